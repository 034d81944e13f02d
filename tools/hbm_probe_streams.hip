// hbm_probe_streams.hip — what the K1b stream mix reaches by load width.  Four read streams per record as in the packed SoA
// (gx 8 B, umi 4 B, meta 4 B, cell scratch 2 B = 18 B) and one 8-byte store for every fifth record, over N records:
//   A  one record per lane and load (8-, 4-, 4- and 2-byte loads; what filter_pack_stream_kernel issues), R rows in flight
//   B  four consecutive records per lane (two 16-byte loads of gx, one each of umi and meta, one 8-byte load of the scratch)
// Best of 6 launches each.  hipcc --offload-arch=gfx950 -O3 -o build/hbm_probe_streams tools/hbm_probe_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x4 __attribute__((ext_vector_type(4)));

template <bool NT, typename T> __device__ __forceinline__ T ld(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }

// A: a wave takes units of 64 * R records, row r of a unit = records base + r * 64 + lane
template <int R, bool NT>
__global__ __launch_bounds__(256) void k_narrow(const u64* __restrict__ gx, const u32* __restrict__ umi, const u32* __restrict__ meta,
                                                const u16* __restrict__ cell, u64* __restrict__ out, size_t n) {
    const size_t per = (size_t)64 * R;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t u = wave; u * per < n; u += n_waves) {
        u64 g[R]; u32 a[R], m[R]; u16 c[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t i = u * per + (size_t)r * 64 + lane;
            const bool ok = i < n;
            g[r] = ok ? ld<NT>(gx + i) : 0; a[r] = ok ? ld<NT>(umi + i) : 0; m[r] = ok ? ld<NT>(meta + i) : 0; c[r] = ok ? ld<NT>(cell + i) : (u16)0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const size_t i = u * per + (size_t)r * 64 + lane;
            if (i < n && i % 5 == 0) out[i / 5] = g[r] ^ ((u64)a[r] << 20) ^ m[r] ^ ((u64)c[r] << 40);
        }
    }
}

// B: a wave takes units of 256 records, lane l holds records base + 4 l .. 4 l + 3
template <bool NT>
__global__ __launch_bounds__(256) void k_wide(const u64* __restrict__ gx, const u32* __restrict__ umi, const u32* __restrict__ meta,
                                              const u16* __restrict__ cell, u64* __restrict__ out, size_t n) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t u = wave; u * 256 < n; u += n_waves) {
        const size_t i = u * 256 + (size_t)lane * 4;
        if (i + 4 > n) continue;                                   // (n is a multiple of 256 here)
        const u64x2 g0 = ld<NT>((const u64x2*)(gx + i)), g1 = ld<NT>((const u64x2*)(gx + i + 2));
        const u32x4 a = ld<NT>((const u32x4*)(umi + i)), m = ld<NT>((const u32x4*)(meta + i));
        const u16x4 c = ld<NT>((const u16x4*)(cell + i));
        const u64 g[4] = {g0.x, g0.y, g1.x, g1.y};
        const u32 av[4] = {a.x, a.y, a.z, a.w}, mv[4] = {m.x, m.y, m.z, m.w};
        const u16 cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((i + j) % 5 == 0) out[(i + j) / 5] = g[j] ^ ((u64)av[j] << 20) ^ mv[j] ^ ((u64)cv[j] << 40);
    }
}

// K1a's mix: 8 bytes read, 2 bytes written per record; lanes take pairs of records (16-byte loads, 4-byte stores) as the kernel does
template <int U, bool NT>
__global__ __launch_bounds__(1024) void k_k1a(const u64* __restrict__ cb, u16* __restrict__ cell, size_t n) {
    const size_t pairs = n / 2, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t p0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p0 < pairs; p0 += stride * U) {
        u64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t p = p0 + (size_t)u * stride; if (p < pairs) v[u] = ld<NT>((const u64x2*)cb + p); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t p = p0 + (size_t)u * stride;
            if (p < pairs) ((u32*)cell)[p] = (u32)(v[u].x * 0x9E3779B97F4A7C15ull >> 48) | ((u32)(v[u].y * 0x9E3779B97F4A7C15ull >> 48) << 16);
        }
    }
}

template <typename F>
static int timeit(const char* name, size_t bytes, F launch, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        OK(hipEventRecord(e0, 0)); launch(); OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    printf("  %-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
    return 0;
}

int main() {
    const size_t n = 200u * 1000 * 1000 / 256 * 256;
    u64 *gx, *out; u32 *umi, *meta; u16* cell;
    OK(hipMalloc(&gx, n * 8)); OK(hipMalloc(&umi, n * 4)); OK(hipMalloc(&meta, n * 4)); OK(hipMalloc(&cell, n * 2)); OK(hipMalloc(&out, (n / 5 + 1) * 8));
    OK(hipMemset(gx, 1, n * 8)); OK(hipMemset(umi, 2, n * 4)); OK(hipMemset(meta, 3, n * 4)); OK(hipMemset(cell, 4, n * 2));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    const size_t bytes = n * 18 + n / 5 * 8;
    printf("%zu records, %.2f GB per launch (18 B read per record + 8 B written per five)\n", n, bytes / 1e9);
    for (int grid : {1024, 2048, 4096, 8192}) {
        printf(" grid %d x 256\n", grid);
        timeit("A narrow loads, 1 row in flight", bytes, [&] { hipLaunchKernelGGL((k_narrow<1, false>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
        timeit("A narrow loads, 4 rows in flight", bytes, [&] { hipLaunchKernelGGL((k_narrow<4, false>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
        timeit("A narrow loads, 4 rows in flight, nt", bytes, [&] { hipLaunchKernelGGL((k_narrow<4, true>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
        timeit("A narrow loads, 8 rows in flight, nt", bytes, [&] { hipLaunchKernelGGL((k_narrow<8, true>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
        timeit("B wide loads (4 records per lane)", bytes, [&] { hipLaunchKernelGGL((k_wide<false>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
        timeit("B wide loads (4 records per lane), nt", bytes, [&] { hipLaunchKernelGGL((k_wide<true>), dim3(grid), dim3(256), 0, 0, gx, umi, meta, cell, out, n); }, e0, e1);
    }
    const size_t b1 = n * 10;
    printf("K1a mix: %.2f GB per launch (8 B read + 2 B written per record)\n", b1 / 1e9);
    for (int grid : {256, 512, 1024, 2048}) {
        char nm[64];
        snprintf(nm, sizeof nm, "grid %d x 1024, 2 pairs in flight", grid);
        timeit(nm, b1, [&] { hipLaunchKernelGGL((k_k1a<2, false>), dim3(grid), dim3(1024), 0, 0, gx, cell, n); }, e0, e1);
        snprintf(nm, sizeof nm, "grid %d x 1024, 4 pairs in flight, nt", grid);
        timeit(nm, b1, [&] { hipLaunchKernelGGL((k_k1a<4, true>), dim3(grid), dim3(1024), 0, 0, gx, cell, n); }, e0, e1);
    }
    return 0;
}
