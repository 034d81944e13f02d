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

// C: BLOCKED layout — per 256-record unit one contiguous run  gx[256] (2048 B) | umi[256] (1024 B) | meta[256] (1024 B) | cell[256] (512 B)
//    = 4608 bytes: the same 18 bytes per record, read as ONE stream.  A wave takes a unit; row r of it = records 64 r + lane
//    (WIDE: lane l holds records 4 l .. 4 l + 3, 16-byte loads).  K1a's half of that layout: the 8-byte cb slice of a unit is read
//    and the 2-byte scratch is written into the unit's run (k_k1a_blocked): a unit is then cb | gx | umi | meta with the scratch
//    overwriting the front of the dead cb slice — probed as 6656-byte runs.
constexpr size_t BLK = 4608;
template <bool WIDE, bool NT, bool WITH_CB = false>
__global__ __launch_bounds__(256) void k_blocked(const unsigned char* __restrict__ blk, u64* __restrict__ out, size_t n, size_t run) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t u = wave; u * 256 < n; u += n_waves) {
        const unsigned char* b = blk + u * run;
        // WITH_CB: 6656-byte runs  scratch (over the front of the dead cb slice) .. | gx | umi | meta
        const u64* gx = (const u64*)(b + (WITH_CB ? 2048 : 0)); const u32* umi = (const u32*)(b + (WITH_CB ? 4096 : 2048));
        const u32* meta = (const u32*)(b + (WITH_CB ? 5120 : 3072)); const u16* cell = (const u16*)(b + (WITH_CB ? 0 : 4096));
        u64 g[4]; u32 a[4], m[4]; u16 c[4];
        if constexpr (WIDE) {
            const u64x2 g0 = ld<NT>((const u64x2*)(gx + 4 * lane)), g1 = ld<NT>((const u64x2*)(gx + 4 * lane + 2));
            const u32x4 av = ld<NT>((const u32x4*)(umi + 4 * lane)), mv = ld<NT>((const u32x4*)(meta + 4 * lane));
            const u16x4 cv = ld<NT>((const u16x4*)(cell + 4 * lane));
            g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y; a[0] = av.x; a[1] = av.y; a[2] = av.z; a[3] = av.w;
            m[0] = mv.x; m[1] = mv.y; m[2] = mv.z; m[3] = mv.w; c[0] = cv.x; c[1] = cv.y; c[2] = cv.z; c[3] = cv.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { g[r] = ld<NT>(gx + r * 64 + lane); a[r] = ld<NT>(umi + r * 64 + lane); m[r] = ld<NT>(meta + r * 64 + lane); c[r] = ld<NT>(cell + r * 64 + lane); }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t i = u * 256 + (WIDE ? (size_t)lane * 4 + r : (size_t)r * 64 + lane);
            if (i % 5 == 0) out[i / 5] = g[r] ^ ((u64)a[r] << 20) ^ m[r] ^ ((u64)c[r] << 40);
        }
    }
}
// K1a on the blocked layout: reads the unit's cb slice (2048 B at the front of a 6656-byte run), writes the 512-byte scratch over its front
template <bool NT>
__global__ __launch_bounds__(1024) void k_k1a_blocked(unsigned char* __restrict__ blk, size_t n, size_t run) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t u = wave; u * 256 < n; u += n_waves) {
        unsigned char* b = blk + u * run;
        const u64x2 v0 = ld<NT>((const u64x2*)b + lane), v1 = ld<NT>((const u64x2*)b + 64 + lane);
        const u32 c0 = (u32)(v0.x * 0x9E3779B97F4A7C15ull >> 48) | ((u32)(v0.y * 0x9E3779B97F4A7C15ull >> 48) << 16);
        const u32 c1 = (u32)(v1.x * 0x9E3779B97F4A7C15ull >> 48) | ((u32)(v1.y * 0x9E3779B97F4A7C15ull >> 48) << 16);
        // the scratch goes where the kernel has just read (same lines): u32 pair index lane, 64 + lane
        ((u32*)b)[lane] = c0; ((u32*)b)[64 + lane] = c1;
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
    {   // the same bytes from the blocked layout (one stream of 4608-byte runs; and 6656-byte runs whose first 2048 bytes — the cb slice K1a has
        // consumed, its front now holding the scratch — are skipped except for the 512 scratch bytes)
        unsigned char* blk; OK(hipMalloc(&blk, n / 256 * 6656)); OK(hipMemset(blk, 5, n / 256 * 6656));
        printf("BLOCKED layout, same %.2f GB per launch\n", bytes / 1e9);
        for (int grid : {1024, 2048, 4096, 8192}) {
            printf(" grid %d x 256\n", grid);
            timeit("C blocked 4608-B runs, narrow loads", bytes, [&] { hipLaunchKernelGGL((k_blocked<false, false>), dim3(grid), dim3(256), 0, 0, blk, out, n, BLK); }, e0, e1);
            timeit("C blocked 4608-B runs, narrow loads, nt", bytes, [&] { hipLaunchKernelGGL((k_blocked<false, true>), dim3(grid), dim3(256), 0, 0, blk, out, n, BLK); }, e0, e1);
            timeit("C blocked 4608-B runs, wide loads", bytes, [&] { hipLaunchKernelGGL((k_blocked<true, false>), dim3(grid), dim3(256), 0, 0, blk, out, n, BLK); }, e0, e1);
            timeit("C blocked 4608-B runs, wide loads, nt", bytes, [&] { hipLaunchKernelGGL((k_blocked<true, true>), dim3(grid), dim3(256), 0, 0, blk, out, n, BLK); }, e0, e1);
            timeit("C' 6656-B runs (1536 dead bytes), narrow, nt", bytes, [&] { hipLaunchKernelGGL((k_blocked<false, true, true>), dim3(grid), dim3(256), 0, 0, blk, out, n, (size_t)6656); }, e0, e1);
            timeit("C' 6656-B runs (1536 dead bytes), wide, nt", bytes, [&] { hipLaunchKernelGGL((k_blocked<true, true, true>), dim3(grid), dim3(256), 0, 0, blk, out, n, (size_t)6656); }, e0, e1);
        }
        printf("K1a on the blocked layout (cb slice read, scratch written over its front; 6656-B runs): %.2f GB per launch\n", n * 10 / 1e9);
        for (int grid : {256, 512, 1024}) {
            char nm[64]; snprintf(nm, sizeof nm, "grid %d x 1024", grid);
            timeit(nm, n * 10, [&] { hipLaunchKernelGGL((k_k1a_blocked<false>), dim3(grid), dim3(1024), 0, 0, blk, n, (size_t)6656); }, e0, e1);
            snprintf(nm, sizeof nm, "grid %d x 1024, nt", grid);
            timeit(nm, n * 10, [&] { hipLaunchKernelGGL((k_k1a_blocked<true>), dim3(grid), dim3(1024), 0, 0, blk, n, (size_t)6656); }, e0, e1);
        }
        OK(hipFree(blk));
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
