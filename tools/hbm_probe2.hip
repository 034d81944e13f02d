// hbm_probe2.hip — streaming-read variants: loads in flight per thread, non-temporal hint, workgroup size, contiguous
// per-workgroup chunks vs grid-stride.  Reports the best of 6 launches per variant on 4 GiB and 300 MiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, int TH, bool CHUNK>
__global__ __launch_bounds__(TH) void k_read(const u32x4* __restrict__ in, size_t n, unsigned long long* out) {
    u32x4 acc = {0, 0, 0, 0};
    size_t i0, i1, stride;
    if (CHUNK) {          // workgroup b reads [b * per, (b + 1) * per) front to back
        const size_t per = ((n + gridDim.x - 1) / gridDim.x + (size_t)TH * U - 1) / ((size_t)TH * U) * ((size_t)TH * U);
        i0 = (size_t)blockIdx.x * per + threadIdx.x; i1 = std::min(n, (size_t)(blockIdx.x + 1) * per); stride = (size_t)TH * U;
    } else { i0 = (size_t)blockIdx.x * TH * U + threadIdx.x; i1 = n; stride = (size_t)gridDim.x * TH * U; }
    for (size_t i = i0; i < i1; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = i + (size_t)u * TH;
            if (k < i1) v[u] = NT ? __builtin_nontemporal_load(in + k) : in[k]; else v[u] = acc;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345677u) *out = 1;
}

template <int U, bool NT, int TH, bool CHUNK>
static int run(const char* name, const void* a, size_t bytes, int grid, unsigned long long* flag, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        OK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_read<U, NT, TH, CHUNK>), dim3(grid), dim3(TH), 0, 0, (const u32x4*)a, bytes / 16, flag);
        OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    printf("  %-34s grid %5d: %.3f ms %.2f TB/s\n", name, grid, best, bytes / best / 1e9);
    return 0;
}
int main() {
    const size_t max_bytes = (size_t)4 << 30;
    void* a; unsigned long long* flag;
    OK(hipMalloc(&a, max_bytes)); OK(hipMalloc(&flag, 8)); OK(hipMemset(a, 1, max_bytes));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    for (size_t bytes : {(size_t)4 << 30, (size_t)300 << 20}) {
        printf("read of %zu MiB\n", bytes >> 20);
        for (int g : {2048, 8192}) {
            run<4, false, 256, false>("U4 256thr stride", a, bytes, g, flag, e0, e1);
            run<8, false, 256, false>("U8 256thr stride", a, bytes, g, flag, e0, e1);
            run<16, false, 256, false>("U16 256thr stride", a, bytes, g, flag, e0, e1);
            run<4, true, 256, false>("U4 256thr stride nt", a, bytes, g, flag, e0, e1);
            run<8, true, 256, false>("U8 256thr stride nt", a, bytes, g, flag, e0, e1);
            run<4, false, 256, true>("U4 256thr chunk", a, bytes, g, flag, e0, e1);
            run<8, true, 256, true>("U8 256thr chunk nt", a, bytes, g, flag, e0, e1);
        }
        for (int g : {512, 1024, 2048}) {
            run<4, false, 1024, false>("U4 1024thr stride", a, bytes, g, flag, e0, e1);
            run<8, true, 1024, false>("U8 1024thr stride nt", a, bytes, g, flag, e0, e1);
            run<4, false, 1024, true>("U4 1024thr chunk", a, bytes, g, flag, e0, e1);
        }
    }
    return 0;
}
