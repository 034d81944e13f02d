// hbm_probe.hip — what a plain streaming kernel reaches on this GPU: read-only, write-only and copy, at the sizes the
// hot-path kernels move.  Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_probe.hip -o gpurun_out/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ in, size_t n, unsigned long long* out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        uint4 a = in[i], b = i + 256 < n ? in[i + 256] : acc, c = i + 512 < n ? in[i + 512] : acc, d = i + 768 < n ? in[i + 768] : acc;
        acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345677u) *out = 1;
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        o[i] = v; if (i + 256 < n) o[i + 256] = v; if (i + 512 < n) o[i + 512] = v; if (i + 768 < n) o[i + 768] = v;
    }
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ in, uint4* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        uint4 a = in[i], b, c, d;
        const bool hb = i + 256 < n, hc = i + 512 < n, hd = i + 768 < n;
        if (hb) b = in[i + 256]; if (hc) c = in[i + 512]; if (hd) d = in[i + 768];
        o[i] = a; if (hb) o[i + 256] = b; if (hc) o[i + 512] = c; if (hd) o[i + 768] = d;
    }
}
int main() {
    const size_t max_bytes = (size_t)4 << 30;
    void *a, *b; unsigned long long* flag;
    OK(hipMalloc(&a, max_bytes)); OK(hipMalloc(&b, max_bytes)); OK(hipMalloc(&flag, 8));
    OK(hipMemset(a, 1, max_bytes)); OK(hipMemset(b, 2, max_bytes));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)300 << 20, (size_t)1 << 30, (size_t)4 << 30};
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
    for (size_t bytes : sizes) for (int g : grids) {
        const size_t n = bytes / 16;
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 6; ++rep) for (int k = 0; k < 3; ++k) {
            OK(hipEventRecord(e0, 0));
            if (k == 0) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, (const uint4*)a, n, flag);
            if (k == 1) hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, (uint4*)b, n);
            if (k == 2) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n / 2);
            OK(hipEventRecord(e1, 0)); OK(hipEventSynchronize(e1));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1)); best[k] = std::min(best[k], ms);
        }
        printf("bytes %5zu MB grid %5d: read %.3f ms %.2f TB/s | write %.3f ms %.2f TB/s | copy (r+w same total) %.3f ms %.2f TB/s\n", bytes >> 20, g,
               best[0], bytes / best[0] / 1e9, best[1], bytes / best[1] / 1e9, best[2], bytes / best[2] / 1e9);
    }
    return 0;
}
