// h2d_2d_probe.hip — can a host batch (the SoA of fastf_batch_t, pinned) land in the engine's BLOCKED device layout at PCIe rate?
//   A  four plain hipMemcpyAsync (cb, gx, umi, meta): today's push path
//   B  cb plain + three pitched copies (hipMemcpy2DAsync: rows of 2048 / 1024 / 1024 bytes, destination pitch 4608)
// per 8 M-record chunk, best of 5.   hipcc --offload-arch=gfx950 -O3 -o tools/bin/h2d_2d_probe tools/h2d_2d_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <chrono>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t n = 8u << 20, units = n / 256, run = 4608;
    unsigned char *h, *d_soa, *d_blk, *d_cb;
    OK(hipHostMalloc((void**)&h, n * 24, hipHostMallocDefault)); memset(h, 7, n * 24);
    OK(hipMalloc((void**)&d_soa, n * 24)); OK(hipMalloc((void**)&d_blk, units * run)); OK(hipMalloc((void**)&d_cb, n * 8));
    hipStream_t s; OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const unsigned char *cb = h, *gx = h + n * 8, *umi = h + n * 16, *meta = h + n * 20;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double best_a = 1e9, best_b = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        double t0 = now();
        OK(hipMemcpyAsync(d_soa, cb, n * 8, hipMemcpyHostToDevice, s)); OK(hipMemcpyAsync(d_soa + n * 8, gx, n * 8, hipMemcpyHostToDevice, s));
        OK(hipMemcpyAsync(d_soa + n * 16, umi, n * 4, hipMemcpyHostToDevice, s)); OK(hipMemcpyAsync(d_soa + n * 20, meta, n * 4, hipMemcpyHostToDevice, s));
        OK(hipStreamSynchronize(s));
        best_a = std::min(best_a, now() - t0);
        t0 = now();
        OK(hipMemcpyAsync(d_cb, cb, n * 8, hipMemcpyHostToDevice, s));
        OK(hipMemcpy2DAsync(d_blk, run, gx, 2048, 2048, units, hipMemcpyHostToDevice, s));
        OK(hipMemcpy2DAsync(d_blk + 2048, run, umi, 1024, 1024, units, hipMemcpyHostToDevice, s));
        OK(hipMemcpy2DAsync(d_blk + 3072, run, meta, 1024, 1024, units, hipMemcpyHostToDevice, s));
        OK(hipStreamSynchronize(s));
        best_b = std::min(best_b, now() - t0);
    }
    printf("%zu records (%.0f MB):  A four plain copies %.2f ms = %.1f GB/s   B plain + three pitched copies %.2f ms = %.1f GB/s\n",
           n, n * 24 / 1e6, best_a * 1e3, n * 24 / best_a / 1e9, best_b * 1e3, n * 24 / best_b / 1e9);
    return 0;
}
