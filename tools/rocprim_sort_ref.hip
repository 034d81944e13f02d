// Calibration only (not part of the product): rocPRIM radix_sort_keys on the same key shape,
// to know what a tuned library sort reaches on this GPU.  hipcc --offload-arch=gfx950 -O3
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char** argv) {
    size_t n = argc > 1 ? atol(argv[1]) : 10000000;
    int bits = argc > 2 ? atoi(argv[2]) : 56;
    std::vector<unsigned long long> h(n);
    std::mt19937_64 g(1);
    for (auto& x : h) x = g() & ((bits == 64) ? ~0ull : ((1ull << bits) - 1));
    unsigned long long *in, *out; void* tmp = nullptr; size_t tb = 0;
    hipMalloc(&in, n * 8); hipMalloc(&out, n * 8);
    hipMemcpy(in, h.data(), n * 8, hipMemcpyHostToDevice);
    rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, bits);
    hipMalloc(&tmp, tb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, bits);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int R = 20;
    for (int it = 0; it < R; ++it) rocprim::radix_sort_keys(tmp, tb, in, out, n, 0, bits);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("rocprim radix_sort_keys n=%zu bits=%d: %.3f ms per sort, %.2f Gkeys/s, tmp=%zu bytes\n", n, bits, ms / R, n / (ms / R) / 1e6, tb);
    return 0;
}
