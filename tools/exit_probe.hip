// exit_probe.hip — what a process pays between its last useful instruction and its exit, by what it still holds:
//   exit_probe <device MiB> <pinned-registered MiB> <file to mmap or -> <free first: 0|1, 2 = hipDeviceReset before the exit>
// prints the time of the explicit frees (if asked); the caller times the whole process (tools/exit_probe.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t dev_mib = argc > 1 ? atol(argv[1]) : 0, pin_mib = argc > 2 ? atol(argv[2]) : 0;
    const char* path = argc > 3 ? argv[3] : "-";
    const int free_first = argc > 4 ? atoi(argv[4]) : 0;
    const double t0 = now();
    void* d = nullptr; void* h = nullptr; void* m = nullptr; size_t mlen = 0;
    hipStream_t s; if (hipStreamCreate(&s) != hipSuccess) { printf("no device\n"); return 1; }
    if (dev_mib) { if (hipMalloc(&d, dev_mib << 20) != hipSuccess) return 2; (void)hipMemsetAsync(d, 1, dev_mib << 20, s); }
    if (pin_mib) { h = aligned_alloc(4096, pin_mib << 20); memset(h, 1, pin_mib << 20); if (hipHostRegister(h, pin_mib << 20, hipHostRegisterDefault) != hipSuccess) return 3; }
    if (strcmp(path, "-")) {
        int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); mlen = st.st_size;
        m = mmap(nullptr, mlen, PROT_READ, MAP_PRIVATE, fd, 0); close(fd);
        volatile unsigned char acc = 0; for (size_t i = 0; i < mlen; i += 4096) acc += ((unsigned char*)m)[i];
    }
    (void)hipStreamSynchronize(s);
    const double t1 = now();
    if (free_first == 2) {                                 // the runtime torn down explicitly: how much of the exit is that?
        (void)hipDeviceReset();
        printf("setup %.3f s; hipDeviceReset %.3f s\n", t1 - t0, now() - t1);
    } else if (free_first) {
        if (m) munmap(m, mlen);
        const double a = now();
        if (h) { (void)hipHostUnregister(h); free(h); }
        const double b = now();
        if (d) (void)hipFree(d);
        const double c = now();
        printf("setup %.3f s; munmap %.3f unregister+free %.3f hipFree %.3f\n", t1 - t0, a - t1, b - a, c - b);
    } else printf("setup %.3f s\n", t1 - t0);
    fflush(stdout);
    _exit(0);
}
