/* thp_probe.c — cost of touching and freeing a big anonymous buffer with 4 KiB pages and with transparent huge pages */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
    const size_t len = (size_t)640 << 20;
    for (int huge = 0; huge <= 1; huge++) for (int rep = 0; rep < 2; rep++) {
        void *p = NULL;
        if (posix_memalign(&p, (size_t)2 << 20, len)) return 1;
        if (huge) { if (madvise(p, len, MADV_HUGEPAGE)) perror("madvise(MADV_HUGEPAGE)"); }
        else (void)madvise(p, len, MADV_NOHUGEPAGE);
        double t0 = now();
        for (size_t i = 0; i < len; i += 4096) ((volatile char *)p)[i] = 1;
        double t1 = now();
        free(p);
        double t2 = now();
        printf("%s: touch %.3f s, free %.3f s\n", huge ? "MADV_HUGEPAGE  " : "MADV_NOHUGEPAGE", t1 - t0, t2 - t1);
    }
    return 0;
}
