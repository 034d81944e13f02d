/* pinwatch — LD_PRELOAD ledger of the host memory the HIP runtime has been told about, for the WHOLE process (this library,
 * torch, RCCL): diagnostic tooling for the GPU memory fault of rounds 5/6, not product code.
 *
 *   LD_PRELOAD=build/libpinwatch.so PINWATCH_LOG=gpurun_out/r6/pinwatch.log python3 -m pytest ...
 *
 * What it keeps: every live hipHostRegister range and every live hipHostMalloc / hipHostAlloc block.
 * What it reports (one line each, with a backtrace, flushed at once):
 *   FREE-WHILE-REGISTERED   free() / realloc() of a heap chunk, or munmap / madvise(MADV_DONTNEED|MADV_FREE) / mremap of a range,
 *                           that overlaps a live hipHostRegister range: the runtime keeps a GPU mapping of pages that are going
 *                           away — the next owner of those addresses inherits a stale registration
 *   STALE-REGISTRATION      a copy whose host side the RUNTIME reports as registered (hipPointerGetAttributes) while the ledger
 *                           holds nothing there: a registration that outlived its memory, or one made behind the ledger's back
 *   REGISTER-OVERLAP        hipHostRegister of a range that overlaps (page-wise) a live one of another base
 *   UNREGISTER-UNKNOWN      hipHostUnregister of a pointer the ledger does not hold
 * What it counts (summary at exit and on SIGABRT, which is how a GPU fault ends the process): registrations, pinned
 * allocations, copies by kind of host memory (registered / runtime-allocated / PAGEABLE, the big ones by calling module),
 * and the last 48 events before the end.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <malloc.h>
#include <signal.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct { int type; int device; void *devicePointer; void *hostPointer; int isManaged; unsigned allocationFlags; } pw_attr_t;

extern void __libc_free(void *);
extern void *__libc_realloc(void *, size_t);

#define MAXR 512
typedef struct { uintptr_t a, b; int kind; } range_t;           /* kind 1: hipHostRegister, 2: hipHostMalloc */
static range_t g_r[MAXR];
static volatile int g_n, g_nreg;                                 /* entries in use (dense prefix), live registrations */
static volatile int g_lock;
static int g_fd = -1;
static __thread int g_inside;                                    /* re-entrancy guard (backtrace, dlsym and the runtime itself call free) */
static unsigned long c_reg, c_unreg, c_hmalloc, c_hfree, c_copy_reg, c_copy_rt, c_copy_page_small, c_copy_page_big, c_viol;
static unsigned long long b_copy_page_big;
#define NMOD 16
static struct { char name[96]; unsigned long n; unsigned long long bytes; } g_mod[NMOD];
#define NEV 48
static char g_ev[NEV][160];
static volatile unsigned g_evi;

static void lock(void) { while (__atomic_exchange_n(&g_lock, 1, __ATOMIC_ACQUIRE)) ; }
static void unlock(void) { __atomic_store_n(&g_lock, 0, __ATOMIC_RELEASE); }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void out(const char *fmt, ...)
{
    char buf[512]; va_list ap; va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (n > (int)sizeof buf - 1) n = sizeof buf - 1;
    if (g_fd >= 0) (void)!write(g_fd, buf, (size_t)n);
}
static void event(const char *fmt, ...)
{
    unsigned i = __atomic_fetch_add(&g_evi, 1, __ATOMIC_RELAXED) % NEV;
    va_list ap; va_start(ap, fmt);
    int k = snprintf(g_ev[i], sizeof g_ev[i], "%.3f t%ld ", now_s(), (long)syscall(SYS_gettid));
    vsnprintf(g_ev[i] + k, sizeof g_ev[i] - (size_t)k, fmt, ap); va_end(ap);
}
static void violation(const char *what, uintptr_t a, uintptr_t b, const range_t *r)
{
    c_viol++;
    out("%s [%#lx, %#lx) against %s [%#lx, %#lx)  (t=%.3f, thread %ld)\n", what, a, b,
        r ? (r->kind == 1 ? "registered" : "pinned allocation") : "-", r ? r->a : 0, r ? r->b : 0, now_s(), (long)syscall(SYS_gettid));
    void *bt[24]; int n = backtrace(bt, 24);
    if (g_fd >= 0) backtrace_symbols_fd(bt, n, g_fd);
}

static void summary(const char *why)
{
    out("---- pinwatch summary (%s): %lu registrations (%lu unregistered, %d live), %lu pinned allocations (%lu freed); copies: %lu registered host memory, "
        "%lu runtime-allocated pinned, %lu pageable below 1 MiB, %lu pageable of 1 MiB or more (%.1f MB); %lu violations\n", why,
        c_reg, c_unreg, g_nreg, c_hmalloc, c_hfree, c_copy_reg, c_copy_rt, c_copy_page_small, c_copy_page_big, b_copy_page_big / 1e6, c_viol);
    for (int i = 0; i < NMOD && g_mod[i].n; i++) out("     big pageable copies from %s: %lu (%.1f MB)\n", g_mod[i].name, g_mod[i].n, g_mod[i].bytes / 1e6);
    out("     last events:\n");
    for (unsigned k = 0; k < NEV; k++) { const char *e = g_ev[(g_evi + k) % NEV]; if (e[0]) out("       %s\n", e); }
}
static void on_abort(int sig) { summary("SIGABRT"); signal(sig, SIG_DFL); raise(sig); }
static void at_exit(void) { summary("exit"); }

static void init(void)
{
    static int done;
    if (done) return;
    done = 1;
    const char *p = getenv("PINWATCH_LOG");
    char path[512];
    if (p) { snprintf(path, sizeof path, "%s.%d", p, (int)getpid()); g_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644); }
    else g_fd = 2;
    signal(SIGABRT, on_abort);
    atexit(at_exit);
    out("pinwatch: process %d\n", (int)getpid());
}

static range_t *find_overlap(uintptr_t a, uintptr_t b, int kind_mask)      /* caller holds the lock */
{
    for (int i = 0; i < g_n; i++) if (g_r[i].kind && (g_r[i].kind & kind_mask) && a < g_r[i].b && g_r[i].a < b) return &g_r[i];
    return NULL;
}
static void add(uintptr_t a, uintptr_t b, int kind)
{
    lock();
    int i = 0;
    for (; i < g_n; i++) if (!g_r[i].kind) break;
    if (i < MAXR) { g_r[i].a = a; g_r[i].b = b; g_r[i].kind = kind; if (i == g_n) g_n = i + 1; if (kind == 1) g_nreg++; }
    unlock();
}
static int drop(uintptr_t a, int kind)
{
    int found = 0;
    lock();
    for (int i = 0; i < g_n; i++) if (g_r[i].kind == kind && g_r[i].a == a) { g_r[i].kind = 0; found = 1; if (kind == 1) g_nreg--; break; }
    unlock();
    return found;
}

/* ---- memory going away ---- */
static void check_gone(const char *what, uintptr_t a, uintptr_t b)
{
    if (!g_nreg || g_inside) return;
    g_inside = 1;
    /* page-wise: the runtime pins whole pages */
    const uintptr_t pa = a & ~(uintptr_t)4095, pb = (b + 4095) & ~(uintptr_t)4095;
    lock();
    range_t *r = find_overlap(pa, pb, 1), copy;
    /* a heap chunk that merely shares its first or last page with a registered neighbour is not going away page-wise */
    if (r && (what[0] == 'f' || what[0] == 'r') && !(a < r->b && r->a < b)) r = NULL;
    if (r) copy = *r;
    unlock();
    if (r) { init(); char w[64]; snprintf(w, sizeof w, "FREE-WHILE-REGISTERED (%s)", what); violation(w, a, b, &copy); }
    g_inside = 0;
}
void free(void *p)
{
    if (p && g_nreg && !g_inside) check_gone("free", (uintptr_t)p, (uintptr_t)p + malloc_usable_size(p));
    __libc_free(p);
}
void *realloc(void *p, size_t n)
{
    if (p && g_nreg && !g_inside) check_gone("realloc", (uintptr_t)p, (uintptr_t)p + malloc_usable_size(p));
    return __libc_realloc(p, n);
}
int munmap(void *p, size_t n) { check_gone("munmap", (uintptr_t)p, (uintptr_t)p + n); return (int)syscall(SYS_munmap, p, n); }
int madvise(void *p, size_t n, int adv)
{
    if (adv == MADV_DONTNEED || adv == MADV_FREE || adv == MADV_REMOVE) check_gone(adv == MADV_DONTNEED ? "madvise(DONTNEED)" : "madvise(FREE/REMOVE)", (uintptr_t)p, (uintptr_t)p + n);
    return (int)syscall(SYS_madvise, p, n, adv);
}

/* ---- the runtime's entry points ---- */
static void *next(const char *name) { void *f = dlsym(RTLD_NEXT, name); if (!f) { out("pinwatch: no %s behind me\n", name); } return f; }
#define NEXT(var, name) if (!var) { g_inside++; var = (__typeof__(var))next(name); g_inside--; }

hipError_t hipHostRegister(void *p, size_t n, unsigned flags)
{
    static hipError_t (*real)(void *, size_t, unsigned); NEXT(real, "hipHostRegister");
    init();
    hipError_t e = real(p, n, flags);
    if (e == 0) {
        c_reg++;
        const uintptr_t a = (uintptr_t)p, b = a + n;
        lock(); range_t *r = find_overlap(a & ~(uintptr_t)4095, (b + 4095) & ~(uintptr_t)4095, 1), copy; if (r) copy = *r; unlock();
        if (r) violation("REGISTER-OVERLAP", a, b, &copy);
        add(a, b, 1);
        event("hipHostRegister [%#lx, %#lx) %zu bytes", a, b, n);
    } else event("hipHostRegister(%p, %zu) -> error %d", p, n, e);
    return e;
}
hipError_t hipHostUnregister(void *p)
{
    static hipError_t (*real)(void *); NEXT(real, "hipHostUnregister");
    init();
    const int known = drop((uintptr_t)p, 1);
    hipError_t e = real(p);
    c_unreg += known;
    if (!known && e == 0) violation("UNREGISTER-UNKNOWN", (uintptr_t)p, (uintptr_t)p, NULL);
    event("hipHostUnregister %p -> %d%s", p, e, known ? "" : " (not in the ledger)");
    return e;
}
static hipError_t host_alloc(const char *name, hipError_t (*real)(void **, size_t, unsigned), void **pp, size_t n, unsigned flags)
{
    init();
    hipError_t e = real(pp, n, flags);
    if (e == 0 && *pp) { c_hmalloc++; add((uintptr_t)*pp, (uintptr_t)*pp + n, 2); event("%s %p %zu bytes", name, *pp, n); }
    return e;
}
hipError_t hipHostMalloc(void **pp, size_t n, unsigned flags) { static hipError_t (*real)(void **, size_t, unsigned); NEXT(real, "hipHostMalloc"); return host_alloc("hipHostMalloc", real, pp, n, flags); }
hipError_t hipHostAlloc(void **pp, size_t n, unsigned flags) { static hipError_t (*real)(void **, size_t, unsigned); NEXT(real, "hipHostAlloc"); return host_alloc("hipHostAlloc", real, pp, n, flags); }
hipError_t hipHostFree(void *p)
{
    static hipError_t (*real)(void *); NEXT(real, "hipHostFree");
    init();
    if (p) { c_hfree += (unsigned long)drop((uintptr_t)p, 2); event("hipHostFree %p", p); }
    return real(p);
}
hipError_t hipFreeHost(void *p) { return hipHostFree(p); }

/* one side of a copy that may be host memory */
static void host_side(const char *api, const void *hp, size_t n, void *ret_addr)
{
    if (!hp || g_inside) return;
    static hipError_t (*attrs)(pw_attr_t *, const void *); static hipError_t (*lasterr)(void);
    if (!attrs) { g_inside++; attrs = (hipError_t (*)(pw_attr_t *, const void *))next("hipPointerGetAttributes"); lasterr = (hipError_t (*)(void))next("hipGetLastError"); g_inside--; }
    if (!attrs) return;
    g_inside++;
    pw_attr_t a; memset(&a, 0, sizeof a);
    const hipError_t e = attrs(&a, hp);
    if (e != 0 && lasterr) (void)lasterr();
    const int rt_knows = e == 0 && a.type != 0;                 /* 0: hipMemoryTypeUnregistered (runtime 7.x); an error: unknown to it as well */
    const int rt_host = rt_knows && a.type == 1;
    if (rt_knows && !rt_host) { g_inside--; return; }            /* device / managed memory */
    const uintptr_t x = (uintptr_t)hp;
    lock(); range_t *r = find_overlap(x, x + (n ? n : 1), 3); int kind = r ? r->kind : 0; unlock();
    if (rt_host && !kind) violation("STALE-REGISTRATION (the runtime reports host memory the ledger does not hold)", x, x + n, NULL);
    if (kind == 1) c_copy_reg++;
    else if (kind == 2 || rt_host) c_copy_rt++;
    else if (n < ((size_t)1 << 20)) c_copy_page_small++;
    else {
        c_copy_page_big++; b_copy_page_big += n;
        Dl_info di; const char *m = "?";
        if (dladdr(ret_addr, &di) && di.dli_fname) { m = strrchr(di.dli_fname, '/'); m = m ? m + 1 : di.dli_fname; }
        for (int i = 0; i < NMOD; i++) {
            if (!g_mod[i].n) { snprintf(g_mod[i].name, sizeof g_mod[i].name, "%s", m); }
            if (!strcmp(g_mod[i].name, m)) { g_mod[i].n++; g_mod[i].bytes += n; break; }
        }
        event("%s: PAGEABLE host memory %p, %zu bytes, called from %s", api, hp, n, m);
    }
    g_inside--;
}
/* hipMemcpyKind: 0 H2H, 1 H2D, 2 D2H, 3 D2D, 4 default */
static void copy_sides(const char *api, void *dst, const void *src, size_t n, int kind, void *ra)
{
    if (kind == 1 || kind == 0 || kind == 4) host_side(api, src, n, ra);
    if (kind == 2 || kind == 0 || kind == 4) host_side(api, dst, n, ra);
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, int kind)
{
    static hipError_t (*real)(void *, const void *, size_t, int); NEXT(real, "hipMemcpy");
    init(); copy_sides("hipMemcpy", dst, src, n, kind, __builtin_return_address(0));
    return real(dst, src, n, kind);
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, int kind, hipStream_t s)
{
    static hipError_t (*real)(void *, const void *, size_t, int, hipStream_t); NEXT(real, "hipMemcpyAsync");
    init(); copy_sides("hipMemcpyAsync", dst, src, n, kind, __builtin_return_address(0));
    return real(dst, src, n, kind, s);
}
hipError_t hipMemcpyWithStream(void *dst, const void *src, size_t n, int kind, hipStream_t s)
{
    static hipError_t (*real)(void *, const void *, size_t, int, hipStream_t); NEXT(real, "hipMemcpyWithStream");
    init(); copy_sides("hipMemcpyWithStream", dst, src, n, kind, __builtin_return_address(0));
    return real(dst, src, n, kind, s);
}
hipError_t hipMemcpyDtoH(void *dst, void *src, size_t n)
{
    static hipError_t (*real)(void *, void *, size_t); NEXT(real, "hipMemcpyDtoH");
    init(); host_side("hipMemcpyDtoH", dst, n, __builtin_return_address(0));
    return real(dst, src, n);
}
hipError_t hipMemcpyHtoD(void *dst, void *src, size_t n)
{
    static hipError_t (*real)(void *, void *, size_t); NEXT(real, "hipMemcpyHtoD");
    init(); host_side("hipMemcpyHtoD", src, n, __builtin_return_address(0));
    return real(dst, src, n);
}
hipError_t hipMemcpyDtoHAsync(void *dst, void *src, size_t n, hipStream_t s)
{
    static hipError_t (*real)(void *, void *, size_t, hipStream_t); NEXT(real, "hipMemcpyDtoHAsync");
    init(); host_side("hipMemcpyDtoHAsync", dst, n, __builtin_return_address(0));
    return real(dst, src, n, s);
}
hipError_t hipMemcpyHtoDAsync(void *dst, void *src, size_t n, hipStream_t s)
{
    static hipError_t (*real)(void *, void *, size_t, hipStream_t); NEXT(real, "hipMemcpyHtoDAsync");
    init(); host_side("hipMemcpyHtoDAsync", src, n, __builtin_return_address(0));
    return real(dst, src, n, s);
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, int kind, hipStream_t s)
{
    static hipError_t (*real)(void *, size_t, const void *, size_t, size_t, size_t, int, hipStream_t); NEXT(real, "hipMemcpy2DAsync");
    init();
    if (kind == 1 || kind == 4) host_side("hipMemcpy2DAsync", src, height ? (height - 1) * spitch + width : 0, __builtin_return_address(0));
    if (kind == 2 || kind == 4) host_side("hipMemcpy2DAsync", dst, height ? (height - 1) * dpitch + width : 0, __builtin_return_address(0));
    return real(dst, dpitch, src, spitch, width, height, kind, s);
}
