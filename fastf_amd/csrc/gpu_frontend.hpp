// gpu_frontend.hpp — device side of the BAM front end: BGZF inflate of a whole reader window on the GPU.
// Included by umi_engine.hip; the decoder itself is gpu_inflate.hpp (one wavefront per BGZF block).
//
// Reference step replaced: sam_read1() -> bgzf_read -> inflate, bam2db_ds.c:360 (htslib/zlib on one core).  In the
// host reader (host_io.c) inflate is the longest stage of an end-to-end run on real-shaped BAMs even on 16 threads;
// the window's compressed bytes are a quarter of the inflated ones, so they cross PCIe cheaply, are decoded by as many
// wavefronts as the window has blocks, and come back into the reader's (pinned) window buffer, where the host goes on
// with the record hop and the tag packing.  Every block's CRC-32 is still checked on the host, and a block the device
// declines or gets wrong is inflated again by zlib — the same referee rule as for the host's own decoder.
//
// The window is cut into slices that run H2D -> kernel -> D2H on alternating streams, so the three stages of
// neighbouring slices overlap.
#include "gpu_inflate.hpp"

struct GiBlock { u64 coff; u32 clen, isize; u64 uoff; };          // offsets into the slice's compressed / inflated bytes

__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const GiBlock* __restrict__ blk, u32 n_blk, const uint8_t* __restrict__ comp,
                                                          uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    __shared__ gi::Work w;
    const u32 b = blockIdx.x;
    if (b >= n_blk) return;
    const GiBlock k = blk[b];
    int rc = 0;
    if (k.isize) rc = gi::inflate_block(w, comp + k.coff, k.clen, out + k.uoff, k.isize);
    if (gi_lane0()) status[b] = (uint8_t)rc;
}

struct fastf_gpuinf {
    int device = 0;
    static constexpr int NS = 3;                                   // slices in flight
    hipStream_t s[NS] = {nullptr, nullptr, nullptr};
    GiBlock* h_blk[NS] = {nullptr, nullptr, nullptr}; size_t h_blk_cap[NS] = {0, 0, 0};
    uint8_t* h_status[NS] = {nullptr, nullptr, nullptr};
    DevBuf d_comp[NS], d_out[NS], d_blk[NS], d_status[NS];
    hipEvent_t ev0 = nullptr, ev_done[NS] = {nullptr, nullptr, nullptr};
    size_t first[NS + 1] = {0, 0, 0, 0}; size_t pending_n = 0; int pending_slices = 0;
    u64 n_blocks = 0, n_declined = 0;
};

extern "C" fastf_gpuinf_t* fastf_gpuinf_create(int device) FASTF_TRY {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_err("no HIP device for the BGZF inflate"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    fastf_gpuinf* g = new fastf_gpuinf();
    g->device = device;
    bool ok = hipEventCreate(&g->ev0) == hipSuccess;
    for (int i = 0; i < fastf_gpuinf::NS && ok; ++i)
        ok = hipStreamCreateWithFlags(&g->s[i], hipStreamNonBlocking) == hipSuccess && hipEventCreate(&g->ev_done[i]) == hipSuccess;
    if (!ok) { fastf_gpuinf_destroy(g); set_err("stream/event creation failed"); return nullptr; }
    return g;
} FASTF_CATCH_ZERO

extern "C" void fastf_gpuinf_destroy(fastf_gpuinf_t* g) FASTF_TRY {
    if (!g) return;
    (void)hipSetDevice(g->device);
    for (int i = 0; i < fastf_gpuinf::NS; ++i) {
        if (g->s[i]) { (void)hipStreamSynchronize(g->s[i]); (void)hipStreamDestroy(g->s[i]); }
        if (g->ev_done[i]) (void)hipEventDestroy(g->ev_done[i]);
        if (g->h_blk[i]) (void)hipHostFree(g->h_blk[i]);
        if (g->h_status[i]) (void)hipHostFree(g->h_status[i]);
        g->d_comp[i].release(); g->d_out[i].release(); g->d_blk[i].release(); g->d_status[i].release();
    }
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    delete g;
} FASTF_CATCH_VOID

extern "C" void fastf_gpuinf_stats(const fastf_gpuinf_t* g, uint64_t* n_blocks, uint64_t* n_declined) FASTF_TRY {
    if (n_blocks) *n_blocks = g ? g->n_blocks : 0;
    if (n_declined) *n_declined = g ? g->n_declined : 0;
} FASTF_CATCH_VOID

// blocks [0, n): compressed payload of block i at comp + blk[i].coff (clen bytes), inflated to out + blk[i].uoff (isize
// bytes).  `comp` (readable 64 bytes past the last block) and `out` must be pinned host memory (fastf_pinned_alloc /
// fastf_pinned_register): both directions are plain asynchronous copies.
// submit() only queues the work — up to three slices on alternating streams, so that the H2D of one slice, the kernel of
// another and the D2H of a third overlap — and returns; the caller is free to inflate other blocks on the host
// meanwhile.  wait() returns when everything has landed: status[i] != 0 means the device declined block i (the caller
// inflates it on the host); *device_ms = time from the first copy to the last.
extern "C" int fastf_gpuinf_submit(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                                   unsigned char* out) FASTF_TRY {
    if (!g) return set_err("null inflate handle");
    g->pending_n = 0;
    if (n == 0) return 0;
    HIP_OK(hipSetDevice(g->device));
    constexpr int NS = fastf_gpuinf::NS;
    // A block takes one wave several milliseconds (the decode is a chain of dependent table look-ups), so throughput is
    // the number of blocks in flight over that latency: slices of at least 4096 blocks, at most NS of them.
    const size_t per = std::max<size_t>(std::min<size_t>(4096, n), (n + NS - 1) / NS);
    int n_slices = 0;
    for (size_t a = 0; a < n; a += per) g->first[n_slices++] = a;
    g->first[n_slices] = n;
    HIP_OK(hipEventRecord(g->ev0, g->s[0]));
    for (int sl = 0; sl < n_slices; ++sl) {
        const int q = sl;
        const size_t a = g->first[sl], b = g->first[sl + 1], nb = b - a;
        const u64 c0 = blk[a].coff & ~(u64)63, c1 = blk[b - 1].coff + blk[b - 1].clen;
        const u64 u0 = blk[a].uoff; u64 u1 = u0;
        for (size_t i = a; i < b; ++i) u1 = std::max<u64>(u1, blk[i].uoff + blk[i].isize);
        const size_t cbytes = (size_t)(c1 - c0) + 64, ubytes = (size_t)(u1 - u0);
        if (nb > g->h_blk_cap[q]) {
            if (g->h_blk[q]) (void)hipHostFree(g->h_blk[q]);
            if (g->h_status[q]) (void)hipHostFree(g->h_status[q]);
            g->h_blk[q] = nullptr; g->h_status[q] = nullptr; g->h_blk_cap[q] = 0;
            const size_t cap = nb + nb / 4 + 64;
            HIP_OK(hipHostMalloc((void**)&g->h_blk[q], cap * sizeof(GiBlock), hipHostMallocDefault));
            HIP_OK(hipHostMalloc((void**)&g->h_status[q], cap, hipHostMallocDefault));
            g->h_blk_cap[q] = cap;
        }
        for (size_t i = a; i < b; ++i) g->h_blk[q][i - a] = GiBlock{blk[i].coff - c0, blk[i].clen, blk[i].isize, blk[i].uoff - u0};
        if (g->d_comp[q].ensure(cbytes) || g->d_out[q].ensure(std::max<size_t>(ubytes, 64)) || g->d_blk[q].ensure(nb * sizeof(GiBlock)) ||
            g->d_status[q].ensure(nb))
            return 1;
        hipStream_t s = g->s[q];
        if (q) HIP_OK(hipStreamWaitEvent(s, g->ev0, 0));
        HIP_OK(hipMemcpyAsync(g->d_comp[q].p, comp + c0, cbytes, hipMemcpyHostToDevice, s));
        HIP_OK(hipMemcpyAsync(g->d_blk[q].p, g->h_blk[q], nb * sizeof(GiBlock), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((u32)nb), dim3(64), 0, s, (const GiBlock*)g->d_blk[q].p, (u32)nb,
                           (const uint8_t*)g->d_comp[q].p, (uint8_t*)g->d_out[q].p, (uint8_t*)g->d_status[q].p);
        HIP_OK(hipGetLastError());
        if (ubytes) HIP_OK(hipMemcpyAsync(out + u0, g->d_out[q].p, ubytes, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(g->h_status[q], g->d_status[q].p, nb, hipMemcpyDeviceToHost, s));
        HIP_OK(hipEventRecord(g->ev_done[q], s));
    }
    g->pending_n = n; g->pending_slices = n_slices;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_gpuinf_wait(fastf_gpuinf_t* g, uint8_t* status, double* device_ms) FASTF_TRY {
    if (!g) return set_err("null inflate handle");
    if (device_ms) *device_ms = 0;
    if (!g->pending_n) return 0;
    HIP_OK(hipSetDevice(g->device));
    float worst = 0;
    for (int sl = 0; sl < g->pending_slices; ++sl) {
        HIP_OK(hipEventSynchronize(g->ev_done[sl]));
        memcpy(status + g->first[sl], g->h_status[sl], g->first[sl + 1] - g->first[sl]);
        float ms = 0;
        if (hipEventElapsedTime(&ms, g->ev0, g->ev_done[sl]) == hipSuccess && ms > worst) worst = ms;
    }
    if (device_ms) *device_ms = worst;
    g->n_blocks += g->pending_n;
    for (size_t i = 0; i < g->pending_n; ++i) g->n_declined += status[i] != 0;
    g->pending_n = 0;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_gpuinf_run(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                                unsigned char* out, uint8_t* status) FASTF_TRY {
    if (fastf_gpuinf_submit(g, comp, blk, n, out)) return 1;
    return fastf_gpuinf_wait(g, status, nullptr);
} FASTF_CATCH_INT
