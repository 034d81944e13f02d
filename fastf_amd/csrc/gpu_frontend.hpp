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
#include "gpu_inflate2.hpp"
#include "gpu_records.hpp"

struct GiBlock { u64 coff; u32 clen, isize; u64 uoff; u64 toff; };   // offsets into the slice's compressed / inflated bytes / token lists

#ifndef FASTF_GI_MINBLOCKS
#define FASTF_GI_MINBLOCKS 5      // waves per SIMD: 96 VGPRs instead of 109, twenty blocks per CU as the LDS allows (25.6 -> 26.0 GB/s)
#endif
__global__ __launch_bounds__(64, FASTF_GI_MINBLOCKS) void bgzf_inflate_kernel(const GiBlock* __restrict__ blk, u32 n_blk, const uint8_t* __restrict__ comp,
                                                          uint8_t* __restrict__ out, uint8_t* __restrict__ status) {
    __shared__ gi::Work w;
    const u32 b = blockIdx.x;
    if (b >= n_blk) return;
    const GiBlock k = blk[b];
    int rc = 0;
    if (k.isize) rc = gi::inflate_block(w, comp + k.coff, k.clen, out + k.uoff, k.isize);
    if (gi_lane0()) status[b] = (uint8_t)rc;
}

// an aligned dword past the CU's L1 (agent scope: this wave's own stores of a round ago are in L2, not necessarily in its L1)
__device__ __forceinline__ u32 gi_l2_load32(const uint8_t* p) {
    return __hip_atomic_load(reinterpret_cast<const u32*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the two-kernel inflate (round 5; gpu_inflate2.hpp has the why) ----
// Phase 1: ONE LANE per BGZF block, GI2_LPW blocks per wave (their working sets side by side in LDS: 580 bytes each).  A lane
// decodes its block's tokens, stores the literals where they belong and lists the matches as 4-byte tokens.
#ifndef FASTF_GI2_LPW
#define FASTF_GI2_LPW 64
#endif
constexpr int GI2_LPW = FASTF_GI2_LPW;
static_assert(GI2_LPW >= 1 && GI2_LPW <= 64 && sizeof(gi2::Work) * GI2_LPW <= 40 * 1024, "blocks per wave: four waves' working sets share a CU's LDS");
#ifndef FASTF_GI2_LDS_MULT
#define FASTF_GI2_LDS_MULT 1                 // (probe builds: 2 asks for twice the LDS a workgroup needs — half the blocks in flight per CU, same code)
#endif
constexpr size_t GI2_LDS_BYTES = sizeof(gi2::Work) * GI2_LPW * FASTF_GI2_LDS_MULT;
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI2_STAMPS)
__device__ unsigned long long gi2_stamp_acc[8];
#endif
__global__ __launch_bounds__(GI2_LPW) void bgzf_decode_kernel(const GiBlock* __restrict__ blk, u32 n_blk, const uint8_t* __restrict__ comp,
                                                               uint8_t* __restrict__ out, u32* __restrict__ tokens, u32* __restrict__ tok_count,
                                                               uint8_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gi2_smem[];
    gi2::Work* const ws = reinterpret_cast<gi2::Work*>(gi2_smem);
    const u32 b = blockIdx.x * (u32)GI2_LPW + threadIdx.x;
    if (b >= n_blk) return;
    const GiBlock k = blk[b];
    int rc = 0; u32 nt = 0;
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI2_STAMPS)
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k.isize) rc = gi2::inflate_tokens(ws[threadIdx.x], comp + k.coff, k.clen, out + k.uoff, k.isize, tokens + k.toff, &nt, st);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&gi2_stamp_acc[i], (unsigned long long)st[i]);
#else
    if (k.isize) rc = gi2::inflate_tokens(ws[threadIdx.x], comp + k.coff, k.clen, out + k.uoff, k.isize, tokens + k.toff, &nt);
#endif
    status[b] = (uint8_t)rc;
    tok_count[b] = rc ? 0u : nt;
}
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI2_STAMPS)
extern "C" int fastf_debug_gi2_stamps(unsigned long long out[8]) FASTF_TRY {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gi2_stamp_acc), 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
} FASTF_CATCH_INT
#endif
// Phase 2: one WAVE per block resolves its matches, 64 tokens at a time.  A wave scan over (literals + length) gives every
// match its place.  A match may copy once the bytes it READS are final: the literals are (phase 1), everything in front of the
// batch is, so what it has to wait for are the matches of its own batch whose output overlaps its source — a contiguous run of
// lanes (the outputs lie in order), found once per batch by two binary searches over the lanes' positions and kept as a 64-bit
// mask: ready = none of those lanes is still pending.  Rounds per batch = the depth of the batch's dependency chains (a field
// copied from the record before, which copied it from the one before: two to four on BAM payloads), not the number of matches
// that have a near source (the first form of this kernel waited for "everything below the first pending match": a round per
// such match, 35 ms for a 2.3 GB window).  Inside a round every ready lane copies 16 bytes per step AS DWORDS — five aligned
// loads shifted into place, four unaligned stores: what bounds this kernel is the number of memory requests a CU can issue
// (a lane per match means a cache line per lane: 64 requests per instruction), and a byte per request was 10 ms per 21 000
// blocks — with no wait between a match's steps.  A match that overlaps its own output (distance < length) repeats its first
// `distance` bytes and reads only those.  The sources are read past the CU's L1 (this
// wave stored some of them a round ago: gi_coherent_load8) after the stores before have been acknowledged.
__global__ __launch_bounds__(256) void bgzf_resolve_kernel(const GiBlock* __restrict__ blk, u32 n_blk, uint8_t* __restrict__ out,
                                                           const u32* __restrict__ tokens, const u32* __restrict__ tok_count,
                                                           const uint8_t* __restrict__ status) {
    const int lane = lane_id();
    const u32 b = blockIdx.x * 4u + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= n_blk) return;
    if (status[b]) return;                                             // (declined in phase 1: the host inflates it)
    const GiBlock k = blk[b];
    const u32 n_tok = tok_count[b];
    uint8_t* const o = out + k.uoff;
    const u32* const tk = tokens + k.toff;
    const u64 below_me = lane ? (~0ull >> (64 - lane)) : 0ull;
    u32 pos = 0;                                                       // (uniform) output position in front of this batch
    for (u32 t0 = 0; t0 < n_tok; t0 += 64) {
        const u32 t = t0 + (u32)lane < n_tok ? tk[t0 + (u32)lane] : gi2::TOK_SKIP;
        const bool skip = (t & gi2::TOK_SKIP) != 0;
        const u32 lit = skip ? t >> 16 : t >> 24, len = skip ? 0u : ((t >> 16) & 255u) + 3u, dist = (t & 0x7FFFu) + 1u;
        const u32 incl = wave_incl_scan32(lit + len, lane);
        const u32 dend = pos + incl, dst = dend - len, src = dst - dist;   // (phase 1 checked: dist <= dst, dst + len <= isize)
        pos += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u32 rd_end = src + (len < dist ? len : dist);            // the bytes this match reads: [src, rd_end)
        // lanes whose output overlaps them: those with dend > src and dst < rd_end (both arrays ascend with the lane)
        u32 i0 = 0, i1 = 0;
#pragma unroll
        for (u32 s_ = 32; s_ >= 1; s_ >>= 1) {
            const u32 e = (u32)__shfl((int)dend, (int)(i0 + s_ - 1u), WAVE), d = (u32)__shfl((int)dst, (int)(i1 + s_ - 1u), WAVE);
            if (e <= src) i0 += s_;
            if (d < rd_end) i1 += s_;
        }
        const u64 upto1 = i1 >= 64u ? ~0ull : ((1ull << i1) - 1ull), upto0 = (1ull << i0) - 1ull;     // (i0 <= 63)
        u64 pending = __ballot(len != 0);
        const u64 deps = upto1 & ~upto0 & below_me;
        // how a match copies: 0 = its source lies wholly in front of its output (distance >= length: nearly all) — 16 bytes per
        // step as four dwords; 1 = it repeats its last 1..3 bytes (a run of one quality value: distance < 4 <= length) — the
        // pattern is built once, in registers; 2 = it repeats a longer stretch (4 <= distance < length: rare) — byte by byte
        const u32 mode = dist >= len ? 0u : (dist < 4u ? 1u : 2u);
        while (pending) {
            const bool ready = ((pending >> lane) & 1ull) && (pending & deps) == 0ull;   // (the first pending match always is)
            const u64 rm = __ballot(ready);
            const u32 maxlen = (u32)__builtin_amdgcn_readfirstlane((int)wave_max32(ready ? len : 0u));
            const uint8_t* const sp = o + src;
            uint8_t* const dp = o + dst;
            // the pattern of a mode-1 match as three dwords (bytes 0..11 of its output; byte j = source byte j mod distance)
            u32 pat0 = 0, pat1 = 0, pat2 = 0;
            if (__ballot(ready && mode == 1u)) {
                if (ready && mode == 1u) {
                    const u32 sh = (u32)(reinterpret_cast<uintptr_t>(sp) & 3u);
                    const u32 w0 = gi_l2_load32(sp - sh), w1 = gi_l2_load32(sp - sh + 4);
                    const u32 v = __builtin_amdgcn_alignbyte(w1, w0, sh);
                    const u32 b0 = v & 255u, b1 = dist > 1u ? (v >> 8) & 255u : b0, b2 = dist > 2u ? (v >> 16) & 255u : (dist > 1u ? b0 : b0);
                    // distance 1: b0 b0 b0 ..; 2: b0 b1 b0 b1 ..; 3: b0 b1 b2 b0 ..
                    const u32 c0 = b0, c1 = b1, c2 = dist == 2u ? b0 : b2;        // bytes 0, 1, 2 of the sequence
                    const u32 c3 = dist == 3u ? b0 : (dist == 2u ? b1 : b0);
                    pat0 = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
                    if (dist == 3u) { pat1 = b1 | (b2 << 8) | (b0 << 16) | (b1 << 24); pat2 = b2 | (b0 << 8) | (b1 << 16) | (b2 << 24); }
                    else { pat1 = pat0; pat2 = pat0; }
                }
            }
            u32 m = 0;                                                 // mode 2: position inside the repeating source
            for (u32 kk = 0; kk < maxlen; kk += 16) {
                const bool act = ready && kk < len;
                const u32 n = act ? (len - kk < 16u ? len - kk : 16u) : 0u;       // bytes of this step
                u32 d0 = pat0, d1 = pat1, d2 = pat2, d3 = pat0;        // (mode 1: four dwords of the sequence, rotated below)
                if (__ballot(act && mode == 0u)) {
                    // five aligned dwords around [sp + kk, sp + kk + 16), shifted into place (the buffer is readable a few
                    // bytes past the block: the window buffers carry slack)
                    const u32 sh = (u32)(reinterpret_cast<uintptr_t>(sp + kk) & 3u);
                    const uint8_t* const a = sp + kk - sh;
                    u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0;
                    if (act && mode == 0u) {
                        w0 = gi_l2_load32(a); w1 = gi_l2_load32(a + 4);
                        if (sh + n > 8u) w2 = gi_l2_load32(a + 8);
                        if (sh + n > 12u) w3 = gi_l2_load32(a + 12);
                        if (sh + n > 16u) w4 = gi_l2_load32(a + 16);
                        d0 = __builtin_amdgcn_alignbyte(w1, w0, sh); d1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
                        d2 = __builtin_amdgcn_alignbyte(w3, w2, sh); d3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
                    }
                }
                if (act && mode != 2u) {
                    uint8_t* const q = dp + kk;
                    // whole dwords (unaligned stores: the hardware splits them), then the last 1..3 bytes
                    if (n >= 4u) __builtin_memcpy(q, &d0, 4);
                    if (n >= 8u) __builtin_memcpy(q + 4, &d1, 4);
                    if (n >= 12u) __builtin_memcpy(q + 8, &d2, 4);
                    if (n >= 16u) __builtin_memcpy(q + 12, &d3, 4);
                    const u32 full = n & ~3u, tail = n & 3u;
                    const u32 dt = full == 0u ? d0 : full == 4u ? d1 : full == 8u ? d2 : d3;
                    if (tail >= 1u) q[full] = (uint8_t)dt;
                    if (tail >= 2u) q[full + 1u] = (uint8_t)(dt >> 8);
                    if (tail >= 3u) q[full + 2u] = (uint8_t)(dt >> 16);
                }
                if (mode == 1u) {                                      // the next step's sequence begins 16 bytes on: 16 mod 3 = 1 dword further
                    const u32 t_ = pat0; pat0 = pat1; pat1 = pat2; pat2 = t_;
                }
                if (__ballot(act && mode == 2u)) {
                    uint8_t v[16];
#pragma unroll
                    for (u32 u = 0; u < 16; ++u) {
                        const bool a2 = act && mode == 2u && u < n;
                        v[u] = a2 ? gi_coherent_load8(sp + m) : (uint8_t)0;
                        m = m + 1u == dist ? 0u : m + 1u;
                    }
#pragma unroll
                    for (u32 u = 0; u < 16; ++u) if (act && mode == 2u && u < n) dp[kk + u] = v[u];
                }
                // (no wait between the steps of a match: it reads [src, src + min(len, dist)) only, which lies in front of its
                //  own output and was final when the round began)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the round's stores have landed before the next round reads them
            pending &= ~rm;
        }
    }
}

struct fastf_gpuinf {
    int device = 0;
    static constexpr int NS = 3;                                   // slices in flight
    hipStream_t s[NS] = {nullptr, nullptr, nullptr};
    GiBlock* h_blk[NS] = {nullptr, nullptr, nullptr}; size_t h_blk_cap[NS] = {0, 0, 0};
    uint8_t* h_status[NS] = {nullptr, nullptr, nullptr};
    DevBuf d_comp[NS], d_out[NS], d_blk[NS], d_status[NS];
    DevBuf d_tok[NS], d_tokn[NS];                                  // two-kernel inflate: the blocks' token lists and their lengths
    bool wave_kernel = false;                                      // FASTF_GI_KERNEL=wave: round 4's one-wavefront-per-block kernel
    hipEvent_t ev0 = nullptr, ev_done[NS] = {nullptr, nullptr, nullptr};
    size_t first[NS + 1] = {0, 0, 0, 0}; size_t pending_n = 0; int pending_slices = 0;
    u64 n_blocks = 0, n_declined = 0;
    // keep mode (fastf_gpuinf_submit_keep): the inflated bytes stay on the device, in one window buffer per parity that
    // mirrors the host window's offsets; CRC-32 per block on the device; fastf_gpurec_parse hops and packs from there
    DevBuf d_win[2]; GrCrcBlock* h_crc[NS] = {nullptr, nullptr, nullptr}; DevBuf d_crc[NS]; DevBuf d_crctab;
    int keep_parity = -1;
    hipStream_t s_parse = nullptr;
    DevBuf d_seg, d_offs, d_result, d_soa[2]; u64 soa_cap[2] = {0, 0};
    u64* h_result = nullptr;
    u64 n_parsed_windows = 0, n_parse_fallbacks = 0, n_parse_repairs = 0;      // (repairs: segments whose chain gr_repair_kernel walked again)
};

extern "C" fastf_gpuinf_t* fastf_gpuinf_create(int device) FASTF_TRY {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { set_err("no HIP device for the BGZF inflate"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    fastf_gpuinf* g = new fastf_gpuinf();
    g->device = device;
    { const char* kv = getenv("FASTF_GI_KERNEL"); g->wave_kernel = kv && !strcmp(kv, "wave"); }
    if (!g->wave_kernel && hipFuncSetAttribute((const void*)bgzf_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GI2_LDS_BYTES) != hipSuccess) {
        delete g; set_err("cannot raise the dynamic LDS limit of bgzf_decode_kernel to %zu bytes", GI2_LDS_BYTES); return nullptr;
    }
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
        if (g->h_crc[i]) (void)hipHostFree(g->h_crc[i]);
        g->d_comp[i].release(); g->d_out[i].release(); g->d_blk[i].release(); g->d_status[i].release(); g->d_crc[i].release();
        g->d_tok[i].release(); g->d_tokn[i].release();
    }
    if (g->s_parse) { (void)hipStreamSynchronize(g->s_parse); (void)hipStreamDestroy(g->s_parse); }
    if (g->h_result) (void)hipHostFree(g->h_result);
    g->d_win[0].release(); g->d_win[1].release(); g->d_crctab.release(); g->d_seg.release(); g->d_offs.release(); g->d_result.release();
    g->d_soa[0].release(); g->d_soa[1].release();
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
static int gpuinf_submit_impl(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                              unsigned char* out, int keep_parity, const uint32_t* crc);
extern "C" int fastf_gpuinf_submit(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                                   unsigned char* out) FASTF_TRY {
    return gpuinf_submit_impl(g, comp, blk, n, out, -1, nullptr);
} FASTF_CATCH_INT
// keep mode: the inflated bytes of blocks [0, n) stay on the device, in the window buffer of `parity` (block i at its host
// offset blk[i].uoff), and every block's CRC-32 is compared with crc[i] (its trailer) on the device: status bit 1.
// Nothing is copied back; fastf_gpurec_parse / fastf_gpurec_fetch read the window buffer afterwards.
extern "C" int fastf_gpuinf_submit_keep(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                                        int parity, const uint32_t* crc) FASTF_TRY {
    if (parity < 0 || parity > 1 || !crc) return set_err("bad keep-mode arguments");
    return gpuinf_submit_impl(g, comp, blk, n, nullptr, parity, crc);
} FASTF_CATCH_INT

static void gr_crc_tables(uint32_t* t) {                     // [0, 256): byte table; [256, 288): x^(2^i) mod P
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = c & 1 ? (c >> 1) ^ gr::CRC_POLY : c >> 1; t[i] = c; }
    uint32_t p = 1u << 30;                                   // x^1
    t[256] = p;
    for (int i = 1; i < 32; ++i) t[256 + i] = p = gr::crc_multmodp(p, p);
}

// pinned descriptor / status / CRC arrays of slice q for nb blocks
static int gpuinf_slice_host(fastf_gpuinf_t* g, int q, size_t nb) {
    if (nb <= g->h_blk_cap[q]) return 0;
    if (g->h_blk[q]) (void)hipHostFree(g->h_blk[q]);
    if (g->h_status[q]) (void)hipHostFree(g->h_status[q]);
    if (g->h_crc[q]) (void)hipHostFree(g->h_crc[q]);
    g->h_blk[q] = nullptr; g->h_status[q] = nullptr; g->h_crc[q] = nullptr; g->h_blk_cap[q] = 0;
    const size_t cap = nb + nb / 4 + 64;
    HIP_OK(hipHostMalloc((void**)&g->h_blk[q], cap * sizeof(GiBlock), hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&g->h_status[q], cap, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&g->h_crc[q], cap * sizeof(GrCrcBlock), hipHostMallocDefault));
    g->h_blk_cap[q] = cap;
    return 0;
}
static int gpuinf_crc_table(fastf_gpuinf_t* g) {
    if (g->d_crctab.p) return 0;
    uint32_t t[288]; gr_crc_tables(t);
    if (g->d_crctab.ensure(sizeof t)) return 1;
    if (copy_h2d(g->d_crctab.p, t, sizeof t)) return 1;
    return 0;
}
static int gpurec_parse_buffers(fastf_gpuinf_t* g, int parity, u64 bytes) {       // for a parse over `bytes` of window
    if (!g->s_parse) HIP_OK(hipStreamCreateWithFlags(&g->s_parse, hipStreamNonBlocking));
    if (!g->h_result) HIP_OK(hipHostMalloc((void**)&g->h_result, 8 * sizeof(u64), hipHostMallocDefault));
    const u32 n_seg = (u32)((bytes + GR_SEG - 1) / GR_SEG);
    const u64 cap = bytes / 36 + 1;
    if (g->d_seg.ensure((size_t)n_seg * sizeof(GrSeg)) || g->d_offs.ensure((size_t)n_seg * GR_SEG_RECS * sizeof(u32)) || g->d_result.ensure(8 * sizeof(u64))) return 1;
    if (g->soa_cap[parity] < cap) {
        if (g->d_soa[parity].ensure((size_t)cap * 24)) return 1;
        g->soa_cap[parity] = cap;
    }
    return 0;
}

// Everything a keep-mode window of up to `window_bytes` inflated / `comp_bytes` compressed bytes in `n_blocks` blocks needs,
// allocated ahead of the first window (the reader's init thread calls this while the host threads inflate alone): the two
// window buffers, the slices' staging and descriptor arrays, the CRC table, the parse buffers.  Without it the first shared
// window pays ~15 ms of allocations and every growth of the device's share another (hipFree waits for the device).
extern "C" int fastf_gpuinf_reserve(fastf_gpuinf_t* g, size_t window_bytes, size_t comp_bytes, size_t n_blocks) FASTF_TRY {
    if (!g) return set_err("null inflate handle");
    HIP_OK(hipSetDevice(g->device));
    if (gpuinf_crc_table(g)) return 1;
    for (int parity = 0; parity < 2; ++parity) {
        if (g->d_win[parity].ensure(window_bytes + 4096)) return 1;
        if (gpurec_parse_buffers(g, parity, window_bytes)) return 1;
    }
    for (int q = 0; q < fastf_gpuinf::NS; ++q) {
        if (gpuinf_slice_host(g, q, n_blocks)) return 1;
        const size_t cap = g->h_blk_cap[q];
        if (g->d_comp[q].ensure(comp_bytes + 1024) || g->d_blk[q].ensure(cap * sizeof(GiBlock)) || g->d_status[q].ensure(cap) ||
            g->d_crc[q].ensure(cap * sizeof(GrCrcBlock)))
            return 1;
        // token lists of a slice (a third to a half of a window): at most a token per three bytes (gi2::token_cap)
        if (!g->wave_kernel && (g->d_tok[q].ensure((window_bytes / 2 / 3 + window_bytes / 2 / 256 + 4 * cap) * sizeof(u32)) || g->d_tokn[q].ensure(cap * sizeof(u32)))) return 1;
    }
    return 0;
} FASTF_CATCH_INT

static int gpuinf_submit_impl(fastf_gpuinf_t* g, const unsigned char* comp, const fastf_gpuinf_blk_t* blk, size_t n,
                              unsigned char* out, int keep_parity, const uint32_t* crc) {
    if (!g) return set_err("null inflate handle");
    g->pending_n = 0;
    g->keep_parity = keep_parity;
    if (n == 0) return 0;
    HIP_OK(hipSetDevice(g->device));
    const bool keep = keep_parity >= 0;
    // FASTF_DEBUG_PINS: the compressed bytes go to the device straight from `comp`, the inflated ones come back straight into
    // `out` (copy-back mode): both must be memory the runtime has been told about (the reader's pinned staging / pinned window front)
    if (debug_known_memory(comp + blk[0].coff, blk[0].clen, "fastf_gpuinf_submit comp") ||
        (!keep && out && debug_known_memory(out + blk[0].uoff, blk[0].isize ? blk[0].isize : 1, "fastf_gpuinf_submit out"))) return 1;
    if (keep) {
        u64 wend = 0;
        for (size_t i = 0; i < n; ++i) wend = std::max<u64>(wend, blk[i].uoff + blk[i].isize);
        // (room for a growing share: a buffer that is replaced costs a hipFree, which waits for the whole device)
        if (g->d_win[keep_parity].bytes < (size_t)wend + 4096 && g->d_win[keep_parity].ensure((size_t)wend + (size_t)wend / 4 + 4096)) return 1;
        if (gpuinf_crc_table(g)) return 1;
    }
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
        if (gpuinf_slice_host(g, q, nb)) return 1;
        u64 n_tok_cap = 0;
        for (size_t i = a; i < b; ++i) {
            g->h_blk[q][i - a] = GiBlock{blk[i].coff - c0, blk[i].clen, blk[i].isize, blk[i].uoff - u0, n_tok_cap};
            n_tok_cap += gi2::token_cap(blk[i].isize);
        }
        if (g->d_comp[q].ensure(cbytes + 512) || (!keep && g->d_out[q].ensure(std::max<size_t>(ubytes, 64) + 64)) || g->d_blk[q].ensure(nb * sizeof(GiBlock)) ||
            g->d_status[q].ensure(nb))
            return 1;
        if (!g->wave_kernel && (g->d_tok[q].ensure((size_t)n_tok_cap * sizeof(u32)) || g->d_tokn[q].ensure(nb * sizeof(u32)))) return 1;
        uint8_t* const d_dst = keep ? (uint8_t*)g->d_win[keep_parity].p + u0 : (uint8_t*)g->d_out[q].p;     // block i lands at d_dst + (uoff - u0)
        hipStream_t s = g->s[q];
        if (q) HIP_OK(hipStreamWaitEvent(s, g->ev0, 0));
        // (the lane decoder reading the compressed bytes straight out of the caller's pinned buffer — no copy, 32 bytes per lane
        //  over PCIe an epoch ahead of their use — was measured and loses: 420-470 against 610-740 blocks per ms end to end,
        //  profiles/r5_notes/e2e_windows_lane_decoder.txt)
        const uint8_t* d_src = (const uint8_t*)g->d_comp[q].p;
        HIP_OK(hipMemcpyAsync(g->d_comp[q].p, comp + c0, cbytes, hipMemcpyHostToDevice, s));
        HIP_OK(hipMemcpyAsync(g->d_blk[q].p, g->h_blk[q], nb * sizeof(GiBlock), hipMemcpyHostToDevice, s));
        if (g->wave_kernel)
            hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((u32)nb), dim3(64), 0, s, (const GiBlock*)g->d_blk[q].p, (u32)nb,
                               (const uint8_t*)g->d_comp[q].p, d_dst, (uint8_t*)g->d_status[q].p);
        else {
            hipLaunchKernelGGL(bgzf_decode_kernel, dim3((u32)((nb + GI2_LPW - 1) / GI2_LPW)), dim3(GI2_LPW), GI2_LDS_BYTES, s, (const GiBlock*)g->d_blk[q].p, (u32)nb,
                               d_src, d_dst, (u32*)g->d_tok[q].p, (u32*)g->d_tokn[q].p, (uint8_t*)g->d_status[q].p);
            hipLaunchKernelGGL(bgzf_resolve_kernel, dim3((u32)((nb + 3) / 4)), dim3(256), 0, s, (const GiBlock*)g->d_blk[q].p, (u32)nb, d_dst,
                               (const u32*)g->d_tok[q].p, (const u32*)g->d_tokn[q].p, (const uint8_t*)g->d_status[q].p);
        }
        HIP_OK(hipGetLastError());
        if (keep) {
            for (size_t i = a; i < b; ++i) g->h_crc[q][i - a] = GrCrcBlock{blk[i].uoff - u0, blk[i].isize, crc[i]};
            if (g->d_crc[q].ensure(nb * sizeof(GrCrcBlock))) return 1;
            HIP_OK(hipMemcpyAsync(g->d_crc[q].p, g->h_crc[q], nb * sizeof(GrCrcBlock), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(gr_crc_kernel, dim3((u32)nb), dim3(64), 0, s, (const GrCrcBlock*)g->d_crc[q].p, (u32)nb, (const uint8_t*)d_dst,
                               (const uint32_t*)g->d_crctab.p, (uint8_t*)g->d_status[q].p);
            HIP_OK(hipGetLastError());
        } else if (ubytes) HIP_OK(hipMemcpyAsync(out + u0, g->d_out[q].p, ubytes, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(g->h_status[q], g->d_status[q].p, nb, hipMemcpyDeviceToHost, s));
        HIP_OK(hipEventRecord(g->ev_done[q], s));
    }
    g->pending_n = n; g->pending_slices = n_slices;
    return 0;
}

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
    // (tests and tools call this with ordinary memory; the reader's own buffers are pinned.  Registered for the call: the runtime
    //  is told about the ranges instead of pinning them on the fly behind our back — umi_engine.hip, "pageable host memory")
    size_t cend = 0, uend = 0;
    for (size_t i = 0; i < n; ++i) { cend = std::max<size_t>(cend, blk[i].coff + blk[i].clen); uend = std::max<size_t>(uend, blk[i].uoff + blk[i].isize); }
    const bool reg_c = g && n && cend && pin_reg((void*)comp, cend + 64) == 0;   // (+ 64: the submit copies the readable slack behind the last block too)
    const bool reg_u = g && n && uend && pin_reg((void*)out, uend) == 0;
    int rc = fastf_gpuinf_submit(g, comp, blk, n, out);
    if (!rc) rc = fastf_gpuinf_wait(g, status, nullptr);
    if (reg_c) pin_unreg((void*)comp);
    if (reg_u) pin_unreg((void*)out);
    return rc;
} FASTF_CATCH_INT

// ---- second stage: record hop + tag extraction + key packing on the window buffer of `parity` (keep mode) ----
// The window buffer mirrors the host window's offsets.  `tail` (tail_len bytes, a record boundary at its first byte: what
// the host carried over from the window before) is placed in front of offset `data_off` (= the host window's reserve), the
// records that START in [data_off - tail_len, end) and are complete there are hopped and packed.
//   out->status     0 ok; 1 = the chains of the segments did not line up (a record longer than a segment, a wrong guess):
//                   nothing was packed, the caller fetches the bytes and parses on the host
//   out->n          records packed, as SoA in device memory (out->batch: valid until the second next parse of this parity)
//   out->handover   offset behind the last complete record: the host goes on from there (fastf_gpurec_fetch brings the
//                   bytes [handover, end) over)
static void gr_dict_from_view(gr::Dict& d, const fastf_keydict_view_t& v) {
    memset(&d, 0, sizeof d);
    d.n_prefix = v.n_prefix;
    for (uint32_t i = 0; i < v.n_prefix && i < (uint32_t)gr::MAX_PREFIX; ++i) {
        d.plen[i] = v.prefix_len[i]; d.pid[i] = v.prefix_id[i];
        memcpy(d.ptext[i], v.prefix[i], gr::MAX_PREFIX_LEN);
    }
}
extern "C" int fastf_gpurec_parse(fastf_gpuinf_t* g, int parity, const unsigned char* tail, size_t tail_len, uint64_t data_off,
                                  uint64_t end, uint32_t n_ref, const fastf_keydict_view_t* cells, const fastf_keydict_view_t* feats,
                                  fastf_gpurec_result_t* out) FASTF_TRY {
    if (!g || !out || !cells || !feats || parity < 0 || parity > 1) return set_err("bad fastf_gpurec_parse arguments");
    if (tail_len > data_off || end < data_off || !g->d_win[parity].p || g->d_win[parity].bytes < end) return set_err("fastf_gpurec_parse: window not on the device");
    HIP_OK(hipSetDevice(g->device));
    const u64 start = data_off - tail_len;
    memset(out, 0, sizeof *out);
    out->handover = start;
    if (end - start < 36) return 0;                                    // not even a fixed part: all of it is the host's
    if (gpurec_parse_buffers(g, parity, end - start)) return 1;
    hipStream_t s = g->s_parse;
    uint8_t* win = (uint8_t*)g->d_win[parity].p;
    if (tail_len && copy_h2d_on(win + start, tail, tail_len, s)) return 1;       // (the tail lies in the reader's host window: ordinary memory)
    const u32 n_seg = (u32)((end - start + GR_SEG - 1) / GR_SEG);
    const u64 sc = g->soa_cap[parity];
    u64* cb = (u64*)g->d_soa[parity].p; u64* gx = cb + sc; u32* umi = (u32*)(gx + sc); u32* meta = umi + sc;
    gr::Dict dc, df;
    gr_dict_from_view(dc, *cells); gr_dict_from_view(df, *feats);
    hipLaunchKernelGGL(gr_hop_kernel, dim3(n_seg), dim3(64), 0, s, (const uint8_t*)win, start, end, n_ref, (GrSeg*)g->d_seg.p, (u32*)g->d_offs.p);
    hipLaunchKernelGGL(gr_stitch_kernel, dim3(1), dim3(1024), 0, s, (GrSeg*)g->d_seg.p, n_seg, end, (u64*)g->d_result.p);
    HIP_OK(hipMemcpyAsync(g->h_result, g->d_result.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, s));
    HIP_OK(hipStreamSynchronize(s));
    g->n_parsed_windows++;
    if (g->h_result[0]) {
        // chains that do not meet (a guessed start inside a record, a record longer than a segment): walked again from where the
        // chain in front ends, then stitched again — the window stays on the device
        hipLaunchKernelGGL(gr_repair_kernel, dim3(1), dim3(64), 0, s, (const uint8_t*)win, start, end, (GrSeg*)g->d_seg.p, (u32*)g->d_offs.p, n_seg, (u64*)g->d_result.p + 5);
        hipLaunchKernelGGL(gr_stitch_kernel, dim3(1), dim3(1024), 0, s, (GrSeg*)g->d_seg.p, n_seg, end, (u64*)g->d_result.p);
        HIP_OK(hipMemcpyAsync(g->h_result, g->d_result.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, s));
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(s));
        g->n_parse_repairs += g->h_result[5];
    }
    if (g->h_result[0]) {
        out->status = 1; g->n_parse_fallbacks++;
        const char* pv = getenv("FASTF_BAM_PROFILE");
        if (pv && pv[0] == '2') {                                      // where the chains parted
            std::vector<GrSeg> hs(n_seg);
            if (copy_d2h(hs.data(), g->d_seg.p, (size_t)n_seg * sizeof(GrSeg)) == 0)
                for (u32 i = 0; i < n_seg; ++i)
                    if (hs[i].first == GR_NONE || (i > 0 && hs[i - 1].exit_off != hs[i].first)) {
                        fprintf(stderr, "[bam] device parse: window [%llu, %llu), %u segments: segment %u begins at %lld (lo %llu), the chain before it ends at %llu\n",
                                (unsigned long long)start, (unsigned long long)end, n_seg, i, hs[i].first == GR_NONE ? -1ll : (long long)hs[i].first,
                                (unsigned long long)(start + (u64)i * GR_SEG), i ? (unsigned long long)hs[i - 1].exit_off : 0ull);
                        break;
                    }
        }
        return 0;
    }
    const u64 n_rec = g->h_result[1];
    if (n_rec > sc) return set_err("internal: %llu records in a window sized for %llu", (unsigned long long)n_rec, (unsigned long long)sc);
    if (n_rec) {
        hipLaunchKernelGGL(gr_pack_kernel, dim3(n_seg), dim3(GR_PACK_THREADS), 0, s, (const uint8_t*)win, start, end, (const GrSeg*)g->d_seg.p, (const u32*)g->d_offs.p,
                           dc, df, cb, gx, umi, meta, sc, (u64*)g->d_result.p);
        HIP_OK(hipMemcpyAsync(g->h_result, g->d_result.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, s));
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(s));
    }
    out->n = n_rec; out->handover = g->h_result[2]; out->no_xf = g->h_result[3]; out->no_gx = g->h_result[4];
    out->batch.cb_key = (const uint64_t*)cb; out->batch.gx_key = (const uint64_t*)gx; out->batch.umi = umi; out->batch.meta = meta; out->batch.n = (size_t)n_rec;
    return 0;
} FASTF_CATCH_INT

// bytes [from, to) of the window buffer of `parity` into dst (dst[0] = byte `from`); dst pinned or not
extern "C" int fastf_gpurec_fetch(fastf_gpuinf_t* g, int parity, unsigned char* dst, uint64_t from, uint64_t to) FASTF_TRY {
    if (!g || parity < 0 || parity > 1 || to < from || !g->d_win[parity].p || g->d_win[parity].bytes < to) return set_err("bad fastf_gpurec_fetch arguments");
    if (to == from) return 0;
    HIP_OK(hipSetDevice(g->device));
    if (copy_d2h(dst, (const uint8_t*)g->d_win[parity].p + from, (size_t)(to - from))) return 1;       // (dst: the reader's host window, pinned or not)
    return 0;
} FASTF_CATCH_INT

extern "C" uint64_t fastf_gpurec_repairs(const fastf_gpuinf_t* g) { return g ? g->n_parse_repairs : 0; }
extern "C" void fastf_gpurec_stats(const fastf_gpuinf_t* g, uint64_t* windows, uint64_t* fallbacks) FASTF_TRY {
    if (windows) *windows = g ? g->n_parsed_windows : 0;
    if (fallbacks) *fallbacks = g ? g->n_parse_fallbacks : 0;
} FASTF_CATCH_VOID
