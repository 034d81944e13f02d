// umi_engine.hip — the engine behind the C ABI of include/fastf_amd.h (sections 2b and 3).
//
// Replaces the reference's record loop + SQLite aggregate (bam2db_ds.c:360-438, 480-483,
// 539-542) with: H2D of packed SoA batches on a copy stream → K1 probe/filter/pack →
// (finish) K2 LSD radix sort → K3 segmented unique/reduce → COO D2H.
// There is NO CPU fallback: every entry point fails loudly when HIP is unusable.
#include "umi_kernels.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

extern "C" {
#include "fastf_amd.h"
#include "host_io.h"
#include "mt_jump.h"
}

using namespace fastf;

// The last error of the process, whichever thread raised it: the BAM reader's prefetch and inflate workers and the
// decoder thread of bam2db() fail on their own threads, and the caller of fastf_last_error() must still see why.
// Readers get a thread-local snapshot, so the returned pointer stays valid while other threads keep working.
static std::mutex g_err_mu;
static char g_err[512] = "";

static int set_err(const char* fmt, ...) {
    char buf[sizeof g_err];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(g_err, buf, sizeof g_err);
    return 1;
}
// Exception barrier of the C ABI.  The reference contract is "return 1 + a message on stderr" (bam2db_ds.h:60); a C
// caller cannot catch, so a std::bad_alloc / std::length_error from a host-side container (row buffers sized from
// device counters, per-device lists of the multi-device engine) that left an extern "C" function would end in
// std::terminate and SIGABRT inside the caller's process.  Every extern "C" function of this translation unit with a
// body of more than one statement is a function-try-block closed by one of these (tests/test_abi.py checks that none
// is left out, and drives fastf_debug_raise_ through it).
static int api_caught(const char* fn, const char* what) {
    return set_err("%s: %s", fn, what ? what : "unknown C++ exception");
}
#define FASTF_TRY try
#define FASTF_CATCH_(ret_stmt)                                                                       \
    catch (const std::bad_alloc&) { (void)api_caught(__func__, "out of host memory (std::bad_alloc)"); ret_stmt; }   \
    catch (const std::exception& x_) { (void)api_caught(__func__, x_.what()); ret_stmt; }             \
    catch (...) { (void)api_caught(__func__, nullptr); ret_stmt; }
#define FASTF_CATCH_INT  FASTF_CATCH_(return 1)
#define FASTF_CATCH_ZERO FASTF_CATCH_(return 0)
#define FASTF_CATCH_VOID FASTF_CATCH_(return)

// test hook (tests/test_abi.py): raises inside the barrier the way a failing allocation would
extern "C" int fastf_debug_raise_(int kind) FASTF_TRY {
    if (kind == 0) { std::vector<u64> v; v.resize(~(size_t)0 / 2); return (int)v.size(); }   // std::length_error
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) throw 42;                                                                   // not a std::exception
    if (kind == 3) { std::vector<std::vector<u64>> at(4); at.at(7).push_back(1); }             // std::out_of_range
    return 0;
} FASTF_CATCH_INT

extern "C" const char* fastf_last_error(void) {
    static thread_local char snap[sizeof g_err];
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(snap, g_err, sizeof snap);
    return snap;
}
#ifdef FASTF_EXPERIMENT
extern "C" const char* fastf_version(void) { return "fastf_amd 0.4 (gfx950) EXPERIMENT BUILD: results may be wrong"; }
#else
extern "C" const char* fastf_version(void) { return "fastf_amd 0.4 (gfx950)"; }
#endif
extern "C" void fastf_set_error_(const char* msg) { set_err("%s", msg); }

#define HIP_OK(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static u32 bits_for(u64 max_value) {   // bits needed to hold 0..max_value
    u32 b = 0;
    while (b < 64 && (max_value >> b)) b++;
    return b ? b : 1;
}

static u32 sort_passes(u32 key_bits, u32 low_bit) { return key_bits > low_bit ? (key_bits - low_bit + 7) / 8 : 0; }

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        HIP_OK(hipMalloc(&p, need));
        bytes = need;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// ---- host memory and the HIP runtime ----
// Three rules, each with its own mechanism:
//  1. Host memory the runtime has not been told about (malloc, std::vector, the caller's arrays) is never handed to hipMemcpy*: it
//     goes through a pinned bounce buffer of the library's own.  For unknown memory of a megabyte or more the runtime pins the range
//     on the fly, has the GPU read or write the caller's pages directly and lets go again; all three GPU memory faults of rounds
//     5/6 were GPU writes to heap addresses while such a copy was running (profiles/r6_notes/gpu_fault_chain.md).  Off the hot
//     path: tables at create, rows of a first finish into a buffer that is not pinned yet, -u rows, a reader window's tail.
//  2. Every hipHostRegister / hipHostUnregister of the library goes through pin_reg / pin_unreg below and so through the ledger
//     of host_io.c: a registered range is never released (fastf_big_free / fastf_big_drop check), the ledger is empty when the
//     last handle is gone (tests/test_gpu_parity.py::test_no_registration_outlives_its_buffer).
//  3. Registered memory is a mapping of its own (fastf_big_alloc), never a piece of the malloc heap.
// FASTF_DEBUG_PINS=1: a ledger violation aborts, and every ABI entry that takes pinned host pointers (fastf_engine_push_pinned,
// fastf_engine_lend_rows, fastf_dev_rows_gather) asks the runtime whether it knows the memory (hipPointerGetAttributes).
static int pin_reg(void* p, size_t bytes) {
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
    fastf_pin_ledger_add(p, bytes);
    return 0;
}
static void pin_unreg(void* p) {
    if (!p) return;
    (void)fastf_pin_ledger_remove(p);
    if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError();
}
static bool pins_debug() { static const bool v = [] { const char* e = getenv("FASTF_DEBUG_PINS"); return e && e[0] == '1'; }(); return v; }
// debug builds of a run: [p, p + bytes) must be memory the runtime knows (pinned host or device memory)
static int debug_known_memory(const void* p, size_t bytes, const char* what) {
    if (!pins_debug() || !p || !bytes) return 0;
    for (const char* q : {(const char*)p, (const char*)p + bytes - 1}) {
        hipPointerAttribute_t a; memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, q) != hipSuccess || a.type == hipMemoryTypeUnregistered) {
            (void)hipGetLastError();
            return set_err("FASTF_DEBUG_PINS: %s: %p (+%zu) is host memory the HIP runtime has not been told about", what, p, bytes);
        }
    }
    return 0;
}
struct HostBounce {
    static constexpr size_t BYTES = (size_t)8 << 20;
    std::mutex mu; void* p = nullptr;
    int ready() {                                       // (caller holds mu)
        if (p) return 0;
        if (hipHostMalloc(&p, BYTES, hipHostMallocPortable) != hipSuccess) { p = nullptr; return set_err("hipHostMalloc(bounce buffer) failed: %s", hipGetErrorString(hipGetLastError())); }
        return 0;
    }
};
static HostBounce g_bounce;
// synchronous; the current device must be the one d_dst / d_src lives on
static int copy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_bounce.mu);
    if (g_bounce.ready()) return 1;
    for (size_t o = 0; o < bytes; o += HostBounce::BYTES) {
        const size_t n = std::min(bytes - o, HostBounce::BYTES);
        memcpy(g_bounce.p, (const char*)h_src + o, n);
        HIP_OK(hipMemcpy((char*)d_dst + o, g_bounce.p, n, hipMemcpyHostToDevice));
    }
    return 0;
}
static int copy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_bounce.mu);
    if (g_bounce.ready()) return 1;
    for (size_t o = 0; o < bytes; o += HostBounce::BYTES) {
        const size_t n = std::min(bytes - o, HostBounce::BYTES);
        HIP_OK(hipMemcpy(g_bounce.p, (const char*)d_src + o, n, hipMemcpyDeviceToHost));
        memcpy((char*)h_dst + o, g_bounce.p, n);
    }
    return 0;
}
// the same onto stream s (and waited for: the bounce buffer is free again when this returns)
static int copy_h2d_on(void* d_dst, const void* h_src, size_t bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_bounce.mu);
    if (g_bounce.ready()) return 1;
    for (size_t o = 0; o < bytes; o += HostBounce::BYTES) {
        const size_t n = std::min(bytes - o, HostBounce::BYTES);
        memcpy(g_bounce.p, (const char*)h_src + o, n);
        HIP_OK(hipMemcpyAsync((char*)d_dst + o, g_bounce.p, n, hipMemcpyHostToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
    }
    return 0;
}

struct KernelTimer {           // optional per-kernel HIP-event timing (bench roofline leg)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
};

struct fastf_multi;
struct fastf_engine {
    fastf_multi* multi = nullptr;        // n_devices > 1: this handle only dispatches (multi_engine.hpp)
    int device = 0;
    hipStream_t s_compute = nullptr, s_copy = nullptr, s_mt = nullptr;   // s_mt: the draw stream's generator (mt_fill_kernel), beside the copies
    hipEvent_t ev_mt = nullptr;
    KeyLayout L{};
    u32 cell_bits = 0, feat_bits = 0, n_cells = 0, n_features = 0;
    u64 threshold = 0;
    u32 n_shards = 1, shard_rank = 0;
    DevBuf tab_cells, tab_feats;
    Table cells{}, feats{};
    DevBuf img_cells, img_genes;         // LDS table images (fast path, when the lists allow it)
    DevBuf d_cell_filter; MissFilter cell_filter{nullptr, 0};        // miss filter in front of the L2 cell table
    CellLds lds_cells{}; GeneLds lds_genes{};
    bool cell16 = false;                 // K1a -> K1b scratch holds u16 cell indices (n_cells <= 65535)
    bool use_lds_cells = false, use_lds_genes = false; u32 genes_blocks_per_cu = 1, cells_blocks_per_cu = 1;
    // draw stream: draws live in a device ring indexed by the ABSOLUTE hit rank since the last reset; the hit-rank
    // base of a chunk is a device-side running total, so a push never waits for the counts of the chunk before it
    fastf_mt_t mt{};
    u32 mt_seed0 = 0; u64 mt_skip0 = 0;  // where the engine-owned stream starts (create / reseed)
    u64 mt_hits = 0;                     // hits served from the engine-owned stream so far
    u64 mt_abs0 = 0;                     // absolute hit rank a  <->  draw number a - mt_abs0 of the engine-owned stream
    bool mt_live = false;                // the generator stands at draw number draws_up - mt_abs0
    DevBuf d_ring; u64 ring_len = 0;     // the decision stream: ring_len bits (u32[ring_len / 32]), ring_len a power of two
    u32 ring_carry = 0;                  // host-packed decisions: the bits of the word draws_up stands in (below bit draws_up & 31)
    DevBuf d_dbits;                      // device-level calls that bring 32-bit draws: their decisions (draw_bits_kernel)
    DevBuf d_mtwords;                    // the generator's words of one launch, between mt_fill_kernel and draw_bits_kernel
    DevBuf d_mt; bool mt_on_device = false;  // the engine-owned stream continues on the device (mt_fill_kernel): state words + read index
    u32 mt_dev_idx = MT_N;                   // ... and where in its block that stream stands, as the host knows it (the parallel generator starts at a block boundary)
    DevBuf d_mtsub, d_mtpoly, d_mtseat, d_mtseq;   // parallel generator (jump-ahead): the sub-streams' states; the jump polynomials; the state fastf_dev_mt_decisions seats; the sources' sequences
    u64 draws_up = 0;                    // absolute ranks below this are (being) uploaded
    u64 draws_valid = 0;                 // ranks below this carry a real draw (caller-supplied streams can run short)
    // staging: chunks in flight, double buffered
    u64 batch_cap = 0;
    struct Slot {
        void* h_stage = nullptr;         // pinned: cb | gx | umi | meta  (only pageable input is staged)
        void* h_draws = nullptr;         // pinned: the draws uploaded with this chunk
        DevBuf d_stage;
        hipEvent_t ev_copy = nullptr, ev_done = nullptr;
        u64* h_small = nullptr;          // pinned: d_small as it stood when this chunk's K1 had finished
        bool busy = false; u64 n = 0; bool external = false;
    } slot[2];
    int cur = 0;
    const u32* cur_umi_ext = nullptr;    // (umi_max_bases > 16) the chunk being pushed: bases 17.. of its UMIs in the device staging
    u64 inflight_records = 0;            // records of the chunks not yet retired
    // key store + results
    u64 key_cap = 0;
    DevBuf d_keys, d_tmp;
    DevBuf d_small;              // key_counts[8] | counters[4] | nnz | nrows_u | n_tmp  (u64 each)
    u64* h_small = nullptr;      // pinned mirror
    DevBuf d_feature, d_cell, d_count, d_ukeys, d_ncopy;
    u32* h_coo = nullptr; u64 h_coo_cap = 0, h_nnz = 0;      // feature | cell | count, h_coo_cap rows each; pageable at first, pinned from its second use on
    bool h_coo_pinned = false; u32 h_coo_uses = 0;
    u32* lent_rows = nullptr; u64 lent_cap = 0;              // fastf_engine_lend_rows: pinned memory of the caller's, lent_cap rows per array
    u32* rows_at = nullptr; u64 rows_stride = 0;             // where the rows of the last finish lie (h_coo or the loan)
    std::vector<u32> h_ufeature, h_ucell, h_uumi, h_ncopy;
    std::vector<uint8_t> h_unonnull;
    u64 total_records = 0, hits_so_far = 0, keys_so_far = 0;      // hits / keys: as of the last retired chunk
    u64 c_sampled = 0, c_valid = 0;
    bool finished = false; int sorted_in_tmp = 0; u64 n_sorted = 0;
    u32 skip_bits = 0;           // low key bits the matrix path leaves unsorted (dedup needs adjacency of equal keys only)
    bool fully_sorted = false;
    // workspace
    DevBuf d_cellidx, d_tilecnt, d_tilebase, d_binbase, d_cnt;   // d_binbase: per-pass bin totals
    DevBuf d_halfhits;                   // K1a: hits per 256-record unit (the streaming K1b's rank bases)
    DevBuf d_segkeys;                    // several shards: the streaming K1b's unsharded output (workgroup regions) in front of shard_partition_kernel
    DevBuf d_segcount, d_segprefix, d_tileseg;   // segmented key buffer left by the streaming K1b: counts, prefix sums, first region of each sort tile
    u32 giant_parity = 0;                        // which of the two giant-item counters the next group-only reduce counts up
    u32 seg_n = 0; u64 seg_stride = 0;           // valid for the key buffer of the last FASTF_PROBE_SEGMENTED call (seg_stride 0: regions of several launches, no common stride)
    // per region of the streaming K1b: its first slot in the key store and the histogram of the sort's first digit (scatter_regions_kernel);
    // d_segcount holds the key counts
    DevBuf d_rgn_phys, d_rgn_hist, d_rgn_blk;
    u32 rgn_hist_n = 0, rgn_hist_shift = 0;      // regions whose histograms are in place and unconsumed (0: none), and the digit they count
    // the push path in stream mode (single shard, gene list in LDS, keys of at most 64 bits): every chunk is staged in the blocked
    // layout and runs the streaming K1b; its regions are appended to the key store, which holds SLOTS (regions with gaps) until
    // fastf_engine_finish sorts out of them
    bool stream_mode = false, store_regions = false;
    u64 slots_used = 0, slot_cap = 0; u32 rgn_n = 0, rgn_cap = 0;
    DevBuf d_scanblk;                    // chunk totals of a scan over more than 16 384 tiles
    // K3: row regions (one slot per key: a workgroup's rows go to the slots of its own chunk), rows per chunk and their bases
    DevBuf d_rg_feature, d_rg_cell, d_rg_count, d_rg_ukeys, d_spanrows, d_spanbase, d_giant;
    u32 rg_n = 0; bool rg_umi = false;   // chunks of the last reduce (0: none) and its kind
    // keys wider than 64 bits (cell bits + feature bits + UMI field > 64): the sort key is the GROUP (cell << feat_bits |
    // feature, group_bits wide), the rest of the key (NULL flag, UMI, length: feat_shift bits) travels beside it as a value
    bool wide = false, long_umi = false; u32 group_bits = 0;
    // umi_max_bases > 24: the rest of the key is more than the 52 bits reduce_hashed_kernel<true, true> holds exactly — its top
    // sub_bits (the UMI's first bases, umi_kernels.hpp make_val_sub) sit in the sorted word below the feature (group_bits counts
    // them), K3 emits a row per (cell, feature, sub-group) split into two 32-bit halves at row_split, merge_sub_rows sums them
    u32 sub_bits = 0, row_split = 0;
    bool finish_queued = false;        // finish_queue() ran for the keys in the store; fastf_engine_finish waits and collects
    DevBuf d_vals, d_vtmp;
    // timing
    bool timing = false;
    hipEvent_t t_ev[2] = {nullptr, nullptr};
    double t_scatter_ms = 0; u64 t_scatter_n = 0;
    double t_k1_ms = 0; u64 t_k1_n = 0;
    double t_k1b_ms = 0; u64 t_k1b_n = 0;
    double t_k3_ms = 0; u64 t_k3_n = 0;
    double t_count_ms = 0; u64 t_count_n = 0;
    double t_gather_ms = 0; u64 t_gather_n = 0;
};

// key_counts (one returning atomic per tile) and counters (three atomics per tile) sit on different 256-B segments
enum { SM_KEYCOUNT = 0, SM_COUNTERS = 32, SM_NNZ = 64, SM_NROWS_U = 65, SM_N = 66, SM_RUNNING = 96, SM_DRAWBASE = 128, SM_WORDS = 160 };

static int set_scatter_lds_limit();
static u32 g_cu_count = 256;
// multi-device engine (multi_engine.hpp, included at the end of this file)
static int multi_create(const fastf_engine_config_t* cfg, fastf_engine* e);
static void multi_destroy(fastf_engine* e);
static int multi_push(fastf_engine* e, const fastf_batch_t* b, bool pinned);
static int multi_finish(fastf_engine* e, fastf_coo_t* coo, uint64_t counters[3]);
static int multi_umi_rows(fastf_engine* e, fastf_umi_rows_t* rows);
static int multi_reset(fastf_engine* e, bool reseed, u32 seed, u64 skip);
static int multi_device_records(const fastf_engine* e, uint64_t* records, u32 n);
static size_t scatter_smem_bytes(u32 ipt = SORT_IPT) {        // LDS follows the tile size: smaller tiles → more workgroups per CU
    return (size_t)ipt * SORT_THREADS * 8 + (size_t)SORT_WAVES * RADIX * 4 + RADIX * 4 * 2 + 64;     // 64: s_wtot[8] + slack
}

// ------------------------------------------------------------------------------------
// table build (host) + upload
// ------------------------------------------------------------------------------------
static u32 h_slot_hash(u64 key) {          // host mirror of slot_hash() in umi_kernels.hpp
    const u32 h = (u32)key * 0x9E3779B1u + (u32)(key >> 32) * 0x85EBCA77u;
    return h ^ (h >> 15);
}

static int build_table(const u64* keys, u32 n, DevBuf& buf, Table& t, const char* what) {
    u32 cap = 64;
    while (cap < 2ull * n) cap <<= 1;
    std::vector<uint4> slots(cap, make_uint4(0, 0, 0, 0));
    for (u32 i = 0; i < n; ++i) {
        const u64 k = keys[i];
        if (k == 0) return set_err("%s key %u is 0 (unpackable string?)", what, i);
        u32 h = h_slot_hash(k) & (cap - 1);
        for (;;) {
            uint4& s = slots[h];
            const u64 sk = ((u64)s.y << 32) | s.x;
            if (sk == 0) { s.x = (u32)k; s.y = (u32)(k >> 32); s.z = i + 1; break; }
            if (sk == k) return set_err("duplicate %s key at index %u (first occurrence wins in the reference; dedup before create)", what, i);
            h = (h + 1) & (cap - 1);
        }
    }
    if (buf.ensure((size_t)cap * sizeof(uint4))) return 1;
    if (copy_h2d(buf.p, slots.data(), (size_t)cap * sizeof(uint4))) return 1;
    t.slots = (const uint4*)buf.p;
    t.mask = cap - 1;
    return 0;
}

// LDS images (see umi_kernels.hpp "LDS-resident tables"); both return 0 and leave use_* false when not eligible
static u32 h_filter_bit(u64 key) {          // host mirror of filter_bit() in umi_kernels.hpp
    const u32 h = (u32)key * 0x85EBCA77u + (u32)(key >> 32) * 0xC2B2AE3Du;
    return h ^ (h >> 13);
}

// bit set over the listed barcode keys, about 10 bits per key (a tenth of the unlisted keys still reach the table), 4-32 KB
static int build_cell_filter(fastf_engine* e, const u64* keys, u32 n) {
    if (n == 0 || getenv("FASTF_NO_CELL_FILTER")) return 0;
    u32 bits = 1u << 15;
    while (bits < 10ull * n && bits < (1u << 18)) bits <<= 1;
    std::vector<u32> w(bits / 32, 0u);
    for (u32 i = 0; i < n; ++i) { const u32 b = h_filter_bit(keys[i]) & (bits - 1); w[b >> 5] |= 1u << (b & 31); }
    if (e->d_cell_filter.ensure(bits / 8)) return 1;
    if (copy_h2d(e->d_cell_filter.p, w.data(), bits / 8)) return 1;
    e->cell_filter.bits = (const u32*)e->d_cell_filter.p; e->cell_filter.mask = bits - 1;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_cells_filtered_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(bits / 8)) != hipSuccess) { e->cell_filter.bits = nullptr; }
    return 0;
}

// hash-and-displace with quotienting over the 32-bit barcode codes (see "LDS-resident tables" in umi_kernels.hpp):
// buckets by the low bits of the mixed code, largest first; a bucket's displacement is any value that puts all of its
// keys on free slots (a single key goes straight to a free slot).  Two keys of one bucket with the same hi part can
// never be placed — about one such pair is expected at 25 k keys, so the mix is re-seeded until there is none.
struct CellImage { std::vector<unsigned char> img; u32 slot_bits = 0, bucket_mask = 0, family = 0, bytes = 0, seed = 0; };
// pure host code (no HIP): true when the list qualifies and a displacement set was found
static bool make_cell_image(const u64* keys, u32 n, CellImage& out) {
    u32 max_n = 32767;
    { const char* mx = getenv("FASTF_LDS_CELLS_MAX"); if (mx) max_n = std::min<u32>(max_n, (u32)atoi(mx)); }
    if (n == 0 || n > max_n) return false;
    const u64 fam = keys[0] >> 49;
    std::vector<u32> code(n);
    for (u32 i = 0; i < n; ++i) {
        const u64 k = keys[i];
        if ((k >> 62) != 1 || (k >> 49) != fam || (k & 0xFFFFu) != 0 || ((k >> 57) & 31) > 16) return false;
        code[i] = (u32)(k >> 16);
    }
    u32 S = 10;
    while ((1u << S) <= n) ++S;                                          // n <= 2^S - 1: index 0 marks an empty slot
    const u32 slots = 1u << S, smask = slots - 1u, lo_mask = (1u << (32u - S)) - 1u;
    u32 B = 256;
    while (B * 3u < n && B < 8192u) B <<= 1;                            // about 2-4 keys per bucket; 16 KB of displacements at most
    const size_t bytes = (size_t)slots * 4 + (size_t)B * 2;
    std::vector<u32> h(n), order(B), head(B + 1), items(n), free_list; std::vector<u32> slot(slots);
    std::vector<unsigned short> disp(B);
    for (u32 attempt = 0; attempt < 64; ++attempt) {
        const u32 seed = attempt * 0x632BE5ABu;
        std::fill(head.begin(), head.end(), 0u);
        for (u32 i = 0; i < n; ++i) { h[i] = cell_mix(code[i], seed); head[(h[i] & lo_mask & (B - 1u)) + 1]++; }
        for (u32 b = 0; b < B; ++b) head[b + 1] += head[b];
        { std::vector<u32> fill(head.begin(), head.end() - 1); for (u32 i = 0; i < n; ++i) items[fill[h[i] & lo_mask & (B - 1u)]++] = i; }
        for (u32 b = 0; b < B; ++b) order[b] = b;
        std::sort(order.begin(), order.end(), [&](u32 x, u32 y) { const u32 sx = head[x + 1] - head[x], sy = head[y + 1] - head[y]; return sx != sy ? sx > sy : x < y; });
        std::fill(slot.begin(), slot.end(), 0u); std::fill(disp.begin(), disp.end(), (unsigned short)0);
        bool ok = true; u32 placed_keys = 0, rover = 0;
        for (u32 oi = 0; oi < B && ok; ++oi) {
            const u32 b = order[oi], lo_i = head[b], sz = head[b + 1] - lo_i;
            if (sz == 0) break;
            if (sz == 1) {                                               // any free slot will do: walk a rover over the table
                while (slot[rover]) rover = (rover + 1) & smask;
                const u32 it = items[lo_i], hi = h[it] >> (32u - S);
                disp[b] = (unsigned short)((rover - hi) & smask);
                slot[rover] = ((h[it] & lo_mask) << S) | (it + 1);
                placed_keys++;
                continue;
            }
            for (u32 x = lo_i; x < lo_i + sz && ok; ++x)                 // equal hi parts inside a bucket collide for every displacement
                for (u32 y = x + 1; y < lo_i + sz; ++y)
                    if ((h[items[x]] >> (32u - S)) == (h[items[y]] >> (32u - S))) { ok = false; break; }
            if (!ok) break;
            bool placed = false;
            u32 d = (b * 0x9E3779B1u) >> (32u - S);                      // spread the starting points
            for (u32 t = 0; t < slots && !placed; ++t, d = (d + 1) & smask) {
                bool fits = true;
                for (u32 x = lo_i; x < lo_i + sz; ++x) if (slot[((h[items[x]] >> (32u - S)) + d) & smask]) { fits = false; break; }
                if (!fits) continue;
                for (u32 x = lo_i; x < lo_i + sz; ++x) { const u32 it = items[x]; slot[((h[it] >> (32u - S)) + d) & smask] = ((h[it] & lo_mask) << S) | (it + 1); }
                disp[b] = (unsigned short)d; placed = true; placed_keys += sz;
            }
            if (!placed) ok = false;
        }
        if (!ok || placed_keys != n) continue;
        out.img.assign(((bytes + 15) & ~(size_t)15) + 16, 0);
        memcpy(out.img.data(), slot.data(), (size_t)slots * 4);
        memcpy(out.img.data() + (size_t)slots * 4, disp.data(), (size_t)B * 2);
        out.slot_bits = S; out.bucket_mask = B - 1u; out.family = (u32)fam; out.bytes = (u32)bytes; out.seed = seed;
        return true;
    }
    return false;                                                        // no displacement set found: stay on the L2 table
}

static int build_cell_lds(fastf_engine* e, const u64* keys, u32 n) {
    CellImage ci;
    if (!make_cell_image(keys, n, ci)) return 0;
    if (e->img_cells.ensure(ci.img.size())) return 1;
    if (copy_h2d(e->img_cells.p, ci.img.data(), ci.img.size())) return 1;
    e->lds_cells.image = (const u32*)e->img_cells.p; e->lds_cells.slot_bits = ci.slot_bits; e->lds_cells.bucket_mask = ci.bucket_mask;
    e->lds_cells.family = ci.family; e->lds_cells.bytes = ci.bytes; e->lds_cells.seed = ci.seed;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(probe_cells_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ci.bytes) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe_cells_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ci.bytes) != hipSuccess) return 0;
    e->cells_blocks_per_cu = ci.bytes + 64 <= 80 * 1024 ? 2u : 1u;
    e->use_lds_cells = true;
    return 0;
}

// test hook (host only, no device needed): the LDS image of a barcode list and its parameters; returns the image size
// in bytes (0: the list does not qualify), copies at most cap bytes
extern "C" size_t fastf_debug_cell_image(const uint64_t* cell_keys, uint32_t n, void* image_out, size_t cap, uint32_t params[5]) FASTF_TRY {
    CellImage ci;
    if (!make_cell_image((const u64*)cell_keys, n, ci)) return 0;
    if (image_out) memcpy(image_out, ci.img.data(), std::min(cap, (size_t)ci.bytes));
    if (params) { params[0] = ci.slot_bits; params[1] = ci.bucket_mask; params[2] = ci.family; params[3] = ci.bytes; params[4] = ci.seed; }
    return ci.bytes;
} FASTF_CATCH_ZERO

static int build_gene_lds(fastf_engine* e, const u64* keys, u32 n) {
    if (n == 0 || n > 65535) return 0;
    // most common ID-form family
    std::vector<u64> fams;
    for (u32 i = 0; i < n; ++i) if ((keys[i] >> 62) == 2) fams.push_back(keys[i] >> 44);
    if (fams.size() * 2 < n) return 0;
    std::sort(fams.begin(), fams.end());
    u64 best = fams[0]; size_t best_n = 0;
    for (size_t i = 0; i < fams.size();) { size_t j = i; while (j < fams.size() && fams[j] == fams[i]) ++j; if (j - i > best_n) { best_n = j - i; best = fams[i]; } i = j; }
    if (best_n * 2 < n) return 0;
    const u64 VM = 0xFFFFFFFFFFFull;
    u64 vmin = ~0ull, vmax = 0;
    for (u32 i = 0; i < n; ++i) if ((keys[i] >> 44) == best) { const u64 v = keys[i] & VM; vmin = std::min(vmin, v); vmax = std::max(vmax, v); }
    const u64 range = vmax - vmin + 1;
    if (range > (1ull << 22)) return 0;
    const u32 words = (u32)((range + 31) / 32), rank_pad = (words + 1u) & ~1u;
    size_t bytes = (((size_t)words * 4 + (size_t)rank_pad * 2 + best_n * 2) + 15) & ~(size_t)15;
    // a dense id range is cheaper as a direct table (one LDS read per lookup instead of bitmap + rank + permutation)
    const size_t direct_bytes = ((size_t)range * 2 + 15) & ~(size_t)15;
    const bool direct = direct_bytes <= bytes + bytes / 8 && !getenv("FASTF_GENES_NO_DIRECT");
    if (direct) bytes = direct_bytes;
    {   // FASTF_GENES_LDS_MAX_KB: largest LDS image (default 152 KB: one workgroup per CU; a real human Ensembl list needs ~128 KB)
        const char* mk = getenv("FASTF_GENES_LDS_MAX_KB");
        if (bytes > (size_t)(mk ? atoi(mk) : 152) * 1024) return 0;
    }
    std::vector<u32> img(bytes / 4 + 4, 0u);
    if (direct) {
        unsigned short* tab = reinterpret_cast<unsigned short*>(img.data());
        for (u32 i = 0; i < n; ++i) if ((keys[i] >> 44) == best) tab[(keys[i] & VM) - vmin] = (unsigned short)(i + 1);
    }
    u32* bitmap = img.data();
    unsigned short* rank = reinterpret_cast<unsigned short*>(img.data() + words);
    unsigned short* perm = rank + rank_pad;
    if (!direct) {
        std::vector<std::pair<u64, u32>> vals; vals.reserve(best_n);
        for (u32 i = 0; i < n; ++i) if ((keys[i] >> 44) == best) vals.push_back({(keys[i] & VM) - vmin, i + 1});
        std::sort(vals.begin(), vals.end());
        for (size_t r = 0; r < vals.size(); ++r) { bitmap[vals[r].first >> 5] |= 1u << (vals[r].first & 31); perm[r] = (unsigned short)vals[r].second; }
        u32 acc = 0;
        for (u32 w = 0; w < words; ++w) { rank[w] = (unsigned short)acc; acc += (u32)__builtin_popcount(bitmap[w]); }
    }
    if (e->img_genes.ensure(bytes)) return 1;
    if (copy_h2d(e->img_genes.p, img.data(), bytes)) return 1;
    e->lds_genes.image = (const u32*)e->img_genes.p; e->lds_genes.words = words; e->lds_genes.n_perm = (u32)best_n;
    e->lds_genes.family = (u32)best; e->lds_genes.vmin = vmin; e->lds_genes.range = range; e->lds_genes.bytes = (u32)bytes; e->lds_genes.direct = direct ? 1u : 0u;
    {
        const void* fns[] = {(const void*)filter_pack_kernel<true, false>, (const void*)filter_pack_kernel<true, true>,
                             (const void*)filter_pack_stream_kernel<false, false, false>, (const void*)filter_pack_stream_kernel<false, false, true>,
                             (const void*)filter_pack_stream_kernel<false, true, false>, (const void*)filter_pack_stream_kernel<false, true, true>,
                             (const void*)filter_pack_stream_kernel<true, false, false>, (const void*)filter_pack_stream_kernel<true, false, true>,
                             (const void*)filter_pack_stream_kernel<true, true, false>, (const void*)filter_pack_stream_kernel<true, true, true>,
                             (const void*)filter_pack_stream_kernel<false, false, false, true>, (const void*)filter_pack_stream_kernel<false, false, true, true>,
                             (const void*)filter_pack_stream_kernel<false, true, false, true>, (const void*)filter_pack_stream_kernel<false, true, true, true>,
                             (const void*)filter_pack_stream_kernel<true, false, false, true>, (const void*)filter_pack_stream_kernel<true, false, true, true>,
                             (const void*)filter_pack_stream_kernel<true, true, false, true>, (const void*)filter_pack_stream_kernel<true, true, true, true>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return 0;
    }
    const size_t per_block = bytes + 1024;                           // + the kernel's static LDS (about 0.5 KB)
    e->genes_blocks_per_cu = (u32)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / per_block));
    e->use_lds_genes = true;
    return 0;
}

// ------------------------------------------------------------------------------------
// create / destroy
// ------------------------------------------------------------------------------------
extern "C" int fastf_engine_create(const fastf_engine_config_t* cfg, fastf_engine_t** out) FASTF_TRY {
    if (!cfg || !out) return set_err("null argument");
    *out = nullptr;
    const char* pf_ = getenv("FASTF_PROFILE");
    const bool lap_on = pf_ && pf_[0] == '2';
    const auto t_0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (lap_on) fprintf(stderr, "[engine create] %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count()); };
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    lap("device count");
    if (he != hipSuccess || ndev == 0)
        return set_err("no HIP device available (%s): the engine has no CPU fallback", hipGetErrorString(he));
    if (cfg->umi_max_bases < 1 || cfg->umi_max_bases > 32) return set_err("umi_max_bases must be 1..32");
    if (cfg->n_devices > 1 || (cfg->n_devices == 1 && getenv("FASTF_FORCE_MULTI"))) {
        fastf_engine* me = new fastf_engine();
        if (multi_create(cfg, me)) { multi_destroy(me); delete me; return 1; }
        *out = me;
        return 0;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return set_err("device %d out of range (have %d)", cfg->device, ndev);
    if (cfg->n_shards < 1 || cfg->n_shards > 8 || cfg->shard_rank >= cfg->n_shards) return set_err("bad shard config");
    HIP_OK(hipSetDevice(cfg->device));
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, cfg->device) == hipSuccess && pr.multiProcessorCount > 0) g_cu_count = (u32)pr.multiProcessorCount; }
    lap("device set, properties");

    fastf_engine* e = new fastf_engine();
    e->device = cfg->device;
    e->n_cells = cfg->n_cells; e->n_features = cfg->n_features;
    e->cell16 = cfg->n_cells <= 65535u && !getenv("FASTF_CELL_SCRATCH_32");
    e->cell_bits = bits_for(cfg->n_cells);
    e->feat_bits = bits_for(cfg->n_features);
    e->L.umi_bits = 2 * cfg->umi_max_bases;
    e->L.umi_max_bytes = (cfg->umi_max_bases + 3) / 4;
    e->L.len_bits = bits_for(e->L.umi_max_bytes);
    if (cfg->umi_max_bases > 24) {                       // 25..32 bases: one layout, the value = flag | low 51 bits of [bases: 64][blob bytes: 4]
        e->sub_bits = WIDE_SUB_BITS; e->L.umi_max_bytes = 8; e->L.len_bits = WIDE_SUB_LEN_BITS; e->L.umi_bits = WIDE_SUB_VAL_BITS - WIDE_SUB_LEN_BITS;
    }
    e->L.feat_shift = 1 + e->L.umi_bits + e->L.len_bits;
    e->L.cell_shift = e->L.feat_shift + e->feat_bits;
    e->L.total_bits = e->L.cell_shift + e->cell_bits;
    e->wide = e->L.total_bits > 64 || cfg->umi_max_bases > 16 || getenv("FASTF_FORCE_WIDE_KEYS") != nullptr;
    e->long_umi = cfg->umi_max_bases > 16;               // batches carry bases 17.. in fastf_batch_t.umi_ext
    e->group_bits = e->cell_bits + e->feat_bits + e->sub_bits;
    e->row_split = e->sub_bits ? std::min<u32>(32u, e->feat_bits + e->sub_bits) : e->feat_bits;
    if (e->group_bits > 64) {
        const u32 gb = e->group_bits; delete e;
        return set_err("umi_max_bases > 24 with %u bits of (cell, feature): the sorted word would need %u bits (> 64)", gb - WIDE_SUB_BITS, gb);
    }
    e->threshold = cfg->draw_threshold;
    {   // Matrix path: only (cell, feature) is sorted — every bit below feat_shift may stay unsorted, K3 (reduce_hashed_kernel)
        // finds the distinct UMIs of a group through its window set.  The digit grid is anchored at the TOP of the key, so
        // the number of 8-bit passes is the minimum for that range whatever the key width (a byte-aligned grid spends a whole
        // pass on the one or two top bits of a 57/58-bit key); everything below the grid stays unsorted.
        const char* sk = getenv("FASTF_SORT_SKIP_BITS");
        const u32 fs = e->L.feat_shift, kb = e->L.total_bits;
        const u32 passes = (kb - fs + 7) / 8;
        e->skip_bits = sk ? (u32)atoi(sk) : (kb > 8 * passes ? kb - 8 * passes : 0);
        if (e->skip_bits >= fs) e->skip_bits = 0;
        if (e->wide) e->skip_bits = 0;                               // the group word is sorted whole
    }
    e->n_shards = cfg->n_shards; e->shard_rank = cfg->shard_rank;
    e->mt_seed0 = cfg->mt_seed; e->mt_skip0 = cfg->mt_skip;

    int rc = 0;
    do {
        // (FASTF_STREAM_PRIORITY=1: the engine's streams at the device's highest priority, so that a push's K1 kernels get the next free
        //  CU in front of the reader's queued inflate workgroups.  Measured twice on the 80 M-record file: 1022 against 933 blocks/ms
        //  in one session, 1007 against 1018 in the next, the device side ready 0.1 s later — a knob, off:
        //  profiles/r5_notes/window_period_at_high_share.txt)
        int prio_lo = 0, prio_hi = 0;
        const char* sp = getenv("FASTF_STREAM_PRIORITY");
        const bool high = sp && sp[0] == '1' && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_hi != prio_lo;
        auto mk = [&](hipStream_t* s) { return high ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(s, hipStreamNonBlocking); };
        if (mk(&e->s_compute) != hipSuccess ||
            mk(&e->s_copy) != hipSuccess ||
            mk(&e->s_mt) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_mt, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->slot[0].ev_copy, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->slot[1].ev_copy, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->slot[0].ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->slot[1].ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreate(&e->t_ev[0]) != hipSuccess || hipEventCreate(&e->t_ev[1]) != hipSuccess) {
            rc = set_err("stream/event creation failed"); break;
        }
        lap("streams and events");
        if ((rc = build_table((const u64*)cfg->cell_keys, cfg->n_cells, e->tab_cells, e->cells, "cell"))) break;
        if ((rc = build_table((const u64*)cfg->feature_keys, cfg->n_features, e->tab_feats, e->feats, "feature"))) break;
        {
            const char* lt = getenv("FASTF_LDS_TABLES");             // "0" keeps both lookups on the L2 tables
            if (!(lt && lt[0] == '0')) {
                const char* lc = getenv("FASTF_LDS_CELLS");
                if (!(lc && lc[0] == '0') && (rc = build_cell_lds(e, (const u64*)cfg->cell_keys, cfg->n_cells))) break;
                if (!e->use_lds_cells && (rc = build_cell_filter(e, (const u64*)cfg->cell_keys, cfg->n_cells))) break;
                if ((rc = build_gene_lds(e, (const u64*)cfg->feature_keys, cfg->n_features))) break;
            }
        }
        lap("tables built and uploaded");
        if ((rc = e->d_small.ensure(SM_WORDS * sizeof(u64)))) break;
        if (hipHostMalloc((void**)&e->h_small, SM_WORDS * sizeof(u64), hipHostMallocDefault) != hipSuccess) {
            rc = set_err("hipHostMalloc failed"); break;
        }
        if (hipMemset(e->d_small.p, 0, SM_WORDS * sizeof(u64)) != hipSuccess) { rc = set_err("memset failed"); break; }
        lap("counters allocated and cleared");
        if (set_scatter_lds_limit()) {
            rc = set_err("cannot raise dynamic LDS limit to %zu bytes", scatter_smem_bytes()); break;
        }
        lap("kernel attributes (code object loaded)");
        e->batch_cap = cfg->batch_records ? cfg->batch_records : (4ull << 20);
        e->key_cap = cfg->key_capacity;
        // the push path stages blocked and runs the streaming K1b whenever the engine can (FASTF_PUSH_TILE_FORM=1: SoA staging and
        // the tile form, as lists that keep the gene table in L2, wide keys and sharded engines use)
        e->stream_mode = e->use_lds_genes && !e->wide && e->n_shards == 1 && !getenv("FASTF_NO_STREAM_K1B") && !getenv("FASTF_PUSH_TILE_FORM");
    } while (0);
    if (rc) { fastf_engine_destroy(e); return 1; }
    *out = e;
    return 0;
} FASTF_CATCH_INT

extern "C" void fastf_engine_destroy(fastf_engine_t* e) FASTF_TRY {
    if (!e) return;
    if (e->multi) { multi_destroy(e); delete e; return; }
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 2; ++i) {
        fastf_engine::Slot& sl = e->slot[i];
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        if (sl.h_draws) (void)hipHostFree(sl.h_draws);
        if (sl.h_small) (void)hipHostFree(sl.h_small);
        sl.d_stage.release();
        if (sl.ev_copy) (void)hipEventDestroy(sl.ev_copy);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (e->t_ev[i]) (void)hipEventDestroy(e->t_ev[i]);
    }
    if (e->h_small) (void)hipHostFree(e->h_small);
    if (e->h_coo) { if (e->h_coo_pinned) pin_unreg(e->h_coo); fastf_big_free(e->h_coo, (size_t)e->h_coo_cap * 12); }
    e->d_ring.release(); e->d_mt.release(); e->d_dbits.release(); e->d_mtwords.release(); e->d_mtsub.release(); e->d_mtpoly.release(); e->d_mtseat.release(); e->d_mtseq.release();
    DevBuf* all[] = {&e->tab_cells, &e->tab_feats, &e->img_cells, &e->img_genes, &e->d_cell_filter, &e->d_keys, &e->d_tmp, &e->d_small, &e->d_feature, &e->d_cell,
                     &e->d_count, &e->d_ukeys, &e->d_ncopy, &e->d_cellidx, &e->d_tilecnt, &e->d_tilebase, &e->d_binbase, &e->d_cnt, &e->d_rg_feature, &e->d_rg_cell, &e->d_rg_count, &e->d_rg_ukeys, &e->d_spanrows, &e->d_spanbase, &e->d_giant, &e->d_scanblk,
                     &e->d_halfhits, &e->d_segcount, &e->d_segprefix, &e->d_tileseg, &e->d_segkeys, &e->d_vals, &e->d_vtmp,
                     &e->d_rgn_phys, &e->d_rgn_hist, &e->d_rgn_blk};
    for (DevBuf* b : all) b->release();
    if (e->s_compute) (void)hipStreamDestroy(e->s_compute);
    if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
    if (e->s_mt) (void)hipStreamDestroy(e->s_mt);
    if (e->ev_mt) (void)hipEventDestroy(e->ev_mt);
    delete e;
} FASTF_CATCH_VOID

extern "C" int fastf_engine_table_modes(const fastf_engine_t* e, int* cells_in_lds, int* genes_in_lds) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (cells_in_lds) *cells_in_lds = e->use_lds_cells;
    if (genes_in_lds) *genes_in_lds = e->use_lds_genes ? (e->lds_genes.direct ? 2 : 1) : 0;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_engine_cell_scratch_bytes(const fastf_engine_t* e, uint32_t* bytes) FASTF_TRY {
    if (!e || !bytes) return set_err("null argument");
    *bytes = (e->multi ? e->n_cells <= 65535u : e->cell16) ? 2u : 4u;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_engine_sort_passes(const fastf_engine_t* e, uint32_t flags, uint32_t* passes) FASTF_TRY {
    if (!e || !passes) return set_err("null argument");
    *passes = sort_passes(e->L.total_bits, (flags & FASTF_SORT_SKIP_LOW) ? e->skip_bits : 0);
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_engine_skip_bits(const fastf_engine_t* e, uint32_t* bits) FASTF_TRY {
    if (!e || !bits) return set_err("null argument");
    *bits = e->skip_bits;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_engine_key_bits(const fastf_engine_t* e, uint32_t* cell_bits, uint32_t* feature_bits,
                                     uint32_t* umi_bits, uint32_t* total_bits) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (cell_bits) *cell_bits = e->cell_bits;
    if (feature_bits) *feature_bits = e->feat_bits;
    if (umi_bits) *umi_bits = 1 + e->L.umi_bits + e->L.len_bits;
    if (total_bits) *total_bits = e->L.total_bits;
    return 0;
} FASTF_CATCH_INT

// ------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------
static u64 max_tiles_for(u64 n, u64 tile) { return (n + tile - 1) / tile + 1; }

// keys per thread of the sort tiles (tile = ipt x 512 keys).  At most 7, so that the scatter kernel's LDS
// (tile x 8 B + 10 KB) lets four workgroups share a CU (8 waves per SIMD), and the smallest value that needs no more
// rounds over those 4 x CU slots than ipt = 7 would, so every round is full.  Measured at 10 M keys with the
// XCD-contiguous tile mapping: 34.4-35.5 us per scatter pass for ipt 6-7, 37 us for 9-10 (three workgroups per CU).
static u32 choose_sort_ipt(u64 n) {
    const char* f = getenv("FASTF_SORT_IPT");
    if (f) { int v = atoi(f); if (v >= 1 && v <= SORT_IPT) return (u32)v; }
    const u64 max_ipt = std::min<u64>(7, SORT_IPT), slots = 4ull * g_cu_count;
    const u64 per_round_max = slots * SORT_THREADS * max_ipt;
    const u64 rounds = std::max<u64>(1, (n + per_round_max - 1) / per_round_max);
    const u64 per_thread = (n + rounds * slots * SORT_THREADS - 1) / (rounds * slots * SORT_THREADS);
    return (u32)std::min<u64>(max_ipt, std::max<u64>(1, per_thread));
}

// max_records: records whose cell indices go to the engine's own scratch array; tile_records (>= max_records): records whose
// K1 tile counts are kept (a blocked pass keeps its cell indices in the caller's buffer)
static int reserve_workspace(fastf_engine* e, u64 max_records, u64 max_keys, u64 tile_records = 0) {
    const u64 t1 = max_tiles_for(std::max(max_records, tile_records), K1_TILE);
    const u64 ts = max_tiles_for(max_keys, (u64)choose_sort_ipt(max_keys) * SORT_THREADS);
    if (max_records && e->d_cellidx.ensure(max_records * sizeof(u32))) return 1;
    if (e->d_tilecnt.bytes < t1 * sizeof(u32)) {                     // (re)allocated: establish the all-zero invariant
        if (e->d_tilecnt.ensure(t1 * sizeof(u32))) return 1;
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemset(e->d_tilecnt.p, 0, e->d_tilecnt.bytes));
    }
    if (e->d_tilebase.ensure(t1 * sizeof(u64))) return 1;
    if (e->d_halfhits.ensure(t1 * 16 * sizeof(u32))) return 1;
    if (e->d_tileseg.ensure((ts + 8) * sizeof(TileSeg))) return 1;
    if (e->d_binbase.ensure(RADIX * sizeof(u32))) return 1;
    if (e->d_cnt.ensure((ts + 4) * RADIX * sizeof(u32))) return 1;     // rows padded to a multiple of 4 tiles
    return 0;
}

extern "C" int fastf_dev_reserve(fastf_engine_t* e, uint64_t max_records, uint64_t max_keys) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_reserve: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    return reserve_workspace(e, max_records, max_keys);
} FASTF_CATCH_INT

// FASTF_DEBUG_SYNC=1: wait after every stage and name it on stderr — when a kernel faults, the last line says which stage
// had completed (diagnostic runs only: the waits serialise everything)
static bool dbg_sync_on() { static int v = -1; if (v < 0) v = getenv("FASTF_DEBUG_SYNC") != nullptr; return v != 0; }
static void dbg_sync(hipStream_t s, const char* what) {
    if (!dbg_sync_on()) return;
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipGetLastError();
    fprintf(stderr, "[sync] %s: %s %s\n", what, hipGetErrorString(e1), e2 == hipSuccess ? "" : hipGetErrorString(e2));
    fflush(stderr);
}

// per-kernel timing helpers (only active in timing mode; adds event records to the stream)
static void t_begin(fastf_engine* e, hipStream_t s) { if (e->timing) (void)hipEventRecord(e->t_ev[0], s); }
static void t_end(fastf_engine* e, hipStream_t s, double* acc, u64* cnt) {
    if (!e->timing) return;
    (void)hipEventRecord(e->t_ev[1], s);
    (void)hipEventSynchronize(e->t_ev[1]);
    float ms = 0;
    if (hipEventElapsedTime(&ms, e->t_ev[0], e->t_ev[1]) == hipSuccess) { *acc += ms; *cnt += 1; }
}

extern "C" int fastf_engine_set_timing(fastf_engine_t* e, int on) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_engine_set_timing: device-level calls take a single-device engine");
    e->timing = on != 0;
    e->t_scatter_ms = e->t_k1_ms = e->t_k1b_ms = e->t_k3_ms = e->t_count_ms = e->t_gather_ms = 0;
    e->t_scatter_n = e->t_k1_n = e->t_k1b_n = e->t_k3_n = e->t_count_n = e->t_gather_n = 0;
    return 0;
} FASTF_CATCH_INT
// which: 0 = K1a probe_cells, 1 = K2 scatter, 2 = K3 (reduce_windows + span_scan), 3 = K2 tile_count,
//        4 = K1b filter_pack, 5 = rows_gather (concatenation of K3's row regions)
extern "C" int fastf_engine_get_timing(fastf_engine_t* e, int which, double* total_ms, uint64_t* launches) FASTF_TRY {
    if (!e) return set_err("null engine");
    const double ms[6] = {e->t_k1_ms, e->t_scatter_ms, e->t_k3_ms, e->t_count_ms, e->t_k1b_ms, e->t_gather_ms};
    const u64 n[6] = {e->t_k1_n, e->t_scatter_n, e->t_k3_n, e->t_count_n, e->t_k1b_n, e->t_gather_n};
    if (which < 0 || which > 5) return set_err("bad timer index");
    *total_ms = ms[which]; *launches = n[which];
    return 0;
} FASTF_CATCH_INT

// ------------------------------------------------------------------------------------
// device-level entry points
// ------------------------------------------------------------------------------------
static u64* g_k1_stamps = nullptr;   // diagnostic builds only
extern "C" void fastf_debug_set_k1_stamps(void* p) { g_k1_stamps = (u64*)p; }

// K1a + scan: cell index per record, hit-rank base per tile, total hits
// one workgroup scans the tile counts (see scan_tiles_kernel for the entries-per-thread choice)
// slot: the K1 scan (0) and the K3 scan (1) may run side by side on two streams (pipelined sharded pass): each has its own chunk totals
static void launch_scan_tiles(fastf_engine* e, int slot, hipStream_t s, u32* in, u64* out, u32 T, u64* total_out, u64* running, u64* base_out) {
    constexpr u32 CHUNK = 16u * 1024u;
    const u32 n_blk = (T + CHUNK - 1) / CHUNK;
    if (n_blk <= 1 || n_blk > 64 || e->d_scanblk.ensure(2 * 64 * sizeof(u64))) {
        hipLaunchKernelGGL(scan_tiles_kernel<16>, dim3(1), dim3(1024), 0, s, in, out, T, total_out, running, base_out, (u64*)nullptr);
        return;
    }
    u64* const blk = (u64*)e->d_scanblk.p + 64 * slot;
    hipLaunchKernelGGL(scan_tiles_kernel<16>, dim3(n_blk), dim3(1024), 0, s, in, out, T, (u64*)nullptr, (u64*)nullptr, (u64*)nullptr, blk);
    hipLaunchKernelGGL(scan_fix_kernel<16>, dim3(n_blk), dim3(1024), 0, s, out, T, (const u64*)blk, n_blk, total_out, running, base_out);
}

// blk: the blocked buffer of these records (K1a writes its cell indices into the units' scratch slices), or nullptr
static int launch_probe_cells(fastf_engine* e, const u64* cb, u64 n, u64* d_total_out, hipStream_t s,
                              u64* d_running = nullptr, u64* d_base_out = nullptr, void* blk = nullptr) {
    if (reserve_workspace(e, blk ? 0 : n, 0, n)) return 1;
    const u32 tiles = (u32)((n + K1_TILE - 1) / K1_TILE);
    const CellOut cout = blk ? CellOut{blk, blk_run_bytes(e->cell16)} : CellOut{e->d_cellidx.p, 0u};
    t_begin(e, s);
    if (e->use_lds_cells) {                          // tile counts are all-zero here: scan_tiles_kernel clears what it reads
        const u32 grid = std::min<u32>(e->cells_blocks_per_cu * g_cu_count, (tiles + 1) / 2);
        if (e->cells_blocks_per_cu >= 2)
            hipLaunchKernelGGL(probe_cells_lds_kernel<true>, dim3(grid), dim3(1024), e->lds_cells.bytes, s, cb, n, e->lds_cells,
                               cout, e->cell16, (u32*)e->d_tilecnt.p, (u32*)e->d_halfhits.p, tiles);
        else
            hipLaunchKernelGGL(probe_cells_lds_kernel<false>, dim3(grid), dim3(1024), e->lds_cells.bytes, s, cb, n, e->lds_cells,
                               cout, e->cell16, (u32*)e->d_tilecnt.p, (u32*)e->d_halfhits.p, tiles);
    } else if (e->cell_filter.bits) {
        const u32 grid = std::min<u32>(tiles, 4 * g_cu_count);
        hipLaunchKernelGGL(probe_cells_filtered_kernel, dim3(grid), dim3(K1_THREADS), (e->cell_filter.mask + 1u) / 8u, s, cb, n,
                           e->cells, e->cell_filter, cout, e->cell16, (u32*)e->d_tilecnt.p, (u32*)e->d_halfhits.p, tiles);
    } else {
        hipLaunchKernelGGL(probe_cells_kernel<0>, dim3(tiles), dim3(K1_THREADS), 0, s, cb, n, e->cells,
                           cout, e->cell16, (u32*)e->d_tilecnt.p, (u32*)e->d_halfhits.p);
    }
    t_end(e, s, &e->t_k1_ms, &e->t_k1_n);
    dbg_sync(s, "K1a probe_cells");
    launch_scan_tiles(e, 0, s, (u32*)e->d_tilecnt.p, (u64*)e->d_tilebase.p, tiles, d_total_out, d_running, d_base_out);
    dbg_sync(s, "K1a tile scan");
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int fastf_dev_count_hits(fastf_engine_t* e, const uint64_t* d_cb_key, uint64_t n,
                                    uint64_t* d_hits_out, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_count_hits: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { HIP_OK(hipMemsetAsync(d_hits_out, 0, sizeof(u64), s)); return 0; }
    return launch_probe_cells(e, (const u64*)d_cb_key, n, (u64*)d_hits_out, s);
} FASTF_CATCH_INT

static bool stream_k1b_possible(const fastf_engine* e) {
    return e->use_lds_genes && e->n_shards <= (u32)MAX_SHARDS && !getenv("FASTF_NO_STREAM_K1B");
}

extern "C" int fastf_dev_block_bytes(const fastf_engine_t* e, uint64_t n, uint64_t* bytes) FASTF_TRY {
    if (!e || !bytes) return set_err("null argument");
    *bytes = 0;
    if (e->multi || e->wide || !stream_k1b_possible(e)) return 0;
    *bytes = ((n + BLK_RECS - 1) / BLK_RECS) * (u64)blk_run_bytes(e->cell16);
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_dev_block_records(fastf_engine_t* e, const uint64_t* d_gx_key, const uint32_t* d_umi, const uint32_t* d_meta,
                                       uint64_t n, void* d_blocked, void* stream) FASTF_TRY {
    if (!e || !d_blocked) return set_err("null argument");
    if (e->multi) return set_err("fastf_dev_block_records: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    if (n == 0) return 0;
    const u64 units = (n + BLK_RECS - 1) / BLK_RECS;
    hipLaunchKernelGGL(block_records_kernel, dim3((u32)std::min<u64>((units + 3) / 4, 16ull * g_cu_count)), dim3(256), 0, (hipStream_t)stream,
                       (const u64*)d_gx_key, d_umi, d_meta, (u64)n, (unsigned char*)d_blocked, blk_run_bytes(e->cell16));
    HIP_OK(hipGetLastError());
    dbg_sync((hipStream_t)stream, "block_records");
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_dev_count_hits_blocked(fastf_engine_t* e, const uint64_t* d_cb_key, uint64_t n, void* d_blocked,
                                            uint64_t* d_hits_out, void* stream) FASTF_TRY {
    if (!e || !d_blocked) return set_err("null argument");
    if (e->multi) return set_err("fastf_dev_count_hits_blocked: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { HIP_OK(hipMemsetAsync(d_hits_out, 0, sizeof(u64), s)); return 0; }
    return launch_probe_cells(e, (const u64*)d_cb_key, n, (u64*)d_hits_out, s, nullptr, nullptr, d_blocked);
} FASTF_CATCH_INT

// draws[i] < threshold as bit (first + i) of a ring of decisions (ring_mask = bits - 1; ~0: a linear array that starts at bit 0)
static int launch_draw_bits(u64 threshold, const u32* d_draws, u64 n, u32* d_bits, hipStream_t s, u64 first = 0, u64 ring_mask = ~0ull) {
    if (n == 0) return 0;
    const u32 grid = (u32)std::min<u64>((n + 255) / 256 + 1, 8ull * g_cu_count);
    hipLaunchKernelGGL(draw_bits_kernel, dim3(grid), dim3(256), 0, s, d_draws, n, threshold, d_bits, first, ring_mask);
    HIP_OK(hipGetLastError());
    return 0;
}
// continue the MT19937 stream in d_mt by `count` draws and leave their DECISIONS at the ranks first .. first + count of the ring:
// the generator writes words into `words` (grown to fit), draw_bits_kernel packs them behind it on the same stream
static int launch_mt_decisions(hipStream_t s, u32* d_mt, DevBuf& words, u32* d_ring, u64 first, u64 count, u64 ring_mask, u64 threshold) {
    if (count == 0) return 0;
    if (words.bytes < count * 4) {
        // (the launches that read the old buffer are queued on this stream: let them finish before it is freed)
        if (words.p) HIP_OK(hipStreamSynchronize(s));
        if (words.ensure(std::max<u64>(count * 4, 1u << 20))) return 1;
    }
    hipLaunchKernelGGL(mt_fill_kernel, dim3(1), dim3(256), 0, s, d_mt, (u32*)words.p, 0ull, count, ~0ull);
    HIP_OK(hipGetLastError());
    return launch_draw_bits(threshold, (const u32*)words.p, count, d_ring, s, first, ring_mask);
}
// where a stream that stood at read index idx of its block stands after n more draws
static u32 mt_idx_after(u32 idx, u64 n) {
    if (n == 0) return idx;
    if (idx + n <= MT_N) return (u32)(idx + n);
    return (u32)((idx + n - 1) % MT_N) + 1u;
}
// The same as launch_mt_decisions, by MANY workgroups (mt_seq_kernel + mt_conv_kernel / mt_fill_multi_kernel: jump-ahead).  *idx: the read index of
// the stream in d_mt as the host knows it (the jumps start at a block boundary: the rest of the block the stream stands in is
// handed out by the one-workgroup kernel first); updated.  Calls below MT_PAR_MIN draws take the one-workgroup kernel: seating
// the sub-streams costs a handful of launches.
constexpr u64 MT_PAR_MIN = 4ull * MT_SUB_DRAWS;
static int launch_mt_decisions_par(fastf_engine* e, hipStream_t s, u32* d_mt, u32* idx, DevBuf& words, u32* d_ring, u64 first, u64 count,
                                   u64 ring_mask, u64 threshold) {
    if (count == 0) return 0;
    static const bool no_par = getenv("FASTF_MT_SERIAL") != nullptr;
    const u64 head = *idx < MT_N ? std::min<u64>(count, MT_N - *idx) : 0;          // to the next block boundary
    if (count < MT_PAR_MIN || no_par) {
        if (launch_mt_decisions(s, d_mt, words, d_ring, first, count, ring_mask, threshold)) return 1;
        *idx = mt_idx_after(*idx, count);
        return 0;
    }
    if (words.bytes < count * 4) {
        if (words.p) HIP_OK(hipStreamSynchronize(s));
        if (words.ensure(std::max<u64>(count * 4, 1u << 20))) return 1;
    }
    if (!e->d_mtpoly.p) {                                      // the polynomials: constants of the generator (mt_jump.c), once per engine
        static_assert(MT_POLY_WORDS == FASTF_MT_POLY_WORDS && MT_SUB_DRAWS == FASTF_MT_SUB_DRAWS && MT_JUMP_R == FASTF_MT_JUMP_R, "kernel and table agree");
        if (e->d_mtpoly.ensure((size_t)FASTF_MT_JUMP_POLYS * MT_POLY_WORDS * sizeof(u64))) return 1;
        if (copy_h2d(e->d_mtpoly.p, fastf_mt_jump_table(), (size_t)FASTF_MT_JUMP_POLYS * MT_POLY_WORDS * sizeof(u64))) return 1;
        HIP_OK(hipFuncSetAttribute((const void*)mt_seq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MT_SEQ_WORDS * sizeof(u32))));
        HIP_OK(hipFuncSetAttribute((const void*)mt_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MT_CONV_LDS_WORDS * sizeof(u32))));
        if (e->d_mtseq.ensure((size_t)MT_JUMP_R * MT_SEQ_STRIDE * sizeof(u32))) return 1;      // the sequences of up to R sources
    }
    u32* const w = (u32*)words.p;
    if (head) {                                                // the rest of the block the stream stands in
        hipLaunchKernelGGL(mt_fill_kernel, dim3(1), dim3(256), 0, s, d_mt, w, 0ull, head, ~0ull);
    }
    // rounds of at most R^2 sub-streams: a round's last sub-stream, generated to its end, is where the next round starts
    constexpr u64 ROUND_DRAWS = (u64)MT_JUMP_R * MT_JUMP_R * MT_SUB_DRAWS;
    u64 at = head, last = 0;
    while (at < count) {
        const u64 body = std::min<u64>(count - at, ROUND_DRAWS);
        const u32 S = (u32)((body + MT_SUB_DRAWS - 1) / MT_SUB_DRAWS);
        if (e->d_mtsub.bytes < (size_t)S * MT_STATE_WORDS * 4) {
            if (e->d_mtsub.p) HIP_OK(hipStreamSynchronize(s));
            if (e->d_mtsub.ensure((size_t)std::max<u32>(S, 64) * MT_STATE_WORDS * 4)) return 1;
        }
        u32* const sub = (u32*)e->d_mtsub.p;
        u32* const seq = (u32*)e->d_mtseq.p;
        HIP_OK(hipMemcpyAsync(sub, d_mt, MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, s));
        if (S > 1) HIP_OK(hipMemsetAsync(sub + MT_STATE_WORDS, 0, (size_t)(S - 1) * MT_STATE_WORDS * 4, s));   // the halves of a jump XOR into zeros
        const u32 n_coarse = (S + MT_JUMP_R - 1) / MT_JUMP_R;            // sub-streams 0, R, 2R, ..
        if (n_coarse > 1) {
            hipLaunchKernelGGL(mt_seq_kernel, dim3(1), dim3(256), MT_SEQ_WORDS * sizeof(u32), s, (const u32*)sub, seq, 1u, S);
            hipLaunchKernelGGL(mt_conv_kernel, dim3(2 * (n_coarse - 1)), dim3(512), MT_CONV_LDS_WORDS * sizeof(u32), s, sub, (const u32*)seq, (const u64*)e->d_mtpoly.p, 1u, S);
        }
        if (S > 1) {
            hipLaunchKernelGGL(mt_seq_kernel, dim3(n_coarse), dim3(256), MT_SEQ_WORDS * sizeof(u32), s, (const u32*)sub, seq, MT_JUMP_R, S);
            hipLaunchKernelGGL(mt_conv_kernel, dim3(2 * n_coarse * (MT_JUMP_R - 1)), dim3(512), MT_CONV_LDS_WORDS * sizeof(u32), s, sub, (const u32*)seq, (const u64*)e->d_mtpoly.p, 0u, S);
        }
        hipLaunchKernelGGL(mt_fill_multi_kernel, dim3(S), dim3(256), 0, s, sub, w + at, 0ull, body, ~0ull);
        HIP_OK(hipMemcpyAsync(d_mt, sub + (size_t)(S - 1) * MT_STATE_WORDS, MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, s));
        HIP_OK(hipGetLastError());
        last = body - (u64)(S - 1) * MT_SUB_DRAWS;               // draws of the round's last sub-stream
        at += body;
    }
    *idx = mt_idx_after(MT_N, last);
    return launch_draw_bits(threshold, (const u32*)w, count, d_ring, s, first, ring_mask);
}

// The decisions of `n_draws` draws of the stream init_genrand(seed) + skip, from the device's own generator (the parallel
// one for large counts), as a linear array of bits — what a resident pass hands to fastf_dev_probe_pack with
// FASTF_PROBE_DRAW_BITS (bench.py: the step with its draw generation inside the clock; tests).
extern "C" int fastf_dev_mt_decisions(fastf_engine_t* e, uint32_t seed, uint64_t skip, uint64_t n_draws, uint32_t* d_bits_out, void* stream) FASTF_TRY {
    if (!e || (n_draws && !d_bits_out)) return set_err("null argument");
    if (e->multi) return set_err("fastf_dev_mt_decisions: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    fastf_mt_t mt; fastf_mt_seed(&mt, seed); fastf_mt_skip(&mt, skip);
    // (the seated state goes into the first slot of the sub-stream array's tail: a buffer the engine keeps)
    if (e->d_mtseat.ensure(sizeof mt)) return 1;
    if (copy_h2d_on(e->d_mtseat.p, &mt, sizeof mt, (hipStream_t)stream)) return 1;
    u32 idx = (u32)mt.idx;
    if (launch_mt_decisions_par(e, (hipStream_t)stream, (u32*)e->d_mtseat.p, &idx, e->d_mtwords, d_bits_out, 0, n_draws, ~0ull, e->threshold)) return 1;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return set_err("the generator kernels failed: %s", hipGetErrorString(hipGetLastError()));
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_dev_draw_bits(fastf_engine_t* e, const uint32_t* d_draws, uint64_t n_draws, uint32_t* d_bits_out, void* stream) FASTF_TRY {
    if (!e || (n_draws && (!d_draws || !d_bits_out))) return set_err("null argument");
    if (e->multi) return set_err("fastf_dev_draw_bits: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    return launch_draw_bits(e->threshold, d_draws, n_draws, d_bits_out, (hipStream_t)stream);
} FASTF_CATCH_INT

#define NO_WIDE(e, what) do { if ((e)->wide) return set_err(what ": this engine's keys are wider than 64 bits — this call moves one 64 bits word per key (fastf_dev_probe_pack_wide / fastf_dev_adopt_wide + fastf_engine_finish and the host-buffer API take wide keys)"); } while (0)

// the streaming K1b over `tiles` K1a tiles: workgroups (at most two per CU: 12 or 16 waves each) and the key slots of one
// workgroup's region (every record of the units its waves walk: waves x rounds units)
// (a workgroup fills K1S_SUB regions by turns, one unit each: a region holds every record of the units that go to it; and a
//  wave should find about four units to walk — the loop is pipelined across units — before more workgroups are started: a short
//  chunk of the push path then leaves few, well filled regions instead of many nearly empty ones)
static void stream_geometry(const fastf_engine* e, u64 tiles, u32* grid, u64* region) {
    const u32 waves = (e->genes_blocks_per_cu >= 2 ? K1S_THREADS : K1S_THREADS_ROOMY) / WAVE;
    const u64 units = std::max<u64>(tiles, 1) * (K1_TILE / K1S_UNIT);
    const u64 g = std::max<u64>(1, std::min<u64>((units + 4 * waves - 1) / (4 * waves), (u64)std::min<u32>(e->genes_blocks_per_cu, 2) * g_cu_count));
    const u64 rounds = (units + g * waves - 1) / (g * waves);
    *grid = (u32)g; *region = ((rounds + K1S_SUB - 1) / K1S_SUB) * waves * K1S_UNIT;
}
// regions of several launches behind each other in one key store (the push path): where this launch's begin
struct RegionAppend { u64 slot0; u32 rgn0; };

static int grow_regions(fastf_engine* e, u32 need);
static int launch_probe(fastf_engine* e, const u64* cb, const u64* gx, const u32* umi, const u32* meta, u64 n,
                        const u32* dbits, u64 n_draws, const u64* draw_base, u64* keys, u64 stride, u64* key_counts,
                        u64* counters, bool reuse_hits, hipStream_t s, u64 draw_mask = ~0ull, u64* d_running = nullptr,
                        bool segmented = false, void* blk = nullptr, const RegionAppend* app = nullptr,
                        u64* wide_vals = nullptr, const u32* wide_ext = nullptr) {
    if (segmented && !app) { e->seg_n = 0; e->rgn_hist_n = 0; }
    if (n == 0) return 0;
    // K1a, unless the caller states that fastf_dev_count_hits just ran on these very records (same stream order).
    // d_running: the scan leaves the running hit total of the earlier chunks at draw_base and adds this chunk's hits.
    if (!reuse_hits && launch_probe_cells(e, cb, n, nullptr, s, d_running, d_running ? const_cast<u64*>(draw_base) : nullptr, blk)) return 1;
    const u32 tiles = (u32)((n + K1_TILE - 1) / K1_TILE);
    PackParams p{};
    p.cell = e->d_cellidx.p; p.cell16 = e->cell16; p.gx = gx; p.umi = umi; p.meta = meta; p.n = n;
    p.tile_base = (const u64*)e->d_tilebase.p;
    // (the streaming K1b reads a unit's decision words unconditionally: a stream without a decision still needs a word to read)
    p.dbits = dbits && n_draws ? dbits : (const u32*)e->d_small.p; p.n_draws = dbits ? n_draws : 0; p.draw_base = draw_base; p.draw_mask = draw_mask;
    p.feats = e->feats;
    p.L = e->L;
    p.n_shards = e->n_shards;
    p.keys = keys; p.shard_stride = stride; p.key_counts = key_counts; p.counters = counters;
    p.stamps = g_k1_stamps;
    p.n_tiles = tiles;
    p.genes = e->lds_genes;
    p.blk = (const unsigned char*)blk;
    if (e->wide) {
        // wide keys (tile form): group word into keys[], the rest into vals[] — the engine's own store (one shard), or
        // wide_vals[n_shards][stride] beside keys[n_shards][stride] (the multi-device engine's per-destination buffers)
        if (segmented || blk) return set_err("internal error: wide keys take the tile form");
        if (!wide_vals && (e->n_shards != 1 || keys != (u64*)e->d_keys.p)) return set_err("internal error: wide keys go through the engine's own key store");
        p.vals = wide_vals ? wide_vals : (u64*)e->d_vals.p; p.wide_feat_bits = e->feat_bits; p.wide_sub_bits = e->sub_bits;
        p.umi_ext = wide_vals ? wide_ext : e->cur_umi_ext;
    }
    // several shards: the streaming kernel writes unsharded into a scratch buffer of workgroup regions, shard_partition_kernel
    // deals the keys to the per-destination buffers (same interface as the tile form: keys[G][stride], key_counts[G] += ...)
    const bool no_stream = getenv("FASTF_NO_STREAM_K1B") != nullptr;
    const bool stream_shards = !segmented && !e->wide && e->n_shards > 1 && e->n_shards <= (u32)MAX_SHARDS && e->use_lds_genes && !no_stream;
    if (blk && !(segmented || stream_shards)) return set_err("FASTF_PROBE_BLOCKED needs the streaming K1b (fastf_dev_block_bytes returns 0 otherwise)");
    t_begin(e, s);
    if (segmented || stream_shards) {
        // streaming form: every wave on its own, keys into one private region per workgroup (see filter_pack_stream_kernel)
        u32 grid; u64 region;
        stream_geometry(e, tiles, &grid, &region);
        const u32 n_rgn = grid * (u32)K1S_SUB, rgn0 = app ? app->rgn0 : 0u;
        if (stream_shards) {
            if (e->d_segkeys.ensure((size_t)n_rgn * region * sizeof(u64))) return 1;
            p.keys = (u64*)e->d_segkeys.p;
        } else if ((u64)n_rgn * region > stride) return set_err("segmented key output needs %llu slots, the buffer has %llu (fastf_dev_probe_capacity)",
                                                                (unsigned long long)((u64)n_rgn * region), (unsigned long long)stride);
        if (app) {                                       // the push path has grown the tables (their contents stay)
            if (e->rgn_cap < rgn0 + n_rgn) return set_err("internal error: region tables too small");
        } else {
            if (e->rgn_n) return set_err("device-level K1 on an engine that holds pushed chunks: fastf_engine_reset first");
            if (grow_regions(e, n_rgn)) return 1;
        }
        if (e->d_segprefix.ensure(((size_t)n_rgn + 1) * sizeof(u64))) return 1;
        StreamParams sp{(const u32*)e->d_halfhits.p, region, (u64*)e->d_segcount.p + rgn0,
                        stream_shards ? nullptr : (u32*)e->d_rgn_hist.p + (size_t)rgn0 * RADIX, (u64*)e->d_rgn_phys.p + rgn0,
                        app ? app->slot0 : 0ull, e->skip_bits};
        // compile-time: roomy (one workgroup per CU: 128 VGPRs), width of the cell scratch, form of the gene image
        const int variant = (e->genes_blocks_per_cu >= 2 ? 0 : 4) | (e->cell16 ? 2 : 0) | (e->lds_genes.direct ? 1 : 0);
#define FPS(R, C, D) do { if (blk) hipLaunchKernelGGL((filter_pack_stream_kernel<R, C, D, true>), dim3(grid), dim3(R ? K1S_THREADS_ROOMY : K1S_THREADS), e->lds_genes.bytes, s, p, sp); \
                          else hipLaunchKernelGGL((filter_pack_stream_kernel<R, C, D, false>), dim3(grid), dim3(R ? K1S_THREADS_ROOMY : K1S_THREADS), e->lds_genes.bytes, s, p, sp); } while (0)
        switch (variant) {
        case 0: FPS(false, false, false); break; case 1: FPS(false, false, true); break;
        case 2: FPS(false, true, false); break;  case 3: FPS(false, true, true); break;
        case 4: FPS(true, false, false); break;  case 5: FPS(true, false, true); break;
        case 6: FPS(true, true, false); break;   default: FPS(true, true, true); break;
        }
#undef FPS
        if (stream_shards) {
            hipLaunchKernelGGL(shard_partition_kernel, dim3(n_rgn), dim3(SP_THREADS), 0, s, (const u64*)e->d_segkeys.p, (const u64*)e->d_segcount.p,
                               region, e->L.cell_shift, e->n_shards, keys, stride, key_counts, counters + 3);
        } else if (app) {
            // (the prefix sums are of no use here — the first sort pass walks the regions — the running key count is)
            hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, s, (const u64*)e->d_segcount.p + rgn0, n_rgn, (u64*)e->d_segprefix.p, key_counts, true);
        } else {
            hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, s, (const u64*)e->d_segcount.p, n_rgn, (u64*)e->d_segprefix.p, key_counts, false);
            e->seg_n = n_rgn; e->seg_stride = region;
            e->rgn_hist_n = n_rgn; e->rgn_hist_shift = e->skip_bits;
        }
    } else if (e->use_lds_genes) {
        const u32 grid = std::min<u32>(tiles, e->genes_blocks_per_cu * g_cu_count);
        if (e->genes_blocks_per_cu >= 2) hipLaunchKernelGGL((filter_pack_kernel<true, false>), dim3(grid), dim3(K1B_THREADS), e->lds_genes.bytes, s, p);
        else hipLaunchKernelGGL((filter_pack_kernel<true, true>), dim3(grid), dim3(K1B_THREADS), e->lds_genes.bytes, s, p);
    } else {
        hipLaunchKernelGGL(filter_pack_kernel<false>, dim3(tiles), dim3(K1B_THREADS), 0, s, p);
    }
    HIP_OK(hipGetLastError());
    t_end(e, s, &e->t_k1b_ms, &e->t_k1b_n);
    dbg_sync(s, blk ? "K1b filter_pack (blocked)" : "K1b filter_pack");
    return 0;
}

extern "C" int fastf_dev_probe_pack(fastf_engine_t* e, const uint64_t* d_cb_key, const uint64_t* d_gx_key,
                                    const uint32_t* d_umi, const uint32_t* d_meta, uint64_t n,
                                    const uint32_t* d_draws, uint64_t n_draws, const uint64_t* d_draw_base,
                                    uint64_t* d_keys_out, uint64_t shard_stride, uint64_t* d_key_counts,
                                    uint64_t* d_counters, uint32_t flags, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_probe_pack: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    NO_WIDE(e, "fastf_dev_probe_pack");
    const bool seg = (flags & FASTF_PROBE_SEGMENTED) != 0, blocked = (flags & FASTF_PROBE_BLOCKED) != 0;
    if (seg && !(e->n_shards == 1 && e->use_lds_genes))
        return set_err("FASTF_PROBE_SEGMENTED needs a single shard and the gene table in LDS (fastf_dev_probe_capacity returns 0 otherwise)");
    if (blocked && !d_gx_key) return set_err("FASTF_PROBE_BLOCKED: d_gx_key must point at the blocked buffer");
    if (n && n_draws && !d_draws) return set_err("null d_draws");
    if (!(flags & FASTF_PROBE_DRAW_BITS) && n && n_draws) {
        // 32-bit draws: K1b reads decisions — one pass over the draws on the caller's stream first (a caller that runs the same
        // stream again and again converts it once with fastf_dev_draw_bits and passes FASTF_PROBE_DRAW_BITS)
        if (e->d_dbits.ensure(((n_draws + 63) / 64) * 8)) return 1;
        if (launch_draw_bits(e->threshold, d_draws, n_draws, (u32*)e->d_dbits.p, (hipStream_t)stream)) return 1;
        d_draws = (const uint32_t*)e->d_dbits.p;
    }
    return launch_probe(e, (const u64*)d_cb_key, blocked ? nullptr : (const u64*)d_gx_key, d_umi, d_meta, n, d_draws, n_draws,
                        (const u64*)d_draw_base, (u64*)d_keys_out, shard_stride, (u64*)d_key_counts, (u64*)d_counters,
                        (flags & FASTF_PROBE_REUSE_HITS) != 0, (hipStream_t)stream, ~0ull, nullptr, seg,
                        blocked ? const_cast<uint64_t*>(d_gx_key) : nullptr);
} FASTF_CATCH_INT

extern "C" int fastf_engine_is_wide(const fastf_engine_t* e) { return e && !e->multi && e->wide ? 1 : 0; }

extern "C" int fastf_dev_probe_pack_wide(fastf_engine_t* e, const uint64_t* d_cb_key, const uint64_t* d_gx_key, const uint32_t* d_umi,
                                         const uint32_t* d_meta, const uint32_t* d_umi_ext, uint64_t n,
                                         const uint32_t* d_draws, uint64_t n_draws, const uint64_t* d_draw_base,
                                         uint64_t* d_keys_out, uint64_t* d_vals_out, uint64_t shard_stride, uint64_t* d_key_counts,
                                         uint64_t* d_counters, uint32_t flags, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_probe_pack_wide: device-level calls take a single-device engine");
    if (!e->wide) return set_err("fastf_dev_probe_pack_wide: this engine's keys fit 64 bits (fastf_dev_probe_pack)");
    if (flags & (FASTF_PROBE_SEGMENTED | FASTF_PROBE_BLOCKED)) return set_err("fastf_dev_probe_pack_wide: wide keys take the tile form (no SEGMENTED / BLOCKED)");
    if (!d_keys_out || !d_vals_out) return set_err("null key or value output");
    if (e->long_umi && !d_umi_ext && n) return set_err("this engine was created with umi_max_bases > 16: d_umi_ext (bases 17.. of every UMI, zeros where none) is needed");
    HIP_OK(hipSetDevice(e->device));
    if (n && n_draws && !d_draws) return set_err("null d_draws");
    if (!(flags & FASTF_PROBE_DRAW_BITS) && n && n_draws) {
        if (e->d_dbits.ensure(((n_draws + 63) / 64) * 8)) return 1;
        if (launch_draw_bits(e->threshold, d_draws, n_draws, (u32*)e->d_dbits.p, (hipStream_t)stream)) return 1;
        d_draws = (const uint32_t*)e->d_dbits.p;
    }
    return launch_probe(e, (const u64*)d_cb_key, (const u64*)d_gx_key, d_umi, d_meta, n, d_draws, n_draws, (const u64*)d_draw_base,
                        (u64*)d_keys_out, shard_stride, (u64*)d_key_counts, (u64*)d_counters, (flags & FASTF_PROBE_REUSE_HITS) != 0,
                        (hipStream_t)stream, ~0ull, nullptr, false, nullptr, nullptr, (u64*)d_vals_out, d_umi_ext);
} FASTF_CATCH_INT

extern "C" int fastf_dev_probe_capacity(const fastf_engine_t* e, uint64_t n, uint64_t* key_slots) FASTF_TRY {
    if (!e || !key_slots) return set_err("null argument");
    *key_slots = 0;
    if (e->wide) return 0;
    if (!(e->n_shards == 1 && e->use_lds_genes) || getenv("FASTF_NO_STREAM_K1B")) return 0;
    u32 grid; u64 region;
    stream_geometry(e, (n + K1_TILE - 1) / K1_TILE, &grid, &region);
    *key_slots = (u64)grid * K1S_SUB * region;
    return 0;
} FASTF_CATCH_INT

static u64* g_stamps = nullptr;   // diagnostic builds only (-DFASTF_STAMPS): per-tile phase timestamps of the last scatter
extern "C" void fastf_debug_set_stamps(void* p) { g_stamps = (u64*)p; }
// grid of the tile kernels: whole rounds over the 8 XCDs, capped — beyond the cap the workgroups walk their tiles
static u32 tile_grid(u32 T) {
    static int cap = -1;
    if (cap < 0) { const char* c = getenv("FASTF_TILE_GRID_CAP"); cap = c ? atoi(c) : 32 * (int)g_cu_count; if (cap < 8) cap = 8; }
    return std::min<u32>((T + 7u) & ~7u, (u32)cap & ~7u);
}
static size_t scatter_smem_bytes_vals(u32 ipt) { return scatter_smem_bytes(ipt) + (size_t)ipt * SORT_THREADS * 8; }
static void launch_scatter(u32 shift, u32 T, hipStream_t s, const u64* src, u64* dst, const u64* d_n, const u32* cnt,
                           const u32* bin_tot, u32 ipt, const SegMap seg = SegMap{nullptr, nullptr, 0, 0},
                           const u64* vsrc = nullptr, u64* vdst = nullptr) {
    T = tile_grid(T);
    if (vsrc) {                         // keys with values (wide keys): one instantiation, run-time shift
        hipLaunchKernelGGL((scatter_kernel<-1, false, true>), dim3(T), dim3(SORT_THREADS), scatter_smem_bytes_vals(ipt), s, src, dst, d_n, cnt, bin_tot, ipt, shift,
                           SegMap{nullptr, nullptr, 0, 0}, (u64*)nullptr, vsrc, vdst);
        return;
    }
#define SC(SH) hipLaunchKernelGGL(scatter_kernel<SH>, dim3(T), dim3(SORT_THREADS), scatter_smem_bytes(ipt), s, src, dst, d_n, cnt, bin_tot, ipt, shift, seg, g_stamps)
    if (seg.prefix) {                   // the one segmented pass: run-time shift, map lookup compiled in
        hipLaunchKernelGGL((scatter_kernel<-1, true>), dim3(T), dim3(SORT_THREADS), scatter_smem_bytes(ipt), s, src, dst, d_n, cnt, bin_tot, ipt, shift, seg, g_stamps);
        return;
    }
    const bool rt = (shift & 7u) != 0 || getenv("FASTF_SORT_RUNTIME_SHIFT");
    switch (rt ? 64u : shift) {
    case 0: SC(0); break;  case 8: SC(8); break;  case 16: SC(16); break; case 24: SC(24); break;
    case 32: SC(32); break; case 40: SC(40); break; case 48: SC(48); break; case 56: SC(56); break;
    default: SC(-1); break;
    }
#undef SC
}

static int set_scatter_lds_limit() {
    const void* fns[10] = {(const void*)scatter_kernel<0>, (const void*)scatter_kernel<8>, (const void*)scatter_kernel<16>,
                           (const void*)scatter_kernel<24>, (const void*)scatter_kernel<32>, (const void*)scatter_kernel<40>,
                           (const void*)scatter_kernel<48>, (const void*)scatter_kernel<56>, (const void*)scatter_kernel<-1>,
                           (const void*)scatter_kernel<-1, true>};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_smem_bytes()) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)scatter_kernel<-1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_smem_bytes_vals(SORT_IPT)) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void*)scatter_regions_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_smem_bytes()) != hipSuccess) return 1;
    return 0;
}

// 8-bit LSD passes over the key bits [low_bit, key_bits): pass q sorts the digit at low_bit + 8q (the last one may reach
// past key_bits, where every key holds zeros).  low_bit = 0 is the full sort.
// vals / vtmp: 64-bit values that move with the keys (wide keys), ping-pong like the keys
static int launch_sort(fastf_engine* e, u64* keys, u64* tmp, const u64* d_n, u64 max_n, u32 key_bits, u32 low_bit,
                       int* sorted_in_tmp, hipStream_t s, bool segmented_input = false, u64* vals = nullptr, u64* vtmp = nullptr) {
    if (key_bits > 64) return set_err("key_bits %u > 64", key_bits);
    const u32 passes = sort_passes(key_bits, low_bit);
    *sorted_in_tmp = (int)(passes & 1);
    if (max_n == 0) return 0;
    if (max_n >= (1ull << 32)) return set_err("sort of %llu keys: limit is 2^32-1 per shard", (unsigned long long)max_n);
    if (reserve_workspace(e, 0, max_n)) return 1;
    const u32 ipt = choose_sort_ipt(max_n);
    const u32 T = (u32)((max_n + (u64)ipt * SORT_THREADS - 1) / ((u64)ipt * SORT_THREADS));
    u32* bintot = (u32*)e->d_binbase.p; u32* cnt = (u32*)e->d_cnt.p;
    SegMap seg{nullptr, nullptr, 0, 0};
    bool walk = false;
    if (segmented_input) {
        // the keys lie in the regions the streaming K1b left (or fastf_dev_set_regions named)
        if (!e->seg_n) return set_err("FASTF_SORT_SEGMENTED without a preceding FASTF_PROBE_SEGMENTED probe_pack");
        if (passes == 0) return set_err("segmented keys need at least one sort pass");
        // K1b left a histogram of this very digit per region: the first pass walks the regions, nothing is counted
        static const bool no_walk = getenv("FASTF_NO_REGION_WALK") != nullptr;
        walk = e->rgn_hist_n == e->seg_n && e->rgn_hist_shift == low_bit && !no_walk && !vals;
        if (!walk) {
            // otherwise (another digit grid, regions without histograms): count per tile, reading through the region map
            if (!e->seg_stride) return set_err("internal error: regions without a common stride need their histograms");
            seg = SegMap{(const u64*)e->d_segprefix.p, (const TileSeg*)e->d_tileseg.p, e->seg_n, e->seg_stride};
            hipLaunchKernelGGL(seg_tiles_kernel, dim3(std::min<u32>((T + 255) / 256, 1024)), dim3(256), 0, s, seg.prefix, seg.n_seg,
                               seg.stride, ipt * SORT_THREADS, (TileSeg*)e->d_tileseg.p);
        }
    }
    const SegMap none{nullptr, nullptr, 0, 0};
    for (u32 q = 0; q < passes; ++q) {
        const u64* src = (q & 1) ? tmp : keys;
        u64* dst = (q & 1) ? keys : tmp;
        const u32 shift = low_bit + 8 * q;
        if (q == 0 && walk) {
            const u32 R = e->seg_n, n_blk = (R + RGN_BLK - 1) / RGN_BLK;
            if (e->d_rgn_blk.ensure((size_t)n_blk * RADIX * sizeof(u32))) return 1;
            hipLaunchKernelGGL(rgn_scan_blocks_kernel, dim3(n_blk), dim3(RADIX), 0, s, (u32*)e->d_rgn_hist.p, R, (u32*)e->d_rgn_blk.p);
            hipLaunchKernelGGL(rgn_scan_fix_kernel, dim3(n_blk), dim3(RADIX), 0, s, (u32*)e->d_rgn_hist.p, R, (const u32*)e->d_rgn_blk.p, n_blk, bintot);
            t_begin(e, s);
            hipLaunchKernelGGL(scatter_regions_kernel, dim3(tile_grid(R)), dim3(SORT_THREADS), scatter_smem_bytes(ipt), s, src, dst,
                               (const u64*)e->d_rgn_phys.p, (const u64*)e->d_segcount.p, (const u32*)e->d_rgn_hist.p, (const u32*)bintot, R, ipt, shift,
                               max_n >= (24ull << 20));
            t_end(e, s, &e->t_scatter_ms, &e->t_scatter_n);
            e->rgn_hist_n = 0;                           // (the histograms are offsets now: consumed)
            dbg_sync(s, "K2 sort pass (region walk)");
            continue;
        }
        t_begin(e, s);
        if (q == 0 && seg.prefix) hipLaunchKernelGGL(tile_count_kernel<true>, dim3(tile_grid(T)), dim3(SORT_THREADS), 0, s, src, d_n, shift, cnt, ipt, seg);
        else hipLaunchKernelGGL(tile_count_kernel<false>, dim3(tile_grid(T)), dim3(SORT_THREADS), 0, s, src, d_n, shift, cnt, ipt, none);
        t_end(e, s, &e->t_count_ms, &e->t_count_n);
        hipLaunchKernelGGL(row_scan_kernel, dim3(RADIX), dim3(1024), 0, s, cnt, d_n, bintot, ipt);
        t_begin(e, s);
        if (vals) launch_scatter(shift, T, s, src, dst, d_n, (const u32*)cnt, (const u32*)bintot, ipt, none, (q & 1) ? vtmp : vals, (q & 1) ? vals : vtmp);
        else launch_scatter(shift, T, s, src, dst, d_n, (const u32*)cnt, (const u32*)bintot, ipt, q == 0 ? seg : none);
        t_end(e, s, &e->t_scatter_ms, &e->t_scatter_n);
        dbg_sync(s, "K2 sort pass");
    }
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int fastf_dev_set_regions(fastf_engine_t* e, const uint64_t* d_counts, uint32_t n_regions, uint64_t stride,
                                     uint64_t* d_n_out, void* stream) FASTF_TRY {
    if (!e || !d_counts || !d_n_out) return set_err("null argument");
    if (e->multi) return set_err("fastf_dev_set_regions: device-level calls take a single-device engine");
    if (n_regions == 0 || stride == 0) return set_err("fastf_dev_set_regions: no regions");
    HIP_OK(hipSetDevice(e->device));
    if (e->d_segprefix.ensure(((size_t)n_regions + 1) * sizeof(u64))) return 1;
    hipLaunchKernelGGL(seg_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const u64*)d_counts, (u32)n_regions, (u64*)e->d_segprefix.p, (u64*)d_n_out);
    HIP_OK(hipGetLastError());
    e->seg_n = n_regions; e->seg_stride = stride; e->rgn_hist_n = 0;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_dev_sort(fastf_engine_t* e, uint64_t* d_keys, uint64_t* d_tmp, const uint64_t* d_n,
                              uint64_t max_n, uint32_t key_bits, uint32_t flags, int* sorted_in_tmp, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_sort: device-level calls take a single-device engine");
    NO_WIDE(e, "fastf_dev_sort");
    HIP_OK(hipSetDevice(e->device));
    int dummy = 0;
    return launch_sort(e, (u64*)d_keys, (u64*)d_tmp, (const u64*)d_n, max_n, key_bits,
                       (flags & FASTF_SORT_SKIP_LOW) ? e->skip_bits : 0,
                       sorted_in_tmp ? sorted_in_tmp : &dummy, (hipStream_t)stream, (flags & FASTF_SORT_SEGMENTED) != 0);
} FASTF_CATCH_INT

// K3 grid: every workgroup owns one contiguous chunk of the keys, so the launch is one round of resident workgroups
// (LDS: 21 KB per workgroup, 37 KB with the hash set of DEDUP 2; 64 VGPRs: eight waves per SIMD, four workgroups per CU)
static u32 k3_grid(u64 max_n, int dedup) {
    static int per_cu[3] = {-1, -1, -1};
    if (per_cu[dedup] < 0) {
        const char* c = getenv("FASTF_K3_PER_CU");
        per_cu[dedup] = c ? atoi(c) : 4;
        if (per_cu[dedup] < 1) per_cu[dedup] = 1;
    }
    const u64 tiles = (max_n + K3_TILE - 1) / K3_TILE;
    return (u32)std::min<u64>(std::max<u64>(tiles, 1), std::min<u64>((u64)per_cu[dedup] * g_cu_count, 4096));
}

// rows into the engine's row regions (one per workgroup chunk), chunk row counts -> bases, total -> *nrows
// wide_vals: the values beside `sorted` (wide keys: sorted = group words); the matrix rows only
template <bool UMI_ROWS>
static int launch_reduce_regions(fastf_engine* e, const u64* sorted, const u64* d_n, u64 max_n, u64* nrows, u32 low_skip, hipStream_t s,
                                 const u64* wide_vals = nullptr) {
    e->rg_n = 0;
    if (max_n == 0) { HIP_OK(hipMemsetAsync(nrows, 0, sizeof(u64), s)); return 0; }
    if (wide_vals && UMI_ROWS) return set_err("internal error: -u rows of wide keys");
    const int dedup = wide_vals ? 2 : (UMI_ROWS || low_skip == 0) ? 0 : 2;      // 2: keys sorted on (cell, feature) only (reduce_hashed_kernel)
    u32 G = k3_grid(max_n, dedup);
    if (dedup == 2) G = std::min<u32>(G, (wide_vals ? 2u : 3u) * g_cu_count);      // reduce_hashed_kernel: 40-58 KB of LDS and 80+ VGPRs, three workgroups per CU (wide keys: two)
    if (e->d_rg_count.ensure(max_n * 4)) return 1;
    if (UMI_ROWS ? e->d_rg_ukeys.ensure(max_n * 8) : (e->d_rg_feature.ensure(max_n * 4) || e->d_rg_cell.ensure(max_n * 4))) return 1;
    if (e->d_spanrows.ensure(4096 * sizeof(u32)) || e->d_spanbase.ensure(4097 * sizeof(u64))) return 1;
    if (dedup == 2 && !e->d_giant.p) {                  // work items of groups longer than a window + their counter (live, frozen)
        if (e->d_giant.ensure((size_t)GIANT_LIST_CAP * GIANT_ITEM_WORDS * sizeof(u64) + 64)) return 1;
        HIP_OK(hipMemsetAsync(e->d_giant.p, 0, e->d_giant.bytes, s));       // the list too: no slot of it is ever read unwritten
    }
    // the item counter: two of them by turns — giant_groups_kernel reads the one this launch counts up and clears the other
    u32* const giant_cnt = dedup == 2 ? (u32*)((char*)e->d_giant.p + (size_t)GIANT_LIST_CAP * GIANT_ITEM_WORDS * sizeof(u64)) : nullptr;
    const u32 parity = dedup == 2 ? (e->giant_parity ^= 1u) : 0u;
    u32* const giant_n = giant_cnt ? giant_cnt + parity : nullptr;
    ReduceParams p{};
    p.keys = sorted; p.n_ptr = d_n; p.L = e->L; p.feat_mask = (u32)((1ull << (wide_vals ? e->row_split : e->feat_bits)) - 1);
    p.err = (u64*)e->d_small.p + SM_COUNTERS + 3;
    p.feature = (u32*)e->d_rg_feature.p; p.cell = (u32*)e->d_rg_cell.p; p.count = (u32*)e->d_rg_count.p; p.ukeys = (u64*)e->d_rg_ukeys.p;
    p.span_rows = (u32*)e->d_spanrows.p;
    p.giant_list = (u64*)e->d_giant.p; p.giant_n = giant_n;
    // groups beyond giant_max send the caller to the full sort; wide keys have no full-sort path, and giant_groups_kernel takes
    // any length (one work item per ~1536 keys, GIANT_LIST_CAP items per launch): 4 M reads of one gene in one cell
    p.giant_max = wide_vals ? (1u << 22) : GIANT_MAX;
    p.vals = wide_vals; p.wide_feat_bits = e->row_split;       // (the rows' two halves: (cell, feature) but for engines with sub-groups)
    t_begin(e, s);
    if (UMI_ROWS) hipLaunchKernelGGL((reduce_windows_kernel<true>), dim3(G), dim3(K3_THREADS), 0, s, p);
    else if (dedup == 0) hipLaunchKernelGGL((reduce_windows_kernel<false>), dim3(G), dim3(K3_THREADS), 0, s, p);
    else if (wide_vals) hipLaunchKernelGGL((reduce_hashed_kernel<true, true>), dim3(G), dim3(K3H_THREADS), 0, s, p);
    else if (e->L.feat_shift > 27) hipLaunchKernelGGL((reduce_hashed_kernel<true>), dim3(G), dim3(K3H_THREADS), 0, s, p);    // UMIs beyond 12 bases: 64-bit slots
    else hipLaunchKernelGGL((reduce_hashed_kernel<false>), dim3(G), dim3(K3H_THREADS), 0, s, p);
    if (dedup == 2)       // (its last workgroup scans the chunks' row counts: no span_scan launch of its own)
        hipLaunchKernelGGL(giant_groups_kernel, dim3(g_cu_count + 1), dim3(512), 0, s, wide_vals ? wide_vals : sorted, (const u64*)e->d_giant.p, giant_cnt, parity, e->L,
                           (u32*)e->d_rg_count.p, (u64*)e->d_small.p + SM_COUNTERS + 3, (const u32*)e->d_spanrows.p, G, (u64*)e->d_spanbase.p, nrows);
    else
        hipLaunchKernelGGL(span_scan_kernel, dim3(1), dim3(1024), 0, s, (const u32*)e->d_spanrows.p, G, (u64*)e->d_spanbase.p, nrows);
    HIP_OK(hipGetLastError());
    if (!UMI_ROWS) t_end(e, s, &e->t_k3_ms, &e->t_k3_n);
    dbg_sync(s, "K3 reduce + span scan + giant groups");
    e->rg_n = G; e->rg_umi = UMI_ROWS;
    return 0;
}

// concatenation of the regions of the last launch_reduce_regions on this engine (same d_n); the destinations are device
// arrays or pinned host memory
template <bool UMI_ROWS>
static int launch_rows_gather(fastf_engine* e, const u64* d_n, u32* feature, u32* cell, u32* count, u64* ukeys, hipStream_t s) {
    if (!e->rg_n) return 0;                             // nothing was reduced (max_n == 0)
    if (e->rg_umi != UMI_ROWS) return set_err("rows gather: the last reduce on this engine was of the other kind");
    t_begin(e, s);
    hipLaunchKernelGGL(rows_gather_kernel<UMI_ROWS>, dim3(e->rg_n), dim3(256), 0, s, (const u32*)e->d_rg_feature.p, (const u32*)e->d_rg_cell.p,
                       (const u32*)e->d_rg_count.p, (const u64*)e->d_rg_ukeys.p, (const u32*)e->d_spanrows.p, (const u64*)e->d_spanbase.p,
                       d_n, e->rg_n, feature, cell, count, ukeys);
    HIP_OK(hipGetLastError());
    if (!UMI_ROWS) t_end(e, s, &e->t_gather_ms, &e->t_gather_n);
    return 0;
}

template <bool UMI_ROWS>
static int launch_reduce(fastf_engine* e, const u64* sorted, const u64* d_n, u64 max_n, u32* feature, u32* cell,
                         u32* count, u64* ukeys, u64* nrows, u32 low_skip, hipStream_t s) {
    if (launch_reduce_regions<UMI_ROWS>(e, sorted, d_n, max_n, nrows, low_skip, s)) return 1;
    return launch_rows_gather<UMI_ROWS>(e, d_n, feature, cell, count, ukeys, s);
}

extern "C" int fastf_dev_reduce(fastf_engine_t* e, const uint64_t* d_sorted, const uint64_t* d_n, uint64_t max_n,
                                uint32_t* d_feature, uint32_t* d_cell, uint32_t* d_count, uint64_t* d_nnz, uint32_t flags,
                                void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_reduce: device-level calls take a single-device engine");
    NO_WIDE(e, "fastf_dev_reduce");
    HIP_OK(hipSetDevice(e->device));
    const u32 low = (flags & FASTF_SORT_SKIP_LOW) ? e->skip_bits : 0;
    if (flags & FASTF_REDUCE_SEGMENTED)
        return launch_reduce_regions<false>(e, (const u64*)d_sorted, (const u64*)d_n, max_n, (u64*)d_nnz, low, (hipStream_t)stream);
    return launch_reduce<false>(e, (const u64*)d_sorted, (const u64*)d_n, max_n, d_feature, d_cell, d_count, nullptr,
                                (u64*)d_nnz, low, (hipStream_t)stream);
} FASTF_CATCH_INT

extern "C" int fastf_dev_rows_gather(fastf_engine_t* e, const uint64_t* d_n, uint32_t* feature, uint32_t* cell, uint32_t* count,
                                     void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_rows_gather: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    if (debug_known_memory(feature, 4, "fastf_dev_rows_gather feature") || debug_known_memory(cell, 4, "fastf_dev_rows_gather cell") || debug_known_memory(count, 4, "fastf_dev_rows_gather count")) return 1;
    return launch_rows_gather<false>(e, (const u64*)d_n, feature, cell, count, nullptr, (hipStream_t)stream);
} FASTF_CATCH_INT

extern "C" int fastf_dev_umi_rows(fastf_engine_t* e, const uint64_t* d_sorted, const uint64_t* d_n, uint64_t max_n,
                                  uint64_t* d_ukeys, uint32_t* d_ncopy, uint64_t* d_nrows, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_umi_rows: device-level calls take a single-device engine");
    NO_WIDE(e, "fastf_dev_umi_rows");
    HIP_OK(hipSetDevice(e->device));
    return launch_reduce<true>(e, (const u64*)d_sorted, (const u64*)d_n, max_n, nullptr, nullptr, d_ncopy,
                               (u64*)d_ukeys, (u64*)d_nrows, 0, (hipStream_t)stream);
} FASTF_CATCH_INT

extern "C" int fastf_dev_error_bits(fastf_engine_t* e, uint64_t* bits) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_error_bits: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipDeviceSynchronize());
    if (copy_d2h(bits, (u64*)e->d_small.p + SM_COUNTERS + 3, sizeof(u64))) return 1;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_dev_clear_error_bits(fastf_engine_t* e, uint64_t mask, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_clear_error_bits: device-level calls take a single-device engine");
    HIP_OK(hipSetDevice(e->device));
    // stream-ordered atomicAnd on the device: kernels raise bits with atomicOr, a host read-modify-write would race them
    hipLaunchKernelGGL(clear_bits_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (u64*)e->d_small.p + SM_COUNTERS + 3, (u64)mask);
    HIP_OK(hipGetLastError());
    return 0;
} FASTF_CATCH_INT

extern "C" const char* fastf_kernel_names(void) FASTF_TRY {
    return "probe_cells_kernel,probe_cells_lds_kernel,probe_cells_filtered_kernel,scan_tiles_kernel,filter_pack_kernel,filter_pack_stream_kernel,"
           "block_records_kernel,tile_count_kernel,row_scan_kernel,scatter_kernel,reduce_windows_kernel,reduce_hashed_kernel,span_scan_kernel,"
           "giant_groups_kernel,rows_gather_kernel";
} FASTF_CATCH_ZERO

// ------------------------------------------------------------------------------------
// host-buffer streaming API
// ------------------------------------------------------------------------------------
static const char* err_bits_text(u64 bits) {
    static thread_local char buf[256];
    snprintf(buf, sizeof buf, "device error bits 0x%llx:%s%s%s%s%s", (unsigned long long)bits,
             (bits & 1) ? " (unknown bit 0);" : "",
             (bits & ERR_DRAWS_SHORT) ? " draw stream shorter than CB hits;" : "",
             (bits & ERR_UMI_TOOLONG) ? " UMI longer than umi_max_bases (raise it: up to 32, with fastf_batch_t.umi_ext from 17 on);" : "",
             (bits & ERR_KEYS_FULL) ? " key store full;" : "",
             (bits & ERR_RUN_TOO_LONG) ? " unsorted run longer than the group-only path handles: sort fully (drop FASTF_SORT_SKIP_LOW);" : "");
    return buf;
}

// SoA: cb | gx | umi | meta (+ 4: umi_ext of engines with umi_max_bases > 16); blocked: cb | one run of 4608 or 5120 bytes per 256
// records (8 + 20 bytes per record at most, and the last run whole)
static size_t stage_bytes(u64 cap) { return (size_t)cap * (8 + 8 + 4 + 4 + 4) + 8192; }

// Streaming push path.
//   chunk i:  [host staging, pageable input only]  ->  H2D on s_copy  ->  K1a + scan + K1b on s_compute
// Nothing in it waits for the chunk before: the hit-rank base of a chunk is a device-side running total (the scan
// kernel hands it to K1b), the draws live in a device ring addressed by absolute hit rank, and the host keeps that ring
// filled up to the highest rank the chunk could possibly reach (hits retired so far + every record still in flight).
// A chunk is retired — its counters read — only when its slot is needed again, two chunks later.

static int slot_alloc(fastf_engine* e, fastf_engine::Slot& sl, bool need_host_stage) {
    if (!sl.h_small) HIP_OK(hipHostMalloc((void**)&sl.h_small, SM_WORDS * sizeof(u64), hipHostMallocDefault));
    if (sl.d_stage.ensure(stage_bytes(e->batch_cap))) return 1;
    if (need_host_stage && !sl.h_stage) HIP_OK(hipHostMalloc(&sl.h_stage, stage_bytes(e->batch_cap), hipHostMallocDefault));
    return 0;
}

// wait for a chunk and fold its counters into the host-side running state
static int retire_slot(fastf_engine* e, fastf_engine::Slot& sl) {
    if (!sl.busy) return 0;
    HIP_OK(hipEventSynchronize(sl.ev_done));
    sl.busy = false;
    e->inflight_records -= sl.n;
    const u64* c = sl.h_small + SM_COUNTERS;
    if (!sl.external) e->mt_hits += c[0] - e->hits_so_far;
    e->hits_so_far = c[0];
    e->c_sampled = c[1]; e->c_valid = c[2];
    e->keys_so_far = sl.h_small[SM_KEYCOUNT];
    if (c[3]) return set_err("%s", err_bits_text(c[3]));
    return 0;
}
static int retire_all(fastf_engine* e) {
    // oldest first: the running state must end at the newest chunk's counters
    const int first = e->cur;              // slot[cur] holds the older of the two chunks in flight
    if (retire_slot(e, e->slot[first])) return 1;
    return retire_slot(e, e->slot[first ^ 1]);
}

// stream mode: the key store holds SLOTS (the regions of the chunks, with gaps): room for `need` of them
static int grow_slots(fastf_engine* e, u64 need) {
    if (need <= e->slot_cap && e->d_keys.p) return 0;
    if (retire_all(e)) return 1;
    HIP_OK(hipStreamSynchronize(e->s_compute));
    const u64 ncap = std::max<u64>(need, std::max<u64>(e->slot_cap * 2, 1u << 22));
    void* np = nullptr;
    HIP_OK(hipMalloc(&np, ncap * sizeof(u64)));
    if (e->d_keys.p && e->slots_used) HIP_OK(hipMemcpy(np, e->d_keys.p, e->slots_used * sizeof(u64), hipMemcpyDeviceToDevice));
    e->d_keys.release();
    e->d_keys.p = np; e->d_keys.bytes = ncap * sizeof(u64);
    e->slot_cap = ncap;
    return 0;
}
// ... and the tables of its regions (first slot, key count, histogram of the first sort digit): room for `need` regions
static int grow_regions(fastf_engine* e, u32 need) {
    if (need <= e->rgn_cap && e->d_rgn_hist.p) return 0;
    if (retire_all(e)) return 1;
    HIP_OK(hipStreamSynchronize(e->s_compute));
    const u32 ncap = std::max<u32>(need, std::max<u32>(e->rgn_cap * 2, 8192u));
    DevBuf* bufs[3] = {&e->d_segcount, &e->d_rgn_phys, &e->d_rgn_hist};
    const size_t per[3] = {sizeof(u64), sizeof(u64), RADIX * sizeof(u32)};
    for (int i = 0; i < 3; ++i) {
        void* np = nullptr;
        HIP_OK(hipMalloc(&np, (size_t)ncap * per[i]));
        if (bufs[i]->p && e->rgn_n) HIP_OK(hipMemcpy(np, bufs[i]->p, (size_t)e->rgn_n * per[i], hipMemcpyDeviceToDevice));
        bufs[i]->release();
        bufs[i]->p = np; bufs[i]->bytes = (size_t)ncap * per[i];
    }
    e->rgn_cap = ncap;
    return 0;
}

static int grow_keys(fastf_engine* e, u64 need) {
    if (need <= e->key_cap && (e->stream_mode ? e->d_tmp.p : e->d_keys.p)) return 0;
    if (retire_all(e)) return 1;           // the copy below needs the exact key count and a quiet key store
    HIP_OK(hipStreamSynchronize(e->s_compute));
    u64 ncap = std::max<u64>(need, std::max<u64>(e->key_cap * 2, 1u << 20));
    if (e->stream_mode) {
        // the keys themselves live in the slot store (grow_slots); this is the bound on their NUMBER, which sizes what
        // fastf_engine_finish works in
        e->key_cap = ncap;
        if (e->d_tmp.ensure(ncap * sizeof(u64)) || e->d_feature.ensure(ncap * 4) || e->d_cell.ensure(ncap * 4) || e->d_count.ensure(ncap * 4)) return 1;
        if (e->d_rg_feature.ensure(ncap * 4) || e->d_rg_cell.ensure(ncap * 4) || e->d_rg_count.ensure(ncap * 4)) return 1;
        return reserve_workspace(e, 0, ncap);
    }
    void* np = nullptr;
    HIP_OK(hipMalloc(&np, ncap * sizeof(u64)));
    if (e->d_keys.p && e->keys_so_far)
        HIP_OK(hipMemcpy(np, e->d_keys.p, e->keys_so_far * sizeof(u64), hipMemcpyDeviceToDevice));
    e->d_keys.release();
    e->d_keys.p = np; e->d_keys.bytes = ncap * sizeof(u64);
    if (e->wide) {                                       // the rest of every key, beside it
        void* nv = nullptr;
        HIP_OK(hipMalloc(&nv, ncap * sizeof(u64)));
        if (e->d_vals.p && e->keys_so_far) HIP_OK(hipMemcpy(nv, e->d_vals.p, e->keys_so_far * sizeof(u64), hipMemcpyDeviceToDevice));
        e->d_vals.release();
        e->d_vals.p = nv; e->d_vals.bytes = ncap * sizeof(u64);
        if (e->d_vtmp.ensure(ncap * sizeof(u64))) return 1;
    }
    e->key_cap = ncap;
    // what fastf_engine_finish needs for this many keys (sort buffer, row arrays, tile workspace) is allocated here, while
    // the caller is still decoding, and not on the way out (a dozen hipMallocs: 10-15 ms of a 0.5 s end-to-end run)
    if (e->d_tmp.ensure(ncap * sizeof(u64)) || e->d_feature.ensure(ncap * 4) || e->d_cell.ensure(ncap * 4) || e->d_count.ensure(ncap * 4)) return 1;
    if (e->d_rg_feature.ensure(ncap * 4) || e->d_rg_cell.ensure(ncap * 4) || e->d_rg_count.ensure(ncap * 4)) return 1;   // K3's row regions
    return reserve_workspace(e, 0, ncap);
}

// the draw source of the chunks being pushed: the engine-owned MT19937 stream, or an array the caller supplied
struct DrawSource { const u32* ext; u64 ext_abs0; u64 ext_n; };

// make the ring hold every absolute rank below `upto`
static bool device_mt_wanted() { const char* h = getenv("FASTF_HOST_DRAWS"); return !(h && h[0] == '1'); }   // (read where a stream is positioned: once per job)

static int upload_draws(fastf_engine* e, fastf_engine::Slot& sl, const DrawSource& src, u64 upto) {
    if (!src.ext && e->mt_on_device) {
        // the engine-owned stream lives on the device: one launch continues it by exactly the ranks that are missing and leaves
        // their decisions in the ring (on a stream of its own: one workgroup walks the stream block by block, about a
        // nanosecond per two draws — beside the record copies, not in front of them; K1 waits for both)
        if (e->draws_up < upto) {
            if (launch_mt_decisions_par(e, e->s_mt, (u32*)e->d_mt.p, &e->mt_dev_idx, e->d_mtwords, (u32*)e->d_ring.p, e->draws_up, upto - e->draws_up, e->ring_len - 1, e->threshold)) return 1;
            HIP_OK(hipEventRecord(e->ev_mt, e->s_mt));
            HIP_OK(hipStreamWaitEvent(e->s_compute, e->ev_mt, 0));
            e->draws_up = upto;
        }
        return 0;
    }
    // host-generated or caller-supplied draws: the decisions are packed here and only they go up — whole words; the word a
    // chunk ends in is kept (ring_carry) and goes up again, completed, with the next one (a kernel that is still reading its
    // low bits sees the same bits)
    if (!sl.h_draws) HIP_OK(hipHostMalloc(&sl.h_draws, e->batch_cap * 4, hipHostMallocDefault));
    const u64 ring_words = e->ring_len >> 5;
    while (e->draws_up < upto) {
        const u64 n = std::min<u64>(upto - e->draws_up, e->batch_cap);
        u32* h = (u32*)sl.h_draws;
        if (src.ext) {
            const u64 a = e->draws_up - src.ext_abs0;                  // index into the caller's array
            const u64 have = a < src.ext_n ? std::min<u64>(src.ext_n - a, n) : 0;
            if (have) memcpy(h, src.ext + a, have * 4);
            if (have < n) memset(h + have, 0, (n - have) * 4);
        } else {
            fastf_mt_fill(&e->mt, h, n);
        }
        // in place: word (s + i) >> 5 is written after draw i was read, and never ahead of it
        const u32 s0 = (u32)(e->draws_up & 31);
        const u64 thr = e->threshold;
        u32 wcur = s0 ? e->ring_carry : 0u;
        u64 nw = 0;
        for (u64 i = 0; i < n; ++i) {
            const u32 b = (u32)((s0 + i) & 31);
            wcur |= (u32)((u64)h[i] < thr) << b;
            if (b == 31) { h[nw++] = wcur; wcur = 0; }
        }
        e->ring_carry = wcur;
        if ((s0 + n) & 31) h[nw++] = wcur;
        const u64 wpos = (e->draws_up & (e->ring_len - 1)) >> 5, first = std::min<u64>(nw, ring_words - wpos);
        HIP_OK(hipMemcpyAsync((u32*)e->d_ring.p + wpos, h, first * 4, hipMemcpyHostToDevice, e->s_copy));
        if (first < nw) HIP_OK(hipMemcpyAsync((u32*)e->d_ring.p, h + first, (nw - first) * 4, hipMemcpyHostToDevice, e->s_copy));
        e->draws_up += n;
        if (e->draws_up < upto) HIP_OK(hipStreamSynchronize(e->s_copy));   // the staging buffer is about to be refilled (uneven chunk sizes only)
    }
    return 0;
}

// pinned == true: the batch arrays are pinned host memory and go to the device straight from where they are
static int push_chunk(fastf_engine* e, const fastf_batch_t* b, size_t off, size_t n, const DrawSource& src, bool pinned) {
    fastf_engine::Slot& sl = e->slot[e->cur];
    if (retire_slot(e, sl)) return 1;                                   // the chunk pushed two chunks ago
    if (slot_alloc(e, sl, !pinned)) return 1;
    // upper bounds for what the device will have seen when this chunk's K1 starts
    const u64 hits_ub = e->hits_so_far + e->inflight_records, keys_ub = e->keys_so_far + e->inflight_records;
    if (grow_keys(e, keys_ub + n)) return 1;
    const u64 cap = e->batch_cap;
    char* ds = (char*)sl.d_stage.p;
    // stream mode: this chunk's regions behind those of the chunks before
    u32 s_grid = 0; u64 s_region = 0, s_slots = 0;
    if (e->stream_mode) {
        stream_geometry(e, (n + K1_TILE - 1) / K1_TILE, &s_grid, &s_region);
        s_slots = (u64)s_grid * K1S_SUB * s_region;
        if (grow_slots(e, e->slots_used + s_slots) || grow_regions(e, e->rgn_n + s_grid * (u32)K1S_SUB)) return 1;
    }
    const size_t o_gx = cap * 8, o_umi = cap * 16, o_meta = cap * 20, o_ext = cap * 24;
    const void *s_cb = b->cb_key + off, *s_gx = b->gx_key + off, *s_umi = b->umi + off, *s_meta = b->meta + off;
    const void *s_ext = (e->long_umi && b->umi_ext) ? b->umi_ext + off : nullptr;
    if (!pinned) {
        // staging: three streams of 8 bytes per record, copied side by side
        char* hs = (char*)sl.h_stage;
        std::thread t_gx, t_um;
        const bool par = n >= (1u << 18);
        if (par) {
            t_gx = std::thread([=] { memcpy(hs + o_gx, s_gx, n * 8); });
            t_um = std::thread([=] { memcpy(hs + o_umi, s_umi, n * 4); memcpy(hs + o_meta, s_meta, n * 4); });
        } else {
            memcpy(hs + o_gx, s_gx, n * 8); memcpy(hs + o_umi, s_umi, n * 4); memcpy(hs + o_meta, s_meta, n * 4);
        }
        memcpy(hs, s_cb, n * 8);
        if (s_ext) { memcpy(hs + o_ext, s_ext, n * 4); s_ext = hs + o_ext; }
        if (par) { t_gx.join(); t_um.join(); }
        s_cb = hs; s_gx = hs + o_gx; s_umi = hs + o_umi; s_meta = hs + o_meta;
    }
    hipStream_t sc = e->s_copy, sk = e->s_compute;
    // (hipMemcpyDefault: a "pinned" batch may also be DEVICE memory — the records the device-side BAM front end packed)
    HIP_OK(hipMemcpyAsync(ds, s_cb, n * 8, hipMemcpyDefault, sc));
    char* const blk = ds + cap * 8;                                     // stream mode: the blocked runs behind the cb keys
    if (e->stream_mode) {
        // The batch lands in the BLOCKED layout (umi_kernels.hpp: per 256 records one run gx | umi | meta | cell scratch): three
        // pitched copies — a row is one unit's slice of an array, the destination pitch the run — and three plain ones for the
        // records of a last, partial unit.  Same rate as plain copies (tools/h2d_2d_probe.hip).
        const u32 run = blk_run_bytes(e->cell16);
        const u64 full = n / BLK_RECS, tail = n % BLK_RECS;
        if (full) {
            HIP_OK(hipMemcpy2DAsync(blk + BLK_GX, run, s_gx, BLK_RECS * 8, BLK_RECS * 8, full, hipMemcpyDefault, sc));
            HIP_OK(hipMemcpy2DAsync(blk + BLK_UMI, run, s_umi, BLK_RECS * 4, BLK_RECS * 4, full, hipMemcpyDefault, sc));
            HIP_OK(hipMemcpy2DAsync(blk + BLK_META, run, s_meta, BLK_RECS * 4, BLK_RECS * 4, full, hipMemcpyDefault, sc));
        }
        if (tail) {
            char* const last = blk + full * run;
            HIP_OK(hipMemcpyAsync(last + BLK_GX, (const char*)s_gx + full * BLK_RECS * 8, tail * 8, hipMemcpyDefault, sc));
            HIP_OK(hipMemcpyAsync(last + BLK_UMI, (const char*)s_umi + full * BLK_RECS * 4, tail * 4, hipMemcpyDefault, sc));
            HIP_OK(hipMemcpyAsync(last + BLK_META, (const char*)s_meta + full * BLK_RECS * 4, tail * 4, hipMemcpyDefault, sc));
        }
    } else {
    HIP_OK(hipMemcpyAsync(ds + o_gx, s_gx, n * 8, hipMemcpyDefault, sc));
    HIP_OK(hipMemcpyAsync(ds + o_umi, s_umi, n * 4, hipMemcpyDefault, sc));
    HIP_OK(hipMemcpyAsync(ds + o_meta, s_meta, n * 4, hipMemcpyDefault, sc));
    }
    if (e->long_umi) {                                                  // bases 17.. (zeros when the batch has none)
        if (s_ext) HIP_OK(hipMemcpyAsync(ds + o_ext, s_ext, n * 4, hipMemcpyDefault, sc));
        else HIP_OK(hipMemsetAsync(ds + o_ext, 0, n * 4, sc));
    }
    e->cur_umi_ext = e->long_umi ? (const u32*)(ds + o_ext) : nullptr;
    if (upload_draws(e, sl, src, hits_ub + n)) return 1;               // generated while the record copies are in flight
    if (!src.ext) e->draws_valid = e->draws_up;
    HIP_OK(hipEventRecord(sl.ev_copy, sc));
    HIP_OK(hipStreamWaitEvent(sk, sl.ev_copy, 0));
    u64* small = (u64*)e->d_small.p;
    if (e->stream_mode) {
        // K1a writes the cell indices into the runs' scratch slices, the streaming K1b reads the runs and appends this chunk's
        // regions (keys, counts, first-digit histograms) to the store: what bench.py's resident pass runs, chunk by chunk
        const RegionAppend app{e->slots_used, e->rgn_n};
        if (launch_probe(e, (const u64*)ds, nullptr, nullptr, nullptr, n, (const u32*)e->d_ring.p, std::min(e->draws_up, e->draws_valid),
                         small + SM_DRAWBASE, (u64*)e->d_keys.p + e->slots_used, s_slots, small + SM_KEYCOUNT, small + SM_COUNTERS, false, sk,
                         e->ring_len - 1, small + SM_RUNNING, true, blk, &app))
            return 1;
        e->slots_used += s_slots; e->rgn_n += s_grid * (u32)K1S_SUB; e->store_regions = true;
    } else
    if (launch_probe(e, (const u64*)ds, (const u64*)(ds + o_gx), (const u32*)(ds + o_umi), (const u32*)(ds + o_meta), n,
                     (const u32*)e->d_ring.p, std::min(e->draws_up, e->draws_valid), small + SM_DRAWBASE, (u64*)e->d_keys.p, e->key_cap,
                     small + SM_KEYCOUNT, small + SM_COUNTERS, false, sk, e->ring_len - 1, small + SM_RUNNING))
        return 1;
    HIP_OK(hipMemcpyAsync(sl.h_small, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, sk));
    HIP_OK(hipEventRecord(sl.ev_done, sk));
    sl.busy = true; sl.n = n; sl.external = src.ext != nullptr;
    e->inflight_records += n;
    e->total_records += n;
    e->cur ^= 1;
    e->finished = false;
    return 0;
}

// stream mode, a push AFTER fastf_engine_finish (no reset in between: the job goes on): the sort passes have rewritten the store,
// so the keys so far become regions again — runs of 64 K keys of the sorted (or last permuted) array, their histograms counted
// by a kernel of its own (rare: nothing on the hot path does this)
static int rebase_store(fastf_engine* e) {
    const u64 n = e->n_sorted;
    HIP_OK(hipStreamSynchronize(e->s_compute));
    if (e->store_regions && n)                           // one pass only: the keys in the store are still the regions, the array is d_tmp's
        HIP_OK(hipMemcpy(e->d_keys.p, e->d_tmp.p, n * sizeof(u64), hipMemcpyDeviceToDevice));
    constexpr u64 RUN = 1u << 16;
    const u32 R = (u32)((n + RUN - 1) / RUN);
    e->rgn_n = 0; e->slots_used = 0; e->store_regions = false; e->rgn_hist_n = 0;
    if (R == 0) return 0;
    if (grow_regions(e, R)) return 1;
    std::vector<u64> phys(R), cnt(R);
    for (u32 i = 0; i < R; ++i) { phys[i] = (u64)i * RUN; cnt[i] = std::min<u64>(RUN, n - phys[i]); }
    if (copy_h2d(e->d_rgn_phys.p, phys.data(), R * sizeof(u64))) return 1;
    if (copy_h2d(e->d_segcount.p, cnt.data(), R * sizeof(u64))) return 1;
    hipLaunchKernelGGL(rgn_hist_kernel, dim3(std::min<u32>(R, 8 * g_cu_count)), dim3(256), 0, e->s_compute, (const u64*)e->d_keys.p, (const u64*)e->d_rgn_phys.p,
                       (const u64*)e->d_segcount.p, R, e->skip_bits, (u32*)e->d_rgn_hist.p);
    HIP_OK(hipGetLastError());
    e->rgn_n = R; e->slots_used = n; e->store_regions = true;
    return 0;
}

static int push_impl(fastf_engine_t* e, const fastf_batch_t* batch, const uint32_t* draws, size_t n_draws, bool pinned) {
    if (!e || !batch) return set_err("null argument");
    if (e->multi) {
        if (draws) return set_err("caller-supplied draws are not supported by the multi-device engine");
        return multi_push(e, batch, pinned);
    }
    if (e->n_shards != 1) return set_err("fastf_engine_push drives a single shard; use the fastf_dev_* calls for sharded runs");
    HIP_OK(hipSetDevice(e->device));
    if (batch->n == 0) return 0;
    if (e->stream_mode && e->finished && rebase_store(e)) return 1;
    if (!e->d_ring.p) {
        u64 r = 1024; while (r < 4 * e->batch_cap) r <<= 1;            // three chunks of ranks can be live at once
        if (e->d_ring.ensure(r / 8)) return 1;                         // one bit per rank
        e->ring_len = r;
    }
    DrawSource src{nullptr, 0, 0};
    if (draws) {
        // caller-supplied draws: draws[i] belongs to the i-th CB hit of this batch.  The ring restarts at the exact hit count.
        if (retire_all(e)) return 1;
        HIP_OK(hipStreamSynchronize(e->s_copy));
        src.ext = draws; src.ext_abs0 = e->hits_so_far; src.ext_n = n_draws;
        e->draws_up = e->hits_so_far; e->draws_valid = e->hits_so_far + n_draws;
        e->ring_carry = 0;                                              // (the bits below draws_up in its word belong to ranks nobody reads again)
        e->mt_live = false; e->mt_on_device = false;
    } else if (!e->mt_live) {
        // (re)position the engine-owned stream: rank a takes draw number a - mt_abs0, and mt_hits draws are spent
        if (retire_all(e)) return 1;
        HIP_OK(hipStreamSynchronize(e->s_copy));
        fastf_mt_seed(&e->mt, e->mt_seed0);
        fastf_mt_skip(&e->mt, e->mt_skip0 + e->mt_hits);
        e->mt_abs0 = e->hits_so_far - e->mt_hits;
        e->draws_up = e->hits_so_far; e->draws_valid = e->hits_so_far;
        e->ring_carry = 0;
        e->mt_live = true;
        // from here on the stream continues on the device (FASTF_HOST_DRAWS=1: on the host, 4 bytes per hit over PCIe)
        e->mt_on_device = false;
        if (device_mt_wanted()) {
            static_assert(sizeof(fastf_mt_t) == (MT_N + 1) * 4, "fastf_mt_t: 624 state words + the read index");
            HIP_OK(hipStreamSynchronize(e->s_mt));                          // (idle already: every chunk that waited for it is retired)
            // (the generator's scratch for the most ranks one launch can be asked for — everything the ring holds — so that no
            //  push stops to grow it)
            if (e->d_mt.ensure(sizeof(fastf_mt_t)) || e->d_mtwords.ensure(e->ring_len * 4)) return 1;
            if (copy_h2d(e->d_mt.p, &e->mt, sizeof(fastf_mt_t))) return 1;
            e->mt_dev_idx = (u32)e->mt.idx;
            e->mt_on_device = true;
        }
    }
    size_t off = 0;
    while (off < batch->n) {
        const size_t n = std::min<size_t>(batch->n - off, e->batch_cap);
        if (push_chunk(e, batch, off, n, src, pinned)) return 1;
        off += n;
    }
    if (draws) {                                                        // the caller's array must not be needed after return
        if (retire_all(e)) return 1;
        HIP_OK(hipStreamSynchronize(e->s_copy));
    } else if (!pinned) {
        // the batch may be reused as soon as the call returns: every chunk was staged, nothing left to wait for
    }
    return 0;
}

// test hook: the device's MT19937 (mt_fill_kernel) from init_genrand(seed) advanced by `skip` draws on the host, continued by
// n_calls launches of counts[i] draws each; all draws, in order, into out (host memory, sum of counts words)
extern "C" int fastf_debug_mt_fill(int device, uint32_t seed, uint64_t skip, const uint64_t* counts, uint32_t n_calls, uint32_t* out) FASTF_TRY {
    HIP_OK(hipSetDevice(device));
    fastf_mt_t mt; fastf_mt_seed(&mt, seed); fastf_mt_skip(&mt, skip);
    u64 total = 0;
    for (u32 i = 0; i < n_calls; ++i) total += counts[i];
    DevBuf st, buf;
    int rc = 0;
    do {
        if (st.ensure(sizeof mt) || buf.ensure(std::max<u64>(total, 1) * 4)) { rc = 1; break; }
        if (copy_h2d(st.p, &mt, sizeof mt)) { rc = 1; break; }
        u64 at = 0;
        for (u32 i = 0; i < n_calls; ++i) {
            hipLaunchKernelGGL(mt_fill_kernel, dim3(1), dim3(256), 0, (hipStream_t)0, (u32*)st.p, (u32*)buf.p, at, counts[i], ~0ull);
            at += counts[i];
        }
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rc = set_err("mt_fill_kernel failed"); break; }
        if (total && copy_d2h(out, buf.p, total * 4)) { rc = 1; break; }
    } while (0);
    st.release(); buf.release();
    return rc;
} FASTF_CATCH_INT

// test hook: what the product does with that kernel (launch_mt_decisions) — the DECISIONS (draw < threshold) of the stream from init_genrand(seed)
// advanced by `skip` draws, n_calls launches of counts[i] draws each, the first of them at absolute rank `first`, into a ring of
// ring_bits bits (a power of two >= 64) that starts out as 0xFF bytes; the ring's words into out (ring_bits / 32 words)
extern "C" int fastf_debug_mt_fill_bits(int device, uint32_t seed, uint64_t skip, uint64_t first, const uint64_t* counts, uint32_t n_calls,
                                        uint64_t threshold, uint64_t ring_bits, uint32_t* out) FASTF_TRY {
    if (ring_bits < 64 || (ring_bits & (ring_bits - 1))) return set_err("ring_bits must be a power of two >= 64");
    HIP_OK(hipSetDevice(device));
    fastf_mt_t mt; fastf_mt_seed(&mt, seed); fastf_mt_skip(&mt, skip);
    DevBuf st, buf, words;
    int rc = 0;
    do {
        if (st.ensure(sizeof mt) || buf.ensure(ring_bits / 8)) { rc = 1; break; }
        if (copy_h2d(st.p, &mt, sizeof mt) || hipMemset(buf.p, 0xFF, ring_bits / 8) != hipSuccess) { rc = set_err("copy failed"); break; }
        u64 at = first;
        for (u32 i = 0; i < n_calls; ++i) {
            if (launch_mt_decisions((hipStream_t)0, (u32*)st.p, words, (u32*)buf.p, at, counts[i], ring_bits - 1, threshold)) { rc = 1; break; }
            at += counts[i];
        }
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { rc = set_err("mt_fill_kernel failed"); break; }
        if (copy_d2h(out, buf.p, ring_bits / 8)) { rc = 1; break; }
    } while (0);
    st.release(); buf.release(); words.release();
    return rc;
} FASTF_CATCH_INT

extern "C" int fastf_engine_push(fastf_engine_t* e, const fastf_batch_t* batch) FASTF_TRY {
    return push_impl(e, batch, nullptr, 0, false);
} FASTF_CATCH_INT

extern "C" int fastf_engine_push_draws(fastf_engine_t* e, const fastf_batch_t* batch, const uint32_t* draws, size_t n_draws) FASTF_TRY {
    if (!draws && batch && batch->n) return set_err("null draws");
    return push_impl(e, batch, draws, n_draws, false);
} FASTF_CATCH_INT

extern "C" int fastf_engine_push_pinned(fastf_engine_t* e, const fastf_batch_t* batch) FASTF_TRY {
    if (batch && batch->n && (debug_known_memory(batch->cb_key, batch->n * 8, "fastf_engine_push_pinned cb_key") || debug_known_memory(batch->gx_key, batch->n * 8, "fastf_engine_push_pinned gx_key") ||
                              debug_known_memory(batch->umi, batch->n * 4, "fastf_engine_push_pinned umi") || debug_known_memory(batch->meta, batch->n * 4, "fastf_engine_push_pinned meta")))
        return 1;
    return push_impl(e, batch, nullptr, 0, true);
} FASTF_CATCH_INT

extern "C" int fastf_engine_wait_input(fastf_engine_t* e) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return 0;                  // a multi-device push returns after its rounds have consumed the input
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipStreamSynchronize(e->s_copy));
    return 0;
} FASTF_CATCH_INT

extern "C" void* fastf_pinned_alloc(size_t bytes) FASTF_TRY {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc of %zu bytes failed", bytes); return nullptr; }
    return p;
} FASTF_CATCH_ZERO
extern "C" void fastf_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
extern "C" int fastf_pinned_register(void* p, size_t bytes) FASTF_TRY {
    if (pin_reg(p, bytes)) return set_err("hipHostRegister of %zu bytes at %p failed", bytes, p);
    return 0;
} FASTF_CATCH_INT
extern "C" void fastf_pinned_unregister(void* p) { pin_unreg(p); }
extern "C" int fastf_debug_live_registrations(void) { return fastf_pin_ledger_live(); }

// Engines with sub-groups (umi_max_bases > 24): K3's rows are one per (cell, feature, first bases of the UMI), the sorted word split
// into two halves at row_split, ascending.  One row per (cell, feature), the counts summed: UMIs of different sub-groups differ.
static u64 merge_sub_rows(fastf_engine* e, u64 nnz) {
    u32 *h_f = e->rows_at, *h_c = e->rows_at + e->rows_stride, *h_k = e->rows_at + 2 * e->rows_stride;
    const u32 fmask = (u32)((1ull << e->feat_bits) - 1);
    u64 out = 0, prev = ~0ull;
    for (u64 i = 0; i < nnz; ++i) {
        const u64 g = ((((u64)h_c[i]) << e->row_split) | h_f[i]) >> e->sub_bits;          // cell << feat_bits | feature
        if (g == prev) { h_k[out - 1] += h_k[i]; continue; }
        prev = g;
        h_f[out] = (u32)g & fmask; h_c[out] = (u32)(g >> e->feat_bits); h_k[out] = h_k[i];
        ++out;
    }
    return out;
}

// The device work of fastf_engine_finish up to the read-back of the counters, queued and not waited for (the multi-device
// engine queues it on every device before it waits for any)
static int finish_queue(fastf_engine* e) {
    hipStream_t s = e->s_compute;
    u64* small = (u64*)e->d_small.p;
    const u64 n = e->keys_so_far;
    if (n) {
        if (e->d_tmp.ensure(e->key_cap * sizeof(u64))) return 1;
        if (e->d_feature.ensure(n * 4) || e->d_cell.ensure(n * 4) || e->d_count.ensure(n * 4)) return 1;
        // the matrix only needs equal keys to be neighbours inside a (cell, feature) group: the lowest digit
        // passes are skipped and K3 resolves the short unsorted runs (fastf_engine_umi_rows sorts fully)
        if (e->wide) {
            // keys wider than 64 bits: the group words are sorted (the rest of each key moves with its group word), K3 finds
            // the distinct rest-of-keys of a group in its window set
            if (e->d_vtmp.ensure(e->key_cap * sizeof(u64))) return 1;
            if (launch_sort(e, (u64*)e->d_keys.p, (u64*)e->d_tmp.p, small + SM_KEYCOUNT, n, e->group_bits, 0, &e->sorted_in_tmp, s, false,
                            (u64*)e->d_vals.p, (u64*)e->d_vtmp.p))
                return 1;
            e->fully_sorted = false;
            if (launch_reduce_regions<false>(e, e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p, small + SM_KEYCOUNT, n, small + SM_NNZ, 0, s,
                                             e->sorted_in_tmp ? (u64*)e->d_vtmp.p : (u64*)e->d_vals.p))
                return 1;
        } else {
        if (e->stream_mode && e->store_regions) {
            // the keys lie in the regions the chunks' K1b left, each with the histogram of the first digit: the first pass
            // walks the regions (no counting pass), the later ones see contiguous buffers
            e->seg_n = e->rgn_n; e->seg_stride = 0; e->rgn_hist_n = e->rgn_n; e->rgn_hist_shift = e->skip_bits;
            if (launch_sort(e, (u64*)e->d_keys.p, (u64*)e->d_tmp.p, small + SM_KEYCOUNT, n, e->L.total_bits, e->skip_bits,
                            &e->sorted_in_tmp, s, true))
                return 1;
            if (sort_passes(e->L.total_bits, e->skip_bits) >= 2) e->store_regions = false;   // the second pass wrote the store from its front
        } else
        if (launch_sort(e, (u64*)e->d_keys.p, (u64*)e->d_tmp.p, small + SM_KEYCOUNT, n, e->L.total_bits, e->skip_bits,
                        &e->sorted_in_tmp, s))
            return 1;
        e->fully_sorted = e->skip_bits == 0;
        const u64* sorted = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p;
        // the rows stay in K3's regions until the host buffer is known to be large enough (the gather below)
        if (launch_reduce_regions<false>(e, sorted, small + SM_KEYCOUNT, n, small + SM_NNZ, e->skip_bits, s)) return 1;
        }
    } else {
        HIP_OK(hipMemsetAsync(small + SM_NNZ, 0, sizeof(u64), s));
    }
    HIP_OK(hipMemcpyAsync(e->h_small, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, s));
    e->finish_queued = true;
    return 0;
}

extern "C" int fastf_engine_finish(fastf_engine_t* e, fastf_coo_t* coo, uint64_t counters[3]) FASTF_TRY {
    if (!e || !coo) return set_err("null argument");
    if (e->multi) return multi_finish(e, coo, counters);
    HIP_OK(hipSetDevice(e->device));
    const bool prof = getenv("FASTF_PROFILE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (prof) fprintf(stderr, "[finish] %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_0).count()); };
    if (retire_all(e)) return 1;
    lap("chunks retired");
    hipStream_t s = e->s_compute;
    u64* small = (u64*)e->d_small.p;
    const u64 n = e->keys_so_far;
    if (!e->finished) {
        if (!e->finish_queued && finish_queue(e)) return 1;
        e->finish_queued = false;
        HIP_OK(hipStreamSynchronize(s));
        if (n && e->wide && (e->h_small[SM_COUNTERS + 3] & ERR_RUN_TOO_LONG))
            return set_err("a (cell, feature) group of more than 4 M reads (or more than 6 M reads in groups beyond 2048) with keys wider than 64 bits: not supported");
        if (n && (e->h_small[SM_COUNTERS + 3] & ERR_RUN_TOO_LONG)) {
            // runs of keys that agree on the sorted bits are long in this data (very deep (cell, feature) groups):
            // finish the sort and reduce exactly; every key is still in the store, permuted
            u64* from = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p;
            u64* other = e->sorted_in_tmp ? (u64*)e->d_keys.p : (u64*)e->d_tmp.p;
            int in_other = 0;
            const u64 keep = e->h_small[SM_COUNTERS + 3] & ~ERR_RUN_TOO_LONG;
            if (copy_h2d_on(small + SM_COUNTERS + 3, &keep, sizeof(u64), s)) return 1;
            if (launch_sort(e, from, other, small + SM_KEYCOUNT, n, e->L.total_bits, 0, &in_other, s)) return 1;
            if (in_other) e->sorted_in_tmp = !e->sorted_in_tmp;
            e->fully_sorted = true; e->store_regions = false;
            const u64* sorted = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p;
            if (launch_reduce_regions<false>(e, sorted, small + SM_KEYCOUNT, n, small + SM_NNZ, 0, s)) return 1;
            HIP_OK(hipMemcpyAsync(e->h_small, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
        }
        lap("sorted and reduced");
        if (e->h_small[SM_COUNTERS + 3]) return set_err("%s", err_bits_text(e->h_small[SM_COUNTERS + 3]));
        const u64 nnz = e->h_small[SM_NNZ];
        if (nnz > n) return set_err("internal error: %llu matrix rows out of %llu keys", (unsigned long long)nnz, (unsigned long long)n);
        // The row buffer is ordinary memory the first time it is used: pinning 23 MB takes 15 ms, the pageable copy of
        // 2 M rows 2 ms, and a one-shot run (the CLI) never uses it twice.  An engine that finishes again (reset + push:
        // benchmarks, sharded passes) pins it then and gets its rows at PCIe rate from there on.
        const bool on_loan = e->lent_rows && nnz <= e->lent_cap;
        if (on_loan) {
            // pinned memory the caller lent (bam2db(): the decoder's slab, idle by now): no allocation, no first-touch page
            // faults, and the gather kernel writes it directly
            e->rows_at = e->lent_rows; e->rows_stride = e->lent_cap;
        } else {
            if (nnz > e->h_coo_cap) {
                if (e->h_coo) { if (e->h_coo_pinned) pin_unreg(e->h_coo); fastf_big_free(e->h_coo, (size_t)e->h_coo_cap * 12); }
                e->h_coo = nullptr; e->h_coo_cap = 0; e->h_coo_pinned = false; e->h_coo_uses = 0;
                const u64 cap = nnz + nnz / 8 + 1024;
                void* m = fastf_big_alloc((size_t)cap * 12);               // 2 MiB aligned, transparent huge pages (host_io.c)
                if (!m) return set_err("out of memory (%llu matrix rows)", (unsigned long long)nnz);
                e->h_coo = (u32*)m; e->h_coo_cap = cap;
            }
            if (!e->h_coo_pinned && e->h_coo && e->h_coo_uses++ >= 1 && pin_reg(e->h_coo, (size_t)e->h_coo_cap * 12) == 0)
                e->h_coo_pinned = true;
            e->rows_at = e->h_coo; e->rows_stride = e->h_coo_cap;
        }
        lap("row buffer ready");
        if (nnz) {
            u32 *h_f = e->rows_at, *h_c = e->rows_at + e->rows_stride, *h_k = e->rows_at + 2 * e->rows_stride;
            if (on_loan || e->h_coo_pinned) {
                // pinned row buffer: the gather of K3's row regions writes it directly — that kernel is the D2H copy
                if (launch_rows_gather<false>(e, small + SM_KEYCOUNT, h_f, h_c, h_k, nullptr, s)) return 1;
            } else {
                if (launch_rows_gather<false>(e, small + SM_KEYCOUNT, (u32*)e->d_feature.p, (u32*)e->d_cell.p, (u32*)e->d_count.p, nullptr, s)) return 1;
                HIP_OK(hipStreamSynchronize(s));               // (the gather; the row buffer is ordinary memory: through the bounce buffer)
                if (copy_d2h(h_f, e->d_feature.p, nnz * 4) || copy_d2h(h_c, e->d_cell.p, nnz * 4) || copy_d2h(h_k, e->d_count.p, nnz * 4)) return 1;
            }
            HIP_OK(hipStreamSynchronize(s));
        }
        lap("rows on the host");
        e->h_nnz = e->sub_bits ? merge_sub_rows(e, nnz) : nnz;
        e->n_sorted = n;
        e->finished = true;
    }
    coo->feature = e->rows_at; coo->cell = e->rows_at ? e->rows_at + e->rows_stride : nullptr; coo->count = e->rows_at ? e->rows_at + 2 * e->rows_stride : nullptr;
    coo->nnz = e->h_nnz;
    if (counters) { counters[0] = e->total_records; counters[1] = e->c_sampled; counters[2] = e->c_valid; }
    return 0;
} FASTF_CATCH_INT

// -u rows with keys wider than 64 bits: sort the pairs fully (by the rest of the key, then by the group word: two stable LSD
// sorts), one row per distinct pair (pair_*_kernel)
static int umi_rows_wide(fastf_engine* e, fastf_umi_rows_t* rows) {
    hipStream_t s = e->s_compute;
    u64* small = (u64*)e->d_small.p;
    const u64 n = e->n_sorted;
    u64 nrows = 0;
    std::vector<u64> uk, uv;
    if (n) {
        u64 *kc = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p, *ko = e->sorted_in_tmp ? (u64*)e->d_keys.p : (u64*)e->d_tmp.p;
        u64 *vc = e->sorted_in_tmp ? (u64*)e->d_vtmp.p : (u64*)e->d_vals.p, *vo = e->sorted_in_tmp ? (u64*)e->d_vals.p : (u64*)e->d_vtmp.p;
        if (!e->fully_sorted) {
            int f = 0;
            if (launch_sort(e, vc, vo, small + SM_KEYCOUNT, n, e->L.feat_shift, 0, &f, s, false, kc, ko)) return 1;   // by the rest of the key (the group words ride along)
            if (f) { std::swap(kc, ko); std::swap(vc, vo); }
            if (launch_sort(e, kc, ko, small + SM_KEYCOUNT, n, e->group_bits, 0, &f, s, false, vc, vo)) return 1;      // then, stably, by the group word
            if (f) { std::swap(kc, ko); std::swap(vc, vo); }
            e->sorted_in_tmp = kc == (u64*)e->d_tmp.p;
            e->fully_sorted = true;
        }
        const u64 T = (n + UW_TILE - 1) / UW_TILE;
        // rows: the group words into d_ukeys, the rests and the head positions into two halves of one scratch buffer
        if (e->d_ukeys.ensure(n * 8) || e->d_ncopy.ensure(n * 4) || e->d_rg_ukeys.ensure(2 * n * 8) || e->d_scanblk.ensure(2 * 64 * sizeof(u64))) return 1;
        DevBuf cnt, base;
        if (cnt.ensure((T + 16) * 4) || base.ensure((T + 16) * 8)) { cnt.release(); base.release(); return 1; }
        int rc = 0;
        do {
            if (hipMemsetAsync(cnt.p, 0, (T + 16) * 4, s) != hipSuccess) { rc = set_err("memset failed"); break; }
            const u32 grid = (u32)std::min<u64>(T, 16ull * g_cu_count);
            hipLaunchKernelGGL(pair_heads_kernel, dim3(grid), dim3(UW_THREADS), 0, s, (const u64*)kc, (const u64*)vc, (const u64*)(small + SM_KEYCOUNT), (u32*)cnt.p);
            launch_scan_tiles(e, 1, s, (u32*)cnt.p, (u64*)base.p, (u32)T, small + SM_NROWS_U, nullptr, nullptr);
            u64* uvals = (u64*)e->d_rg_ukeys.p; u64* hpos = uvals + n;
            hipLaunchKernelGGL(pair_rows_kernel, dim3(grid), dim3(UW_THREADS), 0, s, (const u64*)kc, (const u64*)vc, (const u64*)(small + SM_KEYCOUNT),
                               (const u64*)base.p, (u64*)e->d_ukeys.p, uvals, hpos);
            hipLaunchKernelGGL(pair_copies_kernel, dim3(grid), dim3(UW_THREADS), 0, s, (const u64*)hpos, (const u64*)(small + SM_NROWS_U),
                               (const u64*)(small + SM_KEYCOUNT), (u32*)e->d_ncopy.p);
            if (hipGetLastError() != hipSuccess) { rc = set_err("kernel launch failed (-u rows of wide keys)"); break; }
            if (hipMemcpyAsync(e->h_small, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = set_err("device error in the -u rows of wide keys"); break; }
            if (e->h_small[SM_COUNTERS + 3]) { rc = set_err("%s", err_bits_text(e->h_small[SM_COUNTERS + 3])); break; }
            nrows = e->h_small[SM_NROWS_U];
            if (nrows > n) { rc = set_err("internal error: %llu -u rows out of %llu keys", (unsigned long long)nrows, (unsigned long long)n); break; }
            uk.resize(nrows); uv.resize(nrows); e->h_ncopy.resize(nrows);
            if (nrows) {
                if (copy_d2h(uk.data(), e->d_ukeys.p, nrows * 8) || copy_d2h(uv.data(), uvals, nrows * 8) || copy_d2h(e->h_ncopy.data(), e->d_ncopy.p, nrows * 4)) { rc = 1; break; }
            }
        } while (0);
        cnt.release(); base.release();
        if (rc) return 1;
    } else {
        e->h_ncopy.clear();
    }
    e->h_ufeature.resize(nrows); e->h_ucell.resize(nrows); e->h_uumi.resize(nrows); e->h_unonnull.resize(nrows);
    const u32 fmask = (u32)((1ull << e->feat_bits) - 1);
    const u64 umask = e->L.umi_bits >= 64 ? ~0ull : ((1ull << e->L.umi_bits) - 1);
    for (u64 i = 0; i < nrows; ++i) {
        if (e->sub_bits) {           // (cell, feature, the UMI's first bases) in the sorted word: rows in blob order as they are
            e->h_ufeature[i] = (u32)(uk[i] >> e->sub_bits) & fmask;
            e->h_ucell[i] = (u32)(uk[i] >> (e->feat_bits + e->sub_bits));
            e->h_unonnull[i] = (uint8_t)((uv[i] >> WIDE_SUB_VAL_BITS) & 1);
            const u64 bases = ((uk[i] & ((1ull << e->sub_bits) - 1)) << (WIDE_SUB_VAL_BITS - WIDE_SUB_LEN_BITS)) | ((uv[i] & ((1ull << WIDE_SUB_VAL_BITS) - 1)) >> WIDE_SUB_LEN_BITS);
            e->h_uumi[i] = (u32)(bases >> 32);
            continue;
        }
        e->h_ufeature[i] = (u32)uk[i] & fmask;
        e->h_ucell[i] = (u32)(uk[i] >> e->feat_bits);
        e->h_unonnull[i] = (uint8_t)((uv[i] >> (e->L.umi_bits + e->L.len_bits)) & 1);
        const u64 field = (uv[i] >> e->L.len_bits) & umask;              // the first 16 bases are what fastf_umi_rows_t carries
        e->h_uumi[i] = e->L.umi_bits > 32 ? (u32)(field >> (e->L.umi_bits - 32)) : (u32)(field << (32 - e->L.umi_bits));
    }
    rows->feature = e->h_ufeature.data(); rows->cell = e->h_ucell.data(); rows->n_copy = e->h_ncopy.data();
    rows->umi = e->h_uumi.data(); rows->nonnull = e->h_unonnull.data(); rows->n = nrows;
    return 0;
}

extern "C" int fastf_engine_umi_rows(fastf_engine_t* e, fastf_umi_rows_t* rows) FASTF_TRY {
    if (!e || !rows) return set_err("null argument");
    if (e->multi) return multi_umi_rows(e, rows);
    if (!e->finished) return set_err("call fastf_engine_finish first");
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = e->s_compute;
    u64* small = (u64*)e->d_small.p;
    const u64 n = e->n_sorted;
    u64 nrows = 0;
    if (e->wide) return umi_rows_wide(e, rows);
    std::vector<u64> ukeys;
    if (n) {
        if (e->d_ukeys.ensure(n * 8) || e->d_ncopy.ensure(n * 4)) return 1;
        if (!e->fully_sorted) {            // -u rows are ordered by blob: finish the sort (every key is still there, permuted)
            u64* from = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p;
            u64* other = e->sorted_in_tmp ? (u64*)e->d_keys.p : (u64*)e->d_tmp.p;
            int in_other = 0;
            if (launch_sort(e, from, other, small + SM_KEYCOUNT, n, e->L.total_bits, 0, &in_other, s)) return 1;
            if (in_other) e->sorted_in_tmp = !e->sorted_in_tmp;
            e->fully_sorted = true; e->store_regions = false;
        }
        const u64* sorted = e->sorted_in_tmp ? (u64*)e->d_tmp.p : (u64*)e->d_keys.p;
        if (launch_reduce<true>(e, sorted, small + SM_KEYCOUNT, n, nullptr, nullptr, (u32*)e->d_ncopy.p,
                                (u64*)e->d_ukeys.p, small + SM_NROWS_U, 0, s))
            return 1;
        HIP_OK(hipMemcpyAsync(e->h_small, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        if (e->h_small[SM_COUNTERS + 3]) return set_err("%s", err_bits_text(e->h_small[SM_COUNTERS + 3]));
        nrows = e->h_small[SM_NROWS_U];
        if (nrows > n) return set_err("internal error: %llu -u rows out of %llu keys", (unsigned long long)nrows, (unsigned long long)n);
        ukeys.resize(nrows);
        e->h_ncopy.resize(nrows);
        if (nrows) {
            if (copy_d2h(ukeys.data(), e->d_ukeys.p, nrows * 8) || copy_d2h(e->h_ncopy.data(), e->d_ncopy.p, nrows * 4)) return 1;
        }
    } else {
        e->h_ncopy.clear();
    }
    e->h_ufeature.resize(nrows); e->h_ucell.resize(nrows); e->h_uumi.resize(nrows); e->h_unonnull.resize(nrows);
    const u32 fmask = (u32)((1ull << e->feat_bits) - 1);
    const u64 umask = e->L.umi_bits >= 64 ? ~0ull : ((1ull << e->L.umi_bits) - 1);
    for (u64 i = 0; i < nrows; ++i) {
        const u64 k = ukeys[i];
        e->h_ufeature[i] = (u32)(k >> e->L.feat_shift) & fmask;
        e->h_ucell[i] = (u32)(k >> e->L.cell_shift);
        e->h_unonnull[i] = (uint8_t)((k >> (e->L.umi_bits + e->L.len_bits)) & 1);
        const u64 field = (k >> e->L.len_bits) & umask;
        e->h_uumi[i] = (u32)(field << (32 - e->L.umi_bits));
    }
    rows->feature = e->h_ufeature.data(); rows->cell = e->h_ucell.data(); rows->n_copy = e->h_ncopy.data();
    rows->umi = e->h_uumi.data(); rows->nonnull = e->h_unonnull.data(); rows->n = nrows;
    return 0;
} FASTF_CATCH_INT

// records each device of a multi-device engine has been given since the last reset (a single-device engine: its total)
extern "C" int fastf_engine_device_records(const fastf_engine_t* e, uint64_t* records, uint32_t n) FASTF_TRY {
    if (!e || !records) return set_err("null argument");
    return multi_device_records(e, records, n);
} FASTF_CATCH_INT

extern "C" int fastf_dev_adopt_wide(fastf_engine_t* e, const uint64_t* d_keys, const uint64_t* d_vals, uint64_t n, void* stream) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return set_err("fastf_dev_adopt_wide: device-level calls take a single-device engine");
    if (!e->wide) return set_err("fastf_dev_adopt_wide: this engine's keys fit 64 bits");
    if (n && (!d_keys || !d_vals)) return set_err("null keys or values");
    HIP_OK(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (grow_keys(e, std::max<u64>(n, 1))) return 1;
    if (n) {
        HIP_OK(hipMemcpyAsync(e->d_keys.p, d_keys, n * sizeof(u64), hipMemcpyDeviceToDevice, s));
        HIP_OK(hipMemcpyAsync(e->d_vals.p, d_vals, n * sizeof(u64), hipMemcpyDeviceToDevice, s));
    }
    const u64 cnt = n;
    if (copy_h2d_on((u64*)e->d_small.p + SM_KEYCOUNT, &cnt, sizeof(u64), s)) return 1;
    HIP_OK(hipStreamSynchronize(s));                       // (fastf_engine_finish works on the engine's own stream; cnt is a local)
    e->keys_so_far = n; e->finished = false; e->finish_queued = false; e->fully_sorted = false; e->sorted_in_tmp = 0;
    return 0;
} FASTF_CATCH_INT

extern "C" int fastf_engine_reset(fastf_engine_t* e) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return multi_reset(e, false, 0, 0);
    HIP_OK(hipSetDevice(e->device));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemset(e->d_small.p, 0, SM_WORDS * sizeof(u64)));
    memset(e->h_small, 0, SM_WORDS * sizeof(u64));
    for (auto& sl : e->slot) sl.busy = false;
    e->inflight_records = 0;
    e->total_records = e->hits_so_far = e->keys_so_far = e->c_sampled = e->c_valid = 0;
    e->finished = false; e->finish_queued = false; e->h_nnz = 0;
    e->slots_used = 0; e->rgn_n = 0; e->store_regions = false; e->rgn_hist_n = 0; e->seg_n = 0;
    e->lent_rows = nullptr; e->lent_cap = 0; e->rows_at = nullptr; e->rows_stride = 0;     // a loan lasts for one finish
    e->draws_up = e->draws_valid = 0;
    e->mt_live = false;                  // the engine-owned stream goes on after the last draw a hit consumed
    return 0;
} FASTF_CATCH_INT

// Pinned host memory of the caller's for the rows of the next fastf_engine_finish: three arrays of bytes / 12 rows each.
// finish() uses it when the matrix fits (and its own row buffer otherwise); the coo pointers then point into it.
extern "C" int fastf_engine_lend_rows(fastf_engine_t* e, void* pinned, size_t bytes) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return 0;                                  // the multi-device merge writes its own buffer
    if (e->finished) return set_err("fastf_engine_lend_rows: the rows of this pass are out already (reset first)");
    if (!pinned || bytes < 12 || ((uintptr_t)pinned & 3u)) { e->lent_rows = nullptr; e->lent_cap = 0; return pinned ? set_err("fastf_engine_lend_rows: needs 4-byte aligned memory of at least 12 bytes") : 0; }
    if (debug_known_memory(pinned, bytes, "fastf_engine_lend_rows")) return 1;
    e->lent_rows = (u32*)pinned; e->lent_cap = bytes / 12;
    return 0;
} FASTF_CATCH_INT

// re-position the engine-owned draw stream (tests; bam2db() sets it through the config)
extern "C" int fastf_engine_reseed(fastf_engine_t* e, uint32_t seed, uint64_t skip) FASTF_TRY {
    if (!e) return set_err("null engine");
    if (e->multi) return multi_reset(e, true, seed, skip);
    HIP_OK(hipSetDevice(e->device));
    if (retire_all(e)) return 1;
    e->mt_seed0 = seed; e->mt_skip0 = skip; e->mt_hits = 0;
    e->mt_live = false;
    return 0;
} FASTF_CATCH_INT

// ------------------------------------------------------------------------------------
// tag histogram (crb / extract): same translation unit, reuses K2 and K3u
// ------------------------------------------------------------------------------------
#include "tag_hist.hpp"

// ------------------------------------------------------------------------------------
// one process, several devices: same translation unit, drives sub-engines through the launch_* functions above
// ------------------------------------------------------------------------------------
#include "multi_engine.hpp"

// ------------------------------------------------------------------------------------
// device side of the BAM front end (BGZF inflate of reader windows)
// ------------------------------------------------------------------------------------
#include "gpu_frontend.hpp"
