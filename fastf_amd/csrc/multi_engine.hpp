// multi_engine.hpp — one host process driving G devices behind the same fastf_engine_* calls
// (fastf_engine_config_t.n_devices > 1; bam2db() reads FASTF_DEVICES).  Included by umi_engine.hip.
//
// The path shards by cell: GROUP BY cell_index, feature_index (bam2db_ds.c:480-483) never joins rows of
// different cells, so every key goes to the device that owns its cell (murmur(cell_index) mod G, shard_of()).
//
//   push     chunks of the record stream are dealt round-robin to the devices — chunk i of the STREAM goes to device
//            i mod G, across push calls (a caller that pushes one batch at a time still feeds every device).  A chunk is
//              1. staged (pageable input only), copied (s_copy) and counted: K1a + scan on s_compute, the hit count and the
//                 device's key counts come back into a pinned snapshot, an event marks it               [enqueue_chunk]
//              2. retired, in stream order, once its count is in: the host adds the counts up into hit-rank bases (a
//                 draw is consumed per CB hit in record order, bam2db_ds.c:385), generates exactly the chunk's draws, sends
//                 them and queues K1b: keys land in G per-destination buffers on the device that filtered them [retire_chunk]
//            Nothing waits for a round: the host blocks only on the count of the OLDEST chunk in flight, and only when the
//            device it lives on is needed again (or at finish), so the copy of chunk i+G overlaps K1b of chunk i and the
//            other devices' work.  Stage slots are double-buffered per device; a slot is refilled behind the K1b that read
//            it (event per device and slot).  The per-destination buffers grow ahead of need from upper bounds, on the
//            stream, and the old buffer is kept until the streams are known to be idle — no hipMalloc/hipFree stall.
//   finish   ONE exchange: buffer h of device g -> device h (RCCL send/recv in one group over xGMI, or peer copies
//            when devices alias — the one-GPU rehearsal — or FASTF_EXCHANGE=peer), then K2 + K3 locally on every
//            device, rows back, and a merge of the G row lists by cell (each cell lives on exactly one device).
// Nothing but the keys crosses devices; the three counters are added up on the host.
// Keys wider than 64 bits: K1b (tile form) writes the group word and the rest of the key into two arrays per
// destination, both cross in the same exchange and land in the RECEIVER's own wide key store — from there on the sub-engine
// does what the single-device engine does (finish_queue on every device, then fastf_engine_finish / _umi_rows collect).
#include <dlfcn.h>
#include <rccl/rccl.h>

// rows on the host: three arrays of `cap` u32 in one allocation of 2 MiB-aligned, huge-page backed memory (fastf_big_alloc: no
// zero fill, a page fault per 2 MiB) — what the single-device engine's own row buffer is
struct HostRows {
    u32* p = nullptr; u64 cap = 0;
    int ensure(u64 n) {
        if (n <= cap) return 0;
        release();
        const u64 c = n + n / 8 + 1024;
        p = (u32*)fastf_big_alloc((size_t)c * 12);
        if (!p) return set_err("out of memory (%llu matrix rows)", (unsigned long long)n);
        cap = c;
        return 0;
    }
    void release() { if (p) fastf_big_free(p, (size_t)cap * 12); p = nullptr; cap = 0; }
    u32* f() const { return p; } u32* c() const { return p + cap; } u32* k() const { return p + 2 * cap; }
};

struct MultiDev {
    fastf_engine* e = nullptr;                 // sub-engine on this device: tables, workspace, streams (n_shards = G, shard_rank = g)
    int dev = 0;
    void* h_stage[2] = {nullptr, nullptr};     // pinned chunk staging (pageable input only)
    DevBuf d_stage[2];
    // per stage slot: copy done / K1a counted and the snapshot is on the host / K1b done (the slot may be refilled)
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_cnt[2] = {nullptr, nullptr}, ev_k1b[2] = {nullptr, nullptr}, ev_sent = nullptr;
    bool k1b_queued[2] = {false, false};
    u32* h_draws[2] = {nullptr, nullptr}; DevBuf d_draws[2];     // FASTF_HOST_DRAWS=1: the draws of the chunk in that slot, generated on the host
    // the draw stream on the device: EVERY device continues the one MT19937 stream by the hits of EVERY chunk, in stream order
    // (mt_fill_kernel on its compute stream: a few hundred microseconds per chunk), into a ring of decisions addressed by absolute hit rank,
    // so the draws of its own chunks are there when its K1b runs and nothing about them crosses PCIe or waits for a host loop
    DevBuf d_mt, d_ring, d_mtwords; u64 ring_len = 0;
    u32 mt_idx = MT_N;                         // where in its 624-word block this device's copy of the stream stands (launch_mt_decisions_par)
    u64* h_base[2] = {nullptr, nullptr};       // pinned: absolute hit-rank base of the chunk in that slot (copied beside the K1b launch)
    u64* h_cnt[2] = {nullptr, nullptr};        // pinned snapshot of the sub-engine's d_small behind that slot's K1a
    int next_slot = 0;
    u64* h_info = nullptr;                     // pinned mirror of the sub-engine's d_small (finish)
    DevBuf d_shard; u64 stride = 0;            // keys by destination: [G][stride] (wide keys: the values behind them, another [G][stride])
    std::vector<void*> old_shards;             // replaced while work was queued: freed when the streams are idle
    u64 cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // keys per destination, exact as of the last read-back
    u64 keys_exact = 0, recs_since = 0;        // most keys in one destination as of the last snapshot; records counted since
    u64 records = 0;                           // records this device has been given (tests: every device takes part)
    // receive side
    DevBuf d_recv, d_tmp, d_f, d_c, d_k, d_ukeys, d_ncopy;
    u64 n_recv = 0; int sorted_in_tmp = 0; bool fully_sorted = false;
    HostRows rows;                             // this shard's rows, ascending (cell, feature): pf/pc/pk[0 .. nnz) — in `rows`, or
    const u32 *pf = nullptr, *pc = nullptr, *pk = nullptr; u64 nnz = 0;     // (wide keys) the sub-engine's own row buffer
    std::vector<u64> ukeys; std::vector<u32> ncopy;
};

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct fastf_multi {
    u32 G = 0;
    std::vector<MultiDev> d;
    u64 cap = 0, chunks = 0;                                     // chunks dealt so far: chunk i lives on device i mod G
    struct InFlight { u32 dev; int slot; u64 n; };
    std::vector<InFlight> fifo;                                  // counted, not yet retired (stream order)
    fastf_mt_t mt{}; u32 mt_seed0 = 0; u64 mt_skip0 = 0;
    bool device_mt = false, mt_uploaded = false;                 // the stream continues on the devices (else: on the host, FASTF_HOST_DRAWS=1)
    u64 hits = 0, total_records = 0, c_sampled = 0, c_valid = 0;
    bool finished = false, aliased = false;
    bool keycount_lent = false;                                  // wide keys: the sub-engines' key-count words hold what they received (multi_finish_wide)
    int use_rccl = 0; RcclApi rccl; std::vector<ncclComm_t> comms;
    HostRows merged; u64 merged_n = 0;                           // merged rows
    std::vector<u32> ufeature, ucell, uumi, ncopy; std::vector<uint8_t> unonnull;
};

// librccl is loaded once per process and never unloaded: the library keeps threads, thread-local state and exit
// handlers of its own, and unmapping its code under them is not safe.
static int multi_load_rccl(fastf_multi* m) {
    static std::mutex mu;
    static RcclApi shared;
    std::lock_guard<std::mutex> lk(mu);
    if (!shared.lib) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return set_err("cannot load librccl.so (%s); FASTF_EXCHANGE=peer uses device-to-device copies instead", dlerror());
        RcclApi r;
        // (a library without one of these is of no use and has run nothing yet: it may be unloaded again)
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { dlclose(h); return set_err("librccl.so lacks %s", name); }
        SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        r.lib = h;
        shared = r;
    }
    m->rccl = shared;
    return 0;
}
#define NCCL_OK(m, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return set_err("%s failed: %s", #call, (m)->rccl.GetErrorString(r_)); } while (0)
// inside ncclGroupStart .. ncclGroupEnd: the group is closed before the error is reported
#define NCCL_OK_IN_GROUP(m, call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { (void)(m)->rccl.GroupEnd(); return set_err("%s failed: %s", #call, (m)->rccl.GetErrorString(r_)); } } while (0)

static void multi_free_old_shards(MultiDev& md);
static int multi_retire_all(fastf_multi* m);
static int multi_return_keycount(fastf_multi* m);
static void multi_destroy(fastf_engine* e) {
    fastf_multi* m = e->multi;
    if (!m) return;
    for (MultiDev& md : m->d) {                             // nothing of the exchange may still be queued when its communicator goes
        (void)hipSetDevice(md.dev);
        if (md.e) (void)hipStreamSynchronize(md.e->s_compute);
    }
    for (auto& c : m->comms) if (c && m->rccl.CommDestroy) (void)m->rccl.CommDestroy(c);
    for (MultiDev& md : m->d) {
        (void)hipSetDevice(md.dev);
        if (md.e) (void)hipStreamSynchronize(md.e->s_copy);
        for (int i = 0; i < 2; ++i) {
            if (md.h_stage[i]) (void)hipHostFree(md.h_stage[i]);
            md.d_stage[i].release(); md.d_draws[i].release();
            if (md.h_base[i]) (void)hipHostFree(md.h_base[i]);
            if (md.ev_in[i]) (void)hipEventDestroy(md.ev_in[i]);
            if (md.ev_cnt[i]) (void)hipEventDestroy(md.ev_cnt[i]);
            if (md.ev_k1b[i]) (void)hipEventDestroy(md.ev_k1b[i]);
            if (md.h_draws[i]) (void)hipHostFree(md.h_draws[i]);
            if (md.h_cnt[i]) (void)hipHostFree(md.h_cnt[i]);
        }
        if (md.ev_sent) (void)hipEventDestroy(md.ev_sent);
        if (md.h_info) (void)hipHostFree(md.h_info);
        multi_free_old_shards(md);
        DevBuf* all[] = {&md.d_shard, &md.d_recv, &md.d_tmp, &md.d_f, &md.d_c, &md.d_k, &md.d_ukeys, &md.d_ncopy, &md.d_mt, &md.d_ring, &md.d_mtwords};
        for (DevBuf* b : all) b->release();
        md.rows.release();
        if (md.e) fastf_engine_destroy(md.e);
    }
    m->merged.release();
    delete m;
    e->multi = nullptr;
}

static int multi_create(const fastf_engine_config_t* cfg, fastf_engine* e) {
    const u32 G = cfg->n_devices;
    if (G > 8) return set_err("n_devices %u: at most 8 devices (one node)", G);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    fastf_multi* m = new fastf_multi();
    e->multi = m;
    m->G = G; m->d.resize(G);
    m->cap = cfg->batch_records ? cfg->batch_records : (4ull << 20);
    m->mt_seed0 = cfg->mt_seed; m->mt_skip0 = cfg->mt_skip;
    fastf_mt_seed(&m->mt, cfg->mt_seed); fastf_mt_skip(&m->mt, cfg->mt_skip);
    m->device_mt = device_mt_wanted();
    for (u32 g = 0; g < G; ++g) {
        const int dev = cfg->devices ? cfg->devices[g] : (int)g;
        if (dev < 0 || dev >= ndev) return set_err("device %d out of range (have %d)", dev, ndev);
        for (u32 q = 0; q < g; ++q) if (m->d[q].dev == dev) m->aliased = true;
        m->d[g].dev = dev;
    }
    {   // the exchange: RCCL between distinct devices, device-to-device copies when devices repeat or on request
        const char* x = getenv("FASTF_EXCHANGE");
        m->use_rccl = x ? (strcmp(x, "rccl") == 0) : !m->aliased;
        if (m->use_rccl && m->aliased) return set_err("FASTF_EXCHANGE=rccl needs distinct devices");
        if (m->use_rccl) {
            // communicators first: when RCCL cannot be loaded or initialised and nobody insisted on it, the exchange falls
            // back to device-to-device copies (peer access is set up per device below)
            std::vector<int> devs(G);
            for (u32 g = 0; g < G; ++g) devs[g] = m->d[g].dev;
            m->comms.assign(G, nullptr);
            bool ok = multi_load_rccl(m) == 0;                       // (sets the error text itself)
            if (ok) {
                const ncclResult_t ir = m->rccl.CommInitAll(m->comms.data(), (int)G, devs.data());
                if (ir != ncclSuccess) {
                    ok = false;
                    set_err("ncclCommInitAll over %u devices failed: %s", G, m->rccl.GetErrorString(ir));
                    for (auto& c : m->comms) if (c) { (void)m->rccl.CommDestroy(c); c = nullptr; }     // what it had created
                }
            }
            if (!ok) {
                if (x) { const std::string why = fastf_last_error(); return set_err("FASTF_EXCHANGE=rccl: RCCL is not usable here (%s)", why.c_str()); }
                fprintf(stderr, "Warning: RCCL is not usable here (%s); the key exchange uses device-to-device copies\n", fastf_last_error());
                m->comms.clear();
                m->use_rccl = 0;
            }
        }
    }
    for (u32 g = 0; g < G; ++g) {
        MultiDev& md = m->d[g];
        fastf_engine_config_t sub = *cfg;
        sub.n_devices = 0; sub.devices = nullptr;
        sub.n_shards = G; sub.shard_rank = g; sub.device = md.dev;
        if (fastf_engine_create(&sub, &md.e)) return 1;
        HIP_OK(hipSetDevice(md.dev));
        HIP_OK(hipHostMalloc((void**)&md.h_info, SM_WORDS * sizeof(u64), hipHostMallocDefault));
        for (int i = 0; i < 2; ++i) {
            HIP_OK(hipEventCreateWithFlags(&md.ev_in[i], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&md.ev_cnt[i], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&md.ev_k1b[i], hipEventDisableTiming));
        }
        HIP_OK(hipEventCreateWithFlags(&md.ev_sent, hipEventDisableTiming));
        if (!m->use_rccl) {                             // peer copies: let the devices reach each other's memory
            for (u32 q = 0; q < G; ++q) {
                const int other = cfg->devices ? cfg->devices[q] : (int)q;
                if (other == md.dev) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, md.dev, other) == hipSuccess && can) {
                    hipError_t pe = hipDeviceEnablePeerAccess(other, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        }
    }
    // the parent handle answers the layout queries from shard 0
    e->L = m->d[0].e->L; e->cell_bits = m->d[0].e->cell_bits; e->feat_bits = m->d[0].e->feat_bits;
    e->skip_bits = m->d[0].e->skip_bits; e->n_cells = cfg->n_cells; e->n_features = cfg->n_features;
    e->use_lds_cells = m->d[0].e->use_lds_cells; e->use_lds_genes = m->d[0].e->use_lds_genes; e->lds_genes = m->d[0].e->lds_genes;
    e->wide = m->d[0].e->wide; e->long_umi = m->d[0].e->long_umi; e->sub_bits = m->d[0].e->sub_bits; e->group_bits = m->d[0].e->group_bits;
    return 0;
}

// per-destination key buffers of one device: make room for `need` keys per destination.  The new buffer is filled by
// copies queued on the compute stream (whole old regions: the exact counts are not known here) and the old one is kept
// until the streams are idle — growing never waits for the device.
static int multi_grow_shards(fastf_multi* m, MultiDev& md, u64 need) {
    if (need <= md.stride && md.d_shard.p) return 0;
    const u64 ns = std::max<u64>(need, std::max<u64>(md.stride * 2, 1u << 20));
    const u32 rows = md.e->wide ? 2 * m->G : m->G;             // (wide keys: row G + h holds the values of row h)
    void* np = nullptr;
    HIP_OK(hipMalloc(&np, (size_t)rows * ns * sizeof(u64)));
    if (md.d_shard.p) {
        for (u32 h = 0; h < rows; ++h)
            HIP_OK(hipMemcpyAsync((u64*)np + (u64)h * ns, (u64*)md.d_shard.p + (u64)h * md.stride, md.stride * sizeof(u64), hipMemcpyDeviceToDevice, md.e->s_compute));
        md.old_shards.push_back(md.d_shard.p);
    }
    md.d_shard.p = np; md.d_shard.bytes = (size_t)rows * ns * sizeof(u64);
    md.stride = ns;
    return 0;
}
static void multi_free_old_shards(MultiDev& md) {          // caller: the device's streams are idle
    for (void* q : md.old_shards) (void)hipFree(q);
    md.old_shards.clear();
}

// the oldest chunk in flight: its hit count -> hit-rank base, its draws, K1b
static int multi_retire_chunk(fastf_multi* m) {
    const fastf_multi::InFlight c = m->fifo.front();
    m->fifo.erase(m->fifo.begin());
    MultiDev& md = m->d[c.dev];
    fastf_engine* se = md.e;
    const u64 cap = m->cap;
    const size_t o_gx = cap * 8, o_umi = cap * 16, o_meta = cap * 20, o_ext = cap * 24;
    HIP_OK(hipSetDevice(md.dev));
    HIP_OK(hipEventSynchronize(md.ev_cnt[c.slot]));
    const u64* snap = md.h_cnt[c.slot];
    const u64 hits = snap[SM_N];
    if (hits > c.n) return set_err("internal: more hits than records");
    // the snapshot was taken behind this chunk's K1a, i.e. behind K1b of every earlier chunk of this device: exact counts
    u64 most = 0;
    for (u32 h = 0; h < m->G; ++h) { md.cnt[h] = snap[SM_KEYCOUNT + h]; most = std::max(most, md.cnt[h]); }
    md.keys_exact = most; md.recs_since = c.n;             // (later chunks of this device are not in flight: see enqueue)
    if (multi_grow_shards(m, md, md.keys_exact + md.recs_since)) return 1;
    // exactly this chunk's draws, in stream order (the chunks are retired in stream order)
    u64* small = (u64*)se->d_small.p;
    char* ds = (char*)md.d_stage[c.slot].p;
    // wide keys: the values of destination h behind the keys of all destinations; UMIs beyond 16 bases: the chunk's fifth array
    u64* const wide_vals = se->wide ? (u64*)md.d_shard.p + (u64)m->G * md.stride : nullptr;
    const u32* const wide_ext = se->long_umi ? (const u32*)(ds + o_ext) : nullptr;
    if (m->device_mt) {
        // every device moves the stream on by this chunk's hits (its own copy of the state, its own ring); the owner of the
        // chunk then runs K1b against its ring, hit ranks counted from the stream's start
        const u64 base = m->hits;
        if (hits) for (u32 g = 0; g < m->G; ++g) {
            MultiDev& mg = m->d[g];
            HIP_OK(hipSetDevice(mg.dev));
            if (!m->mt_uploaded || !mg.d_mt.p) {
                u64 r = 1024; while (r < 2 * cap) r <<= 1;
                if (mg.d_mt.ensure(sizeof(fastf_mt_t)) || mg.d_ring.ensure(r / 8) || mg.d_mtwords.ensure(cap * 4)) return 1;   // the decision stream: one bit per rank
                mg.ring_len = r;
                if (copy_h2d_on(mg.d_mt.p, &m->mt, sizeof(fastf_mt_t), mg.e->s_compute)) return 1;       // (m->mt is ordinary memory; once per stream position)
                mg.mt_idx = (u32)m->mt.idx;
            }
            // (a chunk with many hits: sub-streams seated by jump-ahead, generated by many workgroups — the device's serial
            //  2.5 G draws/s no longer caps this form; few hits: the one-workgroup kernel)
            if (launch_mt_decisions_par(mg.e, mg.e->s_compute, (u32*)mg.d_mt.p, &mg.mt_idx, mg.d_mtwords, (u32*)mg.d_ring.p, base, hits, mg.ring_len - 1, mg.e->threshold)) return 1;
            HIP_OK(hipGetLastError());
        }
        if (hits) m->mt_uploaded = true;
        m->hits += hits;
        HIP_OK(hipSetDevice(md.dev));
        if (!md.h_base[c.slot]) HIP_OK(hipHostMalloc((void**)&md.h_base[c.slot], 64, hipHostMallocDefault));
        if (md.k1b_queued[c.slot]) HIP_OK(hipEventSynchronize(md.ev_k1b[c.slot]));      // (the copy below that last read this word has long gone)
        *md.h_base[c.slot] = base;
        HIP_OK(hipMemcpyAsync(small + SM_DRAWBASE, md.h_base[c.slot], sizeof(u64), hipMemcpyHostToDevice, se->s_compute));
        if (launch_probe(se, (const u64*)ds, (const u64*)(ds + o_gx), (const u32*)(ds + o_umi), (const u32*)(ds + o_meta), c.n,
                         (const u32*)md.d_ring.p, base + hits, small + SM_DRAWBASE, (u64*)md.d_shard.p, md.stride, small + SM_KEYCOUNT,
                         small + SM_COUNTERS, true, se->s_compute, md.ring_len ? md.ring_len - 1 : ~0ull, nullptr, false, nullptr, nullptr,
                         wide_vals, wide_ext))
            return 1;
    } else {
    if (md.d_draws[c.slot].ensure(std::max<u64>(cap, 1) * 4)) return 1;
    if (!md.h_draws[c.slot]) HIP_OK(hipHostMalloc((void**)&md.h_draws[c.slot], std::max<u64>(cap, 1) * 4, hipHostMallocDefault));
    fastf_mt_fill(&m->mt, md.h_draws[c.slot], hits);
    m->hits += hits;
    {   // only the decisions go up: bit i = draw i < threshold, packed in place (word i >> 5 is written after draw i was read)
        u32* h = md.h_draws[c.slot];
        const u64 thr = se->threshold;
        u32 wcur = 0;
        for (u64 i = 0; i < hits; ++i) {
            wcur |= (u32)((u64)h[i] < thr) << (i & 31);
            if ((i & 31) == 31) { h[i >> 5] = wcur; wcur = 0; }
        }
        if (hits & 31) h[hits >> 5] = wcur;
    }
    if (hits) HIP_OK(hipMemcpyAsync(md.d_draws[c.slot].p, md.h_draws[c.slot], ((hits + 31) / 32) * 4, hipMemcpyHostToDevice, se->s_compute));
    if (launch_probe(se, (const u64*)ds, (const u64*)(ds + o_gx), (const u32*)(ds + o_umi), (const u32*)(ds + o_meta), c.n,
                     (const u32*)md.d_draws[c.slot].p, hits, nullptr, (u64*)md.d_shard.p, md.stride, small + SM_KEYCOUNT,
                     small + SM_COUNTERS, true, se->s_compute, ~0ull, nullptr, false, nullptr, nullptr, wide_vals, wide_ext))
        return 1;
    }
    HIP_OK(hipEventRecord(md.ev_k1b[c.slot], se->s_compute));
    md.k1b_queued[c.slot] = true;
    return 0;
}

// one chunk (n <= cap records at offset off of the caller's batch) onto its device: stage, copy, count
static int multi_enqueue_chunk(fastf_multi* m, const fastf_batch_t* b, size_t off, size_t n, bool pinned) {
    const u32 dev = (u32)(m->chunks % m->G);
    MultiDev& md = m->d[dev];
    fastf_engine* se = md.e;
    const u64 cap = m->cap;
    const size_t o_gx = cap * 8, o_umi = cap * 16, o_meta = cap * 20, o_ext = cap * 24;
    // K1a of this chunk overwrites the device's K1 scratch (cell indices, tile bases): K1b of the chunk this device took
    // before must be queued first, i.e. that chunk — and, for the stream order of the draws, every older one — retired
    for (;;) {
        bool pending = false;
        for (const auto& c : m->fifo) pending = pending || c.dev == dev;
        if (!pending) break;
        if (multi_retire_chunk(m)) return 1;
    }
    HIP_OK(hipSetDevice(md.dev));
    const int slot = md.next_slot; md.next_slot ^= 1;
    if (md.d_stage[slot].ensure(stage_bytes(cap))) return 1;
    if (!md.h_cnt[slot]) HIP_OK(hipHostMalloc((void**)&md.h_cnt[slot], SM_WORDS * sizeof(u64), hipHostMallocDefault));
    // the slot is refilled behind the K1b that read it (two chunks of this device ago)
    if (md.k1b_queued[slot]) HIP_OK(hipStreamWaitEvent(se->s_copy, md.ev_k1b[slot], 0));
    const void *s_cb = b->cb_key + off, *s_gx = b->gx_key + off, *s_umi = b->umi + off, *s_meta = b->meta + off;
    const void *s_ext = (se->long_umi && b->umi_ext) ? b->umi_ext + off : nullptr;      // bases 17.. (engines with umi_max_bases > 16)
    if (!pinned) {
        if (!md.h_stage[slot]) HIP_OK(hipHostMalloc(&md.h_stage[slot], stage_bytes(cap), hipHostMallocDefault));
        else HIP_OK(hipEventSynchronize(md.ev_in[slot]));   // the copy that last read this staging buffer has left it
        char* hs = (char*)md.h_stage[slot];
        memcpy(hs, s_cb, n * 8); memcpy(hs + o_gx, s_gx, n * 8); memcpy(hs + o_umi, s_umi, n * 4); memcpy(hs + o_meta, s_meta, n * 4);
        if (s_ext) { memcpy(hs + o_ext, s_ext, n * 4); s_ext = hs + o_ext; }
        s_cb = hs; s_gx = hs + o_gx; s_umi = hs + o_umi; s_meta = hs + o_meta;
    }
    char* ds = (char*)md.d_stage[slot].p;
    // (hipMemcpyDefault: a "pinned" batch may also be DEVICE memory — the records the BAM front end packed on the first device —
    // then this is a device-to-device copy, over xGMI when the chunk's device is another one)
    HIP_OK(hipMemcpyAsync(ds, s_cb, n * 8, hipMemcpyDefault, se->s_copy));
    HIP_OK(hipMemcpyAsync(ds + o_gx, s_gx, n * 8, hipMemcpyDefault, se->s_copy));
    HIP_OK(hipMemcpyAsync(ds + o_umi, s_umi, n * 4, hipMemcpyDefault, se->s_copy));
    HIP_OK(hipMemcpyAsync(ds + o_meta, s_meta, n * 4, hipMemcpyDefault, se->s_copy));
    if (se->long_umi) {
        if (s_ext) HIP_OK(hipMemcpyAsync(ds + o_ext, s_ext, n * 4, hipMemcpyDefault, se->s_copy));
        else HIP_OK(hipMemsetAsync(ds + o_ext, 0, n * 4, se->s_copy));                 // (a batch without the fifth array: no UMI beyond 16 bases)
    }
    HIP_OK(hipEventRecord(md.ev_in[slot], se->s_copy));
    HIP_OK(hipStreamWaitEvent(se->s_compute, md.ev_in[slot], 0));
    u64* small = (u64*)se->d_small.p;
    if (launch_probe_cells(se, (const u64*)ds, n, small + SM_N, se->s_compute)) return 1;
    HIP_OK(hipMemcpyAsync(md.h_cnt[slot], small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, se->s_compute));
    HIP_OK(hipEventRecord(md.ev_cnt[slot], se->s_compute));
    m->fifo.push_back({dev, slot, (u64)n});
    m->chunks++;
    md.records += n;
    m->total_records += n;
    m->finished = false;
    return 0;
}

static int multi_retire_all(fastf_multi* m) {
    while (!m->fifo.empty()) if (multi_retire_chunk(m)) return 1;
    return 0;
}

static int multi_push(fastf_engine* e, const fastf_batch_t* b, bool pinned) {
    fastf_multi* m = e->multi;
    if (multi_return_keycount(m)) return 1;
    for (size_t off = 0; off < b->n;) {
        const size_t n = std::min<size_t>(b->n - off, m->cap);
        if (multi_enqueue_chunk(m, b, off, n, pinned)) return 1;
        off += n;
    }
    if (pinned) {
        // the caller's arrays may change once this returns (fastf_engine_wait_input has nothing left to wait for): the
        // copies out of them must have left — the K1 work behind them stays queued
        for (u32 g = 0; g < m->G; ++g) {
            MultiDev& md = m->d[g];
            HIP_OK(hipSetDevice(md.dev));
            HIP_OK(hipStreamSynchronize(md.e->s_copy));
        }
    }
    return 0;
}

// rows of G shards -> one list in (cell, feature) order: each cell lives on one shard, so whole cell runs are copied
template <class Row>
static void multi_merge_by_cell(u32 G, const std::vector<const u32*>& cell, const std::vector<u64>& n, Row&& copy_run) {
    std::vector<u64> p(G, 0);
    for (;;) {
        u32 best = G; u32 bc = ~0u;
        for (u32 g = 0; g < G; ++g) if (p[g] < n[g] && cell[g][p[g]] < bc) { bc = cell[g][p[g]]; best = g; }
        if (best == G) break;
        u64 q = p[best];
        while (q < n[best] && cell[best][q] == bc) ++q;
        copy_run(best, p[best], q);
        p[best] = q;
    }
}

static int multi_exchange(fastf_multi* m) {
    const u32 G = m->G;
    const bool wide = m->d[0].e->wide;
    if (multi_retire_all(m)) return 1;
    // exact key counts
    for (u32 g = 0; g < G; ++g) {
        MultiDev& md = m->d[g];
        HIP_OK(hipSetDevice(md.dev));
        HIP_OK(hipMemcpyAsync(md.h_info, md.e->d_small.p, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, md.e->s_compute));
        HIP_OK(hipStreamSynchronize(md.e->s_compute));
        for (u32 h = 0; h < G; ++h) md.cnt[h] = md.h_info[SM_KEYCOUNT + h];
        if (md.h_info[SM_COUNTERS + 3]) return set_err("%s", err_bits_text(md.h_info[SM_COUNTERS + 3]));
        multi_free_old_shards(md);                                          // the compute stream has just been waited for
    }
    m->c_sampled = m->c_valid = 0;
    for (u32 g = 0; g < G; ++g) { m->c_sampled += m->d[g].h_info[SM_COUNTERS + 1]; m->c_valid += m->d[g].h_info[SM_COUNTERS + 2]; }
    std::vector<std::vector<u64>> at(G, std::vector<u64>(G, 0));            // at[h][g]: where g's keys start in h's receive buffer
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        u64 r = 0;
        for (u32 g = 0; g < G; ++g) { at[h][g] = r; r += m->d[g].cnt[h]; }
        mh.n_recv = r;
        HIP_OK(hipSetDevice(mh.dev));
        if (wide) { if (grow_keys(mh.e, std::max<u64>(r, 1))) return 1; }      // the receiver's own wide store (keys, values, sort and row buffers)
        else if (mh.d_recv.ensure(std::max<u64>(r, 1) * 8) || mh.d_tmp.ensure(std::max<u64>(r, 1) * 8)) return 1;
    }
    // where device h receives (wide keys: two arrays, same offsets) and where device g's values for destination h lie
    auto recv_keys = [&](u32 h) { return wide ? (u64*)m->d[h].e->d_keys.p : (u64*)m->d[h].d_recv.p; };
    auto recv_vals = [&](u32 h) { return (u64*)m->d[h].e->d_vals.p; };
    auto send_vals = [&](u32 g, u32 h) { return (const u64*)m->d[g].d_shard.p + (u64)(G + h) * m->d[g].stride; };
    if (m->use_rccl) {
        // the single all-to-all of the path: one group of G x G send/recv pairs, each device's calls on its own stream
        NCCL_OK(m, m->rccl.GroupStart());
        for (u32 g = 0; g < G; ++g) {
            MultiDev& mg = m->d[g];
            for (u32 h = 0; h < G; ++h) {
                if (mg.cnt[h]) NCCL_OK_IN_GROUP(m, m->rccl.Send((const u64*)mg.d_shard.p + (u64)h * mg.stride, mg.cnt[h], ncclUint64, (int)h, m->comms[g], mg.e->s_compute));
                if (m->d[h].cnt[g]) NCCL_OK_IN_GROUP(m, m->rccl.Recv(recv_keys(g) + at[g][h], m->d[h].cnt[g], ncclUint64, (int)h, m->comms[g], mg.e->s_compute));
                if (wide && mg.cnt[h]) NCCL_OK_IN_GROUP(m, m->rccl.Send(send_vals(g, h), mg.cnt[h], ncclUint64, (int)h, m->comms[g], mg.e->s_compute));
                if (wide && m->d[h].cnt[g]) NCCL_OK_IN_GROUP(m, m->rccl.Recv(recv_vals(g) + at[g][h], m->d[h].cnt[g], ncclUint64, (int)h, m->comms[g], mg.e->s_compute));
            }
        }
        NCCL_OK(m, m->rccl.GroupEnd());
    } else {
        for (u32 g = 0; g < G; ++g) {
            MultiDev& mg = m->d[g];
            HIP_OK(hipSetDevice(mg.dev));
            for (u32 h = 0; h < G; ++h) {
                if (!mg.cnt[h]) continue;
                MultiDev& mh = m->d[h];
                const u64* src = (const u64*)mg.d_shard.p + (u64)h * mg.stride;
                u64* dst = recv_keys(h) + at[h][g];
                if (mh.dev == mg.dev) HIP_OK(hipMemcpyAsync(dst, src, mg.cnt[h] * 8, hipMemcpyDeviceToDevice, mg.e->s_compute));
                else HIP_OK(hipMemcpyPeerAsync(dst, mh.dev, src, mg.dev, mg.cnt[h] * 8, mg.e->s_compute));
                if (wide) {
                    if (mh.dev == mg.dev) HIP_OK(hipMemcpyAsync(recv_vals(h) + at[h][g], send_vals(g, h), mg.cnt[h] * 8, hipMemcpyDeviceToDevice, mg.e->s_compute));
                    else HIP_OK(hipMemcpyPeerAsync(recv_vals(h) + at[h][g], mh.dev, send_vals(g, h), mg.dev, mg.cnt[h] * 8, mg.e->s_compute));
                }
            }
            HIP_OK(hipEventRecord(mg.ev_sent, mg.e->s_compute));
        }
        for (u32 h = 0; h < G; ++h) {                                       // a receiver sorts after every sender has delivered
            HIP_OK(hipSetDevice(m->d[h].dev));
            for (u32 g = 0; g < G; ++g) HIP_OK(hipStreamWaitEvent(m->d[h].e->s_compute, m->d[g].ev_sent, 0));
        }
    }
    return 0;
}

// K2 + K3 on every device over what it received; low_bit = bits left unsorted (0: full sort)
static int multi_sort_reduce(fastf_multi* m, u32 low_bit, bool resort) {
    const u32 G = m->G;
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        fastf_engine* se = mh.e;
        HIP_OK(hipSetDevice(mh.dev));
        u64* small = (u64*)se->d_small.p;
        const u64 n = mh.n_recv;
        mh.h_info[SM_N] = n;
        HIP_OK(hipMemcpyAsync(small + SM_N, mh.h_info + SM_N, sizeof(u64), hipMemcpyHostToDevice, se->s_compute));
        if (!n) { HIP_OK(hipMemsetAsync(small + SM_NNZ, 0, sizeof(u64), se->s_compute)); continue; }
        if (mh.d_f.ensure(n * 4) || mh.d_c.ensure(n * 4) || mh.d_k.ensure(n * 4)) return 1;
        u64* from = (u64*)mh.d_recv.p; u64* other = (u64*)mh.d_tmp.p;
        if (resort && mh.sorted_in_tmp) std::swap(from, other);
        int in_other = 0;
        if (launch_sort(se, from, other, small + SM_N, n, se->L.total_bits, low_bit, &in_other, se->s_compute)) return 1;
        mh.sorted_in_tmp = resort ? (in_other ? !mh.sorted_in_tmp : mh.sorted_in_tmp) : in_other;
        mh.fully_sorted = low_bit == 0;
        const u64* sorted = mh.sorted_in_tmp ? (u64*)mh.d_tmp.p : (u64*)mh.d_recv.p;
        if (launch_reduce<false>(se, sorted, small + SM_N, n, (u32*)mh.d_f.p, (u32*)mh.d_c.p, (u32*)mh.d_k.p, nullptr, small + SM_NNZ,
                                 low_bit, se->s_compute))
            return 1;
    }
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        HIP_OK(hipSetDevice(mh.dev));
        HIP_OK(hipMemcpyAsync(mh.h_info, mh.e->d_small.p, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, mh.e->s_compute));
        HIP_OK(hipStreamSynchronize(mh.e->s_compute));
    }
    return 0;
}

// Keys wider than 64 bits: every device has received into its sub-engine's own store.  The sort and the reduce are queued on all
// devices (finish_queue), then each sub-engine's fastf_engine_finish waits for its own and brings its rows to the host.
static int multi_finish_wide(fastf_multi* m) {
    const u32 G = m->G;
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        fastf_engine* se = mh.e;
        HIP_OK(hipSetDevice(mh.dev));
        se->keys_so_far = mh.n_recv; se->finished = false; se->finish_queued = false; se->fully_sorted = false;
        // (the count the sub-engine's sort reads: its own key count word — K1b's count for destination 0 until now, put back below)
        mh.h_info[SM_KEYCOUNT] = mh.n_recv;
        HIP_OK(hipMemcpyAsync((u64*)se->d_small.p + SM_KEYCOUNT, mh.h_info + SM_KEYCOUNT, sizeof(u64), hipMemcpyHostToDevice, se->s_compute));
        if (finish_queue(se)) return 1;
    }
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        fastf_coo_t part;
        if (fastf_engine_finish(mh.e, &part, nullptr)) return 1;
        mh.pf = part.feature; mh.pc = part.cell; mh.pk = part.count; mh.nnz = part.nnz;       // (the sub-engine's buffer: valid until its reset)
    }
    m->keycount_lent = true;           // (-u rows read the same word; the next push puts K1b's count back: multi_return_keycount)
    return 0;
}

// a push after a wide finish: K1b goes on counting keys for destination 0 in the word the sub-engine's sort was lent
static int multi_return_keycount(fastf_multi* m) {
    if (!m->keycount_lent) return 0;
    for (MultiDev& md : m->d) {
        HIP_OK(hipSetDevice(md.dev));
        md.h_info[SM_KEYCOUNT] = md.cnt[0];
        HIP_OK(hipMemcpyAsync((u64*)md.e->d_small.p + SM_KEYCOUNT, md.h_info + SM_KEYCOUNT, sizeof(u64), hipMemcpyHostToDevice, md.e->s_compute));
        HIP_OK(hipStreamSynchronize(md.e->s_compute));
    }
    m->keycount_lent = false;
    return 0;
}

static int multi_finish(fastf_engine* e, fastf_coo_t* coo, uint64_t counters[3]) {
    fastf_multi* m = e->multi;
    const u32 G = m->G;
    if (!m->finished) {
        if (multi_exchange(m)) return 1;
        if (m->d[0].e->wide) { if (multi_finish_wide(m)) return 1; } else {
        const u32 skip = m->d[0].e->skip_bits;
        if (multi_sort_reduce(m, skip, false)) return 1;
        bool too_long = false;
        for (u32 h = 0; h < G; ++h) too_long = too_long || (m->d[h].h_info[SM_COUNTERS + 3] & ERR_RUN_TOO_LONG);
        if (too_long) {                                  // deep (cell, feature) groups: finish the sort everywhere and reduce exactly
            for (u32 h = 0; h < G; ++h) {
                MultiDev& mh = m->d[h];
                HIP_OK(hipSetDevice(mh.dev));
                if (fastf_dev_clear_error_bits(mh.e, ERR_RUN_TOO_LONG, mh.e->s_compute)) return 1;
            }
            if (multi_sort_reduce(m, 0, true)) return 1;
        }
        for (u32 h = 0; h < G; ++h) {
            MultiDev& mh = m->d[h];
            if (mh.h_info[SM_COUNTERS + 3]) return set_err("%s", err_bits_text(mh.h_info[SM_COUNTERS + 3]));
            const u64 nnz = mh.n_recv ? mh.h_info[SM_NNZ] : 0;
            if (mh.rows.ensure(std::max<u64>(nnz, 1))) return 1;
            mh.pf = mh.rows.f(); mh.pc = mh.rows.c(); mh.pk = mh.rows.k(); mh.nnz = nnz;
            if (nnz) {
                HIP_OK(hipSetDevice(mh.dev));
                if (copy_d2h(mh.rows.f(), mh.d_f.p, nnz * 4) || copy_d2h(mh.rows.c(), mh.d_c.p, nnz * 4) || copy_d2h(mh.rows.k(), mh.d_k.p, nnz * 4)) return 1;
            }
        }
        }
        u64 total = 0;
        std::vector<const u32*> cells(G); std::vector<u64> ns(G);
        for (u32 h = 0; h < G; ++h) { cells[h] = m->d[h].pc; ns[h] = m->d[h].nnz; total += ns[h]; }
        if (m->merged.ensure(std::max<u64>(total, 1))) return 1;
        u32 *of = m->merged.f(), *oc = m->merged.c(), *ok = m->merged.k();
        u64 w = 0;
        multi_merge_by_cell(G, cells, ns, [&](u32 g, u64 a, u64 b) {
            const MultiDev& mg = m->d[g];
            memcpy(of + w, mg.pf + a, (b - a) * 4);
            memcpy(oc + w, mg.pc + a, (b - a) * 4);
            memcpy(ok + w, mg.pk + a, (b - a) * 4);
            w += b - a;
        });
        m->merged_n = total;
        m->finished = true;
    }
    coo->feature = m->merged.f(); coo->cell = m->merged.c(); coo->count = m->merged.k(); coo->nnz = m->merged_n;
    if (counters) { counters[0] = m->total_records; counters[1] = m->c_sampled; counters[2] = m->c_valid; }
    return 0;
}

static int multi_umi_rows(fastf_engine* e, fastf_umi_rows_t* rows) {
    fastf_multi* m = e->multi;
    if (!m->finished) return set_err("call fastf_engine_finish first");
    const u32 G = m->G;
    if (m->d[0].e->wide) {                                // every sub-engine's own -u rows (umi_rows_wide), merged by cell
        std::vector<fastf_umi_rows_t> part(G);
        for (u32 h = 0; h < G; ++h) if (fastf_engine_umi_rows(m->d[h].e, &part[h])) return 1;
        u64 total = 0;
        std::vector<const u32*> cells(G); std::vector<u64> ns(G);
        for (u32 h = 0; h < G; ++h) { cells[h] = part[h].cell; ns[h] = part[h].n; total += ns[h]; }
        m->ufeature.resize(total); m->ucell.resize(total); m->uumi.resize(total); m->unonnull.resize(total); m->ncopy.resize(total);
        u64 w = 0;
        multi_merge_by_cell(G, cells, ns, [&](u32 g, u64 a, u64 b) {
            const fastf_umi_rows_t& r = part[g];
            for (u64 i = a; i < b; ++i, ++w) {
                m->ufeature[w] = r.feature[i]; m->ucell[w] = r.cell[i]; m->unonnull[w] = r.nonnull[i]; m->uumi[w] = r.umi[i]; m->ncopy[w] = r.n_copy[i];
            }
        });
        rows->feature = m->ufeature.data(); rows->cell = m->ucell.data(); rows->n_copy = m->ncopy.data();
        rows->umi = m->uumi.data(); rows->nonnull = m->unonnull.data(); rows->n = total;
        return 0;
    }
    for (u32 h = 0; h < G; ++h) {                         // -u rows are ordered by blob: every shard needs the full sort
        MultiDev& mh = m->d[h];
        fastf_engine* se = mh.e;
        mh.ukeys.clear(); mh.ncopy.clear();
        if (!mh.n_recv) continue;
        HIP_OK(hipSetDevice(mh.dev));
        u64* small = (u64*)se->d_small.p;
        if (mh.d_ukeys.ensure(mh.n_recv * 8) || mh.d_ncopy.ensure(mh.n_recv * 4)) return 1;
        if (!mh.fully_sorted) {
            u64* from = mh.sorted_in_tmp ? (u64*)mh.d_tmp.p : (u64*)mh.d_recv.p;
            u64* other = mh.sorted_in_tmp ? (u64*)mh.d_recv.p : (u64*)mh.d_tmp.p;
            int in_other = 0;
            if (launch_sort(se, from, other, small + SM_N, mh.n_recv, se->L.total_bits, 0, &in_other, se->s_compute)) return 1;
            if (in_other) mh.sorted_in_tmp = !mh.sorted_in_tmp;
            mh.fully_sorted = true;
        }
        const u64* sorted = mh.sorted_in_tmp ? (u64*)mh.d_tmp.p : (u64*)mh.d_recv.p;
        if (launch_reduce<true>(se, sorted, small + SM_N, mh.n_recv, nullptr, nullptr, (u32*)mh.d_ncopy.p, (u64*)mh.d_ukeys.p,
                                small + SM_NROWS_U, 0, se->s_compute))
            return 1;
        HIP_OK(hipMemcpyAsync(mh.h_info, small, SM_WORDS * sizeof(u64), hipMemcpyDeviceToHost, se->s_compute));
    }
    std::vector<std::vector<u32>> ucell(G);
    for (u32 h = 0; h < G; ++h) {
        MultiDev& mh = m->d[h];
        if (!mh.n_recv) continue;
        HIP_OK(hipSetDevice(mh.dev));
        HIP_OK(hipStreamSynchronize(mh.e->s_compute));
        if (mh.h_info[SM_COUNTERS + 3]) return set_err("%s", err_bits_text(mh.h_info[SM_COUNTERS + 3]));
        const u64 nr = mh.h_info[SM_NROWS_U];
        mh.ukeys.resize(nr); mh.ncopy.resize(nr);
        if (nr) {
            if (copy_d2h(mh.ukeys.data(), mh.d_ukeys.p, nr * 8) || copy_d2h(mh.ncopy.data(), mh.d_ncopy.p, nr * 4)) return 1;
        }
        ucell[h].resize(nr);
        for (u64 i = 0; i < nr; ++i) ucell[h][i] = (u32)(mh.ukeys[i] >> e->L.cell_shift);
    }
    u64 total = 0;
    std::vector<const u32*> cells(G); std::vector<u64> ns(G);
    for (u32 h = 0; h < G; ++h) { cells[h] = ucell[h].data(); ns[h] = ucell[h].size(); total += ns[h]; }
    m->ufeature.resize(total); m->ucell.resize(total); m->uumi.resize(total); m->unonnull.resize(total); m->ncopy.resize(total);
    const u32 fmask = (u32)((1ull << e->feat_bits) - 1);
    const u64 umask = e->L.umi_bits >= 64 ? ~0ull : ((1ull << e->L.umi_bits) - 1);
    u64 w = 0;
    multi_merge_by_cell(G, cells, ns, [&](u32 g, u64 a, u64 b) {
        const MultiDev& mg = m->d[g];
        for (u64 i = a; i < b; ++i, ++w) {
            const u64 k = mg.ukeys[i];
            m->ufeature[w] = (u32)(k >> e->L.feat_shift) & fmask;
            m->ucell[w] = (u32)(k >> e->L.cell_shift);
            m->unonnull[w] = (uint8_t)((k >> (e->L.umi_bits + e->L.len_bits)) & 1);
            m->uumi[w] = (u32)(((k >> e->L.len_bits) & umask) << (32 - e->L.umi_bits));
            m->ncopy[w] = mg.ncopy[i];
        }
    });
    rows->feature = m->ufeature.data(); rows->cell = m->ucell.data(); rows->n_copy = m->ncopy.data();
    rows->umi = m->uumi.data(); rows->nonnull = m->unonnull.data(); rows->n = total;
    return 0;
}

static int multi_reset(fastf_engine* e, bool reseed, u32 seed, u64 skip) {
    fastf_multi* m = e->multi;
    if (reseed) {
        if (multi_retire_all(m)) return 1;                   // chunks in flight take their draws from the stream as it was
        m->mt_seed0 = seed; m->mt_skip0 = skip; fastf_mt_seed(&m->mt, seed); fastf_mt_skip(&m->mt, skip);
        m->mt_uploaded = false;                              // the devices take the new position with the next chunk; ranks count from it
        m->hits = 0;
        return 0;
    }
    if (multi_retire_all(m)) return 1;                        // (what is in flight is finished, then forgotten)
    for (MultiDev& md : m->d) {
        if (fastf_engine_reset(md.e)) return 1;              // synchronises the device
        multi_free_old_shards(md);
        for (u64& c : md.cnt) c = 0;
        md.n_recv = 0; md.keys_exact = md.recs_since = md.records = 0;
    }
    m->chunks = 0;
    m->hits = m->total_records = m->c_sampled = m->c_valid = 0;
    m->finished = false; m->keycount_lent = false;
    return 0;
}

static int multi_device_records(const fastf_engine* e, uint64_t* records, u32 n) {
    if (!e->multi) { if (n < 1) return set_err("room for one count needed"); records[0] = e->total_records; return 0; }
    if (n < e->multi->G) return set_err("room for %u counts needed", e->multi->G);
    for (u32 g = 0; g < e->multi->G; ++g) records[g] = e->multi->d[g].records;
    return 0;
}
