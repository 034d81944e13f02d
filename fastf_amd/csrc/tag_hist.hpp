// tag_hist.hpp — device side of the tag-histogram paths (`crb`, `extract`), part of the
// umi_engine.hip translation unit (it reuses K2 = the LSD radix sort and K3u = run-length rows).
//
// The reference builds an insertion-order binary search tree per tag value (filter.c:105-124 insert_tree;
// extract.c:3-31 insert_CB_node for the two-level CB → CR tree) and prints it in pre-order.  The shape of such a
// tree — hence the output order — is a function of (a) the set of distinct values, (b) how often each occurs and
// (c) the position of each value's FIRST occurrence in the record stream (the tree is the Cartesian tree of the
// values in strcmp order with "first seen" as heap priority).  The device computes exactly (a), (b), (c):
//
//   single tag   keys → sort → run-length (distinct key, copies) → first-occurrence index by binary search + atomicMin
//   tag pair     distinct key1 / distinct key2 (two sorts) → per record the dense pair code (rank1 << 32 | rank2)
//                → sort → run-length → first-occurrence index; the CB level follows from the pairs
//
// Records whose tag is absent carry key 0; in the sort copy they are replaced by one key that does occur so that
// constant key digits stay constant (passes over constant digits are skipped) and its count is corrected on the host.
#pragma once

namespace fastf {

// sort copy: absent (0) keys take the value of the record at *fill_idx (a present one)
__global__ __launch_bounds__(256) void th_fill_copy_kernel(const u64* __restrict__ in, u64* __restrict__ out, u64 n, u64 fill_idx) {
    const u64 fill = in[fill_idx];
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 k = in[i];
        out[i] = k ? k : fill;
    }
}

// index of key in the ascending array u[0..m) (the key is known to be present)
__device__ __forceinline__ u32 th_lower_bound(const u64* __restrict__ u, u32 m, u64 key) {
    u32 lo = 0, hi = m;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (u[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// pair code of every record: (rank of key1 among the distinct key1 + 1) << 32 | rank of key2; 0 if either is absent
__global__ __launch_bounds__(256) void th_pair_code_kernel(const u64* __restrict__ k1, const u64* __restrict__ k2, u64 n,
                                                           const u64* __restrict__ u1, const u64* __restrict__ n1_ptr,
                                                           const u64* __restrict__ u2, const u64* __restrict__ n2_ptr,
                                                           u64* __restrict__ code) {
    const u32 n1 = (u32)*n1_ptr, n2 = (u32)*n2_ptr;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 a = k1[i], b = k2[i];
        u64 c = 0;
        if (a && b) c = ((u64)(th_lower_bound(u1, n1, a) + 1) << 32) | th_lower_bound(u2, n2, b);
        code[i] = c;
    }
}

// first[d] = smallest record index whose key is u[d].  Records are visited in ascending order by ascending
// workgroups, so after the first wave of a hot key the (possibly stale, hence only ever too large) plain read of
// first[d] filters nearly every atomic away.
__global__ __launch_bounds__(256) void th_first_index_kernel(const u64* __restrict__ keys, u64 n, const u64* __restrict__ u,
                                                             const u64* __restrict__ m_ptr, u32* __restrict__ first) {
    const u32 m = (u32)*m_ptr;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 k = keys[i];
        if (!k) continue;
        const u32 d = th_lower_bound(u, m, k);
        if ((u32)i < __builtin_nontemporal_load(&first[d])) atomicMin(&first[d], (u32)i);
    }
}

}  // namespace fastf

using namespace fastf;

// the little of std::vector the result arrays need, without value-initialisation on resize
template <typename T> struct RawVec {
    T* p = nullptr; size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec&) = delete; RawVec& operator=(const RawVec&) = delete;
    ~RawVec() { free(p); }
    void resize(size_t k) {
        if (k > cap) { free(p); p = nullptr; cap = 0; p = (T*)malloc((k ? k : 1) * sizeof(T)); if (!p) throw std::bad_alloc(); cap = k; }
        n = k;
    }
    void assign(size_t k, T v) { resize(k); for (size_t i = 0; i < k; ++i) p[i] = v; }
    void clear() { n = 0; }
    T* data() { return p; } const T* data() const { return p; }
    T* begin() { return p; } T* end() { return p + n; }
    T& operator[](size_t i) { return p[i]; } const T& operator[](size_t i) const { return p[i]; }
};
// fn(lo, hi) over [0, n) on the host threads (results post-processing: widening 20 M counters is not a job for one thread)
template <typename F> static void th_par_for(u64 n, F fn) {
    unsigned T = n < (1u << 18) ? 1u : (unsigned)fastf_host_thread_count();
    if (T > 32) T = 32;
    if (T <= 1) { fn((u64)0, n); return; }
    // a thread that cannot be started (EAGAIN, a cgroup's limit) must not unwind past joinable threads — that is
    // std::terminate, which no exception barrier of the C ABI can turn into an error code: its slice runs here instead
    std::vector<std::thread> th;
    th.reserve(T);
    std::vector<unsigned> inline_slices;
    for (unsigned t = 1; t < T; ++t) {
        try { th.emplace_back([=] { fn(n * t / T, n * (t + 1) / T); }); }
        catch (...) { inline_slices.push_back(t); }
    }
    fn((u64)0, n / T);
    for (unsigned t : inline_slices) fn(n * t / T, n * (t + 1) / T);
    for (auto& x : th) x.join();
}

struct fastf_taghist {
    fastf_engine* ws = nullptr;          // workspace engine: streams + sort/reduce scratch
    int device = 0;
    bool pair = false, mode_set = false;
    u64 n = 0, cap = 0;                  // records pushed / capacity of d_k1, d_k2
    DevBuf d_k1, d_k2;
    u64 or1 = 0, and1 = ~0ull, or2 = 0, and2 = ~0ull;      // over the present keys
    u64 first_present1 = ~0ull, first_present2 = ~0ull, first_valid_pair = ~0ull;
    u64 n_present1 = 0, n_present2 = 0, n_valid = 0;
    void* h_stage = nullptr; size_t h_stage_bytes = 0;      // pinned bounce buffer
    // device results
    DevBuf d_a, d_b, d_code, d_u1, d_c1, d_u2, d_c2, d_up, d_cp, d_first, d_small;
    // host results (RawVec: resize does not touch the memory — 20 M (CB, CR) pairs are 0.9 GB of arrays that a std::vector
    // would fill with zeros before the copies from the device overwrite them)
    RawVec<u64> h_key1, h_count1, h_first1, h_pair_key2, h_pair_count, h_pair_first, h_u2, h_up;
    RawVec<u32> h_pair_k1, h_tmp32;
};

extern "C" int fastf_taghist_create(int device, fastf_taghist_t** out) FASTF_TRY {
    if (!out) return set_err("null argument");
    *out = nullptr;
    static const u64 one = (1ull << 62) | (1ull << 57);        // any non-zero key: the workspace engine's lists are unused
    fastf_engine_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.cell_keys = (const uint64_t*)&one; cfg.n_cells = 1; cfg.feature_keys = (const uint64_t*)&one; cfg.n_features = 1;
    cfg.draw_threshold = 1ull << 32; cfg.umi_max_bases = 1; cfg.n_shards = 1; cfg.device = device;
    fastf_engine_t* ws = nullptr;
    if (fastf_engine_create(&cfg, &ws)) return 1;
    fastf_taghist* h = new fastf_taghist();
    h->ws = ws; h->device = device;
    if (h->d_small.ensure(8 * sizeof(u64))) { fastf_engine_destroy(ws); delete h; return 1; }
    *out = h;
    return 0;
} FASTF_CATCH_INT

extern "C" void fastf_taghist_destroy(fastf_taghist_t* h) FASTF_TRY {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    DevBuf* all[] = {&h->d_k1, &h->d_k2, &h->d_a, &h->d_b, &h->d_code, &h->d_u1, &h->d_c1, &h->d_u2, &h->d_c2,
                     &h->d_up, &h->d_cp, &h->d_first, &h->d_small};
    for (DevBuf* b : all) b->release();
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    fastf_engine_destroy(h->ws);
    delete h;
} FASTF_CATCH_VOID

static int th_grow(fastf_taghist* h, u64 need) {
    if (need <= h->cap) return 0;
    u64 ncap = std::max<u64>(need, std::max<u64>(h->cap * 2, 1ull << 22));
    hipStream_t s = h->ws->s_compute;
    for (int which = 0; which < (h->pair ? 2 : 1); ++which) {
        DevBuf& b = which ? h->d_k2 : h->d_k1;
        DevBuf nb;
        if (nb.ensure(ncap * sizeof(u64))) return 1;
        if (h->n) HIP_OK(hipMemcpyAsync(nb.p, b.p, h->n * sizeof(u64), hipMemcpyDeviceToDevice, s));
        HIP_OK(hipStreamSynchronize(s));
        b.release();
        b = nb;
    }
    h->cap = ncap;
    return 0;
}

extern "C" int fastf_taghist_push(fastf_taghist_t* h, const uint64_t* key1, const uint64_t* key2, size_t n) FASTF_TRY {
    if (!h) return set_err("null histogram");
    if (!key1) return set_err("null keys");
    if (h->mode_set && h->pair != (key2 != nullptr)) return set_err("single-tag and tag-pair pushes cannot be mixed");
    h->pair = key2 != nullptr; h->mode_set = true;
    if (n == 0) return 0;
    if (h->n + n >= (1ull << 32) - 1) return set_err("tag histogram: more than 2^32-2 records");
    HIP_OK(hipSetDevice(h->device));
    if (th_grow(h, h->n + n)) return 1;
    hipStream_t s = h->ws->s_compute;
    // summary of the present keys (which digits vary, first present record) while the copy is staged
    const size_t chunk = 1u << 22;
    if (!h->h_stage) {
        HIP_OK(hipHostMalloc(&h->h_stage, 2 * chunk * sizeof(u64), hipHostMallocDefault));
        h->h_stage_bytes = 2 * chunk * sizeof(u64);
    }
    for (size_t off = 0; off < n; off += chunk) {
        const size_t m = std::min(chunk, n - off);
        u64* st1 = (u64*)h->h_stage; u64* st2 = st1 + chunk;
        HIP_OK(hipStreamSynchronize(s));                      // the bounce buffer is free again
        u64 o1 = h->or1, a1 = h->and1, o2 = h->or2, a2 = h->and2;
        for (size_t i = 0; i < m; ++i) {
            const u64 k = key1[off + i];
            st1[i] = k;
            if (k) { o1 |= k; a1 &= k; h->n_present1++; if (h->first_present1 == ~0ull) h->first_present1 = h->n + off + i; }
            if (key2) {
                const u64 q = key2[off + i];
                st2[i] = q;
                if (q) { o2 |= q; a2 &= q; h->n_present2++; if (h->first_present2 == ~0ull) h->first_present2 = h->n + off + i; }
                if (k && q) { h->n_valid++; if (h->first_valid_pair == ~0ull) h->first_valid_pair = h->n + off + i; }
            } else if (k) h->n_valid++;
        }
        h->or1 = o1; h->and1 = a1; h->or2 = o2; h->and2 = a2;
        HIP_OK(hipMemcpyAsync((u64*)h->d_k1.p + h->n + off, st1, m * sizeof(u64), hipMemcpyHostToDevice, s));
        if (key2) HIP_OK(hipMemcpyAsync((u64*)h->d_k2.p + h->n + off, st2, m * sizeof(u64), hipMemcpyHostToDevice, s));
    }
    h->n += n;
    return 0;
} FASTF_CATCH_INT

// LSD passes over the digits named in pass_mask only (the others are constant over all keys)
static int th_sort(fastf_engine* e, u64* keys, u64* tmp, const u64* d_n, u64 n, u32 pass_mask, int* in_tmp, hipStream_t s) {
    *in_tmp = 0;
    if (n == 0 || pass_mask == 0) return 0;
    if (reserve_workspace(e, 0, n)) return 1;
    const u32 ipt = choose_sort_ipt(n);
    const u32 T = (u32)((n + (u64)ipt * SORT_THREADS - 1) / ((u64)ipt * SORT_THREADS));
    u32* bintot = (u32*)e->d_binbase.p; u32* cnt = (u32*)e->d_cnt.p;
    int flip = 0;
    for (u32 q = 0; q < 8; ++q) {
        if (!((pass_mask >> q) & 1)) continue;
        const u64* src = flip ? tmp : keys;
        u64* dst = flip ? keys : tmp;
        hipLaunchKernelGGL(tile_count_kernel<false>, dim3(tile_grid(T)), dim3(SORT_THREADS), 0, s, src, d_n, 8 * q, cnt, ipt, SegMap{nullptr, nullptr, 0, 0});
        hipLaunchKernelGGL(row_scan_kernel, dim3(RADIX), dim3(1024), 0, s, cnt, d_n, bintot, ipt);
        launch_scatter(8 * q, T, s, src, dst, d_n, (const u32*)cnt, (const u32*)bintot, ipt);
        flip ^= 1;
    }
    HIP_OK(hipGetLastError());
    *in_tmp = flip;
    return 0;
}

static u32 th_pass_mask(u64 varying) {
    u32 m = 0;
    for (u32 q = 0; q < 8; ++q) if ((varying >> (8 * q)) & 255) m |= 1u << q;
    return m;
}

// distinct values (ascending) + copies of the n keys in `src` whose absent entries are filled from record fill_idx.
// d_u / d_c must hold n entries; *d_m receives the number of distinct values.
static int th_distinct(fastf_taghist* h, const u64* src, u64 fill_idx, u64 varying, u64* d_u, u32* d_c, u64* d_m, hipStream_t s) {
    fastf_engine* e = h->ws;
    const u64 n = h->n;
    u64* d_nn = (u64*)h->d_small.p + 7;
    if (copy_h2d_on(d_nn, &h->n, sizeof(u64), s)) return 1;      // (a word of the handle: through the bounce buffer, like every host word)
    const u32 grid = (u32)std::min<u64>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(th_fill_copy_kernel, dim3(grid), dim3(256), 0, s, src, (u64*)h->d_a.p, n, fill_idx);
    int in_tmp = 0;
    if (th_sort(e, (u64*)h->d_a.p, (u64*)h->d_b.p, d_nn, n, th_pass_mask(varying), &in_tmp, s)) return 1;
    const u64* sorted = in_tmp ? (const u64*)h->d_b.p : (const u64*)h->d_a.p;
    return launch_reduce<true>(e, sorted, d_nn, n, nullptr, nullptr, d_c, d_u, d_m, 0, s);
}

extern "C" int fastf_taghist_finish(fastf_taghist_t* h, fastf_taghist_result_t* r) FASTF_TRY {
    if (!h || !r) return set_err("null argument");
    memset(r, 0, sizeof *r);
    r->n_records = h->n; r->n_valid = h->n_valid; r->n_key1_present = h->n_present1;
    h->h_key1.clear(); h->h_count1.clear(); h->h_first1.clear();
    h->h_pair_k1.clear(); h->h_pair_key2.clear(); h->h_pair_count.clear(); h->h_pair_first.clear();
    if (h->n_valid == 0) return 0;
    HIP_OK(hipSetDevice(h->device));
    hipStream_t s = h->ws->s_compute;
    const u64 n = h->n;
    const size_t nb = n * sizeof(u64);
    if (h->d_a.ensure(nb) || h->d_b.ensure(nb) || h->d_u1.ensure(nb) || h->d_c1.ensure(n * sizeof(u32)) ||
        h->d_first.ensure(n * sizeof(u32))) return 1;
    u64* d_m1 = (u64*)h->d_small.p; u64* d_m2 = d_m1 + 1; u64* d_mp = d_m1 + 2;
    const u32 grid = (u32)std::min<u64>((n + 255) / 256, 8192);

    // level 1: distinct key1
    if (th_distinct(h, (const u64*)h->d_k1.p, h->first_present1, h->or1 ^ h->and1, (u64*)h->d_u1.p, (u32*)h->d_c1.p, d_m1, s)) return 1;
    u64 m1 = 0;
    if (!h->pair) {
        HIP_OK(hipMemsetAsync(h->d_first.p, 0xff, n * sizeof(u32), s));
        hipLaunchKernelGGL(th_first_index_kernel, dim3(grid), dim3(256), 0, s, (const u64*)h->d_k1.p, n, (const u64*)h->d_u1.p,
                           (const u64*)d_m1, (u32*)h->d_first.p);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(s));
        if (copy_d2h(&m1, d_m1, sizeof(u64))) return 1;
        h->h_key1.resize(m1); h->h_count1.resize(m1); h->h_first1.resize(m1); h->h_tmp32.resize(m1);
        if (copy_d2h(h->h_key1.data(), h->d_u1.p, m1 * sizeof(u64))) return 1;
        if (copy_d2h(h->h_tmp32.data(), h->d_c1.p, m1 * sizeof(u32))) return 1;
        { u64* dst = h->h_count1.data(); const u32* src = h->h_tmp32.data(); th_par_for(m1, [=](u64 lo, u64 hi) { for (u64 i = lo; i < hi; ++i) dst[i] = src[i]; }); }
        if (copy_d2h(h->h_tmp32.data(), h->d_first.p, m1 * sizeof(u32))) return 1;
        { u64* dst = h->h_first1.data(); const u32* src = h->h_tmp32.data(); th_par_for(m1, [=](u64 lo, u64 hi) { for (u64 i = lo; i < hi; ++i) dst[i] = src[i]; }); }
        // the absent records were counted under the fill key
        const u64 n_absent = n - h->n_present1;
        if (n_absent) {
            u64 fill = 0;
            if (copy_d2h(&fill, (const u64*)h->d_k1.p + h->first_present1, sizeof(u64))) return 1;
            const size_t at = (size_t)(std::lower_bound(h->h_key1.begin(), h->h_key1.end(), fill) - h->h_key1.begin());
            if (at >= m1 || h->h_key1[at] != fill || h->h_count1[at] <= n_absent) return set_err("tag histogram: fill key lost");
            h->h_count1[at] -= n_absent;
        }
        r->key1 = (const uint64_t*)h->h_key1.data(); r->count1 = (const uint64_t*)h->h_count1.data(); r->first1 = (const uint64_t*)h->h_first1.data(); r->n1 = m1;
        return 0;
    }

    // pair mode: distinct key2, dense pair codes, distinct pairs
    if (h->d_u2.ensure(nb) || h->d_c2.ensure(n * sizeof(u32)) || h->d_code.ensure(nb) || h->d_up.ensure(nb) ||
        h->d_cp.ensure(n * sizeof(u32))) return 1;
    if (th_distinct(h, (const u64*)h->d_k2.p, h->first_present2, h->or2 ^ h->and2, (u64*)h->d_u2.p, (u32*)h->d_c2.p, d_m2, s)) return 1;
    hipLaunchKernelGGL(th_pair_code_kernel, dim3(grid), dim3(256), 0, s, (const u64*)h->d_k1.p, (const u64*)h->d_k2.p, n,
                       (const u64*)h->d_u1.p, (const u64*)d_m1, (const u64*)h->d_u2.p, (const u64*)d_m2, (u64*)h->d_code.p);
    HIP_OK(hipGetLastError());
    u64 m2 = 0;
    HIP_OK(hipStreamSynchronize(s));
    if (copy_d2h(&m1, d_m1, sizeof(u64)) || copy_d2h(&m2, d_m2, sizeof(u64))) return 1;
    // digits of the pair code that can vary: rank2 in the low word, rank1 + 1 in the high word
    const u64 varying = ((1ull << bits_for(m2 ? m2 - 1 : 0)) - 1) | (((1ull << bits_for(m1)) - 1) << 32);
    if (th_distinct(h, (const u64*)h->d_code.p, h->first_valid_pair, varying, (u64*)h->d_up.p, (u32*)h->d_cp.p, d_mp, s)) return 1;
    HIP_OK(hipMemsetAsync(h->d_first.p, 0xff, n * sizeof(u32), s));
    hipLaunchKernelGGL(th_first_index_kernel, dim3(grid), dim3(256), 0, s, (const u64*)h->d_code.p, n, (const u64*)h->d_up.p,
                       (const u64*)d_mp, (u32*)h->d_first.p);
    HIP_OK(hipGetLastError());
    u64 mp = 0;
    HIP_OK(hipStreamSynchronize(s));
    if (copy_d2h(&mp, d_mp, sizeof(u64))) return 1;
    h->h_key1.resize(m1); h->h_u2.resize(m2); h->h_up.resize(mp);
    if (copy_d2h(h->h_key1.data(), h->d_u1.p, m1 * sizeof(u64))) return 1;
    if (copy_d2h(h->h_u2.data(), h->d_u2.p, m2 * sizeof(u64))) return 1;
    if (copy_d2h(h->h_up.data(), h->d_up.p, mp * sizeof(u64))) return 1;
    h->h_pair_count.resize(mp); h->h_pair_first.resize(mp); h->h_pair_k1.resize(mp); h->h_pair_key2.resize(mp);
    h->h_tmp32.resize(mp);
    if (copy_d2h(h->h_tmp32.data(), h->d_cp.p, mp * sizeof(u32))) return 1;
    { u64* dst = h->h_pair_count.data(); const u32* src = h->h_tmp32.data(); th_par_for(mp, [=](u64 lo, u64 hi) { for (u64 i = lo; i < hi; ++i) dst[i] = src[i]; }); }
    if (copy_d2h(h->h_tmp32.data(), h->d_first.p, mp * sizeof(u32))) return 1;
    { u64* dst = h->h_pair_first.data(); const u32* src = h->h_tmp32.data(); th_par_for(mp, [=](u64 lo, u64 hi) { for (u64 i = lo; i < hi; ++i) dst[i] = src[i]; }); }
    const u64 n_invalid = n - h->n_valid;
    if (n_invalid) {
        u64 fill = 0;
        if (copy_d2h(&fill, (const u64*)h->d_code.p + h->first_valid_pair, sizeof(u64))) return 1;
        const size_t at = (size_t)(std::lower_bound(h->h_up.begin(), h->h_up.end(), fill) - h->h_up.begin());
        if (at >= mp || h->h_up[at] != fill || h->h_pair_count[at] <= n_invalid) return set_err("tag histogram: fill pair lost");
        h->h_pair_count[at] -= n_invalid;
    }
    // key1 values that never occur in a valid pair (their key2 was always absent) drop out of level 1
    h->h_count1.assign(m1, 0); h->h_first1.assign(m1, ~0ull);
    {
        // the pairs are sorted by code, i.e. grouped by key1: a slice starts where key1 changes, so that the level-1 sums of
        // one CB are one thread's
        const u64* up = h->h_up.data(); const u64* u2 = h->h_u2.data(); const u64 *pc = h->h_pair_count.data(), *pf = h->h_pair_first.data();
        u32* pk1 = h->h_pair_k1.data(); u64 *pk2 = h->h_pair_key2.data(), *c1 = h->h_count1.data(), *f1 = h->h_first1.data();
        int bad = 0; int* badp = &bad;
        th_par_for(mp, [=](u64 lo, u64 hi) {
            // both ends move forward to the next change of key1 (the same rule on either side of a boundary)
            while (lo > 0 && lo < mp && (up[lo] >> 32) == (up[lo - 1] >> 32)) ++lo;
            while (hi > 0 && hi < mp && (up[hi] >> 32) == (up[hi - 1] >> 32)) ++hi;
            for (u64 i = lo; i < hi; ++i) {
                const u64 c = up[i];
                const u32 a = (u32)(c >> 32) - 1, b = (u32)c;
                if (a >= m1 || b >= m2) { __atomic_store_n(badp, 1, __ATOMIC_RELAXED); return; }
                pk1[i] = a; pk2[i] = u2[b];
                c1[a] += pc[i];
                f1[a] = std::min(f1[a], pf[i]);
            }
        });
        if (bad) return set_err("tag histogram: pair code out of range");
    }
    r->key1 = (const uint64_t*)h->h_key1.data(); r->count1 = (const uint64_t*)h->h_count1.data(); r->first1 = (const uint64_t*)h->h_first1.data(); r->n1 = m1;
    r->pair_k1 = h->h_pair_k1.data(); r->pair_key2 = (const uint64_t*)h->h_pair_key2.data(); r->pair_count = (const uint64_t*)h->h_pair_count.data();
    r->pair_first = (const uint64_t*)h->h_pair_first.data(); r->n_pairs = mp;
    return 0;
} FASTF_CATCH_INT
