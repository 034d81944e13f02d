// gpu_records.hpp — device side of the BAM front end, second stage: the record hop, the tag extraction and the key packing
// on the bytes the device has just inflated (gpu_inflate.hpp), so that those bytes never travel back to the host.
//
// Reference steps replaced (bam2db_ds.c:360-417): sam_read1() walking the block_size chain, bam_aux_get() x4 / bam_aux2Z /
// bam_aux2i scanning the aux block, hash() + encode_DNA() on the tag strings — restated on the host in host_io.c
// (hop_window, pack_record) and host_prims.c (fastf_keydict_pack, fastf_pack_umi).  This file restates THOSE, byte for
// byte: the per-record functions below compile for the host as well (GR_HOST, one lane) and are compared with the host
// packer on the CPU (tests/test_gpu_records_host.py); the kernels are compared with it on the device.
//
//   hop     the block_size chain is serial by nature.  The inflated window is cut into segments of GR_SEG bytes; one
//           wavefront per segment guesses the first record start in its segment (64 candidate offsets at a time, an offset
//           counts when three structurally valid, plausible records follow each other from it) and one lane hops from there
//           to the end of the segment, leaving the record offsets.  Segment 0 starts at the true chain.  A second kernel
//           accepts the guesses only if every chain ENDS exactly where the next one BEGINS — two chains that share an offset
//           are the same chain from there on, so by induction every accepted offset lies on the true chain; anything else
//           (a record longer than a segment, a wrong guess) makes the caller fall back to the host hop for that window.
//   pack    one lane per record: fixed fields -> aux block -> first CB / xf / GX / UB tag -> exact 64-bit keys (DNA form,
//           DNA+N form, ID form over the dictionary's prefix table) + 2-bit UMI + meta bits, 24 bytes per record straight
//           into the SoA the engine reads.  Strings that are none of these forms can only equal a listed string through the
//           ESCAPE dictionary; a reader whose lists registered an escape string keeps the host packer.
//   crc     CRC-32 of every inflated block, one wavefront per block (64 slices, combined with zlib's x^n mod P products),
//           against the block's trailer: what the host checked on the copied-back bytes.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef GR_HOST
#define GR_FN static inline
#define GR_HD static inline
#else
#define GR_FN __device__ __forceinline__
#define GR_HD __host__ __device__ __forceinline__
#endif

namespace gr {

constexpr uint32_t META_XF_OK_ = 1, META_HAS_UB_ = 2, META_UMI_NONNULL_ = 4, META_UMI_TOOLONG_ = 8, META_LEN_SHIFT_ = 4;
constexpr int MAX_PREFIX = 8, MAX_PREFIX_LEN = 32;

// the ID-form prefixes of one key dictionary (fastf_keydict_t): what fastf_keydict_pack needs besides the string itself
struct Dict {
    uint32_t n_prefix;
    uint32_t plen[MAX_PREFIX];
    uint64_t pid[MAX_PREFIX];
    uint8_t  ptext[MAX_PREFIX][MAX_PREFIX_LEN];
};

GR_FN uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
GR_FN uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

// base -> 2-bit code, 4 = not a base (encode_base, bam2db_ds.c:5-20: upper-case ACGT only)
GR_FN uint32_t base_code(uint8_t c) { return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u; }

// '-' canonical-decimal 0..254 behind the bases: returns suffix+1, or ~0u when malformed (s[n] must be '-')
GR_FN uint32_t dna_suffix(const uint8_t* s, uint32_t n, uint32_t len) {
    if (s[n] != '-') return ~0u;
    const uint32_t d = n + 1, nd = len - d;
    if (nd == 0 || nd > 3) return ~0u;
    if (s[d] == '0' && nd > 1) return ~0u;
    uint32_t v = 0;
    for (uint32_t i = d; i < len; i++) {
        if (s[i] < '0' || s[i] > '9') return ~0u;
        v = v * 10 + (uint32_t)(s[i] - '0');
    }
    return v > 254 ? ~0u : v + 1;
}

// host_prims.c pack_dna
GR_FN uint64_t pack_dna(const uint8_t* s, uint32_t len) {
    uint64_t bases = 0;
    uint32_t n = 0;
    while (n < len) {
        const uint64_t c = base_code(s[n]);
        if (c & 4) break;
        if (n == 24) return 0;
        bases = (bases << 2) | c;
        n++;
    }
    if (n == 0) return 0;
    bases <<= 48 - 2 * n;
    uint64_t suffix = 0;
    if (n < len) {
        const uint32_t sf = dna_suffix(s, n, len);
        if (sf == ~0u) return 0;
        suffix = sf;
    }
    return (1ull << 62) | ((uint64_t)n << 57) | (suffix << 49) | bases;
}

// host_prims.c pack_dna_n
GR_FN uint64_t pack_dna_n(const uint8_t* s, uint32_t len) {
    uint64_t bases = 0;
    uint32_t n = 0;
    int has_n = 0;
    while (n < len) {
        uint64_t c = base_code(s[n]);
        if (c & 4) {
            if (s[n] != 'N') break;
            c = 4; has_n = 1;
        }
        if (n == 16) return 0;
        bases = (bases << 3) | c;
        n++;
    }
    if (n == 0 || !has_n) return 0;
    bases <<= 48 - 3 * n;
    uint64_t suffix = 0;
    if (n < len) {
        const uint32_t sf = dna_suffix(s, n, len);
        if (sf == ~0u) return 0;
        suffix = sf;
    }
    return (3ull << 62) | (1ull << 61) | ((uint64_t)n << 56) | (suffix << 48) | bases;
}

// host_prims.c fastf_keydict_pack, for a dictionary without escape strings
GR_FN uint64_t pack_key(const Dict& d, const uint8_t* s, uint32_t len) {
    uint64_t k = pack_dna(s, len);
    if (k) return k;
    if ((k = pack_dna_n(s, len))) return k;
    uint32_t nd = 0;
    while (nd < len && s[len - 1 - nd] >= '0' && s[len - 1 - nd] <= '9') nd++;
    if (nd == 0 || nd > 13) return 0;
    uint64_t v = 0;
    for (uint32_t i = len - nd; i < len; i++) v = v * 10 + (uint64_t)(s[i] - '0');
    const uint32_t plen = len - nd;
    for (uint32_t q = 0; q < d.n_prefix; q++) {
        if (d.plen[q] != plen) continue;
        bool same = true;
        for (uint32_t i = 0; i < plen; i++) same = same && d.ptext[q][i] == s[i];
        if (same) return (2ull << 62) | (d.pid[q] << 48) | ((uint64_t)nd << 44) | v;
    }
    return 0;
}

// host_prims.c fastf_pack_umi
GR_FN uint32_t pack_umi(const uint8_t* ub, uint32_t len, uint32_t* umi_out) {
    uint32_t meta = META_HAS_UB_;
    *umi_out = 0;
    if (len > 16) return meta | META_UMI_TOOLONG_;
    uint32_t packed = 0, bad = 0;
    for (uint32_t i = 0; i < len; i++) {
        const uint32_t code = base_code(ub[i]);
        bad |= code;
        packed = (packed << 2) | (code & 3);
    }
    if (len) packed <<= 32 - 2 * len;
    if (!(bad & 4)) { meta |= META_UMI_NONNULL_; *umi_out = packed; }
    meta |= ((len + 3) / 4) << META_LEN_SHIFT_;
    return meta;
}

// host_io.c aux_skip: size of one aux value (p at the type byte), -1 when malformed
GR_FN long aux_skip(const uint8_t* p, const uint8_t* end) {
    if (p >= end) return -1;
    switch (*p) {
    case 'A': case 'c': case 'C': return 2;
    case 's': case 'S': return 3;
    case 'i': case 'I': case 'f': return 5;
    case 'd': return 9;
    case 'Z': case 'H': {
        for (const uint8_t* q = p + 1; q < end; q++) if (*q == 0) return (long)(q - p) + 1;
        return -1;
    }
    case 'B': {
        if (end - p < 6) return -1;
        uint64_t esz;
        switch (p[1]) { case 'c': case 'C': esz = 1; break; case 's': case 'S': esz = 2; break;
                        case 'i': case 'I': case 'f': esz = 4; break; default: return -1; }
        const uint64_t cnt = rd32(p + 2), tot = 6 + cnt * esz;
        return tot <= (uint64_t)(end - p) ? (long)tot : -1;
    }
    default: return -1;
    }
}

// host_io.c aux_int (bam_aux2i: integer types convert, anything else yields 0)
GR_FN int64_t aux_int(const uint8_t* p) {
    switch (*p) {
    case 'c': return (int8_t)p[1];
    case 'C': return p[1];
    case 's': return (int16_t)rd16(p + 1);
    case 'S': return rd16(p + 1);
    case 'i': return (int32_t)rd32(p + 1);
    case 'I': return rd32(p + 1);
    default: return 0;
    }
}

// bam_aux2Z (htslib): 'Z' and 'H' values are NUL-terminated strings
GR_FN bool aux_is_string(const uint8_t* p) { return *p == 'Z' || *p == 'H'; }
GR_FN uint32_t zlen(const uint8_t* s) { uint32_t n = 0; while (s[n]) n++; return n; }      // (aux_skip has found the NUL)

// host_io.c pack_worker + pack_record: the record whose block_size field is at p (a structurally valid record: rec_end)
GR_FN void pack_record(const uint8_t* p, const Dict& cells, const Dict& feats,
                       uint64_t* cb_key, uint64_t* gx_key, uint32_t* umi, uint32_t* meta, uint32_t* no_xf, uint32_t* no_gx) {
    const uint8_t* rec = p + 4;
    const uint32_t bs = rd32(p);
    const uint32_t l_read_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
    uint64_t fixed = 32ull + l_read_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
    if (fixed > bs) fixed = bs;
    const uint8_t* aux = rec + fixed;
    const uint8_t* end = rec + bs;
    const uint8_t *cb = nullptr, *xf = nullptr, *gx = nullptr, *ub = nullptr;
    while (end - aux >= 3) {
        const uint8_t* val = aux + 2;
        const long sz = aux_skip(val, end);
        if (sz < 0) break;
        if (aux[0] == 'C' && aux[1] == 'B') { if (!cb) cb = val; }
        else if (aux[0] == 'x' && aux[1] == 'f') { if (!xf) xf = val; }
        else if (aux[0] == 'G' && aux[1] == 'X') { if (!gx) gx = val; }
        else if (aux[0] == 'U' && aux[1] == 'B') { if (!ub) ub = val; }
        aux = val + sz;
    }
    uint32_t m = 0;
    *cb_key = 0; *gx_key = 0; *umi = 0;
    if (cb && aux_is_string(cb)) *cb_key = pack_key(cells, cb + 1, zlen(cb + 1));
    if (xf) { const int64_t q = aux_int(xf); if (q == 25 || q == 17) m |= META_XF_OK_; }
    else if (*cb_key) (*no_xf)++;
    if (gx && aux_is_string(gx)) *gx_key = pack_key(feats, gx + 1, zlen(gx + 1));
    else if (!gx && (m & META_XF_OK_) && *cb_key) (*no_gx)++;
    if (ub && aux_is_string(ub)) m |= pack_umi(ub + 1, zlen(ub + 1), umi);
    *meta = m;
}

// host_io.c rec_end: offset just past the record at o if a whole, structurally valid record starts there; 0 otherwise
GR_FN uint64_t rec_end(const uint8_t* buf, uint64_t o, uint64_t end) {
    if (end - o < 36) return 0;
    const uint8_t* p = buf + o;
    const uint32_t bs = rd32(p);
    if (bs < 32 || (uint64_t)bs + 4 > end - o) return 0;
    const uint64_t fixed = 32ull + p[12] + 4ull * rd16(p + 16) + ((uint64_t)rd32(p + 20) + 1) / 2 + rd32(p + 20);
    return fixed <= bs ? o + 4 + (uint64_t)bs : 0;
}

// host_io.c rec_plausible: does the record at o look like an alignment record (only used to GUESS chain starts)
GR_FN bool rec_plausible(const uint8_t* buf, uint64_t o, uint64_t end, uint32_t n_ref) {
    if (!rec_end(buf, o, end)) return false;
    const uint8_t* p = buf + o;
    const int32_t ref = (int32_t)rd32(p + 4), pos = (int32_t)rd32(p + 8), nref = (int32_t)rd32(p + 24), npos = (int32_t)rd32(p + 28);
    const uint32_t l_name = p[12];
    if (ref < -1 || (uint32_t)(ref + 1) > n_ref || nref < -1 || (uint32_t)(nref + 1) > n_ref || pos < -1 || npos < -1) return false;
    if (l_name < 1 || p[36 + l_name - 1] != 0) return false;
    for (uint32_t i = 0; i + 1 < l_name; i++) if (p[36 + i] < 33 || p[36 + i] > 126) return false;
    return true;
}

// a chain of plausible records that stops at q because the WINDOW ends there: less than a fixed part is left, or the
// record at q announces a sane size that reaches beyond `end` (the device's share of a window ends in the middle of a
// record as a rule; that record is the host's).  Only used to GUESS chain starts, like rec_plausible.
GR_FN bool chain_meets_window_end(const uint8_t* buf, uint64_t q, uint64_t end) {
    if (end - q < 36) return true;
    const uint32_t bs = rd32(buf + q);
    return bs >= 32 && bs <= (64u << 20) && (uint64_t)bs + 4 > end - q;
}

// ---- CRC-32 (IEEE 802.3, reflected, as zlib): slices of a block are hashed side by side and combined ----
constexpr uint32_t CRC_POLY = 0xEDB88320u;
// a(x) * b(x) mod P(x) in the reflected representation (zlib crc32.c multmodp)
GR_HD uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = b & 1 ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(n * 2^k) mod P(x); x2n[i] = x^(2^i) mod P(x) (zlib crc32.c x2nmodp)
GR_FN uint32_t crc_x2nmodp(uint64_t n, unsigned k, const uint32_t* x2n) {
    uint32_t p = 1u << 31;
    while (n) {
        if (n & 1) p = crc_multmodp(x2n[k & 31], p);
        n >>= 1;
        k++;
    }
    return p;
}
// crc of A || B from crc(A), crc(B) and len(B) (zlib crc32_combine)
GR_FN uint32_t crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b, const uint32_t* x2n) {
    return crc_multmodp(crc_x2nmodp(len_b, 3, x2n), crc_a) ^ crc_b;
}
// plain byte-at-a-time CRC of one slice over the 256-entry table
GR_FN uint32_t crc_bytes(const uint32_t* tab, const uint8_t* p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

}  // namespace gr

#ifndef GR_HOST
// ------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------
constexpr uint32_t GR_SEG = 32u << 10;                     // bytes per hop segment
constexpr uint32_t GR_SEG_RECS = GR_SEG / 36 + 2;          // most records that can START in one segment
constexpr uint64_t GR_NONE = ~0ull;
constexpr int GR_PLAUSIBLE = 3;                            // HOP_PLAUSIBLE of host_io.c

struct GrSeg { uint64_t first, exit_off; uint32_t n, base; };       // chain of one segment: first record, offset behind its last, count; base: records in front

// hop: one wavefront per segment of [start, end) (offsets into buf).  offs: GR_SEG_RECS slots per segment.
__global__ __launch_bounds__(64) void gr_hop_kernel(const uint8_t* __restrict__ buf, uint64_t start, uint64_t end, uint32_t n_ref,
                                                    GrSeg* __restrict__ seg, uint32_t* __restrict__ offs) {
    const uint32_t s = blockIdx.x;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint64_t lo = start + (uint64_t)s * GR_SEG, hi = lo + GR_SEG < end ? lo + GR_SEG : end;
    uint64_t first = GR_NONE;
    if (s == 0) first = start;                                      // the true chain enters here
    else {
        for (uint64_t o0 = lo; o0 < hi && first == GR_NONE; o0 += 64) {
            const uint64_t o = o0 + (uint64_t)lane;
            bool ok = false;
            if (o < hi) {
                uint64_t q = o; int k = 0;
                while (k < GR_PLAUSIBLE && gr::rec_plausible(buf, q, end, n_ref)) { q = gr::rec_end(buf, q, end); k++; }
                ok = k == GR_PLAUSIBLE || (k > 0 && gr::chain_meets_window_end(buf, q, end));   // fewer only at the very end of the window
            }
            const uint64_t m = __ballot(ok);
            if (m) first = o0 + (uint64_t)__builtin_ctzll(m);
        }
    }
    uint32_t n = 0;
    uint64_t o = first;
    if (first != GR_NONE && lane == 0) {
        uint32_t* out = offs + (size_t)s * GR_SEG_RECS;
        while (o < hi) {
            const uint64_t e = gr::rec_end(buf, o, end);
            if (!e) break;
            out[n++] = (uint32_t)(o - lo);
            o = e;
        }
        seg[s].first = first; seg[s].exit_off = o; seg[s].n = n; seg[s].base = 0;
    } else if (lane == 0) {
        seg[s].first = GR_NONE; seg[s].exit_off = lo; seg[s].n = 0; seg[s].base = 0;
    }
}

// stitch: one workgroup.  result[0] = status (0 = every chain ends where the next begins), [1] = records, [2] = offset
// behind the last complete record (the hand-over point: an incomplete record, or `end`), [3..4] = no_xf / no_gx (pack adds)
__global__ __launch_bounds__(1024) void gr_stitch_kernel(GrSeg* __restrict__ seg, uint32_t n_seg, uint64_t end, unsigned long long* __restrict__ result) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry, s_bad;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_carry = 0; s_bad = 0; }
    __syncthreads();
    for (uint32_t s0 = 0; s0 < n_seg; s0 += 1024) {
        const uint32_t s = s0 + threadIdx.x;
        uint32_t v = 0;
        if (s < n_seg) {
            v = seg[s].n;
            // the chain in front must end exactly at this chain's first record; a chain that stopped early (incomplete or
            // invalid record) is only legal in the last segment
            if (s > 0 && seg[s - 1].exit_off != seg[s].first) atomicOr(&s_bad, 1u);
            if (seg[s].first == GR_NONE) atomicOr(&s_bad, 1u);
        }
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t off = s_carry;
        for (int i = 0; i < w; ++i) off += s_w[i];
        if (s < n_seg) seg[s].base = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result[0] = s_bad; result[1] = s_carry; result[2] = n_seg ? seg[n_seg - 1].exit_off : end; result[3] = 0; result[4] = 0;
    }
}

// repair: one wavefront, run when the stitch found chains that do not meet.  A segment whose guessed chain start is not where
// the chain in front of it ends (bytes inside a record that look like three records in a row; a record longer than a segment,
// inside which no start can be guessed at all) is walked again from that end — the true chain, by induction from segment 0,
// whose chain begins at `start` — until a segment's own guess is met again.  The scan for the next such segment takes 64
// segments a step; the walk itself is lane 0's, serial like the chain.  gr_stitch_kernel then runs again (bases, status).
__global__ __launch_bounds__(64) void gr_repair_kernel(const uint8_t* __restrict__ buf, uint64_t start, uint64_t end, GrSeg* __restrict__ seg,
                                                       uint32_t* __restrict__ offs, uint32_t n_seg, unsigned long long* __restrict__ n_repaired) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t s = 1, repaired = 0;
    while (s < n_seg) {
        const uint32_t idx = s + (uint32_t)lane;
        bool bad = false;
        if (idx < n_seg) { const uint64_t f = seg[idx].first; bad = f == GR_NONE || seg[idx - 1].exit_off != f; }
        const uint64_t m = __ballot(bad);
        if (!m) { s += 64; continue; }
        uint32_t sb = s + (uint32_t)__builtin_ctzll(m);
        uint32_t next = 0;                                            // (lane 0) where the scan goes on
        if (lane == 0) {
            uint64_t o = seg[sb - 1].exit_off;                        // the true chain enters segment sb here
            for (;;) {
                const uint64_t lo = start + (uint64_t)sb * GR_SEG, hi = lo + GR_SEG < end ? lo + GR_SEG : end;
                uint32_t* out = offs + (size_t)sb * GR_SEG_RECS;
                const uint64_t first = o;
                uint32_t n = 0;
                while (o >= lo && o < hi) {                            // (o >= hi: a record that began earlier runs through the whole segment)
                    const uint64_t e = gr::rec_end(buf, o, end);
                    if (!e) break;
                    out[n++] = (uint32_t)(o - lo);
                    o = e;
                }
                seg[sb].first = first; seg[sb].exit_off = o; seg[sb].n = n; seg[sb].base = 0;
                ++repaired;
                ++sb;
                if (sb >= n_seg || seg[sb].first == o) break;          // the next segment's own chain starts where this one ends: in step again
            }
            next = sb + 1;
        }
        s = (uint32_t)__shfl((int)next, 0, 64);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");            // lane 0's table entries, before the other lanes read them
    }
    if (lane == 0) *n_repaired = repaired;
}

// pack: one workgroup (two wavefronts) per segment, a lane per record.  The segment's bytes — and GR_STAGE_OVER beyond it, for
// the records that begin in it and end behind it — are staged in LDS with 16-byte loads first: a record is then parsed byte by
// byte out of LDS (a lane walks ~15 tags, a few hundred dependent byte reads: from memory they were the kernel's whole time,
// 120 GB/s of a 3 GB window).  A record that reaches beyond what is staged (longer than GR_STAGE_OVER) is read from memory.
constexpr uint32_t GR_STAGE_OVER = 4096, GR_STAGE = GR_SEG + GR_STAGE_OVER + 16, GR_PACK_THREADS = 128;
__global__ __launch_bounds__(GR_PACK_THREADS) void gr_pack_kernel(const uint8_t* __restrict__ buf, uint64_t start, uint64_t end, const GrSeg* __restrict__ seg,
                                                     const uint32_t* __restrict__ offs, gr::Dict cells, gr::Dict feats,
                                                     unsigned long long* __restrict__ cb_key, unsigned long long* __restrict__ gx_key,
                                                     uint32_t* __restrict__ umi, uint32_t* __restrict__ meta, uint64_t cap,
                                                     unsigned long long* __restrict__ result) {
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[GR_STAGE];
    const uint32_t s = blockIdx.x;
    const GrSeg g = seg[s];
    if (g.n == 0) return;                                             // (workgroup-uniform: a segment inside one long record)
    const uint64_t lo = start + (uint64_t)s * GR_SEG;
    // staged: [a0, a0 + n_stage) of the window, a0 = lo rounded down to 16 bytes (buf itself is an allocation's start)
    const uint64_t a0 = lo & ~15ull;
    const uint32_t shift = (uint32_t)(lo - a0);
    const uint64_t want_end = lo + GR_SEG + GR_STAGE_OVER < end ? lo + GR_SEG + GR_STAGE_OVER : end;
    const uint32_t n_stage = (uint32_t)(want_end - a0);               // <= GR_STAGE - 1
    for (uint32_t i = threadIdx.x * 16u; i < n_stage; i += GR_PACK_THREADS * 16u)      // (the last 16 bytes may run past `end`: inside the allocation, never parsed)
        *reinterpret_cast<uint4*>(s_buf + i) = *reinterpret_cast<const uint4*>(buf + a0 + i);
    __syncthreads();
    const uint32_t* in = offs + (size_t)s * GR_SEG_RECS;
    uint32_t no_xf = 0, no_gx = 0;
    for (uint32_t k = threadIdx.x; k < g.n; k += GR_PACK_THREADS) {
        const uint64_t i = (uint64_t)g.base + k;
        if (i >= cap) break;
        uint64_t c, x; uint32_t u, m;
        const uint32_t at = shift + in[k];                            // the record's block_size field in s_buf
        const uint32_t bs = gr::rd32(s_buf + at);
        if ((uint64_t)at + 4u + bs <= n_stage) gr::pack_record(s_buf + at, cells, feats, &c, &x, &u, &m, &no_xf, &no_gx);
        else gr::pack_record(buf + lo + in[k], cells, feats, &c, &x, &u, &m, &no_xf, &no_gx);
        cb_key[i] = c; gx_key[i] = x; umi[i] = u; meta[i] = m;
    }
    if (no_xf) atomicAdd(&result[3], (unsigned long long)no_xf);
    if (no_gx) atomicAdd(&result[4], (unsigned long long)no_gx);
}

// crc: one wavefront per block; status[b] |= 2 when the CRC-32 of the inflated bytes differs from the block's trailer
struct GrCrcBlock { uint64_t uoff; uint32_t isize, crc; };
__global__ __launch_bounds__(64) void gr_crc_kernel(const GrCrcBlock* __restrict__ blk, uint32_t n_blk, const uint8_t* __restrict__ out,
                                                    const uint32_t* __restrict__ tab_g, uint8_t* __restrict__ status) {
    __shared__ uint32_t tab[256];
    __shared__ uint32_t x2n[32];
    const uint32_t b = blockIdx.x;
    if (b >= n_blk) return;
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (int i = lane; i < 256; i += 64) tab[i] = tab_g[i];
    if (lane < 32) x2n[lane] = tab_g[256 + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const GrCrcBlock k = blk[b];
    if (!k.isize) return;
    const uint32_t per = (k.isize + 63) / 64;                       // bytes per lane; the last lanes may hold less or nothing
    const uint32_t lo = (uint32_t)lane * per < k.isize ? (uint32_t)lane * per : k.isize;
    const uint32_t hi = lo + per < k.isize ? lo + per : k.isize;
    uint32_t crc = gr::crc_bytes(tab, out + k.uoff + lo, hi - lo);   // (an empty slice hashes to 0, the neutral element of combine)
    uint32_t len = hi - lo;
    // tree: lane l takes over lane l + d (the bytes behind its own)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t crc_r = (uint32_t)__shfl_down((int)crc, d, 64), len_r = (uint32_t)__shfl_down((int)len, d, 64);
        if ((lane & (2 * d - 1)) == 0 && lane + d < 64) {
            if (len_r) crc = len ? gr::crc_combine(crc, crc_r, len_r, x2n) : crc_r;
            len += len_r;
        }
    }
    if (lane == 0 && crc != k.crc) status[b] |= 2;
}
#endif  // !GR_HOST
