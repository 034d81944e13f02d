// gpu_inflate.hpp — raw DEFLATE (RFC 1951) decode of ONE BGZF block by ONE wavefront.
//
// The step in front of the bam2db hot path is sam_read1() through htslib/zlib (bam2db_ds.c:360): inflate of
// independent <= 64 KiB BGZF blocks.  On the host that is the longest stage of an end-to-end run;
// here the blocks of a window are decoded on the device, one wavefront per block:
//   * control flow is wave-uniform: all 64 lanes run the same symbol loop on the same bit buffer (the SIMD executes 64
//     lanes anyway), and the lanes split the work wherever there is any — table fill, match copy, output flush;
//   * Huffman tables live in LDS: a 10-bit (literal/length) and an 8-bit (distance) direct table of u16 entries; the rare
//     longer codes go through the canonical count/offset walk (one bit at a time) over the sorted symbol list;
//   * the compressed bytes are fetched 256 at a time, one dword per lane, into a register; the bit buffer is refilled
//     with v_readlane from it (no memory access on the symbol path), and the next 256 bytes are loaded a chunk ahead;
//   * the last 4 KiB of output live in an LDS ring: literals and matches write there, every completed 64-byte line is
//     flushed to memory by the whole wave in one store, and a match reads its source from the ring (distance <= 4032) or,
//     for the far ones (most matches of a BAM payload), from memory with L2-coherent loads — those bytes were stored at
//     least 62 line stores ago and the wave keeps at most four memory operations outstanding at that point
//     (s_waitcnt vmcnt(4)): 7.4 KB of LDS per wave, twenty blocks in flight per CU.
// The same source compiles for the host (GI_HOST: one lane) so that the decoder is fuzzed against zlib on the CPU
// (tests/test_gpu_inflate_host.py, tools/gi_host.cpp); the device build is checked against zlib on real BGZF blocks.
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef GI_HOST
#define GI_FN static inline
constexpr int GI_LANES = 1;
#define GI_LANE() 0
#define GI_COHERENT_LOAD8(p) (*(p))
#define GI_WAVE_SYNC() do { } while (0)
#define GI_UNIFORM(x) (x)
#define GI_UNIFORM64(x) (x)
#else
#define GI_FN __device__ __forceinline__
constexpr int GI_LANES = 64;
#define GI_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
// byte read that must see what this wave stored earlier: agent-scope load of the containing dword (bypasses the CU's L1)
// (the dword's address is formed by pointer arithmetic, not through an integer: a pointer that went through uintptr_t has lost
//  its address space and the access becomes a FLAT one, which counts on the LDS counter too — every wait for an LDS read then
//  also waits for these loads)
__device__ __forceinline__ uint8_t gi_coherent_load8(const uint8_t* p) {
    const uint32_t k = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(p - k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint8_t)(w >> (8 * k));
}
#define GI_COHERENT_LOAD8(p) gi_coherent_load8(p)
#define gi_lane0() (GI_LANE() == 0)
// a value every lane holds alike, moved to a scalar register: what follows from it (bit-buffer arithmetic, branches) runs
// on the scalar unit instead of 64 identical vector lanes
#ifdef GI_NO_UNIFORM
#define GI_UNIFORM(x) (x)
#define GI_UNIFORM64(x) (x)
#else
#define GI_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define GI_UNIFORM64(x) (((uint64_t)GI_UNIFORM((uint32_t)((uint64_t)(x) >> 32)) << 32) | (uint64_t)GI_UNIFORM((uint32_t)(x)))
#endif
// lanes of one wave exchange data through LDS without a workgroup barrier: keep the compiler from moving LDS accesses
// across the hand-over (the hardware runs a wave's LDS instructions in order)
#define GI_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

#ifndef GI_STATS_MATCH            /* tools: token statistics of the host build */
#define GI_STATS_MATCH(len, dist) do { } while (0)
#define GI_STATS_LIT() do { } while (0)
#endif

namespace gi {

constexpr int LIT_TB = 10, DIST_TB = 8;
#ifndef GI_RING_BYTES
#define GI_RING_BYTES 4096            /* output ring per wave (a power of two): matches that reach further back read memory */
#endif
constexpr uint32_t RING = GI_RING_BYTES, RMASK = RING - 1, RING_SAFE = RING - 64;
static_assert((RING & (RING - 1)) == 0 && RING >= 1024, "ring: a power of two");
constexpr uint16_t LONG_CODE = 0xFFFF, NO_CODE = 0xFFFE;      // both >= 0xFFFE: one compare on the symbol path

// error codes (0 = ok)
enum { OK = 0, E_BTYPE = 1, E_STORED = 2, E_LENS = 3, E_CODE = 4, E_DIST = 5, E_OVERRUN = 6, E_INPUT = 7, E_SIZE = 8 };

// per-wave working set (LDS on the device)
struct Work {
    uint16_t lit_tab[1 << LIT_TB];      // (symbol << 4) | code length; NO_CODE; LONG_CODE: longer than the table
    uint16_t dist_tab[1 << DIST_TB];    // same; also holds the 7-bit table of the code-length code while a header is read
    uint16_t lit_sorted[288], dist_sorted[32];   // symbols by (length, symbol): the canonical order, for the long-code walk
    uint16_t lit_count[16], dist_count[16];
    uint8_t lens[320];
    uint32_t t_cnt[16], t_next[16], t_offs[16];  // table construction scratch (kept out of registers: 48 live values otherwise)
    uint8_t ring[RING];
};

// Bit reader.  `pos` counts the dwords handed to the bit buffer, from the 4-byte-aligned address at or below the block's
// first byte; on the device the dwords come out of a per-lane register (lane l holds dword 64 c + l of chunk c).
struct Bits {
    const uint32_t* in32; uint32_t pos, n_words; uint64_t buf; uint32_t cnt;
#ifndef GI_HOST
    uint32_t cur, nxt;                         // this lane's dword of the current and of the next 256-byte chunk
#endif
};

#ifdef GI_HOST
GI_FN uint32_t fetch_word(Bits& b) { return b.in32[b.pos++]; }
GI_FN void bits_open(Bits& b, const uint8_t* in, uint32_t in_len) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(in);
    b.in32 = reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
    b.pos = 0; b.n_words = (uint32_t)((a & 3) + in_len + 3) / 4 + 2; b.buf = 0; b.cnt = 0;
}
#else
GI_FN uint32_t chunk_load(const Bits& b, uint32_t chunk) {
    const uint32_t i = chunk * 64u + (uint32_t)GI_LANE();
    return i < b.n_words ? __builtin_nontemporal_load(b.in32 + i) : 0u;   // (the words past the block are never consumed)
}
GI_FN uint32_t fetch_word(Bits& b) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)b.cur, (int)(b.pos & 63u));
    if ((++b.pos & 63u) == 0) { b.cur = b.nxt; b.nxt = chunk_load(b, (b.pos >> 6) + 1u); }
    return w;
}
GI_FN void bits_open(Bits& b, const uint8_t* in, uint32_t in_len) {
    const uint32_t k = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);
    b.in32 = reinterpret_cast<const uint32_t*>(in - k);       // (pointer arithmetic keeps the address space: global loads, not flat ones)
    b.pos = 0; b.n_words = GI_UNIFORM((uint32_t)(k + in_len + 3) / 4 + 2); b.buf = 0; b.cnt = 0;
    b.cur = chunk_load(b, 0); b.nxt = chunk_load(b, 1);
}
#endif

// at least 32 valid bits afterwards (may take words past the block: the caller's buffer is padded by 8 bytes; such bits
// are never consumed by a well-formed stream, and a malformed one is caught by the position check at the end)
GI_FN void refill(Bits& b) {
    if (b.cnt <= 32) { b.buf |= (uint64_t)fetch_word(b) << b.cnt; b.cnt += 32; }
}
GI_FN uint32_t peek(const Bits& b, int n) { return (uint32_t)b.buf & ((1u << n) - 1u); }
GI_FN void drop(Bits& b, int n) { b.buf >>= n; b.cnt -= (uint32_t)n; }
GI_FN uint32_t take(Bits& b, int n) { const uint32_t v = peek(b, n); drop(b, n); return v; }

GI_FN uint32_t rev(uint32_t c, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

// canonical code -> direct table over the low `tb` stream bits + sorted symbol list and per-length counts.
// Returns 0, or E_LENS for an over-subscribed set (or an incomplete one with more than one code).
GI_FN int build(Work& w, const uint8_t* lens, int n_sym, int tb, uint16_t* tab, uint16_t* sorted, uint16_t* count) {
    const int lane = GI_LANE();
    uint32_t* cnt = w.t_cnt; uint32_t* next = w.t_next; uint32_t* offs = w.t_offs;
    // per-length counts: lane l counts the symbols of length l (16 lanes; one lane walks all on the host)
    for (int l = lane; l < 16; l += GI_LANES) {
        uint32_t c = 0;
        for (int s = 0; s < n_sym; ++s) c += lens[s] == l;
        cnt[l] = c;
    }
    GI_WAVE_SYNC();
    const uint32_t n_codes = (uint32_t)n_sym - cnt[0];
    int left = 1;
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - (int)cnt[l]; if (left < 0) return E_LENS; }
    if (left > 0 && n_codes > 1) return E_LENS;
    if (lane == 0) {
        uint32_t code = 0, o = 0, prev = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + prev) << 1; next[l] = code; offs[l] = o; o += cnt[l]; prev = cnt[l]; }
    }
    for (int l = lane; l < 16; l += GI_LANES) count[l] = l ? (uint16_t)cnt[l] : (uint16_t)0;
    for (int i = lane; i < (1 << tb); i += GI_LANES) tab[i] = NO_CODE;
    GI_WAVE_SYNC();
    for (int s = 0; s < n_sym; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t c = next[l], at = offs[l];
        GI_WAVE_SYNC();
        if (lane == 0) { sorted[at] = (uint16_t)s; next[l] = c + 1; offs[l] = at + 1; }
        if (l <= tb) {
            const uint32_t r = rev(c, l);
            const uint16_t e = (uint16_t)((s << 4) | l);
            for (uint32_t i = r + ((uint32_t)lane << l); i < (1u << tb); i += (uint32_t)GI_LANES << l) tab[i] = e;
        } else if (lane == 0) {
            tab[rev(c >> (l - tb), tb)] = LONG_CODE;         // the first tb bits of the code, as they arrive in the stream
        }
        GI_WAVE_SYNC();
    }
    return OK;
}

// one symbol: direct table, or the count/offset walk for codes longer than the table (RFC 1951 3.2.2; one bit per step)
GI_FN int decode(Bits& b, const uint16_t* tab, int tb, const uint16_t* sorted, const uint16_t* count) {
    const uint32_t e = GI_UNIFORM(tab[peek(b, tb)]);
    if (e < 0xFFFEu) {                                    // the common case: the code is in the table
        drop(b, (int)(e & 15u));
        return (int)(e >> 4);
    }
    if (e == NO_CODE) return -1;
    uint32_t code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int l = 1; l <= 15; ++l) {
        code |= (uint32_t)bits & 1u; bits >>= 1;
        const uint32_t c = GI_UNIFORM(count[l]);
        if (code < first + c) { drop(b, l); return (int)GI_UNIFORM(sorted[index + (code - first)]); }
        index += c; first = (first + c) << 1; code <<= 1;
    }
    return -1;
}

// base value and number of extra bits of a length symbol (257..285 -> ls = 0..28) and of a distance symbol (0..29),
// RFC 1951 3.2.5, computed instead of looked up: a table in memory costs a load (and a wait for every store in flight)
// per match
GI_FN void len_code(uint32_t ls, uint32_t& base, uint32_t& extra) {
    if (ls < 8u) { base = 3u + ls; extra = 0; }
    else if (ls == 28u) { base = 258u; extra = 0; }
    else { extra = (ls - 4u) >> 2; base = 3u + ((4u + (ls & 3u)) << extra); }
}
GI_FN void dist_code(uint32_t ds, uint32_t& base, uint32_t& extra) {
    if (ds < 4u) { base = 1u + ds; extra = 0; }
    else { extra = (ds >> 1) - 1u; base = 1u + ((2u + (ds & 1u)) << extra); }
}

struct Out { uint8_t* out; uint32_t cap, op, flushed; uint32_t over; };   // over: output ran past cap (nothing is stored beyond it)

// every completed 64-byte line of the ring goes to memory, one byte per lane
GI_FN void flush_lines(Work& w, Out& o, bool all) {
    const int lane = GI_LANE();
    GI_WAVE_SYNC();
    o.op = GI_UNIFORM(o.op); o.flushed = GI_UNIFORM(o.flushed);          // (wave-uniform by construction: tell the compiler)
    while (o.op - o.flushed >= 64u || (all && o.op > o.flushed)) {
        const uint32_t n = o.op - o.flushed >= 64u ? 64u : o.op - o.flushed;
        if (o.flushed + n > o.cap) { o.over = 1; o.flushed += n; continue; }      // malformed stream: more output than ISIZE
        for (uint32_t i = (uint32_t)lane; i < n; i += GI_LANES) o.out[o.flushed + i] = w.ring[(o.flushed + i) & RMASK];
        o.flushed += n;
    }
}

GI_FN int copy_match(Work& w, Out& o, uint32_t len, uint32_t dist) {
    const int lane = GI_LANE();
    len = GI_UNIFORM(len); dist = GI_UNIFORM(dist); o.op = GI_UNIFORM(o.op);   // wave-uniform by construction: scalar control flow below
    if (dist > o.op) return E_DIST;                        // (dist >= 1 by construction)
    if (o.over) return E_OVERRUN;
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI_NO_COPY)
    // timing probe (wrong output): the token is decoded, its bytes are never copied — what the symbol loop costs on its own
    o.op += len; o.flushed = o.op & ~63u;
    return OK;
#endif
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI_NEAR_ONLY)
    // timing probe (wrong output): every far match reads the ring instead of memory — what the far sources' round trips cost
    if (dist > RING_SAFE) dist = (dist & (RING_SAFE / 2 - 1)) + 64u;
#endif
#ifndef GI_HOST
    // The source left the ring, i.e. it went to memory at least 62 line stores ago.  Memory operations of a wave complete
    // in order, so with at most four of them still outstanding every older store has landed — in BAM payloads most matches
    // reach further back than the ring, and waiting for ALL stores (vmcnt(0)) here cost more than the decoding itself.
    if (dist > RING_SAFE) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
    while (len) {
        const uint32_t n = len < (uint32_t)GI_LANES ? len : (uint32_t)GI_LANES;
        if (dist > RING_SAFE) {                            // far: from memory (dist >= n always here)
            for (uint32_t i = (uint32_t)lane; i < n; i += GI_LANES) w.ring[(o.op + i) & RMASK] = GI_COHERENT_LOAD8(o.out + (o.op - dist + i));
        } else if (dist >= n) {
            for (uint32_t i = (uint32_t)lane; i < n; i += GI_LANES) w.ring[(o.op + i) & RMASK] = w.ring[(o.op - dist + i) & RMASK];
        } else {                                           // the match overlaps its own output: the pattern repeats
            for (uint32_t i = (uint32_t)lane; i < n; i += GI_LANES) {
                uint32_t r = i;                            // i mod dist without a division: dist < n <= 64, a few subtractions
                while (r >= dist) r -= dist;
                w.ring[(o.op + i) & RMASK] = w.ring[(o.op - dist + r) & RMASK];
            }
        }
        GI_WAVE_SYNC();
        o.op += n; len -= n;
        flush_lines(w, o, false);
    }
    return OK;
}

// out must hold `isize` bytes; `in` must be readable 8 bytes past in_len
GI_FN int inflate_block(Work& w, const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t isize) {
    static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int lane = GI_LANE();
    Bits b;
    bits_open(b, in, in_len);
    uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);         // bytes in front of `in` inside its first dword
    refill(b);
    drop(b, (int)(8u * lead));
    Out o{out, isize, 0, 0, 0};
    for (;;) {
        refill(b);
        const uint32_t final = take(b, 1), type = take(b, 2);
        if (type == 0) {                                                   // stored
            drop(b, (int)(b.cnt & 7u));
            refill(b);
            const uint32_t n = take(b, 16), nn = take(b, 16);
            if ((n ^ nn) != 0xFFFFu) return E_STORED;
            if (o.op + n > o.cap) return E_OVERRUN;
            // byte position of the next unread input byte, relative to `in`
            const uint32_t p = b.pos * 4u - b.cnt / 8u - lead;
            if (p + n > in_len) return E_INPUT;
            uint32_t left = n, q = p;
            while (left) {
                const uint32_t k = left < (uint32_t)GI_LANES ? left : (uint32_t)GI_LANES;
                for (uint32_t i = (uint32_t)lane; i < k; i += GI_LANES) w.ring[(o.op + i) & RMASK] = in[q + i];
                GI_WAVE_SYNC();
                o.op += k; q += k; left -= k;
                flush_lines(w, o, false);
            }
            // restart the bit reader behind the stored bytes
            bits_open(b, in + q, in_len - q);
            // (positions are now relative to in + q: keep the end check right by shrinking the block)
            in += q; in_len -= q;
            lead = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 3u);
            refill(b);
            drop(b, (int)(8u * lead));
        } else if (type == 1 || type == 2) {
            int n_lit, n_dist;
            if (type == 1) {                                               // fixed code (RFC 1951 3.2.6)
                for (int i = lane; i < 288; i += GI_LANES) w.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                for (int i = lane; i < 32; i += GI_LANES) w.lens[288 + i] = 5;
                GI_WAVE_SYNC();
                n_lit = 288; n_dist = 30;
                // (symbols 30, 31 of the fixed distance code never occur; building over 32 keeps the code complete)
                if (build(w, w.lens, 288, LIT_TB, w.lit_tab, w.lit_sorted, w.lit_count)) return E_LENS;
                if (build(w, w.lens + 288, 32, DIST_TB, w.dist_tab, w.dist_sorted, w.dist_count)) return E_LENS;
            } else {                                                       // dynamic code: the header
                n_lit = (int)take(b, 5) + 257; n_dist = (int)take(b, 5) + 1;
                const int n_cl = (int)take(b, 4) + 4;
                if (n_lit > 286 || n_dist > 30) return E_LENS;
                // the code-length code: 7-bit direct table in the distance table's space (rebuilt right after)
                for (int i = lane; i < 19; i += GI_LANES) w.lens[300 + i] = 0;
                GI_WAVE_SYNC();
                for (int i = 0; i < n_cl; ++i) { refill(b); const uint8_t v = (uint8_t)take(b, 3); if (lane == 0) w.lens[300 + CLORD[i]] = v; }
                GI_WAVE_SYNC();
                if (build(w, w.lens + 300, 19, 7, w.dist_tab, w.dist_sorted, w.dist_count)) return E_LENS;
                uint8_t prev = 0;
                int i = 0;
                uint8_t* L = w.lens;                                       // (n_lit + n_dist <= 316: the 19 entries at 300.. are dead by then — the table is built)
                while (i < n_lit + n_dist) {
                    refill(b);
                    const uint32_t e = GI_UNIFORM(w.dist_tab[peek(b, 7)]);
                    if (e == NO_CODE || e == LONG_CODE) return E_LENS;
                    drop(b, (int)(e & 15u));
                    const int sym = (int)(e >> 4);
                    if (sym < 16) { if (lane == 0) L[i] = (uint8_t)sym; prev = (uint8_t)sym; ++i; continue; }
                    int rep; uint8_t v;
                    if (sym == 16) { if (i == 0) return E_LENS; rep = 3 + (int)take(b, 2); v = prev; }
                    else if (sym == 17) { rep = 3 + (int)take(b, 3); v = 0; }
                    else { rep = 11 + (int)take(b, 7); v = 0; }
                    if (i + rep > n_lit + n_dist) return E_LENS;
                    for (int k = lane; k < rep; k += GI_LANES) L[i + k] = v;
                    i += rep; prev = v;
                }
                GI_WAVE_SYNC();
                if (w.lens[256] == 0) return E_LENS;                       // no end-of-block code
                if (build(w, w.lens, n_lit, LIT_TB, w.lit_tab, w.lit_sorted, w.lit_count)) return E_LENS;
                if (build(w, w.lens + n_lit, n_dist, DIST_TB, w.dist_tab, w.dist_sorted, w.dist_count)) return E_LENS;
            }
            // the symbols: at most two refills per symbol (>= 33 bits after each: code <= 15 + extra <= 13 bits)
            for (;;) {
#ifdef GI_SCALAR_STATE
                // the reader's state is wave-uniform by construction; say so once per symbol, so that the bit arithmetic and
                // the branches on it are scalar instructions instead of 64 identical lanes under an execution mask
                b.buf = GI_UNIFORM64(b.buf); b.cnt = GI_UNIFORM(b.cnt); b.pos = GI_UNIFORM(b.pos); o.op = GI_UNIFORM(o.op);
#endif
                refill(b);
                const int sym = decode(b, w.lit_tab, LIT_TB, w.lit_sorted, w.lit_count);
                if (sym < 256) {
                    if (sym < 0) return E_CODE;
                    GI_STATS_LIT();
                    if (lane == 0) w.ring[o.op & RMASK] = (uint8_t)sym;
                    o.op++;
                    if ((o.op & 63u) == 0) { flush_lines(w, o, false); if (o.over) return E_OVERRUN; }
                    continue;
                }
                if (sym == 256) break;
                if (sym >= 286) return E_CODE;
                uint32_t lb, le, db, de;
                len_code((uint32_t)sym - 257u, lb, le);
                const uint32_t len = lb + take(b, (int)le);
                refill(b);
                const int ds = decode(b, w.dist_tab, DIST_TB, w.dist_sorted, w.dist_count);
                if (ds < 0 || ds >= 30) return E_CODE;
                dist_code((uint32_t)ds, db, de);
                const uint32_t dist = db + take(b, (int)de);
                GI_STATS_MATCH(len, dist);
                const int rc = copy_match(w, o, len, dist);
                if (rc) return rc;
            }
        } else return E_BTYPE;
        if (final) break;
    }
    flush_lines(w, o, true);
    if (o.over) return E_OVERRUN;
    if (o.op != isize) return E_SIZE;
    // every consumed bit must lie inside the block's input (the bit buffer may have read ahead)
    if (b.pos * 4u - b.cnt / 8u > in_len + lead) return E_INPUT;
    return OK;
}

}  // namespace gi
