// gpu_inflate2.hpp — raw DEFLATE (RFC 1951) of one BGZF block by ONE LANE, into literals-in-place + match tokens.
//
// The step in front of the bam2db hot path is sam_read1() through htslib/zlib (bam2db_ds.c:360): inflate of independent
// <= 64 KiB BGZF blocks.  Round 2-4 ran one wavefront per block (gpu_inflate.hpp): all 64 lanes walk the same symbol loop on
// the same bit buffer, so a block costs a whole wave ~82 instructions per token, and the probe of round 5
// (profiles/r5_notes/inflate_probe_decode_vs_copy.txt) says that loop — not the copying — is what a block's 14 ms are made of.
// Huffman decoding is serial per block, so the way to make it cheap is to decode MANY blocks per wave: here a block is one
// lane's work.  What a lane cannot do well is copy — so it does not:
//   phase 1 (this file, bgzf_decode_kernel)   the lane decodes its block's tokens; a literal goes straight to its final place in
//       the output (the lane knows the position), a match becomes a 4-byte TOKEN (literals since the last match, length,
//       distance) in the block's token list.
//       NO DECODE TABLES.  What limits this kernel is how many blocks a CU holds at once: a block's tokens come one after the
//       other, a token-step is a few hundred dependent instructions whatever one does, so the rate is blocks in flight over
//       that latency, and the blocks' Huffman state lives in LDS.  Direct tables (10-bit + 8-bit, 3.7 KB a block: the first
//       build) allowed 32 blocks per CU.  Here a code is decoded from its CANONICAL form: per code length l one 32-bit word
//       (end of the codes of length <= l, left-aligned to 15 bits) << 16 | (symbols of length <= l) — fifteen words per
//       alphabet, in REGISTERS.  Fifteen compares give the length, the last word that was passed gives where the length's
//       codes begin and how many symbols lie in front, and ONE LDS read fetches the symbol from the list sorted by (length,
//       symbol).  580 bytes of LDS per block (that list, packed, + the code lengths while a header is read): 64 blocks per
//       wave, four waves = 256 blocks per CU.
//       MEMORY TRAFFIC GOES IN EPOCHS.  A wave runs its lanes in lock step, so ONE lane waiting for memory stalls every block
//       of the wave, and on gfx9 loads and stores share one in-order counter (vmcnt): a lane that waits for its next input
//       words also waits for every literal any lane has just stored.  The first build of this kernel did exactly that and
//       took 2.4 us per token (profiles/r5_notes/inflate_two_kernels.txt).  So inside the symbol loop nothing touches memory:
//       literals and tokens are staged in LDS, the input comes out of a 3 x 16-byte register queue, and every EPOCH tokens
//       (the same count in every lane: they enter the loop together) the wave does all of it at once — takes over the quad
//       it asked for an epoch ago (long since there), stores what it staged, asks for the next quad.  A lane that runs dry
//       in between (header parsing, a pathological stream) loads on the spot: correct, just slow.
//   phase 2 (gpu_frontend.hpp, bgzf_resolve_kernel)   one wave per block takes 64 tokens at a time: a wave scan over
//       (literals + length) places every match, and the matches whose source bytes are complete copy at once, the others in the
//       next round — no Huffman state anywhere near it.
// The function below is plain single-threaded C++: the same source is compiled for the host and fuzzed against zlib there
// (tools/gi2_host.cpp, tests/test_gpu_inflate_host.py), on the device every lane of a wave runs it on its own block.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define GI2_FN __host__ __device__ __forceinline__
#else
#define GI2_FN static inline
#endif
#if defined(__clang__)
#define GI2_UNROLL _Pragma("unroll")
#else
#define GI2_UNROLL
#endif

// timing probe (experiment builds): cycles of a wave per section of the symbol loop, summed into gi2_stamp_acc[] by one lane
#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_GI2_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define GI2_STAMP(k) do { const uint64_t t__ = __builtin_amdgcn_s_memtime(); st_acc[k] += t__ - st_last; st_last = t__; } while (0)
#define GI2_STAMP_DECL uint64_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_last = __builtin_amdgcn_s_memtime()
#define GI2_STAMP_OUT(dst) do { for (int k__ = 0; k__ < 8; ++k__) (dst)[k__] = st_acc[k__]; } while (0)
#else
#define GI2_STAMP(k) do { } while (0)
#define GI2_STAMP_DECL do { } while (0)
#define GI2_STAMP_OUT(dst) do { } while (0)
#endif

namespace gi2 {

enum { OK = 0, E_BTYPE = 1, E_STORED = 2, E_LENS = 3, E_CODE = 4, E_DIST = 5, E_OVERRUN = 6, E_INPUT = 7, E_SIZE = 8, E_TOKENS = 9 };

// A token (u32):  bits 0..14 distance - 1 | bit 15 = 0 | bits 16..23 length - 3 | bits 24..31 literals in front of the match
//                 bit 15 = 1: no match, bits 16..31 = literals to step over (runs of more than 255 literals, stored blocks)
constexpr uint32_t TOK_SKIP = 1u << 15;
// tokens a block of `isize` bytes can need: every match covers at least 3 bytes, a step-over token at least 256
GI2_FN uint32_t token_cap(uint32_t isize) { return isize / 3u + isize / 256u + 4u; }

// per-block working set (device: one per lane in LDS; the size is an ODD number of 32-bit words, so the lanes' copies of a
// member lie in different banks)
struct Work {
    uint8_t lit_lo[288];                // literal/length symbols by (code length, symbol) — the canonical order —, their low 8 bits
    uint32_t lit_hi[9];                 // ... and the ninth bit (length symbols, end of block), one bit per place
    uint8_t dist_sorted[32];            // distance symbols likewise; the code-length code's 19 symbols while a header is read
    uint16_t count[16], offs[16];       // construction scratch: symbols per length, next free place per length
    union {
        uint8_t lens[160];              // code lengths of the block's two alphabets, two per byte (header and construction only)
        struct { uint32_t lit[8], tok[16]; } stg;   // symbol loop: the running epoch's literals ((position << 8) | byte) and tokens
    };
};
static_assert(sizeof(Work) % 8 == 4 && sizeof(Work) == 580, "an odd number of words per block");
GI2_FN uint32_t get_len(const Work& w, uint32_t i) { return (w.lens[i >> 1] >> ((i & 1u) * 4u)) & 15u; }
GI2_FN void set_len(Work& w, uint32_t i, uint32_t v) {
    const uint32_t sh = (i & 1u) * 4u;
    w.lens[i >> 1] = (uint8_t)((w.lens[i >> 1] & ~(15u << sh)) | (v << sh));
}
constexpr uint32_t EPOCH = 8, STG_LIT = 8, STG_TOK = 16;     // tokens per epoch (at most one literal or two tokens each)

// Bit reader: the compressed bytes as 32-bit words from the 16-byte-aligned address at or below the block's first byte, QW
// words at a time (two 16-byte loads).  Three stages of registers: q is being consumed, n stands ready behind it, t is in
// flight (asked for at an epoch boundary, taken over at the next one).  After a boundary a lane holds at least QW words = 256
// bits behind the ones it is consuming; EPOCH tokens take 8 x 13 bits on BAM payloads and 8 x 48 at the very most, so running
// dry inside an epoch (-> a load on the spot, which stalls the wave) is left to pathological streams.
constexpr uint32_t QW = 8;
struct Bits {
    const uint32_t* in32; uint32_t pos, n_words; uint64_t buf; uint32_t cnt;
    uint32_t q[QW], n[QW], t[QW];
    uint32_t next_at;                   // word index of the next QW words to ask for
    bool n_valid, t_valid;
};
GI2_FN void stage_load(const Bits& b, uint32_t at, uint32_t (&a)[QW]) {
    // (words past the block are never consumed by a well-formed stream; the caller's buffer is readable 192 bytes past its
    //  last block, and a stream that runs on is stopped by the position check of the symbol loop)
    const uint32_t* p = b.in32 + at;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; a[4] = x.x; a[5] = x.y; a[6] = x.z; a[7] = x.w;
#else
    for (uint32_t i = 0; i < QW; ++i) a[i] = p[i];
#endif
}
GI2_FN uint32_t fetch_word(Bits& b) {
    const uint32_t w = b.q[0];
GI2_UNROLL
    for (uint32_t i = 0; i + 1 < QW; ++i) b.q[i] = b.q[i + 1];
    if ((++b.pos & (QW - 1u)) == 0) {                      // q is used up: the stage behind it
        if (b.n_valid) {
GI2_UNROLL
            for (uint32_t i = 0; i < QW; ++i) b.q[i] = b.n[i];
            b.n_valid = false;
        } else if (b.t_valid) {                            // (waits for it: off the fast path)
GI2_UNROLL
            for (uint32_t i = 0; i < QW; ++i) b.q[i] = b.t[i];
            b.t_valid = false;
        } else {                                           // (run dry: load on the spot)
            stage_load(b, b.next_at, b.q); b.next_at += QW;
#if defined(__HIP_DEVICE_COMPILE__)
            // the wait for this load belongs INSIDE the branch: left to the first use of q behind it, every refill of the fast
            // path would carry an s_waitcnt vmcnt(0) "in case the branch was taken" — and wait for the epoch's prefetch
            asm volatile("" :: "v"(b.q[0]), "v"(b.q[1]), "v"(b.q[2]), "v"(b.q[3]), "v"(b.q[4]), "v"(b.q[5]), "v"(b.q[6]), "v"(b.q[7]));
#endif
        }
    }
    return w;
}
// epoch boundary, input side.  In this order: what was asked for an epoch ago moves up (the one wait of the epoch, for a load
// that has had EPOCH tokens to arrive) — the caller's stores go here — and then the next words are asked for, as the LAST memory
// operation of the epoch: nothing is issued behind it until the next boundary, so the wait there is for it alone.
GI2_FN void epoch_take(Bits& b) {
    if (!b.n_valid && b.t_valid) {
GI2_UNROLL
        for (uint32_t i = 0; i < QW; ++i) b.n[i] = b.t[i];
        b.n_valid = true; b.t_valid = false;
    }
}
GI2_FN void epoch_ask(Bits& b) {
    if (!b.t_valid) { stage_load(b, b.next_at, b.t); b.next_at += QW; b.t_valid = true; }
}
// at least 33 valid bits afterwards
GI2_FN void refill(Bits& b) {
    if (b.cnt <= 32) { b.buf |= (uint64_t)fetch_word(b) << b.cnt; b.cnt += 32; }
}
GI2_FN uint32_t peek(const Bits& b, int n) { return (uint32_t)b.buf & ((1u << n) - 1u); }
GI2_FN void drop(Bits& b, uint32_t n) { b.buf >>= n; b.cnt -= n; }
GI2_FN uint32_t take(Bits& b, int n) { const uint32_t v = peek(b, n); drop(b, (uint32_t)n); return v; }
// `in` readable from its 16-byte-aligned floor to 192 bytes past in + in_len.  (What is read at most: every loop that consumes
// input stops once pos has passed n_words = the block's words + 2, a header in front of the first check is 9 bytes, and the three
// stages reach 3 x QW words ahead of pos — 124 bytes past the end with QW = 8.)
GI2_FN uint32_t bits_open(Bits& b, const uint8_t* in, uint32_t in_len) {
    const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(in) & 15u);     // bytes in front of `in` inside its first quad
    b.in32 = reinterpret_cast<const uint32_t*>(in - lead);
    b.pos = 0; b.n_words = (lead + in_len + 3u) / 4u + 2u; b.buf = 0; b.cnt = 0;
    stage_load(b, 0, b.q);
    stage_load(b, QW, b.n);
    stage_load(b, 2 * QW, b.t);
    b.next_at = 3 * QW; b.n_valid = true; b.t_valid = true;
    for (uint32_t left = lead; left;) {                          // step over the bytes in front
        refill(b);
        const uint32_t k = left < 4u ? left : 4u;
        drop(b, 8u * k); left -= k;
    }
    return lead;
}

GI2_FN uint32_t rev32(uint32_t c) {                          // all 32 bits in reverse order
#if defined(__clang__)
    return __builtin_bitreverse32(c);
#else
    c = ((c >> 1) & 0x55555555u) | ((c & 0x55555555u) << 1);
    c = ((c >> 2) & 0x33333333u) | ((c & 0x33333333u) << 2);
    c = ((c >> 4) & 0x0F0F0F0Fu) | ((c & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(c);
#endif
}

// A canonical Huffman code as fifteen words, v[l - 1] for code length l:
//     (first code of length l + number of codes of length l) << (15 - l)    in bits 16..31   (<= 32768: the END of the codes of
//                                                                            length <= l on the 15-bit left-aligned code line)
//     l                                                                       in bits 12..15
//     number of symbols of length <= l (at most 288)                          in bits 0..11
// The ends never decrease with l, so a 15-bit left-aligned code c has length 1 + #{l : c >= end(l)}, the codes of its length
// begin at the last end it has passed, and the symbols in front of them are that word's low half.
struct Canon { uint32_t v[15]; };

// the code lengths w.lens[at .. at + n_sym) -> the sorted symbol list (LIT: lit_lo / lit_hi, else dist_sorted) and the fifteen
// words.  0, or E_LENS for an over-subscribed set, or an incomplete one — which zlib (inftrees.c) accepts in one form only: a
// literal/length or distance alphabet holding a single code of length 1 (never for the code-length alphabet: `complete_only`).
template <bool LIT>
GI2_FN int build(Work& w, uint32_t at0, int n_sym, Canon& c, bool complete_only = false) {
    for (int l = 0; l < 16; ++l) w.count[l] = 0;
    for (int s = 0; s < n_sym; ++s) w.count[get_len(w, at0 + (uint32_t)s)]++;
    const uint32_t n_codes = (uint32_t)n_sym - w.count[0];
    int left = 1;
    bool bad = false;
    uint32_t first = 0, cum = 0, prev = 0;
GI2_UNROLL
    for (int l = 1; l <= 15; ++l) {
        const uint32_t n = w.count[l];
        left = (left << 1) - (int)n; bad |= left < 0;
        first = (first + prev) << 1;                       // first code of length l
        w.offs[l] = (uint16_t)cum;
        cum += n;
        c.v[l - 1] = (((first + n) << (15 - l)) << 16) | ((uint32_t)l << 12) | cum;
        prev = n;
    }
    if (bad) return E_LENS;
    if (left > 0 && (complete_only || n_codes > 1 || (n_codes == 1 && w.count[1] != 1))) return E_LENS;
    if (LIT) for (int i = 0; i < 9; ++i) w.lit_hi[i] = 0;
    for (int s = 0; s < n_sym; ++s) {
        const uint32_t l = get_len(w, at0 + (uint32_t)s);
        if (l) {
            const uint32_t at = w.offs[l];
            if (LIT) { w.lit_lo[at] = (uint8_t)s; if (s >= 256) w.lit_hi[at >> 5] |= 1u << (at & 31u); }
            else w.dist_sorted[at] = (uint8_t)s;
            w.offs[l] = (uint16_t)(at + 1u);
        }
    }
    return OK;
}

// one symbol (>= 15 valid bits in the buffer); -1: no such code
template <bool LIT>
GI2_FN int decode(Bits& b, const Canon& c, const Work& w) {
    const uint32_t code = rev32((uint32_t)b.buf) >> 17;    // the next 15 stream bits, first bit on top
    const uint32_t key = (code << 16) | 0xFFFFu;           // (>= a word iff the code has reached that word's end)
    // the last of the fifteen ascending words the key has reached: it carries where its length's codes begin, the symbols in
    // front, and its own number (n: the code's length is n + 1).  Fifteen INDEPENDENT compare-and-selects and a tree of maxima:
    // a wave has its SIMD to itself, so what a token-step costs is the length of its dependent chains, not its instruction
    // count — a running count over fifteen compares (a chain of adds) and a four-level binary search (a chain of compares)
    // were both measured slower (profiles/r5_notes/inflate_two_kernels.txt).
    uint32_t cnd[15];
GI2_UNROLL
    for (int i = 0; i < 15; ++i) cnd[i] = key >= c.v[i] ? c.v[i] : 0u;
#define GI2_MAX3(a, b, d) ((a) > (b) ? ((a) > (d) ? (a) : (d)) : ((b) > (d) ? (b) : (d)))
    const uint32_t m0 = GI2_MAX3(cnd[0], cnd[1], cnd[2]), m1 = GI2_MAX3(cnd[3], cnd[4], cnd[5]), m2 = GI2_MAX3(cnd[6], cnd[7], cnd[8]),
                   m3 = GI2_MAX3(cnd[9], cnd[10], cnd[11]), m4 = GI2_MAX3(cnd[12], cnd[13], cnd[14]);
    const uint32_t u0 = GI2_MAX3(m0, m1, m2), u1 = m3 > m4 ? m3 : m4;
    const uint32_t sel = u0 > u1 ? u0 : u1;
#undef GI2_MAX3
    const uint32_t n = (sel >> 12) & 15u;
    const uint32_t l = n + 1u;
    if (l > 15u) return -1;                                // beyond the end of the last length: not a code of this set
    const uint32_t idx = ((code - (sel >> 16)) >> (15u - l)) + (sel & 0xFFFu);
    drop(b, l);
    if (LIT) return (int)(w.lit_lo[idx] | (((w.lit_hi[idx >> 5] >> (idx & 31u)) & 1u) << 8));
    return (int)w.dist_sorted[idx];
}

// base value and number of extra bits of a length symbol (257..285 -> ls = 0..28) and of a distance symbol (0..29), RFC 1951 3.2.5
GI2_FN void len_code(uint32_t ls, uint32_t& base, uint32_t& extra) {
    if (ls < 8u) { base = 3u + ls; extra = 0; }
    else if (ls == 28u) { base = 258u; extra = 0; }
    else { extra = (ls - 4u) >> 2; base = 3u + ((4u + (ls & 3u)) << extra); }
}
GI2_FN void dist_code(uint32_t ds, uint32_t& base, uint32_t& extra) {
    if (ds < 4u) { base = 1u + ds; extra = 0; }
    else { extra = (ds >> 1) - 1u; base = 1u + ((2u + (ds & 1u)) << extra); }
}

// n_tok counts the tokens handed out, of which the last s_tok are still staged (Work::stg); s_lit literals are staged
struct Out { uint8_t* out; uint32_t cap, op; uint32_t* tok; uint32_t n_tok, tok_cap, lit_run, s_lit, s_tok; };

// epoch boundary, output side: the staged literals to their places, the staged tokens behind the block's list
GI2_FN void flush_staged(Work& w, Out& o) {
    for (uint32_t i = 0; i < o.s_lit; ++i) { const uint32_t e = w.stg.lit[i]; o.out[e >> 8] = (uint8_t)e; }
    const uint32_t t0 = o.n_tok - o.s_tok;
    for (uint32_t i = 0; i < o.s_tok; ++i) o.tok[t0 + i] = w.stg.tok[i];
    o.s_lit = 0; o.s_tok = 0;
}
GI2_FN void put_literal(Work& w, Out& o, uint32_t byte) {  // (the caller has checked o.op < o.cap)
    if (o.s_lit == STG_LIT) flush_staged(w, o);            // (cannot happen inside the symbol loop: an epoch has EPOCH tokens)
    w.stg.lit[o.s_lit++] = (o.op << 8) | byte;
    o.op++; o.lit_run++;
}
GI2_FN int put_token(Work& w, Out& o, uint32_t t) {
    if (o.n_tok >= o.tok_cap) return E_TOKENS;
    if (o.s_tok == STG_TOK) flush_staged(w, o);
    w.stg.tok[o.s_tok++] = t; o.n_tok++;
    return OK;
}
GI2_FN int put_step_over(Work& w, Out& o) {                // the literals so far as a token of their own (two for a block of 65 536 literals)
    int rc = OK;
    while (o.lit_run && !rc) {
        const uint32_t n = o.lit_run < 65535u ? o.lit_run : 65535u;
        rc = put_token(w, o, TOK_SKIP | (n << 16));
        o.lit_run -= n;
    }
    return rc;
}
GI2_FN int put_match(Work& w, Out& o, uint32_t len, uint32_t dist) {
    if (dist > o.op) return E_DIST;                       // (dist >= 1 by construction)
    if (o.op + len > o.cap) return E_OVERRUN;
    if (o.lit_run > 255u) { const int rc = put_step_over(w, o); if (rc) return rc; }
    const int rc = put_token(w, o, (dist - 1u) | ((len - 3u) << 16) | (o.lit_run << 24));
    o.lit_run = 0; o.op += len;
    return rc;
}

// One BGZF block: out must hold `isize` bytes; literals are written to their places in out, the bytes of the matches are
// LEFT OUT — tok[0 .. *n_tok) says where they are and where they come from (resolve() below, bgzf_resolve_kernel on the device).
// tok must hold token_cap(isize) entries.
GI2_FN int inflate_tokens(Work& w, const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t isize, uint32_t* tok, uint32_t* n_tok,
                          uint64_t* stamps_out = nullptr) {
    GI2_STAMP_DECL;
    (void)stamps_out;
    const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    Bits b;
    uint32_t lead = bits_open(b, in, in_len);
    Out o{out, isize, 0, tok, 0, token_cap(isize), 0, 0, 0};
    int err = OK;
    bool final = false;
    while (!final && !err) {
        refill(b);
        final = take(b, 1) != 0;
        const uint32_t type = take(b, 2);
        if (type == 0) {                                                   // stored
            drop(b, b.cnt & 7u);
            refill(b);
            const uint32_t n = take(b, 16);
            refill(b);
            const uint32_t nn = take(b, 16);
            // byte position of the next unread input byte, relative to `in`
            const uint32_t p = b.pos * 4u - b.cnt / 8u - lead;
            if ((n ^ nn) != 0xFFFFu) err = E_STORED;
            else if (o.op + n > o.cap) err = E_OVERRUN;
            else if (p + n > in_len) err = E_INPUT;
            else {
                for (uint32_t i = 0; i < n; ++i) out[o.op + i] = in[p + i];
                o.op += n; o.lit_run += n;
                if (o.lit_run > 60000u) err = put_step_over(w, o);
                // restart the bit reader behind the stored bytes (positions are relative to the new start from here on)
                in += p + n; in_len -= p + n;
                lead = bits_open(b, in, in_len);
            }
        } else if (type == 1 || type == 2) {
            flush_staged(w, o);                                            // (the header's code lengths take the staging space)
            int n_lit = 288, n_dist = 32;
            Canon cl, cd;                                                  // the block's two codes (cl: first the code-length code)
            if (type == 1) {                                               // fixed code (RFC 1951 3.2.6)
                for (uint32_t i = 0; i < 288; ++i) set_len(w, i, i < 144 ? 8u : i < 256 ? 9u : i < 280 ? 7u : 8u);
                for (uint32_t i = 0; i < 32; ++i) set_len(w, 288u + i, 5u);    // (symbols 30, 31 never occur; building over 32 keeps the code complete)
            } else {                                                       // dynamic code: the header
                n_lit = (int)take(b, 5) + 257; n_dist = (int)take(b, 5) + 1;
                const int n_cl = (int)take(b, 4) + 4;
                if (n_lit > 286 || n_dist > 30) err = E_LENS;
                // the code-length code: its symbols sorted in the distance list's space (rebuilt right after)
                for (uint32_t i = 0; i < 19; ++i) set_len(w, 300u + i, 0);
                for (int i = 0; i < n_cl; ++i) { refill(b); set_len(w, 300u + CLORD[i], take(b, 3)); }
                if (b.pos > b.n_words) err = E_INPUT;                      // (a header that runs on past its block)
                if (!err && build<false>(w, 300, 19, cl, true)) err = E_LENS;
                uint32_t prev = 0;
                int i = 0;
                const int n_all = n_lit + n_dist;                          // (<= 316: the 19 entries at 300.. are dead by the time they are overwritten — the code is built)
                while (i < n_all && !err) {
                    if (b.pos > b.n_words) { err = E_INPUT; break; }       // every loop that consumes input checks: the over-read stays within the slack
                    refill(b);
                    const int cs = decode<false>(b, cl, w);
                    if (cs < 0) { err = E_LENS; break; }
                    const uint32_t sym = (uint32_t)cs;
                    if (sym < 16u) { set_len(w, (uint32_t)i++, sym); prev = sym; }
                    else {
                        uint32_t rep, v;
                        if (sym == 16u) { rep = 3u + take(b, 2); v = prev; if (i == 0) err = E_LENS; }
                        else if (sym == 17u) { rep = 3u + take(b, 3); v = 0; }
                        else { rep = 11u + take(b, 7); v = 0; }
                        if (i + (int)rep > n_all) err = E_LENS;
                        else { for (uint32_t k = 0; k < rep; ++k) set_len(w, (uint32_t)i + k, v); i += (int)rep; prev = v; }
                    }
                }
                if (!err && get_len(w, 256) == 0) err = E_LENS;            // no end-of-block code
            }
            if (!err && build<true>(w, 0, n_lit, cl)) err = E_LENS;
            if (!err && build<false>(w, (uint32_t)n_lit, n_dist, cd)) err = E_LENS;
            // (from here on w.lens is dead: its space holds the epochs' staged literals and tokens)
            // the symbols: at most two refills per token (>= 33 bits after each: code <= 15 + extra <= 13 bits)
            uint32_t step = 0;
            GI2_STAMP(0);                                                  // (0: headers and code construction)
            while (!err) {
                if (b.pos > b.n_words) { err = E_INPUT; break; }           // a stream that runs on past its block
                if ((++step & (EPOCH - 1u)) == 0) { epoch_take(b); flush_staged(w, o); epoch_ask(b); }   // all of the wave's memory traffic, every EPOCH tokens
                GI2_STAMP(1);                                              // (1: loop head + epochs)
                refill(b);
                GI2_STAMP(2);                                              // (2: refill)
                const int sym = decode<true>(b, cl, w);
                GI2_STAMP(3);                                              // (3: literal/length symbol)
                if (sym < 256) {
                    if (sym < 0) { err = E_CODE; break; }
                    if (o.op >= o.cap) { err = E_OVERRUN; break; }
                    put_literal(w, o, (uint32_t)sym);
                    GI2_STAMP(4);                                          // (4: literal staged)
                    continue;
                }
                if (sym == 256) break;
                if (sym >= 286) { err = E_CODE; break; }
                uint32_t lb, le, db, de;
                len_code((uint32_t)sym - 257u, lb, le);
                const uint32_t len = lb + take(b, (int)le);
                refill(b);
                const int ds = decode<false>(b, cd, w);
                if (ds < 0 || ds >= 30) { err = E_CODE; break; }
                dist_code((uint32_t)ds, db, de);
                const uint32_t dist = db + take(b, (int)de);
                GI2_STAMP(5);                                              // (5: length extra bits, distance symbol, its extra bits)
                err = put_match(w, o, len, dist);
                GI2_STAMP(6);                                              // (6: token staged)
            }
        } else err = E_BTYPE;
    }
    flush_staged(w, o);
    GI2_STAMP(7);
    GI2_STAMP_OUT(stamps_out);
    *n_tok = o.n_tok;
    if (err) return err;
    if (o.op != isize) return E_SIZE;
    // every consumed bit must lie inside the block's input (the bit buffer may have read ahead)
    if (b.pos * 4u - b.cnt / 8u > in_len + lead) return E_INPUT;
    return OK;
}

// the matches of a token list, one after the other (the host's reference of what bgzf_resolve_kernel does in parallel)
GI2_FN int resolve(uint8_t* out, uint32_t isize, const uint32_t* tok, uint32_t n_tok) {
    uint32_t op = 0;
    for (uint32_t t = 0; t < n_tok; ++t) {
        const uint32_t k = tok[t];
        if (k & TOK_SKIP) { op += k >> 16; continue; }
        const uint32_t lit = k >> 24, len = ((k >> 16) & 255u) + 3u, dist = (k & 0x7FFFu) + 1u;
        op += lit;
        if (dist > op || op + len > isize) return E_DIST;
        for (uint32_t i = 0; i < len; ++i) out[op + i] = out[op - dist + i];
        op += len;
    }
    return op <= isize ? OK : E_SIZE;
}

}  // namespace gi2
