/*
 * inflate_fast.c — raw DEFLATE (RFC 1951) decoder for whole BGZF blocks.
 *
 * BAM decoding is the host-side step in front of the device path (reference: sam_read1() through
 * htslib/zlib, bam2db_ds.c:360).  A BGZF block is at most 64 KiB, its inflated size is known up front
 * (ISIZE) and its CRC32 is checked by the caller, so the decoder can be a single straight loop over one
 * input and one output buffer: 64-bit bit buffer refilled 8 bytes at a time, 11-bit primary Huffman
 * tables with sub-tables, matches copied in 8-byte words.  It never writes outside [out, out+out_len)
 * and never reads outside [in, in+in_len); any malformed stream returns non-zero and the caller falls
 * back to zlib.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define LIT_BITS   11
#define DIST_BITS  8
#define MAX_LIT_ENTRIES  ((1 << LIT_BITS) + 1334)    /* primary + worst-case sub-tables (zlib ENOUGH bounds) */
#define MAX_DIST_ENTRIES ((1 << DIST_BITS) + 402)

/* table entry: [31:16] value  [15:8] extra-bit count (or sub-table index bits)  [7:4] kind  [3:0] code length */
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
#define ENTRY(val, extra, kind, len) (((uint32_t)(val) << 16) | ((uint32_t)(extra) << 8) | ((uint32_t)(kind) << 4) | (uint32_t)(len))

static const uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t rev_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

static uint32_t sym_entry(int is_dist, int sym, int len)
{
    if (is_dist) {
        if (sym >= 30) return ENTRY(0, 0, K_BAD, len);
        return ENTRY(k_dist_base[sym], k_dist_extra[sym], K_DIST, len);
    }
    if (sym < 256) return ENTRY(sym, 0, K_LIT, len);
    if (sym == 256) return ENTRY(0, 0, K_EOB, len);
    if (sym >= 286) return ENTRY(0, 0, K_BAD, len);
    return ENTRY(k_len_base[sym - 257], k_len_extra[sym - 257], K_LEN, len);
}

/* canonical Huffman → lookup table (bit-reversed codes, LSB-first stream).  Returns 0 ok, 1 invalid set. */
static int build_table(const uint8_t *lens, int n_sym, int is_dist, int tbits, uint32_t *tab, int max_entries)
{
    int count[16] = {0};
    for (int i = 0; i < n_sym; i++) count[lens[i]]++;
    if (count[0] == n_sym) {                        /* no codes at all: legal for distances in a literal-only block */
        for (int i = 0; i < (1 << tbits); i++) tab[i] = ENTRY(0, 0, K_BAD, 1);
        return 0;
    }
    /* over-subscribed / incomplete check (an incomplete set is allowed only for a single code) */
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return 1; }
    int n_codes = n_sym - count[0];
    if (left > 0 && !(n_codes == 1)) return 1;

    uint32_t next_code[16]; uint32_t code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }

    const int psize = 1 << tbits;
    for (int i = 0; i < psize; i++) tab[i] = ENTRY(0, 0, K_BAD, 1);
    int next_free = psize;
    /* sub-table bookkeeping: for each primary prefix that needs one, its slot and width */
    /* pass 1: short codes */
    for (int s = 0; s < n_sym; s++) {
        int len = lens[s];
        if (!len || len > tbits) continue;
        uint32_t c = rev_bits(next_code[len]++, len);
        uint32_t e = sym_entry(is_dist, s, len);
        for (uint32_t i = c; i < (uint32_t)psize; i += 1u << len) tab[i] = e;
    }
    /* pass 2: long codes, grouped by their low `tbits` bits (the reversed prefix) */
    /* first find, per prefix, the longest code to size the sub-table */
    uint8_t sub_bits[1 << LIT_BITS]; memset(sub_bits, 0, (size_t)psize);
    uint32_t nc2[16]; memcpy(nc2, next_code, sizeof nc2);
    for (int s = 0; s < n_sym; s++) {
        int len = lens[s];
        if (len <= tbits) continue;
        uint32_t c = rev_bits(nc2[len]++, len);
        uint32_t pre = c & (uint32_t)(psize - 1);
        if (len - tbits > sub_bits[pre]) sub_bits[pre] = (uint8_t)(len - tbits);
    }
    uint16_t sub_off[1 << LIT_BITS];
    for (int pfx = 0; pfx < psize; pfx++) {
        if (!sub_bits[pfx]) continue;
        int sz = 1 << sub_bits[pfx];
        if (next_free + sz > max_entries) return 1;
        sub_off[pfx] = (uint16_t)next_free;
        for (int i = 0; i < sz; i++) tab[next_free + i] = ENTRY(0, 0, K_BAD, 1);
        tab[pfx] = ENTRY(next_free, sub_bits[pfx], K_SUB, tbits);
        next_free += sz;
    }
    for (int s = 0; s < n_sym; s++) {
        int len = lens[s];
        if (len <= tbits) continue;
        uint32_t c = rev_bits(next_code[len]++, len);
        uint32_t pre = c & (uint32_t)(psize - 1);
        uint32_t hi = c >> tbits;
        int sb = sub_bits[pre], sl = len - tbits;
        uint32_t e = sym_entry(is_dist, s, sl);     /* length field = bits consumed INSIDE the sub-table */
        for (uint32_t i = hi; i < (1u << sb); i += 1u << sl) tab[sub_off[pre] + i] = e;
    }
    return 0;
}

typedef struct { const uint8_t *in, *in_end; uint64_t buf; int cnt; } bitrd;

static inline __attribute__((always_inline)) void refill(bitrd *b)
{
    if (b->in_end - b->in >= 8) {
        uint64_t w; memcpy(&w, b->in, 8);
        b->buf |= w << b->cnt;
        int take = (63 - b->cnt) >> 3;              /* whole bytes that fit */
        b->in += take; b->cnt += take * 8;
    } else {
        while (b->cnt <= 56 && b->in < b->in_end) { b->buf |= (uint64_t)*b->in++ << b->cnt; b->cnt += 8; }
    }
}
static inline __attribute__((always_inline)) uint32_t peek(const bitrd *b, int n) { return (uint32_t)(b->buf & ((1ull << n) - 1)); }
static inline __attribute__((always_inline)) void drop(bitrd *b, int n) { b->buf >>= n; b->cnt -= n; }

static inline __attribute__((always_inline)) int inflate_body(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    bitrd b = { in, in + in_len, 0, 0 };
    uint8_t *op = out, *const oend = out + out_len;
    uint32_t lit[MAX_LIT_ENTRIES], dst[MAX_DIST_ENTRIES];
    int have_fixed = 0;
    uint32_t flit[MAX_LIT_ENTRIES], fdst[MAX_DIST_ENTRIES];
    int final = 0;

    while (!final) {
        refill(&b);
        if (b.cnt < 3) return 1;
        final = (int)peek(&b, 1); drop(&b, 1);
        int type = (int)peek(&b, 2); drop(&b, 2);
        const uint32_t *L, *D;
        if (type == 0) {                                        /* stored */
            drop(&b, b.cnt & 7);
            /* give whole unread bytes back to the input pointer */
            b.in -= b.cnt >> 3; b.buf = 0; b.cnt = 0;
            if (b.in_end - b.in < 4) return 1;
            uint32_t len = (uint32_t)b.in[0] | (uint32_t)b.in[1] << 8, nlen = (uint32_t)b.in[2] | (uint32_t)b.in[3] << 8;
            b.in += 4;
            if ((len ^ nlen) != 0xffff) return 1;
            if ((size_t)(b.in_end - b.in) < len || (size_t)(oend - op) < len) return 1;
            memcpy(op, b.in, len); op += len; b.in += len;
            continue;
        } else if (type == 1) {                                 /* fixed codes */
            if (!have_fixed) {
                uint8_t l[288];
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                if (build_table(l, 288, 0, LIT_BITS, flit, MAX_LIT_ENTRIES)) return 1;
                for (int i = 0; i < 32; i++) l[i] = 5;           /* 30 and 31 never occur: K_BAD entries */
                if (build_table(l, 32, 1, DIST_BITS, fdst, MAX_DIST_ENTRIES)) return 1;
                have_fixed = 1;
            }
            L = flit; D = fdst;
        } else if (type == 2) {                                 /* dynamic codes */
            refill(&b);
            if (b.cnt < 14) return 1;
            int hlit = (int)peek(&b, 5) + 257; drop(&b, 5);
            int hdist = (int)peek(&b, 5) + 1; drop(&b, 5);
            int hclen = (int)peek(&b, 4) + 4; drop(&b, 4);
            if (hlit > 286 || hdist > 30) return 1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19]; memset(cl, 0, sizeof cl);
            for (int i = 0; i < hclen; i++) { refill(&b); if (b.cnt < 3) return 1; cl[order[i]] = (uint8_t)peek(&b, 3); drop(&b, 3); }
            uint32_t ct[(1 << 7) + 64];
            if (build_table(cl, 19, 0, 7, ct, (1 << 7) + 64)) return 1;     /* symbols 0..18 come back as K_LIT values */
            uint8_t lens[286 + 30]; int n = 0;
            while (n < hlit + hdist) {
                refill(&b);
                uint32_t e = ct[peek(&b, 7)];
                int kind = (e >> 4) & 15, len = e & 15;
                if (kind != K_LIT || len > b.cnt) return 1;
                drop(&b, len);
                int sym = (int)(e >> 16);
                if (sym < 16) lens[n++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (n == 0 || b.cnt < 2) return 1; val = lens[n - 1]; rep = 3 + (int)peek(&b, 2); drop(&b, 2); }
                    else if (sym == 17) { if (b.cnt < 3) return 1; rep = 3 + (int)peek(&b, 3); drop(&b, 3); }
                    else { if (b.cnt < 7) return 1; rep = 11 + (int)peek(&b, 7); drop(&b, 7); }
                    if (n + rep > hlit + hdist) return 1;
                    while (rep--) lens[n++] = (uint8_t)val;
                }
            }
            if (lens[256] == 0) return 1;                        /* no end-of-block code */
            if (build_table(lens, hlit, 0, LIT_BITS, lit, MAX_LIT_ENTRIES)) return 1;
            if (build_table(lens + hlit, hdist, 1, DIST_BITS, dst, MAX_DIST_ENTRIES)) return 1;
            L = lit; D = dst;
        } else return 1;

        /* ---- symbols of one block ---- */
        /* Fast loop: while at least 16 input bytes and 320 output bytes remain, a refill always delivers >= 56 bits and
         * no symbol can overrun the output (a match is at most 258 bytes, its word copy may run 7 bytes past it, and up to
         * three literals precede it), so nothing inside checks a bound.  56 bits cover a whole length/distance pair
         * (15 + 5 + 15 + 13 = 48), so one refill per iteration is enough. */
        int eob = 0;
        while (b.in_end - b.in >= 16 && oend - op >= 320) {
            refill(&b);
            uint32_t e = L[peek(&b, LIT_BITS)];
            int kind = (e >> 4) & 15;
            if (kind == K_LIT) {                                 /* up to three literals per refill (3 x 15 bits at most) */
                drop(&b, e & 15); *op++ = (uint8_t)(e >> 16);
                e = L[peek(&b, LIT_BITS)]; kind = (e >> 4) & 15;
                if (kind == K_LIT) {
                    drop(&b, e & 15); *op++ = (uint8_t)(e >> 16);
                    e = L[peek(&b, LIT_BITS)]; kind = (e >> 4) & 15;
                    if (kind == K_LIT) { drop(&b, e & 15); *op++ = (uint8_t)(e >> 16); continue; }
                }
                if (b.cnt < 48) refill(&b);                      /* a length/distance pair may follow */
            }
            if (kind == K_SUB) {
                drop(&b, LIT_BITS);
                e = L[(e >> 16) + peek(&b, (e >> 8) & 255)];
                kind = (e >> 4) & 15;
            }
            drop(&b, e & 15);
            if (kind == K_LIT) { *op++ = (uint8_t)(e >> 16); continue; }
            if (kind == K_EOB) { eob = 1; break; }
            if (kind != K_LEN) return 1;
            const int xb = (e >> 8) & 255;
            const uint32_t len = (e >> 16) + peek(&b, xb); drop(&b, xb);
            uint32_t d = D[peek(&b, DIST_BITS)];
            int dk = (d >> 4) & 15;
            if (dk == K_SUB) { drop(&b, DIST_BITS); d = D[(d >> 16) + peek(&b, (d >> 8) & 255)]; dk = (d >> 4) & 15; }
            if (dk != K_DIST) return 1;
            drop(&b, d & 15);
            const int dxb = (d >> 8) & 255;
            const uint32_t dist = (d >> 16) + peek(&b, dxb); drop(&b, dxb);
            if (b.cnt < 0 || dist > (size_t)(op - out)) return 1;
            const uint8_t *src = op - dist;
            if (dist >= 16) {                                    /* 16 bytes at a time, the first two without a loop test */
                uint8_t *q = op; const uint8_t *sp = src;
                uint64_t w0, w1;
                memcpy(&w0, sp, 8); memcpy(&w1, sp + 8, 8); memcpy(q, &w0, 8); memcpy(q + 8, &w1, 8);
                if (len > 16) {
                    memcpy(&w0, sp + 16, 8); memcpy(&w1, sp + 24, 8); memcpy(q + 16, &w0, 8); memcpy(q + 24, &w1, 8);
                    for (uint32_t done = 32; done < len; done += 16) {
                        memcpy(&w0, sp + done, 8); memcpy(&w1, sp + done + 8, 8); memcpy(q + done, &w0, 8); memcpy(q + done + 8, &w1, 8);
                    }
                }
            } else if (dist >= 8) {
                uint8_t *q = op; const uint8_t *sp = src; uint32_t left = len;
                do { uint64_t w; memcpy(&w, sp, 8); memcpy(q, &w, 8); q += 8; sp += 8; left = left > 8 ? left - 8 : 0; } while (left);
            } else if (dist == 1) {
                memset(op, *src, len);
            } else {
                for (uint32_t i = 0; i < len; i++) op[i] = src[i];
            }
            op += len;
        }
        /* careful loop: the ends of the buffers */
        while (!eob) {
            refill(&b);                                          /* >= 56 bits unless the input is nearly exhausted */
            uint32_t e = L[peek(&b, LIT_BITS)];
            int kind = (e >> 4) & 15;
            if (kind == K_SUB) {
                drop(&b, LIT_BITS);
                e = L[(e >> 16) + peek(&b, (e >> 8) & 255)];
                kind = (e >> 4) & 15;
            }
            int cl = e & 15;
            if (cl > b.cnt) return 1;
            drop(&b, cl);
            if (kind == K_LIT) {
                if (op >= oend) return 1;
                *op++ = (uint8_t)(e >> 16);
                continue;
            }
            if (kind == K_EOB) break;
            if (kind != K_LEN) return 1;
            int xb = (e >> 8) & 255;
            if (xb > b.cnt) return 1;
            uint32_t len = (e >> 16) + peek(&b, xb); drop(&b, xb);
            if (b.cnt < 32) refill(&b);
            uint32_t d = D[peek(&b, DIST_BITS)];
            int dk = (d >> 4) & 15;
            if (dk == K_SUB) { drop(&b, DIST_BITS); d = D[(d >> 16) + peek(&b, (d >> 8) & 255)]; dk = (d >> 4) & 15; }
            if (dk != K_DIST || (int)(d & 15) > b.cnt) return 1;
            drop(&b, d & 15);
            int dxb = (d >> 8) & 255;
            if (dxb > b.cnt) return 1;
            uint32_t dist = (d >> 16) + peek(&b, dxb); drop(&b, dxb);
            if (dist > (size_t)(op - out) || len > (size_t)(oend - op)) return 1;
            const uint8_t *src = op - dist;
            if (dist >= 8 && (size_t)(oend - op) >= len + 8) {     /* word copies may run up to 7 bytes past len */
                uint8_t *q = op; const uint8_t *s = src; uint32_t left = len;
                do { uint64_t w; memcpy(&w, s, 8); memcpy(q, &w, 8); q += 8; s += 8; left = left > 8 ? left - 8 : 0; } while (left);
            } else if (dist == 1) {
                memset(op, *src, len);
            } else {
                for (uint32_t i = 0; i < len; i++) op[i] = src[i];
            }
            op += len;
        }
    }
    return op == oend ? 0 : 1;
}

/* The same body compiled twice: with BMI2 the variable shifts and masks of the bit reader are single instructions
 * (shrx / bzhi); picked once per process by CPUID. */
#if defined(__x86_64__)
__attribute__((target("bmi2"))) static int inflate_bmi2(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    return inflate_body(in, in_len, out, out_len);
}
#endif
static int inflate_generic(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    return inflate_body(in, in_len, out, out_len);
}

int fastf_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
#if defined(__x86_64__)
    static int have_bmi2 = -1;                                  /* set once, by whichever worker thread comes first: atomics, not a plain int */
    int h = __atomic_load_n(&have_bmi2, __ATOMIC_RELAXED);
    if (h < 0) { h = __builtin_cpu_supports("bmi2") ? 1 : 0; __atomic_store_n(&have_bmi2, h, __ATOMIC_RELAXED); }
    if (h) return inflate_bmi2(in, in_len, out, out_len);
#endif
    return inflate_generic(in, in_len, out, out_len);
}
