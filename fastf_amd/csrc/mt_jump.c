/* mt_jump.c — jump-ahead for MT19937 (mt19937ar.c:105-140), host side.
 *
 * The depth draws of bam2db (bam2db_ds.c:385) are ONE MT19937 stream, one draw per CB hit in record order.  A stream is serial
 * by nature — mt_fill_kernel continues it at 2.5 G draws/s on one workgroup — but its state transition is linear over GF(2), so
 * the state J draws ahead is a fixed linear function of the state now: with F = "advance by one word" and phi = the
 * characteristic polynomial of F (degree 19937),  F^J = g(F)  for  g(x) = x^J mod phi(x),  a polynomial that depends on J alone
 * (no seed in it).  Seating S sub-streams J draws apart then costs log2(S) rounds of "apply g" and every sub-stream is generated
 * by a workgroup of its own (umi_kernels.hpp: mt_jump_kernel, mt_fill_multi_kernel).
 *
 * Representation.  At a block boundary the generator's array a[0..623] (all of it handed out) IS the state; X_0..X_623 = a and
 * X_{t+624} = X_{t+397} ^ mix(X_t, X_{t+1}) continues it as a flat word sequence (mt19937ar.c:113-123 regenerates 624 of these
 * at a time).  The window W_t = X[t .. t+623] is the state t words on, W_{t+1} = F(W_t), and
 *     W_J = g(F) W_0 = XOR over the set coefficients k of g of W_k,       i.e.  W_J[i] = XOR_k X[k + i]:
 * a word-wise convolution of the sequence with the polynomial's bits.  (The low 31 bits of X_t are not part of the 19937-bit
 * state and come out arbitrary in W_J[0]; nothing reads them: word 0 of a handed-out block only lends its top bit.)
 *
 * phi comes from Berlekamp-Massey on 2 x 19937 output bits; x^J mod phi from squarings.  tools/mt_jump_gen.c runs this at
 * build time and writes the table of polynomials the library carries (mt_jump_table.h); tests pin fastf_mt_jump_apply against
 * fastf_mt_skip. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fastf_amd.h"
#include "mt_jump.h"

enum { DEG = 19937, NW = FASTF_MT_POLY_WORDS, N2 = 2 * NW + 2 };

static inline uint32_t jm_mix(uint32_t hi, uint32_t lo)
{
    const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

static inline int bit_get(const uint64_t *a, size_t i) { return (int)((a[i >> 6] >> (i & 63)) & 1u); }
static inline void bit_flip(uint64_t *a, size_t i) { a[i >> 6] ^= 1ull << (i & 63); }

/* dst ^= src << sh (bits), src of n words, dst long enough */
static void xor_shifted(uint64_t *dst, const uint64_t *src, size_t n, size_t sh)
{
    const size_t w = sh >> 6, b = sh & 63;
    if (b == 0) { for (size_t i = 0; i < n; i++) dst[w + i] ^= src[i]; return; }
    uint64_t carry = 0;
    for (size_t i = 0; i < n; i++) { dst[w + i] ^= (src[i] << b) | carry; carry = src[i] >> (64 - b); }
    dst[w + n] ^= carry;
}

/* the characteristic polynomial of the one-word step, bit k = coefficient of x^k, bit 19937 set; 0 on success */
static int char_poly(uint64_t phi[NW + 1])
{
    /* 2 x 19937 bits of output (the low bit of every tempered draw of a fixed stream: a linear functional of the state) */
    enum { NB = 2 * DEG, SW = (NB + 63) / 64 + 1 };
    uint64_t *rv = (uint64_t *)calloc(SW + NW + 4, 8);              /* the sequence REVERSED: bit j = s_{NB-1-j} */
    enum { PW = 2 * NW + 8 };                                         /* room for B shifted by up to 19937 bits */
    uint64_t *C = (uint64_t *)calloc(PW, 8), *B = (uint64_t *)calloc(PW, 8), *T = (uint64_t *)calloc(PW, 8);
    if (!rv || !C || !B || !T) { free(rv); free(C); free(B); free(T); return 1; }
    fastf_mt_t mt; fastf_mt_seed(&mt, 5489u);
    for (size_t t = 0; t < NB; t++) if (fastf_mt_next(&mt) & 1u) bit_flip(rv, NB - 1 - t);
    /* Berlekamp-Massey over GF(2): C(x) = 1 + c_1 x + ... + c_L x^L with s_n = c_1 s_{n-1} ^ ... ^ c_L s_{n-L} */
    C[0] = 1; B[0] = 1;
    size_t L = 0, m = 1;
    for (size_t n = 0; n < NB; n++) {
        /* d = XOR_{i=0..L} C_i s_{n-i};  s_{n-i} = rv[(NB-1-n) + i]: the window of rv that starts at bit NB-1-n */
        const size_t off = NB - 1 - n, ow = off >> 6, ob = off & 63, lw = (L >> 6) + 1;
        uint64_t acc = 0;
        if (ob == 0) for (size_t i = 0; i < lw; i++) acc ^= C[i] & rv[ow + i];
        else for (size_t i = 0; i < lw; i++) acc ^= C[i] & ((rv[ow + i] >> ob) | (rv[ow + i + 1] << (64 - ob)));
        if (!__builtin_parityll(acc)) { m++; continue; }
        if (m > (size_t)DEG + 64) break;                              /* (cannot happen on this sequence) */
        if (2 * L <= n) {
            memcpy(T, C, (NW + 2) * 8);
            xor_shifted(C, B, NW + 1, m);
            L = n + 1 - L;
            memcpy(B, T, (NW + 2) * 8);
            m = 1;
        } else {
            xor_shifted(C, B, NW + 1, m);
            m++;
        }
    }
    int rc = L == DEG ? 0 : 1;
    /* phi(x) = x^L C(1/x) = x^L + c_1 x^{L-1} + ... + c_L */
    memset(phi, 0, (NW + 1) * 8);
    if (!rc) for (size_t i = 0; i <= L; i++) if (bit_get(C, i)) bit_flip(phi, L - i);
    free(rv); free(C); free(B); free(T);
    return rc;
}

/* r (2 NW + 2 words, degree < 2 * 19937) reduced mod phi in place: the result is in its low NW words */
static void reduce(uint64_t *r, const uint64_t phis[64][NW + 2])
{
    /* phis[b] = phi << b: the XOR for a high bit at position p = 19937 + q uses phi << q, a whole-word copy of phis[q & 63] */
    for (size_t p = 2 * DEG - 2 + 1; p-- > DEG;) {
        if (!bit_get(r, p)) continue;
        const size_t q = p - DEG, w = q >> 6;
        const uint64_t *f = phis[q & 63];
        for (size_t i = 0; i < NW + 2; i++) r[w + i] ^= f[i];
    }
}

static void square_mod(uint64_t p[NW], const uint64_t phis[64][NW + 2])
{
    static const uint16_t spread[256] = {
#define S1(x) (uint16_t)((((x) & 1) | (((x) & 2) << 1) | (((x) & 4) << 2) | (((x) & 8) << 3) | (((x) & 16) << 4) | (((x) & 32) << 5) | (((x) & 64) << 6) | (((x) & 128) << 7)))
#define S4(x) S1(x), S1((x) + 1), S1((x) + 2), S1((x) + 3)
#define S16(x) S4(x), S4((x) + 4), S4((x) + 8), S4((x) + 12)
#define S64(x) S16(x), S16((x) + 16), S16((x) + 32), S16((x) + 48)
        S64(0), S64(64), S64(128), S64(192)
    };
    uint64_t r[N2 + NW + 4];
    memset(r, 0, sizeof r);
    for (size_t i = 0; i < NW; i++) {
        uint64_t lo = 0, hi = 0;
        for (int b = 0; b < 4; b++) { lo |= (uint64_t)spread[(p[i] >> (8 * b)) & 255] << (16 * b); hi |= (uint64_t)spread[(p[i] >> (32 + 8 * b)) & 255] << (16 * b); }
        r[2 * i] = lo; r[2 * i + 1] = hi;
    }
    reduce(r, phis);
    memcpy(p, r, NW * 8);
    p[NW - 1] &= (1ull << (DEG - 64 * (NW - 1))) - 1ull;            /* (bits 19937.. are zero after the reduction anyway) */
}

int fastf_mt_jump_polys(uint64_t stride_words, uint32_t n_levels, uint64_t *out)
{
    if (stride_words == 0 || n_levels == 0 || !out) return 1;
    static uint64_t phi[NW + 1];
    static uint64_t (*phis)[NW + 2] = NULL;
    if (!phis) {
        if (char_poly(phi)) return 1;
        uint64_t (*ps)[NW + 2] = (uint64_t (*)[NW + 2])calloc(64, sizeof *ps);
        if (!ps) return 1;
        for (int b = 0; b < 64; b++) xor_shifted(ps[b], phi, NW + 1, (size_t)b);
        phis = ps;
    }
    /* x^stride: the odd part by multiplying x in (shift + reduce), the power of two by squarings */
    uint64_t odd = stride_words; uint32_t twos = 0;
    while (!(odd & 1)) { odd >>= 1; twos++; }
    uint64_t p[NW];
    memset(p, 0, sizeof p);
    if (odd < DEG) bit_flip(p, (size_t)odd);
    else {                                                            /* square-and-multiply on the bits of the odd part */
        p[0] = 2;                                                     /* x */
        int top = 63; while (!((odd >> top) & 1)) top--;
        for (int b = top - 1; b >= 0; b--) {
            square_mod(p, (const uint64_t (*)[NW + 2])phis);
            if ((odd >> b) & 1) {                                     /* times x */
                uint64_t r[N2 + NW + 4]; memset(r, 0, sizeof r);
                xor_shifted(r, p, NW, 1);
                reduce(r, (const uint64_t (*)[NW + 2])phis);
                memcpy(p, r, NW * 8);
            }
        }
    }
    for (uint32_t i = 0; i < twos; i++) square_mod(p, (const uint64_t (*)[NW + 2])phis);
    for (uint32_t l = 0; l < n_levels; l++) {
        memcpy(out + (size_t)l * NW, p, NW * 8);
        if (l + 1 < n_levels) square_mod(p, (const uint64_t (*)[NW + 2])phis);
    }
    return 0;
}

void fastf_mt_jump_apply(const uint32_t a[624], const uint64_t poly[FASTF_MT_POLY_WORDS], uint32_t out[624])
{
    enum { SEQ = 624 + DEG + 1 };
    uint32_t *x = (uint32_t *)malloc((size_t)SEQ * 4);
    if (!x) { memset(out, 0, 624 * 4); return; }
    memcpy(x, a, 624 * 4);
    for (size_t t = 0; t + 624 < SEQ; t++) x[t + 624] = x[t + 397] ^ jm_mix(x[t], x[t + 1]);
    uint32_t acc[624];
    memset(acc, 0, sizeof acc);
    for (size_t w = 0; w < NW; w++) {
        uint64_t bits = poly[w];
        while (bits) {
            const size_t k = (w << 6) + (size_t)__builtin_ctzll(bits);
            bits &= bits - 1;
            const uint32_t *s = x + k;
            for (int i = 0; i < 624; i++) acc[i] ^= s[i];
        }
    }
    memcpy(out, acc, sizeof acc);
    free(x);
}

#ifndef FASTF_MT_JUMP_NO_TABLE
#include "mt_jump_table.h"                 /* build/obj/mt_jump_table.h, written by tools/mt_jump_gen.c (fastf_amd/csrc/Makefile) */
const uint64_t *fastf_mt_jump_table(void) { return fastf_mt_jump_table_; }
#endif
