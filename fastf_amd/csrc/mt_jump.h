/* mt_jump.h — jump-ahead for MT19937: polynomials x^J mod phi and their application (mt_jump.c) */
#ifndef FASTF_MT_JUMP_H
#define FASTF_MT_JUMP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define FASTF_MT_POLY_WORDS 312            /* 19937 coefficients, bit k of word k >> 6 = coefficient of x^k */
#define FASTF_MT_SUB_BLOCKS 256            /* a sub-stream of the parallel generator: 256 blocks of 624 draws */
#define FASTF_MT_SUB_DRAWS (624u * FASTF_MT_SUB_BLOCKS)
#define FASTF_MT_JUMP_R 32                 /* two-level seating: sub-streams 0, R, 2R, .. from the stream's state, then the R - 1 behind each */
#define FASTF_MT_JUMP_POLYS (2 * (FASTF_MT_JUMP_R - 1))   /* x^(iJ), i = 1..R-1, then x^(iRJ), i = 1..R-1: up to R^2 = 1024 sub-streams (163 M draws) per round */
/* out[l * 312 ..]: x^(stride_words * 2^l) mod phi for l = 0 .. n_levels-1 (computed: Berlekamp-Massey once per process, then
 * squarings, a few ms each).  0 on success. */
int fastf_mt_jump_polys(uint64_t stride_words, uint32_t n_levels, uint64_t *out);
/* a[0..623]: the generator's array at a block boundary (all 624 words handed out) -> the array `stride` words later, for the
 * polynomial x^stride mod phi (the host's reference of mt_jump_kernel) */
void fastf_mt_jump_apply(const uint32_t a[624], const uint64_t poly[FASTF_MT_POLY_WORDS], uint32_t out[624]);
/* out[(i-1) * 312 ..] = x^(i stride) mod phi, out[(R-1 + i-1) * 312 ..] = x^(i R stride) mod phi, i = 1 .. R-1 (R = FASTF_MT_JUMP_R) */
int fastf_mt_jump_table_compute(uint64_t stride_words, uint64_t *out);
/* the table the library carries (tools/mt_jump_gen.c at build time): fastf_mt_jump_table_compute(FASTF_MT_SUB_DRAWS) */
const uint64_t *fastf_mt_jump_table(void);
#ifdef __cplusplus
}
#endif
#endif
