/* mt_jump.h — jump-ahead for MT19937: polynomials x^J mod phi and their application (mt_jump.c) */
#ifndef FASTF_MT_JUMP_H
#define FASTF_MT_JUMP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define FASTF_MT_POLY_WORDS 312            /* 19937 coefficients, bit k of word k >> 6 = coefficient of x^k */
#define FASTF_MT_SUB_BLOCKS 512            /* a sub-stream of the parallel generator: 512 blocks of 624 draws */
#define FASTF_MT_SUB_DRAWS (624u * FASTF_MT_SUB_BLOCKS)
#define FASTF_MT_JUMP_LEVELS 12            /* polynomials for strides of 1, 2, 4 .. 2048 sub-streams: up to 1.3 G draws per call */
/* out[l * 312 ..]: x^(stride_words * 2^l) mod phi for l = 0 .. n_levels-1 (computed: Berlekamp-Massey once per process, then
 * squarings, a few ms each).  0 on success. */
int fastf_mt_jump_polys(uint64_t stride_words, uint32_t n_levels, uint64_t *out);
/* a[0..623]: the generator's array at a block boundary (all 624 words handed out) -> the array `stride` words later, for the
 * polynomial x^stride mod phi (the host's reference of mt_jump_kernel) */
void fastf_mt_jump_apply(const uint32_t a[624], const uint64_t poly[FASTF_MT_POLY_WORDS], uint32_t out[624]);
/* the table the library carries (tools/mt_jump_gen.c at build time): FASTF_MT_JUMP_LEVELS polynomials for FASTF_MT_SUB_DRAWS */
const uint64_t *fastf_mt_jump_table(void);
#ifdef __cplusplus
}
#endif
#endif
