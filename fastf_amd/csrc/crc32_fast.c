/*
 * crc32_fast.c — CRC-32 (IEEE 802.3, the gzip/BGZF checksum) by carry-less multiplication.
 *
 * Every BGZF block carries the CRC of its inflated bytes, and zlib 1.2.x computes it at about 1 GB/s per core — next to an
 * inflate that runs at 0.6 GB/s that is a third of the decode time of a BAM.  This is the folding scheme of Gopal et al.,
 * "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ" (Intel, 2009) for the reflected polynomial 0xEDB88320:
 * four 128-bit lanes folded by 512 bits per step, reduced to 128, then to 64 bits, then Barrett reduction to 32 bits.
 * Used for the 16-byte-aligned bulk of a buffer when the CPU has PCLMULQDQ; the tail (and everything on other CPUs) goes
 * through zlib's crc32().  fastf_crc32_selftest() compares both routes once per process and disables this one on any
 * difference.
 */
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>

__attribute__((target("pclmul,sse4.1")))
static uint32_t crc32_clmul(const unsigned char *buf, size_t len, uint32_t crc)     /* len >= 64, len % 16 == 0; raw state in/out */
{
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;

    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;

    while (len >= 64) {                                  /* fold four lanes by 512 bits */
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    /* four lanes → one */
    x0 = _mm_load_si128((const __m128i *)k3k4);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                  /* remaining 16-byte blocks */
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    /* 128 → 64 bits */
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    /* Barrett reduction to 32 bits */
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

static int g_clmul = -1;            /* -1 unknown, 0 off, 1 on */

static uint32_t crc32_route(const unsigned char *buf, size_t len, int use_clmul)
{
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    if (use_clmul && len >= 64) {
        const size_t bulk = len & ~(size_t)15;
        crc = ~crc32_clmul(buf, bulk, ~crc);
        buf += bulk; len -= bulk;
    }
    while (len) {                                        /* zlib takes uInt lengths */
        const uInt n = len > (1u << 30) ? (1u << 30) : (uInt)len;
        crc = (uint32_t)crc32(crc, buf, n);
        buf += n; len -= n;
    }
    return crc;
}

int fastf_crc32_selftest(void)
{
    { const int c = __atomic_load_n(&g_clmul, __ATOMIC_RELAXED); if (c >= 0) return c; }
    int ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (ok) {
        unsigned char t[1024 + 37];
        uint32_t s = 0x9E3779B9u;
        for (size_t i = 0; i < sizeof t; i++) { s = s * 1664525u + 1013904223u; t[i] = (unsigned char)(s >> 24); }
        static const size_t lens[] = {64, 65, 79, 80, 128, 255, 256, 1000, 1024 + 37};
        for (size_t k = 0; k < sizeof lens / sizeof lens[0] && ok; k++)
            for (size_t off = 0; off < 3 && ok; off++)
                if (crc32_route(t + off, lens[k] - off, 1) != crc32_route(t + off, lens[k] - off, 0)) ok = 0;
    }
    __atomic_store_n(&g_clmul, ok, __ATOMIC_RELAXED);
    return ok;
}

uint32_t fastf_crc32(const unsigned char *buf, size_t len)
{
    const int c = __atomic_load_n(&g_clmul, __ATOMIC_RELAXED);      /* (several worker threads may run the self-test at once: same answer) */
    return crc32_route(buf, len, c < 0 ? fastf_crc32_selftest() : c);
}

#else   /* other CPUs: zlib */

int fastf_crc32_selftest(void) { return 0; }
uint32_t fastf_crc32(const unsigned char *buf, size_t len)
{
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
    while (len) { const uInt n = len > (1u << 30) ? (1u << 30) : (uInt)len; crc = (uint32_t)crc32(crc, buf, n); buf += n; len -= n; }
    return crc;
}
#endif
