/* ASan/UBSan harness for inflate_fast.c (CPU only): random data -> zlib deflate -> bit flips / truncation -> our decoder.
 * gcc -O1 -g -fsanitize=address,undefined tools/fuzz_inflate.c fastf_amd/csrc/inflate_fast.c -lz -o build/fuzz_inflate */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
int fastf_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
static uint64_t s = 88172645463325252ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 20000, ok = 0, bad = 0, good_match = 0;
    for (int it = 0; it < iters; it++) {
        size_t n = rnd() % 70000;
        uint8_t *data = malloc(n + 1);
        int mode = rnd() % 4;
        for (size_t i = 0; i < n; i++) data[i] = mode == 0 ? (uint8_t)rnd() : mode == 1 ? (uint8_t)("ACGT"[rnd() & 3]) : mode == 2 ? (uint8_t)(rnd() % 7) : (uint8_t)(i / 300);
        z_stream z; memset(&z, 0, sizeof z);
        int strat[] = {Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE};
        deflateInit2(&z, rnd() % 10, Z_DEFLATED, -15, 1 + rnd() % 9, strat[rnd() % 4]);
        size_t cap = deflateBound(&z, n) + 16;
        uint8_t *comp = malloc(cap);
        z.next_in = data; z.avail_in = (uInt)n; z.next_out = comp; z.avail_out = (uInt)cap;
        deflate(&z, Z_FINISH); size_t cl = z.total_out; deflateEnd(&z);
        /* exact-size heap buffers: any out-of-bounds access trips ASan */
        uint8_t *in = malloc(cl ? cl : 1); memcpy(in, comp, cl);
        uint8_t *out = malloc(n ? n : 1);
        if (fastf_inflate_raw(in, cl, out, n) == 0 && memcmp(out, data, n) == 0) good_match++;
        else { fprintf(stderr, "MISMATCH on valid stream it=%d n=%zu\n", it, n); return 1; }
        for (int k = 0; k < 4; k++) {
            size_t cl2 = cl; 
            if (cl && rnd() % 3 == 0) cl2 = rnd() % cl;                         /* truncate */
            uint8_t *in2 = malloc(cl2 ? cl2 : 1); memcpy(in2, comp, cl2);
            for (int f = rnd() % 4; f > 0 && cl2; f--) in2[rnd() % cl2] ^= (uint8_t)(1u << (rnd() & 7));
            size_t n2 = rnd() % 5 == 0 ? rnd() % 70000 : n;
            uint8_t *out2 = malloc(n2 ? n2 : 1);
            if (fastf_inflate_raw(in2, cl2, out2, n2) == 0) ok++; else bad++;
            free(in2); free(out2);
        }
        free(data); free(comp); free(in); free(out);
    }
    printf("valid streams decoded identically: %d; mutated: %d accepted, %d rejected; no sanitizer report\n", good_match, ok, bad);
    return 0;
}
