/* huff_sync_probe — VERDICT r5 item 6 (iii), the probe it asks for BEFORE anything is built: the device inflate's floor is ONE
 * block's serial symbol loop (a lane decodes ~10 800 tokens one after the other); k lanes could start at k guessed bit offsets
 * inside a deflate block and keep only the chains that fall into step with the true token sequence — IF a decoder that starts at
 * a wrong bit falls into step quickly.  This measures exactly that on a BAM's own BGZF payloads (host only, no GPU):
 *
 *   for every deflate block: the true token starts (bit offsets of every literal/length code);
 *   for `trials` random bit offsets inside the block that are NOT token starts: decode from there with the block's own two
 *   Huffman codes until the decoder's next literal/length code begins at a true token start (from then on it IS the true
 *   sequence), or until it meets an invalid code / the end-of-block symbol / the end of the block (a failed chain).
 *
 *   gcc -O2 -o build/huff_sync_probe tools/huff_sync_probe.c && build/huff_sync_probe in.bam [max_bgzf_blocks] [trials_per_block]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const uint8_t *p; size_t nbits; size_t pos; } bitr;
static inline int getbit(bitr *b) { if (b->pos >= b->nbits) return -1; int v = (b->p[b->pos >> 3] >> (b->pos & 7)) & 1; b->pos++; return v; }
static inline long getbits(bitr *b, int n) { long v = 0; for (int i = 0; i < n; i++) { int t = getbit(b); if (t < 0) return -1; v |= (long)t << i; } return v; }

typedef struct { short count[16], symbol[288]; } huff;
/* puff.c's canonical construction; returns < 0 over-subscribed, > 0 incomplete, 0 complete */
static int construct(huff *h, const short *length, int n)
{
    short offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int s = 0; s < n; s++) h->count[length[s]]++;
    if (h->count[0] == n) return 0;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= h->count[l]; if (left < 0) return left; }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + h->count[l];
    for (int s = 0; s < n; s++) if (length[s]) h->symbol[offs[length[s]]++] = (short)s;
    return left;
}
static int decode(bitr *b, const huff *h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        int t = getbit(b); if (t < 0) return -2;
        code |= t;
        int count = h->count[l];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -1;                                                    /* no such code */
}

static const short LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const short DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* one token from b; 0 literal, 1 match, 2 end of block, < 0 broken */
static int token(bitr *b, const huff *lc, const huff *dc)
{
    int s = decode(b, lc);
    if (s < 0) return -1;
    if (s < 256) return 0;
    if (s == 256) return 2;
    s -= 257; if (s >= 29) return -1;
    if (getbits(b, LEXT[s]) < 0) return -1;
    int d = decode(b, dc);
    if (d < 0 || d >= 30) return -1;
    if (getbits(b, DEXT[d]) < 0) return -1;
    return 1;
}

static int cmp_sz(const void *a, const void *b) { size_t x = *(const size_t *)a, y = *(const size_t *)b; return (x > y) - (x < y); }
static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static size_t *g_sync_tok, *g_sync_bits, g_n, g_cap, g_fail_code, g_fail_eob, g_fail_end, g_blocks, g_tokens_total, g_bits_total;
static size_t *g_blk_tokens, g_nblk, g_blkcap;
static void record(size_t t, size_t bits) { if (g_n == g_cap) { g_cap = g_cap ? g_cap * 2 : 4096; g_sync_tok = realloc(g_sync_tok, g_cap * sizeof(size_t)); g_sync_bits = realloc(g_sync_bits, g_cap * sizeof(size_t)); } g_sync_tok[g_n] = t; g_sync_bits[g_n++] = bits; }

/* one deflate stream (a BGZF payload): every block of it */
static void probe_stream(const uint8_t *p, size_t len, int trials)
{
    bitr b = {p, len * 8, 0};
    static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (;;) {
        long last = getbits(&b, 1), type = getbits(&b, 2);
        if (last < 0 || type < 0) return;
        if (type == 0) {                                           /* stored */
            b.pos = (b.pos + 7) & ~(size_t)7;
            long n = getbits(&b, 16); if (n < 0 || getbits(&b, 16) < 0) return;
            b.pos += (size_t)n * 8;
        } else if (type == 1 || type == 2) {
            huff lc, dc; short lengths[320];
            if (type == 1) {
                int s = 0; for (; s < 144; s++) lengths[s] = 8; for (; s < 256; s++) lengths[s] = 9; for (; s < 280; s++) lengths[s] = 7; for (; s < 288; s++) lengths[s] = 8;
                construct(&lc, lengths, 288);
                for (s = 0; s < 30; s++) lengths[s] = 5;
                construct(&dc, lengths, 30);
            } else {
                long nlen = getbits(&b, 5) + 257, ndist = getbits(&b, 5) + 1, ncode = getbits(&b, 4) + 4;
                if (nlen > 286 || ndist > 30) return;
                huff cl; int i;
                for (i = 0; i < ncode; i++) lengths[order[i]] = (short)getbits(&b, 3);
                for (; i < 19; i++) lengths[order[i]] = 0;
                if (construct(&cl, lengths, 19) != 0) return;
                i = 0;
                while (i < nlen + ndist) {
                    int s = decode(&b, &cl); if (s < 0) return;
                    if (s < 16) lengths[i++] = (short)s;
                    else {
                        int l = 0, rep;
                        if (s == 16) { if (i == 0) return; l = lengths[i - 1]; rep = 3 + (int)getbits(&b, 2); }
                        else if (s == 17) rep = 3 + (int)getbits(&b, 3);
                        else rep = 11 + (int)getbits(&b, 7);
                        if (i + rep > nlen + ndist) return;
                        while (rep--) lengths[i++] = (short)l;
                    }
                }
                if (construct(&lc, lengths, (int)nlen) < 0) return;
                if (construct(&dc, lengths + nlen, (int)ndist) < 0) return;
            }
            /* the true token starts of this block */
            size_t cap = 1 << 14, nt = 0; size_t *start = malloc(cap * sizeof *start);
            const size_t sym0 = b.pos;
            for (;;) {
                if (nt == cap) { cap *= 2; start = realloc(start, cap * sizeof *start); }
                start[nt++] = b.pos;
                int t = token(&b, &lc, &dc);
                if (t < 0) { free(start); return; }
                if (t == 2) break;
            }
            const size_t sym1 = b.pos;                             /* behind the end-of-block code */
            g_blocks++; g_tokens_total += nt; g_bits_total += sym1 - sym0;
            if (g_nblk == g_blkcap) { g_blkcap = g_blkcap ? g_blkcap * 2 : 1024; g_blk_tokens = realloc(g_blk_tokens, g_blkcap * sizeof(size_t)); }
            g_blk_tokens[g_nblk++] = nt;
            if (nt > 64) for (int tr = 0; tr < trials; tr++) {
                size_t s = sym0 + (size_t)(rnd() % (sym1 - sym0 - 32));
                if (bsearch(&s, start, nt, sizeof *start, cmp_sz)) s++;          /* a true start: one bit further (a code has >= 1 bit... a token >= 2 bits here) */
                if (bsearch(&s, start, nt, sizeof *start, cmp_sz)) continue;
                bitr w = {p, sym1, s};                              /* (the chain may not read past the block's last bit) */
                size_t n = 0; int how = 0;
                for (;;) {
                    int t = token(&w, &lc, &dc);
                    n++;
                    if (t < 0) { how = w.pos >= sym1 ? 3 : 1; break; }
                    if (t == 2) { how = 2; break; }
                    if (bsearch(&w.pos, start, nt, sizeof *start, cmp_sz)) break;
                }
                if (how == 0) record(n, w.pos - s);
                else if (how == 1) g_fail_code++; else if (how == 2) g_fail_eob++; else g_fail_end++;
            }
            free(start);
        } else return;
        if (last) return;
    }
}

static size_t pct(size_t *a, size_t n, double q) { return n ? a[(size_t)(q * (double)(n - 1))] : 0; }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s in.bam [max_bgzf_blocks=2000] [trials_per_deflate_block=64]\n", argv[0]); return 2; }
    const long max_blocks = argc > 2 ? atol(argv[2]) : 2000; const int trials = argc > 3 ? atoi(argv[3]) : 64;
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 1; }
    uint8_t hdr[18]; long nb = 0;
    uint8_t *buf = malloc(1 << 16);
    while (nb < max_blocks && fread(hdr, 1, 18, f) == 18) {
        if (hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[12] != 'B' || hdr[13] != 'C') { fprintf(stderr, "not a BGZF block at block %ld\n", nb); break; }
        const size_t bsize = (size_t)hdr[16] + ((size_t)hdr[17] << 8) + 1, payload = bsize - 18 - 8;
        if (fread(buf, 1, payload + 8, f) != payload + 8) break;
        if (nb > 0 || payload > 100) probe_stream(buf, payload, trials);        /* (the first block is mostly the header text) */
        nb++;
    }
    fclose(f);
    qsort(g_sync_tok, g_n, sizeof(size_t), cmp_sz); qsort(g_sync_bits, g_n, sizeof(size_t), cmp_sz); qsort(g_blk_tokens, g_nblk, sizeof(size_t), cmp_sz);
    const size_t fails = g_fail_code + g_fail_eob + g_fail_end, all = g_n + fails;
    printf("%ld BGZF blocks, %zu deflate blocks, %.0f tokens and %.0f bits of symbols per deflate block on average (tokens per block: median %zu, p90 %zu, max %zu)\n",
           nb, g_blocks, (double)g_tokens_total / (double)(g_blocks ? g_blocks : 1), (double)g_bits_total / (double)(g_blocks ? g_blocks : 1),
           pct(g_blk_tokens, g_nblk, 0.5), pct(g_blk_tokens, g_nblk, 0.9), pct(g_blk_tokens, g_nblk, 1.0));
    printf("%zu chains started at a random bit that is not a token start:\n", all);
    printf("  fell into step: %zu (%.2f %%)   after tokens: median %zu, p90 %zu, p99 %zu, p99.9 %zu, max %zu   after bits: median %zu, p90 %zu, p99 %zu, max %zu\n",
           g_n, 100.0 * (double)g_n / (double)(all ? all : 1), pct(g_sync_tok, g_n, 0.5), pct(g_sync_tok, g_n, 0.9), pct(g_sync_tok, g_n, 0.99), pct(g_sync_tok, g_n, 0.999),
           pct(g_sync_tok, g_n, 1.0), pct(g_sync_bits, g_n, 0.5), pct(g_sync_bits, g_n, 0.9), pct(g_sync_bits, g_n, 0.99), pct(g_sync_bits, g_n, 1.0));
    printf("  failed: %zu (%.2f %%): met an invalid code %zu, met the end-of-block symbol %zu, ran to the end of the block out of step %zu\n",
           fails, 100.0 * (double)fails / (double)(all ? all : 1), g_fail_code, g_fail_eob, g_fail_end);
    return 0;
}
