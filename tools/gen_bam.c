/* gen_bam.c — fast synthetic BAM generator for end-to-end benchmarks (test tooling, not product).
 *   gen_bam <out.bam> <barcodes.tsv> <features.tsv> <n_records> [seed] [umi_len] [seq_len] [threads] [repeat] [bodies]
 * repeat > 1 writes the n_records-record body (the 64 pieces: whole BGZF blocks, no record straddles a piece) that many
 * times behind the one header: a file of n_records x repeat records for the price of generating n_records.
 * bodies > 1 generates that many DIFFERENT bodies, one after the other, each from its own seed (and each written `repeat`
 * times): n_records x bodies distinct records — the matrix, the depth of its groups and the output files are those of a
 * file of that size, which a repeated body's are not.
 * The record stream is cut into 64 pieces, each with its own generator state and its own run of BGZF blocks, so the
 * file is the same for every thread count.
 * seq_len > 0 gives records the size and content mix of a Cell Ranger BAM: mapped reads with one CIGAR word, seq_len
 * packed bases, binned qualities (long runs of 'F' with ':' and ','), and the CR/CY/UR/UY/NH/AS/RG tags in front of
 * CB/xf/GX/UB; blocks are then deflated at level 6 as samtools does.
 * Every record: unmapped, tags CB:Z (95 % from the list, 5 % random), xf:C (85 % 25/17), GX:Z, UB:Z.
 * gcc -O2 -o gen_bam gen_bam.c -lz -lpthread */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <pthread.h>

typedef struct { uint64_t s[2]; unsigned char blk[0xff00]; size_t blen; unsigned char *out; size_t olen, ocap; } piece_t;
static __thread piece_t *P;
#define s (P->s)
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }

static char **read_col1(const char *path, size_t *n)
{
    FILE *f = fopen(path, "r"); if (!f) { perror(path); exit(1); }
    size_t cap = 1024; char **v = malloc(cap * sizeof *v); char line[2048]; *n = 0;
    while (fgets(line, sizeof line, f)) { line[strcspn(line, "\t\r\n")] = 0; if (*n == cap) { cap *= 2; v = realloc(v, cap * sizeof *v); } v[(*n)++] = strdup(line); }
    fclose(f); return v;
}

static FILE *out; static int g_level = 1;
#define blk (P->blk)
#define blen (P->blen)
static void flush_block(void)
{
    unsigned char comp[0x10000 + 64]; z_stream z; memset(&z, 0, sizeof z);
    deflateInit2(&z, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    z.next_in = blk; z.avail_in = (uInt)blen; z.next_out = comp + 18; z.avail_out = sizeof comp - 26;
    deflate(&z, Z_FINISH); size_t clen = z.total_out; deflateEnd(&z);
    static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    memcpy(comp, hdr, 16); uint16_t bsize = (uint16_t)(clen + 25); memcpy(comp + 16, &bsize, 2);
    uint32_t crc = (uint32_t)crc32(crc32(0, NULL, 0), blk, (uInt)blen), isz = (uint32_t)blen;
    memcpy(comp + 18 + clen, &crc, 4); memcpy(comp + 22 + clen, &isz, 4);
    if (P->olen + clen + 26 > P->ocap) { P->ocap = P->ocap ? P->ocap * 2 : (1 << 22); P->out = realloc(P->out, P->ocap); if (!P->out) { perror("realloc"); exit(1); } }
    memcpy(P->out + P->olen, comp, clen + 26); P->olen += clen + 26; blen = 0;
}
static void put(const void *p, size_t n)
{
    const unsigned char *q = p;
    while (n) { size_t k = sizeof blk - blen; if (k > n) k = n; memcpy(blk + blen, q, k); blen += k; q += k; n -= k; if (blen == sizeof blk) flush_block(); }
}

enum { PIECES = 64 };
static piece_t *g_piece[PIECES + 1];
static size_t g_n, g_nb, g_ng, g_name0; static char **g_bar, **g_gen; static uint64_t g_seed; static int g_ul, g_sl; static int g_next;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static void gen_piece(int pc);
static void *worker(void *vp)
{
    (void)vp;
    for (;;) {
        pthread_mutex_lock(&g_mu); int pc = g_next++; pthread_mutex_unlock(&g_mu);
        if (pc >= PIECES) return NULL;
        gen_piece(pc);
    }
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: gen_bam out.bam barcodes features n [seed] [umi_len] [seq_len] [threads] [repeat] [bodies]\n"); return 1; }
    g_bar = read_col1(argv[2], &g_nb); g_gen = read_col1(argv[3], &g_ng);
    g_n = strtoull(argv[4], NULL, 10); g_seed = argc > 5 ? strtoull(argv[5], NULL, 10) : 1; g_ul = argc > 6 ? atoi(argv[6]) : 10;
    g_sl = argc > 7 ? atoi(argv[7]) : 0; if (g_sl > 150) g_sl = 150; if (g_sl > 0) g_level = 6;
    int nt = argc > 8 ? atoi(argv[8]) : 1; if (nt < 1) nt = 1; if (nt > 64) nt = 64;
    int repeat = argc > 9 ? atoi(argv[9]) : 1; if (repeat < 1) repeat = 1;
    int bodies = argc > 10 ? atoi(argv[10]) : 1; if (bodies < 1) bodies = 1;
    out = fopen(argv[1], "wb"); if (!out) { perror(argv[1]); return 1; }
    for (int i = 0; i <= PIECES; i++) { g_piece[i] = calloc(1, sizeof(piece_t)); if (!g_piece[i]) { perror("calloc"); return 1; } }
    /* the header is a block run of its own (piece PIECES, written first) */
    P = g_piece[PIECES];
    const char *text = "@HD\tVN:1.6\tSO:unsorted\n"; int32_t lt = (int32_t)strlen(text), nref = 0;
    put("BAM\1", 4); put(&lt, 4); put(text, lt);
    if (g_sl > 0) { nref = 1; put(&nref, 4); int32_t ln = 5, lref = 248956422; put(&ln, 4); put("chr1", 5); put(&lref, 4); }   /* mapped reads name refID 0 */
    else put(&nref, 4);
    if (blen) flush_block();
    fwrite(g_piece[PIECES]->out, 1, g_piece[PIECES]->olen, out);
    const uint64_t seed0 = g_seed;
    for (int b = 0; b < bodies; b++) {
        g_seed = seed0 + 7919ull * (uint64_t)b; g_name0 = g_n * (size_t)b; g_next = 0;
        for (int i = 0; i < PIECES; i++) { P = g_piece[i]; P->olen = 0; blen = 0; }
        pthread_t th[64];
        for (int t = 0; t < nt; t++) pthread_create(&th[t], NULL, worker, NULL);
        for (int t = 0; t < nt; t++) pthread_join(th[t], NULL);
        for (int r = 0; r < repeat; r++)
            for (int i = 0; i < PIECES; i++) fwrite(g_piece[i]->out, 1, g_piece[i]->olen, out);
    }
    static const unsigned char eof[28] = {0x1f,0x8b,8,4,0,0,0,0,0,0xff,6,0,'B','C',2,0,0x1b,0,3,0,0,0,0,0,0,0,0,0};
    fwrite(eof, 1, 28, out); fclose(out);
    return 0;
}

static void gen_piece(int pc)
{
    P = g_piece[pc];
    const size_t n0 = g_n * (size_t)pc / PIECES, n1 = g_n * (size_t)(pc + 1) / PIECES, nb = g_nb, ng = g_ng;
    char **bar = g_bar, **gen = g_gen; const int ul = g_ul, sl = g_sl; const uint64_t seed = g_seed * 1000003ull + (uint64_t)pc;
    s[0] = seed * 0x9E3779B97F4A7C15ull + 1; s[1] = seed ^ 0xD1B54A32D192ED03ull; for (int i = 0; i < 8; i++) rnd();
    unsigned char rec[1024];
    for (size_t i = n0; i < n1; i++) {
        unsigned char *p = rec + 4; int32_t m1 = -1, z = 0; char name[24]; int nl = snprintf(name, sizeof name, "r%zu", g_name0 + i) + 1;
        memcpy(p, &m1, 4); memcpy(p + 4, &m1, 4); p[8] = (unsigned char)nl; p[9] = 0; uint16_t bin = 4680, nc = 0, fl = 4;
        memcpy(p + 10, &bin, 2); memcpy(p + 12, &nc, 2); memcpy(p + 14, &fl, 2); memcpy(p + 16, &z, 4);
        memcpy(p + 20, &m1, 4); memcpy(p + 24, &m1, 4); memcpy(p + 28, &z, 4);
        if (sl > 0) {                               /* mapped read: refID 0, a position, one CIGAR word, bases, qualities */
            int32_t ref = 0, pos = (int32_t)(i * 37 % 100000000); uint16_t one = 1, f0 = (uint16_t)((rnd() & 1) ? 16 : 0); int32_t lseq = sl;
            memcpy(p, &ref, 4); memcpy(p + 4, &pos, 4); p[9] = 255; memcpy(p + 12, &one, 2); memcpy(p + 14, &f0, 2); memcpy(p + 16, &lseq, 4);
        }
        p += 32; memcpy(p, name, nl); p += nl;
        if (sl > 0) {
            uint32_t cig = ((uint32_t)sl << 4) | 0; memcpy(p, &cig, 4); p += 4;
            for (int k = 0; k < (sl + 1) / 2; k++) { uint64_t q = rnd(); *p++ = (unsigned char)(((1 << (q & 3)) << 4) | (1 << ((q >> 2) & 3))); }
            for (int k = 0; k < sl;) { uint64_t q = rnd(); int run = 1 + (int)(q & 31); unsigned char qv = (q >> 5) % 10 < 8 ? 37 : ((q >> 9) & 1 ? 25 : 11);
                                       while (run-- && k < sl) { *p++ = qv; k++; } }
            uint64_t q = rnd();
            memcpy(p, "NHC\1HIC\1ASC", 11); p += 11; *p++ = (unsigned char)(sl - (q & 3)); memcpy(p, "nMC", 3); p += 3; *p++ = (unsigned char)(q >> 4 & 1);
            memcpy(p, "RGZsample:0:1:HXXXXXXXX:1", 26); p += 26;
            *p++ = 'C'; *p++ = 'R'; *p++ = 'Z'; for (int k = 0; k < 16; k++) *p++ = "ACGT"[(q >> (8 + 2 * k)) & 3]; *p++ = 0;
            *p++ = 'C'; *p++ = 'Y'; *p++ = 'Z'; memset(p, 'F', 16); p += 16; *p++ = 0;
            q = rnd(); *p++ = 'U'; *p++ = 'R'; *p++ = 'Z'; for (int k = 0; k < ul; k++) *p++ = "ACGT"[(q >> (2 * k)) & 3]; *p++ = 0;
            *p++ = 'U'; *p++ = 'Y'; *p++ = 'Z'; memset(p, 'F', ul); p += ul; *p++ = 0;
        }
        uint64_t r = rnd();
        *p++ = 'C'; *p++ = 'B'; *p++ = 'Z';
        if ((r & 1023) < 51) { for (int k = 0; k < 16; k++) *p++ = "ACGT"[(rnd() >> 7) & 3]; memcpy(p, "-1", 3); p += 3; }
        else { const char *b = bar[(r >> 10) % nb]; size_t l = strlen(b) + 1; memcpy(p, b, l); p += l; }
        r = rnd(); unsigned xf = (r & 1023) < 870 ? ((r & 1024) ? 25 : 17) : (unsigned)((r >> 12) & 7);
        *p++ = 'x'; *p++ = 'f'; *p++ = 'C'; *p++ = (unsigned char)xf;
        { const char *g = gen[(r >> 16) % ng]; size_t l = strlen(g) + 1; *p++ = 'G'; *p++ = 'X'; *p++ = 'Z'; memcpy(p, g, l); p += l; }
        r = rnd(); *p++ = 'U'; *p++ = 'B'; *p++ = 'Z'; for (int k = 0; k < ul; k++) { *p++ = "ACGT"[r & 3]; r >>= 2; } *p++ = 0;
        int32_t bs = (int32_t)(p - rec - 4); memcpy(rec, &bs, 4); put(rec, (size_t)(p - rec));
    }
    if (blen) flush_block();
}
