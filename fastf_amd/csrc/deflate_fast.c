/*
 * deflate_fast.c — the deflate encoder behind the .gz writers (matrix.mtx.gz, barcodes/features/umi.tsv.gz, crb output).
 *
 * The reference writes its outputs through zlib's gzprintf at the default level (bam2db_ds.c:453-476, 575-650), row by
 * row; parity is defined on the DECOMPRESSED bytes, so the writer may emit any valid deflate stream.  zlib
 * level 6 spends its time walking hash chains (~40 MB/s per core on digit rows): for tens of millions of "f c n\n"
 * rows that made the writer the longest host stage of a run.  This encoder is the usual fast shape instead:
 *   - LZ77 with ONE hash probe per position (4-byte hash, 32 K entries, most recent occurrence): rows of a sorted
 *     matrix repeat their cell column and most of the feature digits from the row before, which is exactly what the
 *     most recent occurrence finds;
 *   - per 128 K-token block a dynamic Huffman code (RFC 1951 section 3.2.7) built from the block's own statistics —
 *     digits and a handful of match lengths/distances code in 2-4 bits;
 *   - a 64-bit bit buffer flushed a byte multiple at a time.
 * Output is a complete gzip member (RFC 1952): header, deflate blocks, CRC-32, ISIZE.
 * Checked against zlib's inflate on structured and random inputs (tests/test_writers.py, tools/fuzz_deflate.c).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

uint32_t fastf_crc32(const unsigned char *buf, size_t len);

enum { HASH_BITS = 15, HASH_SIZE = 1 << HASH_BITS, MAX_MATCH = 258, MIN_MATCH = 4, WINDOW = 32768,
       BLOCK_TOKENS = 1 << 17, N_LITLEN = 288, N_DIST = 30, MAX_BITS = 15 };

/* length -> (symbol, extra bits, base), distance likewise (RFC 1951 section 3.2.5) */
static const uint16_t k_len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t k_dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                         4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint8_t g_len_sym[259];          /* match length 3..258 -> symbol index 0..28 */
static uint8_t g_dist_sym_lo[256];      /* distance 1..256 -> symbol */
static uint8_t g_dist_sym_hi[256];      /* (distance - 1) >> 7 -> symbol, distances 257..32768 */
static int g_tables_ready;

static void init_tables(void)
{
    for (int s = 0; s < 29; s++) {
        int hi = s == 28 ? 258 : k_len_base[s] + (1 << k_len_extra[s]) - 1;
        for (int l = k_len_base[s]; l <= hi && l <= 258; l++) g_len_sym[l] = (uint8_t)s;
    }
    g_len_sym[258] = 28;
    for (int s = 0; s < 30; s++) {
        int lo = k_dist_base[s], hi = lo + (1 << k_dist_extra[s]) - 1;
        for (int d = lo; d <= hi; d++) {
            if (d <= 256) g_dist_sym_lo[d - 1] = (uint8_t)s;
            else g_dist_sym_hi[(d - 1) >> 7] = (uint8_t)s;
        }
    }
    __atomic_store_n(&g_tables_ready, 1, __ATOMIC_RELEASE);
}
static inline unsigned dist_sym(unsigned d) { return d <= 256 ? g_dist_sym_lo[d - 1] : g_dist_sym_hi[(d - 1) >> 7]; }

/* ---- bit writer ---- */
typedef struct { unsigned char *p, *end; uint64_t acc; unsigned n; int overflow; } bitw;
static inline void bw_put(bitw *w, uint32_t v, unsigned k)          /* k <= 32; at most 31 bits pending on entry */
{
    w->acc |= (uint64_t)v << w->n; w->n += k;
    if (w->n >= 32) {
        if (w->end - w->p < 8) { w->overflow = 1; w->n = 0; w->acc = 0; return; }
        memcpy(w->p, &w->acc, 4); w->p += 4; w->acc >>= 32; w->n -= 32;
    }
}
static inline void bw_flush_bytes(bitw *w)
{
    while (w->n > 0) {
        if (w->p >= w->end) { w->overflow = 1; return; }
        *w->p++ = (unsigned char)w->acc; w->acc >>= 8; w->n = w->n > 8 ? w->n - 8 : 0;
    }
}

/* ---- length-limited Huffman code lengths ----
 * Plain Huffman by two-queue merge over the sorted symbols; depths beyond MAX limit are cut back the way zlib's
 * gen_bitlen does (move overflowing leaves up, pay with the deepest leaves that still have room). */
typedef struct { uint32_t freq; uint16_t sym; } hsym;
static int cmp_hsym(const void *a, const void *b)
{
    const hsym *x = (const hsym *)a, *y = (const hsym *)b;
    if (x->freq != y->freq) return x->freq < y->freq ? -1 : 1;
    return (int)x->sym - (int)y->sym;
}

static void huff_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *len)
{
    hsym leaves[N_LITLEN]; int nl = 0;
    memset(len, 0, (size_t)n);
    for (int i = 0; i < n; i++) if (freq[i]) { leaves[nl].freq = freq[i]; leaves[nl].sym = (uint16_t)i; nl++; }
    if (nl == 0) return;
    if (nl == 1) { len[leaves[0].sym] = 1; return; }
    qsort(leaves, (size_t)nl, sizeof leaves[0], cmp_hsym);
    /* nodes: 0..nl-1 leaves, nl.. internal; parent[] */
    uint64_t w[2 * N_LITLEN]; int parent[2 * N_LITLEN];
    for (int i = 0; i < nl; i++) w[i] = leaves[i].freq;
    int a = 0, b = nl, next = nl;                        /* a: next unused leaf, b: next unused internal node */
    for (int k = 0; k < nl - 1; k++) {
        int pick[2];
        for (int t = 0; t < 2; t++) {
            if (a < nl && (b >= next || w[a] <= w[b])) pick[t] = a++;
            else pick[t] = b++;
        }
        w[next] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = next; parent[pick[1]] = next;
        next++;
    }
    int depth[2 * N_LITLEN];
    depth[next - 1] = 0;
    for (int i = next - 2; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
    int bl_count[64]; memset(bl_count, 0, sizeof bl_count);
    int overflow = 0;
    for (int i = 0; i < nl; i++) bl_count[depth[i] > max_bits ? max_bits : depth[i]]++;
    /* every node below the limit, leaf or internal, costs one unit: a subtree of L leaves hanging under a node at the
     * limit holds 2L - 2 such nodes and needs L - 1 of the steps below (each step settles two units) */
    for (int i = 0; i < next - 1; i++) overflow += depth[i] > max_bits;
    if (overflow) {
        /* Kraft sum is now too large: repeatedly take a leaf from the deepest level that still has one above max_bits-1
         * ... zlib's gen_bitlen loop */
        do {
            int bits = max_bits - 1;
            while (bl_count[bits] == 0) bits--;
            bl_count[bits]--;                 /* move one leaf down the tree */
            bl_count[bits + 1] += 2;          /* move one overflow item as its brother */
            bl_count[max_bits]--;
            overflow -= 2;
        } while (overflow > 0);
    }
    /* hand the lengths out: leaves are sorted by ascending frequency, so the rarest get the longest codes */
    int i = 0;
    for (int bits = max_bits; bits >= 1; bits--)
        for (int c = bl_count[bits]; c > 0; c--) len[leaves[i++].sym] = (uint8_t)bits;
}

/* a prefix code exists for these lengths iff the Kraft sum does not exceed 1 (checked in units of 2^-15) */
static int kraft_ok(const uint8_t *len, int n)
{
    uint32_t k = 0;
    for (int i = 0; i < n; i++) if (len[i]) k += 1u << (MAX_BITS - len[i]);
    return k <= (1u << MAX_BITS);
}

static void huff_codes(const uint8_t *len, int n, uint16_t *code)
{
    int bl_count[MAX_BITS + 1] = {0}; uint16_t next_code[MAX_BITS + 2];
    for (int i = 0; i < n; i++) bl_count[len[i]]++;
    bl_count[0] = 0;
    uint16_t c = 0;
    for (int b = 1; b <= MAX_BITS; b++) { c = (uint16_t)((c + bl_count[b - 1]) << 1); next_code[b] = c; }
    for (int i = 0; i < n; i++) {
        if (!len[i]) { code[i] = 0; continue; }
        uint16_t v = next_code[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; b++) { r = (uint16_t)((r << 1) | (v & 1)); v >>= 1; }   /* codes go out LSB first: store them reversed */
        code[i] = r;
    }
}

/* ---- one block: tokens -> dynamic-Huffman deflate block ---- */
typedef struct {
    uint32_t *tok; size_t n;                      /* literal: byte; match: 0x80000000 | (len << 16) | dist  (dist 1..32768 stored as dist - 1) */
    uint32_t lfreq[N_LITLEN], dfreq[N_DIST];
} block_t;

static void write_block(bitw *w, block_t *b, int final)
{
    uint8_t llen[N_LITLEN], dlen[N_DIST]; uint16_t lcode[N_LITLEN], dcode[N_DIST];
    b->lfreq[256] = 1;
    huff_lengths(b->lfreq, N_LITLEN, MAX_BITS, llen);
    huff_lengths(b->dfreq, N_DIST, MAX_BITS, dlen);
    int nd_used = 0; for (int i = 0; i < N_DIST; i++) nd_used += dlen[i] != 0;
    if (nd_used == 0) dlen[0] = 1;                 /* at least one distance code must be described */
#ifdef DEFLATE_DEBUG
    { double kl = 0, kd = 0; int ml = 0; for (int i = 0; i < N_LITLEN; i++) if (llen[i]) { kl += 1.0 / (1 << llen[i]); if (llen[i] > ml) ml = llen[i]; }
      for (int i = 0; i < N_DIST; i++) if (dlen[i]) kd += 1.0 / (1 << dlen[i]);
      fprintf(stderr, "block n=%zu kraft litlen %.6f (max %d) dist %.6f\n", b->n, kl, ml, kd); }
#endif
    huff_codes(llen, N_LITLEN, lcode);
    huff_codes(dlen, N_DIST, dcode);
    int hlit = N_LITLEN - 2;                       /* symbols 286, 287 never occur */
    while (hlit > 257 && llen[hlit - 1] == 0) hlit--;
    int hdist = N_DIST; while (hdist > 1 && dlen[hdist - 1] == 0) hdist--;
    /* code-length alphabet with run-length symbols 16/17/18 */
    uint8_t seq[N_LITLEN + N_DIST]; int ns = 0;
    memcpy(seq, llen, (size_t)hlit); ns = hlit; memcpy(seq + ns, dlen, (size_t)hdist); ns += hdist;
    uint8_t cl_sym[N_LITLEN + N_DIST]; uint8_t cl_ext[N_LITLEN + N_DIST]; int ncl = 0;
    uint32_t clfreq[19] = {0};
    for (int i = 0; i < ns;) {
        int v = seq[i], run = 1;
        while (i + run < ns && seq[i + run] == v) run++;
        i += run;
        if (v == 0) {
            while (run >= 11) { int r = run > 138 ? 138 : run; cl_sym[ncl] = 18; cl_ext[ncl++] = (uint8_t)(r - 11); clfreq[18]++; run -= r; }
            if (run >= 3) { cl_sym[ncl] = 17; cl_ext[ncl++] = (uint8_t)(run - 3); clfreq[17]++; run = 0; }
            while (run-- > 0) { cl_sym[ncl] = 0; cl_ext[ncl++] = 0; clfreq[0]++; }
        } else {
            cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; clfreq[v]++; run--;
            while (run >= 3) { int r = run > 6 ? 6 : run; cl_sym[ncl] = 16; cl_ext[ncl++] = (uint8_t)(r - 3); clfreq[16]++; run -= r; }
            while (run-- > 0) { cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; clfreq[v]++; }
        }
    }
    uint8_t cllen[19]; uint16_t clcode[19];
    huff_lengths(clfreq, 19, 7, cllen);
    /* belt and braces: lengths that admit no prefix code make the caller fall back to zlib for this member */
    if (!kraft_ok(llen, N_LITLEN) || !kraft_ok(dlen, N_DIST) || !kraft_ok(cllen, 19)) { w->overflow = 1; b->n = 0; return; }
#ifdef DEFLATE_DEBUG
    { double k = 0; for (int i = 0; i < 19; i++) if (cllen[i]) k += 1.0 / (1 << cllen[i]);
      fprintf(stderr, "  kraft cl %.6f:", k); for (int i = 0; i < 19; i++) fprintf(stderr, " %d(%u)", cllen[i], clfreq[i]); fprintf(stderr, "\n"); }
#endif
    huff_codes(cllen, 19, clcode);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19; while (hclen > 4 && cllen[order[hclen - 1]] == 0) hclen--;
    bw_put(w, (uint32_t)(final ? 1 : 0) | (2u << 1), 3);
    bw_put(w, (uint32_t)(hlit - 257), 5); bw_put(w, (uint32_t)(hdist - 1), 5); bw_put(w, (uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; i++) bw_put(w, cllen[order[i]], 3);
    for (int i = 0; i < ncl; i++) {
        bw_put(w, clcode[cl_sym[i]], cllen[cl_sym[i]]);
        if (cl_sym[i] == 16) bw_put(w, cl_ext[i], 2);
        else if (cl_sym[i] == 17) bw_put(w, cl_ext[i], 3);
        else if (cl_sym[i] == 18) bw_put(w, cl_ext[i], 7);
    }
    /* the tokens */
    for (size_t i = 0; i < b->n; i++) {
        const uint32_t t = b->tok[i];
        if (!(t & 0x80000000u)) { bw_put(w, lcode[t], llen[t]); continue; }
        const unsigned len = (t >> 16) & 0x1FF, dist = (t & 0xFFFF) + 1;
        const unsigned ls = g_len_sym[len];
        bw_put(w, lcode[257 + ls], llen[257 + ls]);
        if (k_len_extra[ls]) bw_put(w, len - k_len_base[ls], k_len_extra[ls]);
        const unsigned ds = dist_sym(dist);
        bw_put(w, dcode[ds], dlen[ds]);
        if (k_dist_extra[ds]) bw_put(w, dist - k_dist_base[ds], k_dist_extra[ds]);
    }
    bw_put(w, lcode[256], llen[256]);
    b->n = 0; memset(b->lfreq, 0, sizeof b->lfreq); memset(b->dfreq, 0, sizeof b->dfreq);
}

static inline uint32_t rd32u(const unsigned char *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash4(uint32_t v) { return (v * 0x9E3779B1u) >> (32 - HASH_BITS); }

/* upper bound of the gzip member for `len` input bytes: with this token format a dynamic block never expands beyond
 * ~9 bits per literal + a 200-byte table, but be generous */
size_t fastf_gz_bound(size_t len) { return len + len / 4 + 1024; }

/* text -> one complete gzip member in out[0, cap); returns its size, 0 when cap was too small */
size_t fastf_gz_member_fast(const unsigned char *in, size_t len, unsigned char *out, size_t cap)
{
    {   /* sixteen writer threads compress their first members at the same moment: the tables are built by one of them */
        static pthread_once_t once = PTHREAD_ONCE_INIT;
        if (!__atomic_load_n(&g_tables_ready, __ATOMIC_ACQUIRE)) (void)pthread_once(&once, init_tables);
    }
    if (cap < 32) return 0;
    static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 0xff};     /* XFL 4: fastest algorithm; OS unknown */
    memcpy(out, hdr, 10);
    bitw w = { out + 10, out + cap - 8, 0, 0, 0 };
    block_t blk; memset(&blk, 0, sizeof blk);
    blk.tok = (uint32_t *)malloc((size_t)BLOCK_TOKENS * sizeof(uint32_t));
    uint32_t *head = (uint32_t *)malloc((size_t)HASH_SIZE * sizeof(uint32_t));      /* position + 1 of the last occurrence; 0 = none */
    if (!blk.tok || !head) { free(blk.tok); free(head); return 0; }
    memset(head, 0, (size_t)HASH_SIZE * sizeof(uint32_t));
    size_t pos = 0;
    /* positions are kept as 32-bit offsets from `base` so that inputs beyond 4 GiB keep working */
    const unsigned char *base = in; size_t base_off = 0;
    while (pos < len) {
        if (pos - base_off > 0xF0000000u) {                                            /* rebase the hash table */
            memset(head, 0, (size_t)HASH_SIZE * sizeof(uint32_t)); base = in + pos; base_off = pos;
        }
        if (len - pos < MIN_MATCH + 4) {                                               /* tail: literals */
            blk.tok[blk.n++] = in[pos]; blk.lfreq[in[pos]]++; pos++;
        } else {
            const uint32_t v = rd32u(in + pos), h = hash4(v);
            const uint32_t cand = head[h];
            head[h] = (uint32_t)(pos - base_off) + 1;
            size_t cpos = base_off + cand - 1;
            if (cand && pos - cpos <= WINDOW && rd32u(in + cpos) == v) {
                size_t maxl = len - pos < MAX_MATCH ? len - pos : MAX_MATCH, l = 4;
                while (l + 8 <= maxl) {
                    uint64_t x, y; memcpy(&x, in + cpos + l, 8); memcpy(&y, in + pos + l, 8);
                    if (x != y) { l += (size_t)(__builtin_ctzll(x ^ y) >> 3); goto matched; }
                    l += 8;
                }
                while (l < maxl && in[cpos + l] == in[pos + l]) l++;
            matched:
                if (l > maxl) l = maxl;
                blk.tok[blk.n++] = 0x80000000u | ((uint32_t)l << 16) | (uint32_t)(pos - cpos - 1);
                blk.lfreq[257 + g_len_sym[l]]++; blk.dfreq[dist_sym((unsigned)(pos - cpos))]++;
                /* keep the table fresh inside the match: the position after its start and its last four bytes */
                if (len - (pos + l) >= 8) {
                    head[hash4(rd32u(in + pos + 1))] = (uint32_t)(pos + 1 - base_off) + 1;
                    head[hash4(rd32u(in + pos + l - 3))] = (uint32_t)(pos + l - 3 - base_off) + 1;
                    head[hash4(rd32u(in + pos + l - 2))] = (uint32_t)(pos + l - 2 - base_off) + 1;
                    head[hash4(rd32u(in + pos + l - 1))] = (uint32_t)(pos + l - 1 - base_off) + 1;
                }
                pos += l;
            } else {
                blk.tok[blk.n++] = in[pos]; blk.lfreq[in[pos]]++; pos++;
            }
        }
        if (blk.n == BLOCK_TOKENS) write_block(&w, &blk, 0);
    }
    (void)base;
    write_block(&w, &blk, 1);                                                           /* final block (possibly empty) */
    bw_flush_bytes(&w);
    free(blk.tok); free(head);
    if (w.overflow) return 0;
    const uint32_t crc = fastf_crc32(in, len), isz = (uint32_t)len;
    memcpy(w.p, &crc, 4); memcpy(w.p + 4, &isz, 4);
    return (size_t)(w.p + 8 - out);
}
