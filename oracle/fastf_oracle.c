/*
 * fastf_oracle.c — CPU oracle (TEST INFRASTRUCTURE, see fastf_oracle.h).
 *
 * Plain-C restatement of the reference bam2db path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/src).  The
 * reference stores rows in SQLite and lets it aggregate; SQLite is a third-
 * party dependency (unpinned system library, 3.3x), so the aggregate is
 * restated from its documented semantics and cross-checked against the real
 * library in tests/test_oracle_pins.py::test_aggregate_matches_sqlite.
 *
 * Pinning status.  Pinned against the reference's own sources compiled in place (oracle/_ref, `make -C oracle ref`):
 * the MT19937 stream and genrand_real1, SampleInt + the generator state after it, the first-wins / exact-match hash
 * table (tests/test_oracle_pins.py).  PARITY UNPINNED for what restates bam2db_ds.c itself — the record loop, the 2-bit
 * codec, the header text and the writers: that file needs htslib, which is neither in this image nor vendored, and the
 * reference ships no golden vectors for the path.  Those parts rest on the line-by-line restatement, on the known
 * answers recorded in SURVEY.md section 8c (tests/golden/survey_8c.json) and on real SQLite running the reference's two
 * aggregate statements over the oracle's rows.
 */
#define _GNU_SOURCE
#include "fastf_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* MT19937 — mt19937ar.c:44-57 (parameters, global state)              */
/* ------------------------------------------------------------------ */
#define MT_N 624
#define MT_M 397
static uint32_t g_mt[MT_N];
static int g_mti = MT_N + 1;

/* mt19937ar.c:60-73 */
void oracle_init_genrand(uint32_t s)
{
    g_mt[0] = s;
    for (g_mti = 1; g_mti < MT_N; g_mti++)
        g_mt[g_mti] = 1812433253u * (g_mt[g_mti - 1] ^ (g_mt[g_mti - 1] >> 30)) + (uint32_t)g_mti;
}

/* mt19937ar.c:105-140 */
uint32_t oracle_genrand_int32(void)
{
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    uint32_t y;
    if (g_mti >= MT_N) {
        int kk;
        if (g_mti == MT_N + 1) oracle_init_genrand(5489u);
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (g_mt[kk] & 0x80000000u) | (g_mt[kk + 1] & 0x7fffffffu);
            g_mt[kk] = g_mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; kk++) {
            y = (g_mt[kk] & 0x80000000u) | (g_mt[kk + 1] & 0x7fffffffu);
            g_mt[kk] = g_mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (g_mt[MT_N - 1] & 0x80000000u) | (g_mt[0] & 0x7fffffffu);
        g_mt[MT_N - 1] = g_mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        g_mti = 0;
    }
    y = g_mt[g_mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* test convenience, no algorithm of its own: init_genrand(seed), `skip` draws thrown away, then n calls of genrand_int32() into
 * out[] (what a Python loop over oracle_genrand_int32() returns, without its million ctypes calls); state[0..623] (if not NULL):
 * the generator's array afterwards */
void oracle_mt_draws(uint32_t seed, uint64_t skip, uint64_t n, uint32_t *out, uint32_t *state)
{
    oracle_init_genrand(seed);
    for (uint64_t i = 0; i < skip; i++) (void)oracle_genrand_int32();
    for (uint64_t i = 0; i < n; i++) out[i] = oracle_genrand_int32();
    if (state) for (int i = 0; i < MT_N; i++) state[i] = g_mt[i];
}

/* mt19937ar.c:149-153 */
double oracle_genrand_real1(void)
{
    return oracle_genrand_int32() * (1.0 / 4294967295.0);
}

/* bam2db_ds.c:385-390: `rand_depth >= rate_depth` → skip (float promoted to double) */
int oracle_keep_draw(uint32_t draw, float rate_depth)
{
    double r = draw * (1.0 / 4294967295.0);
    return (r >= rate_depth) ? 0 : 1;
}

/* ------------------------------------------------------------------ */
/* cell sub-sampling — utils.c:3-27 (GetSeqInt), :29-75 (SampleInt),    */
/* :86-89 (vsI); call site bam2db_ds.c:240-244                          */
/* ------------------------------------------------------------------ */
size_t oracle_n_cells_sampled(size_t n_cells, float rate_cell)
{
    /* bam2db_ds.c:241: size_t * float → float arithmetic, truncation */
    return (size_t)(n_cells * rate_cell);
}

static int cmp_u64(const void *a, const void *b)
{
    /* utils.c:86-89 returns the (int-truncated) difference; identical ordering
     * for line numbers < 2^31 */
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

int oracle_sample_cells(size_t n_total, size_t n_sample, unsigned int seed, uint64_t *out)
{
    oracle_init_genrand(seed);                       /* utils.c:32 */
    if (n_sample > n_total) return 1;                /* utils.c:38-43 (reference exits) */
    uint64_t *pool = (uint64_t *)malloc((n_total ? n_total : 1) * sizeof(uint64_t));
    if (!pool) return 1;
    for (size_t i = 0; i < n_total; i++) pool[i] = i;   /* GetSeqInt(0, n-1, 1) */
    if (n_total == n_sample) {                       /* utils.c:48-51: identity, no draws */
        memcpy(out, pool, n_total * sizeof(uint64_t));
        free(pool);
        return 0;
    }
    size_t live = n_total;
    for (size_t i = 0; i < n_sample; i++) {          /* utils.c:53-62 */
        size_t idx = oracle_genrand_int32() % live;
        out[i] = pool[idx];
        if (idx != live - 1) pool[idx] = pool[live - 1];
        live--;
    }
    free(pool);
    qsort(out, n_sample, sizeof(uint64_t), cmp_u64); /* bam2db_ds.c:244 */
    return 0;
}

/* ------------------------------------------------------------------ */
/* djb2 + 2-bit codec — bam2db_ds.c:5-104                               */
/* ------------------------------------------------------------------ */
uint64_t oracle_djb2(const char *s, size_t len)
{
    uint64_t h = 5381;
    for (size_t i = 0; i < len; i++) h = ((h << 5) + h) + (uint64_t)(int64_t)s[i]; /* char is signed */
    return h;
}

static int base_code(char c)    /* bam2db_ds.c:5-20 */
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
    return -1;
}

int oracle_encode_dna(const char *seq, uint8_t *out, size_t out_cap, size_t *nbytes)
{
    size_t len = (uint16_t)strlen(seq);              /* bam2db_ds.c:25: uint16_t len_DNA */
    size_t nb = (strlen(seq) + 3) / 4;               /* bam2db_ds.c:419 */
    if (nb > out_cap) return -2;
    memset(out, 0, out_cap);
    for (size_t i = 0; i < len; i++) {               /* bam2db_ds.c:34-49, MSB first */
        int c = base_code(seq[i]);
        if (c < 0) return -1;                        /* NULL blob */
        out[i / 4] |= (uint8_t)(c << (6 - 2 * (i % 4)));
    }
    *nbytes = nb;
    return 0;
}

void oracle_decode_dna(const uint8_t *blob, size_t n_bases, char *out)
{
    for (size_t i = 0; i < n_bases; i++)             /* bam2db_ds.c:64-91 */
        out[i] = "ACGT"[(blob[i / 4] >> (6 - 2 * (i % 4))) & 3];
    out[n_bases] = '\0';
}

/* ------------------------------------------------------------------ */
/* string → index table — hashtable.c:3-7,70-115 (chained, 2^20        */
/* buckets, head insertion, first key wins, strcmp verification)        */
/* ------------------------------------------------------------------ */
typedef struct ht_ent { char *key; int64_t val; struct ht_ent *next; } ht_ent;
typedef struct { uint32_t size; ht_ent **b; } ht_t;

static ht_t *ht_new(uint32_t size)
{
    ht_t *h = (ht_t *)malloc(sizeof(ht_t));
    h->size = size;
    h->b = (ht_ent **)calloc(size, sizeof(ht_ent *));
    return h;
}
static ht_ent *ht_find(const ht_t *h, const char *key)
{
    ht_ent *e = h->b[oracle_djb2(key, strlen(key)) % h->size];
    while (e && strcmp(e->key, key) != 0) e = e->next;
    return e;
}
static int ht_put(ht_t *h, const char *key, int64_t val)   /* 1 inserted, 0 duplicate */
{
    if (ht_find(h, key)) return 0;
    size_t i = oracle_djb2(key, strlen(key)) % h->size;
    ht_ent *e = (ht_ent *)malloc(sizeof(ht_ent));
    e->key = strdup(key); e->val = val; e->next = h->b[i]; h->b[i] = e;
    return 1;
}
static void ht_free(ht_t *h)
{
    if (!h) return;
    for (uint32_t i = 0; i < h->size; i++)
        for (ht_ent *e = h->b[i]; e;) { ht_ent *n = e->next; free(e->key); free(e); e = n; }
    free(h->b); free(h);
}

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */
typedef struct { char *p; size_t len, cap; } sbuf;
static void sb_printf(sbuf *s, const char *fmt, ...)
{
    va_list ap;
    for (;;) {
        va_start(ap, fmt);
        int n = vsnprintf(s->p + s->len, s->cap - s->len, fmt, ap);
        va_end(ap);
        if (n >= 0 && (size_t)n < s->cap - s->len) { s->len += (size_t)n; return; }
        s->cap = s->cap ? s->cap * 2 : 4096;
        if (n > 0 && s->cap < s->len + (size_t)n + 1) s->cap = s->len + (size_t)n + 1;
        s->p = (char *)realloc(s->p, s->cap);
    }
}
static void sb_int(sbuf *s, int v)          /* fast "%d" for the row writer */
{
    if (s->cap - s->len < 16) { s->cap = s->cap ? s->cap * 2 : 4096; s->p = (char *)realloc(s->p, s->cap); }
    char tmp[12]; int n = 0; unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) s->p[s->len++] = '-';
    while (n) s->p[s->len++] = tmp[--n];
}
static void sb_ch(sbuf *s, char c)
{
    if (s->cap - s->len < 2) { s->cap = s->cap ? s->cap * 2 : 4096; s->p = (char *)realloc(s->p, s->cap); }
    s->p[s->len++] = c;
}

/* gzgets() over a memory image: at most cap-1 chars, stops after '\n' */
static int mem_gets(const char *buf, size_t len, size_t *pos, char *line, size_t cap)
{
    if (*pos >= len) return 0;
    size_t n = 0;
    while (n < cap - 1 && *pos < len) {
        char c = buf[(*pos)++];
        line[n++] = c;
        if (c == '\n') break;
    }
    line[n] = '\0';
    return 1;
}

/* ------------------------------------------------------------------ */
/* aggregate — bam2db_ds.c:480-483 (GROUP BY cell, feature;            */
/* COUNT(DISTINCT blob)); :539-542 for -u.  SQLite semantics relied on: */
/* sorter output ascending by the GROUP BY terms; NULL excluded from    */
/* COUNT(DISTINCT); NULL sorts before BLOB; BLOB order = memcmp over    */
/* the common prefix, then length.                                      */
/* ------------------------------------------------------------------ */
typedef struct { const oracle_result_t *r; } sort_ctx;

static int row_cmp(const void *pa, const void *pb, void *vctx)
{
    const oracle_result_t *r = ((sort_ctx *)vctx)->r;
    size_t a = *(const size_t *)pa, b = *(const size_t *)pb;
    if (r->row_cell[a] != r->row_cell[b]) return r->row_cell[a] < r->row_cell[b] ? -1 : 1;
    if (r->row_feature[a] != r->row_feature[b]) return r->row_feature[a] < r->row_feature[b] ? -1 : 1;
    int la = r->row_blob_len[a], lb = r->row_blob_len[b];
    if (la < 0 || lb < 0) return (la < 0 && lb < 0) ? 0 : (la < 0 ? -1 : 1);
    int c = memcmp(r->row_blob + a * ORC_MAX_BLOB, r->row_blob + b * ORC_MAX_BLOB, (size_t)(la < lb ? la : lb));
    if (c) return c < 0 ? -1 : 1;
    return (la > lb) - (la < lb);
}

static int fail(oracle_result_t *out, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(out->err, sizeof out->err, fmt, ap); va_end(ap);
    return 1;
}

int oracle_bam2db(const char *barcodes, size_t barcodes_len,
                  const char *features, size_t features_len,
                  size_t n_rec, const uint8_t *flags, const int32_t *xf,
                  const char *cb, size_t cb_stride,
                  const char *gx, size_t gx_stride,
                  const char *ub, size_t ub_stride,
                  float rate_cell, float rate_depth, unsigned int seed,
                  const char *bam_label, int umi_copies,
                  oracle_result_t *out)
{
    memset(out, 0, sizeof *out);
    sbuf s_bar = {0}, s_feat = {0}, s_mtx = {0}, s_umi = {0};
    char line[1024];
    int rc = 1;
    ht_t *ht_cell = NULL, *ht_feature = NULL;
    size_t *order = NULL;

    oracle_init_genrand(seed);                           /* bam2db_ds.c:122 */
    ht_cell = ht_new(1u << 20);                          /* :213 */
    ht_feature = ht_new(1u << 20);                       /* :221 */

    /* ---- phase C: cell sub-sampling, bam2db_ds.c:229-292 ---- */
    size_t n_cells = 0, pos = 0;
    while (mem_gets(barcodes, barcodes_len, &pos, line, sizeof line)) n_cells++;   /* :233-236 */
    if (!(rate_cell >= 0.0f)) { fail(out, "rate_cell %g undefined in reference", rate_cell); goto done; }
    size_t n_sampled = oracle_n_cells_sampled(n_cells, rate_cell);                 /* :241 */
    if (n_sampled > n_cells) { fail(out, "Sample size must be smaller than population size"); goto done; }
    out->sampled_lines = (uint64_t *)malloc((n_cells ? n_cells : 1) * sizeof(uint64_t));
    if (oracle_sample_cells(n_cells, n_sampled, seed, out->sampled_lines)) { fail(out, "sample failed"); goto done; }
    out->n_sampled = n_sampled;

    size_t cell_index = 1, nth = 0;
    pos = 0;
    while (mem_gets(barcodes, barcodes_len, &pos, line, sizeof line) && cell_index <= n_sampled) { /* :255 */
        nth++;
        if (nth - 1 != out->sampled_lines[cell_index - 1]) continue;               /* :260 */
        line[strcspn(line, "\n\r\t")] = '\0';                                      /* :265 */
        if (ht_put(ht_cell, line, (int64_t)cell_index)) {                          /* :268 */
            sb_printf(&s_bar, "%s\n", line);           /* cell table row → table2gz(:520) */
            cell_index++;
        }   /* duplicate: cell_index not advanced (:281-285) — later lines never match :260 again */
    }
    out->n_barcode = cell_index - 1;

    /* ---- phase D: feature table, bam2db_ds.c:296-337 ---- */
    int feature_index = 1;
    pos = 0;
    while (mem_gets(features, features_len, &pos, line, sizeof line)) {             /* :304 */
        char *id = strtok(line, "\t");                                             /* :306 */
        char *name = strtok(NULL, "\t");
        char *type = strtok(NULL, "\t");
        if (!id || !name || !type) { fail(out, "feature line with <3 columns: reference dereferences NULL"); goto done; }
        type[strcspn(type, "\n\r\t")] = '\0';                                      /* :309 */
        if (ht_put(ht_feature, line, feature_index)) {   /* key = buffer start, :313 */
            sb_printf(&s_feat, "%s\t%s\t%s\n", id, name, type);
            feature_index++;
        }
    }
    out->n_feature = (size_t)(feature_index - 1);

    /* ---- phase E: record loop, bam2db_ds.c:360-438 ---- */
    size_t cap = n_rec ? n_rec : 1;
    out->row_cell = (int32_t *)malloc(cap * sizeof(int32_t));
    out->row_feature = (int32_t *)malloc(cap * sizeof(int32_t));
    out->row_blob_len = (int16_t *)malloc(cap * sizeof(int16_t));
    out->row_blob = (uint8_t *)calloc(cap, ORC_MAX_BLOB);
    size_t n_rows = 0;
    for (size_t i = 0; i < n_rec; i++) {
        out->total_reads++;                                            /* E1 :363 */
        if (!(flags[i] & ORC_HAS_CB)) continue;                        /* E2 :366-371 */
        ht_ent *ce = ht_find(ht_cell, cb + i * cb_stride);             /* E3 :374-380 */
        if (!ce) continue;
        double r = oracle_genrand_real1();                             /* E4 :385 */
        if (r >= rate_depth) continue;                                 /* E5 :387-390 */
        out->sampled_reads++;                                          /* E6 :392 */
        if (!(flags[i] & ORC_HAS_XF)) { out->undefined_records++; continue; }   /* NULL deref in reference */
        if (!(xf[i] == 25 || xf[i] == 17)) continue;                   /* E7 :397-400 */
        if (!(flags[i] & ORC_HAS_GX)) { out->undefined_records++; continue; }   /* NULL deref in reference */
        ht_ent *fe = ht_find(ht_feature, gx + i * gx_stride);          /* E8 :403-410 */
        if (!fe) continue;
        if (!(flags[i] & ORC_HAS_UB)) continue;                        /* E9 :412-416 */
        size_t nb = 0;
        int er = oracle_encode_dna(ub + i * ub_stride, out->row_blob + n_rows * ORC_MAX_BLOB, ORC_MAX_BLOB, &nb); /* E10 */
        if (er == -2) { fail(out, "UMI longer than %d bases unsupported by oracle", ORC_MAX_BLOB * 4); goto done; }
        out->row_cell[n_rows] = (int32_t)ce->val;                      /* E11 :421-424 */
        out->row_feature[n_rows] = (int32_t)fe->val;
        out->row_blob_len[n_rows] = (er == -1) ? (int16_t)-1 : (int16_t)nb;
        n_rows++;
        out->sampled_valid_reads++;                                    /* E12 :435 */
    }
    out->n_rows = n_rows;

    /* ---- phase F: aggregate, bam2db_ds.c:480-483 (+ :539-542) ---- */
    order = (size_t *)malloc((n_rows ? n_rows : 1) * sizeof(size_t));
    for (size_t i = 0; i < n_rows; i++) order[i] = i;
    sort_ctx ctx = { out };
    qsort_r(order, n_rows, sizeof(size_t), row_cmp, &ctx);

    size_t rcap = n_rows ? n_rows : 1;
    out->mtx_feature = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->mtx_cell = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->mtx_count = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->umi_feature = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->umi_cell = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->umi_ncopy = (int32_t *)malloc(rcap * sizeof(int32_t));
    out->umi_text = (char *)calloc(rcap, 11);
    size_t nnz = 0, nu = 0;
    for (size_t i = 0; i < n_rows;) {
        size_t a = order[i];
        int32_t c = out->row_cell[a], f = out->row_feature[a];
        int distinct = 0;
        size_t j = i;
        while (j < n_rows && out->row_cell[order[j]] == c && out->row_feature[order[j]] == f) {
            size_t k = j;                       /* run of equal blobs */
            while (k < n_rows && row_cmp(&order[j], &order[k], &ctx) == 0) k++;
            size_t e = order[j];
            if (out->row_blob_len[e] >= 0) distinct++;          /* NULL not counted */
            out->umi_feature[nu] = f; out->umi_cell[nu] = c; out->umi_ncopy[nu] = (int32_t)(k - j);
            if (out->row_blob_len[e] < 0) strcpy(out->umi_text + nu * 11, "NULL");             /* :634-636 */
            else {
                /* table2gz decodes exactly 10 bases (:629); blobs shorter than 3 bytes are read
                 * out of bounds by the reference — the oracle's rows are zero-padded to 16 bytes */
                oracle_decode_dna(out->row_blob + e * ORC_MAX_BLOB, 10, out->umi_text + nu * 11);
            }
            nu++;
            j = k;
        }
        out->mtx_feature[nnz] = f; out->mtx_cell[nnz] = c; out->mtx_count[nnz] = distinct;
        nnz++;
        i = j;
    }
    out->nnz = nnz;
    out->n_umi_rows = nu;

    /* ---- phase G: writers, bam2db_ds.c:498-525, table2gz :575-650 ---- */
    /* first line: format "%%%MatrixMarket" renders as "%%MatrixMarket" with glibc */
    sb_printf(&s_mtx, "%s", "%%MatrixMarket matrix coordinate integer general\n");
    sb_printf(&s_mtx,
              "%%metadata_json: \n"
              "%%{\n"
              "%%\t\"software_version\": \"fastF-1.0.0\",\n"
              "%%\t\"format_version\": 1,\n"
              "%%\t\"parent_bam\": \"%s\",\n"
              "%%\t\"rate_cell\": %.3f,\n"
              "%%\t\"rate_depth\": %.3f,\n"
              "%%\t\"total_n_FastQ\": %zu,\n"
              "%%\t\"sampled_n_FastQ\": %zu,\n"
              "%%\t\"sampled_valid_n_FastQ\": %zu\n"
              "%%}\n",
              bam_label, rate_cell, rate_depth,
              (size_t)out->total_reads, (size_t)out->sampled_reads, (size_t)out->sampled_valid_reads);
    sb_printf(&s_mtx, "%zu %zu %zu\n", out->n_feature, out->n_barcode, out->nnz);   /* :513 */
    for (size_t i = 0; i < nnz; i++) {                                              /* :516 */
        sb_int(&s_mtx, out->mtx_feature[i]); sb_ch(&s_mtx, ' ');
        sb_int(&s_mtx, out->mtx_cell[i]);    sb_ch(&s_mtx, ' ');
        sb_int(&s_mtx, out->mtx_count[i]);   sb_ch(&s_mtx, '\n');
    }
    if (umi_copies) {                                                               /* :527-556 */
        for (size_t i = 0; i < nu; i++)
            sb_printf(&s_umi, "%d\t%d\t%s\t%d\n", out->umi_feature[i], out->umi_cell[i],
                      out->umi_text + i * 11, out->umi_ncopy[i]);
    }
    sb_ch(&s_mtx, '\0');  s_mtx.len--;
    sb_ch(&s_bar, '\0');  s_bar.len--;
    sb_ch(&s_feat, '\0'); s_feat.len--;
    sb_ch(&s_umi, '\0');  s_umi.len--;
    out->matrix_txt = s_mtx.p;   out->matrix_len = s_mtx.len;
    out->barcodes_txt = s_bar.p; out->barcodes_len = s_bar.len;
    out->features_txt = s_feat.p; out->features_len = s_feat.len;
    out->umi_txt = s_umi.p;      out->umi_len = s_umi.len;
    s_mtx.p = s_bar.p = s_feat.p = s_umi.p = NULL;
    rc = 0;
done:
    free(order);
    ht_free(ht_cell); ht_free(ht_feature);
    free(s_mtx.p); free(s_bar.p); free(s_feat.p); free(s_umi.p);
    return rc;
}

void oracle_result_free(oracle_result_t *r)
{
    free(r->mtx_feature); free(r->mtx_cell); free(r->mtx_count);
    free(r->umi_feature); free(r->umi_cell); free(r->umi_ncopy); free(r->umi_text);
    free(r->matrix_txt); free(r->barcodes_txt); free(r->features_txt); free(r->umi_txt);
    free(r->row_cell); free(r->row_feature); free(r->row_blob_len); free(r->row_blob);
    free(r->sampled_lines);
    memset(r, 0, sizeof *r);
}
