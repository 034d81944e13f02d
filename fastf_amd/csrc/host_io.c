/*
 * host_io.c — host front end (lists, BAM decode, tag extraction, packing) and back end
 * (MatrixMarket / TSV gz writers) of bam2db.  Plain C; zlib is the only dependency.
 *
 * htslib is not available in this image and is not needed: a BAM file is a series of
 * BGZF blocks (gzip members with a "BC" extra field) whose payload is the record stream
 * the reference reads through sam_read1() (bam2db_ds.c:360).  The reader here inflates
 * blocks with zlib and walks the records and their aux fields directly.
 */
#define _GNU_SOURCE
#include "host_io.h"

#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static int io_err(const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    fastf_set_error_(buf);
    return 1;
}

/* ================================================================== */
/* lists                                                              */
/* ================================================================== */

/* whole (possibly gz) file into memory; gzopen reads plain files transparently, as the
 * reference relies on (bam2db_ds.c:153,165) */
static int slurp_gz(const char *path, char **out, size_t *len)
{
    gzFile f = gzopen(path, "r");
    if (!f) return io_err("cannot open %s", path);
    size_t cap = 1 << 20, n = 0;
    char *buf = (char *)malloc(cap);
    for (;;) {
        if (cap - n < (1 << 16)) { cap *= 2; buf = (char *)realloc(buf, cap); }
        int r = gzread(f, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (r < 0) { gzclose(f); free(buf); return io_err("read error on %s", path); }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(f);
    *out = buf; *len = n;
    return 0;
}

/* gzgets(…, 1024) over a memory image: at most 1023 chars, stops after '\n' */
static int next_line(const char *buf, size_t len, size_t *pos, char *line)
{
    if (*pos >= len) return 0;
    size_t n = 0;
    while (n < 1023 && *pos < len) {
        char c = buf[(*pos)++];
        line[n++] = c;
        if (c == '\n') break;
    }
    line[n] = '\0';
    return 1;
}

typedef struct { uint64_t *k; size_t cap, n; } keyset;
static int keyset_add(keyset *s, uint64_t key)      /* 1 = new, 0 = already there */
{
    if ((s->n + 1) * 2 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 1024;
        uint64_t *nk = (uint64_t *)calloc(nc, sizeof *nk);
        for (size_t i = 0; i < s->cap; i++)
            if (s->k[i]) { size_t j = (s->k[i] * 0x9E3779B97F4A7C15ull) >> 20 & (nc - 1); while (nk[j]) j = (j + 1) & (nc - 1); nk[j] = s->k[i]; }
        free(s->k); s->k = nk; s->cap = nc;
    }
    size_t j = (key * 0x9E3779B97F4A7C15ull) >> 20 & (s->cap - 1);
    while (s->k[j]) { if (s->k[j] == key) return 0; j = (j + 1) & (s->cap - 1); }
    s->k[j] = key; s->n++;
    return 1;
}

int fastf_lists_load_mem(const char *barcodes, size_t barcodes_len, const char *features, size_t features_len,
                         float rate_cell, unsigned int seed, fastf_lists_t *out)
{
    memset(out, 0, sizeof *out);
    char line[1024];
    keyset seen = {0};
    uint64_t *lines = NULL;
    int rc = 1;

    out->cell_dict = fastf_keydict_create();
    out->feat_dict = fastf_keydict_create();

    /* pass 1: count lines (bam2db_ds.c:233-236) */
    size_t pos = 0, n_lines = 0;
    while (next_line(barcodes, barcodes_len, &pos, line)) n_lines++;
    out->n_lines_barcodes = n_lines;

    lines = (uint64_t *)malloc((n_lines ? n_lines : 1) * sizeof *lines);
    size_t want = 0;
    if (fastf_sample_cells(n_lines, rate_cell, seed, lines, &want, &out->mt_skip)) {
        io_err("cell rate %g: sample size must lie in [0, number of barcodes]", (double)rate_cell);
        goto done;
    }
    out->n_sampled_target = want;
    out->barcode = (char **)calloc(want ? want : 1, sizeof(char *));
    out->cell_key = (uint64_t *)calloc(want ? want : 1, sizeof(uint64_t));

    /* pass 2: insert the sampled lines in file order (bam2db_ds.c:255-286).  A duplicate
     * barcode does not advance cell_index, so the cursor into the sampled line numbers
     * stalls and nothing after it is inserted — same as the reference. */
    size_t cell_index = 1, nth = 0;
    pos = 0;
    while (next_line(barcodes, barcodes_len, &pos, line) && cell_index <= want) {
        nth++;
        if (nth - 1 != lines[cell_index - 1]) continue;
        line[strcspn(line, "\n\r\t")] = '\0';
        uint64_t key = fastf_keydict_add(out->cell_dict, line, strlen(line));
        if (keyset_add(&seen, key)) {
            out->barcode[cell_index - 1] = strdup(line);
            out->cell_key[cell_index - 1] = key;
            cell_index++;
        } else {
            out->dup_barcodes++;
        }
    }
    out->n_cells = cell_index - 1;
    free(seen.k); memset(&seen, 0, sizeof seen);

    /* features (bam2db_ds.c:296-337): 3 tab-separated columns, key = start of the line
     * buffer as cut by strtok, first occurrence wins */
    size_t fcap = 1024;
    out->feat_id = (char **)malloc(fcap * sizeof(char *));
    out->feat_name = (char **)malloc(fcap * sizeof(char *));
    out->feat_type = (char **)malloc(fcap * sizeof(char *));
    out->feature_key = (uint64_t *)malloc(fcap * sizeof(uint64_t));
    pos = 0;
    while (next_line(features, features_len, &pos, line)) {
        char *save = NULL;
        char *id = strtok_r(line, "\t", &save);
        char *name = strtok_r(NULL, "\t", &save);
        char *type = strtok_r(NULL, "\t", &save);
        if (!id || !name || !type) {
            io_err("feature list: a line has fewer than 3 tab-separated columns (the reference crashes on it)");
            goto done;
        }
        type[strcspn(type, "\n\r\t")] = '\0';
        uint64_t key = fastf_keydict_add(out->feat_dict, line, strlen(line));
        if (!keyset_add(&seen, key)) { out->dup_features++; continue; }
        if (out->n_features == fcap) {
            fcap *= 2;
            out->feat_id = (char **)realloc(out->feat_id, fcap * sizeof(char *));
            out->feat_name = (char **)realloc(out->feat_name, fcap * sizeof(char *));
            out->feat_type = (char **)realloc(out->feat_type, fcap * sizeof(char *));
            out->feature_key = (uint64_t *)realloc(out->feature_key, fcap * sizeof(uint64_t));
        }
        size_t i = out->n_features++;
        out->feat_id[i] = strdup(id); out->feat_name[i] = strdup(name); out->feat_type[i] = strdup(type);
        out->feature_key[i] = key;
    }
    rc = 0;
done:
    free(seen.k);
    free(lines);
    if (rc) fastf_lists_free(out);
    return rc;
}

int fastf_lists_load(const char *barcodes_file, const char *features_file, float rate_cell, unsigned int seed,
                     fastf_lists_t *out)
{
    char *b = NULL, *f = NULL;
    size_t bl = 0, fl = 0;
    if (slurp_gz(barcodes_file, &b, &bl)) return 1;
    if (slurp_gz(features_file, &f, &fl)) { free(b); return 1; }
    int rc = fastf_lists_load_mem(b, bl, f, fl, rate_cell, seed, out);
    free(b); free(f);
    return rc;
}

void fastf_lists_free(fastf_lists_t *l)
{
    if (!l) return;
    for (size_t i = 0; i < l->n_cells && l->barcode; i++) free(l->barcode[i]);
    for (size_t i = 0; i < l->n_features; i++) {
        if (l->feat_id) free(l->feat_id[i]);
        if (l->feat_name) free(l->feat_name[i]);
        if (l->feat_type) free(l->feat_type[i]);
    }
    free(l->barcode); free(l->cell_key);
    free(l->feat_id); free(l->feat_name); free(l->feat_type); free(l->feature_key);
    fastf_keydict_destroy(l->cell_dict); fastf_keydict_destroy(l->feat_dict);
    memset(l, 0, sizeof *l);
}

/* ================================================================== */
/* BAM reader                                                         */
/* ================================================================== */
struct fastf_bam {
    FILE *fp;
    unsigned char *cbuf;        /* one compressed block (<= 64 KiB) */
    unsigned char *ubuf;        /* inflated stream window */
    size_t ulen, upos, ucap;
    int eof;
    uint64_t n_records, n_no_xf, n_no_gx;
    z_stream zs; int zs_init;
};

static inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
static inline uint32_t rd16(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }

/* inflate the next BGZF block and append it to the window; 0 ok, 1 EOF, -1 error */
static int bgzf_next(fastf_bam_t *b)
{
    unsigned char hdr[12];
    size_t r = fread(hdr, 1, 12, b->fp);
    if (r == 0) return 1;
    if (r != 12 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) { io_err("not a BGZF block"); return -1; }
    uint32_t xlen = rd16(hdr + 10);
    unsigned char extra[65536];
    if (fread(extra, 1, xlen, b->fp) != xlen) { io_err("truncated BGZF header"); return -1; }
    int bsize = -1;
    for (uint32_t i = 0; i + 4 <= xlen;) {
        uint32_t slen = rd16(extra + i + 2);
        if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = (int)rd16(extra + i + 4);
        i += 4 + slen;
    }
    if (bsize < 0) { io_err("BGZF block without BC field"); return -1; }
    long clen = (long)bsize + 1 - 12 - (long)xlen;            /* deflate data + crc32 + isize */
    if (clen < 8) { io_err("bad BGZF block size"); return -1; }
    if (fread(b->cbuf, 1, (size_t)clen, b->fp) != (size_t)clen) { io_err("truncated BGZF block"); return -1; }
    uint32_t isize = rd32(b->cbuf + clen - 4);
    /* compact the window, make room */
    if (b->upos) { memmove(b->ubuf, b->ubuf + b->upos, b->ulen - b->upos); b->ulen -= b->upos; b->upos = 0; }
    if (b->ulen + isize > b->ucap) { b->ucap = (b->ulen + isize) * 2; b->ubuf = (unsigned char *)realloc(b->ubuf, b->ucap); }
    if (isize) {
        z_stream *z = &b->zs;
        if (!b->zs_init) { memset(z, 0, sizeof *z); if (inflateInit2(z, -15) != Z_OK) { io_err("inflateInit2"); return -1; } b->zs_init = 1; }
        else inflateReset(z);
        z->next_in = b->cbuf; z->avail_in = (uInt)(clen - 8);
        z->next_out = b->ubuf + b->ulen; z->avail_out = isize;
        int zr = inflate(z, Z_FINISH);
        if (zr != Z_STREAM_END || z->avail_out != 0) { io_err("inflate failed (%d)", zr); return -1; }
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), b->ubuf + b->ulen, isize) != rd32(b->cbuf + clen - 8)) { io_err("BGZF CRC mismatch"); return -1; }
        b->ulen += isize;
    }
    return 0;
}

/* make at least `need` bytes available at upos; 0 ok, 1 clean EOF (no bytes), -1 error/truncated */
static int bam_need(fastf_bam_t *b, size_t need)
{
    while (b->ulen - b->upos < need) {
        if (b->eof) return (b->ulen == b->upos) ? 1 : -1;
        int r = bgzf_next(b);
        if (r < 0) return -1;
        if (r == 1) b->eof = 1;
    }
    return 0;
}

fastf_bam_t *fastf_bam_open(const char *path, int n_threads)
{
    (void)n_threads;
    FILE *fp = fopen(path, "rb");
    if (!fp) { io_err("Fail to open BAM file %s", path); return NULL; }
    fastf_bam_t *b = (fastf_bam_t *)calloc(1, sizeof *b);
    b->fp = fp;
    setvbuf(fp, NULL, _IOFBF, 1 << 22);
    b->cbuf = (unsigned char *)malloc(1 << 16);
    b->ucap = 1 << 20; b->ubuf = (unsigned char *)malloc(b->ucap);
    /* header: magic, l_text, text, n_ref, {l_name, name, l_ref}* — what sam_hdr_read() consumes (bam2db_ds.c:340) */
    if (bam_need(b, 12) || memcmp(b->ubuf + b->upos, "BAM\1", 4) != 0) { io_err("%s is not a BAM file", path); fastf_bam_close(b); return NULL; }
    uint32_t l_text = rd32(b->ubuf + b->upos + 4);
    if (bam_need(b, 12 + (size_t)l_text)) { io_err("truncated BAM header"); fastf_bam_close(b); return NULL; }
    uint32_t n_ref = rd32(b->ubuf + b->upos + 8 + l_text);
    b->upos += 12 + (size_t)l_text;
    for (uint32_t i = 0; i < n_ref; i++) {
        if (bam_need(b, 4)) { io_err("truncated BAM header"); fastf_bam_close(b); return NULL; }
        uint32_t l_name = rd32(b->ubuf + b->upos);
        if (bam_need(b, 8 + (size_t)l_name)) { io_err("truncated BAM header"); fastf_bam_close(b); return NULL; }
        b->upos += 8 + (size_t)l_name;
    }
    return b;
}

void fastf_bam_close(fastf_bam_t *b)
{
    if (!b) return;
    if (b->zs_init) inflateEnd(&b->zs);
    if (b->fp) fclose(b->fp);
    free(b->cbuf); free(b->ubuf); free(b);
}

void fastf_bam_stats(const fastf_bam_t *b, uint64_t *n_records, uint64_t *n_no_xf, uint64_t *n_no_gx)
{
    if (n_records) *n_records = b->n_records;
    if (n_no_xf) *n_no_xf = b->n_no_xf;
    if (n_no_gx) *n_no_gx = b->n_no_gx;
}

/* size of one aux value, or -1 when malformed; p points at the type byte */
static long aux_skip(const unsigned char *p, const unsigned char *end)
{
    if (p >= end) return -1;
    switch (*p) {
    case 'A': case 'c': case 'C': return 2;
    case 's': case 'S': return 3;
    case 'i': case 'I': case 'f': return 5;
    case 'd': return 9;
    case 'Z': case 'H': {
        const unsigned char *q = (const unsigned char *)memchr(p + 1, 0, (size_t)(end - p - 1));
        return q ? (long)(q - p) + 1 : -1;
    }
    case 'B': {
        if (end - p < 6) return -1;
        size_t esz;
        switch (p[1]) { case 'c': case 'C': esz = 1; break; case 's': case 'S': esz = 2; break;
                        case 'i': case 'I': case 'f': esz = 4; break; default: return -1; }
        uint64_t cnt = rd32(p + 2);
        uint64_t tot = 6 + cnt * esz;
        return tot <= (uint64_t)(end - p) ? (long)tot : -1;
    }
    default: return -1;
    }
}

/* bam_aux2i (htslib): integer types convert, anything else yields 0 */
static int64_t aux_int(const unsigned char *p)
{
    switch (*p) {
    case 'c': return (int8_t)p[1];
    case 'C': return p[1];
    case 's': return (int16_t)rd16(p + 1);
    case 'S': return rd16(p + 1);
    case 'i': return (int32_t)rd32(p + 1);
    case 'I': return rd32(p + 1);
    default: return 0;
    }
}

/* one record's aux block → packed fields; first occurrence of a tag wins (bam_aux_get) */
static void pack_record(fastf_bam_t *b, const unsigned char *aux, const unsigned char *end,
                        const fastf_keydict_t *cells, const fastf_keydict_t *feats,
                        uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta)
{
    const unsigned char *cb = NULL, *xf = NULL, *gx = NULL, *ub = NULL;
    while (end - aux >= 3) {
        const unsigned char *val = aux + 2;
        long sz = aux_skip(val, end);
        if (sz < 0) break;                         /* malformed tail: stop scanning, as a failed bam_aux_get */
        if (aux[0] == 'C' && aux[1] == 'B') { if (!cb) cb = val; }
        else if (aux[0] == 'x' && aux[1] == 'f') { if (!xf) xf = val; }
        else if (aux[0] == 'G' && aux[1] == 'X') { if (!gx) gx = val; }
        else if (aux[0] == 'U' && aux[1] == 'B') { if (!ub) ub = val; }
        aux = val + sz;
    }
    uint32_t m = 0;
    *cb_key = 0; *gx_key = 0; *umi = 0;
    /* bam_aux2Z returns NULL for a non-Z tag → hash_table_lookup(NULL) misses (hashtable.c:100) */
    if (cb && *cb == 'Z') {
        const char *s = (const char *)cb + 1;
        *cb_key = fastf_keydict_pack(cells, s, strlen(s));
    }
    if (xf) { int64_t q = aux_int(xf); if (q == 25 || q == 17) m |= FASTF_META_XF_OK; }
    else if (*cb_key) b->n_no_xf++;               /* reference would dereference NULL if this record is kept */
    if (gx && *gx == 'Z') {
        const char *s = (const char *)gx + 1;
        *gx_key = fastf_keydict_pack(feats, s, strlen(s));
    } else if (!gx && (m & FASTF_META_XF_OK) && *cb_key) b->n_no_gx++;
    if (ub) {
        if (*ub == 'Z') { const char *s = (const char *)ub + 1; m |= fastf_pack_umi(s, strlen(s), umi); }
        /* a non-Z UB makes bam_aux2Z return NULL and encode_DNA(NULL) crash in the reference: treated as absent */
    }
    *meta = m;
}

long fastf_bam_read_batch(fastf_bam_t *b, const fastf_keydict_t *cells, const fastf_keydict_t *feats,
                          uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta, size_t cap)
{
    size_t n = 0;
    while (n < cap) {
        int r = bam_need(b, 4);
        if (r == 1) break;                                     /* clean EOF */
        if (r < 0) { fprintf(stderr, "Warning: truncated BAM stream after %llu records\n", (unsigned long long)b->n_records); break; }
        uint32_t bs = rd32(b->ubuf + b->upos);
        if (bs < 32) { io_err("corrupt BAM record (block_size %u)", bs); return -1; }
        if (bam_need(b, 4 + (size_t)bs)) { fprintf(stderr, "Warning: truncated BAM record after %llu records\n", (unsigned long long)b->n_records); break; }
        const unsigned char *rec = b->ubuf + b->upos + 4;
        uint32_t l_read_name = rec[8], n_cigar = rd16(rec + 12), l_seq = rd32(rec + 16);
        uint64_t fixed = 32ull + l_read_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
        if (fixed > bs) { io_err("corrupt BAM record (fields exceed block_size)"); return -1; }
        pack_record(b, rec + fixed, rec + bs, cells, feats, cb_key + n, gx_key + n, umi + n, meta + n);
        b->upos += 4 + (size_t)bs;
        b->n_records++;
        n++;
    }
    return (long)n;
}

/* ================================================================== */
/* writers                                                            */
/* ================================================================== */
typedef struct { char *p; size_t len, cap; } obuf;
static void ob_room(obuf *o, size_t need)
{
    if (o->cap - o->len >= need) return;
    size_t nc = o->cap ? o->cap : (1 << 16);
    while (nc - o->len < need) nc *= 2;
    o->p = (char *)realloc(o->p, nc); o->cap = nc;
}
static void ob_str(obuf *o, const char *s) { size_t n = strlen(s); ob_room(o, n + 1); memcpy(o->p + o->len, s, n); o->len += n; }
static void ob_u32(obuf *o, uint32_t v)
{
    /* table2gz prints INTEGER columns with "%d" of sqlite3_column_int (bam2db_ds.c:619) */
    char t[12]; int n = 0; int32_t sv = (int32_t)v;
    uint32_t u = sv < 0 ? 0u - (uint32_t)sv : (uint32_t)sv;
    ob_room(o, 13);
    if (sv < 0) o->p[o->len++] = '-';
    do { t[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) o->p[o->len++] = t[--n];
}
static void ob_ch(obuf *o, char c) { ob_room(o, 1); o->p[o->len++] = c; }

int fastf_format_matrix(const char *bam_label, float rate_cell, float rate_depth, const uint64_t counters[3],
                        size_t n_feature, size_t n_barcode, const fastf_coo_t *coo, char **out, size_t *out_len)
{
    obuf o = {0};
    char hdr[2048];
    /* bam2db_ds.c:498-513.  The reference's first line is written with the format
     * "%%%MatrixMarket", which glibc renders as two percent signs followed by the word. */
    int n = snprintf(hdr, sizeof hdr,
                     "%%%%MatrixMarket matrix coordinate integer general\n"
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
                     "%%}\n"
                     "%zu %zu %zu\n",
                     bam_label, rate_cell, rate_depth, (size_t)counters[0], (size_t)counters[1], (size_t)counters[2],
                     n_feature, n_barcode, coo->nnz);
    if (n < 0 || (size_t)n >= sizeof hdr) return io_err("matrix header too long");
    ob_str(&o, hdr);
    for (size_t i = 0; i < coo->nnz; i++) {                 /* table2gz(db, "mtx", …, " ") :516 */
        ob_u32(&o, coo->feature[i]); ob_ch(&o, ' ');
        ob_u32(&o, coo->cell[i]);    ob_ch(&o, ' ');
        ob_u32(&o, coo->count[i]);   ob_ch(&o, '\n');
    }
    ob_room(&o, 1); o.p[o.len] = '\0';
    *out = o.p; *out_len = o.len;
    return 0;
}

int fastf_format_umi_rows(const fastf_umi_rows_t *rows, char **out, size_t *out_len)
{
    obuf o = {0};
    for (size_t i = 0; i < rows->n; i++) {                  /* numi table, "\t" delimiter :552 */
        ob_u32(&o, rows->feature[i]); ob_ch(&o, '\t');
        ob_u32(&o, rows->cell[i]);    ob_ch(&o, '\t');
        if (!rows->nonnull[i]) ob_str(&o, "NULL");          /* :634-636 */
        else {
            /* decode_DNA(blob, 10) (:629): always ten bases, whatever the blob length.  Blobs
             * shorter than 3 bytes are read out of bounds by the reference; here the missing
             * bits are zero ('A'). */
            char t[11];
            for (int k = 0; k < 10; k++) t[k] = "ACGT"[(rows->umi[i] >> (30 - 2 * k)) & 3];
            t[10] = '\0';
            ob_str(&o, t);
        }
        ob_ch(&o, '\t');
        ob_u32(&o, rows->n_copy[i]); ob_ch(&o, '\n');
    }
    ob_room(&o, 1); o.p[o.len] = '\0';
    *out = o.p; *out_len = o.len;
    return 0;
}

static int write_gz(const char *dir, const char *name, const char *buf, size_t len)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    gzFile f = gzopen(path, "wb");
    if (!f) { fprintf(stderr, "\x1b[31mError:\x1b[0m can not open file %s\n", path); return io_err("can not open file %s", path); }
    gzbuffer(f, 1 << 20);
    size_t off = 0;
    while (off < len) {
        unsigned chunk = (unsigned)(len - off > (1u << 30) ? (1u << 30) : len - off);
        if (gzwrite(f, buf + off, chunk) != (int)chunk) { gzclose(f); return io_err("write error on %s", path); }
        off += chunk;
    }
    if (gzclose(f) != Z_OK) return io_err("close error on %s", path);
    return 0;
}

int fastf_write_outputs(const char *path_out, const char *bam_label, float rate_cell, float rate_depth,
                        const uint64_t counters[3], const fastf_lists_t *lists, const fastf_coo_t *coo,
                        const fastf_umi_rows_t *umi_rows)
{
    char *mtx = NULL; size_t mlen = 0;
    if (fastf_format_matrix(bam_label, rate_cell, rate_depth, counters, lists->n_features, lists->n_cells, coo, &mtx, &mlen)) return 1;
    int rc = write_gz(path_out, "matrix.mtx.gz", mtx, mlen);
    free(mtx);
    if (rc) return 1;
    printf("matrix.mtx.gz is generated.\n");

    obuf o = {0};
    for (size_t i = 0; i < lists->n_cells; i++) { ob_str(&o, lists->barcode[i]); ob_ch(&o, '\n'); }   /* :520 */
    rc = write_gz(path_out, "barcodes.tsv.gz", o.p ? o.p : "", o.len);
    free(o.p);
    if (rc) return 1;
    printf("barcodes.tsv.gz is generated.\n");

    memset(&o, 0, sizeof o);
    for (size_t i = 0; i < lists->n_features; i++) {                                                     /* :524 */
        ob_str(&o, lists->feat_id[i]); ob_ch(&o, '\t');
        ob_str(&o, lists->feat_name[i]); ob_ch(&o, '\t');
        ob_str(&o, lists->feat_type[i]); ob_ch(&o, '\n');
    }
    rc = write_gz(path_out, "features.tsv.gz", o.p ? o.p : "", o.len);
    free(o.p);
    if (rc) return 1;
    printf("features.tsv.gz is generated.\n");

    if (umi_rows) {                                                                                      /* :527-556 */
        char *u = NULL; size_t ulen = 0;
        if (fastf_format_umi_rows(umi_rows, &u, &ulen)) return 1;
        rc = write_gz(path_out, "umi.tsv.gz", u ? u : "", ulen);
        free(u);
        if (rc) return 1;
        printf("umi.tsv.gz is generated.\n");
    }
    return 0;
}

/* ================================================================== */
/* bulk packing of string-level records (same rules as pack_record)    */
/* ================================================================== */
/* flags: 1 = CB present, 2 = xf present, 4 = GX present, 8 = UB present; strings are
 * fixed-stride and NUL-terminated.  Used by the synthetic generators and parity tests so
 * that the very strings the CPU oracle sees go through the product's packer. */
void fastf_pack_records(const fastf_keydict_t *cells, const fastf_keydict_t *feats, size_t n,
                        const uint8_t *flags, const int32_t *xf,
                        const char *cb, size_t cb_stride, const char *gx, size_t gx_stride,
                        const char *ub, size_t ub_stride,
                        uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta)
{
    for (size_t i = 0; i < n; i++) {
        uint32_t m = 0;
        cb_key[i] = 0; gx_key[i] = 0; umi[i] = 0;
        if (flags[i] & 1) { const char *s = cb + i * cb_stride; cb_key[i] = fastf_keydict_pack(cells, s, strnlen(s, cb_stride)); }
        if ((flags[i] & 2) && (xf[i] == 25 || xf[i] == 17)) m |= FASTF_META_XF_OK;
        if (flags[i] & 4) { const char *s = gx + i * gx_stride; gx_key[i] = fastf_keydict_pack(feats, s, strnlen(s, gx_stride)); }
        if (flags[i] & 8) { const char *s = ub + i * ub_stride; m |= fastf_pack_umi(s, strnlen(s, ub_stride), &umi[i]); }
        meta[i] = m;
    }
}
