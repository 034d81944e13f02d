/*
 * fastf_oracle.h — CPU oracle for the fastF `bam2db` UMI-counting path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (yuw444/fastF, src/bam2db_ds.c + hashtable.c + mt19937ar.c +
 * utils.c).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it, and only as the checker.  Nothing under
 * fastf_amd/ links, imports or calls it.
 *
 * Pinning status: the reference's bam2db_ds.c needs
 * htslib, which this image does not have, so the reference pipeline itself is
 * unbuildable here.  The oracle is pinned by
 *   (1) oracle/_ref: the reference's own libc-only files (mt19937ar.c,
 *       utils.c, hashtable.c) compiled in place → MT stream, SampleInt and
 *       hash-table first-wins semantics compared call by call;
 *   (2) the known-answer outputs of the reference recorded in SURVEY.md §8c
 *       (tests/golden/survey_8c_*.json);
 *   (3) SQLite itself (python stdlib, same library the reference links)
 *       executing the reference's aggregate statement on the oracle's rows;
 *   (4) numpy RandomState (independent MT19937).
 */
#ifndef FASTF_ORACLE_H
#define FASTF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-record presence flags */
#define ORC_HAS_CB 1u
#define ORC_HAS_XF 2u
#define ORC_HAS_GX 4u
#define ORC_HAS_UB 8u

typedef struct oracle_result {
    /* counters printed in the MatrixMarket header (bam2db_ds.c:342-344,508-510) */
    uint64_t total_reads;
    uint64_t sampled_reads;
    uint64_t sampled_valid_reads;
    /* records where the reference would dereference NULL (bam2db_ds.c:394-395,403-404);
     * the oracle treats them as "skip" and counts them here */
    uint64_t undefined_records;
    /* dims line */
    size_t n_feature, n_barcode, nnz;
    /* COO rows in output order: ascending (cell, feature) */
    int32_t *mtx_feature, *mtx_cell, *mtx_count;
    /* -u rows (bam2db_ds.c:539-552): one per distinct (cell, feature, blob) */
    size_t n_umi_rows;
    int32_t *umi_feature, *umi_cell, *umi_ncopy;
    char *umi_text;              /* n_umi_rows x 11 bytes, "NULL" for NULL blobs */
    /* decompressed bytes of the output files */
    char *matrix_txt;   size_t matrix_len;
    char *barcodes_txt; size_t barcodes_len;
    char *features_txt; size_t features_len;
    char *umi_txt;      size_t umi_len;     /* only when umi_copies != 0 */
    /* the rows handed to the aggregate, in insertion order (for SQLite cross-check) */
    size_t n_rows;
    int32_t *row_cell, *row_feature;
    int16_t *row_blob_len;       /* -1 = NULL blob */
    uint8_t *row_blob;           /* n_rows x ORC_MAX_BLOB */
    /* sampled barcode line numbers (0-based), sorted — output of SampleInt+qsort */
    size_t n_sampled; uint64_t *sampled_lines;
    char err[256];
} oracle_result_t;

#define ORC_MAX_BLOB 16          /* UMIs up to 64 bases */

/* Restatement of bam2db() (bam2db_ds.c:106-573) on in-memory inputs.
 * barcodes/features: decompressed content of the two list files.
 * Records are SoA with fixed-stride NUL-terminated strings.
 * Returns 0 on success, 1 on failure (message in out->err). */
int oracle_bam2db(const char *barcodes, size_t barcodes_len,
                  const char *features, size_t features_len,
                  size_t n_rec, const uint8_t *flags, const int32_t *xf,
                  const char *cb, size_t cb_stride,
                  const char *gx, size_t gx_stride,
                  const char *ub, size_t ub_stride,
                  float rate_cell, float rate_depth, unsigned int seed,
                  const char *bam_label, int umi_copies,
                  oracle_result_t *out);
void oracle_result_free(oracle_result_t *r);

/* --- primitives, exported for known-answer pins ------------------------- */
void     oracle_init_genrand(uint32_t s);            /* mt19937ar.c:60-73   */
uint32_t oracle_genrand_int32(void);                 /* mt19937ar.c:105-140 */
double   oracle_genrand_real1(void);                 /* mt19937ar.c:149-153 */
void     oracle_mt_draws(uint32_t seed, uint64_t skip, uint64_t n, uint32_t *out, uint32_t *state);  /* loop over the two above */
/* SampleInt(iota(n_total), n_total, n_sample, 0, seed) then qsort; utils.c:29-75,
 * bam2db_ds.c:240-244.  out must hold n_sample entries (n_total if equal). */
int      oracle_sample_cells(size_t n_total, size_t n_sample, unsigned int seed, uint64_t *out);
size_t   oracle_n_cells_sampled(size_t n_cells, float rate_cell);   /* bam2db_ds.c:241 */
uint64_t oracle_djb2(const char *s, size_t len);     /* bam2db_ds.c:96-104  */
/* encode_DNA (bam2db_ds.c:22-51) + bound size rule (:419).  Returns 0 and fills
 * out/nbytes, or -1 for a NULL blob (non-ACGT). */
int      oracle_encode_dna(const char *seq, uint8_t *out, size_t out_cap, size_t *nbytes);
void     oracle_decode_dna(const uint8_t *blob, size_t n_bases, char *out); /* :53-93 */
/* the depth test of bam2db_ds.c:385-390 on one raw draw: 1 = record kept */
int      oracle_keep_draw(uint32_t draw, float rate_depth);

/* --- crb / extract, fastf_oracle_tags.c ------------------- */
/* extract_bam (extract.c:135-216): tag histogram as an insertion-order BST printed in pre-order,
 * "value,count\n".  type 0 = string tag, type 1 = sprintf("%d", bam_aux2i()).  *total_printed is the
 * doubled record count the reference prints (extract.c:163,165). */
int oracle_extract(size_t n, const uint8_t *present, const char *vals, size_t stride, const int64_t *ivals, int type,
                   char **csv_out, size_t *csv_len, uint64_t *total_printed, uint64_t *valid);
/* read_bam + print_CB_node (extract.c:64-133, 47-62): "CB;CR,count;CR,count;...\n" rows, both trees pre-order. */
int oracle_crb(size_t n, const uint8_t *has_cb, const uint8_t *has_cr, const char *cb, size_t cb_stride,
               const char *cr, size_t cr_stride, char **txt_out, size_t *txt_len, uint64_t *read_count, uint64_t *undefined);
void oracle_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
