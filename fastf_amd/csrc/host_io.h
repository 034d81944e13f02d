/* host_io.h — internal host-side front/back end of bam2db (lists, BAM reader, writers).
 * These symbols are exported from libfastf_amd.so for the tests and the CLI, but the
 * stable ABI is include/fastf_amd.h. */
#ifndef FASTF_HOST_IO_H
#define FASTF_HOST_IO_H

#include "fastf_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

void fastf_set_error_(const char *msg);
/* deflate_fast.c: text -> one complete gzip member; 0 when cap (>= fastf_gz_bound(len)) was too small */
size_t fastf_gz_bound(size_t len);
size_t fastf_gz_member_fast(const unsigned char *in, size_t len, unsigned char *out, size_t cap);

/* ---- barcode / feature lists: bam2db_ds.c:229-337 ---- */
typedef struct fastf_lists {
    size_t    n_lines_barcodes;    /* "Total number of cells"                    (:233-237) */
    size_t    n_sampled_target;    /* (size_t)(n_cells * rate_cell)              (:241)     */
    size_t    n_cells;             /* rows of the cell table = barcodes inserted (:268-280) */
    char    **barcode;             /* [n_cells] text, file order                              */
    uint64_t *cell_key;            /* [n_cells]                                               */
    size_t    n_features;          /* rows of the feature table                  (:313-327) */
    char    **feat_id, **feat_name, **feat_type;
    uint64_t *feature_key;
    fastf_keydict_t *cell_dict, *feat_dict;
    uint64_t  mt_skip;             /* MT draws SampleInt consumed after the re-seed (utils.c:32,55) */
    size_t    dup_barcodes, dup_features;
} fastf_lists_t;

int  fastf_lists_load(const char *barcodes_file, const char *features_file,
                      float rate_cell, unsigned int seed, fastf_lists_t *out);
int  fastf_lists_load_mem(const char *barcodes, size_t barcodes_len,
                          const char *features, size_t features_len,
                          float rate_cell, unsigned int seed, fastf_lists_t *out);
void fastf_lists_free(fastf_lists_t *l);

/* ---- BAM front end: replaces sam_open/sam_hdr_read/sam_read1/bam_aux_get/bam_aux2Z/
 *      bam_aux2i (bam2db_ds.c:141,340,360-417) for BGZF-compressed BAM ---- */
typedef struct fastf_bam fastf_bam_t;
/* gpu_frontend.hpp (HIP side of the library): BGZF inflate of a reader window on the device.  `out` is pinned host
 * memory; status[i] != 0 = the device declined block i (the host inflates it).  FASTF_GPU_INFLATE=1 turns it on. */
typedef struct fastf_gpuinf fastf_gpuinf_t;
typedef struct { uint64_t coff; uint32_t clen, isize; uint64_t uoff; } fastf_gpuinf_blk_t;
fastf_gpuinf_t *fastf_gpuinf_create(int device);
void fastf_gpuinf_destroy(fastf_gpuinf_t *g);
int  fastf_gpuinf_submit(fastf_gpuinf_t *g, const unsigned char *comp, const fastf_gpuinf_blk_t *blk, size_t n, unsigned char *out);
int  fastf_gpuinf_wait(fastf_gpuinf_t *g, uint8_t *status, double *device_ms);
int  fastf_gpuinf_run(fastf_gpuinf_t *g, const unsigned char *comp, const fastf_gpuinf_blk_t *blk, size_t n,
                      unsigned char *out, uint8_t *status);
/* allocate ahead what keep-mode windows of this size need (window buffers, slice staging, parse buffers) */
int  fastf_gpuinf_reserve(fastf_gpuinf_t *g, size_t window_bytes, size_t comp_bytes, size_t n_blocks);
void fastf_gpuinf_stats(const fastf_gpuinf_t *g, uint64_t *n_blocks, uint64_t *n_declined);
/* keep mode + second stage (gpu_records.hpp): the inflated bytes stay on the device (window buffer per parity, block i at its
 * host offset), CRC-32 per block on the device (status bit 1), then record hop + tag extraction + key packing there */
int  fastf_gpuinf_submit_keep(fastf_gpuinf_t *g, const unsigned char *comp, const fastf_gpuinf_blk_t *blk, size_t n,
                              int parity, const uint32_t *crc);
/* what a key dictionary needs on the device: its ID-form prefixes.  fastf_keydict_export returns 1 (and leaves the view
 * unusable) when the dictionary holds escape strings, more than 8 prefixes or a prefix longer than 32 bytes */
typedef struct {
    uint32_t n_prefix; uint32_t prefix_len[8]; uint64_t prefix_id[8]; unsigned char prefix[8][32];
} fastf_keydict_view_t;
int  fastf_keydict_export(const fastf_keydict_t *d, fastf_keydict_view_t *v);
typedef struct {
    int status;                    /* 0 ok; 1: the segment chains did not line up — parse this window on the host */
    uint64_t n, handover, no_xf, no_gx;
    fastf_batch_t batch;           /* DEVICE pointers: n packed records */
} fastf_gpurec_result_t;
int  fastf_gpurec_parse(fastf_gpuinf_t *g, int parity, const unsigned char *tail, size_t tail_len, uint64_t data_off,
                        uint64_t end, uint32_t n_ref, const fastf_keydict_view_t *cells, const fastf_keydict_view_t *feats,
                        fastf_gpurec_result_t *out);
int  fastf_gpurec_fetch(fastf_gpuinf_t *g, int parity, unsigned char *dst, uint64_t from, uint64_t to);
void fastf_gpurec_stats(const fastf_gpuinf_t *g, uint64_t *windows, uint64_t *fallbacks);
uint64_t fastf_gpurec_repairs(const fastf_gpuinf_t *g);   /* hop segments whose guessed chain did not meet the one before and were walked again on the device */
void *fastf_pinned_alloc(size_t bytes);
void fastf_pinned_free(void *p);
int  fastf_pinned_register(void *p, size_t bytes);
void fastf_pinned_unregister(void *p);
void fastf_bam_print_profile(const fastf_bam_t *b);   /* FASTF_BAM_PROFILE=1: stage times of the reader on stderr */
fastf_bam_t *fastf_bam_open(const char *path, int n_threads);
/* gpu_inflate: what to do when FASTF_GPU_INFLATE is unset — 0/-1 host threads only, 1 inflate shared with the device, 2 wait for
 * the device; | 4: the caller will call fastf_bam_enable_device_parse (nothing is pinned for copy-back windows up front);
 * | (ordinal + 1) << 8: the device the reader's device side runs on (0 in those bits: FASTF_DEVICE, else device 0);
 * | (ordinal + 1) << 16: a SECOND device — the reader then keeps an inflate/parse context on each and deals its windows to them
 *   by turns (two windows in flight; 0 in those bits: FASTF_BAM_DEVICE2, else one device) */
fastf_bam_t *fastf_bam_open2(const char *path, int n_threads, int gpu_inflate);
/* Decodes up to cap records into packed SoA; returns the count, 0 at EOF, -1 on error. */
long fastf_bam_read_batch(fastf_bam_t *b, const fastf_keydict_t *cells, const fastf_keydict_t *feats,
                          uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta, size_t cap);
/* device-side parse of the windows the device inflates (gpu_records.hpp): 0 = on, 1 = this reader keeps the host parser */
int  fastf_bam_enable_device_parse(fastf_bam_t *b, const fastf_keydict_t *cells, const fastf_keydict_t *feats);
/* as fastf_bam_read_batch; a batch either fills the host arrays (*on_device = 0) or lies packed in device memory already
 * (*on_device = 1, dev: valid until the second next call) */
/* the next fastf_bam_read_batch / _dev call packs UMIs of up to 32 bases: bases 1..16 into umi[i] as always, bases 17..32 into
 * umi_ext[i] (same index).  NULL (the default): fastf_pack_umi, 16 bases at most.  Host-packed batches only. */
void fastf_bam_set_umi_ext(fastf_bam_t *b, uint32_t *umi_ext);
long fastf_bam_read_batch_dev(fastf_bam_t *b, const fastf_keydict_t *cells, const fastf_keydict_t *feats,
                              uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta, size_t cap,
                              int *on_device, fastf_batch_t *dev);
/* Tag reader of the histogram paths (crb / extract): per record the key of tag1 (and tag2, or NULL) — type 0 string
 * (interned in dict), type 1 integer; 0 = absent.  Returns the count, 0 at EOF, -1 on error; *n_undefined is
 * incremented for string-mode tags that are not of type Z (NULL dereference in the reference, extract.c:102-103,189). */
long fastf_bam_read_tags(fastf_bam_t *b, fastf_keydict_t *dict, const char *tag1, const char *tag2, int type,
                         uint64_t *key1, uint64_t *key2, size_t cap, uint64_t *n_undefined);
/* counts of records whose reference behaviour is undefined (xf / GX missing where the
 * reference dereferences NULL, bam2db_ds.c:394-395,403-404); they are skipped here */
void fastf_bam_stats(const fastf_bam_t *b, uint64_t *n_records, uint64_t *n_no_xf, uint64_t *n_no_gx);
void fastf_bam_close(fastf_bam_t *b);

/* ---- writers: bam2db_ds.c:453-556, table2gz :575-650 ---- */
int fastf_write_outputs(const char *path_out, const char *bam_label, float rate_cell, float rate_depth,
                        const uint64_t counters[3], const fastf_lists_t *lists,
                        const fastf_coo_t *coo, const fastf_umi_rows_t *umi_rows /* NULL unless -u */);
/* the decompressed bytes of matrix.mtx (header + rows) into a malloc'ed buffer */
int fastf_format_matrix(const char *bam_label, float rate_cell, float rate_depth, const uint64_t counters[3],
                        size_t n_feature, size_t n_barcode, const fastf_coo_t *coo, char **out, size_t *out_len);
int fastf_format_umi_rows(const fastf_umi_rows_t *rows, char **out, size_t *out_len);

/* extract.c:47-62 (declared here because of zlib's gzFile) */
#include <zlib.h>
void print_CB_node(CB_node *root, gzFile fp);

/* pre-order of the insertion-order BST of m distinct strings with first-occurrence positions first[] (tag_cmds.c) */
void fastf_tag_tree_preorder(const char *const *strs, const uint64_t *first, uint32_t m, uint32_t *order);

/* text → gzip file (chunk-parallel members) */
int fastf_write_gz_text(const char *path, const char *text, size_t len);

/* string-level records → packed SoA (flags: 1 CB, 2 xf, 4 GX, 8 UB present) */
void fastf_pack_records_ext(const fastf_keydict_t *cells, const fastf_keydict_t *feats, size_t n,
                            const uint8_t *flags, const int32_t *xf,
                            const char *cb, size_t cb_stride, const char *gx, size_t gx_stride,
                            const char *ub, size_t ub_stride,
                            uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta, uint32_t *umi_ext);
void fastf_pack_records(const fastf_keydict_t *cells, const fastf_keydict_t *feats, size_t n,
                        const uint8_t *flags, const int32_t *xf,
                        const char *cb, size_t cb_stride, const char *gx, size_t gx_stride,
                        const char *ub, size_t ub_stride,
                        uint64_t *cb_key, uint64_t *gx_key, uint32_t *umi, uint32_t *meta);

#ifdef __cplusplus
}
#endif
/* host buffers the GPU may get to see, and big short-lived ones: private mappings (never malloc heap), page granular, 2 MiB aligned
 * with transparent huge pages from 2 MiB on; release with fastf_big_free() only (host_io.c) */
void fastf_par_run(int n_threads, void (*fn)(void *, int), void *arg);   /* fn(arg, worker index) on n_threads threads, joined */
int  fastf_host_thread_count(void);                                      /* FASTF_HOST_THREADS, else the cores (at most 16) */
/* ledger of live hipHostRegister ranges (host_io.c): every release of a big buffer is checked against it */
void  fastf_pin_ledger_add(const void *p, size_t bytes);
int   fastf_pin_ledger_remove(const void *p);     /* 1 if the ledger held it */
int   fastf_pin_ledger_live(void);
void *fastf_big_alloc(size_t bytes);
void  fastf_big_drop(void *p, size_t bytes);      /* pages back, range stays allocated */
void  fastf_big_free(void *p, size_t bytes);      /* pages back slice by slice (MADV_DONTNEED), then munmap; heap memory passed here is free()d */

#endif
