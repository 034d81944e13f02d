/*
 * fastf_amd.h — C ABI of the MI355X-native bam2db UMI-counting engine.
 *
 * Three layers, all plain C (pointers + sizes, no C++/torch types):
 *
 *   1. OUTER drop-in symbols — what the reference's own main.c binds
 *        int bam2db(...)            replaces  src/bam2db_ds.c:106-573 (decl. bam2db_ds.h:62-70)
 *        int _umi_copies_flag       replaces  src/bam2db_ds.c:3       (decl. bam2db_ds.h:23)
 *        int cmd_bam2db(argc,argv)  replaces  src/main.c:288-362
 *      and, for the tag-histogram commands,
 *        extract_bam / read_bam / print_CB_node / free_CB_node   replace  src/extract.c (decl. extract.h:15-26)
 *        cmd_crb / cmd_extract                                   replace  src/main.c:231-286, 364-402
 *
 *   2. INNER seam (host buffers in, COO out) — what replaces the reference's
 *      per-record loop + SQLite aggregate (bam2db_ds.c:360-438, 480-483):
 *        fastf_engine_create / _push / _finish / _umi_rows / _reset / _destroy
 *      plus the host-side exact string→key packer that replaces
 *      hash()+hash_table_lookup() key handling (bam2db_ds.c:96-104, hashtable.c:98-115).
 *
 *   3. DEVICE-level entry points (device pointers + a hipStream_t passed as void*)
 *      for callers that already own HBM buffers (bench.py, the multi-GPU host in
 *      fastf_amd/dist.py, kernel parity tests).
 *
 * Every function returns 0 on success and non-zero on failure unless stated;
 * the message is available from fastf_last_error() — the last error of the process, whichever
 * thread raised it (reader workers fail on their own threads); the returned pointer is a
 * per-thread snapshot.
 * The reference convention "0 ok / 1 fail + message on stderr" (bam2db_ds.h:60)
 * is kept by the outer layer.
 */
#ifndef FASTF_AMD_H
#define FASTF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== */
/* 1. outer drop-in symbols                                              */
/* ===================================================================== */

/* bam2db_ds.h:23 / bam2db_ds.c:3 — set by `-u/--umicopies` (main.c:309) */
extern int _umi_copies_flag;

/* bam2db_ds.h:62-70.  db_file is accepted and ignored (no SQLite). */
int bam2db(char *bam_file, char *db_file, char *path_out, char *barcodes_file,
           char *features_file, float rate_cell, float rate_depth, unsigned int seed);

/* main.c:288-362; argv[0] == "bam2db". */
int cmd_bam2db(int argc, const char **argv);
/* which device(s) bam2db() uses, from the values of FASTF_DEVICES ("a,b,.." or a count) and FASTF_DEVICE (either may be NULL):
 * *dev0 = the engine's and the reader's device, *dev_second = the second listed device or -1 */
void fastf_pick_devices(const char *devices, const char *device, int *dev0, int *dev_second);

/* --- crb / extract: src/extract.c.  The tree types are the reference's (filter.h:28-34, extract.h:8-13); they are
 * declared here under guards so that this header can be included next to the reference's own. --- */
#ifndef FASTF_NO_TREE_TYPES
#if !defined(FASTQ_FILTER_H)
typedef struct node { char *data; long int count; struct node *left; struct node *right; } node;
#endif
#if !defined(EXTRACT_H)
typedef struct CB_node { node *CR; char *CB; struct CB_node *left; struct CB_node *right; } CB_node;
#endif
/* extract.c:135-216 — histogram of one tag (type 0: string, 1: integer) → ./tag_summary.csv, counters on stdout */
void extract_bam(char *bam_file, const char *tag, int type);
/* extract.c:64-133 — the CB → CR tree of the reference, node for node (malloc'ed; release with free_CB_node) */
CB_node *read_bam(char *bam_file);
/* extract.c:33-45.  print_CB_node(CB_node *, gzFile) (extract.c:47-62) is exported too; it is declared in
 * host_io.h / the reference's extract.h because its second parameter is zlib's gzFile. */
void free_CB_node(CB_node *root);
#endif
int cmd_crb(int argc, const char **argv);       /* main.c:231-286; argv[0] == "crb"     */
int cmd_extract(int argc, const char **argv);   /* main.c:364-402; argv[0] == "extract" */
/* the same results as bytes: the decompressed content of `crb`'s output file / of tag_summary.csv (malloc'ed) */
int fastf_crb_text(const char *bam_file, char **txt, size_t *txt_len, uint64_t *n_records);
int fastf_extract_text(const char *bam_file, const char *tag, int type, char **csv, size_t *csv_len,
                       uint64_t *n_records, uint64_t *n_valid);

const char *fastf_last_error(void);
const char *fastf_version(void);

/* ===================================================================== */
/* 2a. host-side primitives of the path                                   */
/* ===================================================================== */

/* MT19937 stream — replaces the global-state generator of mt19937ar.c:56-140
 * with an explicit state object; bit-identical output. */
typedef struct fastf_mt { uint32_t s[624]; int idx; } fastf_mt_t;
void     fastf_mt_seed(fastf_mt_t *mt, uint32_t seed);           /* mt19937ar.c:60-73   */
uint32_t fastf_mt_next(fastf_mt_t *mt);                          /* mt19937ar.c:105-140 */
void     fastf_mt_fill(fastf_mt_t *mt, uint32_t *out, size_t n); /* n consecutive draws */
void     fastf_mt_skip(fastf_mt_t *mt, uint64_t n);

/* Integer form of `genrand_real1() >= rate_depth → drop` (bam2db_ds.c:385-390):
 * a record is kept iff draw < fastf_draw_threshold(rate). */
uint64_t fastf_draw_threshold(float rate_depth);   /* 0 .. 2^32 */

/* Cell sub-sampling — bam2db_ds.c:240-244 + utils.c:29-75.
 * n_sampled = (size_t)(n_cells * rate_cell); writes the sorted 0-based line numbers.
 * *draws_used = MT draws consumed (0 when n_sampled == n_cells). Returns non-zero if
 * the reference would exit (n_sampled > n_cells) or the rate is negative/NaN. */
int fastf_sample_cells(size_t n_cells, float rate_cell, unsigned int seed,
                       uint64_t *lines_out, size_t *n_sampled, uint64_t *draws_used);

/* 2-bit UMI codec — bam2db_ds.c:5-51 (+ size rule :419).  Packs up to 16 bases
 * MSB-first into a left-aligned u32.  Returns meta bits (FASTF_META_*). */
uint32_t fastf_pack_umi(const char *ub, size_t len, uint32_t *umi_out);
/* The same for UMIs of up to 32 bases: bases 1..16 into *umi_out, bases 17..32 left-aligned into *ext_out (the reference
 * encodes a UMI of any length, bam2db_ds.c:417-419; an engine created with umi_max_bases 17..32 takes these) */
uint32_t fastf_pack_umi_long(const char *ub, size_t len, uint32_t *umi_out, uint32_t *ext_out);

#define FASTF_META_XF_OK      0x01u   /* xf present and in {25,17}   (bam2db_ds.c:397)  */
#define FASTF_META_HAS_UB     0x02u   /* UB tag present              (bam2db_ds.c:412)  */
#define FASTF_META_UMI_NONNULL 0x04u  /* every base in ACGT          (bam2db_ds.c:36-41)*/
#define FASTF_META_UMI_TOOLONG 0x08u  /* > 16 bases (fastf_pack_umi) / > 32 (_long): not representable, the engine errors */
#define FASTF_META_LEN_SHIFT  4       /* bits 4..7: blob byte length (len+3)/4, 0..8     */
#define FASTF_META_LEN_MASK   0xF0u

/* Exact string → 64-bit key packer.  Two strings get the same key iff they are
 * equal (for every string registered with _add and every string later passed to
 * _pack), which is what the reference's strcmp-verified table guarantees
 * (hashtable.c:98-115).  Key 0 means "cannot match any registered string". */
typedef struct fastf_keydict fastf_keydict_t;
fastf_keydict_t *fastf_keydict_create(void);
void     fastf_keydict_destroy(fastf_keydict_t *d);
uint64_t fastf_keydict_add(fastf_keydict_t *d, const char *s, size_t len);
uint64_t fastf_keydict_pack(const fastf_keydict_t *d, const char *s, size_t len);
/* pack, registering the string when it is new (thread-safe; the tag-histogram paths have no list up front) */
uint64_t fastf_keydict_intern(fastf_keydict_t *d, const char *s, size_t len);
/* key → string; returns the length or -1 (unknown key / buffer too small; cap counts the NUL) */
long     fastf_keydict_decode(const fastf_keydict_t *d, uint64_t key, char *buf, size_t cap);
/* fixed-stride NUL-terminated strings; present may be NULL (all present) */
void     fastf_keydict_pack_many(const fastf_keydict_t *d, const char *strs, size_t stride,
                                 size_t n, const uint8_t *present, uint64_t *out);

/* ===================================================================== */
/* 2b. inner seam: the engine                                             */
/* ===================================================================== */

typedef struct fastf_engine fastf_engine_t;

typedef struct fastf_engine_config {
    const uint64_t *cell_keys;     /* key of cell_index i+1 (sampled barcodes, file order) */
    uint32_t        n_cells;
    const uint64_t *feature_keys;  /* key of feature_index i+1                             */
    uint32_t        n_features;
    uint64_t        draw_threshold;/* fastf_draw_threshold(rate_depth), 0..2^32            */
    uint32_t        umi_max_bases; /* 1..32; widths of the packed sort key follow from it.  A key of more than 64 bits (many
                                      barcodes x many features x long UMIs; always from 17 bases on, which also need
                                      fastf_batch_t.umi_ext) runs on the single-device engine: it sorts the (cell, feature)
                                      word and carries the rest of the key beside it.  From 25 bases on the rest of the key
                                      is itself more than the 52 bits the reduce holds exactly: its top 17 bits (the UMI's
                                      first bases) join the sorted word, the matrix rows of one (cell, feature) are summed
                                      on the host */
    uint32_t        mt_seed;       /* engine-owned draw stream: init_genrand(seed) …       */
    uint64_t        mt_skip;       /* … advanced by the draws SampleInt consumed           */
    uint32_t        n_shards;      /* cell-hash shards (1 = single GPU)                    */
    uint32_t        shard_rank;    /* which shard this engine sorts/reduces                */
    int32_t         device;        /* HIP device ordinal                                   */
    uint64_t        batch_records; /* staging capacity (records per push), 0 = default     */
    uint64_t        key_capacity;  /* initial key-store capacity, 0 = default; grows       */
    /* One host process, several devices (SURVEY 8b inner seam / 8e): n_devices >= 2 makes this ONE engine drive that
     * many GPUs behind the same push / finish / umi_rows calls — chunks of the record stream are dealt to the devices,
     * keys travel to the device that owns their cell in a single exchange (RCCL send/recv group over xGMI; peer
     * copies when devices repeat), every device sorts and reduces its cells, the rows are merged by cell on the
     * host.  n_shards / shard_rank / device are then ignored.  0 or 1: the single-device engine.
     * devices: n_devices HIP ordinals, NULL = 0 .. n_devices-1; ordinals may repeat (all shards on one GPU: the
     * rehearsal the one-GPU tests run). */
    uint32_t        n_devices;
    const int32_t  *devices;
} fastf_engine_config_t;

/* One SoA batch of packed records (host memory; pinned is faster, any works). */
typedef struct fastf_batch {
    const uint64_t *cb_key;   /* 0 = CB absent / cannot match                 */
    const uint64_t *gx_key;   /* 0 = GX absent / cannot match                 */
    const uint32_t *umi;      /* fastf_pack_umi()                             */
    const uint32_t *meta;     /* FASTF_META_*                                 */
    size_t          n;
    const uint32_t *umi_ext;  /* NULL, or bases 17..32 of every UMI (fastf_pack_umi_long) for an engine with umi_max_bases > 16 */
} fastf_batch_t;

typedef struct fastf_coo {
    const uint32_t *feature;  /* 1-based feature_index */
    const uint32_t *cell;     /* 1-based cell_index    */
    const uint32_t *count;    /* COUNT(DISTINCT non-NULL umi) — may be 0 */
    size_t          nnz;      /* rows, ascending (cell, feature) */
} fastf_coo_t;

typedef struct fastf_umi_rows {     /* -u output, ascending (cell, feature, blob) */
    const uint32_t *feature, *cell, *n_copy;
    const uint32_t *umi;            /* left-aligned 2-bit bases; valid iff nonnull[i] */
    const uint8_t  *nonnull;
    size_t          n;
} fastf_umi_rows_t;

int  fastf_engine_create(const fastf_engine_config_t *cfg, fastf_engine_t **out);
void fastf_engine_destroy(fastf_engine_t *e);
/* H2D on the engine's copy stream, probe/filter/pack on its compute stream.  The
 * batch may be reused by the caller as soon as the call returns. */
int  fastf_engine_push(fastf_engine_t *e, const fastf_batch_t *batch);
/* Same, with caller-supplied draws (draws[i] belongs to the i-th CB hit of this batch). */
int  fastf_engine_push_draws(fastf_engine_t *e, const fastf_batch_t *batch,
                             const uint32_t *draws, size_t n_draws);
/* Zero-copy variant: the four arrays of the batch are PINNED host memory (fastf_pinned_alloc, or any memory passed
 * through fastf_pinned_register) and go to the device straight from where they are — no staging copy.  The call
 * returns once the copies are queued; the arrays must stay untouched until fastf_engine_wait_input() or
 * fastf_engine_finish() has returned.  This is the path SURVEY 8d calls "device-path": pinned SoA batches ->
 * hipMemcpyAsync on the copy stream -> kernels.  No push waits for the result of an earlier one: the hit-rank base
 * of a chunk (bam2db_ds.c:385 consumes one draw per CB hit) is carried on the device.
 * The arrays may also be DEVICE memory of the engine's device (single-device engines): records packed on the device by
 * the BAM front end (fastf_bam_read_batch_dev) are pushed this way. */
int  fastf_engine_push_pinned(fastf_engine_t *e, const fastf_batch_t *batch);
int  fastf_engine_wait_input(fastf_engine_t *e);   /* every host-to-device copy queued so far has completed */
void *fastf_pinned_alloc(size_t bytes);            /* hipHostMalloc; NULL on failure */
void  fastf_pinned_free(void *p);
/* pin memory the caller allocated (hipHostRegister).  The caller MUST unregister before the memory is freed, unmapped or its pages
 * dropped: the runtime keeps its entry for the addresses otherwise, and the next owner of those addresses inherits it (a later copy
 * into them faults on the GPU).  Prefer memory that is a mapping of its own over malloc heap memory.  The library keeps a ledger of
 * its registrations; FASTF_DEBUG_PINS=1 makes a violation abort and checks the pointers of the *_pinned / lend / gather entries. */
int   fastf_pinned_register(void *p, size_t bytes);
void  fastf_pinned_unregister(void *p);
int   fastf_debug_live_registrations(void);        /* registrations made through this library that are still live (tests) */
/* Sort + segmented unique/reduce over everything pushed; results stay valid until
 * reset/destroy.  counters = {total, sampled, sampled_valid} (bam2db_ds.c:342-344). */
int  fastf_engine_finish(fastf_engine_t *e, fastf_coo_t *coo, uint64_t counters[3]);
/* Optional, before fastf_engine_finish: PINNED host memory of the caller's (fastf_pinned_alloc / fastf_pinned_register)
 * for the matrix rows — three arrays of bytes / 12 entries.  When the matrix fits, finish() writes the rows there from
 * the device (no row buffer of its own to allocate and fault in) and the coo pointers point into it; the caller keeps
 * the memory alive and untouched until it is done with the rows.  NULL withdraws the loan; fastf_engine_reset ends it.
 * bam2db() lends one half of its decoder slab, idle once the last record is on the device. */
int  fastf_engine_lend_rows(fastf_engine_t *e, void *pinned, size_t bytes);
int  fastf_engine_umi_rows(fastf_engine_t *e, fastf_umi_rows_t *rows);
/* records each device of a multi-device engine (n_devices > 1) has been given since the last reset, records[n_devices];
 * a single-device engine reports its total in records[0] */
int  fastf_engine_device_records(const fastf_engine_t *e, uint64_t *records, uint32_t n);
int  fastf_engine_reset(fastf_engine_t *e);
/* key layout chosen at create time */
int  fastf_engine_key_bits(const fastf_engine_t *e, uint32_t *cell_bits, uint32_t *feature_bits,
                           uint32_t *umi_bits, uint32_t *total_bits);

/* ===================================================================== */
/* 2c. inner seam of crb / extract: the tag histogram                     */
/* ===================================================================== */
/* Replaces the per-record insert_tree / insert_CB_node loops (extract.c:88-108, 161-197; filter.c:105-124):
 * host buffers of exact 64-bit tag keys in (0 = tag absent), per distinct value its count and the index of the
 * record it first occurred on out — the three things the shape of the reference's insertion-order tree depends on. */
typedef struct fastf_taghist fastf_taghist_t;
typedef struct fastf_taghist_result {
    uint64_t n_records;              /* records pushed                                                     */
    uint64_t n_valid;                /* records with key1 != 0 (pair mode: and key2 != 0)                   */
    uint64_t n_key1_present;         /* records with key1 != 0                                              */
    /* level 1: distinct key1, ascending key value.  Pair mode: count1/first1 cover valid records only, and an
     * entry whose key1 never met a key2 has count1 == 0. */
    const uint64_t *key1, *count1, *first1; size_t n1;
    /* pair mode: distinct (key1, key2) of valid records, grouped by key1 in level-1 order */
    const uint32_t *pair_k1;         /* index into key1[]                                                   */
    const uint64_t *pair_key2, *pair_count, *pair_first; size_t n_pairs;
} fastf_taghist_result_t;
int  fastf_taghist_create(int device, fastf_taghist_t **out);
void fastf_taghist_destroy(fastf_taghist_t *h);
/* key2 == NULL: single-tag histogram; otherwise the two-level (key1 → key2) histogram.  One mode per object. */
int  fastf_taghist_push(fastf_taghist_t *h, const uint64_t *key1, const uint64_t *key2, size_t n);
/* result arrays are owned by the object and stay valid until the next finish / destroy */
int  fastf_taghist_finish(fastf_taghist_t *h, fastf_taghist_result_t *result);

/* ===================================================================== */
/* 3. device-level entry points (all pointers are device pointers)        */
/* ===================================================================== */

/* number of CB hits in a record range (multi-GPU: draw-rank base of each shard).  Runs K1a and leaves the cell
 * index of every record in the engine's scratch: a fastf_dev_probe_pack over the very same records, next on the same
 * stream, may pass FASTF_PROBE_REUSE_HITS to skip its own K1a.  The caller vouches for "same records": nothing is
 * cached by pointer. */
int fastf_dev_count_hits(fastf_engine_t *e, const uint64_t *d_cb_key, uint64_t n,
                         uint64_t *d_hits_out, void *stream);

/* K1: probe + filter + pack.  d_keys_out holds n_shards buffers of `shard_stride`
 * keys each; d_key_counts[n_shards] (u64) are appended to (not reset);
 * d_counters[4] = {hits, sampled, sampled_valid, error bits} are added to.
 * The i-th CB hit of this range uses d_draws[*d_draw_base + i] (d_draw_base may be NULL = 0):
 * a sharded run keeps the whole draw stream resident and passes its hit-rank base here. */
int fastf_dev_probe_pack(fastf_engine_t *e,
                         const uint64_t *d_cb_key, const uint64_t *d_gx_key,
                         const uint32_t *d_umi, const uint32_t *d_meta, uint64_t n,
                         const uint32_t *d_draws, uint64_t n_draws, const uint64_t *d_draw_base,
                         uint64_t *d_keys_out, uint64_t shard_stride,
                         uint64_t *d_key_counts, uint64_t *d_counters, uint32_t flags, void *stream);
#define FASTF_PROBE_REUSE_HITS 1u  /* fastf_dev_count_hits(e, d_cb_key, n, …) was the previous call on this stream */
/* Streaming form of K1b (single shard, gene list in LDS): every wave filters and packs on its own and writes into
 * private regions of d_keys_out (two per workgroup, filled by turns) — no barrier and no global atomic in the loop.  The
 * key buffer is then SEGMENTED (regions with gaps): d_key_counts[0] is SET to the number of keys, the buffer must hold
 * fastf_dev_probe_capacity() slots, and the next fastf_dev_sort over it must carry FASTF_SORT_SEGMENTED.  The kernel also
 * leaves, per region, the histogram of the first digit of the FASTF_SORT_SKIP_LOW sort: such a sort's first pass walks the
 * regions and counts nothing (scatter_regions_kernel); a sort on another digit grid counts per tile through the region map
 * the engine keeps.  The sorted result is contiguous as always.  One sort per probe_pack: the histograms are used up. */
#define FASTF_PROBE_SEGMENTED 2u
/* key slots a FASTF_PROBE_SEGMENTED call over n records needs in d_keys_out (a little more than n);
 * 0 = this engine cannot run the streaming form (several shards, or a gene list that does not fit LDS) */
int fastf_dev_probe_capacity(const fastf_engine_t *e, uint64_t n, uint64_t *key_slots);

/* BLOCKED record layout — the engine's own device staging.  Per 256-record unit one contiguous run
 *     gx u64[256] | umi u32[256] | meta u32[256] | cell scratch (u16[256], or u32[256] when n_cells > 65535)
 * of 4608 (5120) bytes; the cb keys stay an array of their own.  K1a writes a unit's scratch slice, the streaming K1b reads a
 * unit as ONE stream instead of four distant ones (the same 18 bytes per record move 18 % faster that way:
 * tools/hbm_probe_streams.hip).  fastf_batch_t is untouched: fastf_engine_push / _push_pinned land every chunk of a batch in
 * this layout themselves — three hipMemcpy2DAsync per chunk (a row = one unit's slice of gx / umi / meta, destination pitch =
 * the run) plus three plain copies for a last partial unit, umi_engine.hip push_chunk — and run the same K1a + streaming K1b
 * on it as the device-level calls below; engines whose gene list stays in L2, keys wider than 64 bits and sharded engines
 * stage SoA and run the tile form of K1b.  Device-resident SoA gets here through fastf_dev_block_records.
 * fastf_dev_block_bytes: size of the buffer for n records; 0 = this engine cannot run the streaming K1b (gene list not in
 * LDS, FASTF_NO_STREAM_K1B): use the SoA form. */
int fastf_dev_block_bytes(const fastf_engine_t *e, uint64_t n, uint64_t *bytes);
int fastf_dev_block_records(fastf_engine_t *e, const uint64_t *d_gx_key, const uint32_t *d_umi, const uint32_t *d_meta,
                            uint64_t n, void *d_blocked, void *stream);
/* fastf_dev_probe_pack: d_gx_key points at a blocked buffer of the n records (d_umi, d_meta are ignored).  The cell
 * scratch lives in that buffer, so with FASTF_PROBE_REUSE_HITS the preceding count must have been
 * fastf_dev_count_hits_blocked on the same buffer. */
#define FASTF_PROBE_BLOCKED 8u
int fastf_dev_count_hits_blocked(fastf_engine_t *e, const uint64_t *d_cb_key, uint64_t n, void *d_blocked,
                                 uint64_t *d_hits_out, void *stream);
/* The decision stream.  All K1b wants from the draw of a CB hit is one bit — the read is kept iff
 * genrand_real1() < rate (the reference drops on `>=`, bam2db_ds.c:385-390), i.e. draw < the engine's integer threshold
 * (fastf_draw_threshold) — so that is what it reads:
 * bit (i & 31) of word i >> 5 = draws[i] < threshold.  fastf_dev_probe_pack takes 32-bit draws and converts them on
 * every call (one extra pass over them); a caller that runs the same stream more than once converts it once with
 * fastf_dev_draw_bits (d_bits_out: 8-byte aligned, (n_draws + 63) / 64 * 8 bytes, the tail of the last 64 bits zero) and passes the result
 * as d_draws together with FASTF_PROBE_DRAW_BITS — n_draws and *d_draw_base keep counting draws. */
#define FASTF_PROBE_DRAW_BITS 16u
int fastf_dev_draw_bits(fastf_engine_t *e, const uint32_t *d_draws, uint64_t n_draws, uint32_t *d_bits_out, void *stream);
/* The decision stream from the DEVICE's generator: bit i of d_bits_out = draw i of init_genrand(seed) advanced by `skip` draws is
 * below the engine's threshold (mt19937ar.c:105-140 + bam2db_ds.c:385-390) — same layout and size rule as fastf_dev_draw_bits.
 * Large counts run on many workgroups at once: the stream's state is linear, sub-streams 624 x 256 draws apart are seated by
 * jump-ahead (mt_jump.c) and generated side by side.  Synchronises the stream. */
int fastf_dev_mt_decisions(fastf_engine_t *e, uint32_t seed, uint64_t skip, uint64_t n_draws, uint32_t *d_bits_out, void *stream);

/* Keys wider than 64 bits on a SHARDED engine (n_shards > 1; one process per GPU: fastf_amd/dist.py).  The calls above take
 * 64-bit keys; an engine whose keys are wider (fastf_engine_is_wide: many barcodes x many features x long UMIs, or
 * umi_max_bases > 16) runs the same E1-E12 chain (bam2db_ds.c:360-438) with the key in two words:
 *   fastf_dev_probe_pack_wide   as fastf_dev_probe_pack (tile form; flags FASTF_PROBE_REUSE_HITS / _DRAW_BITS), the group word of
 *                               every surviving record into d_keys_out[shard][..], the rest of its key into d_vals_out[shard][..]
 *                               (same slot), d_umi_ext = bases 17.. of the UMIs or NULL;
 *   [the caller exchanges both arrays: keys and values of shard s to rank s]
 *   fastf_dev_adopt_wide        the n pairs this shard owns into the engine's own key store (synchronises the stream);
 *                               fastf_engine_finish / fastf_engine_umi_rows then give this shard's rows (hashtable.c:70-115 and
 *                               bam2db_ds.c:417-419 take any list size and UMI length; the GROUP BY of :480-483 never joins cells). */
int fastf_engine_is_wide(const fastf_engine_t *e);
int fastf_dev_probe_pack_wide(fastf_engine_t *e, const uint64_t *d_cb_key, const uint64_t *d_gx_key, const uint32_t *d_umi,
                              const uint32_t *d_meta, const uint32_t *d_umi_ext, uint64_t n,
                              const uint32_t *d_draws, uint64_t n_draws, const uint64_t *d_draw_base,
                              uint64_t *d_keys_out, uint64_t *d_vals_out, uint64_t shard_stride, uint64_t *d_key_counts,
                              uint64_t *d_counters, uint32_t flags, void *stream);
int fastf_dev_adopt_wide(fastf_engine_t *e, const uint64_t *d_keys, const uint64_t *d_vals, uint64_t n, void *stream);

/* K2: LSD radix sort of the low `key_bits` bits of n keys (n read from *d_n on the
 * device, at most max_n).  d_keys and d_tmp are ping-pong buffers of max_n keys;
 * *sorted_in_tmp tells where the result landed. */
#define FASTF_SORT_SKIP_LOW   2u   /* leave the low fastf_engine_skip_bits() bits unsorted: enough for the matrix
                                      (equal keys stay neighbours of their (cell, feature, top-UMI-bits) run);
                                      pass the same flag to fastf_dev_reduce.  Not for fastf_dev_umi_rows.        */
#define FASTF_SORT_SEGMENTED  4u   /* d_keys is the output of the last FASTF_PROBE_SEGMENTED fastf_dev_probe_pack */
/* The keys of the next FASTF_SORT_SEGMENTED fastf_dev_sort lie in n_regions rows of `stride` slots of its d_keys, row r
 * holding d_counts[r] keys at its front (device array, u64): the receive buffer of a fixed-capacity key exchange is sorted
 * where it landed, no compaction pass.  Writes the total to *d_n_out on the device. */
int fastf_dev_set_regions(fastf_engine_t *e, const uint64_t *d_counts, uint32_t n_regions, uint64_t stride,
                          uint64_t *d_n_out, void *stream);
int fastf_engine_skip_bits(const fastf_engine_t *e, uint32_t *bits);
/* number of 8-bit LSD passes fastf_dev_sort runs for this engine's keys with the given flags: the digit grid starts at
 * the skip bit (not necessarily a byte boundary) when FASTF_SORT_SKIP_LOW is set, at bit 0 otherwise */
int fastf_engine_sort_passes(const fastf_engine_t *e, uint32_t flags, uint32_t *passes);
/* which lookup structure the lists qualified for: 1 = LDS-resident (barcodes: perfect hash of 32-bit codes; genes:
 * bitmap + rank + permutation over one id family; genes: 2 = direct index table over a dense id range),
 * 0 = open-addressed table in L2 */
int fastf_engine_table_modes(const fastf_engine_t *e, int *cells_in_lds, int *genes_in_lds);
/* width of one entry of the cell-index scratch K1a leaves for K1b: 2 bytes when every cell index fits 16 bits, else 4 */
int fastf_engine_cell_scratch_bytes(const fastf_engine_t *e, uint32_t *bytes);
int fastf_dev_sort(fastf_engine_t *e, uint64_t *d_keys, uint64_t *d_tmp,
                   const uint64_t *d_n, uint64_t max_n, uint32_t key_bits, uint32_t flags,
                   int *sorted_in_tmp, void *stream);

/* K3: segmented unique/reduce of sorted keys → COO (SoA, capacity max_n rows);
 * *d_nnz (u64) receives the row count. */
#define FASTF_REDUCE_SEGMENTED 8u   /* leave the rows in the engine's row regions (one per workgroup chunk, each in
                                       order, the regions in order) and only report *d_nnz: d_feature/d_cell/d_count are
                                       not touched; fastf_dev_rows_gather concatenates the regions later */
int fastf_dev_reduce(fastf_engine_t *e, const uint64_t *d_sorted, const uint64_t *d_n,
                     uint64_t max_n, uint32_t *d_feature, uint32_t *d_cell,
                     uint32_t *d_count, uint64_t *d_nnz, uint32_t flags, void *stream);
/* Concatenates the row regions of the last fastf_dev_reduce on this engine (same d_n) into feature/cell/count: device
 * arrays, or PINNED HOST memory (hipHostMalloc / fastf_pinned_alloc) — then this kernel is the device-to-host copy of
 * the matrix rows.  A fastf_dev_reduce without FASTF_REDUCE_SEGMENTED runs it itself. */
int fastf_dev_rows_gather(fastf_engine_t *e, const uint64_t *d_n, uint32_t *feature, uint32_t *cell,
                          uint32_t *count, void *stream);

/* K3u: run-length rows for -u (capacity max_n rows). */
int fastf_dev_umi_rows(fastf_engine_t *e, const uint64_t *d_sorted, const uint64_t *d_n,
                       uint64_t max_n, uint64_t *d_ukeys, uint32_t *d_ncopy,
                       uint64_t *d_nrows, void *stream);

/* bytes of device workspace the engine holds for sorts of up to max_n keys
 * (grown on demand by the calls above; call once up front to avoid growth in a
 * timed region) */
int fastf_dev_reserve(fastf_engine_t *e, uint64_t max_records, uint64_t max_keys);

/* Error bits raised by device kernels since the last clear (synchronises).  Bit 16 (FASTF_ERR_RUN_TOO_LONG) after
 * a FASTF_SORT_SKIP_LOW reduce means the result of that reduce is not valid: sort fully and reduce again
 * (fastf_engine_finish does this by itself). */
#define FASTF_ERR_RUN_TOO_LONG 16u
int fastf_dev_error_bits(fastf_engine_t *e, uint64_t *bits);
/* stream-ordered (an atomicAnd on the device; no synchronisation) */
int fastf_dev_clear_error_bits(fastf_engine_t *e, uint64_t mask, void *stream);

/* name of the dominant kernel symbols, for profile post-processing */
const char *fastf_kernel_names(void);

#ifdef __cplusplus
}
#endif
#endif /* FASTF_AMD_H */
