/*
 * bam2db_main.c — the outer drop-in layer: bam2db(), _umi_copies_flag, cmd_bam2db().
 *
 *   bam2db()      same signature/return convention as the reference (bam2db_ds.h:62-70,
 *                 bam2db_ds.c:106-573); db_file is accepted and ignored.
 *   cmd_bam2db()  same flags as main.c:288-362 (-b -f -a -d -c -r -o -s -u), values as
 *                 `-x v`, `-xv`, `--long v`, `--long=v`; floats via strtof, ints via
 *                 strtol(s, &e, 0) like argparse.c:88-108.
 * The device engine does the work; there is no CPU fallback.
 */
#define _GNU_SOURCE
#include "host_io.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>

int _umi_copies_flag = 0;
int fastf_process_is_exiting_ = 0;     /* set by fastf_cli.c only: bam2db() may leave its resources to the exiting process */

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

/* decoder thread: BAM → packed SoA batches, double-buffered, while the caller pushes the previous batch
 * (and, for the first batch, while the HIP runtime and the engine come up) */
typedef struct { uint64_t *cb, *gx; uint32_t *umi, *meta, *ext; long n; int on_device; fastf_batch_t dev; } dec_slot;   /* on_device: the batch lies packed in device memory (dev); ext: bases 17.. of the UMIs (runs with UMIs beyond 16 bases only) */
#define DEC_SLOTS 6                      /* ring of decoder slots; only those the decoder reaches are ever touched or pinned */
typedef struct {
    fastf_bam_t *bam; const fastf_lists_t *lists; size_t cap;
    dec_slot slot[DEC_SLOTS]; int filled[DEC_SLOTS]; int n_slots;
    /* pinning (the pin thread): begun[k] = the decoder has started on slot k for the first time; pinned[k] = 1 registered
     * with the HIP runtime, -1 registration failed or switched off (that slot is pushed through the engine's staging) */
    int begun[DEC_SLOTS], pinned[DEC_SLOTS], pin_stop;
    unsigned char *slab; size_t slot_bytes;
    pthread_mutex_t mu; pthread_cond_t cv;
    int stop; double t_decode;
    int engine_up;                       /* set by the pusher once it can take batches */
    /* phase accounting (FASTF_PROFILE): records decoded so far, how many of them before the decoder first saw the engine up,
     * and when that was / when the last record was decoded (seconds since bam2db() was entered) */
    uint64_t n_decoded, n_before_engine; double t_saw_engine, t_last_record;
    int trace; double t_origin;          /* FASTF_PROFILE=2: one line per batch, times since bam2db() was entered */
} dec_ctx;

/* hand slot k to the pusher */
static void dec_deliver(dec_ctx *d, int k, long n, double t_start)
{
    if (d->trace) fprintf(stderr, "[trace] %.3f decoder: slot %d filled, %ld records%s, %.3f s\n", now_s() - d->t_origin, k, n,
                          d->slot[k].on_device ? " (packed on the device)" : "", now_s() - t_start);
    pthread_mutex_lock(&d->mu);
    d->slot[k].n = n; d->filled[k] = 1;
    pthread_cond_broadcast(&d->cv);
    pthread_mutex_unlock(&d->mu);
}
/* wait until the pusher has given slot k back and at most `ahead` other slots are still waiting for it; 0 when told to stop */
static int dec_wait_free(dec_ctx *d, int k, int ahead)
{
    pthread_mutex_lock(&d->mu);
    for (;;) {
        int out = 0;
        for (int i = 0; i < d->n_slots; i++) out += d->filled[i];
        if ((!d->filled[k] && out <= ahead) || d->stop) break;
        pthread_cond_wait(&d->cv, &d->mu);
    }
    const int stop = d->stop;
    if (!stop && !d->begun[k]) { d->begun[k] = 1; pthread_cond_broadcast(&d->cv); }     /* the pin thread may take it now */
    pthread_mutex_unlock(&d->mu);
    if (!stop && d->trace) fprintf(stderr, "[trace] %.3f decoder: slot %d free\n", now_s() - d->t_origin, k);
    return !stop;
}

/* pin thread: registers the slots with the HIP runtime in the order the decoder reaches them (23 ms per 100 MB slot —
 * beside the engine's start-up and the pushes, not in front of the first push) */
static void *pin_main(void *vp)
{
    dec_ctx *d = (dec_ctx *)vp;
    for (int k = 0; k < d->n_slots; k++) {
        pthread_mutex_lock(&d->mu);
        while (!d->begun[k] && !d->pin_stop) pthread_cond_wait(&d->cv, &d->mu);
        const int stop = d->pin_stop;
        pthread_mutex_unlock(&d->mu);
        if (stop) break;
        const int ok = fastf_pinned_register(d->slab + (size_t)k * d->slot_bytes, d->slot_bytes) == 0;
        if (d->trace) fprintf(stderr, "[trace] %.3f pin thread: slot %d %s\n", now_s() - d->t_origin, k, ok ? "registered" : "not registered");
        pthread_mutex_lock(&d->mu);
        d->pinned[k] = ok ? 1 : -1;
        pthread_cond_broadcast(&d->cv);
        pthread_mutex_unlock(&d->mu);
    }
    return NULL;
}

static void *decoder_main(void *vp)
{
    dec_ctx *d = (dec_ctx *)vp;
    /* The reader hands out one window's records per call.  Until the engine is up nobody takes a slot, so the slot being
     * filled goes on taking windows, and the decoder walks on through the ring (DEC_SLOTS slots of `cap` records are what
     * the host can decode ahead while the HIP runtime starts: a skinny BAM runs at 80 M records/s on the host alone).
     * That ends with the first batch the device packed: its arrays stay valid for one more window of that parity only, so
     * from then on every call is handed over at once and the decoder stays one slot ahead of the pusher, no more — by then
     * the engine is up or about to be. */
    int accumulate = 1;
    for (int k = 0;; k = (k + 1) % d->n_slots) {
        if (!dec_wait_free(d, k, accumulate ? d->n_slots : 1)) break;
        double t = now_s();
        dec_slot *sl = &d->slot[k];
        size_t fill = 0;
        long n;
        int on_dev = 0;
        fastf_batch_t dev; memset(&dev, 0, sizeof dev);
        for (;;) {
            if (sl->ext) fastf_bam_set_umi_ext(d->bam, sl->ext + fill);
            n = fastf_bam_read_batch_dev(d->bam, d->lists->cell_dict, d->lists->feat_dict, sl->cb + fill, sl->gx + fill,
                                         sl->umi + fill, sl->meta + fill, d->cap - fill, &on_dev, &dev);
            if (n <= 0 || on_dev) break;
            fill += (size_t)n;
            if (!accumulate || __atomic_load_n(&d->engine_up, __ATOMIC_ACQUIRE) || d->cap - fill < ((size_t)1 << 16)) break;
        }
        d->t_decode += now_s() - t;
        {   const uint64_t got = (uint64_t)fill + (n > 0 && on_dev ? (uint64_t)n : 0);
            if (d->t_saw_engine == 0 && __atomic_load_n(&d->engine_up, __ATOMIC_ACQUIRE)) {
                d->t_saw_engine = now_s() - d->t_origin; d->n_before_engine = d->n_decoded + got;   /* (this call's records were under way) */
            }
            d->n_decoded += got;
            if (got) d->t_last_record = now_s() - d->t_origin; }
        if (on_dev) accumulate = 0;
        if (fill) {
            sl->on_device = 0;
            dec_deliver(d, k, (long)fill, t);
            if (n > 0 && !on_dev) continue;
            /* the call that ended the filling brought something else: the end of the input, an error, or a batch that lies on
             * the device — it goes into the next slot, behind the records above */
            k = (k + 1) % d->n_slots;
            if (!dec_wait_free(d, k, accumulate ? d->n_slots : 1)) break;
            sl = &d->slot[k];
            t = now_s();
        }
        sl->on_device = on_dev; sl->dev = dev;
        dec_deliver(d, k, n, t);
        if (n <= 0) break;
    }
    return NULL;
}

/* release thread: once the last record has been pushed, the reader (pinned window buffers, the pinned staging of the
 * compressed bytes, the BAM mapping, ~1 GB of touched pages) and the pinned slab are of no use any more.  Unpinning and
 * unmapping them takes 0.1-0.15 s that the exiting process used to pay after its outputs were closed; here it runs
 * beside the sort, the reduce and the writers. */
typedef struct {
    pthread_t dec_thread, pin_thread; int pin_started; dec_ctx *dec; fastf_bam_t *bam; int keep_first;
    uint64_t no_xf, no_gx; double t_release;
} rel_ctx;

/* the pin thread has stopped: unregister what it registered (but the first keep_first slots: on loan), free the slab (if none is) */
static void slab_release(dec_ctx *d, int keep_first)
{
    for (int k = keep_first; k < d->n_slots; k++)
        if (d->pinned[k] == 1) { fastf_pinned_unregister(d->slab + (size_t)k * d->slot_bytes); d->pinned[k] = 0; }
    if (!keep_first) { fastf_big_free(d->slab, (size_t)d->n_slots * d->slot_bytes); d->slab = NULL; }
    else if (keep_first < d->n_slots)
        fastf_big_drop(d->slab + (size_t)keep_first * d->slot_bytes, (size_t)(d->n_slots - keep_first) * d->slot_bytes);      /* the other slots' pages, at least */
}
static void pin_thread_stop(dec_ctx *d, pthread_t th)
{
    pthread_mutex_lock(&d->mu); d->pin_stop = 1; pthread_cond_broadcast(&d->cv); pthread_mutex_unlock(&d->mu);
    pthread_join(th, NULL);
}

static void *release_main(void *vp)
{
    rel_ctx *r = (rel_ctx *)vp;
    const double t0 = now_s();
    pthread_join(r->dec_thread, NULL);                  /* it has delivered its end-of-file batch and is on its way out */
    fastf_bam_stats(r->bam, NULL, &r->no_xf, &r->no_gx);
    fastf_bam_close(r->bam);                            /* prints the reader's profile lines first */
    /* keep_first: that many slots at the front of the slab are on loan to the engine as its row buffer until the outputs are written (bam2db() below) */
    if (r->pin_started) pin_thread_stop(r->dec, r->pin_thread);
    slab_release(r->dec, r->keep_first);
    r->t_release = now_s() - t0;
    return NULL;
}

/* Which device(s) a run uses.  FASTF_DEVICES = "a,b,.." (explicit ordinals: the first is the engine's and the reader's device, the
 * second the one a second inflate context may use) or a count n (devices 0..n-1; n == 1: one device, FASTF_DEVICE says which);
 * without FASTF_DEVICES, FASTF_DEVICE; without both, device 0.  Exported for the host-only test of the parsing. */
void fastf_pick_devices(const char *devices, const char *device, int *dev0, int *dev_second)
{
    int d0 = 0, d2 = -1;
    if (devices && *devices) {
        const char *c = strchr(devices, ',');
        if (c) { d0 = (int)strtol(devices, NULL, 10); d2 = (int)strtol(c + 1, NULL, 10); }
        else {
            const int n = atoi(devices);
            if (n >= 2) d2 = 1;                                    /* devices 0..n-1 */
            else if (device) d0 = atoi(device);                    /* one device: FASTF_DEVICE says which */
        }
    } else if (device) d0 = atoi(device);
    if (d0 < 0 || d0 > 254) d0 = 0;
    if (d2 < 0 || d2 > 254 || d2 == d0) d2 = -1;
    *dev0 = d0; *dev_second = d2;
}

static uint32_t bits_for(uint64_t v) { uint32_t b = 0; while (b < 64 && (v >> b)) b++; return b ? b : 1; }

/* umi_bases: 0 = choose (16 bases if the packed key then fits 64 bits, else 12 with a second run at 16 — keys wider than 64
 * bits — should the file hold longer UMIs: *longer_umis is set and everything is torn down for that run) */
static int bam2db_run(char *bam_file, char *path_out, char *barcodes_file, char *features_file,
                      float rate_cell, float rate_depth, unsigned int seed, uint32_t umi_bases, int *longer_umis)
{
    int rc = 1;
    {   extern int fastf_worker_nice_;
        const char *wn = getenv("FASTF_WORKER_NICE");
        fastf_worker_nice_ = wn ? atoi(wn) : 0; }
    const int prof = getenv("FASTF_PROFILE") != NULL;
    double t0 = now_s(), t_lists = 0, t_engine = 0, t_decode = 0, t_push = 0, t_finish = 0, t_write = 0, t_wait_release = 0, tt;
    fastf_lists_t lists; memset(&lists, 0, sizeof lists);
    fastf_bam_t *bam = NULL;
    fastf_engine_t *eng = NULL;
    dec_ctx dec; pthread_t dec_thread, pin_thread; int dec_started = 0, pin_started = 0; double t_wait = 0;
    rel_ctx rel; pthread_t rel_thread; int rel_started = 0;
    memset(&dec, 0, sizeof dec); memset(&rel, 0, sizeof rel);

    /* bam2db needs a device anyway: unless told otherwise (FASTF_GPU_INFLATE=0) the BGZF inflate of big windows is
     * shared between the host threads and the device (host_io.c: hybrid inflate, every block CRC-checked on the host) */
    /* ONE device ordinal for the reader's device side and the engine (several devices: the first of them): FASTF_DEVICES
     * ("0,1,2,3", or a count = devices 0..count-1), else FASTF_DEVICE, else device 0 */
    int dev0 = 0, dev_second = -1;         /* dev_second: the engine's second device, for the reader's second inflate context (host_io.c) */
    {   const char *dvs = getenv("FASTF_DEVICES"), *dv1 = getenv("FASTF_DEVICE");
        fastf_pick_devices(dvs, dv1, &dev0, &dev_second); }
    {   /* | 4: the device-side parse will be asked for (below, once the lists are known) unless it is switched off */
        const char *gp = getenv("FASTF_GPU_PARSE");
        bam = fastf_bam_open2(bam_file, 0, 1 | ((gp && gp[0] == '0') ? 0 : 4) | ((dev0 + 1) << 8) | ((dev_second + 1) << 16));
    }
    if (!bam) { fprintf(stderr, "Fail to open BAM file %s (%s)\n", bam_file, fastf_last_error()); goto done; }
    fprintf(stderr, "Opened BAM file %s successfully\n", bam_file);

    tt = now_s();
    if (fastf_lists_load(barcodes_file, features_file, rate_cell, seed, &lists)) {
        fprintf(stderr, "Fail to load barcode/feature lists: %s\n", fastf_last_error());
        goto done;
    }
    t_lists = now_s() - tt;
    printf("Total number of cells: %zu\n", lists.n_lines_barcodes);
    printf("Actual number of sampled cell barcodes: %zu\n", lists.n_sampled_target);
    for (size_t i = 0; i < lists.dup_barcodes; i++) printf("Warning: Duplicate cell barcodes were found in %s!\n", barcodes_file);
    for (size_t i = 0; i < lists.dup_features; i++) printf("Warning: Duplicate feature names were found in %s!\n", features_file);

    /* start decoding right away; the engine (HIP init, table upload) comes up meanwhile */
    const char *bs = getenv("FASTF_BATCH_RECORDS");
    const size_t cap = bs ? (size_t)strtoull(bs, NULL, 0) : ((size_t)4 << 20);
    memset(&dec, 0, sizeof dec);
    dec.bam = bam; dec.lists = &lists; dec.cap = cap;
    {   const char *pf = getenv("FASTF_PROFILE"); dec.trace = pf && pf[0] == '2'; dec.t_origin = t0; }
    double t_engine_up = 0, t_closed = 0;
    pthread_mutex_init(&dec.mu, NULL); pthread_cond_init(&dec.cv, NULL);
    /* one slab for the ring of decoder slots (untouched pages cost nothing): each slot is pinned when the decoder reaches
     * it, so that the engine copies the packed records to the device straight from where the decoder wrote them */
    /* UMIs beyond 16 bases (the third attempt of bam2db(), or FASTF_UMI_MAX_BASES): the slots carry a fifth array, the host packs */
    const int long_umis = umi_bases > 16 || (getenv("FASTF_UMI_MAX_BASES") && atoi(getenv("FASTF_UMI_MAX_BASES")) > 16);
    dec.n_slots = DEC_SLOTS; dec.slot_bytes = (cap * (long_umis ? 28 : 24) + 4095) & ~(size_t)4095;     /* whole pages: two slots' registrations never share one */
    if (!(dec.slab = (unsigned char *)fastf_big_alloc((size_t)dec.n_slots * dec.slot_bytes))) { fprintf(stderr, "out of memory\n"); goto done; }
    for (int k = 0; k < dec.n_slots; k++) {
        unsigned char *base = dec.slab + (size_t)k * dec.slot_bytes;
        dec.slot[k].cb = (uint64_t *)base; dec.slot[k].gx = (uint64_t *)(base + cap * 8);
        dec.slot[k].umi = (uint32_t *)(base + cap * 16); dec.slot[k].meta = (uint32_t *)(base + cap * 20);
        dec.slot[k].ext = long_umis ? (uint32_t *)(base + cap * 24) : NULL;
    }
    /* the windows the device inflates are hopped and packed there as well (one device; lists the device packer can hold) */
    /* (several devices: the batches the first device packs go to whichever device their chunk is dealt to, device to device) */
    if (!long_umis) (void)fastf_bam_enable_device_parse(bam, lists.cell_dict, lists.feat_dict);
    printf("Start to convert bam file to UMI keys on the device...\n");
    if (pthread_create(&dec_thread, NULL, decoder_main, &dec) != 0) { fprintf(stderr, "cannot start decoder thread\n"); goto done; }
    dec_started = 1;
    {   const char *zc = getenv("FASTF_ZERO_COPY");                 /* "0": stage every batch through the engine's own pinned buffers */
        if (!(zc && zc[0] == '0') && pthread_create(&pin_thread, NULL, pin_main, &dec) == 0) pin_started = 1;
        else for (int k = 0; k < dec.n_slots; k++) dec.pinned[k] = -1;
    }

    fastf_engine_config_t cfg; memset(&cfg, 0, sizeof cfg);
    cfg.cell_keys = lists.cell_key; cfg.n_cells = (uint32_t)lists.n_cells;
    cfg.feature_keys = lists.feature_key; cfg.n_features = (uint32_t)lists.n_features;
    cfg.draw_threshold = fastf_draw_threshold(rate_depth);
    cfg.mt_seed = seed; cfg.mt_skip = lists.mt_skip;
    cfg.n_shards = 1; cfg.shard_rank = 0;
    cfg.device = dev0;
    /* FASTF_DEVICES: "4" = devices 0..3, or an explicit list "0,1,2,3" (ordinals may repeat: all shards on one GPU).
     * Two or more entries make this one process drive that many GPUs (cell-hash shards, one key exchange). */
    int32_t dev_list[8]; uint32_t n_dev = 0;
    const char *dv = getenv("FASTF_DEVICES");
    if (dv && *dv) {
        if (strchr(dv, ',')) {
            for (const char *q = dv; *q && n_dev < 8;) { dev_list[n_dev++] = (int32_t)strtol(q, (char **)&q, 10); while (*q == ',' || *q == ' ') q++; }
        } else {
            n_dev = (uint32_t)atoi(dv); if (n_dev > 8) n_dev = 8;
            for (uint32_t i = 0; i < n_dev; i++) dev_list[i] = (int32_t)i;
        }
        if (n_dev >= 2) { cfg.n_devices = n_dev; cfg.devices = dev_list; }
    }
    /* 16-base UMIs (36 key bits) when the packed key then fits 64 bits; else 12 bases (27 bits: what 10x chemistry writes) if
     * THAT fits — the fast 64-bit path — with a second run should the file turn out to hold longer UMIs (umi_bases = 16 then);
     * else 16 bases with keys wider than 64 bits (the engine sorts the (cell, feature) word and carries the rest beside it) */
    const char *ul = getenv("FASTF_UMI_MAX_BASES");
    const uint32_t group_bits = bits_for(cfg.n_cells) + bits_for(cfg.n_features);
    if (umi_bases) cfg.umi_max_bases = umi_bases;
    else if (ul) cfg.umi_max_bases = (uint32_t)atoi(ul);
    else cfg.umi_max_bases = (group_bits + 36 <= 64 || group_bits + 27 > 64) ? 16 : 12;
    const int may_rerun = !ul && cfg.umi_max_bases < 32;
    cfg.batch_records = cap;
    tt = now_s();
    if (fastf_engine_create(&cfg, &eng)) { fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error()); goto done; }
    t_engine = now_s() - tt;
    __atomic_store_n(&dec.engine_up, 1, __ATOMIC_RELEASE);
    t_engine_up = now_s() - t0;
    if (dec.trace) fprintf(stderr, "[trace] %.3f main: engine up\n", now_s() - t0);

    for (int k = 0;; k = (k + 1) % dec.n_slots) {
        tt = now_s();
        pthread_mutex_lock(&dec.mu);
        while (!dec.filled[k]) pthread_cond_wait(&dec.cv, &dec.mu);
        /* records in the slab wait for their slot's registration (the pin thread is at most a few slots behind) */
        while (dec.slot[k].n > 0 && !dec.slot[k].on_device && dec.pinned[k] == 0) pthread_cond_wait(&dec.cv, &dec.mu);
        const int slot_pinned = dec.pinned[k] == 1;
        pthread_mutex_unlock(&dec.mu);
        t_wait += now_s() - tt;
        const long n = dec.slot[k].n;
        if (n < 0) { fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error()); goto done; }
        if (n == 0) break;
        fastf_batch_t batch = { dec.slot[k].cb, dec.slot[k].gx, dec.slot[k].umi, dec.slot[k].meta, (size_t)n, dec.slot[k].ext };
        const int on_dev = dec.slot[k].on_device;
        if (on_dev) batch = dec.slot[k].dev;                  /* packed on the device: a device-to-device copy into the engine's staging */
        tt = now_s();
        /* pinned slots: queue the copies, and hand the slot back to the decoder once they have left it */
        if ((slot_pinned || on_dev) ? (fastf_engine_push_pinned(eng, &batch) || fastf_engine_wait_input(eng)) : fastf_engine_push(eng, &batch)) {
            if (may_rerun && strstr(fastf_last_error(), "UMI longer")) { *longer_umis = (int)cfg.umi_max_bases; goto done; }
            fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error()); goto done;
        }
        t_push += now_s() - tt;
        if (dec.trace) fprintf(stderr, "[trace] %.3f main: slot %d pushed (%ld records) in %.3f s\n", now_s() - t0, k, n, now_s() - tt);
        pthread_mutex_lock(&dec.mu);
        dec.filled[k] = 0;
        pthread_cond_broadcast(&dec.cv);
        pthread_mutex_unlock(&dec.mu);
    }
    t_decode = dec.t_decode;
    /* every record is on the device (each push waited for its copies): the reader and the slab go now, beside what follows */
    {   extern int fastf_reader_leaves_device_side_;
        fastf_reader_leaves_device_side_ = fastf_process_is_exiting_; }
    /* the matrix rows go into the first decoder slot — pinned already, its pages touched, idle from here on: finish() then
     * has no row buffer to allocate and fault in (0.03-0.05 s beside the release thread's unmapping) and the device writes
     * the rows at PCIe rate.  A matrix that does not fit takes the engine's own buffer. */
    const char *lr = getenv("FASTF_LEND_ROWS");                                  /* "0": the engine's own row buffer (A/B) */
    pthread_mutex_lock(&dec.mu);
    /* (every slot the decoder reached is pinned and idle by now: all of them from the front of the slab — 12 bytes a row; a matrix
     * of 40 M rows in pageable memory came over at 1-10 GB/s, its page faults behind the release thread's unmapping) */
    int n_lend = 0;
    while (n_lend < dec.n_slots && dec.pinned[n_lend] == 1) n_lend++;
    pthread_mutex_unlock(&dec.mu);
    const int lend = (n_lend > 0 && !(lr && lr[0] == '0') && fastf_engine_lend_rows(eng, dec.slab, (size_t)n_lend * dec.slot_bytes) == 0) ? n_lend : 0;
    rel.dec_thread = dec_thread; rel.pin_thread = pin_thread; rel.pin_started = pin_started; rel.dec = &dec; rel.bam = bam; rel.keep_first = lend;
    if (pthread_create(&rel_thread, NULL, release_main, &rel) == 0) {
        /* the release thread stops the pin thread and gives the slab back, but for the slots on loan */
        rel_started = 1; dec_started = 0; pin_started = 0; bam = NULL;
    }
    fastf_coo_t coo; uint64_t counters[3];
    tt = now_s();
    if (fastf_engine_finish(eng, &coo, counters)) {
        if (may_rerun && strstr(fastf_last_error(), "UMI longer")) { *longer_umis = (int)cfg.umi_max_bases; goto done; }
        fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error()); goto done;
    }
    t_finish = now_s() - tt;
    printf("In %s, total fastQ reads: %zu\n", bam_file, (size_t)counters[0]);
    printf("In %s, sampled fastQ reads: %zu\n", bam_file, (size_t)counters[1]);
    printf("In %s, sampled and valid fastQ reads: %zu\n", bam_file, (size_t)counters[2]);
    if (!rel_started) fastf_bam_stats(bam, NULL, &rel.no_xf, &rel.no_gx);

    fastf_umi_rows_t urows; memset(&urows, 0, sizeof urows);
    if (_umi_copies_flag && fastf_engine_umi_rows(eng, &urows)) { fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error()); goto done; }
    tt = now_s();
    if (fastf_write_outputs(path_out, bam_file, rate_cell, rate_depth, counters, &lists, &coo,
                            _umi_copies_flag ? &urows : NULL)) {
        fprintf(stderr, "\x1b[31mError:\x1b[0m %s\n", fastf_last_error());
        goto done;
    }
    t_write = now_s() - tt;
    rc = 0;
    t_closed = now_s() - t0;
    if (prof) { struct timespec rt; clock_gettime(CLOCK_REALTIME, &rt); fprintf(stderr, "[bam2db] outputs closed at %.6f (unix time)\n", rt.tv_sec + rt.tv_nsec * 1e-9); }
    if (rel_started) { tt = now_s(); pthread_join(rel_thread, NULL); rel_started = 0; t_wait_release = now_s() - tt; }
    if (rel.no_xf || rel.no_gx)
        fprintf(stderr, "Note: %llu records with a CB but no xf tag and %llu with a valid xf but no GX tag were skipped "
                        "(the reference dereferences NULL on them).\n", (unsigned long long)rel.no_xf, (unsigned long long)rel.no_gx);
    if (prof) {
        /* the three pieces of a run: until the device takes work (the host threads decode alone meanwhile), the steady state
         * (records decoded from then on, reader windows shared with the device), and what is left after the last record */
        const double t_se = dec.t_saw_engine > 0 ? dec.t_saw_engine : dec.t_last_record;
        const uint64_t n_steady = dec.n_decoded - (dec.t_saw_engine > 0 ? dec.n_before_engine : dec.n_decoded);
        fprintf(stderr, "[bam2db] phases: engine_up_s=%.3f decoder_saw_engine_s=%.3f records_before=%llu steady_records=%llu steady_s=%.3f "
                        "last_record_s=%.3f outputs_closed_s=%.3f\n", t_engine_up, t_se,
                (unsigned long long)(dec.n_decoded - n_steady), (unsigned long long)n_steady,
                dec.t_last_record > t_se ? dec.t_last_record - t_se : 0.0, dec.t_last_record, t_closed);
    }
    if (prof)
        fprintf(stderr, "[bam2db] lists %.3f s, engine create %.3f s, BAM decode+pack %.3f s (decoder thread; main waited %.3f s), "
                        "push (stage+H2D+K1 enqueue) %.3f s, finish (sort+reduce+D2H) %.3f s, write %.3f s, reader+slab release %.3f s "
                        "(beside finish and write; waited %.3f s for it), total so far %.3f s\n",
                t_lists, t_engine, t_decode, t_wait, t_push, t_finish, t_write, rel.t_release, t_wait_release, now_s() - t0);
done:
    tt = now_s();
    if (rel_started) pthread_join(rel_thread, NULL);    /* (an error after the end of the input) */
    if (dec_started) {
        pthread_mutex_lock(&dec.mu); dec.stop = 1; memset(dec.filled, 0, sizeof dec.filled);
        pthread_cond_broadcast(&dec.cv); pthread_mutex_unlock(&dec.mu);
        pthread_join(dec_thread, NULL);
    }
    if (pin_started) { pin_thread_stop(&dec, pin_thread); pin_started = 0; }
    if (fastf_process_is_exiting_ && !*longer_umis) {
        /* the fastF CLI leaves through _exit() right after this call: device memory, pinned pages and the BAM mapping
         * go back with the process, and unmapping them one by one first costs 0.1-0.2 s */
        if (bam) fastf_bam_print_profile(bam);
        if (prof) fprintf(stderr, "[bam2db] engine teardown skipped (process exits), %.3f s\n", now_s() - tt);
        return rc;
    }
    if (eng) fastf_engine_destroy(eng);
    if (dec.slab) slab_release(&dec, 0);                /* (whatever the release thread left: the lent slot, or all of it) */
    if (bam) fastf_bam_close(bam);
    fastf_lists_free(&lists);
    if (prof) fprintf(stderr, "[bam2db] teardown (engine, pinned slab, BAM mapping, lists) %.3f s\n", now_s() - tt);
    return rc;
}

int bam2db(char *bam_file, char *db_file, char *path_out, char *barcodes_file, char *features_file,
           float rate_cell, float rate_depth, unsigned int seed)
{
    (void)db_file;                      /* no SQLite in this engine */
    /* the reference takes a UMI of any length (bam2db_ds.c:417-419).  The first run assumes what keeps the key within 64 bits
     * where it can (12 or 16 bases); a file with longer UMIs is run again with room for 16, then 32 bases — keys wider than 64
     * bits, and from 17 bases on a fifth array beside the packed records.  Beyond 32 bases: an error. */
    int longer_umis = 0;
    int rc = bam2db_run(bam_file, path_out, barcodes_file, features_file, rate_cell, rate_depth, seed, 0, &longer_umis);
    while (longer_umis) {
        const uint32_t next = longer_umis < 16 ? 16 : 32;
        fprintf(stderr, "Note: UMIs longer than %d bases in %s: running again with room for %u (keys wider than 64 bits)\n", longer_umis, bam_file, next);
        longer_umis = 0;
        rc = bam2db_run(bam_file, path_out, barcodes_file, features_file, rate_cell, rate_depth, seed, next, &longer_umis);
    }
    return rc;
}

/* ------------------------------------------------------------------ */
/* CLI                                                                 */
/* ------------------------------------------------------------------ */
static void usage_bam2db(FILE *f)
{
    fprintf(f,
            "Usage: fastF bam2db [options]\n\n"
            "Filter bam file with desired cell proportion and read depth on the GPU, then summarise it into UMI matrix.\n\n"
            "    -h, --help            show this help message and exit\n"
            "    -b, --bam=<str>       path to bam file\n"
            "    -f, --feature=<str>   path to feature list file\n"
            "    -a, --barcode=<str>   path to barcode list file\n"
            "    -d, --dbname=<str>    name of database (accepted for compatibility, ignored)\n"
            "    -c, --cell=<flt>      rate of cell barcode (default 1.0)\n"
            "    -r, --depth=<flt>     rate of depth (default 1.0)\n"
            "    -o, --out=<str>       path to output directory (default .)\n"
            "    -s, --seed=<int>      seed for random number generator (default 926)\n"
            "    -u, --umicopies       whether to store umi.tsv.gz\n");
}

struct opt { char s; const char *l; int has_arg; };
static const struct opt k_opts[] = {
    {'h', "help", 0}, {'b', "bam", 1}, {'f', "feature", 1}, {'a', "barcode", 1}, {'d', "dbname", 1},
    {'c', "cell", 1}, {'r', "depth", 1}, {'o', "out", 1}, {'s', "seed", 1}, {'u', "umicopies", 0}, {0, NULL, 0}};

int cmd_bam2db(int argc, const char **argv)
{
    const char *bam = NULL, *feat = NULL, *bar = NULL, *db = NULL, *out = ".";
    /* the reference leaves the two rates uninitialised when omitted (main.c:293-294); 1.0 here */
    float rate_cell = 1.0f, rate_depth = 1.0f;
    unsigned int seed = 926;

    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const struct opt *o = NULL;
        const char *val = NULL;
        if (a[0] != '-' || !a[1]) break;                       /* first non-option stops parsing */
        if (a[1] == '-') {
            if (!a[2]) break;                                  /* "--" */
            const char *eq = strchr(a + 2, '=');
            size_t nl = eq ? (size_t)(eq - a - 2) : strlen(a + 2);
            for (const struct opt *k = k_opts; k->l; k++)
                if (strlen(k->l) == nl && strncmp(k->l, a + 2, nl) == 0) { o = k; break; }
            if (o && eq) val = eq + 1;
        } else {
            for (const struct opt *k = k_opts; k->l; k++) if (k->s == a[1]) { o = k; break; }
            if (o && o->has_arg && a[2]) val = a + 2;          /* -xVALUE */
        }
        /* messages and exit status as argparse.c:36-46, 274-277: the option is named by its own spelling (`--cell`,
         * `-c`), never with the value glued to it, and the process ends with EXIT_FAILURE */
        if (!o) { fprintf(stderr, "error: unknown option `%s`\n", a); usage_bam2db(stderr); exit(EXIT_FAILURE); }
        char oname[32];
        if (a[1] == '-') snprintf(oname, sizeof oname, "--%s", o->l); else snprintf(oname, sizeof oname, "-%c", o->s);
        if (o->has_arg && !val) {
            if (i + 1 >= argc) { fprintf(stderr, "error: option `%s` requires a value\n", oname); exit(EXIT_FAILURE); }
            val = argv[++i];
        }
        char *end = NULL;
        switch (o->s) {
        case 'h': usage_bam2db(stdout); exit(0);
        case 'b': bam = val; break;
        case 'f': feat = val; break;
        case 'a': bar = val; break;
        case 'd': db = val; break;
        case 'o': out = val; break;
        /* argparse.c:88-108: ERANGE first, then trailing characters; an empty value parses as 0 */
        case 'c': errno = 0; rate_cell = strtof(val, &end); if (errno == ERANGE) { fprintf(stderr, "error: option `%s` numerical result out of range\n", oname); exit(EXIT_FAILURE); } if (*end) { fprintf(stderr, "error: option `%s` expects a numerical value\n", oname); exit(EXIT_FAILURE); } break;
        case 'r': errno = 0; rate_depth = strtof(val, &end); if (errno == ERANGE) { fprintf(stderr, "error: option `%s` numerical result out of range\n", oname); exit(EXIT_FAILURE); } if (*end) { fprintf(stderr, "error: option `%s` expects a numerical value\n", oname); exit(EXIT_FAILURE); } break;
        case 's': errno = 0; seed = (unsigned int)strtol(val, &end, 0); if (errno == ERANGE) { fprintf(stderr, "error: option `%s` numerical result out of range\n", oname); exit(EXIT_FAILURE); } if (*end) { fprintf(stderr, "error: option `%s` expects an integer value\n", oname); exit(EXIT_FAILURE); } break;
        case 'u': _umi_copies_flag = 1; break;
        }
    }

    /* existence checks of main.c:323-345 */
    if (!bam || access(bam, F_OK) == -1) { fprintf(stderr, "\x1b[31mError:\x1b[0m bam file: %s does not exist.\n", bam ? bam : "(null)"); exit(1); }
    if (!feat || access(feat, F_OK) == -1) { fprintf(stderr, "\x1b[31mError:\x1b[0m feature file: %s does not exist.\n", feat ? feat : "(null)"); exit(1); }
    if (!bar || access(bar, F_OK) == -1) { fprintf(stderr, "\x1b[31mError:\x1b[0m barcode file: %s does not exist.\n", bar ? bar : "(null)"); exit(1); }
    if (db && access(db, F_OK) != -1) {
        fprintf(stderr, "\x1b[31mError:\x1b[0m database: %s already exists, change the name of database in -d argument!\n", db);
        exit(1);
    }
    if (bam2db((char *)bam, (char *)db, (char *)out, (char *)bar, (char *)feat, rate_cell, rate_depth, seed)) {
        fprintf(stderr, "\x1b[31mError:\x1b[0m bam2db failed.\n");
        return 1;
    }
    return 0;
}
