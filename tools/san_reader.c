/* diagnostic: the BAM reader's batch path (windows, progressive unmapping of the file, the spare window buffer going back at end
 * of file, huge-page allocations, the lists + key dictionaries) under sanitizers — CPU build only, the device side stubbed out:
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude -Ifastf_amd/csrc tools/san_reader.c fastf_amd/csrc/{host_io,host_prims,inflate_fast,crc32_fast,deflate_fast}.c -lz -lpthread -o build/san_reader
 *   (and -fsanitize=thread);  build/san_reader file.bam barcodes.tsv features.tsv   -> prints a checksum of the packed records per thread count */
#include "host_io.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
void fastf_set_error_(const char *m) { fprintf(stderr, "err: %s\n", m); }
const char *fastf_last_error(void) { return ""; }
/* the HIP side of the library is not part of this build: no device, nothing pinned */
fastf_gpuinf_t *fastf_gpuinf_create(int d) { (void)d; return NULL; }
int fastf_gpuinf_reserve(fastf_gpuinf_t *g, size_t w, size_t c, size_t n) { (void)g; (void)w; (void)c; (void)n; return 1; }
void fastf_gpuinf_destroy(fastf_gpuinf_t *g) { (void)g; }
int fastf_gpuinf_submit(fastf_gpuinf_t *g, const unsigned char *c, const fastf_gpuinf_blk_t *b, size_t n, unsigned char *o) { (void)g; (void)c; (void)b; (void)n; (void)o; return 1; }
int fastf_gpuinf_submit_keep(fastf_gpuinf_t *g, const unsigned char *c, const fastf_gpuinf_blk_t *b, size_t n, int p, const uint32_t *crc) { (void)g; (void)c; (void)b; (void)n; (void)p; (void)crc; return 1; }
int fastf_gpuinf_wait(fastf_gpuinf_t *g, uint8_t *s, double *ms) { (void)g; (void)s; (void)ms; return 1; }
void fastf_gpuinf_stats(const fastf_gpuinf_t *g, uint64_t *a, uint64_t *b) { (void)g; if (a) *a = 0; if (b) *b = 0; }
int fastf_gpurec_parse(fastf_gpuinf_t *g, int p, const unsigned char *t, size_t tl, uint64_t d, uint64_t e, uint32_t nr, const fastf_keydict_view_t *c, const fastf_keydict_view_t *f, fastf_gpurec_result_t *o) { (void)g; (void)p; (void)t; (void)tl; (void)d; (void)e; (void)nr; (void)c; (void)f; (void)o; return 1; }
int fastf_gpurec_fetch(fastf_gpuinf_t *g, int p, unsigned char *d, uint64_t a, uint64_t b) { (void)g; (void)p; (void)d; (void)a; (void)b; return 1; }
void fastf_gpurec_stats(const fastf_gpuinf_t *g, uint64_t *a, uint64_t *b) { (void)g; if (a) *a = 0; if (b) *b = 0; }
void *fastf_pinned_alloc(size_t n) { return malloc(n); }
void fastf_pinned_free(void *p) { free(p); }
int fastf_pinned_register(void *p, size_t n) { (void)p; (void)n; return 1; }
void fastf_pinned_unregister(void *p) { (void)p; }

int main(int argc, char **argv) {
    if (argc < 4) return 9;
    fastf_lists_t lists; memset(&lists, 0, sizeof lists);
    if (fastf_lists_load(argv[2], argv[3], 1.0f, 926, &lists)) return 3;
    for (int threads = 1; threads <= 8; threads *= 2) {
        fastf_bam_t *b = fastf_bam_open(argv[1], threads);
        if (!b) return 1;
        const size_t cap = 7001;
        uint64_t *cb = malloc(cap * 8), *gx = malloc(cap * 8); uint32_t *um = malloc(cap * 4), *me = malloc(cap * 4);
        uint64_t n = 0, sum = 0;
        for (;;) {
            int on_dev = 0; fastf_batch_t dev;
            long m = fastf_bam_read_batch_dev(b, lists.cell_dict, lists.feat_dict, cb, gx, um, me, cap, &on_dev, &dev);
            if (m < 0) return 2;
            if (m == 0) break;
            for (long i = 0; i < m; i++) sum = sum * 1099511628211ull + (cb[i] ^ (gx[i] << 1) ^ um[i] ^ ((uint64_t)me[i] << 40));
            n += (uint64_t)m;
        }
        printf("threads %d: %llu records, checksum %016llx\n", threads, (unsigned long long)n, (unsigned long long)sum);
        free(cb); free(gx); free(um); free(me);
        fastf_bam_close(b);
    }
    fastf_lists_free(&lists);
    return 0;
}
