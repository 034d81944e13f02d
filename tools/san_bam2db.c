/* diagnostic: bam2db()'s host side — the decoder thread and its ring of slots, the pin thread, the release thread, the scout, the
 * writers with their side thread — under sanitizers, CPU build only.  The engine is a stub that comes up late (as the HIP runtime
 * does), takes every batch in the order it is pushed and checks that order with a running checksum, and hands back a small fixed
 * matrix in the memory it was lent; no device, no kernels (the real engine is tested on the GPU: tests/test_gpu_*.py).
 *   gcc -O1 -g -fsanitize=thread -Iinclude -Ifastf_amd/csrc -Itools tools/san_bam2db.c fastf_amd/csrc/{bam2db_main,host_io,host_prims,inflate_fast,crc32_fast,deflate_fast}.c -lz -lpthread -o build/san_bam2db
 *   build/san_bam2db file.bam barcodes.tsv features.tsv outdir   -> "pushed <n> records in <k> batches, checksum <c>" */
#define SAN_PINNED_REGISTER_RC 0
#include "fastf_amd.h"
#include "san_stubs.h"
#include <string.h>
#include <time.h>

extern int fastf_process_is_exiting_;   /* bam2db_main.c */

struct fastf_engine { uint64_t n, batches, sum; void *lent; size_t lent_bytes; uint32_t rows[3][4]; int finished; };
static struct fastf_engine g_eng;
static void nap_ms(long ms) { struct timespec t = { ms / 1000, (ms % 1000) * 1000000L }; nanosleep(&t, NULL); }

int fastf_engine_create(const fastf_engine_config_t *cfg, fastf_engine_t **out)
{
    (void)cfg;
    nap_ms(150);                         /* the device context takes its time: the decoder runs ahead meanwhile */
    memset(&g_eng, 0, sizeof g_eng);
    *out = &g_eng;
    return 0;
}
void fastf_engine_destroy(fastf_engine_t *e) { (void)e; }
static int take(fastf_engine_t *e, const fastf_batch_t *b)
{
    for (size_t i = 0; i < b->n; i++)
        e->sum = e->sum * 1099511628211ull + (b->cb_key[i] ^ (b->gx_key[i] << 1) ^ b->umi[i] ^ ((uint64_t)b->meta[i] << 40));
    e->n += b->n; e->batches++;
    return 0;
}
int fastf_engine_push(fastf_engine_t *e, const fastf_batch_t *b) { return take(e, b); }
int fastf_engine_push_pinned(fastf_engine_t *e, const fastf_batch_t *b) { return take(e, b); }
int fastf_engine_wait_input(fastf_engine_t *e) { (void)e; return 0; }
int fastf_engine_lend_rows(fastf_engine_t *e, void *p, size_t bytes) { e->lent = p; e->lent_bytes = bytes; return 0; }
int fastf_engine_finish(fastf_engine_t *e, fastf_coo_t *coo, uint64_t counters[3])
{
    /* four rows, written where a real finish would put them: into the loan when there is one */
    const size_t cap = e->lent && e->lent_bytes >= 48 ? e->lent_bytes / 12 : 4;
    uint32_t *base = e->lent && e->lent_bytes >= 48 ? (uint32_t *)e->lent : &e->rows[0][0];
    for (uint32_t i = 0; i < 4; i++) { base[i] = 1 + i; base[cap + i] = 1; base[2 * cap + i] = i; }
    coo->feature = base; coo->cell = base + cap; coo->count = base + 2 * cap; coo->nnz = 4;
    counters[0] = e->n; counters[1] = e->n / 2; counters[2] = e->n / 4;
    e->finished = 1;
    return 0;
}
int fastf_engine_umi_rows(fastf_engine_t *e, fastf_umi_rows_t *rows) { (void)e; memset(rows, 0, sizeof *rows); return 0; }

int main(int argc, char **argv)
{
    if (argc < 5) return 9;
    fastf_process_is_exiting_ = argc > 5 && argv[5][0] == 'x';     /* 'x': leave like the CLI does (the release paths differ) */
    const int rc = bam2db(argv[1], NULL, argv[4], argv[2], argv[3], 0.5f, 0.5f, 926);
    printf("bam2db returned %d: pushed %llu records in %llu batches, checksum %016llx\n", rc, (unsigned long long)g_eng.n,
           (unsigned long long)g_eng.batches, (unsigned long long)g_eng.sum);
    return rc;
}
