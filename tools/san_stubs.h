/* tools/san_stubs.h — the HIP side of the library as seen by the CPU-only sanitizer harnesses (tools/san_reader.c, tools/san_bam2db.c):
 * no device, nothing inflated or parsed on one; "pinned" memory is ordinary memory. */
#ifndef FASTF_SAN_STUBS_H
#define FASTF_SAN_STUBS_H
#include "host_io.h"
#include <stdio.h>
#include <stdlib.h>
#ifndef SAN_PINNED_REGISTER_RC
#define SAN_PINNED_REGISTER_RC 1      /* fastf_pinned_register: 1 = "cannot pin" (the reader harness), 0 = pretend it worked */
#endif
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
uint64_t fastf_gpurec_repairs(const fastf_gpuinf_t *g) { (void)g; return 0; }
void *fastf_pinned_alloc(size_t n) { return malloc(n); }
void fastf_pinned_free(void *p) { free(p); }
int fastf_pinned_register(void *p, size_t n) { (void)p; (void)n; return SAN_PINNED_REGISTER_RC; }
void fastf_pinned_unregister(void *p) { (void)p; }

#endif
