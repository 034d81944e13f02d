/* diagnostic: the BAM tag reader + registering key dictionary under sanitizers (CPU build only):
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude -Ifastf_amd/csrc tools/san_tags.c fastf_amd/csrc/{host_io,host_prims,inflate_fast,crc32_fast}.c -lz -lpthread -o build/san_tags
 *   (and -fsanitize=thread);  build/san_tags file.bam CB CR */
#include "host_io.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
void fastf_set_error_(const char *m) { fprintf(stderr, "err: %s\n", m); }
const char *fastf_last_error(void) { return ""; }
int main(int argc, char **argv) {
    for (int threads = 1; threads <= 8; threads *= 2) {
        fastf_bam_t *b = fastf_bam_open(argv[1], threads);
        if (!b) return 1;
        fastf_keydict_t *d = fastf_keydict_create();
        size_t cap = 20000; uint64_t *k1 = malloc(cap * 8), *k2 = malloc(cap * 8); uint64_t undef = 0, n = 0, x = 0;
        for (;;) {
            long m = fastf_bam_read_tags(b, d, argv[2], argc > 3 ? argv[3] : NULL, 0, k1, k2, cap, &undef);
            if (m <= 0) break;
            char buf[128];
            for (long i = 0; i < m; i++) if (k1[i]) { long l = fastf_keydict_decode(d, k1[i], buf, sizeof buf); if (l < 0) return 2; x += (uint64_t)l; }
            n += (uint64_t)m;
        }
        printf("threads %d: %llu records, decoded bytes %llu, undefined %llu\n", threads, (unsigned long long)n, (unsigned long long)x, (unsigned long long)undef);
        free(k1); free(k2); fastf_keydict_destroy(d); fastf_bam_close(b);
    }
    return 0;
}
