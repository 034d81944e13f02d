/* diagnostic: the BAM reader's batch path (windows, progressive unmapping of the file, the spare window buffer going back at end
 * of file, huge-page allocations, the lists + key dictionaries) under sanitizers — CPU build only, the device side stubbed out:
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude -Ifastf_amd/csrc tools/san_reader.c fastf_amd/csrc/{host_io,host_prims,inflate_fast,crc32_fast,deflate_fast}.c -lz -lpthread -o build/san_reader
 *   (and -fsanitize=thread);  build/san_reader file.bam barcodes.tsv features.tsv   -> prints a checksum of the packed records per thread count */
#include "host_io.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "san_stubs.h"

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
