// fuzz_records.cpp — the device-side record functions (gpu_records.hpp, host build: tools/gr_host.cpp) against the host reader on a
// damaged payload, under AddressSanitizer (tools/fuzz_reader.py drives it):
//   fuzz_records <file.bam = the payload in valid BGZF> <payload.bin = the same bytes, inflated> <first record offset> barcodes features
// The host reader delivers the records in front of the first structurally invalid one, window by window; the device functions walk
// the same chain over a buffer that ends exactly where the payload ends (a read beyond it is an ASan report).  Every record the
// reader delivered must be packed identically.  Exit status 0 and a line "records host <n> device <m> equal <k>"; 3 on a mismatch.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
extern "C" {
#include "host_io.h"
}
struct View { uint32_t n_prefix; uint32_t prefix_len[8]; uint64_t prefix_id[8]; unsigned char prefix[8][32]; };
extern "C" long gr_host_parse(const uint8_t* buf, uint64_t start, uint64_t end, const View* cells, const View* feats,
                              uint64_t* cb, uint64_t* gx, uint32_t* umi, uint32_t* meta, uint64_t cap, uint32_t* no_xf, uint32_t* no_gx, uint64_t* handover);
extern "C" uint64_t gr_host_guess(const uint8_t* buf, uint64_t lo, uint64_t hi, uint64_t end, uint32_t n_ref);

int main(int argc, char** argv) {
    if (argc < 6) return 9;
    fastf_lists_t lists; memset(&lists, 0, sizeof lists);
    if (fastf_lists_load(argv[4], argv[5], 1.0f, 926, &lists)) return 8;
    fastf_keydict_view_t vc, vf;
    if (fastf_keydict_export(lists.cell_dict, &vc) || fastf_keydict_export(lists.feat_dict, &vf)) return 7;
    static_assert(sizeof(View) == sizeof(fastf_keydict_view_t), "view layout");
    // device functions over the inflated payload, in a buffer of exactly its size
    FILE* f = fopen(argv[2], "rb"); if (!f) return 6;
    fseek(f, 0, SEEK_END); const long len = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)malloc((size_t)len); if (fread(buf, 1, (size_t)len, f) != (size_t)len) return 6; fclose(f);
    const uint64_t off = strtoull(argv[3], nullptr, 10), cap = (uint64_t)len / 36 + 1;
    std::vector<uint64_t> dcb(cap), dgx(cap); std::vector<uint32_t> dum(cap), dme(cap);
    uint32_t nx = 0, ng = 0; uint64_t hand = 0;
    const long m = gr_host_parse(buf, off, (uint64_t)len, (const View*)&vc, (const View*)&vf, dcb.data(), dgx.data(), dum.data(), dme.data(), cap, &nx, &ng, &hand);
    // the chain-start guess from a few places (what a hop segment does): must stay inside the buffer too
    uint64_t acc = 0;
    for (uint64_t lo = off; lo + 64 < (uint64_t)len; lo += 4099) acc += gr_host_guess(buf, lo, lo + 64, (uint64_t)len, 2);
    // the host reader on the same bytes
    std::vector<uint64_t> hcb, hgx; std::vector<uint32_t> hum, hme;
    fastf_bam_t* b = fastf_bam_open(argv[1], 4);
    long n = 0;
    if (b) {
        const size_t bc = 5000;
        std::vector<uint64_t> cb(bc), gx(bc); std::vector<uint32_t> um(bc), me(bc);
        for (;;) {
            const long k = fastf_bam_read_batch(b, lists.cell_dict, lists.feat_dict, cb.data(), gx.data(), um.data(), me.data(), bc);
            if (k <= 0) break;
            hcb.insert(hcb.end(), cb.begin(), cb.begin() + k); hgx.insert(hgx.end(), gx.begin(), gx.begin() + k);
            hum.insert(hum.end(), um.begin(), um.begin() + k); hme.insert(hme.end(), me.begin(), me.begin() + k);
            n += k;
        }
        fastf_bam_close(b);
    }
    long eq = 0;
    const long both = n < m ? n : m;
    for (long i = 0; i < both; i++) eq += hcb[i] == dcb[i] && hgx[i] == dgx[i] && hum[i] == dum[i] && hme[i] == dme[i];
    printf("records host %ld device %ld equal %ld (guess sum %llx)\n", n, m, eq, (unsigned long long)acc);
    free(buf); fastf_lists_free(&lists);
    return (eq == both && n <= m) ? 0 : 3;
}
