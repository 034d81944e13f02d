// gr_host.cpp — fastf_amd/csrc/gpu_records.hpp compiled for the host (one lane): the device-side record hop, tag
// extraction, key packing and CRC slicing checked on the CPU against the host reader (tests/test_gpu_records_host.py).
#define GR_HOST
#include "../fastf_amd/csrc/gpu_records.hpp"
#include <string.h>
#include <vector>

struct View { uint32_t n_prefix; uint32_t prefix_len[8]; uint64_t prefix_id[8]; unsigned char prefix[8][32]; };   // fastf_keydict_view_t
static void to_dict(gr::Dict& d, const View& v) {
    memset(&d, 0, sizeof d);
    d.n_prefix = v.n_prefix;
    for (uint32_t i = 0; i < v.n_prefix && i < 8; ++i) { d.plen[i] = v.prefix_len[i]; d.pid[i] = v.prefix_id[i]; memcpy(d.ptext[i], v.prefix[i], 32); }
}

extern "C" {
// the chain from `start` over [start, end): every complete record packed; returns the count, *handover = offset behind the last
long gr_host_parse(const uint8_t* buf, uint64_t start, uint64_t end, const View* cells, const View* feats,
                   uint64_t* cb, uint64_t* gx, uint32_t* umi, uint32_t* meta, uint64_t cap, uint32_t* no_xf, uint32_t* no_gx, uint64_t* handover) {
    gr::Dict dc, df; to_dict(dc, *cells); to_dict(df, *feats);
    uint64_t o = start; long n = 0;
    *no_xf = *no_gx = 0;
    while ((uint64_t)n < cap) {
        const uint64_t e = end - o >= 36 ? gr::rec_end(buf, o, end) : 0;
        if (!e) break;
        gr::pack_record(buf + o, dc, df, cb + n, gx + n, umi + n, meta + n, no_xf, no_gx);
        o = e; n++;
    }
    *handover = o;
    return n;
}
// first offset in [lo, hi) from which three plausible records follow each other (the hop kernel's guess); ~0 if none
uint64_t gr_host_guess(const uint8_t* buf, uint64_t lo, uint64_t hi, uint64_t end, uint32_t n_ref) {
    for (uint64_t o = lo; o < hi; o++) {
        uint64_t q = o; int k = 0;
        while (k < 3 && gr::rec_plausible(buf, q, end, n_ref)) { q = gr::rec_end(buf, q, end); k++; }
        if (k == 3 || (k > 0 && gr::chain_meets_window_end(buf, q, end))) return o;
    }
    return ~0ull;
}
// CRC-32 of p[0, n) computed as the kernel does: `slices` slices hashed on their own, combined pairwise in a tree
uint32_t gr_host_crc(const uint8_t* p, uint32_t n, uint32_t slices) {
    uint32_t tab[256], x2n[32];
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = c & 1 ? (c >> 1) ^ gr::CRC_POLY : c >> 1; tab[i] = c; }
    uint32_t q = 1u << 30; x2n[0] = q;
    for (int i = 1; i < 32; ++i) x2n[i] = q = gr::crc_multmodp(q, q);
    std::vector<uint32_t> crc(slices), len(slices);
    const uint32_t per = (n + slices - 1) / slices;
    for (uint32_t l = 0; l < slices; ++l) {
        const uint32_t lo = l * per < n ? l * per : n, hi = lo + per < n ? lo + per : n;
        crc[l] = gr::crc_bytes(tab, p + lo, hi - lo); len[l] = hi - lo;
    }
    for (uint32_t d = 1; d < slices; d <<= 1)
        for (uint32_t l = 0; l + d < slices; l += 2 * d) {
            if (len[l + d]) crc[l] = len[l] ? gr::crc_combine(crc[l], crc[l + d], len[l + d], x2n) : crc[l + d];
            len[l] += len[l + d];
        }
    return n ? crc[0] : 0;
}
uint64_t gr_host_pack_key(const View* v, const uint8_t* s, uint32_t len) { gr::Dict d; to_dict(d, *v); return gr::pack_key(d, s, len); }
}
