// distance / length statistics of the tokens in the BGZF blocks of a BAM (host build of the device decoder, instrumented)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#define GI_HOST
static unsigned long long g_lit, g_match, g_match_bytes, g_d[6], g_len_hist[9], g_blocks;
#define GI_STATS_MATCH(len, dist) do { g_match++; g_match_bytes += (len); g_d[(dist) <= 64 ? 0 : (dist) <= 4032 ? 1 : (dist) <= 8128 ? 2 : (dist) <= 16320 ? 3 : 4]++; g_len_hist[(len) < 4 ? 0 : (len) < 6 ? 1 : (len) < 9 ? 2 : (len) < 17 ? 3 : (len) < 33 ? 4 : (len) < 65 ? 5 : (len) < 129 ? 6 : 7]++; } while (0)
#define GI_STATS_LIT() do { g_lit++; } while (0)
#include "../fastf_amd/csrc/gpu_inflate.hpp"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); std::vector<uint8_t> raw(400u << 20); size_t n = fread(raw.data(), 1, raw.size(), f); fclose(f);
    size_t pos = 0; static gi::Work w; std::vector<uint8_t> out(70000);
    while (pos + 18 < n && g_blocks < 3000) {
        unsigned xlen = raw[pos + 10] | raw[pos + 11] << 8, bsize = (raw[pos + 16] | raw[pos + 17] << 8) + 1;
        if (pos + bsize > n) break;
        uint32_t isize; memcpy(&isize, &raw[pos + bsize - 4], 4);
        if (isize) { int rc = gi::inflate_block(w, &raw[pos + 12 + xlen], bsize - 12 - xlen - 8, out.data(), isize); if (rc) { printf("rc %d\n", rc); return 1; } }
        pos += bsize; g_blocks++;
    }
    printf("blocks %llu literals %llu matches %llu match bytes %llu (avg len %.1f) -> tokens per block %.0f, literal share of bytes %.2f\n", g_blocks, g_lit, g_match, g_match_bytes,
           (double)g_match_bytes / g_match, (double)(g_lit + g_match) / g_blocks, (double)g_lit / (g_lit + g_match_bytes));
    const char* dn[] = {"<=64", "<=4032", "<=8128", "<=16320", "<=32768"};
    for (int i = 0; i < 5; i++) printf("  dist %-8s %.3f\n", dn[i], (double)g_d[i] / g_match);
    const char* ln[] = {"3", "4-5", "6-8", "9-16", "17-32", "33-64", "65-128", ">128"};
    for (int i = 0; i < 8; i++) printf("  len %-7s %.3f\n", ln[i], (double)g_len_hist[i] / g_match);
}
