// host build of the lane-per-block inflate (gpu_inflate2.hpp: plain single-threaded C++) for CPU fuzzing against zlib:
// phase 1 (literals in place + match tokens) and the sequential reference of phase 2 (gi2::resolve)
#include "../fastf_amd/csrc/gpu_inflate2.hpp"
#include <stdlib.h>
#include <vector>
// returns the decoder's code (0 ok); *n_tok_out = tokens written; out holds the inflated bytes after the resolve
extern "C" int gi2_host_inflate(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t isize, uint32_t* n_tok_out) {
    static thread_local gi2::Work* w = nullptr;
    if (!w) w = (gi2::Work*)malloc(sizeof(gi2::Work));
    // the reader takes 32-byte stages from the aligned floor of `in` to 192 bytes past its end: give it a padded, aligned copy
    std::vector<uint8_t> buf((size_t)in_len + 16 + 32 + 256, 0);
    uint8_t* base = buf.data();
    base += (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
    uint8_t* at = base + 16 + 5;                                   // (an odd offset inside a quad: the lead-byte path is exercised)
    memcpy(at, in, in_len);
    std::vector<uint32_t> tok(gi2::token_cap(isize));
    uint32_t n_tok = 0;
    const int rc = gi2::inflate_tokens(*w, at, in_len, out, isize, tok.data(), &n_tok);
    if (n_tok_out) *n_tok_out = n_tok;
    if (rc) return rc;
    return gi2::resolve(out, isize, tok.data(), n_tok);
}
