// host build of the device inflate (gpu_inflate.hpp with GI_HOST: one lane) for CPU fuzzing against zlib
#define GI_HOST 1
#include "../fastf_amd/csrc/gpu_inflate.hpp"
#include <stdlib.h>
extern "C" int gi_host_inflate(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t isize) {
    static thread_local gi::Work* w = nullptr;
    if (!w) w = (gi::Work*)malloc(sizeof(gi::Work));
    return gi::inflate_block(*w, in, in_len, out, isize);
}
