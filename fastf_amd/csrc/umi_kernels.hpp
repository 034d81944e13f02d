// umi_kernels.hpp — hand-written HIP kernels of the bam2db hot path for gfx950 (CDNA4).
//
//   mt_fill / draw_bits           the MT19937 draw stream (mt19937ar.c:105-140) continued on the device and packed to one
//                                 keep/drop decision bit per CB hit (bam2db_ds.c:385-390)
//   K1a probe_cells_kernel        CB probe, E2/E3 of the reference loop (bam2db_ds.c:366-380)
//   K1b filter_pack_kernel        E4..E12 (:385-435): depth decision by hit rank, xf, GX probe,
//                                 UB, packed 64-bit (cell, feature, umi) key
//   K1b filter_pack_stream_kernel the same per wave on 256-record units (a hand-pipelined loop: a unit's inputs are requested
//                                 during the unit before), keys into workgroup regions (+ shard_partition_kernel when the pass
//                                 is sharded)
//   K2  tile_count / row_scan / scatter: 64-bit LSD radix sort, 8-bit digits — stands in for SQLite's sorter
//                                 behind GROUP BY (bam2db_ds.c:480-483); the matrix path sorts (cell, feature) only
//   K3  reduce_hashed / giant_groups (+ the span scan) / rows_gather   [reduce_windows / span_scan for fully sorted keys]
//                                 COUNT(DISTINCT umi) GROUP BY cell, feature: the keys read once, rows into per-workgroup
//                                 regions, distinct UMIs of a group through an exact hash set in LDS
//   K3u (same, UMI_ROWS=true)     COUNT(*) GROUP BY cell, feature, umi (-u, :539-542), fully sorted keys
//
// All of it is integer indexing: wave64 ballots/mbcnt for ranking, LDS for the
// tile-local reorder, coalesced 8-byte streams to HBM.  No MFMA by design.
#pragma once
// Build knobs.  -DFASTF_<NAME> macros of this file tune or instrument (tile shapes, non-temporal hints, time stamps): the
// results stay bit-exact.  A knob that CHANGES results (elimination and timing experiments: "what does this kernel cost
// without its hash set") must be named FASTF_X_<NAME> and tested as `#if defined(FASTF_EXPERIMENT) && defined(FASTF_X_...)`:
// `make all` never defines FASTF_EXPERIMENT, a library built with it says so in fastf_version(), and tests/test_abi.py checks
// all three (round 3's FASTF_K3_NOHASH / _HASH_NOFLAG / _DEBUG went with the kernel modes they lived in).
#if !defined(FASTF_EXPERIMENT)
#  if defined(FASTF_X_ANY)
#    error "FASTF_X_* knobs need -DFASTF_EXPERIMENT (experiment builds only: tools/build_variant.sh)"
#  endif
#endif
#ifndef FASTF_K1_IPT
#define FASTF_K1_IPT 8
#endif
#ifndef FASTF_K1_THREADS
#define FASTF_K1_THREADS 512
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fastf {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int WAVE = 64;

// error bits accumulated in counters[3]
constexpr u64 ERR_DRAWS_SHORT  = 2;
constexpr u64 ERR_UMI_TOOLONG  = 4;
constexpr u64 ERR_KEYS_FULL    = 8;
constexpr u64 ERR_RUN_TOO_LONG = 16;       // group-only sort: a (cell, feature) group beyond what the hash path takes → the caller must sort fully

// meta bits (mirror of fastf_amd.h)
constexpr u32 META_XF_OK = 1, META_HAS_UB = 2, META_UMI_NONNULL = 4, META_UMI_TOOLONG = 8;
constexpr u32 META_LEN_SHIFT = 4, META_LEN_MASK = 0xF0;

// ------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() {
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of m strictly below this lane
__device__ __forceinline__ u32 rank_below(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// Inputs that are read exactly once are loaded with the non-temporal hint: a plain 4 GiB streaming read reaches
// 7.1 TB/s with it against 6.2 TB/s without (tools/hbm_probe2.hip).  FASTF_NT_* = 0 gives plain loads (A/B builds).
#ifndef FASTF_NT_ALL
#define FASTF_NT_ALL 1
#endif
#ifndef FASTF_NT_K1A
#define FASTF_NT_K1A FASTF_NT_ALL
#endif
#ifndef FASTF_NT_K1B
#define FASTF_NT_K1B FASTF_NT_ALL
#endif
#ifndef FASTF_NT_SORT
#define FASTF_NT_SORT FASTF_NT_ALL
#endif
#ifndef FASTF_NT_K3
#define FASTF_NT_K3 FASTF_NT_ALL
#endif
typedef u64 u64x2_t __attribute__((ext_vector_type(2)));
template <bool NT, class T> __device__ __forceinline__ T ld_once(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ ulonglong2 ld_once2(const u64* p) {          // 16-byte aligned pair
    if constexpr (NT) { const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p)); return make_ulonglong2(v.x, v.y); }
    else return *reinterpret_cast<const ulonglong2*>(p);
}
// (Non-temporal STORES were tried in K1a, K1b, the scatter and K3 and lost or changed nothing — scatter 0.127 -> 0.23 ms —
// so results are written with plain stores: profiles/r2_notes/ab_nontemporal_stores.txt.)
__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ u32 wave_sum32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// inclusive scan across the wave
__device__ __forceinline__ u32 wave_incl_scan32(u32 v, int lane) {
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        u32 t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    return v;
}
// a wave-uniform 64-bit value into scalar registers (readfirstlane returns int: the halves are widened unsigned)
__device__ __forceinline__ u64 uniform64(u64 v) {
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__global__ void clear_bits_kernel(u64* word, u64 mask) { atomicAnd(word, ~mask); }

// ------------------------------------------------------------------------------------
// MT19937 on the device (mt19937ar.c:105-140): the depth draws of the CB hits (bam2db_ds.c:385) are generated where they are
// consumed — no host loop (about a nanosecond per draw on one core), no bytes per hit over PCIe.  One workgroup continues
// the stream held in `state` (the 624 words + the read index, fastf_mt_t's layout: the host seeds and skips, rarely) by
// `count` draws and writes the tempered values to out[(first + i) & mask].  A block of 624 words is regenerated in three
// dependent sweeps — words 0..226 need old words only, 227..453 the new 0..226, 454..623 the new 227..396 (and word 623 the
// new word 0) — each sweep reading everything it needs before any of it is overwritten.
//
// What K1b wants from a draw is ONE BIT — keep the read iff genrand_real1() <= rate, i.e. draw < threshold
// (bam2db_ds.c:385-390, fastf_depth_threshold) — and that bit is what it reads: the DECISION STREAM, bit (r & 31) of 32-bit
// word ((r & mask) >> 5) for absolute hit rank r (configs[2]: 0.72 GB of draws per 200 M records become 23 MB, one word per
// lane and unit instead of four).  The generator's four waves are a serial critical path (three barriers per 624 draws):
// tempering, comparing and packing inside them, or beside them behind the same barriers, cost more than half its rate
// (0.65–1.08 against 2.5 G draws/s, profiles/r4_notes/mt_fill_decisions.txt).  So the generator writes plain words into a
// scratch buffer and draw_bits_kernel — as parallel as one likes — turns them into ring bits behind it on the same stream.
// ------------------------------------------------------------------------------------
constexpr u32 MT_N = 624, MT_M = 397;
__device__ __forceinline__ u32 mt_mix(u32 hi, u32 lo) {
    const u32 y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}
__device__ __forceinline__ u32 mt_temper(u32 y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}
__device__ __forceinline__ void mt_fill_body(u32 (&buf)[2][MT_N], u32* __restrict__ state, u32* __restrict__ out, u64 first, u64 count, u64 mask) {
    // two copies of the state: a block is regenerated FROM one INTO the other, so a sweep never overwrites what it still
    // reads: three barriers per block.  A word goes out (tempered, to its rank's ring slot) the moment it is computed — no
    // second pass over the block; only what is left of the block the kernel starts in is handed out by a loop of its own.
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < MT_N; i += 256) buf[0][i] = state[i];
    u32 idx = state[MT_N];
    u32 cur = 0;                                             // buf[cur] holds the current block
    __syncthreads();
    u64 done = 0;
    if (idx < MT_N && count) {                               // the rest of the block the stream stands in
        const u32 take = (u32)(count < (u64)(MT_N - idx) ? count : (u64)(MT_N - idx));
        for (u32 i = tid; i < take; i += 256) out[(first + i) & mask] = mt_temper(buf[0][idx + i]);
        idx += take; done = take;
    }
    while (done < count) {                                   // (uniform) whole blocks; the last one may be handed out in part
        const u32* o = buf[cur]; u32* n = buf[cur ^ 1];
        const u64 left = count - done;                       // words of this block that go out: k < left
        const u64 at = first + done;
        // sweep 1: k = 0..226   n[k] = o[k + 397] ^ mix(o[k], o[k + 1])
        if (tid < MT_N - MT_M) { const u32 v = o[tid + MT_M] ^ mt_mix(o[tid], o[tid + 1]); n[tid] = v; if (tid < left) out[(at + tid) & mask] = mt_temper(v); }
        __syncthreads();
        // sweep 2: k = 227..453  n[k] = n[k - 227] ^ mix(o[k], o[k + 1])
        const u32 k2 = tid + (MT_N - MT_M);
        if (tid < MT_N - MT_M) { const u32 v = n[k2 - (MT_N - MT_M)] ^ mt_mix(o[k2], o[k2 + 1]); n[k2] = v; if (k2 < left) out[(at + k2) & mask] = mt_temper(v); }
        __syncthreads();
        // sweep 3: k = 454..622 the same with n[k - 227] from sweep 2; k = 623: n[623] = n[396] ^ mix(o[623], n[0])
        const u32 k3 = tid + 2 * (MT_N - MT_M);
        if (k3 < MT_N) {
            const u32 v = k3 < MT_N - 1 ? n[k3 - (MT_N - MT_M)] ^ mt_mix(o[k3], o[k3 + 1]) : n[MT_M - 1] ^ mt_mix(o[MT_N - 1], n[0]);
            n[k3] = v;
            if (k3 < left) out[(at + k3) & mask] = mt_temper(v);
        }
        __syncthreads();
        cur ^= 1;
        idx = (u32)(left < (u64)MT_N ? left : (u64)MT_N);
        done += idx;
    }
    for (u32 i = tid; i < MT_N; i += 256) state[i] = buf[cur][i];
    if (tid == 0) state[MT_N] = idx;
}
__global__ __launch_bounds__(256) void mt_fill_kernel(u32* __restrict__ state, u32* __restrict__ out, u64 first, u64 count, u64 mask) {
    __shared__ u32 buf[2][MT_N];
    mt_fill_body(buf, state, out, first, count, mask);
}

// ------------------------------------------------------------------------------------
// The same stream from MANY workgroups: jump-ahead (mt_jump.c has the algebra).  The stream's state is linear over GF(2), so
// the state J draws on is g(F) applied to the state now, with g = x^J mod the characteristic polynomial — a constant of the
// generator, no seed in it.  Sub-stream j of a call starts J = 624 x 256 draws behind sub-stream j - 1:
//   mt_seq_kernel         one workgroup per SOURCE: its array (a block boundary: all 624 words handed out) is continued as a flat
//                         word sequence X_{t+624} = X_{t+397} ^ mix(X_t, X_{t+1}) for 19 976 words more, in LDS (sweeps of 227
//                         words, a barrier each), and written out (82 KB)
//   mt_conv_kernel        two workgroups per JUMP, half the polynomial each: the array J words on is the XOR of the windows
//                         X[k .. k+623] over the set coefficients k of g — a lane keeps ten outputs, the polynomial's bits walk in
//                         scalar registers, eight waves split the half's words, the halves meet in the destination by atomic XOR
//                         Two rounds of these seat up to R^2 = 1024 sub-streams: sub-streams R, 2R, .. straight from the stream's
//                         state (polynomials x^(i R J)), then the R - 1 behind every one of those (x^(i J)) — every workgroup
//                         of a launch depends on the launch before only.  (Round 5 doubled: sub-streams 2^l apart from those
//                         2^(l+1) apart, nine dependent launches of 105 us for 283 sub-streams, 0.94 of the 1.37 ms a job's draws cost.)
//   mt_fill_multi_kernel  workgroup j continues sub-stream j by its share of the call's draws (mt_fill_body above).
// 90 M draws (one configs[2] job): 566 sub-streams; sequences: 1 + 18 workgroups, convolutions: 34 + 1 096 half-jumps; one fill launch.
// ------------------------------------------------------------------------------------
constexpr u32 MT_SUB_DRAWS = 624u * 256u, MT_POLY_WORDS = 312, MT_DEG = 19937, MT_STATE_WORDS = MT_N + 1;
constexpr u32 MT_JUMP_SWEEPS = 88, MT_SEQ_WORDS = MT_N + MT_JUMP_SWEEPS * (MT_N - MT_M);   // 20 600 words >= 624 + 19 936: every window of the convolution
static_assert(MT_SEQ_WORDS >= MT_N + MT_DEG + 16, "the sequence covers the windows of every coefficient (and the sixteen outputs beyond 624 a lane reads along)");
constexpr u32 MT_SEQ_STRIDE = 20736;                             // words between two sequences in memory (a half's bulk load reads a few words past 20 600: never used)
constexpr u32 MT_JUMP_R = 32;
// Round 6, second step: the SEQUENCE of a source is computed once (mt_seq_kernel: one workgroup per source, 88 barrier sweeps in
// LDS, then out to memory — 82 KB) instead of by every jump that starts there (31 of them per source), and a jump's CONVOLUTION is
// split in two halves of the polynomial (mt_conv_kernel): a half needs 10 624 words of the sequence, 63 KB of LDS with its eight
// waves' partial arrays — two workgroups per CU where the whole jump (123 KB) allowed one — and XORs its 624 words into the
// zeroed destination with atomics.  274 jumps = 548 halves over 512 places instead of 274 workgroups over 256.
constexpr u32 MT_HALF_PW = MT_POLY_WORDS / 2, MT_HALF_X = MT_HALF_PW * 64u + 640u, MT_CONV_WAVES = 8;    // 156 polynomial words, 10 624 sequence words per half
constexpr u32 MT_CONV_LDS_WORDS = MT_HALF_X + 16u + MT_CONV_WAVES * 640u;
static_assert(MT_POLY_WORDS % 2 == 0 && MT_HALF_PW * 64u + MT_HALF_X <= MT_SEQ_STRIDE, "a half's bulk load stays inside its sequence's stride");
// workgroup b: the sequence of sub-stream b * src_stride (a block boundary: all 624 words handed out) continued as a flat word
// sequence X_{t+624} = X_{t+397} ^ mix(X_t, X_{t+1}) for 19 976 words more, to seq + b * MT_SEQ_STRIDE
__global__ __launch_bounds__(256) void mt_seq_kernel(const u32* __restrict__ states, u32* __restrict__ seq, u32 src_stride, u32 n_sub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mtj_smem[];
    u32* const X = reinterpret_cast<u32*>(mtj_smem);
    const u32 src = blockIdx.x * src_stride;
    if (src >= n_sub) return;                                  // (block-uniform)
    const u32 tid = threadIdx.x;
    const u32* const a = states + (u64)src * MT_STATE_WORDS;
    for (u32 i = tid; i < MT_N; i += 256) X[i] = a[i];
    __syncthreads();
    // words t + 624 for t in [base, base + 227) need nothing beyond X[base + 623]
    for (u32 sw = 0; sw < MT_JUMP_SWEEPS; ++sw) {
        if (tid < MT_N - MT_M) { const u32 t = sw * (MT_N - MT_M) + tid; X[t + MT_N] = X[t + MT_M] ^ mt_mix(X[t], X[t + 1]); }
        __syncthreads();
    }
    u32* const out = seq + (u64)blockIdx.x * MT_SEQ_STRIDE;
    for (u32 i = tid; i < MT_SEQ_WORDS; i += 256) out[i] = X[i];
}
// one 8-byte LDS read that the compiler cannot pair with a neighbour (its 32-bit LDS address passes through an empty asm)
__device__ __forceinline__ u64 lds_read_u64_alone(const u64* p) {
    typedef __attribute__((address_space(3))) const u64 lds_u64;
    u32 a = (u32)(uintptr_t)(lds_u64*)p;
    asm volatile("" : "+v"(a));
    return *(lds_u64*)(uintptr_t)a;
}
// One HALF of a jump.  coarse: job g seats sub-stream (g + 1) R from sub-stream 0 (sequence 0, polynomial x^((g + 1) R J): the
// table's second half); else job g = c (R - 1) + (i - 1) seats sub-stream c R + i from sub-stream c R (sequence c, polynomial x^(i J)).
// W_J[i] = XOR over the set coefficients k of X[k + i]: wave w takes the half's polynomial words w, w + 8, .., every lane ten of the 624
// outputs (i = lane + 64 q), the eight partial arrays meet in LDS, and the half's 624 words go into the destination by atomic XOR
// (the destinations were zeroed; the other half adds its own).
__global__ __launch_bounds__(512, 2) void mt_conv_kernel(u32* __restrict__ states, const u32* __restrict__ seq, const u64* __restrict__ polys, u32 coarse, u32 n_sub) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mtj_smem[];
    u32* const X = reinterpret_cast<u32*>(mtj_smem);
    const u32 job = blockIdx.x >> 1, half = blockIdx.x & 1u;
    const u32 c = coarse ? 0u : job / (MT_JUMP_R - 1u), i = coarse ? job + 1u : job % (MT_JUMP_R - 1u) + 1u;
    const u32 dst = coarse ? i * MT_JUMP_R : c * MT_JUMP_R + i;
    if (dst >= n_sub) return;                                  // (block-uniform)
    const u64* const poly = polys + (u64)((coarse ? MT_JUMP_R - 1u : 0u) + i - 1u) * MT_POLY_WORDS + (u64)half * MT_HALF_PW;
    const u32 tid = threadIdx.x;
    {   // this half's stretch of the sequence: 16-byte loads (the stride and the half's offset are multiples of four words)
        const uint4* const g = reinterpret_cast<const uint4*>(seq + (u64)c * MT_SEQ_STRIDE + (u64)half * MT_HALF_PW * 64u);
        uint4* const l = reinterpret_cast<uint4*>(X);
        for (u32 q = tid; q < MT_HALF_X / 4u; q += 512) l[q] = g[q];
    }
    __syncthreads();
    u32* const part = X + MT_HALF_X + 16u;                        // [8][640]
    {
        // A lane keeps five PAIRS of outputs and reads them with ds_read_b64 — 256 bytes a clock from the LDS where ds_read_b32
        // moves 128 (MI355X_MICROARCH.md, LDS) — which wants 8-byte aligned addresses: for an even coefficient k the pairs are the
        // outputs (2m, 2m + 1), for an odd one (2m - 1, 2m), so that the first word read, X[k + i], always has an even index
        // (X[k - 1 + 2m] for odd k: pair 0's first word belongs to no output).  Two sets of accumulators; the odd set is moved
        // down by one output when the wave is done (a DPP shift and four readlanes).
        const u32 lane = tid & 63u, w = tid >> 6;
        u64 ae[5], ao[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { ae[q] = 0; ao[q] = 0; }
        const u64* const X64 = reinterpret_cast<const u64*>(X);
        // the even coefficients of a polynomial word, then the odd ones, FOUR coefficients a turn: twenty 8-byte reads in flight per
        // lane (a lone half-jump took 56 us at two a turn — 215 cycles per coefficient and wave: the wave waited for each turn's reads)
        auto xor3 = [](u64 a, u64 b, u64 c) -> u64 {              // a ^ b ^ c in one instruction per 32 bits (v_bitop3, truth table 0x96)
            const u32 lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0x96);
            const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0x96);
            return (u64)lo | ((u64)hi << 32);
        };
        auto walk = [&](u64 bits, const u64* const base, u64 (&acc)[5]) {
            while (bits) {                                         // (uniform inside the wave)
                const u32 n = (u32)__builtin_popcountll(bits);
                if (n >= 4) {
                    const u32 k0 = (u32)__builtin_ctzll(bits); bits &= bits - 1;
                    const u32 k1 = (u32)__builtin_ctzll(bits); bits &= bits - 1;
                    const u32 k2 = (u32)__builtin_ctzll(bits); bits &= bits - 1;
                    const u32 k3 = (u32)__builtin_ctzll(bits); bits &= bits - 1;
                    const u64 *x0 = base + (k0 >> 1), *x1 = base + (k1 >> 1), *x2 = base + (k2 >> 1), *x3 = base + (k3 >> 1);
                    u64 t0[5], t1[5], t2[5], t3[5];
#pragma unroll
                    // (every address made opaque: separate ds_read_b64 — 256 bytes a clock — instead of the ds_read2st64_b64
                    //  pairs the compiler merges neighbours into, which the LDS serves at 128)
                    for (int q = 0; q < 5; ++q) {
                        t0[q] = lds_read_u64_alone(x0 + 64 * q); t1[q] = lds_read_u64_alone(x1 + 64 * q);
                        t2[q] = lds_read_u64_alone(x2 + 64 * q); t3[q] = lds_read_u64_alone(x3 + 64 * q);
                    }
#pragma unroll
                    for (int q = 0; q < 5; ++q) acc[q] = xor3(xor3(acc[q], t0[q], t1[q]), t2[q], t3[q]);
                } else {
                    const u32 k0 = (u32)__builtin_ctzll(bits); bits &= bits - 1;
                    const u64* const x0 = base + (k0 >> 1);
#pragma unroll
                    for (int q = 0; q < 5; ++q) acc[q] ^= x0[64 * q];
                }
            }
        };
        for (u32 pw = w; pw < MT_HALF_PW; pw += MT_CONV_WAVES) {
            const u64 bits = uniform64(poly[pw]);
            const u64* const base = X64 + (pw << 5) + lane;          // X[64 pw + 2m ..]: even k reads X[k + 2m ..], odd k X[k - 1 + 2m ..]
            walk(bits & 0x5555555555555555ull, base, ae);
            walk(bits & 0xAAAAAAAAAAAAAAAAull, base, ao);
        }
        // output 2m = ae[m].lo ^ ao[m].hi, output 2m + 1 = ae[m].hi ^ ao[m + 1].lo  (pair m = lane + 64 q; beyond pair 319: 0)
        u32 carry = 0;
#pragma unroll
        for (int q = 4; q >= 0; --q) {
            const u32 o_lo = (u32)ao[q];
            // lane l: ao.lo of pair m + 1 (lane 63: pair 0 of the turn above) — DPP wave_shl:1
            const u32 from_above = (u32)__builtin_amdgcn_update_dpp((int)carry, (int)o_lo, 0x130, 0xF, 0xF, false);
            carry = (u32)__builtin_amdgcn_readlane((int)o_lo, 0);
            const u32 v_lo = (u32)ae[q] ^ (u32)(ao[q] >> 32), v_hi = (u32)(ae[q] >> 32) ^ from_above;
            *reinterpret_cast<u64*>(part + w * 640u + 2u * (lane + 64u * q)) = (u64)v_lo | ((u64)v_hi << 32);
        }
    }
    __syncthreads();
    for (u32 t = tid; t < MT_N; t += 512) {
        u32 v = 0;
#pragma unroll
        for (u32 w = 0; w < MT_CONV_WAVES; ++w) v ^= part[w * 640u + t];
        if (v) atomicXor(&states[(u64)dst * MT_STATE_WORDS + t], v);
    }
    if (tid == 0 && half == 0) states[(u64)dst * MT_STATE_WORDS + MT_N] = MT_N;   // a block boundary: the next draw regenerates
}
// sub-stream j: draws [j * MT_SUB_DRAWS, ...) of the call's `count`, to out[(first + that) & mask]
__global__ __launch_bounds__(256) void mt_fill_multi_kernel(u32* __restrict__ states, u32* __restrict__ out, u64 first, u64 count, u64 mask) {
    __shared__ u32 buf[2][MT_N];
    const u64 off = (u64)blockIdx.x * MT_SUB_DRAWS;
    if (off >= count) return;
    const u64 n = count - off < (u64)MT_SUB_DRAWS ? count - off : (u64)MT_SUB_DRAWS;
    mt_fill_body(buf, states + (u64)blockIdx.x * MT_STATE_WORDS, out, first + off, n, mask);
}

// draws -> decisions: bit (first + i) of the ring (ring_mask = its size in bits - 1, a power of two >= 64 — or ~0: a linear
// array) = draws[i] < threshold, i < n.  A wave owns one 64-bit word of the ring per round: its ballot IS the word.  The word
// `first` stands in keeps its bits below `first` (the launch before wrote them: same stream), the word the range ends in is
// zero above the end (the next launch completes it).  Caller-supplied draws, the host's own stream (FASTF_HOST_DRAWS=1: packed
// on the host instead) and mt_fill_kernel's words all become decisions here.
__global__ __launch_bounds__(256) void draw_bits_kernel(const u32* __restrict__ draws, u64 n, u64 threshold, u32* __restrict__ bits, u64 first, u64 ring_mask) {
    const int lane = lane_id();
    u64* const bits64 = reinterpret_cast<u64*>(bits);
    const u64 wmask = ring_mask >> 6, w_first = first >> 6, end = first + n;
    const u64 n_words = n ? ((end + 63) >> 6) - w_first : 0;
    const u64 waves = (u64)gridDim.x * (256 / WAVE), w0 = (u64)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    // four ring words per wave and turn: four loads in flight per lane (one per turn ran at 2.6 TB/s: a lane waited for each draw)
    constexpr u32 U = 4;
    for (u64 c0 = w0 * U; c0 < n_words; c0 += waves * U) {
        u32 d[U]; bool in[U];
#pragma unroll
        for (u32 u = 0; u < U; ++u) {
            const u64 r = ((w_first + c0 + u) << 6) + (u64)lane;  // this lane's rank in word c0 + u
            in[u] = c0 + u < n_words && r >= first && r < end;
            d[u] = draws[in[u] ? r - first : 0];
        }
#pragma unroll
        for (u32 u = 0; u < U; ++u) {
            const u64 c = c0 + u;
            const u64 m = __ballot(in[u] && (u64)d[u] < threshold);
            if (lane == 0 && c < n_words) {
                u64* const w = bits64 + ((w_first + c) & wmask);
                const u32 low = c == 0 ? (u32)(first & 63u) : 0u;                // bits of this word that belong to earlier ranks
                *w = low ? (*w & ((1ull << low) - 1ull)) | m : m;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// device-resident open-addressed table: 16-byte slots {key lo, key hi, value, 0},
// key 0 = empty, linear probing, load <= 0.5.  A few hundred KB: lives in L2.
// ------------------------------------------------------------------------------------
struct Table { const uint4* slots; u32 mask; };

// slot hash: two 32-bit multiplies (integer multiplies are quarter-rate; the murmur finaliser costs eight) —
// the key halves carry the entropy (bases / gene number) in different bits, the sum of the two products mixes them
__device__ __forceinline__ u32 slot_hash(u64 key) {
    const u32 h = (u32)key * 0x9E3779B1u + (u32)(key >> 32) * 0x85EBCA77u;
    return h ^ (h >> 15);
}

__device__ __forceinline__ u32 table_probe(const Table t, u64 key) {
    if (key == 0) return 0;
    u32 h = slot_hash(key) & t.mask;
    for (u32 i = 0; i <= t.mask; ++i) {
        const uint4 s = t.slots[h];
        const u64 k = ((u64)s.y << 32) | s.x;
        if (k == key) return s.z;
        if (k == 0) return 0;
        h = (h + 1) & t.mask;
    }
    return 0;
}

// ------------------------------------------------------------------------------------
// packed sort key layout:  [cell][feature][nonnull:1][umi: 2*Lmax bits][len]
// ------------------------------------------------------------------------------------
struct KeyLayout {
    u32 umi_bits;      // 2 * umi_max_bases
    u32 len_bits;      // 2 (Lmax <= 12) or 3
    u32 feat_shift;    // = 1 + umi_bits + len_bits
    u32 cell_shift;    // = feat_shift + feature_bits
    u32 total_bits;
    u32 umi_max_bytes;
};

__device__ __forceinline__ u64 make_key(const KeyLayout L, u32 cell, u32 feat, u32 umi, u32 meta) {
    u64 k = ((u64)cell << L.cell_shift) | ((u64)feat << L.feat_shift);
    if (meta & META_UMI_NONNULL) {
        const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
        k |= (1ULL << (L.umi_bits + L.len_bits)) | ((u64)(umi >> (32 - L.umi_bits)) << L.len_bits) | len;
    }
    return k;
}

// the same two, straight-line (the streaming K1b: an exec-mask branch per record condition costs more than the few
// instructions it skips)
__device__ __forceinline__ u64 make_key_flat(const KeyLayout L, u32 cell, u32 feat, u32 umi, u32 meta) {
    const u32 nn = (meta / META_UMI_NONNULL) & 1u, len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    const u64 low = ((u64)nn << (L.umi_bits + L.len_bits)) | ((u64)(umi >> (32 - L.umi_bits)) << L.len_bits) | len;
    return ((u64)cell << L.cell_shift) | ((u64)feat << L.feat_shift) | (low & (0ull - (u64)nn));
}
__device__ __forceinline__ bool umi_overflows_flat(const KeyLayout L, u32 umi, u32 meta) {
    const u32 tail = L.umi_bits < 32 ? (1u << (32 - L.umi_bits)) - 1u : 0u;        // (scalar) the bits of umi the field cannot hold
    const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    return ((meta & META_UMI_TOOLONG) != 0) | (((meta & META_UMI_NONNULL) != 0) & ((len > L.umi_max_bytes) | ((umi & tail) != 0)));
}

// the rest of a wide key (everything below the group word) from a UMI of up to 32 bases in 64 bits (first base on top)
__device__ __forceinline__ u64 make_val64(const KeyLayout L, u64 umi, u32 meta) {
    if (!(meta & META_UMI_NONNULL)) return 0;
    const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    return (1ULL << (L.umi_bits + L.len_bits)) | ((umi >> (64 - L.umi_bits)) << L.len_bits) | len;
}
// UMIs of up to 32 bases (engines with umi_max_bases > 24).  The rest of the key, [bases: 64 bits][blob bytes: 4 bits] under the
// non-NULL flag, is 69 bits and the reduce holds 52 exactly: the low 51 under the flag are the value, the top WIDE_SUB_BITS (the
// first bases) go into the sorted word below the feature — keys that differ there are different UMIs whatever the rest, so the
// distinct count of a (cell, feature) is the sum over its sub-groups (summed on the host: umi_engine.hip merge_sub_rows)
constexpr u32 WIDE_SUB_BITS = 17, WIDE_SUB_VAL_BITS = 51, WIDE_SUB_LEN_BITS = 4;
__device__ __forceinline__ u64 make_val_sub(u64 umi, u32 meta) {
    if (!(meta & META_UMI_NONNULL)) return 0;
    const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    return (1ULL << WIDE_SUB_VAL_BITS) | (((umi << WIDE_SUB_LEN_BITS) | len) & ((1ULL << WIDE_SUB_VAL_BITS) - 1));
}
// the sub-group: bits 68..52 of [bases][blob bytes] = the top WIDE_SUB_BITS of the bases = of their first 32-bit word
__device__ __forceinline__ u32 wide_sub_of(u32 umi_first_word, u32 meta) {
    static_assert(WIDE_SUB_VAL_BITS - WIDE_SUB_LEN_BITS + WIDE_SUB_BITS == 64, "the sub-group and the value's bases make up the 64 bits of bases");
    return (meta & META_UMI_NONNULL) ? umi_first_word >> (32 - WIDE_SUB_BITS) : 0u;
}
__device__ __forceinline__ bool umi_overflows_sub(u32 meta) {
    return (meta & META_UMI_TOOLONG) || ((meta & META_UMI_NONNULL) && ((meta & META_LEN_MASK) >> META_LEN_SHIFT) > 8u);
}
__device__ __forceinline__ bool umi_overflows64(const KeyLayout L, u64 umi, u32 meta) {
    if (meta & META_UMI_TOOLONG) return true;
    if (!(meta & META_UMI_NONNULL)) return false;
    const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    return len > L.umi_max_bytes || (L.umi_bits < 64 && (umi << L.umi_bits) != 0);
}

// a UMI the chosen layout cannot hold exactly (longer than umi_max_bases with non-zero tail,
// or more blob bytes than the length field encodes)
__device__ __forceinline__ bool umi_overflows(const KeyLayout L, u32 umi, u32 meta) {
    if (meta & META_UMI_TOOLONG) return true;
    if (!(meta & META_UMI_NONNULL)) return false;
    const u32 len = (meta & META_LEN_MASK) >> META_LEN_SHIFT;
    return len > L.umi_max_bytes || (L.umi_bits < 32 && (umi << L.umi_bits) != 0);
}

// ------------------------------------------------------------------------------------
// batched probe: N independent first-slot loads in flight per lane (the tables live in L2,
// so the cost is latency, not bytes); the rare collision continues in a scalar loop.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ u32 table_probe_from(const Table t, u64 key, u32 h) {
    for (u32 i = 0; i <= t.mask; ++i) {
        const uint4 s = t.slots[h];
        const u64 k = ((u64)s.y << 32) | s.x;
        if (k == key) return s.z;
        if (k == 0) return 0;
        h = (h + 1) & t.mask;
    }
    return 0;
}

typedef __attribute__((__vector_size__(4 * sizeof(int)))) int i32x4_t;

template <int N, int AUX = 0>
__device__ __forceinline__ void table_probe_batch(const Table t, const u64 (&key)[N], u32 (&val)[N]) {
    u32 h[N]; uint4 s[N];
    // wave-uniform buffer descriptor over the table: 32-bit offsets, cache policy in AUX
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)t.slots, 0, (t.mask + 1u) * 16u, 0x00020000);
#pragma unroll
    for (int j = 0; j < N; ++j) h[j] = key[j] ? (slot_hash(key[j]) & t.mask) : 0u;   // dead lanes share slot 0
#pragma unroll
    for (int j = 0; j < N; ++j) {                              // unconditional: N loads back to back, one wait
        const i32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(h[j] * 16u), 0, AUX);
        s[j] = make_uint4((u32)v[0], (u32)v[1], (u32)v[2], (u32)v[3]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const u64 k = ((u64)s[j].y << 32) | s[j].x;
        u32 v = 0;
        if (key[j] != 0) {
            if (k == key[j]) v = s[j].z;
            else if (k != 0) v = table_probe_from(t, key[j], (h[j] + 1) & t.mask);
        }
        val[j] = v;
    }
}

// ------------------------------------------------------------------------------------
// exclusive scan of per-tile counts (one workgroup; T is a few thousand entries, four per thread).
// The input is cleared once read: K1a accumulates its hit counts with atomics and needs zeros there, so the buffer
// returns to all-zero after every use and no memset is launched per step.
// ------------------------------------------------------------------------------------
// running / base_out (optional): a device-side running total across launches — *base_out = *running before this
// launch's total is added (the streaming push path: hit-rank base of a chunk without a host round trip).
// PER = entries per thread and round (a thread owns PER consecutive entries: 16; 64 in one round measured slower, each
// of a thread's 48 memory instructions touches 64 different lines).  One workgroup scans 16 384 entries per round.  Longer
// arrays (200 M records: 48 829 tiles, three rounds, 56 us on one CU) are cut into one chunk per workgroup — blk_tot !=
// nullptr: workgroup b scans chunk b and leaves its total — and scan_fix_kernel adds the totals of the chunks in front.
template <u32 PER>
__global__ __launch_bounds__(1024) void scan_tiles_kernel(u32* __restrict__ in, u64* __restrict__ out, u32 T,
                                                          u64* __restrict__ total_out, u64* __restrict__ running = nullptr,
                                                          u64* __restrict__ base_out = nullptr, u64* __restrict__ blk_tot = nullptr) {
    __shared__ u32 s_w[16];
    __shared__ u64 s_carry;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const u32 t_first = blk_tot ? blockIdx.x * PER * 1024u : 0u, t_end = blk_tot ? (t_first + PER * 1024u < T ? t_first + PER * 1024u : T) : T;
    for (u32 t0 = t_first; t0 < t_end; t0 += PER * 1024) {
        const u32 t = t0 + PER * threadIdx.x;
        u32 v[PER];
        if (t + PER <= T) {                              // (in is 16-byte aligned: hipMalloc; t is a multiple of PER)
#pragma unroll
            for (u32 q = 0; q < PER / 4; ++q) {
                const uint4 x = reinterpret_cast<const uint4*>(in + t)[q];
                v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
                reinterpret_cast<uint4*>(in + t)[q] = make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (u32 k = 0; k < PER; ++k) { v[k] = t + k < T ? in[t + k] : 0; if (t + k < T) in[t + k] = 0; }
        }
        u32 sum = 0;
#pragma unroll
        for (u32 k = 0; k < PER; ++k) sum += v[k];
        const u32 inc = wave_incl_scan32(sum, lane);
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        u64 off = s_carry;
        for (int i = 0; i < w; ++i) off += s_w[i];
        u64 e = off + inc - sum;
        if (t + PER <= T) {
#pragma unroll
            for (u32 q = 0; q < PER / 2; ++q) {
                reinterpret_cast<ulonglong2*>(out + t)[q] = make_ulonglong2(e, e + v[2 * q]);
                e += (u64)v[2 * q] + v[2 * q + 1];
            }
        } else {
#pragma unroll
            for (u32 k = 0; k < PER; ++k) { if (t + k < T) out[t + k] = e; e += v[k]; }
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (blk_tot) { if (threadIdx.x == 0) blk_tot[blockIdx.x] = s_carry; return; }
    if (threadIdx.x == 0 && total_out) *total_out = s_carry;
    if (threadIdx.x == 0 && running) { const u64 r = *running; *base_out = r; *running = r + s_carry; }
}

// second half of the chunked scan: chunk b's entries get the totals of the chunks in front; workgroup 0 hands out the total
template <u32 PER>
__global__ __launch_bounds__(1024) void scan_fix_kernel(u64* __restrict__ out, u32 T, const u64* __restrict__ blk_tot, u32 n_blk,
                                                        u64* __restrict__ total_out, u64* __restrict__ running, u64* __restrict__ base_out) {
    u64 off = 0, total = 0;
    for (u32 i = 0; i < n_blk; ++i) { const u64 v = blk_tot[i]; if (i < blockIdx.x) off += v; total += v; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (total_out) *total_out = total;
        if (running) { const u64 r = *running; *base_out = r; *running = r + total; }
    }
    if (blockIdx.x == 0 || off == 0) return;
    const u32 t_first = blockIdx.x * PER * 1024u;
    for (u32 k = threadIdx.x; k < PER * 1024u && t_first + k < T; k += 1024) out[t_first + k] += off;
}

// ------------------------------------------------------------------------------------
// K1a: CB probe (E2/E3, bam2db_ds.c:366-380) → cell index per record + hits per tile
// K1b: depth draw by hit rank, xf, feature probe, UB, key pack (E4..E12, :385-435)
// The hit rank of a record (= its position in the MT draw stream) is
//   tile_base[tile] (scan of K1a's counts) + rank inside the tile (ballots, record order).
// ------------------------------------------------------------------------------------
constexpr int K1_THREADS = FASTF_K1_THREADS, K1_IPT = FASTF_K1_IPT, K1_TILE = K1_THREADS * K1_IPT, K1_WAVES = K1_THREADS / WAVE;
// K1a lays a tile out wave by wave: wave w of a 512-thread group owns the records [512 w, 512 w + 512) of its tile
static_assert(K1_THREADS == 512 && K1_IPT == 8, "K1a layout: 8 waves x 8 items x 64 lanes per tile");

// the cell-index scratch between K1a and K1b holds u16 entries when every cell index fits (n_cells <= 65535): 2 instead of
// 4 bytes written and read back per record
// BLOCKED record layout (the engine's own device staging; the SoA of the ABI stays what callers hand over): per 256-record
// unit ONE contiguous run   gx u64[256] | umi u32[256] | meta u32[256] | cell scratch (u16[256] or u32[256])
// = 4608 (5120) bytes.  K1a fills the scratch slice of a unit, K1b then reads a unit as one stream instead of four distant
// ones: tools/hbm_probe_streams.hip measures the same 18 bytes per record at 6.3 TB/s from such runs against 5.2 TB/s from
// four arrays (profiles/r4_notes/hbm_probe_blocked_layout.txt; a run with dead bytes in it — the cb slice kept inside —
// loses the gain, which is why cb stays an array of its own).
constexpr u32 BLK_RECS = 256, BLK_GX = 0, BLK_UMI = 2048, BLK_META = 3072, BLK_CELL = 4096;
__host__ __device__ __forceinline__ u32 blk_run_bytes(bool c16) { return BLK_CELL + (c16 ? 2u : 4u) * BLK_RECS; }
// where K1a leaves the cell index of record idx: a plain array (run == 0) or the scratch slices of a blocked buffer
struct CellOut { void* p; u32 run; };
__device__ __forceinline__ void put_cell(const CellOut o, bool c16, u64 idx, u32 v) {
    unsigned char* at = reinterpret_cast<unsigned char*>(o.p);
    u64 i = idx;
    if (o.run) { at += (idx / BLK_RECS) * o.run + BLK_CELL; i = idx % BLK_RECS; }
    if (c16) reinterpret_cast<unsigned short*>(at)[i] = (unsigned short)v; else reinterpret_cast<u32*>(at)[i] = v;
}
__device__ __forceinline__ u32 get_cell(const void* __restrict__ in, bool c16, u64 idx) {
    return c16 ? (u32)reinterpret_cast<const unsigned short*>(in)[idx] : reinterpret_cast<const u32*>(in)[idx];
}

__device__ __forceinline__ u32 get_cell_once(const void* __restrict__ in, bool c16, u64 idx) {      // streaming K1b: read once
    return c16 ? (u32)ld_once<FASTF_NT_K1B != 0>(reinterpret_cast<const unsigned short*>(in) + idx)
               : ld_once<FASTF_NT_K1B != 0>(reinterpret_cast<const u32*>(in) + idx);
}

__device__ __forceinline__ u32 shard_of(u32 cell, u32 n_shards) {
    return (u32)((mix64((u64)cell) >> 32) % n_shards);
}

// half_hits[16 t + h]: hits among records [256 h, 256 h + 256) of tile t — the hit-rank base of a 256-record unit is
// tile_base[t] + the halves in front of it, which lets every WAVE of K1b work on its own (filter_pack_stream_kernel)
template <int AUX>
__global__ __launch_bounds__(K1_THREADS) void probe_cells_kernel(const u64* __restrict__ cb, u64 n, Table cells,
                                                                 const CellOut cell_out, bool c16, u32* __restrict__ tile_hits,
                                                                 u32* __restrict__ half_hits) {
    __shared__ u32 s_w[K1_WAVES];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u64 base = (u64)blockIdx.x * K1_TILE + (u64)w * (K1_IPT * WAVE);
    u64 key[K1_IPT]; u32 cell[K1_IPT];
#pragma unroll
    for (int j = 0; j < K1_IPT; ++j) {
        const u64 idx = base + (u64)j * WAVE + lane;
        key[j] = idx < n ? cb[idx] : 0;
    }
    table_probe_batch<K1_IPT, AUX>(cells, key, cell);
    u32 hits = 0, hits_lo = 0;
#pragma unroll
    for (int j = 0; j < K1_IPT; ++j) {
        const u64 idx = base + (u64)j * WAVE + lane;
        if (idx < n) put_cell(cell_out, c16, idx, cell[j]);
        hits += (u32)__popcll(__ballot(cell[j] != 0));
        if (j == K1_IPT / 2 - 1) hits_lo = hits;
    }
    if (lane == 0) {
        s_w[w] = hits;
        half_hits[((u64)blockIdx.x * K1_WAVES + w) * 2] = hits_lo; half_hits[((u64)blockIdx.x * K1_WAVES + w) * 2 + 1] = hits - hits_lo;
    }
    __syncthreads();
    if (tid == 0) {
        u32 t = 0;
        for (int i = 0; i < K1_WAVES; ++i) t += s_w[i];
        tile_hits[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------
// LDS-resident tables (the fast path when the lists allow it; the L2 tables above stay the
// general path).  Exactness rests on the key packing, not on hashing:
//   cells: every registered barcode is of the DNA form with the same length (<= 16 bases) and the
//          same "-N" suffix, so key bits 63:49 are one constant ("family") and bits 47:16 hold the
//          bases: a 32-bit code identifies the barcode.  The table is hash-and-displace with
//          quotienting: h = a BIJECTIVE 32-bit mix of the code, split into hi (S bits) and lo
//          (32 - S bits); slot = (hi + disp[bucket(lo)]) mod 2^S, and the slot keeps lo next to
//          the cell index.  A probe that finds its lo in its slot has the same bucket, hence the
//          same displacement, hence the same hi — the whole code is verified by the 32 - S stored
//          bits, so a slot is 4 bytes: (lo << S) | index, 0 = empty.  32 767 cells fit 128 KB of
//          slots + 16 KB of displacements (one 1024-thread workgroup per CU), 16 383 cells 72 KB
//          (two per CU).  A tag key outside the family cannot be a registered barcode.
//   genes: registered feature ids of one ID-form family (<prefix><n digits>, key bits 63:44) map
//          through a bitmap over [vmin, vmax] + per-word rank + permutation to the feature index:
//          exact hit or exact miss with three LDS reads; keys of other families use the L2 table.
// ------------------------------------------------------------------------------------
struct CellLds { const u32* image; u32 slot_bits; u32 bucket_mask; u32 family; u32 bytes; u32 seed; };
                                                                      // image: u32 slot[1 << slot_bits] | u16 disp[bucket_mask + 1]
struct GeneLds { const u32* image; u32 words; u32 n_perm; u32 family; u64 vmin; u64 range; u32 bytes; u32 direct; };
                                                                      // image: u32 bitmap[words] | u16 rank[words] (padded) | u16 perm[n_perm]
                                                                      // direct (dense id range): u16 index[range], 0 = not a listed id

// bijection on 32 bits (odd multipliers and xor-shifts are invertible): distinct codes have distinct mixes
__host__ __device__ __forceinline__ u32 cell_mix(u32 code, u32 seed) {
    u32 h = (code ^ seed) * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return h;
}


// K1a with a miss filter (barcode lists too large for the LDS perfect hash): a bit set over a second hash of the listed
// keys sits in LDS (persistent workgroups, four per CU); a key whose bit is clear cannot be listed and never reaches the
// L2 table.  Every listed key passes, so the result is the table's; what changes is the number of L2 lines gathered:
// with --cell 0.5 half of the CB tags are unsampled barcodes, and BAMs carry reads of unlisted barcodes anyway.
struct MissFilter { const u32* bits; u32 mask; };          // mask = number of bits - 1 (power of two)
__device__ __forceinline__ u32 filter_bit(u64 key) {
    const u32 h = (u32)key * 0x85EBCA77u + (u32)(key >> 32) * 0xC2B2AE3Du;
    return h ^ (h >> 13);
}

__global__ __launch_bounds__(K1_THREADS, 8) void probe_cells_filtered_kernel(const u64* __restrict__ cb, u64 n, Table cells, MissFilter f,
                                                                             const CellOut cell_out, bool c16, u32* __restrict__ tile_hits,
                                                                             u32* __restrict__ half_hits, u32 n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ u32 s_w[K1_WAVES];
    u32* s_bits = reinterpret_cast<u32*>(smem);
    {
        const uint4* src = reinterpret_cast<const uint4*>(f.bits);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (u32 i = threadIdx.x; i < (f.mask + 1u) / 128u; i += K1_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    for (u32 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u64 base = (u64)tile * K1_TILE + (u64)w * (K1_IPT * WAVE);
        u64 key[K1_IPT]; u32 cell[K1_IPT];
#pragma unroll
        for (int j = 0; j < K1_IPT; ++j) {
            const u64 idx = base + (u64)j * WAVE + lane;
            key[j] = idx < n ? cb[idx] : 0;
        }
#pragma unroll
        for (int j = 0; j < K1_IPT; ++j) {
            const u32 b = filter_bit(key[j]) & f.mask;
            if (!((s_bits[b >> 5] >> (b & 31u)) & 1u)) key[j] = 0;        // not listed: no probe (dead lanes share slot 0)
        }
        table_probe_batch<K1_IPT, 0>(cells, key, cell);
        u32 hits = 0, hits_lo = 0;
#pragma unroll
        for (int j = 0; j < K1_IPT; ++j) {
            const u64 idx = base + (u64)j * WAVE + lane;
            if (idx < n) put_cell(cell_out, c16, idx, cell[j]);
            hits += (u32)__popcll(__ballot(cell[j] != 0));
            if (j == K1_IPT / 2 - 1) hits_lo = hits;
        }
        if (lane == 0) {
            s_w[w] = hits;
            half_hits[((u64)tile * K1_WAVES + w) * 2] = hits_lo; half_hits[((u64)tile * K1_WAVES + w) * 2 + 1] = hits - hits_lo;
        }
        __syncthreads();
        if (tid == 0) {
            u32 t = 0;
            for (int i = 0; i < K1_WAVES; ++i) t += s_w[i];
            tile_hits[tile] = t;
        }
        __syncthreads();
    }
}

// K1a, LDS mode: persistent 1024-thread workgroups (one or two per CU, by the size of the table image); every wave
// walks its own 512-record chunks — each lane loads and stores PAIRS of neighbouring records (16-byte loads, 8-byte
// stores) — and adds its hit count to tile_hits[] (all-zero on entry) with one atomic: no barrier after the image is
// in LDS.
// TWO_PER_CU: the image leaves room for two workgroups per CU, which takes 64 VGPRs at most
template <bool TWO_PER_CU>
__global__ __launch_bounds__(1024, TWO_PER_CU ? 8 : 4) void probe_cells_lds_kernel(const u64* __restrict__ cb, u64 n, CellLds c,
                                                               const CellOut cell_dst, bool c16, u32* __restrict__ tile_hits,
                                                               u32* __restrict__ half_hits, u32 n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 S = c.slot_bits, smask = (1u << S) - 1u, lo_mask = (1u << (32u - S)) - 1u;
    const u32* s_slot = reinterpret_cast<const u32*>(smem);
    const unsigned short* s_disp = reinterpret_cast<const unsigned short*>(smem + ((size_t)4 << S));
    {
        const uint4* src = reinterpret_cast<const uint4*>(c.image);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (u32 i = threadIdx.x; i < (c.bytes + 15u) / 16u; i += 1024) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = lane_id();
    constexpr int PAIRS = K1_IPT / 2;                                    // 4 x 64 lanes x 2 records = one 512-record chunk
    const u32 n_chunks = n_tiles * (u32)K1_WAVES;
    void* const cell_out = cell_dst.p;
    const bool even = (reinterpret_cast<uintptr_t>(cb) & 15u) == 0 && (reinterpret_cast<uintptr_t>(cell_out) & 7u) == 0;
    for (u32 chunk = blockIdx.x * 16u + (threadIdx.x >> 6); chunk < n_chunks; chunk += gridDim.x * 16u) {
        const u64 base = (u64)chunk * (K1_IPT * WAVE);
        u64 key[K1_IPT]; u32 cell[K1_IPT];
        const bool whole = even && base + K1_IPT * WAVE <= n;            // wave-uniform
        if (whole) {
#pragma unroll
            for (int j = 0; j < PAIRS; ++j) {
                const ulonglong2 v = ld_once2<FASTF_NT_K1A != 0>(cb + base + 2ull * (j * WAVE + lane));
                key[2 * j] = v.x; key[2 * j + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K1_IPT; ++j) {
                const u64 idx = base + 2ull * ((j >> 1) * WAVE + lane) + (j & 1);
                key[j] = idx < n ? cb[idx] : 0;
            }
        }
        u32 hits = 0, hits_lo = 0;                                        // key[2 jj], key[2 jj + 1]: records 128 jj + 2 lane + {0, 1}
#pragma unroll
        for (int j = 0; j < K1_IPT; ++j) {
            const bool fam = (u32)(key[j] >> 49) == c.family && (key[j] & 0xFFFFu) == 0;
            const u32 h = cell_mix((u32)(key[j] >> 16), c.seed);
            const u32 lo = h & lo_mask;
            const u32 e = s_slot[((h >> (32u - S)) + s_disp[lo & c.bucket_mask]) & smask];
            const u32 v = (fam && (e >> S) == lo) ? (e & smask) : 0u;          // empty slots carry index 0
            cell[j] = v;
            hits += (u32)__popcll(__ballot(v != 0));
            if (j == K1_IPT / 2 - 1) hits_lo = hits;
        }
        if (whole && cell_dst.run) {
            // blocked buffer: the chunk is two units; pairs 0, 1 lie in the first, pairs 2, 3 in the second (scalar unit base)
            static_assert(K1_IPT * WAVE == 2 * (int)BLK_RECS && PAIRS == 4, "a K1a chunk is two blocked units");
            unsigned char* const u0 = reinterpret_cast<unsigned char*>(cell_out) + (base / BLK_RECS) * cell_dst.run + BLK_CELL;
#pragma unroll
            for (int j = 0; j < PAIRS; ++j) {
                unsigned char* const ub = u0 + (j >> 1) * cell_dst.run;
                const u32 rec = 2u * ((u32)(j & 1) * WAVE + (u32)lane);
                if (c16) *reinterpret_cast<u32*>(ub + 2u * rec) = cell[2 * j] | (cell[2 * j + 1] << 16);
                else *reinterpret_cast<uint2*>(ub + 4u * rec) = make_uint2(cell[2 * j], cell[2 * j + 1]);
            }
        } else if (whole) {
            if (c16) {
#pragma unroll
                for (int j = 0; j < PAIRS; ++j)
                    *reinterpret_cast<u32*>(reinterpret_cast<unsigned short*>(cell_out) + base + 2ull * (j * WAVE + lane)) = cell[2 * j] | (cell[2 * j + 1] << 16);
            } else {
#pragma unroll
                for (int j = 0; j < PAIRS; ++j)
                    *reinterpret_cast<uint2*>(reinterpret_cast<u32*>(cell_out) + base + 2ull * (j * WAVE + lane)) = make_uint2(cell[2 * j], cell[2 * j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K1_IPT; ++j) {
                const u64 idx = base + 2ull * ((j >> 1) * WAVE + lane) + (j & 1);
                if (idx < n) put_cell(cell_dst, c16, idx, cell[j]);
            }
        }
        if (lane == 0) {
            if (hits) atomicAdd(&tile_hits[chunk / (u32)K1_WAVES], hits);
            *reinterpret_cast<uint2*>(half_hits + 2ull * chunk) = make_uint2(hits_lo, hits - hits_lo);
        }
    }
}

// K1b geometry: the same 4096-record tiles as K1a, as THREADS x IPT
// 1024 threads x 4 records at 8 waves per SIMD (64 VGPRs): measured 87 us vs 99 us for 512 x 8 at 4 waves per SIMD
#ifndef FASTF_K1B_THREADS
#define FASTF_K1B_THREADS 1024
#endif
#ifndef FASTF_K1B_MINWAVES
#define FASTF_K1B_MINWAVES 8
#endif
#ifndef FASTF_K1B_MINWAVES_L2
#define FASTF_K1B_MINWAVES_L2 8
#endif
constexpr int K1B_THREADS = FASTF_K1B_THREADS, K1B_IPT = K1_TILE / K1B_THREADS, K1B_WAVES = K1B_THREADS / WAVE;
static_assert(K1B_THREADS * K1B_IPT == K1_TILE, "K1b walks K1a's tiles");

struct PackParams {
    const void* cell; bool cell16;  // K1a's scratch: u16 or u32 entries
    const unsigned char* blk;      // streaming form, BLOCKED: the units' runs (gx | umi | meta | scratch); cell/gx/umi/meta unused
    const u64* gx; const u32* umi; const u32* meta; u64 n;
    const u64* tile_base;          // exclusive scan of tile_hits
    const u32* dbits; u64 n_draws; // the decision stream (mt_fill_kernel<true>, draw_bits_kernel): hit rank r is kept iff bit (r & 31) of
                                   // dbits[(r & draw_mask) >> 5] is set; valid while r < n_draws
    u64 draw_mask;                 // ~0 for a linear array; ring size (in ranks, a power of two >= 32) - 1 for the streaming push path
    const u64* draw_base;          // optional device-side offset into draws (sharded runs, streaming pushes)
    Table feats;
    GeneLds genes;                 // LDS fast path of the feature lookup (LDS_GENES instantiation)
    u32 n_tiles;
    KeyLayout L;
    u32 n_shards;
    u64* keys; u64 shard_stride;   // keys + s*shard_stride
    // keys wider than 64 bits (tile form): keys[] takes the GROUP (cell << wide_feat_bits | feature) and vals[]
    // the rest of the key — NULL flag, UMI, length, laid out as the low feat_shift bits of a narrow key; vals == nullptr: narrow
    u64* vals; u32 wide_feat_bits;
    u32 wide_sub_bits;             // 0, or WIDE_SUB_BITS (umi_max_bases > 24): keys[] = (group << wide_sub_bits) | the UMI's first bases (make_val_sub)
    const u32* umi_ext;            // wide keys, umi_max_bases > 16: bases 17.. of every UMI (nullptr: none)
    u64* key_counts;               // [n_shards], appended
    u64* counters;                 // {hits, sampled, valid, err}
    u64* stamps;                   // diagnostic builds only (-DFASTF_STAMPS)
};

#ifdef FASTF_STAMPS
#define K1STAMP(i) do { __builtin_amdgcn_s_waitcnt(0); if (p.stamps && threadIdx.x == 0) p.stamps[(u64)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define K1STAMP(i) do { } while (0)
#endif

// ROOMY: the gene image leaves room for one workgroup per CU only (sparse id ranges of real lists: >78 KB) — then four
// waves per SIMD is all there is, and the kernel may as well use 128 VGPRs (112 us vs 125 us with the 64-VGPR build)
template <bool LDS_GENES, bool ROOMY = false>
__global__ __launch_bounds__(K1B_THREADS, ROOMY ? 4 : (LDS_GENES ? FASTF_K1B_MINWAVES : FASTF_K1B_MINWAVES_L2)) void filter_pack_kernel(const PackParams p) {
    __shared__ u32 s_cnt[K1B_IPT * K1B_WAVES];       // hits per (item, wave), then exclusive
    __shared__ u64 s_tot[3];                         // hits, sampled, valid of this workgroup's tiles so far
    __shared__ u32 s_shard_cnt[8];
    __shared__ u64 s_shard_base[8];
    __shared__ u32 s_err;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // LDS_GENES: the gene image

    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u32* s_bitmap = reinterpret_cast<const u32*>(smem);
    const unsigned short* s_rank = reinterpret_cast<const unsigned short*>(smem + (size_t)p.genes.words * 4);
    const unsigned short* s_perm = s_rank + ((p.genes.words + 1u) & ~1u);
    const unsigned short* s_direct = reinterpret_cast<const unsigned short*>(smem);      // direct mode: u16 index[range]
    if (LDS_GENES) {
        const uint4* src = reinterpret_cast<const uint4*>(p.genes.image);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (u32 i = tid; i < (p.genes.bytes + 15u) / 16u; i += K1B_THREADS) dst[i] = src[i];
    }

    // LDS_GENES: persistent workgroups walk the tiles; otherwise one tile per workgroup (grid = n_tiles)
    auto do_tile = [&](const u32 tile) {
    const u64 base = (u64)tile * K1_TILE;

    if (tid < 8) s_shard_cnt[tid] = 0;

    K1STAMP(0);
    // ---- loads; hit ranks in record order (item-major, then wave, then lane) ----
    u64 gxk[K1B_IPT]; u32 umi[K1B_IPT], meta[K1B_IPT], cell[K1B_IPT], hrank[K1B_IPT];
#pragma unroll
    for (int j = 0; j < K1B_IPT; ++j) {
        const u64 idx = base + (u64)j * K1B_THREADS + tid;
        const bool in = idx < p.n;
        cell[j] = in ? get_cell(p.cell, p.cell16, idx) : 0;
        gxk[j]  = in ? p.gx[idx] : 0;
        umi[j]  = in ? p.umi[idx] : 0;
        meta[j] = in ? p.meta[idx] : 0;
    }
#pragma unroll
    for (int j = 0; j < K1B_IPT; ++j) {
        const u64 hm = __ballot(cell[j] != 0);
        hrank[j] = rank_below(hm);
        if (lane == 0) s_cnt[j * K1B_WAVES + w] = (u32)__popcll(hm);
    }
    __syncthreads();
    if (w == 0) {
        static_assert(K1B_IPT * K1B_WAVES <= WAVE, "one wave scans the (item, wave) counts");
        const u32 c = lane < K1B_IPT * K1B_WAVES ? s_cnt[lane] : 0;
        const u32 inc = wave_incl_scan32(c, lane);
        if (lane < K1B_IPT * K1B_WAVES) s_cnt[lane] = inc - c;
    }
    __syncthreads();
    K1STAMP(1);
    const u64 tile_base = p.tile_base[tile] + (p.draw_base ? *p.draw_base : 0);

    // ---- depth draw (E4/E5 :385-390): the loads are issued here and consumed after the feature lookup ----
    u32 n_hit = 0, n_samp = 0, n_valid = 0, errs = 0;
    u32 draw[K1B_IPT], dbit[K1B_IPT];                                 // the decision word of a hit's rank and the bit in it
#pragma unroll
    for (int j = 0; j < K1B_IPT; ++j) {
        draw[j] = 0; dbit[j] = 0;
        if (cell[j] != 0) {
            const u64 r = tile_base + s_cnt[j * K1B_WAVES + w] + hrank[j];
            if (r < p.n_draws) { draw[j] = p.dbits[(r & p.draw_mask) >> 5]; dbit[j] = (u32)r & 31u; }
            else { cell[j] = 0; n_hit++; errs |= (u32)ERR_DRAWS_SHORT; }
        }
    }
    K1STAMP(2);
    u64 fkey[K1B_IPT]; u32 feat[K1B_IPT];
    if (LDS_GENES) {
        // E8 :403-410 ahead of E5/E7: the LDS lookup does not depend on the draw, so it runs while the draw loads are
        // in flight (for every CB hit with a good xf; a record the draw drops just wastes one lookup)
#pragma unroll
        for (int j = 0; j < K1B_IPT; ++j) fkey[j] = (cell[j] != 0 && (meta[j] & META_XF_OK)) ? gxk[j] : 0;
        if (p.genes.direct) {                                        // dense id range: one table read
#pragma unroll
            for (int j = 0; j < K1B_IPT; ++j) {
                u32 f = 0;
                const u64 k = fkey[j];
                if (k != 0) {
                    if ((u32)(k >> 44) == p.genes.family) {
                        const u64 v = (k & 0xFFFFFFFFFFFull) - p.genes.vmin;   // wraps to huge when below vmin
                        if (v < p.genes.range) f = s_direct[(u32)v];
                    } else f = table_probe(p.feats, k);              // other id families / escaped strings
                }
                feat[j] = f;
            }
        } else {                                                     // sparse range: bitmap word + rank (one wait), then the permutation
#pragma unroll
            for (int j = 0; j < K1B_IPT; ++j) {
                u32 f = 0;
                const u64 k = fkey[j];
                if (k != 0) {
                    if ((u32)(k >> 44) == p.genes.family) {
                        const u64 v = (k & 0xFFFFFFFFFFFull) - p.genes.vmin;
                        if (v < p.genes.range) {
                            const u32 wd = s_bitmap[(u32)v >> 5], rk = s_rank[(u32)v >> 5], bit = (u32)v & 31u;
                            if ((wd >> bit) & 1u) f = s_perm[rk + __popc(wd & ((1u << bit) - 1u))];
                        }
                    } else f = table_probe(p.feats, k);
                }
                feat[j] = f;
            }
        }
#pragma unroll
        for (int j = 0; j < K1B_IPT; ++j) {
            bool alive = cell[j] != 0;
            n_hit += alive;
            alive = alive && ((draw[j] >> dbit[j]) & 1u);
            n_samp += alive;                                         // E6 :392
            if (!alive) feat[j] = 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < K1B_IPT; ++j) {
            bool alive = cell[j] != 0;
            n_hit += alive;
            alive = alive && ((draw[j] >> dbit[j]) & 1u);
            n_samp += alive;                                         // E6 :392
            alive = alive && (meta[j] & META_XF_OK);                 // E7 :394-400
            fkey[j] = alive ? gxk[j] : 0;
        }
        table_probe_batch<K1B_IPT>(p.feats, fkey, feat);              // E8 :403-410
    }
    K1STAMP(3);

    // ---- UB (E9 :412-416), key (E10/E11), slot in the tile-local shard list ----
    u64 key[K1B_IPT]; u32 pos[K1B_IPT]; u32 shard[K1B_IPT]; bool emit[K1B_IPT];
#pragma unroll
    for (int j = 0; j < K1B_IPT; ++j) {
        key[j] = 0; pos[j] = 0; shard[j] = 0;
        const bool alive = feat[j] != 0 && (meta[j] & META_HAS_UB);
        u64 umi_w = 0;                                                 // wide keys: the UMI in 64 bits
        if (p.vals) {
            const u64 idx = base + (u64)j * K1B_THREADS + tid;
            umi_w = ((u64)umi[j] << 32) | ((p.umi_ext && alive && idx < p.n) ? p.umi_ext[idx] : 0u);
        }
        if (alive && (p.vals ? (p.wide_sub_bits ? umi_overflows_sub(meta[j]) : umi_overflows64(p.L, umi_w, meta[j])) : umi_overflows(p.L, umi[j], meta[j])))
            errs |= (u32)ERR_UMI_TOOLONG;
        if (alive) {
            n_valid++;                                               // E12 :435
            // (wide keys: the low part only here — cell and feature go into the group word at the store below)
            shard[j] = p.n_shards > 1 ? shard_of(cell[j], p.n_shards) : 0;
            if (p.vals && p.wide_sub_bits) key[j] = make_val_sub(umi_w, meta[j]);                  // (the sub-group: at the store below)
            else key[j] = p.vals ? make_val64(p.L, umi_w, meta[j]) : make_key(p.L, cell[j], feat[j], umi[j], meta[j]);
        }
        emit[j] = alive;
        // order inside a shard list is irrelevant: the keys are sorted next
        if (p.n_shards == 1) {
            const u64 m = __ballot(alive);
            if (m) {
                u32 b = 0;
                if (lane == 0) b = atomicAdd(&s_shard_cnt[0], (u32)__popcll(m));
                b = __shfl(b, 0, WAVE);
                if (alive) pos[j] = b + rank_below(m);
            }
        } else if (alive) {
            pos[j] = atomicAdd(&s_shard_cnt[shard[j]], 1u);          // one returning LDS atomic per key (8 ballot rounds cost more)
        }
    }

    K1STAMP(4);
    // ---- counters + global slot reservation ----
    // the three counters and the error bits gather in LDS; a persistent workgroup sends them to memory once, after its
    // last tile (same-address device-scope atomics serialise at about 12 ns each: three per tile were a sixth of this kernel)
    n_hit = wave_sum32(n_hit); n_samp = wave_sum32(n_samp); n_valid = wave_sum32(n_valid);
    if (lane == 0) {
        if (n_hit) atomicAdd(&s_tot[0], (u64)n_hit);
        if (n_samp) atomicAdd(&s_tot[1], (u64)n_samp);
        if (n_valid) atomicAdd(&s_tot[2], (u64)n_valid);
    }
    if (errs) atomicOr(&s_err, errs);
    __syncthreads();
    if (tid >= 64 && tid < 64 + (int)p.n_shards) {
        const u32 s = tid - 64;
        const u32 c = s_shard_cnt[s];
        u64 b = c ? atomicAdd(&p.key_counts[s], (u64)c) : 0;
        if (b + c > p.shard_stride) { atomicOr(&p.counters[3], ERR_KEYS_FULL); b = ~0ULL; }
        s_shard_base[s] = b;
    }
    __syncthreads();
    K1STAMP(5);
#pragma unroll
    for (int j = 0; j < K1B_IPT; ++j) {
        if (emit[j]) {
            const u64 b = s_shard_base[shard[j]];
            if (b != ~0ULL) {
                if (p.vals) {                                            // wide key: group and rest side by side
                    const u32 sub = p.wide_sub_bits ? wide_sub_of(umi[j], meta[j]) : 0u;
                    const u64 at = (u64)shard[j] * p.shard_stride + b + pos[j];
                    p.vals[at] = key[j];
                    p.keys[at] = ((((u64)cell[j] << p.wide_feat_bits) | feat[j]) << p.wide_sub_bits) | sub;
                } else p.keys[(u64)shard[j] * p.shard_stride + b + pos[j]] = key[j];
            }
        }
    }
    K1STAMP(6);
    };  // do_tile
    if (tid < 3) s_tot[tid] = 0;
    if (tid == 0) s_err = 0;
    if constexpr (LDS_GENES) {
        __syncthreads();                           // gene image is in LDS
        for (u32 tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
            do_tile(tile);
            __syncthreads();                       // the next tile re-initialises the shared scalars
        }
    } else {
        __syncthreads();
        do_tile(blockIdx.x);
        __syncthreads();
    }
    if (tid < 3 && s_tot[tid]) atomicAdd(&p.counters[tid], s_tot[tid]);
    if (tid == 3 && s_err) atomicOr(&p.counters[3], (u64)s_err);
}

// ------------------------------------------------------------------------------------
// K1b, streaming form (single shard, gene table in LDS): every WAVE works on its own 256-record units — no barrier,
// no workgroup-wide scan, no global atomic anywhere in the loop.
//   hit-rank base of a unit   tile_base[t] + the half_hits in front of it inside its tile (K1a wrote both)
//   output slot of a key      the workgroup owns a private region of the key buffer (capacity = the records it walks);
//                             a wave takes its slots with ONE returning LDS atomic per unit
//   counters                  kept per wave, added up in LDS once, three global atomics per workgroup at the end
// The key buffer is therefore SEGMENTED after this kernel: region b holds seg_count[b] keys at keys + b * region_stride.
// seg_scan_kernel turns the counts into prefix sums and the total; the first pass of the sort reads through that map
// (SegMap below), every later pass sees a contiguous buffer again.
// Measured motive (s_memtime stamps on the configs[2] shape, tile form): 26 % of a tile waited on the returning
// device-scope atomic that reserved its output slots, 22 % on tile_base + the draw gather behind the workgroup scan,
// and same-address device-scope atomics serialise at ~12 ns each.
// ------------------------------------------------------------------------------------
constexpr int K1S_IPT = 4, K1S_UNIT = K1S_IPT * WAVE;                       // 256 records per wave and step
static_assert(K1_TILE % K1S_UNIT == 0 && K1_TILE / K1S_UNIT == 16, "16 units per K1a tile");
// Waves per workgroup.  Two workgroups share a CU when the gene image allows it (2 x 73 KB on configs[2]); with 16 waves each
// that is 8 waves per SIMD and a budget of 64 VGPRs and 78 SGPRs per wave (the trap handler keeps 16 of the 96): the loop wants
// about 70 of the first and 120 of the second, so three vector registers went to scratch and 45 scalars into the lanes of a
// vector register — 90 v_readlane per unit, a quarter of the loop's vector instructions, only to fetch them back.  12 waves
// per workgroup = 6 per SIMD = 80 VGPRs and 102 SGPRs: nothing spills.
#ifndef FASTF_K1S_THREADS
#define FASTF_K1S_THREADS 768
#endif
constexpr int K1S_THREADS = FASTF_K1S_THREADS, K1S_THREADS_ROOMY = 1024;
static_assert(K1S_THREADS % (4 * WAVE) == 0 && K1S_THREADS <= 1024, "whole waves on every SIMD");

// Every workgroup owns K1S_SUB regions of the key buffer and fills them by turns, one unit into the one, the next into the
// other (the loop body exists twice anyway: the register sets of the hand-pipelined loop): region 2 b + s, region_stride slots
// each.  Twice the regions of half the size make the first sort pass — which walks REGIONS, not tiles — a full round of
// workgroups at four per CU.  Per region the kernel also leaves the histogram of the sort's first digit (the 8 key bits from
// hist_shift up; LDS atomics on the keys it has in registers anyway): rgn_hist[256 r + d], and with it the first pass needs no
// counting pass over the keys (scatter_regions_kernel; round 4 read all keys once more for that: tile_count_kernel<true>).
constexpr int K1S_SUB = 2, K1S_BINS = 256;
struct StreamParams {
    const u32* half_hits;          // [16 * n_tiles]
    u64 region_stride;             // key slots per region
    u64* seg_count;                // [K1S_SUB * gridDim.x] out: keys per region
    u32* rgn_hist;                 // [K1S_SUB * gridDim.x][256] out, or nullptr: no histogram wanted (sharded passes)
    u64* rgn_phys;                 // [K1S_SUB * gridDim.x] out (with rgn_hist): first slot of each region, counted from slot0
    u64 slot0;                     //   the slot number p.keys stands at in the caller's key store
    u32 hist_shift;
};

// sum over the lanes 0..15 of one 16-lane row (DPP row shifts, no LDS traffic); the total is returned to every lane
__device__ __forceinline__ u32 row16_sum_lane15(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);      // row_shr:8
    return (u32)__builtin_amdgcn_readlane((int)x, 15);
}

// C16: the cell scratch holds u16 entries; DIRECT: the gene image is the dense u16 table.  Both are compile-time so that
// the loop is straight-line code: with run-time flags every load sat in its own branch, with waits between them
// (PMC on the configs[2] shape: 430 vector + 325 scalar instructions per 256 records, the vector ALU busy 60 % of the kernel).
template <bool ROOMY, bool C16, bool DIRECT, bool BLOCKED = false>
__global__ __launch_bounds__(ROOMY ? K1S_THREADS_ROOMY : K1S_THREADS, ROOMY ? 4 : 2 * K1S_THREADS / (4 * WAVE)) void filter_pack_stream_kernel(const PackParams p, const StreamParams sp) {
    constexpr int THREADS = ROOMY ? K1S_THREADS_ROOMY : K1S_THREADS, WAVES = THREADS / WAVE;
    __shared__ u64 s_tot[3];
    __shared__ u32 s_cursor[K1S_SUB], s_err;
    __shared__ u32 s_hist[K1S_SUB * K1S_BINS];                             // first sort digit of the keys of each region
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the gene image
    const int tid = threadIdx.x, lane = lane_id();
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                // scalar: the unit's addresses are scalar base + lane
    const u32* s_bitmap = reinterpret_cast<const u32*>(smem);
    const unsigned short* s_rank = reinterpret_cast<const unsigned short*>(smem + (size_t)p.genes.words * 4);
    const unsigned short* s_perm = s_rank + ((p.genes.words + 1u) & ~1u);
    const unsigned short* s_direct = reinterpret_cast<const unsigned short*>(smem);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.genes.image);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (u32 i = tid; i < (p.genes.bytes + 15u) / 16u; i += THREADS) dst[i] = src[i];
    }
    if (tid < 3) s_tot[tid] = 0;
    if (tid < K1S_SUB) s_cursor[tid] = 0;
    if (tid == 0) s_err = 0;
    for (int i = tid; i < K1S_SUB * K1S_BINS; i += THREADS) s_hist[i] = 0;
    __syncthreads();

    constexpr bool NT = FASTF_NT_K1B != 0;
    u64* const region0 = p.keys + (u64)blockIdx.x * K1S_SUB * sp.region_stride;
    const bool hs_hi = sp.hist_shift >= 32u;                               // (uniform) the digit lies in the key's upper word
    const u64 draw_off = p.draw_base ? *p.draw_base : 0;
    u32 w_hit = 0, w_samp = 0, w_valid = 0, errs = 0;                      // wave-uniform running counts
    // the workgroup's waves take consecutive units per round (unit u: tile u >> 4, place u & 15 in it): the same locality as
    // the tile form.
    //
    // The loop is software-pipelined BY HAND: a unit's inputs are requested during the unit before —
    //   cell, umi, meta and the unit's     at the top of the unit before, into a second set of registers (13 of them: with 12
    //   decision words                     waves per workgroup the budget is 80 and the loop needs 60)
    //   gx                                 behind the gene lookup of the unit before, into the registers it just vacated
    //   tile base, hit counts of the tile  two units ahead: they place the unit in the decision stream when its words are asked for
    // — and every one of those loads is unconditional (a unit that does not exist reads the last one that does, a lane beyond
    // the records its unit's last record): the compiler's wait counters then stay exact, each wait covers the loads it needs
    // and not the ones just issued.  Before this a wave asked for its 4.6 KB, waited two to three microseconds with nothing
    // else of its own in flight, and worked for six: eight waves per SIMD covered that only in part (12 waves per workgroup
    // without a spill ran no faster than 16 with 90 spill reloads per unit — the loop was latency-bound,
    // profiles/r4_notes/ab_k1b_pipelined.txt).
    const u32 n_units = p.n_tiles * (u32)(K1_TILE / K1S_UNIT), ustep = gridDim.x * (u32)WAVES;
    const u32 n_live = (u32)((p.n + K1S_UNIT - 1) / K1S_UNIT);            // units that hold records (the tiles are whole: the rest is empty); >= 1
    u32 u = blockIdx.x * (u32)WAVES + (u32)w;                              // scalar
    // one unit's inputs apart from gx: the records' cell / umi / meta, the unit's decision words, and (wave-uniform) the
    // decisions left in the stream from its first rank on, that rank's bit in its word, the records the unit holds
    struct UnitRegs { u32 cell[K1S_IPT], umi[K1S_IPT], meta[K1S_IPT], dw, avail, sbit, nvalid; };
    UnitRegs ra, rb;                                                       // the unit in work and the one after it, by turns
    u64 gxk[K1S_IPT];
    u64 tb = 0; u32 hha = 0;                                               // small inputs of the unit after it (tb: the same value in every lane)
    u32 zlane = 0;
    asm volatile("" : "+v"(zlane));                                        // an opaque zero per lane, see small_inputs
    // records a unit holds (0: none), and the unit whose memory stands in for it when it holds none
    auto records_of = [&](const u32 uu) { return uu < n_live ? (u32)(p.n - (u64)uu * K1S_UNIT < (u64)K1S_UNIT ? p.n - (u64)uu * K1S_UNIT : (u64)K1S_UNIT) : 0u; };
    auto unit_run = [&](const u32 uu) { return BLOCKED ? p.blk + (u64)uu * blk_run_bytes(C16) : nullptr; };
    // (nv >= 1 records of unit uu exist; a blocked run is whole even when its unit is not)
    auto load_gx = [&](u64 (&g)[K1S_IPT], const u32 uu, const u32 nv) {
        const u64* const gx_u = BLOCKED ? reinterpret_cast<const u64*>(unit_run(uu) + BLK_GX) : p.gx + (u64)uu * K1S_UNIT;
#pragma unroll
        for (int j = 0; j < K1S_IPT; ++j) {
            const u32 o = (u32)j * WAVE + (u32)lane;
            g[j] = ld_once<NT>(gx_u + (BLOCKED || o < nv ? o : nv - 1u));
        }
    };
    auto load_rest = [&](u32 (&c)[K1S_IPT], u32 (&um)[K1S_IPT], u32 (&me)[K1S_IPT], const u32 uu, const u32 nv) {
        const unsigned char* const run = unit_run(uu);
        const u64 base = (u64)uu * K1S_UNIT;
        const u32* const umi_u = BLOCKED ? reinterpret_cast<const u32*>(run + BLK_UMI) : p.umi + base;
        const u32* const meta_u = BLOCKED ? reinterpret_cast<const u32*>(run + BLK_META) : p.meta + base;
        const unsigned short* const c16_u = BLOCKED ? reinterpret_cast<const unsigned short*>(run + BLK_CELL) : reinterpret_cast<const unsigned short*>(p.cell) + base;
        const u32* const c32_u = BLOCKED ? reinterpret_cast<const u32*>(run + BLK_CELL) : reinterpret_cast<const u32*>(p.cell) + base;
#pragma unroll
        for (int j = 0; j < K1S_IPT; ++j) {
            const u32 o = (u32)j * WAVE + (u32)lane, oc = BLOCKED || o < nv ? o : nv - 1u;
            c[j] = C16 ? (u32)ld_once<NT>(c16_u + oc) : ld_once<NT>(c32_u + oc);
            um[j] = ld_once<NT>(umi_u + oc);
            me[j] = ld_once<NT>(meta_u + oc);
        }
    };
    // place a unit in the decision stream from its tile's small inputs and ask for its decision words: bits rank0 .. rank0 +
    // its hits of the stream — nine 32-bit words at most, word i in lane i (the lanes behind repeat them; one or two cache lines).
    // avail: decisions left in the stream from the unit's first rank on (capped: a unit has 256 hits at most) — ranks beyond
    // the stream are caught in the loop, the load stays in bounds (the stream holds at least one word: launch_probe)
    struct Place { u32 avail, sbit; };
    auto place_unit = [&](u32& words, const u32 uu, const u64 tbv, const u32 hhu) {
        const u32 place = uu & 15u;
        const u64 rank0 = uniform64(tbv) + draw_off + row16_sum_lane15((u32)lane < place ? hhu : 0u);
        Place q;
        q.avail = rank0 < p.n_draws ? (u32)(p.n_draws - rank0 < 1024 ? p.n_draws - rank0 : 1024) : 0u;
        q.sbit = (u32)rank0 & 31u;
        // (word numbers in 32 bits: a stream of up to 2^37 decisions; the clamp to the last word that holds one is scalar)
        const u64 last = p.n_draws ? (p.n_draws - 1) >> 5 : 0, first = rank0 >> 5;
        const u32 room = first < last ? (u32)(last - first < 15 ? last - first : 15) : 0u;
        const u32 off = (u32)lane & 15u;
        const u32 wi = ((u32)(first < last ? first : last) + (off < room ? off : room)) & (u32)(p.draw_mask >> 5);
        words = p.dbits[wi];
        return q;
    };
    // (asked for as VECTOR loads, the tile base through an opaque zero in its address: a load the compiler knows to be
    //  uniform is moved to scalar registers right behind its issue, and that wait would cover every load in flight — the
    //  records just asked for; lanes 16.. repeat the hit counts, place_unit() reads lanes < 16 only)
    auto small_inputs = [&](const u32 uu) {
        const u32 t = (uu < n_units ? uu : n_units - 1u) >> 4;
        tb = p.tile_base[t + zlane];
        hha = sp.half_hits[16ull * t + ((u32)lane & 15u)];
    };
    if (u < n_units) {                                                     // the first unit: everything at once
        small_inputs(u);
        ra.nvalid = records_of(u);
        const u32 us = ra.nvalid ? u : n_live - 1u, ns = ra.nvalid ? ra.nvalid : records_of(n_live - 1u);
        load_rest(ra.cell, ra.umi, ra.meta, us, ns);
        const Place q = place_unit(ra.dw, u, tb, hha);
        ra.avail = q.avail; ra.sbit = q.sbit;
        small_inputs(u + ustep);
        load_gx(gxk, us, ns);                                              // (last, as in the loop: the same loads are in flight behind the small inputs either way in)
    }
    // one unit: `cur` holds it, `nxt` receives the unit after it.  The loop below runs the body twice per turn with the two
    // register sets exchanged — a copy at the end of the body would have to wait for the loads it copies
    auto one_unit = [&](UnitRegs& cur, UnitRegs& nxt, auto sub_c) {
        constexpr int SUB = decltype(sub_c)::value;                        // which of the workgroup's regions this unit's keys go to
        u32 (&cell)[K1S_IPT] = cur.cell; u32 (&umi)[K1S_IPT] = cur.umi; u32 (&meta)[K1S_IPT] = cur.meta;
        const u32 dw = cur.dw, avail = cur.avail, sbit = cur.sbit, nvalid = cur.nvalid;
        const u32 un = u + ustep;                                          // the unit after this one
        const u32 nv_n = un < n_units ? records_of(un) : 0u;
        // the unit whose memory the loads for `un` read: itself, or (no such unit, no records in it) the last one with records
        const u32 us = nv_n ? un : n_live - 1u, ns = nv_n ? nv_n : records_of(n_live - 1u);
        // ---- the unit after this one: its place in the decision stream and its decision words (from the small inputs asked
        //      for a unit ago), cell / umi / meta into the second register set, the small inputs of the unit after that ----
        const Place qn = place_unit(nxt.dw, un < n_units ? un : u, tb, hha);
        nxt.avail = qn.avail; nxt.sbit = qn.sbit; nxt.nvalid = nv_n;
        load_rest(nxt.cell, nxt.umi, nxt.meta, us, ns);
        small_inputs(un + ustep);
        __builtin_amdgcn_sched_barrier(0);
        if (nvalid < (u32)K1S_UNIT) {                                      // (uniform) the last unit with records, or an empty one
#pragma unroll
            for (int j = 0; j < K1S_IPT; ++j) cell[j] = (u32)j * WAVE + (u32)lane < nvalid ? cell[j] : 0u;
        }
        // ---- depth draw (E4/E5): ranks in record order; a record's decision comes from the lane that fetched its word ----
        u32 keep[K1S_IPT];
        {
            u64 hm[K1S_IPT]; u32 H = 0;
#pragma unroll
            for (int j = 0; j < K1S_IPT; ++j) { hm[j] = __ballot(cell[j] != 0); H += (u32)__popcll(hm[j]); }
            w_hit += H;
            u32 pre = 0;                                                   // hits of the unit in front of item j
#pragma unroll
            for (int j = 0; j < K1S_IPT; ++j) {
                const u32 rl = pre + rank_below(hm[j]);                    // local hit rank (meaningful on hit lanes)
                const u32 bpos = sbit + rl;                                // < 32 + 256: words 0..8
                keep[j] = ((u32)__builtin_amdgcn_ds_bpermute((int)((bpos >> 5) << 2), (int)dw) >> (bpos & 31u)) & 1u;    // (only read where cell[j] != 0)
                if (H > avail) cell[j] = rl >= avail ? 0u : cell[j];       // (uniform, never on a complete stream) a hit beyond the stream: dropped
                pre += (u32)__popcll(hm[j]);
            }
            if (H > avail) errs |= (u32)ERR_DRAWS_SHORT;
        }
        // ---- E8: gene lookup in LDS ----
        // straight-line: every lane reads the table (index 0 when it has no business there) and the result is selected — an
        // exec-mask branch per condition costs more instructions than the lookup; ids of other families (rare: the L2 table)
        // are left to a wave-uniform branch
        u32 feat[K1S_IPT];
#pragma unroll
        for (int j = 0; j < K1S_IPT; ++j) {
            const u64 k = gxk[j];
            const bool want = (cell[j] != 0) & ((meta[j] & META_XF_OK) != 0) & (k != 0);
            const bool fam = (u32)(k >> 44) == p.genes.family;
            const u64 v = (k & 0xFFFFFFFFFFFull) - p.genes.vmin;                         // wraps to huge when below vmin
            const bool in = want & fam & (v < p.genes.range);
            const u32 vi = in ? (u32)v : 0u;
            u32 f;
            if (DIRECT) f = s_direct[vi];
            else {
                const u32 wd = s_bitmap[vi >> 5], rk = s_rank[vi >> 5], bit = vi & 31u;
                const u32 pi = ((wd >> bit) & 1u) ? rk + __popc(wd & ((1u << bit) - 1u)) : 0u;
                f = ((wd >> bit) & 1u) ? s_perm[pi] : 0u;
            }
            f = in ? f : 0u;
            if (__ballot(want & !fam)) { if (want & !fam) f = table_probe(p.feats, k); }  // other id families / escaped strings
            feat[j] = f;
        }
        // gx has done its work: the gx of the unit after this one into its registers
        __builtin_amdgcn_sched_barrier(0);
        load_gx(gxk, us, ns);
        __builtin_amdgcn_sched_barrier(0);
        // ---- E5..E12: keep/drop, key; slots of the wave inside the workgroup's region ----
        u64 key[K1S_IPT], em[K1S_IPT];
        u32 n_keys = 0;
#pragma unroll
        for (int j = 0; j < K1S_IPT; ++j) {
            bool alive = (cell[j] != 0) & (keep[j] != 0);
            w_samp += (u32)__popcll(__ballot(alive));                                      // E6
            alive = alive & (feat[j] != 0) & ((meta[j] & META_HAS_UB) != 0);
            if (__ballot(alive & umi_overflows_flat(p.L, umi[j], meta[j]))) errs |= (u32)ERR_UMI_TOOLONG;      // (uniform branch, never taken on good data)
            key[j] = make_key_flat(p.L, cell[j], feat[j], umi[j], meta[j]);                // (stored where alive only)
            em[j] = __ballot(alive);
            n_keys += (u32)__popcll(em[j]);
        }
        w_valid += n_keys;                                                                 // E12
        if (n_keys) {
            u32 pos0 = 0;
            if (lane == 0) pos0 = atomicAdd(&s_cursor[SUB], n_keys);
            pos0 = (u32)__builtin_amdgcn_readfirstlane((int)pos0);
            if ((u64)pos0 + n_keys > sp.region_stride) errs |= (u32)ERR_KEYS_FULL;         // cannot happen: the region holds every record that goes to it
            else {
                u64* const region = region0 + (u64)SUB * sp.region_stride;
#pragma unroll
                for (int j = 0; j < K1S_IPT; ++j) {
                    if (__builtin_amdgcn_inverse_ballot_w64(em[j])) {
                        region[pos0 + rank_below(em[j])] = key[j];                         // (32-bit index off the region's scalar base)
                        // the sort's first digit of the key, counted per region (scatter_regions_kernel); always: a sharded pass
                        // has no use for it, and a test in the loop would cost it a scalar register it does not have
                        const u32 dg = hs_hi ? (u32)(key[j] >> 32) >> (sp.hist_shift - 32u) : __builtin_amdgcn_alignbit((u32)(key[j] >> 32), (u32)key[j], sp.hist_shift);
                        (void)__hip_atomic_fetch_add(&s_hist[SUB * K1S_BINS + (dg & (K1S_BINS - 1))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    pos0 += (u32)__popcll(em[j]);
                }
            }
        }
        u = un;
    };
    while (u < n_units) {
        one_unit(ra, rb, std::integral_constant<int, 0>{});
        if (u >= n_units) break;
        one_unit(rb, ra, std::integral_constant<int, 1>{});
    }
    if (lane == 0) {
        if (w_hit) atomicAdd(&s_tot[0], (u64)w_hit);
        if (w_samp) atomicAdd(&s_tot[1], (u64)w_samp);
        if (w_valid) atomicAdd(&s_tot[2], (u64)w_valid);
    }
    {
        u32 e = errs;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e |= __shfl_xor(e, o, WAVE);
        if (lane == 0 && e) atomicOr(&s_err, e);
    }
    __syncthreads();
    // (the thread index from the wave number and the lane: `tid` itself would have to live through the loop in a register the
    //  loop does not have — it went to scratch memory)
    const int te = w * WAVE + lane;
    if (te < 3 && s_tot[te]) atomicAdd(&p.counters[te], s_tot[te]);
    if (te == 3 && s_err) atomicOr(&p.counters[3], (u64)s_err);
    if (te >= 64 && te < 64 + K1S_SUB) {
        const u32 r = blockIdx.x * K1S_SUB + (u32)(te - 64);
        sp.seg_count[r] = s_cursor[te - 64];
        if (sp.rgn_hist) sp.rgn_phys[r] = sp.slot0 + (u64)r * sp.region_stride;
    }
    if (sp.rgn_hist)                                                       // one contiguous 2 KB run per workgroup
        for (int i = te; i < K1S_SUB * K1S_BINS; i += THREADS) sp.rgn_hist[(u64)blockIdx.x * (K1S_SUB * K1S_BINS) + i] = s_hist[i];
}

// SoA on the device -> blocked runs (callers that hold device-resident SoA and want the blocked K1 path; the push path lands
// its host batches in this layout with pitched copies instead).  One wave per unit and round.
__global__ __launch_bounds__(256) void block_records_kernel(const u64* __restrict__ gx, const u32* __restrict__ umi, const u32* __restrict__ meta,
                                                            u64 n, unsigned char* __restrict__ blk, u32 run) {
    const u64 units = (n + BLK_RECS - 1) / BLK_RECS, waves = (u64)gridDim.x * (blockDim.x / WAVE);
    const int lane = lane_id();
    for (u64 u = (u64)blockIdx.x * (blockDim.x / WAVE) + (threadIdx.x >> 6); u < units; u += waves) {
        unsigned char* const r = blk + u * run;
#pragma unroll
        for (int j = 0; j < (int)BLK_RECS / WAVE; ++j) {
            const u32 o = (u32)j * WAVE + (u32)lane;
            const u64 i = u * BLK_RECS + o;
            const bool in = i < n;
            reinterpret_cast<u64*>(r + BLK_GX)[o] = in ? gx[i] : 0;
            reinterpret_cast<u32*>(r + BLK_UMI)[o] = in ? umi[i] : 0;
            reinterpret_cast<u32*>(r + BLK_META)[o] = in ? meta[i] : 0;
        }
    }
}

// Several shards (multi-GPU): the streaming K1b above writes its keys unsharded into the workgroup regions of a scratch
// buffer, and this kernel deals them to the per-destination buffers — workgroup b takes region b, counts its keys per
// destination in LDS (one returning LDS atomic per key), reserves the space of each destination with ONE device-scope
// atomic per destination and tile, and writes every destination's keys as one dense run.  8 bytes read + 8 written per key
// (0.19 keys per record on the configs[2] shape) against the streaming kernel's 1.5x advantage over the tile form on
// 24 bytes per record.
constexpr int SP_THREADS = 256, SP_IPT = 8, SP_TILE = SP_THREADS * SP_IPT, MAX_SHARDS = 8;
__global__ __launch_bounds__(SP_THREADS) void shard_partition_kernel(const u64* __restrict__ seg_keys, const u64* __restrict__ seg_count,
                                                                     u64 region_stride, u32 cell_shift, u32 n_shards,
                                                                     u64* __restrict__ keys_out, u64 shard_stride,
                                                                     u64* __restrict__ key_counts, u64* __restrict__ err) {
    __shared__ u32 s_cnt[MAX_SHARDS];
    __shared__ u64 s_base[MAX_SHARDS];
    const int tid = threadIdx.x;
    const u64 cnt = seg_count[blockIdx.x];
    const u64* src = seg_keys + (u64)blockIdx.x * region_stride;
    for (u64 i0 = 0; i0 < cnt; i0 += SP_TILE) {
        if (tid < MAX_SHARDS) s_cnt[tid] = 0;
        __syncthreads();
        u64 key[SP_IPT]; u32 sh[SP_IPT], lp[SP_IPT];
#pragma unroll
        for (int j = 0; j < SP_IPT; ++j) {
            const u64 i = i0 + (u64)j * SP_THREADS + tid;
            key[j] = i < cnt ? src[i] : 0;
            sh[j] = 0; lp[j] = 0;
            if (i < cnt) { sh[j] = shard_of((u32)(key[j] >> cell_shift), n_shards); lp[j] = atomicAdd(&s_cnt[sh[j]], 1u); }
        }
        __syncthreads();
        if (tid < (int)n_shards) {
            const u32 c = s_cnt[tid];
            u64 bs = c ? atomicAdd(&key_counts[tid], (u64)c) : 0;
            if (bs + c > shard_stride) { atomicOr(err, ERR_KEYS_FULL); bs = ~0ull; }
            s_base[tid] = bs;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SP_IPT; ++j) {
            const u64 i = i0 + (u64)j * SP_THREADS + tid;
            if (i < cnt && s_base[sh[j]] != ~0ull) keys_out[(u64)sh[j] * shard_stride + s_base[sh[j]] + lp[j]] = key[j];
        }
        __syncthreads();
    }
}

// the map from the logical key index (regions back to back) to the segmented buffer.  Per sort tile one 32-byte entry
// {delta1, hi1, delta2, hi2}: keys of the tile with logical index < hi1 live at index + delta1 (the region of the tile's
// first key), those below hi2 at index + delta2 (the next region that holds keys); only a tile that runs through more
// than two regions (regions of a few keys) searches prefix[].
struct TileSeg { u64 delta1, hi1, delta2, hi2; };
struct SegMap { const u64* prefix; const TileSeg* tile_seg; u32 n_seg; u64 stride; };   // prefix == nullptr: contiguous buffer

// prefix[b] = keys in the regions before b, prefix[n_seg] = total, also stored to *n_out (the key count the sort reads)
// add: *n_out grows by the total instead (the push path: regions of one chunk behind those of the chunks before)
__global__ __launch_bounds__(1024) void seg_scan_kernel(const u64* __restrict__ seg_count, u32 n_seg, u64* __restrict__ prefix,
                                                        u64* __restrict__ n_out, bool add = false) {
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 b0 = 0; b0 < n_seg; b0 += 1024) {
        const u32 b = b0 + threadIdx.x;
        const u64 v = b < n_seg ? seg_count[b] : 0;
        u64 inc = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) { const u64 x = __shfl_up(inc, o, WAVE); if (lane >= o) inc += x; }
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        u64 off = s_carry;
        for (int i = 0; i < w; ++i) off += s_w[i];
        if (b < n_seg) prefix[b] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) { prefix[n_seg] = s_carry; *n_out = add ? *n_out + s_carry : s_carry; }
}

// largest b with prefix[b] <= idx and a non-empty region (idx < total)
__device__ __forceinline__ u32 seg_region_of(const u64* __restrict__ prefix, u32 n_seg, u64 idx) {
    u32 lo = 0, hi = n_seg;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (prefix[mid] <= idx) lo = mid; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(256) void seg_tiles_kernel(const u64* __restrict__ prefix, u32 n_seg, u64 stride, u32 tile_keys,
                                                        TileSeg* __restrict__ tile_seg) {
    const u64 total = prefix[n_seg];
    const u64 T = (total + tile_keys - 1) / tile_keys;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < T; t += (u64)gridDim.x * 256) {
        const u64 first = t * tile_keys;
        const u32 b = seg_region_of(prefix, n_seg, first);
        TileSeg ts;
        ts.delta1 = (u64)b * stride - prefix[b]; ts.hi1 = prefix[b + 1];
        u32 b2 = b + 1;                                   // the next region that holds keys (if any)
        while (b2 < n_seg && prefix[b2 + 1] == ts.hi1) ++b2;
        if (b2 < n_seg) { ts.delta2 = (u64)b2 * stride - prefix[b2]; ts.hi2 = prefix[b2 + 1]; }
        else { ts.delta2 = 0; ts.hi2 = ts.hi1; }
        tile_seg[t] = ts;
    }
}

__device__ __forceinline__ u64 seg_phys(const SegMap m, u64 idx, const TileSeg ts) {
    if (idx < ts.hi1) return idx + ts.delta1;
    if (idx < ts.hi2) return idx + ts.delta2;
    const u32 b = seg_region_of(m.prefix, m.n_seg, idx);
    return (u64)b * m.stride + (idx - m.prefix[b]);
}

// ------------------------------------------------------------------------------------
// K2: LSD radix sort, 8-bit digits.  Per pass: tile_count → row_scan → scatter.
// ------------------------------------------------------------------------------------
constexpr int RADIX = 256;
// keys per thread of a sort tile, compile-time bound: 8 keeps the scatter at 47 VGPRs (8 waves per SIMD)
#ifndef FASTF_SORT_IPT_MAX
#define FASTF_SORT_IPT_MAX 8
#endif
#ifndef FASTF_SORT_THREADS
#define FASTF_SORT_THREADS 512
#endif
constexpr int SORT_THREADS = FASTF_SORT_THREADS, SORT_IPT = FASTF_SORT_IPT_MAX, SORT_TILE = SORT_THREADS * SORT_IPT, SORT_WAVES = SORT_THREADS / WAVE;

// tile size is chosen per sort (ipt = keys per thread, 1..SORT_IPT) so that the tiles fill whole rounds of the
// resident workgroup slots: at 10 M keys a fixed 8192-key tile leaves the third round 38 % full
__device__ __forceinline__ u32 num_tiles(u64 n, u32 ipt) { const u64 t = (u64)ipt * SORT_THREADS; return (u32)((n + t - 1) / t); }
// rows of cnt[d][tile] are padded to a multiple of 4 tiles so a row scan can use 16-byte accesses
__device__ __forceinline__ u32 row_stride(u32 T) { return (T + 3u) & ~3u; }

// A sort whose keys outgrow the 256 MB Infinity Cache streams them (non-temporal loads: +14 % on tile_count at 38 M keys);
// a smaller one finds the keys of the pass before still cached, and plain loads are faster there (10 M keys: scatter
// 38 us plain, 43 us streamed).
__device__ __forceinline__ bool sort_streams(u64 n_keys) { return n_keys >= (24ull << 20); }

// XCD-contiguous tile of this workgroup.  The ranges are cut from the ACTUAL tile count T (the grid is sized for the
// caller's upper bound on the key count and may be several times larger: cutting the ranges from gridDim would leave
// most XCDs without a tile when the bound is loose).  Returns a value >= T for workgroups without a tile.
// The grid is capped (launch_sort): a workgroup takes the tiles r, r + gridDim/8, ... of its XCD's range, so a loose
// bound on the key count costs no empty workgroups, each of which would still have to fetch the key count.
__device__ __forceinline__ u32 xcd_tile(u32 T, u32 r) {
    const u32 chunk = (T + 7u) >> 3;
    return r < chunk ? (blockIdx.x & 7u) * chunk + r : ~0u;
}

// per-tile digit counts: cnt[d * T + tile].  Four LDS copies of the histogram (lane & 3) keep the
// same-address atomic conflicts of skewed digits (e.g. the constant length bits) four times shorter.
// At 38 M keys the kernel takes 63 us for 307 MB, which is what a plain cold read of that size takes on this chip
// (tools/hbm_probe.hip: 74 us).  Tried without gain: 8 or 16 histogram copies, per-wave copies (slower), a persistent grid that
// requests the next tile's keys before counting the current one (69 us), counting from one digit byte per key that the
// scatter before leaves beside the keys (52 us, but the scatter pays 20 us for the byte stores).
template <bool SEG>
__global__ __launch_bounds__(SORT_THREADS) void tile_count_kernel(const u64* __restrict__ keys, const u64* __restrict__ n_ptr,
                                                                  u32 shift, u32* __restrict__ cnt, u32 ipt, const SegMap seg) {
    __shared__ u32 s_h[4 * RADIX];
    const u64 n = *n_ptr;
    const u32 T = num_tiles(n, ipt);
    // same XCD-contiguous tile mapping as the scatter: the 4-byte counts of neighbouring tiles share cache lines in
    // every digit row of cnt[][]
    for (u32 r = blockIdx.x >> 3;; r += gridDim.x >> 3) {
        const u32 tile = xcd_tile(T, r);
        if (tile >= T) return;                               // block-uniform
        for (int i = threadIdx.x; i < 4 * RADIX; i += SORT_THREADS) s_h[i] = 0;
        __syncthreads();
        const u64 base = (u64)tile * ipt * SORT_THREADS;
        u32* my = s_h + (threadIdx.x & 3) * RADIX;
        u64 k[SORT_IPT];
        if constexpr (SEG) {                                 // first pass over the segmented output of the streaming K1b
            const TileSeg ts = seg.tile_seg[tile];
#pragma unroll
            for (int j = 0; j < SORT_IPT; ++j) {
                const u64 idx = base + (u64)j * SORT_THREADS + threadIdx.x;
                k[j] = (j < (int)ipt && idx < n) ? ld_once<FASTF_NT_SORT != 0>(keys + seg_phys(seg, idx, ts)) : 0;
            }
        } else if (sort_streams(n)) {                        // block-uniform
#pragma unroll
            for (int j = 0; j < SORT_IPT; ++j) {
                const u64 idx = base + (u64)j * SORT_THREADS + threadIdx.x;
                k[j] = (j < (int)ipt && idx < n) ? ld_once<FASTF_NT_SORT != 0>(keys + idx) : 0;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SORT_IPT; ++j) {
                const u64 idx = base + (u64)j * SORT_THREADS + threadIdx.x;
                k[j] = (j < (int)ipt && idx < n) ? keys[idx] : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < SORT_IPT; ++j) {
            const u64 idx = base + (u64)j * SORT_THREADS + threadIdx.x;
            if (j < (int)ipt && idx < n) atomicAdd(&my[(k[j] >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x < RADIX) {
            const int d = threadIdx.x;
            cnt[(u64)d * row_stride(T) + tile] = s_h[d] + s_h[RADIX + d] + s_h[2 * RADIX + d] + s_h[3 * RADIX + d];
        }
        __syncthreads();                                     // the next tile clears the histograms
    }
}

// row d: exclusive scan of cnt[d][0..T) in place, and the row total (= keys with digit d) into bin_tot[d];
// 4 tiles per thread (one uint4).  The scatter turns bin_tot into bin bases itself (a 256-wide scan per tile).
__global__ __launch_bounds__(1024) void row_scan_kernel(u32* __restrict__ cnt, const u64* __restrict__ n_ptr,
                                                        u32* __restrict__ bin_tot, u32 ipt) {
    __shared__ u32 s_w[16];
    __shared__ u32 s_carry;
    const u32 T = num_tiles(*n_ptr, ipt), S = row_stride(T);
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint4* row = reinterpret_cast<uint4*>(cnt + (u64)blockIdx.x * S);
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 q0 = 0; q0 * 4 < T; q0 += 1024) {
        const u32 q = q0 + threadIdx.x;                        // quad index; tiles 4q .. 4q+3
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q * 4 < T) v = row[q];                             // padding tiles are never written: mask them
        if (q * 4 + 1 >= T) v.y = 0;
        if (q * 4 + 2 >= T) v.z = 0;
        if (q * 4 + 3 >= T) v.w = 0;
        if (q * 4 >= T) v.x = 0;
        const u32 sum = v.x + v.y + v.z + v.w;
        const u32 inc = wave_incl_scan32(sum, lane);
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        u32 off = s_carry;
        for (int i = 0; i < w; ++i) off += s_w[i];
        const u32 e0 = off + inc - sum;
        if (q * 4 < T) row[q] = make_uint4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bin_tot[blockIdx.x] = s_carry;
}

// match-any over the 8 digit bits: mask of the lanes holding the same digit.
// Per bit: one sign-extending bit-field extract (0 / -1), one compare that yields the ballot,
// and one v_bitop3 per mask half computing  m & ~(ballot ^ ext)  (truth table 0x90).
__device__ __forceinline__ void match_digit(u32 d, u32& mlo, u32& mhi) {
    mlo = ~0u; mhi = ~0u;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int ext = __builtin_amdgcn_sbfe((int)d, b, 1);
        const u64 bal = __builtin_amdgcn_ballot_w64(ext != 0);
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (u32)bal, (u32)ext, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (u32)(bal >> 32), (u32)ext, 0x90);
    }
}

// digit of a pass: byte-aligned shifts are template parameters (one v_bfe on the right key half); SHIFT < 0 takes the
// shift at run time (digit grids that do not start on a byte boundary, see sort_low_bit() in umi_engine.hip)
template <int SHIFT>
__device__ __forceinline__ u32 digit_of(u64 key, u32 rshift) {
    if constexpr (SHIFT >= 0) return (u32)(key >> SHIFT) & 255u;
    else return (u32)(key >> rshift) & 255u;
}

// scatter: stable within the tile (wave-major, item, lane == memory order).  FULL tiles skip every
// bounds check (only the last tile of a pass is partial).
// VALS: every key carries a 64-bit value (vin -> vout) that moves with it (keys wider than 64 bits: the group is sorted, the
// rest of the key rides along); the values take a second tile-sized LDS array behind everything else
// g_off (threads < 256): where in bin `tid` this tile's keys of that digit begin — asked for by the caller ahead of the call,
// needed only after the ranking.  Returns (threads < 256) the tile's number of keys with digit `tid`.
template <int SHIFT, bool FULL, bool SEG, bool VALS = false>
__device__ __forceinline__ u32 scatter_tile(const u64* __restrict__ in, u64* __restrict__ out, u64 base, u32 n_valid,
                                            u32 tile, const u32 g_off, const u32* __restrict__ bin_tot,
                                            const int ipt, unsigned char* smem, const u32 rshift, const SegMap seg, const int tid,
                                            const bool streams, u64* stamps = nullptr,
                                            const u64* __restrict__ vin = nullptr, u64* __restrict__ vout = nullptr) {
#ifdef FASTF_STAMPS
#define STAMP(i) do { if (stamps && threadIdx.x == 0) stamps[(u64)tile * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    STAMP(0);
    u64* s_keys = reinterpret_cast<u64*>(smem);                                    // SORT_TILE keys
    u32* s_whist = reinterpret_cast<u32*>(smem + (size_t)ipt * SORT_THREADS * 8);  // [WAVES][256]
    u32* s_delta = s_whist + SORT_WAVES * RADIX;                                   // [256] global - local start
    u32* s_start = s_delta + RADIX;                                                // [256] local bin start
    u32* s_wtot  = s_start + RADIX;                                                // [4] tile-local, [4] global
    u64* s_vals  = reinterpret_cast<u64*>(s_wtot + 16);                            // VALS: SORT_TILE values (8-byte aligned: all the sizes above are multiples of 8)
    const int lane = tid & (WAVE - 1), w = tid >> 6;       // (tid comes in opaque, see scatter_kernel)

    for (int i = tid; i < SORT_WAVES * RADIX; i += SORT_THREADS) s_whist[i] = 0;
    const u32 g_tot = tid < RADIX ? bin_tot[tid] : 0u;     // keys of the whole pass with digit tid → bin base by a scan below

    // wave-striped load: wave w owns [w*IPT*64, (w+1)*IPT*64) of the tile
    u64 key[SORT_IPT];
    const u32 wbase = (u32)w * (u32)ipt * WAVE;
    if constexpr (SEG) {                                       // first pass over the segmented output of the streaming K1b
        const TileSeg ts = seg.tile_seg[tile];
#pragma unroll
        for (int j = 0; j < SORT_IPT; ++j) {
            const u32 li = wbase + j * WAVE + lane;
            key[j] = (j < ipt && (FULL || li < n_valid)) ? ld_once<FASTF_NT_SORT != 0>(in + seg_phys(seg, base + li, ts)) : ~0ULL;
        }
    } else if (streams) {                                      // block-uniform
#pragma unroll
        for (int j = 0; j < SORT_IPT; ++j) {
            const u32 li = wbase + j * WAVE + lane;
            key[j] = (j < ipt && (FULL || li < n_valid)) ? ld_once<FASTF_NT_SORT != 0>(in + base + li) : ~0ULL;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SORT_IPT; ++j) {
            const u32 li = wbase + j * WAVE + lane;
            key[j] = (j < ipt && (FULL || li < n_valid)) ? in[base + li] : ~0ULL;
        }
    }
    u64 val[SORT_IPT];
    if constexpr (VALS) {
#pragma unroll
        for (int j = 0; j < SORT_IPT; ++j) {
            const u32 li = wbase + j * WAVE + lane;
            val[j] = (j < ipt && (FULL || li < n_valid)) ? vin[base + li] : 0;
        }
    }
    __syncthreads();
#ifdef FASTF_STAMPS
    asm volatile("" :: "v"(key[0]));
    __builtin_amdgcn_s_waitcnt(0);
#endif
    STAMP(1);

    // rank inside the wave: per-wave histogram, no atomics (one leader per digit group)
    u32 rnk[SORT_IPT];
    u32* wh = s_whist + w * RADIX;
#pragma unroll
    for (int j = 0; j < SORT_IPT; ++j) {
        if (j >= ipt) break;                               // wave-uniform
        const u32 li = wbase + j * WAVE + lane;
        const u32 d = (FULL || li < n_valid) ? digit_of<SHIFT>(key[j], rshift) : 255u;
        u32 mlo, mhi;
        match_digit(d, mlo, mhi);
        const u32 before = wh[d];
        const u32 r = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        __builtin_amdgcn_wave_barrier();
        if (r == 0) wh[d] = before + (u32)__popc(mlo) + (u32)__popc(mhi);
        __builtin_amdgcn_wave_barrier();
        rnk[j] = before + r;
    }
    __syncthreads();
    STAMP(2);

    // per digit: exclusive offsets of the waves, tile totals → local bin starts
    u32 run = 0, inc = 0, ginc = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int i = 0; i < SORT_WAVES; ++i) { const u32 c = s_whist[i * RADIX + tid]; s_whist[i * RADIX + tid] = run; run += c; }
        inc = wave_incl_scan32(run, lane);                 // over the 64 digits of this wave
        ginc = wave_incl_scan32(g_tot, lane);
        if (lane == WAVE - 1) { s_wtot[w] = inc; s_wtot[4 + w] = ginc; }
    }
    __syncthreads();
    if (tid < RADIX) {
        u32 o = 0, go = 0;
        for (int i = 0; i < w; ++i) { o += s_wtot[i]; go += s_wtot[4 + i]; }
        const u32 lstart = o + inc - run;                  // first tile-local slot of digit tid
        s_start[tid] = lstart;
        s_delta[tid] = (go + ginc - g_tot) + g_off - lstart;   // global slot = bin base + offset in the bin + local slot - local start
    }
    __syncthreads();
    STAMP(3);

    // tile-local reorder through LDS
#pragma unroll
    for (int j = 0; j < SORT_IPT; ++j) {
        if (j >= ipt) break;
        const u32 li = wbase + j * WAVE + lane;
        const u32 d = (FULL || li < n_valid) ? digit_of<SHIFT>(key[j], rshift) : 255u;
        const u32 p = s_start[d] + wh[d] + rnk[j];
        s_keys[p] = key[j];
        if constexpr (VALS) s_vals[p] = val[j];
    }
    __syncthreads();
    STAMP(4);

    // coalesced write-out: consecutive threads → consecutive slots of the same bin
#pragma unroll
    for (int j = 0; j < SORT_IPT; ++j) {
        if (j >= ipt) break;
        const u32 pidx = j * SORT_THREADS + tid;
        if (FULL || pidx < n_valid) {
            const u64 k = s_keys[pidx];
            const u32 d = digit_of<SHIFT>(k, rshift);
            out[(u64)(u32)(s_delta[d] + pidx)] = k;
            if constexpr (VALS) vout[(u64)(u32)(s_delta[d] + pidx)] = s_vals[pidx];
        }
    }
    STAMP(5);
#ifdef FASTF_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(6);
#endif
#undef STAMP
    return run;
}

// SEG: the input is the segmented key buffer of the streaming K1b (first pass only; its own instantiation, so that the
// map lookup does not cost the other passes registers: 47 VGPRs keep four workgroups on a CU)
template <int SHIFT, bool SEG = false, bool VALS = false>
__global__ __launch_bounds__(SORT_THREADS, VALS ? 4 : 8) void scatter_kernel(const u64* __restrict__ in, u64* __restrict__ out,
                                                               const u64* __restrict__ n_ptr,
                                                               const u32* __restrict__ off, const u32* __restrict__ bin_tot,
                                                               u32 ipt, u32 rshift, const SegMap seg, u64* stamps,
                                                               const u64* __restrict__ vin = nullptr, u64* __restrict__ vout = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u64 n = *n_ptr;
    const u32 T = num_tiles(n, ipt);
    // Workgroups are dealt to the 8 XCDs round-robin by blockIdx.  Consecutive tiles write consecutive runs of every
    // bin, so neighbouring tiles share the cache lines at their run boundaries: give each XCD a contiguous range of
    // tiles and those partial lines merge in that XCD's L2 instead of going to HBM twice.
    const u32 tile_keys = ipt * SORT_THREADS;
    const bool streams = sort_streams(n);
    for (u32 r = blockIdx.x >> 3;; r += gridDim.x >> 3) {
#ifdef FASTF_NO_XCD_SWIZZLE
        const u32 tile = (r << 3) | (blockIdx.x & 7u);
#else
        const u32 tile = xcd_tile(T, r);
#endif
        if (tile >= T) return;                               // block-uniform
        const u64 base = (u64)tile * tile_keys;
        const u32 n_valid = (u32)((n - base) < (u64)tile_keys ? (n - base) : (u64)tile_keys);
        // The thread index is made opaque per tile: otherwise every per-thread address of the tile body is hoisted out of
        // the loop and stays live across it (75 VGPRs, three workgroups per CU, instead of 47 and four).
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        // this tile's global bin offsets: issued now, needed only after the ranking
        const u32 g_off = tid < RADIX ? off[(u64)tid * row_stride(T) + tile] : 0u;
        // (no barrier between tiles: whatever a tile reads last from LDS is rewritten only after two barriers of the next)
        if (n_valid == tile_keys) (void)scatter_tile<SHIFT, true, SEG, VALS>(in, out, base, n_valid, tile, g_off, bin_tot, (int)ipt, smem, rshift, seg, tid, streams, stamps, vin, vout);
        else (void)scatter_tile<SHIFT, false, SEG, VALS>(in, out, base, n_valid, tile, g_off, bin_tot, (int)ipt, smem, rshift, seg, tid, streams, stamps, vin, vout);
        if constexpr (VALS) __syncthreads();                 // the value array is read last and rewritten before the next tile's second barrier
    }
}

// ------------------------------------------------------------------------------------
// The FIRST pass of a sort over the regions the streaming K1b left (filter_pack_stream_kernel): no counting pass.  K1b kept a
// histogram of this pass's digit per region (rgn_hist[r][256]); two small kernels turn those into exclusive offsets over the
// regions, per digit, and the bin totals — region-major, so a workgroup flushes and fetches one contiguous KB — and
// scatter_regions_kernel has every workgroup WALK a region tile by tile, carrying the count of each digit it has sent out so
// far: a tile's place in a bin is the region's offset plus what the tiles of the region in front of it put there.  The order
// of the keys inside a bin is region by region, tile by tile: as good as any other for the first pass of an LSD sort.
// Round 4 read all keys once more to count them per tile (tile_count_kernel<true>, 0.074 ms of the 2.1 ms step on configs[2]).
// ------------------------------------------------------------------------------------
constexpr u32 RGN_BLK = 64;            // regions per workgroup of the offset scan
static_assert(K1S_BINS == RADIX, "K1b's region histograms are histograms of a sort digit");
// per block of RGN_BLK regions: exclusive offsets inside the block (in place), the block's totals per digit
__global__ __launch_bounds__(RADIX) void rgn_scan_blocks_kernel(u32* __restrict__ hist, u32 R, u32* __restrict__ blk_sum) {
    const u32 d = threadIdx.x, r0 = blockIdx.x * RGN_BLK, r1 = r0 + RGN_BLK < R ? r0 + RGN_BLK : R;
    u32 acc = 0;
    for (u32 r = r0; r < r1; r += 8) {                       // eight loads in flight, then the running sum
        u32 v[8];
#pragma unroll
        for (u32 k = 0; k < 8; ++k) v[k] = r + k < r1 ? hist[(u64)(r + k) * RADIX + d] : 0u;
#pragma unroll
        for (u32 k = 0; k < 8; ++k) { if (r + k < r1) hist[(u64)(r + k) * RADIX + d] = acc; acc += v[k]; }
    }
    blk_sum[(u64)blockIdx.x * RADIX + d] = acc;
}
// the blocks in front added in; workgroup 0 also leaves the totals per digit (the bin sizes of the pass)
__global__ __launch_bounds__(RADIX) void rgn_scan_fix_kernel(u32* __restrict__ hist, u32 R, const u32* __restrict__ blk_sum, u32 n_blk,
                                                             u32* __restrict__ bin_tot) {
    const u32 d = threadIdx.x, w = blockIdx.x;
    u32 off = 0, tot = 0;
    for (u32 i = 0; i < n_blk; ++i) { const u32 v = blk_sum[(u64)i * RADIX + d]; off += i < w ? v : 0u; tot += v; }
    if (w == 0) bin_tot[d] = tot;
    if (off == 0) return;
    const u32 r0 = w * RGN_BLK, r1 = r0 + RGN_BLK < R ? r0 + RGN_BLK : R;
    for (u32 r = r0; r < r1; ++r) hist[(u64)r * RADIX + d] += off;
}
// region r: rgn_count[r] keys from keys + rgn_phys[r]; hist_excl[r][d]: keys with digit d in the regions in front of r
__global__ __launch_bounds__(SORT_THREADS, 8) void scatter_regions_kernel(const u64* __restrict__ keys, u64* __restrict__ out,
                                                                           const u64* __restrict__ rgn_phys, const u64* __restrict__ rgn_count,
                                                                           const u32* __restrict__ hist_excl, const u32* __restrict__ bin_tot,
                                                                           u32 R, u32 ipt, u32 rshift, bool streams) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 tile_keys = ipt * SORT_THREADS;
    const SegMap none{nullptr, nullptr, 0, 0};
    // neighbouring regions write neighbouring runs of every bin: a contiguous range of regions per XCD, as for the tiles
    for (u32 i = blockIdx.x >> 3;; i += gridDim.x >> 3) {
        const u32 r = xcd_tile(R, i);
        if (r >= R) return;                                  // block-uniform
        const u64 cnt = rgn_count[r];
        const u64* const in = keys + rgn_phys[r];
        u32 acc = threadIdx.x < RADIX ? hist_excl[(u64)r * RADIX + threadIdx.x] : 0u;
        for (u64 base = 0; base < cnt; base += tile_keys) {
            const u32 n_valid = (u32)(cnt - base < (u64)tile_keys ? cnt - base : (u64)tile_keys);
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));                    // (see scatter_kernel: keeps the tile body's addresses out of the loop's live set)
            u32 run;
            if (n_valid == tile_keys) run = scatter_tile<-1, true, false>(in, out, base, n_valid, r, acc, bin_tot, (int)ipt, smem, rshift, none, tid, streams);
            else run = scatter_tile<-1, false, false>(in, out, base, n_valid, r, acc, bin_tot, (int)ipt, smem, rshift, none, tid, streams);
            acc += run;
        }
    }
}
// histograms of regions whose keys are already in memory (an engine that is pushed to again after a finish: rare)
__global__ __launch_bounds__(256) void rgn_hist_kernel(const u64* __restrict__ keys, const u64* __restrict__ rgn_phys, const u64* __restrict__ rgn_count,
                                                       u32 R, u32 shift, u32* __restrict__ hist) {
    __shared__ u32 s_h[RADIX];
    for (u32 r = blockIdx.x; r < R; r += gridDim.x) {
        s_h[threadIdx.x] = 0;
        __syncthreads();
        const u64 cnt = rgn_count[r];
        const u64* const in = keys + rgn_phys[r];
        for (u64 i = threadIdx.x; i < cnt; i += 256) atomicAdd(&s_h[(u32)(in[i] >> shift) & 255u], 1u);
        __syncthreads();
        hist[(u64)r * RADIX + threadIdx.x] = s_h[threadIdx.x];
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------
// K3 / K3u over FULLY SORTED keys — reduce_windows_kernel — the keys are read ONCE.
//   UMI_ROWS = false: one row per (cell, feature); count = distinct non-NULL umi keys (a key is new iff it differs from its
//                     neighbour in front): the exact fallback when the group-only path raises ERR_RUN_TOO_LONG
//   UMI_ROWS = true : one row per distinct key;    count = copies of that key (-u)
// (The matrix normally comes from keys sorted on (cell, feature) only: reduce_hashed_kernel below.)
// Each workgroup owns a contiguous chunk of the key array and walks it in windows of K3_TILE keys.  A window always
// starts at the head of a group and is CUT at a head: the group that is still open at the end of a window is not
// processed, the next window starts at its head (those few keys are read again: they are the tail of a tile this
// workgroup has just touched).  A chunk starts at the first head at or after its nominal start and ends at the first
// head at or after its nominal end, so groups never straddle windows or chunks: no carries, no fix-up pass, and the row
// count of a chunk is known to the workgroup alone.  Rows therefore go to a REGION per workgroup (the slots of its own
// keys: a chunk has at most as many rows as keys) with coalesced plain stores, span_rows[] keeps the counts,
// span_scan_kernel turns them into row bases + the total, and rows_gather_kernel concatenates the regions wherever the
// rows are wanted — into device arrays, or straight into pinned host memory, where it IS the device-to-host copy.
// Traffic: 8 bytes per key in, 12 (16 for -u) bytes per row out: SURVEY 8d's K3 bytes.
// A group longer than a window (only one head in it) is carried as an "open row" across windows.
// ------------------------------------------------------------------------------------
struct ReduceParams {
    const u64* keys; const u64* n_ptr;
    KeyLayout L; u32 feat_mask;
    u64* err;                              // error bits (ERR_RUN_TOO_LONG)
    u32* feature; u32* cell; u32* count;   // row REGIONS (capacity: one slot per key); UMI_ROWS: feature/cell unused
    u64* ukeys;                            // UMI_ROWS only
    u32* span_rows;                        // rows per workgroup chunk [gridDim.x]
    // DEDUP 2: groups longer than a window are counted by giant_groups_kernel: one work item per (group, hash partition)
    u64* giant_list; u32* giant_n;         // items {start, len, row slot, partition | partitions << 32} (GIANT_ITEM_WORDS u64 each)
    u32 giant_max;                         // longest group handed to giant_groups_kernel (beyond: ERR_RUN_TOO_LONG)
    // keys wider than 64 bits (reduce_hashed_kernel<true, true>): keys[] = the group (cell << wide_feat_bits | feature), sorted;
    // vals[] = the rest of each key (NULL flag, UMI, length: the low feat_shift bits of a narrow key), in the keys' order
    const u64* vals; u32 wide_feat_bits;
};

// a (cell, feature) group longer than a window (DEDUP 2) is cut into hash partitions of about GIANT_PART keys: each partition
// is one work item of giant_groups_kernel, which counts its distinct keys in an LDS hash set of full keys; longer than
// GIANT_MAX (or more than GIANT_LIST_CAP items in one launch) raises ERR_RUN_TOO_LONG: sort fully instead
constexpr u32 GIANT_MAX = 1u << 16, GIANT_PART = 1536, GIANT_LIST_CAP = 4096, GIANT_ITEM_WORDS = 4, GIANT_TAB = 4096;

// 4 keys per thread (2048-key windows): 324 us vs 352 us for 8 on the configs[2] shape (two-pass form of round 2)
#ifndef FASTF_K3_IPT
#define FASTF_K3_IPT 4
#endif
#ifndef FASTF_K3_THREADS
#define FASTF_K3_THREADS 512
#endif
constexpr int K3_THREADS = FASTF_K3_THREADS, K3_IPT = FASTF_K3_IPT, K3_TILE = K3_THREADS * K3_IPT, K3_WAVES = K3_THREADS / WAVE;
constexpr int K3_UNITS = K3_IPT * K3_WAVES;
static_assert(K3_UNITS <= WAVE, "one wave scans the (item, wave) units");

// nominal chunk of workgroup b of G: whole tiles, spread evenly; [start, end) in keys
__host__ __device__ __forceinline__ u64 k3_chunk_start(u64 n, u32 b, u32 G) {
    const u64 T = (n + K3_TILE - 1) / K3_TILE;
    const u64 s = (T * b / G) * K3_TILE;
    return s < n ? s : n;
}

__device__ __forceinline__ u32 wave_min32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)v, o, WAVE); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ u32 wave_max32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)v, o, WAVE); v = t > v ? t : v; }
    return v;
}

// workgroup barrier that waits for this wave's LDS traffic only (not for its global loads and stores)
__device__ __forceinline__ void k3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// register budget: 64 VGPRs (eight waves per SIMD, four workgroups per CU)
#ifndef FASTF_K3_PD32
typedef unsigned short k3_pd_t;
#else
typedef u32 k3_pd_t;
#endif
static_assert(K3_TILE <= 65535 || sizeof(k3_pd_t) == 4, "distinct prefixes of a window must fit k3_pd_t");
template <bool UMI_ROWS>
__global__ __launch_bounds__(K3_THREADS, 8) void reduce_windows_kernel(const ReduceParams p) {
    // (item, wave) units in window order: heads / distinct flags per unit, then their exclusive scans
    __shared__ u32 s_h[K3_UNITS], s_d[K3_UNITS];
    __shared__ u64 s_hb[K3_UNITS];         // head ballots of the units (the cut is found from these)
    __shared__ u32 s_tot[2];
    __shared__ k3_pd_t s_pd[K3_TILE + 2];  // by local row: distinct-prefix at the head (at most K3_TILE: 16 bits do)
    __shared__ u64 s_id[K3_TILE];          // the window's keys, later the row identities (UMI_ROWS: the key; else (cell << 32) | feature)
    __shared__ u32 s_first;                // chunk start search

    // everything that is the same for the whole wave is kept in scalar registers (readfirstlane): the wave index, what
    // comes back from LDS broadcasts and cross-lane reductions, the open row — left to itself the compiler keeps those
    // in vector registers and the kernel spills at its 64-VGPR budget
    const int tid = threadIdx.x, lane = lane_id(), w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 n = *p.n_ptr;
    const u32 gshift = UMI_ROWS ? 0u : p.L.feat_shift;
    const u32 nn_shift = p.L.umi_bits + p.L.len_bits;
    const u32 G = gridDim.x, b = blockIdx.x;
    const u64 nom_start = k3_chunk_start(n, b, G), nom_end = k3_chunk_start(n, b + 1, G);
    u32 rows_so_far = 0;                   // (uniform) rows of this chunk written so far

    // ---- chunk start: the first head at or after the nominal start (a group that began earlier belongs to the chunk before) ----
    u64 cursor = nom_start;
    if (b > 0 && nom_start < nom_end) {
        for (;;) {
            if (tid == 0) s_first = ~0u;
            __syncthreads();
            const u64 idx = cursor + tid;
            bool head = false;
            if (idx < nom_end) head = idx == 0 || (p.keys[idx] >> gshift) != (p.keys[idx - 1] >> gshift);
            const u64 m = __ballot(head);
            if (m && lane == 0) __hip_atomic_fetch_min(&s_first, (u32)(w * WAVE + __builtin_ctzll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            const u32 f = __builtin_amdgcn_readfirstlane(s_first);
            __syncthreads();
            if (f != ~0u) { cursor += f; break; }
            cursor += K3_THREADS;
            if (cursor >= nom_end) break;
        }
    }
    if (cursor >= nom_end) { if (tid == 0) p.span_rows[b] = 0; return; }   // no group starts in this chunk (or it is empty)

    const u64 region = nom_start;          // rows of this chunk go to the slots of its own keys
    bool open_valid = false;               // (uniform) a group longer than a window is being carried
    u64 open_id = 0; u32 open_cnt = 0;
    bool done = false;
    // The loop is a chain of phases separated by barriers; what crosses a barrier is in LDS only, so the barriers wait for
    // LDS traffic alone (k3_barrier) and global loads and stores stay in flight across them: the keys of the NEXT window are
    // requested as soon as the cut of this one is known and arrive while its rows are scanned, staged and written, and the
    // row stores are never waited for.  (With __syncthreads each window paid three memory round trips one after the other:
    // its keys, the key in front of it, its stores.)
    u64 nkey[K3_IPT];                      // keys of the window about to be processed
    {
        const u32 W0 = (u32)(n - cursor < (u64)K3_TILE ? n - cursor : (u64)K3_TILE);
#pragma unroll
        for (int j = 0; j < K3_IPT; ++j) {
            const u32 loc = (u32)j * K3_THREADS + tid;
            nkey[j] = loc < W0 ? ld_once<FASTF_NT_K3 != 0>(p.keys + cursor + loc) : 0;
        }
    }
    u64 prev0 = cursor > 0 ? p.keys[cursor - 1] : 0;       // (uniform) the key in front of the window
    while (!done) {
        const u64 base = cursor;
        const u32 W = (u32)(n - base < (u64)K3_TILE ? n - base : (u64)K3_TILE);
        u64 key[K3_IPT], hm[K3_IPT], dm[K3_IPT];
#pragma unroll
        for (int j = 0; j < K3_IPT; ++j) {
            key[j] = nkey[j];
            s_id[(u32)j * K3_THREADS + tid] = key[j];
        }
        k3_barrier();
#pragma unroll
        for (int j = 0; j < K3_IPT; ++j) {
            const u32 loc = (u32)j * K3_THREADS + tid;
            const u64 idx = base + loc;
            const bool valid = loc < W;
            const u64 k = key[j];
            const u64 prev = (valid && idx > 0) ? (loc > 0 ? s_id[loc - 1] : prev0) : ~k;
            const bool head = valid && (idx == 0 || (k >> gshift) != (prev >> gshift));
            bool dist;
            if (UMI_ROWS) dist = valid;
            else {
                dist = valid && ((k >> nn_shift) & 1);
                if (dist && idx > 0) dist = k != prev;                 // fully sorted keys: new iff it differs from the key in front
            }
            hm[j] = __ballot(head); dm[j] = __ballot(dist);
            if (lane == 0) s_hb[j * K3_WAVES + w] = hm[j];
        }
        k3_barrier();
        // ---- the cut (every wave works it out for itself from the 32 head ballots) ----
        //   stop: the first head at or beyond the chunk's nominal end — it and everything after it belong to the next chunk
        //   else, at the end of the data, the whole window; else the last head of the window (its group is left to the
        //   next window); else (one group fills the window) the whole window, the group stays open
        const u32 lim = nom_end > base ? (u32)(nom_end - base < (u64)K3_TILE ? nom_end - base : (u64)K3_TILE) : 0u;
        u32 c_stop = ~0u, c_last = 0;
        if (lane < K3_UNITS) {
            const u64 hb = s_hb[lane];
            const u32 ub = (u32)(lane / K3_WAVES) * K3_THREADS + (u32)(lane % K3_WAVES) * WAVE;    // first position of the unit
            u64 at_or_after = hb;
            if (lim > ub) at_or_after = lim - ub >= 64 ? 0ull : hb & ~((1ull << (lim - ub)) - 1);
            if (at_or_after) c_stop = ub + (u32)__builtin_ctzll(at_or_after);
            const u64 not_first = ub == 0 ? hb & ~1ull : hb;
            if (not_first) c_last = ub + 63u - (u32)__builtin_clzll(not_first);
        }
        c_stop = __builtin_amdgcn_readfirstlane(wave_min32(c_stop)); c_last = __builtin_amdgcn_readfirstlane(wave_max32(c_last));
        u32 cut; bool closed;
        if (c_stop != ~0u) { cut = c_stop; closed = true; done = true; }
        else if (base + W == n) { cut = W; closed = true; done = true; }
        else if (c_last > 0) { cut = c_last; closed = true; }
        else { cut = W; closed = false; }
        // the next window starts at the cut: request its keys now, and keep the key in front of it (s_id is restaged below)
        if (!done) {
            const u64 nb = base + cut;
            const u32 Wn = (u32)(n - nb < (u64)K3_TILE ? n - nb : (u64)K3_TILE);
#pragma unroll
            for (int j = 0; j < K3_IPT; ++j) {
                const u32 loc = (u32)j * K3_THREADS + tid;
                nkey[j] = loc < Wn ? ld_once<FASTF_NT_K3 != 0>(p.keys + nb + loc) : 0;
            }
            const u64 pk = s_id[cut - 1];                  // cut >= 1 whenever the loop goes on
            prev0 = uniform64(pk);
        }
        // keys at or beyond the cut are not this window's
#pragma unroll
        for (int j = 0; j < K3_IPT; ++j) {
            const u32 ub = (u32)j * K3_THREADS + (u32)w * WAVE;
            const u64 below = cut <= ub ? 0ull : (cut - ub >= 64 ? ~0ull : (1ull << (cut - ub)) - 1);
            hm[j] &= below; dm[j] &= below;
            if (lane == 0) { s_h[j * K3_WAVES + w] = (u32)__popcll(hm[j]); s_d[j * K3_WAVES + w] = (u32)__popcll(dm[j]); }
        }
        k3_barrier();
        if (w == 0) {
            const u32 h = lane < K3_UNITS ? s_h[lane] : 0u, d = lane < K3_UNITS ? s_d[lane] : 0u;
            const u32 hi = wave_incl_scan32(h, lane), di = wave_incl_scan32(d, lane);
            if (lane < K3_UNITS) { s_h[lane] = hi - h; s_d[lane] = di - d; }
            if (lane == WAVE - 1) { s_tot[0] = hi; s_tot[1] = di; s_pd[hi] = (k3_pd_t)di; }   // sentinel: all distinct flags of the window
        }
        k3_barrier();
        const u32 n_rows = __builtin_amdgcn_readfirstlane(s_tot[0]), d_all = __builtin_amdgcn_readfirstlane(s_tot[1]);
#pragma unroll
        for (int j = 0; j < K3_IPT; ++j) {
            if ((hm[j] >> lane) & 1) {
                const u32 r = s_h[j * K3_WAVES + w] + rank_below(hm[j]);
                s_pd[r] = (k3_pd_t)(s_d[j * K3_WAVES + w] + rank_below(dm[j]));
                s_id[r] = UMI_ROWS ? key[j]
                                   : (((u64)(u32)(key[j] >> p.L.cell_shift)) << 32) | ((u32)(key[j] >> p.L.feat_shift) & p.feat_mask);
            }
        }
        k3_barrier();
        // distinct flags in front of the window's first head belong to the open row (none unless a group is being carried)
        const u32 lead = n_rows ? (u32)__builtin_amdgcn_readfirstlane((int)s_pd[0]) : d_all;
        if (open_valid) open_cnt += lead;
        u32 first_row = 0, last_row = n_rows;              // rows [first_row, last_row) of the window are written now
        if (open_valid && (n_rows > 0 || done)) {
            // the carried group has ended: its row goes out in front of the window's own
            if (tid == 0) {
                const u64 at = region + rows_so_far;
                p.count[at] = open_cnt;
                if (UMI_ROWS) p.ukeys[at] = open_id;
                else { p.feature[at] = (u32)open_id; p.cell[at] = (u32)(open_id >> 32); }
            }
            rows_so_far += 1; open_valid = false;
        }
        if (!closed && n_rows > 0) {                       // one group fills the window and goes on: carry it
            last_row = n_rows - 1;
            const u64 oid = s_id[last_row];
            open_valid = true;
            open_id = uniform64(oid);
            open_cnt = (u32)__builtin_amdgcn_readfirstlane((int)((u32)s_pd[n_rows] - (u32)s_pd[last_row]));
        }
        const u64 row_base = region + rows_so_far;
        for (u32 r = first_row + tid; r < last_row; r += K3_THREADS) {
            const u32 c = (u32)s_pd[r + 1] - (u32)s_pd[r];
            const u64 id = s_id[r];
            p.count[row_base + r] = c;
            if (UMI_ROWS) p.ukeys[row_base + r] = id;
            else { p.feature[row_base + r] = (u32)id; p.cell[row_base + r] = (u32)(id >> 32); }
        }
        rows_so_far += last_row - first_row;
        cursor = base + cut;
        k3_barrier();                                  // the next window restages the shared arrays
    }
    if (open_valid) {                                  // (cannot happen: the last window closes the group — kept as a guard)
        if (tid == 0) {
            const u64 at = region + rows_so_far;
            p.count[at] = open_cnt;
            if (UMI_ROWS) p.ukeys[at] = open_id;
            else { p.feature[at] = (u32)open_id; p.cell[at] = (u32)(open_id >> 32); }
        }
        rows_so_far += 1;
    }
    if (tid == 0) p.span_rows[b] = rows_so_far;
}

// ------------------------------------------------------------------------------------
// K3h: the matrix rows from keys sorted on (cell, feature) ONLY — reduce_hashed_kernel.
// Same chunks, windows and row regions as reduce_windows_kernel (a window starts at a group head and is cut at one, so
// no group straddles windows or chunks), rebuilt around what the PMC counters of round 3 said about the hash mode of that
// kernel: it was bound by instruction issue (134 scalar + 110 vector + 20 LDS instructions per 64 keys, five barriers per
// window), not by bytes.  Here
//   * a wave owns 256 CONSECUTIVE keys of the window: the neighbour in front comes over DPP (wave_shr:1), the key in front
//     of the wave from one extra load — the keys are never staged in LDS;
//   * a key's group is named by its ROW RANK r inside the window (heads in front of it: two mbcnt), which is also where
//     its count goes;
//   * the window-local set is EXACT in one 32-bit word per slot: x = the key bits below feat_shift (NULL flag, UMI,
//     length; at most 27 bits) splits into lo (12 bits) and hi; slot = (lo ^ f(hi) ^ g(r)) + i * step(hi) and the slot
//     holds (hi, r, i + 1): two keys that meet in a slot with the same word have the same r, hi, i and home slot, hence
//     the same lo — the same key of the same group.  One compare-and-swap per probe: 0 back = first occurrence, the own
//     word back = duplicate, anything else = another key, next slot.  No second read, no 64-bit multiply.
//     (UMIs beyond 12 bases: 64-bit slots holding (r, x) whole — SLOT64.)
//   * counts: D = distinct keys of the unit in front of a lane (mbcnt).  A head lane adds +D to the row in front of its
//     own and -D to its own, the unit adds its total to its last row: every row ends up with D(next head) - D(own head),
//     across units and waves, through LDS atomics (a head may add its -D before the +D of its successor arrives: the sums are
//     exact modulo 2^32);
//   * two barriers per window (heads published / rows complete), both waiting for LDS traffic only.
// Groups longer than a window go to giant_groups_kernel as before.
// ------------------------------------------------------------------------------------
constexpr int K3H_THREADS = 512, K3H_WAVES = K3H_THREADS / WAVE, K3H_IPT = 4, K3H_TILE = K3H_THREADS * K3H_IPT, K3H_UNITS = K3H_TILE / WAVE;
constexpr u32 K3H_TAB = 4096, K3H_WAVE_KEYS = K3H_IPT * WAVE;      // (slots of the 64-bit forms; the 32-bit form has K3H_TAB32)
#ifndef FASTF_K3H_TAB32
#define FASTF_K3H_TAB32 8192
#endif
constexpr u32 K3H_TAB32 = FASTF_K3H_TAB32, K3H_LO32 = K3H_TAB32 == 8192 ? 13u : 12u;   // slots of the 32-bit form and the key bits its slot number stands for
static_assert(K3H_TAB32 == 4096 || K3H_TAB32 == 8192, "32-bit slots: (27 - lo) + 11 + 6 bits");
static_assert(K3H_TILE == K3_TILE, "same chunk geometry as reduce_windows_kernel (k3_chunk_start, rows_gather_kernel)");
static_assert(K3H_UNITS == 32 && K3H_TILE <= 2048, "32 head ballots per window; row ranks fit 11 bits");

__device__ __forceinline__ u32 dpp_wave_shr1(u32 lane0, u32 x) {          // lane l gets x of lane l - 1, lane 0 gets lane0
    return (u32)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ u64 readlane64(u64 v, u32 l) {                 // l wave-uniform
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), (int)l) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)v, (int)l);
}
// inclusive sum over each row of 16 lanes (DPP row shifts); lanes 15 and 31 hold the totals of rows 0 and 1
__device__ __forceinline__ u32 row16_incl_sum(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    return x;
}
// Register budget and occupancy (A/B on configs[2], profiles/r4_notes/ab_k3h_occupancy.txt): 64 VGPRs / four workgroups per
// CU spill six registers in the window loop and take 0.215 ms; 80 VGPRs / three per CU 0.200; and with the LDS that frees,
// one 32-bit count per row instead of two 16-bit halves per word (fewer address and shift instructions around every LDS
// atomic) 0.192 ms.  FASTF_K3H_MINW=8 gives the 64-register build back.
#ifndef FASTF_K3H_MINW
#define FASTF_K3H_MINW 6
#endif
__device__ __forceinline__ void k3h_add(u32* cnt, u32 row, u32 v) {        // v may be "negative": sums are exact modulo 2^32
    (void)__hip_atomic_fetch_add(&cnt[row], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// WIDE (keys wider than 64 bits; always with SLOT64): the group word is compared whole, everything below the group comes
// from vals[]; a slot holds (x >> 12, row rank, probe number + 1) — the quotient form of the 32-bit slots with 40 bits of x
template <bool SLOT64, bool WIDE = false>
__global__ __launch_bounds__(K3H_THREADS, WIDE ? 4 : (SLOT64 ? 6 : FASTF_K3H_MINW)) void reduce_hashed_kernel(const ReduceParams p) {
    static_assert(!WIDE || SLOT64, "wide keys use the 64-bit slots");
    typedef typename std::conditional<SLOT64, u64, u32>::type slot_t;
    __shared__ u64 s_hb[K3H_UNITS];        // head ballots of the 32 units of the window
    __shared__ u32 s_cnt[K3H_TILE];        // distinct counts by row; all-zero between windows
    constexpr u32 TAB = SLOT64 ? K3H_TAB : K3H_TAB32;
    __shared__ slot_t s_tab[TAB];          // the set; all-zero between windows
    // (the row identities go to memory straight from the head lanes: the 16 KB two arrays of them took are the set's now —
    //  twice the slots, half the load, shorter probe chains; a wave waits for its longest chain)
    __shared__ u32 s_first;
    __shared__ u64 s_g0;

    const int tid = threadIdx.x, lane = lane_id(), w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 n = *p.n_ptr;
    const u32 gshift = WIDE ? 0u : p.L.feat_shift, nn_shift = p.L.umi_bits + p.L.len_bits;
    const u32 G = gridDim.x, b = blockIdx.x;
    const u64 nom_start = k3_chunk_start(n, b, G), nom_end = k3_chunk_start(n, b + 1, G);
    for (u32 i = tid; i < TAB; i += K3H_THREADS) s_tab[i] = 0;
    for (u32 i = tid; i < K3H_TILE; i += K3H_THREADS) s_cnt[i] = 0;

    // ---- chunk start: the first head at or after the nominal start (a group that began earlier belongs to the chunk before) ----
    u64 cursor = nom_start;
    if (b > 0 && nom_start < nom_end) {
        for (;;) {
            if (tid == 0) s_first = ~0u;
            __syncthreads();
            const u64 idx = cursor + tid;
            bool head = false;
            // (nom_start can be 0 for b > 0: the grid is sized for the caller's bound on the key count, the chunks are cut from
            //  the actual one, and the first few chunks of a small n are empty)
            if (idx < nom_end) head = idx == 0 || (p.keys[idx] >> gshift) != (p.keys[idx - 1] >> gshift);
            const u64 m = __ballot(head);
            if (m && lane == 0) __hip_atomic_fetch_min(&s_first, (u32)(w * WAVE + __builtin_ctzll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            const u32 f = __builtin_amdgcn_readfirstlane(s_first);
            __syncthreads();
            if (f != ~0u) { cursor += f; break; }
            cursor += K3H_THREADS;
            if (cursor >= nom_end) break;
        }
    }
    __syncthreads();                                                       // the cleared arrays
    if (cursor >= nom_end) { if (tid == 0) p.span_rows[b] = 0; return; }   // no group starts in this chunk (or it is empty)
    cursor = uniform64(cursor);

    const u64 region = nom_start;          // rows of this chunk go to the slots of its own keys
    const u32 pbase = (u32)w * K3H_WAVE_KEYS;                              // first window position of this wave
    u32 rows_so_far = 0;
    bool done = false;
    // keys of the window about to be processed (lane l of nkey[j]: position pbase + 64 j + l) and the key in front of the wave
    u64 nkey[K3H_IPT], npk = 0;
    u64 nval[WIDE ? K3H_IPT : 1], npv = 0;                               // WIDE: the rest of each key, and of the key in front of the wave
    auto request = [&](const u64 at) {
        const u32 Wn = (u32)(n - at < (u64)K3H_TILE ? n - at : (u64)K3H_TILE);
#pragma unroll
        for (int j = 0; j < K3H_IPT; ++j) {
            const u32 pos = pbase + (u32)j * WAVE + (u32)lane;
            nkey[j] = pos < Wn ? ld_once<FASTF_NT_K3 != 0>(p.keys + at + pos) : 0;
            if constexpr (WIDE) nval[j] = pos < Wn ? ld_once<FASTF_NT_K3 != 0>(p.vals + at + pos) : 0;
        }
        // the key in front of the wave: the same address in every lane, but asked for as a VECTOR load (an opaque zero in the
        // address) — a scalar load counts on lgkmcnt, and the LDS-only barriers below would wait for its memory round trip
        u32 z = 0;
        asm volatile("" : "+v"(z));
        npk = (w > 0 && pbase < Wn) ? p.keys[at + pbase - 1 + z] : 0;
        if constexpr (WIDE) npv = (w > 0 && pbase < Wn) ? p.vals[at + pbase - 1 + z] : 0;
    };
    request(cursor);
    while (!done) {
        const u64 base = cursor;
        const u32 W = (u32)(n - base < (u64)K3H_TILE ? n - base : (u64)K3H_TILE);
        u64 key[K3H_IPT];
        u64 val[WIDE ? K3H_IPT : 1];
        u32 fl = 0;                                                        // per lane: bit j = key j is a head, bit 4 + j = it equals its neighbour in front
        // ---- heads (and copies of the neighbour in front) from registers ----
        {
            u64 carry = uniform64(npk), vcarry = WIDE ? uniform64(npv) : 0;
#pragma unroll
            for (int j = 0; j < K3H_IPT; ++j) {
                const u32 pos = pbase + (u32)j * WAVE + (u32)lane;
                const u64 k = key[j] = nkey[j];
                const u64 prev = ((u64)dpp_wave_shr1((u32)(carry >> 32), (u32)(k >> 32)) << 32) | dpp_wave_shr1((u32)carry, (u32)k);
                const bool valid = pos < W;
                // (bitwise, not short-circuit: straight-line code instead of an exec-mask branch per term)
                const bool head = valid & ((pos == 0) | (((k ^ prev) >> gshift) != 0));        // position 0 is a head by construction
                const u64 hmj = __ballot(head);
                if (lane == 0) s_hb[w * K3H_IPT + j] = hmj;
                fl |= (head ? 1u : 0u) << j;
                bool same = k == prev;                                     // a copy of the neighbour in front
                if constexpr (WIDE) {
                    const u64 v = val[j] = nval[j];
                    const u64 pv = ((u64)dpp_wave_shr1((u32)(vcarry >> 32), (u32)(v >> 32)) << 32) | dpp_wave_shr1((u32)vcarry, (u32)v);
                    same = v == pv;                                        // (the group is the same: not a head)
                    vcarry = readlane64(v, 63);
                }
                fl |= ((valid & !head & same) ? 16u : 0u) << j;
                carry = readlane64(k, 63);
            }
        }
        k3_barrier();
        // ---- the cut, the row ranks in front of this wave, the rows of the window: every wave from the 32 ballots ----
        //   stop: the first head at or beyond the chunk's nominal end — it and everything after it belong to the next chunk
        //   else, at the end of the data, the whole window; else the last head of the window (its group is left to the
        //   next window); else one group fills the window: giant_groups_kernel counts it
        u32 cut; bool giant = false;
        u32 R0, n_rows;
        {
            const u64 hb = lane < K3H_UNITS ? s_hb[lane] : 0ull;
            const u32 lim = nom_end > base ? (u32)(nom_end - base < (u64)K3H_TILE ? nom_end - base : (u64)K3H_TILE) : 0u;
            const u32 ub = (u32)lane * WAVE;
            u64 at_or_after = 0;                                           // heads at or beyond the chunk's nominal end (its last windows only)
            if (lim < (u32)K3H_TILE) {                                     // (uniform)
                at_or_after = hb;
                if (lim > ub) at_or_after = lim - ub >= 64 ? 0ull : hb & ~((1ull << (lim - ub)) - 1);
            }
            const u64 not_first = lane == 0 ? hb & ~1ull : hb;
            const u64 m_stop = lim < (u32)K3H_TILE ? __ballot(at_or_after != 0) : 0ull, m_last = __ballot(not_first != 0);
            if (m_stop) { const u32 fu = (u32)__builtin_ctzll(m_stop); cut = fu * WAVE + (u32)__builtin_ctzll(readlane64(at_or_after, fu)); done = true; }
            else if (base + W == n) { cut = W; done = true; }
            else if (m_last) { const u32 lu = 63u - (u32)__builtin_clzll(m_last); cut = lu * WAVE + 63u - (u32)__builtin_clzll(readlane64(not_first, lu)); }
            else { cut = W; giant = true; }
            // heads below the cut, per unit; low half: all of them (rows of the window), high half: those in front of this wave
            const u64 below = cut <= ub ? 0ull : (cut - ub >= 64 ? ~0ull : (1ull << (cut - ub)) - 1);
            const u32 c = (u32)__popcll(hb & below);
            const u32 sc = row16_incl_sum(c | (lane < w * K3H_IPT ? c << 16 : 0u));
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)sc, 15) + (u32)__builtin_amdgcn_readlane((int)sc, 31);
            n_rows = tot & 0xFFFFu; R0 = tot >> 16;
        }
        if (giant) {
            // One group fills the window (and the window starts at its head).  Its row goes out with a count of zero, the next
            // head is looked for, and giant_groups_kernel counts the group's distinct keys — one work item per hash partition,
            // added up with atomics on that row's count.
            if (tid == 0) s_g0 = key[0];
            u64 pos = base + W;
            u32 found = ~0u;
            for (;;) {
                if (tid == 0) s_first = ~0u;
                __syncthreads();
                const u64 g0 = s_g0 >> gshift;
                const u64 idx = pos + tid;
                const bool h = idx < n && (p.keys[idx] >> gshift) != g0;
                const u64 m = __ballot(h);
                if (m && lane == 0) __hip_atomic_fetch_min(&s_first, (u32)(w * WAVE + __builtin_ctzll(m)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __syncthreads();
                found = __builtin_amdgcn_readfirstlane(s_first);
                __syncthreads();
                if (found != ~0u || pos + K3H_THREADS >= n) break;
                pos += K3H_THREADS;
            }
            const u64 next = uniform64(found != ~0u ? pos + found : n);
            const u64 len = next - base;
            const u32 parts = (u32)((len + GIANT_PART - 1) / GIANT_PART);
            const u64 row = region + rows_so_far;
            if (tid == 0) {
                const u64 k0 = s_g0;
                p.count[row] = 0;
                p.feature[row] = (WIDE ? (u32)k0 : (u32)(k0 >> p.L.feat_shift)) & p.feat_mask;
                p.cell[row] = (u32)(k0 >> (WIDE ? p.wide_feat_bits : p.L.cell_shift));
            }
            if (len > p.giant_max) { if (tid == 0) atomicOr(p.err, ERR_RUN_TOO_LONG); }
            else {
                if (tid == 0) s_first = atomicAdd(p.giant_n, parts);
                __syncthreads();
                const u32 at = __builtin_amdgcn_readfirstlane(s_first);
                if (at + parts > GIANT_LIST_CAP) { if (tid == 0) atomicOr(p.err, ERR_RUN_TOO_LONG); }
                else {
                    // (strided: the wide path takes groups of up to 4 M reads = 2731 items, more than the workgroup has threads — ADVICE r4)
                    for (u32 t = (u32)tid; t < parts; t += K3H_THREADS) {
                        u64* it = p.giant_list + (u64)(at + t) * GIANT_ITEM_WORDS;
                        it[0] = base; it[1] = len; it[2] = row; it[3] = (u64)t | ((u64)parts << 32);
                    }
                }
                __syncthreads();                                           // s_first is reused by the next search
            }
            rows_so_far += 1;
            cursor = next;
            done = cursor >= nom_end || cursor >= n;
            if (!done) request(cursor);
            continue;
        }
        // the next window starts at the cut: request its keys now
        if (!done) request(base + cut);
        // ---- the set, the counts, the row identities ----
        // per unit j: sl[j] = the slot being tried, in the end the slot this key holds (~0: none); st bit j = the key counts as
        // distinct, bit 4 + j = it still has to be looked up
        u32 sl[K3H_IPT], row[K3H_IPT], st = 0;
        const u64 row_base = region + rows_so_far;                         // the window's rows in the chunk's region
        {
            u32 stp[K3H_IPT]; slot_t wd[K3H_IPT], od[K3H_IPT];
            u32 Ru = R0;
#pragma unroll
            for (int j = 0; j < K3H_IPT; ++j) {
                const u32 pos = pbase + (u32)j * WAVE + (u32)lane;
                const bool valid = pos < cut;
                const bool is_head = valid && ((fl >> j) & 1u);
                const u64 h = __ballot(is_head);
                const bool single = __builtin_amdgcn_inverse_ballot_w64(h & (h >> 1));      // a head whose neighbour behind is a head too
                const u64 k = WIDE ? val[j] : key[j];                                        // where the NULL flag, the UMI and the length are
                const u32 r = Ru + rank_below(h) + (is_head ? 1u : 0u) - 1u;                 // row of the key's group (lanes in front of
                row[j] = r;                                                                  // the unit's first head: the row before)
                const bool dist = valid && ((k >> nn_shift) & 1) && !((fl >> (4 + j)) & 1u);
                const bool probe = dist && !single;
                st |= (dist ? 1u : 0u) << j; st |= (probe ? 16u : 0u) << j;
                if constexpr (WIDE) {
                    const u64 x = k, hi = x >> 12;                         // x < 2^52 (24 bases; longer UMIs: make_val_sub keeps 52 bits here)
                    const u32 f = (u32)hi * 0x9E3779B1u + (u32)(hi >> 32) * 0x85EBCA77u;
                    sl[j] = ((u32)x ^ (f >> 8) ^ __umul24(r, 0x9E5u)) & (TAB - 1);
                    stp[j] = ((f >> 20) & 62u) | 1u;
                    wd[j] = hi | ((u64)r << 40) | (1ull << 51);            // hi: 40 bits, row rank: 11, probe number + 1: 6
                } else if constexpr (SLOT64) {
                    const u64 x = k & ((1ull << gshift) - 1);
                    u32 hh = (u32)x * 0x9E3779B1u + (u32)(x >> 32) * 0x85EBCA77u + r * 0xC2B2AE3Du;
                    hh ^= hh >> 15;
                    sl[j] = hh & (TAB - 1); stp[j] = ((hh >> 20) & 62u) | 1u;
                    wd[j] = (1ull << 63) | ((u64)r << 40) | x;
                } else {
                    const u32 x = (u32)k & ((1u << gshift) - 1u);          // key bits below the group: NULL flag, UMI, length
                    const u32 hi = x >> K3H_LO32, f = __umul24(hi, 0x9E3779u);
                    sl[j] = (x ^ (f >> 8) ^ __umul24(r, 0x9E5u)) & (TAB - 1);
                    stp[j] = ((f >> 20) & 62u) | 1u;                       // odd: the walk visits every slot
                    wd[j] = hi | (r << 15) | (1u << 26);                   // hi: 27 - lo <= 15 bits, row rank: 11, probe number + 1: 6
                }
                Ru += (u32)__popcll(h);
            }
            // first probes of the four units back to back (one LDS round trip for all of them) ...
#pragma unroll
            for (int j = 0; j < K3H_IPT; ++j) {
                od[j] = 0;
                if ((st >> (4 + j)) & 1u) (void)__hip_atomic_compare_exchange_strong(&s_tab[sl[j]], &od[j], wd[j], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // ... then whoever met another key walks on: 0 back = first occurrence (the slot is this key's until the window
            // ends), the own word back = seen before, anything else = another key
            u32 overflow = 0;
#pragma unroll
            for (int j = 0; j < K3H_IPT; ++j) {
                bool go = ((st >> (4 + j)) & 1u) && od[j] != 0 && od[j] != wd[j];
                u32 tries = 1;
#pragma nounroll
                while (go) {
                    sl[j] = (sl[j] + stp[j]) & (TAB - 1);
                    if constexpr (WIDE) wd[j] += 1ull << 51;
                    else if constexpr (!SLOT64) wd[j] += 1u << 26;
                    od[j] = 0;
                    (void)__hip_atomic_compare_exchange_strong(&s_tab[sl[j]], &od[j], wd[j], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    go = od[j] != 0 && od[j] != wd[j] && ++tries < 63;
                }
                const bool tried = (st >> (4 + j)) & 1u;
                const bool mine = tried && od[j] == 0, seen = tried && od[j] == wd[j];
                if (seen) st &= ~(1u << j);
                if (tried && !mine && !seen) overflow = 1;                 // (a table this crowded is not this path's kind of data)
                if (!mine) sl[j] = ~0u;
            }
            if (__any(overflow) && lane == 0) atomicOr(p.err, ERR_RUN_TOO_LONG);
            Ru = R0;
#pragma unroll
            for (int j = 0; j < K3H_IPT; ++j) {
                const u32 pos = pbase + (u32)j * WAVE + (u32)lane;
                const bool is_head = pos < cut && ((fl >> j) & 1u);
                const u64 h = __ballot(is_head);
                const u64 dmj = __ballot((st >> j) & 1u);
                const u32 D = rank_below(dmj);
                const u32 r = row[j];
                if (is_head) {
                    if (D) { k3h_add(s_cnt, r - 1u, D); k3h_add(s_cnt, r, 0u - D); }
                    p.feature[row_base + r] = (WIDE ? (u32)key[j] : (u32)(key[j] >> p.L.feat_shift)) & p.feat_mask;
                    p.cell[row_base + r] = (u32)(key[j] >> (WIDE ? p.wide_feat_bits : p.L.cell_shift));
                }
                const u32 heads = (u32)__popcll(h), tot = (u32)__popcll(dmj);
                if (lane == 0 && tot) k3h_add(s_cnt, Ru + heads - 1u, tot);  // (Ru + heads >= 1: position 0 of the window is a head)
                Ru += heads;
            }
        }
        k3_barrier();
        // ---- the rows' counts out (coalesced; their identities went out from the head lanes), the set and the counts back to all-zero ----
#pragma unroll
        for (int j = 0; j < K3H_IPT; ++j) if (sl[j] != ~0u) s_tab[sl[j]] = 0;
        for (u32 r = tid; r < n_rows; r += K3H_THREADS) {
            p.count[row_base + r] = s_cnt[r];
            s_cnt[r] = 0;
        }
        rows_so_far += n_rows;
        cursor = base + cut;
        // (no barrier here: the next window's first LDS writes are its head ballots, last read before the barrier above; its
        //  atomics and probes come after its own first barrier, which every wave reaches after the stores and clears above)
    }
    if (tid == 0) p.span_rows[b] = rows_so_far;
}

// exclusive scan of the chunks' row counts (G <= 4096) -> row bases + the total; one workgroup of THREADS threads
template <int THREADS>
__device__ __forceinline__ void span_scan_body(const u32* __restrict__ span_rows, u32 G, u64* __restrict__ span_base, u64* __restrict__ total_out) {
    constexpr u32 PER = 4096 / THREADS;
    __shared__ u32 s_w[THREADS / WAVE];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    u32 v[PER]; u32 sum = 0;
#pragma unroll
    for (u32 k = 0; k < PER; ++k) { const u32 i = threadIdx.x * PER + k; v[k] = i < G ? span_rows[i] : 0u; sum += v[k]; }
    const u32 inc = wave_incl_scan32(sum, lane);
    if (lane == WAVE - 1) s_w[w] = inc;
    __syncthreads();
    u64 off = 0;
    for (int i = 0; i < w; ++i) off += s_w[i];
    u64 e = off + inc - sum;
#pragma unroll
    for (u32 k = 0; k < PER; ++k) { const u32 i = threadIdx.x * PER + k; if (i < G) span_base[i] = e; e += v[k]; }
    if (threadIdx.x == THREADS - 1) { span_base[G] = e; if (total_out) *total_out = e; }
}

// distinct non-NULL keys of one hash partition of a group longer than a window (work items left by reduce_hashed_kernel).
// The LAST workgroup of the grid does something else: the scan of the chunks' row counts (what span_scan_kernel does for the
// other reduce kernel — one launch less behind the reduce) and the reset of the item counter the NEXT reduce will use
// (items[0 / 1] by turns: this launch reads n_items[parity], which the reduce before it counted up).
__global__ __launch_bounds__(512) void giant_groups_kernel(const u64* __restrict__ keys, const u64* __restrict__ list, u32* __restrict__ n_items, u32 parity,
                                                           KeyLayout L, u32* __restrict__ count, u64* __restrict__ err,
                                                           const u32* __restrict__ span_rows, u32 G, u64* __restrict__ span_base, u64* __restrict__ total_out) {
    if (blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 0) n_items[parity ^ 1u] = 0;
        span_scan_body<512>(span_rows, G, span_base, total_out);
        return;
    }
    __shared__ u64 s_set[GIANT_TAB];
    __shared__ u32 s_cnt;
    // more items than the list holds: a group that straddled the end of the list left its slots unwritten, and the caller
    // sorts fully and reduces again anyway (ERR_RUN_TOO_LONG is up) — walk nothing
    const u32 have = n_items[parity];
    const u32 items = have <= GIANT_LIST_CAP ? have : 0u;
    const u32 nn_shift = L.umi_bits + L.len_bits;
    for (u32 g = blockIdx.x; g < items; g += gridDim.x - 1) {
        const u64* it = list + (u64)g * GIANT_ITEM_WORDS;
        const u64 start = it[0], len = it[1], row = it[2];
        const u32 part = (u32)it[3], parts = (u32)(it[3] >> 32);
        for (u32 i = threadIdx.x; i < GIANT_TAB; i += 512) s_set[i] = 0;
        if (threadIdx.x == 0) s_cnt = 0;
        u64 k[4], kn[4];                                                   // this round's keys and the next round's, asked for a round ahead
#pragma unroll
        for (int u = 0; u < 4; ++u) { const u64 i = (u64)u * 512 + threadIdx.x; kn[u] = i < len ? keys[start + i] : 0; }
        __syncthreads();
        u32 mine = 0; bool full = false;
        for (u64 i0 = 0; i0 < len; i0 += 4 * 512) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { k[u] = kn[u]; const u64 i = i0 + 4 * 512 + (u64)u * 512 + threadIdx.x; kn[u] = i < len ? keys[start + i] : 0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!((k[u] >> nn_shift) & 1)) continue;                  // NULL umi (and the padding zeros): not counted
                const u64 hh = k[u] * 0x9E3779B97F4A7C15ull;
                if ((u32)(((hh >> 32) * parts) >> 32) != part) continue;   // another partition's key
                u32 h = (u32)(hh >> 7) & (GIANT_TAB - 1);
                for (u32 step = 0;; ++step) {
                    u64 v = s_set[h];
                    if (v == 0) {
                        if (__hip_atomic_compare_exchange_strong(&s_set[h], &v, k[u], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { ++mine; break; }
                    }
                    if (v == k[u]) break;
                    if (step >= GIANT_TAB) { full = true; break; }        // (a partition far larger than its share: cannot hold it)
                    h = (h + 1) & (GIANT_TAB - 1);
                }
            }
        }
        mine = wave_sum32(mine);
        if (lane_id() == 0 && mine) atomicAdd(&s_cnt, mine);
        if (full) atomicOr(err, ERR_RUN_TOO_LONG);
        __syncthreads();
        if (threadIdx.x == 0 && s_cnt) atomicAdd(&count[row], s_cnt);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// -u rows of keys wider than 64 bits: the pairs (group word, rest of the key) are fully sorted (two LSD sorts: by the rest,
// then by the group word); one row per distinct pair, with the number of its copies.  Three small kernels around the tile
// scan — this path is about not refusing such inputs, not about speed:
//   pair_heads_kernel   heads (a pair that differs from the one in front) per 2048-pair tile
//   pair_rows_kernel    row r of a head: its pair and its position
//   pair_copies_kernel  copies of row r = position of row r + 1 (or n) - position of row r
// ------------------------------------------------------------------------------------
constexpr int UW_THREADS = 256, UW_IPT = 8, UW_TILE = UW_THREADS * UW_IPT;
__device__ __forceinline__ bool pair_head(const u64* __restrict__ keys, const u64* __restrict__ vals, u64 i) {
    return i == 0 || keys[i] != keys[i - 1] || vals[i] != vals[i - 1];
}
__global__ __launch_bounds__(UW_THREADS) void pair_heads_kernel(const u64* __restrict__ keys, const u64* __restrict__ vals, const u64* __restrict__ n_ptr,
                                                                u32* __restrict__ tile_cnt) {
    const u64 n = *n_ptr, T = (n + UW_TILE - 1) / UW_TILE;
    for (u64 t = blockIdx.x; t < T; t += gridDim.x) {
        u32 c = 0;
#pragma unroll
        for (int j = 0; j < UW_IPT; ++j) { const u64 i = t * UW_TILE + (u64)threadIdx.x * UW_IPT + j; if (i < n && pair_head(keys, vals, i)) ++c; }
        c = wave_sum32(c);
        if (lane_id() == 0 && c) atomicAdd(&tile_cnt[t], c);
    }
}
__global__ __launch_bounds__(UW_THREADS) void pair_rows_kernel(const u64* __restrict__ keys, const u64* __restrict__ vals, const u64* __restrict__ n_ptr,
                                                               const u64* __restrict__ tile_base, u64* __restrict__ ukeys, u64* __restrict__ uvals,
                                                               u64* __restrict__ hpos) {
    __shared__ u32 s_w[UW_THREADS / WAVE];
    const u64 n = *n_ptr, T = (n + UW_TILE - 1) / UW_TILE;
    const int lane = lane_id(), w = threadIdx.x >> 6;
    for (u64 t = blockIdx.x; t < T; t += gridDim.x) {
        const u64 i0 = t * UW_TILE + (u64)threadIdx.x * UW_IPT;
        u32 flags = 0, c = 0;
#pragma unroll
        for (int j = 0; j < UW_IPT; ++j) if (i0 + j < n && pair_head(keys, vals, i0 + j)) { flags |= 1u << j; ++c; }
        const u32 inc = wave_incl_scan32(c, lane);
        if (lane == WAVE - 1) s_w[w] = inc;
        __syncthreads();
        u64 r = tile_base[t] + inc - c;
        for (int x = 0; x < w; ++x) r += s_w[x];
#pragma unroll
        for (int j = 0; j < UW_IPT; ++j) if ((flags >> j) & 1u) { ukeys[r] = keys[i0 + j]; uvals[r] = vals[i0 + j]; hpos[r] = i0 + j; ++r; }
        __syncthreads();
    }
}
__global__ __launch_bounds__(UW_THREADS) void pair_copies_kernel(const u64* __restrict__ hpos, const u64* __restrict__ nrows_ptr, const u64* __restrict__ n_ptr,
                                                                 u32* __restrict__ ncopy) {
    const u64 rows = *nrows_ptr, n = *n_ptr;
    for (u64 r = (u64)blockIdx.x * UW_THREADS + threadIdx.x; r < rows; r += (u64)gridDim.x * UW_THREADS)
        ncopy[r] = (u32)((r + 1 < rows ? hpos[r + 1] : n) - hpos[r]);
}

// row bases of the chunks (exclusive scan of span_rows, G <= 4096) and the total — behind reduce_windows_kernel; behind
// reduce_hashed_kernel the last workgroup of giant_groups_kernel does it
__global__ __launch_bounds__(1024) void span_scan_kernel(const u32* __restrict__ span_rows, u32 G, u64* __restrict__ span_base,
                                                         u64* __restrict__ total_out) {
    span_scan_body<1024>(span_rows, G, span_base, total_out);
}

// concatenation of the chunks' row regions: one workgroup per chunk.  The destinations may be device arrays or pinned
// host memory (then this kernel is the device-to-host copy of the rows).
template <bool UMI_ROWS>
__global__ __launch_bounds__(256) void rows_gather_kernel(const u32* __restrict__ rg_feature, const u32* __restrict__ rg_cell,
                                                          const u32* __restrict__ rg_count, const u64* __restrict__ rg_ukeys,
                                                          const u32* __restrict__ span_rows, const u64* __restrict__ span_base,
                                                          const u64* __restrict__ n_ptr, u32 G,
                                                          u32* __restrict__ feature, u32* __restrict__ cell, u32* __restrict__ count,
                                                          u64* __restrict__ ukeys) {
    const u64 n = *n_ptr;
    for (u32 b = blockIdx.x; b < G; b += gridDim.x) {
        const u32 rows = span_rows[b];
        const u64 src = k3_chunk_start(n, b, G), dst = span_base[b];
        for (u32 r = threadIdx.x; r < rows; r += 256) {
            count[dst + r] = rg_count[src + r];
            if (UMI_ROWS) ukeys[dst + r] = rg_ukeys[src + r];
            else { feature[dst + r] = rg_feature[src + r]; cell[dst + r] = rg_cell[src + r]; }
        }
    }
}

}  // namespace fastf
