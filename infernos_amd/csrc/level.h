// level.h -- parameters and helpers of the C = 32 level kernel (level.hip), shared with the experiment kept under
// tools/exp/level_pipe.hip (the convolutions software-pipelined over a completion counter in LDS instead of one barrier each:
// measured equal, round 6, profiles/NOTES.md)
#pragma once
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "igemm.h"

namespace ifh {

struct LevelParams {
    const uint16_t *x;
    int64_t x_bstride;
    uint16_t *out;
    int64_t out_bstride;
    const uint16_t *w[3];        // per block: packed fragments of its six convolutions (ops.w_chain_pack layout)
    const float *bias[3];        // per block: [6][C]
    int T, nbatch, tiles_per_seq, ntiles;
    float slope, out_scale;
    int accumulate;              // the first block adds to `out` as well
    unsigned long long *prof;    // ABL & 16 builds: shader-clock sums of wave 0 of every workgroup, per phase of a convolution
    // POST instantiations (ifh_level_desc.post_w): the folded conv_post
    const float *post_w;         // [7][32]
    float post_bias, post_slope;
    uint16_t *audio;             // bf16 [nbatch][T]
    uint16_t *mean_ws;           // per workgroup: the running mean of the tile being worked on, [grid][RT][C] bf16
};

template <int... I, class F>
__device__ __forceinline__ void lv_static_for_impl(std::integer_sequence<int, I...>, F &&f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void lv_static_for(F &&f)
{
    lv_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// LeakyReLU of four packed bf16 rounded back to bf16 -- the values of chain_lrelu4 / lrelu8 (max(p, p * slope), 0 < slope <= 1),
// bit for bit: a non-negative p is itself, a negative one is bf16(p * slope); the choice is made on the packed halves (arithmetic
// shift of the signs + bit select), so no maximum, no canonicalisation and no inline asm with its padding: 6 VALU per pair.
__device__ __forceinline__ uint32_t lv_lrelu2(uint32_t x, float slope)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 p = {__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u)};
    const f32x2 q = p * slope;
    const uint32_t qk = f32x2_to_bf16x2(q.x, q.y);
    // 0xffff per negative half, then a bit select -- as inline asm: from the C form hipcc rebuilt a compare + select + byte permute per
    // half (15 VALU per pair)
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, %1, %3\n\tv_bfi_b32 %0, %0, %2, %3" : "=&v"(r) : "s"(0x000f000fu), "v"(qk), "v"(x));
    return r;
}
__device__ __forceinline__ uint2 lv_lrelu4(uint2 v, float slope) { return make_uint2(lv_lrelu2(v.x, slope), lv_lrelu2(v.y, slope)); }

constexpr int lv_log2(int v) { return v <= 1 ? 0 : 1 + lv_log2(v / 2); }
constexpr int lv_min(int a, int b) { return a < b ? a : b; }


// tools/exp/level_pipe.hip (linked by tools/r06_level_variants.sh only)
int launch_level_pipe(LevelParams &p, int t0, int t1, int t2, hipStream_t st);

}  // namespace ifh
