// attn_core.h -- the arithmetic of one decode-attention query row (online softmax over KU keys per iteration), shared by the
// decode kernels of attn.hip and the resident decode step of step.hip so that both produce the same bits.
#pragma once
#include "common.h"

namespace ifh {

// K / V cache rows of a decode step are read once per step and are far larger than the caches (2.4 GB per 5-beam Whisper-base
// step at 128 utterances, 0.1 GB per SpeechT5 layer at 300 rows): non-temporal loads, so that they do not evict the decoders'
// weights from L2 / Infinity Cache under the latency-bound step GEMMs that run beside them (MI355X_MICROARCH.md, nt on once-read
// streams).
#ifndef IFH_ATTN_NT
#define IFH_ATTN_NT 1
#endif
// Sum over the 8 lanes that share (lane >> 3), every lane gets it: s + s[lane^1], + [lane^2], + [lane^4], in that order -- what three
// __shfl_xor steps compute, bit for bit (after the second step the four lanes of a quad hold one value, so the mirrored lane 7 - i
// of the third step holds what lane i ^ 4 does) -- but as DPP operands of the adds instead of ds_bpermute round trips through the
// LDS crossbar (60 of them, each behind an s_waitcnt, per 8 KB of K/V in the 5-beam kernel, whose loop is VALU-bound).
__device__ __forceinline__ float sum8_dpp(float s)
{
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return s;
}

// Packed f32 FMAs of the decode kernels' inner loops, c = fma(broadcast(a.x or a.y), b, c) on both halves of a register pair.  Written as
// scalar or vector C the compiler packs only a part of them (the loops are VALU-bound: 5 query rows x 4 keys x 16 FMAs per 128 bytes of
// K/V per lane); every FMA is the same fused multiply-add as before, so the bits do not change.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 a, f32x2 b, f32x2 c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 a, f32x2 b, f32x2 c)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(b));
    return c;
}
// One online-softmax update of a query row over the KU keys of an iteration: q2 = the row's 8 dims as 4 pairs; klo / khi[p][e] = dims
// 2e / 2e+1 of keys 2p and 2p+1; vp[u][e] = dims (2e, 2e+1) of key u's value; valid[u] = key u exists.  Per key the score is the chain
// fma(q[0], k[0], 0), fma(q[1], k[1], .), ... summed over the 8 dim-lanes, as in rounds 1-2.
template <int KU>
__device__ __forceinline__ void attn_row_update(const f32x2 (&q2)[4], const f32x2 (&klo)[KU / 2][4], const f32x2 (&khi)[KU / 2][4],
                                                const f32x2 (&vp)[KU][4], const bool (&valid)[KU], float &m, float &l, f32x2 (&o2)[4])
{
    float sc[KU];
    float mn = m;
#pragma unroll
    for (int p = 0; p < KU / 2; p++) {
        f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            s2 = pk_fma_lo(q2[e], klo[p][e], s2);
            s2 = pk_fma_hi(q2[e], khi[p][e], s2);
        }
        const float t0 = sum8_dpp(s2.x), t1 = sum8_dpp(s2.y);       // (unconditionally: DPP under a branch costs an exec round trip)
        sc[2 * p] = valid[2 * p] ? t0 : -1e30f;
        sc[2 * p + 1] = valid[2 * p + 1] ? t1 : -1e30f;
        mn = fmaxf(mn, fmaxf(sc[2 * p], sc[2 * p + 1]));
    }
    const float a = __expf(m - mn);
    l *= a;
#pragma unroll
    for (int e = 0; e < 4; e++) o2[e] *= a;
#pragma unroll
    for (int p = 0; p < KU / 2; p++) {
        f32x2 pr2;
        pr2.x = valid[2 * p] ? __expf(sc[2 * p] - mn) : 0.0f;
        pr2.y = valid[2 * p + 1] ? __expf(sc[2 * p + 1] - mn) : 0.0f;
        l += pr2.x;
        l += pr2.y;
#pragma unroll
        for (int e = 0; e < 4; e++) o2[e] = pk_fma_lo(pr2, vp[2 * p][e], o2[e]);
#pragma unroll
        for (int e = 0; e < 4; e++) o2[e] = pk_fma_hi(pr2, vp[2 * p + 1][e], o2[e]);
    }
    m = mn;
}
// the iteration's K / V rows unpacked into the operand pairs of attn_row_update
template <int KU>
__device__ __forceinline__ void attn_unpack(const uint4 (&kk)[KU], const uint4 (&vv)[KU], f32x2 (&klo)[KU / 2][4], f32x2 (&khi)[KU / 2][4],
                                            f32x2 (&vp)[KU][4])
{
#pragma unroll
    for (int p = 0; p < KU / 2; p++) {
        const uint32_t *k0 = reinterpret_cast<const uint32_t *>(&kk[2 * p]), *k1 = reinterpret_cast<const uint32_t *>(&kk[2 * p + 1]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            klo[p][e] = (f32x2){__uint_as_float(k0[e] << 16), __uint_as_float(k1[e] << 16)};
            khi[p][e] = (f32x2){__uint_as_float(k0[e] & 0xffff0000u), __uint_as_float(k1[e] & 0xffff0000u)};
        }
    }
#pragma unroll
    for (int u = 0; u < KU; u++) {
        const uint32_t *vu = reinterpret_cast<const uint32_t *>(&vv[u]);
#pragma unroll
        for (int e = 0; e < 4; e++) vp[u][e] = (f32x2){__uint_as_float(vu[e] << 16), __uint_as_float(vu[e] & 0xffff0000u)};
    }
}

__device__ __forceinline__ uint4 ld_stream16(const uint16_t *ptr)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#if IFH_ATTN_NT
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ptr));
#else
    const u32x4 v = *reinterpret_cast<const u32x4 *>(ptr);
#endif
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return f32x2_to_bf16x2(a, b);      // one v_cvt_pk_bf16_f32 (two conversions + shift + or before: same rounding)
}

}  // namespace ifh
