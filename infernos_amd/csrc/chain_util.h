// chain_util.h -- helpers shared by the fused HiFi-GAN residual-block kernels (chain.hip, seq.hip): the LeakyReLU of packed bf16,
// hand-counted LDS waits and the spreading of a k-step's fragment reads over its MFMAs.
#pragma once
#include "igemm.h"

namespace ifh {

// LeakyReLU of four packed bf16, rounded back to bf16: max(a, a * slope) for 0 < slope <= 1 (the values of lrelu8 /
// fmaxf(a, a * slope), bit for bit).  fmaxf() costs a canonicalising v_max x,x per operand that comes out of bit
// operations; the epilogues are VALU-issue-bound and run in no MFMA's shadow, so the maximum is one v_max_f32 by hand
// and the products are packed (v_pk_mul_f32).
__device__ __forceinline__ float chain_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint2 chain_lrelu4(uint2 v, float slope)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 lo = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u)};
    const f32x2 hi = {__uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
    const f32x2 ls = lo * slope, hs = hi * slope;
    return make_uint2(f32x2_to_bf16x2(chain_max(lo.x, ls.x), chain_max(lo.y, ls.y)),
                      f32x2_to_bf16x2(chain_max(hi.x, hs.x), chain_max(hi.y, hs.y)));
}

// s_waitcnt lgkmcnt(n) with n known only after loop unrolling (the immediate must be a literal)
__device__ __forceinline__ void wait_lgkm(int n)
{
    switch (n) {
    case 0: asm volatile("s_waitcnt lgkmcnt(0)"); break;
    case 1: asm volatile("s_waitcnt lgkmcnt(1)"); break;
    case 2: asm volatile("s_waitcnt lgkmcnt(2)"); break;
    case 3: asm volatile("s_waitcnt lgkmcnt(3)"); break;
    case 4: asm volatile("s_waitcnt lgkmcnt(4)"); break;
    case 5: asm volatile("s_waitcnt lgkmcnt(5)"); break;
    case 6: asm volatile("s_waitcnt lgkmcnt(6)"); break;
    case 7: asm volatile("s_waitcnt lgkmcnt(7)"); break;
    case 8: asm volatile("s_waitcnt lgkmcnt(8)"); break;
    case 9: asm volatile("s_waitcnt lgkmcnt(9)"); break;
    case 10: asm volatile("s_waitcnt lgkmcnt(10)"); break;
    case 11: asm volatile("s_waitcnt lgkmcnt(11)"); break;
    case 12: asm volatile("s_waitcnt lgkmcnt(12)"); break;
    default: asm volatile("s_waitcnt lgkmcnt(13)"); break;
    }
}
// The NR fragment reads of the next k-step are spread evenly over the NM MFMAs of this one: read i sits in front of
// MFMA i*NM/NR.  (In front of the first NR MFMAs they kept the LDS 100 % busy for two thirds of the k-step -- every
// wave issues at the same time -- and idle for the rest; an LDS instruction that finds the queue full stalls its wave's
// MFMA issue.)  rd_at(k): the read in front of MFMA k or -1; rd_before(k): reads issued in front of MFMAs 0..k-1.
constexpr int rd_at(int k, int nr, int nm)
{
    for (int i = 0; i < nr; i++)
        if (i * nm / nr == k) return i;
    return -1;
}
constexpr int rd_before(int k, int nr, int nm)
{
    int n = 0;
    for (int i = 0; i < nr; i++)
        if (i * nm / nr < k) n++;
    return n;
}

}  // namespace ifh
