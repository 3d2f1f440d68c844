// igemm.h -- parameters and fused epilogue shared by the GEMM-shaped kernels (nn.hip, conv.hip)
#pragma once
#include <math.h>

#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ifh {

struct IgemmParams {
    const uint16_t *x;
    int64_t x_bstride;
    int lda;
    int Cin, taps, stride, dil, pad;
    int T_in, T_out, nbatch;
    const uint16_t *w;
    int K, N;
    const float *bias;
    const uint8_t *colmask;
    float pre_slope;
    int act;
    float act_slope;
    const uint16_t *resid;
    int64_t resid_bstride;
    int resid_ld;
    float out_scale;
    int accumulate;
    void *out;
    int out_f32;
    int64_t out_bstride;
    int ldc, ostride, ooff;
    int vec_ok;
    int res_vec_ok;           // bias/colmask/resid can be read as 16/4/8-byte vectors
    int fast_epi;             // every 4-group of this launch can take the vector epilogue
    int store16;              // conv.hip: outputs are 16-byte addressable -> LDS-transposed full-row stores
    int n_split;              // columns >= n_split (if > 0) go to the second output region
    void *out2;
    int64_t out2_bstride;
    int ldc2, ooff2, dyn_ooff2_mul;
    const int32_t *dyn;       // optional device scalar (e.g. decoder position), or one value per output row (dyn_stride = 1)
    int dyn_stride;           // value for output row m = dyn[m * dyn_stride]
    int dyn_ooff_mul;         // ooff += dyn[0] * dyn_ooff_mul
    int64_t dyn_resid_mul;    // resid += dyn[0] * dyn_resid_mul (elements)
    // LayerNorm folded around a skinny GEMM (decode steps; see ifh_conv_desc):
    const void *aln_stats;    // int64 [M][2] fixed-point (2^16) (sum, sumsq) of the A rows: out = rstd*(acc - mean*aln_c1[n]) (+bias ...)
    const float *aln_c1;      // [N] row sums of the gamma-folded weights
    const void *rln_stats;    // int64 [M][2] stats of the residual rows: resid -> (r - mean)*rstd*rln_gamma[n] + rln_beta[n]
    const float *rln_gamma, *rln_beta;
    void *stats_out;          // int64 [M][2] += fixed-point (sum, sumsq) of the stored output rows (caller zeroes)
    int ln_dim;               // LayerNorm width
    float ln_eps;
    int ln_rms;               // RMSNorm instead of LayerNorm in the folded modes (mean term dropped)
    unsigned long long *amax_keys;   // k_gemm_m64: per-row arg-max keys (ifh_conv_desc.argmax_keys)
    int whole_chip;           // ifh_conv_desc.whole_chip
    int zt_cout;              // > 0: fused ConvTranspose1d(k8,s4,p2) form (3 taps, 4*zt_cout columns): columns of phases 0-1 have an
                              // all-zero tap 2, of phases 2-3 an all-zero tap 0 -- a column tile inside one half skips those k-steps
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANH = 3, ACT_LRELU = 4, ACT_SIGMOID = 5, ACT_SILU_GLU = 6 };

// GELU (exact erf form) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the stored
// result): one reciprocal, one exp2 and a 5-term polynomial instead of the library erff's branchy ~35 instructions.  The
// large-GEMM epilogue applies it to 64 outputs per thread; with erff the activation cost as much as the K = 512 loop it follows.
__device__ __forceinline__ float gelu_fast(float v)
{
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float pl = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    pl = __builtin_fmaf(t, pl, 1.421413741f);
    pl = __builtin_fmaf(t, pl, -0.284496736f);
    pl = __builtin_fmaf(t, pl, 0.254829592f);
    pl *= t;
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erfabs = __builtin_fmaf(-pl, e, 1.0f);
    return 0.5f * v * (1.0f + copysignf(erfabs, v));
}

__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_GELU: return gelu_fast(v);        // (one GELU everywhere: the decode kernels' epilogues spent 3.5 us of a 16 us fc1 launch in erff)
    case ACT_TANH: return tanhf(v);
    case ACT_LRELU: return v > 0.0f ? v : v * slope;
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}

// activation chosen at compile time (ACTC >= 0) or by the runtime switch (ACTC < 0)
template <int ACTC>
__device__ __forceinline__ float apply_act_c(float v, int act, float slope)
{
    if (ACTC == ACT_NONE) return v;
    if (ACTC == ACT_GELU) return gelu_fast(v);
    if (ACTC == ACT_RELU) return fmaxf(v, 0.0f);
    return apply_act(v, act, slope);
}

__device__ __forceinline__ uint4 lrelu8(uint4 v, float slope)
{
    uint32_t *u = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float lo = __uint_as_float(u[i] << 16), hi = __uint_as_float(u[i] & 0xffff0000u);
        lo = fmaxf(lo, lo * slope);      // 0 < slope < 1
        hi = fmaxf(hi, hi * slope);
        u[i] = f32x2_to_bf16x2(lo, hi);
    }
    return v;
}


// Output stores of the fused epilogues.  The resident decode step (step.hip) hands a phase's outputs to OTHER workgroups of the same
// launch and compiles these with IFH_EPI_SC1: stores are write-through (sc1) unless the workgroup has established that its whole
// cluster runs on one XCD, i.e. shares one L2 (g_epi_plain: plain stores that stay in that L2); the consumers read with sc1 loads
// (never served by a CU's L1) after the producers' arrival flags (MI355X_MICROARCH.md, inter-workgroup visibility).  Its LayerNorm
// row statistics are accumulated per workgroup in LDS (g_epi_stats) and handed over as per-workgroup partial sums -- 64-bit integer
// sums, so the total is the one the atomics of the launch chain produce.  Everywhere else: plain stores, atomics.
#ifdef IFH_EPI_SC1
__shared__ int g_epi_plain;
__shared__ unsigned long long g_epi_stats[32][2];      // [row of the 32-row block][sum, sum of squares], fixed point 2^16
#endif
__device__ __forceinline__ void epi_store8(void *o, uint2 v)
{
#ifdef IFH_EPI_SC1
    if (g_epi_plain)
        *reinterpret_cast<uint2 *>(o) = v;
    else
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(o), (unsigned long long)v.x | ((unsigned long long)v.y << 32),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *reinterpret_cast<uint2 *>(o) = v;
#endif
}
__device__ __forceinline__ void epi_store16f(void *o, float4 v)
{
#ifdef IFH_EPI_SC1
    if (g_epi_plain) {
        *reinterpret_cast<float4 *>(o) = v;
    } else {
        const f32x4 t = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(t) : "memory");
    }
#else
    *reinterpret_cast<float4 *>(o) = v;
#endif
}
__device__ __forceinline__ void epi_stats_add(void *stats_out, int m, unsigned long long v1, unsigned long long v2)
{
#ifdef IFH_EPI_SC1
    atomicAdd(&g_epi_stats[m & 31][0], v1);
    atomicAdd(&g_epi_stats[m & 31][1], v2);
#else
    unsigned long long *so = reinterpret_cast<unsigned long long *>(stats_out) + 2 * m;
    atomicAdd(so, v1);
    atomicAdd(so + 1, v2);
#endif
}

// ---- fused epilogue for 4 consecutive output channels n..n+3 of output row m.
// Two forms: the vector form (16-byte bias load, 8-byte residual load, 8/16-byte store) is inlined
// into the kernels' unrolled accumulator loops and must stay SMALL -- when it grew, hipcc stopped
// unrolling those loops and put the accumulator tiles in scratch memory (2x slower kernels).
// The general form (ragged N, unaligned pointers, ...) lives in separate kernel instantiations
// (template parameter FAST = false), chosen by the host when p.fast_epi == 0.
__device__ __forceinline__ int dyn_value(const IgemmParams &p, int m)
{
    return p.dyn ? p.dyn[(int64_t)m * p.dyn_stride] : 0;
}

struct EpiRow {
    int64_t obase, rbase;
    void *outp;
};

__device__ __forceinline__ EpiRow epi_row(const IgemmParams &p, int m, int n, int dynv)
{
    const int b = p.T_out == 1 ? m : m / p.T_out, t = m - b * p.T_out;      // (decode steps: one row per batch entry, no division)
    const bool second = p.n_split > 0 && n >= p.n_split;      // n_split % 16 == 0: a 4-group never straddles
    const int64_t orow = second ? ((int64_t)t * p.ostride + p.ooff2 + (int64_t)dynv * p.dyn_ooff2_mul)
                                : ((int64_t)t * p.ostride + p.ooff + (int64_t)dynv * p.dyn_ooff_mul);
    EpiRow r;
    r.obase = second ? ((int64_t)b * p.out2_bstride + orow * p.ldc2 - p.n_split) : ((int64_t)b * p.out_bstride + orow * p.ldc);
    r.outp = second ? p.out2 : p.out;
    r.rbase = (int64_t)b * p.resid_bstride + orow * p.resid_ld + (int64_t)dynv * p.dyn_resid_mul;
    return r;
}

// HAVE_R: the residual 4-vector was loaded by the caller ahead of its K loop (rpre)
// STORE = false: bf16 outputs only; the rounded 4-vector is returned instead of written (the caller
// transposes it through LDS into full-row stores).
// HAVE_B: same for the bias 4-vector (bpre; ignored when p.bias is null)
template <bool HAVE_R = false, bool STORE = true, bool HAVE_B = false, int ACTC = -1>
__device__ __forceinline__ uint2 igemm_store4_fast(const IgemmParams &p, int m, int n, f32x4 acc, int dynv,
                                                   uint2 rpre = make_uint2(0, 0),
                                                   float4 bpre = make_float4(0.f, 0.f, 0.f, 0.f))
{
    const EpiRow e = epi_row(p, m, n, dynv);
    float v0 = acc[0], v1 = acc[1], v2 = acc[2], v3 = acc[3];
    if (p.bias) {
        const float4 bv = HAVE_B ? bpre : *reinterpret_cast<const float4 *>(p.bias + n);
        v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
    }
    if (ACTC >= 0 ? ACTC != ACT_NONE : p.act != ACT_NONE) {
        v0 = apply_act_c<ACTC>(v0, p.act, p.act_slope);
        v1 = apply_act_c<ACTC>(v1, p.act, p.act_slope);
        v2 = apply_act_c<ACTC>(v2, p.act, p.act_slope);
        v3 = apply_act_c<ACTC>(v3, p.act, p.act_slope);
    }
    if (p.colmask) {
        const uint32_t mk = *reinterpret_cast<const uint32_t *>(p.colmask + n);
        v0 = (mk & 0xffu) ? v0 * 2.0f : 0.0f;
        v1 = (mk & 0xff00u) ? v1 * 2.0f : 0.0f;
        v2 = (mk & 0xff0000u) ? v2 * 2.0f : 0.0f;
        v3 = (mk & 0xff000000u) ? v3 * 2.0f : 0.0f;
    }
    if (p.resid) {
        const uint2 rv = HAVE_R ? rpre : *reinterpret_cast<const uint2 *>(p.resid + e.rbase + n);
        v0 += __uint_as_float(rv.x << 16);
        v1 += __uint_as_float(rv.x & 0xffff0000u);
        v2 += __uint_as_float(rv.y << 16);
        v3 += __uint_as_float(rv.y & 0xffff0000u);
    }
    v0 *= p.out_scale; v1 *= p.out_scale; v2 *= p.out_scale; v3 *= p.out_scale;
    if (STORE && p.out_f32) {
        float *o = reinterpret_cast<float *>(e.outp) + e.obase + n;
        if (p.accumulate) {
            const float4 prev = *reinterpret_cast<const float4 *>(o);
            v0 += prev.x; v1 += prev.y; v2 += prev.z; v3 += prev.w;
        }
        epi_store16f(o, make_float4(v0, v1, v2, v3));
        return make_uint2(0, 0);
    }
    uint16_t *o = reinterpret_cast<uint16_t *>(e.outp) + e.obase + n;
    if (p.accumulate) {
        const uint2 pv = *reinterpret_cast<const uint2 *>(o);
        v0 += __uint_as_float(pv.x << 16);
        v1 += __uint_as_float(pv.x & 0xffff0000u);
        v2 += __uint_as_float(pv.y << 16);
        v3 += __uint_as_float(pv.y & 0xffff0000u);
    }
    uint2 pk;
    pk.x = f32x2_to_bf16x2(v0, v1);
    pk.y = f32x2_to_bf16x2(v2, v3);
    if (STORE) epi_store8(o, pk);
    return pk;
}

__device__ __forceinline__ void igemm_store4_general(const IgemmParams &p, int m, int n, float a0, float a1,
                                                               float a2, float a3, int dynv)
{
    const EpiRow e = epi_row(p, m, n, dynv);
#define IFH_EPI1(V, R)                                                                                     \
    if (n + R < p.N) {                                                                                     \
        float v = V;                                                                                       \
        if (p.bias) v += p.bias[n + R];                                                                    \
        v = apply_act(v, p.act, p.act_slope);                                                              \
        if (p.colmask) v = p.colmask[n + R] ? v * 2.0f : 0.0f;                                             \
        if (p.resid) v += bf16_to_f32(p.resid[e.rbase + n + R]);                                           \
        v *= p.out_scale;                                                                                  \
        if (p.out_f32) {                                                                                   \
            float *o = reinterpret_cast<float *>(e.outp) + e.obase + n + R;                                \
            *o = v + (p.accumulate ? *o : 0.0f);                                                           \
        } else {                                                                                           \
            uint16_t *o = reinterpret_cast<uint16_t *>(e.outp) + e.obase + n + R;                          \
            *o = f32_to_bf16(v + (p.accumulate ? bf16_to_f32(*o) : 0.0f));                                 \
        }                                                                                                  \
    }
    IFH_EPI1(a0, 0) IFH_EPI1(a1, 1) IFH_EPI1(a2, 2) IFH_EPI1(a3, 3)
#undef IFH_EPI1
}

template <bool FAST>
__device__ __forceinline__ void igemm_store4(const IgemmParams &p, int m, int n, f32x4 acc, int dynv)
{
    if (FAST)
        (void)igemm_store4_fast(p, m, n, acc, dynv);
    else
        igemm_store4_general(p, m, n, acc[0], acc[1], acc[2], acc[3], dynv);
}

// ---- LayerNorm-folded epilogue of the decode-step GEMMs (ifh_conv_desc.aln_* / rln_* / stats_out), shared by the
// weight-streaming kernel (k_gemm_skinny) and the LDS-tiled one (k_gemm_dec) so that both produce the same bits from the same
// accumulated 4-vector.  `s` = D[n .. n+3][m]; operands were fetched by the caller ahead of its K loop.
struct LnRow {
    float a_mean, a_rstd, r_mean, r_rstd;
};

__device__ __forceinline__ LnRow ln_row(const IgemmParams &p, longlong2 st_a, longlong2 st_r)
{
    const float fx = (1.0f / 65536.0f) / (float)p.ln_dim;
    LnRow r;
    r.a_mean = p.ln_rms ? 0.0f : (float)st_a.x * fx;
    r.r_mean = (float)st_r.x * fx;
    r.a_rstd = rsqrtf(fmaxf((float)st_a.y * fx - r.a_mean * r.a_mean, 0.0f) + p.ln_eps);
    r.r_rstd = rsqrtf(fmaxf((float)st_r.y * fx - r.r_mean * r.r_mean, 0.0f) + p.ln_eps);
    return r;
}

// xok: row m is a real row; lanes fg = 0..3 of a 16-lane group hold columns n = nb + 4*fg of the same row (all 64 lanes of the
// wave must call this: the row statistics are reduced across them by shuffles)
__device__ __forceinline__ void ln_epi4(const IgemmParams &p, int m, int n, bool xok, f32x4 s, int dynv, const LnRow &lr,
                                        float4 c1, float4 bias, float4 gam, float4 beta, uint2 resid, int fg)
{
    float v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
    const bool ok = xok && n < p.N;
    if (ok) {
        const EpiRow er = epi_row(p, m, n, dynv);
        if (p.aln_stats) {
            const float mean = lr.a_mean, rstd = lr.a_rstd;
            v0 = rstd * (v0 - mean * c1.x);
            v1 = rstd * (v1 - mean * c1.y);
            v2 = rstd * (v2 - mean * c1.z);
            v3 = rstd * (v3 - mean * c1.w);
        }
        if (p.bias) {
            v0 += bias.x; v1 += bias.y; v2 += bias.z; v3 += bias.w;
        }
        if (p.act != ACT_NONE) {
            v0 = apply_act(v0, p.act, p.act_slope);
            v1 = apply_act(v1, p.act, p.act_slope);
            v2 = apply_act(v2, p.act, p.act_slope);
            v3 = apply_act(v3, p.act, p.act_slope);
        }
        if (p.resid) {
            float r0 = __uint_as_float(resid.x << 16), r1 = __uint_as_float(resid.x & 0xffff0000u);
            float r2 = __uint_as_float(resid.y << 16), r3 = __uint_as_float(resid.y & 0xffff0000u);
            if (p.rln_stats) {
                const float mean = lr.r_mean, rstd = lr.r_rstd;
                r0 = (r0 - mean) * rstd * gam.x + beta.x;
                r1 = (r1 - mean) * rstd * gam.y + beta.y;
                r2 = (r2 - mean) * rstd * gam.z + beta.z;
                r3 = (r3 - mean) * rstd * gam.w + beta.w;
            }
            v0 += r0; v1 += r1; v2 += r2; v3 += r3;
        }
        v0 *= p.out_scale; v1 *= p.out_scale; v2 *= p.out_scale; v3 *= p.out_scale;
        if (p.out_f32) {
            epi_store16f(reinterpret_cast<float *>(er.outp) + er.obase + n, make_float4(v0, v1, v2, v3));
        } else {
            uint2 pk;
            pk.x = f32x2_to_bf16x2(v0, v1);
            pk.y = f32x2_to_bf16x2(v2, v3);
            epi_store8(reinterpret_cast<uint16_t *>(er.outp) + er.obase + n, pk);
            v0 = __uint_as_float(pk.x << 16); v1 = __uint_as_float(pk.x & 0xffff0000u);   // what consumers will read
            v2 = __uint_as_float(pk.y << 16); v3 = __uint_as_float(pk.y & 0xffff0000u);
        }
    }
    if (p.stats_out) {
        float s1 = ok ? (v0 + v1) + (v2 + v3) : 0.0f;
        float s2 = ok ? (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3) : 0.0f;
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);      // the 4 lanes fg = 0..3 share row m
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (fg == 0 && xok)
            epi_stats_add(p.stats_out, m, (unsigned long long)__double2ll_rn((double)s1 * 65536.0),
                          (unsigned long long)__double2ll_rn((double)s2 * 65536.0));
    }
}


// conv.hip: LDS-resident-input convolution for the stride-1 residual-block shapes.
// Returns true if it took the launch.
bool try_launch_conv_direct(const IgemmParams &p, bool pre, hipStream_t st);
// gemm_big.hip: DMA-ring matrix product for thousands of rows x whole 256 x 128 tiles.  Returns true if it took the launch.
bool try_launch_gemm_big(const IgemmParams &p, bool pre, hipStream_t st, float *ws = nullptr, int64_t ws_floats = 0);
bool try_launch_gemm_m64d(const IgemmParams &p, hipStream_t st);
int try_launch_gemm_m64d_splitk(const IgemmParams &p, float *ws, int64_t ws_floats, hipStream_t st);

}  // namespace ifh
