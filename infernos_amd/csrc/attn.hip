// attn.hip -- attention for the speech models on gfx950 (head_dim 64 everywhere:
// Whisper tiny/base 384/6, 512/8; SpeechT5 768/12).
//   * k_attn_prefill: flash-style (online softmax) non-causal attention on the matrix cores,
//     64 queries per block (16 per wave), 64-key tiles staged in LDS (K row-major, V
//     transposed).  Products are issued swapped (S^T = K.Q^T, O^T = V^T.P^T) so the softmax
//     row of a query lives on one lane column and P feeds the second MFMA with no lane
//     movement.  Optional key-length mask and SpeechT5 relative-position bias
//     (modeling_speecht5.py:938-945) supplied as a precomputed table R[q][rel].
//   * k_attn_decode: one query token against a KV cache; one wave per (batch, head).
// q is expected pre-scaled by head_dim^-0.5 (folded into the q projection weights).
#include <math.h>

#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ifh {

constexpr int HD = 64;
constexpr int KT = 64;       // keys per tile
constexpr int KLD = 72;      // padded LDS row (elements)

struct AttnParams {
    const uint16_t *q, *k, *v;
    uint16_t *out;
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;  // element strides (batch, token); head h at +h*64
    int Tq, Tk, H;
    const int32_t *key_len;  // [B] or null
    const float *relbias;    // [B][Tq][H][nrel] or null
    int nrel;
};

__device__ __forceinline__ uint32_t pack2(float a, float b)
{
    return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}

__global__ __launch_bounds__(256) void k_attn_prefill(const AttnParams p)
{
    __shared__ __attribute__((aligned(16))) uint16_t Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) uint16_t Vt[HD * KLD];
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int qi = blockIdx.x * 64 + wid * 16 + fr;  // this lane's query row
    const int klen = p.key_len ? p.key_len[b] : p.Tk;

    // Q fragments (B operand of S^T = K.Q^T): Q[qi][d = 32*s + 8*fg .. +8]
    bf16x8_t qf[2];
    {
        const uint16_t *qp = p.q + (int64_t)b * p.q_bs + (int64_t)(qi < p.Tq ? qi : 0) * p.q_ts + h * HD;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (qi < p.Tq) t = *reinterpret_cast<const uint4 *>(qp + 32 * s + 8 * fg);
            qf[s] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -1e30f, lrun = 0.0f;
    const float *rb = p.relbias ? p.relbias + (((int64_t)b * p.Tq + (qi < p.Tq ? qi : 0)) * p.H + h) * p.nrel : nullptr;
    const int half = p.nrel >> 1;

    const int ntile = (klen + KT - 1) / KT;
    for (int kt = 0; kt < ntile; kt++) {
        const int kbase = kt * KT;
        __syncthreads();
        // stage K [64 keys][64 d] and V^T [64 d][64 keys]; 2 x 16-byte vectors each per thread
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int vv = tid + 256 * i;
            const int key = vv >> 3, dv = (vv & 7) * 8;
            const int kg = kbase + key;
            uint4 kk = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
            if (kg < klen) {
                kk = *reinterpret_cast<const uint4 *>(p.k + (int64_t)b * p.k_bs + (int64_t)kg * p.k_ts + h * HD + dv);
                vx = *reinterpret_cast<const uint4 *>(p.v + (int64_t)b * p.v_bs + (int64_t)kg * p.v_ts + h * HD + dv);
            }
            *reinterpret_cast<uint4 *>(&Ks[key * KLD + dv]) = kk;
            const uint16_t *ve = reinterpret_cast<const uint16_t *>(&vx);
#pragma unroll
            for (int e = 0; e < 8; e++) Vt[(dv + e) * KLD + key] = ve[e];
        }
        __syncthreads();
        // S^T tiles: 4 x (16 keys x 16 queries)
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 2; ds++) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t *>(&Ks[(c * 16 + fr) * KLD + ds * 32 + fg * 8]);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ds], s[c], 0, 0, 0);
            }
        }
        // bias, mask, running max
        float mloc = -1e30f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int kidx = kbase + c * 16 + 4 * fg + r;
                float val = s[c][r];
                if (rb) {
                    int rel = qi - kidx;
                    rel = rel < -half ? -half : (rel > half - 1 ? half - 1 : rel);
                    val += rb[rel + half];
                }
                val = (kidx < klen) ? val : -1e30f;
                s[c][r] = val;
                mloc = fmaxf(mloc, val);
            }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        const float alpha = __expf(mrun - mnew);
        float lsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = __expf(s[c][r] - mnew);
                s[c][r] = pv;
                lsum += pv;
            }
        }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        lrun = lrun * alpha + lsum;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o[i][0] *= alpha;
            o[i][1] *= alpha;
            o[i][2] *= alpha;
            o[i][3] *= alpha;
        }
        // O^T += V^T . P^T ; k-step ks covers keys 32*ks..32*ks+31, element j <-> key 16*(j>>2) + 4*fg + (j&3)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint4 pb;
            pb.x = pack2(s[2 * ks][0], s[2 * ks][1]);
            pb.y = pack2(s[2 * ks][2], s[2 * ks][3]);
            pb.z = pack2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
            pb.w = pack2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pb);
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                const uint16_t *vr = &Vt[(dt * 16 + fr) * KLD + ks * 32 + 4 * fg];
                uint4 va;
                const uint2 lo = *reinterpret_cast<const uint2 *>(vr);
                const uint2 hi = *reinterpret_cast<const uint2 *>(vr + 16);
                va.x = lo.x;
                va.y = lo.y;
                va.z = hi.x;
                va.w = hi.y;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), pf, o[dt], 0, 0, 0);
            }
        }
    }
    if (qi < p.Tq) {
        const float inv = lrun > 0.0f ? 1.0f / lrun : 0.0f;
        uint16_t *op = p.out + (int64_t)b * p.o_bs + (int64_t)qi * p.o_ts + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
            uint2 pk;
            pk.x = pack2(o[dt][0] * inv, o[dt][1] * inv);
            pk.y = pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2 *>(op + dt * 16 + 4 * fg) = pk;
        }
    }
}

// One query token per (batch, head); scores in LDS.  S <= smax (dynamic LDS floats).
__global__ __launch_bounds__(64) void k_attn_decode(const uint16_t *__restrict__ q, int64_t q_bs,
                                                    const uint16_t *__restrict__ k, const uint16_t *__restrict__ v,
                                                    int64_t kv_bs, int64_t kv_ts, uint16_t *__restrict__ out,
                                                    int64_t o_bs, const int32_t *__restrict__ key_len, int S)
{
    extern __shared__ __attribute__((aligned(16))) float sc[];
    const int b = blockIdx.y, h = blockIdx.x, lane = threadIdx.x;
    const int klen = key_len ? key_len[b] : S;
    float qv[HD];
    {
        const uint16_t *qp = q + (int64_t)b * q_bs + h * HD;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 t = *reinterpret_cast<const uint4 *>(qp + 8 * i);
            const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qv[8 * i + 2 * e] = __uint_as_float(u[e] << 16);
                qv[8 * i + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
            }
        }
    }
    const uint16_t *kb = k + (int64_t)b * kv_bs + h * HD;
    const uint16_t *vb = v + (int64_t)b * kv_bs + h * HD;
    float mloc = -1e30f;
    for (int key = lane; key < klen; key += 64) {
        const uint16_t *kr = kb + (int64_t)key * kv_ts;
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 t = *reinterpret_cast<const uint4 *>(kr + 8 * i);
            const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                acc = __fmaf_rn(qv[8 * i + 2 * e], __uint_as_float(u[e] << 16), acc);
                acc = __fmaf_rn(qv[8 * i + 2 * e + 1], __uint_as_float(u[e] & 0xffff0000u), acc);
            }
        }
        sc[key] = acc;
        mloc = fmaxf(mloc, acc);
    }
    const float m = wave_max(mloc);
    float lsum = 0.0f;
    for (int key = lane; key < klen; key += 64) {
        const float pv = __expf(sc[key] - m);
        sc[key] = pv;
        lsum += pv;
    }
    const float l = wave_sum(lsum);
    __syncthreads();
    float acc = 0.0f;
    for (int key = 0; key < klen; key++) acc = __fmaf_rn(sc[key], bf16_to_f32(vb[(int64_t)key * kv_ts + lane]), acc);
    out[(int64_t)b * o_bs + h * HD + lane] = f32_to_bf16(l > 0.0f ? acc / l : 0.0f);
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_attn_prefill_bf16(const ifh_attn_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d && d->q && d->k && d->v && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->nheads > 0 && d->tq >= 0 && d->tk >= 1);
    if (d->nbatch == 0 || d->tq == 0) return IFH_OK;
    IFH_CHECK_ARG(d->head_dim == HD);
    IFH_CHECK_ARG(d->q_ts % 8 == 0 && d->k_ts % 8 == 0 && d->v_ts % 8 == 0 && d->o_ts % 4 == 0);
    IFH_CHECK_ARG(d->q_bs % 8 == 0 && d->k_bs % 8 == 0 && d->v_bs % 8 == 0 && d->o_bs % 4 == 0);
    IFH_CHECK_ARG(d->relbias == nullptr || (d->nrel > 0 && d->nrel % 2 == 0));
    IFH_CHECK_ARG(d->nbatch < 65536 && d->nheads < 65536);
    AttnParams p;
    p.q = (const uint16_t *)d->q;
    p.k = (const uint16_t *)d->k;
    p.v = (const uint16_t *)d->v;
    p.out = (uint16_t *)d->out;
    p.q_bs = d->q_bs;
    p.q_ts = d->q_ts;
    p.k_bs = d->k_bs;
    p.k_ts = d->k_ts;
    p.v_bs = d->v_bs;
    p.v_ts = d->v_ts;
    p.o_bs = d->o_bs;
    p.o_ts = d->o_ts;
    p.Tq = d->tq;
    p.Tk = d->tk;
    p.H = d->nheads;
    p.key_len = d->key_len;
    p.relbias = d->relbias;
    p.nrel = d->nrel;
    dim3 grid((d->tq + 63) / 64, d->nheads, d->nbatch);
    hipLaunchKernelGGL(k_attn_prefill, grid, dim3(256), 0, as_stream(stream), p);
    IFH_LAUNCH_CHECK("attn_prefill");
    return IFH_OK;
}

extern "C" int ifh_attn_decode_bf16(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs,
                                    int64_t kv_ts, void *out, int64_t o_bs, const int32_t *key_len, int max_keys,
                                    int nbatch, int nheads, int head_dim, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(q && k && v && out && nheads > 0 && head_dim == HD && max_keys >= 1 && max_keys <= 8192);
    IFH_CHECK_ARG(q_bs % 8 == 0 && kv_bs % 8 == 0 && kv_ts % 8 == 0 && nbatch < 65536);
    dim3 grid(nheads, nbatch);
    hipLaunchKernelGGL(k_attn_decode, grid, dim3(64), (size_t)max_keys * sizeof(float), as_stream(stream),
                       (const uint16_t *)q, q_bs, (const uint16_t *)k, (const uint16_t *)v, kv_bs, kv_ts, (uint16_t *)out,
                       o_bs, key_len, max_keys);
    IFH_LAUNCH_CHECK("attn_decode");
    return IFH_OK;
}
