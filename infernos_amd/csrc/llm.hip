// Decoder-only LLM step kernels for InfernLLMWorker (Cluster/InfernLLMWorker.py:60-119: Qwen2.5 through
// transformers' generate): the pieces around the GEMMs (nn.hip) of a Qwen2 layer
// (transformers/models/qwen2/modeling_qwen2.py: Qwen2RMSNorm, apply_rotary_pos_emb, Qwen2Attention with grouped-query
// heads, Qwen2MLP).
//   k_rmsnorm        one wave per token row: x * rsqrt(mean(x^2) + eps) * gamma, fp32 statistics.
//   k_rope_append    rotary embedding of the fused q|k|v projection: q rotated in place, k rotated and v copied into the
//                    KV cache at the token's own position (ragged prompts: every row has its own length).
//   k_attn_gqa       one query token per (token, kv head): the G query heads of a kv head share every K/V load; keys
//                    0 .. key_len[token]-1 of the token's cache row.  The same kernel serves prefill (one launch over all
//                    prompt tokens, key_len = position + 1: causal) and decode.
//   k_attn_gqa_prefill  the same attention for the prompt (tokens_per_row >= 16) on the matrix cores: a workgroup = 16 consecutive query
//                    tokens of one row x one kv head, one wave per query head of the group; 64-key K/V tiles in LDS shared by the
//                    waves; flash-style online softmax in the layout of attn.hip's k_attn_prefill (S^T = K.Q^T, O^T = V^T.P^T).
//   k_silu_mul       silu(gate) * up on the fused gate|up projection.
#include <stdlib.h>

#include "common.h"
#include "attn_core.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ifh {

__global__ __launch_bounds__(256) void k_rmsnorm(const uint16_t *__restrict__ x, const float *__restrict__ gamma,
                                                 uint16_t *__restrict__ out, int rows, int dim, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const uint4 *xr = reinterpret_cast<const uint4 *>(x + (int64_t)row * dim);
    uint4 *orow = reinterpret_cast<uint4 *>(out + (int64_t)row * dim);
    const int nv = dim >> 3;
    float ss = 0.0f;
    for (int j = lane; j < nv; j += 64) {
        const uint4 t = xr[j];
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float a = __uint_as_float(u[e] << 16), b = __uint_as_float(u[e] & 0xffff0000u);
            ss = __fmaf_rn(a, a, ss);
            ss = __fmaf_rn(b, b, ss);
        }
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)dim + eps);
    for (int j = lane; j < nv; j += 64) {
        const uint4 t = xr[j];
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
        const float4 g0 = *reinterpret_cast<const float4 *>(gamma + 8 * j);
        const float4 g1 = *reinterpret_cast<const float4 *>(gamma + 8 * j + 4);
        uint4 o;
        o.x = f32x2_to_bf16x2(__uint_as_float(u[0] << 16) * r * g0.x, __uint_as_float(u[0] & 0xffff0000u) * r * g0.y);
        o.y = f32x2_to_bf16x2(__uint_as_float(u[1] << 16) * r * g0.z, __uint_as_float(u[1] & 0xffff0000u) * r * g0.w);
        o.z = f32x2_to_bf16x2(__uint_as_float(u[2] << 16) * r * g1.x, __uint_as_float(u[2] & 0xffff0000u) * r * g1.y);
        o.w = f32x2_to_bf16x2(__uint_as_float(u[3] << 16) * r * g1.z, __uint_as_float(u[3] & 0xffff0000u) * r * g1.w);
        orow[j] = o;
    }
}

// token i = b * T + t sits at position pos0[b] + t of cache row b; tokens with t >= nvalid[b] are padding (skipped).
// One thread per (token, head slot, 8 consecutive pairs j .. j + 7 < hd/2): 16-byte loads of the two halves of the pairs (round 5: one
// pair per thread with 2-byte accesses took 52 us per layer on a 12 288-token prompt); head slots: nh query heads, nkv key heads, nkv
// value heads.  Per element the arithmetic and rounding are unchanged.
__global__ __launch_bounds__(256) void k_rope_append(uint16_t *__restrict__ qkv, int64_t qkv_ld,
                                                     const float *__restrict__ cs /* [max_pos][hd/2][2] cos, sin */,
                                                     uint16_t *__restrict__ cache, int64_t cache_bs, int64_t cache_ts,
                                                     const int32_t *__restrict__ pos0, const int32_t *__restrict__ nvalid,
                                                     int T, int nh, int nkv, int hd, int max_pos, int64_t total)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int half = hd >> 1, groups = half >> 3;
    const int j = (int)(idx % groups) * 8;
    const int slot = (int)((idx / groups) % (nh + 2 * nkv));
    const int64_t i = idx / ((int64_t)groups * (nh + 2 * nkv));
    const int b = (int)(i / T), t = (int)(i % T);
    if (t >= nvalid[b]) return;
    const int pos = pos0[b] + t;
    if (pos >= max_pos) return;
    uint16_t *src = qkv + i * qkv_ld + (int64_t)slot * hd;
    const uint4 lo = *reinterpret_cast<const uint4 *>(src + j), hi = *reinterpret_cast<const uint4 *>(src + j + half);
    if (slot >= nh + nkv) {                 // value head: plain copy
        uint16_t *dst = cache + (int64_t)b * cache_bs + (int64_t)pos * cache_ts + (int64_t)(slot - nh) * hd;
        *reinterpret_cast<uint4 *>(dst + j) = lo;
        *reinterpret_cast<uint4 *>(dst + j + half) = hi;
        return;
    }
    const float *cp = cs + ((int64_t)pos * half + j) * 2;
    uint4 olo, ohi;
    const uint32_t *ul = reinterpret_cast<const uint32_t *>(&lo), *uh = reinterpret_cast<const uint32_t *>(&hi);
    uint32_t *pl = reinterpret_cast<uint32_t *>(&olo), *ph = reinterpret_cast<uint32_t *>(&ohi);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float4 c4 = *reinterpret_cast<const float4 *>(cp + 4 * e);       // cos, sin of pairs 2 e, 2 e + 1
        const float a0 = __uint_as_float(ul[e] << 16), a1 = __uint_as_float(ul[e] & 0xffff0000u);
        const float b0 = __uint_as_float(uh[e] << 16), b1 = __uint_as_float(uh[e] & 0xffff0000u);
        const uint32_t r00 = f32_to_bf16(a0 * c4.x - b0 * c4.y), r01 = f32_to_bf16(a1 * c4.z - b1 * c4.w);
        const uint32_t r10 = f32_to_bf16(b0 * c4.x + a0 * c4.y), r11 = f32_to_bf16(b1 * c4.z + a1 * c4.w);
        pl[e] = r00 | (r01 << 16);
        ph[e] = r10 | (r11 << 16);
    }
    uint16_t *dst = slot < nh ? src : cache + (int64_t)b * cache_bs + (int64_t)pos * cache_ts + (int64_t)(slot - nh) * hd;
    *reinterpret_cast<uint4 *>(dst + j) = olo;
    *reinterpret_cast<uint4 *>(dst + j + half) = ohi;
}

// Grouped-query decode attention.  A wave is 8 key-groups x 8 lanes; a lane owns 8*HDV of the 64*HDV head dims (HDV
// 16-byte loads per key for K and for V).  Structure of k_attn_decode (attn.hip): per key-group online softmax over its
// keys, merged over the 8 groups and the NW waves at the end.
// ROPE (head_dim 128, one token per row: a decode step): the launch also does what k_rope_append does for the step's token -- the
// rotary embedding of q (in registers; a lane's 8 + 8 dims are exactly the pairs (j, j + 64)) and of the token's k, rounded to bf16 at
// the same points, the rotated k and the v appended to the cache at position key_len - 1 (by the first workgroup of the kv head) -- and
// every workgroup uses the fresh k / v for that key instead of reading the cache: the same bits as the two launches, one launch less
// per layer.
template <int NW, int G, int HDV, bool ROPE = false>
__global__ __launch_bounds__(NW * 64) void k_attn_gqa(const uint16_t *__restrict__ q, int64_t q_ts,
                                                      uint16_t *__restrict__ cache, int64_t cache_bs,
                                                      int64_t cache_ts, int v_off, uint16_t *__restrict__ out,
                                                      int64_t o_ts, const int32_t *__restrict__ key_len, int T,
                                                      float scale, int gtot, int nsplit, const float *__restrict__ rope_cs = nullptr,
                                                      int k_col = 0, int v_col = 0)
{
    static_assert(!ROPE || HDV == 2, "the fused rotary embedding pairs dims (j, j + 64) inside a lane: head_dim 128");
    // G = the query heads THIS workgroup serves: a kv head's gtot query heads are cut into nsplit workgroups of G (each re-reads the
    // K/V rows; a decode step of 64 tokens x 2 kv heads is otherwise 128 workgroups on 256 CUs, and VALU-bound at G = 6).  The
    // arithmetic of a query head does not depend on the cut.
    constexpr int HD = 64 * HDV, DV = 8 * HDV;
    __shared__ float comb[NW][G][8][2 + DV];
    const int i = blockIdx.y, kvh = blockIdx.x / nsplit;
    const int h0 = kvh * gtot + (blockIdx.x - kvh * nsplit) * G;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 7, g = lane >> 3;
    const int klen = key_len[i];
    float qv[G][DV];
    float rc[8], rs[8];                                 // ROPE: cos / sin of this lane's eight pairs at the token's position
    uint4 kfresh[HDV], vfresh[HDV];                     // ROPE: the step's own key (rotated) and value rows, this lane's dims
    // a pair (a = x[j], b = x[j + 64]) of packed bf16 rows lo / hi -> rotated, rounded as k_rope_append rounds
    auto rot8 = [&](const uint4 lo, const uint4 hi, uint4 &olo, uint4 &ohi) {
        const uint32_t *ul = reinterpret_cast<const uint32_t *>(&lo), *uh = reinterpret_cast<const uint32_t *>(&hi);
        uint32_t *pl = reinterpret_cast<uint32_t *>(&olo), *ph = reinterpret_cast<uint32_t *>(&ohi);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float a0 = __uint_as_float(ul[e] << 16), a1 = __uint_as_float(ul[e] & 0xffff0000u);
            const float b0 = __uint_as_float(uh[e] << 16), b1 = __uint_as_float(uh[e] & 0xffff0000u);
            const uint32_t r00 = f32_to_bf16(a0 * rc[2 * e] - b0 * rs[2 * e]), r01 = f32_to_bf16(a1 * rc[2 * e + 1] - b1 * rs[2 * e + 1]);
            const uint32_t r10 = f32_to_bf16(b0 * rc[2 * e] + a0 * rs[2 * e]), r11 = f32_to_bf16(b1 * rc[2 * e + 1] + a1 * rs[2 * e + 1]);
            pl[e] = r00 | (r01 << 16);
            ph[e] = r10 | (r11 << 16);
        }
    };
    if (ROPE) {
        const float *cp = rope_cs + ((int64_t)(klen - 1) * 64 + 8 * c) * 2;
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) {
            const float4 t4 = *reinterpret_cast<const float4 *>(cp + 4 * e4);
            rc[2 * e4] = t4.x; rs[2 * e4] = t4.y; rc[2 * e4 + 1] = t4.z; rs[2 * e4 + 1] = t4.w;
        }
        const uint16_t *kr = q + (int64_t)i * q_ts + k_col + kvh * HD + 8 * c, *vr = q + (int64_t)i * q_ts + v_col + kvh * HD + 8 * c;
        rot8(*reinterpret_cast<const uint4 *>(kr), *reinterpret_cast<const uint4 *>(kr + 64), kfresh[0], kfresh[HDV - 1]);
        vfresh[0] = *reinterpret_cast<const uint4 *>(vr);
        vfresh[HDV - 1] = *reinterpret_cast<const uint4 *>(vr + 64);
        if (blockIdx.x == kvh * nsplit && wid == 0 && g == 0) {      // one workgroup per kv head appends the token to the cache
            uint16_t *kd = cache + (int64_t)(i / T) * cache_bs + (int64_t)(klen - 1) * cache_ts + kvh * HD + 8 * c;
            *reinterpret_cast<uint4 *>(kd) = kfresh[0];
            *reinterpret_cast<uint4 *>(kd + 64) = kfresh[HDV - 1];
            *reinterpret_cast<uint4 *>(kd + v_off) = vfresh[0];
            *reinterpret_cast<uint4 *>(kd + v_off + 64) = vfresh[HDV - 1];
        }
    }
#pragma unroll
    for (int r = 0; r < G; r++) {
        uint4 tq[HDV];
#pragma unroll
        for (int hv = 0; hv < HDV; hv++) tq[hv] = *reinterpret_cast<const uint4 *>(q + (int64_t)i * q_ts + (h0 + r) * HD + hv * 64 + 8 * c);
        if (ROPE) rot8(tq[0], tq[HDV - 1], tq[0], tq[HDV - 1]);
#pragma unroll
        for (int hv = 0; hv < HDV; hv++) {
            const uint32_t *u = reinterpret_cast<const uint32_t *>(&tq[hv]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                qv[r][hv * 8 + 2 * e] = __uint_as_float(u[e] << 16) * scale;
                qv[r][hv * 8 + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u) * scale;
            }
        }
    }
    const uint16_t *kb = cache + (int64_t)(i / T) * cache_bs + kvh * HD + 8 * c;
    const uint16_t *vb = kb + v_off;
    float m[G], l[G], o[G][DV];
#pragma unroll
    for (int r = 0; r < G; r++) {
        m[r] = -1e30f;
        l[r] = 0.0f;
#pragma unroll
        for (int d = 0; d < DV; d++) o[r][d] = 0.0f;
    }
    constexpr int KU = 2;
    for (int key0 = wid * 8 + g; key0 < klen; key0 += 8 * NW * KU) {
        uint4 kk[KU][HDV], vv[KU][HDV];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int key = key0 + u * 8 * NW;
#pragma unroll
            for (int hv = 0; hv < HDV; hv++) {
                kk[u][hv] = make_uint4(0, 0, 0, 0);
                vv[u][hv] = make_uint4(0, 0, 0, 0);
                if (key < klen) {
                    kk[u][hv] = *reinterpret_cast<const uint4 *>(kb + (int64_t)key * cache_ts + hv * 64);
                    vv[u][hv] = *reinterpret_cast<const uint4 *>(vb + (int64_t)key * cache_ts + hv * 64);
                }
                if (ROPE && key == klen - 1) {               // the step's own token: from registers, whatever the cache row holds yet
                    kk[u][hv] = kfresh[hv];
                    vv[u][hv] = vfresh[hv];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < G; r++) {
            float sc[KU];
            float mn = m[r];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                float s = 0.0f;
#pragma unroll
                for (int hv = 0; hv < HDV; hv++) {
                    const uint32_t *ku = reinterpret_cast<const uint32_t *>(&kk[u][hv]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        s = __fmaf_rn(qv[r][hv * 8 + 2 * e], __uint_as_float(ku[e] << 16), s);
                        s = __fmaf_rn(qv[r][hv * 8 + 2 * e + 1], __uint_as_float(ku[e] & 0xffff0000u), s);
                    }
                }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s = (key0 + u * 8 * NW < klen) ? s : -1e30f;
                sc[u] = s;
                mn = fmaxf(mn, s);
            }
            const float a = __expf(m[r] - mn);
            l[r] *= a;
#pragma unroll
            for (int d = 0; d < DV; d++) o[r][d] *= a;
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const float pr = (key0 + u * 8 * NW < klen) ? __expf(sc[u] - mn) : 0.0f;
                l[r] += pr;
#pragma unroll
                for (int hv = 0; hv < HDV; hv++) {
                    const uint32_t *vu = reinterpret_cast<const uint32_t *>(&vv[u][hv]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        o[r][hv * 8 + 2 * e] = __fmaf_rn(pr, __uint_as_float(vu[e] << 16), o[r][hv * 8 + 2 * e]);
                        o[r][hv * 8 + 2 * e + 1] = __fmaf_rn(pr, __uint_as_float(vu[e] & 0xffff0000u), o[r][hv * 8 + 2 * e + 1]);
                    }
                }
            }
            m[r] = mn;
        }
    }
#pragma unroll
    for (int r = 0; r < G; r++) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float m2 = __shfl_xor(m[r], off, 64), l2 = __shfl_xor(l[r], off, 64);
            const float mn = fmaxf(m[r], m2);
            const float a = __expf(m[r] - mn), a2 = __expf(m2 - mn);
            l[r] = l[r] * a + l2 * a2;
#pragma unroll
            for (int d = 0; d < DV; d++) o[r][d] = o[r][d] * a + __shfl_xor(o[r][d], off, 64) * a2;
            m[r] = mn;
        }
    }
    if (NW > 1) {
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < G; r++) {
                comb[wid][r][c][0] = m[r];
                comb[wid][r][c][1] = l[r];
#pragma unroll
                for (int d = 0; d < DV; d++) comb[wid][r][c][2 + d] = o[r][d];
            }
        }
        __syncthreads();
        if (wid == 0 && g == 0) {
#pragma unroll
            for (int r = 0; r < G; r++) {
#pragma unroll
                for (int w = 1; w < NW; w++) {
                    const float m2 = comb[w][r][c][0], l2 = comb[w][r][c][1];
                    const float mn = fmaxf(m[r], m2);
                    const float a = __expf(m[r] - mn), a2 = __expf(m2 - mn);
                    l[r] = l[r] * a + l2 * a2;
#pragma unroll
                    for (int d = 0; d < DV; d++) o[r][d] = o[r][d] * a + comb[w][r][c][2 + d] * a2;
                    m[r] = mn;
                }
            }
        }
    }
    if (wid == 0 && g == 0) {
#pragma unroll
        for (int r = 0; r < G; r++) {
            const float inv = l[r] > 0.0f ? 1.0f / l[r] : 0.0f;
#pragma unroll
            for (int hv = 0; hv < HDV; hv++) {
                uint4 pk;
                pk.x = f32x2_to_bf16x2(o[r][hv * 8 + 0] * inv, o[r][hv * 8 + 1] * inv);
                pk.y = f32x2_to_bf16x2(o[r][hv * 8 + 2] * inv, o[r][hv * 8 + 3] * inv);
                pk.z = f32x2_to_bf16x2(o[r][hv * 8 + 4] * inv, o[r][hv * 8 + 5] * inv);
                pk.w = f32x2_to_bf16x2(o[r][hv * 8 + 6] * inv, o[r][hv * 8 + 7] * inv);
                *reinterpret_cast<uint4 *>(out + (int64_t)i * o_ts + (h0 + r) * HD + hv * 64 + 8 * c) = pk;
            }
        }
    }
}

// interleaved = 0: gate_up row = [gate (ffn) | up (ffn)];  1: [gate_0, up_0, gate_1, up_1, ...] (the row order of a weight
// packed for the fused IFH_ACT_SILU_GLU epilogue)
__global__ __launch_bounds__(256) void k_silu_mul(const uint16_t *__restrict__ gu, uint16_t *__restrict__ out, int64_t rows,
                                                  int ffn, int interleaved)
{
    const int nv = ffn >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * nv) return;
    const int64_t row = idx / nv;
    const int j = (int)(idx % nv);
    uint32_t r[4];
    if (interleaved) {
        const uint4 a = *reinterpret_cast<const uint4 *>(gu + row * 2 * ffn + 16 * j);
        const uint4 b = *reinterpret_cast<const uint4 *>(gu + row * 2 * ffn + 16 * j + 8);
        const uint32_t u[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};       // word e = (gate, up) pair e
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g0 = __uint_as_float(u[2 * e] << 16), u0 = __uint_as_float(u[2 * e] & 0xffff0000u);
            const float g1 = __uint_as_float(u[2 * e + 1] << 16), u1 = __uint_as_float(u[2 * e + 1] & 0xffff0000u);
            r[e] = f32x2_to_bf16x2(g0 / (1.0f + __expf(-g0)) * u0, g1 / (1.0f + __expf(-g1)) * u1);
        }
    } else {
        const uint4 a = *reinterpret_cast<const uint4 *>(gu + row * 2 * ffn + 8 * j);
        const uint4 b = *reinterpret_cast<const uint4 *>(gu + row * 2 * ffn + ffn + 8 * j);
        const uint32_t *ua = reinterpret_cast<const uint32_t *>(&a), *ub = reinterpret_cast<const uint32_t *>(&b);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float g0 = __uint_as_float(ua[e] << 16), g1 = __uint_as_float(ua[e] & 0xffff0000u);
            const float u0 = __uint_as_float(ub[e] << 16), u1 = __uint_as_float(ub[e] & 0xffff0000u);
            r[e] = f32x2_to_bf16x2(g0 / (1.0f + __expf(-g0)) * u0, g1 / (1.0f + __expf(-g1)) * u1);
        }
    }
    *reinterpret_cast<uint4 *>(out + row * ffn + 8 * j) = make_uint4(r[0], r[1], r[2], r[3]);
}

__global__ void k_add_i32_vec(int32_t *__restrict__ v, const int32_t *__restrict__ mask, int n, int delta)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && (!mask || mask[i])) v[i] += delta;
}

// Prompt attention on the matrix cores (Cluster/InfernLLMWorker.py:103-119: the first forward of generate() over the whole context).
// The per-token kernel above spends 476 us per layer on 64 rows x 192 tokens (a quarter of the layer): every query token re-reads its
// row's K/V from L2 and multiplies on the vector ALUs.  Here 16 query tokens x G heads share each 64-key tile: G waves, wave g = query
// head kvh * G + g; lane (fr, fg) holds query fr's scores of keys 16 c + 4 fg + r of a tile (softmax over a query = over 4 lanes and
// 16 registers), P is rounded to bf16 for the second product (as a bf16 model does); masks: key index < key_len[token] (causal and
// padding rows alike), tiles past the block's largest key_len are skipped.  V stays row-major in LDS and is read transposed by
// ds_read_b64_tr_b16, as in attn.hip.
template <int HDV>
__global__ __launch_bounds__(512) void k_attn_gqa_prefill(const uint16_t *__restrict__ q, int64_t q_ts, const uint16_t *__restrict__ cache,
                                                          int64_t cache_bs, int64_t cache_ts, int v_off, uint16_t *__restrict__ out,
                                                          int64_t o_ts, const int32_t *__restrict__ key_len, int T, float scale, int G)
{
    constexpr int HDT = 64 * HDV, KLD = HDT + 8, KT = 64, NS = 2 * HDV, ND = 4 * HDV, RV = HDT / 8;
    __shared__ __attribute__((aligned(16))) uint16_t Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[KT * KLD];
    const int b = blockIdx.z, kvh = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthreads = blockDim.x;
    const int fr = lane & 15, fg = lane >> 4;
    const int t = blockIdx.x * 16 + fr;
    const bool qok = t < T;
    const int64_t tok = (int64_t)b * T + (qok ? t : T - 1);
    const int klen = qok ? key_len[tok] : 0;
    int kmax = klen;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    const int h = kvh * G + wid;

    bf16x8_t qf[NS];
    {
        const uint16_t *qp = q + tok * q_ts + (int64_t)h * HDT;
#pragma unroll
        for (int s = 0; s < NS; s++) qf[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4 *>(qp + 32 * s + 8 * fg));
    }
    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) o[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -1e30f, lrun = 0.0f;
    const uint16_t *kb = cache + (int64_t)b * cache_bs + (int64_t)kvh * HDT, *vb = kb + v_off;
    constexpr float kLog2e = 1.4426950408889634f;
    const float c1 = scale * kLog2e;

    const int ntile = (kmax + KT - 1) / KT;
    for (int kt = 0; kt < ntile; kt++) {
        const int kbase = kt * KT;
        __syncthreads();
        for (int v = tid; v < KT * RV; v += nthreads) {
            const int key = v / RV, dv = (v - key * RV) * 8, kg = kbase + key;
            uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
            if (kg < kmax) {
                kk = *reinterpret_cast<const uint4 *>(kb + (int64_t)kg * cache_ts + dv);
                vv = *reinterpret_cast<const uint4 *>(vb + (int64_t)kg * cache_ts + dv);
            }
            *reinterpret_cast<uint4 *>(&Ks[key * KLD + dv]) = kk;
            *reinterpret_cast<uint4 *>(&Vs[key * KLD + dv]) = vv;
        }
        __syncthreads();
        // S^T tiles: 4 x (16 keys x 16 queries)
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < NS; ds++) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t *>(&Ks[(c * 16 + fr) * KLD + ds * 32 + fg * 8]);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ds], s[c], 0, 0, 0);
            }
        }
        float mloc = -1e30f;
        if (kbase + KT > klen) {           // (lane-wise: a query whose keys end inside this tile, or before it)
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kidx = kbase + c * 16 + 4 * fg + r;
                    s[c][r] = kidx < klen ? s[c][r] : -1e30f;
                }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) mloc = fmaxf(fmaxf(mloc, fmaxf(s[c][0], s[c][1])), fmaxf(s[c][2], s[c][3]));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        // exp(scale (x - m)) = exp2(x c1 - m c1): one fma + the hardware's native exp2 per score
        const float mscaled = mnew * c1;
        const float alpha = __builtin_amdgcn_exp2f(mrun * c1 - mscaled);
        float lsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], c1, -mscaled));
                s[c][r] = pv;
                lsum += pv;
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        lrun = lrun * alpha + lsum;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < ND; i++) {
            o[i][0] *= alpha;
            o[i][1] *= alpha;
            o[i][2] *= alpha;
            o[i][3] *= alpha;
        }
        // O^T += V^T . P^T (attn.hip, k_attn_prefill: the operand layout and the transposed LDS read)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint4 pb;
            pb.x = pack2(s[2 * ks][0], s[2 * ks][1]);
            pb.y = pack2(s[2 * ks][2], s[2 * ks][3]);
            pb.z = pack2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
            pb.w = pack2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pb);
            const uint32_t vaddr = (uint32_t)(uintptr_t)(&Vs[(ks * 32 + 4 * fg + (fr >> 2)) * KLD + 4 * (fr & 3)]);
#pragma unroll
            for (int d4 = 0; d4 < ND; d4 += 4) {
                uint2 lo[4], hi[4];
#pragma unroll
                for (int dt = 0; dt < 4; dt++) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dt]) : "v"(vaddr), "n"((d4 + dt) * 32) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(vaddr), "n"((d4 + dt) * 32 + 16 * KLD * 2) : "memory");
                }
                // (the wait names the eight results as in/out operands: the MFMAs below depend on IT, not only on the read statements)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                             :
                             : "memory");
#pragma unroll
                for (int dt = 0; dt < 4; dt++) {
                    uint4 va;
                    va.x = lo[dt].x;
                    va.y = lo[dt].y;
                    va.z = hi[dt].x;
                    va.w = hi[dt].y;
                    o[d4 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), pf, o[d4 + dt], 0, 0, 0);
                }
            }
        }
    }
    if (qok) {
        const float inv = lrun > 0.0f ? 1.0f / lrun : 0.0f;
        uint16_t *op = out + tok * o_ts + (int64_t)h * HDT;
#pragma unroll
        for (int dt = 0; dt < ND; dt++) {
            uint2 pk;
            pk.x = pack2(o[dt][0] * inv, o[dt][1] * inv);
            pk.y = pack2(o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<uint2 *>(op + dt * 16 + 4 * fg) = pk;
        }
    }
}

// Decode attention of a head_dim-128 model on the matrix cores: ONE workgroup per (token, kv head), the group's G <= 16 query heads as
// the 16 "query rows" of k_attn_gqa_prefill's layout, so the K / V rows of the token's cache row are read ONCE for all of them (the
// per-head workgroups of k_attn_gqa re-read them per query head: 79 MB per layer at 64 tokens x 200 keys, 18 us).  Four waves take
// the 32-key tiles in turn (wave-private LDS tiles, no barrier in the loop) and merge their (max, sum, O) at the end.  ROPE as in
// k_attn_gqa: rotary embedding of q (a lane's fragments s and s + 2 are the pairs (j, j + 64)) and of the step's own key, which the
// wave that owns its tile writes into the tile (and into the cache) in place of the cache row.
template <bool ROPE>
__global__ __launch_bounds__(256, 2) void k_attn_gqa_decode_mfma(const uint16_t *__restrict__ q, int64_t q_ts, uint16_t *__restrict__ cache,
                                                                 int64_t cache_bs, int64_t cache_ts, int v_off, uint16_t *__restrict__ out,
                                                                 int64_t o_ts, const int32_t *__restrict__ key_len, float scale, int G,
                                                                 const float *__restrict__ rope_cs, int k_col, int v_col)
{
    constexpr int HDT = 128, KLD = HDT + 8, KT = 32, NS = 4, ND = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int i = blockIdx.y, kvh = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    uint16_t *Ks = reinterpret_cast<uint16_t *>(smem) + wid * (2 * KT * KLD), *Vs = Ks + KT * KLD;
    const int klen = key_len[i];
    const bool hok = fr < G;
    const int h = kvh * G + (hok ? fr : 0);
    const uint16_t *qrow = q + (int64_t)i * q_ts;

    // Q fragments (B operand of S^T = K.Q^T): head fr, dims 32 s + 8 fg .. + 8; rows past the group are zero
    bf16x8_t qf[NS];
    {
        uint4 tq[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) tq[s] = hok ? *reinterpret_cast<const uint4 *>(qrow + (int64_t)h * HDT + 32 * s + 8 * fg) : make_uint4(0, 0, 0, 0);
        if (ROPE) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const float *cp = rope_cs + ((int64_t)(klen - 1) * 64 + 32 * s + 8 * fg) * 2;
                uint32_t *ul = reinterpret_cast<uint32_t *>(&tq[s]), *uh = reinterpret_cast<uint32_t *>(&tq[s + 2]);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float4 cs4 = *reinterpret_cast<const float4 *>(cp + 4 * e);       // cos, sin of dims 2 e, 2 e + 1
                    const float a0 = __uint_as_float(ul[e] << 16), a1 = __uint_as_float(ul[e] & 0xffff0000u);
                    const float b0 = __uint_as_float(uh[e] << 16), b1 = __uint_as_float(uh[e] & 0xffff0000u);
                    const uint32_t r00 = f32_to_bf16(a0 * cs4.x - b0 * cs4.y), r01 = f32_to_bf16(a1 * cs4.z - b1 * cs4.w);
                    const uint32_t r10 = f32_to_bf16(b0 * cs4.x + a0 * cs4.y), r11 = f32_to_bf16(b1 * cs4.z + a1 * cs4.w);
                    ul[e] = r00 | (r01 << 16);
                    uh[e] = r10 | (r11 << 16);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; s++) qf[s] = __builtin_bit_cast(bf16x8_t, tq[s]);
    }
    f32x4 o[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -1e30f, lrun = 0.0f;
    const uint16_t *kb = cache + (int64_t)i * cache_bs + (int64_t)kvh * HDT, *vb = kb + v_off;
    constexpr float kLog2e = 1.4426950408889634f;
    const float c1 = scale * kLog2e;
    const int ntile = (klen + KT - 1) / KT;

    for (int kt = wid; kt < ntile; kt += 4) {
        const int kbase = kt * KT;
        // the tile: 32 keys x 16 chunks of 16 bytes for K and for V; lane l takes chunks l % 16 of keys l / 16 + 4 it
        uint4 kk[8], vv[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int key = (lane >> 4) + 4 * it, kg = kbase + key, dv = (lane & 15) * 8;
            kk[it] = make_uint4(0, 0, 0, 0);
            vv[it] = kk[it];
            if (kg < klen && !(ROPE && kg == klen - 1)) {
                kk[it] = *reinterpret_cast<const uint4 *>(kb + (int64_t)kg * cache_ts + dv);
                vv[it] = *reinterpret_cast<const uint4 *>(vb + (int64_t)kg * cache_ts + dv);
            }
        }
        if (ROPE && klen - 1 >= kbase && klen - 1 < kbase + KT) {
            // the step's own token: chunk c = lane % 16 of its rotated key (partner chunk c ^ 8) and of its value, by the lanes that
            // hold that key's slot (lane / 16 + 4 it == (klen - 1) - kbase); appended to the cache as well
            const int rowl = klen - 1 - kbase, c = lane & 15, cl = c & 7;
            if ((lane >> 4) == (rowl & 3)) {
                const uint16_t *kr = qrow + k_col + kvh * HDT, *vr = qrow + v_col + kvh * HDT;
                const uint4 lo = *reinterpret_cast<const uint4 *>(kr + 8 * cl), hi = *reinterpret_cast<const uint4 *>(kr + 64 + 8 * cl);
                const float *cp = rope_cs + ((int64_t)(klen - 1) * 64 + 8 * cl) * 2;
                const uint32_t *ul = reinterpret_cast<const uint32_t *>(&lo), *uh = reinterpret_cast<const uint32_t *>(&hi);
                uint4 rk;
                uint32_t *pr = reinterpret_cast<uint32_t *>(&rk);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float4 cs4 = *reinterpret_cast<const float4 *>(cp + 4 * e);
                    const float a0 = __uint_as_float(ul[e] << 16), a1 = __uint_as_float(ul[e] & 0xffff0000u);
                    const float b0 = __uint_as_float(uh[e] << 16), b1 = __uint_as_float(uh[e] & 0xffff0000u);
                    const uint32_t r00 = f32_to_bf16(a0 * cs4.x - b0 * cs4.y), r01 = f32_to_bf16(a1 * cs4.z - b1 * cs4.w);
                    const uint32_t r10 = f32_to_bf16(b0 * cs4.x + a0 * cs4.y), r11 = f32_to_bf16(b1 * cs4.z + a1 * cs4.w);
                    pr[e] = c < 8 ? (r00 | (r01 << 16)) : (r10 | (r11 << 16));
                }
                const uint4 rv = *reinterpret_cast<const uint4 *>(vr + 8 * c);
                uint16_t *kd = cache + (int64_t)i * cache_bs + (int64_t)(klen - 1) * cache_ts + kvh * HDT + 8 * c;
                *reinterpret_cast<uint4 *>(kd) = rk;
                *reinterpret_cast<uint4 *>(kd + v_off) = rv;
#pragma unroll
                for (int it = 0; it < 8; it++)
                    if (it == (rowl >> 2)) { kk[it] = rk; vv[it] = rv; }
            }
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int key = (lane >> 4) + 4 * it, dv = (lane & 15) * 8;
            *reinterpret_cast<uint4 *>(&Ks[key * KLD + dv]) = kk[it];
            *reinterpret_cast<uint4 *>(&Vs[key * KLD + dv]) = vv[it];
        }
        // (wave-private tile: the wave's own LDS writes are ordered before its reads)
        f32x4 s[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < NS; ds++) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t *>(&Ks[(c * 16 + fr) * KLD + ds * 32 + fg * 8]);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ds], s[c], 0, 0, 0);
            }
        }
        float mloc = -1e30f;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int kidx = kbase + c * 16 + 4 * fg + r;
                s[c][r] = kidx < klen ? s[c][r] : -1e30f;
                mloc = fmaxf(mloc, s[c][r]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        const float mscaled = mnew * c1;
        const float alpha = __builtin_amdgcn_exp2f(mrun * c1 - mscaled);
        float lsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], c1, -mscaled));
                s[c][r] = pv;
                lsum += pv;
            }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        lrun = lrun * alpha + lsum;
        mrun = mnew;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
        }
        uint4 pb;
        pb.x = pack2(s[0][0], s[0][1]);
        pb.y = pack2(s[0][2], s[0][3]);
        pb.z = pack2(s[1][0], s[1][1]);
        pb.w = pack2(s[1][2], s[1][3]);
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pb);
        const uint32_t vaddr = (uint32_t)(uintptr_t)(&Vs[(4 * fg + (fr >> 2)) * KLD + 4 * (fr & 3)]);
#pragma unroll
        for (int d4 = 0; d4 < ND; d4 += 4) {
            uint2 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dt]) : "v"(vaddr), "n"((d4 + dt) * 32) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(vaddr), "n"((d4 + dt) * 32 + 16 * KLD * 2) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                         :
                         : "memory");
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                uint4 va;
                va.x = lo[dt].x; va.y = lo[dt].y; va.z = hi[dt].x; va.w = hi[dt].y;
                o[d4 + dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), pf, o[d4 + dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile's reads are done before the next tile overwrites it
    }

    // merge the four waves: (m, l) per head and O^T through LDS (the tiles are idle), wave 0 finishes
    __syncthreads();
    float *mg = reinterpret_cast<float *>(smem);                 // [4 waves][64 lanes][2 + 32]
    {
        float *mp = mg + (wid * 64 + lane) * 34;
        mp[0] = mrun;
        mp[1] = lrun;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            mp[2 + 4 * d] = o[d][0]; mp[3 + 4 * d] = o[d][1]; mp[4 + 4 * d] = o[d][2]; mp[5 + 4 * d] = o[d][3];
        }
    }
    __syncthreads();
    if (wid == 0) {
        float mall = mrun;
#pragma unroll
        for (int w = 1; w < 4; w++) mall = fmaxf(mall, mg[(w * 64 + lane) * 34]);
        float lall = 0.0f;
        f32x4 acc[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) acc[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float *mp = mg + (w * 64 + lane) * 34;
            const float a = __builtin_amdgcn_exp2f((mp[0] - mall) * c1);
            lall += mp[1] * a;
#pragma unroll
            for (int d = 0; d < ND; d++) {
                acc[d][0] += mp[2 + 4 * d] * a; acc[d][1] += mp[3 + 4 * d] * a; acc[d][2] += mp[4 + 4 * d] * a; acc[d][3] += mp[5 + 4 * d] * a;
            }
        }
        if (hok) {
            const float inv = lall > 0.0f ? 1.0f / lall : 0.0f;
            uint16_t *op = out + (int64_t)i * o_ts + (int64_t)h * HDT;
#pragma unroll
            for (int d = 0; d < ND; d++) {
                uint2 pk;
                pk.x = pack2(acc[d][0] * inv, acc[d][1] * inv);
                pk.y = pack2(acc[d][2] * inv, acc[d][3] * inv);
                *reinterpret_cast<uint2 *>(op + d * 16 + 4 * fg) = pk;
            }
        }
    }
}

template <int G, int HDV>
static void attn_gqa_launch(const ifh_gqa_desc *d, hipStream_t st, int gtot)
{
    const int nsplit = gtot / G;
    dim3 grid(d->nkv * nsplit, d->ntokens);
    if constexpr (HDV == 2) {
        if (d->rope_cos_sin) {
            const int k_col = d->nheads * d->head_dim, v_col = (d->nheads + d->nkv) * d->head_dim;
            if (d->max_keys > 256)
                hipLaunchKernelGGL((k_attn_gqa<4, G, HDV, true>), grid, dim3(256), 0, st, (const uint16_t *)d->q, d->q_ts, (uint16_t *)d->cache,
                                   d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->tokens_per_row, d->scale, gtot,
                                   nsplit, d->rope_cos_sin, k_col, v_col);
            else
                hipLaunchKernelGGL((k_attn_gqa<1, G, HDV, true>), grid, dim3(64), 0, st, (const uint16_t *)d->q, d->q_ts, (uint16_t *)d->cache,
                                   d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->tokens_per_row, d->scale, gtot,
                                   nsplit, d->rope_cos_sin, k_col, v_col);
            return;
        }
    }
    if (d->max_keys > 256)      // (8 waves per (token, kv head) measured slower: 27.8 vs 21.5 us at 192-256 keys, G = 6)
        hipLaunchKernelGGL((k_attn_gqa<4, G, HDV>), grid, dim3(256), 0, st, (const uint16_t *)d->q, d->q_ts,
                           (uint16_t *)d->cache, d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts,
                           d->key_len, d->tokens_per_row, d->scale, gtot, nsplit);
    else
        hipLaunchKernelGGL((k_attn_gqa<1, G, HDV>), grid, dim3(64), 0, st, (const uint16_t *)d->q, d->q_ts,
                           (uint16_t *)d->cache, d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts,
                           d->key_len, d->tokens_per_row, d->scale, gtot, nsplit);
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_rmsnorm_bf16(const void *x, const float *gamma, void *out, int rows, int dim, float eps,
                                ifh_stream_t stream)
{
    IFH_CHECK_ARG(rows >= 0);
    if (rows == 0) return IFH_OK;
    IFH_CHECK_ARG(x && gamma && out && dim > 0 && dim % 8 == 0 && eps > 0.0f);
    IFH_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)gamma)) & 15) == 0);
    hipLaunchKernelGGL(k_rmsnorm, dim3((rows + 3) / 4), dim3(256), 0, as_stream(stream), (const uint16_t *)x, gamma,
                       (uint16_t *)out, rows, dim, eps);
    IFH_LAUNCH_CHECK("rmsnorm");
    return IFH_OK;
}

extern "C" int ifh_rope_append_bf16(void *qkv, int64_t qkv_ld, const float *cos_sin, int max_pos, void *cache,
                                    int64_t cache_bs, int64_t cache_ts, const int32_t *pos0, const int32_t *nvalid,
                                    int nrows, int tokens_per_row, int nheads, int nkv, int head_dim, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0 && tokens_per_row >= 1);
    if (nrows == 0) return IFH_OK;
    IFH_CHECK_ARG(qkv && cos_sin && cache && pos0 && nvalid && max_pos >= 1);
    IFH_CHECK_ARG(nheads >= 1 && nkv >= 1 && head_dim >= 2 && head_dim % 2 == 0);
    IFH_CHECK_ARG(qkv_ld >= (int64_t)(nheads + 2 * nkv) * head_dim && cache_ts >= (int64_t)2 * nkv * head_dim &&
                  cache_bs >= cache_ts * max_pos);
    IFH_CHECK_ARG(head_dim % 16 == 0 && qkv_ld % 8 == 0 && cache_bs % 8 == 0 && cache_ts % 8 == 0 &&
                  ((((uintptr_t)qkv) | ((uintptr_t)cache) | ((uintptr_t)cos_sin)) & 15) == 0);      // 16-byte accesses
    const int64_t total = (int64_t)nrows * tokens_per_row * (nheads + 2 * nkv) * (head_dim / 16);
    IFH_CHECK_ARG(total / 256 < (1ll << 31));
    hipLaunchKernelGGL(k_rope_append, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (uint16_t *)qkv, qkv_ld, cos_sin, (uint16_t *)cache, cache_bs, cache_ts, pos0, nvalid,
                       tokens_per_row, nheads, nkv, head_dim, max_pos, total);
    IFH_LAUNCH_CHECK("rope_append");
    return IFH_OK;
}

extern "C" int ifh_attn_gqa_bf16(const ifh_gqa_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d && d->ntokens >= 0);
    if (d->ntokens == 0) return IFH_OK;
    IFH_CHECK_ARG(d->q && d->cache && d->out && d->key_len && d->nheads >= 1 && d->nkv >= 1 && d->nheads % d->nkv == 0);
    IFH_CHECK_ARG(d->head_dim == 64 || d->head_dim == 128);
    IFH_CHECK_ARG(d->tokens_per_row >= 1 && d->max_keys >= 1 && d->ntokens < 65536 * 16);
    IFH_CHECK_ARG(d->q_ts % 8 == 0 && d->o_ts % 8 == 0 && d->cache_bs % 8 == 0 && d->cache_ts % 8 == 0 && d->v_off % 8 == 0);
    const int G = d->nheads / d->nkv;
    IFH_CHECK_ARG(G <= 8);
    // the fused rotary embedding + KV append: a decode step (one token per row) at head_dim 128, a 16-byte addressable table
    IFH_CHECK_ARG(!d->rope_cos_sin || (d->tokens_per_row == 1 && d->head_dim == 128 && (((uintptr_t)d->rope_cos_sin) & 15) == 0));
    hipStream_t st = as_stream(stream);
    // a decode step at head_dim 128: one workgroup per (token, kv head), the group's heads as the query rows of the matrix-core layout
    constexpr int dec_mfma = 1;       // fixed by measurement (profiles/NOTES.md)
    if (dec_mfma && d->tokens_per_row == 1 && d->head_dim == 128) {
        constexpr int bytes = 4 * 2 * 32 * 136 * 2;
        static DeviceOnce attr_once;
        int attr_dev = 0;
        if (attr_once.needed(&attr_dev)) {
            hipError_t e = hipFuncSetAttribute((const void *)k_attn_gqa_decode_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_attn_gqa_decode_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return check_hip(e, "attn_gqa_decode lds attr");
            attr_once.done(attr_dev);
        }
        const dim3 grid((unsigned)d->nkv, (unsigned)d->ntokens);
        const int k_col = d->nheads * d->head_dim, v_col = (d->nheads + d->nkv) * d->head_dim;
        if (d->rope_cos_sin)
            hipLaunchKernelGGL(k_attn_gqa_decode_mfma<true>, grid, dim3(256), bytes, st, (const uint16_t *)d->q, d->q_ts, (uint16_t *)d->cache, d->cache_bs,
                               d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->scale, G, d->rope_cos_sin, k_col, v_col);
        else
            hipLaunchKernelGGL(k_attn_gqa_decode_mfma<false>, grid, dim3(256), bytes, st, (const uint16_t *)d->q, d->q_ts, (uint16_t *)d->cache, d->cache_bs,
                               d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->scale, G, (const float *)nullptr, 0, 0);
        IFH_LAUNCH_CHECK("attn_gqa_decode");
        return IFH_OK;
    }
    // a prompt (16 or more tokens per row): the matrix-core kernel, 16 query tokens x G heads per workgroup
    constexpr int mfma_on = 1;       // fixed by measurement (profiles/NOTES.md)
    if (mfma_on && d->tokens_per_row >= 16 && d->ntokens % d->tokens_per_row == 0) {
        const dim3 grid((unsigned)((d->tokens_per_row + 15) / 16), (unsigned)d->nkv, (unsigned)(d->ntokens / d->tokens_per_row));
        if (d->head_dim == 128)
            hipLaunchKernelGGL(k_attn_gqa_prefill<2>, grid, dim3(G * 64), 0, st, (const uint16_t *)d->q, d->q_ts, (const uint16_t *)d->cache,
                               d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->tokens_per_row, d->scale, G);
        else
            hipLaunchKernelGGL(k_attn_gqa_prefill<1>, grid, dim3(G * 64), 0, st, (const uint16_t *)d->q, d->q_ts, (const uint16_t *)d->cache,
                               d->cache_bs, d->cache_ts, d->v_off, (uint16_t *)d->out, d->o_ts, d->key_len, d->tokens_per_row, d->scale, G);
        IFH_LAUNCH_CHECK("attn_gqa_prefill");
        return IFH_OK;
    }
    // query heads per workgroup: the whole group while that fills the chip twice over, else the largest divisor of G that does (a
    // decode step: few tokens); IFH_GQA_GS overrides (tuning switch)
    constexpr int gs_env = 0;
    int gs = G;
    if (gs_env > 0 && G % gs_env == 0)
        gs = gs_env;
    else
        while (gs > 1 && (int64_t)d->nkv * (G / gs) * d->ntokens < 2 * device_cu_count_physical()) {
            int nx = gs - 1;
            while (nx > 1 && G % nx) nx--;
            gs = nx;
        }
#define IFH_GQA(GG)                                                      \
    case GG:                                                             \
        if (d->head_dim == 128) attn_gqa_launch<GG, 2>(d, st, G);        \
        else attn_gqa_launch<GG, 1>(d, st, G);                           \
        break;
    switch (gs) {
        IFH_GQA(1) IFH_GQA(2) IFH_GQA(3) IFH_GQA(4) IFH_GQA(5) IFH_GQA(6) IFH_GQA(7) IFH_GQA(8)
    }
#undef IFH_GQA
    IFH_LAUNCH_CHECK("attn_gqa");
    return IFH_OK;
}

extern "C" int ifh_silu_mul_bf16(const void *gate_up, void *out, int64_t rows, int ffn, int interleaved, ifh_stream_t stream)
{
    IFH_CHECK_ARG(rows >= 0);
    if (rows == 0) return IFH_OK;
    IFH_CHECK_ARG(gate_up && out && ffn > 0 && ffn % 8 == 0);
    const int64_t total = rows * (ffn / 8);
    IFH_CHECK_ARG(total / 256 < (1ll << 31));
    hipLaunchKernelGGL(k_silu_mul, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       (const uint16_t *)gate_up, (uint16_t *)out, rows, ffn, interleaved);
    IFH_LAUNCH_CHECK("silu_mul");
    return IFH_OK;
}

extern "C" int ifh_add_i32_vec(int32_t *values, const int32_t *mask, int n, int delta, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(values);
    hipLaunchKernelGGL(k_add_i32_vec, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), values, mask, n, delta);
    IFH_LAUNCH_CHECK("add_i32_vec");
    return IFH_OK;
}
