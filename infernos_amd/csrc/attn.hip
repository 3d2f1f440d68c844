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
#include "attn_core.h"

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

__global__ __launch_bounds__(256) void k_attn_prefill(const AttnParams p)
{
    __shared__ __attribute__((aligned(16))) uint16_t Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[KT * KLD];      // V row-major like K: the PV operand is read TRANSPOSED by
                                                                        // ds_read_b64_tr_b16 (round 3; the V^T image of rounds 1-2 cost
                                                                        // sixteen 2-byte LDS stores per thread and tile)
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
    const bool has_rb = p.relbias != nullptr;
    const float *rb = has_rb ? p.relbias + (((int64_t)b * p.Tq + (qi < p.Tq ? qi : 0)) * p.H + h) * p.nrel : nullptr;
    const int half = p.nrel >> 1;

    const int ntile = (klen + KT - 1) / KT;
    // K/V tiles travel through registers one tile ahead: the loads of tile kt+1 are in flight while tile kt is being
    // multiplied (a load -> LDS -> barrier sequence per tile left the block waiting on HBM/L2 24 times at T = 1500)
    uint4 kk0 = make_uint4(0, 0, 0, 0), kk1 = kk0, vx0 = kk0, vx1 = kk0;
#define AT_LOAD(KB)                                                                                                  \
    {                                                                                                                \
        const int kg0 = (KB) + (tid >> 3), kg1 = (KB) + ((tid + 256) >> 3);                                          \
        const int dv = (tid & 7) * 8;                                                                                \
        kk0 = kk1 = vx0 = vx1 = make_uint4(0, 0, 0, 0);                                                              \
        if (kg0 < klen) {                                                                                            \
            kk0 = *reinterpret_cast<const uint4 *>(p.k + (int64_t)b * p.k_bs + (int64_t)kg0 * p.k_ts + h * HD + dv); \
            vx0 = *reinterpret_cast<const uint4 *>(p.v + (int64_t)b * p.v_bs + (int64_t)kg0 * p.v_ts + h * HD + dv); \
        }                                                                                                            \
        if (kg1 < klen) {                                                                                            \
            kk1 = *reinterpret_cast<const uint4 *>(p.k + (int64_t)b * p.k_bs + (int64_t)kg1 * p.k_ts + h * HD + dv); \
            vx1 = *reinterpret_cast<const uint4 *>(p.v + (int64_t)b * p.v_bs + (int64_t)kg1 * p.v_ts + h * HD + dv); \
        }                                                                                                            \
    }
#define AT_COMMIT(KKV, VXV, I)                                                                                       \
    {                                                                                                                \
        const int vv = tid + 256 * I;                                                                                \
        const int key = vv >> 3, dv = (vv & 7) * 8;                                                                  \
        *reinterpret_cast<uint4 *>(&Ks[key * KLD + dv]) = KKV;                                                       \
        *reinterpret_cast<uint4 *>(&Vs[key * KLD + dv]) = VXV;                                                       \
    }
    if (ntile > 0) AT_LOAD(0)
    for (int kt = 0; kt < ntile; kt++) {
        const int kbase = kt * KT;
        __syncthreads();
        // stage K [64 keys][64 d] and V^T [64 d][64 keys] from the prefetched registers
        AT_COMMIT(kk0, vx0, 0)
        AT_COMMIT(kk1, vx1, 1)
        __syncthreads();
        if (kt + 1 < ntile) AT_LOAD(kbase + KT)
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
        // bias, mask, running max.  The kernel is VALU-issue-bound (16 scores per lane against 16 MFMAs per tile), so a tile that
        // needs neither the relative-position bias nor the key-length mask -- every tile but the last of a Whisper window --
        // takes the short form: one max per score here, one fma + one exp2 below.
        float mloc = -1e30f;
        if (has_rb || kbase + KT > klen) {
            // the 16 bias values of this lane's scores go out together under the wave-uniform flag (as `if (rb) val += rb[..]` on the
            // per-lane pointer each load had its own exec branch and s_waitcnt vmcnt(0): 16 round trips in series per tile)
            float rbv[4][4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    rbv[c][r] = 0.0f;
                    if (has_rb) {
                        int rel = qi - (kbase + c * 16 + 4 * fg + r);
                        rel = rel < -half ? -half : (rel > half - 1 ? half - 1 : rel);
                        rbv[c][r] = rb[rel + half];
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int kidx = kbase + c * 16 + 4 * fg + r;
                    float val = s[c][r];
                    if (has_rb) val += rbv[c][r];
                    val = (kidx < klen) ? val : -1e30f;
                    s[c][r] = val;
                    mloc = fmaxf(mloc, val);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) mloc = fmaxf(fmaxf(mloc, fmaxf(s[c][0], s[c][1])), fmaxf(s[c][2], s[c][3]));
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(mrun, mloc);
        // exp(x - m) = exp2(x * log2e - m * log2e): one fma + the hardware's native exp2 per score
        constexpr float kLog2e = 1.4426950408889634f;
        const float mscaled = mnew * kLog2e;
        const float alpha = __builtin_amdgcn_exp2f(mrun * kLog2e - mscaled);
        float lsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], kLog2e, -mscaled));
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
        // O^T += V^T . P^T ; k-step ks covers keys 32*ks..32*ks+31, element j <-> key 16*(j>>2) + 4*fg + (j&3).
        // A operand V^T[d = 16 dt + fr][those 8 keys] from the row-major V image: ds_read_b64_tr_b16 hands lane i of a 16-lane
        // group column i of a 4-row x 16-column block (lane 4q + p of the group supplies the address of row q, columns 4p..4p+3):
        // rows = keys 32 ks + 4 fg + (0..3) (+16 for the second half), columns = d 16 dt .. 16 dt + 15.  Every lane is active here.
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint4 pb;
            pb.x = pack2(s[2 * ks][0], s[2 * ks][1]);
            pb.y = pack2(s[2 * ks][2], s[2 * ks][3]);
            pb.z = pack2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
            pb.w = pack2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pb);
            const uint32_t vaddr = (uint32_t)(uintptr_t)(&Vs[(ks * 32 + 4 * fg + (fr >> 2)) * KLD + 4 * (fr & 3)]);
            uint2 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[dt]) : "v"(vaddr), "n"(dt * 32) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[dt]) : "v"(vaddr), "n"(dt * 32 + 16 * KLD * 2) : "memory");
            }
            // (the wait names the eight results as in/out operands: the MFMAs below depend on IT, not only on the read statements,
            // so the scheduler cannot hoist them above the wait)
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

// k_attn_prefill2: the same arithmetic (bit for bit: the order of every sum is k_attn_prefill's) for long query sequences --
// the Whisper encoder's 1500 x 1500 windows.  What k_attn_prefill pays per 64-key tile and wave -- sixteen MFMAs against eight
// ds_read_b128 + sixteen transposed reads, four ds_write_b128 of the register-staged tile and two barriers -- is halved here:
//   * 32 queries per wave (two 16-query blocks share every K / V fragment read), 128 per workgroup;
//   * K / V tiles come by LDS-DMA (global_load_lds_dwordx4: a wave-instruction = 8 keys x 128 B) into two 16 KB buffers, tile
//     kt + 1 in flight while tile kt is multiplied, ONE barrier per tile (own pieces landed -> barrier -> everyone's landed and
//     the other buffer is free);
//   * 128-byte LDS rows, 16-byte chunk c of key r at chunk c ^ 2 ((r / 2) % 4) (applied on the source side of the DMA): the
//     permutation of gemm_big8.hip, conflict-free for the ds_read_b128 rows of K and for the ds_read_b64_tr_b16 blocks of V
//     (it keeps 32-byte pairs together).  Keys past key_len are fetched from the last valid key and masked as before.
constexpr int A2_BUF = 16384, A2_V = 8192;

// the reductions over the four 16-lane rows of a wave (lane ^ 16, lane ^ 32) by row swaps instead of ds_bpermute: after
// v_permlane16_swap of two copies one holds rows (0, 0, 2, 2), the other (1, 1, 3, 3); the sums are the __shfl_xor ones
// with the operands of an addition exchanged on half of the lanes (same bits)
typedef unsigned a2_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float a2_max_rows(float m)
{
    // (the builtins, not asm: a VALU write followed by a row swap of that register needs wait states hipcc only inserts for
    // instructions it can see)
    const unsigned u = __float_as_uint(m);
    const a2_u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    float a;
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(r.x)), "v"(__uint_as_float(r.y)));
    const unsigned w = __float_as_uint(a);
    const a2_u32x2 t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(t.x)), "v"(__uint_as_float(t.y)));
    return a;
}
__device__ __forceinline__ float a2_sum_rows(float l)
{
    const unsigned u = __float_as_uint(l);
    const a2_u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float a = __uint_as_float(r.x) + __uint_as_float(r.y);
    const unsigned w = __float_as_uint(a);
    const a2_u32x2 t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(t.x) + __uint_as_float(t.y);
}

__global__ __launch_bounds__(256, 3) void k_attn_prefill2(const AttnParams p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds2[2 * A2_BUF];
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int qb0 = blockIdx.x * 128 + wid * 32 + fr;     // this lane's query rows: qb0, qb0 + 16
    const int klen = p.key_len ? p.key_len[b] : p.Tk;
    const unsigned lbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds2);

    bf16x8_t qf[2][2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int qi = qb0 + 16 * j;
        const uint16_t *qp = p.q + (int64_t)b * p.q_bs + (int64_t)(qi < p.Tq ? qi : 0) * p.q_ts + h * HD;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint4 t = make_uint4(0, 0, 0, 0);
            if (qi < p.Tq) t = *reinterpret_cast<const uint4 *>(qp + 32 * s + 8 * fg);
            qf[j][s] = __builtin_bit_cast(bf16x8_t, t);
        }
    }
    f32x4 o[2][4];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) o[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-1e30f, -1e30f}, lrun[2] = {0.0f, 0.0f};
    const int ntile = (klen + KT - 1) / KT;

    // DMA: this wave fetches keys 16 wid .. 16 wid + 15 of a tile (two pieces of K, two of V); lane l -> key l / 8 of the piece,
    // LDS chunk l % 8 <- source chunk (l % 8) ^ 2 ((key / 2) % 4)
    const int drow = lane >> 3, dch = (lane & 7) ^ (((drow >> 1) & 3) << 1);
    const unsigned char *ksrc = reinterpret_cast<const unsigned char *>(p.k + (int64_t)b * p.k_bs + h * HD);
    const unsigned char *vsrc = reinterpret_cast<const unsigned char *>(p.v + (int64_t)b * p.v_bs + h * HD);
    const unsigned kts2 = (unsigned)p.k_ts * 2u, vts2 = (unsigned)p.v_ts * 2u;
#define A2_DMA(VOFF, BASE, DST)                                                                                          \
    do {                                                                                                                 \
        unsigned keep_;                                                                                                  \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(VOFF), "s"(BASE), "s"(DST) : "memory");                                       \
    } while (0)
    auto issue_tile = [&](int kt) {
        const unsigned dst = lbase + (unsigned)(kt & 1) * A2_BUF + (unsigned)wid * 2048;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int key = kt * KT + wid * 16 + j * 8 + drow;
            key = key < klen ? key : klen - 1;
            const unsigned ko = (unsigned)key * kts2 + (unsigned)dch * 16u, vo = (unsigned)key * vts2 + (unsigned)dch * 16u;
            A2_DMA(ko, ksrc, dst + j * 1024);
            A2_DMA(vo, vsrc, dst + A2_V + j * 1024);
        }
    };
    // fragment addresses (buffer 0; the other buffer, the key block c and the k step are immediates or one scalar add)
    const int fsw = ((fr & 7) >> 1) << 1;
    const unsigned ka0 = (unsigned)(fr * 128 + ((fg ^ fsw) << 4)), ka1 = (unsigned)(fr * 128 + (((4 + fg) ^ fsw) << 4));
    unsigned va[4];
    {
        const int row = 4 * fg + (fr >> 2), sw = ((row >> 1) & 3) << 1;
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            va[dt] = lbase + A2_V + (unsigned)(row * 128 + (((dt * 2 + ((fr & 3) >> 1)) ^ sw) << 4) + (fr & 1) * 8);
    }

    // the q loads are waited for HERE: left to hipcc, their s_waitcnt vmcnt(0) lands at the first MFMA inside the loop, right
    // behind the next tile's DMA
    asm volatile("" : : "v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));
    if (ntile > 0) issue_tile(0);
    for (int kt = 0; kt < ntile; kt++) {
        const int kbase = kt * KT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < ntile) issue_tile(kt + 1);
        const unsigned bo = (unsigned)(kt & 1) * A2_BUF;
        const unsigned char *kb = lds2 + bo;
        // S^T tiles of both query blocks: 4 x (16 keys x 16 queries) each, every K fragment read once
        f32x4 s[2][4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[0][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s[1][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t *>(kb + c * 2048 + ka0);
            const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t *>(kb + c * 2048 + ka1);
            s[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0][0], s[0][c], 0, 0, 0);
            s[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[1][0], s[1][c], 0, 0, 0);
            s[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[0][1], s[0][c], 0, 0, 0);
            s[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1][1], s[1][c], 0, 0, 0);
        }
        constexpr float kLog2e = 1.4426950408889634f;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float mloc = -1e30f;
            if (kbase + KT > klen) {       // wave-uniform: the last, partial tile
                const int lim = klen - kbase - 4 * fg;      // key c * 16 + r of this lane is valid below lim
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) s[j][c][r] = (c * 16 + r < lim) ? s[j][c][r] : -1e30f;
                }
            }
            // (asm: fmaxf on MFMA results costs a canonicalising v_max each under hipcc)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mloc) : "v"(s[j][c][0]), "v"(s[j][c][1]));
                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mloc) : "v"(s[j][c][2]), "v"(s[j][c][3]));
            }
            mloc = a2_max_rows(mloc);
            const float mnew = fmaxf(mrun[j], mloc);
            const float mscaled = mnew * kLog2e;
            const float alpha = __builtin_amdgcn_exp2f(mrun[j] * kLog2e - mscaled);
            float lsum = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][c][r], kLog2e, -mscaled));
                    s[j][c][r] = pv;
                    lsum += pv;
                }
            }
            lsum = a2_sum_rows(lsum);
            lrun[j] = lrun[j] * alpha + lsum;
            mrun[j] = mnew;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o[j][i][0] *= alpha;
                o[j][i][1] *= alpha;
                o[j][i][2] *= alpha;
                o[j][i][3] *= alpha;
            }
        }
        // O^T += V^T . P^T: every transposed V fragment feeds both query blocks
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8_t pf[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                uint4 pb;
                pb.x = pack2(s[j][2 * ks][0], s[j][2 * ks][1]);
                pb.y = pack2(s[j][2 * ks][2], s[j][2 * ks][3]);
                pb.z = pack2(s[j][2 * ks + 1][0], s[j][2 * ks + 1][1]);
                pb.w = pack2(s[j][2 * ks + 1][2], s[j][2 * ks + 1][3]);
                pf[j] = __builtin_bit_cast(bf16x8_t, pb);
            }
            uint2 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                const unsigned a = va[dt] + bo;
                if (ks == 0) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[dt]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi[dt]) : "v"(a) : "memory");
                } else {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(lo[dt]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(hi[dt]) : "v"(a) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                         :
                         : "memory");
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                uint4 vv;
                vv.x = lo[dt].x;
                vv.y = lo[dt].y;
                vv.z = hi[dt].x;
                vv.w = hi[dt].y;
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, vv);
                o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0], o[0][dt], 0, 0, 0);
                o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1], o[1][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int qi = qb0 + 16 * j;
        if (qi < p.Tq) {
            const float inv = lrun[j] > 0.0f ? 1.0f / lrun[j] : 0.0f;
            uint16_t *op = p.out + (int64_t)b * p.o_bs + (int64_t)qi * p.o_ts + h * HD;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                uint2 pk;
                pk.x = pack2(o[j][dt][0] * inv, o[j][dt][1] * inv);
                pk.y = pack2(o[j][dt][2] * inv, o[j][dt][3] * inv);
                *reinterpret_cast<uint2 *>(op + dt * 16 + 4 * fg) = pk;
            }
        }
    }
#undef A2_DMA
}

// k_attn_prefill_few: at most 16 query rows per (batch, head) -- the beams of a Whisper utterance over its 1500 cross-attention keys
// (5376 launches per bench cycle).  k_attn_prefill runs that as a 64-query block: three of its four waves multiply and exponentiate
// padding rows, and the tile in flight is one.  Here ONE wave owns the (batch, head): the same per-tile arithmetic on its one
// 16-query block (bit-identical to k_attn_prefill), K / V tiles by LDS-DMA into a wave-private ring of AF_RING 16 KB buffers (AF_RING - 1
// tiles in flight behind the one being multiplied; four by default), no barrier anywhere -- own-wave LDS reads behind an LDS-DMA are ordered by vmcnt alone.
template <int AF_RING>
__global__ __launch_bounds__(64) void k_attn_prefill_few(const AttnParams p)
{
    __shared__ __attribute__((aligned(1024))) unsigned char ldsf[AF_RING * A2_BUF];
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x;
    const int fr = lane & 15, fg = lane >> 4;
    const int qi = fr;                                   // this lane's query row (Tq <= 16)
    const int klen = p.key_len ? p.key_len[b] : p.Tk;
    const unsigned lbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ldsf);

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
    const int ntile = (klen + KT - 1) / KT;

    // DMA: piece q of a tile = keys 8 q .. 8 q + 7 (eight pieces of K, eight of V); lane l -> key l / 8 of the piece, LDS chunk l % 8 <-
    // source chunk (l % 8) ^ 2 ((key / 2) % 4): k_attn_prefill2's image
    const int drow = lane >> 3, dch = (lane & 7) ^ (((drow >> 1) & 3) << 1);
    const unsigned char *ksrc = reinterpret_cast<const unsigned char *>(p.k + (int64_t)b * p.k_bs + h * HD);
    const unsigned char *vsrc = reinterpret_cast<const unsigned char *>(p.v + (int64_t)b * p.v_bs + h * HD);
    const unsigned kts2 = (unsigned)p.k_ts * 2u, vts2 = (unsigned)p.v_ts * 2u;
#define AF_DMA(VOFF, BASE, DST)                                                                                          \
    do {                                                                                                                 \
        unsigned keep_;                                                                                                  \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(VOFF), "s"(BASE), "s"(DST) : "memory");                                       \
    } while (0)
    auto issue_tile = [&](int kt, int slot) {
        const unsigned dst = lbase + (unsigned)slot * A2_BUF;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            int key = kt * KT + q * 8 + drow;
            key = key < klen ? key : klen - 1;
            const unsigned ko = (unsigned)key * kts2 + (unsigned)dch * 16u, vo = (unsigned)key * vts2 + (unsigned)dch * 16u;
            AF_DMA(ko, ksrc, dst + q * 1024);
            AF_DMA(vo, vsrc, dst + A2_V + q * 1024);
        }
    };
    const int fsw = ((fr & 7) >> 1) << 1;
    const unsigned ka0 = (unsigned)(fr * 128 + ((fg ^ fsw) << 4)), ka1 = (unsigned)(fr * 128 + (((4 + fg) ^ fsw) << 4));
    unsigned va[4];
    {
        const int row = 4 * fg + (fr >> 2), sw = ((row >> 1) & 3) << 1;
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
            va[dt] = lbase + A2_V + (unsigned)(row * 128 + (((dt * 2 + ((fr & 3) >> 1)) ^ sw) << 4) + (fr & 1) * 8);
    }
    asm volatile("" : : "v"(qf[0]), "v"(qf[1]));      // (the q loads are waited for here, not behind the first DMA)
#pragma unroll
    for (int t = 0; t < AF_RING - 1; t++)
        if (t < ntile) issue_tile(t, t);
    int slot = 0;
    for (int kt = 0; kt < ntile; kt++) {
        const int kbase = kt * KT;
        // tile kt + AF_RING - 1 goes into the buffer tile kt - 1 was multiplied from (its reads were waited for before its MFMAs); 16
        // DMA instructions per tile: tile kt has landed when at most the later tiles' are outstanding
        if (kt + AF_RING - 1 < ntile) {
            issue_tile(kt + AF_RING - 1, slot == 0 ? AF_RING - 1 : slot - 1);
            if (AF_RING == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (AF_RING == 3) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            if (AF_RING == 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        } else {
            const int later = ntile - 1 - kt;           // tiles requested and not yet needed: 0 .. AF_RING - 2
            if (later >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned bo = (unsigned)slot * A2_BUF;
        const unsigned char *kb = ldsf + bo;
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            s[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const bf16x8_t kf0 = *reinterpret_cast<const bf16x8_t *>(kb + c * 2048 + ka0);
            const bf16x8_t kf1 = *reinterpret_cast<const bf16x8_t *>(kb + c * 2048 + ka1);
            s[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, qf[0], s[c], 0, 0, 0);
            s[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, qf[1], s[c], 0, 0, 0);
        }
        constexpr float kLog2e = 1.4426950408889634f;
        float mloc = -1e30f;
        if (kbase + KT > klen) {       // wave-uniform: the last, partial tile
            const int lim = klen - kbase - 4 * fg;
#pragma unroll
            for (int c = 0; c < 4; c++) {
#pragma unroll
                for (int r = 0; r < 4; r++) s[c][r] = (c * 16 + r < lim) ? s[c][r] : -1e30f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mloc) : "v"(s[c][0]), "v"(s[c][1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(mloc) : "v"(s[c][2]), "v"(s[c][3]));
        }
        mloc = a2_max_rows(mloc);
        const float mnew = fmaxf(mrun, mloc);
        const float mscaled = mnew * kLog2e;
        const float alpha = __builtin_amdgcn_exp2f(mrun * kLog2e - mscaled);
        float lsum = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[c][r], kLog2e, -mscaled));
                s[c][r] = pv;
                lsum += pv;
            }
        }
        lsum = a2_sum_rows(lsum);
        lrun = lrun * alpha + lsum;
        mrun = mnew;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            o[i][0] *= alpha;
            o[i][1] *= alpha;
            o[i][2] *= alpha;
            o[i][3] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            uint4 pb;
            pb.x = pack2(s[2 * ks][0], s[2 * ks][1]);
            pb.y = pack2(s[2 * ks][2], s[2 * ks][3]);
            pb.z = pack2(s[2 * ks + 1][0], s[2 * ks + 1][1]);
            pb.w = pack2(s[2 * ks + 1][2], s[2 * ks + 1][3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pb);
            uint2 lo[4], hi[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                const unsigned a = va[dt] + bo;
                if (ks == 0) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[dt]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi[dt]) : "v"(a) : "memory");
                } else {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(lo[dt]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(hi[dt]) : "v"(a) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                         :
                         : "memory");
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                uint4 vv;
                vv.x = lo[dt].x;
                vv.y = lo[dt].y;
                vv.z = hi[dt].x;
                vv.w = hi[dt].y;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, o[dt], 0, 0, 0);
            }
        }
        slot = slot == AF_RING - 1 ? 0 : slot + 1;
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
#undef AF_DMA
}

// One query token per (batch, head) against a KV cache.  A wave is 8 key-groups x 8 lanes; a lane owns
// 8 of the 64 head dims (one 16-byte load per key for K and for V, 128 B coalesced per key).
// Every key-group runs its own online softmax over keys g, g+8*NW, ...; the groups (and the NW
// waves of the block, for long caches) are merged at the end with the usual (m, l, o) rescale.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_attn_decode(const uint16_t *__restrict__ q, int64_t q_bs,
                                                         const uint16_t *__restrict__ k, const uint16_t *__restrict__ v,
                                                         int64_t kv_bs, int64_t kv_ts, uint16_t *__restrict__ out,
                                                         int64_t o_bs, const int32_t *__restrict__ key_len, int S,
                                                         const int32_t *__restrict__ dyn_len, int dyn_add, int kv_group)
{
    __shared__ float comb[NW][8][10];
    const int b = blockIdx.y, h = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 7, g = lane >> 3;
    int klen = key_len ? key_len[b] + dyn_add : (dyn_len ? dyn_len[0] + dyn_add : S);
    float qv[8];
    {
        const uint4 t = *reinterpret_cast<const uint4 *>(q + (int64_t)b * q_bs + h * HD + 8 * c);
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            qv[2 * e] = __uint_as_float(u[e] << 16);
            qv[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
        }
    }
    const uint16_t *kb = k + (int64_t)(b / kv_group) * kv_bs + h * HD + 8 * c;      // kv_group query rows share a cache row
    const uint16_t *vb = v + (int64_t)(b / kv_group) * kv_bs + h * HD + 8 * c;
    float m = -1e30f, l = 0.0f;
    f32x2 q2[4], o2[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        q2[e] = (f32x2){qv[2 * e], qv[2 * e + 1]};
        o2[e] = (f32x2){0.0f, 0.0f};
    }
    // KU keys per key-group per iteration: 2*KU independent 16-byte loads in flight per lane (the loop is
    // pure latency: few hundred keys at most), scores reduced over the 8 dim-lanes, one online-softmax
    // update per iteration.
    constexpr int KU = 4;
    for (int key0 = wid * 8 + g; key0 < klen; key0 += 8 * NW * KU) {
        uint4 kk[KU], vv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int key = key0 + u * 8 * NW;
            kk[u] = make_uint4(0, 0, 0, 0);
            vv[u] = make_uint4(0, 0, 0, 0);
            if (key < klen) {
                kk[u] = ld_stream16(kb + (int64_t)key * kv_ts);
                vv[u] = ld_stream16(vb + (int64_t)key * kv_ts);
            }
        }
        f32x2 klo[KU / 2][4], khi[KU / 2][4], vp[KU][4];
        bool valid[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) valid[u] = key0 + u * 8 * NW < klen;
        attn_unpack<KU>(kk, vv, klo, khi, vp);
        attn_row_update<KU>(q2, klo, khi, vp, valid, m, l, o2);
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        o[2 * e] = o2[e].x;
        o[2 * e + 1] = o2[e].y;
    }
    // merge the 8 key-groups of the wave
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
        const float mn = fmaxf(m, m2);
        const float a = __expf(m - mn), a2 = __expf(m2 - mn);
        l = l * a + l2 * a2;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = o[i] * a + __shfl_xor(o[i], off, 64) * a2;
        m = mn;
    }
    if (NW > 1) {
        if (g == 0) {
            comb[wid][c][0] = m;
            comb[wid][c][1] = l;
#pragma unroll
            for (int i = 0; i < 8; i++) comb[wid][c][2 + i] = o[i];
        }
        __syncthreads();
        if (wid == 0 && g == 0) {
#pragma unroll
            for (int w = 1; w < NW; w++) {
                const float m2 = comb[w][c][0], l2 = comb[w][c][1];
                const float mn = fmaxf(m, m2);
                const float a = __expf(m - mn), a2 = __expf(m2 - mn);
                l = l * a + l2 * a2;
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = o[i] * a + comb[w][c][2 + i] * a2;
                m = mn;
            }
        }
    }
    if (wid == 0 && g == 0) {
        const float inv = l > 0.0f ? 1.0f / l : 0.0f;
        uint4 pk;
        pk.x = pack2(o[0] * inv, o[1] * inv);
        pk.y = pack2(o[2] * inv, o[3] * inv);
        pk.z = pack2(o[4] * inv, o[5] * inv);
        pk.w = pack2(o[6] * inv, o[7] * inv);
        *reinterpret_cast<uint4 *>(out + (int64_t)b * o_bs + h * HD + 8 * c) = pk;
    }
}

// k_attn_decode for G query rows that share one cache row (the beams of an utterance over its cross-attention K/V):
// one workgroup per (cache row, head) serves the G queries, so every key and value is loaded once instead of G times.
// Key assignment, iteration order and arithmetic per query are those of k_attn_decode: same bits.
template <int NW, int G>
__global__ __launch_bounds__(NW * 64) void k_attn_decode_group(const uint16_t *__restrict__ q, int64_t q_bs,
                                                               const uint16_t *__restrict__ k,
                                                               const uint16_t *__restrict__ v, int64_t kv_bs,
                                                               int64_t kv_ts, uint16_t *__restrict__ out, int64_t o_bs,
                                                               int S)
{
    __shared__ float comb[NW][G][8][10];
    const int bg = blockIdx.y, h = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = lane & 7, g = lane >> 3;
    const int klen = S;
    float qv[G][8];
#pragma unroll
    for (int r = 0; r < G; r++) {
        const uint4 t = *reinterpret_cast<const uint4 *>(q + (int64_t)(bg * G + r) * q_bs + h * HD + 8 * c);
        const uint32_t *u = reinterpret_cast<const uint32_t *>(&t);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            qv[r][2 * e] = __uint_as_float(u[e] << 16);
            qv[r][2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
        }
    }
    const uint16_t *kb = k + (int64_t)bg * kv_bs + h * HD + 8 * c;
    const uint16_t *vb = v + (int64_t)bg * kv_bs + h * HD + 8 * c;
    float m[G], l[G];
    f32x2 q2[G][4], o2[G][4];
#pragma unroll
    for (int r = 0; r < G; r++) {
        m[r] = -1e30f;
        l[r] = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            q2[r][e] = (f32x2){qv[r][2 * e], qv[r][2 * e + 1]};
            o2[r][e] = (f32x2){0.0f, 0.0f};
        }
    }
    constexpr int KU = 4;
    for (int key0 = wid * 8 + g; key0 < klen; key0 += 8 * NW * KU) {
        uint4 kk[KU], vv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int key = key0 + u * 8 * NW;
            kk[u] = make_uint4(0, 0, 0, 0);
            vv[u] = make_uint4(0, 0, 0, 0);
            if (key < klen) {
                kk[u] = ld_stream16(kb + (int64_t)key * kv_ts);
                vv[u] = ld_stream16(vb + (int64_t)key * kv_ts);
            }
        }
        f32x2 klo[KU / 2][4], khi[KU / 2][4], vp[KU][4];
        bool valid[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) valid[u] = key0 + u * 8 * NW < klen;
        attn_unpack<KU>(kk, vv, klo, khi, vp);
#pragma unroll
        for (int r = 0; r < G; r++) attn_row_update<KU>(q2[r], klo, khi, vp, valid, m[r], l[r], o2[r]);
    }
    float o[G][8];
#pragma unroll
    for (int r = 0; r < G; r++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            o[r][2 * e] = o2[r][e].x;
            o[r][2 * e + 1] = o2[r][e].y;
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
            for (int i = 0; i < 8; i++) o[r][i] = o[r][i] * a + __shfl_xor(o[r][i], off, 64) * a2;
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
                for (int i = 0; i < 8; i++) comb[wid][r][c][2 + i] = o[r][i];
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
                    for (int i = 0; i < 8; i++) o[r][i] = o[r][i] * a + comb[w][r][c][2 + i] * a2;
                    m[r] = mn;
                }
            }
        }
    }
    if (wid == 0 && g == 0) {
#pragma unroll
        for (int r = 0; r < G; r++) {
            const float inv = l[r] > 0.0f ? 1.0f / l[r] : 0.0f;
            uint4 pk;
            pk.x = pack2(o[r][0] * inv, o[r][1] * inv);
            pk.y = pack2(o[r][2] * inv, o[r][3] * inv);
            pk.z = pack2(o[r][4] * inv, o[r][5] * inv);
            pk.w = pack2(o[r][6] * inv, o[r][7] * inv);
            *reinterpret_cast<uint4 *>(out + (int64_t)(bg * G + r) * o_bs + h * HD + 8 * c) = pk;
        }
    }
}

template <int G>
static void attn_decode_group_launch(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs, int64_t kv_ts,
                                     void *out, int64_t o_bs, int max_keys, int ngroups, int nheads, hipStream_t st)
{
    dim3 grid(nheads, ngroups);
    if (max_keys > 256)
        hipLaunchKernelGGL((k_attn_decode_group<4, G>), grid, dim3(256), 0, st, (const uint16_t *)q, q_bs, (const uint16_t *)k,
                           (const uint16_t *)v, kv_bs, kv_ts, (uint16_t *)out, o_bs, max_keys);
    else
        hipLaunchKernelGGL((k_attn_decode_group<1, G>), grid, dim3(64), 0, st, (const uint16_t *)q, q_bs, (const uint16_t *)k,
                           (const uint16_t *)v, kv_bs, kv_ts, (uint16_t *)out, o_bs, max_keys);
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
    // long query sequences (the Whisper encoder) take the 128-query form; IFH_ATTN_PREFILL2 = 0 / 1 forces one of the two
    // (read per call: the parity test compares the two forms in one process)
    const char *env2 = getenv("IFH_ATTN_PREFILL2");
    const int force2 = env2 && *env2 ? atoi(env2) : -1;
    const bool fits2 = (int64_t)d->tk * d->k_ts * 2 < (int64_t(1) << 31) && (int64_t)d->tk * d->v_ts * 2 < (int64_t(1) << 31);
    // at most 16 query rows (the beams of an utterance over its cross-attention keys): one wave per (batch, head); IFH_ATTN_FEW = 0 / 1
    const char *envf = getenv("IFH_ATTN_FEW");
    const int forcef = envf && *envf ? atoi(envf) : -1;
    if (fits2 && !d->relbias && d->tq <= 16 && (forcef < 0 ? d->tk >= 256 : forcef > 0)) {
        // (IFH_ATTN_FEW_RING: buffers per wave, 2 .. 4; tuning switch)
        const char *envr = getenv("IFH_ATTN_FEW_RING");
        // alone: 2 -> 76 us, 3 -> 91 (768 slots for 1 024 workgroups), 4 -> 77 (its fourth tile makes 64 LDS-DMA pieces outstanding
        // against a 6-bit vmcnt: never fully ahead); C3 equal within the spread for all three: the smallest ring, 32 KB per wave
        const int ring = envr && *envr ? atoi(envr) : 2;
        const dim3 gridf(1, d->nheads, d->nbatch);
        if (ring == 2) hipLaunchKernelGGL(k_attn_prefill_few<2>, gridf, dim3(64), 0, as_stream(stream), p);
        else if (ring == 4) hipLaunchKernelGGL(k_attn_prefill_few<4>, gridf, dim3(64), 0, as_stream(stream), p);
        else hipLaunchKernelGGL(k_attn_prefill_few<3>, gridf, dim3(64), 0, as_stream(stream), p);
    } else if (fits2 && !d->relbias && (force2 < 0 ? d->tq >= 256 : force2 > 0)) {
        dim3 grid((d->tq + 127) / 128, d->nheads, d->nbatch);
        hipLaunchKernelGGL(k_attn_prefill2, grid, dim3(256), 0, as_stream(stream), p);
    } else {
        dim3 grid((d->tq + 63) / 64, d->nheads, d->nbatch);
        hipLaunchKernelGGL(k_attn_prefill, grid, dim3(256), 0, as_stream(stream), p);
    }
    IFH_LAUNCH_CHECK("attn_prefill");
    return IFH_OK;
}

static int attn_decode_launch(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs, int64_t kv_ts,
                              void *out, int64_t o_bs, const int32_t *key_len, int max_keys, int nbatch, int nheads,
                              int head_dim, const int32_t *dyn_len, int dyn_add, int kv_group, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(q && k && v && out && nheads > 0 && head_dim == HD && max_keys >= 1 && kv_group >= 1);
    IFH_CHECK_ARG(q_bs % 8 == 0 && kv_bs % 8 == 0 && kv_ts % 8 == 0 && o_bs % 8 == 0 && nbatch < 65536);
    IFH_CHECK_ARG(kv_group == 1 || !key_len);       // per-row key counts belong to query rows, not to shared cache rows
    dim3 grid(nheads, nbatch);
    // 4 waves per (batch, head) for long caches (measured faster from ~160 keys up) -- and for EVERY growing cache (a decode
    // loop's self-attention: dyn_len, or per-row positions key_len + dyn_add): the waves' partial softmaxes are merged in a
    // different order than one wave's, so the choice must not depend on the cache CAPACITY, which differs between a frozen batch
    // (sized by its text length) and the continuous batch (sized for the longest text it admits) holding the same row
    if (max_keys > 256 || dyn_len || (key_len && dyn_add))
        hipLaunchKernelGGL(k_attn_decode<4>, grid, dim3(256), 0, as_stream(stream), (const uint16_t *)q, q_bs,
                           (const uint16_t *)k, (const uint16_t *)v, kv_bs, kv_ts, (uint16_t *)out, o_bs, key_len, max_keys,
                           dyn_len, dyn_add, kv_group);
    else
        hipLaunchKernelGGL(k_attn_decode<1>, grid, dim3(64), 0, as_stream(stream), (const uint16_t *)q, q_bs,
                           (const uint16_t *)k, (const uint16_t *)v, kv_bs, kv_ts, (uint16_t *)out, o_bs, key_len, max_keys,
                           dyn_len, dyn_add, kv_group);
    IFH_LAUNCH_CHECK("attn_decode");
    return IFH_OK;
}

extern "C" int ifh_attn_decode_bf16(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs,
                                    int64_t kv_ts, void *out, int64_t o_bs, const int32_t *key_len, int max_keys,
                                    int nbatch, int nheads, int head_dim, const int32_t *dyn_len, int dyn_add,
                                    ifh_stream_t stream)
{
    return attn_decode_launch(q, q_bs, k, v, kv_bs, kv_ts, out, o_bs, key_len, max_keys, nbatch, nheads, head_dim, dyn_len,
                              dyn_add, 1, stream);
}

extern "C" int ifh_attn_decode_shared_bf16(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs,
                                           int64_t kv_ts, void *out, int64_t o_bs, int max_keys, int nbatch, int nheads,
                                           int head_dim, int kv_group, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0 && kv_group >= 1);
    if (nbatch == 0) return IFH_OK;
    if (kv_group == 1 || kv_group > 8 || nbatch % kv_group != 0)       // general form: one workgroup per query row
        return attn_decode_launch(q, q_bs, k, v, kv_bs, kv_ts, out, o_bs, nullptr, max_keys, nbatch, nheads, head_dim,
                                  nullptr, 0, kv_group, stream);
    IFH_CHECK_ARG(q && k && v && out && nheads > 0 && head_dim == HD && max_keys >= 1);
    IFH_CHECK_ARG(q_bs % 8 == 0 && kv_bs % 8 == 0 && kv_ts % 8 == 0 && o_bs % 8 == 0 && nbatch / kv_group < 65536);
    const int ng = nbatch / kv_group;
    hipStream_t st = as_stream(stream);
    switch (kv_group) {
#define IFH_AG(G)                                                                                             \
    case G:                                                                                                   \
        attn_decode_group_launch<G>(q, q_bs, k, v, kv_bs, kv_ts, out, o_bs, max_keys, ng, nheads, st);        \
        break;
        IFH_AG(2) IFH_AG(3) IFH_AG(4) IFH_AG(5) IFH_AG(6) IFH_AG(7) IFH_AG(8)
#undef IFH_AG
    }
    IFH_LAUNCH_CHECK("attn_decode_group");
    return IFH_OK;
}
