// nn.hip -- dense building blocks of the speech models on gfx950:
//   * k_igemm: bf16 implicit-GEMM on the matrix cores (v_mfma_f32_16x16x32_bf16) covering
//     Linear, Conv1d (any taps/stride/dilation, channels-last) and the phases of
//     ConvTranspose1d, with fused input LeakyReLU and a fused epilogue (bias, activation,
//     dropout column mask, residual, scale, accumulate).
//   * k_layernorm: wave-per-row LayerNorm with optional fused residual add.
//   * k_transpose: [B][R][C] -> [B][C][R] with cast to bf16 (layout glue).
// The MFMA is issued "swapped" (A = weight rows, B = activation rows) so that a lane ends
// up holding 4 consecutive output channels of one output row -> 8/16-byte stores.
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "igemm.h"

namespace ifh {

// ---- skinny GEMM for decode steps (M <= 256 rows, taps == 1): latency-bound weight streaming.
// One block = 16 output channels x 16 rows; its NW waves split K so that every lane has all of
// its 16-byte weight/activation fragments in flight at once (one memory round trip), straight
// from global/L2 (each weight byte is used once per block: no LDS staging); the NW partial
// tiles are summed through LDS.  Grid (N/16, M/16): 192..768 blocks for the decoder shapes.
// NT = 2: two column tiles per block (32 x 32 outputs): each activation fragment meets two weight fragments and vice
// versa, a third fewer L2 -> CU bytes per output than the 32 x 16 block.
template <int NW, int U, int MT, int NT = 1>
__global__ __launch_bounds__(NW * 64) void k_gemm_skinny(const IgemmParams p)
{
    static_assert(MT == 1 || MT == 2, "row tiles per block");
    static_assert(NT == 1 || (NT == 2 && MT == 2), "column tiles per block");
    // output tile t (row tile t % MT, column tile t / MT) is finished by wave t % NW: TPW tiles per wave.  With NW = 2 and
    // four tiles (the 32 x 32 block of a big grid) a wave finishes two -- the K split stays 2-way, so the bits of a row do
    // not depend on how many rows the launch has (a row of a 640-row ragged decode batch == the same row in a 64-row batch)
    constexpr int TPW = (MT * NT + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) float red[NW][MT * NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // (an XCD-contiguous remap of the column tiles -- 1/8 of W per L2 -- was measured: no change at 64 or 256 rows)
    const int n0 = blockIdx.x * 16 * NT, m0 = blockIdx.y * 16 * MT;
    const int M = p.nbatch * p.T_out;
    const int nk = (p.K + 31) / 32;
    const int per = (nk + NW - 1) / NW;
    const int kt0 = wid * per, kt1 = min(nk, kt0 + per);
    const int nrow = n0 + fr;
    const bool wok = nrow < p.N;
    const uint16_t *wrow = p.w + (int64_t)(wok ? nrow : 0) * p.K + fg * 8;
    const bool wok1 = NT > 1 && nrow + 16 < p.N;
    const uint16_t *wrow1 = p.w + (int64_t)(wok1 ? nrow + 16 : 0) * p.K + fg * 8;
    // row tile 0 (and 1 when MT == 2: the weight fragment is fetched once for both); named variables, not arrays
    const bool xok0 = m0 + fr < M, xok1 = MT > 1 && m0 + 16 + fr < M;
    const uint16_t *xrow, *xrow1;
    {
        const int mm = xok0 ? m0 + fr : 0;
        const int bb = mm / p.T_out, tt = mm - bb * p.T_out;
        xrow = p.x + (int64_t)bb * p.x_bstride + (int64_t)tt * p.lda + fg * 8;
        const int mm1 = xok1 ? m0 + 16 + fr : 0;
        const int bb1 = mm1 / p.T_out, tt1 = mm1 - bb1 * p.T_out;
        xrow1 = p.x + (int64_t)bb1 * p.x_bstride + (int64_t)tt1 * p.lda + fg * 8;
    }
    // LayerNorm folding (ifh_conv_desc.aln_* / rln_*): row statistics are two 64-bit fixed-point sums per row
    // ([rows][2] int64, scale 2^16) that producers build with integer atomics -- integer addition commutes,
    // so unlike float atomics the result is bit-reproducible.  One 16-byte load per lane, issued before the
    // weight stream and consumed in the epilogue.
    // Epilogue operands of the LN mode are requested here as well (ahead of the weight stream): the
    // decoder step is a chain of ~50 of these launches, each only a few microseconds long, and a second
    // dependent round trip after the K loop is a visible fraction of it.
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    int em[TPW], enb[TPW], edyn[TPW];
    bool eown[TPW], exok[TPW];
    longlong2 st_a[TPW], st_r[TPW];
    float4 pc1[TPW], pbias[TPW], pgam[TPW], pbeta[TPW];
    uint2 presid[TPW];
#pragma unroll
    for (int e = 0; e < TPW; e++) {
        const int t = wid + e * NW;
        eown[e] = t < MT * NT;
        em[e] = m0 + 16 * (eown[e] ? t % MT : 0) + fr;
        enb[e] = n0 + 16 * (eown[e] ? t / MT : 0);            // first column of this tile
        exok[e] = eown[e] && em[e] < M;
        edyn[e] = dyn_value(p, exok[e] ? em[e] : 0);          // per output row when dyn_stride = 1 (ragged decode positions)
        st_a[e] = make_longlong2(0, 0);
        st_r[e] = make_longlong2(0, 0);
        pc1[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        pbias[e] = pc1[e]; pgam[e] = pc1[e]; pbeta[e] = pc1[e];
        presid[e] = make_uint2(0, 0);
        if (exok[e]) {
            if (p.aln_stats) st_a[e] = reinterpret_cast<const longlong2 *>(p.aln_stats)[em[e]];
            if (p.rln_stats) st_r[e] = reinterpret_cast<const longlong2 *>(p.rln_stats)[em[e]];
        }
        if (ln_mode && exok[e] && enb[e] + 4 * fg < p.N) {
            const int n = enb[e] + 4 * fg;
            if (p.aln_stats && !p.ln_rms) pc1[e] = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
            if (p.bias) pbias[e] = *reinterpret_cast<const float4 *>(p.bias + n);
            if (p.resid) {
                presid[e] = *reinterpret_cast<const uint2 *>(p.resid + epi_row(p, em[e], n, edyn[e]).rbase + n);
                if (p.rln_stats) {
                    pgam[e] = *reinterpret_cast<const float4 *>(p.rln_gamma + n);
                    pbeta[e] = *reinterpret_cast<const float4 *>(p.rln_beta + n);
                }
            }
        }
    }
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc, acc3 = acc;
    for (int kt = kt0; kt < kt1; kt += U) {
        uint4 wv[U], xv[U], xw[MT > 1 ? U : 1], wv1[NT > 1 ? U : 1];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = kt + u;
            const bool kok = kk < kt1 && (kk * 32 + fg * 8) < p.K;      // K % 8 == 0
            wv[u] = make_uint4(0, 0, 0, 0);
            xv[u] = make_uint4(0, 0, 0, 0);
            if (wok && kok) wv[u] = *reinterpret_cast<const uint4 *>(wrow + kk * 32);
            if (xok0 && kok) xv[u] = *reinterpret_cast<const uint4 *>(xrow + kk * 32);
            if (MT > 1) {
                xw[u] = make_uint4(0, 0, 0, 0);
                if (xok1 && kok) xw[u] = *reinterpret_cast<const uint4 *>(xrow1 + kk * 32);
            }
            if (NT > 1) {
                wv1[u] = make_uint4(0, 0, 0, 0);
                if (wok1 && kok) wv1[u] = *reinterpret_cast<const uint4 *>(wrow1 + kk * 32);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv[u]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, xv[u]), acc, 0, 0, 0);
            if (MT > 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, xw[u]), acc1, 0, 0, 0);
            if (NT > 1) {
                const bf16x8_t wf1 = __builtin_bit_cast(bf16x8_t, wv1[u]);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, __builtin_bit_cast(bf16x8_t, xv[u]), acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1, __builtin_bit_cast(bf16x8_t, xw[u]), acc3, 0, 0, 0);
            }
        }
    }
    *reinterpret_cast<f32x4 *>(&red[wid][0][lane][0]) = acc;
    if (MT > 1) *reinterpret_cast<f32x4 *>(&red[wid][MT - 1][lane][0]) = acc1;
    if (NT > 1) {
        *reinterpret_cast<f32x4 *>(&red[wid][MT * (NT - 1)][lane][0]) = acc2;
        *reinterpret_cast<f32x4 *>(&red[wid][MT * NT - 1][lane][0]) = acc3;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < TPW; e++) {
        if (!eown[e]) continue;
        const int tile = wid + e * NW;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; w++) s += *reinterpret_cast<const f32x4 *>(&red[w][tile][lane][0]);
        const int n = enb[e] + 4 * fg;
        if (ln_mode) {
            // LayerNorm folded around the GEMM (host guarantees the vector epilogue conditions, N % 16 == 0)
            ln_epi4(p, em[e], n, exok[e], s, edyn[e], ln_row(p, st_a[e], st_r[e]), pc1[e], pbias[e], pgam[e], pbeta[e], presid[e], fg);
        } else if (exok[e] && n < p.N) {
            if (p.fast_epi)
                igemm_store4<true>(p, em[e], n, s, edyn[e]);
            else
                igemm_store4<false>(p, em[e], n, s, edyn[e]);
        }
    }
}

// ---- LDS-tiled GEMM for decode steps of a few hundred rows (continuous batching: 192..1024 rows per step, the 640 rows of a
// 5-beam search).  k_gemm_skinny re-streams the weights for every 32 rows from L2 -- 16 FLOP per L2 byte, fine while the step is
// launch-bound at <= 128 rows, 20-40 us per launch at 512.  Here a 64 x BN output tile stages its operands through LDS, but a
// decode-step GEMM is a LATENCY problem (K = 512..3072: a handful of microseconds of work), so the staging is built around
// memory round trips, not around MFMA rate: K is walked in chunks of KC = 256 (8 k-steps of 32), all 16-byte loads of a
// chunk are in flight at once (into registers, one chunk ahead of the one being multiplied), and a chunk costs two barriers.
// K = 768 is three round trips instead of the 24 of a 32-wide double-buffered loop (the first version of this kernel: 23 us per
// launch whatever the row count).  The ARITHMETIC is the streaming kernel's: K is accumulated in `ksplit` chains over the same
// contiguous k ranges the streaming kernel's waves take (2 below K = 2048, 4 from there), the chains are added in wave order,
// and the epilogue is the shared ln_epi4 / igemm_store4 -- so a row's bits are the same whichever kernel the row count selects
// (tests/test_nn_gpu.py::test_gemm_dec_is_bit_identical_to_skinny).  taps == 1, K % 32 == 0, N % 16 == 0 in the LN modes.
// 16 bytes as a native vector: a uint4 (a struct) copied global -> array -> LDS stays two memcpys through a stack slot that the
// compiler does not promote (seen as scratch stores behind an s_waitcnt right after the loads)
__device__ __forceinline__ uint4 ld_u32x4(const uint16_t *ptr)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *reinterpret_cast<const u32x4 *>(ptr);
    return make_uint4(v.x, v.y, v.z, v.w);
}

#ifdef IFH_DEC_PROF
__device__ unsigned long long g_dec_prof[8];
#define DEC_STAMP(I)                                                                   \
    {                                                                                  \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                  \
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&g_dec_prof[I], now_ - tprev_); \
        tprev_ = now_;                                                                 \
    }
#else
#define DEC_STAMP(I) {}
#endif
// BM = 32 (round 3): a launch of fewer workgroups than CUs is one workgroup per CU whose four waves move in lock-step through
// issue / wait / LDS store / fragment reads (profiles/NOTES.md): half the rows per workgroup = twice the workgroups, each phase 2/3
// as long.
// SPLITZ (round 4; LLM down projection, K = 8960 at <= 64 rows): blockIdx.z = one accumulation chain -- the k range that wave z of
// the streaming kernel walks -- and instead of the epilogue the workgroup leaves its f32 partial tile in `ws` [chains][M][N];
// k_splitk_finish adds the chains in chain order and runs the shared epilogue: the streaming kernel's bits from 4 x the workgroups.
template <int BN, int BM = 64, bool SPLITZ = false>
__global__ __launch_bounds__(256, 2) void k_gemm_dec(const IgemmParams p, const int ksplit, float *__restrict__ ws = nullptr)
{
#ifdef IFH_DEC_PROF
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
#endif
    constexpr int KC = 256, LDK = KC + 8;        // row stride 528 B: the 16 rows of a fragment read fall on 16 distinct 16-byte slots
    constexpr int WGM = 2, WGN = 2;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int AV = BM * (KC / 8) / 256, BV = BN * (KC / 8) / 256;      // 16-byte vectors per thread per chunk
    static_assert((BN == 32 || BN == 64) && (BM == 64 || BM == 32), "tile");
    __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDK];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BN * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WGM, wn = wid / WGM;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int M = p.nbatch * p.T_out;
    // thread -> (row, 16-byte column) of the operand chunks: 32 consecutive threads read one row's 512 contiguous bytes
    const int lrow = tid >> 5, lcol = (tid & 31) * 8;
    const uint16_t *arow[AV], *brow[BV];
    bool aval[AV], bval[BV];
    // rows / columns beyond the edge are clamped to the last valid one (loaded, multiplied, never stored) and, when K is a whole
    // number of chunks (every hot shape: 256, 512, 768, 1280, 2048, 3072), nothing is predicated: as one exec branch + range test
    // + 64-bit address per 16-byte load the first chunk took 2 000 shader clocks (1 us) to ISSUE (tools/probe_dec_phases.py)
    const bool t1 = p.T_out == 1;
#pragma unroll
    for (int i = 0; i < AV; i++) {
        const int m = m0 + lrow + 8 * i;
        aval[i] = m < M;
        const int mm = aval[i] ? m : M - 1;
        const int b = t1 ? mm : mm / p.T_out, t = mm - b * p.T_out;
        arow[i] = p.x + (int64_t)b * p.x_bstride + (int64_t)t * p.lda + lcol;
    }
#pragma unroll
    for (int i = 0; i < BV; i++) {
        const int n = n0 + lrow + 8 * i;
        bval[i] = n < p.N;
        brow[i] = p.w + (int64_t)(bval[i] ? n : p.N - 1) * p.K + lcol;
    }
    // the k range of this workgroup: all of K, or (SPLITZ) chain blockIdx.z of ksplit
    const int nk_all = p.K / 32;
    const int per_z = (nk_all + ksplit - 1) / ksplit;
    const int kbeg = SPLITZ ? (int)blockIdx.z * per_z * 32 : 0;
    const int kend = SPLITZ ? min(p.K, kbeg + per_z * 32) : p.K;
    const bool whole = (kend - kbeg) % KC == 0;
    uint4 ra[AV], rb[BV];
#define IFH_DEC_LOAD(K0)                                                                                       \
    if (whole) {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < AV; i++) ra[i] = ld_u32x4(arow[i] + (K0));                       \
        _Pragma("unroll") for (int i = 0; i < BV; i++) rb[i] = ld_u32x4(brow[i] + (K0));                       \
    } else {                                                                                                   \
        const bool kin_ = (K0) + lcol < kend;                                                                  \
        _Pragma("unroll") for (int i = 0; i < AV; i++)                                                         \
            ra[i] = kin_ ? ld_u32x4(arow[i] + (K0)) : make_uint4(0, 0, 0, 0);                                  \
        _Pragma("unroll") for (int i = 0; i < BV; i++)                                                         \
            rb[i] = kin_ ? ld_u32x4(brow[i] + (K0)) : make_uint4(0, 0, 0, 0);                                  \
    }
#define IFH_DEC_STORE()                                                                                        \
    {                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < AV; i++)                                                         \
            *reinterpret_cast<uint4 *>(&As[(lrow + 8 * i) * LDK + lcol]) = ra[i];                              \
        _Pragma("unroll") for (int i = 0; i < BV; i++)                                                         \
            *reinterpret_cast<uint4 *>(&Bs[(lrow + 8 * i) * LDK + lcol]) = rb[i];                              \
    }
    DEC_STAMP(0)
    IFH_DEC_LOAD(kbeg)
    DEC_STAMP(1)
    // epilogue operands, requested behind the first chunk (as the streaming kernel requests them ahead of its K loop)
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    int em[MT], edyn[MT];
    bool exok[MT];
    longlong2 st_a[MT], st_r[MT];
#pragma unroll
    for (int j = 0; j < MT; j++) {
        em[j] = m0 + wm * WM + j * 16 + fr;
        exok[j] = em[j] < M;
        edyn[j] = dyn_value(p, exok[j] ? em[j] : 0);
        st_a[j] = make_longlong2(0, 0);
        st_r[j] = make_longlong2(0, 0);
        if (exok[j]) {
            if (p.aln_stats) st_a[j] = reinterpret_cast<const longlong2 *>(p.aln_stats)[em[j]];
            if (p.rln_stats) st_r[j] = reinterpret_cast<const longlong2 *>(p.rln_stats)[em[j]];
        }
    }
    float4 pc1[NT], pbias[NT], pgam[NT], pbeta[NT];
    uint2 presid[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
        pc1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        pbias[i] = pc1[i]; pgam[i] = pc1[i]; pbeta[i] = pc1[i];
        if (ln_mode && n < p.N) {
            if (p.aln_stats && !p.ln_rms) pc1[i] = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
            if (p.bias) pbias[i] = *reinterpret_cast<const float4 *>(p.bias + n);
            if (p.resid && p.rln_stats) {
                pgam[i] = *reinterpret_cast<const float4 *>(p.rln_gamma + n);
                pbeta[i] = *reinterpret_cast<const float4 *>(p.rln_beta + n);
            }
        }
#pragma unroll
        for (int j = 0; j < MT; j++) {
            presid[i][j] = make_uint2(0, 0);
            if (ln_mode && p.resid && exok[j] && n < p.N)
                presid[i][j] = *reinterpret_cast<const uint2 *>(p.resid + epi_row(p, em[j], n, edyn[j]).rbase + n);
        }
    }
    f32x4 acc[NT][MT], sum[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) {
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            sum[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    const int nk = (kend - kbeg) / 32;
    const int per = SPLITZ ? nk : (nk + ksplit - 1) / ksplit;      // k-steps per accumulation chain (= per wave of the streaming kernel)
    int next_flush = per;
    const int nchunk = (kend - kbeg + KC - 1) / KC;
    DEC_STAMP(2)
    for (int c = 0; c < nchunk; c++) {
        IFH_DEC_STORE();
        __syncthreads();
        if (c == 0) {
            DEC_STAMP(3)
        }
        if (c + 1 < nchunk) {                                   // the next chunk's round trip overlaps this chunk's MFMAs
            IFH_DEC_LOAD(kbeg + (c + 1) * KC)
        }
        const int ks1 = min(KC / 32, nk - c * (KC / 32));
        for (int ks = 0; ks < ks1; ks++) {
            bf16x8_t fa[NT], fb[MT];
#pragma unroll
            for (int i = 0; i < NT; i++)
                fa[i] = *reinterpret_cast<const bf16x8_t *>(&Bs[(wn * WN + i * 16 + fr) * LDK + ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < MT; j++)
                fb[j] = *reinterpret_cast<const bf16x8_t *>(&As[(wm * WM + j * 16 + fr) * LDK + ks * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int j = 0; j < MT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            const int kt1 = c * (KC / 32) + ks + 1;
            if (kt1 == next_flush || kt1 == nk) {               // end of a chain: add it to the running sum, in chain order
#pragma unroll
                for (int i = 0; i < NT; i++)
#pragma unroll
                    for (int j = 0; j < MT; j++) {
                        sum[i][j] += acc[i][j];
                        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                next_flush += per;
            }
        }
        __syncthreads();
    }
    DEC_STAMP(4)
#undef IFH_DEC_LOAD
#undef IFH_DEC_STORE
    if (SPLITZ) {
#pragma unroll
        for (int j = 0; j < MT; j++)
#pragma unroll
            for (int i = 0; i < NT; i++) {
                const int n = n0 + wn * WN + i * 16 + 4 * fg;
                if (exok[j] && n < p.N)
                    *reinterpret_cast<float4 *>(ws + ((int64_t)blockIdx.z * M + em[j]) * p.N + n) =
                        make_float4(sum[i][j][0], sum[i][j][1], sum[i][j][2], sum[i][j][3]);
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < MT; j++) {
        const LnRow lr = ln_row(p, st_a[j], st_r[j]);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = n0 + wn * WN + i * 16 + 4 * fg;
            if (ln_mode) {
                ln_epi4(p, em[j], n, exok[j], sum[i][j], edyn[j], lr, pc1[i], pbias[i], pgam[i], pbeta[i], presid[i][j], fg);
            } else if (exok[j] && n < p.N) {
                if (p.fast_epi)
                    igemm_store4<true>(p, em[j], n, sum[i][j], edyn[j]);
                else
                    igemm_store4<false>(p, em[j], n, sum[i][j], edyn[j]);
            }
        }
    }
    DEC_STAMP(5)
#ifdef IFH_DEC_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(&g_dec_prof[7], 1ull);
#endif
}

// the chains of a SPLITZ launch added in chain order + the streaming kernel's epilogue: one wave per 16 x 16 output tile, a lane holds
// 4 consecutive columns of one row exactly as in k_gemm_skinny, whose epilogue operands it fetches the same way
__global__ __launch_bounds__(64) void k_splitk_finish(const IgemmParams p, const float *__restrict__ ws, const int nchains)
{
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int M = p.nbatch * p.T_out;
    const int em = m0 + fr, n = n0 + 4 * fg;
    const bool exok = em < M;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (exok && n < p.N)
        for (int z = 0; z < nchains; z++) {
            const float4 v = *reinterpret_cast<const float4 *>(ws + ((int64_t)z * M + em) * p.N + n);
            s += (f32x4){v.x, v.y, v.z, v.w};
        }
    const int edyn = dyn_value(p, exok ? em : 0);
    const bool ln_mode = p.aln_stats || p.rln_stats || p.stats_out;
    if (ln_mode) {
        longlong2 st_a = make_longlong2(0, 0), st_r = make_longlong2(0, 0);
        float4 pc1 = make_float4(0.f, 0.f, 0.f, 0.f), pbias = pc1, pgam = pc1, pbeta = pc1;
        uint2 presid = make_uint2(0, 0);
        if (exok) {
            if (p.aln_stats) st_a = reinterpret_cast<const longlong2 *>(p.aln_stats)[em];
            if (p.rln_stats) st_r = reinterpret_cast<const longlong2 *>(p.rln_stats)[em];
        }
        if (exok && n < p.N) {
            if (p.aln_stats && !p.ln_rms) pc1 = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
            if (p.bias) pbias = *reinterpret_cast<const float4 *>(p.bias + n);
            if (p.resid) {
                presid = *reinterpret_cast<const uint2 *>(p.resid + epi_row(p, em, n, edyn).rbase + n);
                if (p.rln_stats) {
                    pgam = *reinterpret_cast<const float4 *>(p.rln_gamma + n);
                    pbeta = *reinterpret_cast<const float4 *>(p.rln_beta + n);
                }
            }
        }
        ln_epi4(p, em, n, exok, s, edyn, ln_row(p, st_a, st_r), pc1, pbias, pgam, pbeta, presid, fg);
    } else if (exok && n < p.N) {
        if (p.fast_epi)
            igemm_store4<true>(p, em, n, s, edyn);
        else
            igemm_store4<false>(p, em, n, s, edyn);
    }
}

// ---- weight-streaming GEMM for decode steps of LLM-sized layers (M <= 64 rows, N x K in the tens of MB) ----
// k_gemm_skinny gives every 16 x 16 output tile its own block: at M = 64 a weight fragment is fetched by four blocks and an
// activation fragment by N/16 of them, fine for 768-wide speech decoders whose weights live in L2, ~1 TB/s of weight
// streaming on an 8960 x 1536 layer.  Here one block owns NTB column tiles x ALL (up to 4) row tiles, so every weight
// byte crosses HBM -> CU exactly once; its four waves split K into contiguous quarters (sequential addresses per
// wave), keep U k-steps of 16-byte fragment loads in flight, and sum their partial tiles through LDS.
template <int NTB, int U>
__global__ __launch_bounds__(256) void k_gemm_m64(const IgemmParams p)
{
    constexpr int NW = 4, MT = 4;
    __shared__ __attribute__((aligned(16))) float red[NW][MT * NTB][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16 * NTB;
    const int M = p.nbatch * p.T_out;
    const int nk = (p.K + 31) / 32;
    const int per = (nk + NW - 1) / NW;
    const int kt0 = wid * per, kt1 = min(nk, kt0 + per);
    const uint16_t *wrow[NTB];
    bool wok[NTB];
#pragma unroll
    for (int t = 0; t < NTB; t++) {
        const int nrow = n0 + 16 * t + fr;
        wok[t] = nrow < p.N;
        wrow[t] = p.w + (int64_t)(wok[t] ? nrow : 0) * p.K + fg * 8;
    }
    const uint16_t *xrow[MT];
    bool xok[MT];
#pragma unroll
    for (int r = 0; r < MT; r++) {
        const int mm0 = 16 * r + fr;
        xok[r] = mm0 < M;
        const int mm = xok[r] ? mm0 : 0;
        const int bb = mm / p.T_out, tt = mm - bb * p.T_out;
        xrow[r] = p.x + (int64_t)bb * p.x_bstride + (int64_t)tt * p.lda + fg * 8;
    }
    f32x4 acc[NTB][MT];
#pragma unroll
    for (int t = 0; t < NTB; t++)
#pragma unroll
        for (int r = 0; r < MT; r++) acc[t][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = kt0; kt < kt1; kt += U) {
        uint4 wv[U][NTB], xv[U][MT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = kt + u;
            const bool kok = kk < kt1 && (kk * 32 + fg * 8) < p.K;      // K % 8 == 0
#pragma unroll
            for (int t = 0; t < NTB; t++) {
                wv[u][t] = make_uint4(0, 0, 0, 0);
                if (wok[t] && kok) wv[u][t] = *reinterpret_cast<const uint4 *>(wrow[t] + kk * 32);
            }
#pragma unroll
            for (int r = 0; r < MT; r++) {
                xv[u][r] = make_uint4(0, 0, 0, 0);
                if (xok[r] && kok) xv[u][r] = *reinterpret_cast<const uint4 *>(xrow[r] + kk * 32);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < NTB; t++) {
                const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv[u][t]);
#pragma unroll
                for (int r = 0; r < MT; r++)
                    acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8_t, xv[u][r]), acc[t][r], 0, 0, 0);
            }
    }
#pragma unroll
    for (int t = 0; t < NTB; t++)
#pragma unroll
        for (int r = 0; r < MT; r++) *reinterpret_cast<f32x4 *>(&red[wid][t * MT + r][lane][0]) = acc[t][r];
    __syncthreads();
    // wave w finishes tiles w, w + 4, ...: column tile = tile / MT, row tile = tile % MT = w -- the rows 16 w .. 16 w + 15 over all
    // of the block's columns, so a row's largest value inside the block is found by its four lanes alone (arg-max keys)
    unsigned long long best = 0;
#pragma unroll
    for (int i = 0; i < NTB; i++) {
        const int tile = wid + NW * i;
        f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; w++) s += *reinterpret_cast<const f32x4 *>(&red[w][tile][lane][0]);
        const int m = 16 * (tile % MT) + fr, n = n0 + 16 * (tile / MT) + 4 * fg;
        if (p.amax_keys && m < M) {
            // (the launcher admits no bias / activation / residual / scale here, and of the folded normalisations only the RMS form: s is
            // what igemm_store4 stores up to a positive factor per row, so its order is the stored values' order)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (n + e < p.N) {
                    const uint32_t u = __float_as_uint(s[e]);
                    const uint32_t ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                    const unsigned long long key = ((unsigned long long)ord << 32) | (uint32_t)(0xffffffffu - (uint32_t)(n + e));
                    best = key > best ? key : best;
                }
            }
        }
        if (m < M && n < p.N) {
            const int dynv = dyn_value(p, m);
            if (p.aln_stats) {
                // normalisation of the A rows folded in (ifh_conv_desc.aln_*): the producer left (sum, sum of squares)
                const longlong2 st = reinterpret_cast<const longlong2 *>(p.aln_stats)[m];
                const float fx = (1.0f / 65536.0f) / (float)p.ln_dim;
                const float mean = p.ln_rms ? 0.0f : (float)st.x * fx;
                const float rstd = rsqrtf(fmaxf((float)st.y * fx - mean * mean, 0.0f) + p.ln_eps);
                float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!p.ln_rms) c1 = *reinterpret_cast<const float4 *>(p.aln_c1 + n);
                s[0] = rstd * (s[0] - mean * c1.x);
                s[1] = rstd * (s[1] - mean * c1.y);
                s[2] = rstd * (s[2] - mean * c1.z);
                s[3] = rstd * (s[3] - mean * c1.w);
            }
            if (p.act == ACT_SILU_GLU) {
                // interleaved (gate, up) weight rows: this lane holds two pairs -> two outputs of the half-width result
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n);
                    s[0] += bv.x; s[1] += bv.y; s[2] += bv.z; s[3] += bv.w;
                }
                const float o0 = s[0] / (1.0f + __expf(-s[0])) * s[1], o1 = s[2] / (1.0f + __expf(-s[2])) * s[3];
                *reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(p.out) + (int64_t)m * p.ldc + (n >> 1)) =
                    f32x2_to_bf16x2(o0, o1);
            } else if (p.fast_epi)
                igemm_store4<true>(p, m, n, s, dynv);
            else
                igemm_store4<false>(p, m, n, s, dynv);
        }
    }
    if (p.amax_keys) {
        unsigned long long o = __shfl_xor(best, 16, 64);
        best = o > best ? o : best;
        o = __shfl_xor(best, 32, 64);
        best = o > best ? o : best;
        if (fg == 0 && 16 * wid + fr < M && best) atomicMax(p.amax_keys + 16 * wid + fr, best);
    }
}

__global__ void k_argmax_keys_finish(unsigned long long *__restrict__ keys, int32_t *__restrict__ tokens, int n)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) {
        const unsigned long long k = keys[i];
        tokens[i] = k ? (int32_t)(0xffffffffu - (uint32_t)k) : 0;
        keys[i] = 0;
    }
}

#ifndef IFH_IGEMM_MINB
#define IFH_IGEMM_MINB 3
#endif
template <int BM, int BN, int WGM, bool PRE, bool FAST, int NWV = 4, int KT = 32, bool PLAIN = false, bool ALN = false>
// (three 4-wave workgroups per CU: left alone the compiler takes 116 VGPRs + 64 AGPRs = two per CU, and at K = 512 the k-loop is a
// chain of 16 load -> barrier round trips whose latency only other resident workgroups cover)
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? IFH_IGEMM_MINB : 2) void k_igemm(const IgemmParams p)
{
    constexpr int NTH = 64 * NWV;              // threads: 4 waves, or 8 for the 256 x 128 tile
    constexpr int WGN = NWV / WGM;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int MT = WM / 16, NT = WN / 16;
    // KT = 32: 64-byte row pieces, double-buffered K tiles, one barrier per k-step.  KT = 64 (round 3): a row piece is a whole
    // 128-byte line -- with 64-byte pieces every line of A and W is requested from L2 twice, a k-step apart, and the vector L1 does
    // not keep it (3 x 16 KB of tiles per CU pass in between): TCP -> TCC read requests per launch were 2 x bytes / 128 at the
    // Whisper-base encoder shapes, 43 % of the L2s' request rate -- one LDS buffer (two barriers per 64 of K, the same per K).
    constexpr int KV = KT / 8, NBUF = KT == 32 ? 2 : 1;
    constexpr int LD = KT + 8;  // 8 pad (80- / 144-byte rows)
    constexpr int AV = BM * KV / NTH, BV = (BN * KV + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) uint16_t As2[NBUF][BM * LD];
    __shared__ __attribute__((aligned(16))) uint16_t Bs2[NBUF][BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid % WGM, wn = wid / WGM;
    // Workgroup -> tile, XCD-aware and column-tile-fastest: consecutive workgroup ids go to the 8 XCDs in turn, each with its own
    // L2, so XCD x takes the contiguous range [x * per, (x + 1) * per) of the tile list and walks it with the column tile running
    // fastest -- the ntn workgroups that share a row tile (all of A's 128 x K rows) run together on ONE L2, which fetches that tile
    // once.  With the row tile on blockIdx.x (rounds 1-2) the same A rows came back from HBM once per column tile: at Whisper-base's
    // encoder shapes (192 000 rows, K = 512, N = 512..2048) that was 4-16 passes over a 197 MB activation matrix per GEMM.
    int m0, n0;
    {
        const int ntn = (p.N + BN - 1) / BN, ntm = (p.nbatch * p.T_out + BM - 1) / BM;
        const int per = (ntm * ntn + 7) >> 3;
        const int t = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= per || t >= ntm * ntn) return;
        m0 = (t / ntn) * BM;
        n0 = (t % ntn) * BN;
    }
    const int M = p.nbatch * p.T_out;
    const bool uniform_tap = (p.Cin % KT) == 0;

    // per-thread activation-row descriptors (fixed over the K loop)
    const uint16_t *arow[AV];
    int abase_t[AV];
    bool avalid[AV];
#pragma unroll
    for (int i = 0; i < AV; i++) {
        const int v = tid + NTH * i;
        const int m = m0 + v / KV;
        avalid[i] = m < M;
        const int mm = avalid[i] ? m : 0;
        const int b = mm / p.T_out, t = mm - b * p.T_out;
        arow[i] = p.x + (int64_t)b * p.x_bstride;
        abase_t[i] = t * p.stride - p.pad;
    }
    // PLAIN (round 3): a matrix product (one tap, stride 1, no padding, K a multiple of KT).  The general tile load below is ~230
    // instructions per k-step around 16 MFMAs -- tap / channel arithmetic, four range tests and an exec branch per 16-byte load,
    // kernel arguments re-read from memory -- and with three waves per SIMD the issue slots, not memory, bound the k-loop (the
    // Whisper-base encoder GEMMs ran at 0.4-0.5 PFLOP/s with 92 % L2 hits).  Here a thread's loads are pointer + k: rows and
    // columns beyond the edge are clamped to the last valid one (loaded, multiplied, never stored), so nothing is predicated.
    const uint16_t *aptr[AV], *wptr[BV];
    if (PLAIN) {
#pragma unroll
        for (int i = 0; i < AV; i++) {
            const int v = tid + NTH * i;
            const int mm = min(m0 + v / KV, M - 1);
            const int b = mm / p.T_out, t = mm - b * p.T_out;
            aptr[i] = p.x + (int64_t)b * p.x_bstride + (int64_t)t * p.lda + (v % KV) * 8;
        }
#pragma unroll
        for (int i = 0; i < BV; i++) {
            const int v = min(tid + NTH * i, BN * KV - 1);
            wptr[i] = p.w + (int64_t)min(n0 + v / KV, p.N - 1) * p.K + (v % KV) * 8;
        }
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NOTE: plain macros, not lambdas: arrays captured by reference in a lambda were placed in
    // scratch memory by hipcc (272 B/lane of spills in the 128x128 tile, ~2x slower).
    uint4 ra[AV], rb[BV];
#define IFH_LOAD_TILES(K0)                                                                               \
    if (PLAIN) {                                                                                         \
        const int k0_ = (K0);                                                                            \
        _Pragma("unroll") for (int i = 0; i < AV; i++) ra[i] = ld_u32x4(aptr[i] + k0_);                  \
        _Pragma("unroll") for (int i = 0; i < BV; i++) rb[i] = ld_u32x4(wptr[i] + k0_);                  \
    } else {                                                                                             \
        const int k0_ = (K0);                                                                            \
        int tap_u = 0, ci_u = 0;                                                                         \
        if (uniform_tap) {                                                                               \
            tap_u = k0_ / p.Cin;                                                                         \
            ci_u = k0_ - tap_u * p.Cin;                                                                  \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < AV; i++)                                                   \
        {                                                                                                \
            const int v = tid + NTH * i;                                                                 \
            const int k = k0_ + (v % KV) * 8;                                                            \
            int tap, ci;                                                                                 \
            if (uniform_tap) {                                                                           \
                tap = tap_u;                                                                             \
                ci = ci_u + (v % KV) * 8;                                                                \
            } else {                                                                                     \
                tap = k / p.Cin;                                                                         \
                ci = k - tap * p.Cin;                                                                    \
            }                                                                                            \
            const int tin = abase_t[i] + tap * p.dil;                                                    \
            uint4 val = make_uint4(0, 0, 0, 0);                                                          \
            if (avalid[i] && k < kend && tin >= 0 && tin < p.T_in)                                       \
                val = *reinterpret_cast<const uint4 *>(arow[i] + (int64_t)tin * p.lda + ci);             \
            ra[i] = val;                                                                                 \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < BV; i++)                                                   \
        {                                                                                                \
            const int v = tid + NTH * i;                                                                 \
            const int n = n0 + v / KV;                                                                   \
            const int k = k0_ + (v % KV) * 8;                                                            \
            uint4 val = make_uint4(0, 0, 0, 0);                                                          \
            if (v < BN * KV && n < p.N && k < kend) val = *reinterpret_cast<const uint4 *>(p.w + (int64_t)n * p.K + k); \
            rb[i] = val;                                                                                 \
        }                                                                                                \
    }
    // the fused input LeakyReLU is applied at LDS-store time, after the previous tile's MFMAs, so the
    // global loads of this tile stayed in flight behind them (zeros stay zeros)
#define IFH_STORE_TILES(BUF)                                                                             \
    {                                                                                                    \
        uint16_t *As_ = As2[BUF], *Bs_ = Bs2[BUF];                                                       \
        _Pragma("unroll") for (int i = 0; i < AV; i++)                                                   \
        {                                                                                                \
            const int v = tid + NTH * i;                                                                 \
            *reinterpret_cast<uint4 *>(&As_[(v / KV) * LD + (v % KV) * 8]) = PRE ? lrelu8(ra[i], p.pre_slope) : ra[i]; \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < BV; i++)                                                   \
        {                                                                                                \
            const int v = tid + NTH * i;                                                                 \
            if ((BN * KV) % NTH == 0 || v < BN * KV) *reinterpret_cast<uint4 *>(&Bs_[(v / KV) * LD + (v % KV) * 8]) = rb[i]; \
        }                                                                                                \
    }

    // fused transposed convolution: this column tile's structurally zero tap is not multiplied (adding 0 * x leaves every
    // accumulator as it is, so the bits do not change)
    int kbeg = 0, kend = p.K;
    if (p.zt_cout) {
        const int r_lo = n0 / p.zt_cout, r_hi = (min(n0 + BN, p.N) - 1) / p.zt_cout;
        if (r_hi < 2) kend = 2 * p.Cin;
        else if (r_lo >= 2) kbeg = p.Cin;
    }
    const int nk = (kend - kbeg + KT - 1) / KT;
    IFH_LOAD_TILES(kbeg)
    const int fr = lane & 15, fg = lane >> 4;
    // bias 4-vectors of this lane's column groups, requested ahead of the K loop (vector epilogue only)
    float4 bpre[NT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int n = n0 + wn * WN + i * 16 + 4 * fg;
        bpre[i] = (FAST && p.bias && n < p.N) ? *reinterpret_cast<const float4 *>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    IFH_STORE_TILES(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const uint16_t *As = As2[NBUF == 2 ? (kt & 1) : 0], *Bs = Bs2[NBUF == 2 ? (kt & 1) : 0];
        if (kt + 1 < nk) {
            IFH_LOAD_TILES(kbeg + (kt + 1) * KT)
        }
#pragma unroll
        for (int ks = 0; ks < KT / 32; ks++) {
            bf16x8_t fa[NT], fb[MT];
#pragma unroll
            for (int i = 0; i < NT; i++)
                fa[i] = *reinterpret_cast<const bf16x8_t *>(&Bs[(wn * WN + i * 16 + fr) * LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < MT; j++)
                fb[j] = *reinterpret_cast<const bf16x8_t *>(&As[(wm * WM + j * 16 + fr) * LD + ks * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int j = 0; j < MT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (NBUF == 2) {
            // the other buffer was last read one iteration ago and every wave has passed the barrier since
            if (kt + 1 < nk) IFH_STORE_TILES((kt + 1) & 1);
            __syncthreads();
        } else if (kt + 1 < nk) {
            __syncthreads();                           // every wave has read this tile's fragments
            IFH_STORE_TILES(0);
            __syncthreads();
        }
    }

#undef IFH_LOAD_TILES
#undef IFH_STORE_TILES
    // ---- epilogue: lane holds D[n = 4*fg + r][m = fr] of each 16x16 tile
    // The activation is uniform over the launch: choosing it once, outside the 4 x 4 tile loops, instead of through the runtime
    // switch on each of a thread's 64 outputs (measured on the Whisper-base fc1 shape, 192 000 x 512 x 2048: 1.07 ms without an
    // activation, 1.66 ms with ReLU through the switch, 1.78 ms with the library GELU).
#define IFH_IGEMM_EPI(ACTC)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < MT; j++)                                                                   \
    {                                                                                                                \
        const int m = m0 + wm * WM + j * 16 + fr;                                                                    \
        if (m >= M) continue;                                                                                        \
        const int dynv = dyn_value(p, m);                                                                            \
        float a_mean = 0.0f, a_rstd = 1.0f;                                                                          \
        if (ALN) {      /* LayerNorm of the A rows folded in (ifh_conv_desc.aln_*): ln_row's arithmetic */           \
            const LnRow lr = ln_row(p, reinterpret_cast<const longlong2 *>(p.aln_stats)[m], make_longlong2(0, 0));   \
            a_mean = lr.a_mean;                                                                                      \
            a_rstd = lr.a_rstd;                                                                                      \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NT; i++)                                                               \
        {                                                                                                            \
            const int n = n0 + wn * WN + i * 16 + 4 * fg;                                                            \
            if (n >= p.N) continue;                                                                                  \
            if (ALN) {                                                                                               \
                float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
                if (!p.ln_rms) c1 = *reinterpret_cast<const float4 *>(p.aln_c1 + n);                                 \
                acc[i][j][0] = a_rstd * (acc[i][j][0] - a_mean * c1.x);                                              \
                acc[i][j][1] = a_rstd * (acc[i][j][1] - a_mean * c1.y);                                              \
                acc[i][j][2] = a_rstd * (acc[i][j][2] - a_mean * c1.z);                                              \
                acc[i][j][3] = a_rstd * (acc[i][j][3] - a_mean * c1.w);                                              \
            }                                                                                                        \
            if (FAST)                                                                                                \
                (void)igemm_store4_fast<false, true, true, ACTC>(p, m, n, acc[i][j], dynv, make_uint2(0, 0), bpre[i]); \
            else                                                                                                     \
                igemm_store4<false>(p, m, n, acc[i][j], dynv);                                                       \
        }                                                                                                            \
    }
    if (!FAST || (p.act != ACT_NONE && p.act != ACT_GELU && p.act != ACT_RELU)) {
        IFH_IGEMM_EPI(-1)
    } else if (p.act == ACT_NONE) {
        IFH_IGEMM_EPI(ACT_NONE)
    } else if (p.act == ACT_GELU) {
        IFH_IGEMM_EPI(ACT_GELU)
    } else {
        IFH_IGEMM_EPI(ACT_RELU)
    }
#undef IFH_IGEMM_EPI
}

// ---- LayerNorm: RPW rows per wave, optional residual add first; D <= 256 * NCH (NCH = 4: 1024; 8: 2048 -- Whisper large-v3's 1280), D % 4 == 0.
// Lane l holds elements 4 l .. 4 l + 3 of every 256-element chunk; every load of a wave's rows goes out before the first use (inside the
// per-lane `if (e < D)` of the first form hipcc put each load in its own exec branch with an s_waitcnt vmcnt(0) behind it: 2-4 serial memory
// latencies per launch, 3.7 TB/s on the encoder's 192 000 x 512 stream against 5.3 for a copy), lanes past D load element 0 and select
// zeros.  RPW = 4 for long streams: the shuffle reductions of the rows interleave and gamma / beta are fetched once per wave.  The order of
// every sum is the same for both RPW (same bits).
template <int NCH, int RPW, bool RESID>
__global__ __launch_bounds__(256) void k_layernorm(const uint16_t *__restrict__ x, const uint16_t *__restrict__ resid,
                                                   const float *__restrict__ gamma, const float *__restrict__ beta,
                                                   uint16_t *__restrict__ out, int rows, int D, float eps)
{
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    const int lane = threadIdx.x & 63;
    if (row0 >= rows) return;
    uint2 xa[RPW][NCH], xc[RPW][NCH];
    float4 g[NCH], bt[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int e = lane * 4 + 256 * i;
        g[i] = *reinterpret_cast<const float4 *>(gamma + (e < D ? e : 0));
        bt[i] = *reinterpret_cast<const float4 *>(beta + (e < D ? e : 0));
    }
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;      // (a clamped row is computed again, not stored)
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const int e = lane * 4 + 256 * i;
            const int64_t off = (int64_t)row * D + (e < D ? e : 0);
            xa[r][i] = *reinterpret_cast<const uint2 *>(x + off);
            if (RESID) xc[r][i] = *reinterpret_cast<const uint2 *>(resid + off);
        }
    }
    // (gamma / beta pinned up here: left alone hipcc sinks their loads behind the reductions, one more exposed latency)
#pragma unroll
    for (int i = 0; i < NCH; i++)
        asm volatile("" : "+v"(g[i].x), "+v"(g[i].y), "+v"(g[i].z), "+v"(g[i].w), "+v"(bt[i].x), "+v"(bt[i].y), "+v"(bt[i].z), "+v"(bt[i].w),
                          "+v"(xa[0][0].x));       // (tied to the first row value: the statement cannot sink below the first sum)
    float v[RPW][NCH][4];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        s[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const bool ok = lane * 4 + 256 * i < D;
            v[r][i][0] = __uint_as_float(xa[r][i].x << 16);
            v[r][i][1] = __uint_as_float(xa[r][i].x & 0xffff0000u);
            v[r][i][2] = __uint_as_float(xa[r][i].y << 16);
            v[r][i][3] = __uint_as_float(xa[r][i].y & 0xffff0000u);
            if (RESID) {
                v[r][i][0] += __uint_as_float(xc[r][i].x << 16);
                v[r][i][1] += __uint_as_float(xc[r][i].x & 0xffff0000u);
                v[r][i][2] += __uint_as_float(xc[r][i].y << 16);
                v[r][i][3] += __uint_as_float(xc[r][i].y & 0xffff0000u);
            }
            const float t = s[r] + ((v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]));
            s[r] = ok ? t : s[r];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; r++) s[r] += __shfl_xor(s[r], o, 64);
    float mean[RPW], q[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) {
        mean[r] = s[r] / (float)D;
        q[r] = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH; i++) {
            const bool ok = lane * 4 + 256 * i < D;
            float t = q[r];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float d = v[r][i][c] - mean[r];
                t += d * d;
            }
            q[r] = ok ? t : q[r];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < RPW; r++) q[r] += __shfl_xor(q[r], o, 64);
    float rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; r++) rstd[r] = rsqrtf(q[r] / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NCH; i++) {
        const int e = lane * 4 + 256 * i;
        const bool ok = e < D;
#pragma unroll
        for (int r = 0; r < RPW; r++) {
            const float o0 = (v[r][i][0] - mean[r]) * rstd[r] * g[i].x + bt[i].x, o1 = (v[r][i][1] - mean[r]) * rstd[r] * g[i].y + bt[i].y;
            const float o2 = (v[r][i][2] - mean[r]) * rstd[r] * g[i].z + bt[i].z, o3 = (v[r][i][3] - mean[r]) * rstd[r] * g[i].w + bt[i].w;
            uint2 pk;
            pk.x = f32x2_to_bf16x2(o0, o1);
            pk.y = f32x2_to_bf16x2(o2, o3);
            if (ok && row0 + r < rows) *reinterpret_cast<uint2 *>(out + (int64_t)(row0 + r) * D + e) = pk;
        }
    }
}

// ---- transpose the last two dims, cast to bf16: in [B][R][C] (f32 or bf16) -> out [B][C][R]
template <bool IN_F32>
__global__ __launch_bounds__(256) void k_transpose(const void *__restrict__ in, uint16_t *__restrict__ out, int R, int C)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        float v = 0.0f;
        if (r < R && c < C) {
            const int64_t idx = ((int64_t)b * R + r) * C + c;
            v = IN_F32 ? reinterpret_cast<const float *>(in)[idx] : bf16_to_f32(reinterpret_cast<const uint16_t *>(in)[idx]);
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < C) out[((int64_t)b * C + c) * R + r] = f32_to_bf16(tile[tx][k]);
    }
}

template <int BM, int BN, int WGM, int NWV = 4, int KT = 32>
static int launch_igemm(const IgemmParams &p, bool pre, hipStream_t st)
{
    const int M = p.nbatch * p.T_out;
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    dim3 grid((unsigned)(((tiles + 7) / 8) * 8));          // 1-D, a multiple of 8: see the tile map at the top of k_igemm
    constexpr bool no_plain = false;     // fixed by measurement (profiles/NOTES.md)
    const bool plain = p.fast_epi && !pre && p.taps == 1 && p.stride == 1 && p.pad == 0 && p.K % KT == 0 && p.T_out <= p.T_in && !p.zt_cout;
    if (p.aln_stats) {                       // the caller (ifh_conv_bf16) has checked `plain` and the tile shape
        if (BM == 128 && BN == 128 && NWV == 4 && KT == 32)
            hipLaunchKernelGGL((k_igemm<BM, BN, WGM, false, true, NWV, KT, true, (BM == 128 && BN == 128 && NWV == 4 && KT == 32)>), grid, dim3(64 * NWV), 0, st, p);
        else {
            return fail(IFH_EINVAL, "k_igemm: the LayerNorm-folded epilogue exists for the 128 x 128 x 4-wave x 32 tile only");
        }
    } else if (plain && !no_plain)
        hipLaunchKernelGGL((k_igemm<BM, BN, WGM, false, true, NWV, KT, true>), grid, dim3(64 * NWV), 0, st, p);
    else if (p.fast_epi) {
        if (pre)
            hipLaunchKernelGGL((k_igemm<BM, BN, WGM, true, true, NWV, KT>), grid, dim3(64 * NWV), 0, st, p);
        else
            hipLaunchKernelGGL((k_igemm<BM, BN, WGM, false, true, NWV, KT>), grid, dim3(64 * NWV), 0, st, p);
    } else {
        if (pre)
            hipLaunchKernelGGL((k_igemm<BM, BN, WGM, true, false, NWV, KT>), grid, dim3(64 * NWV), 0, st, p);
        else
            hipLaunchKernelGGL((k_igemm<BM, BN, WGM, false, false, NWV, KT>), grid, dim3(64 * NWV), 0, st, p);
    }
    return IFH_OK;
}

}  // namespace ifh

using namespace ifh;

static int igemm_aln_rows()
{
    constexpr int v = 256;      // fixed by measurement (profiles/NOTES.md)
    return v;
}

extern "C" int ifh_conv_argmax_supported(int rows, int n, int k)
{
    return rows > 16 && rows <= 64 && n >= 8192 && (int64_t)n * k >= (int64_t)4096 * 1024 && k % 8 == 0;
}

extern "C" int ifh_argmax_keys_finish(void *keys, int32_t *tokens, int n, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(keys && tokens && (((uintptr_t)keys) & 7) == 0);
    hipLaunchKernelGGL(k_argmax_keys_finish, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), (unsigned long long *)keys, tokens, n);
    IFH_LAUNCH_CHECK("argmax_keys_finish");
    return IFH_OK;
}

extern "C" int ifh_conv_bf16(const ifh_conv_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && d->w && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t_out >= 0);
    if (d->nbatch == 0 || d->t_out == 0) return IFH_OK;
    IFH_CHECK_ARG(d->cin > 0 && d->cin % 8 == 0 && d->taps >= 1 && d->stride >= 1 && d->dil >= 1);
    IFH_CHECK_ARG(d->lda % 8 == 0 && d->x_bstride % 8 == 0 && (((uintptr_t)d->x) & 15) == 0 && (((uintptr_t)d->w) & 15) == 0);
    IFH_CHECK_ARG(d->n > 0 && d->t_in > 0 && d->ldc > 0 && d->ostride >= 1 && d->ooff >= 0);
    IFH_CHECK_ARG(d->act >= 0 && d->act <= 6);
    IgemmParams p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.lda = d->lda;
    p.Cin = d->cin;
    p.taps = d->taps;
    p.stride = d->stride;
    p.dil = d->dil;
    p.pad = d->pad;
    p.T_in = d->t_in;
    p.T_out = d->t_out;
    p.nbatch = d->nbatch;
    p.w = (const uint16_t *)d->w;
    p.K = d->taps * d->cin;
    p.N = d->n;
    p.bias = d->bias;
    p.colmask = d->colmask;
    p.pre_slope = d->pre_slope;
    p.act = d->act;
    p.act_slope = d->act_slope;
    p.resid = (const uint16_t *)d->resid;
    p.resid_bstride = d->resid_bstride;
    p.resid_ld = d->resid_ld;
    p.out_scale = d->out_scale;
    p.accumulate = d->accumulate;
    p.out = d->out;
    p.out_f32 = d->out_f32;
    p.out_bstride = d->out_bstride;
    p.ldc = d->ldc;
    p.ostride = d->ostride;
    p.ooff = d->ooff;
    const int esz = d->out_f32 ? 4 : 2;
    p.vec_ok = (d->ldc % 4 == 0) && (d->out_bstride % 4 == 0) && ((((uintptr_t)d->out) % (4 * esz)) == 0);
    p.res_vec_ok = (!d->bias || (((uintptr_t)d->bias) & 15) == 0) && (!d->colmask || (((uintptr_t)d->colmask) & 3) == 0) &&
                   (!d->resid || ((((uintptr_t)d->resid) & 7) == 0 && d->resid_ld % 4 == 0 && d->resid_bstride % 4 == 0 &&
                                  d->dyn_resid_mul % 4 == 0));
    p.fast_epi = 0;
    p.n_split = d->n_split;
    p.out2 = d->out2;
    p.out2_bstride = d->out2_bstride;
    p.ldc2 = d->ldc2;
    p.ooff2 = d->ooff2;
    p.dyn_ooff2_mul = d->dyn_ooff2_mul;
    if (d->n_split) {
        IFH_CHECK_ARG(d->n_split % 16 == 0 && d->out2 && d->ldc2 > 0 && d->n_split < d->n);
        p.vec_ok = p.vec_ok && (d->ldc2 % 4 == 0) && (d->out2_bstride % 4 == 0) && ((((uintptr_t)d->out2) % (4 * esz)) == 0);
    }
    p.fast_epi = p.vec_ok && p.res_vec_ok && (d->n % 4 == 0);
    p.aln_stats = d->aln_stats;
    p.aln_c1 = d->aln_c1;
    p.rln_stats = d->rln_stats;
    p.rln_gamma = d->rln_gamma;
    p.rln_beta = d->rln_beta;
    p.stats_out = d->stats_out;
    p.ln_dim = d->ln_dim;
    p.ln_eps = d->ln_eps;
    p.ln_rms = d->ln_rms;
    if (d->aln_stats || d->rln_stats || d->stats_out) {
        IFH_CHECK_ARG((int64_t)d->nbatch * d->t_out <= 1024 && d->taps == 1 && d->stride == 1 && d->pad == 0 && d->pre_slope == 1.0f);
        IFH_CHECK_ARG(d->n % 16 == 0 && d->ln_dim > 0 && !d->colmask && !d->accumulate && p.vec_ok && p.res_vec_ok);
        IFH_CHECK_ARG(!d->aln_stats || d->aln_c1 || d->ln_rms);
        IFH_CHECK_ARG(!d->ln_rms || !d->rln_stats);
        IFH_CHECK_ARG(!d->rln_stats || (d->rln_gamma && d->rln_beta && d->resid));
    }
    p.zt_cout = 0;
    p.amax_keys = (unsigned long long *)d->argmax_keys;
    p.whole_chip = d->whole_chip;
    if (d->convt_cout) {
        IFH_CHECK_ARG(d->taps == 3 && d->n == 4 * d->convt_cout && d->cin % 32 == 0 && d->stride == 1 && d->dil == 1);
        p.zt_cout = d->convt_cout;
    }
    p.dyn = d->dyn_pos;
    p.dyn_stride = d->dyn_stride;
    IFH_CHECK_ARG(d->dyn_stride == 0 || d->dyn_stride == 1);
    p.dyn_ooff_mul = d->dyn_ooff_mul;
    p.dyn_resid_mul = d->dyn_resid_mul;
    const bool pre = d->pre_slope != 1.0f;
    const int64_t M = (int64_t)d->nbatch * d->t_out;
    IFH_CHECK_ARG(M < (1ll << 31));
    hipStream_t st = as_stream(stream);
    const bool ln_fold = d->aln_stats || d->rln_stats || d->stats_out;
    const bool aln_only = d->aln_stats && !d->rln_stats && !d->stats_out;     // k_gemm_m64 can consume row statistics, not produce them
    const bool glu = d->act == IFH_ACT_SILU_GLU;
    if (glu && M > 64) {
        // thousands of rows (LLM prefill): the SiLU-gate epilogue exists in the DMA-ring kernel (gemm_big.hip) for whole 256 x 128 tiles
        IFH_CHECK_ARG(!ln_fold && !d->bias);
        if (try_launch_gemm_big(p, pre, st, (float *)d->splitk_ws, d->splitk_ws_floats)) {
            IFH_LAUNCH_CHECK("conv_bf16");
            return IFH_OK;
        }
        return fail(IFH_EINVAL, "conv_bf16: the SiLU-gate epilogue takes 17..64 rows, or whole 256 x 128 tiles from 4096 rows up");
    }
    if (glu)
        IFH_CHECK_ARG(M <= 64 && M > 16 && d->n >= 8192 && d->n % 16 == 0 && !d->resid && !d->accumulate && !d->out_f32 &&
                      !d->colmask && d->n_split == 0 && d->t_out == (int)M && d->ostride == 1 && d->ooff == 0 && !d->dyn_pos &&
                      d->out_scale == 1.0f && !d->rln_stats && !d->stats_out && d->taps == 1 && d->stride == 1 && d->pad == 0 && !pre);
    const bool wide_m64 = M <= 64 && M > 16 && d->taps == 1 && d->stride == 1 && d->pad == 0 && !pre && (!ln_fold || aln_only) && d->n >= 8192 &&
                          ((int64_t)d->n * p.K >= (int64_t)4096 * 1024 || glu);
    if (d->argmax_keys)       // arg-max keys exist in k_gemm_m64's epilogue only, on the values as stored (ifh_conv_argmax_supported).  A folded
                              // normalisation must be the RMS form: k_gemm_m64 takes the keys from the raw sums, which a positive per-row
                              // scale leaves in order and a LayerNorm fold's per-column mean term does not (k_gemm_m64d keys the stored values)
        IFH_CHECK_ARG(wide_m64 && !glu && d->out_f32 && !d->bias && !d->resid && !d->accumulate && !d->colmask && d->act == IFH_ACT_NONE &&
                      d->out_scale == 1.0f && d->n_split == 0 && d->nbatch == 1 && (((uintptr_t)d->argmax_keys) & 7) == 0 &&
                      (!d->aln_stats || d->ln_rms));
    if (wide_m64) {
        // LLM-sized wide layer at decode batch (gate|up 17920 x 1536, the vocabulary head): every weight byte once
        // (k_gemm_m64).  Narrow deep layers (down 1536 x 8960: 96 column tiles) stay with the 16 x 16-tile kernel below:
        // one block per column tile leaves 160 CUs idle and measured slower (41.8 vs 34.2 us).
        if (try_launch_gemm_m64d(p, st)) {            // whole-line DMA form (gemm_m64d.hip)
            IFH_LAUNCH_CHECK("conv_bf16");
            return IFH_OK;
        }
        const int ct = (d->n + 15) / 16;
        // column tiles per block: 4 wherever that still gives a block per CU.  Fewer (more, smaller blocks) was measured
        // slower at every size (gate|up 27.9 / 29.2 / 37.4 us, head 159 / 200 / 296 us for 4 / 2 / 1): each block re-reads the
        // whole activation matrix from L2, and L2 -> CU bytes (~4 TB/s here), not HBM, are what bounds this kernel
        if (ct >= 4 * 256)
            hipLaunchKernelGGL((k_gemm_m64<4, 2>), dim3((ct + 3) / 4), dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((k_gemm_m64<2, 3>), dim3((ct + 1) / 2), dim3(256), 0, st, p);
    } else if (aln_only && !glu && M >= igemm_aln_rows() && d->n >= 8192 && p.fast_epi && !pre && d->taps == 1 && d->stride == 1 && d->pad == 0 &&
               p.K % 32 == 0 && d->t_out <= d->t_in && !d->accumulate && !d->resid) {
        // a wide LayerNorm-folded head over hundreds of rows (the vocabulary projection of a 5-beam Whisper step: 640 x 51 865,
        // K = 512) is a throughput GEMM, not a decode-step one: 128 x 128 tiles of k_igemm with the normalisation applied in its
        // epilogue (133 us as 8 110 tiles of 64 x 64 in k_gemm_dec, which moves twice the operand bytes from L2).  One accumulation
        // chain instead of two: the logits' last bits differ from what the same rows give below this row count.
        if (int rc = launch_igemm<128, 128, 2>(p, pre, st)) return rc;
    } else if ((M <= 256 || ((ln_fold || d->decode_step) && M <= 1024)) && d->taps == 1 && d->stride == 1 && d->pad == 0 && !pre) {
        // (LayerNorm-folded launches exist only in this kernel: up to 1024 rows -- the 640 decode rows of a 5-beam search)
        const dim3 grid((d->n + 15) / 16, (unsigned)((M + 15) / 16));
        // waves per block = K split: 2 (12 k-steps each at K = 768) / 4 for deep K.  With several decode loops in
        // flight (SpeechPipeline TTS lanes) fewer, longer waves beat 4/8 short ones by ~6 % end to end; alone the
        // launch takes the same time either way.  Above 64 rows a block takes two row tiles per weight fragment
        // (same K split, hence the same bits): the step time grows by the L2 re-reads of W per 16-row tile.
        constexpr bool one_tile = false;              // fixed by measurement (profiles/NOTES.md)
        // From a few hundred rows up the step is no longer launch-bound and the streaming kernel's L2 traffic (all of W per 32
        // rows) is what a launch costs: the LDS-tiled kernel with the same accumulation chains takes over (same bits).
        constexpr int dec_rows = 128;   // fixed by measurement (profiles/NOTES.md)
        // a deep narrow layer at decode batch (the LLM's down projection: 1536 x 8960 at 64 rows) is 96 x 4 workgroups of the
        // streaming kernel re-reading 4 x the weights and 96 x the activations from L2 (34 us for 27.5 MB of weights): the LDS-tiled
        // kernel with one of the four accumulation chains per workgroup (blockIdx.z) + a finishing pass -- the same chains added in
        // the same order, hence the same bits
        constexpr int splitk_on = 1;        // fixed by measurement (profiles/NOTES.md)
        if (splitk_on && M > 16 && M <= 64 && p.K >= 4096 && p.K % 32 == 0 && d->n % 32 == 0 && d->n >= 512 && p.vec_ok && !glu) {
            // whole-line DMA form (gemm_m64d.hip): as many K parts as the caller's workspace holds, one chain each, added in part order
            // by the same finishing pass (its sums differ in the last bits from the four-chain forms below)
            if (d->splitk_ws && !d->argmax_keys) {
                const int parts = try_launch_gemm_m64d_splitk(p, (float *)d->splitk_ws, d->splitk_ws_floats, st);
                if (parts > 0) {
                    hipLaunchKernelGGL(k_splitk_finish, dim3((d->n + 15) / 16, (unsigned)((M + 15) / 16)), dim3(64), 0, st, p, (const float *)d->splitk_ws, parts);
                    IFH_LAUNCH_CHECK("conv_bf16");
                    return IFH_OK;
                }
            }
            const int chains = 4;
            // the workspace belongs to the caller's decode state (ifh_conv_desc.splitk_ws): captured graphs that replay concurrently on
            // different streams each carry their own; the library keeps none
            float *ws = (d->splitk_ws && d->splitk_ws_floats >= (int64_t)chains * M * d->n && (((uintptr_t)d->splitk_ws) & 15) == 0)
                            ? (float *)d->splitk_ws : nullptr;
            if (ws) {
                hipLaunchKernelGGL((k_gemm_dec<32, 64, true>), dim3(1, (d->n + 31) / 32, chains), dim3(256), 0, st, p, chains, ws);
                hipLaunchKernelGGL(k_splitk_finish, dim3((d->n + 15) / 16, (unsigned)((M + 15) / 16)), dim3(64), 0, st, p, (const float *)ws, chains);
                IFH_LAUNCH_CHECK("conv_bf16");
                return IFH_OK;
            }
        }
        if (M >= dec_rows && p.K % 32 == 0 && p.K >= 64 && (!ln_fold || d->n % 16 == 0)) {
            const int ksplit = p.K >= 2048 ? 4 : 2;
            // 64 x 64 tiles; 64 x 32 where that is what it takes to give every CU a workgroup
            const int64_t t64 = ((M + 63) / 64) * ((d->n + 63) / 64);
            // (k_gemm_dec_deep -- the two halves of K in two wave groups of one workgroup: a stage alone 2.5-5 % faster, the pipelined
            // cycle 2 % slower -- was a switch of rounds 3-5, off by default; removed in round 6, profiles/NOTES.md "Round 3, late")
            constexpr int64_t bm32_max = 200;   // fixed by measurement (profiles/NOTES.md): 32-row tiles up to this many 64 x 32 workgroups (pipelined C3, three alternating runs each: 0 -> 10 006 x, 100 -> 10 001, 150 -> 10 177-10 279, 200 -> 10 313-10 326, 300 -> 10 194, 500 -> 10 101)
            if (t64 >= 200)
                hipLaunchKernelGGL((k_gemm_dec<64>), dim3((M + 63) / 64, (d->n + 63) / 64), dim3(256), 0, st, p, ksplit);
            else if (((M + 63) / 64) * ((d->n + 31) / 32) <= bm32_max)
                hipLaunchKernelGGL((k_gemm_dec<32, 32>), dim3((M + 31) / 32, (d->n + 31) / 32), dim3(256), 0, st, p, ksplit);
            else
                hipLaunchKernelGGL((k_gemm_dec<32>), dim3((M + 63) / 64, (d->n + 31) / 32), dim3(256), 0, st, p, ksplit);
        } else if (M > 64 && !one_tile) {
            const dim3 grid2((d->n + 15) / 16, (unsigned)((M + 31) / 32));
            // Grids beyond what the chip holds at once (qkv, ff1 at 192-256 rows: 1152-1536 blocks against 4 x 256
            // co-resident at 200 registers) take the half-depth load batch: 128 registers, twice the blocks per CU --
            // ff1 20.2 -> 15.4 us, qkv 14.6 -> 13.1 us at 256 rows, same bits (the k order of a wave is unchanged).
            // Measured without effect on the deep-K GEMM (ff2, 19 us at 256 rows): 8 waves, one row tile, shallower batches.
            constexpr bool u6 = true;      // fixed by measurement (profiles/NOTES.md)
            constexpr int nt2 = 3;            // fixed by measurement (profiles/NOTES.md): bit 0 deep K, bit 1 big grids
            const bool big = (int64_t)grid2.x * grid2.y > 1024;
            const dim3 grid3((d->n + 31) / 32, (unsigned)((M + 31) / 32));
            if (p.K >= 2048) {
                if (nt2 & 1) hipLaunchKernelGGL((k_gemm_skinny<4, 6, 2, 2>), grid3, dim3(256), 0, st, p);
                else hipLaunchKernelGGL((k_gemm_skinny<4, 12, 2>), grid2, dim3(256), 0, st, p);
            } else if (big && (nt2 & 2))     // 32 x 32 block, K still split 2-way (the bits do not depend on the row count)
                hipLaunchKernelGGL((k_gemm_skinny<2, 6, 2, 2>), grid3, dim3(128), 0, st, p);
            else if (big && (nt2 & 4))       // (tuning: the 4-way split of round 2 -- a different summation order)
                hipLaunchKernelGGL((k_gemm_skinny<4, 6, 2, 2>), grid3, dim3(256), 0, st, p);
            else if (u6 && big)
                hipLaunchKernelGGL((k_gemm_skinny<2, 6, 2>), grid2, dim3(128), 0, st, p);
            else
                hipLaunchKernelGGL((k_gemm_skinny<2, 12, 2>), grid2, dim3(128), 0, st, p);
        } else if (p.K >= 2048)
            hipLaunchKernelGGL((k_gemm_skinny<4, 12, 1>), grid, dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL((k_gemm_skinny<2, 12, 1>), grid, dim3(128), 0, st, p);
    } else if (try_launch_gemm_big(p, pre, st, (float *)d->splitk_ws, d->splitk_ws_floats)) {
        // thousands of rows x whole 256 x 128 tiles: DMA ring, two workgroups per CU (gemm_big.hip)
    } else if (try_launch_conv_direct(p, pre, st)) {
        // residual-block shapes: input tile resident in LDS (conv.hip)
    } else if (d->n <= 32)
        launch_igemm<128, 32, 4>(p, pre, st);
    else if (M <= 64)
        launch_igemm<64, 32, 2>(p, pre, st);
    else if (d->n <= 64)
        launch_igemm<128, 64, 2>(p, pre, st);
    else {
        // a few hundred rows x a narrow layer (the 640 decode rows of a 5-beam search over 128 utterances: 20 tiles of
        // 128 x 128 at n = 512) leaves most of the 256 CUs idle: take the tile that yields at least one workgroup per CU.
        // The k order per output element does not depend on the tile: same bits.
        const int64_t mt = (M + 127) / 128;
        // thousands of row tiles (encoders, prefill): the 256 x 128 tile of eight waves moves a quarter fewer operand bytes per
        // FLOP through L2 -> CU, which is what these GEMMs wait for at K = 512..2048 (same k order per element: same bits)
        constexpr int big_rows = 0x7fffffff;   // fixed by measurement (profiles/NOTES.md): measured SLOWER (encoder 30.0 vs 28.2 ms), off
        if (M >= big_rows && (mt / 2) * ((d->n + 127) / 128) >= 512)
            launch_igemm<256, 128, 4, 8>(p, pre, st);
        else if (mt * ((d->n + 127) / 128) >= 256) {
            // three or more tiles per workgroup slot: 128-byte row pieces (KT = 64), +9-12 % at the Whisper-base encoder shapes
            constexpr int kt64_rows = 0;
            if (M >= kt64_rows && p.Cin % 64 == 0 && mt * ((d->n + 127) / 128) >= 3 * 768)
                launch_igemm<128, 128, 2, 4, 64>(p, pre, st);
            else
                launch_igemm<128, 128, 2>(p, pre, st);
        }
        else if (mt * ((d->n + 63) / 64) >= 256 || M > 4096)
            launch_igemm<128, 64, 2>(p, pre, st);
        else
            launch_igemm<64, 32, 2>(p, pre, st);
    }
    IFH_LAUNCH_CHECK("conv_bf16");
    return IFH_OK;
}

extern "C" int ifh_layernorm_bf16(const void *x, const void *resid, const float *gamma, const float *beta, void *out,
                                  int rows, int dim, float eps, ifh_stream_t stream)
{
    IFH_CHECK_ARG(rows >= 0);
    if (rows == 0) return IFH_OK;
    IFH_CHECK_ARG(x && gamma && beta && out && dim > 0 && dim <= 2048 && dim % 4 == 0);
    // (IFH_LN_RPW = 1: one row per wave at every size; read per call for the parity test)
    const char *rpw_env = getenv("IFH_LN_RPW");
    const int rpw = rpw_env && *rpw_env ? atoi(rpw_env) : 4;
    const bool many = rows >= 8192 && rpw == 4;
#define IFH_LN(NCH, RPW, RESID)                                                                                                   \
    hipLaunchKernelGGL((k_layernorm<NCH, RPW, RESID>), dim3((rows + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, as_stream(stream),  \
                       (const uint16_t *)x, (const uint16_t *)resid, gamma, beta, (uint16_t *)out, rows, dim, eps)
#define IFH_LN_R(NCH, RPW)                                                  \
    do {                                                                    \
        if (resid) IFH_LN(NCH, RPW, true); else IFH_LN(NCH, RPW, false);    \
    } while (0)
#define IFH_LN_N(NCH)                                                       \
    do {                                                                    \
        if (many) IFH_LN_R(NCH, 4); else IFH_LN_R(NCH, 1);                  \
    } while (0)
    // chunks of 256 elements per row: inactive chunks contribute nothing, so the count only decides how many dummy loads there are
    if (dim <= 512) IFH_LN_N(2);
    else if (dim <= 768) IFH_LN_N(3);
    else if (dim <= 1024) IFH_LN_N(4);
    else IFH_LN_R(8, 1);
#undef IFH_LN_N
#undef IFH_LN_R
#undef IFH_LN
    IFH_LAUNCH_CHECK("layernorm");
    return IFH_OK;
}

extern "C" int ifh_transpose_to_bf16(const void *in, int in_f32, void *out, int nbatch, int rows, int cols,
                                     ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0 && rows >= 0 && cols >= 0);
    if (nbatch == 0 || rows == 0 || cols == 0) return IFH_OK;
    IFH_CHECK_ARG(in && out && nbatch < 65536);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, nbatch);
    if (in_f32)
        hipLaunchKernelGGL(k_transpose<true>, grid, dim3(256), 0, as_stream(stream), in, (uint16_t *)out, rows, cols);
    else
        hipLaunchKernelGGL(k_transpose<false>, grid, dim3(256), 0, as_stream(stream), in, (uint16_t *)out, rows, cols);
    IFH_LAUNCH_CHECK("transpose");
    return IFH_OK;
}

#ifdef IFH_DEC_PROF
extern "C" int ifh_debug_dec_prof(unsigned long long *out8, int reset)
{
    if (out8) hipMemcpyFromSymbol(out8, HIP_SYMBOL(ifh::g_dec_prof), 64);
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(ifh::g_dec_prof), z, 64);
    }
    return 0;
}
#endif
