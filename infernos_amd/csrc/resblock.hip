// resblock.hip -- one HiFi-GAN residual pair in a single launch:
//     out = (x + conv2(lrelu(conv1(lrelu(x); k, dilation d)); k, dilation 1)) * out_scale  (+ out)
// (transformers modeling_speecht5.py HifiGanResidualBlock.forward, the loop body; Cin = Cout = C).
//
// The two convolutions of a pair are HBM-bound at C <= 128 when launched separately (conv.hip): each one
// reads its input tile and writes its output tile, and the second also re-reads x for the residual --
// five activation passes per pair.  Here the intermediate tile never leaves the CU: conv1 is evaluated on
// BM1 = BM + 16 rows (the rows conv2 needs on either side, k <= 17), rounded to bf16 and LeakyReLU'd
// exactly as the separate launches do (so results are bit-identical to them), kept in LDS, and conv2
// runs on it.  Two activation passes per pair (x in, out out).
//
// MFMA: v_mfma_f32_16x16x32_bf16, swapped operands (A = weight rows, B = activation rows) as in conv.hip.
// Weights: LDS-resident for C = 32, otherwise one stream of K-chunks (conv1's then conv2's) through a
// register prefetch.  All global operands of the epilogue (bias, residual rows, accumulate rows) are
// requested before the input tile so that no load sits behind a dependent wait.
#include <stdlib.h>

#include "igemm.h"

namespace ifh {

struct PairParams {
    const uint16_t *x;
    int64_t x_bstride;
    const uint16_t *w1, *w2;     // [C][taps][C]
    const float *b1, *b2;        // [C] or null
    int taps, dil, T, nbatch;
    int rows_per_block;          // valid output rows per block (<= BM)
    float slope, out_scale;
    int accumulate;
    uint16_t *out;
    int64_t out_bstride;
};

__device__ __forceinline__ uint2 lrelu4(uint2 v, float slope)
{
    float a = __uint_as_float(v.x << 16), b = __uint_as_float(v.x & 0xffff0000u);
    float c = __uint_as_float(v.y << 16), d = __uint_as_float(v.y & 0xffff0000u);
    a = fmaxf(a, a * slope);
    b = fmaxf(b, b * slope);
    c = fmaxf(c, c * slope);
    d = fmaxf(d, d * slope);
    return make_uint2(f32x2_to_bf16x2(a, b), f32x2_to_bf16x2(c, d));
}

template <int CIN, int WGM, int MT1, int NT, bool RESIDENT, int KC, int EPB, bool RES_LDS, int NWV, int MINW>
__global__ __launch_bounds__(64 * NWV, MINW) void k_resblock_pair(const PairParams p)
{
    constexpr int NTHR = 64 * NWV;            // 8 waves: two per SIMD inside the one block a CU holds at C = 128
    constexpr int WGN = NWV / WGM;
    static_assert(!RESIDENT || NWV == 4, "the resident-weight transfers assume 256 threads");
    constexpr bool LATE_ACC = RESIDENT || (CIN == 64 && !RES_LDS) || NWV > 8;       // C = 32: one exposed load in the 2-of-18 accumulating launches buys a third block per CU
    constexpr bool LATE_RES = !RES_LDS && (CIN == 64 || NWV > 8);      // residual rows read (L2-hot) in the epilogue: 28 fewer live registers
    constexpr int TPE = WGM * MT1 / EPB;      // conv1 tiles per batch entry
    constexpr int BM1E = TPE * 16;            // conv1 rows per entry
    constexpr int BME = BM1E - 16;            // conv2 (output) rows per entry
    constexpr int BN = WGN * NT * 16;
    static_assert(BN == CIN, "a block covers every channel");
    static_assert(EPB == 1 || WGM == 1, "several entries per block: one wave row");
    static_assert(TPE * EPB == WGM * MT1, "tiles split evenly over the entries");
    constexpr int XS = CIN + 8;
    constexpr int WV = RESIDENT ? 0 : (BN * KC / 8 + NTHR - 1) / NTHR;
    static_assert(WV <= 16, "prefetch registers");
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wid % WGM, wn = wid / WGM;
    const int b0 = blockIdx.y * EPB, t0 = blockIdx.x * p.rows_per_block;
    const int K = p.taps * CIN;
    const int halo1 = (p.taps - 1) * p.dil, h1 = halo1 / 2, h2 = (p.taps - 1) / 2;
    const int R1 = BM1E + halo1;              // input rows per entry
    const int KW = RESIDENT ? K : KC;
    const int WS = KW + 8;
    // LDS: [ input tile | later the intermediate tile | later the output tile ] [ weights ] [ raw residual rows ]
    uint16_t *Xs = lds;
    uint16_t *Ms = lds;
    uint16_t *Ws = Xs + ((EPB * R1 * XS + 7) & ~7);
    uint16_t *Rs = Ws + BN * WS;
    const int nvalid = min(p.rows_per_block, p.T - t0);

    // ---- epilogue operands first: bias vectors, rows of `out` to accumulate onto, and (when they are not
    // kept in LDS) the residual rows
    float4 bp1[NT], bp2[NT];
    uint2 rpre[NT][MT1], apre[NT][MT1];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int n = (wn * NT + i) * 16 + 4 * fg;
        bp1[i] = p.b1 ? *reinterpret_cast<const float4 *>(p.b1 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bp2[i] = p.b2 ? *reinterpret_cast<const float4 *>(p.b2 + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < MT1; j++) {
        const int g = wm * MT1 + j, e = g / TPE, lt = g - e * TPE;
        const int t = min(t0 + lt * 16 + fr, p.T - 1);
        const int be = min(b0 + e, p.nbatch - 1);
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = (wn * NT + i) * 16 + 4 * fg;
            rpre[i][j] = (RES_LDS || LATE_RES) ? make_uint2(0, 0)
                                 : *reinterpret_cast<const uint2 *>(p.x + (int64_t)be * p.x_bstride + (int64_t)t * CIN + n);
            apre[i][j] = (p.accumulate && !LATE_ACC)      // LATE_ACC reads them in the epilogue instead: fewer live registers
                             ? *reinterpret_cast<const uint2 *>(p.out + (int64_t)be * p.out_bstride + (int64_t)t * CIN + n)
                             : make_uint2(0, 0);
        }
    }
    // ---- weights.  Streamed: chunk c < nchunk belongs to conv1, the rest to conv2 (named registers: see
    // conv.hip).  RESIDENT (C = 32): conv1's weights go to LDS now, conv2's wait in registers.
    const int nchunk = RESIDENT ? 1 : K / KC;
    uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15;
    w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = w8 = w9 = w10 = w11 = w12 = w13 = w14 = w15 = make_uint4(0, 0, 0, 0);
#define RB_W1(I, REG, SRC, K0)                                                                   \
    if (I < WV && ((I + 1) * NTHR <= BN * KC / 8 || tid + NTHR * I < BN * KC / 8)) {             \
        const int v = tid + NTHR * I;                                                            \
        REG = *reinterpret_cast<const uint4 *>((SRC) + (int64_t)(v / (KC / 8)) * K + (K0) + (v % (KC / 8)) * 8); \
    }
#define RB_W_PREFETCH(CH)                                                                        \
    if (!RESIDENT) {                                                                             \
        const uint16_t *src_ = (CH) < nchunk ? p.w1 : p.w2;                                      \
        const int k0_ = ((CH) < nchunk ? (CH) : (CH) - nchunk) * KC;                             \
        RB_W1(0, w0, src_, k0_) RB_W1(1, w1, src_, k0_) RB_W1(2, w2, src_, k0_) RB_W1(3, w3, src_, k0_)     \
        RB_W1(4, w4, src_, k0_) RB_W1(5, w5, src_, k0_) RB_W1(6, w6, src_, k0_) RB_W1(7, w7, src_, k0_)     \
        RB_W1(8, w8, src_, k0_) RB_W1(9, w9, src_, k0_) RB_W1(10, w10, src_, k0_) RB_W1(11, w11, src_, k0_) \
        RB_W1(12, w12, src_, k0_) RB_W1(13, w13, src_, k0_) RB_W1(14, w14, src_, k0_) RB_W1(15, w15, src_, k0_) \
    }
#define RB_C1(I, REG)                                                                            \
    if (I < WV && ((I + 1) * NTHR <= BN * KC / 8 || tid + NTHR * I < BN * KC / 8)) {             \
        const int v = tid + NTHR * I;                                                            \
        *reinterpret_cast<uint4 *>(&Ws[(v / (KC / 8)) * WS + (v % (KC / 8)) * 8]) = REG;         \
    }
#define RB_W_COMMIT()                                                                            \
    if (!RESIDENT) {                                                                             \
        RB_C1(0, w0) RB_C1(1, w1) RB_C1(2, w2) RB_C1(3, w3) RB_C1(4, w4) RB_C1(5, w5) RB_C1(6, w6) RB_C1(7, w7)      \
        RB_C1(8, w8) RB_C1(9, w9) RB_C1(10, w10) RB_C1(11, w11) RB_C1(12, w12) RB_C1(13, w13) RB_C1(14, w14) RB_C1(15, w15) \
    }
    // whole-matrix transfers for RESIDENT: vector v of [BN][K/8] (at most 8 per thread: taps <= 17 at C = 32)
#define RB_R1(I, REG, SRC)                                                                       \
    {                                                                                            \
        const int v = tid + 256 * I;                                                             \
        if (v < BN * (K / 8)) REG = *reinterpret_cast<const uint4 *>((SRC) + (int64_t)v * 8);    \
    }
#define RB_RC1(I, REG)                                                                           \
    {                                                                                            \
        const int v = tid + 256 * I;                                                             \
        if (v < BN * (K / 8)) *reinterpret_cast<uint4 *>(&Ws[(v / (K / 8)) * WS + (v % (K / 8)) * 8]) = REG; \
    }
    if (RESIDENT) {
        RB_R1(0, w0, p.w1) RB_R1(1, w1, p.w1) RB_R1(2, w2, p.w1) RB_R1(3, w3, p.w1)
        RB_R1(4, w4, p.w1) RB_R1(5, w5, p.w1)
        RB_R1(0, w8, p.w2) RB_R1(1, w9, p.w2) RB_R1(2, w10, p.w2) RB_R1(3, w11, p.w2)
        RB_R1(4, w12, p.w2) RB_R1(5, w13, p.w2)                                         // <= 6 vectors: taps <= 12 at C = 32
    } else {
        RB_W_PREFETCH(0)
    }
    // ---- input tile: row q of entry e <-> time t0 - 8 - h1 + q, LeakyReLU applied once here; the raw rows
    // of the output range also go to Rs (the residual operand).  All loads in flight together.
    {
        constexpr int VPR = CIN / 8;
        constexpr int XVB = (EPB * (BM1E + 50) * VPR + NTHR - 1) / NTHR;
        const int tx0 = t0 - 8 - h1;
        const int nvec = EPB * R1 * VPR;
        for (int base = 0; base < nvec; base += NTHR * XVB) {
            uint4 xv[XVB];
#pragma unroll
            for (int i = 0; i < XVB; i++) {
                const int v = base + tid + NTHR * i;
                const int rr = v / VPR, c = (v - rr * VPR) * 8;
                const int e = EPB == 1 ? 0 : rr / R1;
                const int tin = tx0 + (rr - e * R1);
                const bool ok = v < nvec && tin >= 0 && tin < p.T && b0 + e < p.nbatch;
                xv[i] = *reinterpret_cast<const uint4 *>(p.x + (ok ? (int64_t)(b0 + e) * p.x_bstride + (int64_t)tin * CIN + c : 0));
                if (!ok) xv[i] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < XVB; i++) {
                const int v = base + tid + NTHR * i;
                const int rr = v / VPR, c = (v - rr * VPR) * 8;
                if (v < nvec) {
                    *reinterpret_cast<uint4 *>(&Xs[rr * XS + c]) = lrelu8(xv[i], p.slope);
                    if (RES_LDS) {
                        const int e = EPB == 1 ? 0 : rr / R1;
                        const int q = rr - e * R1 - 8 - h1;
                        if (q >= 0 && q < BME) *reinterpret_cast<uint4 *>(&Rs[(e * BME + q) * XS + c]) = xv[i];
                    }
                }
            }
        }
    }
    if (RESIDENT) {
        RB_RC1(0, w0) RB_RC1(1, w1) RB_RC1(2, w2) RB_RC1(3, w3) RB_RC1(4, w4) RB_RC1(5, w5)
    }

    f32x4 acc[NT][MT1];
    const int brow0 = (wn * NT * 16 + fr) * WS + fg * 8;
    // =========================== conv1 (dilation d) on BM1 rows ===========================
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT1; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        int xrow[MT1];
#pragma unroll
        for (int j = 0; j < MT1; j++) {
            const int g = wm * MT1 + j, e = g / TPE, lt = g - e * TPE;
            xrow[j] = (e * R1 + lt * 16 + fr) * XS + fg * 8;
        }
        for (int ch = 0; ch < nchunk; ch++) {
            if (!RESIDENT) {
                if (ch > 0) __syncthreads();
                RB_W_COMMIT()
            }
            __syncthreads();
            if (!RESIDENT) RB_W_PREFETCH(ch + 1)          // conv2's first chunk follows conv1's last
#define RB_KSTEP1(ks)                                                                             \
            {\
                const int k = ch * KC + ks * 32; \
                const int tap = k / CIN, ci = k - tap * CIN; \
                const int aoff = tap * p.dil * XS + ci; \
                bf16x8_t fa[NT], fb[MT1]; \
_Pragma("unroll") \
                for (int i = 0; i < NT; i++) fa[i] = *reinterpret_cast<const bf16x8_t *>(&Ws[brow0 + i * 16 * WS + ks * 32]); \
_Pragma("unroll") \
                for (int j = 0; j < MT1; j++) fb[j] = *reinterpret_cast<const bf16x8_t *>(&Xs[xrow[j] + aoff]); \
_Pragma("unroll") \
                for (int i = 0; i < NT; i++) \
_Pragma("unroll") \
                    for (int j = 0; j < MT1; j++) \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0); \
            }
            // streamed weights: the k-steps of a chunk are fully unrolled so that the next step's fragment
            // loads overlap this step's MFMAs (one wave per SIMD: nothing else hides the LDS latency)
            if (RESIDENT) {
#pragma unroll 2
                for (int ks = 0; ks < KW / 32; ks++) RB_KSTEP1(ks)
            } else {
#pragma unroll
                for (int ks = 0; ks < KC / 32; ks++) RB_KSTEP1(ks)
            }
#undef RB_KSTEP1
        }
    }
    // conv1 epilogue: + bias, round to bf16 (what the separate launch stores), LeakyReLU (what the next
    // launch applies on load); rows outside [0, T) are conv2's zero padding.  The tile replaces the input tile.
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MT1; j++) {
        const int g = wm * MT1 + j, e = g / TPE, lt = g - e * TPE;
        const int tmid = t0 - 8 + lt * 16 + fr;
        const bool ok = tmid >= 0 && tmid < p.T;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = (wn * NT + i) * 16 + 4 * fg;
            const f32x4 a = acc[i][j];
            uint2 pk = make_uint2(f32x2_to_bf16x2(a[0] + bp1[i].x, a[1] + bp1[i].y), f32x2_to_bf16x2(a[2] + bp1[i].z, a[3] + bp1[i].w));
            pk = lrelu4(pk, p.slope);
            *reinterpret_cast<uint2 *>(&Ms[(g * 16 + fr) * XS + n]) = ok ? pk : make_uint2(0, 0);
            acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // =========================== conv2 (dilation 1) on the first TPE-1 tiles of every entry ===========================
    {
        const int arow0 = (wm * MT1 * 16 + fr + 8 - h2) * XS + fg * 8;
        for (int ch = 0; ch < nchunk; ch++) {
            __syncthreads();                               // previous chunk consumed (first: conv1's last chunk)
            if (!RESIDENT) {
                RB_W_COMMIT()
            } else {
                RB_RC1(0, w8) RB_RC1(1, w9) RB_RC1(2, w10) RB_RC1(3, w11) RB_RC1(4, w12) RB_RC1(5, w13)
            }
            __syncthreads();                               // (first: also publishes Ms)
            if (!RESIDENT && ch + 1 < nchunk) RB_W_PREFETCH(nchunk + ch + 1)
#define RB_KSTEP2(ks)                                                                             \
            {\
                const int k = ch * KC + ks * 32; \
                const int tap = k / CIN, ci = k - tap * CIN; \
                const int aoff = arow0 + tap * XS + ci; \
                bf16x8_t fa[NT], fb[MT1]; \
_Pragma("unroll") \
                for (int i = 0; i < NT; i++) fa[i] = *reinterpret_cast<const bf16x8_t *>(&Ws[brow0 + i * 16 * WS + ks * 32]); \
_Pragma("unroll") \
                for (int j = 0; j < MT1; j++) \
                    if ((wm * MT1 + j) % TPE != TPE - 1) fb[j] = *reinterpret_cast<const bf16x8_t *>(&Ms[aoff + j * 16 * XS]); \
_Pragma("unroll") \
                for (int i = 0; i < NT; i++) \
_Pragma("unroll") \
                    for (int j = 0; j < MT1; j++) \
                        if ((wm * MT1 + j) % TPE != TPE - 1) \
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0); \
            }
            if (RESIDENT) {
#pragma unroll 2
                for (int ks = 0; ks < KW / 32; ks++) RB_KSTEP2(ks)
            } else {
#pragma unroll
                for (int ks = 0; ks < KC / 32; ks++) RB_KSTEP2(ks)
            }
#undef RB_KSTEP2
        }
    }
#undef RB_W_PREFETCH
#undef RB_W_COMMIT
#undef RB_W1
#undef RB_C1
#undef RB_R1
#undef RB_RC1
    // conv2 epilogue (+ bias, + x, * scale, + previous out) -> LDS (over the intermediate tile) -> full-row 16-byte stores
    constexpr int OS = BN + 8;
    __syncthreads();
    uint16_t *Os = lds;
#pragma unroll
    for (int j = 0; j < MT1; j++) {
        const int g = wm * MT1 + j, e = g / TPE, lt = g - e * TPE;
        if (lt == TPE - 1) continue;
        const int orow = e * BME + lt * 16 + fr;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = (wn * NT + i) * 16 + 4 * fg;
            const f32x4 a = acc[i][j];
            const uint2 rv = RES_LDS ? *reinterpret_cast<const uint2 *>(&Rs[orow * XS + n])
                             : LATE_RES ? *reinterpret_cast<const uint2 *>(p.x + (int64_t)min(b0 + e, p.nbatch - 1) * p.x_bstride +
                                                                           (int64_t)min(t0 + lt * 16 + fr, p.T - 1) * CIN + n)
                                        : rpre[i][j];
            float v0 = a[0] + bp2[i].x, v1 = a[1] + bp2[i].y, v2 = a[2] + bp2[i].z, v3 = a[3] + bp2[i].w;
            v0 += __uint_as_float(rv.x << 16);
            v1 += __uint_as_float(rv.x & 0xffff0000u);
            v2 += __uint_as_float(rv.y << 16);
            v3 += __uint_as_float(rv.y & 0xffff0000u);
            v0 *= p.out_scale; v1 *= p.out_scale; v2 *= p.out_scale; v3 *= p.out_scale;
            if (p.accumulate) {
                const uint2 pv = LATE_ACC ? *reinterpret_cast<const uint2 *>(
                                                p.out + (int64_t)min(b0 + e, p.nbatch - 1) * p.out_bstride + (int64_t)min(t0 + lt * 16 + fr, p.T - 1) * CIN + n)
                                          : apre[i][j];
                v0 += __uint_as_float(pv.x << 16);
                v1 += __uint_as_float(pv.x & 0xffff0000u);
                v2 += __uint_as_float(pv.y << 16);
                v3 += __uint_as_float(pv.y & 0xffff0000u);
            }
            *reinterpret_cast<uint2 *>(&Os[orow * OS + n]) = make_uint2(f32x2_to_bf16x2(v0, v1), f32x2_to_bf16x2(v2, v3));
        }
    }
    __syncthreads();
    constexpr int VPRO = BN / 8;
#pragma unroll 4
    for (int v = tid; v < EPB * BME * VPRO; v += NTHR) {
        const int orow = v / VPRO, c = (v - orow * VPRO) * 8;
        const int e = EPB == 1 ? 0 : orow / BME;
        const int row = orow - e * BME;
        if (row < nvalid && b0 + e < p.nbatch)
            *reinterpret_cast<uint4 *>(p.out + (int64_t)(b0 + e) * p.out_bstride + (int64_t)(t0 + row) * CIN + c) =
                *reinterpret_cast<const uint4 *>(&Os[orow * OS + c]);
    }
}

template <int CIN, int WGM, int MT1, int NT, bool RESIDENT, int EPB, bool RES_LDS, int KC = 64, int NWV = 4, int MINW = 1>
static int launch_pair(PairParams &p, hipStream_t st)
{
    constexpr int WGN = NWV / WGM;
    constexpr int TPE = WGM * MT1 / EPB, BM1E = TPE * 16, BME = BM1E - 16, BN = WGN * NT * 16;
    constexpr int XS = CIN + 8;
    const int K = p.taps * CIN;
    const int R1 = BM1E + (p.taps - 1) * p.dil;
    const int KW = RESIDENT ? K : KC;
    if (!RESIDENT && K % KC != 0) return fail(IFH_EINVAL, "resblock_pair: taps*c must be a multiple of 64");
    if (RESIDENT && BN * (K / 8) > 6 * 256) return fail(IFH_EINVAL, "resblock_pair: at most 11 taps at c = 32");
    const size_t bytes = ((size_t)((EPB * R1 * XS + 7) & ~7) + (size_t)BN * (KW + 8) + (RES_LDS ? (size_t)EPB * BME * XS : 0)) * sizeof(uint16_t);
    if (bytes > 160 * 1024) return fail(IFH_EINVAL, "resblock_pair: tile does not fit in LDS (taps*dil too large)");
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        hipError_t e = hipFuncSetAttribute((const void *)k_resblock_pair<CIN, WGM, MT1, NT, RESIDENT, KC, EPB, RES_LDS, NWV, MINW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return check_hip(e, "resblock_pair lds attr");
        attr_once.done(attr_dev);
    }
    const int nblk = (p.T + BME - 1) / BME;
    p.rows_per_block = (p.T + nblk - 1) / nblk;
    dim3 grid(nblk, (p.nbatch + EPB - 1) / EPB);
    hipLaunchKernelGGL((k_resblock_pair<CIN, WGM, MT1, NT, RESIDENT, KC, EPB, RES_LDS, NWV, MINW>), grid, dim3(64 * NWV), bytes, st, p);
    return IFH_OK;
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_resblock_pair_bf16(const ifh_resblock_desc *d, ifh_stream_t stream)
{
    IFH_CHECK_ARG(d);
    IFH_CHECK_ARG(d->x && d->w1 && d->w2 && d->out);
    IFH_CHECK_ARG(d->nbatch >= 0 && d->t >= 0);
    if (d->nbatch == 0 || d->t == 0) return IFH_OK;
    IFH_CHECK_ARG(d->c == 32 || d->c == 64 || d->c == 128 || d->c == 256);
    IFH_CHECK_ARG(d->taps >= 1 && d->taps <= 15 && (d->taps & 1) == 1 && d->dil >= 1 && d->nbatch < 65536);
    IFH_CHECK_ARG((((uintptr_t)d->x) & 15) == 0 && (((uintptr_t)d->out) & 15) == 0 && (((uintptr_t)d->w1) & 15) == 0 &&
                  (((uintptr_t)d->w2) & 15) == 0 && d->x_bstride % 8 == 0 && d->out_bstride % 8 == 0);
    IFH_CHECK_ARG((!d->bias1 || (((uintptr_t)d->bias1) & 15) == 0) && (!d->bias2 || (((uintptr_t)d->bias2) & 15) == 0));
    IFH_CHECK_ARG(d->slope > 0.0f && d->slope <= 1.0f);
    PairParams p;
    p.x = (const uint16_t *)d->x;
    p.x_bstride = d->x_bstride;
    p.w1 = (const uint16_t *)d->w1;
    p.w2 = (const uint16_t *)d->w2;
    p.b1 = d->bias1;
    p.b2 = d->bias2;
    p.taps = d->taps;
    p.dil = d->dil;
    p.T = d->t;
    p.nbatch = d->nbatch;
    p.rows_per_block = 0;
    p.slope = d->slope;
    p.out_scale = d->out_scale;
    p.accumulate = d->accumulate;
    p.out = (uint16_t *)d->out;
    p.out_bstride = d->out_bstride;
    hipStream_t st = as_stream(stream);
    constexpr int nwv8 = 1;      // fixed by measurement (profiles/NOTES.md): 8-wave blocks at C = 128
    constexpr int c64v = 1;      // fixed by measurement (profiles/NOTES.md)
    constexpr int c32v = 1;      // fixed by measurement (profiles/NOTES.md)
    int rc;
    switch (d->c) {                                                                // <C, WGM, MT1, NT, resident W, entries/block, residual rows in LDS>
    case 256: rc = launch_pair<256, 1, 4, 4, false, 1, true, 128>(p, st); break;   // conv1 64 rows, out 48; 128-wide weight chunks
    case 128:                                                                      // conv1 224 rows, out 208; 128-wide chunks (7-10 % over 64)
        // the tile takes 109 KB of LDS (one block per CU): 8 waves give every SIMD a second wave to cover the two
        // barriers per weight chunk -- 10-13 % over 4 waves at 1024 chunks (88 -> 77, 141 -> 127, 195 -> 177 us for
        // 3 / 7 / 11 taps).  At C = 64 (2 blocks per CU already) the same split costs 35 % (NT = 1 doubles the LDS reads per MFMA).
        // (14 waves -- 7 row groups x 2 -- measured 3-4 % slower than 8)
        rc = nwv8 ? launch_pair<128, 2, 7, 2, false, 1, false, 128, 8>(p, st) : launch_pair<128, 2, 7, 4, false, 1, false, 128>(p, st);
        break;
    case 64:                                                                       // conv1 224 rows, out 208 (K = 64*taps: 64-wide chunks)
        // residual rows re-read from L2 in the epilogue instead of kept in LDS (49 KB per block) and the kernel compiled
        // for 168 registers: 3 blocks per CU instead of 2 -- 15-19 % per launch (97 -> 82, 152 -> 127, 210 -> 171 us)
        rc = c64v ? launch_pair<64, 2, 7, 2, false, 1, false, 64, 4, 3>(p, st) : launch_pair<64, 2, 7, 2, false, 1, true>(p, st);
        break;
    default:                                                                       // conv1 256 rows, out 240
        // compiled for 128 registers (126 used, no spills; 160 without the bound): 4 blocks per CU where LDS allows
        // (3 / 7 taps), 7-13 % per launch (139 -> 128, 189 -> 166, 238 -> 210 us at 1024 chunks).  A 96-register build spills.
        rc = c32v ? launch_pair<32, 4, 4, 2, true, 1, false, 64, 4, 4>(p, st) : launch_pair<32, 4, 4, 2, true, 1, false>(p, st);
        break;
    }
    if (rc != IFH_OK) return rc;
    IFH_LAUNCH_CHECK("resblock_pair_bf16");
    return IFH_OK;
}
