// conv.hip -- stride-1 Conv1d with the input tile resident in LDS, for the HiFi-GAN residual
// blocks (Cin = Cout in {32, 64, 128, 256}, kernel 3/7/11, dilation 1/3/5), which carry
// ~97 % of the vocoder FLOPs (66.06 of 68.3 MFLOP per mel frame per level).
//
// Versus the generic implicit GEMM (nn.hip:k_igemm) the activation rows of a block
// (BM + (taps-1)*dil rows x Cin) are staged ONCE -- with the fused LeakyReLU applied once per
// element instead of once per tap -- and every tap's MFMA operand is a shifted row window of
// that tile.  Weights: fully LDS-resident for Cin <= 64 (the K loop then runs barrier-free),
// streamed in 64-wide K chunks through a register prefetch for Cin >= 128.
// MFMA: v_mfma_f32_16x16x32_bf16, swapped operands (A = weight rows, B = activation rows), same
// fused epilogue as k_igemm (bias, residual, scale, accumulate).
#include <stdlib.h>

#include "igemm.h"

namespace ifh {

// EPB > 1: one block covers EPB consecutive batch rows (vocoder chunks), one 4-wave group each, sharing the streamed
// weight chunks in LDS -- at Cin 256 a 48-row block re-reads all of W from L2 and the launch is bound by the L2 -> CU
// rate (measured 6.4 us per tap at 768 chunks = 100 MB of W per tap = 15.7 TB/s); two chunks per block halve that.
template <int CIN, int WGM, int MT, int NT, bool RESIDENT, int KC, int EPB>
__global__ __launch_bounds__(256 * EPB) void k_conv_direct(const IgemmParams p)
{
    constexpr int NTHR = 256 * EPB;
    constexpr int WGN = 4 / WGM;
    constexpr int BM = WGM * MT * 16, BN = WGN * NT * 16;
    constexpr int XS = CIN + 8;   // LDS row stride (elements): 16-byte pad
    constexpr int WV = RESIDENT ? 0 : (BN * KC / 8 / NTHR);   // prefetch vectors per thread (<= 16)
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, grp = tid >> 8, gt = tid & 255;
    const int fr = lane & 15, fg = lane >> 4;
    const int wm = wid % WGM, wn = wid / WGM;
    const bool live = (int)blockIdx.y * EPB + grp < p.nbatch;       // a group past the last batch row computes on a copy
    const int b = live ? blockIdx.y * EPB + grp : p.nbatch - 1, t0 = blockIdx.x * BM;
    const int halo = (p.taps - 1) * p.dil;
    const int R = BM + halo;
    const int KW = RESIDENT ? p.K : KC;
    const int WS = KW + 8;
    const int XT = (R * XS + 7) & ~7;
    uint16_t *Xs = lds + grp * XT;
    uint16_t *Ws = lds + EPB * XT;

    // residual operand of the epilogue, requested first (ahead of the input tile) and consumed after the K loop
    const int dynv = p.dyn ? p.dyn[0] : 0;
    uint2 rpre[NT][MT];
    float4 bpre[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
        bpre[i] = p.bias ? *reinterpret_cast<const float4 *>(p.bias + (wn * NT + i) * 16 + 4 * fg) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.resid) {
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int t = min(t0 + (wm * MT + j) * 16 + fr, p.T_out - 1);
            const EpiRow e = epi_row(p, b * p.T_out + t, 0, dynv);
#pragma unroll
            for (int i = 0; i < NT; i++)
                rpre[i][j] = *reinterpret_cast<const uint2 *>(p.resid + e.rbase + (wn * NT + i) * 16 + 4 * fg);
        }
    }

    // ---- stage the input rows once (zero outside [0, T_in)), LeakyReLU fused here.  All of a thread's
    // loads are issued before the first LDS write (XVB covers a halo of 50 rows = 11 taps x dilation 5 in
    // one batch): a load-use-load loop would expose one HBM round trip per vector.
    {
        constexpr int VPR = CIN / 8;
        constexpr int XVB = ((BM + 50) * VPR + 255) / 256;
        const uint16_t *xb = p.x + (int64_t)b * p.x_bstride;
        const bool pre = p.pre_slope != 1.0f;
        for (int base = 0; base < R * VPR; base += 256 * XVB) {
            uint4 xv[XVB];
#pragma unroll
            for (int i = 0; i < XVB; i++) {
                const int v = base + gt + 256 * i;
                const int r = v / VPR, c = (v - r * VPR) * 8;
                const int tin = t0 - p.pad + r;
                const bool ok = v < R * VPR && tin >= 0 && tin < p.T_in;
                xv[i] = *reinterpret_cast<const uint4 *>(xb + (ok ? (int64_t)tin * p.lda + c : 0));
                if (!ok) xv[i] = make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < XVB; i++) {
                const int v = base + gt + 256 * i;
                const int r = v / VPR, c = (v - r * VPR) * 8;
                if (v < R * VPR) *reinterpret_cast<uint4 *>(&Xs[r * XS + c]) = pre ? lrelu8(xv[i], p.pre_slope) : xv[i];
            }
        }
    }
    // Prefetch registers are NAMED scalars driven by macros: as an array (or captured in a lambda)
    // hipcc kept them in scratch memory.
    uint4 w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15;
    w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = w8 = w9 = w10 = w11 = w12 = w13 = w14 = w15 = make_uint4(0, 0, 0, 0);
#define IFH_W1(I, REG, K0)                                                                       \
    if (I < WV) {                                                                                \
        const int v = tid + NTHR * I;                                                            \
        REG = *reinterpret_cast<const uint4 *>(p.w + (int64_t)(v / (KC / 8)) * p.K + (K0) + (v % (KC / 8)) * 8); \
    }
#define IFH_W_PREFETCH(K0)                                                                       \
    if (!RESIDENT) {                                                                             \
        IFH_W1(0, w0, K0) IFH_W1(1, w1, K0) IFH_W1(2, w2, K0) IFH_W1(3, w3, K0)                  \
        IFH_W1(4, w4, K0) IFH_W1(5, w5, K0) IFH_W1(6, w6, K0) IFH_W1(7, w7, K0)                  \
        IFH_W1(8, w8, K0) IFH_W1(9, w9, K0) IFH_W1(10, w10, K0) IFH_W1(11, w11, K0)              \
        IFH_W1(12, w12, K0) IFH_W1(13, w13, K0) IFH_W1(14, w14, K0) IFH_W1(15, w15, K0)          \
    }
#define IFH_C1(I, REG)                                                                           \
    if (I < WV) {                                                                                \
        const int v = tid + NTHR * I;                                                            \
        *reinterpret_cast<uint4 *>(&Ws[(v / (KC / 8)) * WS + (v % (KC / 8)) * 8]) = REG;         \
    }
#define IFH_W_COMMIT()                                                                           \
    if (!RESIDENT) {                                                                             \
        IFH_C1(0, w0) IFH_C1(1, w1) IFH_C1(2, w2) IFH_C1(3, w3)                                  \
        IFH_C1(4, w4) IFH_C1(5, w5) IFH_C1(6, w6) IFH_C1(7, w7)                                  \
        IFH_C1(8, w8) IFH_C1(9, w9) IFH_C1(10, w10) IFH_C1(11, w11)                              \
        IFH_C1(12, w12) IFH_C1(13, w13) IFH_C1(14, w14) IFH_C1(15, w15)                          \
    }
    if (RESIDENT) {
        const int vpr = p.K / 8;
        for (int v = tid; v < BN * vpr; v += NTHR) {
            const int n = v / vpr, kv = (v - n * vpr) * 8;
            *reinterpret_cast<uint4 *>(&Ws[n * WS + kv]) = *reinterpret_cast<const uint4 *>(p.w + (int64_t)n * p.K + kv);
        }
    } else {
        IFH_W_PREFETCH(0)
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < MT; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nchunk = RESIDENT ? 1 : p.K / KC;
    const int arow0 = (wm * MT * 16 + fr) * XS + fg * 8;
    const int brow0 = (wn * NT * 16 + fr) * WS + fg * 8;
    for (int ch = 0; ch < nchunk; ch++) {
        if (!RESIDENT) {
            if (ch > 0) __syncthreads();      // everyone is done reading the previous chunk
            IFH_W_COMMIT()
        }
        __syncthreads();
        if (!RESIDENT && ch + 1 < nchunk) IFH_W_PREFETCH((ch + 1) * KC)
        const int ksteps = KW / 32;
        for (int ks = 0; ks < ksteps; ks++) {
            const int k = ch * KC + ks * 32;
            const int tap = k / CIN, ci = k - tap * CIN;
            const int aoff = arow0 + tap * p.dil * XS + ci;
            bf16x8_t fa[NT], fb[MT];
#pragma unroll
            for (int i = 0; i < NT; i++) fa[i] = *reinterpret_cast<const bf16x8_t *>(&Ws[brow0 + i * 16 * WS + ks * 32]);
#pragma unroll
            for (int j = 0; j < MT; j++) fb[j] = *reinterpret_cast<const bf16x8_t *>(&Xs[aoff + j * 16 * XS]);
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int j = 0; j < MT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

#undef IFH_W_PREFETCH
#undef IFH_W_COMMIT
#undef IFH_W1
#undef IFH_C1
    if (p.store16) {
        // Full-row stores: the rounded tile goes through LDS (over the dead input tile) so that every
        // global store is 16 bytes per lane and a wave writes whole contiguous rows.  The per-lane layout of
        // the MFMA result (4 channels x 16 different rows) stored directly costs about as much as the rest
        // of the kernel (8-byte pieces of 16 rows per instruction).
        constexpr int OS = BN + 8;
        __syncthreads();
        uint16_t *Os = Xs;
#pragma unroll
        for (int j = 0; j < MT; j++) {
            const int tl = (wm * MT + j) * 16 + fr;
            if (t0 + tl >= p.T_out) continue;
            const int m = b * p.T_out + t0 + tl;
#pragma unroll
            for (int i = 0; i < NT; i++) {
                const int n = (wn * NT + i) * 16 + 4 * fg;
                const uint2 pk = igemm_store4_fast<true, false, true>(p, m, n, acc[i][j], dynv, rpre[i][j], bpre[i]);
                *reinterpret_cast<uint2 *>(&Os[tl * OS + n]) = pk;
            }
        }
        __syncthreads();
        constexpr int VPR = BN / 8;
        uint16_t *outp = reinterpret_cast<uint16_t *>(p.out);
#pragma unroll 4
        for (int v = gt; v < BM * VPR; v += 256) {
            const int row = v / VPR, c = (v - row * VPR) * 8;
            if (live && t0 + row < p.T_out) {
                const EpiRow e = epi_row(p, b * p.T_out + t0 + row, 0, dynv);
                *reinterpret_cast<uint4 *>(outp + e.obase + c) = *reinterpret_cast<const uint4 *>(&Os[row * OS + c]);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < MT; j++) {
        const int t = t0 + (wm * MT + j) * 16 + fr;
        if (t >= p.T_out || !live) continue;
        const int m = b * p.T_out + t;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int n = (wn * NT + i) * 16 + 4 * fg;
            (void)igemm_store4_fast<true, true, true>(p, m, n, acc[i][j], dynv, rpre[i][j], bpre[i]);
        }
    }
}

template <int CIN, int WGM, int MT, int NT, bool RESIDENT, int KC = 64, int EPB = 1>
static bool launch_direct(const IgemmParams &p, hipStream_t st)
{
    constexpr int WGN = 4 / WGM;
    constexpr int BM = WGM * MT * 16, BN = WGN * NT * 16;
    constexpr int XS = CIN + 8;
    const int R = BM + (p.taps - 1) * p.dil;
    const int KW = RESIDENT ? p.K : KC;
    if (!RESIDENT && p.K % KC != 0) return false;
    static_assert(EPB == 1 || BM * (BN + 8) <= BM * XS, "the store16 staging of a group lies inside its own input tile");
    const size_t bytes = ((size_t)EPB * ((R * XS + 7) & ~7) + (size_t)BN * (KW + 8)) * sizeof(uint16_t);
    if (bytes > 160 * 1024) return false;
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (bytes > 64 * 1024 && attr_once.needed(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)k_conv_direct<CIN, WGM, MT, NT, RESIDENT, KC, EPB>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return false;
        attr_once.done(attr_dev);
    }
    dim3 grid((p.T_out + BM - 1) / BM, (p.nbatch + EPB - 1) / EPB);
    hipLaunchKernelGGL((k_conv_direct<CIN, WGM, MT, NT, RESIDENT, KC, EPB>), grid, dim3(256 * EPB), bytes, st, p);
    return true;
}

bool try_launch_conv_direct(const IgemmParams &p_, bool pre, hipStream_t st)
{
    IgemmParams p = p_;
    if (p.dyn && p.dyn_stride != 0) return false;          // per-row dynamic offsets: the generic kernels (nn.hip)
    constexpr bool no16 = false;      // fixed by measurement (profiles/NOTES.md)
    p.store16 = !no16 && !p.out_f32 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && p.ldc % 8 == 0 &&
                p.out_bstride % 8 == 0 && ((int64_t)p.ooff * p.ldc) % 8 == 0 && ((int64_t)p.ostride * p.ldc) % 8 == 0 &&
                (p.dyn == nullptr || ((int64_t)p.dyn_ooff_mul * p.ldc) % 8 == 0);
    (void)pre;
    constexpr int mask = 15;   // fixed by measurement (profiles/NOTES.md)
    if (!((p.Cin == 256 && (mask & 1)) || (p.Cin == 128 && (mask & 2)) || (p.Cin == 64 && (mask & 4)) || (p.Cin == 32 && (mask & 8))))
        return false;
    if (p.stride != 1 || p.taps < 2 || p.N != p.Cin || p.n_split != 0 || !p.fast_epi) return false;
    if (p.T_out != p.T_in + 2 * p.pad - (p.taps - 1) * p.dil) return false;
    if (p.nbatch >= 65536) return false;
    constexpr int epb = 2;       // fixed by measurement (profiles/NOTES.md)
    constexpr int kc = 64;      // fixed by measurement (profiles/NOTES.md)
    switch (p.Cin) {
    case 256:                                                                        // BM 48  x BN 256
        if (p.T_out < 32) return false;
        // Two chunks per block (shared W chunks: half the L2 -> CU weight traffic, the bound of this shape) when the
        // block count still fills the 256 CUs evenly, or when one 48-row block alone already takes more than half a
        // CU's LDS (11 taps x dilation 5).  Measured per 9-conv set: 1024 chunks 778 -> 655 us; 768 chunks
        // (384 blocks = 1.5 per CU) no gain except the 11x5 conv, 119 -> 94 us.
        if (epb == 2 && p.T_out <= 48 && p.nbatch >= 512) {
            const int half = (p.nbatch + 1) / 2;
            const bool lds_bound = (48 + (p.taps - 1) * p.dil) * (256 + 8) * 2 + 256 * (64 + 8) * 2 > 80 * 1024;
            if ((half % 256 == 0 || half >= 1024 || lds_bound) && launch_direct<256, 1, 3, 4, false, 64, 2>(p, st)) return true;
        }
        return kc == 128 ? launch_direct<256, 1, 3, 4, false, 128>(p, st) : launch_direct<256, 1, 3, 4, false, 64>(p, st);
    case 128:                                                                        // BM 192 x BN 128
        if (p.T_out < 64) return false;
        return kc == 128 ? launch_direct<128, 2, 6, 4, false, 128>(p, st) : launch_direct<128, 2, 6, 4, false, 64>(p, st);
    case 64: return p.T_out >= 128 && launch_direct<64, 2, 8, 2, false>(p, st);      // BM 256 x BN 64, streamed W (3 blocks/CU)
    case 32: return p.T_out >= 128 && launch_direct<32, 4, 4, 2, true>(p, st);       // BM 256 x BN 32
    default: return false;
    }
}

}  // namespace ifh
