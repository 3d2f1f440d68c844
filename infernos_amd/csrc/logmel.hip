// logmel.hip -- Whisper log-mel front end for gfx950.
//
// Replaces Cluster/InfernSTTWorker.py:114 (WhisperProcessor -> WhisperFeatureExtractor;
// transformers feature_extraction_whisper.py:135-168): zero-pad to 30 s, reflect-centred
// STFT (n_fft 400, hop 160, periodic Hann), power, slaney mel filterbank, log10,
// per-utterance max-8 clamp, (x+4)/4.
//
// Kernel 1 (k_logmel_dft): one block = 128 frames of one utterance, 7 waves.  The audio
// tile (20.7k samples) is staged once in LDS (index-padded so frame-strided reads are
// conflict-free); the 400-point real DFT is done as an exact-f32 MFMA contraction
// (v_mfma_f32_32x32x2_f32) against a window-folded cos/sin table, using the even/odd fold
// e[n]=x[n]+x[400-n], o[n]=x[n]-x[400-n] that halves K to 201.  Wave w owns bins
// 32w..32w+31 for both cos and sin, so power is formed in registers; it is then parked in
// LDS (overlaying the audio tile) for the sparse mel projection + log10 and a per-utterance
// atomic max.  Kernel 2 applies the clamp/scale (needs the completed max).
// Algorithmic bytes per 30 s window: 480000*4 read + 80*3000*4 written = 2.88 MB.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace ifh {

constexpr int kNfft = 400, kHop = 160, kBins = 201, kFrames = 3000, kNsamp = 480000;
constexpr int kFT = 128;            // frames per block
constexpr int kTabCols = 448;       // 224 cos bins | 224 sin bins
constexpr int kTabRows = 202;       // folded n = 0..201 (201 is a zero row)
constexpr int kTile = (kFT - 1) * kHop + kNfft + 1;      // 20721 samples (+1: index 400 of the last frame)
constexpr int kTileLds = kTile + kTile / kHop + 2;       // padded index space
constexpr int kPLds = kBins * kFT;                       // power tile overlay
constexpr int kLdsFloats = (kTileLds > kPLds) ? kTileLds : kPLds;

__device__ __forceinline__ int pad_idx(int m) { return m + m / kHop; }

__global__ __launch_bounds__(448) void k_logmel_dft(const float *__restrict__ audio, int64_t stride,
                                                    const int32_t *__restrict__ lens,
                                                    const float *__restrict__ tab,
                                                    const int32_t *__restrict__ mel_lo,
                                                    const int32_t *__restrict__ mel_cnt,
                                                    const int32_t *__restrict__ mel_off,
                                                    const float *__restrict__ mel_w, int n_mel,
                                                    float *__restrict__ raw /* [B][n_mel][3000] */,
                                                    int *__restrict__ gmax)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * kFT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..6, bin tile
    int len = lens ? lens[b] : kNsamp;
    len = len > kNsamp ? kNsamp : (len < 0 ? 0 : len);
    const float *x = audio + (int64_t)b * stride;
    float *rawb = raw + (int64_t)b * n_mel * kFrames;
    const int a0 = f0 * kHop - kNfft / 2;  // absolute index of tile sample 0 in the unpadded signal

    float lmax = -INFINITY;
    const int a_last = a0 + kTile - 1;
    const bool tail_clear = (a_last < kNsamp) || (2 * (kNsamp - 1) - a_last >= len);
    if (a0 >= len && a0 >= 0 && tail_clear) {
        // every sample this block would read is zero: power 0 -> log10(clamp 1e-10)
        const float v0 = log10f(fmaxf(0.0f * (float)len, 1e-10f));
        for (int idx = tid; idx < n_mel * kFT; idx += 448) {
            const int m = idx >> 7, f = f0 + (idx & (kFT - 1));
            if (f < kFrames) rawb[(int64_t)m * kFrames + f] = v0;
        }
        if (tid == 0) atomicMax(gmax + b, float_to_ordered(v0));
        return;
    }

    // ---- stage the audio tile (reflect at the 30 s boundaries, zero beyond len)
    for (int m = tid; m < kTile; m += 448) {
        int a = a0 + m;
        if (a < 0) a = -a;
        if (a >= kNsamp) a = 2 * (kNsamp - 1) - a;
        float v = 0.0f;
        if (a >= 0 && a < len) v = x[a];
        lds[pad_idx(m)] = v;
    }
    __syncthreads();

    // ---- folded DFT on the f32 matrix pipe
    const int i = lane & 31, kh = lane >> 5;
    f32x16 accc[4], accs[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            accc[q][r] = 0.0f;
            accs[q][r] = 0.0f;
        }
    }
    const float *tc = tab + 32 * w + i;
    const float *ts = tab + 224 + 32 * w + i;
    const int fb = 161 * i;  // pad_idx(f*160) for f = i (+ 161*32*q per frame tile)
    // The table operands come from L2 (362 KB table, shared by every block).  With one block per CU
    // nothing else hides that latency, so the loop is software-pipelined by hand: the operands of the
    // next PF steps are in flight (registers) while the current PF steps feed the matrix pipe.
    constexpr int PF = 8;
    constexpr int NSTEP = kTabRows / 2;          // 101
    float pc[PF], ps[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) {
        const int n = 2 * u + kh;
        pc[u] = tc[n * kTabCols];
        ps[u] = ts[n * kTabCols];
    }
    for (int s0 = 0; s0 < NSTEP; s0 += PF) {
        float cc[PF], cs[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            cc[u] = pc[u];
            cs[u] = ps[u];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {          // prefetch the following group (clamped: row 201 is all zeros)
            int n = 2 * (s0 + PF + u) + kh;
            n = n < kTabRows ? n : kTabRows - 1;
            pc[u] = tc[n * kTabCols];
            ps[u] = ts[n * kTabCols];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int step = s0 + u;
            if (step < NSTEP) {
                const int n = 2 * step + kh;
                const int n2 = kNfft - n;
                const int o1 = n + (n >= 160) + (n >= 320);
                const int o2 = n2 + (n2 >= 160) + (n2 >= 320);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int base = fb + 161 * 32 * q;
                    const float xa = lds[base + o1];
                    const float xb = lds[base + o2];
                    const float e = xa + xb, o = xa - xb;
                    accc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(cc[u], e, accc[q], 0, 0, 0);
                    accs[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(cs[u], o, accs[q], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();  // everyone is done with the audio tile; overlay the power tile

    // D layout: col = lane&31 (frame), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (bin in tile)
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int bin = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (bin < kBins) {
                const float c = accc[q][r], s = accs[q][r];
                lds[bin * kFT + 32 * q + i] = __fmaf_rn(c, c, s * s);
            }
        }
    }
    __syncthreads();

    // ---- sparse mel projection + log10; (mel, frame) pairs, frame fastest
    for (int idx = tid; idx < n_mel * kFT; idx += 448) {
        const int m = idx >> 7, fl = idx & (kFT - 1);
        const int lo = mel_lo[m], cnt = mel_cnt[m];
        const float *wv = mel_w + mel_off[m];
        float acc = 0.0f;
        for (int c = 0; c < cnt; c++) acc = __fmaf_rn(wv[c], lds[(lo + c) * kFT + fl], acc);
        const float v = log10f(fmaxf(acc, 1e-10f));
        const int f = f0 + fl;
        if (f < kFrames) {
            rawb[(int64_t)m * kFrames + f] = v;
            lmax = fmaxf(lmax, v);
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -INFINITY) atomicMax(gmax + b, float_to_ordered(lmax));
}


// ---------------------------------------------------------------------------------------------
// k_logmel_dft2: the same folded DFT on the bf16 matrix pipe with exact 3-way operand splitting.
// An f32 value v is written v = v0 + v1 + v2 with v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1)
// (24 mantissa bits in all).  table*audio is then the sum of the six products
//   t0a0 + t0a1 + t1a0 + t1a1 + t0a2 + t2a0           (dropped terms are <= 2^-24 relative)
// each an exact bf16 product accumulated in f32 by v_mfma_f32_32x32x16_bf16: f32-grade accuracy at
// 6/16 of the f32-MFMA cost (2.7x fewer matrix-pipe cycles).  Geometry: block = 128 frames, 4 waves,
// wave w owns frames 32w..32w+31 and ALL 14 bin tiles (7 cos + 7 sin, 224 accumulator registers), so
// its audio fragments are formed and split once per k-step; the table slice of a k-step (448 rows x
// 16 k x 3 splits, 48-byte padded rows: conflict-free ds_read_b128) is staged in LDS for all waves
// through a register prefetch.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int kK2 = 208;                    // folded n padded to 13 k-steps of 16
constexpr int kKS2 = kK2 / 16;
constexpr int kRowB = 48;                   // LDS bytes per table row (32 data + 16 pad)
constexpr int kTabSliceB = 3 * 448 * kRowB; // 64512
constexpr int kTab2LdsOff = ((kTileLds * 4 + 15) / 16) * 16;
constexpr int kLds2Bytes = (kTab2LdsOff + kTabSliceB > kPLds * 4) ? (kTab2LdsOff + kTabSliceB) : (kPLds * 4);

__device__ __forceinline__ void split3(float v, uint16_t &a, uint16_t &b, uint16_t &c)
{
    a = f32_to_bf16(v);
    const float r1 = v - bf16_to_f32(a);
    b = f32_to_bf16(r1);
    const float r2 = r1 - bf16_to_f32(b);
    c = f32_to_bf16(r2);
}

__global__ __launch_bounds__(256) void k_logmel_dft2(const float *__restrict__ audio, int64_t stride,
                                                     const int32_t *__restrict__ lens,
                                                     const uint16_t *__restrict__ tab2 /* [13][3][448][16] bf16 */,
                                                     const int32_t *__restrict__ mel2 /* packed filters, see create */,
                                                     int mel2_words, int n_mel,
                                                     float *__restrict__ raw, int *__restrict__ gmax,
                                                     unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const long long tp0 = clock64();
    unsigned char *ldsb = reinterpret_cast<unsigned char *>(lds);
    unsigned char *tabl = ldsb + kTab2LdsOff;
    int32_t *mell = reinterpret_cast<int32_t *>(ldsb + kLds2Bytes);
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * kFT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    int len = lens ? lens[b] : kNsamp;
    len = len > kNsamp ? kNsamp : (len < 0 ? 0 : len);
    const float *x = audio + (int64_t)b * stride;
    float *rawb = raw + (int64_t)b * n_mel * kFrames;
    const int a0 = f0 * kHop - kNfft / 2;
    float lmax = -INFINITY;
    const int a_last = a0 + kTile - 1;
    const bool tail_clear = (a_last < kNsamp) || (2 * (kNsamp - 1) - a_last >= len);
    if (len == 0 || (a0 >= len && a0 >= 0 && tail_clear)) {
        const float v0 = log10f(fmaxf(0.0f * (float)len, 1e-10f));
        for (int idx = tid; idx < n_mel * kFT; idx += 256) {
            const int m = idx >> 7, f = f0 + (idx & (kFT - 1));
            if (f < kFrames) rawb[(int64_t)m * kFrames + f] = v0;
        }
        if (tid == 0) atomicMax(gmax + b, float_to_ordered(v0));
        return;
    }
    // ---- stage the audio tile with every load of the thread in flight at once (one wave per SIMD:
    // nothing else hides the HBM latency).  Interior, 16-byte aligned tiles take the float4 path.
    const bool interior = a0 >= 0 && a_last < len && ((reinterpret_cast<uintptr_t>(x + a0) & 15) == 0);
    if (interior) {
        constexpr int NV4 = kTile / 4;          // 5180 float4 (+1 scalar tail)
        constexpr int NI = (NV4 + 255) / 256;   // 21
        const float4 *x4 = reinterpret_cast<const float4 *>(x + a0);
        float4 v[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int q = tid + 256 * i;
            v[i] = x4[q < NV4 ? q : NV4 - 1];
        }
        if (tid == 0) lds[pad_idx(kTile - 1)] = x[a0 + kTile - 1];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int q = tid + 256 * i;
            if (q < NV4) {
                const int pi = pad_idx(4 * q);   // 4q..4q+3 never straddle a multiple of 160
                lds[pi] = v[i].x;
                lds[pi + 1] = v[i].y;
                lds[pi + 2] = v[i].z;
                lds[pi + 3] = v[i].w;
            }
        }
    } else {
        constexpr int NI = (kTile + 255) / 256;  // 81
        float v[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            int a = a0 + tid + 256 * i;
            if (a < 0) a = -a;
            if (a >= kNsamp) a = 2 * (kNsamp - 1) - a;
            const bool ok = a >= 0 && a < len;
            const float t_ = x[ok ? a : 0];
            v[i] = ok ? t_ : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int m = tid + 256 * i;
            if (m < kTile) lds[pad_idx(m)] = v[i];
        }
    }
    for (int i = tid; i < mel2_words; i += 256) mell[i] = mel2[i];
    // table slice prefetch registers (named: arrays tend to end up in scratch)
    constexpr int NV = 2688;                  // 16-byte vectors per slice (3*448*2)
    uint4 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10;
    p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = p8 = p9 = p10 = make_uint4(0, 0, 0, 0);
#define LM_PF1(I, REG, KS)                                                                              \
    {                                                                                                   \
        const int v = tid + 256 * I;                                                                    \
        if (v < NV) REG = reinterpret_cast<const uint4 *>(tab2 + (int64_t)(KS) * (NV * 8))[v];          \
    }
#define LM_PREFETCH(KS)                                                                                 \
    LM_PF1(0, p0, KS) LM_PF1(1, p1, KS) LM_PF1(2, p2, KS) LM_PF1(3, p3, KS) LM_PF1(4, p4, KS) LM_PF1(5, p5, KS) \
    LM_PF1(6, p6, KS) LM_PF1(7, p7, KS) LM_PF1(8, p8, KS) LM_PF1(9, p9, KS) LM_PF1(10, p10, KS)
#define LM_CM1(I, REG)                                                                                  \
    {                                                                                                   \
        const int v = tid + 256 * I;                                                                    \
        if (v < NV) *reinterpret_cast<uint4 *>(tabl + (v >> 1) * kRowB + (v & 1) * 16) = REG;           \
    }
#define LM_COMMIT()                                                                                     \
    LM_CM1(0, p0) LM_CM1(1, p1) LM_CM1(2, p2) LM_CM1(3, p3) LM_CM1(4, p4) LM_CM1(5, p5) LM_CM1(6, p6)   \
    LM_CM1(7, p7) LM_CM1(8, p8) LM_CM1(9, p9) LM_CM1(10, p10)
    LM_PREFETCH(0)
    __syncthreads();                           // audio tile staged
    const long long tp1 = clock64();

    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[14];
#pragma unroll
    for (int t = 0; t < 14; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[t][q] = 0.0f;
    const int fbase = 161 * (32 * w + r);      // pad_idx(frame*160)
    // audio fragments: e = x[n] + x[400-n], o = x[n] - x[400-n], each split 3 ways.  The fragments of
    // k-step ks+1 are formed between the MFMAs of k-step ks.
    uint16_t e0[8], e1[8], e2[8], o0[8], o1[8], o2[8];
#define LM_ELEM(KS, J)                                                                                  \
    {                                                                                                   \
        const int n = (KS) * 16 + 8 * h + (J);     /* <= 223; table rows > 200 are zero */             \
        const int n2 = kNfft - n;                                                                       \
        const float xa = lds[fbase + n + (n >= 160)];                                                   \
        const float xb = lds[fbase + n2 + (n2 >= 160) + (n2 >= 320)];                                   \
        split3(xa + xb, e0[J], e1[J], e2[J]);                                                           \
        split3(xa - xb, o0[J], o1[J], o2[J]);                                                           \
    }
#define LM_PACK(DST, SRC)                                                                               \
    {                                                                                                   \
        uint4 t_;                                                                                       \
        t_.x = (uint32_t)SRC[0] | ((uint32_t)SRC[1] << 16);                                             \
        t_.y = (uint32_t)SRC[2] | ((uint32_t)SRC[3] << 16);                                             \
        t_.z = (uint32_t)SRC[4] | ((uint32_t)SRC[5] << 16);                                             \
        t_.w = (uint32_t)SRC[6] | ((uint32_t)SRC[7] << 16);                                             \
        DST = __builtin_bit_cast(bf16x8_t, t_);                                                         \
    }
    // Fragment registers are ping-ponged (A: even k-steps, B: odd): the set being re-packed was last read by
    // MFMAs two barriers ago.  Re-packing the set in use right behind its last MFMA gave run-to-run
    // different results in lanes 16..31/48..63 of the B operand (gfx950 reads the 4-VGPR operand of the
    // K=16 MFMA over more than one beat; the compiler does not guard that write-after-read).
    bf16x8_t EA0, EA1, EA2, OA0, OA1, OA2, EB0, EB1, EB2, OB0, OB1, OB2;
    LM_ELEM(0, 0) LM_ELEM(0, 1) LM_ELEM(0, 2) LM_ELEM(0, 3) LM_ELEM(0, 4) LM_ELEM(0, 5) LM_ELEM(0, 6) LM_ELEM(0, 7)
    LM_PACK(EA0, e0) LM_PACK(EA1, e1) LM_PACK(EA2, e2) LM_PACK(OA0, o0) LM_PACK(OA1, o1) LM_PACK(OA2, o2)
#define LM_KSTEP(KS, CE0, CE1, CE2, CO0, CO1, CO2, NE0, NE1, NE2, NO0, NO1, NO2)                        \
    {                                                                                                   \
        if ((KS) > 0) __syncthreads(); /* previous slice fully consumed */                              \
        LM_COMMIT()                                                                                     \
        __syncthreads();                                                                                \
        if ((KS) + 1 < kKS2) LM_PREFETCH((KS) + 1)                                                      \
        _Pragma("unroll") for (int t = 0; t < 14; t++)                                                  \
        {                                                                                               \
            const unsigned char *rowp = tabl + (32 * t + r) * kRowB + h * 16;                           \
            const bf16x8_t T0 = *reinterpret_cast<const bf16x8_t *>(rowp);                              \
            const bf16x8_t T1 = *reinterpret_cast<const bf16x8_t *>(rowp + 448 * kRowB);                \
            const bf16x8_t T2 = *reinterpret_cast<const bf16x8_t *>(rowp + 2 * 448 * kRowB);            \
            const bf16x8_t B0 = t < 7 ? CE0 : CO0, B1 = t < 7 ? CE1 : CO1, B2 = t < 7 ? CE2 : CO2;      \
            f32x16 a = acc[t];                                                                          \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T2, B0, a, 0, 0, 0); /* small terms first */    \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T0, B2, a, 0, 0, 0);                            \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T1, B1, a, 0, 0, 0);                            \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T1, B0, a, 0, 0, 0);                            \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T0, B1, a, 0, 0, 0);                            \
            a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T0, B0, a, 0, 0, 0);                            \
            acc[t] = a;                                                                                 \
            if (t < 8) LM_ELEM((KS) + 1, t)                                                             \
        }                                                                                               \
        LM_PACK(NE0, e0) LM_PACK(NE1, e1) LM_PACK(NE2, e2) LM_PACK(NO0, o0) LM_PACK(NO1, o1) LM_PACK(NO2, o2) \
    }
    for (int ks = 0; ks + 1 < kKS2; ks += 2) {
        LM_KSTEP(ks, EA0, EA1, EA2, OA0, OA1, OA2, EB0, EB1, EB2, OB0, OB1, OB2)
        LM_KSTEP(ks + 1, EB0, EB1, EB2, OB0, OB1, OB2, EA0, EA1, EA2, OA0, OA1, OA2)
    }
    static_assert(kKS2 % 2 == 1, "tail k-step uses the A set");
    LM_KSTEP(kKS2 - 1, EA0, EA1, EA2, OA0, OA1, OA2, EB0, EB1, EB2, OB0, OB1, OB2)
#undef LM_KSTEP
#undef LM_PF1
#undef LM_PREFETCH
#undef LM_CM1
#undef LM_COMMIT
#undef LM_PACK
#undef LM_ELEM
    __syncthreads();  // audio tile and table slice are dead: overlay the power tile
    const long long tp2 = clock64();
#pragma unroll
    for (int t = 0; t < 7; t++) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int bin = 32 * t + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (bin < kBins) {
                const float c = acc[t][q], sn = acc[7 + t][q];
                lds[bin * kFT + 32 * w + r] = __fmaf_rn(c, c, sn * sn);
            }
        }
    }
    __syncthreads();
    // ---- sparse mel projection: thread = (frame, mel parity), four filters in flight per iteration
    // (filter weights padded to groups of 4 in LDS; group index past a filter's end reads zero weights)
    {
        const int fl = tid & (kFT - 1);
        const int f = f0 + fl;
        const float4 *w4 = reinterpret_cast<const float4 *>(mell + 384);
        const int zero_g = (mel2_words - 384) / 4 - 1;   // the last group is all-zero
        for (int mb = tid >> 7; mb < n_mel; mb += 8) {
            int lo[4], c4[4], o4[4];
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            int gmaxn = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int m = mb + 2 * u;
                lo[u] = mell[m];
                c4[u] = mell[128 + m];
                o4[u] = mell[256 + m];
                gmaxn = max(gmaxn, c4[u]);
            }
            for (int g = 0; g < gmaxn; g++) {
                float4 wv[4];
                float q[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool on = g < c4[u];
                    wv[u] = w4[on ? o4[u] + g : zero_g];
                    const int k = on ? lo[u] + 4 * g : 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) q[u][e] = lds[min(k + e, kBins - 1) * kFT + fl];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a[u] = __fmaf_rn(wv[u].x, q[u][0], a[u]);
                    a[u] = __fmaf_rn(wv[u].y, q[u][1], a[u]);
                    a[u] = __fmaf_rn(wv[u].z, q[u][2], a[u]);
                    a[u] = __fmaf_rn(wv[u].w, q[u][3], a[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float v = log10f(fmaxf(a[u], 1e-10f));
                if (f < kFrames) {
                    rawb[(int64_t)(mb + 2 * u) * kFrames + f] = v;
                    lmax = fmaxf(lmax, v);
                }
            }
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -INFINITY) atomicMax(gmax + b, float_to_ordered(lmax));
    if (prof && tid == 0) {
        const long long tp3 = clock64();
        atomicAdd(prof + 0, (unsigned long long)(tp1 - tp0));
        atomicAdd(prof + 1, (unsigned long long)(tp2 - tp1));
        atomicAdd(prof + 2, (unsigned long long)(tp3 - tp2));
        atomicAdd(prof + 3, 1ull);
    }
}


// ---------------------------------------------------------------------------------------------
// k_logmel_fft: the power spectrum through a real FFT instead of a DFT-as-GEMM (which cannot reach the HBM roof: its
// matrix work alone is >= 83 us per 64 windows).  Real 400-point transform via a complex 200-point one, z[n] = x[2n] +
// i x[2n+1], 200 = 25 x 8:
//   phase A  thread (frame f, n2 < 8): register DFT-25 (5 x 5) of z[8 n1 + n2], twiddle W200^(n2 k1) -> Y[f][k1][n2] (LDS)
//   phase B  13 tasks per frame (k1 = 0; pairs k1 / 25-k1): the DFT-8s give Z[k1 + 25 k2]; bins k and 200-k come out
//            together as |A +- B|^2 with A = (Z[k] + conj Z[200-k])/2, B = -(i/2) W400^k (Z[k] - conj Z[200-k])
//   then the sparse mel projection + log10 + per-window max as before.
// Block = 32 frames x 8 threads; LDS = 25.7 KB (audio tile, later the power tile) + 51.2 KB (Y): 2 blocks per CU.
// Audio layout: sample m at m + 16*(m/160), so a frame's (even, odd) pairs are aligned 8-byte reads and the 8 x 4
// (frame, n2) lanes of a half-wave hit 64 distinct banks.
// ---------------------------------------------------------------------------------------------
constexpr int kFB = 32;                                 // frames per block
constexpr int kTileF = (kFB - 1) * kHop + kNfft;        // 5360 samples
constexpr int kAudF = kTileF + 16 * ((kTileF - 1) / kHop) + 16;   // padded audio floats (5904)
constexpr int kPS = 201;                                // power tile row stride (odd: conflict-free over frames)
constexpr int kReg0F = (kAudF > kFB * kPS) ? kAudF : kFB * kPS;   // audio tile / power tile overlay
constexpr int kYF = kFB * 25 * 8 * 2;                   // Y buffer floats
constexpr int kLdsFftBytes = (kReg0F + kYF) * 4;

typedef float f2v __attribute__((ext_vector_type(2)));     // native 2-vector: lets the compiler use v_pk_add/mul/fma_f32

struct FftConst {
    f2v w25[5][5];      // W25^(b c)
};
__constant__ FftConst c_fft;

__device__ __forceinline__ f2v mk2(float a, float b) { return (f2v){a, b}; }
__device__ __forceinline__ f2v cmul(f2v a, f2v b) { return a.x * b + a.y * mk2(-b.y, b.x); }     // packed: (ax bx - ay by, ax by + ay bx)
__device__ __forceinline__ f2v cadd(f2v a, f2v b) { return a + b; }
__device__ __forceinline__ f2v csub(f2v a, f2v b) { return a - b; }

// forward 5-point DFT (in place on 5 named values)
#define FFT5(V0, V1, V2, V3, V4)                                                                              \
    {                                                                                                         \
        const float c1 = 0.30901699437494745f, c2 = -0.80901699437494745f;                                    \
        const float s1 = 0.95105651629515353f, s2 = 0.58778525229247314f;                                     \
        const f2v t1 = cadd(V1, V4), t2 = cadd(V2, V3), t3 = csub(V1, V4), t4 = csub(V2, V3);              \
        const f2v m1 = V0 + c1 * t1 + c2 * t2;                                                                \
        const f2v m2 = V0 + c2 * t1 + c1 * t2;                                                                \
        const f2v u1 = s1 * t3 + s2 * t4;                                                                     \
        const f2v u2 = s2 * t3 - s1 * t4;                                                                     \
        V0 = V0 + t1 + t2;                                                                                    \
        V1 = mk2(m1.x + u1.y, m1.y - u1.x);  /* m1 - i u1 */                                          \
        V4 = mk2(m1.x - u1.y, m1.y + u1.x);  /* m1 + i u1 */                                          \
        V2 = mk2(m2.x + u2.y, m2.y - u2.x);                                                           \
        V3 = mk2(m2.x - u2.y, m2.y + u2.x);                                                           \
    }

// forward 8-point DFT of v[0..7] (natural order in, natural order out)
__device__ __forceinline__ void fft8(f2v (&v)[8])
{
    const float h = 0.70710678118654752f;
    // stage 1: pairs (j, j+4)
    f2v a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    f2v a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    f2v a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    f2v a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // twiddle the odd half by W8^j: 1, (1-i)/sqrt2, -i, (-1-i)/sqrt2
    a5 = mk2(h * (a5.x + a5.y), h * (a5.y - a5.x));
    a6 = mk2(a6.y, -a6.x);
    a7 = mk2(h * (a7.y - a7.x), -h * (a7.x + a7.y));
    // two 4-point DFTs: even outputs from a0..a3, odd outputs from a4..a7
    f2v b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    b3 = mk2(b3.y, -b3.x);
    v[0] = cadd(b0, b1); v[4] = csub(b0, b1); v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
    f2v d0 = cadd(a4, a6), d2 = csub(a4, a6), d1 = cadd(a5, a7), d3 = csub(a5, a7);
    d3 = mk2(d3.y, -d3.x);
    v[1] = cadd(d0, d1); v[5] = csub(d0, d1); v[3] = cadd(d2, d3); v[7] = csub(d2, d3);
}

__device__ __forceinline__ void power_pair(f2v zk, f2v zm, f2v tw, float &pk, float &pm)
{
    const float ax = 0.5f * (zk.x + zm.x), ay = 0.5f * (zk.y - zm.y);      // A = (Zk + conj Zm)/2
    const float dx = zk.x - zm.x, dy = zk.y + zm.y;                          // D = Zk - conj Zm
    const float ex = tw.x * dx - tw.y * dy, ey = tw.x * dy + tw.y * dx;      // E = tw D
    const float bx = 0.5f * ey, by = -0.5f * ex;                             // B = -(i/2) E
    const float px = ax + bx, py = ay + by, qx = ax - bx, qy = ay - by;
    pk = __fmaf_rn(px, px, py * py);
    pm = __fmaf_rn(qx, qx, qy * qy);
}

__global__ __launch_bounds__(256) void k_logmel_fft(const float *__restrict__ audio, int64_t stride,
                                                    const int32_t *__restrict__ lens, const float *__restrict__ win,
                                                    const f2v *__restrict__ tw200 /* [8][25] */,
                                                    const f2v *__restrict__ tw400 /* [101] */,
                                                    const int32_t *__restrict__ mel2 /* packed filters, see create */,
                                                    int mel2_words, int n_mel, float *__restrict__ raw,
                                                    int *__restrict__ gmax, unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    long long tq0 = clock64();
    float *aud = lds;                       // later the power tile [kFB][kPS]
    f2v *Yb = reinterpret_cast<f2v *>(lds + kReg0F);
    f2v *tw4l = reinterpret_cast<f2v *>(lds + kReg0F + kYF);          // W400^k, k <= 100 (+1 pad)
    int32_t *mell = reinterpret_cast<int32_t *>(lds + kReg0F + kYF + 204);
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    int len = lens ? lens[b] : kNsamp;
    len = len > kNsamp ? kNsamp : (len < 0 ? 0 : len);
    const float *x = audio + (int64_t)b * stride;
    float *rawb = raw + (int64_t)b * n_mel * kFrames;
    float lmax = -INFINITY;
    for (int i = tid; i < mel2_words; i += 256) mell[i] = mel2[i];
    if (tid < 101) tw4l[tid] = tw400[tid];
    // Persistent over frame tiles (tile, tile + gridDim.x, ...): the next tile's audio is requested into registers
    // before this tile's transforms start, so the block never sits waiting for HBM between tiles (with two blocks per
    // CU the staging wait was half of a block's life).
    constexpr int NV4 = kTileF / 4;          // 1340 float4 per tile
    constexpr int NI4 = (NV4 + 255) / 256;   // 6
    constexpr int ntiles = (kFrames + kFB - 1) / kFB;
    static_assert(NI4 == 6, "six named prefetch registers");
    float4 pv0, pv1, pv2, pv3, pv4, pv5;     // named: the array form was spilled to scratch
    pv0 = pv1 = pv2 = pv3 = pv4 = pv5 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool have_pv = false;
#define LF_PV_LOAD(X4)                                                                                  \
    {                                                                                                  \
        pv0 = (X4)[tid];                                                                               \
        pv1 = (X4)[tid + 256];                                                                         \
        pv2 = (X4)[tid + 512];                                                                         \
        pv3 = (X4)[tid + 768];                                                                         \
        pv4 = (X4)[tid + 1024];                                                                        \
        pv5 = (X4)[tid + 1280 < NV4 ? tid + 1280 : NV4 - 1];                                           \
    }
#define LF_PV_STORE1(I, REG)                                                                           \
    {                                                                                                  \
        const int q = tid + 256 * I;                                                                   \
        if (q < NV4) *reinterpret_cast<float4 *>(&aud[4 * q + 16 * ((4 * q) / kHop)]) = REG;           \
    }
#define LF_TILE_GEOM(TILE)                                                                             \
    const int f0 = (TILE) * kFB;                                                                       \
    const int a0 = f0 * kHop - kNfft / 2;                                                              \
    const int a_last = a0 + kTileF - 1;                                                                \
    const bool tail_clear = (a_last < kNsamp) || (2 * (kNsamp - 1) - a_last >= len);                   \
    const bool zero_tile = len == 0 || (a0 >= len && a0 >= 0 && tail_clear);                           \
    const bool interior = a0 >= 0 && a_last < len && ((reinterpret_cast<uintptr_t>(x + a0) & 15) == 0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    tq0 = clock64();
    LF_TILE_GEOM(tile)
    if (zero_tile) {
        const float v0 = log10f(fmaxf(0.0f * (float)len, 1e-10f));
        for (int idx = tid; idx < n_mel * kFB; idx += 256) {
            const int m = idx / kFB, f = f0 + (idx - m * kFB);
            if (f < kFrames) rawb[(int64_t)m * kFrames + f] = v0;
        }
        lmax = fmaxf(lmax, v0);
        have_pv = false;
        continue;
    }
    __syncthreads();                         // the previous tile's power tile has been consumed
    // ---- stage the audio tile
    if (interior) {
        if (!have_pv) {
            const float4 *x4 = reinterpret_cast<const float4 *>(x + a0);
            LF_PV_LOAD(x4)
        }
        LF_PV_STORE1(0, pv0) LF_PV_STORE1(1, pv1) LF_PV_STORE1(2, pv2) LF_PV_STORE1(3, pv3) LF_PV_STORE1(4, pv4) LF_PV_STORE1(5, pv5)
    } else {
        constexpr int NI = (kTileF + 255) / 256;  // 21
        float v[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            int a = a0 + tid + 256 * i;
            if (a < 0) a = -a;
            if (a >= kNsamp) a = 2 * (kNsamp - 1) - a;
            const bool ok = a >= 0 && a < len;
            const float t_ = x[ok ? a : 0];
            v[i] = ok ? t_ : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int m = tid + 256 * i;
            if (m < kTileF) aud[m + 16 * (m / kHop)] = v[i];
        }
    }
    __syncthreads();
    const long long tq1 = clock64();
    const int f = tid >> 3, sub = tid & 7;
    // ---- phase A: DFT-25 over n1 of z[8 n1 + sub], n = 8 n1 + sub, samples 2n, 2n+1 of frame f
    {
        f2v v0, v1, v2, v3, v4, v5, v6, v7, v8, v9, v10, v11, v12, v13, v14, v15, v16, v17, v18, v19, v20, v21, v22, v23, v24;
#define LF_LOAD(V, N1)                                                                               \
    {                                                                                                \
        const int n = 8 * (N1) + sub;                                                                \
        const f2v xs = *reinterpret_cast<const f2v *>(&aud[176 * f + 2 * n + 16 * ((2 * n) / kHop)]); \
        const f2v wv = *reinterpret_cast<const f2v *>(&win[2 * n]);                            \
        V = mk2(xs.x * wv.x, xs.y * wv.y);                                                   \
    }
        LF_LOAD(v0, 0) LF_LOAD(v1, 1) LF_LOAD(v2, 2) LF_LOAD(v3, 3) LF_LOAD(v4, 4) LF_LOAD(v5, 5) LF_LOAD(v6, 6)
        LF_LOAD(v7, 7) LF_LOAD(v8, 8) LF_LOAD(v9, 9) LF_LOAD(v10, 10) LF_LOAD(v11, 11) LF_LOAD(v12, 12) LF_LOAD(v13, 13)
        LF_LOAD(v14, 14) LF_LOAD(v15, 15) LF_LOAD(v16, 16) LF_LOAD(v17, 17) LF_LOAD(v18, 18) LF_LOAD(v19, 19)
        LF_LOAD(v20, 20) LF_LOAD(v21, 21) LF_LOAD(v22, 22) LF_LOAD(v23, 23) LF_LOAD(v24, 24)
#undef LF_LOAD
        // n1 = 5a + b: for each b a DFT-5 over a (in place: slot 5c + b <- inner[b][c])
        FFT5(v0, v5, v10, v15, v20) FFT5(v1, v6, v11, v16, v21) FFT5(v2, v7, v12, v17, v22)
        FFT5(v3, v8, v13, v18, v23) FFT5(v4, v9, v14, v19, v24)
        // twiddle inner[b][c] by W25^(b c) (b, c >= 1)
#define LF_TW(V, B, C) V = cmul(V, c_fft.w25[B][C]);
        LF_TW(v6, 1, 1) LF_TW(v7, 2, 1) LF_TW(v8, 3, 1) LF_TW(v9, 4, 1)
        LF_TW(v11, 1, 2) LF_TW(v12, 2, 2) LF_TW(v13, 3, 2) LF_TW(v14, 4, 2)
        LF_TW(v16, 1, 3) LF_TW(v17, 2, 3) LF_TW(v18, 3, 3) LF_TW(v19, 4, 3)
        LF_TW(v21, 1, 4) LF_TW(v22, 2, 4) LF_TW(v23, 3, 4) LF_TW(v24, 4, 4)
#undef LF_TW
        // for each c a DFT-5 over b: slot 5c + d <- Y[k1 = c + 5 d]
        FFT5(v0, v1, v2, v3, v4) FFT5(v5, v6, v7, v8, v9) FFT5(v10, v11, v12, v13, v14)
        FFT5(v15, v16, v17, v18, v19) FFT5(v20, v21, v22, v23, v24)
        const f2v *tw = tw200 + sub * 25;
        f2v *yo = Yb + (f * 25) * 8 + sub;
#define LF_OUT(V, C, D) yo[((C) + 5 * (D)) * 8] = cmul(V, tw[(C) + 5 * (D)]);
        LF_OUT(v0, 0, 0) LF_OUT(v1, 0, 1) LF_OUT(v2, 0, 2) LF_OUT(v3, 0, 3) LF_OUT(v4, 0, 4)
        LF_OUT(v5, 1, 0) LF_OUT(v6, 1, 1) LF_OUT(v7, 1, 2) LF_OUT(v8, 1, 3) LF_OUT(v9, 1, 4)
        LF_OUT(v10, 2, 0) LF_OUT(v11, 2, 1) LF_OUT(v12, 2, 2) LF_OUT(v13, 2, 3) LF_OUT(v14, 2, 4)
        LF_OUT(v15, 3, 0) LF_OUT(v16, 3, 1) LF_OUT(v17, 3, 2) LF_OUT(v18, 3, 3) LF_OUT(v19, 3, 4)
        LF_OUT(v20, 4, 0) LF_OUT(v21, 4, 1) LF_OUT(v22, 4, 2) LF_OUT(v23, 4, 3) LF_OUT(v24, 4, 4)
#undef LF_OUT
    }
    have_pv = false;
    {   // request the next tile of this block (if it is an interior one) now that phase A's own global loads (window,
        // twiddles) have been consumed -- loads retire in order, so anything issued later would wait for these; phases B
        // and the mel projection read LDS only and cover the round trip
        const int nt = tile + gridDim.x;
        if (nt < ntiles) {
            const int nf0 = nt * kFB, na0 = nf0 * kHop - kNfft / 2, na_last = na0 + kTileF - 1;
            if (na0 >= 0 && na_last < len && ((reinterpret_cast<uintptr_t>(x + na0) & 15) == 0)) {
                const float4 *x4 = reinterpret_cast<const float4 *>(x + na0);
                LF_PV_LOAD(x4)
                have_pv = true;
            }
        }
    }
    __syncthreads();      // Y complete; the audio tile is dead: its place becomes the power tile
    const long long tq2 = clock64();
    // ---- phase B: tasks sub and sub + 8 (< 13) of frame f
    float *Pt = aud + f * kPS;
    for (int task = sub; task < 13; task += 8) {
        f2v za[8], zb[8];
        const f2v *ya = Yb + (f * 25 + task) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) za[j] = ya[j];
        fft8(za);                                     // Z[task + 25 k2]
        if (task == 0) {
            float pk, pm;
            power_pair(za[0], za[0], tw4l[0], pk, pm);
            Pt[0] = pk;
            Pt[200] = pm;
            power_pair(za[1], za[7], tw4l[25], pk, pm);
            Pt[25] = pk;
            Pt[175] = pm;
            power_pair(za[2], za[6], tw4l[50], pk, pm);
            Pt[50] = pk;
            Pt[150] = pm;
            power_pair(za[3], za[5], tw4l[75], pk, pm);
            Pt[75] = pk;
            Pt[125] = pm;
            power_pair(za[4], za[4], tw4l[100], pk, pm);
            Pt[100] = pk;
        } else {
            const int kp = 25 - task;
            const f2v *yb = Yb + (f * 25 + kp) * 8;
#pragma unroll
            for (int j = 0; j < 8; j++) zb[j] = yb[j];
            fft8(zb);                                 // Z[kp + 25 k2]
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                float pk, pm;
                int k = task + 25 * k2;               // <= 87: partner 200 - k = kp + 25 (7 - k2)
                power_pair(za[k2], zb[7 - k2], tw4l[k], pk, pm);
                Pt[k] = pk;
                Pt[200 - k] = pm;
                k = kp + 25 * k2;                     // <= 99: partner 200 - k = task + 25 (7 - k2)
                power_pair(zb[k2], za[7 - k2], tw4l[k], pk, pm);
                Pt[k] = pk;
                Pt[200 - k] = pm;
            }
        }
    }
    __syncthreads();
    const long long tq3 = clock64();
    // ---- sparse mel projection + log10: thread = (frame fl, mel m = mg + 8 u), four filters in flight; the packed
    // filter table (lo | groups | first group | weights padded to groups of 4, trailing zero group) sits in LDS
    {
        const int fl = tid & (kFB - 1), mg = tid >> 5;
        const float *pr = aud + fl * kPS;
        const int fr_ = f0 + fl;
        const float4 *w4 = reinterpret_cast<const float4 *>(mell + 384);
        const int zero_g = (mel2_words - 384) / 4 - 1;
        for (int mb = mg; mb < n_mel; mb += 32) {
            int lo[4], c4[4], o4[4];
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            int gmaxn = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int m = mb + 8 * u;
                const bool on = m < n_mel;
                lo[u] = on ? mell[m] : 0;
                c4[u] = on ? mell[128 + m] : 0;
                o4[u] = on ? mell[256 + m] : 0;
                gmaxn = max(gmaxn, c4[u]);
            }
            for (int g = 0; g < gmaxn; g++) {
                float4 wv[4];
                float q[4][4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool on = g < c4[u];
                    wv[u] = w4[on ? o4[u] + g : zero_g];
                    const int k = on ? lo[u] + 4 * g : 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) q[u][e] = pr[min(k + e, kBins - 1)];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a[u] = __fmaf_rn(wv[u].x, q[u][0], a[u]);
                    a[u] = __fmaf_rn(wv[u].y, q[u][1], a[u]);
                    a[u] = __fmaf_rn(wv[u].z, q[u][2], a[u]);
                    a[u] = __fmaf_rn(wv[u].w, q[u][3], a[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int m = mb + 8 * u;
                if (m < n_mel && fr_ < kFrames) {
                    const float v = log10f(fmaxf(a[u], 1e-10f));
                    rawb[(int64_t)m * kFrames + fr_] = v;
                    lmax = fmaxf(lmax, v);
                }
            }
        }
    }
    if (prof && tid == 0) {
        const long long tq4 = clock64();
        atomicAdd(prof + 0, (unsigned long long)(tq1 - tq0));
        atomicAdd(prof + 1, (unsigned long long)(tq2 - tq1));
        atomicAdd(prof + 2, (unsigned long long)(tq3 - tq2));
        atomicAdd(prof + 3, (unsigned long long)(tq4 - tq3));
        atomicAdd(prof + 4, 1ull);
    }
    }   // tile loop
#undef LF_TILE_GEOM
#undef LF_PV_LOAD
#undef LF_PV_STORE1
    lmax = wave_max(lmax);
    if (lane == 0 && lmax > -INFINITY) atomicMax(gmax + b, float_to_ordered(lmax));
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_logmel_finish(const float *__restrict__ raw, void *__restrict__ out,
                                                       const int *__restrict__ gmax, int per_utt)
{
    const int b = blockIdx.y;
    const float floorv = ordered_to_float(gmax[b]) - 8.0f;
    const float *src = raw + (int64_t)b * per_utt;
    const int n4 = per_utt >> 2;  // per_utt = n_mel*3000, a multiple of 4
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4 *>(src)[k];
        v.x = (fmaxf(v.x, floorv) + 4.0f) * 0.25f;
        v.y = (fmaxf(v.y, floorv) + 4.0f) * 0.25f;
        v.z = (fmaxf(v.z, floorv) + 4.0f) * 0.25f;
        v.w = (fmaxf(v.w, floorv) + 4.0f) * 0.25f;
        if (BF16) {
            uint2 p;
            p.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
            p.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
            reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(out) + (int64_t)b * per_utt)[k] = p;
        } else {
            reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + (int64_t)b * per_utt)[k] = v;
        }
    }
}

// The same clamp/scale fused into the consumer's layout change: raw f32 [B][n_mel][3000] -> bf16 [B][3000][n_mel], the
// channels-last input of Whisper's first convolution.  With this the normalised f32 plane is never written or re-read
// (k_logmel_finish moved 1.92 MB per window on top of the 2.88 MB the transform itself needs).
__global__ __launch_bounds__(256) void k_logmel_finish_transpose(const float *__restrict__ raw, uint16_t *__restrict__ out,
                                                                 const int *__restrict__ gmax, int n_mel, int frames)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float floorv = ordered_to_float(gmax[b]) - 8.0f;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;          // r: mel band, c: frame
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        float v = 0.0f;
        if (r < n_mel && c < frames) v = (fmaxf(raw[((int64_t)b * n_mel + r) * frames + c], floorv) + 4.0f) * 0.25f;
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < n_mel && c < frames) out[((int64_t)b * frames + c) * n_mel + r] = f32_to_bf16(tile[tx][k]);
    }
}

// transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney'), float64
static double hz_to_mel(double f)
{
    if (f >= 1000.0) return 15.0 + log(f / 1000.0) * (27.0 / log(6.4));
    return 3.0 * f / 200.0;
}
static double mel_to_hz(double m)
{
    if (m >= 15.0) return 1000.0 * exp((log(6.4) / 27.0) * (m - 15.0));
    return 200.0 * m / 3.0;
}

}  // namespace ifh

using namespace ifh;

struct ifh_logmel {
    int n_mel;
    std::vector<float> filters;  // [201][n_mel]
    float *d_tab = nullptr;
    int32_t *d_mel2 = nullptr;    // packed filters for k_logmel_dft2: lo[128] | groups[128] | first group[128] | weights (x4 padded)
    int mel2_words = 0;
    float *d_win = nullptr;       // k_logmel_fft: hann window [400], W200^(n2 k1) [8][25], W400^k [101]
    f2v *d_tw200 = nullptr, *d_tw400 = nullptr;
    uint16_t *d_tab2 = nullptr;   // [13][3][448][16] bf16 splits of the folded DFT table (k_logmel_dft2)
    int32_t *d_lo = nullptr, *d_cnt = nullptr, *d_off = nullptr;
    float *d_w = nullptr;
};

extern "C" int ifh_logmel_create(int n_mel, ifh_logmel_t *out)
{
    IFH_CHECK_ARG(out && (n_mel == 80 || n_mel == 128));
    ifh_logmel *h = new ifh_logmel();
    h->n_mel = n_mel;
    // ---- mel filters (float64 -> float32)
    std::vector<double> pts(n_mel + 2), fr(n_mel + 2);
    const double mlo = hz_to_mel(0.0), mhi = hz_to_mel(8000.0);
    for (int k = 0; k < n_mel + 2; k++) {
        pts[k] = mlo + (mhi - mlo) * (double)k / (double)(n_mel + 1);
        fr[k] = mel_to_hz(pts[k]);
    }
    h->filters.assign((size_t)kBins * n_mel, 0.0f);
    for (int k = 0; k < kBins; k++) {
        const double ff = 8000.0 * (double)k / (double)(kBins - 1);
        for (int m = 0; m < n_mel; m++) {
            const double down = -(fr[m] - ff) / (fr[m + 1] - fr[m]);
            const double up = (fr[m + 2] - ff) / (fr[m + 2] - fr[m + 1]);
            double v = down < up ? down : up;
            if (v < 0.0) v = 0.0;
            v *= 2.0 / (fr[m + 2] - fr[m]);
            h->filters[(size_t)k * n_mel + m] = (float)v;
        }
    }
    std::vector<int32_t> lo(n_mel), cnt(n_mel), off(n_mel);
    std::vector<float> wts;
    for (int m = 0; m < n_mel; m++) {
        int first = -1, last = -1;
        for (int k = 0; k < kBins; k++)
            if (h->filters[(size_t)k * n_mel + m] != 0.0f) {
                if (first < 0) first = k;
                last = k;
            }
        lo[m] = first < 0 ? 0 : first;
        cnt[m] = first < 0 ? 0 : (last - first + 1);
        off[m] = (int32_t)wts.size();
        for (int k = 0; k < cnt[m]; k++) wts.push_back(h->filters[(size_t)(lo[m] + k) * n_mel + m]);
    }
    if (wts.empty()) wts.push_back(0.0f);
    std::vector<int32_t> mel2(384, 0);
    for (int m = 0; m < n_mel; m++) {
        const int groups = (cnt[m] + 3) / 4;
        mel2[m] = lo[m];
        mel2[128 + m] = groups;
        mel2[256 + m] = (int32_t)((mel2.size() - 384) / 4);
        for (int k = 0; k < groups * 4; k++) {
            const float wv = k < cnt[m] ? wts[off[m] + k] : 0.0f;
            int32_t bits;
            memcpy(&bits, &wv, 4);
            mel2.push_back(bits);
        }
    }
    for (int k = 0; k < 4; k++) mel2.push_back(0);       // trailing all-zero weight group
    h->mel2_words = (int)mel2.size();
    // ---- folded DFT table: row n, col c<224: hann[n]*cos(2pi c n/400) (c<=200), col 224+c: sin
    std::vector<float> tab((size_t)kTabRows * kTabCols, 0.0f);
    for (int n = 0; n <= 200; n++) {
        const double win = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / 400.0);
        const double fwin = (double)(float)win * ((n == 200) ? 0.5 : 1.0);
        for (int c = 0; c <= 200; c++) {
            const int ph = (int)(((long long)c * n) % 400);
            const double ang = 2.0 * M_PI * (double)ph / 400.0;
            tab[(size_t)n * kTabCols + c] = (float)(fwin * cos(ang));
            if (n >= 1 && n <= 199 && c >= 1 && c <= 199) tab[(size_t)n * kTabCols + 224 + c] = (float)(fwin * sin(ang));
        }
    }
    // ---- v2 table: rows = output bins (cos 0..223 | sin 224..447), cols = folded n, three bf16 splits
    auto bf16_rn = [](float f) -> uint16_t {
        uint32_t u;
        memcpy(&u, &f, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto bf16_f = [](uint16_t hbits) -> float {
        uint32_t u = (uint32_t)hbits << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    std::vector<uint16_t> tab2((size_t)kKS2 * 3 * 448 * 16, 0);
    for (int bin = 0; bin < 448; bin++)
        for (int n = 0; n < kK2; n++) {
            const float tv = (n < kTabRows) ? tab[(size_t)n * kTabCols + bin] : 0.0f;
            const uint16_t s0 = bf16_rn(tv);
            const float r1 = tv - bf16_f(s0);
            const uint16_t s1 = bf16_rn(r1);
            const float r2 = r1 - bf16_f(s1);
            const uint16_t s2 = bf16_rn(r2);
            const int ks = n / 16, kk = n % 16;
            const uint16_t sp[3] = {s0, s1, s2};
            for (int q = 0; q < 3; q++) tab2[(((size_t)ks * 3 + q) * 448 + bin) * 16 + kk] = sp[q];
        }
    hipError_t e = hipSuccess;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes);
        if (e == hipSuccess) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    {   // FFT path tables (float, computed in double)
        std::vector<float> winf(kNfft);
        for (int n = 0; n < kNfft; n++) winf[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)kNfft));
        std::vector<f2v> t200(8 * 25), t400(101);
        for (int n2 = 0; n2 < 8; n2++)
            for (int k1 = 0; k1 < 25; k1++) {
                const double ang = -2.0 * M_PI * (double)((n2 * k1) % 200) / 200.0;
                t200[n2 * 25 + k1] = (f2v){(float)cos(ang), (float)sin(ang)};
            }
        for (int k = 0; k <= 100; k++) {
            const double ang = -2.0 * M_PI * (double)k / 400.0;
            t400[k] = (f2v){(float)cos(ang), (float)sin(ang)};
        }
        FftConst fc;
        for (int bb = 0; bb < 5; bb++)
            for (int cc = 0; cc < 5; cc++) {
                const double ang = -2.0 * M_PI * (double)((bb * cc) % 25) / 25.0;
                fc.w25[bb][cc] = (f2v){(float)cos(ang), (float)sin(ang)};
            }
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_fft), &fc, sizeof(fc));
        up((void **)&h->d_win, winf.data(), winf.size() * 4);
        up((void **)&h->d_tw200, t200.data(), t200.size() * sizeof(f2v));
        up((void **)&h->d_tw400, t400.data(), t400.size() * sizeof(f2v));
    }
    up((void **)&h->d_tab, tab.data(), tab.size() * 4);
    up((void **)&h->d_tab2, tab2.data(), tab2.size() * 2);
    up((void **)&h->d_mel2, mel2.data(), mel2.size() * 4);
    up((void **)&h->d_lo, lo.data(), lo.size() * 4);
    up((void **)&h->d_cnt, cnt.data(), cnt.size() * 4);
    up((void **)&h->d_off, off.data(), off.size() * 4);
    up((void **)&h->d_w, wts.data(), wts.size() * 4);
    if (e != hipSuccess) {
        ifh_logmel_destroy(h);
        return check_hip(e, "logmel_create");
    }
    *out = h;
    return IFH_OK;
}

extern "C" int ifh_logmel_destroy(ifh_logmel_t h)
{
    if (!h) return IFH_OK;
    if (h->d_tab) (void)hipFree(h->d_tab);
    if (h->d_tab2) (void)hipFree(h->d_tab2);
    if (h->d_win) (void)hipFree(h->d_win);
    if (h->d_tw200) (void)hipFree(h->d_tw200);
    if (h->d_tw400) (void)hipFree(h->d_tw400);
    if (h->d_mel2) (void)hipFree(h->d_mel2);
    if (h->d_lo) (void)hipFree(h->d_lo);
    if (h->d_cnt) (void)hipFree(h->d_cnt);
    if (h->d_off) (void)hipFree(h->d_off);
    if (h->d_w) (void)hipFree(h->d_w);
    delete h;
    return IFH_OK;
}

extern "C" int ifh_logmel_filters_host(ifh_logmel_t h, float *out)
{
    IFH_CHECK_ARG(h && out);
    memcpy(out, h->filters.data(), h->filters.size() * sizeof(float));
    return IFH_OK;
}

extern "C" int64_t ifh_logmel_workspace_floats(ifh_logmel_t h, int nbatch, int out_bf16)
{
    if (!h || nbatch < 0) return -1;
    return (int64_t)nbatch + (out_bf16 ? (int64_t)nbatch * h->n_mel * kFrames : 0);
}

static int logmel_transform(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch, float *raw,
                            int *gmax, hipStream_t st);

extern "C" int ifh_logmel_run(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch,
                              void *out, int out_bf16, float *workspace, ifh_stream_t stream)
{
    IFH_CHECK_ARG(h && nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(audio && out && workspace && stride >= 0);
    hipStream_t st = as_stream(stream);
    int *gmax = reinterpret_cast<int *>(workspace);
    float *raw = out_bf16 ? (workspace + nbatch) : reinterpret_cast<float *>(out);
    const int rc = logmel_transform(h, audio, stride, lens, nbatch, raw, gmax, st);
    if (rc != IFH_OK) return rc;
    const int per_utt = h->n_mel * kFrames;
    dim3 g2((per_utt / 4 + 255) / 256, nbatch);
    if (out_bf16)
        hipLaunchKernelGGL(k_logmel_finish<true>, g2, dim3(256), 0, st, raw, out, gmax, per_utt);
    else
        hipLaunchKernelGGL(k_logmel_finish<false>, g2, dim3(256), 0, st, raw, out, gmax, per_utt);
    IFH_LAUNCH_CHECK("logmel_finish");
    return IFH_OK;
}

extern "C" int ifh_logmel_run_raw(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch,
                                  float *raw, int32_t *win_max, ifh_stream_t stream)
{
    IFH_CHECK_ARG(h && nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(audio && raw && win_max && stride >= 0);
    return logmel_transform(h, audio, stride, lens, nbatch, raw, win_max, as_stream(stream));
}

extern "C" int ifh_logmel_finish_transpose_bf16(ifh_logmel_t h, const float *raw, const int32_t *win_max, int nbatch, void *out,
                                                ifh_stream_t stream)
{
    IFH_CHECK_ARG(h && nbatch >= 0 && nbatch < 65536);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(raw && win_max && out);
    dim3 grid((kFrames + 31) / 32, (h->n_mel + 31) / 32, nbatch);
    hipLaunchKernelGGL(k_logmel_finish_transpose, grid, dim3(256), 0, as_stream(stream), raw, (uint16_t *)out, win_max, h->n_mel, kFrames);
    IFH_LAUNCH_CHECK("logmel_finish_transpose");
    return IFH_OK;
}

static int logmel_transform(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch, float *raw,
                            int *gmax, hipStream_t st)
{
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)gmax, (int)0x80000000, (size_t)nbatch, st);
    if (e != hipSuccess) return check_hip(e, "logmel memset");
    const size_t ldsb = (size_t)kLdsFloats * sizeof(float);
    static unsigned long long attr_mask = 0;
    int attr_dev = 0;
    static const bool use_v1 = getenv("IFH_LOGMEL_V1") != nullptr;      // tuning switch: f32-MFMA formulation
    static const bool use_dft = getenv("IFH_LOGMEL_DFT") != nullptr;    // tuning switch: bf16x3-MFMA DFT instead of the FFT
    if (attr_needed_on_this_device(attr_mask, &attr_dev)) {
        e = hipFuncSetAttribute((const void *)k_logmel_dft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_logmel_dft2, hipFuncAttributeMaxDynamicSharedMemorySize, kLds2Bytes + 8192);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)k_logmel_fft, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsFftBytes + 8192);
        if (e != hipSuccess) return check_hip(e, "logmel set lds attr");
        attr_mask |= 1ull << attr_dev;
    }
    dim3 grid((kFrames + kFT - 1) / kFT, nbatch);
    if (!use_v1 && !use_dft) {
        static unsigned long long *d_proff = nullptr;
        static const bool do_proff = getenv("IFH_LOGMEL_PROF") != nullptr;
        if (do_proff && !d_proff) (void)hipMalloc((void **)&d_proff, 64);
        if (do_proff) (void)hipMemsetAsync(d_proff, 0, 64, st);
        const int ntile = (kFrames + kFB - 1) / kFB;
        int gx = (2 * 256 + nbatch - 1) / nbatch;            // ~2 resident blocks per CU in all, each walking its share of the tiles
        gx = gx < 1 ? 1 : (gx > ntile ? ntile : gx);
        hipLaunchKernelGGL(k_logmel_fft, dim3(gx, nbatch), dim3(256), (size_t)kLdsFftBytes + 816 + (size_t)h->mel2_words * 4, st, audio, stride,
                           lens, h->d_win, h->d_tw200, h->d_tw400, h->d_mel2, h->mel2_words, h->n_mel, raw, gmax,
                           do_proff ? d_proff : nullptr);
        if (do_proff) {
            unsigned long long hp[5];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hp, d_proff, 40, hipMemcpyDeviceToHost);
            if (hp[4]) fprintf(stderr, "logmel fft prof: blocks=%llu stage=%llu A=%llu B=%llu mel=%llu cycles/block\n", hp[4], hp[0] / hp[4], hp[1] / hp[4], hp[2] / hp[4], hp[3] / hp[4]);
        }
    } else if (use_v1)
        hipLaunchKernelGGL(k_logmel_dft, grid, dim3(448), ldsb, st, audio, stride, lens, h->d_tab, h->d_lo, h->d_cnt,
                           h->d_off, h->d_w, h->n_mel, raw, gmax);
    else {
        static unsigned long long *d_prof = nullptr;
        static const bool do_prof = getenv("IFH_LOGMEL_PROF") != nullptr;
        if (do_prof && !d_prof) { (void)hipMalloc((void **)&d_prof, 64); }
        if (do_prof) (void)hipMemsetAsync(d_prof, 0, 64, st);
        hipLaunchKernelGGL(k_logmel_dft2, grid, dim3(256), (size_t)kLds2Bytes + (size_t)h->mel2_words * 4, st, audio, stride, lens,
                           h->d_tab2, h->d_mel2, h->mel2_words, h->n_mel, raw, gmax, do_prof ? d_prof : nullptr);
        if (do_prof) {
            unsigned long long hp[4];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hp, d_prof, 32, hipMemcpyDeviceToHost);
            if (hp[3]) fprintf(stderr, "logmel2 prof: blocks=%llu stage=%llu dft=%llu mel=%llu cycles/block\n", hp[3], hp[0] / hp[3], hp[1] / hp[3], hp[2] / hp[3]);
        }
    }
    IFH_LAUNCH_CHECK("logmel_dft");
    return IFH_OK;
}
