// logmel.hip -- Whisper log-mel front end for gfx950.
//
// Replaces Cluster/InfernSTTWorker.py:114 (WhisperProcessor -> WhisperFeatureExtractor;
// transformers feature_extraction_whisper.py:135-168): zero-pad to 30 s, reflect-centred
// STFT (n_fft 400, hop 160, periodic Hann), power, slaney mel filterbank, log10,
// per-utterance max-8 clamp, (x+4)/4.
//
// k_logmel_fft computes the raw log-mel rows and the per-window maximum (real 400-point FFT, sparse mel projection);
// the clamp / scale is a second small kernel (k_logmel_finish) or is folded into the layout change in front of Whisper's
// conv1 (k_logmel_finish_transpose).  (The two DFT-as-GEMM formulations of round 1 -- exact-f32 MFMA and 3-way bf16 split --
// were removed: their matrix work alone exceeds the FFT kernel's whole time)
// Algorithmic bytes per 30 s window: 480000*4 read + 80*3000*4 written = 2.88 MB.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace ifh {

constexpr int kNfft = 400, kHop = 160, kBins = 201, kFrames = 3000, kNsamp = 480000;

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

// The library is built with -ffp-contract=off (the bit-exact codec / resampler / GEMM-order guarantees).  The transform below is
// held to 1e-3 against the float64 reference instead: fused multiply-adds shorten its dependent chains and round once, not twice.
#pragma clang fp contract(fast)

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

__global__ __launch_bounds__(256, 2) void k_logmel_fft(const float *__restrict__ audio, int64_t stride,
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
    // per-thread constants of phase A, loaded once per block (a thread keeps its n2 = tid & 7 over all tiles): the 25 window
    // pairs w[2n], w[2n+1] of its samples n = 8 n1 + n2.  As per-tile global loads they were 25 L1 round trips per thread per
    // tile in front of the transform.  (The 25 output twiddles W200^(n2 k1) as well would take the kernel past 256 registers.)
    f2v wreg[25];
    {
        const int sub_ = tid & 7;
#pragma unroll
        for (int i = 0; i < 25; i++) wreg[i] = *reinterpret_cast<const f2v *>(&win[2 * (8 * i + sub_)]);
    }
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
        // sample pair n = 8 n1 + n2 of frame f sits at 176 f + 2 n + 16 ((2 n) / 160); 2 n = 16 n1 + 2 n2 with 2 n2 <= 14 never
        // carries into the next multiple of 160, so the padding term is the compile-time 16 (n1 / 10): one base + literals
        const float *abase = aud + 176 * f + 2 * sub;
#define LF_LOAD(V, N1)                                                                               \
    {                                                                                                \
        const f2v xs = *reinterpret_cast<const f2v *>(abase + (16 * (N1) + 16 * ((N1) / 10)));       \
        V = xs * wreg[N1];                                                                           \
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
// W25^(b c) as literals (round 3): as __constant__ loads the 16 twiddles sat in 32 SGPRs for the whole kernel and pushed the
// compiler into 70+ SGPR spills (v_writelane / v_readlane + hazard nops in every tile)
#define LF_W25(B, C) ( \
    (B) == 1 && (C) == 1 ? mk2(9.685831611e-01f, -2.486898872e-01f) : \
    (B) == 1 && (C) == 2 ? mk2(8.763066800e-01f, -4.817536741e-01f) : \
    (B) == 1 && (C) == 3 ? mk2(7.289686274e-01f, -6.845471059e-01f) : \
    (B) == 1 && (C) == 4 ? mk2(5.358267950e-01f, -8.443279255e-01f) : \
    (B) == 2 && (C) == 1 ? mk2(8.763066800e-01f, -4.817536741e-01f) : \
    (B) == 2 && (C) == 2 ? mk2(5.358267950e-01f, -8.443279255e-01f) : \
    (B) == 2 && (C) == 3 ? mk2(6.279051953e-02f, -9.980267284e-01f) : \
    (B) == 2 && (C) == 4 ? mk2(-4.257792916e-01f, -9.048270525e-01f) : \
    (B) == 3 && (C) == 1 ? mk2(7.289686274e-01f, -6.845471059e-01f) : \
    (B) == 3 && (C) == 2 ? mk2(6.279051953e-02f, -9.980267284e-01f) : \
    (B) == 3 && (C) == 3 ? mk2(-6.374239897e-01f, -7.705132428e-01f) : \
    (B) == 3 && (C) == 4 ? mk2(-9.921147013e-01f, -1.253332336e-01f) : \
    (B) == 4 && (C) == 1 ? mk2(5.358267950e-01f, -8.443279255e-01f) : \
    (B) == 4 && (C) == 2 ? mk2(-4.257792916e-01f, -9.048270525e-01f) : \
    (B) == 4 && (C) == 3 ? mk2(-9.921147013e-01f, -1.253332336e-01f) : \
    (B) == 4 && (C) == 4 ? mk2(-6.374239897e-01f, 7.705132428e-01f) : \
    mk2(1.0f, 0.0f))
#define LF_TW(V, B, C) V = cmul(V, LF_W25(B, C));
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
    int32_t *d_mel2 = nullptr;    // packed filters: lo[128] | groups[128] | first group[128] | weights (x4 padded)
    int mel2_words = 0;
    float *d_win = nullptr;       // hann window [400], W200^(n2 k1) [8][25], W400^k [101]
    f2v *d_tw200 = nullptr, *d_tw400 = nullptr;
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
    up((void **)&h->d_mel2, mel2.data(), mel2.size() * 4);
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
    if (h->d_win) (void)hipFree(h->d_win);
    if (h->d_tw200) (void)hipFree(h->d_tw200);
    if (h->d_tw400) (void)hipFree(h->d_tw400);
    if (h->d_mel2) (void)hipFree(h->d_mel2);
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
    static DeviceOnce attr_once;
    int attr_dev = 0;
    if (attr_once.needed(&attr_dev)) {
        e = hipFuncSetAttribute((const void *)k_logmel_fft, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsFftBytes + 8192);
        if (e != hipSuccess) return check_hip(e, "logmel set lds attr");
        attr_once.done(attr_dev);
    }
    static unsigned long long *d_proff = nullptr;
    static const bool do_proff = getenv("IFH_LOGMEL_PROF") != nullptr;      // diagnostic: phase clocks of the kernel to stderr
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
    IFH_LAUNCH_CHECK("logmel_dft");
    return IFH_OK;
}
