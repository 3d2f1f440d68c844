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
    hipError_t e = hipSuccess;
    auto up = [&](void **dst, const void *src, size_t bytes) {
        if (e != hipSuccess) return;
        e = hipMalloc(dst, bytes);
        if (e == hipSuccess) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    up((void **)&h->d_tab, tab.data(), tab.size() * 4);
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

extern "C" int ifh_logmel_run(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch,
                              void *out, int out_bf16, float *workspace, ifh_stream_t stream)
{
    IFH_CHECK_ARG(h && nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(audio && out && workspace && stride >= 0);
    hipStream_t st = as_stream(stream);
    int *gmax = reinterpret_cast<int *>(workspace);
    float *raw = out_bf16 ? (workspace + nbatch) : reinterpret_cast<float *>(out);
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)gmax, (int)0x80000000, (size_t)nbatch, st);
    if (e != hipSuccess) return check_hip(e, "logmel memset");
    const size_t ldsb = (size_t)kLdsFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute((const void *)k_logmel_dft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (e != hipSuccess) return check_hip(e, "logmel set lds attr");
        attr_set = true;
    }
    dim3 grid((kFrames + kFT - 1) / kFT, nbatch);
    hipLaunchKernelGGL(k_logmel_dft, grid, dim3(448), ldsb, st, audio, stride, lens, h->d_tab, h->d_lo, h->d_cnt,
                       h->d_off, h->d_w, h->n_mel, raw, gmax);
    IFH_LAUNCH_CHECK("logmel_dft");
    const int per_utt = h->n_mel * kFrames;
    dim3 g2((per_utt / 4 + 255) / 256, nbatch);
    if (out_bf16)
        hipLaunchKernelGGL(k_logmel_finish<true>, g2, dim3(256), 0, st, raw, out, gmax, per_utt);
    else
        hipLaunchKernelGGL(k_logmel_finish<false>, g2, dim3(256), 0, st, raw, out, gmax, per_utt);
    IFH_LAUNCH_CHECK("logmel_finish");
    return IFH_OK;
}
