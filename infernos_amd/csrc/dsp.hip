// dsp.hip -- G.711 mu-law, sinc resampler, per-tick ingest, VAD state machine / chunk
// assembly for gfx950.  All byte/integer work here is HBM- or latency-bound: the design
// rules are coalesced 16-byte stores, LDS-staged tiles, one block per call for the
// stateful kernels.  Reference lines are cited per kernel; include/infernos_hip.h holds
// the contracts.
#include <math.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace ifh {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return IFH_OK;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return IFH_EHIP;
}

// ---- mu-law closed forms (Core/Codecs/G711.py:7-19 tables via audioop) -------------------
__host__ __device__ constexpr int ulaw2lin(unsigned code)
{
    unsigned u = (~code) & 0xFFu;
    int t = (int)((u & 0x0Fu) << 3) + 0x84;
    t <<= (u & 0x70u) >> 4;
    return (u & 0x80u) ? (0x84 - t) : (t - 0x84);
}

__host__ __device__ inline unsigned lin2ulaw(int v /* int16 range */)
{
    int p = v >> 2;
    const unsigned mask = (p < 0) ? 0x7Fu : 0xFFu;
    p = (p < 0) ? -p : p;
    p = (p > 8159) ? 8159 : p;
    p += 33;  // 33..8192
#if defined(__HIP_DEVICE_COMPILE__)
    const int msb = 31 - __clz(p);
#else
    int msb = 0;
    for (int q = p; q > 1; q >>= 1) msb++;
#endif
    const int seg = msb - 5;  // 0..8
    const unsigned uval = (seg >= 8) ? 0x7Fu : (unsigned)((seg << 4) | ((p >> (seg + 1)) & 0xF));
    return (uval ^ mask) & 0xFFu;
}

struct UlawF32Lut {
    float v[256];
    constexpr UlawF32Lut() : v{}
    {
        for (int i = 0; i < 256; i++) v[i] = (float)ulaw2lin((unsigned)i) / 32767.0f;
    }
};
__constant__ UlawF32Lut c_ulaw_f32 = UlawF32Lut();

__device__ __forceinline__ unsigned encode_sample(float x)
{
    // G711.py:27: clamp(x*32767, -32768, 32767).to(int16) (truncation), then table
    float s = x * 32767.0f;
    s = (s == s) ? s : 0.0f;
    s = fminf(fmaxf(s, -32768.0f), 32767.0f);
    return lin2ulaw((int)s);
}

// ---- bulk decode: u8[n] -> f32[n].  4 B load / 16 B store per lane, fully coalesced.
__global__ __launch_bounds__(256) void k_g711_decode(const uint8_t *__restrict__ in, float *__restrict__ out,
                                                     int64_t n)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = c_ulaw_f32.v[threadIdx.x];
    __syncthreads();
    const int64_t n4 = n >> 2;
    const bool aligned = ((((uintptr_t)in) & 3) == 0) && ((((uintptr_t)out) & 15) == 0);
    if (aligned) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
             i += (int64_t)gridDim.x * blockDim.x) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(in)[i];
            float4 o;
            o.x = lut[w & 0xFF];
            o.y = lut[(w >> 8) & 0xFF];
            o.z = lut[(w >> 16) & 0xFF];
            o.w = lut[(w >> 24) & 0xFF];
            reinterpret_cast<float4 *>(out)[i] = o;
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            out[i] = lut[in[i]];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            out[i] = lut[in[i]];
    }
}

// ---- bulk encode: f32[n] -> u8[n].  16 B load / 4 B store per lane.
__global__ __launch_bounds__(256) void k_g711_encode(const float *__restrict__ in, uint8_t *__restrict__ out,
                                                     int64_t n)
{
    const int64_t n4 = n >> 2;
    const bool aligned = ((((uintptr_t)in) & 15) == 0) && ((((uintptr_t)out) & 3) == 0);
    if (aligned) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
             i += (int64_t)gridDim.x * blockDim.x) {
            const float4 v = reinterpret_cast<const float4 *>(in)[i];
            const uint32_t w = encode_sample(v.x) | (encode_sample(v.y) << 8) | (encode_sample(v.z) << 16) |
                               (encode_sample(v.w) << 24);
            reinterpret_cast<uint32_t *>(out)[i] = w;
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            out[i] = (uint8_t)encode_sample(in[i]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (int64_t)gridDim.x * blockDim.x)
            out[i] = (uint8_t)encode_sample(in[i]);
    }
}

static int grid_for(int64_t work_items, int block)
{
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;  // 8 blocks per CU, grid-stride the rest
    return (int)g;
}

}  // namespace ifh

using namespace ifh;

extern "C" const char *ifh_last_error(void) { return g_err.c_str(); }
extern "C" int ifh_version(void) { return 100; }
extern "C" int ifh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ifh_g711_tables_host(int16_t *out256_host, uint8_t *out65536_host)
{
    IFH_CHECK_ARG(out256_host && out65536_host);
    for (int i = 0; i < 256; i++) out256_host[i] = (int16_t)ulaw2lin((unsigned)i);
    for (int v = -32768; v < 32768; v++) out65536_host[v + 32768] = (uint8_t)lin2ulaw(v);
    return IFH_OK;
}

extern "C" int ifh_g711_decode_u8_f32(const uint8_t *in, float *out, int64_t n, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(in && out);
    hipLaunchKernelGGL(k_g711_decode, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), in, out, n);
    IFH_LAUNCH_CHECK("g711_decode");
    return IFH_OK;
}

extern "C" int ifh_g711_encode_f32_u8(const float *in, uint8_t *out, int64_t n, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(in && out);
    hipLaunchKernelGGL(k_g711_encode, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), in, out, n);
    IFH_LAUNCH_CHECK("g711_encode");
    return IFH_OK;
}

// =========================================================================================
// Sinc resampler (torchaudio Resample restated; Core/AudioChunk.py:19-24)
// =========================================================================================
struct ifh_resampler {
    int orig, nw, ntaps, width;
    std::vector<float> taps;  // [nw][ntaps]
    float *d_taps = nullptr;
};

namespace ifh {

static int gcd_int(int a, int b)
{
    while (b) {
        int t = a % b;
        a = b;
        b = t;
    }
    return a;
}

// torchaudio.functional._get_sinc_resample_kernel, sinc_interp_hann, width 6, rolloff 0.99;
// float64 arithmetic, result cast to float32.
static void build_sinc_kernel(int orig_sr, int new_sr, ifh_resampler *r)
{
    const int g = gcd_int(orig_sr, new_sr);
    const int o = orig_sr / g, n = new_sr / g;
    const double lpw = 6.0, rolloff = 0.99;
    const double base = (double)(o < n ? o : n) * rolloff;
    const int width = (int)ceil(lpw * o / base);
    const int ntaps = 2 * width + o;
    r->orig = o;
    r->nw = n;
    r->ntaps = ntaps;
    r->width = width;
    r->taps.resize((size_t)n * ntaps);
    const double scale = base / o;
    for (int p = 0; p < n; p++) {
        const double ph = (double)(float)((double)(-p) / (double)n);
        for (int j = 0; j < ntaps; j++) {
            double t = (ph + (double)(j - width) / (double)o) * base;
            if (t < -lpw) t = -lpw;
            if (t > lpw) t = lpw;
            const double c = cos(t * M_PI / lpw / 2.0);
            const double window = c * c;
            t *= M_PI;
            const double s = (t == 0.0) ? 1.0 : sin(t) / t;
            r->taps[(size_t)p * ntaps + j] = (float)(s * window * scale);
        }
    }
}

constexpr int kRsTile = 2048;  // outputs per block

// y[n*nw + p] = sum_j taps[p][j] * xpad[n*orig + j], fixed ascending-j fmaf chain
// (bit-identical to oracle/dsp_oracle.c:orc_resample).
__global__ __launch_bounds__(256) void k_resample(const float *__restrict__ in, int64_t in_stride,
                                                  const int32_t *__restrict__ lens, int64_t max_len,
                                                  float *__restrict__ out, int64_t out_stride,
                                                  const float *__restrict__ taps, int orig, int nw, int ntaps,
                                                  int width, int span)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xs = smem;          // [span]
    float *tp = smem + span;   // [nw*ntaps]
    const int row = blockIdx.y;
    const int64_t L = lens ? (int64_t)lens[row] : max_len;
    const int64_t out_len = (L * nw + orig - 1) / orig;
    const int64_t o0 = (int64_t)blockIdx.x * kRsTile;
    if (o0 >= out_len) return;
    const int64_t n0 = o0 / nw;  // kRsTile is a multiple of nw for every ratio accepted at create
    const int64_t x0 = n0 * orig - width;
    const float *x = in + (int64_t)row * in_stride;
    for (int i = threadIdx.x; i < span; i += blockDim.x) {
        const int64_t idx = x0 + i;
        xs[i] = (idx >= 0 && idx < L) ? x[idx] : 0.0f;
    }
    for (int i = threadIdx.x; i < nw * ntaps; i += blockDim.x) tp[i] = taps[i];
    __syncthreads();
    float *y = out + (int64_t)row * out_stride;
    for (int t = threadIdx.x; t < kRsTile; t += blockDim.x) {
        const int64_t o = o0 + t;
        if (o >= out_len) break;
        const int nl = t / nw, p = t - nl * nw;
        const float *k = tp + p * ntaps;
        const float *xv = xs + nl * orig;
        float acc = 0.0f;
        for (int j = 0; j < ntaps; j++) acc = __fmaf_rn(k[j], xv[j], acc);
        y[o] = acc;
    }
}

}  // namespace ifh

extern "C" int ifh_resample_create(int orig_sr, int new_sr, ifh_resampler_t *out)
{
    IFH_CHECK_ARG(out && orig_sr > 0 && new_sr > 0);
    ifh_resampler *r = new ifh_resampler();
    build_sinc_kernel(orig_sr, new_sr, r);
    if (kRsTile % r->nw != 0 || r->nw > 64 || r->orig > 64) {
        delete r;
        return fail(IFH_EINVAL, "ifh_resample_create: unsupported ratio (reduced new must divide 2048, both <= 64)");
    }
    {   // the tile of k_resample: ((2048 / new) * orig + ntaps) input samples + new * ntaps taps, f32, in LDS
        const size_t lds = ((size_t)(kRsTile / r->nw) * r->orig + r->ntaps + (size_t)r->nw * r->ntaps) * sizeof(float);
        if (lds > 64 * 1024) {
            delete r;
            return fail(IFH_EINVAL, "ifh_resample_create: ratio needs more than 64 KB of LDS per tile (orig/new too large, e.g. 8:1)");
        }
    }
    const size_t bytes = r->taps.size() * sizeof(float);
    hipError_t e = hipMalloc((void **)&r->d_taps, bytes);
    if (e != hipSuccess) {
        delete r;
        return check_hip(e, "resample_create hipMalloc");
    }
    e = hipMemcpy(r->d_taps, r->taps.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(r->d_taps);
        delete r;
        return check_hip(e, "resample_create hipMemcpy");
    }
    *out = r;
    return IFH_OK;
}

extern "C" int ifh_resample_destroy(ifh_resampler_t h)
{
    if (!h) return IFH_OK;
    if (h->d_taps) (void)hipFree(h->d_taps);
    delete h;
    return IFH_OK;
}

extern "C" int ifh_resample_info(ifh_resampler_t h, int *orig_r, int *new_r, int *ntaps, int *width, float *taps_host)
{
    IFH_CHECK_ARG(h);
    if (orig_r) *orig_r = h->orig;
    if (new_r) *new_r = h->nw;
    if (ntaps) *ntaps = h->ntaps;
    if (width) *width = h->width;
    if (taps_host) memcpy(taps_host, h->taps.data(), h->taps.size() * sizeof(float));
    return IFH_OK;
}

extern "C" int64_t ifh_resample_out_len(ifh_resampler_t h, int64_t in_len)
{
    if (!h || in_len < 0) return -1;
    return (in_len * h->nw + h->orig - 1) / h->orig;
}

extern "C" int ifh_resample_run(ifh_resampler_t h, const float *in, int64_t in_stride, const int32_t *lens,
                                int64_t max_len, int nrows, float *out, int64_t out_stride, ifh_stream_t stream)
{
    IFH_CHECK_ARG(h && nrows >= 0 && max_len >= 0);
    if (nrows == 0 || max_len == 0) return IFH_OK;
    IFH_CHECK_ARG(in && out);
    const int64_t max_out = ifh_resample_out_len(h, max_len);
    IFH_CHECK_ARG(out_stride >= max_out && in_stride >= max_len);
    const int span = (kRsTile / h->nw) * h->orig + h->ntaps;
    const size_t lds = (size_t)(span + h->nw * h->ntaps) * sizeof(float);
    dim3 grid((unsigned)((max_out + kRsTile - 1) / kRsTile), (unsigned)nrows);
    hipLaunchKernelGGL(k_resample, grid, dim3(256), lds, as_stream(stream), in, in_stride, lens, max_len, out,
                       out_stride, h->d_taps, h->orig, h->nw, h->ntaps, h->width, span);
    IFH_LAUNCH_CHECK("resample");
    return IFH_OK;
}

// =========================================================================================
// Per-tick ingest (VADChannel.ingest + G711Codec.decode, Core/VAD/SileroVAD.py:27-35)
// =========================================================================================
namespace ifh {

__global__ __launch_bounds__(64) void k_ingest_tick(const uint8_t *__restrict__ frames,
                                                    const int32_t *__restrict__ slot, uint8_t *__restrict__ fifo,
                                                    int32_t *__restrict__ fifo_len, float *__restrict__ win,
                                                    int32_t *__restrict__ win_ready, float *__restrict__ hist,
                                                    float *__restrict__ pcm8k, float *__restrict__ pcm16k,
                                                    const float *__restrict__ taps /* [2][15] */, int nt,
                                                    int64_t tick_stride)
{
    __shared__ float xs[15 + 160];
    __shared__ float tp[32];
    __shared__ uint8_t fb[IFH_FIFO_CAP];
    const int i = blockIdx.x, t = threadIdx.x;
    const int s = slot[i];
    uint8_t *ff = fifo + (int64_t)s * IFH_FIFO_CAP;
    if (t < 30) tp[t] = taps[t];
    // nt consecutive ticks of this call (frames of tick k at frames + k*tick_stride): the per-call state goes through
    // global memory between ticks exactly as between launches; pcm8k/pcm16k keep the last tick's samples
    for (int tk = 0; tk < nt; tk++) {
    if (tk) __syncthreads();
    const uint8_t *fr = frames + (int64_t)tk * tick_stride + (int64_t)i * 160;
    const int fl = fifo_len[s];
    if (t < 15) xs[t] = hist[(int64_t)s * 16 + t];
    // existing FIFO bytes -> LDS
    for (int k = t; k < fl; k += 64) fb[k] = ff[k];
    for (int k = t; k < 160; k += 64) {
        const uint8_t b = fr[k];
        const float v = c_ulaw_f32.v[b];
        xs[15 + k] = v;
        pcm8k[(int64_t)i * 160 + k] = v;
        fb[fl + k] = b;
    }
    __syncthreads();
    // streaming 8k->16k: out k uses xs[k/2 + j], phase k&1
    for (int k = t; k < 320; k += 64) {
        const float *kk = tp + (k & 1) * 15;
        const float *xv = xs + (k >> 1);
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < 15; j++) acc = __fmaf_rn(kk[j], xv[j], acc);
        pcm16k[(int64_t)i * 320 + k] = acc;
    }
    if (t < 15) hist[(int64_t)s * 16 + t] = xs[160 + t];
    const int nfl = fl + 160;
    if (nfl >= IFH_VAD_WINDOW) {
        float *w = win + (int64_t)s * IFH_VAD_WINDOW;
        for (int k = t; k < IFH_VAD_WINDOW; k += 64) w[k] = c_ulaw_f32.v[fb[k]];
        const int rem = nfl - IFH_VAD_WINDOW;
        for (int k = t; k < rem; k += 64) ff[k] = fb[IFH_VAD_WINDOW + k];
        if (t == 0) {
            fifo_len[s] = rem;
            win_ready[s] = 1;
        }
    } else {
        for (int k = t; k < 160; k += 64) ff[fl + k] = fb[fl + k];
        if (t == 0) {
            fifo_len[s] = nfl;
            win_ready[s] = 0;
        }
    }
    }
}

}  // namespace ifh

extern "C" int ifh_ingest_tick(const uint8_t *frames, const int32_t *slot, int n, uint8_t *fifo, int32_t *fifo_len,
                               float *win, int32_t *win_ready, float *hist, float *pcm8k, float *pcm16k,
                               ifh_resampler_t rs8to16, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(frames && slot && fifo && fifo_len && win && win_ready && hist && pcm8k && pcm16k && rs8to16);
    IFH_CHECK_ARG(rs8to16->orig == 1 && rs8to16->nw == 2 && rs8to16->ntaps == 15);
    hipLaunchKernelGGL(k_ingest_tick, dim3(n), dim3(64), 0, as_stream(stream), frames, slot, fifo, fifo_len, win,
                       win_ready, hist, pcm8k, pcm16k, rs8to16->d_taps, 1, (int64_t)0);
    IFH_LAUNCH_CHECK("ingest_tick");
    return IFH_OK;
}

// =========================================================================================
// Output side of the per-call loop: OutputMTMuxer.idle mix (Core/OutputMuxer.py:75-85)
// fused with G711Codec.encode (Core/Codecs/G711.py:25-32), batched over calls.
// tracks f32 [n][K][L] (zero-padded blocks, track order = insertion order), present u8 [n][K] (1 if the
// track produced a block), ndiv i32 [n] (= len(self.tracks)); out u8 [n][L]; has_out u8 [n].
//   0 present -> has_out 0 (the reference sends nothing);  1 present -> that block unchanged;
//   >= 2 -> float32 sum in track order, then one IEEE division by ndiv.
// =========================================================================================
namespace ifh {

__global__ __launch_bounds__(256) void k_mux_encode(const float *__restrict__ tracks, const uint8_t *__restrict__ present,
                                                    const int32_t *__restrict__ ndiv, int K, int L,
                                                    uint8_t *__restrict__ out, uint8_t *__restrict__ has_out)
{
    const int c = blockIdx.x;
    int cnt = 0, only = 0;
    for (int k = 0; k < K; k++)
        if (present[(int64_t)c * K + k]) {
            cnt++;
            only = k;
        }
    if (threadIdx.x == 0) has_out[c] = cnt > 0;
    if (cnt == 0) return;
    const float *tb = tracks + (int64_t)c * K * L;
    const float div = (float)ndiv[c];
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        float v;
        if (cnt == 1) {
            v = tb[(int64_t)only * L + i];
        } else {
            v = 0.0f;
            bool first = true;
            for (int k = 0; k < K; k++)
                if (present[(int64_t)c * K + k]) {
                    const float x = tb[(int64_t)k * L + i];
                    v = first ? x : v + x;
                    first = false;
                }
            v = v / div;
        }
        out[(int64_t)c * L + i] = (uint8_t)encode_sample(v);
    }
}

}  // namespace ifh

extern "C" int ifh_mux_encode_f32_u8(const float *tracks, const uint8_t *present, const int32_t *ndiv, int ncalls,
                                     int ntracks, int block_len, uint8_t *out, uint8_t *has_out, ifh_stream_t stream)
{
    IFH_CHECK_ARG(ncalls >= 0 && ntracks >= 1 && block_len >= 1);
    if (ncalls == 0) return IFH_OK;
    IFH_CHECK_ARG(tracks && present && ndiv && out && has_out);
    hipLaunchKernelGGL(k_mux_encode, dim3(ncalls), dim3(256), 0, as_stream(stream), tracks, present, ndiv, ntracks,
                       block_len, out, has_out);
    IFH_LAUNCH_CHECK("mux_encode");
    return IFH_OK;
}

// =========================================================================================
// VAD: stand-in probability model, hysteresis FSM, chunk assembly
// =========================================================================================
namespace ifh {

// Stand-in for the Silero v3.1 network (third party, weights unavailable offline):
// p = sigmoid(0.5 * (10*log10(mean(x^2) + 1e-10) + 30)).  One wave per window.
__global__ __launch_bounds__(64) void k_vad_energy_prob(const float *__restrict__ win,
                                                        const int32_t *__restrict__ slot, float *__restrict__ prob)
{
    const int i = blockIdx.x, t = threadIdx.x;
    const float *w = win + (int64_t)slot[i] * IFH_VAD_WINDOW;
    float acc = 0.0f;
    for (int k = t; k < IFH_VAD_WINDOW; k += 64) acc = __fmaf_rn(w[k], w[k], acc);
    acc = wave_sum(acc);
    if (t == 0) {
        const float db = 10.0f * log10f(acc / (float)IFH_VAD_WINDOW + 1e-10f);
        prob[i] = 1.0f / (1.0f + expf(-0.5f * (db + 30.0f)));
    }
}

// VADIteratorB.__call__ per channel (SileroVADUtils.py:105-130).  Python-float semantics:
// probabilities are f32 values compared in double against thr and thr-0.15.
__device__ __forceinline__ void vad_fsm(double p, int W, int sample_rate, double threshold, int64_t &trig,
                                        int64_t &temp_end, int64_t &cur, int64_t &kind, int64_t &pos)
{
    const double min_sil = sample_rate * 100 / 1000.0, pad = sample_rate * 30 / 1000.0;
    kind = 0;
    pos = 0;
    cur += W;
    if (p >= threshold && temp_end) temp_end = 0;
    if (p >= threshold && !trig) {
        trig = 1;
        const double sp = (cur > W) ? pad : 0.0;
        kind = 1;
        pos = (int64_t)((double)cur - sp - (double)W);
    } else if (p < threshold - 0.15 && trig) {
        if (!temp_end) temp_end = cur;
        if ((double)(cur - temp_end) >= min_sil) {
            kind = 2;
            pos = (int64_t)((double)temp_end + pad - (double)W);
            temp_end = 0;
            trig = 0;
        }
    }
}

// FSM only (no buffers): one thread per channel.
__global__ __launch_bounds__(64) void k_vad_fsm(const float *__restrict__ prob, const int32_t *__restrict__ slot,
                                                int n, int window, int sample_rate, double threshold,
                                                int64_t *__restrict__ st_i64, int64_t *__restrict__ ev2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t *st = st_i64 + (int64_t)slot[i] * 4;
    int64_t trig = st[0], temp_end = st[1], cur = st[2], kind, pos;
    vad_fsm((double)prob[i], window, sample_rate, threshold, trig, temp_end, cur, kind, pos);
    st[0] = trig;
    st[1] = temp_end;
    st[2] = cur;
    ev2[2 * i] = kind;
    ev2[2 * i + 1] = pos;
}

// One block per call.  SileroVADUtils.py:105-130 (FSM) + SileroVAD.py:81-112 (buffers).
__global__ __launch_bounds__(256) void k_vad_step(const float *__restrict__ win, const float *__restrict__ prob,
                                                  const int32_t *__restrict__ slot, int sample_rate,
                                                  double threshold, int64_t *__restrict__ st_i64,
                                                  int32_t *__restrict__ buf_len, float *__restrict__ abuf,
                                                  int64_t *__restrict__ ev, float *__restrict__ emit)
{
    constexpr int W = IFH_VAD_WINDOW;
    const int i = blockIdx.x, t = threadIdx.x;
    const int s = slot[i];
    float *ab = abuf + (int64_t)s * IFH_ABUF_CAP;
    float *em = emit + (int64_t)s * IFH_EMIT_CAP;
    const float *w = win + (int64_t)s * W;
    int64_t *st = st_i64 + (int64_t)s * 4;

    __shared__ int64_t sh[8];  // kind,pos,move_src,move_n,emit_len,new_blen,...
    int blen = buf_len[s];
    // vc.active_buffer = cat(active_buffer, p.audio)
    for (int k = t; k < W; k += 256) ab[blen + k] = w[k];
    blen += W;

    if (t == 0) {
        int64_t trig = st[0], temp_end = st[1], cur = st[2], astart = st[3];
        int64_t kind = 0, pos = 0;
        vad_fsm((double)prob[i], W, sample_rate, threshold, trig, temp_end, cur, kind, pos);
        int64_t err = 0, n_emit = 0, emit_ipos = 0, emit_len = 0, move_src = 0, move_n = 0;
        int64_t nb = blen;
        if (kind == 1) {
            if (astart != -1) err = 1;
            astart = pos;
            const int64_t poff = cur - astart;
            if (!(poff > 0 && poff < nb)) err = 1;
            if (!err) {
                move_src = nb - poff;
                move_n = poff;
                nb = poff;
            }
        } else if (kind == 2) {
            const int64_t aend = pos;
            if (!(astart != -1 && aend > astart)) err = 1;
            if (!(cur > aend)) err = 1;
            const int64_t poff = cur - aend;
            if (!(poff > 0 && poff < nb)) err = 1;
            if (!err && (nb - poff) != (aend - astart)) err = 1;
            if (!err) {
                n_emit = 1;
                emit_ipos = astart;
                emit_len = nb - poff;
                astart = -1;
            }
        }
        if (!err) {
            if (astart == -1) {
                if (nb > 2 * W) nb = 2 * W;
            } else if (nb > IFH_EMIT_CAP) {
                n_emit = 1;
                emit_ipos = astart;
                emit_len = IFH_EMIT_CAP;
                move_src = IFH_EMIT_CAP;
                move_n = nb - IFH_EMIT_CAP;
                nb = move_n;
                astart += IFH_EMIT_CAP;
                if (temp_end != 0 && temp_end < astart) temp_end = astart;
            }
        }
        st[0] = trig;
        st[1] = temp_end;
        st[2] = cur;
        st[3] = astart;
        buf_len[s] = (int)nb;
        int64_t *e = ev + (int64_t)i * 8;
        e[0] = kind;
        e[1] = pos;
        e[2] = (astart != -1) ? 1 : 0;
        e[3] = n_emit;
        e[4] = emit_ipos;
        e[5] = emit_len;
        e[6] = err;
        e[7] = 0;
        sh[0] = n_emit ? emit_len : 0;
        sh[1] = move_src;
        sh[2] = move_n;
    }
    __syncthreads();
    const int64_t elen = sh[0], msrc = sh[1], mn = sh[2];
    if (elen > 0) {
        const int64_t n4 = elen >> 2;  // row bases are 16-byte aligned
        const float4 *s4 = reinterpret_cast<const float4 *>(ab);
        float4 *d4 = reinterpret_cast<float4 *>(em);
        for (int64_t k = t; k < n4; k += 256) d4[k] = s4[k];
        for (int64_t k = (n4 << 2) + t; k < elen; k += 256) em[k] = ab[k];
    }
    if (mn > 0) {
        // move ab[msrc .. msrc+mn) to ab[0 .. mn); mn <= 1008, ranges may overlap: stage in registers
        float r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t k = t + 256 * q;
            r[q] = (k < mn) ? ab[msrc + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int64_t k = t + 256 * q;
            if (k < mn) ab[k] = r[q];
        }
    }
}

}  // namespace ifh

extern "C" int ifh_vad_energy_prob(const float *win, const int32_t *slot, int n, float *prob, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(win && slot && prob);
    hipLaunchKernelGGL(k_vad_energy_prob, dim3(n), dim3(64), 0, as_stream(stream), win, slot, prob);
    IFH_LAUNCH_CHECK("vad_energy_prob");
    return IFH_OK;
}

extern "C" int ifh_vad_fsm_step(const float *prob, const int32_t *slot, int n, int window, int sample_rate,
                                double threshold, int64_t *st_i64, int64_t *ev2, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(prob && slot && st_i64 && ev2 && window > 0);
    IFH_CHECK_ARG(sample_rate == 8000 || sample_rate == 16000);
    hipLaunchKernelGGL(k_vad_fsm, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), prob, slot, n, window,
                       sample_rate, threshold, st_i64, ev2);
    IFH_LAUNCH_CHECK("vad_fsm");
    return IFH_OK;
}

extern "C" int ifh_vad_step(const float *win, const float *prob, const int32_t *slot, int n, int sample_rate,
                            double threshold, int64_t *st_i64, int32_t *buf_len, float *abuf, int64_t *ev,
                            float *emit, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(win && prob && slot && st_i64 && buf_len && abuf && ev && emit);
    IFH_CHECK_ARG(sample_rate == 8000 || sample_rate == 16000);
    hipLaunchKernelGGL(k_vad_step, dim3(n), dim3(256), 0, as_stream(stream), win, prob, slot, sample_rate, threshold,
                       st_i64, buf_len, abuf, ev, emit);
    IFH_LAUNCH_CHECK("vad_step");
    return IFH_OK;
}

// =========================================================================================
// Block driver: T consecutive ticks of n calls from frames already resident in HBM -- the loop of
// RTP/InfernRTPIngest.py:63-100 (per packet: VADChannel.ingest -> SileroVADWorker.process_batch) expressed
// as the same per-tick / per-window launches as ifh_ingest_tick + ifh_vad_energy_prob + ifh_vad_step,
// issued from this one call (no interpreter between launches).  After every window the event table is read
// back (the reference's .tolist() sync); chunks emitted by the state machine are appended to `arena`
// (device) and logged on the host.  All n slots must hold the same number of FIFO bytes on entry (calls
// ticking in lock-step), as they do when every call receives one frame per tick.
// =========================================================================================
// vad_w == nullptr: the energy rule; else the recurrent network (vadnet.hip) with its per-call state vad_h / vad_c [2][n][64]
static int ingest_block_impl(const uint8_t *frames, int nticks, const int32_t *slot, int n, uint8_t *fifo,
                             int32_t *fifo_len, float *win, int32_t *win_ready, float *hist, float *pcm8k,
                             float *pcm16k, ifh_resampler_t rs8to16, float *prob, int sample_rate, double threshold,
                             int64_t *st_i64, int32_t *buf_len, float *abuf, int64_t *ev, float *emit, float *arena,
                             int64_t arena_cap, int64_t *log4, int log_cap, int *nlog, int64_t *arena_used,
                             const float *vad_w, float *vad_h, float *vad_c, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0 && nticks >= 0 && nlog && arena_used);
    *nlog = 0;
    *arena_used = 0;
    if (n == 0 || nticks == 0) return IFH_OK;
    IFH_CHECK_ARG(frames && slot && fifo && fifo_len && win && win_ready && hist && pcm8k && pcm16k && rs8to16);
    IFH_CHECK_ARG(prob && st_i64 && buf_len && abuf && ev && emit && arena && log4 && log_cap > 0 && arena_cap > 0);
    IFH_CHECK_ARG(rs8to16->orig == 1 && rs8to16->nw == 2 && rs8to16->ntaps == 15);
    IFH_CHECK_ARG(sample_rate == 8000 || sample_rate == 16000);
    hipStream_t st = as_stream(stream);
    std::vector<int32_t> hslot(n);
    // event tables come back through pinned memory and are looked at just before the NEXT window's VAD launches: the
    // tick launch enqueued in between keeps the stream busy while the host decides (the tick kernel does not touch
    // the emit rows, so the chunks of window w are copied out before anything can overwrite them).
    struct Pinned {
        int64_t *buf = nullptr;
        size_t cap = 0;
        hipEvent_t ev[2] = {nullptr, nullptr};
    };
    static thread_local Pinned pin;
    if (pin.cap < (size_t)n * 16) {
        if (pin.buf) (void)hipHostFree(pin.buf);
        pin.buf = nullptr;
        pin.cap = 0;
        if (hipHostMalloc((void **)&pin.buf, (size_t)n * 16 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
            return fail(IFH_EHIP, "ingest_block: pinned event buffer");
        pin.cap = (size_t)n * 16;
    }
    for (int k = 0; k < 2; k++)
        if (!pin.ev[k] && hipEventCreateWithFlags(&pin.ev[k], hipEventDisableTiming) != hipSuccess)
            return fail(IFH_EHIP, "ingest_block: event");
    int32_t fill0 = 0;
    hipError_t e = hipMemcpyAsync(hslot.data(), slot, (size_t)n * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipMemcpyAsync(&fill0, fifo_len + hslot[0], 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return check_hip(e, "ingest_block setup");
    int64_t nbytes = fill0, used = 0;
    int nl = 0, nwin = 0;
    // consume the event table of window w (buffer w & 1): log + arena copies of its emitted chunks
    auto consume = [&](int w) -> int {
        hipError_t ee = hipEventSynchronize(pin.ev[w & 1]);
        if (ee != hipSuccess) return check_hip(ee, "ingest_block window");
        const int64_t *hev = pin.buf + (size_t)(w & 1) * n * 8;
        for (int i = 0; i < n; i++) {
            const int64_t *r = hev + (size_t)i * 8;
            if (r[6]) return fail(IFH_EINVAL, "ingest_block: VAD buffer invariant violated (SileroVAD.py:89/95-98)");
            if (!r[3]) continue;
            const int64_t len = r[5];
            if (nl >= log_cap || used + len > arena_cap) return fail(IFH_EINVAL, "ingest_block: chunk log / arena full");
            ee = hipMemcpyAsync(arena + used, emit + (int64_t)hslot[i] * IFH_EMIT_CAP, (size_t)len * 4, hipMemcpyDeviceToDevice, st);
            if (ee != hipSuccess) return check_hip(ee, "ingest_block emit copy");
            log4[4 * nl + 0] = i;
            log4[4 * nl + 1] = r[4];
            log4[4 * nl + 2] = len;
            log4[4 * nl + 3] = used;
            used += len;
            nl++;
        }
        return IFH_OK;
    };
    for (int t = 0; t < nticks;) {
        // the ticks up to (and including) the one that completes the next window go in one launch
        int k = (int)((IFH_VAD_WINDOW - nbytes + 159) / 160);
        k = k < 1 ? 1 : (k > nticks - t ? nticks - t : k);
        hipLaunchKernelGGL(k_ingest_tick, dim3(n), dim3(64), 0, st, frames + (int64_t)t * n * 160, slot, fifo, fifo_len, win,
                           win_ready, hist, pcm8k, pcm16k, rs8to16->d_taps, k, (int64_t)n * 160);
        t += k;
        nbytes += 160 * (int64_t)k;
        if (nbytes < IFH_VAD_WINDOW) continue;
        nbytes -= IFH_VAD_WINDOW;
        if (nwin > 0) {
            const int rc = consume(nwin - 1);
            if (rc != IFH_OK) return rc;
        }
        if (vad_w)
            launch_vadnet_slots(win, slot, n, vad_w, vad_h, vad_c, prob, st);
        else
            hipLaunchKernelGGL(k_vad_energy_prob, dim3(n), dim3(64), 0, st, win, slot, prob);
        hipLaunchKernelGGL(k_vad_step, dim3(n), dim3(256), 0, st, win, prob, slot, sample_rate, threshold, st_i64, buf_len,
                           abuf, ev, emit);
        e = hipMemcpyAsync(pin.buf + (size_t)(nwin & 1) * n * 8, ev, (size_t)n * 64, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipEventRecord(pin.ev[nwin & 1], st);
        if (e != hipSuccess) return check_hip(e, "ingest_block window");
        nwin++;
    }
    if (nwin > 0) {
        const int rc = consume(nwin - 1);
        if (rc != IFH_OK) return rc;
    }
    IFH_LAUNCH_CHECK("ingest_block");
    *nlog = nl;
    *arena_used = used;
    return IFH_OK;
}

extern "C" int ifh_ingest_block(const uint8_t *frames, int nticks, const int32_t *slot, int n, uint8_t *fifo,
                                int32_t *fifo_len, float *win, int32_t *win_ready, float *hist, float *pcm8k,
                                float *pcm16k, ifh_resampler_t rs8to16, float *prob, int sample_rate, double threshold,
                                int64_t *st_i64, int32_t *buf_len, float *abuf, int64_t *ev, float *emit, float *arena,
                                int64_t arena_cap, int64_t *log4, int log_cap, int *nlog, int64_t *arena_used,
                                ifh_stream_t stream)
{
    return ingest_block_impl(frames, nticks, slot, n, fifo, fifo_len, win, win_ready, hist, pcm8k, pcm16k, rs8to16, prob, sample_rate,
                             threshold, st_i64, buf_len, abuf, ev, emit, arena, arena_cap, log4, log_cap, nlog, arena_used, nullptr,
                             nullptr, nullptr, stream);
}

extern "C" int ifh_ingest_block_net(const uint8_t *frames, int nticks, const int32_t *slot, int n, uint8_t *fifo,
                                    int32_t *fifo_len, float *win, int32_t *win_ready, float *hist, float *pcm8k,
                                    float *pcm16k, ifh_resampler_t rs8to16, float *prob, int sample_rate, double threshold,
                                    int64_t *st_i64, int32_t *buf_len, float *abuf, int64_t *ev, float *emit, float *arena,
                                    int64_t arena_cap, int64_t *log4, int log_cap, int *nlog, int64_t *arena_used,
                                    const float *vad_weights, float *vad_h, float *vad_c, ifh_stream_t stream)
{
    IFH_CHECK_ARG(vad_weights && vad_h && vad_c);
    return ingest_block_impl(frames, nticks, slot, n, fifo, fifo_len, win, win_ready, hist, pcm8k, pcm16k, rs8to16, prob, sample_rate,
                             threshold, st_i64, buf_len, abuf, ev, emit, arena, arena_cap, log4, log_cap, nlog, arena_used, vad_weights,
                             vad_h, vad_c, stream);
}
