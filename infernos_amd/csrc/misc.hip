// misc.hip -- small fused glue kernels around the GEMM/attention core:
// token embedding (+positions), greedy argmax / softmax pick over the vocabulary, and the
// TTS streaming glue of HelloSippyRTPipe.infer (HelloSippyRTPipe.py:191-240): stop-rule
// bookkeeping, carry+chunking (with HiFi-GAN input normalisation and the `.view`
// re-interpretation AmendmentNetwork1 applies, HelloSippyRT.py:223-224), the 32->1 output
// conv + tanh of HiFi-GAN, and the amendment gain/tanh epilogue with chunk un-stacking.
#include <math.h>

#include "common.h"

namespace ifh {

// out[i][:] = table[ids[i]][:] + (pos_table ? pos_table[pos0 + (i % T)][:] : 0), D % 8 == 0
__global__ __launch_bounds__(256) void k_embed(const int32_t *__restrict__ ids, const uint16_t *__restrict__ table,
                                               const uint16_t *__restrict__ pos_table, int pos0, int T, int D,
                                               int n, uint16_t *__restrict__ out, const int32_t *__restrict__ dyn,
                                               int dyn_ids_mul)
{
    const int vecs = D / 8;
    if (dyn) {
        pos0 += dyn[0];
        ids += (int64_t)dyn[0] * dyn_ids_mul;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * vecs;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / vecs), c = (int)(e % vecs) * 8;
        uint4 a = *reinterpret_cast<const uint4 *>(table + (int64_t)ids[i] * D + c);
        if (pos_table) {
            const uint4 b = *reinterpret_cast<const uint4 *>(pos_table + (int64_t)(pos0 + i % T) * D + c);
            uint32_t *ua = reinterpret_cast<uint32_t *>(&a);
            const uint32_t *ub = reinterpret_cast<const uint32_t *>(&b);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float lo = __uint_as_float(ua[k] << 16) + __uint_as_float(ub[k] << 16);
                const float hi = __uint_as_float(ua[k] & 0xffff0000u) + __uint_as_float(ub[k] & 0xffff0000u);
                ua[k] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
            }
        }
        *reinterpret_cast<uint4 *>(out + (int64_t)i * D + c) = a;
    }
}

// per row: argmax (first index on ties) and optionally softmax probability of `pick`
__global__ __launch_bounds__(256) void k_argmax_pick(const float *__restrict__ logits, int64_t ld, int V, int pick,
                                                     int32_t *__restrict__ arg, float *__restrict__ prob,
                                                     const int32_t *__restrict__ dyn, int dyn_out_mul)
{
    if (dyn && arg) arg += (int64_t)dyn[0] * dyn_out_mul;
    __shared__ float smax[4];
    __shared__ int sidx[4];
    __shared__ float ssum[4];
    const float *row = logits + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float best = -INFINITY;
    int bi = 0x7fffffff;
#define AM_CMP(VAL, IDX)                                   \
    {                                                      \
        const float v_ = (VAL);                            \
        const int i_ = (IDX);                              \
        if (v_ > best || (v_ == best && i_ < bi)) {        \
            best = v_;                                     \
            bi = i_;                                       \
        }                                                  \
    }
    if (((reinterpret_cast<uintptr_t>(row) | (uintptr_t)(ld * 4)) & 15) == 0) {
        // 16-byte rows: 8 independent float4 loads in flight per thread (a scalar load-compare loop pays one memory
        // round trip per 256 elements: 72 us per 51 865-wide row)
        const float4 *row4 = reinterpret_cast<const float4 *>(row);
        const int V4 = V >> 2;
        for (int base = 0; base < V4; base += 256 * 8) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int j = base + tid + 256 * u;
                q[u] = j < V4 ? row4[j] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i0 = 4 * (base + tid + 256 * u);
                AM_CMP(q[u].x, i0) AM_CMP(q[u].y, i0 + 1) AM_CMP(q[u].z, i0 + 2) AM_CMP(q[u].w, i0 + 3)
            }
        }
        for (int i = 4 * V4 + tid; i < V; i += 256) AM_CMP(row[i], i)
    } else {
        for (int i = tid; i < V; i += 256) AM_CMP(row[i], i)
    }
#undef AM_CMP
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    if (lane == 0) {
        smax[wid] = best;
        sidx[wid] = bi;
    }
    __syncthreads();
    float gm = smax[0];
    int gi = sidx[0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        if (smax[w] > gm || (smax[w] == gm && sidx[w] < gi)) {
            gm = smax[w];
            gi = sidx[w];
        }
    }
    if (tid == 0 && arg) arg[blockIdx.x] = gi;
    if (prob) {
        float s = 0.0f;
        for (int i = tid; i < V; i += 256) s += expf(row[i] - gm);
        s = wave_sum(s);
        if (lane == 0) ssum[wid] = s;
        __syncthreads();
        if (tid == 0) prob[blockIdx.x] = expf(row[pick] - gm) / (ssum[0] + ssum[1] + ssum[2] + ssum[3]);
    }
}

// HelloSippyRTPipe.py:227-228: ends_at = where(ends_at<0 & minlen<=idx & (any(sigmoid>=thr) | maxlen<=idx), idx+2, ends_at)
__global__ void k_tts_stop(const float *__restrict__ logits /* [B][ld], 2 used */, int64_t *__restrict__ ends_at, int n,
                           int idx, int minlen, int maxlen, float thr, int ends_inc, const int32_t *__restrict__ dyn, int ld,
                           const int32_t *__restrict__ dyn_minmax)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    if (dyn) idx = dyn[0];
    if (dyn_minmax) { minlen = dyn_minmax[0]; maxlen = dyn_minmax[1]; }
    const float p0 = 1.0f / (1.0f + expf(-logits[ld * b])), p1 = 1.0f / (1.0f + expf(-logits[ld * b + 1]));
    const bool hit = (ends_at[b] < 0) && (minlen <= idx) && ((p0 >= thr) || (p1 >= thr) || (maxlen <= idx));
    if (hit) ends_at[b] = idx + ends_inc;
}

// single-block form that also advances the device-held step counter afterwards (one launch fewer per step)
__global__ __launch_bounds__(256) void k_tts_stop_advance(const float *__restrict__ logits, int64_t *__restrict__ ends_at,
                                                         int n, int minlen, int maxlen, float thr, int ends_inc,
                                                         int32_t *__restrict__ pos, int ld, uint4 *__restrict__ zbuf, int nz16,
                                                         const int32_t *__restrict__ dyn_minmax)
{
    for (int i = threadIdx.x; i < nz16; i += blockDim.x) zbuf[i] = make_uint4(0, 0, 0, 0);      // next step's LN statistics
    const int idx = pos[0];
    if (dyn_minmax) { minlen = dyn_minmax[0]; maxlen = dyn_minmax[1]; }
    for (int b = threadIdx.x; b < n; b += blockDim.x) {
        const float p0 = 1.0f / (1.0f + expf(-logits[ld * b])), p1 = 1.0f / (1.0f + expf(-logits[ld * b + 1]));
        const bool hit = (ends_at[b] < 0) && (minlen <= idx) && ((p0 >= thr) || (p1 >= thr) || (maxlen <= idx));
        if (hit) ends_at[b] = idx + ends_inc;
    }
    __syncthreads();
    if (threadIdx.x == 0) pos[0] = idx + 1;
}

// the stop rule over a ragged batch: every row has its own position, lengths and liveness (continuous batching)
__global__ __launch_bounds__(256) void k_tts_stop_advance_rows(const float *__restrict__ logits, int64_t *__restrict__ ends_at,
                                                              int n, float thr, int ends_inc, int32_t *__restrict__ pos,
                                                              const uint8_t *__restrict__ active,
                                                              const int32_t *__restrict__ minmax, int ld,
                                                              uint4 *__restrict__ zbuf, int nz16)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int i = tid; i < nz16; i += nt) zbuf[i] = make_uint4(0, 0, 0, 0);      // next step's LN statistics
    for (int b = tid; b < n; b += nt) {
        if (!active[b]) continue;
        const int idx = pos[b], minlen = minmax[2 * b], maxlen = minmax[2 * b + 1];
        const float p0 = 1.0f / (1.0f + expf(-logits[ld * b])), p1 = 1.0f / (1.0f + expf(-logits[ld * b + 1]));
        const bool hit = (ends_at[b] < 0) && (minlen <= idx) && ((p0 >= thr) || (p1 >= thr) || (maxlen <= idx));
        if (hit) ends_at[b] = idx + ends_inc;
        pos[b] = idx + 1;
    }
}

// frame 0 of the current frame buffer <- last frame of the previous one, zeros for rows that start now (pos == 0)
__global__ __launch_bounds__(256) void k_tts_carry_rows(const uint16_t *__restrict__ prev, uint16_t *__restrict__ cur,
                                                        const int32_t *__restrict__ pos, int n, int frames)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per 8 mel bins: 10 per row
    if (i >= n * 10) return;
    const int b = i / 10, c = i - b * 10;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (pos[b] != 0) v = *reinterpret_cast<const uint4 *>(prev + ((int64_t)b * frames + frames - 1) * 80 + 8 * c);
    *reinterpret_cast<uint4 *>(cur + (int64_t)b * frames * 80 + 8 * c) = v;
}

// HelloSippyRTPipe.py:231-235: S = cat(pre_frames[B,4,80], post[B,32,80]); pre_frames <- S[:, -4:];
// chunk i = S[:, 8i:8i+12] stacked chunk-major.  Emits
//   voc_in  [4B][12][80]  = (chunk - mean)/scale            (SpeechT5HifiGan normalize_before)
//   amd_mel [4B][12][80]  with amd_mel[n][t][c] = chunk_flat[n][c*12 + t]   (the .view at HelloSippyRT.py:224)
__global__ __launch_bounds__(256) void k_tts_chunks(uint16_t *__restrict__ pre_frames, const uint16_t *__restrict__ post,
                                                    const float *__restrict__ mean, const float *__restrict__ scale,
                                                    uint16_t *__restrict__ voc_in, uint16_t *__restrict__ amd_mel, int B,
                                                    const uint8_t *__restrict__ fresh)
{
    __shared__ uint16_t S[36 * 80];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool zero_pre = fresh && fresh[b];            // a row that starts with this call: carried frames are zeros
    for (int i = tid; i < 36 * 80; i += 256)
        S[i] = (i < 320) ? (zero_pre ? (uint16_t)0 : pre_frames[(int64_t)b * 320 + i]) : post[(int64_t)b * 2560 + (i - 320)];
    __syncthreads();
    for (int i = tid; i < 320; i += 256) pre_frames[(int64_t)b * 320 + i] = S[32 * 80 + i];
    for (int i = tid; i < 4 * 960; i += 256) {
        const int ch = i / 960, r = i % 960;
        const int64_t n = (int64_t)ch * B + b;
        const int t = r / 80, c = r % 80;
        const float v = bf16_to_f32(S[(8 * ch + t) * 80 + c]);
        voc_in[n * 960 + r] = f32_to_bf16((v - mean[c]) / scale[c]);
        // viewed[c'][t'] = flat[c'*12 + t'] -> channels-last position t'*80 + c'
        const int cp = r / 12, tp = r % 12;
        amd_mel[n * 960 + tp * 80 + cp] = S[8 * ch * 80 + r];
    }
}

// HiFi-GAN tail (modeling_speecht5.py:3058-3060): leaky_relu(0.01) -> Conv1d(32->1, k7, p3) -> tanh
// x bf16 [N][T][32] -> audio bf16 [N][T]
__global__ __launch_bounds__(256) void k_conv_post(const uint16_t *__restrict__ x, const float *__restrict__ w /* [7][32] */,
                                                   float bias, uint16_t *__restrict__ audio, int T, float slope)
{
    __shared__ float ws[7 * 32];
    const int n = blockIdx.y;
    if (threadIdx.x < 224) ws[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const uint16_t *xr = x + (int64_t)n * T * 32;
    float acc = bias;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        const int tt = t + k - 3;
        if (tt < 0 || tt >= T) continue;
        const uint4 *row = reinterpret_cast<const uint4 *>(xr + (int64_t)tt * 32);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 v = row[q];
            const uint32_t *u = reinterpret_cast<const uint32_t *>(&v);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float lo = __uint_as_float(u[e] << 16), hi = __uint_as_float(u[e] & 0xffff0000u);
                lo = lo > 0.0f ? lo : lo * slope;
                hi = hi > 0.0f ? hi : hi * slope;
                acc = __fmaf_rn(ws[k * 32 + q * 8 + 2 * e], lo, acc);
                acc = __fmaf_rn(ws[k * 32 + q * 8 + 2 * e + 1], hi, acc);
            }
        }
    }
    audio[(int64_t)n * T + t] = f32_to_bf16(tanhf(acc));
}

// AmendmentNetwork1 tail (HelloSippyRT.py:233-237) + un-stacking (HelloSippyRTPipe.py:238-239):
// post bf16 [4B][8][256] (channels-last, leaky-relu already applied), audio bf16 [4B][3072]
// out bf16 [B][4*2048]: out[b][ch*2048 + c*8 + t] = tanh(audio[n][512 + c*8 + t] * post[n][t][c]), n = ch*B + b
__global__ __launch_bounds__(256) void k_amend_final(const uint16_t *__restrict__ post, const uint16_t *__restrict__ audio,
                                                     uint16_t *__restrict__ out, int B)
{
    const int n = blockIdx.x;
    const int ch = n / B, b = n % B;
    for (int i = threadIdx.x; i < 2048; i += 256) {
        const int c = i >> 3, t = i & 7;
        const float g = bf16_to_f32(post[(int64_t)n * 2048 + t * 256 + c]);
        const float a = bf16_to_f32(audio[(int64_t)n * 3072 + 512 + i]);
        // the reference multiplies two bf16 tensors (bf16 result) and applies tanh in bf16
        const float prod = bf16_to_f32(f32_to_bf16(a * g));
        out[(int64_t)b * 8192 + ch * 2048 + i] = f32_to_bf16(tanhf(prod));
    }
}

// L2-normalise speaker x-vectors (F.normalize, eps 1e-12) into the concat buffer columns
__global__ __launch_bounds__(64) void k_l2norm_rows(const uint16_t *__restrict__ x, int D, uint16_t *__restrict__ out,
                                                    int ld_out)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    float s = 0.0f;
    for (int i = lane; i < D; i += 64) {
        const float v = bf16_to_f32(x[(int64_t)r * D + i]);
        s += v * v;
    }
    s = wave_sum(s);
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    for (int i = lane; i < D; i += 64) out[(int64_t)r * ld_out + i] = f32_to_bf16(bf16_to_f32(x[(int64_t)r * D + i]) * inv);
}

__global__ __launch_bounds__(256) void k_add_i32(int32_t *p, int delta, uint4 *__restrict__ zbuf, int nz16)
{
    for (int i = threadIdx.x; i < nz16; i += blockDim.x) zbuf[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x == 0) p[0] += delta;
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_add_i32(int32_t *value, int delta, void *zero_buf, int64_t zero_bytes, ifh_stream_t stream)
{
    IFH_CHECK_ARG(value && zero_bytes >= 0 && zero_bytes % 16 == 0 && zero_bytes < (1ll << 30) && (zero_bytes == 0 || zero_buf));
    IFH_CHECK_ARG((((uintptr_t)zero_buf) & 15) == 0);
    hipLaunchKernelGGL(k_add_i32, dim3(1), dim3(zero_bytes ? 256 : 64), 0, as_stream(stream), value, delta, (uint4 *)zero_buf,
                       (int)(zero_bytes / 16));
    IFH_LAUNCH_CHECK("add_i32");
    return IFH_OK;
}

extern "C" int ifh_embed_bf16(const int32_t *ids, const void *table, const void *pos_table, int pos0, int seq_len,
                              int dim, int n, void *out, const int32_t *dyn_pos, int dyn_ids_mul, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(ids && table && out && dim > 0 && dim % 8 == 0 && seq_len > 0);
    const int64_t work = (int64_t)n * (dim / 8);
    int grid = (int)((work + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_embed, dim3(grid), dim3(256), 0, as_stream(stream), ids, (const uint16_t *)table,
                       (const uint16_t *)pos_table, pos0, seq_len, dim, n, (uint16_t *)out, dyn_pos, dyn_ids_mul);
    IFH_LAUNCH_CHECK("embed");
    return IFH_OK;
}

extern "C" int ifh_argmax_pick_f32(const float *logits, int64_t ld, int vocab, int nrows, int pick_token,
                                   int32_t *argmax_out, float *pick_prob_out, const int32_t *dyn_pos, int dyn_out_mul,
                                   ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0);
    if (nrows == 0) return IFH_OK;
    IFH_CHECK_ARG(logits && vocab > 0 && ld >= vocab && (argmax_out || pick_prob_out));
    IFH_CHECK_ARG(!pick_prob_out || (pick_token >= 0 && pick_token < vocab));
    hipLaunchKernelGGL(k_argmax_pick, dim3(nrows), dim3(256), 0, as_stream(stream), logits, ld, vocab, pick_token,
                       argmax_out, pick_prob_out, dyn_pos, dyn_out_mul);
    IFH_LAUNCH_CHECK("argmax_pick");
    return IFH_OK;
}

extern "C" int ifh_tts_stop_update(const float *prob_logits, int64_t *ends_at, int n, int idx, int minlen, int maxlen,
                                   float threshold, int ends_inc, const int32_t *dyn_idx, int logits_ld,
                                   const int32_t *dyn_minmax, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(prob_logits && ends_at && logits_ld >= 2);
    hipLaunchKernelGGL(k_tts_stop, dim3((n + 63) / 64), dim3(64), 0, as_stream(stream), prob_logits, ends_at, n, idx,
                       minlen, maxlen, threshold, ends_inc, dyn_idx, logits_ld, dyn_minmax);
    IFH_LAUNCH_CHECK("tts_stop");
    return IFH_OK;
}

extern "C" int ifh_tts_stop_advance(const float *prob_logits, int64_t *ends_at, int n, int minlen, int maxlen,
                                    float threshold, int ends_inc, int32_t *pos, int logits_ld, void *zero_buf,
                                    int64_t zero_bytes, const int32_t *dyn_minmax, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0 && prob_logits && ends_at && pos && logits_ld >= 2);
    IFH_CHECK_ARG(zero_bytes >= 0 && zero_bytes % 16 == 0 && zero_bytes < (1ll << 30) && (zero_bytes == 0 || zero_buf));
    IFH_CHECK_ARG((((uintptr_t)zero_buf) & 15) == 0);
    hipLaunchKernelGGL(k_tts_stop_advance, dim3(1), dim3(256), 0, as_stream(stream), prob_logits, ends_at, n, minlen, maxlen,
                       threshold, ends_inc, pos, logits_ld, (uint4 *)zero_buf, (int)(zero_bytes / 16), dyn_minmax);
    IFH_LAUNCH_CHECK("tts_stop_advance");
    return IFH_OK;
}

extern "C" int ifh_tts_chunks_rows_bf16(void *pre_frames, const void *post, const float *mean, const float *scale,
                                        void *voc_in, void *amd_mel, const uint8_t *fresh, int nbatch, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(pre_frames && post && mean && scale && voc_in && amd_mel);
    hipLaunchKernelGGL(k_tts_chunks, dim3(nbatch), dim3(256), 0, as_stream(stream), (uint16_t *)pre_frames,
                       (const uint16_t *)post, mean, scale, (uint16_t *)voc_in, (uint16_t *)amd_mel, nbatch, fresh);
    IFH_LAUNCH_CHECK("tts_chunks");
    return IFH_OK;
}

extern "C" int ifh_tts_chunks_bf16(void *pre_frames, const void *post, const float *mean, const float *scale,
                                   void *voc_in, void *amd_mel, int nbatch, ifh_stream_t stream)
{
    return ifh_tts_chunks_rows_bf16(pre_frames, post, mean, scale, voc_in, amd_mel, nullptr, nbatch, stream);
}

extern "C" int ifh_tts_stop_advance_rows(const float *prob_logits, int64_t *ends_at, int n, float threshold, int ends_inc,
                                         int32_t *pos, const uint8_t *active, const int32_t *minmax, int logits_ld,
                                         void *zero_buf, int64_t zero_bytes, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0 && prob_logits && ends_at && pos && active && minmax && logits_ld >= 2);
    IFH_CHECK_ARG(zero_bytes >= 0 && zero_bytes % 16 == 0 && zero_bytes < (1ll << 30) && (zero_bytes == 0 || zero_buf));
    IFH_CHECK_ARG((((uintptr_t)zero_buf) & 15) == 0);
    // a few blocks: the statistics of a 1024-row step are 300 KB to clear
    hipLaunchKernelGGL(k_tts_stop_advance_rows, dim3(zero_bytes > 65536 ? 8 : 1), dim3(256), 0, as_stream(stream), prob_logits,
                       ends_at, n, threshold, ends_inc, pos, active, minmax, logits_ld, (uint4 *)zero_buf, (int)(zero_bytes / 16));
    IFH_LAUNCH_CHECK("tts_stop_advance_rows");
    return IFH_OK;
}

extern "C" int ifh_tts_carry_rows_bf16(const void *prev, void *cur, const int32_t *pos, int n, int frames, ifh_stream_t stream)
{
    IFH_CHECK_ARG(n >= 0);
    if (n == 0) return IFH_OK;
    IFH_CHECK_ARG(prev && cur && pos && frames >= 1 && (((uintptr_t)prev) & 15) == 0 && (((uintptr_t)cur) & 15) == 0);
    hipLaunchKernelGGL(k_tts_carry_rows, dim3((n * 10 + 255) / 256), dim3(256), 0, as_stream(stream), (const uint16_t *)prev,
                       (uint16_t *)cur, pos, n, frames);
    IFH_LAUNCH_CHECK("tts_carry_rows");
    return IFH_OK;
}

extern "C" int ifh_hifigan_post_bf16(const void *x, const float *w7x32, float bias, void *audio, int nrows, int t,
                                     float slope, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0 && t >= 0);
    if (nrows == 0 || t == 0) return IFH_OK;
    IFH_CHECK_ARG(x && w7x32 && audio && nrows < 65536);
    hipLaunchKernelGGL(k_conv_post, dim3((t + 255) / 256, nrows), dim3(256), 0, as_stream(stream), (const uint16_t *)x,
                       w7x32, bias, (uint16_t *)audio, t, slope);
    IFH_LAUNCH_CHECK("hifigan_post");
    return IFH_OK;
}

extern "C" int ifh_amend_final_bf16(const void *post, const void *audio, void *out, int nbatch, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nbatch >= 0);
    if (nbatch == 0) return IFH_OK;
    IFH_CHECK_ARG(post && audio && out);
    hipLaunchKernelGGL(k_amend_final, dim3(4 * nbatch), dim3(256), 0, as_stream(stream), (const uint16_t *)post,
                       (const uint16_t *)audio, (uint16_t *)out, nbatch);
    IFH_LAUNCH_CHECK("amend_final");
    return IFH_OK;
}

extern "C" int ifh_l2norm_rows_bf16(const void *x, int dim, int nrows, void *out, int ld_out, ifh_stream_t stream)
{
    IFH_CHECK_ARG(nrows >= 0);
    if (nrows == 0) return IFH_OK;
    IFH_CHECK_ARG(x && out && dim > 0 && ld_out >= dim);
    hipLaunchKernelGGL(k_l2norm_rows, dim3(nrows), dim3(64), 0, as_stream(stream), (const uint16_t *)x, dim,
                       (uint16_t *)out, ld_out);
    IFH_LAUNCH_CHECK("l2norm_rows");
    return IFH_OK;
}

// ---- spatial partition: CU-range streams and the persistent kernels' CU budget (infernos_hip.h) ----
namespace ifh {
std::atomic<int> g_cu_budget{0};
}

extern "C" int ifh_stream_create_cu_range(int first_cu, int n_cus, ifh_stream_t *stream_out)
{
    IFH_CHECK_ARG(stream_out);
    const int ncu = device_cu_count_physical();
    if (ncu <= 0) return fail(IFH_EHIP, "stream_create_cu_range: device query");
    IFH_CHECK_ARG(first_cu >= 0 && n_cus > 0 && first_cu + n_cus <= ncu);
    uint32_t mask[32] = {0};
    IFH_CHECK_ARG(ncu <= 32 * 32);
    for (int c = first_cu; c < first_cu + n_cus; c++) mask[c >> 5] |= 1u << (c & 31);
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)((ncu + 31) / 32), mask);
    if (e != hipSuccess) return check_hip(e, "hipExtStreamCreateWithCUMask");
    *stream_out = (ifh_stream_t)st;
    return IFH_OK;
}

extern "C" int ifh_stream_destroy(ifh_stream_t stream)
{
    IFH_CHECK_ARG(stream);
    hipError_t e = hipStreamDestroy(as_stream(stream));
    if (e != hipSuccess) return check_hip(e, "hipStreamDestroy");
    return IFH_OK;
}

extern "C" int ifh_set_cu_budget(int n)
{
    IFH_CHECK_ARG(n >= 0);
    g_cu_budget.store(n, std::memory_order_relaxed);
    return IFH_OK;
}
