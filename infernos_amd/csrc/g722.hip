// G.722 sub-band ADPCM (the reference's second negotiated codec: Core/Codecs/G722.py:8-56 over the third-party `G722`
// module, SIP/InfernUAS.py:50), batched over calls: the codec is a per-call recursion (predictor and scale-factor state carried
// from sample to sample), so one thread runs one call's frame and the batch supplies the parallelism -- a 20 ms frame is 160
// dependent steps of ~150 integer operations, ~0.1 ms, whatever the number of calls up to the thread count of the chip.
// Arithmetic: ITU-T G.722 blocks 1L-6L / 1H-5H / 4 and the 24-tap QMF as arranged in the public-domain spandsp / libg722 code;
// the reference constructs G722(8000, 64000): libg722's 8 kHz mode (one code byte per 8 kHz sample, lower band only).  PARITY
// UNPINNED against that module (absent here, no vectors in the reference): see DESIGN.md 2.
#include "common.h"

namespace ifh {

struct band_t {
    int32_t s, sp, sz, r[3], a[3], ap[3], p[3], d[7], b[7], bp[7], sg[7], nb, det;
};
struct g722_state_t {
    band_t band[2];
    int32_t x[24];
    int32_t pad[128 - 2 * 45 - 24];
};
static_assert(sizeof(g722_state_t) == 128 * 4, "state words (include/infernos_hip.h: IFH_G722_STATE_WORDS)");

__constant__ int32_t qmf_fwd[12] = {3, -11, 12, 32, -210, 951, 3876, -805, 362, -156, 53, -11};
__constant__ int32_t qmf_rev[12] = {-11, 53, -156, 362, -805, 3876, 951, -210, 32, 12, -11, 3};
__constant__ int32_t qm2[4] = {-7408, -1616, 7408, 1616};
__constant__ int32_t qm4[16] = {0, -20456, -12896, -8968, -6288, -4240, -2584, -1200, 20456, 12896, 8968, 6288, 4240, 2584, 1200, 0};
__constant__ int32_t qm6[64] = {-136, -136, -136, -136, -24808, -21904, -19008, -16704, -14984, -13512, -12280, -11192, -10232, -9360,
                                -8576, -7856, -7192, -6576, -6000, -5456, -4944, -4464, -4008, -3576, -3168, -2776, -2400, -2032,
                                -1688, -1360, -1040, -728, 24808, 21904, 19008, 16704, 14984, 13512, 12280, 11192, 10232, 9360,
                                8576, 7856, 7192, 6576, 6000, 5456, 4944, 4464, 4008, 3576, 3168, 2776, 2400, 2032, 1688, 1360,
                                1040, 728, 432, 136, -432, -136};
__constant__ int32_t q6[32] = {0, 35, 72, 110, 150, 190, 233, 276, 323, 370, 422, 473, 530, 587, 650, 714, 786, 858, 940, 1023, 1121,
                               1219, 1339, 1458, 1612, 1765, 1980, 2195, 2557, 2919, 0, 0};
__constant__ int32_t iln[32] = {0, 63, 62, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8,
                                7, 6, 5, 4, 0};
__constant__ int32_t ilp[32] = {0, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37,
                                36, 35, 34, 33, 32, 0};
__constant__ int32_t ihn[3] = {0, 1, 0};
__constant__ int32_t ihp[3] = {0, 3, 2};
__constant__ int32_t wl[8] = {-60, -30, 58, 172, 334, 538, 1198, 3042};
__constant__ int32_t rl42[16] = {0, 7, 6, 5, 4, 3, 2, 1, 7, 6, 5, 4, 3, 2, 1, 0};
__constant__ int32_t ilb[32] = {2048, 2093, 2139, 2186, 2233, 2282, 2332, 2383, 2435, 2489, 2543, 2599, 2656, 2714, 2774, 2834, 2896,
                                2960, 3025, 3091, 3158, 3228, 3298, 3371, 3444, 3520, 3597, 3676, 3756, 3838, 3922, 4008};
__constant__ int32_t wh[3] = {0, -214, 798};
__constant__ int32_t rh2[4] = {2, 1, 2, 1};

__device__ __forceinline__ int32_t sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

/* Block 4: predictor adaptation and the next signal estimate of one band, given its quantised difference d */
__device__ void block4(band_t *bd, int32_t d)
{
    int32_t wd1, wd2, wd3;
    int i;
    bd->d[0] = d;
    bd->r[0] = sat16(bd->s + d);
    bd->p[0] = sat16(bd->sz + d);
    for (i = 0; i < 3; i++) bd->sg[i] = bd->p[i] >> 15;
    wd1 = sat16(bd->a[1] * 4);
    wd2 = (bd->sg[0] == bd->sg[1]) ? -wd1 : wd1;
    if (wd2 > 32767) wd2 = 32767;
    wd3 = (bd->sg[0] == bd->sg[2]) ? 128 : -128;
    wd3 += (wd2 >> 7);
    wd3 += (bd->a[2] * 32512) >> 15;
    if (wd3 > 12288) wd3 = 12288;
    else if (wd3 < -12288) wd3 = -12288;
    bd->ap[2] = wd3;
    bd->sg[0] = bd->p[0] >> 15;
    bd->sg[1] = bd->p[1] >> 15;
    wd1 = (bd->sg[0] == bd->sg[1]) ? 192 : -192;
    wd2 = (bd->a[1] * 32640) >> 15;
    bd->ap[1] = sat16(wd1 + wd2);
    wd3 = sat16(15360 - bd->ap[2]);
    if (bd->ap[1] > wd3) bd->ap[1] = wd3;
    else if (bd->ap[1] < -wd3) bd->ap[1] = -wd3;
    wd1 = (d == 0) ? 0 : 128;
    bd->sg[0] = d >> 15;
    for (i = 1; i < 7; i++) {
        bd->sg[i] = bd->d[i] >> 15;
        wd2 = (bd->sg[i] == bd->sg[0]) ? wd1 : -wd1;
        wd3 = (bd->b[i] * 32640) >> 15;
        bd->bp[i] = sat16(wd2 + wd3);
    }
    for (i = 6; i > 0; i--) {
        bd->d[i] = bd->d[i - 1];
        bd->b[i] = bd->bp[i];
    }
    for (i = 2; i > 0; i--) {
        bd->r[i] = bd->r[i - 1];
        bd->p[i] = bd->p[i - 1];
        bd->a[i] = bd->ap[i];
    }
    wd1 = sat16(bd->r[1] + bd->r[1]);
    wd1 = (bd->a[1] * wd1) >> 15;
    wd2 = sat16(bd->r[2] + bd->r[2]);
    wd2 = (bd->a[2] * wd2) >> 15;
    bd->sp = sat16(wd1 + wd2);
    bd->sz = 0;
    for (i = 6; i > 0; i--) {
        wd1 = sat16(bd->d[i] + bd->d[i]);
        bd->sz += (bd->b[i] * wd1) >> 15;
    }
    bd->sz = sat16(bd->sz);
    bd->s = sat16(bd->sp + bd->sz);
}

__device__ __forceinline__ int32_t scalel(int32_t nb, int shift_base)
{
    const int32_t wd1 = (nb >> 6) & 31, wd2 = shift_base - (nb >> 11);
    const int32_t wd3 = (wd2 < 0) ? (ilb[wd1] << -wd2) : (ilb[wd1] >> wd2);
    return wd3 << 2;
}


__global__ void k_g722_init(g722_state_t *st, int n)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    g722_state_t s;
    int32_t *w = reinterpret_cast<int32_t *>(&s);
    for (int i = 0; i < 128; i++) w[i] = 0;
    s.band[0].det = 32;
    s.band[1].det = 8;
    st[c] = s;
}

// IN_F32: samples are f32 in [-1, 1], converted as the reference wrapper does (clamp(x * 32767, -32768, 32767) truncated)
template <bool IN_F32>
__global__ void k_g722_encode(g722_state_t *st, const void *in, int64_t in_stride, int nsamp, int eight_k, uint8_t *out,
                              int64_t out_stride, int n)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    g722_state_t s = st[c];
    const int16_t *pi = reinterpret_cast<const int16_t *>(in) + (int64_t)c * in_stride;
    const float *pf = reinterpret_cast<const float *>(in) + (int64_t)c * in_stride;
    uint8_t *po = out + (int64_t)c * out_stride;
    auto sample = [&](int j) -> int32_t {
        if (IN_F32) {
            float v = pf[j] * 32767.0f;
            v = fminf(fmaxf(v, -32768.0f), 32767.0f);
            return (int32_t)v;
        }
        return pi[j];
    };
    int j = 0, o = 0;
    while (j < nsamp) {
        int32_t xlow, xhigh = 0, el, wd, wd1, wd2, ilow, ihigh = 0, ril, dlow, il4, i;
        if (eight_k) {
            xlow = sample(j++) >> 1;
        } else {
            int32_t sumeven = 0, sumodd = 0;
            for (i = 0; i < 22; i++) s.x[i] = s.x[i + 2];
            s.x[22] = sample(j++);
            s.x[23] = (j < nsamp) ? sample(j++) : 0;
            for (i = 0; i < 12; i++) {
                sumodd += s.x[2 * i] * qmf_fwd[i];
                sumeven += s.x[2 * i + 1] * qmf_rev[i];
            }
            xlow = (sumeven + sumodd) >> 14;
            xhigh = (sumeven - sumodd) >> 14;
        }
        el = sat16(xlow - s.band[0].s);
        wd = (el >= 0) ? el : -(el + 1);
        for (i = 1; i < 30; i++) {
            wd1 = (q6[i] * s.band[0].det) >> 12;
            if (wd < wd1) break;
        }
        ilow = (el < 0) ? iln[i] : ilp[i];
        ril = ilow >> 2;
        wd2 = qm4[ril];
        dlow = (s.band[0].det * wd2) >> 15;
        il4 = rl42[ril];
        wd = (s.band[0].nb * 127) >> 7;
        s.band[0].nb = wd + wl[il4];
        if (s.band[0].nb < 0) s.band[0].nb = 0;
        else if (s.band[0].nb > 18432) s.band[0].nb = 18432;
        s.band[0].det = scalel(s.band[0].nb, 8);
        block4(&s.band[0], dlow);
        if (eight_k) {
            po[o++] = (uint8_t)(0xC0 | ilow);
        } else {
            int32_t eh, mih, dhigh, ih2;
            eh = sat16(xhigh - s.band[1].s);
            wd = (eh >= 0) ? eh : -(eh + 1);
            wd1 = (564 * s.band[1].det) >> 12;
            mih = (wd >= wd1) ? 2 : 1;
            ihigh = (eh < 0) ? ihn[mih] : ihp[mih];
            wd2 = qm2[ihigh];
            dhigh = (s.band[1].det * wd2) >> 15;
            ih2 = rh2[ihigh];
            wd = (s.band[1].nb * 127) >> 7;
            s.band[1].nb = wd + wh[ih2];
            if (s.band[1].nb < 0) s.band[1].nb = 0;
            else if (s.band[1].nb > 22528) s.band[1].nb = 22528;
            s.band[1].det = scalel(s.band[1].nb, 10);
            block4(&s.band[1], dhigh);
            po[o++] = (uint8_t)((ihigh << 6) | ilow);
        }
    }
    st[c] = s;
}

template <bool OUT_F32>
__global__ void k_g722_decode(g722_state_t *st, const uint8_t *in, int64_t in_stride, int nbytes, int eight_k, void *out,
                              int64_t out_stride, int n)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= n) return;
    g722_state_t s = st[c];
    const uint8_t *pi = in + (int64_t)c * in_stride;
    int16_t *po = reinterpret_cast<int16_t *>(out) + (int64_t)c * out_stride;
    float *pf = reinterpret_cast<float *>(out) + (int64_t)c * out_stride;
    int o = 0;
    auto emit = [&](int32_t v) {
        if (OUT_F32) pf[o++] = (float)v / 32767.0f;
        else po[o++] = (int16_t)v;
    };
    for (int j = 0; j < nbytes; j++) {
        const int32_t code = pi[j];
        int32_t wd1 = code & 0x3F, ihigh = (code >> 6) & 3, wd2 = qm6[wd1], rlow, dlowt, rhigh = 0;
        wd1 >>= 2;
        wd2 = (s.band[0].det * wd2) >> 15;
        rlow = s.band[0].s + wd2;
        if (rlow > 16383) rlow = 16383;
        else if (rlow < -16384) rlow = -16384;
        wd2 = qm4[wd1];
        dlowt = (s.band[0].det * wd2) >> 15;
        wd2 = rl42[wd1];
        wd1 = (s.band[0].nb * 127) >> 7;
        wd1 += wl[wd2];
        if (wd1 < 0) wd1 = 0;
        else if (wd1 > 18432) wd1 = 18432;
        s.band[0].nb = wd1;
        s.band[0].det = scalel(s.band[0].nb, 8);
        block4(&s.band[0], dlowt);
        if (eight_k) {
            emit((int16_t)(rlow << 1));
            continue;
        }
        int32_t dhigh, i, xout1 = 0, xout2 = 0;
        wd2 = qm2[ihigh];
        dhigh = (s.band[1].det * wd2) >> 15;
        rhigh = dhigh + s.band[1].s;
        if (rhigh > 16383) rhigh = 16383;
        else if (rhigh < -16384) rhigh = -16384;
        wd2 = rh2[ihigh];
        wd1 = (s.band[1].nb * 127) >> 7;
        wd1 += wh[wd2];
        if (wd1 < 0) wd1 = 0;
        else if (wd1 > 22528) wd1 = 22528;
        s.band[1].nb = wd1;
        s.band[1].det = scalel(s.band[1].nb, 10);
        block4(&s.band[1], dhigh);
        for (i = 0; i < 22; i++) s.x[i] = s.x[i + 2];
        s.x[22] = rlow + rhigh;
        s.x[23] = rlow - rhigh;
        for (i = 0; i < 12; i++) {
            xout2 += s.x[2 * i] * qmf_fwd[i];
            xout1 += s.x[2 * i + 1] * qmf_rev[i];
        }
        emit(sat16(xout1 >> 11));
        emit(sat16(xout2 >> 11));
    }
    st[c] = s;
}

}  // namespace ifh

using namespace ifh;

extern "C" int ifh_g722_init(int32_t *state, int ncalls, ifh_stream_t stream)
{
    IFH_CHECK_ARG(ncalls >= 0);
    if (ncalls == 0) return IFH_OK;
    IFH_CHECK_ARG(state && (((uintptr_t)state) & 15) == 0);
    hipLaunchKernelGGL(k_g722_init, dim3((ncalls + 63) / 64), dim3(64), 0, as_stream(stream), (g722_state_t *)state, ncalls);
    IFH_LAUNCH_CHECK("g722_init");
    return IFH_OK;
}

extern "C" int ifh_g722_encode(int32_t *state, const void *pcm, int pcm_f32, int64_t pcm_stride, int nsamples, int eight_k,
                               uint8_t *code, int64_t code_stride, int ncalls, ifh_stream_t stream)
{
    IFH_CHECK_ARG(ncalls >= 0 && nsamples >= 0);
    if (ncalls == 0 || nsamples == 0) return IFH_OK;
    IFH_CHECK_ARG(state && pcm && code && pcm_stride >= nsamples && (eight_k || nsamples % 2 == 0));
    IFH_CHECK_ARG(code_stride >= (eight_k ? nsamples : nsamples / 2));
    dim3 grid((ncalls + 63) / 64);
    if (pcm_f32)
        hipLaunchKernelGGL(k_g722_encode<true>, grid, dim3(64), 0, as_stream(stream), (g722_state_t *)state, pcm, pcm_stride, nsamples,
                           eight_k, code, code_stride, ncalls);
    else
        hipLaunchKernelGGL(k_g722_encode<false>, grid, dim3(64), 0, as_stream(stream), (g722_state_t *)state, pcm, pcm_stride, nsamples,
                           eight_k, code, code_stride, ncalls);
    IFH_LAUNCH_CHECK("g722_encode");
    return IFH_OK;
}

extern "C" int ifh_g722_decode(int32_t *state, const uint8_t *code, int64_t code_stride, int nbytes, int eight_k, void *pcm,
                               int pcm_f32, int64_t pcm_stride, int ncalls, ifh_stream_t stream)
{
    IFH_CHECK_ARG(ncalls >= 0 && nbytes >= 0);
    if (ncalls == 0 || nbytes == 0) return IFH_OK;
    IFH_CHECK_ARG(state && pcm && code && code_stride >= nbytes && pcm_stride >= (eight_k ? nbytes : 2 * nbytes));
    dim3 grid((ncalls + 63) / 64);
    if (pcm_f32)
        hipLaunchKernelGGL(k_g722_decode<true>, grid, dim3(64), 0, as_stream(stream), (g722_state_t *)state, code, code_stride, nbytes,
                           eight_k, pcm, pcm_stride, ncalls);
    else
        hipLaunchKernelGGL(k_g722_decode<false>, grid, dim3(64), 0, as_stream(stream), (g722_state_t *)state, code, code_stride, nbytes,
                           eight_k, pcm, pcm_stride, ncalls);
    IFH_LAUNCH_CHECK("g722_decode");
    return IFH_OK;
}
