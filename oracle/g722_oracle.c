/* G.722 sub-band ADPCM, CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The reference's Core/Codecs/G722.py:8-56 wraps the third-party `G722` extension module (sippy/libg722, requirements.txt,
 * unpinned; absent from /root/reference and from this image; the reference holds no test or vector for it): constructor
 * G722(8000, 64000), i.e. libg722's 8 kHz mode (G722_SAMPLE_RATE_8000) at 64 kbit/s -- one 8-bit code word per 8 kHz
 * sample, only the lower sub-band coded, no QMF.  PARITY UNPINNED: this file restates the published algorithm (ITU-T G.722
 * blocks 1L-6L and 4 as arranged in the public-domain spandsp/libg722 g722_encode.c / g722_decode.c) from its
 * specification; it cannot be checked against the module offline.  The 16 kHz two-band form (QMF + 2-bit upper band) is
 * included for completeness of the restatement and exercised by the round-trip tests.
 *
 * State layout (int32 words, shared with the HIP kernel): per band b (0 low, 1 high) at 45*b:
 *   s, sp, sz, r[3], a[3], ap[3], p[3], d[7], b[7], bp[7], sg[7], nb, det   = 45 words; then x[24] QMF history = 114 words,
 *   kept as 128 words per direction. */
#include <stdint.h>
#include <string.h>

#define G722_WORDS 128

typedef struct {
    int32_t s, sp, sz, r[3], a[3], ap[3], p[3], d[7], b[7], bp[7], sg[7], nb, det;
} band_t;

typedef struct {
    band_t band[2];
    int32_t x[24];
    int32_t pad[G722_WORDS - 2 * 45 - 24];
} g722_state_t;

static const int32_t qmf_fwd[12] = {3, -11, 12, 32, -210, 951, 3876, -805, 362, -156, 53, -11};
static const int32_t qmf_rev[12] = {-11, 53, -156, 362, -805, 3876, 951, -210, 32, 12, -11, 3};
static const int32_t qm2[4] = {-7408, -1616, 7408, 1616};
static const int32_t qm4[16] = {0, -20456, -12896, -8968, -6288, -4240, -2584, -1200, 20456, 12896, 8968, 6288, 4240, 2584, 1200, 0};
static const int32_t qm6[64] = {-136, -136, -136, -136, -24808, -21904, -19008, -16704, -14984, -13512, -12280, -11192, -10232, -9360,
                                -8576, -7856, -7192, -6576, -6000, -5456, -4944, -4464, -4008, -3576, -3168, -2776, -2400, -2032,
                                -1688, -1360, -1040, -728, 24808, 21904, 19008, 16704, 14984, 13512, 12280, 11192, 10232, 9360,
                                8576, 7856, 7192, 6576, 6000, 5456, 4944, 4464, 4008, 3576, 3168, 2776, 2400, 2032, 1688, 1360,
                                1040, 728, 432, 136, -432, -136};
static const int32_t q6[32] = {0, 35, 72, 110, 150, 190, 233, 276, 323, 370, 422, 473, 530, 587, 650, 714, 786, 858, 940, 1023, 1121,
                               1219, 1339, 1458, 1612, 1765, 1980, 2195, 2557, 2919, 0, 0};
static const int32_t iln[32] = {0, 63, 62, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8,
                                7, 6, 5, 4, 0};
static const int32_t ilp[32] = {0, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37,
                                36, 35, 34, 33, 32, 0};
static const int32_t ihn[3] = {0, 1, 0};
static const int32_t ihp[3] = {0, 3, 2};
static const int32_t wl[8] = {-60, -30, 58, 172, 334, 538, 1198, 3042};
static const int32_t rl42[16] = {0, 7, 6, 5, 4, 3, 2, 1, 7, 6, 5, 4, 3, 2, 1, 0};
static const int32_t ilb[32] = {2048, 2093, 2139, 2186, 2233, 2282, 2332, 2383, 2435, 2489, 2543, 2599, 2656, 2714, 2774, 2834, 2896,
                                2960, 3025, 3091, 3158, 3228, 3298, 3371, 3444, 3520, 3597, 3676, 3756, 3838, 3922, 4008};
static const int32_t wh[3] = {0, -214, 798};
static const int32_t rh2[4] = {2, 1, 2, 1};

static int32_t sat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

/* Block 4: predictor adaptation and the next signal estimate of one band, given its quantised difference d */
static void block4(band_t *bd, int32_t d)
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

static int32_t scalel(int32_t nb, int shift_base)
{
    const int32_t wd1 = (nb >> 6) & 31, wd2 = shift_base - (nb >> 11);
    const int32_t wd3 = (wd2 < 0) ? (ilb[wd1] << -wd2) : (ilb[wd1] >> wd2);
    return wd3 << 2;
}

void orc_g722_init(int32_t *state)
{
    g722_state_t *s = (g722_state_t *)state;
    memset(s, 0, sizeof(*s));
    s->band[0].det = 32;
    s->band[1].det = 8;
}

/* eight_k != 0: n samples at 8 kHz -> n code bytes (lower band only);  else: n (even) samples at 16 kHz -> n/2 bytes */
int64_t orc_g722_encode(int32_t *state, const int16_t *amp, int64_t n, int eight_k, uint8_t *out)
{
    g722_state_t *s = (g722_state_t *)state;
    int64_t j = 0, o = 0;
    while (j < n) {
        int32_t xlow, xhigh = 0, el, wd, wd1, wd2, ilow, ihigh = 0, ril, dlow, il4, i;
        if (eight_k) {
            xlow = amp[j++] >> 1;
        } else {
            int32_t sumeven = 0, sumodd = 0;
            memmove(s->x, s->x + 2, 22 * sizeof(int32_t));
            s->x[22] = amp[j++];
            s->x[23] = (j < n) ? amp[j++] : 0;
            for (i = 0; i < 12; i++) {
                sumodd += s->x[2 * i] * qmf_fwd[i];
                sumeven += s->x[2 * i + 1] * qmf_rev[i];
            }
            xlow = (sumeven + sumodd) >> 14;
            xhigh = (sumeven - sumodd) >> 14;
        }
        /* 1L: difference, 6-bit quantiser */
        el = sat16(xlow - s->band[0].s);
        wd = (el >= 0) ? el : -(el + 1);
        for (i = 1; i < 30; i++) {
            wd1 = (q6[i] * s->band[0].det) >> 12;
            if (wd < wd1) break;
        }
        ilow = (el < 0) ? iln[i] : ilp[i];
        /* 2L: inverse 4-bit quantiser;  3L: scale factor adaptation */
        ril = ilow >> 2;
        wd2 = qm4[ril];
        dlow = (s->band[0].det * wd2) >> 15;
        il4 = rl42[ril];
        wd = (s->band[0].nb * 127) >> 7;
        s->band[0].nb = wd + wl[il4];
        if (s->band[0].nb < 0) s->band[0].nb = 0;
        else if (s->band[0].nb > 18432) s->band[0].nb = 18432;
        s->band[0].det = scalel(s->band[0].nb, 8);
        block4(&s->band[0], dlow);
        if (eight_k) {
            out[o++] = (uint8_t)(0xC0 | ilow);          /* upper-band bits left at "11" */
        } else {
            int32_t eh, mih, dhigh, ih2;
            eh = sat16(xhigh - s->band[1].s);
            wd = (eh >= 0) ? eh : -(eh + 1);
            wd1 = (564 * s->band[1].det) >> 12;
            mih = (wd >= wd1) ? 2 : 1;
            ihigh = (eh < 0) ? ihn[mih] : ihp[mih];
            wd2 = qm2[ihigh];
            dhigh = (s->band[1].det * wd2) >> 15;
            ih2 = rh2[ihigh];
            wd = (s->band[1].nb * 127) >> 7;
            s->band[1].nb = wd + wh[ih2];
            if (s->band[1].nb < 0) s->band[1].nb = 0;
            else if (s->band[1].nb > 22528) s->band[1].nb = 22528;
            s->band[1].det = scalel(s->band[1].nb, 10);
            block4(&s->band[1], dhigh);
            out[o++] = (uint8_t)((ihigh << 6) | ilow);
        }
    }
    return o;
}

/* eight_k != 0: n code bytes -> n samples at 8 kHz;  else: n bytes -> 2n samples at 16 kHz */
int64_t orc_g722_decode(int32_t *state, const uint8_t *code_in, int64_t n, int eight_k, int16_t *amp)
{
    g722_state_t *s = (g722_state_t *)state;
    int64_t j, o = 0;
    for (j = 0; j < n; j++) {
        const int32_t code = code_in[j];
        int32_t wd1 = code & 0x3F, ihigh = (code >> 6) & 3, wd2 = qm6[wd1], rlow, dlowt, rhigh = 0;
        wd1 >>= 2;
        /* 5L: reconstructed low-band sample, 6L: limit */
        wd2 = (s->band[0].det * wd2) >> 15;
        rlow = s->band[0].s + wd2;
        if (rlow > 16383) rlow = 16383;
        else if (rlow < -16384) rlow = -16384;
        wd2 = qm4[wd1];
        dlowt = (s->band[0].det * wd2) >> 15;
        wd2 = rl42[wd1];
        wd1 = (s->band[0].nb * 127) >> 7;
        wd1 += wl[wd2];
        if (wd1 < 0) wd1 = 0;
        else if (wd1 > 18432) wd1 = 18432;
        s->band[0].nb = wd1;
        s->band[0].det = scalel(s->band[0].nb, 8);
        block4(&s->band[0], dlowt);
        if (eight_k) {
            amp[o++] = (int16_t)(rlow << 1);
            continue;
        }
        {
            int32_t dhigh, i, xout1 = 0, xout2 = 0;
            wd2 = qm2[ihigh];
            dhigh = (s->band[1].det * wd2) >> 15;
            rhigh = dhigh + s->band[1].s;
            if (rhigh > 16383) rhigh = 16383;
            else if (rhigh < -16384) rhigh = -16384;
            wd2 = rh2[ihigh];
            wd1 = (s->band[1].nb * 127) >> 7;
            wd1 += wh[wd2];
            if (wd1 < 0) wd1 = 0;
            else if (wd1 > 22528) wd1 = 22528;
            s->band[1].nb = wd1;
            s->band[1].det = scalel(s->band[1].nb, 10);
            block4(&s->band[1], dhigh);
            memmove(s->x, s->x + 2, 22 * sizeof(int32_t));
            s->x[22] = rlow + rhigh;
            s->x[23] = rlow - rhigh;
            for (i = 0; i < 12; i++) {
                xout2 += s->x[2 * i] * qmf_fwd[i];
                xout1 += s->x[2 * i + 1] * qmf_rev[i];
            }
            amp[o++] = (int16_t)sat16(xout1 >> 11);
            amp[o++] = (int16_t)sat16(xout2 >> 11);
        }
    }
    return o;
}
