/*
 * oracle/dsp_oracle.c -- CPU restatement of the reference's DSP arithmetic on the
 * per-call speech path.  TEST INFRASTRUCTURE ONLY: this file is compiled into
 * oracle/liboracle.so and may be called from tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py -- never from the product (infernos_amd/).
 *
 * Pinning: every function here is checked in tests/test_oracle_*.py against golden
 * vectors captured by importing the reference in the build container
 * (tools/gen_golden.py -> tests/golden/).  Exception: the sinc resampler, whose
 * arithmetic lives in torchaudio (absent from the image and from /root/reference;
 * requirements.txt:7 "torchaudio>=2.0.0", unpinned) -- "parity unpinned" for that
 * function: it restates torchaudio's published algorithm
 * (functional._get_sinc_resample_kernel/_apply_sinc_resample_kernel), see
 * orc_resample() below.
 *
 * Build: make -C oracle     (gcc -O2 -ffp-contract=off; fmaf() calls are explicit so
 * that the accumulation order/rounding is identical to the HIP kernels).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- G.711 mu-law -------------------------------------------------------------
 * Reference: Core/Codecs/G711.py:7-19 builds two tables with the stdlib `audioop`
 * (lin2ulaw / ulaw2lin, 16-bit samples).  The closed forms below are the CCITT
 * G.711 algorithm audioop implements; tests check all 256 + 65536 entries against
 * the tables captured from the reference (tests/golden/g711_tables.npz).
 */
int16_t orc_ulaw2lin(uint8_t code)
{
    unsigned u = (~code) & 0xFFu;
    int t = (int)((u & 0x0Fu) << 3) + 0x84;
    t <<= (u & 0x70u) >> 4;
    return (int16_t)((u & 0x80u) ? (0x84 - t) : (t - 0x84));
}

uint8_t orc_lin2ulaw(int16_t v)
{
    static const int seg_end[8] = {0x3F, 0x7F, 0xFF, 0x1FF, 0x3FF, 0x7FF, 0xFFF, 0x1FFF};
    int p = v >> 2;                 /* 16-bit sample to 14-bit magnitude range */
    int mask, seg;
    if (p < 0) { p = -p; mask = 0x7F; } else { mask = 0xFF; }
    if (p > 8159) p = 8159;
    p += 33;
    for (seg = 0; seg < 8; seg++)
        if (p <= seg_end[seg]) break;
    if (seg >= 8) return (uint8_t)(0x7F ^ mask);
    return (uint8_t)(((seg << 4) | ((p >> (seg + 1)) & 0xF)) ^ mask);
}

/* G711Codec.decode (Core/Codecs/G711.py:34-42): table lookup, int16 -> float32,
 * divide by 32767.0 (one IEEE f32 division). */
void orc_g711_decode(const uint8_t *in, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++)
        out[i] = (float)orc_ulaw2lin(in[i]) / 32767.0f;
}

/* G711Codec.encode (Core/Codecs/G711.py:25-32): clamp(x*32767, -32768, 32767),
 * truncate toward zero to int16, +32768, table lookup. NaN clamps propagate in
 * torch; the reference never feeds NaN, the oracle maps NaN to 0 like the kernels. */
void orc_g711_encode(const float *in, uint8_t *out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        float s = in[i] * 32767.0f;
        if (!(s == s)) s = 0.0f;
        if (s < -32768.0f) s = -32768.0f;
        if (s > 32767.0f) s = 32767.0f;
        out[i] = orc_lin2ulaw((int16_t)(int)s);
    }
}

/* ---- polyphase sinc resampler --------------------------------------------------
 * Reference call sites: Core/AudioChunk.py:19-24 -> config/InfernGlobals.py:23-26
 * -> torchaudio.transforms.Resample(orig, new) (sinc_interp_hann, width 6, rolloff
 * 0.99).  Kernel construction (float64, cast to f32) is done by the caller
 * (oracle/dsp.py: sinc_kernel) and passed in as kern[new][ntaps].
 *
 *   xpad = [0]*width ++ x ++ [0]*(width+orig)
 *   y[n*new + p] = sum_{j<ntaps} kern[p][j] * xpad[n*orig + j],   n = 0..L/orig
 *   output truncated to ceil(new*L/orig)
 *
 * Accumulation order is FIXED: acc=0; for j ascending: acc = fmaf(k, x, acc).
 * The HIP kernel uses the same chain, which is what "bit-exact" means for this row.
 */
void orc_resample(const float *x, int64_t L, const float *kern, int orig, int nw,
                  int ntaps, int width, float *out, int64_t out_len)
{
    for (int64_t o = 0; o < out_len; o++) {
        int64_t n = o / nw;
        int p = (int)(o % nw);
        const float *k = kern + (int64_t)p * ntaps;
        float acc = 0.0f;
        int64_t base = n * orig - width;
        for (int j = 0; j < ntaps; j++) {
            int64_t idx = base + j;
            float xv = (idx >= 0 && idx < L) ? x[idx] : 0.0f;
            acc = fmaf(k[j], xv, acc);
        }
        out[o] = acc;
    }
}

/* ---- VAD hysteresis state machine ------------------------------------------------
 * Reference: Core/VAD/SileroVADUtils.py:105-130 (VADIteratorB.__call__ per channel).
 * State is {triggered, temp_end, current_sample}.  Events: ev_kind 0 none, 1 start,
 * 2 end; ev_pos the sample index.  thr=0.5, neg=thr-0.15 evaluated in double like
 * Python floats; min_sil = sr*100/1000, pad = sr*30/1000 (Python floats, here
 * doubles; int() truncation).
 */
typedef struct { int32_t triggered; int64_t temp_end; int64_t current_sample; } orc_vad_state;

void orc_vad_fsm_step(orc_vad_state *st, const double *prob, int n, int window,
                      int sr, double thr, int32_t *ev_kind, int64_t *ev_pos)
{
    const double min_sil = sr * 100 / 1000.0, pad = sr * 30 / 1000.0;
    for (int i = 0; i < n; i++) {
        orc_vad_state *c = &st[i];
        double p = prob[i];
        ev_kind[i] = 0; ev_pos[i] = 0;
        c->current_sample += window;
        if (p >= thr && c->temp_end) c->temp_end = 0;
        if (p >= thr && !c->triggered) {
            c->triggered = 1;
            double sp = (c->current_sample > window) ? pad : 0.0;
            ev_kind[i] = 1; ev_pos[i] = (int64_t)(c->current_sample - sp - window);
            continue;
        }
        if (p < thr - 0.15 && c->triggered) {
            if (!c->temp_end) c->temp_end = c->current_sample;
            if ((double)(c->current_sample - c->temp_end) < min_sil) continue;
            ev_kind[i] = 2; ev_pos[i] = (int64_t)(c->temp_end + pad - window);
            c->temp_end = 0; c->triggered = 0;
        }
    }
}

/* ---- Whisper log-mel, direct (slow, exact-order-free) restatement ------------------
 * Reference: Cluster/InfernSTTWorker.py:114 -> transformers WhisperFeatureExtractor
 * (feature_extraction_whisper.py:135-168, v5.15.0).  Double-precision direct DFT;
 * used to validate oracle/dsp.py:logmel (numpy rfft) on short inputs.
 * audio[L] f32 (already padded/truncated to nsamp by the caller), mel[201][nmel] f64.
 */
void orc_logmel_direct(const float *audio, int64_t nsamp, const double *mel, int nmel,
                       float *out /* [nmel][nframes] */, int64_t nframes)
{
    const int nfft = 400, hop = 160, nb = 201;
    static double win[400], ct[400], stt[400];
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < nfft; n++) {
        win[n] = 0.5 - 0.5 * cos(2.0 * PI * n / nfft);
        ct[n] = cos(2.0 * PI * n / nfft);
        stt[n] = sin(2.0 * PI * n / nfft);
    }
    double gmax = -1e300;
    for (int64_t f = 0; f < nframes; f++) {
        double fr[400], pw[201];
        for (int n = 0; n < nfft; n++) {
            int64_t s = f * hop + n - nfft / 2;
            if (s < 0) s = -s;                         /* reflect */
            if (s >= nsamp) s = 2 * (nsamp - 1) - s;
            fr[n] = (double)audio[s] * (double)(float)win[n];
        }
        for (int k = 0; k < nb; k++) {
            double re = 0, im = 0;
            for (int n = 0; n < nfft; n++) {
                int idx = (int)(((int64_t)k * n) % nfft);
                re += fr[n] * ct[idx]; im -= fr[n] * stt[idx];
            }
            pw[k] = re * re + im * im;
        }
        for (int m = 0; m < nmel; m++) {
            double acc = 0;
            for (int k = 0; k < nb; k++) acc += mel[k * nmel + m] * pw[k];
            if (acc < 1e-10) acc = 1e-10;
            double lg = log10(acc);
            if (lg > gmax) gmax = lg;
            out[(int64_t)m * nframes + f] = (float)lg;
        }
    }
    for (int64_t i = 0; i < (int64_t)nmel * nframes; i++) {
        double v = out[i];
        if (v < gmax - 8.0) v = gmax - 8.0;
        out[i] = (float)((v + 4.0) / 4.0);
    }
}
