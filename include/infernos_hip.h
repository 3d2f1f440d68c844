/*
 * infernos_hip.h -- C ABI of libinfernos_hip.so, the MI355X (gfx950) implementation of
 * the Infernos per-call speech hot path.
 *
 * The reference (sippy/Infernos) has no FFI of its own: the path is reached through
 * duck-typed Python classes.  Every entry point below names the
 * reference interface it sits underneath (file:line under the reference tree); the
 * Python classes in infernos_amd/ keep the reference's names and call these through
 * ctypes on raw device pointers (torch.Tensor.data_ptr()) and the caller's hipStream_t.
 *
 * Conventions
 *   - every function returns 0 on success, a negative IFH_E* code on failure; the text of
 *     the last failure on the calling thread is ifh_last_error().
 *   - all pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void*; NULL is the legacy default stream.
 *   - nothing is allocated, freed or synchronised inside a launch function (graph-capture
 *     safe) unless the comment says "host sync".
 *   - no ownership crosses the boundary; handles are created/destroyed explicitly.
 *   - bf16 tensors are raw uint16_t bit patterns.
 */
#ifndef INFERNOS_HIP_H
#define INFERNOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IFH_OK 0
#define IFH_EINVAL (-1)   /* bad argument */
#define IFH_EHIP (-2)     /* HIP runtime error (text in ifh_last_error) */
#define IFH_ENOMEM (-3)
#define IFH_ERTPPARSE (-4) /* malformed RTP datagram (the reference's RTPParseError); nothing was changed */

typedef void *ifh_stream_t;

const char *ifh_last_error(void);
int ifh_version(void);                 /* 10000*major + 100*minor + patch */
int ifh_device_count(void);            /* does not initialise a context */

/* ---------------------------------------------------------------------------------
 * G.711 mu-law          replaces Core/Codecs/G711.py:7-47
 * ------------------------------------------------------------------------------- */
/* _ulaw_to_pcm_ct / _pcm_to_ulaw_ct (G711.py:7-19): the two tables, computed on the host
 * by the same closed form the kernels use. out256_host: int16[256]; out65536_host: u8[65536]. */
int ifh_g711_tables_host(int16_t *out256_host, uint8_t *out65536_host);

/* ---- G.722 (Core/Codecs/G722.py:8-56: the reference's second negotiated codec, SIP/InfernUAS.py:50; it wraps the third-party
 * `G722` module as G722(8000, 64000): 8 kHz mode, one code byte per 8 kHz sample, lower sub-band only).  Stateful sub-band
 * ADPCM: every call owns IFH_G722_STATE_WORDS int32 of state per direction (device memory, ifh_g722_init), carried from
 * frame to frame; one thread per call.  eight_k = 1: nsamples at 8 kHz <-> nsamples bytes; eight_k = 0 (the codec's native
 * form): nsamples (even) at 16 kHz <-> nsamples / 2 bytes through the 24-tap QMF.  pcm is int16 or (pcm_f32 = 1) f32 in
 * [-1, 1] converted as the reference wrapper does (clamp(x * 32767) truncated; decoded value / 32767).  Strides in elements. */
#define IFH_G722_STATE_WORDS 128
int ifh_g722_init(int32_t *state, int ncalls, ifh_stream_t stream);
int ifh_g722_encode(int32_t *state, const void *pcm, int pcm_f32, int64_t pcm_stride, int nsamples, int eight_k, uint8_t *code,
                    int64_t code_stride, int ncalls, ifh_stream_t stream);
int ifh_g722_decode(int32_t *state, const uint8_t *code, int64_t code_stride, int nbytes, int eight_k, void *pcm, int pcm_f32,
                    int64_t pcm_stride, int ncalls, ifh_stream_t stream);
/* G711Codec.decode (G711.py:34-42), resample=False: u8[n] -> f32[n] = lut[u]/32767.0f */
int ifh_g711_decode_u8_f32(const uint8_t *in, float *out, int64_t n, ifh_stream_t stream);
/* G711Codec.encode (G711.py:25-32): f32[n] -> u8[n] */
int ifh_g711_encode_f32_u8(const float *in, uint8_t *out, int64_t n, ifh_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Sinc resampler        replaces Core/AudioChunk.py:19-24 + config/InfernGlobals.py:23-26
 *                        (torchaudio.transforms.Resample(orig, new), sinc_interp_hann)
 * ------------------------------------------------------------------------------- */
typedef struct ifh_resampler *ifh_resampler_t;
int ifh_resample_create(int orig_sr, int new_sr, ifh_resampler_t *out);       /* host sync */
int ifh_resample_destroy(ifh_resampler_t h);
/* taps_host[new_r*ntaps] float32; returns geometry (orig_r/new_r are gcd-reduced) */
int ifh_resample_info(ifh_resampler_t h, int *orig_r, int *new_r, int *ntaps, int *width,
                      float *taps_host /* may be NULL */);
int64_t ifh_resample_out_len(ifh_resampler_t h, int64_t in_len);              /* ceil(new*L/orig) */
/* N independent rows: in[r*in_stride .. +len_r), out[r*out_stride .. +out_len(len_r)).
 * lens may be NULL (every row has max_len samples).  Zero padding at both row ends, as
 * torchaudio does for a whole chunk. */
int ifh_resample_run(ifh_resampler_t h, const float *in, int64_t in_stride, const int32_t *lens,
                     int64_t max_len, int nrows, float *out, int64_t out_stride, ifh_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Per-tick ingest       replaces RTP/InfernRTPIngest.py:98 -> Core/VAD/SileroVAD.py:27-35
 *                        (VADChannel.ingest: byte FIFO -> 768-sample decoded windows) fused
 *                        with G711Codec.decode and a streaming 8k->16k resample.
 * One call = one 20 ms tick of n calls.  State arrays are indexed by slot[i]:
 *   fifo      u8  [ncap][IFH_FIFO_CAP]   pending encoded bytes
 *   fifo_len  i32 [ncap]
 *   win       f32 [ncap][768]            last completed decoded window (8 kHz)
 *   win_ready i32 [ncap]                 set to 1 when `win` was refreshed by this tick
 *   hist      f32 [ncap][16]             last 15 decoded samples (streaming resample carry)
 * Outputs for this tick, row i = call slot[i]:
 *   pcm8k  f32 [n][160]   decoded frame
 *   pcm16k f32 [n][320]   streaming resample; sample k of tick t is y[2*(160*t-8)+k] of the
 *                         whole-stream resample (8-sample look-ahead delay), bit-identical taps
 * ------------------------------------------------------------------------------- */
#define IFH_FIFO_CAP 1024
#define IFH_VAD_WINDOW 768
int ifh_ingest_tick(const uint8_t *frames /* [n][160] */, const int32_t *slot, int n,
                    uint8_t *fifo, int32_t *fifo_len, float *win, int32_t *win_ready, float *hist,
                    float *pcm8k, float *pcm16k, ifh_resampler_t rs8to16, ifh_stream_t stream);

/* ---------------------------------------------------------------------------------
 * RTP ingress stage     (host only, no stream) replaces the per-packet Python of
 *                        RTP/InfernRTPIngest.py:63-100: `RtpJBuf(jb_size=8).udp_in(data)` (third-party
 *                        rtpsynth, absent: restated from its call site, PARITY UNPINNED),
 *                        the ERS -> `codec.silence(ts_diff)` fill (:84-86) and the hand-over of payload
 *                        bytes to the per-call VAD FIFO (:96), for a table of calls.
 * ifh_rtp_parse          RFC 3550 fixed header (+CSRC list, header extension, padding).  Malformed ->
 *                        IFH_ERTPPARSE (RTPParseError at :76-79).
 * ifh_rtpjb_create       n_streams calls; depth = packets a call may hold while waiting for a gap to fill
 *                        (RTPInStream.jb_size = 8, :32); frame_bytes = bytes per tick frame (160);
 *                        ts_per_byte = RTP timestamp units per payload byte (1 for G.711 at 8 kHz);
 *                        fill_byte = the codec's silence byte (0xFF, G711.py:61-62); fifo_cap = bytes of
 *                        released payload a call may hold before ifh_rtpjb_pop_tick takes them.
 * ifh_rtpjb_push         one datagram of one call.  Released frames, in order, are appended to the call's
 *                        byte FIFO and (when recs != NULL) reported: an in-order packet passes straight
 *                        through followed by every held packet that now continues the sequence; a late or
 *                        duplicate packet is dropped; an early one is held, and once more than `depth` are
 *                        held the gap in front is given up: ONE IFH_RTP_FRAME_ERS record {lseq_start,
 *                        lseq_end, ts_diff = timestamp of the next packet - timestamp expected} and
 *                        ts_diff / ts_per_byte fill bytes, then the held run.  Consecutive records of a
 *                        call always satisfy lseq_start == previous lseq_end + 1 (the assert at :91).
 *                        payload (optional) receives the released RTP payloads back to back.
 * ifh_rtpjb_push_batch   n datagrams (buf + off[n+1], stream[n]); returns how many were refused, codes in status[n].
 *                        Threading: a table has no lock -- push*, pop_tick, reset_stream and stats of ONE table must
 *                        come from one thread (or be serialised by the caller); different tables are independent.
 * ifh_rtpjb_pop_tick     for every call holding >= frame_bytes: one frame into frames[k][frame_bytes] and its
 *                        index into slots[k] -- the host-side inputs of ifh_ingest_tick / ifh_ingest_block.
 * ifh_rtpjb_reset_stream a new jitter buffer for the call (WIStreamUpdate, :66-70); drop_fifo also empties
 *                        its released bytes.
 * ------------------------------------------------------------------------------- */
#define IFH_RTP_MAX_PAYLOAD 1472
#define IFH_RTP_FRAME_RTP 0
#define IFH_RTP_FRAME_ERS 1
typedef struct ifh_rtp_hdr {
    int32_t version, padding, extension, cc, marker, pt;
    uint32_t seq, ts, ssrc;
    int32_t payload_off, payload_len;
} ifh_rtp_hdr;
typedef struct ifh_rtp_rec {
    int32_t stream, type;            /* IFH_RTP_FRAME_* */
    int64_t lseq_start, lseq_end;    /* RTP: both the packet's extended sequence number */
    uint32_t ts, ts_diff;            /* ERS: ts = timestamp of the packet after the gap */
    int64_t payload_off;             /* offset into the payload buffer passed to ifh_rtpjb_push, -1 if none */
    int32_t payload_len;             /* ERS: number of fill bytes appended */
    ifh_rtp_hdr hdr;                 /* RTP only */
} ifh_rtp_rec;
enum {
    IFH_RTP_STAT_RECEIVED = 0, IFH_RTP_STAT_RELEASED, IFH_RTP_STAT_LATE, IFH_RTP_STAT_DUPLICATE, IFH_RTP_STAT_REORDERED,
    IFH_RTP_STAT_ERS_EVENTS, IFH_RTP_STAT_ERS_PACKETS, IFH_RTP_STAT_ERS_BYTES, IFH_RTP_STAT_PARSE_ERRORS,
    IFH_RTP_STAT_OVERFLOW_BYTES, IFH_RTP_STAT_FIFO_BYTES, IFH_RTP_STAT_HELD, IFH_RTP_STAT_LAST_LSEQ, IFH_RTP_NSTATS
};
typedef void *ifh_rtpjb_t;
int ifh_rtp_parse(const uint8_t *pkt, int len, ifh_rtp_hdr *out);
int ifh_rtpjb_create(int n_streams, int depth, int frame_bytes, int ts_per_byte, int fill_byte, int fifo_cap,
                     ifh_rtpjb_t *out);
int ifh_rtpjb_destroy(ifh_rtpjb_t h);
int ifh_rtpjb_reset_stream(ifh_rtpjb_t h, int stream, int drop_fifo);
int ifh_rtpjb_push(ifh_rtpjb_t h, int stream, const uint8_t *pkt, int len, ifh_rtp_rec *recs, int rec_cap,
                   uint8_t *payload, int64_t payload_cap, int *nrec);
int ifh_rtpjb_push_batch(ifh_rtpjb_t h, const uint8_t *buf, const int32_t *off, const int32_t *stream, int n,
                         int32_t *status);
int ifh_rtpjb_pop_tick(ifh_rtpjb_t h, uint8_t *frames, int32_t *slots, int cap, int *n_out);
int ifh_rtpjb_stats(ifh_rtpjb_t h, int stream, int64_t *stats /* [IFH_RTP_NSTATS] */);

/* ---------------------------------------------------------------------------------
 * Output mix + encode    replaces Core/OutputMuxer.py:75-85 (OutputMTMuxer.idle mix: zero-pad,
 *                        sum in track order, divide by the number of tracks) + G711Codec.encode, for n calls at once.
 * tracks f32 [n][ntracks][block_len]; present u8 [n][ntracks]; ndiv i32 [n]; out u8 [n][block_len];
 * has_out u8 [n] (0: no track had a block, nothing to send; 1 track: unchanged; >=2: sum / ndiv).
 * ------------------------------------------------------------------------------- */
int ifh_mux_encode_f32_u8(const float *tracks, const uint8_t *present, const int32_t *ndiv, int ncalls, int ntracks,
                          int block_len, uint8_t *out, uint8_t *has_out, ifh_stream_t stream);

/* ---------------------------------------------------------------------------------
 * VAD                   replaces Core/VAD/SileroVADUtils.py:105-130 (hysteresis FSM) and
 *                        Core/VAD/SileroVAD.py:81-112 (chunk assembly), batched on device.
 * The speech-probability model itself (Silero v3.1 JIT, third party, weights not
 * obtainable offline) is pluggable: callers pass prob[n].  ifh_vad_energy_prob is the
 * documented stand-in used by the benchmarks.
 * Per-slot state (SoA, indexed by slot[i]):
 *   st_i64 [ncap][4]  = {triggered, temp_end, current_sample, active_start(-1 = None)}
 *   buf_len i32[ncap] ; abuf f32 [ncap][IFH_ABUF_CAP]  (VADChannel.active_buffer)
 * Per-call-of-this-function outputs, row i:
 *   ev i64 [n][8] = {kind(0 none,1 start,2 end), pos, active(0/1) AFTER this window,
 *                    n_emit (0,1), emit_ipos, emit_len, err(0 ok / 1 = the reference's
 *                    assert at SileroVAD.py:89/95-98 would have fired), 0}
 *   emit f32 [ncap][IFH_EMIT_CAP] : the emitted VadAudioChunk audio (8 kHz) for slot[i]
 * ------------------------------------------------------------------------------- */
#define IFH_ABUF_CAP (240000 + 768)
#define IFH_EMIT_CAP 240000
int ifh_vad_energy_prob(const float *win /* [ncap][768] */, const int32_t *slot, int n,
                        float *prob /* [n] */, ifh_stream_t stream);
/* A recurrent speech-probability network SHAPED like the reference's detector (Core/VAD/SileroVAD.py:44-45: Silero VAD v3.1,
 * third party, not obtainable offline -- PARITY UNPINNED against it): conv front end + two LSTM(64) layers whose state, two
 * [2][n][64] tensors, is carried from window to window exactly as Core/VAD/SileroVADUtils.py:21-26,99,131 carry the model's.
 * weights: ifh_vadnet_weight_floats() floats in the layout of csrc/vadnet.hip (infernos_amd.weights.synth_vadnet packs it).
 * x f32 [n][768]; h_in / c_in / h_out / c_out f32 [2][n][64] (in and out may alias); prob f32 [n]. */
int ifh_vadnet_weight_floats(void);
int ifh_vadnet_prob(const float *x, int n, const float *weights, const float *h_in, const float *c_in, float *h_out,
                    float *c_out, float *prob, ifh_stream_t stream);
/* FSM only (VADIteratorB.__call__): updates st_i64[slot][0..2]; ev2 i64 [n][2] = {kind, pos} */
int ifh_vad_fsm_step(const float *prob /* [n] */, const int32_t *slot, int n, int window, int sample_rate,
                     double threshold, int64_t *st_i64, int64_t *ev2, ifh_stream_t stream);
/* FSM + chunk assembly for one 768-sample window per listed call */
int ifh_vad_step(const float *win /* [ncap][768] */, const float *prob /* [n] */, const int32_t *slot,
                 int n, int sample_rate, double threshold, int64_t *st_i64, int32_t *buf_len, float *abuf,
                 int64_t *ev, float *emit, ifh_stream_t stream);

/* Block driver for frames that are already resident: nticks consecutive 20 ms ticks of n calls, i.e. the loop of
 * RTP/InfernRTPIngest.py:63-100 over the three entry points above (the ticks up to the next completed 768-sample window
 * in one launch of the ifh_ingest_tick kernel, then ifh_vad_energy_prob + ifh_vad_step), issued from one host call.  frames u8
 * [nticks][n][160].  After every window the event table is read back (the reference's per-batch .tolist()); chunks
 * emitted by the state machine are appended to `arena` (device f32, arena_cap floats) and logged in log4 (host int64
 * [log_cap][4] = row in `slot`, ipos, length, arena offset).  All n slots must enter with equal FIFO fill.
 * Uses the built-in energy probability model; a pluggable network needs the per-window calls instead. */
int ifh_ingest_block(const uint8_t *frames, int nticks, const int32_t *slot, int n, uint8_t *fifo, int32_t *fifo_len,
                     float *win, int32_t *win_ready, float *hist, float *pcm8k, float *pcm16k, ifh_resampler_t rs8to16,
                     float *prob, int sample_rate, double threshold, int64_t *st_i64, int32_t *buf_len, float *abuf,
                     int64_t *ev, float *emit, float *arena, int64_t arena_cap, int64_t *log4, int log_cap, int *nlog,
                     int64_t *arena_used, ifh_stream_t stream);
/* The same driver with the recurrent network of ifh_vadnet_prob as the probability model (the reference runs its network on every
 * window of every call: Core/VAD/SileroVAD.py:78-80, SileroVADUtils.py:99,131): vad_weights = the ifh_vadnet_weight_floats() blob,
 * vad_h / vad_c f32 [2][n][64] = the per-call recurrent state in the order of `slot`, updated in place window by window. */
int ifh_ingest_block_net(const uint8_t *frames, int nticks, const int32_t *slot, int n, uint8_t *fifo, int32_t *fifo_len,
                         float *win, int32_t *win_ready, float *hist, float *pcm8k, float *pcm16k, ifh_resampler_t rs8to16,
                         float *prob, int sample_rate, double threshold, int64_t *st_i64, int32_t *buf_len, float *abuf,
                         int64_t *ev, float *emit, float *arena, int64_t arena_cap, int64_t *log4, int log_cap, int *nlog,
                         int64_t *arena_used, const float *vad_weights, float *vad_h, float *vad_c, ifh_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Whisper log-mel       replaces Cluster/InfernSTTWorker.py:114 (WhisperProcessor ->
 *                        WhisperFeatureExtractor, transformers feature_extraction_whisper.py:135-168)
 * audio f32 rows of `stride` samples, lens[b] valid samples (zero beyond, truncated to 480000);
 * out [B][n_mel][3000] f32 (out_bf16=0) or bf16 (out_bf16=1).  n_mel is 80 or 128.
 * workspace: ifh_logmel_workspace_floats() floats, any contents (per-utterance max; plus
 * the f32 staging plane when the output is bf16).
 * ------------------------------------------------------------------------------- */
typedef struct ifh_logmel *ifh_logmel_t;
int ifh_logmel_create(int n_mel, ifh_logmel_t *out);                           /* host sync */
int ifh_logmel_destroy(ifh_logmel_t h);
int ifh_logmel_filters_host(ifh_logmel_t h, float *out201xnmel_host);          /* [201][n_mel] */
int64_t ifh_logmel_workspace_floats(ifh_logmel_t h, int nbatch, int out_bf16);  /* B (+ B*n_mel*3000 if bf16) */
int ifh_logmel_run(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch,
                   void *out, int out_bf16, float *workspace, ifh_stream_t stream);
/* The same transform with the normalisation left to the consumer: raw f32 [B][n_mel][3000] = log10(max(mel power,
 * 1e-10)) and win_max int32 [B] = the per-utterance maximum of it, in the order-preserving integer encoding of a float
 * (i >= 0 ? i : i ^ 0x7fffffff).  ifh_logmel_finish_transpose_bf16 applies max(x, max - 8), (x + 4) / 4 while it
 * produces the bf16 channels-last [B][3000][n_mel] input of Whisper's first convolution: the normalised f32 plane is
 * never written (1.92 MB of the 4.8 MB per window ifh_logmel_run moves).  Bit-identical to ifh_logmel_run (f32) followed
 * by ifh_transpose_to_bf16. */
int ifh_logmel_run_raw(ifh_logmel_t h, const float *audio, int64_t stride, const int32_t *lens, int nbatch, float *raw,
                       int32_t *win_max, ifh_stream_t stream);
int ifh_logmel_finish_transpose_bf16(ifh_logmel_t h, const float *raw, const int32_t *win_max, int nbatch, void *out,
                                     ifh_stream_t stream);


/* ---------------------------------------------------------------------------------
 * Dense building blocks of the speech models (bf16 storage, fp32 accumulate).  These sit
 * under the third-party engines the reference calls from
 *   Cluster/InfernSTTWorker.py:61-107   (Whisper encoder/decoder; CTranslate2 / HF)
 *   HelloSippyTTSRT/HelloSippyRTPipe.py:111-115,196-237 (SpeechT5, HiFi-GAN; HF transformers)
 *   HelloSippyTTSRT/HelloSippyRT.py:219-237 (AmendmentNetwork1)
 * ------------------------------------------------------------------------------- */
#define IFH_ACT_NONE 0
#define IFH_ACT_RELU 1
#define IFH_ACT_GELU 2      /* exact erf form */
#define IFH_ACT_TANH 3
#define IFH_ACT_LRELU 4     /* slope = act_slope */
#define IFH_ACT_SIGMOID 5
#define IFH_ACT_SILU_GLU 6  /* weight rows interleaved (gate_j, up_j): out[m, j] = silu(acc[2j]) * acc[2j+1], n/2 columns written;
                            * decode-batch launches of wide layers only (16 < rows <= 64, n >= 8192), plain bf16 output */

/* One implicit-GEMM launch on the matrix cores: Linear, Conv1d, or one phase of a
 * ConvTranspose1d, channels-last.
 *   out[b][t*ostride+ooff][n] (+)= scale * ( mask_n( act( bias[n] + sum_{tap,ci} w[n][tap][ci] *
 *        pre(x[b][t*stride - pad + tap*dil][ci]) ) ) + resid )
 * x rows outside [0,t_in) read as zero (after pre()).  pre() = LeakyReLU(pre_slope), 1.0 = identity.
 * Element strides: x row r of batch b is x + b*x_bstride + r*lda; out/resid likewise with
 * (out_bstride, ldc) / (resid_bstride, resid_ld; both 0 broadcast one row).  cin % 8 == 0. */
typedef struct ifh_conv_desc {
    const void *x;          /* bf16 */
    int64_t x_bstride;
    int32_t lda;
    int32_t cin, taps, stride, dil, pad;
    int32_t t_in, t_out, nbatch;
    const void *w;          /* bf16 [n][taps][cin] */
    int32_t n;
    const float *bias;      /* [n] or NULL */
    const uint8_t *colmask; /* [n] or NULL: keep ? 2*v : 0  (SpeechT5 prenet dropout, p = 0.5) */
    float pre_slope;
    int32_t act;
    float act_slope;
    const void *resid;      /* bf16 or NULL */
    int64_t resid_bstride;
    int32_t resid_ld;
    float out_scale;
    int32_t accumulate;     /* add the previous contents of out */
    void *out;              /* bf16, or f32 when out_f32 */
    int32_t out_f32;
    int64_t out_bstride;
    int32_t ldc, ostride, ooff;
    /* optional device-resident scalar (e.g. the decoder position) so that a captured hipGraph can be
     * replayed for every step: effective ooff += dyn_pos[0]*dyn_ooff_mul, resid += dyn_pos[0]*dyn_resid_mul */
    const int32_t *dyn_pos;
    int32_t dyn_ooff_mul;
    int64_t dyn_resid_mul;
    /* optional second output region: columns n >= n_split (n_split % 16 == 0) are written to
     * out2[b*out2_bstride + (t*ostride + ooff2 + dyn_pos[0]*dyn_ooff2_mul)*ldc2 + (n - n_split)] -- lets one
     * launch produce q (scratch) and K|V (appended to the cache) */
    int32_t n_split;
    void *out2;
    int64_t out2_bstride;
    int32_t ldc2, ooff2, dyn_ooff2_mul;
    /* LayerNorm folded around a decode-step GEMM (rows <= 1024, taps == 1), so that the three LayerNorm launches
     * per transformer layer disappear from the latency-bound decode loop:
     *   stats_out int64 [rows][2]: += (sum, sum of squares) of every stored output row in 2^16 fixed point,
     *       by integer atomics (commutative, hence bit-reproducible); the caller zeroes it per step;
     *       consumers receive the same array as aln_stats / rln_stats;
     *   aln_stats/aln_c1: the A rows are LayerNorm inputs; w must hold W*diag(gamma), bias must hold
     *       W*beta + b, aln_c1[n] = sum_k w[n][k]; then out = rstd*(acc - mean*aln_c1[n]) + bias[n];
     *   rln_stats/rln_gamma/rln_beta: the residual rows are LayerNorm inputs, normalised on the fly.
     * ln_dim = LayerNorm width, ln_eps its epsilon. */
    const void *aln_stats;
    const float *aln_c1;
    const void *rln_stats;
    const float *rln_gamma, *rln_beta;
    void *stats_out;
    int32_t ln_dim;
    float ln_eps;
    int32_t ln_rms;        /* 1: the folded normalisation is an RMSNorm -- aln: out = rsqrt(mean(x^2) + eps) * acc (aln_c1 unused);
                            * stats_out as for LayerNorm (only the sum of squares is consumed) */
    int32_t dyn_stride;    /* 0: dyn_pos is one scalar for the launch.  1: one value per output row m = b*t_out + t
                            * (dyn_pos[m]): rows of a ragged decode batch sit at different positions -- each appends its K|V
                            * at its own cache row and adds its own positional-encoding row (continuous batching of the
                            * loop at HelloSippyRTPipe.py:195-229 across utterances that joined at different infer() calls) */
    int32_t decode_step;   /* 1: the launch belongs to a latency-bound decode step of up to 1024 rows: it takes the weight-streaming
                            * kernel whose K split depends on K alone (the one every launch of <= 256 rows takes), so that a row's
                            * bits do not depend on how many rows share the step -- a row of a 640-row ragged batch equals the
                            * same row decoded in a batch of 8 */
    int32_t convt_cout;    /* > 0: w is a ConvTranspose1d(k=8, stride=4, padding=2) in its fused one-launch form -- 3 taps, n =
                            * 4*convt_cout columns, column block r = output phase r, whose unused tap (tap 2 for r < 2, tap 0 for
                            * r >= 2) is all zeros: the kernel then skips those products (same bits, a third less matrix work) */
    void *splitk_ws;       /* optional f32 workspace OWNED BY THE CALLER (its decode state: one per stream / captured graph that may run
                            * concurrently), splitk_ws_floats >= 4 * rows * n: a deep narrow decode launch (17..64 rows, K >= 4096: the
                            * LLM's down projection) then runs as one accumulation chain per workgroup + a finishing pass (same bits as
                            * the streaming kernel, 1.7 x faster).  NULL: the streaming kernel.  The library holds no workspace of its own */
    int64_t splitk_ws_floats;
    void *argmax_keys;     /* optional uint64 [rows], ZERO before the launch: the launch also leaves, per output row, the key
                            * (order-preserving bits of the largest value << 32) | (0xffffffff - its column) -- the greedy pick of a
                            * vocabulary head without a second pass over the logits (Cluster/InfernLLMWorker.py:103-119: generate()'s
                            * arg-max).  Ties go to the lowest column.  Only where ifh_conv_argmax_supported says so (a wide f32-output
                            * matrix product of 17..64 rows without bias / activation / residual; a folded normalisation of the A rows
                            * (aln_stats) must be the RMS form, ln_rms = 1 -- IFH_EINVAL otherwise); ifh_argmax_keys_finish turns the
                            * keys into token ids and zeroes them again */
    int32_t whole_chip;    /* 1: a persistent kernel taking this launch (the 256 x 256 GEMM) uses every CU even when ifh_set_cu_budget
                            * reserves some for other stages -- for a caller that runs alone on the device for the moment (the LLM's
                            * prompt pass of an AI-attendant turn); 0: the budget applies (inside the speech pipeline, where it measured
                            * 1.7 % better) */
} ifh_conv_desc;
int ifh_conv_bf16(const ifh_conv_desc *desc, ifh_stream_t stream);
/* 1 if a launch of this shape (rows x n x k matrix product, f32 output) fills ifh_conv_desc.argmax_keys */
int ifh_conv_argmax_supported(int rows, int n, int k);
/* tokens[i] = the column held in keys[i] (0 if the key is still zero); keys[i] = 0 */
int ifh_argmax_keys_finish(void *keys, int32_t *tokens, int n, ifh_stream_t stream);

/* One HiFi-GAN residual pair in a single launch (transformers modeling_speecht5.py
 * HifiGanResidualBlock.forward loop body, called from SpeechT5HifiGan.forward :3040-3047 via
 * HelloSippyTTSRT/HelloSippyRTPipe.py:236):
 *     out = (x + conv2(lrelu(conv1(lrelu(x); taps, dil)); taps, 1)) * out_scale  (+ out if accumulate)
 * x, out bf16 [nbatch][t][c] channels-last with contiguous rows (batch strides in elements, % 8 == 0),
 * c in {32,64,128,256}, w1/w2 bf16 [c][taps][c], taps odd <= 15 (<= 11 at c = 32), "same" padding.  The intermediate is
 * rounded to bf16 exactly as two ifh_conv_bf16 launches would, so both routes give identical bits. */
typedef struct ifh_resblock_desc {
    const void *x;
    int64_t x_bstride;
    int32_t c, taps, dil, t, nbatch;
    const void *w1;
    const float *bias1;     /* [c] or NULL */
    const void *w2;
    const float *bias2;
    float slope;            /* LeakyReLU slope ahead of both convolutions, (0, 1] */
    float out_scale;
    int32_t accumulate;
    void *out;
    int64_t out_bstride;
} ifh_resblock_desc;
int ifh_resblock_pair_bf16(const ifh_resblock_desc *desc, ifh_stream_t stream);

/* One whole HiFi-GAN residual block (HifiGanResidualBlock.forward, modeling_speecht5.py; three dilation pairs) in a
 * single launch, bit-identical to three ifh_resblock_pair_bf16 launches with dilations 1, 3, 5:
 *     for d in (1, 3, 5): x = x + conv2_d(lrelu(conv1_d(lrelu(x); taps, d)); taps, 1)
 *     out = x * out_scale  (+ out if accumulate)
 * x, out bf16 [nbatch][t][c] channels-last (batch strides in elements, % 4 == 0); c in {32, 64, 128} (t <= 192 at
 * c = 128), taps in {3, 7, 11}.  wstream: the six convolutions' weights as one stream of MFMA fragments, padded with
 * zeros to whole 8 KB units -- for convolution q = 2*pair + {0: dilated, 1: plain}, k-step s (32 consecutive
 * k = tap*c + channel), fragment i (16 output channels), lane l: the 8 bf16
 *     w_q[16*i + (l & 15)][32*s + 8*(l >> 4) + 0..7]           (w_q as [c][taps][c] = [out][tap][in])
 * at element ((ks_q + s) * (c/16) + i) * 512 + l * 8, ks_q = 6 convolutions' k-steps before q; nunits =
 * ceil(6 * taps * (c/32) * (c/16) * 1024 / 8192).  bias f32 [6][c] in the same order.  (host packing:
 * infernos_amd.ops.w_chain_pack) */
typedef struct ifh_chain_desc {
    const void *x;
    int64_t x_bstride;
    int32_t c, taps, t, nbatch;
    const void *wstream;
    int32_t nunits;
    const float *bias;
    float slope;            /* LeakyReLU slope ahead of every convolution, (0, 1] */
    float out_scale;
    int32_t accumulate;
    void *out;
    int64_t out_bstride;
    void *debug_prof;       /* NULL, or device uint64[16] (zeroed by the caller): diagnostic shader-clock sums of wave 0 of every
                             * workgroup -- [0] tile set-up, [1+2q] K loop / [2+2q] epilogue+barrier of convolution q, [13] tiles,
                             * [14] workgroup lifetime, [15] workgroups (tools/probe_chain.py) */
    float post_slope;       /* LeakyReLU applied to what is stored, (0, 1]; 0 or 1 = none.  With the slope the CONSUMER would apply on load
                             * (the next upsampler's 0.1) the stored tensor is what that consumer multiplies, so it can run as a plain matrix
                             * product (round 6: the vocoder's upsamplers over guard rows); same bits either way */
} ifh_chain_desc;
int ifh_resblock_chain_bf16(const ifh_chain_desc *desc, ifh_stream_t stream);
/* The same residual block over WHOLE sequences held in ONE LDS image that every convolution overwrites in place (csrc/seq.hip):
 * nothing is recomputed at tile edges and only the weights stream.  Bit-identical to ifh_resblock_chain_bf16.  Shapes: (c, t) =
 * (64, 768), (128, 192) or (256, 48) -- the HiFi-GAN levels of a 12-frame chunk -- with taps 3, 7 or 11 (not 3 at c = 64:
 * ifh_resblock_seq_supported); wstream = the fragment
 * stream of ops.w_chain_pack zero padded to whole units of ifh_resblock_seq_unit_bytes(c) bytes (8 KB; 16 KB at c = 256), nunits
 * counted in those units. */
typedef struct ifh_seq_desc {
    const void *x;
    int64_t x_bstride;
    int32_t c, taps, t, nbatch;
    const void *wstream;
    int32_t nunits;
    const float *bias;
    float slope;            /* LeakyReLU slope ahead of every convolution, (0, 1] */
    float out_scale;
    int32_t accumulate;
    void *out;
    int64_t out_bstride;
    void *debug_prof;       /* NULL, or device uint64[8] (zeroed by the caller): diagnostic shader-clock sums of wave 0 of every workgroup --
                             * [0] tile top, [1] K loops, [2] waiting for the other waves behind a K loop, [3] conv1 / [4] conv2 epilogues,
                             * [5] last epilogue, [6] tiles, [7] workgroup lifetime (tools/probe_seq.py) */
    float post_slope;       /* as ifh_chain_desc.post_slope */
} ifh_seq_desc;
int ifh_resblock_seq_unit_bytes(int c);
int ifh_resblock_seq_supported(int c, int t, int taps);       /* 1 if ifh_resblock_seq_bf16 serves this shape */
int ifh_resblock_seq_bf16(const ifh_seq_desc *desc, ifh_stream_t stream);

/* The residual blocks of one HiFi-GAN upsampling level in ONE launch (csrc/level.hip), weights stationary in registers:
 *     for block j < nblocks (taps[j] in {3, 7, 11}):  y_j = chain_j(x)        (ifh_resblock_chain_bf16's function)
 *     out = y_0 * out_scale (+ out if accumulate);  out = y_j * out_scale + out  for j >= 1
 * i.e. what nblocks ifh_resblock_chain_bf16 launches with accumulate = (j > 0 || accumulate) compute, bit for bit
 * (transformers modeling_speecht5.py HifiGanResidualBlock / SpeechT5HifiGan.forward: the mean over the three blocks of a
 * level with out_scale = 1/3; reached from HelloSippyTTSRT/HelloSippyRTPipe.py:236).  wstream[j] / bias[j]: block j's
 * fragment stream and biases exactly as ifh_chain_desc takes them (ops.w_chain_pack).  c = 32; taps = (3, 7, 11) or one block. */
typedef struct ifh_level_desc {
    const void *x;
    int64_t x_bstride;
    int32_t c, t, nbatch, nblocks;
    int32_t taps[3];
    int32_t accumulate;
    const void *wstream[3];
    const float *bias[3];
    float slope;            /* LeakyReLU slope ahead of every convolution, (0, 1] */
    float out_scale;
    void *out;
    int64_t out_bstride;
    void *debug_prof;       /* NULL (diagnostic builds: device uint64[16], shader-clock sums per phase of a convolution) */
    /* Optional: the vocoder's last convolution folded in (SpeechT5HifiGan.forward: conv_post, 7 taps, 32 -> 1 channel, behind
     * LeakyReLU(post_slope), then tanh -- what ifh_hifigan_post_bf16 computes from `out`, with the same order of sums: same bits).
     * post_w f32 [7][32] != NULL: nblocks = 3, accumulate = 0; the level's mean stays on the chip (`out` is not written and may be
     * NULL), audio bf16 [nbatch][t] is; mean_ws is scratch of the caller's (one per stream / captured graph that may run
     * concurrently), mean_ws_bytes >= ifh_level_ws_bytes(): the blocks' running mean of the tile a workgroup is working on. */
    const float *post_w;
    float post_bias, post_slope;
    void *audio;
    void *mean_ws;
    int64_t mean_ws_bytes;
} ifh_level_desc;
int ifh_resblock_level_bf16(const ifh_level_desc *desc, ifh_stream_t stream);
int64_t ifh_level_ws_bytes(void);

/* One stride-1 "same" convolution with 256 input and 256 output channels on short sequences (t <= 48: the first
 * HiFi-GAN level), two sequences per workgroup, weights DMA'd as pre-packed fragments (the stream layout of
 * ifh_chain_desc for ONE convolution: taps * 8 k-steps of 16 fragments = taps * 8 units of 16 KB, no padding):
 *     out = ((conv(lrelu(x, pre_slope); taps, dil) + bias) (+ resid)) * out_scale  (+ out if accumulate)
 * i.e. what ifh_conv_bf16 computes for these shapes, with the same k order and the same rounding point (same bits).
 * (taps-1)/2 * dil <= 25. */
typedef struct ifh_ring256_desc {
    const void *x;
    int64_t x_bstride;
    int32_t taps, dil, t, nbatch;
    const void *wstream;
    const float *bias;      /* [256] or NULL */
    float pre_slope;        /* LeakyReLU slope on the input, (0, 1] (1 = none) */
    const void *resid;      /* bf16 [nbatch][t][256] or NULL */
    int64_t resid_bstride;
    float out_scale;
    int32_t accumulate;
    void *out;
    int64_t out_bstride;
} ifh_ring256_desc;
int ifh_conv_ring256_bf16(const ifh_ring256_desc *desc, ifh_stream_t stream);

/* y = LayerNorm(x (+ resid)) * gamma + beta; rows of `dim` bf16, dim <= 2048, dim % 4 == 0 */
int ifh_layernorm_bf16(const void *x, const void *resid, const float *gamma, const float *beta, void *out,
                       int rows, int dim, float eps, ifh_stream_t stream);
/* in [nbatch][rows][cols] (f32 if in_f32 else bf16) -> out bf16 [nbatch][cols][rows] */
int ifh_transpose_to_bf16(const void *in, int in_f32, void *out, int nbatch, int rows, int cols, ifh_stream_t stream);

/* Non-causal multi-head attention, head_dim 64, q pre-scaled.  Element strides (batch, token);
 * head h lives at +64*h.  key_len[b] (>=1) masks keys >= key_len; relbias f32
 * [nbatch][tq][nheads][nrel] adds relbias[q][clip(q-k, -nrel/2, nrel/2-1) + nrel/2]. */
typedef struct ifh_attn_desc {
    const void *q, *k, *v;
    void *out;
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
    int32_t nbatch, nheads, head_dim, tq, tk;
    const int32_t *key_len;
    const float *relbias;
    int32_t nrel;
} ifh_attn_desc;
int ifh_attn_prefill_bf16(const ifh_attn_desc *desc, ifh_stream_t stream);
/* one query token per (batch, head) against a KV cache.  Number of keys: key_len[b] + dyn_add if key_len is given
 * (per-row device counts: padded encoder lengths, or the rows' own decoder positions in a ragged batch), else
 * dyn_len[0] + dyn_add if dyn_len (device scalar; lets a captured graph be replayed per step), else
 * max_keys (which must bound the key count in every case). */
int ifh_attn_decode_bf16(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs, int64_t kv_ts,
                         void *out, int64_t o_bs, const int32_t *key_len, int max_keys, int nbatch, int nheads,
                         int head_dim, const int32_t *dyn_len, int dyn_add, ifh_stream_t stream);

/* ifh_attn_decode_bf16 for query rows that share cache rows: row b reads K/V row b / kv_group (the beams of one
 * utterance over its one cross-attention cache); every row attends over max_keys keys. */
int ifh_attn_decode_shared_bf16(const void *q, int64_t q_bs, const void *k, const void *v, int64_t kv_bs, int64_t kv_ts,
                                void *out, int64_t o_bs, int max_keys, int nbatch, int nheads, int head_dim, int kv_group,
                                ifh_stream_t stream);

/* ---- beam search over the per-token decode step ----
 * The search ctranslate2.models.Whisper.generate runs by default for Cluster/InfernSTTWorker.py:61-75 (beam_size 5,
 * length_penalty 1; the wheel is not vendored by the reference -- the bookkeeping is the one oracle/nn.py:beam_search
 * restates and pins).  Decode rows are batch-major: row = b * beams + k.  One call = one step at cur_len = pos[0]
 * tokens (prompt_len <= cur_len < max_length; other values are a no-op so that graph replays past the end are
 * harmless):
 *   log p = log_softmax(logits[row]) (+ suppress) (+ begin_suppress when cur_len == prompt_len);
 *   the 2*beams best of run_scores[b,k] + log p over (k, token), ties to the lower (k, token);
 *   a candidate ends when its token is eos_id or cur_len + 1 == max_length; the best `beams` that did not end become
 *   the running beams: their columns of toks (int32 [>= max_length][nbatch*beams], position-major) are permuted in
 *   place, the new token is written at toks[cur_len], run_scores updated, beam_src[row] = the row (of this step)
 *   each running row continues -- what ifh_kv_gather_bf16 consumes;
 *   ended candidates ranked inside the first `beams` compete, on score / (cur_len + 1 - prompt_len) ** length_penalty,
 *   with the finished hypotheses kept so far: fin_scores f32 / is_fin u8 / fin_len i32 [nbatch, beams] and fin_seqs i32
 *   [nbatch, beams, max_length - prompt_len] (generated tokens, eos included), best first;
 *   unsat[b] (int32, 1 on entry of the first step) drops to 0 once the best running beam of b cannot beat its worst
 *   finished hypothesis: from then on b's finished list is frozen;
 *   alive[cur_len] = 1 (int32 [max_length], zeroed by the caller) if any b is still unsat and can continue.
 * Caller initialises run_scores = {0, -1e9, ...} per b, fin_scores = -1e9, is_fin = 0, unsat = 1.
 * scratch: nbatch * beams * 132 bytes.  beams <= 8; beams * (2*max_length - prompt_len) * 4 <= 60 KB. */
typedef struct ifh_beam_desc {
    const float *logits;
    int64_t ld;
    int32_t vocab, nbatch, beams;
    const float *suppress;        /* optional f32[vocab], 0 or -inf */
    const float *begin_suppress;  /* optional f32[vocab], applied at the first generated position only */
    int32_t *toks;
    const int32_t *pos;
    int32_t prompt_len, max_length, eos_id;
    float length_penalty;
    float *run_scores;
    float *fin_scores;
    int32_t *fin_seqs;
    int32_t *fin_len;
    uint8_t *is_fin;
    int32_t *unsat;
    int32_t *beam_src;
    int32_t *alive;
    void *scratch;
} ifh_beam_desc;
int ifh_beam_step(const ifh_beam_desc *desc, ifh_stream_t stream);
/* dst[l][row] = src[l][row_src[row]] for the first min(len[0], max_len) tokens of nlayers KV caches laid out
 * [nlayers][nrows][max_len][tok_elems] bf16 (row_stride elements between rows, layer_stride between layers: the decoder's
 * layers in ONE launch per search step); src != dst (ping-pong).  tok_elems % 8 == 0. */
int ifh_kv_gather_bf16(const void *src, void *dst, const int32_t *row_src, const int32_t *len, int max_len, int nrows,
                       int64_t row_stride, int tok_elems, int nlayers, int64_t layer_stride, ifh_stream_t stream);

/* out[i] = table[ids[i]] + pos_table[pos0 + i % seq_len] (pos_table may be NULL); bf16, dim % 8 == 0.
 * dyn_pos (device scalar, optional): pos0 += dyn_pos[0], ids += dyn_pos[0]*dyn_ids_mul */
int ifh_embed_bf16(const int32_t *ids, const void *table, const void *pos_table, int pos0, int seq_len, int dim,
                   int n, void *out, const int32_t *dyn_pos, int dyn_ids_mul, ifh_stream_t stream);
/* per row of f32 logits: argmax (first on ties) and/or softmax probability of pick_token;
 * dyn_pos optional: argmax_out += dyn_pos[0]*dyn_out_mul */
int ifh_argmax_pick_f32(const float *logits, int64_t ld, int vocab, int nrows, int pick_token, int32_t *argmax_out,
                        float *pick_prob_out, const int32_t *dyn_pos, int dyn_out_mul, ifh_stream_t stream);
/* value[0] += delta on the stream (advances a device-resident step counter between graph replays); also clears
 * zero_bytes (% 16 == 0, 16-byte aligned, may be 0) at zero_buf: the LayerNorm statistics the NEXT step accumulates into */
int ifh_add_i32(int32_t *value, int delta, void *zero_buf, int64_t zero_bytes, ifh_stream_t stream);

/* ---- sampling a decode step's logits: what transformers' generate does for a checkpoint whose generation config says
 * do_sample (Qwen2.5-Instruct: repetition_penalty 1.05, temperature 0.7, top_k 20, top_p 0.8), i.e. the reference's
 * InfernLLMWorker.py:113-118 call with no sampling arguments (generation/logits_process.py order). ---- */
/* logits[r, t] = l < 0 ? l * penalty : l / penalty for every token t in history[r, 0 .. lens[r]) (each token once);
 * history int32 [nrows, hist_ld], any length; vocab <= 262144 (one presence bit per entry in LDS) */
int ifh_repetition_penalty_f32(float *logits, int64_t ld, int vocab, int nrows, const int32_t *history, int64_t hist_ld,
                               const int32_t *lens, float penalty, ifh_stream_t stream);
/* per row: logits / temperature, the top_k (<= 32; 0 = 32) best (ties to the lower token id), top-p on their softmax
 * (a candidate stays iff the mass of it and everything below it exceeds 1 - top_p; the best always stays), renormalise,
 * draw by inverse CDF over the kept candidates in descending order with uniform[r] in [0, 1).  out_tokens int32 [nrows].
 * scratch: nrows * 260 bytes.  out_cand / out_probs (optional, [nrows][32]): the sorted candidates and their final
 * probabilities (0 for dropped ones) -- test hooks. */
int ifh_sample_topk_f32(const float *logits, int64_t ld, int vocab, int nrows, float temperature, int top_k, float top_p,
                        const float *uniform, int32_t *out_tokens, void *scratch, int32_t *out_cand, float *out_probs,
                        ifh_stream_t stream);

/* ---- decoder-only LLM step (InfernLLMWorker, Cluster/InfernLLMWorker.py:60-119: Qwen2.5 through transformers' generate;
 * layer maths of transformers/models/qwen2/modeling_qwen2.py).  The projections are ifh_conv_bf16 GEMMs; these are the
 * kernels between them. ---- */
/* out = x * rsqrt(mean(x^2) + eps) * gamma   (Qwen2RMSNorm); x, out bf16 [rows, dim], gamma f32 [dim]; dim % 8 == 0 */
int ifh_rmsnorm_bf16(const void *x, const float *gamma, void *out, int rows, int dim, float eps, ifh_stream_t stream);
/* Rotary embedding + KV append on the fused projection qkv bf16 [nrows * tokens_per_row, qkv_ld] = q (nheads*head_dim) |
 * k (nkv*head_dim) | v (nkv*head_dim).  Token t of row b sits at position pos0[b] + t; tokens with t >= nvalid[b] are
 * padding and are skipped.  q is rotated in place; rotated k and v go to cache bf16 [nrows][max_pos][cache_ts] at
 * (b, position): k heads first, then v heads.  cos_sin f32 [max_pos][head_dim/2][2] (cos, sin of position * theta^(-2j/hd));
 * pairing (j, j + head_dim/2) as apply_rotary_pos_emb / rotate_half. */
int ifh_rope_append_bf16(void *qkv, int64_t qkv_ld, const float *cos_sin, int max_pos, void *cache, int64_t cache_bs,
                         int64_t cache_ts, const int32_t *pos0, const int32_t *nvalid, int nrows, int tokens_per_row,
                         int nheads, int nkv, int head_dim, ifh_stream_t stream);
/* Grouped-query attention of single query tokens against the KV cache: token i (of ntokens = rows * tokens_per_row)
 * attends over keys 0 .. key_len[i]-1 of cache row i / tokens_per_row; the nheads/nkv (<= 8) query heads of a kv head
 * share its K/V loads.  Prefill = one call over all prompt tokens with key_len[i] = position + 1 (causal); decode =
 * tokens_per_row 1.  head_dim 64 or 128.  out[i] bf16 [nheads*head_dim]; softmax(scale * q.k) v, fp32 accumulation. */
typedef struct ifh_gqa_desc {
    const void *q;
    int64_t q_ts;
    const void *cache;
    int64_t cache_bs, cache_ts;
    int32_t v_off;             /* element offset of the v heads inside a cache token (nkv * head_dim) */
    void *out;
    int64_t o_ts;
    const int32_t *key_len;    /* int32 [ntokens] */
    int32_t ntokens, tokens_per_row, nheads, nkv, head_dim, max_keys;
    float scale;
    const float *rope_cos_sin; /* optional ([max_pos][head_dim/2][2] cos, sin; tokens_per_row 1, head_dim 128): the launch also applies the
                                * rotary embedding to q and to the token's k (columns nheads*head_dim.. of its q row: the fused q|k|v
                                * projection) at position key_len[i] - 1 and appends k | v to the cache there -- what ifh_rope_append_bf16
                                * does for a decode step, in the same launch and with the same bits; `cache` is written */
} ifh_gqa_desc;
int ifh_attn_gqa_bf16(const ifh_gqa_desc *desc, ifh_stream_t stream);
/* out[r, j] = silu(gate[r, j]) * up[r, j]   (Qwen2MLP); bf16, ffn % 8 == 0.  gate_up row = [gate | up] (interleaved 0) or
 * [gate_0, up_0, gate_1, up_1, ...] (interleaved 1: the layout of a projection packed for IFH_ACT_SILU_GLU) */
int ifh_silu_mul_bf16(const void *gate_up, void *out, int64_t rows, int ffn, int interleaved, ifh_stream_t stream);
/* values[i] += delta where mask[i] != 0 (mask optional): per-row sequence lengths advanced between graph replays */
int ifh_add_i32_vec(int32_t *values, const int32_t *mask, int n, int delta, ifh_stream_t stream);

/* ---- TTS streaming glue, HelloSippyRTPipe.infer (HelloSippyRTPipe.py:191-240) ---- */
/* stop rule (:227-228) on the 2 stop logits per utterance (row stride logits_ld floats); ends_at int64[n].
 * dyn_minmax (optional, device int32[2] = {minlen, maxlen}) overrides the by-value lengths: a launch captured in a
 * hipGraph then follows the lengths of whichever batch currently occupies the state (HelloSippyRTPipe.py:117-118) */
int ifh_tts_stop_update(const float *prob_logits, int64_t *ends_at, int n, int idx, int minlen, int maxlen,
                        float threshold, int ends_inc, const int32_t *dyn_idx /* overrides idx if set */,
                        int logits_ld, const int32_t *dyn_minmax, ifh_stream_t stream);
/* the same stop rule with idx = pos[0], followed by pos[0] += 1 (one launch; for graph-replayed decode loops); also
 * clears zero_bytes at zero_buf as ifh_add_i32 does */
int ifh_tts_stop_advance(const float *prob_logits, int64_t *ends_at, int n, int minlen, int maxlen, float threshold,
                         int ends_inc, int32_t *pos, int logits_ld, void *zero_buf, int64_t zero_bytes,
                         const int32_t *dyn_minmax, ifh_stream_t stream);
/* ---- the same loop over a RAGGED batch: rows joined at different infer() calls and sit at different decoder
 * positions (continuous batching; engines/speecht5.py:TTSRaggedState).  pos int32[n] per-row position (= the
 * reference's idx), active uint8[n] (0: the slot holds no live utterance -- its position and ends_at stay frozen),
 * minmax int32[n][2] per-row {minlen, maxlen}. */
/* stop rule (:227-228) per row at idx = pos[b], then pos[b] += 1, for rows with active[b] != 0; clears zero_bytes
 * at zero_buf (the next step's LayerNorm statistics) */
int ifh_tts_stop_advance_rows(const float *prob_logits, int64_t *ends_at, int n, float threshold, int ends_inc,
                              int32_t *pos, const uint8_t *active, const int32_t *minmax, int logits_ld, void *zero_buf,
                              int64_t zero_bytes, ifh_stream_t stream);
/* frame 0 of this call's frame buffer = the last frame of the previous call (:217 `spectrum[:, -1:, :]` feeding the
 * next prenet), or zeros for a row that starts now (pos[b] == 0: `output_sequence = zeros`, :119).  prev, cur bf16
 * [n][frames][80]; copies prev[b][frames-1] -> cur[b][0] */
int ifh_tts_carry_rows_bf16(const void *prev, void *cur, const int32_t *pos, int n, int frames, ifh_stream_t stream);
/* ifh_tts_chunks_bf16 where rows with fresh[b] != 0 take zeros as their carried frames (`pre_frames = zeros[B,4,80]`,
 * :78) instead of what the slot's previous occupant left; fresh may be NULL */
int ifh_tts_chunks_rows_bf16(void *pre_frames, const void *post, const float *mean, const float *scale, void *voc_in,
                             void *amd_mel, const uint8_t *fresh, int nbatch, ifh_stream_t stream);
/* carry + 4 overlapped 12-frame chunks (:231-235): pre_frames bf16 [B][4][80] (updated), post bf16
 * [B][32][80] -> voc_in bf16 [4B][12][80] normalised by (x-mean)/scale, amd_mel bf16 [4B][12][80] =
 * channels-last form of the chunk re-viewed as [80][12] (HelloSippyRT.py:224) */
int ifh_tts_chunks_bf16(void *pre_frames, const void *post, const float *mean, const float *scale, void *voc_in,
                        void *amd_mel, int nbatch, ifh_stream_t stream);
/* HiFi-GAN tail: LeakyReLU(slope) -> Conv1d(32->1,k7,p3) -> tanh; x bf16 [nrows][t][32] -> bf16 [nrows][t] */
int ifh_hifigan_post_bf16(const void *x, const float *w7x32, float bias, void *audio, int nrows, int t, float slope,
                          ifh_stream_t stream);
/* AmendmentNetwork1 tail + chunk un-stacking: post bf16 [4B][8][256], audio bf16 [4B][3072] -> out bf16 [B][8192] */
int ifh_amend_final_bf16(const void *post, const void *audio, void *out, int nbatch, ifh_stream_t stream);
/* rows / max(||row||, 1e-12) (F.normalize) written with leading dimension ld_out */
int ifh_l2norm_rows_bf16(const void *x, int dim, int nrows, void *out, int ld_out, ifh_stream_t stream);

/* ---- spatial partition of the GPU between stages (round 3) ----
 * The serving loop (SpeechPipeline.run_steps; no counterpart in the reference, whose stages are separate processes on separate
 * GPUs or time-slice one) runs throughput kernels that fill every CU and its LDS (Whisper encoder GEMMs, HiFi-GAN) beside chains
 * of small dependent decode-step launches; while a chip-filling kernel is resident a chain's next launch finds no CU slot and the
 * chain stalls for the kernel's whole duration.  A stream created here runs its kernels on CUs [first_cu, first_cu + n_cus) only
 * (hipExtStreamCreateWithCUMask), so the throughput stages can be kept off a few CUs that then always have room for the chains. */
int ifh_stream_create_cu_range(int first_cu, int n_cus, ifh_stream_t *stream_out);
int ifh_stream_destroy(ifh_stream_t stream);
/* persistent kernels (one workgroup per CU: ifh_resblock_chain_bf16, ifh_conv_ring256_bf16) size their grids to n CUs instead of
 * the device's count; 0 restores the device's count.  Process-wide. */
int ifh_set_cu_budget(int n);

#ifdef __cplusplus
}
#endif
#endif /* INFERNOS_HIP_H */
