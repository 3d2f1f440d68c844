"""numpy / ctypes front-end of the DSP oracle (see dsp_oracle.c for the arithmetic).

Every function cites the reference line it restates.  Pinned against tests/golden/
by tests/test_oracle_dsp.py; the resampler is "parity unpinned" (torchaudio absent,
see dsp_oracle.c header and DESIGN.md).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.orc_ulaw2lin.restype = ctypes.c_int16
        L.orc_ulaw2lin.argtypes = [ctypes.c_uint8]
        L.orc_lin2ulaw.restype = ctypes.c_uint8
        L.orc_lin2ulaw.argtypes = [ctypes.c_int16]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---- G.711 (Core/Codecs/G711.py:7-47) ------------------------------------------------
def ulaw_to_pcm_table():
    L = lib()
    return np.array([L.orc_ulaw2lin(i) for i in range(256)], dtype=np.int16)


def pcm_to_ulaw_table():
    L = lib()
    return np.array([L.orc_lin2ulaw(i) for i in range(-32768, 32768)], dtype=np.uint8)


def g711_decode(data):
    """bytes/uint8[n] -> float32[n]   (G711Codec.decode, resample=False)"""
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
    out = np.empty(a.size, np.float32)
    lib().orc_g711_decode(_p(a), _p(out), ctypes.c_int64(a.size))
    return out.reshape(a.shape)


def g711_encode(x):
    """float32[n] -> uint8[n]   (G711Codec.encode)"""
    a = np.ascontiguousarray(x, np.float32)
    out = np.empty(a.size, np.uint8)
    lib().orc_g711_encode(_p(a), _p(out), ctypes.c_int64(a.size))
    return out.reshape(a.shape)


# ---- sinc resampler (Core/AudioChunk.py:19-24 -> torchaudio Resample) ------------------
def sinc_kernel(orig, new, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann), float64
    arithmetic then cast to float32.  Returns (kernel[new_r, ntaps] f32, width, orig_r, new_r)."""
    g = math.gcd(int(orig), int(new))
    o, n = int(orig) // g, int(new) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    # torchaudio: arange(0,-new,-1)/new is float32 there; exact for the ratios used here
    ph = (np.arange(0, -n, -1, dtype=np.float64) / n).astype(np.float32).astype(np.float64)[:, None]
    t = (ph + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / o
    with np.errstate(invalid='ignore', divide='ignore'):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    k = k * window * scale
    return k.astype(np.float32), width, o, n


def resample(x, orig, new):
    """float32[L] -> float32[ceil(new*L/orig)]  (torchaudio _apply_sinc_resample_kernel)"""
    x = np.ascontiguousarray(x, np.float32)
    k, width, o, n = sinc_kernel(orig, new)
    L = x.shape[-1]
    out_len = int(math.ceil(n * L / o))
    if L == 0:
        return np.zeros(x.shape[:-1] + (0,), np.float32)
    flat = x.reshape(-1, L)
    out = np.empty((flat.shape[0], out_len), np.float32)
    for r in range(flat.shape[0]):
        lib().orc_resample(_p(flat[r]), ctypes.c_int64(L), _p(k), o, n, k.shape[1], width,
                           _p(out[r]), ctypes.c_int64(out_len))
    return out.reshape(x.shape[:-1] + (out_len,))


# ---- Whisper log-mel (Cluster/InfernSTTWorker.py:114 -> WhisperFeatureExtractor) -------
def hz_to_mel_slaney(f):
    f = np.asarray(f, np.float64)
    mel = 3.0 * f / 200.0
    lr = f >= 1000.0
    with np.errstate(divide='ignore', invalid='ignore'):
        mel = np.where(lr, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), mel)
    return mel


def mel_to_hz_slaney(m):
    m = np.asarray(m, np.float64)
    f = 200.0 * m / 3.0
    lr = m >= 15.0
    return np.where(lr, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), f)


def mel_filter_bank(n_bins=201, n_mel=80, fmin=0.0, fmax=8000.0, sr=16000):
    """transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney') as
    used by WhisperFeatureExtractor.__init__ (feature_extraction_whisper.py:92-100).
    Returns float64 [n_bins, n_mel]."""
    mel_pts = np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mel + 2)
    freqs = mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    fdiff = np.diff(freqs)
    slopes = freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (freqs[2:n_mel + 2] - freqs[:n_mel])
    return fb * enorm[None, :]


def logmel(audio, n_mel=80, nsamp=480000):
    """float32[L] (or [B,L]) @16 kHz -> float32[n_mel, 3000]: zero-pad/truncate to 30 s,
    reflect-centred STFT(400, hop 160, periodic Hann), |.|^2, mel, log10(clamp 1e-10),
    max(x, max-8), (x+4)/4.  (feature_extraction_whisper.py:135-168, torch path)"""
    a = np.asarray(audio, np.float32)
    if a.ndim == 2:
        return np.stack([logmel(r, n_mel, nsamp) for r in a])
    x = np.zeros(nsamp, np.float32)
    x[:min(nsamp, a.size)] = a[:nsamp]
    xp = np.pad(x, 200, mode='reflect')
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(400) / 400)).astype(np.float32)
    nfr = nsamp // 160
    idx = np.arange(nfr)[:, None] * 160 + np.arange(400)[None, :]
    fr = (xp[idx] * win[None, :]).astype(np.float32)
    spec = np.fft.rfft(fr.astype(np.float64), axis=1)
    pw = (spec.real ** 2 + spec.imag ** 2)
    mel = mel_filter_bank(201, n_mel)
    ms = pw @ mel
    lg = np.log10(np.maximum(ms, 1e-10))
    lg = np.maximum(lg, lg.max() - 8.0)
    return ((lg + 4.0) / 4.0).T.astype(np.float32)


def logmel_direct(audio, n_mel=80, nsamp=480000):
    """dsp_oracle.c:orc_logmel_direct -- O(N^2) double DFT cross-check for short nsamp."""
    a = np.asarray(audio, np.float32)
    x = np.zeros(nsamp, np.float32)
    x[:min(nsamp, a.size)] = a[:nsamp]
    nfr = nsamp // 160
    mel = np.ascontiguousarray(mel_filter_bank(201, n_mel), np.float64)
    out = np.empty((n_mel, nfr), np.float32)
    lib().orc_logmel_direct(_p(x), ctypes.c_int64(nsamp), _p(mel), n_mel, _p(out), ctypes.c_int64(nfr))
    return out


# ---- output mix + encode (Core/OutputMuxer.py:75-85 + G711Codec.encode) ---------------------------
def mux_encode(tracks, present, ndiv):
    """tracks f32 [n][K][L], present bool [n][K], ndiv int [n] -> (u8 [n][L], has_out bool [n])"""
    tracks = np.asarray(tracks, np.float32)
    n, K, L = tracks.shape
    out = np.zeros((n, L), np.uint8)
    has = np.zeros(n, bool)
    for c in range(n):
        ks = [k for k in range(K) if present[c][k]]
        if not ks:
            continue
        has[c] = True
        if len(ks) == 1:
            mix = tracks[c, ks[0]]
        else:
            acc = tracks[c, ks[0]].copy()
            for k in ks[1:]:
                acc = (acc + tracks[c, k]).astype(np.float32)
            mix = (acc / np.float32(ndiv[c])).astype(np.float32)
        out[c] = g711_encode(mix)
    return out, has


# ---- G.722 (g722_oracle.c; parity unpinned: see its header) -----------------------------------------------------
def g722_new_state():
    st = np.zeros(128, np.int32)
    lib().orc_g722_init(_p(st))
    return st


def g722_encode(state, pcm16, eight_k=True):
    """pcm16 int16 [n] -> uint8 [n] (eight_k) or [n/2]; `state` (g722_new_state) is advanced"""
    pcm16 = np.ascontiguousarray(pcm16, np.int16)
    out = np.zeros(pcm16.size if eight_k else (pcm16.size + 1) // 2, np.uint8)
    lib().orc_g722_encode.restype = ctypes.c_int64
    n = lib().orc_g722_encode(_p(state), _p(pcm16), ctypes.c_int64(pcm16.size), int(eight_k), _p(out))
    return out[:n]


def g722_decode(state, code, eight_k=True):
    """uint8 [n] -> int16 [n] (eight_k) or [2n]"""
    code = np.ascontiguousarray(code, np.uint8)
    out = np.zeros(code.size if eight_k else 2 * code.size, np.int16)
    lib().orc_g722_decode.restype = ctypes.c_int64
    n = lib().orc_g722_decode(_p(state), _p(code), ctypes.c_int64(code.size), int(eight_k), _p(out))
    return out[:n]
