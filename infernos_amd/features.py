"""Whisper log-mel feature extraction on the HIP device.

Replaces the WhisperProcessor call at Cluster/InfernSTTWorker.py:114 (transformers
WhisperFeatureExtractor: 30 s padding, STFT 400/160 Hann, 80/128 slaney mels, log10,
max-8 clamp, (x+4)/4).  Kernel: csrc/logmel.hip via ifh_logmel_run.
"""
import ctypes

import torch

from . import _lib

N_SAMPLES = 480000
N_FRAMES = 3000


class WhisperLogMel:
    def __init__(self, n_mel: int = 80, device=None):
        self.device = _lib.require_device(device)
        self.n_mel = n_mel
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_logmel_create(n_mel, ctypes.byref(h)), 'ifh_logmel_create')
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.lib().ifh_logmel_destroy(self.handle)
        except Exception:
            pass

    def filters(self):
        import numpy as np
        f = np.zeros((201, self.n_mel), np.float32)
        _lib.check(_lib.lib().ifh_logmel_filters_host(self.handle, f.ctypes.data_as(ctypes.c_void_p)))
        return f

    def __call__(self, audio: torch.Tensor, lens: torch.Tensor = None, out_dtype=torch.float32, out=None):
        """audio f32 [B, L] (L <= 480000 or more; truncated), lens int32[B] valid samples
        (default L) -> [B, n_mel, 3000] float32 or bfloat16."""
        dev = self.device
        x = audio.to(dev, torch.float32)
        if x.dim() == 1:
            x = x[None, :]
        x = x.contiguous()
        B, L = x.shape
        if lens is None:
            lens = torch.full((B,), min(L, N_SAMPLES), dtype=torch.int32, device=dev)
        else:
            lens = lens.to(dev, torch.int32).clamp(max=min(L, N_SAMPLES)).contiguous()
        bf16 = out_dtype == torch.bfloat16
        if out is None:
            out = torch.empty((B, self.n_mel, N_FRAMES), dtype=out_dtype, device=dev)
        L_ = _lib.lib()
        ws = torch.empty(int(L_.ifh_logmel_workspace_floats(self.handle, B, int(bf16))), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L_.ifh_logmel_run(self.handle, _lib.ptr(x), L, _lib.ptr(lens), B, _lib.ptr(out), int(bf16),
                                         _lib.ptr(ws), _lib.stream_ptr(dev)), 'ifh_logmel_run')
        return out

    def raw(self, audio: torch.Tensor, lens: torch.Tensor = None, out=None):
        """The transform without the normalisation pass: -> (raw f32 [B, n_mel, 3000] = log10(max(mel, 1e-10)),
        win_max int32 [B]).  Hand both to `to_conv_input` (or Whisper.encode(raw=...)): the clamp to max - 8 and the
        (x + 4) / 4 scale happen while the consumer re-lays the features out, so the normalised plane never exists."""
        dev = self.device
        x = audio.to(dev, torch.float32)
        if x.dim() == 1:
            x = x[None, :]
        x = x.contiguous()
        B, L = x.shape
        if lens is None:
            lens = torch.full((B,), min(L, N_SAMPLES), dtype=torch.int32, device=dev)
        else:
            lens = lens.to(dev, torch.int32).clamp(max=min(L, N_SAMPLES)).contiguous()
        if out is None:
            out = torch.empty((B, self.n_mel, N_FRAMES), dtype=torch.float32, device=dev)
        wmax = torch.empty(B, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_logmel_run_raw(self.handle, _lib.ptr(x), L, _lib.ptr(lens), B, _lib.ptr(out), _lib.ptr(wmax),
                                                     _lib.stream_ptr(dev)), 'ifh_logmel_run_raw')
        return out, wmax

    def to_conv_input(self, raw: torch.Tensor, wmax: torch.Tensor, out: torch.Tensor = None):
        """(raw, win_max) -> normalised bf16 [B, 3000, n_mel] (channels-last input of Whisper's conv1)"""
        B = raw.size(0)
        if out is None:
            out = torch.empty((B, N_FRAMES, self.n_mel), dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_logmel_finish_transpose_bf16(self.handle, _lib.ptr(raw), _lib.ptr(wmax), B, _lib.ptr(out),
                                                                   _lib.stream_ptr(self.device)), 'ifh_logmel_finish_transpose_bf16')
        return out
