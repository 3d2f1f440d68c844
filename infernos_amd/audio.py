"""AudioChunk / VadAudioChunk and the cached sinc resampler.

Interface of Core/AudioChunk.py:8-47 and config/InfernGlobals.py:23-26.  Audio tensors
live on the HIP device; resampling runs ifh_resample_run (csrc/dsp.hip), which restates
torchaudio.transforms.Resample(orig, new) (sinc_interp_hann, width 6, rolloff 0.99).
"""
import ctypes
from functools import lru_cache

import torch

from . import _lib


class Resampler:
    """Callable like torchaudio.transforms.Resample: f32[..., L] -> f32[..., ceil(new*L/orig)]."""

    def __init__(self, from_sr: int, to_sr: int, device=None):
        self.device = _lib.require_device(device)
        self.from_sr, self.to_sr = int(from_sr), int(to_sr)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_resample_create(self.from_sr, self.to_sr, ctypes.byref(h)), 'ifh_resample_create')
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                _lib.lib().ifh_resample_destroy(self.handle)
        except Exception:
            pass

    def info(self):
        import numpy as np
        o, n, nt, w = (ctypes.c_int32() for _ in range(4))
        L = _lib.lib()
        _lib.check(L.ifh_resample_info(self.handle, ctypes.byref(o), ctypes.byref(n), ctypes.byref(nt), ctypes.byref(w), None))
        taps = np.zeros((n.value, nt.value), np.float32)
        _lib.check(L.ifh_resample_info(self.handle, None, None, None, None, taps.ctypes.data_as(ctypes.c_void_p)))
        return o.value, n.value, nt.value, w.value, taps

    def out_len(self, n: int) -> int:
        return int(_lib.lib().ifh_resample_out_len(self.handle, int(n)))

    def __call__(self, audio: torch.Tensor, lens: torch.Tensor = None) -> torch.Tensor:
        x = audio.to(device=self.device, dtype=torch.float32)
        shape = x.shape
        L = shape[-1] if x.dim() else 0
        if L == 0:
            return torch.zeros(shape, dtype=torch.float32, device=self.device)
        x2 = x.reshape(-1, L).contiguous()
        olen = self.out_len(L)
        out = torch.empty((x2.size(0), olen), dtype=torch.float32, device=self.device)
        if lens is not None:
            lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
            out.zero_()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_resample_run(self.handle, _lib.ptr(x2), L, _lib.ptr(lens), L, x2.size(0),
                                                   _lib.ptr(out), olen, _lib.stream_ptr(self.device)), 'ifh_resample_run')
        return out.reshape(shape[:-1] + (olen,))


@lru_cache(maxsize=8)
def get_resampler(from_sr: int, to_sr: int, device='cuda'):
    """InfernGlobals.get_resampler (config/InfernGlobals.py:23-26)."""
    return Resampler(from_sr, to_sr, device)


class AudioChunk:
    debug: bool = False
    samplerate: int
    audio: torch.Tensor
    track_id: int = 0
    active: bool = True

    def __init__(self, audio: torch.Tensor, samplerate: int):
        assert isinstance(audio, torch.Tensor)
        self.audio = audio
        self.samplerate = samplerate

    def resample(self, sample_rate: int):
        """In place, returns self (AudioChunk.py:19-24).  The result stays on the HIP device."""
        assert sample_rate != self.samplerate
        dev = self.audio.device if self.audio.is_cuda else _lib.require_device()
        rs = get_resampler(self.samplerate, sample_rate, str(dev))
        self.audio = rs(self.audio.to(torch.float)).to(self.audio.dtype)
        self.samplerate = sample_rate
        return self

    def duration(self):
        return self.audio.size(0) / self.samplerate


class VadAudioChunk(AudioChunk):
    debug: bool = False
    ipos: int

    def __init__(self, audio: torch.Tensor, samplerate: int, ipos: int):
        super().__init__(audio, samplerate)
        self.ipos = ipos

    def tpos(self):
        return self.ipos / self.samplerate

    def append(self, other: 'VadAudioChunk'):
        """Concatenate `other`, zero-filling the gap between the two (AudioChunk.py:39-47)."""
        assert self.samplerate == other.samplerate
        gap = other.ipos - (self.ipos + self.audio.size(0))
        assert gap >= 0
        parts = [self.audio]
        if gap > 0:
            parts.append(torch.zeros(gap, dtype=self.audio.dtype, device=self.audio.device))
        parts.append(other.audio.to(self.audio.device))
        self.audio = torch.cat(parts, dim=0)
