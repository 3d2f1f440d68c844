"""G.711 mu-law codec on the HIP device.

Interface of Core/Codecs/GenCodec.py:1-13 and Core/Codecs/G711.py:21-70 (class attrs
srate/crate/ptype/ename, rtpmap(), encode(Tensor)->bytes, decode(bytes, resample,
sample_rate)->AudioChunk, silence, e2d_frames, d2e_frames, to(), device()).
Arithmetic: ifh_g711_{decode_u8_f32,encode_f32_u8} (csrc/dsp.hip).
"""
import ctypes

import torch

from . import _lib
from .audio import AudioChunk

# The reference keeps its two lookup tables as module globals that .to() moves for every
# codec instance at once (G711.py:49-59); the device selection is shared the same way here.
_shared_device = None


class GenCodec:
    srate: int = 8000   # sample rate
    crate: int = 8000   # RTP clock rate
    ptype: int = None   # RTP payload type
    ename: str = None   # encoding name

    def __init__(self):
        assert self.ptype is not None and self.ename is not None

    @classmethod
    def rtpmap(cls):
        assert cls.ptype is not None and cls.ename is not None
        return 'rtpmap:%d %s/%d' % (cls.ptype, cls.ename, cls.crate)


class G711Codec(GenCodec):
    ptype = 0
    ename = 'PCMU'

    # -- device handling ---------------------------------------------------------------
    def device(self):
        global _shared_device
        if _shared_device is None:
            _shared_device = _lib.require_device()
        return _shared_device

    def to(self, device):
        global _shared_device
        _shared_device = _lib.require_device(device)
        return self

    # -- codec -------------------------------------------------------------------------
    def encode(self, audio_tensor: torch.Tensor) -> bytes:
        dev = self.device()
        x = audio_tensor.detach().to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
        out = torch.empty(x.numel(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_g711_encode_f32_u8(_lib.ptr(x), _lib.ptr(out), x.numel(), _lib.stream_ptr(dev)),
                       'ifh_g711_encode_f32_u8')
        return out.cpu().numpy().tobytes()

    def decode_tensor(self, ulaw: torch.Tensor) -> torch.Tensor:
        """uint8 device tensor (any shape) -> float32 tensor of the same shape."""
        dev = self.device()
        u = ulaw.to(device=dev, dtype=torch.uint8).contiguous()
        out = torch.empty(u.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_g711_decode_u8_f32(_lib.ptr(u), _lib.ptr(out), u.numel(), _lib.stream_ptr(dev)),
                       'ifh_g711_decode_u8_f32')
        return out

    def decode(self, ulaw_bytes: bytes, resample: bool = True, sample_rate: int = GenCodec.srate):
        dev = self.device()
        host = torch.frombuffer(bytearray(ulaw_bytes), dtype=torch.uint8) if len(ulaw_bytes) else \
            torch.empty(0, dtype=torch.uint8)
        chunk = AudioChunk(self.decode_tensor(host.to(dev)), self.srate)
        if resample and sample_rate != self.srate:
            chunk.resample(sample_rate)
        return chunk

    # -- frame arithmetic ----------------------------------------------------------------
    def e2d_frames(self, enframes: int, out_srate: int = GenCodec.srate):
        assert out_srate % self.srate == 0
        return enframes * out_srate // self.srate

    def d2e_frames(self, dnframes: int, in_srate: int = GenCodec.srate):
        assert in_srate % self.srate == 0
        return dnframes * self.srate // in_srate

    def silence(self, nframes: int):
        return b'\xff' * nframes


class G722Batch:
    """G.722 encoder + decoder state of `ncalls` calls on the device (ifh_g722_*): frames of all calls per launch, state
    carried between frames.  eight_k=True is the mode the reference constructs (G722(8000, 64000)): 8 kHz samples, one code
    byte per sample; eight_k=False is the codec's native 16 kHz form."""

    def __init__(self, ncalls: int, device=None, eight_k: bool = True):
        self.device = dev = _lib.require_device(device)
        self.n, self.eight_k = ncalls, bool(eight_k)
        self.enc_state = torch.zeros((ncalls, 128), dtype=torch.int32, device=dev)
        self.dec_state = torch.zeros((ncalls, 128), dtype=torch.int32, device=dev)
        self.reset()

    def reset(self):
        with torch.cuda.device(self.device):
            for st in (self.enc_state, self.dec_state):
                _lib.check(_lib.lib().ifh_g722_init(_lib.ptr(st), self.n, _lib.stream_ptr(self.device)), 'ifh_g722_init')

    def encode(self, pcm: torch.Tensor) -> torch.Tensor:
        """pcm [ncalls, S] f32 in [-1, 1] or int16 (device) -> uint8 [ncalls, S] (eight_k) or [ncalls, S / 2]"""
        assert pcm.dim() == 2 and pcm.size(0) == self.n and pcm.dtype in (torch.float32, torch.int16)
        pcm = pcm.to(self.device).contiguous()
        S = pcm.size(1)
        out = torch.empty((self.n, S if self.eight_k else S // 2), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_g722_encode(_lib.ptr(self.enc_state), _lib.ptr(pcm), int(pcm.dtype == torch.float32), S, S,
                                                  int(self.eight_k), _lib.ptr(out), out.size(1), self.n,
                                                  _lib.stream_ptr(self.device)), 'ifh_g722_encode')
        return out

    def decode(self, code: torch.Tensor, f32: bool = True) -> torch.Tensor:
        """uint8 [ncalls, nbytes] -> f32 (value / 32767) or int16 [ncalls, nbytes] (eight_k) or [ncalls, 2 nbytes]"""
        assert code.dim() == 2 and code.size(0) == self.n and code.dtype == torch.uint8
        code = code.to(self.device).contiguous()
        nb = code.size(1)
        out = torch.empty((self.n, nb if self.eight_k else 2 * nb), dtype=torch.float32 if f32 else torch.int16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_g722_decode(_lib.ptr(self.dec_state), _lib.ptr(code), nb, nb, int(self.eight_k), _lib.ptr(out),
                                                  int(f32), out.size(1), self.n, _lib.stream_ptr(self.device)), 'ifh_g722_decode')
        return out


class G722Codec(GenCodec):
    """Core/Codecs/G722.py:8-56 -- one stream's codec object (encoder and decoder state inside, as the wrapped `G722` module
    keeps them): encode(float tensor in [-1, 1] @8 kHz) -> bytes, decode(bytes) -> AudioChunk @8 kHz, optionally resampled."""
    srate: int = 8000
    default_br: int = 64000
    ptype: int = 9
    ename: str = 'G722'

    def __init__(self):
        super().__init__()
        self._device = None
        self.codec = None

    def _batch(self):
        if self.codec is None:
            self.codec = G722Batch(1, self._device, eight_k=(self.srate == 8000))
        return self.codec

    def device(self):
        return self._batch().device

    def to(self, device):
        dev = _lib.require_device(device)
        if self.codec is not None and self.codec.device != dev:
            raise RuntimeError('G722Codec.to(): the stream state already lives on %s' % self.codec.device)
        self._device = dev
        return self

    def encode(self, audio_tensor: torch.Tensor) -> bytes:
        b = self._batch()
        x = audio_tensor.detach()
        x = x.to(b.device, torch.int16 if x.dtype == torch.int16 else torch.float32).reshape(1, -1)
        if x.size(1) == 0:
            return b''
        return b.encode(x)[0].cpu().numpy().tobytes()

    def decode(self, audio_enc: bytes, resample: bool = True, sample_rate: int = srate):
        b = self._batch()
        if len(audio_enc) == 0:
            chunk = AudioChunk(torch.empty(0, dtype=torch.float32, device=b.device), self.srate)
        else:
            code = torch.frombuffer(bytearray(audio_enc), dtype=torch.uint8).reshape(1, -1)
            chunk = AudioChunk(b.decode(code.to(b.device))[0], self.srate)
        if resample and sample_rate != self.srate:
            chunk.resample(sample_rate)
        return chunk

    def silence(self, nframes: int):
        return self.encode(torch.zeros(self.e2d_frames(nframes), dtype=torch.int16))

    def e2d_frames(self, enframes: int, out_srate: int = srate):
        return enframes * (1 if self.srate == 8000 else 2) * out_srate // self.srate

    def d2e_frames(self, dnframes: int, in_srate: int = srate):
        return dnframes * self.srate // ((1 if self.srate == 8000 else 2) * in_srate)


def g711_tables():
    """(int16[256] ulaw->pcm, uint8[65536] pcm->ulaw) as the kernels compute them."""
    import numpy as np
    a = np.zeros(256, np.int16)
    b = np.zeros(65536, np.uint8)
    _lib.check(_lib.lib().ifh_g711_tables_host(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p)))
    return a, b
