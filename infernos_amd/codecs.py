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


def g711_tables():
    """(int16[256] ulaw->pcm, uint8[65536] pcm->ulaw) as the kernels compute them."""
    import numpy as np
    a = np.zeros(256, np.int16)
    b = np.zeros(65536, np.uint8)
    _lib.check(_lib.lib().ifh_g711_tables_host(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p)))
    return a, b
