"""Batched per-tick front end: every call's 20 ms G.711 frame handled by one kernel launch.

This is the MI355X replacement for the per-packet Python of RTP/InfernRTPIngest.py:63-100 +
Core/VAD/SileroVAD.py:27-35 (one GIL-bound thread for all calls): `CallTable.tick` takes
the [N,160] mu-law frame matrix and updates device-resident per-call state (byte FIFO,
768-sample VAD window, streaming-resampler carry) via ifh_ingest_tick.
"""
import torch

from . import _lib
from .audio import get_resampler

FIFO_CAP = 1024
WINDOW = 768


class CallTable:
    def __init__(self, capacity: int, device=None):
        self.device = dev = _lib.require_device(device)
        self.capacity = capacity
        self.fifo = torch.zeros((capacity, FIFO_CAP), dtype=torch.uint8, device=dev)
        self.fifo_len = torch.zeros(capacity, dtype=torch.int32, device=dev)
        self.win = torch.zeros((capacity, WINDOW), dtype=torch.float32, device=dev)
        self.win_ready = torch.zeros(capacity, dtype=torch.int32, device=dev)
        self.hist = torch.zeros((capacity, 16), dtype=torch.float32, device=dev)
        self._rs = get_resampler(8000, 16000, str(dev))

    def tick(self, frames: torch.Tensor, slots: torch.Tensor, pcm8k=None, pcm16k=None, want_ready=True):
        """frames u8 [n,160] (device), slots int32 [n] -> (pcm8k f32 [n,160], pcm16k f32 [n,320],
        win_ready int32 [n] view by slot)."""
        dev = self.device
        n = frames.size(0)
        assert frames.dtype == torch.uint8 and frames.shape == (n, 160) and frames.is_cuda
        frames = frames.contiguous()
        slots = slots.to(dev, torch.int32).contiguous()
        if pcm8k is None:
            pcm8k = torch.empty((n, 160), dtype=torch.float32, device=dev)
        if pcm16k is None:
            pcm16k = torch.empty((n, 320), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_ingest_tick(
                _lib.ptr(frames), _lib.ptr(slots), n, _lib.ptr(self.fifo), _lib.ptr(self.fifo_len), _lib.ptr(self.win),
                _lib.ptr(self.win_ready), _lib.ptr(self.hist), _lib.ptr(pcm8k), _lib.ptr(pcm16k), self._rs.handle,
                _lib.stream_ptr(dev)), 'ifh_ingest_tick')
        return pcm8k, pcm16k, (self.win_ready[slots.long()] if want_ready else None)


def mux_encode(tracks: torch.Tensor, present: torch.Tensor, ndiv: torch.Tensor):
    """Batched output mix + mu-law encode (ifh_mux_encode_f32_u8): tracks f32 [n,K,L], present u8/bool [n,K],
    ndiv int32 [n] -> (u8 [n,L], has_out u8 [n]) on the device."""
    dev = _lib.require_device(tracks.device if tracks.is_cuda else None)
    tracks = tracks.to(dev, torch.float32).contiguous()
    n, K, L = tracks.shape
    present = present.to(dev, torch.uint8).contiguous()
    ndiv = ndiv.to(dev, torch.int32).contiguous()
    out = torch.empty((n, L), dtype=torch.uint8, device=dev)
    has = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ifh_mux_encode_f32_u8(_lib.ptr(tracks), _lib.ptr(present), _lib.ptr(ndiv), n, K, L, _lib.ptr(out),
                                                    _lib.ptr(has), _lib.stream_ptr(dev)), 'ifh_mux_encode_f32_u8')
    return out, has


class TickEgress:
    """Hand-back of a tick's encoded frames to the host: the last step of the per-tick path (the reference hands each call's bytes
    to its RTP sender thread, RTP/RTPOutputWorker.py:84-149; here one [n, L] matrix per tick goes to a pinned host buffer).

    `push(enc)` queues the device-to-host copy on the current stream and records ONE timing event -- a marker packet -- behind it;
    `wait()` blocks until the stream has passed it and returns the host buffer.  The marker is not decoration: with the serving
    engines running beside the tick, a copy that is followed directly by the blocking wait left 1-2 ticks per 100 waiting 35-65 ms
    inside the tick's own hardware queue (round 4: eleven alternating runs, p99 40-57 ms without the marker, 3.6-7.2 ms with it;
    the queue reaches the tick's first packet at once, the stall sits between its last kernel and the completion signal of the
    copy).  What the extra packet changes in the queue processor is not understood; it is kept where every per-tick caller gets it --
    bench.py's tick probe calls this class (`marker = False` takes the event out, for measurements)."""

    def __init__(self, n: int, L: int = 160, device=None):
        self.device = _lib.require_device(device)
        self.host = torch.empty((n, L), dtype=torch.uint8).pin_memory()
        self.marker = True             # the timing event behind the copy (False: the tick tail of round 4, profiles/NOTES.md)
        self._ev = torch.cuda.Event(enable_timing=True) if self.marker else None
        self._stream = None

    def push(self, enc: torch.Tensor):
        assert enc.dtype == torch.uint8 and enc.shape == self.host.shape and enc.is_cuda
        self._stream = torch.cuda.current_stream(self.device)
        self.host.copy_(enc, non_blocking=True)
        if self.marker:
            self._ev.record(self._stream)
        return self

    def wait(self) -> torch.Tensor:
        if self._stream is not None:
            self._stream.synchronize()
        return self.host
