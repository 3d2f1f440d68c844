"""RTP ingress in front of the tick kernel.

The three names the reference imports from the third-party `rtpsynth.RtpJBuf` module (`RtpJBuf`, `RTPFrameType`,
`RTPParseError`, InfernRTPIngest.py:6) over the C-ABI entry points `ifh_rtp_parse` / `ifh_rtpjb_*` (csrc/rtp.hip).
rtpsynth is not in the reference tree nor in the image, so its behaviour is restated from the call site (PARITY
UNPINNED).  The reference's own RTP/InfernRTPIngest.py (per-packet thread, `RTPInStream`) is NOT
mirrored here: it runs unchanged on top of these names once `compat.install()` has aliased `rtpsynth.RtpJBuf`,
`Core.VAD.SileroVAD` and `Core.Codecs.G711`.

(RTP egress -- header synthesis, `rtpsynth.RtpSynth` -- is outside SURVEY.md section 8 and not part of this library.)
`RTPIngestTable` is the batched form the MI355X path uses: datagrams of all calls are pushed into one table and
`pop_tick()` hands `CallTable.tick` / `ifh_ingest_block` the `[n,160]` frame matrix and slot list of the calls
that have a whole 20 ms frame, instead of one Python `VADChannel.ingest` call per packet per call.
"""
import ctypes
from typing import Optional

import numpy as np

from . import _lib


class RTPParseError(Exception):
    pass


class RTPFrameType:
    RTP = 0
    ERS = 1


class _RtpInfo:
    __slots__ = ('lseq', 'seq', 'ts', 'ssrc', 'ptype', 'mbt', 'padding', 'extension', 'cc', 'data_offset', 'data_size',
                 'nsamples')


class _Frame:
    __slots__ = ('rtp',)


class _Content:
    """`.type` plus either `.frame.rtp.*` (RTP) or `.lseq_start/.lseq_end/.ts_diff` (ERS), as read at
    InfernRTPIngest.py:82-92."""
    __slots__ = ('type', 'frame', 'lseq_start', 'lseq_end', 'ts_diff')


class _Ready:
    __slots__ = ('content', 'rtp_data')

    def __repr__(self):
        c = self.content
        if c.type == RTPFrameType.ERS:
            return f'ERS(lseq={c.lseq_start}..{c.lseq_end}, ts_diff={c.ts_diff})'
        return f'RTP(lseq={c.frame.rtp.lseq}, ts={c.frame.rtp.ts}, len={len(self.rtp_data)})'


def _check_parse(rc, what):
    if rc == _lib.IFH_ERTPPARSE:
        raise RTPParseError((_lib.lib().ifh_last_error() or b'').decode())
    _lib.check(rc, what)


def rtp_parse(data: bytes) -> dict:
    """RFC 3550 header fields of one datagram (ifh_rtp_parse); RTPParseError when malformed."""
    hdr = _lib.RtpHdr()
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b'\0')
    _check_parse(_lib.lib().ifh_rtp_parse(buf, len(data), ctypes.byref(hdr)), 'ifh_rtp_parse')
    return {n: getattr(hdr, n) for n, _ in _lib.RtpHdr._fields_}


def _wrap(rec, payload) -> _Ready:
    r, c = _Ready(), _Content()
    c.type = rec.type
    if rec.type == RTPFrameType.ERS:
        c.frame = None
        c.lseq_start, c.lseq_end, c.ts_diff = rec.lseq_start, rec.lseq_end, rec.ts_diff
        r.rtp_data = None
    else:
        info = _RtpInfo()
        info.lseq, info.seq, info.ts, info.ssrc = rec.lseq_start, rec.hdr.seq, rec.hdr.ts, rec.hdr.ssrc
        info.ptype, info.mbt, info.padding, info.extension, info.cc = rec.hdr.pt, rec.hdr.marker, rec.hdr.padding, \
            rec.hdr.extension, rec.hdr.cc
        info.data_offset, info.data_size, info.nsamples = rec.hdr.payload_off, rec.hdr.payload_len, rec.hdr.payload_len
        c.frame = _Frame()
        c.frame.rtp = info
        c.lseq_start = c.lseq_end = c.ts_diff = None
        r.rtp_data = bytes(payload[rec.payload_off:rec.payload_off + rec.payload_len])
    r.content = c
    return r


class _Table:
    """Owner of one ifh_rtpjb_t handle."""

    def __init__(self, n_streams, depth, frame_bytes=160, ts_per_byte=1, fill_byte=0xff, fifo_cap=8192):
        self.n_streams, self.depth, self.frame_bytes = n_streams, depth, frame_bytes
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().ifh_rtpjb_create(n_streams, depth, frame_bytes, ts_per_byte, fill_byte, fifo_cap,
                                               ctypes.byref(h)), 'ifh_rtpjb_create')
        self._h = h
        self._recs = (_lib.RtpRec * (depth + 3))()
        self._payload = (ctypes.c_uint8 * ((depth + 2) * _lib.IFH_RTP_MAX_PAYLOAD))()
        self._nrec = ctypes.c_int32()

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            _lib.lib().ifh_rtpjb_destroy(h)

    def push(self, stream: int, data: bytes):
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b'\0')
        rc = _lib.lib().ifh_rtpjb_push(self._h, stream, buf, len(data), self._recs, len(self._recs), self._payload,
                                       len(self._payload), ctypes.byref(self._nrec))
        _check_parse(rc, 'ifh_rtpjb_push')
        return [_wrap(self._recs[i], self._payload) for i in range(self._nrec.value)]

    def reset(self, stream: int, drop_fifo=False):
        _lib.check(_lib.lib().ifh_rtpjb_reset_stream(self._h, stream, int(drop_fifo)), 'ifh_rtpjb_reset_stream')

    def stats(self, stream: int) -> dict:
        st = (ctypes.c_int64 * len(_lib.RTP_STATS))()
        _lib.check(_lib.lib().ifh_rtpjb_stats(self._h, stream, st), 'ifh_rtpjb_stats')
        return dict(zip(_lib.RTP_STATS, st))


class RtpJBuf:
    """`rtpsynth.RtpJBuf.RtpJBuf(capacity).udp_in(data) -> [frames]` as used at InfernRTPIngest.py:43,68,77."""

    def __init__(self, capacity: int):
        self._t = _Table(1, capacity)

    def udp_in(self, data: bytes):
        return self._t.push(0, data)

    def stats(self):
        return self._t.stats(0)


class RTPIngestTable(_Table):
    """All calls of one ingest thread: push datagrams as they arrive, pop one frame matrix per 20 ms tick."""

    RING = 4

    def __init__(self, n_streams, depth=8, frame_bytes=160, fifo_cap=8192, pin=None):
        super().__init__(n_streams, depth, frame_bytes, 1, 0xff, fifo_cap)
        import torch
        pin = torch.cuda.is_available() if pin is None else pin
        # a small ring of pinned (frames, slots) pairs: the views pop_tick() hands out stay untouched for RING - 1
        # further ticks, so an asynchronous H2D copy of tick t is not overwritten while the host already pops t+1
        # (catching up after a stall).  push_batch / pop_tick must be called from ONE thread (the table has no lock).
        self._ring = [(torch.empty((n_streams, frame_bytes), dtype=torch.uint8, pin_memory=pin),
                       torch.empty(n_streams, dtype=torch.int32, pin_memory=pin)) for _ in range(self.RING)]
        self._tick = 0
        self.frames, self.slots = self._ring[0]
        self._n = ctypes.c_int32()

    def push_batch(self, datagrams, streams):
        """datagrams: list of bytes; streams: their call indices.  Returns the per-datagram status codes
        (0 = taken, IFH_ERTPPARSE = malformed and ignored, like the `except RTPParseError: return` at :78-80)."""
        n = len(datagrams)
        off = np.zeros(n + 1, np.int32)
        np.cumsum([len(d) for d in datagrams], out=off[1:])
        buf = np.frombuffer(b''.join(datagrams) or b'\0', np.uint8)
        sid = np.ascontiguousarray(streams, np.int32)
        status = np.zeros(n, np.int32)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = _lib.lib().ifh_rtpjb_push_batch(self._h, vp(buf), vp(off), vp(sid), n, vp(status))
        if rc < 0:
            _lib.check(rc, 'ifh_rtpjb_push_batch')
        return status

    def pop_tick(self):
        """-> (frames u8 [n,frame_bytes], slots int32 [n]) host views (pinned when a GPU is present) of the calls
        that hold a whole frame; feed `.to(device, non_blocking=True)` of both to CallTable.tick.  The views belong to
        a ring of RING buffers: they are valid until RING - 1 further pop_tick() calls have been made."""
        self.frames, self.slots = self._ring[self._tick % self.RING]
        self._tick += 1
        _lib.check(_lib.lib().ifh_rtpjb_pop_tick(self._h, ctypes.c_void_p(self.frames.data_ptr()),
                                                 ctypes.c_void_p(self.slots.data_ptr()), self.n_streams,
                                                 ctypes.byref(self._n)), 'ifh_rtpjb_pop_tick')
        n = self._n.value
        return self.frames[:n], self.slots[:n]
