"""RTP ingress in front of the tick kernel.

Host mirror of RTP/InfernRTPIngest.py:15-160 (`WIPkt`, `WIStreamUpdate`, `WIStreamConnect`, `RTPInStream`,
`InfernRTPIngest`), RTP/AudioInput.py:3-8 and of the three names the reference imports from the third-party
`rtpsynth.RtpJBuf` module (`RtpJBuf`, `RTPFrameType`, `RTPParseError`, InfernRTPIngest.py:6), all over the
C-ABI entry points `ifh_rtp_parse` / `ifh_rtpjb_*` (csrc/rtp.hip).  rtpsynth is not in the reference tree nor in
the image, so its behaviour is restated from the call site (PARITY UNPINNED).

`RtpSynth` / `RTPEgressTable` are the egress counterpart (`rtpsynth.RtpSynth`, RTP/RTPOutputWorker.py:88,104,136) over
`ifh_rtpsynth_*`.  `RTPIngestTable` is the batched form the MI355X path uses: datagrams of all calls are pushed into one table and
`pop_tick()` hands `CallTable.tick` / `ifh_ingest_block` the `[n,160]` frame matrix and slot list of the calls
that have a whole 20 ms frame, instead of one Python `VADChannel.ingest` call per packet per call.
"""
import ctypes
from queue import Queue
from threading import Lock
from typing import Optional, Union
from uuid import UUID

import numpy as np

from . import _lib
from .workers import InfernWrkThread, RTPWrkTRun


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

    def __init__(self, n_streams, depth=8, frame_bytes=160, fifo_cap=8192, pin=None):
        super().__init__(n_streams, depth, frame_bytes, 1, 0xff, fifo_cap)
        import torch
        pin = torch.cuda.is_available() if pin is None else pin
        self.frames = torch.empty((n_streams, frame_bytes), dtype=torch.uint8, pin_memory=pin)
        self.slots = torch.empty(n_streams, dtype=torch.int32, pin_memory=pin)
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
        that hold a whole frame; feed `.to(device, non_blocking=True)` of both to CallTable.tick."""
        _lib.check(_lib.lib().ifh_rtpjb_pop_tick(self._h, ctypes.c_void_p(self.frames.data_ptr()),
                                                 ctypes.c_void_p(self.slots.data_ptr()), self.n_streams,
                                                 ctypes.byref(self._n)), 'ifh_rtpjb_pop_tick')
        n = self._n.value
        return self.frames[:n], self.slots[:n]


class _SynthTable:
    """Owner of one ifh_rtpsynth_t handle."""

    def __init__(self, n_streams, ts_step, seed=None):
        import os
        self.n_streams, self.ts_step = n_streams, ts_step
        if seed is None:
            seed = int.from_bytes(os.urandom(8), 'little')          # RFC 3550 8.1: random SSRC / initial sequence / timestamp
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().ifh_rtpsynth_create(n_streams, ts_step, seed, ctypes.byref(h)), 'ifh_rtpsynth_create')
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            _lib.lib().ifh_rtpsynth_destroy(h)

    def set(self, stream, ssrc, seq, ts, marker=True):
        _lib.check(_lib.lib().ifh_rtpsynth_set(self._h, stream, ssrc, seq, ts, int(marker)), 'ifh_rtpsynth_set')

    def get(self, stream) -> dict:
        u = [ctypes.c_uint32() for _ in range(3)]
        q = [ctypes.c_int64() for _ in range(2)]
        _lib.check(_lib.lib().ifh_rtpsynth_get(self._h, stream, *[ctypes.byref(x) for x in u + q]), 'ifh_rtpsynth_get')
        return dict(ssrc=u[0].value, seq=u[1].value, ts=u[2].value, sent=q[0].value, skipped=q[1].value)

    def skip(self, stream, nframes):
        _lib.check(_lib.lib().ifh_rtpsynth_skip(self._h, stream, nframes), 'ifh_rtpsynth_skip')


class RtpSynth:
    """`rtpsynth.RtpSynth.RtpSynth(srate, ptime)` as used at RTPOutputWorker.py:88,104,136: `next_pkt(plen, pt, pload=bytes)`
    returns the datagram, `skip(n)` lets n frame times pass without a packet."""

    def __init__(self, srate: int, ptime: int, seed=None):
        self._t = _SynthTable(1, srate * ptime // 1000, seed)
        self._len = ctypes.c_int32()

    def next_pkt(self, plen: int, pt: int, pload: Optional[bytes] = None) -> bytes:
        if pload is not None:
            assert len(pload) == plen
            src = (ctypes.c_uint8 * max(plen, 1)).from_buffer_copy(pload or b'\0')
        else:
            src = None
        out = (ctypes.c_uint8 * (12 + plen))()
        _lib.check(_lib.lib().ifh_rtpsynth_next_batch(self._t._h, src, None, None, 1, plen, pt, out, ctypes.byref(self._len)),
                   'ifh_rtpsynth_next_batch')
        return bytes(out)

    def skip(self, npkts: int):
        self._t.skip(0, npkts)

    def state(self):
        return self._t.get(0)


class RTPEgressTable(_SynthTable):
    """All calls of one output thread: the [n, plen] payload matrix of `frontend.mux_encode` (host copy) -> n datagrams."""

    def __init__(self, n_streams, ts_step=160, pt=0, seed=None):
        super().__init__(n_streams, ts_step, seed)
        self.pt = pt

    def next_batch(self, payload, has=None, slots=None):
        """payload: uint8 tensor/array [n, plen] on the host; has: uint8 [n] (0 = nothing to send this tick) or None;
        slots: int32 [n] call indices or None (= 0..n-1).  -> list of n datagrams (bytes, b'' where nothing is sent)."""
        pl = np.ascontiguousarray(payload.numpy() if hasattr(payload, 'numpy') else payload, np.uint8)
        n, plen = pl.shape
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        hs = None if has is None else np.ascontiguousarray(has.numpy() if hasattr(has, 'numpy') else has, np.uint8)
        sl = None if slots is None else np.ascontiguousarray(slots.numpy() if hasattr(slots, 'numpy') else slots, np.int32)
        out = np.empty((n, 12 + plen), np.uint8)
        ln = np.empty(n, np.int32)
        _lib.check(_lib.lib().ifh_rtpsynth_next_batch(self._h, vp(pl), vp(hs), vp(sl), n, plen, self.pt, vp(out), vp(ln)),
                   'ifh_rtpsynth_next_batch')
        return [out[i, :ln[i]].tobytes() for i in range(n)]


# ------------------------------------------------------------------------------------------------
# The reference's own per-call objects (drop-in for RTP/InfernRTPIngest.py), over RtpJBuf above
# ------------------------------------------------------------------------------------------------
class AudioInput:
    """RTP/AudioInput.py:3-8"""
    vad_chunk_in: Optional[callable]
    audio_in: Optional[callable]

    def __init__(self, audio_in: Optional[callable] = None, vad_chunk_in: Optional[callable] = None):
        self.vad_chunk_in = vad_chunk_in
        self.audio_in = audio_in


class RTPParams:
    """RTP/RTPParams.py:5-13 (the codec class is G.711 here; G.722 is out of scope)."""
    default_ptime: int = 20

    def __init__(self, rtp_target, out_ptime=None, codec=None):
        assert isinstance(rtp_target, tuple) and len(rtp_target) == 2
        from .codecs import G711Codec
        self.rtp_target = rtp_target
        self.out_ptime = out_ptime if out_ptime is not None else self.default_ptime
        self.codec = codec if codec is not None else G711Codec


class WIPkt:
    def __init__(self, stream: 'RTPInStream', data, address, rtime):
        self.stream, self.data, self.address, self.rtime = stream, data, address, rtime


class WIStreamUpdate:
    def __init__(self, stream: 'RTPInStream'):
        self.stream = stream


class WIStreamConnect:
    def __init__(self, stream: 'RTPInStream', ain: AudioInput):
        self.stream, self.ain = stream, ain


class RTPInStream:
    """InfernRTPIngest.py:31-110: one call's jitter buffer -> codec bytes -> VADChannel."""
    jb_size: int = 8
    input_sr: int = 8000
    last_output_lseq: Optional[int] = None
    output_sr: int = 16000
    npkts: int = 0

    def __init__(self, ring: 'InfernRTPIngest', rtp_params, get_direct_soundout: callable):
        from .vad import VADChannel
        self.jbuf = RtpJBuf(self.jb_size)
        self.codec = rtp_params.codec().to(ring.device)
        self.ring = ring
        self.get_direct_soundout = get_direct_soundout
        self.ain = AudioInput()
        self.ain_lock = Lock()
        self.vchan = VADChannel(self.audio_chunk_out, self.vad_chunk_out, self.codec.decode, ring.device)

    def rtp_received(self, data, address, rtime):
        self.ring.pkt_queue.put(WIPkt(self, data, address, rtime))

    def stream_update(self):
        self.ring.pkt_queue.put(WIStreamUpdate(self))

    def stream_connect(self, ain: AudioInput):
        if isinstance(ain.vad_chunk_in, UUID):
            ain.vad_chunk_in = self.get_direct_soundout(ain.vad_chunk_in)
        if isinstance(ain.audio_in, UUID):
            ain.audio_in = self.get_direct_soundout(ain.audio_in)
        self.ring.pkt_queue.put(WIStreamConnect(self, ain))

    def _proc_in_tread(self, wi: Union[WIPkt, WIStreamUpdate, WIStreamConnect], svad):
        if isinstance(wi, WIStreamUpdate):
            self.jbuf = RtpJBuf(self.jb_size)
            self.last_output_lseq = None
            return
        if isinstance(wi, WIStreamConnect):
            with self.ain_lock:
                self.ain = wi.ain
            return
        try:
            res = self.jbuf.udp_in(wi.data)
        except RTPParseError as e:
            self.ring.dprint(f'InfernRTPIngest.run: RTPParseError: {e}')
            return
        self.npkts += 1
        for pkt in res:
            if pkt.content.type == RTPFrameType.ERS:
                self.last_output_lseq = pkt.content.lseq_end
                rtp_data = self.codec.silence(pkt.content.ts_diff)
            else:
                lseq = pkt.content.frame.rtp.lseq
                assert self.last_output_lseq is None or lseq == self.last_output_lseq + 1
                self.last_output_lseq = lseq
                rtp_data = pkt.rtp_data
            self.vchan.ingest(svad, rtp_data, self.codec)

    def audio_chunk_out(self, chunk, active: bool):
        chunk.active = active
        with self.ain_lock:
            if self.ain.audio_in is None:
                return
            self.ain.audio_in(chunk=chunk)

    def vad_chunk_out(self, chunk):
        with self.ain_lock:
            if self.ain.vad_chunk_in is None:
                return
            self.ain.vad_chunk_in(chunk=chunk)


class InfernRTPIngest(InfernWrkThread):
    """InfernRTPIngest.py:112-160: the thread that owns the VAD worker and drains the packet queue."""
    debug = False

    def __init__(self, device: str, vad_factory=None):
        super().__init__()
        self.pkt_queue = Queue()
        self.device = device
        self._vad_factory = vad_factory

    def start(self):
        self._start_queue = Queue()
        super().start()
        r = self._start_queue.get()
        if isinstance(r, Exception):
            super().join()
            raise r
        del self._start_queue

    def dprint(self, *args):
        if self.debug:
            print(*args)

    def run(self):
        super().thread_started()
        try:
            if self._vad_factory is not None:
                svad = self._vad_factory(self.device)
            else:
                from .vad import SileroVADWorker
                svad = SileroVADWorker(self.device)
            svad.start()
        except Exception as e:
            self._start_queue.put(e)
            return
        self._start_queue.put(0)
        while self.get_state() == RTPWrkTRun:
            wi = self.pkt_queue.get()
            if wi is None:
                break
            wi.stream._proc_in_tread(wi, svad)
        svad.stop()

    def stop(self):
        self.pkt_queue.put(None)
        super().stop()
