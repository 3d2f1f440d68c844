"""CPU: the RTP ingress stage (ifh_rtp_parse / ifh_rtpjb_*, host-only C++ in the C-ABI library)
against oracle/rtp.py on scripted and random arrival orders, and the call-site contract of
RTP/InfernRTPIngest.py:76-96.  PARITY UNPINNED vs the third-party rtpsynth the reference uses (see oracle/rtp.py)."""
import os

import numpy as np
import pytest

from oracle import rtp as O


@pytest.fixture(scope='module')
def R(built_lib):
    from infernos_amd import rtp
    return rtp


def _norm(ready, R):
    out = []
    for f in ready:
        c = f.content
        if c.type == R.RTPFrameType.ERS:
            out.append(('ers', c.lseq_start, c.lseq_end, c.ts_diff))
        else:
            out.append(('rtp', c.frame.rtp.lseq, c.frame.rtp.ts, f.rtp_data))
    return out


def test_parse_fields_and_errors(R):
    rng = np.random.default_rng(0)
    pl = bytes(rng.integers(0, 256, 160, dtype=np.uint8))
    cases = [dict(), dict(marker=1, pt=8), dict(csrc=(1, 2, 3)), dict(ext=(0xbede, b'\1\2\3\4' * 2)), dict(pad=4),
             dict(csrc=(7,), ext=(1, b''), pad=1, marker=1, pt=101)]
    for kw in cases:
        pkt = O.build_packet(65535, 0xfffffff0, pl, ssrc=0xdeadbeef, **kw)
        assert R.rtp_parse(pkt) == O.parse(pkt), kw
        assert pkt[R.rtp_parse(pkt)['payload_off']:][:160] == pl
    good = O.build_packet(1, 160, pl, pad=4)
    bad = [b'', good[:11], bytes([0x40]) + good[1:], O.build_packet(1, 0, b'', csrc=(1, 2))[:16],
           O.build_packet(1, 0, b'', ext=(1, b'\0' * 8))[:18], good[:-1] + b'\0', good[:-1] + bytes([200]),
           bytes([0xa0]) + b'\0' * 11]
    for pkt in bad:
        with pytest.raises(O.ParseError):
            O.parse(pkt)
        with pytest.raises(R.RTPParseError):
            R.rtp_parse(pkt)


def test_in_order_stream_passes_straight_through(R):
    jb = R.RtpJBuf(8)
    for i in range(40):
        pl = bytes([i]) * 160
        res = jb.udp_in(O.build_packet(65530 + i, 160 * i, pl))     # wraps through 65535 -> 0
        assert len(res) == 1 and res[0].rtp_data == pl
        assert res[0].content.type == R.RTPFrameType.RTP and res[0].content.frame.rtp.lseq == 65530 + i
    st = jb.stats()
    assert st['released'] == 40 and st['held'] == 0 and st['last_lseq'] == 65569 and st['fifo_bytes'] == 40 * 160


def test_scripted_reorder_loss_duplicate(R):
    """Worked example: depth 2.  0,1 in order; 3 and 4 wait for 2; 2 arrives -> 2,3,4; a second copy of 3 is late;
    6,7 wait (a second copy of 6 is a duplicate); 8 overflows the buffer -> ERS(5..5, ts_diff=160) then 6,7,8;
    5 arrives late -> dropped."""
    jb, oj = R.RtpJBuf(2), O.JBuf(2)
    pk = lambda s: O.build_packet(s, 160 * s, bytes([s]) * 160)
    got = []
    for s in (0, 1, 3, 4, 2, 3, 6, 6, 7, 8, 5):
        r = _norm(jb.udp_in(pk(s)), R)
        assert r == oj.udp_in(pk(s)), s
        got.append([(f[0],) + tuple(f[1:3]) for f in r])
    assert got == [[('rtp', 0, 0)], [('rtp', 1, 160)], [], [], [('rtp', 2, 320), ('rtp', 3, 480), ('rtp', 4, 640)], [], [],
                   [], [], [('ers', 5, 5), ('rtp', 6, 960), ('rtp', 7, 1120), ('rtp', 8, 1280)], []]
    st = jb.stats()
    assert (st['late'], st['duplicate'], st['ers_events'], st['ers_packets'], st['ers_bytes']) == (2, 1, 1, 1, 160)
    assert {k: st[k] for k in oj.counts if k != 'ers_bytes'} == {k: v for k, v in oj.counts.items() if k != 'ers_bytes'}


@pytest.mark.parametrize('seed', range(6))
def test_random_arrivals_match_oracle_and_call_site_contract(R, seed):
    rng = np.random.default_rng(seed)
    depth = int(rng.integers(1, 9))
    n = 600
    start = int(rng.integers(0, 65536))
    ts0 = int(rng.integers(0, 2 ** 32))
    order = np.arange(n) + rng.integers(0, 12, n) * (rng.random(n) < 0.25)        # local displacement
    idx = np.argsort(order, kind='stable')
    idx = idx[rng.random(n) >= 0.08]                                             # loss
    idx = np.concatenate([idx, rng.choice(idx, 20)])                             # duplicates / very late copies
    idx = idx[np.argsort(np.concatenate([np.arange(len(idx) - 20), rng.integers(0, len(idx), 20)]), kind='stable')]
    jb, oj = R.RtpJBuf(depth), O.JBuf(depth)
    frames, last = [], None
    for k, i in enumerate(idx):
        i = int(i)
        pl = bytes(np.random.default_rng(1000 + i).integers(0, 256, 160, dtype=np.uint8))
        pkt = O.build_packet(start + i, ts0 + 160 * i, pl, marker=int(i == 0))
        if k % 97 == 5:                                 # a malformed datagram changes nothing
            with pytest.raises(R.RTPParseError):
                jb.udp_in(pkt[:7])
        r = _norm(jb.udp_in(pkt), R)
        assert r == oj.udp_in(pkt)
        for f in r:                                     # InfernRTPIngest.py:84-92
            lo, hi = (f[1], f[2]) if f[0] == 'ers' else (f[1], f[1])
            assert last is None or lo == last + 1
            assert hi >= lo
            last = hi
        frames += r
    # every byte position of the stream is filled exactly once: payload or silence, aligned on the timestamps
    blob = O.released_bytes(frames)
    first = frames[0][1] - start
    assert len(blob) == (last - start - first + 1) * 160
    for f in frames:
        if f[0] == 'rtp':
            i = f[1] - start
            assert blob[(i - first) * 160:(i - first + 1) * 160] == f[3]
    st = jb.stats()
    assert st['fifo_bytes'] == min(len(blob), 8192) and st['parse_errors'] >= 1


def test_table_pop_tick_frames_and_reset(R):
    import torch
    N = 5
    tab = R.RTPIngestTable(N, depth=4, pin=False)
    rng = np.random.default_rng(3)
    data = {s: bytes(rng.integers(0, 256, 160 * 6, dtype=np.uint8)) for s in range(N)}
    pk = lambda s, i, size=160: O.build_packet(100 * s + i, 160 * i, data[s][160 * i:160 * i + size])
    # tick 0: calls 0..3 deliver packet 0; call 4 delivers nothing; one malformed datagram for call 2
    st = tab.push_batch([pk(0, 0), pk(1, 0), pk(2, 0), b'\x80\0', pk(3, 0)], [0, 1, 2, 2, 3])
    assert list(st) == [0, 0, 0, -4, 0]
    frames, slots = tab.pop_tick()
    assert slots.tolist() == [0, 1, 2, 3] and all(bytes(frames[k].numpy()) == data[s][:160] for k, s in enumerate(slots.tolist()))
    assert tab.pop_tick()[1].numel() == 0
    # call 0: packet 2 before packet 1 -> nothing, then both; call 1: an 80-byte packet is half a frame
    assert list(tab.push_batch([pk(0, 2), pk(1, 1, 80)], [0, 1])) == [0, 0]
    assert tab.pop_tick()[1].numel() == 0
    tab.push_batch([pk(0, 1)], [0])
    f, s = tab.pop_tick()
    assert s.tolist() == [0] and bytes(f[0].numpy()) == data[0][160:320]
    f, s = tab.pop_tick()
    assert s.tolist() == [0] and bytes(f[0].numpy()) == data[0][320:480]
    # call 3 loses packets 1..2 and gives up after `depth` held packets: 320 bytes of 0xFF, then 3,4,5...
    tab.push_batch([O.build_packet(300 + i, 160 * i, bytes([i]) * 160) for i in (3, 4, 5, 6, 7)], [3] * 5)
    got = []
    while True:
        f, s = tab.pop_tick()
        if s.numel() == 0:
            break
        assert s.tolist() == [3]
        got.append(bytes(f[0].numpy()))
    assert got == [b'\xff' * 160] * 2 + [bytes([i]) * 160 for i in (3, 4, 5, 6, 7)]
    assert tab.stats(3)['ers_bytes'] == 320
    # WIStreamUpdate: a fresh buffer accepts an unrelated sequence origin
    tab.reset(0)
    tab.push_batch([O.build_packet(7, 0, b'\x55' * 160)], [0])
    f, s = tab.pop_tick()
    assert s.tolist() == [0] and bytes(f[0].numpy()) == b'\x55' * 160
    assert tab.frames.dtype == torch.uint8 and tab.slots.dtype == torch.int32
    with pytest.raises(Exception):
        tab.push(N, pk(0, 0))


def test_compat_names(R):
    import infernos_amd.compat as compat
    compat.install()
    from rtpsynth.RtpJBuf import RtpJBuf, RTPFrameType, RTPParseError     # noqa: F401  (InfernRTPIngest.py:6)
    assert RtpJBuf is R.RtpJBuf
    # the per-packet thread of RTP/InfernRTPIngest.py is the reference's own file, not mirrored here
    assert 'RTP.InfernRTPIngest' not in compat.MAP and not hasattr(R, 'RTPInStream')


REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference tree (build container only)')
def test_reference_rtp_ingest_runs_unchanged_on_the_aliased_names(R):
    """INTEGRATION.md's claim: the reference's own RTP/InfernRTPIngest.py (loaded from where it lies, nothing copied)
    works once compat.install() has aliased rtpsynth.RtpJBuf / Core.VAD.SileroVAD / Core.Codecs.G711 / Core.AudioChunk /
    Core.InfernWrkThread.  Shuffled / lossy datagrams reach VADChannel.ingest as the in-order payload bytes with
    codec.silence(ts_diff) where a gap was given up (InfernRTPIngest.py:63-100)."""
    import importlib.util
    import sys
    import types
    import infernos_amd.compat as compat
    compat.install()
    saved = {k: sys.modules.get(k) for k in ('Core.Codecs.G722', 'RTP', 'RTP.AudioInput', 'RTP.RTPParams', 'RTP.InfernRTPIngest')}
    try:
        g722 = types.ModuleType('Core.Codecs.G722')          # third-party G722 is absent: name only
        g722.G722Codec = type('G722Codec', (), {})
        sys.modules['Core.Codecs.G722'] = g722
        pkg = types.ModuleType('RTP')
        pkg.__path__ = [os.path.join(REF, 'RTP')]
        sys.modules['RTP'] = pkg
        sys.dont_write_bytecode = True

        def load(name):
            spec = importlib.util.spec_from_file_location('RTP.' + name, os.path.join(REF, 'RTP', name + '.py'))
            m = importlib.util.module_from_spec(spec)
            sys.modules['RTP.' + name] = m
            spec.loader.exec_module(m)
            return m
        load('AudioInput')
        load('RTPParams')
        ing = load('InfernRTPIngest')
        assert ing.RtpJBuf is R.RtpJBuf

        class Codec:
            def to(self, dev): return self
            def decode(self, *a, **k): raise AssertionError('not reached: the fake channel swallows the bytes')
            def silence(self, n): return b'\xff' * n

        class Params:
            codec = Codec

        class Ring:
            device, debug = 'cpu', False
            def __init__(self):
                from queue import Queue
                self.pkt_queue = Queue()
            def dprint(self, *a): pass

        ring = Ring()
        st = ing.RTPInStream(ring, Params(), get_direct_soundout=lambda u: None)
        st.jbuf = R.RtpJBuf(2)                        # small buffer so that a loss is given up on quickly
        fed = []

        class Chan:
            def ingest(self, svad, data, codec):
                fed.append(bytes(data))
        st.vchan = Chan()
        pk = lambda s: O.build_packet(1000 + s, 160 * s, bytes([s]) * 160)
        for s in (0, 2, 1, 3, 5, 6, 7, 8):            # 4 is lost
            st.rtp_received(pk(s), ('10.0.0.1', 5004), 0.0)
        st.rtp_received(b'\x80\x00', ('10.0.0.1', 5004), 0.0)
        while not ring.pkt_queue.empty():
            wi = ring.pkt_queue.get()
            wi.stream._proc_in_tread(wi, svad=None)
        assert b''.join(fed) == b''.join(bytes([s]) * 160 for s in (0, 1, 2, 3)) + b'\xff' * 160 + \
            b''.join(bytes([s]) * 160 for s in (5, 6, 7, 8))
        assert st.last_output_lseq == 1008 and st.npkts == 8
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
