"""GPU parity tests for the DSP rows (a1-a9, a12 log-mel): the HIP path, reached
through the plugin classes and the C ABI, against the CPU oracle and the golden vectors.
Integer/byte work and the FIR are compared bit-exactly; log-mel within 1e-3."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dsp as odsp  # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def dev(built_lib):
    from infernos_amd import _lib
    return _lib.require_device('cuda:0')


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


# ---- G.711 -----------------------------------------------------------------------------------
def test_g711_golden_vectors(dev, golden_dir):
    from infernos_amd.codecs import G711Codec
    g = np.load(os.path.join(golden_dir, 'g711_tables.npz'))
    c = G711Codec().to(dev)
    out = c.decode(g['rand_bytes'].tobytes(), resample=False)
    assert out.samplerate == 8000 and out.audio.is_cuda
    assert np.array_equal(out.audio.cpu().numpy(), g['rand_decoded'])
    assert np.array_equal(c.decode(bytes(range(256)), resample=False).audio.cpu().numpy(), g['all_decoded'])
    assert c.encode(torch.from_numpy(g['edge_in'])) == g['edge_encoded'].tobytes()
    assert c.encode(torch.from_numpy(g['rand_float']).to(dev)) == g['rand_encoded'].tobytes()
    assert c.encode(torch.from_numpy(g['all_decoded'])) == g['roundtrip'].tobytes()
    assert c.decode(b'', resample=False).audio.numel() == 0 and c.encode(torch.zeros(0)) == b''
    assert c.silence(4) == b'\xff' * 4 and c.rtpmap() == 'rtpmap:0 PCMU/8000'


def test_g711_encode_every_int16(dev, golden_dir):
    """All 65536 table entries: feed floats that scale exactly to each int16."""
    from infernos_amd.codecs import G711Codec
    g = np.load(os.path.join(golden_dir, 'g711_tables.npz'))
    c = G711Codec().to(dev)
    ints = np.arange(-32768, 32768, dtype=np.int64)
    x = (ints.astype(np.float64) / 32767.0).astype(np.float32)
    back = np.trunc(np.clip(x * np.float32(32767.0), -32768, 32767)).astype(np.int64)
    got = np.frombuffer(c.encode(torch.from_numpy(x)), dtype=np.uint8)
    assert np.array_equal(got, g['pcm_to_ulaw'][back + 32768])
    assert np.array_equal(got, odsp.g711_encode(x))
    assert len(set(back.tolist())) > 65000


@pytest.mark.parametrize('n', [1, 3, 4, 5, 160, 768, (1 << 20) + 3])
def test_g711_bulk_matches_oracle(dev, n):
    from infernos_amd.codecs import G711Codec
    rng = np.random.default_rng(n)
    c = G711Codec().to(dev)
    b = rng.integers(0, 256, n, dtype=np.uint8)
    d = c.decode_tensor(torch.from_numpy(b)).cpu().numpy()
    assert np.array_equal(d, odsp.g711_decode(b))
    x = (rng.standard_normal(n) * 0.5).astype(np.float32)
    x[:: 97] = np.nan if n > 1000 else x[:: 97]
    assert c.encode(torch.from_numpy(x)) == odsp.g711_encode(x).tobytes()
    # unaligned views
    if n > 8:
        t = torch.from_numpy(x).to(dev)[1:]
        assert c.encode(t) == odsp.g711_encode(x[1:]).tobytes()


# ---- resampler ----------------------------------------------------------------------------------
@pytest.mark.parametrize('orig,new', [(8000, 16000), (16000, 8000)])
def test_resample_bit_exact(dev, orig, new):
    from infernos_amd.audio import Resampler
    rs = Resampler(orig, new, dev)
    o, n, nt, w, taps = rs.info()
    k, ow, oo, on = odsp.sinc_kernel(orig, new)
    assert (o, n, nt, w) == (oo, on, k.shape[1], ow)
    assert np.array_equal(taps, k), 'sinc taps differ from the oracle'
    rng = np.random.default_rng(orig)
    for L in (1, 2, 7, 160, 2047, 2048, 2049, 4097, 240000):
        x = rng.standard_normal(L).astype(np.float32)
        y = rs(torch.from_numpy(x)).cpu().numpy()
        ref = odsp.resample(x, orig, new)
        assert y.shape == ref.shape == (-(-n * L // o),)
        assert np.array_equal(y, ref), (L, np.abs(y - ref).max())
    # ragged batch
    xb = rng.standard_normal((5, 5000)).astype(np.float32)
    lens = np.array([5000, 1, 0, 4097, 333], np.int32)
    yb = rs(torch.from_numpy(xb), lens=torch.from_numpy(lens)).cpu().numpy()
    for r in range(5):
        ref = odsp.resample(xb[r, :lens[r]], orig, new)
        assert np.array_equal(yb[r, :ref.size], ref)
        assert not yb[r, ref.size:].any()
    assert rs(torch.zeros(0)).numel() == 0


def test_audio_chunk_resample_inplace(dev):
    from infernos_amd.audio import AudioChunk
    rng = np.random.default_rng(1)
    x = rng.standard_normal(801).astype(np.float32)
    c = AudioChunk(torch.from_numpy(x), 8000)
    assert c.resample(16000) is c and c.samplerate == 16000 and c.audio.is_cuda
    assert np.array_equal(c.audio.cpu().numpy(), odsp.resample(x, 8000, 16000))
    with pytest.raises(AssertionError):
        c.resample(16000)


# ---- per-tick ingest --------------------------------------------------------------------------
def test_ingest_tick_stream(dev):
    from infernos_amd.frontend import CallTable
    rng = np.random.default_rng(9)
    ncalls, nticks = 37, 23
    frames = rng.integers(0, 256, (nticks, ncalls, 160), dtype=np.uint8)
    ct = CallTable(64, dev)
    slots = torch.from_numpy(rng.permutation(64)[:ncalls].astype(np.int32)).to(dev)
    p8, p16, wins = [], [], [[] for _ in range(ncalls)]
    for t in range(nticks):
        a, b, ready = ct.tick(torch.from_numpy(frames[t]).to(dev), slots)
        p8.append(a.cpu().numpy()); p16.append(b.cpu().numpy())
        r = ready.cpu().numpy()
        w = ct.win[slots.long()].cpu().numpy()
        for i in range(ncalls):
            if r[i]:
                wins[i].append(w[i].copy())
    p8 = np.concatenate(p8, axis=1); p16 = np.concatenate(p16, axis=1)
    for i in range(ncalls):
        stream = frames[:, i].reshape(-1)
        x = odsp.g711_decode(stream)
        assert np.array_equal(p8[i], x)
        ref = odsp.resample(np.concatenate([np.zeros(8, np.float32), x]), 8000, 16000)[: p16.shape[1]]
        assert np.array_equal(p16[i], ref)
        nwin = stream.size // 768
        assert len(wins[i]) == nwin
        for k in range(nwin):
            assert np.array_equal(wins[i][k], x[k * 768:(k + 1) * 768])
    assert ct.fifo_len[slots.long()].cpu().tolist() == [(nticks * 160) % 768] * ncalls


def test_rtp_table_feeds_tick_kernel(dev):
    """8f-2 -> a7: shuffled / lossy RTP arrivals through RTPIngestTable.pop_tick + CallTable.tick give the decoded
    stream of the in-order payloads with 0xFF (decodes to 0.0) where a packet was given up on."""
    from infernos_amd.frontend import CallTable
    from infernos_amd.rtp import RTPIngestTable
    from oracle import rtp as ortp
    rng = np.random.default_rng(21)
    ncalls, npk = 9, 40
    payload = rng.integers(0, 256, (ncalls, npk, 160), dtype=np.uint8)
    tab = RTPIngestTable(ncalls, depth=4)
    ct = CallTable(16, dev)
    lost = {(c, int(k)) for c in range(ncalls) for k in rng.choice(np.arange(2, npk - 8), 2, replace=False)}
    out = [[] for _ in range(ncalls)]
    for t in range(npk + 2):
        dg, sid = [], []
        for c in range(ncalls):
            ks = [t] if t < npk else []
            if t % 5 == 1 and t + 1 < npk:
                ks = [t + 1]                   # swap neighbours: t+1 arrives first ...
            elif t % 5 == 2:
                ks = [t - 1, t]                # ... then t and the tick's own packet
            for k in ks:
                if k < npk and (c, k) not in lost:
                    dg.append(ortp.build_packet(500 * c + k, 160 * k, payload[c, k].tobytes())); sid.append(c)
        if dg:
            assert not tab.push_batch(dg, sid).any()
        while True:
            frames, slots = tab.pop_tick()
            if slots.numel() == 0:
                break
            p8, _, _ = ct.tick(frames.to(dev, non_blocking=True), slots.to(dev, non_blocking=True), want_ready=False)
            p8 = p8.cpu().numpy()
            for j, c in enumerate(slots.tolist()):
                out[c].append(p8[j])
    for c in range(ncalls):
        want = payload[c].copy()
        for (cc, k) in lost:
            if cc == c:
                want[k] = 0xff
        got = np.concatenate(out[c])
        assert got.size == npk * 160
        assert np.array_equal(got, odsp.g711_decode(want.reshape(-1)))
        st = tab.stats(c)
        assert st['ers_packets'] == 2 and st['ers_bytes'] == 320 and st['held'] == 0


# ---- VAD ----------------------------------------------------------------------------------------
class ScriptedModel:
    def __init__(self, dev):
        self.dev, self.script = dev, []

    def reset_states(self):
        pass

    def __call__(self, x, sr):
        p = self.script.pop(0)
        assert len(p) == x.size(0)
        return torch.tensor(p, dtype=torch.float32, device=self.dev)


def _replay(sc, dev):
    from infernos_amd.codecs import G711Codec
    from infernos_amd.vad import SileroVADWorker, VADChannel
    model = ScriptedModel(dev)
    w = SileroVADWorker(dev, 8000, model=model, max_channels=2)
    codec = G711Codec().to(dev)
    rng = np.random.default_rng(sc['seed'])
    events, chans = [], []
    for ci in range(sc['nch']):
        def a_in(chunk, active, ci=ci):
            events.append(['raw', ci, bool(active), int(chunk.audio.size(0)), sha16(chunk.audio.cpu().numpy())])

        def v_in(chunk, ci=ci):
            events.append(['vad', ci, int(chunk.ipos), int(chunk.audio.size(0)), int(chunk.samplerate),
                           sha16(chunk.audio.cpu().numpy())])
        chans.append(VADChannel(a_in, v_in, None, dev))
    pkts = rng.integers(0, 256, (sc['nch'], sc['npkts'], 160), dtype=np.uint8)
    probs = list(sc['probs'])
    for pi in range(sc['npkts']):
        for ci, ch in enumerate(chans):
            ch.ingest(w, pkts[ci, pi].tobytes(), codec)
        wis = []
        while not w.inf_queue.empty():
            wis.append(w.inf_queue.get_nowait())
        if wis:
            order, pr = probs.pop(0)
            assert [chans.index(wi[0]) for wi in wis] == order
            model.script.append(pr)
            try:
                w.process_batch(wis)
            except AssertionError:
                events.append(['assert', pi])
                break
    final = [{'triggered': bool(c.state.triggered), 'temp_end': int(c.state.temp_end),
              'current_sample': int(c.state.current_sample),
              'active_start': None if c.active_start is None else int(c.active_start),
              'buf_len': int(c.buf_len), 'fifo_len': len(c.vad_buffer)} for c in chans]
    return events, final


def test_vad_worker_replays_reference_traces(dev, golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'vad_traces.json')))
    for sc in g['scenarios']:
        events, final = _replay(sc, dev)
        assert events == sc['events'], sc['name']
        if not (events and events[-1][0] == 'assert'):
            assert final == sc['final'], sc['name']
        else:
            for a, b in zip(final, sc['final']):
                assert (a['triggered'], a['current_sample']) == (b['triggered'], b['current_sample'])


class StatefulFakeModel:
    """tools/gen_golden.py:_StatefulFakeSilero on device tensors: h += [first sample > 0], c += 1, p from (h + c) mod 5."""

    def __init__(self, dev):
        import types
        self.dev = dev
        self._c = types.SimpleNamespace(_h=None, _c=None, _last_sr=0, _last_batch_size=0)

    def reset_states(self):
        pass

    def __call__(self, x, sr):
        mc = self._c
        assert mc._last_batch_size == x.size(0) and mc._last_sr == sr and tuple(mc._h.shape) == (2, x.size(0), 64)
        assert mc._h.is_cuda and mc._c.is_cuda
        e = (x[:, 0] > 0).to(torch.float32)
        mc._h = mc._h + e[None, :, None]
        mc._c = mc._c + 1.0
        k = torch.remainder(mc._h[0, :, 0] + mc._c[1, :, 63], 5.0)
        return torch.where(k >= 3.0, 0.9, 0.1).to(torch.float32)


def test_vad_worker_stateful_model_replays_reference_trace(dev, golden_dir):
    """Per-channel recurrent model state is injected before and saved after every model call by slot
    (SileroVADUtils.py:21-26,99,131), with sub-batch membership, order and duplicates changing from batch to batch:
    events, final FSM state and final model state equal the reference run with the same fake stateful model."""
    from infernos_amd.codecs import G711Codec
    from infernos_amd.vad import SileroVADWorker, VADChannel
    g = json.load(open(os.path.join(golden_dir, 'vad_stateful_trace.json')))
    w = SileroVADWorker(dev, 8000, model=StatefulFakeModel(dev), max_channels=2)     # forces two table growths
    codec = G711Codec().to(dev)
    rng = np.random.default_rng(g['seed'])
    nch, npkts = g['nch'], g['npkts']
    events, chans = [], []
    for ci in range(nch):
        def a_in(chunk, active, ci=ci):
            events.append(['raw', ci, bool(active), int(chunk.audio.size(0)), sha16(chunk.audio.cpu().numpy())])

        def v_in(chunk, ci=ci):
            events.append(['vad', ci, int(chunk.ipos), int(chunk.audio.size(0)), int(chunk.samplerate),
                           sha16(chunk.audio.cpu().numpy())])
        chans.append(VADChannel(a_in, v_in, None, dev))
    pkts = rng.integers(0, 256, (nch, npkts, 160), dtype=np.uint8)
    orders = list(g['orders'])
    keep = lambda ci, pi: (pi * 7 + ci * 3) % (ci + 3) != 0
    for pi in range(npkts):
        for ci in (range(nch) if pi % 2 == 0 else reversed(range(nch))):
            if keep(ci, pi):
                chans[ci].ingest(w, pkts[ci, pi].tobytes(), codec)
        if pi % 13 not in (2, 5, 12):
            continue
        wis = []
        while not w.inf_queue.empty():
            wis.append(w.inf_queue.get_nowait())
        if wis:
            assert [chans.index(wi[0]) for wi in wis] == orders.pop(0)
            w.process_batch(wis)
    assert events == g['events']
    final = [{'triggered': bool(c.state.triggered), 'temp_end': int(c.state.temp_end),
              'current_sample': int(c.state.current_sample),
              'active_start': None if c.active_start is None else int(c.active_start),
              'buf_len': int(c.buf_len), 'fifo_len': len(c.vad_buffer),
              'h': float(c.state.model_state[0][0, 0]), 'c': float(c.state.model_state[1][1, 63])} for c in chans]
    assert final == g['final']


def test_vad_worker_slots_follow_concurrent_calls(dev):
    """Rows of the device tables are released with the channel (detach or garbage collection) and reused with a clean
    state: table size follows the concurrent, not the cumulative, number of calls."""
    import gc
    from infernos_amd.codecs import G711Codec
    from infernos_amd.vad import SileroVADWorker, VADChannel
    w = SileroVADWorker(dev, 8000, model=StatefulFakeModel(dev), max_channels=4)
    codec = G711Codec().to(dev)
    rng = np.random.default_rng(3)
    seen = []

    def one_call(n_windows):
        ch = VADChannel(lambda chunk, active: None, lambda chunk: None, None, dev)
        for _ in range(n_windows):
            ch.ingest(w, rng.integers(0, 256, 768, dtype=np.uint8).tobytes(), codec)
            w.process_batch([w.inf_queue.get_nowait()])
        return ch
    for k in range(40):                             # 40 calls one after another, at most 3 alive at a time
        ch = one_call(2 + k % 3)
        assert ch.state.current_sample == 768 * (2 + k % 3)              # a reused row starts from a clean FSM state
        assert float(ch.state.model_state[1][0, 0]) == 2 + k % 3         # ... and a clean model state
        seen.append(ch)
        if len(seen) == 3:
            seen.pop(0).detach()
            del ch
            seen.pop(0)                             # dropped without detach(): the finalizer releases the row
            gc.collect()
    assert w._cap == 4 and w.channels_attached <= 3


def test_vad_iterator_fsm_matches_oracle(dev):
    import ctypes
    from infernos_amd.vad import VADBatchState, VADIteratorB
    model = ScriptedModel(dev)
    it = VADIteratorB(model, sampling_rate=8000)
    rng = np.random.default_rng(4)
    n, steps = 33, 200
    P = rng.random((steps, n)).astype(np.float32)
    P[:, 0] = 0.5; P[:, 1] = np.float32(0.35); P[::7, 2] = 0.9
    bs = VADBatchState(n, device=str(dev))

    class St(ctypes.Structure):
        _fields_ = [('triggered', ctypes.c_int32), ('temp_end', ctypes.c_int64), ('current_sample', ctypes.c_int64)]
    sts = (St * n)()
    L = odsp.lib()
    for s in range(steps):
        model.script.append(P[s].tolist())
        it(torch.zeros(n, 768, device=dev), bstate=bs)
        kind = np.zeros(n, np.int32); pos = np.zeros(n, np.int64)
        pd = P[s].astype(np.float64)
        L.orc_vad_fsm_step(sts, pd.ctypes.data_as(ctypes.c_void_p), n, 768, 8000, ctypes.c_double(0.5),
                           kind.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p))
        for i, c in enumerate(bs.channels):
            assert (int(c.triggered), c.temp_end, c.current_sample) == \
                (sts[i].triggered, sts[i].temp_end, sts[i].current_sample)
            exp = None if kind[i] == 0 else ({'start': int(pos[i])} if kind[i] == 1 else {'end': int(pos[i])})
            assert c.speech == exp


def test_energy_vad_standin_is_monotone(dev):
    from infernos_amd.vad import EnergyVADModel
    m = EnergyVADModel(dev)
    x = torch.randn(4, 768, device=dev) * torch.tensor([1e-4, 1e-3, 1e-2, 0.3], device=dev)[:, None]
    p = m(x, 8000).cpu().numpy()
    assert (np.diff(p) > 0).all() and p[0] < 0.05 and p[-1] > 0.95


# ---- Whisper log-mel -----------------------------------------------------------------------------
def test_logmel_matches_reference_vectors(dev, golden_dir):
    from infernos_amd.features import WhisperLogMel
    from infernos_amd.synth import synth_utterance
    meta = json.load(open(os.path.join(golden_dir, 'logmel_meta.json')))
    g = np.load(os.path.join(golden_dir, 'logmel.npz'))
    lm = WhisperLogMel(80, dev)
    assert hashlib.sha256(odsp.mel_filter_bank().astype(np.float32).tobytes()).hexdigest() == \
        hashlib.sha256(lm.filters().tobytes()).hexdigest()
    auds, seeds = [], []
    for seed, c in meta['cases'].items():
        secs = c['seconds']
        x8 = synth_utterance(int(seed), max(secs, 2.5))[: int(secs * 8000)]
        auds.append(odsp.resample(x8, 8000, 16000)); seeds.append(seed)
    L = max(a.size for a in auds)
    xb = np.zeros((len(auds), L), np.float32)
    for i, a in enumerate(auds):
        xb[i, :a.size] = a
    lens = torch.tensor([a.size for a in auds], dtype=torch.int32)
    out = lm(torch.from_numpy(xb), lens=lens).cpu().numpy()
    for i, seed in enumerate(seeds):
        np.testing.assert_allclose(out[i][:, ::37], g['frames_%s' % seed], rtol=0, atol=1e-3)
        ref = odsp.logmel(auds[i])
        err = np.abs(out[i] - ref).max()
        assert err < 1e-3, (seed, err)
        assert abs(float(out[i].max()) - meta['cases'][seed]['max']) < 1e-3


def test_logmel_edge_cases(dev):
    from infernos_amd.features import WhisperLogMel
    lm = WhisperLogMel(80, dev)
    rng = np.random.default_rng(12)
    full = (0.2 * rng.standard_normal(480000 + 999)).astype(np.float32)     # longer than 30 s: truncated
    xb = np.zeros((4, full.size), np.float32)
    xb[0] = full
    xb[1, :470500] = full[:470500]       # ends inside the last block: reflect tail must see real samples
    xb[2, :1] = 0.5                      # one sample
    lens = torch.tensor([full.size, 470500, 1, 0], dtype=torch.int32)
    out = lm(torch.from_numpy(xb), lens=lens).cpu().numpy()
    for i in range(4):
        ref = odsp.logmel(xb[i, :int(lens[i])])
        assert np.abs(out[i] - ref).max() < 1e-3, i
    assert np.allclose(out[3], (np.log10(1e-10) + 4) / 4, atol=1e-6)       # silence: constant plane
    ob = lm(torch.from_numpy(xb[:2]), lens=lens[:2], out_dtype=torch.bfloat16)
    assert ob.dtype == torch.bfloat16
    assert np.abs(ob.float().cpu().numpy() - out[:2]).max() < 1e-2
    lm128 = WhisperLogMel(128, dev)
    o128 = lm128(torch.from_numpy(xb[:1, :160000])).cpu().numpy()
    assert np.abs(o128[0] - odsp.logmel(xb[0, :160000], n_mel=128)).max() < 1e-3


def test_logmel_of_a_window_does_not_depend_on_the_batch(dev):
    """A size-independent property at the bench configuration's batch (128 windows of 30 s, ragged lengths): the [80, 3000] plane of a
    window -- and the raw plane + window maximum the STT stage consumes -- is the same bits alone, among 3 and among 128."""
    from infernos_amd.features import WhisperLogMel
    lm = WhisperLogMel(80, dev)
    rng = np.random.default_rng(5)
    n = 128
    x = (0.1 * rng.standard_normal((n, 480000))).astype(np.float32)
    lens_np = rng.integers(1, 480001, size=n).astype(np.int32)
    lens_np[:4] = (480000, 160000, 1, 0)
    for i in range(n):
        x[i, lens_np[i]:] = 0.0
    xb, lens = torch.from_numpy(x).to(dev), torch.from_numpy(lens_np)
    full = lm(xb, lens=lens).clone()
    raw_full, wmax_full = lm.raw(xb, lens=lens)
    raw_full, wmax_full = raw_full.clone(), wmax_full.clone()
    assert bool(torch.isfinite(full).all())
    for lo, m in ((0, 1), (2, 1), (3, 1), (n - 1, 1), (60, 3), (0, 64)):
        part = lm(xb[lo:lo + m].contiguous(), lens=lens[lo:lo + m])
        assert torch.equal(part.view(torch.int32), full[lo:lo + m].view(torch.int32)), (lo, m)
        raw, wm = lm.raw(xb[lo:lo + m].contiguous(), lens=lens[lo:lo + m])
        assert torch.equal(raw.view(torch.int32), raw_full[lo:lo + m].view(torch.int32)), ('raw', lo, m)
        assert torch.equal(wm, wmax_full[lo:lo + m]), ('max', lo, m)


# ---- output mix + encode (8f-1) ---------------------------------------------------------------------
def test_mux_encode_matches_oracle(dev):
    from infernos_amd.frontend import mux_encode
    rng = np.random.default_rng(21)
    n, K, L = 37, 3, 800
    tr = (rng.standard_normal((n, K, L)) * 0.4).astype(np.float32)
    present = rng.random((n, K)) < 0.6
    present[0] = False; present[1] = [True, False, False]; present[2] = True
    tr[~present] = 0
    ndiv = np.maximum(present.sum(1), 1).astype(np.int32)
    ndiv[3] = 3                       # a track that exists but produced no block still divides
    ref, has_ref = odsp.mux_encode(tr, present, ndiv)
    out, has = mux_encode(torch.from_numpy(tr), torch.from_numpy(present), torch.from_numpy(ndiv))
    assert np.array_equal(has.cpu().numpy().astype(bool), has_ref)
    o = out.cpu().numpy()
    assert np.array_equal(o[has_ref], ref[has_ref])


# ---- G.722 ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('eight_k', [True, False])
def test_g722_batch_matches_oracle_bit_exact(dev, eight_k):
    """ifh_g722_encode / _decode for 37 calls, three consecutive frames with carried state, int16 and f32 interfaces, against
    the CPU restatement call by call (integer recursion: bit-exact; the f32 interface is the wrapper's conversion)."""
    from infernos_amd.codecs import G722Batch
    N, F = 37, 3
    S = 160 if eight_k else 320
    rng = np.random.default_rng(5 + int(eight_k))
    t = np.arange(F * S)
    pcm = np.stack([np.clip((0.5 * np.sin(t * (0.02 + 0.003 * c)) + 0.1 * rng.standard_normal(F * S)) * (3000 + 700 * c), -32768, 32767)
                    for c in range(N)]).astype(np.int16)
    pcm[5] = 0
    pcm[6, ::2], pcm[6, 1::2] = 32767, -32768
    b16, bf = G722Batch(N, dev, eight_k), G722Batch(N, dev, eight_k)
    enc_st = [odsp.g722_new_state() for _ in range(N)]
    dec_st = [odsp.g722_new_state() for _ in range(N)]
    for f in range(F):
        fr = pcm[:, f * S:(f + 1) * S]
        code = b16.encode(torch.from_numpy(fr.copy()).to(dev))
        ref = np.stack([odsp.g722_encode(enc_st[c], fr[c], eight_k) for c in range(N)])
        assert np.array_equal(code.cpu().numpy(), ref), f
        # the f32 interface: x -> clamp(x * 32767) truncated, as the reference wrapper prepares its int16
        xf = (fr.astype(np.float32) / 32767.0)
        as16 = np.clip(xf * np.float32(32767.0), -32768, 32767).astype(np.int16)
        codef = bf.encode(torch.from_numpy(xf).to(dev))
        if f == 0:
            reff = np.stack([odsp.g722_encode(odsp.g722_new_state(), as16[c], eight_k) for c in range(N)])
            assert np.array_equal(codef.cpu().numpy(), reff)
        dec = b16.decode(code, f32=False).cpu().numpy()
        refd = np.stack([odsp.g722_decode(dec_st[c], ref[c], eight_k) for c in range(N)])
        assert np.array_equal(dec, refd), f
    assert np.array_equal(b16.enc_state.cpu().numpy()[:, :114], np.stack(enc_st)[:, :114])
    d32 = G722Batch(N, dev, eight_k)
    o32 = d32.decode(torch.from_numpy(np.stack([odsp.g722_encode(odsp.g722_new_state(), pcm[c, :S], eight_k) for c in range(N)])).to(dev))
    o16 = np.stack([odsp.g722_decode(odsp.g722_new_state(), odsp.g722_encode(odsp.g722_new_state(), pcm[c, :S], eight_k), eight_k) for c in range(N)])
    assert np.array_equal(o32.cpu().numpy(), o16.astype(np.float32) / np.float32(32767.0))


def test_g722_codec_wrapper(dev):
    """G722Codec as Core/Codecs/G722.py:8-56 uses it: bytes out of encode (state carried across calls), AudioChunk out of
    decode (optionally resampled), silence()."""
    from infernos_amd.audio import AudioChunk
    from infernos_amd.codecs import G722Codec
    c = G722Codec().to(dev)
    rng = np.random.default_rng(9)
    x = (0.3 * np.sin(np.arange(480) * 0.07) + 0.02 * rng.standard_normal(480)).astype(np.float32)
    st = odsp.g722_new_state()
    as16 = np.clip(x * np.float32(32767.0), -32768, 32767).astype(np.int16)
    for f in range(3):
        b = c.encode(torch.from_numpy(x[f * 160:(f + 1) * 160]))
        assert isinstance(b, bytes) and b == odsp.g722_encode(st, as16[f * 160:(f + 1) * 160]).tobytes()
    dst = odsp.g722_new_state()
    code = odsp.g722_encode(odsp.g722_new_state(), as16)
    ch = c.decode(code[:160].tobytes())
    assert isinstance(ch, AudioChunk) and ch.samplerate == 8000 and ch.audio.numel() == 160
    assert np.array_equal(ch.audio.cpu().numpy(), odsp.g722_decode(dst, code[:160]).astype(np.float32) / np.float32(32767.0))
    ch2 = c.decode(code[160:320].tobytes(), resample=True, sample_rate=16000)
    assert ch2.samplerate == 16000 and ch2.audio.numel() == 320
    assert len(c.silence(160)) == 160 and c.decode(b'').audio.numel() == 0 and c.encode(torch.zeros(0)) == b''
