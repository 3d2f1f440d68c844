#!/usr/bin/env python3
"""Capture golden vectors by RUNNING the reference (sippy/Infernos, /root/reference) here.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [section ...]

Sections: g711 vad vad_stateful stt batched muxer logmel tts whisper (default: all).
Only numbers (inputs/expected outputs) are written, to tests/golden/.  The reference
cannot travel to the GPU box; these fixtures can.  Third-party arithmetic the reference
calls (transformers) is exercised through the reference's own call sites wherever the
reference has one; where it has none reachable offline the fixture says so.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402

refload.install()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def dump_json(name, obj):
    with open(os.path.join(GOLD, name), 'w') as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print('wrote', name)


# --------------------------------------------------------------------------------------
def gen_g711():
    from Core.Codecs.G711 import G711Codec, _pcm_to_ulaw_ct, _ulaw_to_pcm_ct
    c = G711Codec()
    rng = np.random.default_rng(0)
    rb = rng.integers(0, 256, 4096, dtype=np.uint8)
    dec = c.decode(rb.tobytes(), resample=False).audio.numpy()
    edge = np.array([0, 1, -1, 0.5, 1e-4, -1e-4, 0.999999, -0.999999, 2.0, -2.0, 1.00001, -1.00003,
                     3.0518e-05, -3.0518e-05, 0.25, -0.75, 6.1e-5, 0.12345, -0.54321], dtype=np.float32)
    enc_edge = np.frombuffer(c.encode(torch.from_numpy(edge)), dtype=np.uint8)
    xr = (rng.standard_normal(8000) * 0.4).astype(np.float32)
    enc_r = np.frombuffer(c.encode(torch.from_numpy(xr)), dtype=np.uint8)
    all_codes = np.arange(256, dtype=np.uint8)
    dec_all = c.decode(all_codes.tobytes(), resample=False).audio.numpy()
    rt = np.frombuffer(c.encode(torch.from_numpy(dec_all)), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, 'g711_tables.npz'),
                        ulaw_to_pcm=_ulaw_to_pcm_ct.numpy(), pcm_to_ulaw=_pcm_to_ulaw_ct.numpy(),
                        rand_bytes=rb, rand_decoded=dec, edge_in=edge, edge_encoded=enc_edge,
                        rand_float=xr, rand_encoded=enc_r, all_decoded=dec_all, roundtrip=rt)
    meta = {
        'source': 'Core/Codecs/G711.py:7-47 run in the build container (python audioop)',
        'sha256_ulaw_to_pcm_i16le': sha(_ulaw_to_pcm_ct.numpy()),
        'sha256_pcm_to_ulaw_u8': sha(_pcm_to_ulaw_ct.numpy()),
        'rtpmap': G711Codec.rtpmap(), 'silence_3': list(c.silence(3)),
        'e2d_160_16000': c.e2d_frames(160, 16000), 'd2e_768_8000': c.d2e_frames(768, 8000),
        'd2e_768_16000': c.d2e_frames(768, 16000),
        'srate': c.srate, 'crate': c.crate, 'ptype': c.ptype, 'ename': c.ename,
    }
    dump_json('g711_meta.json', meta)


# --------------------------------------------------------------------------------------
class _FakeSilero:
    """Stands in for the Silero JIT model object: exposes what VADIteratorB touches
    (SileroVADUtils.py:72,99,103,131) and returns scripted probabilities."""
    def __init__(self):
        import types
        self._c = types.SimpleNamespace(_h=None, _c=None, _last_sr=0, _last_batch_size=0)
        self.script = []

    def reset_states(self):
        pass

    def __call__(self, x, sr):
        assert x.dim() == 2
        p = self.script.pop(0)
        assert len(p) == x.size(0), (len(p), x.size(0))
        return torch.tensor(p, dtype=torch.float32)


def _mk_vad_worker():
    from Core.VAD.SileroVAD import SileroVADWorker
    from Core.VAD.SileroVADUtils import VADIteratorB
    from Cluster.InfernBatchedWorker import InfernBatchedWorker
    w = object.__new__(SileroVADWorker)
    InfernBatchedWorker.__init__(w)
    w.device = 'cpu'
    w.model = _FakeSilero()
    w.vad_iterator = VADIteratorB(w.model, sampling_rate=8000)
    w.window_size_samples = 768
    w.input_sr = 8000
    w.max_vad_frames = 8000 * 30
    return w


def _run_vad_scenario(name, nch, npkts, probs_fn, seed):
    """Feed npkts 160-byte packets per channel, round-robin; whenever >=1 window is
    queued run process_batch on everything queued (as the worker thread would)."""
    from Core.VAD.SileroVAD import VADChannel
    from Core.Codecs.G711 import G711Codec
    import contextlib, io
    w = _mk_vad_worker()
    codec = G711Codec()
    rng = np.random.default_rng(seed)
    events = []
    chans = []
    for ci in range(nch):
        def a_in(chunk, active, ci=ci):
            events.append(['raw', ci, bool(active), int(chunk.audio.size(0)), sha(chunk.audio.numpy())[:16]])

        def v_in(chunk, ci=ci):
            events.append(['vad', ci, int(chunk.ipos), int(chunk.audio.size(0)), int(chunk.samplerate),
                           sha(chunk.audio.numpy())[:16]])
        chans.append(VADChannel(a_in, v_in, None, 'cpu'))
    pkts = rng.integers(0, 256, (nch, npkts, 160), dtype=np.uint8)
    win_ctr = [0] * nch
    all_probs = []
    for pi in range(npkts):
        for ci, ch in enumerate(chans):
            ch.ingest(w, pkts[ci, pi].tobytes(), codec)
        wis = []
        while not w.inf_queue.empty():
            wis.append(w.inf_queue.get_nowait())
        if wis:
            order = [chans.index(wi[0]) for wi in wis]
            pr = []
            for ci in order:
                pr.append(float(probs_fn(ci, win_ctr[ci])))
                win_ctr[ci] += 1
            # reference de-dups channels per sub-batch; with one window per channel per
            # packet round there is exactly one sub-batch
            assert len(set(order)) == len(order)
            w.model.script.append(pr)
            all_probs.append([order, pr])
            try:
                with contextlib.redirect_stdout(io.StringIO()):
                    w.process_batch(wis)
            except AssertionError:
                events.append(['assert', pi])
                break
    final = [{'triggered': bool(c.state.triggered), 'temp_end': int(c.state.temp_end),
              'current_sample': int(c.state.current_sample),
              'active_start': None if c.active_start is None else int(c.active_start),
              'buf_len': int(c.active_buffer.size(0)), 'fifo_len': len(c.vad_buffer)} for c in chans]
    return {'name': name, 'nch': nch, 'npkts': npkts, 'seed': seed, 'probs': all_probs,
            'events': events, 'final': final}


def gen_vad():
    sc = []
    base = [.1, .1, .1, .9, .9, .9, .2, .2, .2, .1, .1, .1]
    sc.append(_run_vad_scenario('survey_b2_worked_example', 1, 60, lambda c, i: base[i], 0))
    sc.append(_run_vad_scenario('speech_from_first_window', 1, 40,
                                lambda c, i: [.9, .9, .1, .1, .1, .6, .1, .1][i % 8], 1))
    sc.append(_run_vad_scenario('speech_from_second_window', 1, 40,
                                lambda c, i: [.1, .9, .9, .1, .1, .1, .6, .1, .1][i % 9], 6))
    rngp = np.random.default_rng(7)
    tab = rngp.random((5, 400))
    tab[:, 0] = 0.1            # speech in the very first window trips the reference's assert (:89)
    sc.append(_run_vad_scenario('five_channels_random_probs', 5, 300, lambda c, i: tab[c, i], 2))
    # blip shorter than min-silence, hysteresis band (0.35..0.5) holds state
    hy = [.1, .6, .4, .4, .45, .3, .6, .2, .2, .2, .2, .7, .36, .34, .34, .9]
    sc.append(_run_vad_scenario('hysteresis_band', 1, 5 * len(hy), lambda c, i: hy[i % len(hy)], 3))
    # 30 s cap: continuous speech for 330 windows (>240000 samples), then silence
    sc.append(_run_vad_scenario('thirty_second_cap', 1, 1800,
                                lambda c, i: .95 if 2 <= i < 335 else .05, 4))
    # cap while temp_end pending
    sc.append(_run_vad_scenario('cap_with_temp_end', 1, 1700,
                                lambda c, i: .05 if (i == 0 or i >= 330) else (.1 if i == 312 else .95), 5))
    dump_json('vad_traces.json', {'source': 'Core/VAD/SileroVAD.py:27-112 + SileroVADUtils.py:74-133 with a '
                                  'scripted-probability model object', 'scenarios': sc})


# --------------------------------------------------------------------------------------

class _StatefulFakeSilero:
    """A fake *stateful* model with the attributes VADIteratorB injects and saves (SileroVADUtils.py:99,131):
    h += [first sample of the window > 0], c += 1 (exact in f32), p = 0.9 where (h + c) mod 5 >= 3 else 0.1.  A wrong
    gather/scatter of the per-channel state changes the probabilities and therefore the events.  The same class is
    restated in tests/test_dsp_gpu.py (on device tensors)."""
    def __init__(self):
        import types
        self._c = types.SimpleNamespace(_h=None, _c=None, _last_sr=0, _last_batch_size=0)

    def reset_states(self):
        pass

    def __call__(self, x, sr):
        mc = self._c
        assert mc._last_batch_size == x.size(0) and mc._last_sr == sr and tuple(mc._h.shape) == (2, x.size(0), 64)
        e = (x[:, 0] > 0).to(torch.float32)
        mc._h = mc._h + e[None, :, None]
        mc._c = mc._c + 1.0
        k = torch.remainder(mc._h[0, :, 0] + mc._c[1, :, 63], 5.0)
        return torch.where(k >= 3.0, torch.tensor(0.9), torch.tensor(0.1))


def stateful_keep(ci, pi):
    """which packets a channel receives: channel-dependent gaps, so sub-batch membership and order keep changing"""
    return (pi * 7 + ci * 3) % (ci + 3) != 0


def gen_vad_stateful():
    from Core.VAD.SileroVAD import VADChannel
    from Core.VAD.SileroVADUtils import VADIteratorB
    from Core.Codecs.G711 import G711Codec
    import contextlib, io
    nch, npkts, seed = 6, 420, 11
    w = _mk_vad_worker()
    w.model = _StatefulFakeSilero()
    w.vad_iterator = VADIteratorB(w.model, sampling_rate=8000)
    codec = G711Codec()
    rng = np.random.default_rng(seed)
    events, chans, orders = [], [], []
    for ci in range(nch):
        def a_in(chunk, active, ci=ci):
            events.append(['raw', ci, bool(active), int(chunk.audio.size(0)), sha(chunk.audio.numpy())[:16]])

        def v_in(chunk, ci=ci):
            events.append(['vad', ci, int(chunk.ipos), int(chunk.audio.size(0)), int(chunk.samplerate),
                           sha(chunk.audio.numpy())[:16]])
        chans.append(VADChannel(a_in, v_in, None, 'cpu'))
    pkts = rng.integers(0, 256, (nch, npkts, 160), dtype=np.uint8)
    for pi in range(npkts):
        for ci in (range(nch) if pi % 2 == 0 else reversed(range(nch))):      # arrival order alternates too
            if stateful_keep(ci, pi):
                chans[ci].ingest(w, pkts[ci, pi].tobytes(), codec)
        if pi % 13 not in (2, 5, 12):     # let windows pile up: a channel may then be queued twice -> two sub-batches
            continue
        wis = []
        while not w.inf_queue.empty():
            wis.append(w.inf_queue.get_nowait())
        if wis:
            orders.append([chans.index(wi[0]) for wi in wis])
            with contextlib.redirect_stdout(io.StringIO()):
                w.process_batch(wis)
    final = [{'triggered': bool(c.state.triggered), 'temp_end': int(c.state.temp_end),
              'current_sample': int(c.state.current_sample),
              'active_start': None if c.active_start is None else int(c.active_start),
              'buf_len': int(c.active_buffer.size(0)), 'fifo_len': len(c.vad_buffer),
              'h': float(c.state.model_state[0][0, 0]), 'c': float(c.state.model_state[1][1, 63])} for c in chans]
    assert any(len(set(o)) < len(o) for o in orders), 'scenario must queue a channel twice in one batch'
    assert len({tuple(sorted(set(o))) for o in orders}) > 3, 'scenario must vary the sub-batch membership'
    assert any(e[0] == 'vad' for e in events)
    dump_json('vad_stateful_trace.json', {
        'source': 'Core/VAD/SileroVAD.py:27-112 + SileroVADUtils.py:21-26,99,131 with a fake STATEFUL model object '
                  '(tools/gen_golden.py:_StatefulFakeSilero)', 'nch': nch, 'npkts': npkts, 'seed': seed,
        'orders': orders, 'events': events, 'final': final})


def gen_stt():
    from Cluster.STTSession import STTSession, STTRequest, STTSentinel
    from Core.AudioChunk import VadAudioChunk, AudioChunk

    class FakeSTT:
        max_chunk_duration = 32.0
        sample_rate = 8000          # avoid the (absent) torchaudio resampler

        def __init__(self):
            self.calls = []

        def infer(self, wi):
            self.calls.append(wi)

    def run(script):
        stt = FakeSTT()
        sess = STTSession(stt, keep_context=False)
        log = []

        def mk_cb(tag):
            def cb(result):
                if isinstance(result, STTSentinel):
                    log.append(['sentinel', tag, result.signal])
                else:
                    log.append(['result', tag, result])
            return cb
        for op in script:
            if op[0] == 'vad':
                _, tag, ipos, n = op
                ch = VadAudioChunk(torch.full((n,), float(tag)), 8000, ipos)
                sess.soundin(STTRequest(ch, mk_cb(tag), 'en'))
            elif op[0] == 'plain':
                _, tag, n = op
                sess.soundin(STTRequest(AudioChunk(torch.full((n,), float(tag)), 8000), mk_cb(tag), 'en'))
            elif op[0] == 'sentinel':
                sess.soundin(STTSentinel(op[1], mk_cb('s' + op[1])))
            elif op[0] == 'complete':
                if not stt.calls:
                    log.append(['nothing_to_complete'])
                    continue
                req, text_cb, ctx = stt.calls.pop(0)
                a = req.chunk.audio
                vals = np.asarray(a)
                # run-length summary of the merged audio (tags / zero gaps)
                rl = []
                for v in vals.tolist():
                    if rl and rl[-1][0] == v:
                        rl[-1][1] += 1
                    else:
                        rl.append([v, 1])
                log.append(['submitted', len(vals), rl, type(a).__name__])
                text_cb(result='R%d' % len(vals))
            log.append(['state', bool(sess.busy), len(sess.pending), len(stt.calls)])
        return log

    scripts = {
        'single': [('vad', 1, 0, 800), ('complete',)],
        'merge_while_busy': [('vad', 1, 0, 800), ('vad', 2, 1000, 400), ('vad', 3, 1400, 800),
                             ('complete',), ('complete',)],
        'no_merge_over_32s': [('vad', 1, 0, 8000), ('vad', 2, 8000 * 20, 8000), ('vad', 3, 8000 * 31 + 1, 8000),
                              ('complete',), ('complete',), ('complete',)],
        'sentinel_alone': [('sentinel', 'flush')],
        'sentinel_behind_request': [('vad', 1, 0, 800), ('sentinel', 'a'), ('complete',)],
        'sentinel_shadowed': [('vad', 1, 0, 800), ('sentinel', 'a'), ('sentinel', 'b'), ('vad', 2, 5000, 800),
                              ('complete',), ('complete',)],
        'plain_chunk_not_merged': [('vad', 1, 0, 800), ('plain', 2, 800), ('vad', 3, 3000, 800),
                                   ('complete',), ('complete',), ('complete',)],
        'merge_skips_sentinel': [('vad', 1, 0, 800), ('vad', 2, 900, 100), ('sentinel', 'x'), ('vad', 3, 2000, 100),
                                 ('complete',), ('complete',)],
    }
    out = {k: run(v) for k, v in scripts.items()}
    dump_json('stt_session_traces.json', {'source': 'Cluster/STTSession.py:63-113 with a recording worker; '
                                          'sample_rate=8000 so no resample', 'scripts': {k: [list(o) for o in v] for k, v in scripts.items()},
                                          'logs': out})


# --------------------------------------------------------------------------------------
def gen_batched():
    from Cluster.InfernBatchedWorker import InfernBatchedWorker

    class W(InfernBatchedWorker):
        max_batch_size = 4

        def process_batch(self, wis):
            pass
    w = W()
    for i in range(10):
        w.infer(i)
    b1 = w.next_batch(); b2 = w.next_batch(); b3 = w.next_batch()
    w.infer(7); w.infer(None); w.infer(8)
    b4 = w.next_batch()
    w2 = W(); w2.infer(None)
    b5 = w2.next_batch()
    dump_json('batched_worker.json', {'source': 'Cluster/InfernBatchedWorker.py:17-28', 'max_batch_size': 4,
                                      'batches': [b1, b2, b3, b4, b5]})


# --------------------------------------------------------------------------------------
def gen_muxer():
    from Core.OutputMuxer import OutputMTMuxer
    from Core.AudioChunk import AudioChunk
    from Core.AStreamMarkers import ASMarkerNewSent
    import contextlib, io
    rng = np.random.default_rng(11)
    log = []

    class Mk(ASMarkerNewSent):
        def __init__(self, tag, **kw):
            super().__init__(**kw); self.tag = tag

        def on_proc(self, w, *a):
            log.append(['marker', self.tag])
    mux = OutputMTMuxer(8000, 800, 'cpu')
    script = [('chunk', 0, 500), ('idle',), ('chunk', 0, 500), ('idle',), ('idle',), ('marker', 0, 'm1'),
              ('chunk', 0, 1000), ('chunk', 1, 300), ('idle',), ('chunk', 1, 900), ('idle',), ('idle',), ('idle',),
              ('marker', 1, 'm2'), ('idle',), ('idle',)]
    data = {}
    for i, op in enumerate(script):
        if op[0] == 'chunk':
            a = rng.standard_normal(op[2]).astype(np.float32)
            data['in_%d' % i] = a
            c = AudioChunk(torch.from_numpy(a.copy()), 8000); c.track_id = op[1]
            mux.chunk_in(c)
        elif op[0] == 'marker':
            mux.chunk_in(Mk(op[2], track_id=op[1]))
        else:
            with contextlib.redirect_stdout(io.StringIO()):
                r = mux.idle(None)
            if r is None:
                log.append(['idle', i, None])
            else:
                data['out_%d' % i] = r.numpy().copy()
                log.append(['idle', i, int(r.size(0))])
    np.savez_compressed(os.path.join(GOLD, 'muxer_data.npz'), **data)
    dump_json('muxer_trace.json', {'source': 'Core/OutputMuxer.py:10-85', 'script': [list(s) for s in script], 'log': log})


# --------------------------------------------------------------------------------------
def synth_utterance(seed, seconds=10.0, sr=8000):
    """SURVEY.md 8(d) synthetic utterance (also implemented in infernos_amd.synth)."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    f0 = rng.uniform(100, 300)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * t)
    env[(t < 1.0) | (t > seconds - 1.0)] = 0.0
    x = 0.3 * env * sum(a * np.sin(2 * np.pi * k * f0 * t) for k, a in ((1, 1.0), (2, 0.5), (3, 0.25)))
    x = x + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)


def gen_logmel():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()          # what WhisperProcessor wraps (InfernSTTWorker.py:52,114)
    out = {}
    meta = {'source': 'transformers 5.15.0 WhisperFeatureExtractor() as called at Cluster/InfernSTTWorker.py:114 '
                      '(return_tensors=np, sampling_rate=16000)', 'cases': {}}
    from oracle import dsp
    for seed, secs in ((1000, 10.0), (1001, 3.3), (1002, 30.0), (1003, 0.5)):
        x8 = synth_utterance(seed, max(secs, 2.5))[: int(secs * 8000)]
        x16 = dsp.resample(x8, 8000, 16000)
        ref = fe([x16], sampling_rate=16000, return_tensors='np').input_features[0]
        out['frames_%d' % seed] = ref[:, ::37].copy()
        meta['cases'][str(seed)] = {'seconds': secs, 'n16': int(x16.size), 'sum': float(ref.astype(np.float64).sum()),
                                    'sumsq': float((ref.astype(np.float64) ** 2).sum()),
                                    'min': float(ref.min()), 'max': float(ref.max()), 'audio_sha': sha(x16)}
    # batch of two: per-utterance max (feature_extraction_whisper.py:159-161)
    meta['mel_filters_sha256_f64'] = sha(fe.mel_filters)
    np.savez_compressed(os.path.join(GOLD, 'logmel.npz'), **out)
    dump_json('logmel_meta.json', meta)


def gen_t2t():
    """Core/T2T/Translator.py and NumbersToWords.py run in place over scripted stand-ins for the packages they wrap
    (argostranslate's package index / installed languages, inflect's number_to_words): what is pinned is the reference's own
    logic -- pair search and pivot order, chaining, the number regex, suffix rules, replacement order, translation cache."""
    import types
    import argostranslate.package as apkg
    import argostranslate.translate as atr
    out = {'translator': [], 'numbers': []}

    def world(pairs):
        log = []

        class Pkg:
            def __init__(self, a, b):
                self.from_code, self.to_code = a, b

            def download(self):
                return '%s_%s.argosmodel' % (self.from_code, self.to_code)

        class Lang:
            def __init__(self, code):
                self.code = code

            def get_translation(self, to):
                frm = self.code
                return types.SimpleNamespace(translate=lambda s, frm=frm, to=to.code: '[%s>%s]%s' % (frm, to, s))
        apkg.update_package_index = lambda: log.append('update')
        apkg.get_available_packages = lambda: [Pkg(a, b) for a, b in pairs]
        apkg.install_from_path = lambda path: log.append('install ' + path)
        atr.get_installed_languages = lambda: [Lang(c) for c in ('en', 'it', 'de', 'ru', 'ja', 'pt')]
        return log
    import importlib
    import Core.T2T.Translator as T
    for name, pairs, frm, to, use_filter in (
            ('direct', [('en', 'ja'), ('ru', 'en')], 'en', 'ja', False),
            ('pivot through the last candidate', [('ru', 'ja'), ('ja', 'it'), ('ru', 'en'), ('en', 'it')], 'ru', 'it', False),
            ('pivot, later candidates missing', [('ru', 'en'), ('en', 'it')], 'ru', 'it', True),
            ('no path', [('en', 'de')], 'ru', 'it', False)):
        log = world(pairs)
        importlib.reload(T)
        rec = {'name': name, 'pairs': pairs, 'from': frm, 'to': to, 'filter': use_filter}
        flt = (lambda text, from_code, to_code, tr: '<%s-%s>' % (from_code, to_code) + tr(text)) if use_filter else None
        import contextlib, io
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                t = T.Translator(frm, to, filter=flt)
            rec['nstages'] = len(t.translators)
            rec['out'] = t.translate('hello')
        except StopIteration:
            rec['raises'] = 'StopIteration'
        rec['log'] = [l for l in log if l != 'update']
        out['translator'].append(rec)
    # NumbersToWords over a scripted number_to_words
    inflect = types.ModuleType('inflect')
    inflect.engine = lambda: types.SimpleNamespace(number_to_words=lambda s: 'N(%s)' % s)
    sys.modules['inflect'] = inflect
    import config.InfernGlobals as IGm
    calls = []
    IGm.InfernGlobals.get_translator = staticmethod(lambda a, b, **k: types.SimpleNamespace(
        translate=lambda s: (calls.append(s), '{%s:%s}' % (b, s))[1]))
    import Core.T2T.NumbersToWords as N
    importlib.reload(N)
    texts = ['I have 3 cats and 2 dogs.', 'I have 3% cats and 2% dogs.', 'I have 30000 cats and 2999 dogs.',
             'I have 50% cats and 29.0% dogs.', 'I have 3,090.6 cats and 21,188,128 dogs.%,', 'I have 3% cats and dogs 2%.',
             'I have 3% cats and dogs 20%, and mice 3.0%.', 'I have 3% cats and dogs since 2024, or 2023.', 'No numbers here!',
             '7 7 7, and 7!', 'x1 2x 3.', '12,5 then 1.5.']
    for lang in ('en', 'de'):
        n2w = N.NumbersToWords(lang)
        del calls[:]
        res = [n2w(t) for t in texts]
        out['numbers'].append({'lang': lang, 'texts': texts, 'out': res, 'translated': list(calls)})
    out['source'] = 'Core/T2T/Translator.py:19-57, Core/T2T/NumbersToWords.py:7-35 run in place over scripted package stand-ins'
    json.dump(out, open(os.path.join(GOLD, 't2t.json'), 'w'), indent=0, sort_keys=True)
    print('wrote t2t.json:', [r.get('out', r.get('raises')) for r in out['translator']])


SECTIONS = {'t2t': gen_t2t, 'g711': gen_g711, 'vad': gen_vad, 'vad_stateful': gen_vad_stateful, 'stt': gen_stt, 'batched': gen_batched, 'muxer': gen_muxer,
            'logmel': gen_logmel}

if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    try:
        import gen_golden_nn
        SECTIONS.update(gen_golden_nn.SECTIONS)
    except ImportError:
        pass
    want = sys.argv[1:] or list(SECTIONS)
    for s in want:
        SECTIONS[s]()
