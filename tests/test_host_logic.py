"""CPU: host-side logic of the plugin interface against traces captured from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from infernos_amd.audio import AudioChunk, VadAudioChunk
from infernos_amd.stt import STTRequest, STTSentinel, STTSession
from infernos_amd.workers import InfernBatchedWorker


def test_batched_worker_batches(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'batched_worker.json')))

    class W(InfernBatchedWorker):
        max_batch_size = g['max_batch_size']

        def process_batch(self, wis):
            pass
    w = W()
    for i in range(10):
        w.infer(i)
    got = [w.next_batch(), w.next_batch(), w.next_batch()]
    w.infer(7); w.infer(None); w.infer(8)
    got.append(w.next_batch())
    w2 = W(); w2.infer(None)
    got.append(w2.next_batch())
    assert got == g['batches']


def test_batched_worker_thread_lifecycle():
    seen = []

    class W(InfernBatchedWorker):
        max_batch_size = 3

        def process_batch(self, wis):
            seen.append(list(wis))

    class Item:
        def __init__(self):
            self.started = 0

        def _proc_start_cb(self):
            self.started += 1
    w = W()
    with pytest.raises(AssertionError):
        w.stop()                      # not running yet (InfernWrkThread.py:63-64)
        w.inf_queue.get_nowait()
    w = W()
    w.start()
    items = [Item() for _ in range(7)]
    for it in items:
        w.infer(it)
    import time
    t0 = time.time()
    while sum(len(b) for b in seen) < 7 and time.time() - t0 < 5:
        time.sleep(0.01)          # stop() does not drain the queue (InfernBatchedWorker.py:33-45)
    w.stop()
    assert sum(len(b) for b in seen) == 7 and all(len(b) <= 3 for b in seen)
    assert all(it.started == 1 for it in items)
    assert not w.is_alive()


def test_vad_audio_chunk_append():
    a = VadAudioChunk(torch.ones(10), 8000, 100)
    b = VadAudioChunk(torch.full((5,), 2.0), 8000, 115)
    a.append(b)
    assert a.audio.tolist() == [1.0] * 10 + [0.0] * 5 + [2.0] * 5
    assert a.tpos() == 100 / 8000 and a.duration() == 20 / 8000
    with pytest.raises(AssertionError):
        a.append(VadAudioChunk(torch.ones(3), 8000, 0))
    with pytest.raises(AssertionError):
        a.append(VadAudioChunk(torch.ones(3), 16000, 1000))


class _RecordingSTT:
    max_chunk_duration = 32.0
    sample_rate = 8000
    wants_numpy = True

    def __init__(self):
        self.calls = []

    def infer(self, wi):
        self.calls.append(wi)


def _run_script(script):
    stt = _RecordingSTT()
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
            sess.soundin(STTRequest(VadAudioChunk(torch.full((n,), float(tag)), 8000, ipos), mk_cb(tag), 'en'))
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
            vals = np.asarray(req.chunk.audio)
            rl = []
            for v in vals.tolist():
                if rl and rl[-1][0] == v:
                    rl[-1][1] += 1
                else:
                    rl.append([v, 1])
            log.append(['submitted', len(vals), rl, type(req.chunk.audio).__name__])
            text_cb(result='R%d' % len(vals))
        log.append(['state', bool(sess.busy), len(sess.pending), len(stt.calls)])
    return log


def test_stt_session_traces(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'stt_session_traces.json')))
    for name, script in g['scripts'].items():
        assert _run_script([tuple(o) for o in script]) == g['logs'][name], name


def test_stt_session_stop_drops_late_results():
    stt = _RecordingSTT()
    sess = STTSession(stt, False)
    got = []
    sess.soundin(STTRequest(AudioChunk(torch.zeros(8), 8000), lambda result: got.append(result), 'en'))
    req, cb, ctx = stt.calls.pop()
    sess.stop()
    cb(result='late')
    assert got == []
