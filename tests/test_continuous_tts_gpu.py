"""GPU: continuous (ragged-position) batching of the TTS decode loop (infernos_amd.tts.ContinuousTTS over
engines/speecht5.py:TTSRaggedState) against the per-batch schedule the reference runs
(Cluster/InfernTTSWorker.py:83-92: freeze a batch, loop HelloSippyRTPipe.infer / unbatch_and_dispatch to the end).
Rows of one step sit at different decoder positions; every row must receive exactly the audio -- byte for byte, dispatch
by dispatch -- it receives when its batch runs alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pipe(dev, stop_bias=None, output_sr=16000, seed=0):
    from infernos_amd.tts import HelloSippyRTPipe
    from infernos_amd.weights import synth_state_dict
    kw = {} if stop_bias is None else {'stop_bias': stop_bias}
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', seed, **kw), 'hifigan': synth_state_dict('hifigan', seed),
         'amendment': synth_state_dict('amendment', seed)}
    pp = HelloSippyRTPipe(dev, weights=W, processor=lambda **k: None, speaker_embeddings=[], output_sr=output_sr)
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(dev)
    pp.mask_source = lambda n: fixed           # the same keep-masks for every infer() call: schedules become comparable
    return pp


class _Sink:
    def __init__(self, n):
        self.rows = [[] for _ in range(n)]

    def cb(self, i):
        return lambda chunk: self.rows[i].append(None if chunk is None else chunk.clone())


def _per_batch(pp, ids, spk, max_calls=None):
    """the reference schedule: this batch alone, infer() / unbatch_and_dispatch() until every row has ended"""
    from infernos_amd.pipeline import _make_state
    state = _make_state(pp, ids, spk)
    sink = _Sink(ids.size(0))
    state.dispatch = [sink.cb(i) for i in range(ids.size(0))]
    calls = 0
    while True:
        pp.infer(state)
        calls += 1
        if not pp.unbatch_and_dispatch(state) or (max_calls is not None and calls >= max_calls):
            break
    return sink.rows, calls


def _same(a, b):
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if (x is None) != (y is None):
            return False
        if x is not None and not torch.equal(x.view(torch.int16), y.view(torch.int16)):
            return False
    return True


def test_ragged_rows_get_the_audio_of_their_own_batch(built_lib):
    """Five batches of different sizes and text lengths (hence different maxlen: they end after 2..5 infer() calls through
    the stop rule's maxlen arm) join a running ContinuousTTS at different infer() boundaries, some into slots that a finished
    batch has just left.  Every dispatch of every row -- tensors and the closing None -- equals the per-batch schedule."""
    from infernos_amd import _lib
    from infernos_amd.tts import ContinuousTTS
    dev = _lib.require_device('cuda:0')
    pp = _pipe(dev, stop_bias=-20.0)
    g = torch.Generator().manual_seed(9)
    batches = []
    for n, T in ((3, 3), (2, 5), (4, 2), (1, 7), (3, 4)):           # maxlen = 10 T decoder steps
        ids = torch.randint(4, 80, (n, T), generator=g, dtype=torch.int32)
        batches.append((ids, torch.randn(n, 512, generator=g)))
    ref = [_per_batch(pp, ids, spk) for ids, spk in batches]
    assert sorted(c for _, c in ref) != [ref[0][1]] * 5, 'the batches should end after different numbers of calls'
    eng = ContinuousTTS(pp, max_rows=16, max_text=16, row_bucket=16)
    sinks, groups = [], []

    def submit(k):
        ids, spk = batches[k]
        s = _Sink(ids.size(0))
        sinks.append(s)
        groups.append(eng.submit(ids, torch.full((ids.size(0),), ids.size(1), dtype=torch.int32), spk,
                                 dispatch=[s.cb(i) for i in range(ids.size(0))]))
    plan = {0: [0], 1: [1, 2], 3: [3], 4: [4]}                       # engine call -> batches that join in front of it
    call = 0
    while call < 40:
        for k in plan.get(call, []):
            submit(k)
        alive = eng.step()
        call += 1
        if not alive and call > max(plan):
            break
    assert all(gr.done.is_set() for gr in groups) and len(groups) == 5
    pos_seen = set()
    for k, (rows, calls) in enumerate(ref):
        assert groups[k].calls == calls, (k, groups[k].calls, calls)
        for i in range(len(rows)):
            assert _same(sinks[k].rows[i], rows[i]), (k, i, [None if c is None else c.numel() for c in sinks[k].rows[i]],
                                                      [None if c is None else c.numel() for c in rows[i]])
        pos_seen.add(tuple(groups[k].slots))
    assert eng.calls_run < sum(c for _, c in ref), 'the batches were supposed to overlap in the engine'
    assert sorted(eng.free) == list(range(16))


@pytest.mark.parametrize('lookahead', [False, True])
def test_ragged_natural_stop_and_slot_reuse(built_lib, lookahead):
    """Seeded weights with the stop head live: rows of one batch end at different steps (sigmoid arm of the rule); a
    second batch takes over the slots of the first.  Same bytes as each batch alone -- also with the engine queueing call c + 1
    before it takes call c's results (`lookahead`: a batch that ends by the stop rule is then computed for one call more than it
    needed; nothing changes for any row)."""
    from infernos_amd import _lib
    from infernos_amd.tts import ContinuousTTS
    dev = _lib.require_device('cuda:0')
    pp = _pipe(dev, stop_bias=-1.5, seed=3)
    g = torch.Generator().manual_seed(21)
    batches = []
    for n, T in ((6, 6), (5, 9)):
        batches.append((torch.randint(4, 80, (n, T), generator=g, dtype=torch.int32), torch.randn(n, 512, generator=g)))
    ref = [_per_batch(pp, ids, spk) for ids, spk in batches]
    eng = ContinuousTTS(pp, max_rows=16, max_text=16, row_bucket=16)
    eng.lookahead = lookahead
    out = []
    for k, (ids, spk) in enumerate(batches):
        s = _Sink(ids.size(0))
        gr = eng.submit(ids, torch.full((ids.size(0),), ids.size(1), dtype=torch.int32), spk,
                        dispatch=[s.cb(i) for i in range(ids.size(0))])
        eng.step()
        out.append((s, gr))
    while eng.step():
        pass
    lens = []
    for k, (rows, calls) in enumerate(ref):
        s, gr = out[k]
        assert gr.done.is_set() and gr.calls == calls
        for i in range(len(rows)):
            assert _same(s.rows[i], rows[i]), (k, i)
            lens.append(sum(c.numel() for c in rows[i] if c is not None))
    print('natural-stop utterance lengths (samples):', lens)


@pytest.mark.parametrize('output_sr', [8000])
def test_continuous_engine_at_bench_rows_matches_lane_schedule(built_lib, output_sr):
    """The size bench.py runs (configuration 3): 128-row utterance batches, four of them in flight in ONE ragged decode
    batch (512-640 row steps, joined at different infer() boundaries), 8 kHz mu-law output as SpeechPipeline asks for.
    Rows 0-3 and 124-127 of every batch against the same utterances run as a per-batch schedule of their own
    (HelloSippyRTPipe decode_chunk + render, 8-row batch): byte-identical mu-law -- the K split of the decode GEMMs does not depend on the
    row count, every other kernel is row-local."""
    from infernos_amd import _lib
    from infernos_amd.tts import ContinuousTTS
    dev = _lib.require_device('cuda:0')
    pp = _pipe(dev, stop_bias=-20.0, output_sr=output_sr)
    N, T, CALLS = 128, 64, 4
    g = torch.Generator().manual_seed(1)
    eng = ContinuousTTS(pp, max_rows=640, max_text=T, row_bucket=128).start()
    try:
        batches = [(torch.randint(4, 80, (N, T), generator=g, dtype=torch.int32), torch.randn(N, 512, generator=g)) for _ in range(6)]
        lens = torch.full((N,), T, dtype=torch.int32)
        groups = []
        import time
        for k, (ids, spk) in enumerate(batches):
            groups.append(eng.submit(ids, lens, spk, max_calls=CALLS, want_ulaw=True))
            time.sleep(0.03 if k % 2 else 0.0)              # some join together, some a call or two later
        for gr in groups:
            gr.result(timeout=300)
    finally:
        eng.stop()
    assert eng.rows_run / eng.calls_run > 128, 'batches were supposed to share engine calls'
    rows = [0, 1, 2, 3, 124, 125, 126, 127]
    A = 8192 // (16000 // output_sr)
    for k, (ids, spk) in enumerate(batches):
        gr = groups[k]
        torch.cuda.current_stream(dev).wait_event(gr.done_event)
        assert gr.ulaw.shape == (N, CALLS * A) and gr.valid.tolist() == [CALLS * A - 512 // (16000 // output_sr)] * N
        from infernos_amd.pipeline import _make_state
        state = _make_state(pp, ids[rows], spk[rows])
        for c in range(CALLS):                               # as SpeechPipeline.synthesize encodes a lane's batch
            par = pp.decode_chunk(state)
            pcm = pp.resampler(pp.render(state.dev, par, use_graphs=True).float())
            ul = torch.empty(pcm.shape, dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().ifh_g711_encode_f32_u8(_lib.ptr(pcm.contiguous()), _lib.ptr(ul), pcm.numel(), _lib.stream_ptr(dev)), 'enc')
            assert torch.equal(ul, gr.ulaw[rows, c * A:(c + 1) * A]), (k, c)
