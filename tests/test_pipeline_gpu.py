"""GPU: the batched serving loop (infernos_amd.pipeline) end to end on a few calls, checked
against the oracle stage by stage on the same intermediate data."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import isolated  # noqa: E402
from oracle import dsp as odsp, nn as onn  # noqa: E402


def test_pipeline_cycle_small(built_lib):
    from infernos_amd import _lib
    from infernos_amd.codecs import G711Codec
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    N = 3
    pipe = SpeechPipeline(N, dev, n_infer=4, n_new_tokens=6)
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, device=dev)
    pipe.tts.mask_source = lambda n: fixed
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
    chunks = pipe.ingest(frames)
    pcm = odsp.g711_decode(ulaw)
    for i in range(N):
        assert len(chunks[i]) >= 1
        tot = sum(c.audio.numel() for c in chunks[i])
        assert 6.5 * 8000 < tot < 9.5 * 8000, tot
        for c in chunks[i]:
            # chunk audio is the decoded stream at [ipos, ipos+len) -- except its first 240 samples (the 30 ms
            # start pad), which the reference takes from the stale head of its idle buffer (SileroVAD.py:90,101-102)
            a = c.audio.cpu().numpy()
            assert np.array_equal(a[240:], pcm[i, c.ipos + 240:c.ipos + a.size])
    spans = [[(c.ipos, c.audio.numel()) for c in lst] for lst in chunks]
    pipe_merged = []
    for lst in chunks:                           # what pipe.stt merges (VadAudioChunk.append semantics)
        lo = lst[0].ipos
        m = np.zeros(lst[-1].ipos + lst[-1].audio.numel() - lo, np.float32)
        for c in lst:
            m[c.ipos - lo:c.ipos - lo + c.audio.numel()] = c.audio.cpu().numpy()
        pipe_merged.append(m)
    toks, nsp, secs = pipe.stt(chunks)
    assert toks.shape == (N, 6) and nsp.shape == (N,)
    # oracle on the same merged audio
    sd = synth_state_dict('whisper_tiny', 0)
    for i in range(N):
        lo = spans[i][0][0]
        hi = spans[i][-1][0] + spans[i][-1][1]
        merged = np.zeros(hi - lo, np.float32)
        for (p, n) in spans[i]:
            merged[p - lo:p - lo + n] = pcm[i, p:p + n]
        merged = pipe_merged[i]
        assert abs(float(secs[i]) - merged.size / 8000.0) < 1e-6
        mel = torch.from_numpy(odsp.logmel(odsp.resample(merged, 8000, 16000)))[None]
        with torch.no_grad():
            o_toks, o_first, _, _ = onn.whisper_greedy(sd, mel, pipe.prompt[:1].long(), 6, 6)
        margin = float(o_first.topk(2).values[0, 0] - o_first.topk(2).values[0, 1])
        if margin > 0.05:
            assert int(toks[i, 0]) == int(o_toks[0, 0]), (i, toks[i].tolist(), o_toks.tolist())
    ul, valid, sp = pipe.synthesize()
    assert ul.shape == (N, 4 * 4096) and ul.dtype == torch.uint8
    assert valid.tolist() == [4 * 4096 - 256] * N          # the first call drops its first 256 samples @8 kHz
    # second and third utterances replay the captured hipGraphs (decoder steps + render): same bytes
    ul2, _, _ = pipe.synthesize()
    ul3, _, _ = pipe.synthesize()
    assert torch.equal(ul2, ul) and torch.equal(ul3, ul)
    audio = odsp.g711_decode(ul.cpu().numpy())
    assert np.isfinite(audio).all() and 1e-4 < np.abs(audio).mean() < 0.5


@pytest.mark.parametrize('lanes,group,fronts', [(2, 1, 1), (3, 2, 2)])      # (round 6: (2, 2, 1) and (1, 3, 1) ran the same code paths)
def test_pipelined_tts_lanes_match_sequential(built_lib, lanes, group, fronts):
    """run_steps with the front-end thread, overlapping TTS lanes and grouped TTS batches (the utterances of
    `group` consecutive cycles synthesised as one batch) returns, cycle by cycle and in order, the bytes of the
    strictly sequential schedule."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 5       # 5 x 64 text rows: no GEMM of the grouped batch crosses a kernel-selection threshold (M <= 256 -> skinny)
    pipe = SpeechPipeline(N, dev, n_infer=3, n_new_tokens=4, tts_lanes=lanes, tts_group=group, front_lanes=fronts)
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, device=dev)
    for lane in pipe.tts_lanes:
        lane.mask_source = lambda n: fixed
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = [torch.from_numpy(np.ascontiguousarray(np.roll(ulaw, k, axis=0).reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
              for k in range(2)]
    ref = []
    for k in range(2):
        r = pipe.run_steps(lambda _k, k=k: frames[k], 1, pipelined=False)
        ref.append((r['ulaw'].clone(), r['tokens'].clone()))
    pipe.prime(frames[0])
    got = []
    pipe.run_steps(lambda k: frames[k % 2], 7, pipelined=True,
                   on_cycle=lambda r: got.append((r['ulaw'].clone(), r['tokens'].clone())))
    torch.cuda.synchronize()
    assert len(got) == 7
    for k, (ul, tk) in enumerate(got):
        assert torch.equal(tk, ref[k % 2][1]), k
        assert torch.equal(ul, ref[k % 2][0]), k


@pytest.mark.parametrize('lanes,fronts,beam', [(4, 3, 5)])       # (round 6: (2, 1, 1) -- greedy, one front lane -- dropped for the suite's time limit)
@isolated
def test_continuous_tts_schedule_matches_sequential_lane_schedule(built_lib, lanes, fronts, beam):
    """The schedule bench.py times -- `fronts` ingest+STT lanes (5-beam search) feeding ONE continuous TTS decode batch
    that holds up to `lanes` utterance batches at different decoder positions -- returns, cycle by cycle and in order, the
    bytes of the strictly sequential per-batch schedule of a separate lane-mode pipeline (one frozen batch at a time, the
    reference's worker loop, Cluster/InfernTTSWorker.py:83-92)."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 5
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, device=dev)
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = [torch.from_numpy(np.ascontiguousarray(np.roll(ulaw, k, axis=0).reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
              for k in range(2)]
    seq = SpeechPipeline(N, dev, n_infer=3, n_new_tokens=4, tts_lanes=1, stt_beam=beam)
    seq.tts.mask_source = lambda n: fixed
    ref = []
    for k in range(2):
        r = seq.run_steps(lambda _k, k=k: frames[k], 1, pipelined=False)
        ref.append((r['ulaw'].clone(), r['tokens'].clone(), r['tts_samples'].clone()))
    del seq
    pipe = SpeechPipeline(N, dev, n_infer=3, n_new_tokens=4, tts_lanes=lanes, front_lanes=fronts, stt_beam=beam,
                          tts_mode='continuous')
    pipe.tts.mask_source = lambda n: fixed
    try:
        pipe.prime(frames[0])
        got = []
        if fronts > 1:
            # 5 calls make an engine call shorter than an STT cycle, so batches would only meet by chance: hold every engine call
            # back a little so that the three front lanes' batches pile up and share decode steps (which is what is under test)
            import time
            step0 = pipe.ctts.step
            pipe.ctts.step = lambda *a, **kw: (time.sleep(0.03), step0(*a, **kw))[1]
        pipe.run_steps(lambda k: frames[k % 2], 9, pipelined=True,
                       on_cycle=lambda r: got.append((r['ulaw'].clone(), r['tokens'].clone(), r['tts_samples'].clone())))
        torch.cuda.synchronize()
        assert len(got) == 9
        for k, (ul, tk, ns) in enumerate(got):
            assert torch.equal(tk, ref[k % 2][1]), k
            assert torch.equal(ns, ref[k % 2][2]), k
            assert torch.equal(ul, ref[k % 2][0]), k
        if fronts > 1:          # (one front lane feeds the engine one batch at a time)
            assert pipe.ctts.rows_run > pipe.ctts.calls_run * pipe.ctts.row_bucket, 'no two batches ever shared a step'
    finally:
        pipe.ctts.stop()


@pytest.mark.parametrize('vad_model', [None, 'recurrent'])
def test_block_ingest_equals_per_tick_ingest(built_lib, vad_model):
    """ifh_ingest_block / ifh_ingest_block_net (the tick loop driven from one host call) emit exactly the chunks of the per-tick path --
    with the energy rule and with the recurrent network (conv + 2 x LSTM(64), per-call state carried window to window:
    Core/VAD/SileroVAD.py:78-80, SileroVADUtils.py:99,131), whose state tables must also end up equal."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 4
    pipe = SpeechPipeline(N, dev, n_infer=1, n_new_tokens=2, tts_lanes=1, vad_model=vad_model)
    assert (pipe.vad.model is None) == (vad_model is None)
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
    res = []
    for block in (False, True):
        pipe.reset_calls()
        a = pipe.ingest(frames[:250], block=block)            # two calls: the FIFO / VAD state carries over
        b = pipe.ingest(frames[250:], block=block)
        res.append([[(c.ipos, c.audio.cpu().numpy()) for c in la + lb] for la, lb in zip(a, b)])
        st = (pipe.vad.st.cpu().numpy().copy(), pipe.vad.blen.cpu().numpy().copy(), pipe.calls.fifo_len.cpu().numpy().copy(),
              pipe.vad.mh.cpu().numpy().copy(), pipe.vad.mc.cpu().numpy().copy())
        res.append(st)
    tick, tick_st, blk, blk_st = res
    assert sum(len(l) for l in tick) >= N
    if vad_model is not None:
        assert np.abs(blk_st[3]).max() > 0              # the network's state moved
    for lt, lb in zip(tick, blk):
        assert [p for p, _ in lt] == [p for p, _ in lb]
        for (_, at), (_, ab) in zip(lt, lb):
            assert np.array_equal(at, ab)
    for u, v in zip(tick_st, blk_st):
        assert np.array_equal(u, v)


@isolated
def test_paced_ticks_beside_the_running_engines_keep_their_tail(built_lib):
    """The per-tick path as a caller of the library runs it -- H2D frame matrix -> CallTable.tick -> mux_encode ->
    frontend.TickEgress (D2H copy + the marker packet behind it) -> wait -- paced at 20 ms on a high-priority stream while a
    pipelined SpeechPipeline (front lanes + the continuous TTS engine) keeps the GPU busy.  Round 4 saw 1-2 ticks per 100 wait
    40-57 ms in their own hardware queue without the marker and 3.6-7.2 ms with it; the marker now lives in the product
    (TickEgress).  The measured figures (p50 0.15-0.4 ms, p99 0.9-3 ms on an otherwise idle box) go to gpurun_out/tick_tail.json;
    what the test asserts are bounds a loaded, shared box still meets -- p50 < 10 ms, p99 < 100 ms over 300 ticks -- and that the
    tick's bytes are right.  Runs in a child process (conftest.isolated)."""
    import json
    import os
    import threading
    import time
    from infernos_amd import _lib
    from infernos_amd.frontend import CallTable, TickEgress, mux_encode
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 64
    x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
    pipe = SpeechPipeline(N, dev, whisper_family='whisper_tiny', n_infer=10, n_new_tokens=16, tts_lanes=3, front_lanes=2,
                          stt_beam=5, tts_mode='continuous', vad_model='recurrent')
    lat = []
    try:
        pipe.prime(frames)
        stop = threading.Event()

        def load():
            torch.cuda.set_device(dev)
            while not stop.is_set():
                pipe.run_steps(lambda k: frames, 4, pipelined=True)
        th = threading.Thread(target=load, daemon=True)
        th.start()
        time.sleep(0.5)
        calls, egress = CallTable(N, dev), TickEgress(N, 160, dev)
        host_in = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).pin_memory()
        stream = torch.cuda.Stream(device=dev, priority=-1)
        slots = torch.arange(N, dtype=torch.int32, device=dev)
        dfr = torch.empty((N, 160), dtype=torch.uint8, device=dev)
        p8, p16 = torch.empty((N, 160), device=dev), torch.empty((N, 320), device=dev)
        present, ndiv = torch.ones((N, 1), dtype=torch.uint8, device=dev), torch.ones(N, dtype=torch.int32, device=dev)
        t0 = time.perf_counter()
        for t in range(300):
            due = t0 + t * 0.020
            now = time.perf_counter()
            if due > now:
                time.sleep(due - now)
            a = time.perf_counter()
            with torch.cuda.stream(stream):
                dfr.copy_(host_in[t % 500], non_blocking=True)
                calls.tick(dfr, slots, p8, p16, want_ready=False)
                enc, _ = mux_encode(p8[:, None, :], present, ndiv)
                egress.push(enc).wait()
            lat.append((time.perf_counter() - a) * 1e3)
        stop.set()
        th.join(120)
        assert np.array_equal(egress.host.numpy(), odsp.g711_encode(odsp.g711_decode(ulaw.reshape(N, 500, 160)[:, 299 % 500])))
    finally:
        pipe.close()
    lat = np.array(lat)
    p50, p99 = float(np.percentile(lat, 50)), float(np.percentile(lat, 99))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump({'ticks': len(lat), 'calls': N, 'p50_ms': p50, 'p99_ms': p99, 'worst_ms': float(lat.max())}, open('gpurun_out/tick_tail.json', 'w'))
    assert p50 < 10.0 and p99 < 100.0, (p50, p99, float(lat.max()))


def test_config4_front_end_256_calls_per_gpu(built_lib):
    """BASELINE config 4 per-GPU size (256 calls): the batched front end against the oracle call by call -- decoded
    PCM and the VAD chunk list of every call equal what the oracle's per-call path gives for that call's own stream
    (calls j and j+128 carry the same audio and must agree bit for bit wherever they sit in the 256-row table)."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 256
    pipe = SpeechPipeline(N, dev, n_infer=1, n_new_tokens=2, tts_lanes=1)
    base = np.stack([synth_utterance(1000 + i, 4.0) for i in range(8)])
    x = base[np.arange(N) % 8]
    x[128:] = x[:128]
    ulaw = odsp.g711_encode(x)
    nt = ulaw.shape[1] // 160
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, nt, 160).transpose(1, 0, 2))).to(dev)
    pipe.reset_calls()
    chunks = pipe.ingest(frames, block=True)
    got = [[(c.ipos, c.audio.cpu().numpy()) for c in lst] for lst in chunks]
    assert sum(len(g) for g in got) >= N // 2
    for j in range(N):
        k = j % 8
        assert [p for p, _ in got[j]] == [p for p, _ in got[k]]
        for (_, a), (_, b) in zip(got[j], got[k]):
            assert np.array_equal(a, b)
    # the first 8 calls against the oracle's decode of their own stream: every emitted chunk is a slice of it, except
    # its 240-sample start pad, which the reference takes from the stale head of its idle buffer (SileroVAD.py:90,101-102)
    for k in range(8):
        pcm = odsp.g711_decode(ulaw[k])
        for ipos, a in got[k]:
            assert np.array_equal(a[240:], pcm[ipos + 240:ipos + a.size])
    assert pipe.calls.fifo_len.cpu().tolist() == [(nt * 160) % 768] * N


def test_full_size_cycle_properties(built_lib):
    """BASELINE config 2 size (64 calls x 10 s), checked through size-independent properties: calls i and i+32 carry
    the same audio, speaker and text, so every stage must give them identical results wherever they sit in the batch
    (batch-position invariance); the mu-law round trip of the input is the identity on codes; STT sees the expected
    amount of speech; the TTS output has the expected length and is finite; a second cycle reproduces the first."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    dev = _lib.require_device('cuda:0')
    N = 64
    pipe = SpeechPipeline(N, dev, tts_lanes=1)
    pipe.speakers[32:] = pipe.speakers[:32]
    pipe.text_ids[32:] = pipe.text_ids[:32]
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, device=dev)
    pipe.tts.mask_source = lambda n: fixed
    x = np.stack([synth_utterance(1000 + (i % 32), 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    assert np.array_equal(odsp.g711_encode(odsp.g711_decode(ulaw)), np.where(ulaw == 0x7f, 0xff, ulaw))   # 0x7F -> 0xFF only
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
    outs = []
    for _ in range(2):
        r = pipe.run_steps(lambda k: frames, 1, pipelined=False)
        outs.append((r['tokens'].cpu(), r['no_speech_prob'].cpu(), r['stt_seconds'].clone(), r['ulaw'].cpu(),
                     r['tts_samples'].clone(), r['chunks']))
    toks, nsp, secs, ul, valid, chunks = outs[0]
    assert torch.equal(toks[:32], toks[32:]) and torch.equal(nsp[:32], nsp[32:])
    assert chunks[:32] == chunks[32:]
    assert torch.equal(ul[:32], ul[32:])
    assert bool((secs > 6.5).all()) and bool((secs < 9.5).all())
    assert ul.shape == (N, 10 * 4096) and valid.tolist() == [10 * 4096 - 256] * N
    audio = odsp.g711_decode(ul.numpy())
    assert np.isfinite(audio).all() and 1e-4 < np.abs(audio).mean() < 0.5
    for a, b in zip(outs[0][:5], outs[1][:5]):
        assert torch.equal(a, b)


class _RecMasks:
    def __init__(self, dev, seed):
        self.g, self.dev, self.log = torch.Generator().manual_seed(seed), dev, []

    def __call__(self, nsteps):
        m = torch.randint(0, 2, (nsteps, 2, 256), generator=self.g, dtype=torch.uint8)
        self.log.append(m)
        return m.to(self.dev)


# (round 6: the third case of rounds 2-5, C3 with greedy decode and the lane schedule, ran Whisper-base as 'C3-as-benched' does and
# greedy + lanes as 'C4-share' does; dropped for the suite's time limit)
@pytest.mark.parametrize('name,N,family,nheads,stt_beam,tts_mode', [('C3-as-benched', 128, 'whisper_base', 8, 5, 'continuous'),
                                                                   ('C4-share', 256, 'whisper_tiny', 6, 1, 'lanes')])
def test_baseline_config_full_cycle(built_lib, name, N, family, nheads, stt_beam, tts_mode):
    """BASELINE config 3 (128 calls, Whisper-base STT -> T2T stub -> TTS) and the per-GPU share of config 4 (256 calls):
    one full 10 s utterance cycle through SpeechPipeline at full size.  (i) size-independent properties: calls i and
    i + N/2 carry the same audio, speaker and text and must give identical results wherever they sit in the batch; the
    expected amount of speech reaches STT; TTS output length; a second cycle reproduces the first.  (ii) three calls
    stage by stage against the oracle on the same intermediate data: VAD chunks are slices of the oracle's decode, the
    first greedy token equals the fp32 oracle's where its top-2 margin exceeds the logit error, and the synthesised
    audio of the first three infer() calls is within the bf16 bar of oracle.tts_infer + resample + mu-law."""
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    H = N // 2
    pipe = SpeechPipeline(N, dev, whisper_family=family, tts_lanes=1, n_new_tokens=8, stt_beam=stt_beam, tts_mode=tts_mode)
    pipe.speakers[H:] = pipe.speakers[:H]
    pipe.text_ids[H:] = pipe.text_ids[:H]
    masks = _RecMasks(dev, 31)
    pipe.tts.mask_source = masks
    x = np.stack([synth_utterance(1000 + (i % H), 10.0) for i in range(N)])
    ulaw = odsp.g711_encode(x)
    frames = torch.from_numpy(np.ascontiguousarray(ulaw.reshape(N, 500, 160).transpose(1, 0, 2))).to(dev)
    outs = []
    for rep in range(2):
        masks.g.manual_seed(31)
        masks.log.clear()
        r = pipe.run_steps(lambda k: frames, 1, pipelined=False)
        outs.append((r['tokens'].cpu(), r['no_speech_prob'].cpu(), r['stt_seconds'].clone(), r['ulaw'].cpu(),
                     r['tts_samples'].clone(), r['chunks']))
    toks, nsp, secs, ul, valid, chunks = outs[0]
    # (i) properties
    assert torch.equal(toks[:H], toks[H:]) and torch.equal(nsp[:H], nsp[H:]) and chunks[:H] == chunks[H:]
    assert torch.equal(ul[:H], ul[H:])
    assert bool((secs > 6.5).all()) and bool((secs < 9.5).all())
    assert ul.shape == (N, 10 * 4096) and valid.tolist() == [10 * 4096 - 256] * N
    for a, b in zip(outs[0][:5], outs[1][:5]):
        assert torch.equal(a, b)
    # (ii) three calls against the oracle
    rows = [0, H // 3, H - 1]
    pcm = odsp.g711_decode(ulaw)
    sd_w = synth_state_dict(family, 0)
    for i in rows:
        lo, hi = chunks[i][0][0], chunks[i][-1][0] + chunks[i][-1][1]
        merged = np.zeros(hi - lo, np.float32)
        for (p, n) in chunks[i]:
            merged[p - lo + 240:p - lo + n] = pcm[i, p + 240:p + n]       # (the 240-sample start pad is stale audio, SileroVAD.py:90)
        assert abs(float(secs[i]) - merged.size / 8000.0) < 1e-6
        mel = torch.from_numpy(odsp.logmel(odsp.resample(merged, 8000, 16000)))[None]
        if stt_beam > 1:
            # the 5-beam search of the timed region (InfernSTTWorker.py:61-75): the device hypothesis is the fp32 oracle's,
            # or -- where bf16 logit error and the <= 240 stale start-pad samples reorder near-tied beams -- scores within that
            # error of the oracle's best when teacher-forced through the fp32 oracle
            with torch.no_grad():
                o_seqs, o_scores, o_enc = onn.whisper_beam(sd_w, mel, pipe.prompt[:1].long(), 8, nheads, stt_beam, 50257)
                mine = toks[i].tolist()
                if 50257 in mine:
                    mine = mine[:mine.index(50257) + 1]
                if mine != o_seqs[0]:
                    caches = [{'self': {}, 'cross': {}} for _ in range(len(pipe.whisper.dec_layers))]
                    full = torch.tensor([pipe.prompt[0].tolist() + mine])
                    lg = onn.whisper_decoder(sd_w, full[:, :-1], 0, o_enc, nheads, caches)[0]
                    lp = torch.log_softmax(lg[3:].float(), -1)
                    ts = float(sum(lp[j, t] for j, t in enumerate(mine))) / len(mine)
                    # per-token log-prob tolerance of a bf16 engine: 1.5 x the 99th percentile of the HF engine's own bf16-vs-fp32
                    # logit difference (tests/golden/whisper_tf.npz: 0.21 -> 0.32), plus the stale start-pad samples' share
                    assert ts > float(o_scores[0]) - 0.32 - 0.3 / len(mine), (name, i, mine, o_seqs[0], ts, float(o_scores[0]))
        else:
            with torch.no_grad():
                o_toks, o_first, _, _ = onn.whisper_greedy(sd_w, mel, pipe.prompt[:1].long(), 2, nheads)
            top2 = o_first.topk(2).values[0]
            if float(top2[0] - top2[1]) > 0.3:                # the start pad differs by <= 240 stale samples: allow for it
                assert int(toks[i, 0]) == int(o_toks[0, 0]), (name, i, toks[i].tolist(), o_toks.tolist())
    W = {k: synth_state_dict(k, 0, **({'stop_bias': -20.0} if k == 'speecht5_tts' else {})) for k in ('speecht5_tts', 'hifigan', 'amendment')}
    ost = onn.TTSState(W['speecht5_tts'], pipe.text_ids[rows].long(), torch.ones(len(rows), pipe.n_text, dtype=torch.int32),
                       pipe.speakers[rows])
    dev_pcm = odsp.g711_decode(ul[rows].numpy())
    for c in range(3):
        with torch.no_grad():
            a = onn.tts_infer(W['speecht5_tts'], W['hifigan'], W['amendment'], ost, masks.log[c])
        ref = odsp.g711_decode(odsp.g711_encode(odsp.resample(a.numpy(), 16000, 8000)))
        got = dev_pcm[:, c * 4096:(c + 1) * 4096]
        lo = 256 if c == 0 else 0                          # the first call's first 512 samples @16 k are never dispatched
        e = float(np.linalg.norm(got[:, lo:] - ref[:, lo:]) / np.linalg.norm(ref[:, lo:]))
        print('%s: TTS call %d decoded-mu-law rel_l2 vs oracle %.3e' % (name, c, e))
        assert e < 6e-2, (name, c, e)          # bf16 model error (<= 1.5 x the reference's own 1.3e-2) + two mu-law quantisations
    if pipe.ctts is not None:
        pipe.ctts.stop()


def test_bench_two_ranks_on_one_gpu_dry_run(built_lib):
    """The N > 1 bench path end to end -- plain `python bench.py --gpus 2` (self-launched ranks), sticky shards, ingress scatter / egress gather
    on two communicators issued from the pipelined schedule, barrier + MAX-over-ranks timing, one JSON line from rank 0 -- on
    a single-GPU box: IFH_DRYRUN_ONE_GPU=1 puts both ranks on cuda:0 and runs the collectives over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IFH_DRYRUN_ONE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    # the plain command the driver uses at N = 1, with --gpus 2: bench.py starts its own ranks (child process)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--config', 'C2', '--calls-per-gpu', '4', '--tts-lanes', '2', '--no-cpu-baseline', '--no-extra-configs']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['gpu_ids'] == [0, 0]          # dry run: both ranks on cuda:0
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['calls_total'] == 8 and d['config']['calls_per_gpu'] == 4
    assert d['value'] > 0 and d['scaling'] == 'weak' and abs(d['value'] - 8 * 10.0 / (d['ms_per_step'] * 1e-3)) < 0.01 * d['value']
    assert d['tts_samples_per_call'] == 10 * 4096 - 256 and 6.5 < d['stt_audio_seconds_per_call'] < 9.5


def test_bench_config5_share_leg(built_lib):
    """bench.py's configuration-5 leg (STT -> LLM -> TTS turn latency for one GPU's sessions) end to end at a small size:
    4 sessions, the test-sized LLM configuration."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--c5-only', '--c5-sessions', '4', '--c5-llm', 'qwen2_tiny64', '--steps', '2',
           '--tts-lanes', '1', '--front-lanes', '1']
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])['C5_share']
    assert d['turns'] == 2 and d['p50_turn_latency_ms'] > 0
    parts = d['stt_ms'] + d['llm_first_sentence_ms'] + d['tts_first_chunk_ms']
    assert abs(parts - d['p50_turn_latency_ms']) < 0.35 * d['p50_turn_latency_ms']
    assert d['llm_reply_ms'] > d['llm_first_sentence_ms'] and d['llm_decode_tokens_per_s'] > 0


def test_rccl_scatter_gather_on_device_tensors_world_1(built_lib):
    """The `nccl` (= RCCL) branch of shard.scatter_frames / gather_rows on DEVICE tensors, both communicators of the pipelined
    schedule, in a world of one rank -- what a single-GPU box can show of the multi-GPU path: the RCCL call signatures execute
    and return the right rows.  (Scaling over xGMI remains unmeasured: no multi-GPU box is reachable from this build.)"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    code = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from infernos_amd.shard import scatter_frames, gather_rows
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
g_in, g_out = dist.new_group(), dist.new_group()
frames = torch.randint(0, 256, (50, 7, 160), dtype=torch.uint8, device=dev)
for k in range(3):
    mine = scatter_frames(frames, 7, 50, dev, group=g_in, always_collective=True)
    assert mine.is_cuda and torch.equal(mine, frames)
    rows = torch.randint(0, 256, (7, 4096), dtype=torch.uint8, device=dev)
    full = gather_rows(rows, 7, group=g_out, always_collective=True)
    assert full.is_cuda and torch.equal(full, rows)
torch.cuda.synchronize()
print('rccl', torch.cuda.nccl.version(), 'backend', dist.get_backend(g_in))
dist.destroy_process_group()
""" % root
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'rccl' in r.stdout and 'nccl' in r.stdout, r.stdout
