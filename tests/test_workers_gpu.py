"""GPU: the drop-in worker threads (InfernSTTWorker / InfernTTSWorker) driven through the session
classes exactly as the reference's actors drive them."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dsp as odsp, nn as onn  # noqa: E402


class StubTokenizer:
    """Whisper tokenizer stand-in (the BPE files are not available offline)."""
    eos_token_id = 50257
    ids = {'<|startoftranscript|>': 50258, '<|en|>': 50259, '<|transcribe|>': 50359, '<|translate|>': 50358,
           '<|notimestamps|>': 50363, '<|nospeech|>': 50362}

    def convert_tokens_to_ids(self, t):
        return self.ids[t] if isinstance(t, str) else [self.ids[x] for x in t]

    def decode(self, ids, skip_special_tokens=True):
        return ' ' + ' '.join(str(i) for i in ids)


class IdsProcessor:
    def __call__(self, text, return_tensors='pt'):
        return {'input_ids': torch.tensor([[int(t) for t in text.split()]], dtype=torch.long)}


def test_stt_worker_through_sessions(built_lib):
    from infernos_amd import _lib
    from infernos_amd.audio import VadAudioChunk
    from infernos_amd.stt import InfernSTTWorker, STTRequest, STTResult, STTSentinel, STTSession
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    sd = synth_state_dict('whisper_tiny', 0)
    w = InfernSTTWorker(dev, weights=sd, tokenizer=StubTokenizer(), fixed_new_tokens=6, beam_size=1)     # the torch engine's path
    w.start()
    try:
        results, done = {}, threading.Event()
        sessions = [STTSession(w, keep_context=(i == 0)) for i in range(3)]
        x = [synth_utterance(1000 + i, 4.0) for i in range(3)]

        def cb(i):
            def f(result):
                results[i] = result
                if len(results) == 4:
                    done.set()
            return f
        for i, s in enumerate(sessions):
            req = STTRequest(VadAudioChunk(torch.from_numpy(x[i][8000:24000]).to(dev), 8000, 8000), cb(i), 'en')
            req.mode = 'translate' if i == 1 else 'transcribe'
            s.soundin(req)
        sessions[2].soundin(STTSentinel('flush', cb(3)))          # echoed after session 2's request completes
        assert done.wait(120)
        for i in range(3):
            r = results[i]
            assert isinstance(r, STTResult) and r.duration == 2 and r.inf_time > 0
            toks = [int(t) for t in r.text.split()]
            assert len(toks) == 6 and not r.text.startswith(' ')
            mel = torch.from_numpy(odsp.logmel(odsp.resample(x[i][8000:24000], 8000, 16000)))[None]
            prompt = torch.tensor([[50258, 50259, 50358 if i == 1 else 50359, 50363]])
            with torch.no_grad():
                o_toks, o_first, o_l0, _ = onn.whisper_greedy(sd, mel, prompt, 6, 6)
            top2 = o_first.topk(2).values[0]
            if float(top2[0] - top2[1]) > 0.05:
                assert toks[0] == int(o_toks[0, 0])
            nsp = float(torch.softmax(o_l0, -1)[0, 50362])
            assert abs(r.no_speech_prob - nsp) <= 0.5 * nsp + 1e-12
        assert isinstance(results[3], STTSentinel) and results[3].signal == 'flush'
        assert sessions[0].context == []                           # (c + t)[:-224] of a short output is empty
        # every row above max_ns_prob -> empty text, nothing generated (InfernSTTWorker.py:91-92)
        got = []
        req = STTRequest(VadAudioChunk(torch.zeros(8000, device=dev), 8000, 0), lambda result: got.append(result), 'en')
        req.max_ns_prob = -1.0
        ev = threading.Event()
        req.text_cb = lambda result: (got.append(result), ev.set())
        sessions[1].soundin(req)
        assert ev.wait(60) and got[0].text == ''
    finally:
        w.stop()


def test_tts_worker_through_sessions(built_lib):
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.muxer import ASMarkerNewSent
    from infernos_amd.tts import InfernTTSWorker, TTSRequest, TTSSession
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0), 'hifigan': synth_state_dict('hifigan', 0),
         'amendment': synth_state_dict('amendment', 0)}
    g = torch.Generator().manual_seed(1)
    voices = [torch.randn(1, 512, generator=g) for _ in range(4)]
    w = InfernTTSWorker('en', 8000, dev, weights=W, processor=IdsProcessor(), speaker_embeddings=voices)
    assert w.max_batch_size == 8 and w.output_sr == 8000 and w.get_voice(2) is voices[2]
    w.start()
    try:
        out = {0: [], 1: []}
        fin = threading.Event()

        def so(i):
            def f(chunk):
                out[i].append(chunk)
                if all(o and isinstance(o[-1], ASMarkerNewSent) for o in out.values()):
                    fin.set()
            return f
        sess = [TTSSession(w, None) for _ in range(2)]
        for i, s in enumerate(sess):
            s.start(so(i))
        sess[0].say(TTSRequest('44 45 46 47 48 49 50 51 52', speaker_id=1))      # stops at step 0 with these weights
        sess[1].say(TTSRequest('5 17 33 8', speaker_id=2))
        assert fin.wait(180)
        for i in range(2):
            chunks = [c for c in out[i] if isinstance(c, AudioChunk)]
            assert chunks and all(c.samplerate == 8000 and c.audio.dim() == 1 and not c.audio.is_cuda for c in chunks)
            assert isinstance(out[i][-1], ASMarkerNewSent)
            assert all(torch.isfinite(c.audio.float()).all() for c in chunks)
    finally:
        w.stop()


def test_tts_worker_continuous_mode_gives_every_session_its_own_batch_audio(built_lib):
    """InfernTTSWorker(continuous=True): requests join the running decode batch at the next infer() boundary (tts.ContinuousTTS)
    instead of queueing behind a frozen batch (Cluster/InfernTTSWorker.py:83-92).  Three sessions speak overlapping utterances of
    different lengths; each receives, chunk for chunk and byte for byte, what the frozen-batch worker sends when the same request
    is synthesised alone, and the end-of-sentence marker after it."""
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.muxer import ASMarkerNewSent
    from infernos_amd.tts import InfernTTSWorker, TTSRequest, TTSSession
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0, stop_bias=-20.0), 'hifigan': synth_state_dict('hifigan', 0),
         'amendment': synth_state_dict('amendment', 0)}
    g = torch.Generator().manual_seed(1)
    voices = [torch.randn(1, 512, generator=g) for _ in range(4)]
    fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(dev)
    texts = ['5 17 33', '44 45 46 47 48 49', '9 8 7 6']             # 3, 6 and 4 tokens: utterances of different lengths (maxlen arm of the stop rule)

    def run(continuous):
        w = InfernTTSWorker('en', 8000, dev, weights=W, processor=IdsProcessor(), speaker_embeddings=voices, continuous=continuous)
        w.tts_engine.mask_source = lambda n: fixed
        w.start()
        out = [[] for _ in texts]
        try:
            evs = [threading.Event() for _ in texts]

            def so(i):
                def f(chunk):
                    out[i].append(chunk)
                    if isinstance(chunk, ASMarkerNewSent):
                        evs[i].set()
                return f
            sess = [TTSSession(w, None) for _ in texts]
            for i, s_ in enumerate(sess):
                s_.start(so(i))
            for i, s_ in enumerate(sess):
                s_.say(TTSRequest(texts[i], speaker_id=i))
                if not continuous:
                    assert evs[i].wait(180)          # one frozen batch of one at a time
            assert all(e.wait(180) for e in evs)
        finally:
            w.stop()
        return out
    ref, got = run(False), run(True)
    for i in range(len(texts)):
        a = [c.audio for c in ref[i] if isinstance(c, AudioChunk)]
        b = [c.audio for c in got[i] if isinstance(c, AudioChunk)]
        assert len(a) == len(b) and len(a) >= 2, (i, len(a), len(b))
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.int16), y.view(torch.int16)), i
        assert isinstance(got[i][-1], ASMarkerNewSent) and sum(isinstance(c, ASMarkerNewSent) for c in got[i]) == 1


def test_stt_worker_out_of_memory_retries_one_by_one(built_lib):
    """The reference's recovery at Cluster/InfernSTTWorker.py:66-72: a batch that runs out of device memory is retried
    request by request after the allocator's cache has been dropped; results and their order are those of the batch."""
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.stt import InfernSTTWorker, STTRequest
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    w = InfernSTTWorker(dev, weights=synth_state_dict('whisper_tiny', 0), tokenizer=StubTokenizer(), fixed_new_tokens=5)
    aud = [torch.from_numpy(odsp.resample(synth_utterance(1000 + i, 3.0)[8000:20000], 8000, 16000)) for i in range(3)]

    def run():
        got = []
        wis = [(STTRequest(AudioChunk(a, 16000), None, 'en'), (lambda result, i=i: got.append((i, result.text, result.no_speech_prob))), None)
               for i, a in enumerate(aud)]
        w.process_batch(wis)
        return got
    ref = run()
    calls, real = [], w.transcribe_batch

    def flaky(audios, prompts, max_nsps):
        calls.append(len(audios))
        if len(audios) > 1:
            raise RuntimeError('HIP out of memory. Tried to allocate 20.00 MiB')
        return real(audios, prompts, max_nsps)
    w.transcribe_batch = flaky
    got = run()
    assert calls == [3, 1, 1, 1]
    assert [g[:2] for g in got] == [r[:2] for r in ref]
    for g, r in zip(got, ref):
        assert abs(g[2] - r[2]) <= 1e-3 * abs(r[2]) + 1e-12
    w.transcribe_batch = lambda *a: (_ for _ in ()).throw(RuntimeError('something else'))
    with pytest.raises(RuntimeError, match='something else'):
        run()


def test_tts_worker_audio_matches_oracle(built_lib):
    """InfernTTSWorker through TTSSession (process_batch -> infer -> unbatch_and_dispatch -> TTSSndDispatch -> soundout):
    the audio that reaches `soundout` is the oracle's tts_infer output for the same text, voice and dropout masks, cut at
    the reference's dispatch offsets, within the bf16 bar of test_tts_pipe_matches_reference_run."""
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.muxer import ASMarkerNewSent
    from infernos_amd.tts import InfernTTSWorker, TTSRequest, TTSSession
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0, stop_bias=-20.0), 'hifigan': synth_state_dict('hifigan', 0),
         'amendment': synth_state_dict('amendment', 0)}
    g = torch.Generator().manual_seed(1)
    voices = [torch.randn(1, 512, generator=g) for _ in range(4)]
    mg = torch.Generator().manual_seed(8)
    masks = []

    def mask_source(n):
        m = torch.randint(0, 2, (n, 2, 256), generator=mg, dtype=torch.uint8)
        masks.append(m)
        return m.to(dev)
    w = InfernTTSWorker('en', 16000, dev, weights=W, processor=IdsProcessor(), speaker_embeddings=voices, mask_source=mask_source)
    w.start()
    try:
        out, fin = [], threading.Event()

        def soundout(chunk):
            out.append(chunk)
            if isinstance(chunk, ASMarkerNewSent):
                fin.set()
        sess = TTSSession(w, None)
        sess.start(soundout)
        text = '5 17 33 8 61'                                   # 5 tokens: the length rule stops it at step 50 (-> ends_at 52)
        sess.say(TTSRequest(text, speaker_id=2))
        assert fin.wait(180)
    finally:
        w.stop()
    got = torch.cat([c.audio.float() for c in out if isinstance(c, AudioChunk)])
    ids = torch.tensor([[int(t) for t in text.split()]])
    st = onn.TTSState(W['speecht5_tts'], ids, torch.ones_like(ids).int(), voices[2])
    ref = []
    with torch.no_grad():
        for c in range(len(masks)):
            a = onn.tts_infer(W['speecht5_tts'], W['hifigan'], W['amendment'], st, masks[c])
            (s0, e0, _), = onn.tts_dispatch_offsets(st.idx, st.starts_at.tolist(), st.ends_at.tolist())[0]
            ref.append(a[0, s0:e0])
    ref = torch.cat(ref)
    assert got.numel() == ref.numel() == (52 - 1) * 512, (got.numel(), ref.numel())
    e = float((got.double() - ref.double()).norm() / ref.double().norm())
    print('TTS worker through session: rel_l2 vs oracle %.3e over %d samples, %d infer calls' % (e, got.numel(), len(masks)))
    assert e < 3e-2, e


def test_stt_worker_default_decode_is_beam_search(built_lib):
    """The worker's default decode is the reference's default engine's: beam search with 5 beams
    (Cluster/InfernSTTWorker.py:61-75 through ctranslate2's defaults), every request decoded whatever its
    max_ns_prob, no_speech_prob reported.  Checked against the oracle's search on the same audio."""
    from infernos_amd import _lib
    from infernos_amd.audio import AudioChunk
    from infernos_amd.stt import InfernSTTWorker, STTRequest
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    dev = _lib.require_device('cuda:0')
    sd = synth_state_dict('whisper_tiny', 0)
    V = 51865
    specials = [i for i in range(50257, V) if i != 50257]
    w = InfernSTTWorker(dev, weights=sd, tokenizer=StubTokenizer(), max_new_tokens=7, suppress_tokens=specials,
                        begin_suppress_tokens=[220, 50257])
    assert w.beam_size == 5
    aud = [odsp.resample(synth_utterance(1000 + i, 3.0)[8000:20000], 8000, 16000) for i in range(3)]
    got = []
    wis = []
    for i, a in enumerate(aud):
        req = STTRequest(AudioChunk(torch.from_numpy(a), 16000), None, 'en')
        req.max_ns_prob = -1.0                 # would suppress generation on the torch engine's path; not here
        wis.append((req, (lambda result, i=i: got.append((i, result))), None))
    w.process_batch(wis)
    assert [g[0] for g in got] == [0, 1, 2]
    sup = torch.zeros(V)
    sup[specials] = float('-inf')
    bs = torch.zeros(V)
    bs[[220, 50257]] = float('-inf')
    same = 0
    for i, r in got:
        toks = [int(t) for t in r.text.split()]
        mel = torch.from_numpy(odsp.logmel(aud[i][None]))
        with torch.no_grad():
            seqs, scores, enc = onn.whisper_beam(sd, mel, torch.tensor([[50258, 50259, 50359, 50363]]), 7, 6, 5, 50257, 1.0,
                                                 suppress=sup, begin_suppress=bs)
            o_l0 = onn.whisper_decoder(sd, torch.tensor([[50258]]), 0, enc, 6, [{'self': {}, 'cross': {}} for _ in range(4)])[:, 0]
        ref = [t for t in seqs[0] if t != 50257]
        same += int(toks == ref)
        assert len(toks) == len(ref) == 7
        nsp = float(torch.softmax(o_l0, -1)[0, 50362])
        assert abs(r.no_speech_prob - nsp) <= 0.5 * nsp + 1e-12
    assert same >= 2, same
