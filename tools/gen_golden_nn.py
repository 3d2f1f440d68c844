"""Golden vectors for the neural rows: run the REFERENCE's own code
(HelloSippyRTPipe.infer / unbatch_and_dispatch, InfernSTTWorker.process_batch) on the
third-party engines it uses (transformers), with the seeded synthetic weights of
infernos_amd.weights loaded into those engines.  Imported by tools/gen_golden.py.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _ids_processor(text, return_tensors='pt'):
    return {'input_ids': torch.tensor([[int(t) for t in text.split()]], dtype=torch.long)}


def _make_pipe(sd_t5, sd_voc, sd_amd, half):
    import HelloSippyTTSRT.HelloSippyRTPipe as mod
    from HelloSippyTTSRT.HelloSippyRT import AmendmentNetwork1, AmendmentNetwork1Config
    from transformers import SpeechT5ForTextToSpeech, SpeechT5Config, SpeechT5HifiGan, SpeechT5HifiGanConfig
    from config.InfernGlobals import InfernGlobals
    if not half:
        mod.maybe_half = lambda x: x
    else:
        import importlib
        importlib.reload(mod)
    pp = object.__new__(mod.HelloSippyRTPipe)
    pp.cuda_lock = InfernGlobals().torcher
    pp.processor = _ids_processor
    model = SpeechT5ForTextToSpeech(SpeechT5Config())
    model.load_state_dict(sd_t5, strict=True)
    voc = SpeechT5HifiGan(SpeechT5HifiGanConfig())
    voc.load_state_dict(sd_voc, strict=True)
    amd = AmendmentNetwork1(AmendmentNetwork1Config())
    amd.load_state_dict(sd_amd, strict=True)
    if half:
        model, voc, amd = mod.maybe_half(model), mod.maybe_half(voc), mod.maybe_half(amd)
    pp.model, pp.vocoder, pp.chunker = model.eval(), voc.eval(), amd.eval()
    pp.resampler = None
    pp.output_sr = 16000
    return mod, pp


def _run_tts(sd_t5, sd_voc, sd_amd, texts, speakers, ncalls, half, seed):
    mod, pp = _make_pipe(sd_t5, sd_voc, sd_amd, half)
    masks = []
    real_bernoulli = torch.bernoulli

    def rec_bernoulli(inp, p=None, **kw):
        m = real_bernoulli(inp, p=p, **kw)
        masks.append(m[-1].to(torch.uint8).clone())
        return m
    dispatched = [[] for _ in texts]
    reqs = []
    for i, (t, s) in enumerate(zip(texts, speakers)):
        def disp(chunk, i=i):
            dispatched[i].append(None if chunk is None else chunk.float().numpy().copy())
        reqs.append(mod.HelloSippyPlayRequest(None, t, s, disp))
    torch.manual_seed(seed)
    torch.bernoulli = rec_bernoulli
    try:
        with torch.no_grad():
            states = [mod.HelloSippyPipeState(pp, r) for r in reqs]
            st = mod.HelloSippyPipeStateBatched(states, pp)
            enc = st.encoder_last_hidden_state.float().clone()
            audios, more, bookkeeping = [], [], []
            for c in range(ncalls):
                pp.infer(st)
                audios.append(st.audio.float().clone())
                bookkeeping.append({'idx': int(st.idx), 'starts_at': st.starts_at.tolist(), 'ends_at': st.ends_at.tolist()})
                with contextlib.redirect_stdout(io.StringIO()):
                    more.append(bool(pp.unbatch_and_dispatch(st)))
    finally:
        torch.bernoulli = real_bernoulli
    m = torch.stack(masks).view(ncalls, 16, 2, 256).numpy()
    return dict(enc=enc, audios=audios, more=more, book=bookkeeping, masks=m, dispatched=dispatched,
                maxlen=st.maxlen, pre_frames=st.pre_frames.float().clone())


def gen_tts():
    from infernos_amd.weights import synth_state_dict
    from oracle import nn as onn
    sd_voc = synth_state_dict('hifigan', 0)
    sd_amd = synth_state_dict('amendment', 0)
    texts = ['5 17 33 8 21 60 4', '9 10 11 12 13 14 15 16 17 18 19 70', '44 45 46 47 48 49 50 51 52']
    g = torch.Generator().manual_seed(2000)
    speakers = [torch.randn(1, 512, generator=g) for _ in texts]
    out = {}
    meta = {'source': 'HelloSippyTTSRT/HelloSippyRTPipe.py:71-121,191-259 run on transformers 5.15.0 modules holding '
                      'infernos_amd.weights.synth_state_dict(seed 0) weights', 'texts': texts, 'speaker_seed': 2000}
    # ---- scenario A: stop head disabled (prob_out.bias=-20), 2 calls, fp32 and the reference's bf16
    sdA = synth_state_dict('speecht5_tts', 0, stop_bias=-20.0)
    A = _run_tts(sdA, sd_voc, sd_amd, texts, speakers, 2, half=False, seed=3000)
    Ab = _run_tts(sdA, sd_voc, sd_amd, texts, speakers, 2, half=True, seed=3000)
    assert np.array_equal(A['masks'], Ab['masks'])
    out['A_masks'] = np.packbits(A['masks'], axis=-1)
    out['A_enc_slice'] = A['enc'][:, :, :16].numpy()
    for c in range(2):
        out['A_audio_%d' % c] = A['audios'][c][:, ::8].numpy()
        out['A_audio_bf16_%d' % c] = Ab['audios'][c][:, ::8].numpy()
    meta['A'] = {'book': A['book'], 'more': A['more'], 'maxlen': A['maxlen'],
                 'dispatch_lens': [[None if d is None else int(d.size) for d in ch] for ch in A['dispatched']],
                 'enc_sum': float(A['enc'].double().sum()), 'audio_abs_mean': [float(a.abs().mean()) for a in A['audios']],
                 'bf16_vs_fp32_max_abs': [float((a - b).abs().max()) for a, b in zip(A['audios'], Ab['audios'])],
                 'bf16_vs_fp32_rel_l2': [float((a - b).norm() / a.norm()) for a, b in zip(A['audios'], Ab['audios'])]}
    # first dispatched chunk of each row == audio[startoff:]
    out['A_first_dispatch_0'] = A['dispatched'][0][0][::8]
    # ---- oracle cross-check while we are here (printed, also asserted in tests)
    ids = [torch.tensor([[int(t) for t in s.split()]]) for s in texts]
    T = max(i.size(1) for i in ids)
    inp = torch.cat([torch.nn.functional.pad(i, (0, T - i.size(1))) for i in ids])
    msk = torch.cat([torch.nn.functional.pad(torch.ones_like(i), (0, T - i.size(1))) for i in ids]).int()
    st = onn.TTSState(sdA, inp, msk, torch.cat(speakers))
    print('enc diff', float((st.enc - A['enc']).abs().max()))
    stages = {}
    for c in range(2):
        a = onn.tts_infer(sdA, sd_voc, sd_amd, st, torch.from_numpy(A['masks'][c]), stages=stages)
        print('call', c, 'audio diff', float((a - A['audios'][c]).abs().max()), 'scale', float(a.abs().mean()),
              'voc scale', float(stages['vocoder'].abs().mean()), 'mel scale', float(stages['postnet'].abs().mean()))
        if c == 0:
            out['A_postnet_0'] = stages['postnet'].numpy()
            out['A_vocoder_0'] = stages['vocoder'][:, ::16].numpy()
    # ---- scenario B: natural stop head (random weights stop early) -> dispatch bookkeeping
    sdB = synth_state_dict('speecht5_tts', 0)
    Bx = _run_tts(sdB, sd_voc, sd_amd, texts, speakers, 3, half=False, seed=3001)
    out['B_masks'] = np.packbits(Bx['masks'], axis=-1)
    meta['B'] = {'book': Bx['book'], 'more': Bx['more'], 'maxlen': Bx['maxlen'],
                 'dispatch_lens': [[None if d is None else int(d.size) for d in ch] for ch in Bx['dispatched']]}
    for i, ch in enumerate(Bx['dispatched']):
        for j, d in enumerate(ch):
            if d is not None:
                out['B_disp_%d_%d' % (i, j)] = d[::8]
    np.savez_compressed(os.path.join(GOLD, 'tts.npz'), **out)
    import json
    json.dump(meta, open(os.path.join(GOLD, 'tts_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote tts.npz / tts_meta.json', meta['A']['bf16_vs_fp32_rel_l2'], meta['B']['book'])


def gen_whisper():
    import json
    import types
    from transformers import WhisperConfig, WhisperForConditionalGeneration, WhisperFeatureExtractor
    from infernos_amd.weights import synth_state_dict
    from Cluster.InfernSTTWorker import InfernSTTWorker
    from Cluster.InfernBatchedWorker import InfernBatchedWorker
    from Cluster.STTSession import STTRequest
    from Core.AudioChunk import AudioChunk
    from oracle import nn as onn, dsp
    sys.path.insert(0, HERE)
    from gen_golden import synth_utterance
    sd = synth_state_dict('whisper_tiny', 0)
    model = WhisperForConditionalGeneration(WhisperConfig())
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    fe = WhisperFeatureExtractor()
    PROMPT = [50258, 50259, 50359, 50363]     # <|startoftranscript|><|en|><|transcribe|><|notimestamps|> (multilingual ids)
    NOSPEECH = 50362

    class Tok:
        pad_token_id = 50257

        def convert_tokens_to_ids(self, toks):
            if isinstance(toks, str):
                return NOSPEECH
            return list(PROMPT[:len(toks)])

    class Proc:
        tokenizer = Tok()

        def __call__(self, audios, sampling_rate=None, return_tensors='pt'):
            return fe(audios, sampling_rate=sampling_rate, return_tensors=return_tensors)

        def batch_decode(self, seqs, skip_special_tokens=True):
            return [' '.join(str(int(t)) for t in s) for s in seqs]
    w = object.__new__(InfernSTTWorker)
    InfernBatchedWorker.__init__(w)
    w.model, w.processor, w.device = model, Proc(), 'cpu'
    w.no_speech_token_id = NOSPEECH
    from functools import partial
    w.process_audios = partial(w.processor, return_tensors='pt')
    w.infer_and_decode = w.infer_and_decode_torch
    auds = [dsp.resample(synth_utterance(1000 + i, 10.0), 8000, 16000) for i in range(2)]
    results = []
    wis = []
    for a in auds:
        ch = AudioChunk(torch.from_numpy(a), 16000)
        ch.audio = ch.audio.numpy()
        req = STTRequest(ch, None, 'en')
        req.max_ns_prob = -1.0            # every no-speech prob is above it -> early return, no generate()
        wis.append((req, lambda result: results.append(result), None))
    w.process_batch(wis)
    nsp = [float(r.no_speech_prob) for r in results]
    # engine-level pins (HF forward with the reference's prompt layout)
    mel = torch.from_numpy(fe(auds, sampling_rate=16000, return_tensors='np').input_features)
    prompt = torch.tensor([PROMPT, PROMPT])
    with torch.no_grad():
        fo = model(input_features=mel, decoder_input_ids=prompt)
        enc = fo.encoder_last_hidden_state
        logits = fo.logits
        # manual greedy, 8 tokens
        toks = prompt.clone()
        outs = []
        for s in range(8):
            lg = model(input_features=mel, decoder_input_ids=toks).logits[:, -1]
            nxt = lg.argmax(-1)
            outs.append(nxt)
            toks = torch.cat([toks, nxt[:, None]], 1)
    greedy = torch.stack(outs, 1)
    o_toks, o_first, o_l0, o_enc = onn.whisper_greedy(sd, mel, prompt, 8, 6)
    print('whisper enc diff', float((o_enc - enc).abs().max()), 'logit diff', float((o_first - logits[:, -1]).abs().max()),
          'tokens equal', bool((o_toks == greedy).all()), 'nsp', nsp,
          float(torch.softmax(o_l0, -1)[0, NOSPEECH]))
    np.savez_compressed(os.path.join(GOLD, 'whisper.npz'), enc_slice=enc[:, ::25, :32].numpy(),
                        first_logits_slice=logits[:, -1, ::97].numpy(), logits0_slice=logits[:, 0, ::97].numpy(),
                        greedy=greedy.numpy())
    json.dump({'source': 'Cluster/InfernSTTWorker.py:77-92,109-123 (process_batch -> infer_and_decode_torch early-return '
                         'path) on transformers 5.15.0 WhisperForConditionalGeneration(WhisperConfig()) holding '
                         'synth_state_dict(whisper_tiny, 0); greedy tokens from the same engine',
               'prompt': PROMPT, 'no_speech_id': NOSPEECH, 'no_speech_prob': nsp, 'audio_seeds': [1000, 1001],
               'enc_sum': float(enc.double().sum()), 'first_logits_max': [float(x) for x in logits[:, -1].max(-1).values]},
              open(os.path.join(GOLD, 'whisper_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote whisper.npz')


def gen_whisper_tf():
    """Teacher-forced decoder logits at every one of 36 positions (4 prompt + 32 forced tokens) from the engine the
    reference's torch path drives (HF WhisperForConditionalGeneration, Cluster/InfernSTTWorker.py:83-88
    `self.model(**inputs, decoder_input_ids=...)`), in fp32 and in bf16 -- the latter is the yardstick for the bf16
    device path: the device is held to <= 1.5x the error the same engine makes when it merely runs in bf16.
    tiny (BASELINE configs 1-2) keeps logit slices; base (config 3) keeps the per-position error norms."""
    import json
    from transformers import WhisperConfig, WhisperForConditionalGeneration, WhisperFeatureExtractor
    from infernos_amd.weights import synth_state_dict
    from oracle import nn as onn, dsp
    sys.path.insert(0, HERE)
    from gen_golden import synth_utterance
    fe = WhisperFeatureExtractor()
    PROMPT = [50258, 50259, 50359, 50363]
    meta, arrays = {}, {}
    base_cfg = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8,
                    encoder_ffn_dim=2048, decoder_ffn_dim=2048)
    for fam, seed, cfg, nh, aseeds in (('whisper_tiny', 0, WhisperConfig(), 6, [1000, 1001]),
                                       ('whisper_base', 1, WhisperConfig(**base_cfg), 8, [1000, 1001])):
        sd = synth_state_dict(fam, seed)
        model = WhisperForConditionalGeneration(cfg)
        model.load_state_dict(sd, strict=True)
        model.eval()
        vocab = sd['model.decoder.embed_tokens.weight'].shape[0]
        g = torch.Generator().manual_seed(4242)
        forced = torch.randint(0, 50257, (len(aseeds), 32), generator=g)
        toks = torch.cat([torch.tensor([PROMPT] * len(aseeds)), forced], 1)               # [2, 36]
        auds = [dsp.resample(synth_utterance(a, 10.0), 8000, 16000) for a in aseeds]
        mel = torch.from_numpy(fe(auds, sampling_rate=16000, return_tensors='np').input_features)
        with torch.no_grad():
            fo32 = model(input_features=mel, decoder_input_ids=toks)
            l32, enc32 = fo32.logits, fo32.encoder_last_hidden_state                      # [2, 36, V], [2, 1500, d]
            mb = model.to(torch.bfloat16)
            fo16 = mb(input_features=mel.to(torch.bfloat16), decoder_input_ids=toks)
            l16, enc16 = fo16.logits.float(), fo16.encoder_last_hidden_state.float()
            model.to(torch.float32)
            # the oracle (oracle/nn.py) on the same inputs: pins it for teacher-forced steps as well
            enc = onn.whisper_encoder(sd, mel, nh)
            nl = len([k for k in sd if k.startswith('model.decoder.layers.') and k.endswith('.fc1.weight')])
            caches = [{'self': {}, 'cross': {}} for _ in range(nl)]
            lo = onn.whisper_decoder(sd, toks, 0, enc, nh, caches)
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        e16 = [rel(l16[:, t], l32[:, t]) for t in range(toks.size(1))]
        eo = max(rel(lo[:, t], l32[:, t]) for t in range(toks.size(1)))
        print(fam, 'oracle vs HF fp32 (max over positions) %.2e ; HF bf16 vs fp32 per position: min %.2e max %.2e' % (eo, min(e16), max(e16)))
        assert eo < 1e-4
        meta[fam] = {'weights_seed': seed, 'audio_seeds': aseeds, 'tokens': toks.tolist(), 'nheads': nh,
                     'bf16_vs_fp32_rel_l2': e16, 'oracle_vs_hf_fp32_rel_l2_max': eo, 'vocab': vocab,
                     # the same engine's encoder output, bf16 run against fp32 run: the bar for the device encoder
                     'enc_bf16_vs_fp32_rel_l2': rel(enc16, enc32), 'oracle_enc_vs_hf_fp32_rel_l2': rel(enc, enc32)}
        if fam == 'whisper_tiny':
            arrays['tiny_logits_fp32_slice'] = l32[:, :, ::97].numpy()
            arrays['tiny_logits_bf16_slice'] = l16[:, :, ::97].numpy()
        else:
            arrays['base_logits_fp32_slice'] = l32[:, :, ::389].numpy()
    meta['source'] = ('transformers 5.15.0 WhisperForConditionalGeneration holding synth_state_dict weights, called as '
                      'Cluster/InfernSTTWorker.py:83-88 does (input_features + decoder_input_ids), fp32 and bf16')
    np.savez_compressed(os.path.join(GOLD, 'whisper_tf.npz'), **arrays)
    json.dump(meta, open(os.path.join(GOLD, 'whisper_tf_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote whisper_tf.npz')


def gen_whisper_beam():
    """Beam-search fixtures: transformers' GenerationMixin.generate(num_beams=K) (the base class, not Whisper's
    long-form override) on the seeded whisper_tiny, with the special-token range suppressed so that the search has
    real alternatives, for several (K, eos, length, length_penalty) cases; oracle/nn.py:beam_search is pinned to
    them."""
    import json
    from transformers import WhisperConfig, WhisperForConditionalGeneration, WhisperFeatureExtractor
    from transformers.generation.utils import GenerationMixin
    from infernos_amd.weights import synth_state_dict
    from oracle import nn as onn, dsp
    sys.path.insert(0, HERE)
    from gen_golden import synth_utterance
    sd = synth_state_dict('whisper_tiny', 0)
    model = WhisperForConditionalGeneration(WhisperConfig())
    model.load_state_dict(sd, strict=True)
    model.eval()
    fe = WhisperFeatureExtractor()
    PROMPT = [50258, 50259, 50359, 50363]
    V = 51865
    auds = [dsp.resample(synth_utterance(1000 + i, 10.0), 8000, 16000) for i in range(2)]
    mel = torch.from_numpy(fe(auds, sampling_rate=16000, return_tensors='np').input_features)
    prompt = torch.tensor([PROMPT, PROMPT])
    cases = [dict(beams=5, eos=50257, n_new=12, lp=1.0, begin=[]),
             dict(beams=5, eos=5880, n_new=12, lp=1.0, begin=[]),
             dict(beams=5, eos=21251, n_new=12, lp=1.0, begin=[]),
             dict(beams=5, eos=21251, n_new=10, lp=2.0, begin=[6435]),
             dict(beams=5, eos=6435, n_new=8, lp=0.5, begin=[220, 6435]),
             dict(beams=3, eos=23249, n_new=9, lp=1.0, begin=[]),
             dict(beams=2, eos=5880, n_new=6, lp=1.0, begin=[5880]),
             dict(beams=4, eos=9923, n_new=16, lp=1.0, begin=[]),
             dict(beams=5, eos=5880, n_new=12, lp=0.0, begin=[]),          # un-normalised scores: short hypotheses win
             dict(beams=5, eos=6435, n_new=12, lp=0.0, begin=[]),
             dict(beams=3, eos=21251, n_new=12, lp=0.0, begin=[]),
             dict(beams=5, eos=5880, n_new=12, lp=0.3, begin=[]),
             dict(beams=5, eos=21251, n_new=12, lp=0.2, begin=[5880])]
    arrays, meta = {}, []
    for ci, c in enumerate(cases):
        sup_ids = [i for i in range(50257, V) if i != c['eos']]
        with torch.no_grad():
            out = GenerationMixin.generate(model, input_features=mel, decoder_input_ids=prompt, num_beams=c['beams'],
                                           do_sample=False, max_new_tokens=c['n_new'], eos_token_id=c['eos'],
                                           pad_token_id=50256, length_penalty=c['lp'], early_stopping=False,
                                           return_dict_in_generate=True, output_scores=True, suppress_tokens=sup_ids,
                                           begin_suppress_tokens=c['begin'] or None, forced_decoder_ids=None)
        seq = out.sequences[:, len(PROMPT):]
        # generated length per row: up to and including the first eos (the rest is padding)
        lens = []
        for r in seq.tolist():
            lens.append(r.index(c['eos']) + 1 if c['eos'] in r else len(r))
        sup = torch.zeros(V)
        sup[sup_ids] = float('-inf')
        bs = None
        if c['begin']:
            bs = torch.zeros(V)
            bs[c['begin']] = float('-inf')
        o_seq, o_sc, _ = onn.whisper_beam(sd, mel, prompt, c['n_new'], 6, c['beams'], c['eos'], c['lp'], suppress=sup,
                                          begin_suppress=bs)
        same = all(o_seq[b] == seq[b, :lens[b]].tolist() for b in range(2))
        print('beam case', ci, c, 'lens', lens, 'tokens equal', same, 'score diff',
              float((o_sc - out.sequences_scores).abs().max()))
        arrays['seq%d' % ci] = seq.numpy().astype(np.int32)
        arrays['len%d' % ci] = np.asarray(lens, np.int32)
        arrays['score%d' % ci] = out.sequences_scores.numpy()
        meta.append(c)
    # ---- Whisper-BASE (BASELINE config 3), the search bench.py times: 5 beams, 32 new tokens, length_penalty 1; four
    # x four utterances; a second case ends early on an eos taken from the first case's output
    base_cfg = dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8, decoder_attention_heads=8,
                    encoder_ffn_dim=2048, decoder_ffn_dim=2048)
    sdb = synth_state_dict('whisper_base', 1)
    mb = WhisperForConditionalGeneration(WhisperConfig(**base_cfg))
    mb.load_state_dict(sdb, strict=True)
    mb.eval()
    bseeds = list(range(1000, 1016))           # 16 utterances x 2 cases = 32 hypotheses (round 5: the exact-match count as a rate)
    bauds = [dsp.resample(synth_utterance(a, 10.0), 8000, 16000) for a in bseeds]
    bmel = torch.from_numpy(fe(bauds, sampling_rate=16000, return_tensors='np').input_features)
    bprompt = torch.tensor([PROMPT] * len(bseeds))
    bcases = [dict(beams=5, eos=50257, n_new=32, lp=1.0, begin=[])]
    bmeta = []
    ci = 0
    while ci < len(bcases):
        c = bcases[ci]
        sup_ids = [i for i in range(50257, V) if i != c['eos']]
        with torch.no_grad():
            out = GenerationMixin.generate(mb, input_features=bmel, decoder_input_ids=bprompt, num_beams=c['beams'],
                                           do_sample=False, max_new_tokens=c['n_new'], eos_token_id=c['eos'],
                                           pad_token_id=50256, length_penalty=c['lp'], early_stopping=False,
                                           return_dict_in_generate=True, output_scores=True, suppress_tokens=sup_ids,
                                           begin_suppress_tokens=None, forced_decoder_ids=None)
        seq = out.sequences[:, len(PROMPT):]
        lens = [r.index(c['eos']) + 1 if c['eos'] in r else len(r) for r in seq.tolist()]
        sup = torch.zeros(V)
        sup[sup_ids] = float('-inf')
        o_seq, o_sc, _ = onn.whisper_beam(sdb, bmel, bprompt, c['n_new'], 8, c['beams'], c['eos'], c['lp'], suppress=sup)
        same = all(o_seq[b] == seq[b, :lens[b]].tolist() for b in range(len(bseeds)))
        print('base beam case', ci, c, 'lens', lens, 'tokens equal', same, 'score diff',
              float((o_sc - out.sequences_scores).abs().max()))
        assert same
        arrays['base_seq%d' % ci] = seq.numpy().astype(np.int32)
        arrays['base_len%d' % ci] = np.asarray(lens, np.int32)
        arrays['base_score%d' % ci] = out.sequences_scores.numpy()
        bmeta.append(c)
        if ci == 0:
            bcases.append(dict(beams=5, eos=int(seq[0, 10]), n_new=32, lp=1.0, begin=[]))
        ci += 1
    np.savez_compressed(os.path.join(GOLD, 'whisper_beam.npz'), **arrays)
    json.dump({'source': 'transformers 5.15.0 GenerationMixin.generate(num_beams=K, do_sample=False, early_stopping=False) '
                         'on WhisperForConditionalGeneration(WhisperConfig()) holding synth_state_dict(whisper_tiny, 0); '
                         'ids >= 50257 other than eos suppressed (suppress_tokens), `begin` = begin_suppress_tokens; '
                         '`base_*`: the same on synth_state_dict(whisper_base, 1) (d 512, 6+6 layers, 8 heads)',
               'prompt': PROMPT, 'audio_seeds': [1000, 1001], 'cases': meta,
               'base': {'weights_seed': 1, 'audio_seeds': bseeds, 'nheads': 8, 'cases': bmeta}},
              open(os.path.join(GOLD, 'whisper_beam_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote whisper_beam.npz')


def _hf_qwen2(family, seed):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    from infernos_amd.weights import synth_state_dict, QWEN2_CONFIGS
    c = QWEN2_CONFIGS[family]
    hc = Qwen2Config(vocab_size=c['vocab'], hidden_size=c['hidden'], intermediate_size=c['ffn'], num_hidden_layers=c['layers'],
                     num_attention_heads=c['heads'], num_key_value_heads=c['kv_heads'], head_dim=c['head_dim'],
                     max_position_embeddings=c['max_pos'], rms_norm_eps=c['rms_eps'], tie_word_embeddings=c['tie'],
                     rope_parameters={'rope_theta': c['rope_theta'], 'rope_type': 'default'}, use_sliding_window=False,
                     attention_dropout=0.0, pad_token_id=0, bos_token_id=None, eos_token_id=None)
    model = Qwen2ForCausalLM(hc)
    sd = synth_state_dict(family, seed)
    full = dict(sd)
    if c['tie']:
        full['lm_head.weight'] = sd['model.embed_tokens.weight']
    model.load_state_dict(full, strict=True)
    model.eval()
    return model, sd, c


def gen_qwen2():
    """Qwen2 fixtures: transformers' Qwen2ForCausalLM holding the seeded weights, called the way
    InfernLLMWorker.process_batch does (Cluster/InfernLLMWorker.py:108-118: a padded batch + attention mask through
    generate); the state-dict schema of the parity-test configurations goes to nn_schema.json."""
    import json
    from oracle import nn as onn
    arrays, meta = {}, {}
    schema_path = os.path.join(ROOT, 'infernos_amd', 'nn_schema.json')
    schema = json.load(open(schema_path))
    for family, seed, prompts in (('qwen2_tiny', 0, [[11, 22, 33, 44, 55, 66, 77], [5, 9, 2], [901, 17, 4, 4, 250]]),
                                  ('qwen2_tiny64', 1, [[3, 1, 4, 1, 5, 9, 2, 6], [700, 2], [10, 20, 30, 40, 50, 60]])):
        model, sd, c = _hf_qwen2(family, seed)
        schema[family] = {k: [list(v.shape), 'float32'] for k, v in model.state_dict().items()
                          if not (c['tie'] and k == 'lm_head.weight')}
        T = max(len(p) for p in prompts)
        ids = torch.zeros((len(prompts), T), dtype=torch.long)
        mask = torch.zeros((len(prompts), T), dtype=torch.long)
        for i, p in enumerate(prompts):                        # left padding: generation continues from the last real token
            ids[i, T - len(p):] = torch.tensor(p)
            mask[i, T - len(p):] = 1
        n_new = 10
        with torch.no_grad():
            out = model.generate(input_ids=ids, attention_mask=mask, max_new_tokens=n_new, do_sample=False,
                                 return_dict_in_generate=True, output_logits=True, pad_token_id=0)
            fwd = model(input_ids=ids, attention_mask=mask).logits
        gen = out.sequences[:, T:]
        step_logits = torch.stack(out.logits, 1)                # [B, n_new, V]
        o_new, o_logs = onn.qwen2_greedy(sd, c, prompts, n_new)
        ok = all(o_new[i] == gen[i].tolist() for i in range(len(prompts)))
        dl = max(float((o_logs[i][len(p) - 1:] - step_logits[i]).abs().max()) for i, p in enumerate(prompts))
        dp = max(float((o_logs[i][:len(p)] - fwd[i, T - len(p):]).abs().max()) for i, p in enumerate(prompts))
        print(family, 'greedy equal', ok, 'max |logit diff| generated', dl, 'prompt', dp, 'logit std', float(fwd.std()))
        arrays[family + '_gen'] = gen.numpy().astype(np.int32)
        arrays[family + '_step_logits'] = step_logits[:, :, ::7].numpy()
        for i, p in enumerate(prompts):
            arrays['%s_prompt_logits%d' % (family, i)] = fwd[i, T - len(p):, ::7].numpy()
        # the same engine in the reference's dtype (torch_dtype="auto" -> bf16 checkpoints): its distance from fp32 is the
        # bar the device engine is held to
        with torch.no_grad():
            fwd16 = model.to(torch.bfloat16)(input_ids=ids, attention_mask=mask).logits.float()
        model.to(torch.float32)
        rel16 = max(float((fwd16[i, T - len(p):] - fwd[i, T - len(p):]).norm() / fwd[i, T - len(p):].norm())
                    for i, p in enumerate(prompts))
        print(family, 'transformers bf16 vs fp32 prompt logits rel-L2', rel16)
        meta[family] = {'seed': seed, 'prompts': prompts, 'n_new': n_new, 'hf_bf16_rel_l2': rel16}
    json.dump(schema, open(schema_path, 'w'), sort_keys=True)
    json.dump(schema, open(os.path.join(GOLD, 'nn_schema.json'), 'w'), sort_keys=True)
    meta['source'] = ('transformers 5.15.0 Qwen2ForCausalLM holding synth_state_dict weights; left-padded batch + attention_mask '
                      'through forward and generate(do_sample=False), fp32')
    np.savez_compressed(os.path.join(GOLD, 'qwen2.npz'), **arrays)
    json.dump(meta, open(os.path.join(GOLD, 'qwen2_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote qwen2.npz')


def gen_llm_host():
    """Host-side fixtures of the LLM row, produced by the REFERENCE's own classes: LLMSession (Cluster/LLMSession.py),
    ResultsStreamer fed scripted token streams, and InfernLLMWorker.process_batch (Cluster/InfernLLMWorker.py:104-118)
    driving transformers' Qwen2ForCausalLM (seeded qwen2_tiny64) with the stand-in chat tokenizer."""
    import json
    from Cluster.LLMSession import LLMSession, LLMSessionParams, LLMRequest, LLMInferRequest
    import Cluster.InfernLLMWorker as ref
    from infernos_amd.synth import CharChatTokenizer
    out = {}
    # ---- 1. session transcript
    class FakeLLM:
        def __init__(self):
            self.q = []

        def infer(self, ireq):
            self.q.append(ireq)
    llm = FakeLLM()
    sess = LLMSession(llm, LLMSessionParams('You are an attendant.'))
    log = []
    snap = lambda tag: log.append([tag, json.loads(json.dumps(sess.context))])
    got = []
    snap('init')
    sess.context_add('<Incoming call>')
    snap('ctx user')
    sess.context_add('second user line')
    snap('ctx user again')
    r1 = LLMRequest('Hello, who is this?', lambda result: got.append(['r1', result.text]))
    sess.textin(r1)
    snap('textin r1')
    q1 = [list(llm.q[-1].context)]
    llm.q[-1].textout_cb(result=ref.LLMResult('This is the attendant.', r1.id))
    snap('textout r1')
    llm.q[-1].textout_cb(result=ref.LLMResult('How can I help?', r1.id))
    snap('textout r1 again')
    r2 = LLMRequest('I need a taxi', lambda result: got.append(['r2', result.text]))
    r2.auto_ctx_add = False
    sess.textin(r2)
    snap('textin r2')
    llm.q[-1].textout_cb(result=ref.LLMResult('Calling one.', r2.id))
    snap('textout r2 (no auto add)')
    sess.context_add('Calling one.', 'assistant')
    sess.context_add('<sentence interrupted>', 'user')
    snap('manual adds')
    out['session'] = {'log': log, 'delivered': got, 'queued_first_context': q1}
    # ---- 2. streamer on scripted token streams
    tok = CharChatTokenizer(777)

    class Upper:
        llm_tokenizer = tok
    g = torch.Generator().manual_seed(7)
    B, steps = 3, 90
    stream = torch.randint(4, 777, (steps, B), generator=g)
    stop = [90, 41, 64]
    for b in range(B):
        stream[stop[b]:, b] = tok.pad_token_id
    calls = []
    wis = []
    for b in range(B):
        ir = LLMInferRequest(LLMRequest('x', None), [{}])
        ir.textout_cb = (lambda result, b=b: calls.append([len(seen), b, result.text]))
        wis.append(ir)
    seen = []
    st = ref.ResultsStreamer(wis, Upper())
    st.put(torch.zeros((B, 5), dtype=torch.long))
    for s in range(steps):
        seen.append(s)
        st.put(stream[s])
    seen.append('end')
    st.end()
    out['streamer'] = {'tokens': stream.tolist(), 'calls': calls}
    # ---- 3. the reference worker's process_batch on the seeded model
    model, sd, c = _hf_qwen2('qwen2_tiny64', 1)
    contexts = [[{'role': 'system', 'content': 'You are an attendant.'}, {'role': 'user', 'content': 'Hello? Who is this.'}],
                [{'role': 'system', 'content': 'Short.'}, {'role': 'user', 'content': 'Hi'}],
                [{'role': 'system', 'content': 'You are an attendant at a hotel.'}, {'role': 'user', 'content': 'I need a room, please!'},
                 {'role': 'assistant', 'content': 'Sure.'}, {'role': 'user', 'content': 'For two nights'}]]
    msgs = [tok.apply_chat_template(cx, tokenize=False, add_generation_prompt=True) for cx in contexts]
    enc = tok(msgs, return_tensors='pt', padding=True)
    with torch.no_grad():
        free = model.generate(**enc, max_new_tokens=80, do_sample=False, pad_token_id=tok.pad_token_id)[:, enc['input_ids'].size(1):]
    eos = sorted({int(free[0, 70]), int(free[1, 33]), int(free[2, 52])})
    model.generation_config.eos_token_id = eos
    model.generation_config.pad_token_id = tok.pad_token_id
    w = object.__new__(ref.InfernLLMWorker)
    from Cluster.InfernBatchedWorker import InfernBatchedWorker
    InfernBatchedWorker.__init__(w)
    w.llm_model, w.llm_tokenizer, w.debug = model, tok, False
    if not hasattr(torch, 'xpu') or not hasattr(torch.xpu, 'synchronize'):
        raise RuntimeError('torch.xpu missing')
    real_sync = torch.xpu.synchronize
    torch.xpu.synchronize = lambda *a, **k: None
    calls, puts = [], []
    real_put = ref.ResultsStreamer.put

    def logging_put(self, token_ids):
        puts.append(token_ids.tolist())
        return real_put(self, token_ids)
    ref.ResultsStreamer.put = logging_put
    wis = []
    for b, cx in enumerate(contexts):
        ir = LLMInferRequest(LLMRequest('x', None), cx)
        ir.textout_cb = (lambda result, b=b: calls.append([b, result.text]))
        wis.append(ir)
    try:
        w.process_batch(wis)
    finally:
        torch.xpu.synchronize = real_sync
        ref.ResultsStreamer.put = real_put
    gen_steps = puts[1:]
    print('llm worker: eos', eos, 'steps', len(gen_steps), 'calls', calls)
    out['worker'] = {'contexts': contexts, 'eos': eos, 'prompt_ids': puts[0], 'step_tokens': gen_steps, 'calls': calls,
                     'family': 'qwen2_tiny64', 'seed': 1}
    out['source'] = ('Cluster/LLMSession.py:34-70, Cluster/InfernLLMWorker.py:15-66 (ResultsStreamer) and :104-118 (process_batch) '
                     'run in place on transformers 5.15.0; tokenizer = infernos_amd.synth.CharChatTokenizer')
    json.dump(out, open(os.path.join(GOLD, 'llm_host.json'), 'w'), indent=0, sort_keys=True)
    print('wrote llm_host.json')


def gen_sampling():
    """transformers' own logits processors / warpers (the chain generate builds for do_sample with Qwen2.5-Instruct's
    generation config) on seeded rows: the distribution the draw is made from."""
    import json
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                         TopKLogitsWarper, TopPLogitsWarper)
    from oracle import nn as onn
    g = torch.Generator().manual_seed(21)
    V, R = 5003, 6
    logits = torch.randn(R, V, generator=g) * 4
    hist = torch.randint(0, V, (R, 40), generator=g)
    cases = [dict(penalty=1.05, temperature=0.7, top_k=20, top_p=0.8), dict(penalty=1.0, temperature=1.0, top_k=5, top_p=1.0),
             dict(penalty=1.3, temperature=0.3, top_k=32, top_p=0.95), dict(penalty=1.0, temperature=1.5, top_k=20, top_p=0.5)]
    arrays = {'logits': logits.numpy(), 'history': hist.numpy().astype(np.int32)}
    for ci, c in enumerate(cases):
        s = RepetitionPenaltyLogitsProcessor(c['penalty'])(hist, logits.clone()) if c['penalty'] != 1.0 else logits.clone()
        s = TemperatureLogitsWarper(c['temperature'])(hist, s)
        s = TopKLogitsWarper(c['top_k'])(hist, s)
        if c['top_p'] < 1.0:
            s = TopPLogitsWarper(c['top_p'])(hist, s)
        p = torch.softmax(s, -1)
        ids = torch.full((R, 32), -1, dtype=torch.int32)
        pr = torch.zeros(R, 32)
        for r in range(R):
            o_ids, o_p = onn.sample_warp(logits[r], hist[r].tolist(), c['penalty'], c['temperature'], c['top_k'], c['top_p'])
            order = torch.argsort(p[r], descending=True, stable=True)
            n = int((p[r] > 0).sum())
            ids[r, :n] = order[:n].int()
            pr[r, :n] = p[r][order[:n]]
            assert o_ids.tolist() == order[:n].tolist() and float((o_p - pr[r, :n]).abs().max()) < 1e-6, (ci, r)
        arrays['ids%d' % ci] = ids.numpy()
        arrays['probs%d' % ci] = pr.numpy()
        print('sampling case', ci, c, 'kept', [(int((ids[r] >= 0).sum())) for r in range(R)])
    np.savez_compressed(os.path.join(GOLD, 'sampling.npz'), **arrays)
    json.dump({'cases': cases, 'source': 'transformers 5.15.0 generation/logits_process.py processors applied in generate\'s order'},
              open(os.path.join(GOLD, 'sampling_meta.json'), 'w'), indent=1, sort_keys=True)
    print('wrote sampling.npz')


SECTIONS = {'tts': gen_tts, 'whisper': gen_whisper, 'whisper_tf': gen_whisper_tf, 'whisper_beam': gen_whisper_beam, 'qwen2': gen_qwen2, 'llm_host': gen_llm_host, 'sampling': gen_sampling}
