"""CPU: the neural oracle (oracle/nn.py) against fixtures produced by the reference's own
HelloSippyRTPipe.infer()/unbatch_and_dispatch() and InfernSTTWorker.process_batch() runs."""
import json
import os

import numpy as np
import pytest
import torch

from infernos_amd.weights import synth_state_dict
from oracle import nn as onn


@pytest.fixture(scope='module')
def tts(golden_dir):
    return (np.load(os.path.join(golden_dir, 'tts.npz')), json.load(open(os.path.join(golden_dir, 'tts_meta.json'))))


def tts_inputs(meta):
    ids = [torch.tensor([[int(t) for t in s.split()]]) for s in meta['texts']]
    T = max(i.size(1) for i in ids)
    inp = torch.cat([torch.nn.functional.pad(i, (0, T - i.size(1))) for i in ids])
    msk = torch.cat([torch.nn.functional.pad(torch.ones_like(i), (0, T - i.size(1))) for i in ids]).int()
    g = torch.Generator().manual_seed(meta['speaker_seed'])
    spk = torch.cat([torch.randn(1, 512, generator=g) for _ in ids])
    return inp, msk, spk


def test_tts_infer_matches_reference_run(tts):
    g, meta = tts
    sd = synth_state_dict('speecht5_tts', 0, stop_bias=-20.0)
    voc, amd = synth_state_dict('hifigan', 0), synth_state_dict('amendment', 0)
    inp, msk, spk = tts_inputs(meta)
    with torch.no_grad():
        st = onn.TTSState(sd, inp, msk, spk)
        assert st.maxlen == meta['A']['maxlen']
        np.testing.assert_allclose(st.enc[:, :, :16].numpy(), g['A_enc_slice'], atol=2e-5)
        masks = np.unpackbits(g['A_masks'], axis=-1)
        for c in range(2):
            stages = {}
            a = onn.tts_infer(sd, voc, amd, st, torch.from_numpy(masks[c]), stages=stages)
            np.testing.assert_allclose(a[:, ::8].numpy(), g['A_audio_%d' % c], atol=5e-6)
            if c == 0:
                np.testing.assert_allclose(stages['postnet'].numpy(), g['A_postnet_0'], atol=5e-5)
                np.testing.assert_allclose(stages['vocoder'][:, ::16].numpy(), g['A_vocoder_0'], atol=2e-5)
            bk = meta['A']['book'][c]
            assert (st.idx, st.starts_at.tolist(), st.ends_at.tolist()) == (bk['idx'], bk['starts_at'], bk['ends_at'])
            # the reference's own bf16 run sits this far from fp32 -- the bar for the bf16 GPU path
            rel = np.linalg.norm(g['A_audio_bf16_%d' % c] - g['A_audio_%d' % c]) / np.linalg.norm(g['A_audio_%d' % c])
            assert 1e-3 < rel < 0.1


def test_tts_dispatch_offsets_match_reference(tts):
    g, meta = tts
    for key in ('A', 'B'):
        m = meta[key]
        lens = [[] for _ in m['book'][0]['starts_at']]
        live = [True] * len(lens)
        for bk, more in zip(m['book'], m['more']):
            offs, mo = onn.tts_dispatch_offsets(bk['idx'], bk['starts_at'], bk['ends_at'])
            assert mo == more
            for i, (s, e, fin) in enumerate(offs):
                if not live[i]:
                    continue
                assert s <= e
                if s != e:
                    lens[i].append(e - s)
                if fin:
                    lens[i].append(None)
                    live[i] = False
        assert lens == m['dispatch_lens'], key


def test_tts_stop_rule_scenario_b(tts):
    g, meta = tts
    sd = synth_state_dict('speecht5_tts', 0)
    voc, amd = synth_state_dict('hifigan', 0), synth_state_dict('amendment', 0)
    inp, msk, spk = tts_inputs(meta)
    masks = np.unpackbits(g['B_masks'], axis=-1)
    with torch.no_grad():
        st = onn.TTSState(sd, inp, msk, spk)
        a = onn.tts_infer(sd, voc, amd, st, torch.from_numpy(masks[0]))
    bk = meta['B']['book'][0]
    assert st.ends_at.tolist() == bk['ends_at']
    offs, _ = onn.tts_dispatch_offsets(st.idx, st.starts_at.tolist(), st.ends_at.tolist())
    for i, (s, e, fin) in enumerate(offs):
        if s != e:
            np.testing.assert_allclose(a[i, s:e][::8].numpy(), g['B_disp_%d_0' % i], atol=5e-6)


def test_whisper_matches_reference_run(golden_dir):
    from oracle import dsp
    from infernos_amd.synth import synth_utterance
    g = np.load(os.path.join(golden_dir, 'whisper.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'whisper_meta.json')))
    sd = synth_state_dict('whisper_tiny', 0)
    auds = [dsp.resample(synth_utterance(s, 10.0), 8000, 16000) for s in meta['audio_seeds']]
    mel = torch.from_numpy(dsp.logmel(np.stack(auds)))
    prompt = torch.tensor([meta['prompt']] * 2)
    with torch.no_grad():
        toks, first, l0, enc = onn.whisper_greedy(sd, mel, prompt, 8, 6)
    np.testing.assert_allclose(enc[:, ::25, :32].numpy(), g['enc_slice'], atol=3e-4)
    np.testing.assert_allclose(first[:, ::97].numpy(), g['first_logits_slice'], atol=2e-3)
    np.testing.assert_allclose(l0[:, ::97].numpy(), g['logits0_slice'], atol=2e-3)
    assert np.array_equal(toks.numpy(), g['greedy'])
    nsp = torch.softmax(l0, -1)[:, meta['no_speech_id']].tolist()
    np.testing.assert_allclose(nsp, meta['no_speech_prob'], rtol=1e-2)


def test_beam_search_matches_transformers(golden_dir):
    """oracle/nn.py:beam_search against transformers' GenerationMixin beam search on the seeded whisper_tiny
    (tools/gen_golden_nn.py:gen_whisper_beam): 13 cases over beams, eos id, length, length_penalty, begin-suppression,
    two of them ending early on a finished hypothesis."""
    from oracle import dsp
    from infernos_amd.synth import synth_utterance
    g = np.load(os.path.join(golden_dir, 'whisper_beam.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'whisper_beam_meta.json')))
    sd = synth_state_dict('whisper_tiny', 0)
    auds = [dsp.resample(synth_utterance(s, 10.0), 8000, 16000) for s in meta['audio_seeds']]
    mel = torch.from_numpy(dsp.logmel(np.stack(auds)))
    prompt = torch.tensor([meta['prompt']] * 2)
    V = 51865
    early = 0
    for ci, c in enumerate(meta['cases']):
        sup = torch.zeros(V)
        sup[50257:] = float('-inf')
        sup[c['eos']] = 0.0
        bs = None
        if c['begin']:
            bs = torch.zeros(V)
            bs[c['begin']] = float('-inf')
        with torch.no_grad():
            seqs, scores, _ = onn.whisper_beam(sd, mel, prompt, c['n_new'], 6, c['beams'], c['eos'], c['lp'], suppress=sup,
                                               begin_suppress=bs)
        lens = g['len%d' % ci]
        for b in range(2):
            assert seqs[b] == g['seq%d' % ci][b, :lens[b]].tolist(), (ci, b)
            early += int(lens[b] < c['n_new'])
        np.testing.assert_allclose(scores.numpy(), g['score%d' % ci], atol=2e-4)
    assert early >= 2


def test_sampling_distribution_matches_transformers_warpers(golden_dir):
    """oracle/nn.py:sample_warp against transformers' RepetitionPenalty / Temperature / TopK / TopP chain
    (tools/gen_golden_nn.py:gen_sampling)"""
    g = np.load(os.path.join(golden_dir, 'sampling.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'sampling_meta.json')))
    logits, hist = torch.from_numpy(g['logits']), g['history']
    for ci, c in enumerate(meta['cases']):
        for r in range(logits.size(0)):
            ids, p = onn.sample_warp(logits[r], hist[r].tolist(), c['penalty'], c['temperature'], c['top_k'], c['top_p'])
            n = int((g['ids%d' % ci][r] >= 0).sum())
            assert ids.tolist() == g['ids%d' % ci][r, :n].tolist()
            np.testing.assert_allclose(p.numpy(), g['probs%d' % ci][r, :n], atol=1e-6)
            assert onn.sample_pick(ids, p, 0.0) == int(ids[0]) and onn.sample_pick(ids, p, 0.999999) == int(ids[-1])


def test_qwen2_bf16_restatement_reproduces_the_fixture_bf16_error(golden_dir):
    """oracle.nn.qwen2_forward with bfloat16 weights follows transformers' Qwen2 modules run in that dtype; its error against the fp32
    run must be the one the transformers fixture recorded for the same seeded weights (hf_bf16_rel_l2, tools/gen_golden_nn.py) -- that
    makes it usable as "the reference engine in bf16" at depths no fixture reaches (tests/test_llm_gpu.py: the 28-layer configuration-5
    share measures its bar with it)."""
    import json
    import os
    from infernos_amd.weights import synth_state_dict, QWEN2_CONFIGS
    meta = json.load(open(os.path.join(golden_dir, 'qwen2_meta.json')))
    for fam in ('qwen2_tiny', 'qwen2_tiny64'):
        m, cfg = meta[fam], QWEN2_CONFIGS[fam]
        sd = synth_state_dict(fam, m['seed'])
        sd16 = onn._cast(sd, torch.bfloat16)
        es = []
        with torch.no_grad():
            for p in m['prompts']:
                a = onn.qwen2_forward(sd, cfg, torch.tensor([p]), 0, [{} for _ in range(cfg['layers'])])[0]
                b = onn.qwen2_forward(sd16, cfg, torch.tensor([p]), 0, [{} for _ in range(cfg['layers'])])[0]
                es.append(float((b.double() - a.double()).norm() / a.double().norm()))
        assert 0.75 * m['hf_bf16_rel_l2'] < max(es) < 1.25 * m['hf_bf16_rel_l2'], (fam, es, m['hf_bf16_rel_l2'])
