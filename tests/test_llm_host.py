"""CPU: host logic of the LLM row against transcripts produced by the reference's own classes
(tests/golden/llm_host.json, tools/gen_golden_nn.py:gen_llm_host): LLMSession context handling, the sentence-boundary
ResultsStreamer on scripted token streams, and the worker's process_batch glue (chat template -> padded tokenisation ->
generate -> streamer protocol) with the engine replaced by a replay of the tokens the reference's run generated."""
import json
import os

import pytest
import torch

from infernos_amd.llm import (InfernLLMWorker, LLMInferRequest, LLMRequest, LLMResult, LLMSession, LLMSessionParams,
                              ResultsStreamer)
from infernos_amd.synth import CharChatTokenizer


@pytest.fixture(scope='module')
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'llm_host.json')))


def test_llm_session_transcript(gold):
    class FakeLLM:
        def __init__(self):
            self.q = []

        def infer(self, ireq):
            self.q.append(ireq)
    llm = FakeLLM()
    sess = LLMSession(llm, LLMSessionParams('You are an attendant.'))
    log, got = [], []
    snap = lambda tag: log.append([tag, json.loads(json.dumps(sess.context))])
    snap('init')
    sess.context_add('<Incoming call>')
    snap('ctx user')
    sess.context_add('second user line')
    snap('ctx user again')
    r1 = LLMRequest('Hello, who is this?', lambda result: got.append(['r1', result.text]))
    sess.textin(r1)
    snap('textin r1')
    assert isinstance(llm.q[-1], LLMInferRequest) and llm.q[-1].req is r1
    q1 = [list(llm.q[-1].context)]
    llm.q[-1].textout_cb(result=LLMResult('This is the attendant.', r1.id))
    snap('textout r1')
    llm.q[-1].textout_cb(result=LLMResult('How can I help?', r1.id))
    snap('textout r1 again')
    r2 = LLMRequest('I need a taxi', lambda result: got.append(['r2', result.text]))
    r2.auto_ctx_add = False
    sess.textin(r2)
    snap('textin r2')
    llm.q[-1].textout_cb(result=LLMResult('Calling one.', r2.id))
    snap('textout r2 (no auto add)')
    sess.context_add('Calling one.', 'assistant')
    sess.context_add('<sentence interrupted>', 'user')
    snap('manual adds')
    assert log == gold['session']['log']
    assert got == gold['session']['delivered']
    assert q1 == gold['session']['queued_first_context']
    sess.stop()
    assert not hasattr(sess, 'llm')


def _wis(n, calls, seen=None):
    wis = []
    for b in range(n):
        ir = LLMInferRequest(LLMRequest('x', None), [{}])
        ir.textout_cb = (lambda result, b=b: calls.append(([len(seen)] if seen is not None else []) + [b, result.text]))
        wis.append(ir)
    return wis


def test_results_streamer_on_scripted_streams(gold):
    tok = CharChatTokenizer(777)

    class Upper:
        llm_tokenizer = tok
    stream = torch.tensor(gold['streamer']['tokens'])
    calls, seen = [], []
    wis = _wis(stream.size(1), calls, seen)
    st = ResultsStreamer(wis, Upper())
    st.put(torch.zeros((stream.size(1), 5), dtype=torch.long))
    for s in range(stream.size(0)):
        seen.append(s)
        st.put(stream[s])
    seen.append('end')
    st.end()
    assert calls == gold['streamer']['calls']
    assert len(calls) > 6 and any(c[0] < stream.size(0) for c in calls)          # mid-stream deliveries happened
    assert all(isinstance(c[2], str) for c in calls)


class ReplayEngine:
    """stands in for engines/qwen2.Qwen2: checks the prompts it is handed and replays the reference run's tokens"""
    max_tokens = 4096

    def __init__(self, w):
        self.w = w
        self.seen = None

    def generate(self, prompts, max_new_tokens, eos_ids=(), pad_id=0, on_tokens=None, sampler=None):
        self.seen = dict(prompts=prompts, max_new_tokens=max_new_tokens, eos=tuple(eos_ids), pad=pad_id)
        for step in self.w['step_tokens']:
            on_tokens(torch.tensor(step, dtype=torch.long))
        return None, []


def test_worker_process_batch_glue_replays_reference_run(gold):
    w = gold['worker']
    tok = CharChatTokenizer(777)
    worker = object.__new__(InfernLLMWorker)             # no device: only the host glue is under test
    worker.llm_tokenizer, worker.debug = tok, False
    worker.llm_model = eng = ReplayEngine(w)
    worker.max_new_tokens, worker.eos_token_ids, worker.pad_token_id, worker.sampler = 16 * 1024, tuple(w['eos']), 0, None
    import contextlib
    worker.device = None
    calls = []
    wis = []
    for b, cx in enumerate(w['contexts']):
        ir = LLMInferRequest(LLMRequest('x', None), cx)
        ir.textout_cb = (lambda result, b=b: calls.append([b, result.text]))
        wis.append(ir)
    ids, prompts = worker.tokenize_batch(wis)
    assert ids.tolist() == w['prompt_ids']                                   # same padded batch as the reference built
    assert prompts == [[t for t in row if t != 0] for row in w['prompt_ids']]
    orig = torch.cuda.device
    torch.cuda.device = lambda d: contextlib.nullcontext()
    try:
        worker.process_batch(wis)
    finally:
        torch.cuda.device = orig
    assert calls == w['calls']
    assert eng.seen['prompts'] == prompts and eng.seen['eos'] == tuple(w['eos']) and eng.seen['max_new_tokens'] == 16 * 1024
    assert all(isinstance(c[1], str) for c in calls)


def test_compat_registers_llm_modules():
    import sys
    from infernos_amd import compat
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split('.')[0] in ('Cluster', 'Core', 'HelloSippyTTSRT', 'safetorch', 'config', 'rtpsynth')}
    try:
        compat.install()
        from Cluster.LLMSession import LLMSession as A
        from Cluster.InfernLLMWorker import InfernLLMWorker as B, ResultsStreamer as C
        assert A is LLMSession and B is InfernLLMWorker and C is ResultsStreamer
    finally:
        for k in [k for k in sys.modules if k.split('.')[0] in ('Cluster', 'Core', 'HelloSippyTTSRT', 'safetorch', 'config', 'rtpsynth')]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})
