"""LLM stage of the AI-attendant path: session objects, the sentence-boundary streamer and the batched worker.

Interfaces of Cluster/LLMSession.py:5-70 (LLMRequest, LLMResult, LLMInferRequest, LLMSessionParams, LLMSession) and
Cluster/InfernLLMWorker.py:15-119 (ResultsStreamer, InfernLLMWorker).  The reference worker loads
Qwen/Qwen2.5-14B-Instruct 4-bit through ipex-llm and calls transformers' generate with the streamer; here the model is
engines/qwen2.py on the HIP device (bf16), and the generate loop hands the streamer the same sequence of `put` calls:
the prompt once, then one int64 [B] tensor of tokens per step (pad for rows that have stopped), then `end`.
"""
from functools import partial
from time import monotonic
from typing import List, Tuple
from uuid import uuid4

import torch

from . import _lib
from .workers import InfernBatchedWorker


class LLMRequest:
    """text for the model + where the answer's pieces go (`textout_cb(result=LLMResult)`); `auto_ctx_add = False` keeps the
    answer out of the session's context (the attendant adds what was actually spoken itself)"""
    auto_ctx_add = True

    def __init__(self, text, textout_cb):
        self.id = uuid4()
        self.text = text
        self.textout_cb = textout_cb


class LLMResult:
    """one delivered piece of an answer, tagged with the request it belongs to"""

    def __init__(self, text, req_id):
        self.req_id = req_id
        self.text = text


class LLMInferRequest:
    """the worker's queue item: the request, a shallow snapshot of the session context (tuple of the same message dicts), and
    -- set by the session -- `textout_cb`"""

    def __init__(self, req, context):
        self.context = tuple(context)
        self.req = req


class LLMSessionParams:
    def __init__(self, system_prompt):
        self.system_prompt = system_prompt


class LLMSession():
    """Per-call chat context (Cluster/LLMSession.py:34-70).  Behaviour, pinned by tests/golden/llm_host.json: a message of the
    role the context already ends with is appended to that message after a space, anything else opens a new message; a
    request is queued with a SHALLOW snapshot of the context (the worker sees later additions to the last message); every
    piece of the answer is added as 'assistant' unless the request opted out (auto_ctx_add), then handed to the caller."""
    debug = False

    def __init__(self, llm, params):
        self.id = uuid4()
        self.llm = llm
        self.context = [dict(role='system', content=params.system_prompt)]

    def context_add(self, content, role='user'):
        tail = self.context[-1] if self.context else None
        if tail is None or tail['role'] != role:
            self.context.append(dict(role=role, content=content))
        else:
            tail['content'] = '%s %s' % (tail['content'], content)
        if self.debug:
            print('%4.3f: LLMSession.context_add -> %r' % (monotonic(), self.context))

    def textin(self, req):
        self.context_add(req.text)
        work = LLMInferRequest(req, self.context)
        work.textout_cb = partial(self.textout, req=req)
        if hasattr(req, '_proc_start_cb'):
            work._proc_start_cb = req._proc_start_cb
        self.llm.infer(work)

    def textout(self, req, result):
        if req.auto_ctx_add:
            self.context_add(result.text, 'assistant')
        req.textout_cb(result=result)

    def stop(self):
        del self.llm


class ResultsStreamer:
    """Sentence-boundary streaming of a batch while it is generated (Cluster/InfernLLMWorker.py:15-66; the protocol is
    transformers' streamer: `put(prompt ids)` once, `put(int64 [B])` per generated token, `end()`).  Behaviour, pinned by
    tests/golden/llm_host.json: the whole batch is re-decoded after every token except when the token count is a multiple of
    `decode_batch_size`; of a row's not yet delivered text, the part in front of the LAST occurrence of the FIRST marker of
    `sync_on` that occurs at all -- minus that marker's final character -- is delivered if it is at least 10 characters
    long, and delivery resumes behind the marker; `end()` delivers whatever is left."""
    debug = False
    sync_on = ('. ', '? ', '! ', '\n')
    decode_batch_size = 8

    def __init__(self, wis: List[LLMInferRequest], upper: 'InfernLLMWorker'):
        self.tokenizer = upper.llm_tokenizer
        self.batch_decode = partial(upper.llm_tokenizer.batch_decode, skip_special_tokens=True)
        self.wi_cbs = tuple(wi.textout_cb for wi in wis)
        self.newLLMResult = tuple(partial(LLMResult, req_id=wi.req.id) for wi in wis)
        self.oposs = [0] * len(wis)            # per row: characters already delivered
        self.current_tokens = None

    def _boundary(self, text: str) -> int:
        """offset just behind the marker the delivery cuts at, or -1"""
        for mark in self.sync_on:
            at = text.rfind(mark)
            if at >= 0:
                return at + len(mark)
        return -1

    def put(self, token_ids):
        if self.current_tokens is None:        # the prompt: only the batch size matters
            self.current_tokens = torch.zeros((token_ids.shape[0], 0), dtype=torch.long)
            return
        step = token_ids if token_ids.dim() == 2 else token_ids.unsqueeze(1)
        self.current_tokens = torch.cat([self.current_tokens, step], dim=1)
        if self.current_tokens.shape[1] % self.decode_batch_size == 0:
            return
        for row, text in enumerate(self.batch_decode(self.current_tokens)):
            start = self.oposs[row]
            fresh = text[start:]
            cut = self._boundary(fresh) if fresh else -1
            if cut < 0 or cut - 1 < 10:
                continue
            self.wi_cbs[row](result=self.newLLMResult[row](fresh[:cut - 1]))
            self.oposs[row] = start + cut

    def end(self):
        for row, text in enumerate(self.batch_decode(self.current_tokens)):
            rest = text[self.oposs[row]:]
            if rest:
                self.wi_cbs[row](result=self.newLLMResult[row](rest))
        del self.current_tokens
        del self.wi_cbs


def qwen2_config_from_hf(hc):
    """transformers Qwen2Config -> the engine's configuration dict (infernos_amd.weights.QWEN2_CONFIGS layout)"""
    rope = getattr(hc, 'rope_parameters', None) or {}
    return dict(vocab=hc.vocab_size, hidden=hc.hidden_size, ffn=hc.intermediate_size, layers=hc.num_hidden_layers,
                heads=hc.num_attention_heads, kv_heads=hc.num_key_value_heads,
                head_dim=getattr(hc, 'head_dim', None) or hc.hidden_size // hc.num_attention_heads,
                rope_theta=float(rope.get('rope_theta', getattr(hc, 'rope_theta', 1.0e6))), rms_eps=hc.rms_norm_eps,
                tie=bool(hc.tie_word_embeddings), max_pos=hc.max_position_embeddings)


class InfernLLMWorker(InfernBatchedWorker):
    """`infer(LLMInferRequest)`; every request's textout_cb receives LLMResult pieces from the worker thread as the
    batch is generated, then the remainder when it ends (InfernLLMWorker.py:104-118).

    Constructor extras (optional): `weights` (HF-format Qwen2 state dict) + `config` (QWEN2_CONFIGS-style dict; default:
    download `model_name`), `tokenizer` (needs apply_chat_template / __call__(padding=True) / batch_decode /
    eos_token_id / pad_token_id; default AutoTokenizer.from_pretrained), `max_new_tokens` (the reference passes
    16 * 1024), `max_tokens` (KV-cache positions per row), `eos_token_ids`, `generation_config` (dict: do_sample,
    temperature, top_k, top_p, repetition_penalty -- what transformers' generate takes from the checkpoint; read from
    the checkpoint when the weights are downloaded, greedy when weights are passed without it).  max_batch_size is the
    reference's knob (8 there); the MI355X default is 64: a decode step streams the weights once whatever the batch."""
    model_name = "Qwen/Qwen2.5-14B-Instruct"
    max_batch_size: int = 64
    debug = False
    llm_model: object
    llm_tokenizer: object

    def __init__(self, device=None, model_name: str = None, weights=None, config=None, tokenizer=None,
                 max_new_tokens: int = 16 * 1024, max_tokens: int = 4096, eos_token_ids=None, generation_config=None):
        super().__init__()
        from .engines.qwen2 import Qwen2
        self.device = dev = _lib.require_device(device if device is not None else 'cuda')      # no CPU fallback
        if model_name is not None:
            self.model_name = model_name
        if weights is None:
            from transformers import AutoConfig, AutoModelForCausalLM
            config = qwen2_config_from_hf(AutoConfig.from_pretrained(self.model_name))
            hf = AutoModelForCausalLM.from_pretrained(self.model_name, torch_dtype='auto')
            weights = hf.state_dict()
            if generation_config is None:
                gc = hf.generation_config
                generation_config = dict(do_sample=bool(gc.do_sample), temperature=gc.temperature or 1.0, top_k=gc.top_k or 0,
                                         top_p=gc.top_p or 1.0, repetition_penalty=gc.repetition_penalty or 1.0)
                if generation_config['do_sample'] and generation_config['top_k'] > 32:
                    # transformers' default top_k (50) on a checkpoint that does not set it: the device sampler keeps the 32
                    # best candidates of a row -- clamp, loudly, rather than die in the sampler's constructor
                    import warnings
                    warnings.warn('InfernLLMWorker: generation_config.top_k=%d clamped to 32 (the device sampler keeps the 32 best '
                                  'candidates per row)' % generation_config['top_k'])
                    generation_config['top_k'] = 32
                if eos_token_ids is None and gc.eos_token_id is not None:
                    eos_token_ids = gc.eos_token_id
            del hf
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(self.model_name)
        self.llm_tokenizer = tokenizer
        with torch.cuda.device(dev):
            self.llm_model = Qwen2(weights, config, dev, max_tokens=max_tokens)
        self.max_new_tokens = max_new_tokens
        eos = eos_token_ids if eos_token_ids is not None else getattr(tokenizer, 'eos_token_id', None)
        self.eos_token_ids = tuple(eos) if isinstance(eos, (list, tuple, set)) else (() if eos is None else (int(eos),))
        pad = getattr(tokenizer, 'pad_token_id', None)
        self.pad_token_id = int(pad) if pad is not None else (self.eos_token_ids[0] if self.eos_token_ids else 0)
        gc = generation_config or {}
        self.sampler = None
        if gc.get('do_sample'):
            from .engines.qwen2 import Sampler
            self.sampler = Sampler(gc.get('temperature', 1.0), gc.get('top_k', 0), gc.get('top_p', 1.0),
                                   gc.get('repetition_penalty', 1.0), gc.get('seed', 0))

    def tokenize_batch(self, wis: List[LLMInferRequest]):
        """chat template + padded tokenisation exactly as InfernLLMWorker.py:108-112, then every row's own token list
        (padding removed through the attention mask): rows are decoded at their own lengths -- the result of a
        left-padded batch in transformers' generate"""
        messages = [self.llm_tokenizer.apply_chat_template(list(r.context), tokenize=False, add_generation_prompt=True)
                    for r in wis]
        enc = self.llm_tokenizer(messages, return_tensors="pt", padding=True)
        ids, mask = enc['input_ids'], enc['attention_mask']
        limit = self.llm_model.max_tokens - 2
        prompts = []
        for row, m in zip(ids.tolist(), mask.tolist()):
            p = [t for t, k in zip(row, m) if k]
            prompts.append(p[-limit:] if len(p) > limit else p)          # longer than the cache: keep the tail
        return ids, prompts

    def process_batch(self, wis: List[LLMInferRequest]):
        if self.debug:
            print(f'InfernLLMWorker.process_batch: got {len(wis)=}')
        streamer = ResultsStreamer(wis, self)
        ids, prompts = self.tokenize_batch(wis)
        with torch.cuda.device(self.device):
            streamer.put(ids)
            self.llm_model.generate(prompts, self.max_new_tokens, eos_ids=self.eos_token_ids, pad_id=self.pad_token_id,
                                    on_tokens=streamer.put, sampler=self.sampler)
        streamer.end()
