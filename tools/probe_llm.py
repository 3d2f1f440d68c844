"""Qwen2.5-shaped decode on one GPU (BASELINE configuration 5's LLM): prefill and per-token step time at B rows.
usage: python tools/probe_llm.py [B] [prompt_len] [new_tokens] [qwen2_1p5b|qwen2_14b]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infernos_amd import _lib                                    # noqa: E402
from infernos_amd.engines.qwen2 import Qwen2                     # noqa: E402
from infernos_amd.weights import QWEN2_CONFIGS, qwen2_random_on_device, qwen2_schema     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
P = int(sys.argv[2]) if len(sys.argv) > 2 else 192
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = _lib.require_device('cuda:0')
FAM = sys.argv[4] if len(sys.argv) > 4 else 'qwen2_1p5b'
cfg = QWEN2_CONFIGS[FAM]


t0 = time.perf_counter()
model = Qwen2(qwen2_random_on_device(cfg, dev), cfg, dev, max_tokens=P + N + 8)
torch.cuda.synchronize()
print('model built in %.1f s' % (time.perf_counter() - t0))
g = torch.Generator().manual_seed(1)
prompts = [torch.randint(10, cfg['vocab'] - 10, (P - (i % 7),), generator=g).tolist() for i in range(B)]
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st, _ = model.prefill(prompts)
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
model.step(st, B)
model.step(st, B)          # captures
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    model.step(st, B)
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / N
params = sum(int(torch.tensor(s).prod()) for s, _ in qwen2_schema(cfg).values())
wbytes = 2 * params
print(FAM + ' shape, B=%d, prompt %d: prefill %.1f ms (%.0f tok/s); decode %.3f ms/step = %.0f tok/s; weights %.2f GB -> '
      '%.2f TB/s of weight streaming (%.0f %% of 8 TB/s)' % (B, P, t_pre * 1e3, B * P / t_pre, t_step * 1e3, B / t_step, wbytes / 1e9,
                                                            wbytes / t_step / 1e12, 100 * wbytes / t_step / 8e12))

# ---- the pieces of a decode step on their own (HIP events, 50 launches each) ----
from infernos_amd import ops                                     # noqa: E402


def ev(fn, n=50):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


L = model.layers[0]
d, ff = model.d, model.ff
kv = st['kv'][0]
parts = {
    'rmsnorm': lambda: ops.rmsnorm(st['x'], model.ones, st['h'], B, d, model.eps),
    'qkv %dx%d' % (model.nq, d): lambda: ops.linear(st['h'], L['wqkv'], L['bqkv'], st['qkv'], rows=B, k=d, n=model.nq),
    'rope': lambda: ops.rope_append(st['qkv'], model.cos_sin, kv, st['lens'][0], st['ones'], nrows=B, tokens_per_row=1, nheads=model.nh,
                                    nkv=model.nkv, head_dim=model.hd, max_pos=model.max_tokens),
    'attn': lambda: ops.attn_gqa(st['qkv'], kv, st['att'], st['lens'][1], ntokens=B, tokens_per_row=1, nheads=model.nh, nkv=model.nkv,
                                 head_dim=model.hd, max_pos=model.max_tokens, max_keys=model.max_tokens),
    'o %dx%d' % (d, d): lambda: ops.linear(st['att'], L['wo'], None, st['x'], rows=B, k=d, n=d, resid=st['x']),
    'gate_up %dx%d' % (2 * ff, d): lambda: ops.linear(st['h'], L['wgu'], None, st['gu'], rows=B, k=d, n=2 * ff),
    'silu_mul': lambda: ops.silu_mul(st['gu'], st['ff'], B, ff, interleaved=True),
    'down %dx%d' % (d, ff): lambda: ops.linear(st['ff'], L['wd'], None, st['x'], rows=B, k=ff, n=d, resid=st['x']),
    'head %dx%d' % (model.vocab, d): lambda: ops.linear(st['h'], model.head, None, st['logits_full'], rows=B, k=d, n=model.vocab, ldc=model.vpad),
    'argmax': lambda: ops.argmax_pick(st['logits_full'], vocab=model.vocab, nrows=B, ld=model.vpad, argmax_out=st['toks']),
}
print('per launch (us): ' + ' | '.join('%s %.1f' % (k, ev(f)) for k, f in parts.items()))

# ---- prefill pieces at B x P tokens (one layer) ----
rows = B * P
e = lambda *s_: torch.empty(s_, dtype=torch.bfloat16, device=dev)
x, h, qkv, att, gu, ffb = e(rows, d), e(rows, d), e(rows, model.nq), e(rows, model.nh * model.hd), e(rows, 2 * ff), e(rows, ff)
x.normal_()
lens_t = torch.full((B,), P, dtype=torch.int32, device=dev)
pos0 = torch.zeros(B, dtype=torch.int32, device=dev)
t_idx = torch.arange(P, dtype=torch.int32, device=dev)[None, :].expand(B, P)
key_len = (t_idx + 1).contiguous()
pp = {
    'rmsnorm': lambda: ops.rmsnorm(x, model.ones, h, rows, d, model.eps),
    'qkv': lambda: ops.linear(h, L['wqkv'], L['bqkv'], qkv, rows=rows, k=d, n=model.nq),
    'rope': lambda: ops.rope_append(qkv, model.cos_sin, kv, pos0, lens_t, nrows=B, tokens_per_row=P, nheads=model.nh, nkv=model.nkv,
                                    head_dim=model.hd, max_pos=model.max_tokens),
    'attn': lambda: ops.attn_gqa(qkv, kv, att, key_len, ntokens=rows, tokens_per_row=P, nheads=model.nh, nkv=model.nkv,
                                 head_dim=model.hd, max_pos=model.max_tokens, max_keys=P),
    'o': lambda: ops.linear(att, L['wo'], None, x, rows=rows, k=d, n=d, resid=x),
    'gate_up': lambda: ops.linear(h, L['wgu'], None, gu, rows=rows, k=d, n=2 * ff),
    'silu_mul': lambda: ops.silu_mul(gu, ffb, rows, ff, interleaved=True),
    'down': lambda: ops.linear(ffb, L['wd'], None, x, rows=rows, k=ff, n=d, resid=x),
}
res = {k: ev(f, 5) for k, f in pp.items()}
print('prefill per layer at %d tokens (us): ' % rows + ' | '.join('%s %.0f' % kv_ for kv_ in res.items()) +
      ' | sum x %d layers = %.1f ms' % (len(model.layers), sum(res.values()) * len(model.layers) / 1e3))
