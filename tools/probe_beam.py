"""Time Whisper decode: greedy vs beam search (B utterances, 32 new tokens) and, under rocprofv3, show where a beam
step goes.  usage: python tools/probe_beam.py [family] [B] [beams]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from infernos_amd import _lib                                    # noqa: E402
from infernos_amd.engines.whisper import Whisper                 # noqa: E402
from infernos_amd.weights import synth_state_dict                # noqa: E402

family = sys.argv[1] if len(sys.argv) > 1 else 'whisper_base'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = _lib.require_device('cuda:0')
model = Whisper(synth_state_dict(family, 1), dev)
g = torch.Generator().manual_seed(0)
enc = (torch.randn(B, 1500, model.d, generator=g) * 0.5).to(dev, torch.bfloat16)
prompt = torch.tensor([[50258, 50259, 50359, 50363]] * B, dtype=torch.int32)


UG = os.environ.get('IFH_NO_GRAPHS') is None


def timed(fn, n=3):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tg = timed(lambda: model.generate(enc, prompt, 32, no_speech_id=50362, use_graphs=UG))
tb = timed(lambda: model.generate_beam(enc, prompt, 32, beams=K, eos_id=50257, no_speech_id=50362, check_every=64, use_graphs=UG))
print('%s B=%d: greedy %.2f ms, beam-%d %.2f ms (%.3f ms/step over 35 steps)' % (family, B, tg, K, tb, tb / 35))

# ---- where a beam step goes: the pieces timed on their own (HIP events, 20 launches each) ----
from infernos_amd import ops                                     # noqa: E402
rows = B * K
bufs = model._dec(rows, K)
st = bufs['beam_state']


def ev(fn, n=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


bufs['pos'].fill_(20)
d, H = model.d, model.h
t_step = ev(lambda: (model.decoder_step(bufs, rows, False), bufs['pos'].fill_(20)))
t_beam = ev(lambda: ops.beam_step(bufs['logits_full'], st, bufs['toks'], bufs['pos'], vocab=model.vocab, ld=model.vpad,
                                  prompt_len=4, max_length=36, eos_id=50257))
t_gather = ev(lambda: [ops.kv_gather(a, b, st.beam_src, bufs['pos'], nrows=rows, max_len=model.max_tokens, tok_elems=2 * d)
                       for a, b in zip(bufs['kv'], bufs['kv2'])])
t_cross = ev(lambda: [model._cross_attn(bufs, li, rows) for li in range(len(model.dec_layers))])
kv = bufs['kv'][0]
t_self = ev(lambda: [ops.attn_decode(bufs['q'], kv, kv, bufs['att'], nbatch=rows, nheads=H, max_keys=model.max_tokens, q_bs=d,
                                     kv_bs=model.max_tokens * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d, dyn_len=bufs['pos'], dyn_add=1)
                     for _ in model.dec_layers])
t_head = ev(lambda: ops.linear(bufs['hn'], model.tok, None, bufs['logits_full'], rows=rows, k=d, n=model.vocab, ldc=model.vpad))
L0 = model.dec_layers[0]
t_ff = ev(lambda: (ops.linear(bufs['hn'], L0['w1'], L0['b1'], bufs['ff'], rows=rows, k=d, n=model.ff),
                   ops.linear(bufs['ff'], L0['w2'], L0['b2'], bufs['x'], rows=rows, k=model.ff, n=d)))
print('rows %d (eager, us): decoder step %.0f | beam_step %.0f | kv gather x%d %.0f | cross attn x%d %.0f | self attn x%d %.0f | '
      'vocab head %.0f | ff1+ff2 (one layer) %.0f' % (rows, t_step, t_beam, len(bufs['kv']), t_gather, len(bufs['kv']), t_cross,
                                                       len(bufs['kv']), t_self, t_head, t_ff))
