"""The 5-beam cross-attention of a Whisper-base decode step (128 utterances x 8 heads, 5 query rows, 1500 keys):
python tools/probe_cross_few.py [B] [H] [Tq] [Tk]   (IFH_ATTN_FEW = 0: k_attn_prefill, 1: k_attn_prefill_few)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Tq = int(sys.argv[3]) if len(sys.argv) > 3 else 5
Tk = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
D = H * 64
NL = 6                                   # six layers' caches, walked in turn: 2.4 GB, nothing stays in a cache between launches
q = (torch.randn(B, Tq, D, device=dev) * 0.5).to(torch.bfloat16)
kvs = [(torch.randn(B, Tk, 2 * D, device=dev)).to(torch.bfloat16) for _ in range(NL)]
out = torch.empty(B, Tq, D, dtype=torch.bfloat16, device=dev)
byt = B * Tk * 2 * D * 2
res = {}
for form in ('0', '1', '0', '1'):
    os.environ['IFH_ATTN_FEW'] = form
    def run(i):
        ops.attn_prefill(q, kvs[i % NL], kvs[i % NL], out, nbatch=B, nheads=H, tq=Tq, tk=Tk, v_off=D, q_ts=D, k_ts=2 * D, v_ts=2 * D, o_ts=D)
    for i in range(6): run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    run(0); torch.cuda.synchronize()
    res[form] = out.clone()
    print(f'form {form}: {us:.1f} us -> {byt/us/1e6:.2f} TB/s of K/V', flush=True)
print('identical:', torch.equal(res['0'].view(torch.int16), res['1'].view(torch.int16)))
