"""Whisper-base encoder attention alone: python tools/probe_attn_prefill.py [B] [H] [T]  (both forms, TF/s)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
D = H * 64
qkv = (torch.randn(B, T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(B, T, D, dtype=torch.bfloat16, device=dev)
flops = 4.0 * B * H * T * T * 64
res = {}
for form in ('0', '1', '0', '1'):
    os.environ['IFH_ATTN_PREFILL2'] = form
    def run():
        ops.attn_prefill(qkv, qkv, qkv, out, nbatch=B, nheads=H, tq=T, tk=T, k_off=D, v_off=2 * D, q_ts=3 * D, k_ts=3 * D,
                         v_ts=3 * D, o_ts=D)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    res[form] = out.clone()
    print(f'form {form}: {ms*1e3:.0f} us -> {flops/ms/1e9:.0f} TF/s', flush=True)
print('identical:', torch.equal(res['0'].view(torch.int16), res['1'].view(torch.int16)))
