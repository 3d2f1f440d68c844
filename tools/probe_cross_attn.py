"""Beam cross-attention (128 utterances x 5 beams x 8 heads over 1500 keys): the VALU decode kernel that shares K/V loads
(ifh_attn_decode_shared_bf16) against the MFMA prefill kernel run with tq = beams query rows per utterance."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
B, K, H, S = 128, 5, 8, 1500
d = H * 64
g = torch.Generator().manual_seed(0)
q = (torch.randn(B * K, d, generator=g) * 0.3).to(dev, BF)
kvs = [torch.randn(B, S, 2 * d, generator=g).to(dev, BF) for _ in range(6)]
o1 = torch.empty(B * K, d, dtype=BF, device=dev)
o2 = torch.empty(B * K, d, dtype=BF, device=dev)


def dec():
    for kv in kvs:
        ops.attn_decode_shared(q, kv, kv, o1, nbatch=B * K, nheads=H, max_keys=S, q_bs=d, kv_bs=S * 2 * d, kv_ts=2 * d, o_bs=d, v_off=d, kv_group=K)


def pre():
    for kv in kvs:
        ops.attn_prefill(q, kv, kv, o2, nbatch=B, nheads=H, tq=K, tk=S, v_off=d, q_ts=d, k_ts=2 * d, v_ts=2 * d, o_ts=d)


for name, fn in (('decode_shared', dec), ('prefill tq=5', pre)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print('%s: %.0f us per 6 layers = %.2f TB/s of K/V' % (name, us, 6 * B * S * 2 * d * 2 / us / 1e6))
a, b = o1.float(), o2.float()
print('rel l2 between the two', float((a - b).norm() / a.norm()))
