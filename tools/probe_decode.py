"""Decode-step kernel timings (graph-captured, GPU-side): python tools/probe_decode.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return min(ts)
B, H, D = 64, 12, 768
q = torch.randn(B, D, device=dev).to(BF); out = torch.empty(B, D, dtype=BF, device=dev)
for S, SM in ((64, 64), (160, 672), (640, 672), (1500, 1500)):
    kv = torch.randn(B, SM, 2 * D, device=dev).to(BF)
    pos = torch.tensor([S - 1], dtype=torch.int32, device=dev)
    for mk in (SM, 1025 if SM <= 1024 else SM):
        t = timeit(lambda: ops.attn_decode(q, kv, kv, out, nbatch=B, nheads=H, max_keys=mk, q_bs=D, kv_bs=SM * 2 * D, kv_ts=2 * D,
                                           o_bs=D, v_off=D, dyn_len=pos, dyn_add=1))
        print(f'attn_decode keys={S:5d} max_keys={mk:5d} ({"NW4" if mk > 1024 else "NW1"}): {t:6.2f} us')
x = torch.randn(B, 3072, device=dev).to(BF)
for (K, N) in ((768, 768), (768, 2304), (768, 3072), (3072, 768), (1280, 768), (80, 256), (768, 160)):
    w = torch.randn(N, K, device=dev).to(BF); b = torch.zeros(N, device=dev); o = torch.empty(B, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.linear(x, w, b, o, rows=B, k=K, n=N, lda=3072))
    print(f'skinny gemm M=64 K={K:5d} N={N:5d}: {t:6.2f} us   weights {N*K*2/1e6:.2f} MB -> {N*K*2/t/1e3:.0f} GB/s')
g_, b_ = torch.ones(D, device=dev), torch.zeros(D, device=dev)
xx = torch.randn(B, D, device=dev).to(BF)
t = timeit(lambda: ops.layernorm(xx, g_, b_, out, B, D))
print(f'layernorm 64x768: {t:6.2f} us')
p = torch.zeros(1, dtype=torch.int32, device=dev)
t = timeit(lambda: ops.add_i32(p, 1))
print(f'add_i32 (empty-ish kernel): {t:6.2f} us')
