"""Per-shape timing of the vocoder convolutions (tuning aid): python tools/probe_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
def timeit(fn, n=20):
    """GPU-side time per launch: n launches captured in a hipGraph (no host launch overhead)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
N = 256
for (T, C) in ((48, 256), (192, 128), (768, 64), (3072, 32)):
    x = torch.randn(N, T, C, device=dev).to(BF); r = torch.randn(N, T, C, device=dev).to(BF)
    out = torch.empty(N, T, C, dtype=BF, device=dev)
    bias = torch.zeros(C, device=dev)
    for k, d in ((3, 1), (7, 1), (11, 1), (11, 5)):
        w = (torch.randn(C, k, C, device=dev) / (C * k) ** 0.5).to(BF)
        t = timeit(lambda: ops.conv(x, w, bias, out, nbatch=N, t_in=T, t_out=T, cin=C, n=C, taps=k, dil=d, pad=(k * d - d) // 2,
                                    pre_slope=0.1, resid=r))
        fl = 2.0 * N * T * C * C * k
        by = 3.0 * N * T * C * 2
        print(f'T={T:5d} C={C:4d} k={k:2d} d={d}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  {by/t/1e9:7.0f} GB/s(min traffic)')
