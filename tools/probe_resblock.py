"""Residual pair: fused launch (ifh_resblock_pair_bf16) vs the two conv launches (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for (T, C) in ((48, 256), (192, 128), (768, 64), (3072, 32)):
    x = torch.randn(N, T, C, device=dev).to(BF)
    h = torch.empty(N, T, C, dtype=BF, device=dev); out = torch.empty(N, T, C, dtype=BF, device=dev)
    b1 = torch.zeros(C, device=dev); b2 = torch.zeros(C, device=dev)
    for k, d in ((3, 1), (3, 5), (7, 3), (11, 1), (11, 5)):
        w1 = (torch.randn(C, k, C, device=dev) / (C * k) ** 0.5).to(BF); w2 = (torch.randn(C, k, C, device=dev) / (C * k) ** 0.5).to(BF)
        def two():
            ops.conv(x, w1, b1, h, nbatch=N, t_in=T, t_out=T, cin=C, n=C, taps=k, dil=d, pad=(k * d - d) // 2, pre_slope=0.1)
            ops.conv(h, w2, b2, out, nbatch=N, t_in=T, t_out=T, cin=C, n=C, taps=k, pad=(k - 1) // 2, pre_slope=0.1, resid=x)
        def one():
            ops.resblock_pair(x, w1, b1, w2, b2, out, nbatch=N, t=T, c=C, taps=k, dil=d)
        t2, t1 = timeit(two), timeit(one)
        fl = 2 * 2.0 * N * T * C * C * k
        print(f'T={T:5d} C={C:4d} k={k:2d} d={d}: two convs {t2*1e6:7.1f} us ({fl/t2/1e12:6.1f} TF/s)   fused {t1*1e6:7.1f} us ({fl/t1/1e12:6.1f} TF/s)  x{t2/t1:.2f}')
