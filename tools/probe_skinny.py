"""Decode-step GEMM shapes of the SpeechT5 decoder at M rows (tuning aid): python tools/probe_skinny.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # extra elements per activation row (row-stride experiment)
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (K, N) in ((768, 2304), (768, 768), (768, 3072), (3072, 768), (256, 768), (768, 160)):
    x = torch.randn(M, 1, K + PAD, device=dev).to(BF)
    w = (torch.randn(N, 1, K, device=dev) / K ** 0.5).to(BF)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, 1, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.conv(x, w, b, out, nbatch=M, t_in=1, t_out=1, cin=K, n=N, taps=1, x_bstride=K + PAD))
    l2 = (M * K * 2 * (N / 16) + N * K * 2 * (M / 32)) / 1e6
    print(f'M={M} K={K:5d} N={N:5d}: {t*1e6:6.2f} us   {2.0*M*K*N/t/1e12:6.1f} TF/s   L2 reads {l2:6.1f} MB -> {l2/1e6/t:5.1f} TB/s')
