"""C=256 residual-block convolutions at the bench's launch-group size (768 chunks x 48 rows): python tools/probe_c256.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T, C = 48, 256
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
x = torch.randn(N, T, C, device=dev).to(BF); r = torch.randn(N, T, C, device=dev).to(BF)
out = torch.empty(N, T, C, dtype=BF, device=dev)
bias = torch.zeros(C, device=dev)
tot = 0.0
for k, d in ((3, 1), (3, 3), (3, 5), (7, 1), (7, 3), (7, 5), (11, 1), (11, 3), (11, 5)):
    w = (torch.randn(C, k, C, device=dev) / (C * k) ** 0.5).to(BF)
    t = timeit(lambda: ops.conv(x, w, bias, out, nbatch=N, t_in=T, t_out=T, cin=C, n=C, taps=k, dil=d, pad=(k * d - d) // 2,
                                pre_slope=0.1, resid=r))
    fl = 2.0 * N * T * C * C * k
    tot += t
    print(f'k={k:2d} d={d}: {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TF/s  sum={float(out.float().abs().sum()):.6e}')
print(f'total {tot*1e6:.1f} us')
