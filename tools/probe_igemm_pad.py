"""Row-stride experiment for the LDS-staged implicit GEMM (encoder shapes): python tools/probe_igemm_pad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (M, K, N) in ((96000, 384, 1536), (96000, 1536, 384), (96000, 384, 1152), (16384, 768, 3072), (16384, 3072, 768)):
    for pad in (0, 8, 32):
        x = torch.randn(M, K + pad, device=dev).to(BF)
        w = (torch.randn(N, 1, K, device=dev) / K ** 0.5).to(BF)
        b = torch.zeros(N, device=dev)
        out = torch.empty(M, N + pad, dtype=BF, device=dev)
        t = timeit(lambda: ops.conv(x, w, b, out, nbatch=1, t_in=M, t_out=M, cin=K, n=N, taps=1, lda=K + pad, ldc=N + pad))
        print(f'M={M} K={K:5d} N={N:5d} pad={pad:2d}: {t*1e6:7.1f} us  {2.0*M*K*N/t/1e12:6.1f} TF/s')
