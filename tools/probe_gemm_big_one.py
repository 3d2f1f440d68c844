"""One gemm_big shape, a few launches (for rocprofv3 --pmc): python tools/probe_gemm_big_one.py K N act resid"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
K, N, act, res = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
M = 192000
x = torch.randn(M, K, device=dev).to(BF)
w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
b = torch.zeros(N, device=dev)
out = torch.empty(M, N, dtype=BF, device=dev)
r = torch.randn(M, N, device=dev).to(BF) if res else None
for _ in range(4):
    ops.linear(x, w, b, out, rows=M, k=K, n=N, act=act, resid=r)
torch.cuda.synchronize()
