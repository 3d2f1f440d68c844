import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
N = 256
for (T, C) in ((768, 64),):
    x = torch.randn(N, T, C, device=dev).to(BF); r = torch.randn(N, T, C, device=dev).to(BF)
    out = torch.empty(N, T, C, dtype=BF, device=dev)
    bias = torch.zeros(C, device=dev)
    for k, d in ((3, 1),):
        w = (torch.randn(C, k, C, device=dev) / (C * k) ** 0.5).to(BF)
        for ub, ur in ((1, 1), (0, 1), (1, 0), (0, 0)):
            print('bias', ub, 'resid', ur, file=sys.stderr, flush=True)
            for _ in range(3):
                ops.conv(x, w, bias if ub else None, out, nbatch=N, t_in=T, t_out=T, cin=C, n=C, taps=k, dil=d, pad=(k * d - d) // 2, pre_slope=0.1, resid=r if ur else None)
            torch.cuda.synchronize()
