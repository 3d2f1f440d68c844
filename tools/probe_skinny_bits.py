"""Diagnostic: bits of the LayerNorm-folded skinny GEMM at M rows against the same rows in 64-row pieces, per mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
D = 768
g = torch.Generator().manual_seed(0)
for M in (128, 256, 512, 640):
    x = torch.randn(M, D, generator=g).to(dev, BF)
    stats = torch.zeros((1, 1024, 2), dtype=torch.int64, device=dev)
    xf = x.float().double()
    stats[0, :M, 0] = torch.round(xf.sum(1) * 65536).long()
    stats[0, :M, 1] = torch.round((xf ** 2).sum(1) * 65536).long()
    for N in (768, 2304, 3072):
        w = (torch.randn(N, D, generator=g) / D ** 0.5)
        b = torch.randn(N, generator=g) * 0.1
        gam, bet = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
        wf, c2, c1 = ops.w_linear_ln(w, b, gam, bet, dev)
        y = torch.empty(M, N, dtype=BF, device=dev)
        ops.linear(x, wf, c2, y, rows=M, k=D, n=N, aln=(stats, 0, c1), ln_dim=D)
        yp = torch.empty(M, N, dtype=BF, device=dev)
        for r0 in range(0, M, 64):
            ops.linear(x[r0:r0 + 64], wf, c2, yp[r0:r0 + 64], rows=64, k=D, n=N, aln=(stats, r0 * 2, c1), ln_dim=D)
        torch.cuda.synchronize()
        print('aln M=%d N=%d bits %s' % (M, N, 'same' if torch.equal(y.view(torch.int16), yp.view(torch.int16)) else 'DIFF %d' % int((y != yp).sum())))
        y2 = torch.empty(M, N, dtype=BF, device=dev)
        ops.linear(x, wf, c2, y2, rows=M, k=D, n=N)
        yp2 = torch.empty(M, N, dtype=BF, device=dev)
        for r0 in range(0, M, 64):
            ops.linear(x[r0:r0 + 64], wf, c2, yp2[r0:r0 + 64], rows=64, k=D, n=N)
        torch.cuda.synchronize()
        print('plain M=%d N=%d bits %s' % (M, N, 'same' if torch.equal(y2.view(torch.int16), yp2.view(torch.int16)) else 'DIFF %d' % int((y2 != yp2).sum())))
