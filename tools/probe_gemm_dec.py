"""GPU time per launch of the decode-step GEMM shapes by row count (50 launches captured in one hipGraph).
python tools/probe_gemm_dec.py   (IFH_GEMM_DEC_ROWS=100000 -> streaming kernel only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
print('IFH_GEMM_DEC_ROWS=%s' % os.environ.get('IFH_GEMM_DEC_ROWS'))
shapes = [('tts qkv', 768, 2304), ('tts wo', 768, 768), ('tts ff1', 768, 3072), ('tts ff2', 3072, 768), ('w-base qkv', 512, 1536),
          ('w-base ff1', 512, 2048), ('w-base ff2', 2048, 512), ('w-base wo', 512, 512)]
for M in (256, 512, 640, 1024):
    line = []
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev).to(BF)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
        b = torch.zeros(N, device=dev)
        stats = torch.zeros((1, 1024, 2), dtype=torch.int64, device=dev)
        stats[0, :, 1] = 65536 * K
        c1 = torch.zeros(N, device=dev)
        out = torch.empty(M, N, dtype=BF, device=dev)
        fn = lambda: ops.linear(x, w, b, out, rows=M, k=K, n=N, aln=(stats, 0, c1), ln_dim=K)
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        line.append('%s %.1f us (%.0f TF/s)' % (name, us, 2.0 * M * K * N / us / 1e6))
    print('M=%d: ' % M + ' | '.join(line))
