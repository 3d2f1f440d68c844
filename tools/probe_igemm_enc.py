"""k_igemm at the Whisper-base encoder shapes (M = 128 x 1500 rows), with and without the epilogue operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
for name, K, N, act, res in (('qkv', 512, 1536, 0, False), ('wo', 512, 512, 0, True), ('wo no-resid', 512, 512, 0, False),
                             ('fc1 gelu', 512, 2048, 2, False), ('fc1 no-act', 512, 2048, 0, False), ('fc1 relu', 512, 2048, 1, False),
                             ('fc2', 2048, 512, 0, True), ('fc2 no-resid', 2048, 512, 0, False)):
    x = torch.randn(M, K, device=dev).to(BF)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF)
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, dtype=BF, device=dev)
    r = torch.randn(M, N, device=dev).to(BF) if res else None
    fn = lambda: ops.linear(x, w, b, out, rows=M, k=K, n=N, act=act, resid=r)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print('%-14s M=%d K=%d N=%d: %.0f us = %.0f TF/s' % (name, M, K, N, us, 2.0 * M * K * N / us / 1e6))
