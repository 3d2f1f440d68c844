#!/usr/bin/env python3
"""Ablation timings of k_gemm_big8 at the Whisper-base encoder shapes -- needs tools/build_gemm_abl.sh's library.
IFH_GEMM_BIG_ABL bits: 1 no DMA after a tile's first stages, 2 no MFMAs, 4 no epilogue, 16 no output stores (results are wrong by
design); 8 = phase clocks of wave 0 (K loop, waits in front of the stage barriers, the barriers, epilogue), printed per launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('GB_CHILD'):
    import torch
    from infernos_amd import _lib, ops
    BF = torch.bfloat16
    dev = _lib.require_device('cuda:0')
    M = 192000
    for name, K, N, act, res in (('qkv', 512, 1536, 0, False), ('fc1 gelu', 512, 2048, 2, False), ('fc2', 2048, 512, 0, False),
                                 ('wo+resid', 512, 512, 0, True)):
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
        print('ABL=%s %-10s %6.0f us = %5.0f TF/s' % (os.environ.get('IFH_GEMM_BIG_ABL', '0'), name, us, 2.0 * M * K * N / us / 1e6))
else:
    runs = [a for a in (sys.argv[1].split(',') if len(sys.argv) > 1 else ('0', '1', '2', '4', '5', '6', '3', '16', '17'))] + ['8']
    for abl in runs:
        env = dict(os.environ, GB_CHILD='1', IFH_GEMM_BIG_ABL=abl)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        print('\n'.join((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-4:]))
        if abl == '8':
            seen = set()
            for ln in r.stderr.splitlines():
                if ln.startswith('k_gemm_big8') and ln.split(':')[0] not in seen:
                    seen.add(ln.split(':')[0])
                    print('   ', ln)
