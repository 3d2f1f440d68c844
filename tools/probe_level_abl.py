#!/usr/bin/env python3
"""Ablation timings of k_resblock_level (k = 11 block, C = 32) -- needs a library whose level.hip was compiled with -DLV_DEV_ABL:
    hipcc ... -DLV_DEV_ABL -c infernos_amd/csrc/level.hip -o infernos_amd/build/level.hip.o && relink (tools/build_level_abl.sh)
IFH_LEVEL_ABL bits: 1 no fragment reloads, 2 no epilogue pieces, 4 no activation reads, 8 no MFMAs (results are wrong by design).
    python tools/probe_level_abl.py [nchunks]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get('LV_CHILD'):
    import torch
    from infernos_amd import _lib, ops
    BF = torch.bfloat16
    n = int(sys.argv[1])
    dev = _lib.require_device('cuda:0')
    g = torch.Generator().manual_seed(0)
    c, T, k = 32, 3072, int(os.environ.get('LV_TAPS', '11'))
    x = torch.randn(n, T, c, generator=g).to(BF).to(dev)
    convs = [((torch.randn(c, c, k, generator=g) / (c * k) ** 0.5).to(BF).float(), torch.randn(c, generator=g) * 0.1) for _ in range(6)]
    ws, nu, bias = ops.w_chain_pack(convs, dev)
    out = torch.zeros_like(x)
    prof = torch.zeros(16, dtype=torch.int64, device=dev) if os.environ.get('IFH_LEVEL_ABL') == '16' else None
    fn = lambda: ops.resblock_level(x, [(k, ws, bias)], out, nbatch=n, t=T, c=c, scale=1 / 3, prof=prof)
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 8 * 1e-3
    print('k=%-2d ABL=%-3s %8.1f us  %7.1f TF/s' % (k, os.environ.get('IFH_LEVEL_ABL', '0'), t * 1e6, 6 * 2.0 * n * T * c * c * k / t / 1e12))
    if prof is not None:
        pr = prof.cpu().tolist()
        nc = max(1, pr[6])
        print('   per convolution (wave 0, shader clocks): setup %d  barrier %d  W-wait %d  main %d  last-rb %d  final-epi %d   | block top %d per block; convs %d' % (
            pr[0] // nc, pr[1] // nc, pr[2] // nc, pr[3] // nc, pr[4] // nc, pr[5] // nc, pr[7] * 6 // nc, nc))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else '1280'
    runs = [('0', '11'), ('1', '11'), ('4', '11'), ('12', '11'), ('13', '11'), ('16', '11'), ('0', '7'), ('16', '7'), ('0', '3'), ('16', '3')]
    for abl, taps in runs:
        env = dict(os.environ, LV_CHILD='1', IFH_LEVEL_ABL=abl, LV_TAPS=taps)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), n], env=env, capture_output=True, text=True)
        print('\n'.join((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-2:]))
