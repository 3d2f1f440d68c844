#!/usr/bin/env python3
"""Time ifh_resblock_seq_bf16 per (C, taps) at the vocoder's level shapes and print its phase clocks (debug_prof):
    python tools/probe_seq.py [nchunks] [C,C,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
BF = torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
cs = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [64, 128, 256]
dev = _lib.require_device('cuda:0')
g = torch.Generator().manual_seed(0)
for c in cs:
    T = {64: 768, 128: 192, 256: 48}[c]
    x = torch.randn(n, T, c, generator=g).to(BF).to(dev)
    out = torch.zeros_like(x)
    for k in (3, 7, 11):
        if not ops.seq_supported(c, T, k):
            continue
        convs = [((torch.randn(c, c, k, generator=g) / (c * k) ** 0.5).to(BF).float(), torch.randn(c, generator=g) * 0.1) for _ in range(6)]
        ws, nu, bias = ops.w_chain_pack(convs, dev, unit_bytes=ops.seq_unit_bytes(c))
        fn = lambda prof=None: ops.resblock_seq(x, ws, nu, bias, out, nbatch=n, t=T, c=c, taps=k, scale=1 / 3, accumulate=True, prof=prof)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        gf = 6 * 2 * T * c * c * k * n / 1e9
        prof = torch.zeros(8, dtype=torch.int64, device=dev)
        fn(prof); torch.cuda.synchronize()
        pr = prof.cpu().tolist()
        nt = max(1, pr[6])
        ideal = 6 * 2 * T * c * c * k * (2 if c != 64 else 1) / 4096        # MFMA clocks per tile (NSEQ sequences) at 4 SIMDs x 1024 FLOP/clk
        print('C=%3d k=%2d: %7.1f us %6.0f TF/s | per tile (wave 0 clocks): top %d  K-loops %d  K-barrier %d  epi1 %d  epi2 %d  last %d  | sum %d  MFMA-ideal %d  (%.0f %%)' % (
            c, k, t * 1e6, gf / t / 1e3, pr[0] // nt, pr[1] // nt, pr[2] // nt, pr[3] // nt, pr[4] // nt, pr[5] // nt, sum(pr[:6]) // nt, ideal, 100.0 * ideal * nt / max(1, sum(pr[:6]))), flush=True)
