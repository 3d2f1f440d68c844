#!/usr/bin/env python3
"""The vocoder's fused upsamplers (ConvTranspose1d(k8, s4, p2) as ONE 3-tap convolution with 4 Cout columns, LeakyReLU on the operand
load) as they run today (k_igemm's general tile loads) against the same product as a PLAIN matrix product over overlapping rows
(a zero guard row between sequences makes the 3-tap window [t-1, t, t+1] of channels-last rows one contiguous K = 3 Cin vector with
row stride Cin; the LeakyReLU would move into the producer): time only (the guard-row operand here is random data).
    python tools/probe_upsampler.py [nchunks]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from infernos_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16


def ev_time(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
    dev = _lib.require_device('cuda:0')
    g = torch.Generator().manual_seed(0)
    tot_a = tot_b = 0.0
    for i, (t, c) in enumerate(((12, 512), (48, 256), (192, 128), (768, 64))):
        co = c // 2
        w = torch.randn(c, co, 8, generator=g) / (c * 2) ** 0.5
        b = torch.randn(co, generator=g) * 0.1
        wf, bf = ops.w_convT_fused(w, b, dev)
        x = torch.randn(n, t, c, generator=g).to(BF).to(dev)
        u = torch.empty(n, 4 * t, co, dtype=BF, device=dev)
        ta = ev_time(lambda: ops.conv(x, wf, bf, u, nbatch=n, t_in=t, t_out=t, cin=c, n=2 * c, taps=3, pad=1, pre_slope=0.1, convt_cout=co))
        # plain form: rows n (t + 2) with guards, K = 3 c contiguous from row m (lda = c), output rows m + 1
        rows = n * (t + 2)
        xg = torch.randn(rows + 2, c, generator=g).to(BF).to(dev)
        ug = torch.empty(rows + 2, 2 * c, dtype=BF, device=dev)
        tb = ev_time(lambda: ops.linear(xg, wf, bf, ug[1:], rows=rows, k=3 * c, n=2 * c, lda=c))
        fl = 2.0 * n * t * (2 * c) * (2 * c)          # useful: two of the three taps per phase
        print('upsampler %d: %4d -> %4d channels, %7d rows: today %6.1f us (%5.0f TF/s useful)   plain over guard rows %6.1f us (%5.0f TF/s useful, x%.2f)' % (
            i, c, co, n * t, ta * 1e6, fl / ta / 1e12, tb * 1e6, fl / tb / 1e12, ta / tb))
        tot_a += ta; tot_b += tb
    print('four upsamplers: today %.1f us, plain %.1f us' % (tot_a * 1e6, tot_b * 1e6))


if __name__ == '__main__':
    main()
