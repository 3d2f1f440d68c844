#!/usr/bin/env python3
"""Time ifh_resblock_chain_bf16 against the three ifh_resblock_pair_bf16 launches it replaces, per (C, taps), at the
vocoder's level shapes (HIP events over graph-free back-to-back launches), and the whole vocoder pass both ways.
    python tools/probe_chain.py [nchunks]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = _lib.require_device('cuda:0')
    g = torch.Generator().manual_seed(0)
    tot_c = tot_p = 0.0
    only = os.environ.get('ONLY')              # e.g. ONLY=128:11 -> that chain alone, 5 launches (for rocprofv3 --pmc)
    if only:
        c, k = (int(v) for v in only.split(':'))
        T = {128: 192, 64: 768, 32: 3072}[c]
        x = torch.randn(n, T, c, generator=g).to(BF).to(dev)
        convs = [((torch.randn(c, c, k, generator=g) / (c * k) ** 0.5).to(BF).float(), torch.randn(c, generator=g) * 0.1) for _ in range(6)]
        ws, nu, bias = ops.w_chain_pack(convs, dev)
        out = torch.zeros_like(x)
        for _ in range(5):
            ops.resblock_chain(x, ws, nu, bias, out, nbatch=n, t=T, c=c, taps=k, scale=1 / 3)
        torch.cuda.synchronize()
        return
    shapes = ((128, 192), (64, 768), (32, 3072))
    if os.environ.get('SHAPES'):
        shapes = tuple(sh for sh in shapes if str(sh[0]) in os.environ['SHAPES'].split(','))
    for c, T in shapes:
        x = torch.randn(n, T, c, generator=g).to(BF).to(dev)
        for k in ((11,) if os.environ.get('K11') else (3, 7, 11)):
            convs, dw = [], []
            for d in (1, 3, 5):
                for _ in range(2):
                    w = (torch.randn(c, c, k, generator=g) / (c * k) ** 0.5).to(BF).float()
                    b = torch.randn(c, generator=g) * 0.1
                    convs.append((w, b)); dw.append((ops.w_conv(w, dev), b.to(dev)))
            ws, nu, bias = ops.w_chain_pack(convs, dev)
            out = torch.zeros_like(x)
            tmp = [torch.empty_like(x), torch.empty_like(x)]

            def pairs():
                cur = x
                for di, d in enumerate((1, 3, 5)):
                    (w1, b1), (w2, b2) = dw[2 * di], dw[2 * di + 1]
                    nxt = out if di == 2 else tmp[di]
                    ops.resblock_pair(cur, w1, b1, w2, b2, nxt, nbatch=n, t=T, c=c, taps=k, dil=d, scale=(1 / 3 if di == 2 else 1.0))
                    cur = nxt
            tp = ev_time(pairs)
            tc = ev_time(lambda: ops.resblock_chain(x, ws, nu, bias, out, nbatch=n, t=T, c=c, taps=k, scale=1 / 3))
            gf = 6 * 2 * T * c * c * k * n / 1e9
            print('C=%3d k=%2d: 3 pairs %7.1f us (%6.0f TF/s)   chain %7.1f us (%6.0f TF/s)   x%.2f' % (
                c, k, tp * 1e6, gf / tp / 1e3, tc * 1e6, gf / tc / 1e3, tp / tc), flush=True)
            tot_c += tc; tot_p += tp
            if os.environ.get('PROF'):
                prof = torch.zeros(16, dtype=torch.int64, device=dev)
                ops.resblock_chain(x, ws, nu, bias, out, nbatch=n, t=T, c=c, taps=k, scale=1 / 3, prof=prof)
                torch.cuda.synchronize()
                pr = prof.cpu().tolist()
                nb, nt = pr[15], pr[13]
                per = [v / max(nt, 1) for v in pr[:13]]
                print('      per tile (shader clocks, wave 0): top %.0f | ' % per[0] +
                      ' '.join('K%d %.0f E%d %.0f' % (q, per[1 + 2 * q], q, per[2 + 2 * q]) for q in range(6)) +
                      ' | sum %.0f, block life/tile %.0f, tiles/block %.1f' % (sum(per), pr[14] / max(nt, 1), nt / max(nb, 1)), flush=True)
    print('levels 1-3, 9 residual blocks: pairs %.3f ms, chain %.3f ms' % (tot_p * 1e3, tot_c * 1e3))
    if os.environ.get('K11'):
        return
    # the C = 256 level: ifh_conv_ring256_bf16 against ifh_conv_bf16, per convolution
    x = torch.randn(n, 48, 256, generator=g).to(BF).to(dev)
    o1, o2 = torch.zeros_like(x), torch.zeros_like(x)
    for k in (3, 7, 11):
        for d in (1, 5):
            w = (torch.randn(256, 256, k, generator=g) / (256 * k) ** 0.5).to(BF).float()
            b = torch.randn(256, generator=g) * 0.1
            wd, bd = ops.w_conv(w, dev), b.to(dev)
            ws, nu, bias = ops.w_chain_pack([(w, b)], dev, unit_bytes=16384)
            t0 = ev_time(lambda: ops.conv(x, wd, bd, o1, nbatch=n, t_in=48, t_out=48, cin=256, n=256, taps=k, dil=d, pad=(k - 1) // 2 * d,
                                          pre_slope=0.1, resid=x))
            t1 = ev_time(lambda: ops.conv_ring256(x, ws, bias.reshape(-1), o2, nbatch=n, t=48, taps=k, dil=d, pre_slope=0.1, resid=x))
            gf = 2 * 48 * 256 * 256 * k * n / 1e9
            print('C=256 k=%2d d=%d: conv %7.1f us (%6.0f TF/s)   ring256 %7.1f us (%6.0f TF/s)   x%.2f  same bits: %s' % (
                k, d, t0 * 1e6, gf / t0 / 1e3, t1 * 1e6, gf / t1 / 1e3, t0 / t1, torch.equal(o1.view(torch.int16), o2.view(torch.int16))), flush=True)
    from infernos_amd.engines.vocoder import HifiGan
    from infernos_amd.weights import synth_state_dict
    voc = HifiGan(synth_state_dict('hifigan', 0), dev)
    vin = torch.randn(n, 12, 80, generator=g).to(BF).to(dev)
    for fused, ring in ((False, False), (True, False), (True, True)):
        voc.fused_chain, voc.fused_ring = fused, ring
        t = ev_time(lambda: voc(vin), n=5)
        print('vocoder pass, %d chunks, chain=%s ring256=%s: %.3f ms = %.0f TFLOP/s (%.1f %% of 2.5 PF)' % (
            n, fused, ring, t * 1e3, n * 3.28 / t / 1e3, n * 3.28 / t / 1e3 / 25))


if __name__ == '__main__':
    main()
