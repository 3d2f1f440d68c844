#!/usr/bin/env python3
"""Time ifh_resblock_level_bf16 (csrc/level.hip) against the ifh_resblock_chain_bf16 launches it replaces at the C = 32 level of
the vocoder (T = 3072 rows per chunk), per block and for the whole level, and check the bits.
    python tools/probe_level.py [nchunks]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
from infernos_amd import _lib, ops  # noqa: E402

BF = torch.bfloat16


def ev_time(fn, n=8):
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
    c, T = 32, 3072
    x = torch.randn(n, T, c, generator=g).to(BF).to(dev)
    blocks = []
    for k in (3, 7, 11):
        convs = [((torch.randn(c, c, k, generator=g) / (c * k) ** 0.5).to(BF).float(), torch.randn(c, generator=g) * 0.1) for _ in range(6)]
        ws, nu, bias = ops.w_chain_pack(convs, dev)
        blocks.append((k, ws, nu, bias))
    ref, out = torch.zeros_like(x), torch.zeros_like(x)

    def chain_all():
        for j, (k, ws, nu, bias) in enumerate(blocks):
            ops.resblock_chain(x, ws, nu, bias, ref, nbatch=n, t=T, c=c, taps=k, scale=1 / 3, accumulate=j > 0)

    def level_all():
        ops.resblock_level(x, [(k, ws, bias) for k, ws, nu, bias in blocks], out, nbatch=n, t=T, c=c, scale=1 / 3)
    chain_all(); level_all(); torch.cuda.synchronize()
    print('bits equal:', bool(torch.equal(ref.view(torch.int16), out.view(torch.int16))))
    fl = lambda k: 6 * 2.0 * n * T * c * c * k
    for k, ws, nu, bias in blocks:
        tc = ev_time(lambda: ops.resblock_chain(x, ws, nu, bias, ref, nbatch=n, t=T, c=c, taps=k, scale=1 / 3))
        tl = ev_time(lambda: ops.resblock_level(x, [(k, ws, bias)], out, nbatch=n, t=T, c=c, scale=1 / 3))
        print('C=%d k=%2d: chain %7.1f us = %6.1f TF/s   level %7.1f us = %6.1f TF/s   x%.2f' % (
            c, k, tc * 1e6, fl(k) / tc / 1e12, tl * 1e6, fl(k) / tl / 1e12, tc / tl))
    # round 6: conv_post + tanh folded into the level launch against level + ifh_hifigan_post_bf16
    pw = (torch.randn(7, 32, generator=g) / 15).to(dev)
    audio, audio2 = torch.empty(n, T, dtype=BF, device=dev), torch.empty(n, T, dtype=BF, device=dev)
    wsb = torch.empty(ops.level_ws_bytes(), dtype=torch.uint8, device=dev)

    def level_then_post():
        level_all()
        _lib.check(_lib.lib().ifh_hifigan_post_bf16(ops._addr(out), ops._addr(pw), 0.03, ops._addr(audio), n, T, 0.01, _lib.stream_ptr(dev)), 'post')

    def level_post():
        ops.resblock_level(x, [(k, ws, bias) for k, ws, nu, bias in blocks], None, nbatch=n, t=T, c=c, scale=1 / 3, post=(pw, 0.03, 0.01, audio2, wsb))
    level_then_post(); level_post(); torch.cuda.synchronize()
    print('folded conv_post bits equal:', bool(torch.equal(audio.view(torch.int16), audio2.view(torch.int16))))
    t2, t1 = ev_time(level_then_post), ev_time(level_post)
    print('C=%d level + conv_post: two launches %7.1f us   folded %7.1f us   x%.2f' % (c, t2 * 1e6, t1 * 1e6, t2 / t1))
    tc, tl = ev_time(chain_all), ev_time(level_all)
    ft = sum(fl(k) for k in (3, 7, 11))
    print('C=%d level (3 blocks): chain launches %7.1f us = %6.1f TF/s   one level launch %7.1f us = %6.1f TF/s   x%.2f' % (
        c, tc * 1e6, ft / tc / 1e12, tl * 1e6, ft / tl / 1e12, tc / tl))


if __name__ == '__main__':
    main()
