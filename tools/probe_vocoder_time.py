#!/usr/bin/env python3
"""Time the HiFi-GAN vocoder pass (HIP events) at a chunk count, with and without the level kernel (IFH_NO_LEVEL=1).
    python tools/probe_vocoder_time.py [chunks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
x = torch.randn(N, 12, 80, device=dev).to(torch.bfloat16)
for _ in range(6):
    voc(x)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        voc(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print('vocoder pass, %d chunks, level=%s: %.3f ms = %.1f TFLOP/s = %.1f %% of 2.5 PF' % (N, voc.fused_level, t * 1e3, N * 3.280 / t / 1e3, N * 3.280 / t / 1e3 / 25))
