"""One HiFi-GAN level's residual blocks alone, as the vocoder runs them (for rocprofv3 --pmc / --stats and HIP-event timing):
    python3 tools/probe_voc_level.py <C: 256|128|64|32> [nchunks] [launches] [taps: 0 = the level's three blocks | 3 | 7 | 11]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
C = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1280; L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = _lib.require_device('cuda:0')
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
lvl = {256: 0, 128: 1, 64: 2, 32: 3}[C]
T = 48 * 4 ** lvl
u = torch.randn(n, T, C, device=dev).to(torch.bfloat16)
B = voc._buffers(n, 12)
def run():
    voc.level(lvl, u, B, n, T, C)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(L):
    run()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / L * 1e-3
gf = 6 * 2 * T * C * C * 21 * n / 1e9
print('level C=%d, %d chunks: %.1f us = %.0f TF/s (%.1f %% of 2.5 PF)' % (C, n, t * 1e6, gf / t / 1e3, gf / t / 1e3 / 25))
