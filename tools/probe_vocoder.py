"""Runs P HiFi-GAN vocoder passes (N chunks x 12 frames, as in bench.py's roofline leg: N = 4 x calls x grouped
cycles = 1024 by default) -- used under rocprofv3 --pmc to collect FETCH_SIZE / WRITE_SIZE per pass:
python3 tools/probe_vocoder.py [passes] [chunks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
x = torch.randn(N, 12, 80, device=dev).to(torch.bfloat16)
for _ in range(P):
    voc(x)
torch.cuda.synchronize()
print('done', P)
