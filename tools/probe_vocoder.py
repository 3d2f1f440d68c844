"""Runs P HiFi-GAN vocoder passes (256 chunks x 12 frames, as in bench.py's roofline leg) -- used under
rocprofv3 --pmc to collect FETCH_SIZE / WRITE_SIZE per pass:  python3 tools/probe_vocoder.py [passes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
voc = HifiGan(synth_state_dict('hifigan', 0), dev)
x = torch.randn(256, 12, 80, device=dev).to(torch.bfloat16)
for _ in range(P):
    voc(x)
torch.cuda.synchronize()
print('done', P)
