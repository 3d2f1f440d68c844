"""Runs P log-mel launches (N x 30 s windows, as in bench.py's roofline_logmel leg) -- used under rocprofv3 --pmc to
collect FETCH_SIZE / WRITE_SIZE per launch:  python3 tools/probe_logmel.py [passes] [windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.features import WhisperLogMel
dev = _lib.require_device('cuda:0')
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lm = WhisperLogMel(80, dev)
x = torch.randn(N, 480000, device=dev) * 0.1
lens = torch.full((N,), 480000, dtype=torch.int32, device=dev)
out = torch.empty(N, 80, 3000, device=dev)
for _ in range(P):
    lm.raw(x, lens=lens, out=out)        # the transform as the STT stage runs it (normalisation fused into conv1's layout change)
torch.cuda.synchronize()
print('done', P)
