"""Whisper encoder alone (B windows): python tools/probe_encoder.py [B] [whisper_tiny|whisper_base]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.engines.whisper import Whisper
from infernos_amd.weights import synth_state_dict
dev = _lib.require_device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fam = sys.argv[2] if len(sys.argv) > 2 else 'whisper_tiny'
GF = {'whisper_tiny': 36.9e9, 'whisper_base': 87.4e9}[fam]
w = Whisper(synth_state_dict(fam, 0), dev)
mel = torch.randn(B, 80, 3000, device=dev)
for _ in range(2): w.encode(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): w.encode(mel)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f'{fam} encoder B={B}: {dt*1e3:.2f} ms -> {B*GF/dt/1e12:.0f} TF/s')
