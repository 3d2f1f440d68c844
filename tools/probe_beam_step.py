"""One 5-beam step over Whisper-sized logits: python tools/probe_beam_step.py [rows] [vocab]  (us per ifh beam step, per-kernel via rocprofv3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 640
V = int(sys.argv[2]) if len(sys.argv) > 2 else 51865
K, B = 5, rows // 5
ld = (V + 3) // 4 * 4
st = ops.BeamState(B, K, 32, dev)
st.reset()
L = 4 + 32
toks = torch.zeros((L + 1, rows), dtype=torch.int32, device=dev)
pos = torch.full((1,), 4, dtype=torch.int32, device=dev)
logits = torch.randn(rows, ld, device=dev) * 3
sup = torch.zeros(V, device=dev); sup[::7] = float('-inf')
bsup = torch.zeros(V, device=dev); bsup[::5] = float('-inf')
def run():
    st.reset()
    ops.beam_step(logits, st, toks, pos, vocab=V, ld=ld, prompt_len=4, max_length=L, eos_id=50257, suppress=sup, begin_suppress=bsup)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f'beam step ({rows} rows x {V}): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
