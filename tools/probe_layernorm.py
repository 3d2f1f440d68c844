"""LayerNorm over the encoder's stream: python tools/probe_layernorm.py [rows] [D]  (one / four rows per wave, and a plain copy of the same bytes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib, ops
dev = _lib.require_device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
out = torch.empty_like(x)
g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
byt = 2 * rows * D * 2
for form in ('1', '4', '1', '4'):
    os.environ['IFH_LN_RPW'] = form
    us = timeit(lambda: ops.layernorm(x, g, b, out, rows, D))
    print(f'rows per wave {form}: {us:.1f} us -> {byt/us/1e6:.2f} TB/s', flush=True)
us = timeit(lambda: out.copy_(x))
print(f'torch copy: {us:.1f} us -> {byt/us/1e6:.2f} TB/s')
