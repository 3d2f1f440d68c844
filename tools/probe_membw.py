"""Memory-rate floor for vocoder-sized tensors (tuning aid): fill / copy / add of 25 MB bf16 buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (25, 100, 400):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).to(torch.bfloat16); r = torch.randn(n, device=dev).to(torch.bfloat16)
    o = torch.empty_like(x)
    t = timeit(lambda: o.zero_()); print(f'{mb} MB zero_: {t*1e6:.1f} us {mb/1024/t/1e3*1.048576:.2f} TB/s written')
    t = timeit(lambda: o.copy_(x)); print(f'{mb} MB copy_: {t*1e6:.1f} us {2*mb*1.048576e6/t/1e12:.2f} TB/s r+w')
    t = timeit(lambda: torch.add(x, r, out=o)); print(f'{mb} MB add:   {t*1e6:.1f} us {3*mb*1.048576e6/t/1e12:.2f} TB/s r+r+w')
    t = timeit(lambda: x.sum()); print(f'{mb} MB sum:   {t*1e6:.1f} us {mb*1.048576e6/t/1e12:.2f} TB/s read')
