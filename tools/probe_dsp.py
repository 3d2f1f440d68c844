import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from infernos_amd import _lib
from infernos_amd.features import WhisperLogMel
from infernos_amd.codecs import G711Codec
from infernos_amd.audio import get_resampler
dev = _lib.require_device('cuda:0')
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
lm = WhisperLogMel(80, dev)
for B, L in ((64, 480000), (64, 160000), (256, 480000)):
    x = torch.randn(B, 480000, device=dev) * 0.1
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    out = torch.empty(B, 80, 3000, device=dev)
    t = timeit(lambda: lm(x, lens=lens, out=out))
    print(f'logmel B={B} len={L}: {t*1e6:.1f} us  -> {B*2.88e6/t/1e9:.1f} GB/s algorithmic, {t/B*1e6:.2f} us/window')
c = G711Codec().to(dev)
n = 1 << 28
b = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
o = torch.empty(n, dtype=torch.float32, device=dev)
L_ = _lib.lib()
t = timeit(lambda: L_.ifh_g711_decode_u8_f32(_lib.ptr(b), _lib.ptr(o), n, _lib.stream_ptr(dev)))
print(f'g711 decode {n} B: {t*1e6:.1f} us -> {n*5/t/1e9:.1f} GB/s')
t = timeit(lambda: L_.ifh_g711_encode_f32_u8(_lib.ptr(o), _lib.ptr(b), n, _lib.stream_ptr(dev)))
print(f'g711 encode {n}: {t*1e6:.1f} us -> {n*5/t/1e9:.1f} GB/s')
rs = get_resampler(8000, 16000, str(dev))
x = torch.randn(256, 240000, device=dev)
t = timeit(lambda: rs(x))
print(f'resample 8k->16k 256x240000: {t*1e6:.1f} us -> {256*240000*12/t/1e9:.1f} GB/s')
