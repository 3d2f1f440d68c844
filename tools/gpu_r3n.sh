cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dsp_gpu.py -q -m gpu -k logmel 2>&1 | tail -5
IFH_LOGMEL_PROF=1 timeout 300 python tools/probe_logmel.py 3 128 2>&1 | tail -4
python3 - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from infernos_amd import _lib
from infernos_amd.features import WhisperLogMel
dev=_lib.require_device('cuda:0')
lm=WhisperLogMel(80, dev)
N=128
x=torch.randn(N,480000,device=dev)*0.1
lens=torch.full((N,),480000,dtype=torch.int32,device=dev)
out=torch.empty(N,80,3000,device=dev)
for _ in range(3): lm.raw(x,lens=lens,out=out)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): lm.raw(x,lens=lens,out=out)
e1.record(); torch.cuda.synchronize()
t=e0.elapsed_time(e1)/10*1e-3
print('logmel 128 windows: %.1f us = %.0f GB/s = %.1f %% of 8 TB/s' % (t*1e6, N*2.88e6/t/1e9, N*2.88e6/t/1e9/80))
PY
