cd $GRAFT_REPO_ROOT
python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range())
for p in (-2,-1,0,1):
    try:
        s=torch.cuda.Stream(priority=p); print(p, 'ok', s.priority)
    except Exception as e: print(p, 'err', e)
"
timeout 600 python -m pytest tests/test_dsp_gpu.py -q -m gpu -k logmel 2>&1 | tail -3
IFH_LOGMEL_PROF=1 timeout 300 python tools/probe_logmel.py 2>&1 | tail -12
