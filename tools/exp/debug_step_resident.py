"""the first k launches of a step, resident against the launch chain: which tensors / rows / columns differ (debugging aid).
python tools/debug_step_resident.py rows cw k0 k1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
from infernos_amd.engines import speecht5
from infernos_amd.engines.speecht5 import TTSRaggedState, ragged_decoder_steps
from test_step_resident_gpu import _model, _tensors, _randomise

dev = _lib.require_device('cuda:0')
pp = _model(dev)
model = pp.model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
cw = int(sys.argv[2]) if len(sys.argv) > 2 else 16
k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 5
k1 = int(sys.argv[4]) if len(sys.argv) > 4 else 13
LIMIT, COUNT = [0], [0]
_conv, _attn = ops.conv, ops.attn_decode


def lim(fn):
    def f(*a, **k):
        COUNT[0] += 1
        if COUNT[0] <= LIMIT[0]:
            return fn(*a, **k)
    return f


ops.conv, ops.attn_decode = lim(_conv), lim(_attn)
model.resident_cw = cw
for K in range(k0, k1 + 1):
    a = TTSRaggedState(model, max_rows=n, max_text=64)
    b = TTSRaggedState(model, max_rows=n, max_text=64)
    masks = _randomise(a, n, 64, seed=n)
    ta, tb = _tensors(a), _tensors(b)
    for k in ta:
        tb[k].copy_(ta[k])
    LIMIT[0], COUNT[0] = K, 0
    ragged_decoder_steps(model, a, masks[:1], n, nsteps=1, use_graphs=False, resident=False)
    COUNT[0] = 0
    ragged_decoder_steps(model, b, masks[:1], n, nsteps=1, resident=True)
    torch.cuda.synchronize()
    print('---- first %d launches (+ stop rule): err %s' % (K, b.step_ctx.status()[0]))
    for k in ta:
        x, y = ta[k], tb[k]
        if torch.equal(x.view(torch.uint8), y.view(torch.uint8)):
            continue
        d = (x.float() != y.float())
        if x.dim() >= 2:
            rows = d.reshape(x.shape[0], -1).any(1).nonzero().flatten().tolist()
            cols = d.reshape(-1, x.shape[-1]).any(0).nonzero().flatten().tolist()
            md = (x.float() - y.float()).abs().max().item()
            print('%-12s shape %s: %d elements differ (max |d| %.3g); rows %s%s; cols %s%s' % (
                k, tuple(x.shape), int(d.sum()), md, rows[:12], '...' if len(rows) > 12 else '', cols[:12], '...' if len(cols) > 12 else ''))
        else:
            print('%-12s %d differ: %s' % (k, int(d.sum()), d.nonzero().flatten().tolist()[:16]))
    if K == 6:
        for r in (0, 7):
            print('row', r, 'klen', int(ta['pos'][r]))
            print(' chain   ', ta['att'][r, :16].float().tolist())
            print(' resident', tb['att'][r, :16].float().tolist())
            print(' v[key0] ', ta['self_kv[0]'][r, 0, 768:784].float().tolist())
            print(' q       ', ta['q'][r, :16].float().tolist())
