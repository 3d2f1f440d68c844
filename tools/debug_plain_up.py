#!/usr/bin/env python3
"""where the plain-upsampler vocoder pass first differs from the round-5 pass (level means and upsampler outputs, layer by layer)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from infernos_amd import _lib
from infernos_amd.engines.vocoder import HifiGan
from infernos_amd.weights import synth_state_dict
BF = torch.bfloat16
dev = _lib.require_device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sd = synth_state_dict('hifigan', 0)
g = torch.Generator().manual_seed(n)
chunks = (torch.randn(n, 12, 80, generator=g) * 0.8).to(BF).float()
voc_in = ((chunks - sd['mean']) / sd['scale']).to(BF).to(dev)
voc = HifiGan(sd, dev)
cp, co = {}, {}
a = voc(voc_in, cache=cp).clone()
voc.plain_up = voc.fused_post = False
b = voc(voc_in, cache=co).clone()
torch.cuda.synchronize()
P, O = cp[(n, 12, 'plain')], co[(n, 12)]
import torch.nn.functional as F
t, c = 48, 256
for i in range(3):
    xg = P['xg%d' % i][1:1 + n * (t + 2)].view(n, t + 2, c)[:, :t]          # rows b (t + 2) + 1 + r -> view row r
    want = F.leaky_relu(O['xn%d' % i].float(), 0.1).to(BF)
    d = (xg.float() - want.float()).abs()
    print('level %d mean (lrelu): differing elements %d of %d, max %g' % (i, int((d > 0).sum()), d.numel(), float(d.max())))
    ug = P['ug%d' % (i + 1)][1:1 + n * (t + 2)].view(n, t + 2, 2 * c)[:, :t].reshape(n, 4 * t, c // 2)
    uo = O['u%d' % (i + 1)]
    d = (ug.float() - uo.float()).abs()
    print('upsampler %d out: differing elements %d of %d, max %g' % (i + 1, int((d > 0).sum()), d.numel(), float(d.max())))
    t, c = t * 4, c // 2
d = (a.float() - b.float()).abs()
print('audio: differing %d of %d max %g' % (int((d > 0).sum()), d.numel(), float(d.max())))
