"""Debug probe: dump the log-mel of a seeded batch (run twice, with and without IFH_LOGMEL_V1, then diff)."""
import sys
import numpy as np
import torch
from infernos_amd.features import WhisperLogMel

out = sys.argv[1]
rng = np.random.default_rng(5)
B, L = 3, 480000
x = (rng.standard_normal((B, L)) * 0.1).astype(np.float32)
x[1, 200000:] *= 1e-4
lens = torch.tensor([480000, 300000, 123457], dtype=torch.int32)
lm = WhisperLogMel(80, torch.device('cuda:0'))
xd = torch.from_numpy(x).cuda()
y = lm(xd, lens=lens).float().cpu().numpy()
for rep in range(20):
    y2 = lm(xd, lens=lens).float().cpu().numpy()
    if not np.array_equal(y, y2):
        print('NONDETERMINISTIC at rep', rep, np.abs(y - y2).max())
        break
else:
    print('20 repeats bit-identical')
np.save(out, y)
if len(sys.argv) > 2:
    a = np.load(sys.argv[2])
    d = np.abs(a - y)
    print('max diff', d.max())
    bad = np.argwhere(d > 1e-3)
    print('n bad', len(bad))
    if len(bad):
        print('bad batch idx', np.unique(bad[:, 0]))
        print('bad mel idx', np.unique(bad[:, 1])[:40])
        fr = np.unique(bad[:, 2])
        print('bad frames', fr[:60], '...', fr[-10:], 'count', fr.size)
        print('frame mod 128', np.unique(fr % 128)[:64])
