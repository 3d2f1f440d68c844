"""Diagnostic: does a row's result depend on how many rows share the ragged TTS step?  Runs the same 128-row batch alone
(n = 128) and next to filler batches (n = 256, 512) and compares the decoder frames and the rendered audio bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib, ops
from infernos_amd.tts import HelloSippyRTPipe, ContinuousTTS
from infernos_amd.weights import synth_state_dict

dev = _lib.require_device('cuda:0')
W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0, stop_bias=-20.0), 'hifigan': synth_state_dict('hifigan', 0),
     'amendment': synth_state_dict('amendment', 0)}
pp = HelloSippyRTPipe(dev, weights=W, processor=lambda **k: None, speaker_embeddings=[], output_sr=8000)
fixed = torch.randint(0, 2, (16, 2, 256), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(dev)
pp.mask_source = lambda n: fixed
pp.model.use_graphs = False
N, T = 128, 64
g = torch.Generator().manual_seed(1)
ids = torch.randint(4, 80, (N, T), generator=g, dtype=torch.int32)
spk = torch.randn(N, 512, generator=g)
lens = torch.full((N,), T, dtype=torch.int32)
res = {}
for fill in (0, 1, 3):
    eng = ContinuousTTS(pp, max_rows=128 * (fill + 1), max_text=T, row_bucket=128)
    grp = eng.submit(ids, lens, spk, max_calls=2, want_ulaw=True)
    for f in range(fill):
        eng.submit(torch.randint(4, 80, (N, T), generator=g, dtype=torch.int32), lens, torch.randn(N, 512, generator=g), max_calls=2, want_ulaw=True)
    outs = []
    for c in range(2):
        eng.step()
        torch.cuda.synchronize()
        st = eng.st
        par = c & 1
        outs.append(dict(spec=st.spec[par][:N].clone(), post=st.post[par][:N].clone(), kv0=st.self_kv[0][:N, :16 * (c + 1)].clone(),
                         t3=st.t3[:N].clone(), audio=eng.render_bufs[eng._bucket() if eng.live else 128 * (fill + 1)]['out'][par][:N].clone(),
                         ulaw=grp.ulaw[:, c * 4096:(c + 1) * 4096].clone()))
    res[fill] = outs
    del eng
for fill in (1, 3):
    for c in range(2):
        for k in res[0][c]:
            a, b = res[0][c][k], res[fill][c][k]
            same = torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
            nd = int((a != b).sum()) if not same else 0
            print('fill %d call %d %-6s %s  differing elements %d / %d' % (fill, c, k, 'same' if same else 'DIFF', nd, a.numel()))
