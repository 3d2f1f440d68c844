import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from infernos_amd.engines.whisper import Whisper
from infernos_amd.features import WhisperLogMel
from infernos_amd.synth import synth_utterance
from infernos_amd.audio import get_resampler
from infernos_amd.weights import synth_state_dict
gd = 'tests/golden'
g = np.load(os.path.join(gd, 'whisper_beam.npz'))
meta = json.load(open(os.path.join(gd, 'whisper_beam_meta.json')))
dev = torch.device('cuda:0')
sd = synth_state_dict('whisper_tiny', 0)
model = Whisper(sd, dev)
rs = get_resampler(8000, 16000, str(dev))
x8 = torch.from_numpy(np.stack([synth_utterance(s, 10.0) for s in meta['audio_seeds']])).to(dev)
mel = WhisperLogMel(80, dev)(rs(x8))
enc = model.encode(mel)
prompt = torch.tensor([meta['prompt']] * 2, dtype=torch.int32)
V = 51865
for ci in [int(a) for a in sys.argv[1:]] or [8]:
    c = meta['cases'][ci]
    sup = torch.zeros(V); sup[50257:] = float('-inf'); sup[c['eos']] = 0.0
    bs = None
    if c['begin']:
        bs = torch.zeros(V); bs[c['begin']] = float('-inf')
    for ug in (False, True):
        toks, lens, scores, nsp = model.generate_beam(enc, prompt, c['n_new'], beams=c['beams'], eos_id=c['eos'],
                                                      length_penalty=c['lp'], suppress=sup, begin_suppress=bs, check_every=4, use_graphs=ug)
        st = model._dec(2 * c['beams'], c['beams'])['beam_state']
        print('case', ci, c, 'graphs', ug)
        print(' toks', toks.tolist(), 'lens', lens.tolist(), 'scores', scores.tolist())
        print(' fin_scores', st.fin_scores.tolist(), 'is_fin', st.is_fin.tolist(), 'fin_len', st.fin_len.tolist())
        print(' fin_seqs', st.fin_seqs[:, :2].tolist())
        print(' run', st.run_scores.tolist(), 'unsat', st.unsat.tolist(), 'alive', st.alive[:20].tolist())
    print(' ref', g['seq%d' % ci].tolist(), g['len%d' % ci].tolist(), g['score%d' % ci].tolist())
