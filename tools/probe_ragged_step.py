"""GPU time of one ragged TTS decoder step (hipGraph replay) by row count.  python tools/probe_ragged_step.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.build()
from infernos_amd import _lib
from infernos_amd.tts import HelloSippyRTPipe, ContinuousTTS
from infernos_amd.engines.speecht5 import ragged_decoder_steps
from infernos_amd.weights import synth_state_dict

dev = _lib.require_device('cuda:0')
W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0, stop_bias=-20.0), 'hifigan': synth_state_dict('hifigan', 0),
     'amendment': synth_state_dict('amendment', 0)}
pp = HelloSippyRTPipe(dev, weights=W, processor=lambda **k: None, speaker_embeddings=[], output_sr=8000)
eng = ContinuousTTS(pp, max_rows=1024, max_text=64, row_bucket=128)
st = eng.st
st.active.fill_(1)
st.enc_len.fill_(64)
st.minmax[:, 1] = 640
masks = torch.zeros((16, 2, 256), dtype=torch.uint8, device=dev)
print('IFH_GEMM_DEC_ROWS=%s' % os.environ.get('IFH_GEMM_DEC_ROWS'))
for n in (128, 256, 384, 512, 640, 768, 1024):
    st.pos.fill_(100)
    for _ in range(3):
        ragged_decoder_steps(pp.model, st, masks, n)
        st.pos.fill_(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ragged_decoder_steps(pp.model, st, masks, n)
        st.pos.fill_(100)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 64
    print('rows %4d: %.3f ms per step = %.2f us per row-step' % (n, ms, ms * 1e3 / n))
