"""GPU time of one ragged TTS decoder step: the launch chain (hipGraph replay) against the resident step (csrc/step.hip) by row
count and workgroups per cluster.  python tools/probe_step_resident.py"""
import sys, os
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
rows = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [128, 256, 384, 512, 640, 1024]
cws = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [8, 16]


def timed(n, **kw):
    st.pos.fill_(100)
    for _ in range(2):
        ragged_decoder_steps(pp.model, st, masks, n, **kw)
        st.pos.fill_(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ragged_decoder_steps(pp.model, st, masks, n, **kw)
        st.pos.fill_(100)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 64


for n in rows:
    line = 'rows %4d: chain %.3f ms' % (n, timed(n, resident=False))
    for cw in cws:
        pp.model.resident_cw = cw
        line += ' | resident cw=%d: %.3f ms' % (cw, timed(n, resident=True))
        err, _ = st.step_ctx.status()
        if err:
            line += ' (WAIT BOUND HIT)'
    print(line, flush=True)

# per-phase times of cluster 0 (100 MHz clock): wait for the cluster / work, by phase
n = rows[-1] if len(rows) == 1 else 512
pp.model.resident_cw = cws[-1]
timed(n, resident=True)
prog = st.progs[(3, 0.5, st.ncalls & 1, n)] if (3, 0.5, st.ncalls & 1, n) in st.progs else next(iter(st.progs.values()))
st.step_ctx.prof()
reps = 20
for _ in range(reps):
    prog.run(st.step_ctx, cw=cws[-1], prof=True, write_through=pp.model.resident_wt)
    st.pos.fill_(100)
pr = st.step_ctx.prof()
print('per phase at %d rows, cw %d (us): wait / work' % (n, cws[-1]))
tw = tb = 0.0
for i in range(prog.nphase):
    w, b = pr[i][0] / reps, pr[i][1] / reps
    tw += w
    tb += b
    print('  %2d: %6.2f %6.2f' % (i, w, b), end='' if i % 4 != 3 else '\n')
print('\n  sum wait %.1f us, work %.1f us' % (tw, tb))
gb = st.step_ctx.gemm_breakdown
ng = max(1, gb[6])
print('  GEMM phases (%d in %d launches), mean us: table+prefetch issue %.2f | wait %.2f | stats+image %.2f | tasks %.2f | epilogue %.2f | partials+arrive %.2f'
      % (ng, reps, gb[0] / ng, gb[1] / ng, gb[2] / ng, gb[3] / ng, gb[4] / ng, gb[5] / ng))
