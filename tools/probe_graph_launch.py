"""Host cost vs GPU time of the decoder-step hipGraphs (is the decode loop launch-bound on the host?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.pipeline import SpeechPipeline, _make_state
dev = _lib.require_device('cuda:0')
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipe = SpeechPipeline(64, dev, tts_lanes=1, tts_group=G)
pipe.prime()
pp = pipe.tts
state = _make_state(pp, pipe.text_ids.repeat(G, 1), pipe.speakers.repeat(G, 1))
print('rows', 64 * G)
st = state.dev
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    pp.decode_chunk(state)            # 16 step graphs, asynchronous
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'decode_chunk (16 steps): host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms -> {1e3*(t2-t0)/16:.3f} ms/step, host {1e3*(t1-t0)/16:.3f} ms/step')
# a single captured step graph replayed back to back
g = next(iter(st.graphs.values()))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'50 replays of one step graph: host {1e3*(t1-t0)/50:.3f} ms/replay, GPU-complete {1e3*(t2-t0)/50:.3f} ms/replay')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f'GPU event time per replay {e0.elapsed_time(e1)/50:.3f} ms')
