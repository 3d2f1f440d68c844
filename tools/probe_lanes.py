"""TTS-only throughput vs number of concurrent lanes (no front end): python tools/probe_lanes.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infernos_amd import _lib
from infernos_amd.pipeline import SpeechPipeline
dev = _lib.require_device('cuda:0')
LMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 6
OVERLAP = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
pipe = SpeechPipeline(64, dev, tts_lanes=LMAX)
pipe.prime()
sys.setswitchinterval(2e-4)
for L in range(1, LMAX + 1):
    NIT = 4
    def work(lane):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(pipe._lane_streams[lane]):
            for _ in range(NIT):
                pipe.synthesize(lane=lane, overlap=OVERLAP)
            torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(l,)) for l in range(L)]
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'overlap={int(OVERLAP)} lanes={L}: {L * NIT / dt:6.2f} TTS batches/s  ({dt / (L * NIT) * 1e3:6.1f} ms per batch of 64 calls, {dt / NIT * 1e3:6.1f} ms per lane-iteration)')
