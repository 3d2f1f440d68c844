"""Where the STT stage's time goes (standalone): python tools/probe_stt.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infernos_amd import _lib
from infernos_amd.pipeline import SpeechPipeline
from infernos_amd.synth import synth_utterance
from infernos_amd.codecs import G711Codec
dev = _lib.require_device('cuda:0')
N = 64
pipe = SpeechPipeline(N, dev, tts_lanes=1)
codec = G711Codec().to(dev)
x = np.stack([synth_utterance(1000 + i, 10.0) for i in range(N)])
ul = np.frombuffer(codec.encode(torch.from_numpy(x)), dtype=np.uint8).reshape(N, 500, 160)
frames = torch.from_numpy(np.ascontiguousarray(ul.transpose(1, 0, 2))).to(dev)
def T(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r
for _ in range(3):
    pipe.reset_calls(); ch = pipe.ingest(frames); pipe.stt(ch)
pipe.reset_calls()
t_ing, chunks = T(lambda: (pipe.reset_calls(), pipe.ingest(frames))[1])
import copy
def merge():
    merged = []
    for lst in chunks:
        head = copy.copy(lst[0]); merged.append(head.audio)
    lens8 = torch.tensor([m.numel() for m in merged], dtype=torch.int32)
    L8 = max(int(lens8.max()), 1)
    x8 = torch.zeros((N, L8), dtype=torch.float32, device=dev)
    for i, m in enumerate(merged):
        x8[i, :m.numel()] = m
    return x8, lens8
t_merge, (x8, lens8) = T(merge)
t_rs, x16 = T(lambda: pipe.up(x8, lens=lens8))
lens16 = (lens8 * 2).to(dev)
t_mel, mel = T(lambda: pipe.logmel(x16, lens=lens16))
t_enc, enc = T(lambda: pipe.whisper.encode(mel))
t_gen, _ = T(lambda: pipe.whisper.generate(enc, pipe.prompt, 32, no_speech_id=50362))
print(f'ingest {t_ing:.2f} ms | merge/pad {t_merge:.2f} | resample {t_rs:.2f} | logmel {t_mel:.2f} | encoder {t_enc:.2f} | generate(32) {t_gen:.2f}')
