#!/usr/bin/env python3
"""bench.py -- real-time-factor x concurrent calls of the Infernos speech hot path on MI355X.

Default workload = BASELINE.json configs[2] ("C3", the largest single-GPU configuration): 128 concurrent synthetic
calls per GPU, each a 10 s G.711 mu-law utterance in 20 ms / 160 B frames, carried through
    ingest (decode + 8k->16k + VAD windows) -> Whisper-BASE STT (log-mel, encoder, 32 tokens; 5-beam search by default, --stt-beam 1 = greedy) -> T2T stub
    -> SpeechT5 + HiFi-GAN + Amendment TTS (10 infer() calls = 5.12 s of speech at T_text = 64) -> 16k->8k -> mu-law,
bf16 models with seeded random weights (no checkpoints offline), synthetic audio.
`--config C2` = 64 calls + Whisper-tiny (configs[1]); `--config C4` = the per-GPU share of configs[3]: 256 calls +
Whisper-tiny.  At N = 1 the default run also measures C2 and the C4 share briefly and reports them as extra keys.

One "step" = one such utterance cycle for every call of every rank.  value = call-seconds of inbound audio fully
processed per wall second = (calls x 10 s) / step time: the aggregate real-time factor.  Inputs are resident in HBM
when the timed region starts.  With N > 1 GPUs calls are sharded (weak scaling); rank 0 scatters the frame matrix and
gathers the encoded output over RCCL inside the timed region.

Batches are formed ACROSS CALLS ONLY (one utterance per call per batch: --tts-group 1); consecutive cycles are
stage-pipelined the way a serving loop is: --front-lanes ingest+STT lanes (5-beam Whisper decode) run ahead of the TTS
stage, which (--tts-mode continuous, the default) is ONE ragged decode batch: every utterance batch in flight (at most
--tts-lanes) is a set of row slots of the same decoder step, each row at its own decoder position; batches join at an
infer() boundary and leave when their utterances end (infernos_amd/tts.py:ContinuousTTS).  --tts-mode lanes is round
2's schedule (one engine clone and launch chain per batch).  Every cycle's whole work -- and the fill and drain of this
pipeline -- is inside the timed K steps; outputs are byte-identical to the sequential per-batch schedule
(tests/test_pipeline_gpu.py, tests/test_continuous_tts_gpu.py).

p50/p99 tick latency is measured INSIDE the timed region, under that load: a tick thread hands one [N,160] mu-law
frame matrix (pinned host memory) to the boundary every 20 ms and waits for that tick's ingest outputs (H2D ->
ifh_ingest_tick -> VAD window/decision when one completes -> ifh_mux_encode_f32_u8 of the next 20 ms of real TTS
output rows -> D2H) to be back on the host.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|C4] [--no-cpu-baseline] [--no-extra-configs]
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

UTT_SECONDS = 10.0
TICKS = 500
VOCODER_GFLOP_PER_CHUNK = 3.280          # dense 2xMAC per 12-frame chunk
LOGMEL_BYTES_PER_WINDOW = 2.88e6         # 480000*4 read + 80*3000*4 written
PEAK_BF16_TFLOPS = 2500.0                # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E spec

CONFIGS = {   # name -> (calls per GPU, Whisper family, description)
    'C3': (128, 'whisper_base', 'C3: %d concurrent synthetic 10 s G.711 calls per GPU through ingest+VAD -> Whisper-base STT '
                                '(32 tokens) -> T2T stub -> SpeechT5+HiFi-GAN TTS (10 infer calls, T_text 64) -> mu-law'),
    'C2': (64, 'whisper_tiny', 'C2: %d concurrent synthetic 10 s G.711 calls per GPU through ingest+VAD -> Whisper-tiny STT '
                               '(32 tokens) -> T2T stub -> SpeechT5+HiFi-GAN TTS (10 infer calls, T_text 64) -> mu-law'),
    'C4': (256, 'whisper_tiny', 'C4 per-GPU share: %d concurrent synthetic 10 s G.711 calls per GPU through ingest+VAD -> '
                                'Whisper-tiny STT (32 tokens) -> SpeechT5+HiFi-GAN TTS (10 infer calls, T_text 64) -> mu-law'),
}


def make_frames(ncalls, first_call, codec_encode):
    """[TICKS, ncalls, 160] u8 of the SURVEY.md 8(d) synthetic utterances (seed 1000+call)."""
    from infernos_amd.synth import synth_utterance
    x = np.stack([synth_utterance(1000 + first_call + i, UTT_SECONDS) for i in range(ncalls)])
    ulaw = codec_encode(x)                                     # [ncalls, 80000] u8
    return np.ascontiguousarray(ulaw.reshape(ncalls, TICKS, 160).transpose(1, 0, 2))


def cpu_baseline(family='whisper_base', ncalls=16, threads=32, beams=5):
    """The oracle (CPU restatement, kind "port") timed on this host, bounded: one 10 s cycle for `ncalls` calls batched
    (the reference's own TTS cap is 8).  Two TTS legs: the reference's dtype (bf16, `maybe_half`,
    HelloSippyRTPipe.py:57; 3 of the 10 infer() calls measured, the other 7 priced at their mean) and fp32 (all 10
    measured); STT is fp32 in both (the reference's CTranslate2 int8 engine is absent).  `value` is the FASTER leg.
    Threads capped (a small-batch graph on 128 threads is slower than on 32)."""
    from oracle import dsp as odsp, nn as onn
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    torch.manual_seed(0)
    nthreads = max(1, min(threads, os.cpu_count() or 1))
    prev = torch.get_num_threads()
    torch.set_num_threads(nthreads)
    nheads = 8 if family == 'whisper_base' else 6
    try:
        x = np.stack([synth_utterance(1000 + i, UTT_SECONDS) for i in range(ncalls)])
        sd_w = synth_state_dict(family, 0)
        sd_t = synth_state_dict('speecht5_tts', 0, stop_bias=-20.0)
        sd_v, sd_a = synth_state_dict('hifigan', 0), synth_state_dict('amendment', 0)
        t0 = time.perf_counter()
        pcm = odsp.g711_decode(odsp.g711_encode(x))
        x16 = odsp.resample(pcm[:, 8000:72000], 8000, 16000)          # ~8 s of speech per call, as the VAD emits
        mel = torch.from_numpy(odsp.logmel(x16))
        with torch.no_grad():                     # the decode of the timed region: beam search (or greedy with --stt-beam 1)
            prompt = torch.tensor([[50258, 50259, 50359, 50363]] * ncalls)
            if beams > 1:
                onn.whisper_beam(sd_w, mel, prompt, 32, nheads, beams, 50257)
            else:
                onn.whisper_greedy(sd_w, mel, prompt, 32, nheads)
        t_stt = time.perf_counter() - t0
        g = torch.Generator().manual_seed(2000)
        ids = torch.randint(4, 80, (ncalls, 64), generator=g)
        spk = torch.randn(ncalls, 512, generator=g)
        masks = (torch.rand(16, 2, 256, generator=g) < 0.5).to(torch.uint8)
        legs = {}
        for name, dtype, nmeas in (('bf16', torch.bfloat16, 3), ('fp32', torch.float32, 10)):
            sdt = onn._cast(sd_t, dtype)
            sdv, sda = onn._cast(sd_v, dtype), onn._cast(sd_a, dtype)
            t1 = time.perf_counter()
            with torch.no_grad():
                st = onn.TTSState(sdt, ids, torch.ones_like(ids).int(), spk, dtype=dtype)
                t_enc = time.perf_counter() - t1
                t2 = time.perf_counter()
                for _ in range(nmeas):
                    a = onn.tts_infer(sdt, sdv, sda, st, masks, dtype=dtype)
                    odsp.g711_encode(odsp.resample(a.float().numpy(), 16000, 8000))
                t_inf = (time.perf_counter() - t2) / nmeas
            legs[name] = dict(total=t_stt + t_enc + 10 * t_inf, t_enc=t_enc, t_inf=t_inf, measured_infer_calls=nmeas)
    finally:
        torch.set_num_threads(prev)
    best = min(legs, key=lambda k: legs[k]['total'])
    return {'value': round(ncalls * UTT_SECONDS / legs[best]['total'], 3), 'unit': 'x real-time (call-seconds/s)', 'cores': nthreads,
            'kind': 'port', 'tts_dtype_of_value': best,
            'legs_x_realtime': {k: round(ncalls * UTT_SECONDS / v['total'], 3) for k, v in legs.items()},
            'sample': ('%d calls, one 10 s cycle on the oracle (oracle/): ingest + log-mel + %s 32 tokens ' % (ncalls, family)) +
                      ('beam search, %d beams, ' % beams if beams > 1 else 'greedy, ') +
                      'fp32 (%.2f s); TTS leg bf16 (the reference\'s dtype): SpeechT5 encoder %.2f s + 10 x infer()+resample+mu-law at %.2f s '
                      '(3 measured); TTS leg fp32: encoder %.2f s + 10 x %.2f s (all measured); value = the faster leg (%s); torch '
                      'threads=%d of os.cpu_count()=%s' % (t_stt, legs['bf16']['t_enc'], legs['bf16']['t_inf'], legs['fp32']['t_enc'],
                                                           legs['fp32']['t_inf'], best, nthreads, os.cpu_count())}


class TickProbe(threading.Thread):
    """The per-tick boundary under load (SURVEY.md 8d metric ii): every `period` seconds one [N,160] mu-law frame matrix goes
    host -> device -> ifh_ingest_tick (+ VAD window step and decision whenever 768 samples complete) and the next 20 ms of
    TTS output goes ifh_mux_encode_f32_u8 -> host; the latency of a tick is hand-over until both are back on the host."""

    def __init__(self, dev, n, host_frames, tts_pcm, period=0.020, vad_model='recurrent'):
        super().__init__(daemon=True)
        from infernos_amd.frontend import CallTable
        from infernos_amd.pipeline import BatchedVAD
        self.dev, self.n, self.period = dev, n, period
        self.host_frames = host_frames                              # pinned u8 [TICKS, n, 160]
        self.tts_pcm = tts_pcm                                      # device f32 [n, S]: real TTS output rows
        with torch.cuda.device(dev):
            # the detector of the per-tick path = the throughput path's (--vad-model): by default the recurrent network shaped like the
            # reference's (conv front end + 2 x LSTM(64), state [2,N,64] x 2 carried per call: csrc/vadnet.hip) with the distilled weights --
            # Silero's are not obtainable offline
            from infernos_amd.vad import RecurrentVADModel
            self.calls = CallTable(n, dev)
            self.vad = BatchedVAD(n, dev, model=RecurrentVADModel(dev, weights='distilled') if vad_model == 'recurrent' else None)
            # the per-tick path is the real-time one: its few small kernels go to a high-priority hardware queue so that they
            # do not wait behind a lane's whole queued decode graph
            # IFH_TICK_CUS="first,n" (tuning switch): a CU-range stream instead -- a hardware queue of its own, so that a tick's
            # launches do not stand behind the decode steps the TTS engine has queued on the shared high-priority queue
            tc = os.environ.get('IFH_TICK_CUS', '')
            if tc:
                from infernos_amd import _lib as _l
                first, ncu = (int(x) for x in tc.split(','))
                self.stream = _l.cu_range_stream(dev, first, ncu)
            else:
                self.stream = torch.cuda.Stream(device=dev, priority=-1)
            self.slots = torch.arange(n, dtype=torch.int32, device=dev)
            self.dfr = torch.empty((n, 160), dtype=torch.uint8, device=dev)
            self.p8, self.p16 = torch.empty((n, 160), device=dev), torch.empty((n, 320), device=dev)
            self.present = torch.ones((n, 1), dtype=torch.uint8, device=dev)
            self.ndiv = torch.ones(n, dtype=torch.int32, device=dev)
            self.enc = torch.empty((n, 160), dtype=torch.uint8, device=dev)
            self.has = torch.empty(n, dtype=torch.uint8, device=dev)
        from infernos_amd.frontend import TickEgress
        self.egress = TickEgress(n, 160, dev)                        # the product's hand-back: D2H copy + marker packet (frontend.py)
        self.host_out = self.egress.host
        self.lat, self.windows, self._nb = [], 0, 0
        self.worst = (0.0, {})
        # extra timing events (marker packets) between the tick's operations, for experiments: bit 0 after the H2D copy, 1 after the
        # ingest kernel, 2 after the mix kernel.  (Round 4: ANY one marker takes the slowest tick from 40-60 ms to 4-8 ms; the one that
        # ships sits behind the D2H copy, inside frontend.TickEgress -- the product's hand-back, which this probe calls.)
        self.evmask = int(os.environ.get('IFH_TICK_EVS', '0'))
        self._halt = threading.Event()

    def tick(self, t):
        from infernos_amd import _lib
        L, dev, n = _lib.lib(), self.dev, self.n
        a = time.perf_counter()
        with torch.cuda.stream(self.stream):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(self.stream)                                 # (completes as soon as the queue reaches it: nothing is in front)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            self.dfr.copy_(self.host_frames[t % TICKS], non_blocking=True)
            if self.evmask & 1: evs[0].record(self.stream)           # after the H2D copy
            self.calls.tick(self.dfr, self.slots, self.p8, self.p16, want_ready=False)
            if self.evmask & 2: evs[1].record(self.stream)           # after the ingest kernel
            b = time.perf_counter()
            self._nb += 160
            if self._nb >= 768:
                self._nb -= 768
                self.vad.step(self.calls.win)                       # includes its host sync: the VAD decision is on the host
                self.windows += 1
            c = time.perf_counter()
            S = self.tts_pcm.size(1)
            off = (t * 160) % (S - 160)
            trk = self.tts_pcm[:, off:off + 160].contiguous()
            _lib.check(L.ifh_mux_encode_f32_u8(_lib.ptr(trk), _lib.ptr(self.present), _lib.ptr(self.ndiv), n, 1, 160,
                                               _lib.ptr(self.enc), _lib.ptr(self.has), _lib.stream_ptr(dev)), 'ifh_mux_encode_f32_u8')
            if self.evmask & 4: evs[2].record(self.stream)           # after the VAD window (if any) and the mix kernel
            self.egress.push(self.enc)                              # D2H copy + the marker packet behind it
            d = time.perf_counter()
            ev1.record(self.stream)
            self.egress.wait()
        e = time.perf_counter()
        ms = (e - a) * 1e3
        if ms > self.worst[0]:        # where the slowest tick spent its time: queueing ingest, VAD window (+ its sync), queueing the mix, final wait;
            # gpu_first_to_last_ms = from the moment the hardware queue reached the tick's first packet to its last kernel's end
            self.worst = (ms, {'queue_ingest_ms': round((b - a) * 1e3, 2), 'vad_window_ms': round((c - b) * 1e3, 2),
                               'queue_mix_ms': round((d - c) * 1e3, 2), 'final_wait_ms': round((e - d) * 1e3, 2),
                               'gpu_first_to_last_ms': round(ev0.elapsed_time(ev1), 2),
                               'gpu_after_h2d_ingest_vadmix_d2h_ms': [round(ev0.elapsed_time(x), 2) if self.evmask >> i & 1 else None for i, x in enumerate(evs)]})
        return ms

    def run(self):
        torch.cuda.set_device(self.dev)
        t, t0 = self._t_next, time.perf_counter() - self._t_next * self.period
        while not self._halt.is_set():
            due = t0 + t * self.period
            now = time.perf_counter()
            if due > now:
                time.sleep(due - now)
            self.lat.append(self.tick(t))
            t += 1

    def stop(self):
        self._halt.set()
        self.join(10)


def time_steps(pipe, frames_for, nsteps, warmup, world, pipelined, egress, dry, dev, probe=None):
    if warmup:
        pipe.run_steps(frames_for, warmup, pipelined=pipelined, on_cycle=egress)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if probe is not None:
        probe.start()
    from infernos_amd import _lib
    mark = os.environ.get('IFH_TRACE_MARK') == '1'        # profiling aid: a recognisable kernel brackets the timed region
    if mark:
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
    c0 = _lib.CALLS[0]
    g0 = _lib.CountedGraph.captures[0]
    e0 = (sum(e.calls_run for e in pipe.ctts_all), sum(e.rows_run for e in pipe.ctts_all)) if pipe.ctts is not None else None
    p0 = [dict(e.prof) for e in pipe.ctts_all] if pipe.ctts is not None else []
    if pipe.ctts is not None:
        pipe.ctts.st.replay_prof = [0.0, 0]
    t0 = time.perf_counter()
    res = pipe.run_steps(frames_for, nsteps, pipelined=pipelined, on_cycle=egress)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if mark:
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
    time_steps.launches_per_step = (_lib.CALLS[0] - c0) / max(1, nsteps)
    time_steps.graph_captures = _lib.CountedGraph.captures[0] - g0
    time_steps.engine_prof = None if e0 is None else {k: round(sum(e.prof.get(k, 0.0) - q.get(k, 0.0) for e, q in zip(pipe.ctts_all, p0)), 4) for k in ('admit', 'steps', 'render', 'ends_wait', 'book')}
    time_steps.replay_prof = None if e0 is None else list(pipe.ctts.st.replay_prof)
    if pipe.ctts is not None:
        pipe.ctts.st.replay_prof = None
    time_steps.engine = None if e0 is None else (sum(e.calls_run for e in pipe.ctts_all) - e0[0], sum(e.rows_run for e in pipe.ctts_all) - e0[1])
    if probe is not None:
        probe.stop()
    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if dry else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(tmax.item()), res



def c5_share(dev, args, build, n_sessions=64, prompt_len=192, first_sentence=16, reply_tokens=48, turns=5, llm_family='qwen2_1p5b'):
    """One GPU's share of BASELINE configuration 5 (512 AI-attendant sessions over 8 GPUs = 64 per GPU): a turn is
    STT (the C3 front end and decode on the caller's 10 s utterance) -> LLM (Qwen2.5-1.5B shape, random weights, a
    192-token chat context per session, reply streamed) -> TTS.  Turn latency = end of the caller's speech (utterance
    ingested, VAD closed) to the first 512 ms of synthesised audio of the reply's first sentence (taken as 16 tokens),
    with all 64 sessions' turns arriving together (the worst case for a batched worker).  Sequential stages, one stream."""
    from infernos_amd.engines.qwen2 import Qwen2
    from infernos_amd.weights import QWEN2_CONFIGS, qwen2_random_on_device
    cfg = QWEN2_CONFIGS[llm_family]
    pipe, frames_all, frames_for, _, _ = build('C3', n_sessions)
    llm = Qwen2(qwen2_random_on_device(cfg, dev), cfg, dev, max_tokens=prompt_len + reply_tokens + 8)
    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(10, cfg['vocab'] - 10, (prompt_len - (i % 9),), generator=g).tolist() for i in range(n_sessions)]
    n_infer = pipe.n_infer
    lat, parts = [], []
    for turn in range(turns + 2):
        fl = pipe.front_lanes[0]
        pipe.reset_calls(fl)
        chunks = pipe.ingest(frames_for(turn), fl)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pipe.stt(chunks, fl)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        marks = []
        llm.generate(prompts, reply_tokens, on_tokens=lambda t: marks.append(time.perf_counter()))
        t_first = marks[first_sentence - 1]
        t2 = time.perf_counter()
        pipe.n_infer = 1                                   # the first 512 ms chunk of the reply
        pipe.synthesize()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        pipe.n_infer = n_infer
        if turn >= 2:                                      # the first two turns load kernels and capture the graphs
            lat.append((t1 - t0) + (t_first - t1) + (t3 - t2))
            parts.append((t1 - t0, t_first - t1, t2 - t1, t3 - t2))
    pa = np.array(parts)
    res = {'workload': '%d AI-attendant sessions per GPU (1/8 of configuration 5): Whisper-base STT + Qwen2.5-1.5B-shaped LLM '
                       '(random weights, %d-token contexts) + SpeechT5/HiFi-GAN TTS, turns of all sessions arriving together' %
                       (n_sessions, prompt_len),
           'p50_turn_latency_ms': round(float(np.percentile(lat, 50)) * 1e3, 1),
           'turn_latency_definition': 'end of caller speech -> first 512 ms of synthesised reply audio (first sentence = %d tokens)' % first_sentence,
           'stt_ms': round(float(pa[:, 0].mean()) * 1e3, 1), 'llm_first_sentence_ms': round(float(pa[:, 1].mean()) * 1e3, 1),
           'llm_reply_ms': round(float(pa[:, 2].mean()) * 1e3, 1), 'tts_first_chunk_ms': round(float(pa[:, 3].mean()) * 1e3, 1),
           'llm_decode_tokens_per_s': round(n_sessions * (reply_tokens - first_sentence) / float((pa[:, 2] - pa[:, 1]).mean()), 0),
           'turns': turns}
    pipe.close()
    del pipe, llm, frames_all
    torch.cuda.empty_cache()
    return res


def tts_harness(dev, n_sessions=50, tokens=64, max_rows=256):
    """The reference's own TTS measurement (HelloSippyTTSRT/HelloSippyRTPipeTest.py:201-208, 226-235): 50 sessions each
    `say` one prompt at once through the worker (here InfernTTSWorker(continuous=True) behind TTSSession, as the SIP side drives it);
    per session  time_to_first_frame = first audio chunk delivered - request made,  time_to_last_frame likewise, and
    rtr = (time_to_last_frame - time_to_first_frame) / (number_of_frames / 8000)  (< 1 = faster than real time).  Prompts are token
    ids (no tokenizer offline), T_text = 64, seeded weights whose stop head is held off (stop_bias -20: the maxlen arm ends the
    utterance after 640 decoder steps = 10.2 s of speech)."""
    from infernos_amd.audio import AudioChunk
    from infernos_amd.muxer import ASMarkerNewSent
    from infernos_amd.tts import InfernTTSWorker, TTSRequest, TTSSession
    from infernos_amd.weights import synth_state_dict

    class IdsProcessor:
        def __call__(self, text, return_tensors='pt'):
            return {'input_ids': torch.tensor([[int(t) for t in text.split()]], dtype=torch.long)}
    W = {'speecht5_tts': synth_state_dict('speecht5_tts', 0, stop_bias=-20.0), 'hifigan': synth_state_dict('hifigan', 0),
         'amendment': synth_state_dict('amendment', 0)}
    g = torch.Generator().manual_seed(7)
    voices = [torch.randn(1, 512, generator=g) for _ in range(8)]
    w = InfernTTSWorker('en', 8000, dev, weights=W, processor=IdsProcessor(), speaker_embeddings=voices, continuous=True,
                        max_rows=max_rows, max_text=tokens)
    w.max_batch_size = n_sessions
    w.start()
    res = {}
    try:
        for rnd in range(2):                              # round 0 loads the kernels and captures the graphs; round 1 is reported
            done = threading.Event()
            stat = [dict(first=None, last=None, frames=0) for _ in range(n_sessions)]
            left = [n_sessions]
            lock = threading.Lock()

            def so(i, t0):
                def f(chunk):
                    now = time.perf_counter() - t0[0]
                    st = stat[i]
                    if isinstance(chunk, AudioChunk):
                        if st['first'] is None:
                            st['first'] = now
                        st['frames'] += int(chunk.audio.numel())
                    elif isinstance(chunk, ASMarkerNewSent):
                        st['last'] = now
                        with lock:
                            left[0] -= 1
                            if left[0] == 0:
                                done.set()
                return f
            sess, t0 = [TTSSession(w, None) for _ in range(n_sessions)], [0.0]
            for i, s_ in enumerate(sess):
                s_.start(so(i, t0))
            prompts = [' '.join(str(int(v)) for v in torch.randint(4, 80, (tokens,), generator=g)) for _ in range(n_sessions)]
            t0[0] = time.perf_counter()
            for i, s_ in enumerate(sess):
                s_.say(TTSRequest(prompts[i], speaker_id=i % 8))
            if not done.wait(600):
                raise RuntimeError('tts_harness: sessions did not finish')
            ttff = np.array([st['first'] for st in stat]) * 1e3
            rtr = np.array([(st['last'] - st['first']) / (st['frames'] / 8000.0) for st in stat])
            res = {'definition': 'HelloSippyTTSRT/HelloSippyRTPipeTest.py:201-208,226-235: per session time_to_first_frame and '
                                 'rtr = (t_last - t_first) / (frames / 8000), all sessions speaking at once',
                   'sessions': n_sessions, 'text_tokens': tokens, 'speech_seconds_per_session': round(float(np.mean([st['frames'] for st in stat])) / 8000.0, 2),
                   'time_to_first_frame_ms': {'p50': round(float(np.percentile(ttff, 50)), 1), 'p99': round(float(np.percentile(ttff, 99)), 1)},
                   'rtr': {'p50': round(float(np.percentile(rtr, 50)), 4), 'p99': round(float(np.percentile(rtr, 99)), 4)},
                   'engine': 'InfernTTSWorker(continuous=True) behind TTSSession; one ragged decode batch of %d rows' % n_sessions}
    finally:
        w.stop()
    return res


def self_launch(ngpus):
    """`python bench.py --gpus N` without a launcher: one rank per GPU through torch.distributed.run on 127.0.0.1 and a
    free port.  stdout of the ranks is passed through (rank 0 prints the one JSON line), stderr too."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    # compile once here (hipcc / make only: nothing in this process touches the GPU), so that the ranks find the library built
    from infernos_amd import build as _b
    _b.build(verbose=False)
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // ngpus)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=24)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='C3')
    ap.add_argument('--calls-per-gpu', type=int, default=0, help='override the configuration\'s call count')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the short C2 / C4-share runs after the main one')
    ap.add_argument('--no-tick-probe', action='store_true', help='do not run the per-tick latency thread inside the timed region')
    ap.add_argument('--breakdown', action='store_true', help='print per-stage wall times to stderr')
    ap.add_argument('--tts-mode', choices=['continuous', 'lanes'], default='continuous',
                    help='continuous: ONE ragged TTS decode batch over every utterance batch in flight (rows at different decoder '
                         'positions, joined at infer() boundaries); lanes: one engine clone and launch chain per batch (round 2)')
    ap.add_argument('--tts-lanes', type=int, default=5, help='utterance batches that may be in flight in the TTS stage together')
    ap.add_argument('--front-lanes', type=int, default=4, help='ingest+STT lanes (cycles k, k+1, k+2 in flight together)')
    ap.add_argument('--tts-group', type=int, default=1, help='utterance cycles of the SAME calls synthesised as one TTS batch '
                    '(> 1 is an offline-throughput mode: a live call cannot have utterance k+1 before k has been spoken)')
    ap.add_argument('--no-tts-overlap', action='store_true', help='render on the lane stream instead of a second stream per lane')
    ap.add_argument('--stt-beam', type=int, default=5, help='Whisper decode: 5 = beam search at the width of the reference default '
                    'engine (ctranslate2 defaults, InfernSTTWorker.py:61-75; transformers formulation of the search, ctranslate2 '
                    'parity unpinned); 1 = greedy (its torch engine)')
    ap.add_argument('--c5-only', action='store_true', help='run only the configuration-5 per-GPU share (STT -> LLM -> TTS turn latency) and print it')
    ap.add_argument('--tts-harness-only', action='store_true', help='run only the reference\'s own TTS measurement (50 sessions: time to first frame, rtr) and print it')
    ap.add_argument('--c5-sessions', type=int, default=64)
    ap.add_argument('--c5-llm', default='qwen2_1p5b', help='infernos_amd.weights.QWEN2_CONFIGS entry (random weights of that shape)')
    ap.add_argument('--cu-reserve', type=int, default=None,
                    help='CUs the persistent (one workgroup per CU) vocoder kernels leave to the decode chains and the per-tick kernels '
                         '(SpeechPipeline.cu_reserve, default 96 / IFH_CU_RESERVE; 0 = none)')
    ap.add_argument('--vad-model', choices=['recurrent', 'energy'], default='recurrent',
                    help='probability model of the VAD step in the throughput path AND the tick probe: recurrent = conv + 2 x LSTM(64) network on '
                         'every 768-sample window of every call with the per-call state carried (csrc/vadnet.hip, distilled weights; the '
                         'reference runs its network there: Core/VAD/SileroVAD.py:78-80); energy = the stateless energy rule')
    ap.add_argument('--no-pipeline', action='store_true', help='run the stages of consecutive cycles strictly one after another')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (never exec) and before this
        # process has made any GPU call; rank 0's JSON line is relayed, the exit code is the launcher's
        # (replica fan-out: Cluster/InfernBenchActor.py:214-221)
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import __graft_entry__ as ge
    ge.build()        # every rank: one compiles under build/.lock, the others wait for it (infernos_amd/build.py)
    # IFH_DRYRUN_ONE_GPU=1: every rank uses cuda:0 and the collectives run over gloo (host-staged) -- lets the
    # N>1 code path (sharding, collective order, stage threads) be exercised on a single-GPU box
    dry = os.environ.get('IFH_DRYRUN_ONE_GPU') == '1'
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    g_in = g_out = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        dist.barrier()
        # ingress (scatter, front-end thread) and egress (gather, main thread) get their own communicators
        g_in, g_out = dist.new_group(), dist.new_group()
    gpu_ids = [local_rank]
    if world > 1:                                       # which device every rank sits on (the line names them)
        gl = [None] * world
        dist.all_gather_object(gl, local_rank)
        gpu_ids = [int(x) for x in gl]
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.shard import gather_rows, scatter_frames
    from infernos_amd.codecs import G711Codec

    codec = G711Codec().to(dev)

    def enc(x):
        return np.frombuffer(codec.encode(torch.from_numpy(x)), dtype=np.uint8).reshape(x.shape)

    def build(cfg, n_local, beam=None):
        _, family, _ = CONFIGS[cfg]
        pipe = SpeechPipeline(n_local, dev, whisper_family=family, tts_lanes=args.tts_lanes, tts_overlap=not args.no_tts_overlap,
                              tts_group=args.tts_group, front_lanes=args.front_lanes, stt_beam=args.stt_beam if beam is None else beam,
                              tts_mode=args.tts_mode, cu_reserve=args.cu_reserve, vad_model=args.vad_model)
        n_total = n_local * world
        # every rank builds its own rows for the N=1 path; with N>1 rank 0 holds all rows and scatters
        if world == 1:
            frames_all = torch.from_numpy(make_frames(n_local, 0, enc)).to(dev)
        else:
            frames_all = torch.from_numpy(make_frames(n_total, 0, enc)).to(dev) if rank == 0 else None

        def frames_for(k):
            # with N>1 the ingress rank scatters this cycle's frame block over RCCL (inside the timed region)
            return scatter_frames(frames_all, n_total, TICKS, dev, group=g_in) if world > 1 else frames_all
        egress = (lambda r: gather_rows(r['ulaw'], n_total, group=g_out)) if world > 1 else None
        # priming (untimed, not part of the W warm-up steps): two sequential cycles load every kernel and
        # capture the hipGraphs of the decode loops, so the timed steps replay them
        last = pipe.run_steps(frames_for, 2, pipelined=False)
        pipe.prime(frames_for(0))
        return pipe, frames_all, frames_for, egress, last

    if args.tts_harness_only:
        if rank == 0:
            print(json.dumps({'tts_harness': tts_harness(dev)}))
        return
    if args.c5_only:
        if rank == 0:
            print(json.dumps({'C5_share': c5_share(dev, args, build, n_sessions=args.c5_sessions, llm_family=args.c5_llm,
                                                   turns=max(1, min(args.steps, 5)))}))
        return
    n_local = args.calls_per_gpu or CONFIGS[args.config][0]
    n_total = n_local * world
    pipe, frames_all, frames_for, egress, primed = build(args.config, n_local)
    probe = None
    if rank == 0 and not args.no_tick_probe:
        host_frames = (frames_all[:, :n_local] if world > 1 else frames_all).cpu().pin_memory()
        ul = primed['ulaw']                                         # u8 [n_local, 40960]: the real TTS rows of a primed cycle
        tts_pcm = torch.empty(ul.shape, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().ifh_g711_decode_u8_f32(_lib.ptr(ul.contiguous()), _lib.ptr(tts_pcm), ul.numel(), _lib.stream_ptr(dev)),
                   'ifh_g711_decode_u8_f32')
        probe = TickProbe(dev, n_local, host_frames, tts_pcm, vad_model=args.vad_model)
        for t in range(8):                                          # load its kernels before the timed region
            probe.tick(t)
        probe._t_next = 8
        probe.lat.clear()
        probe.worst = (0.0, {})
    dt, res = time_steps(pipe, frames_for, args.steps, args.warmup, world, not args.no_pipeline, egress, dry, dev, probe)
    ms_per_step = dt / args.steps * 1e3
    value = n_total * UTT_SECONDS / (dt / args.steps)
    launches_per_cycle, engine = time_steps.launches_per_step, time_steps.engine
    # strictly sequential stage times of one cycle (ingest / STT / TTS one after another on an otherwise idle GPU): their sum
    # against ms_per_step says what the stage pipelining + continuous batching buy
    stage_ms = None
    if rank == 0 and world == 1:
        pipe.reset_calls()
        fr = frames_for(0)
        ts = []
        for _ in range(2):
            pipe.reset_calls()
            torch.cuda.synchronize(); a = time.perf_counter()
            ch = pipe.ingest(fr); torch.cuda.synchronize(); b = time.perf_counter()
            pipe.stt(ch); torch.cuda.synchronize(); c = time.perf_counter()
            pipe.synthesize(); torch.cuda.synchronize(); d = time.perf_counter()
            ts.append(((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3))
        stage_ms = {'ingest': round(ts[-1][0], 2), 'stt': round(ts[-1][1], 2), 'tts': round(ts[-1][2], 2),
                    'sum': round(sum(ts[-1]), 2)}

    if args.breakdown and rank == 0 and getattr(pipe, 'stage_wall', None):
        sw = pipe.stage_wall
        print('pipelined job wall ms: front %.1f (n=%d)  tts %.1f (n=%d)' % (
            1e3 * sum(sw['front']) / max(1, len(sw['front'])), len(sw['front']),
            1e3 * sum(sw['tts']) / max(1, len(sw['tts'])), len(sw['tts'])), file=sys.stderr)
    if args.breakdown and rank == 0 and stage_ms:
        print('sequential stage ms: %r' % (stage_ms,), file=sys.stderr)
    out = None
    if rank == 0:
        # ---- stage timings + rooflines (HIP events on the launch stream = torch's current stream)
        def ev_time(fn, n=3):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e-3
        # the vocoder launch group of the timed region: 4 chunks per TTS row of a render pass -- with the continuous engine the
        # rows of every utterance batch in flight (mean over the timed engine calls, in whole batches)
        tts_rows = n_local * max(1, args.tts_group)
        if engine is not None and engine[0]:
            tts_rows = max(tts_rows, int(round(engine[1] / engine[0] / tts_rows)) * tts_rows)
        nchunks = 4 * tts_rows
        # the kernel rooflines are taken on the WHOLE chip (in the timed region the vocoder ran on 256 - cu_reserve CUs by choice)
        vocoder_cus = torch.cuda.get_device_properties(dev).multi_processor_count - pipe.cu_reserve
        _lib.check(_lib.lib().ifh_set_cu_budget(0), 'ifh_set_cu_budget')
        voc_in = torch.randn(nchunks, 12, 80, device=dev).to(torch.bfloat16)
        t_voc = ev_time(lambda: pipe.tts.vocoder(voc_in), n=5)
        ach_tf = nchunks * VOCODER_GFLOP_PER_CHUNK / t_voc / 1e3
        x16 = torch.randn(n_local, 480000, device=dev) * 0.1
        lens = torch.full((n_local,), 480000, dtype=torch.int32, device=dev)
        mel_out = torch.empty(n_local, 80, 3000, device=dev)
        # the transform as the STT stage runs it (raw log10 plane + per-window maximum; the clamp/scale is fused into the
        # layout change in front of Whisper's conv1, infernos_amd/features.py): reads 1.92 MB, writes 0.96 MB per window
        t_mel = ev_time(lambda: pipe.logmel.raw(x16, lens=lens, out=mel_out), n=5)
        ach_gbs = n_local * LOGMEL_BYTES_PER_WINDOW / t_mel / 1e9
        del voc_in, x16, mel_out

        def latest(name):
            """profiles/rNN_<name> of the newest round that has one"""
            for r in range(9, 0, -1):
                f = os.path.join(ROOT, 'profiles', 'r%02d_%s' % (r, name))
                if os.path.exists(f):
                    return f
            return os.path.join(ROOT, 'profiles', 'r00_' + name)

        def pmc(name, per):
            """HBM bytes per launch from a separate rocprofv3 --pmc run (profiles/), scaled by units if the sizes differ"""
            f = latest(name)
            if not os.path.exists(f):
                return None
            pj = json.load(open(f))
            return pj['hbm_bytes_per_pass'] * per / pj['chunks_per_pass']

        def pmc_field(name, key):
            f = latest(name)
            if not os.path.exists(f):
                return None
            return json.load(open(f)).get(key)
        voc_pmc, lm_pmc = os.path.relpath(latest('vocoder_pmc.json'), ROOT), os.path.relpath(latest('logmel_pmc.json'), ROOT)
        # the TTS decode chain (the chain that sets the cycle): a decoder step of the continuous engine at the rows it ran with, inside
        # the timed region (host seconds the engine thread spent waiting for its 16-step calls -- it waits for the device after step 8,
        # 'steps', and for the end flags after step 16, 'ends_wait' -- per step: the chain's device time under load) and alone on the
        # idle GPU (16 steps on the engine's own state and stream, nothing live: every row is computed, none advances)
        step_ms = None
        if engine is not None and engine[0] and getattr(time_steps, 'engine_prof', None):
            from infernos_amd.engines.speecht5 import ragged_decoder_steps
            eng = pipe.ctts
            nrows = eng._bucket() if eng.live else max(16, int(round(engine[1] / engine[0] / 16.0)) * 16)
            nrows = min(nrows, eng.st.R)
            with torch.cuda.device(dev), torch.cuda.stream(eng.main):
                masks = eng.pp.mask_source(16).to(dev).contiguous()
                eng.st.active.zero_()
                for _ in range(2):
                    ragged_decoder_steps(eng.pp.model, eng.st, masks, nrows, nsteps=16, threshold=eng.pp.threshold)
                eng.main.synchronize()
                ta = time.perf_counter()
                for _ in range(4):
                    ragged_decoder_steps(eng.pp.model, eng.st, masks, nrows, nsteps=16, threshold=eng.pp.threshold)
                eng.main.synchronize()
                idle = (time.perf_counter() - ta) / 64 * 1e3
            step_ms = {'in_pipeline': round((time_steps.engine_prof['steps'] + time_steps.engine_prof['ends_wait']) / (engine[0] * 16) * 1e3, 3),
                       'idle_gpu': round(idle, 3),
                       'rows_idle_measurement': nrows, 'steps_per_cycle': round(engine[0] * 16 / max(1, args.steps), 1),
                       'launches_per_step': 55,
                       'host_ms_in_graph_launch_per_step': (round(time_steps.replay_prof[0] / max(1, time_steps.replay_prof[1]) * 1e3, 3)
                                                            if getattr(time_steps, 'replay_prof', None) else None)}
        lat = np.array(probe.lat) if probe is not None and probe.lat else None
        out = {
            'metric': 'real-time-factor x concurrent calls (STT+TTS on 20 ms G.711 frames)',
            'value': round(value, 2), 'unit': 'x real-time (call-seconds/s)', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': (CONFIGS[args.config][2] % n_local).replace('ingest+VAD', 'ingest+VAD (LSTM detector on every window)'
                                                                                if args.vad_model == 'recurrent' else 'ingest+VAD (energy rule)'),
                       'calls_per_gpu': n_local, 'calls_total': n_total,
                       'utterance_seconds': UTT_SECONDS, 'weights': 'seeded random (HF shapes)',
                       'parallelism': 'calls sharded %d/GPU, models replicated; RCCL scatter/gather of frames/output' % n_local,
                       'batching': 'across calls only (one utterance per call per batch)' if args.tts_group == 1 else
                                   'OFFLINE mode: %d consecutive utterances of the same calls per TTS batch' % args.tts_group,
                       'vocoder_cus_in_timed_region': vocoder_cus,
                       'stage_pipelining': not args.no_pipeline, 'front_lanes': args.front_lanes, 'tts_lanes': args.tts_lanes,
                       'tts_rows_per_batch': n_local * args.tts_group, 'tts_mode': args.tts_mode,
                       'tts_rows_per_decode_step': (round(engine[1] / engine[0], 1) if engine is not None and engine[0] else
                                                    n_local * args.tts_group),
                       'vad_model': ('conv + 2 x LSTM(64) network (csrc/vadnet.hip: k_vadnet on every 768-sample window of every call inside '
                                     'ifh_ingest_block_net, per-call state [2,N,64] x 2 carried; weights distilled from the energy rule on synthetic '
                                     'call audio -- Silero v3.1 itself, Core/VAD/SileroVAD.py:44, is not obtainable offline)'
                                     if args.vad_model == 'recurrent' else 'stateless energy rule (ifh_vad_energy_prob)'),
                       'stt_decode': ('beam search, %d beams (%d decode rows), 32 tokens' % (args.stt_beam, n_local * args.stt_beam))
                                     if args.stt_beam > 1 else 'greedy, 32 tokens'},
            'rccl_version': (list(torch.cuda.nccl.version()) if world > 1 and not dry else None),
            'gpu_ids': gpu_ids,
            'launches_per_cycle': round(launches_per_cycle, 1),
            'launches_note': 'calls into stream-taking C-ABI entry points per utterance cycle inside the timed region, hipGraph replays '
                             'counted by the launches they hold (infernos_amd/_lib.py:CALLS)',
            'sequential_stage_ms': stage_ms,
            'graph_captures_in_timed_region': getattr(time_steps, 'graph_captures', None),
            'tts_engine_host_seconds': getattr(time_steps, 'engine_prof', None),
            'tts_decode_step_ms': step_ms,
            'stt_audio_seconds_per_call': round(float(res['stt_seconds'].mean()), 3),
            'tts_samples_per_call': int(res['tts_samples'].float().mean()),
            'roofline': {'kernel': 'HiFi-GAN vocoder pass (%d chunks x 12 frames per launch group)' % nchunks,
                         'bound': 'mfma', 'achieved': round(ach_tf, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach_tf / PEAK_BF16_TFLOPS, 4), 'traffic': pmc('vocoder_pmc.json', nchunks),
                         'traffic_note': 'HBM bytes per pass from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, ' + voc_pmc + ' (1280-chunk pass; scaled by chunks if the pass sizes differ; the x2 over-counts the 8-byte-per-lane loads of the residual-block kernels, see the note in that file)',
                         'seconds_per_vocoder_pass': t_voc},
            'roofline_logmel': {'kernel': 'k_logmel_fft (%d x 30 s windows -> raw log-mel [80,3000] f32 + window maximum)' % n_local, 'bound': 'hbm',
                                'achieved': round(ach_gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                'frac': round(ach_gbs / PEAK_HBM_GBS, 4), 'traffic': pmc('logmel_pmc.json', n_local),
                                'traffic_note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, ' + lm_pmc,
                                'valu_frac': pmc_field('logmel_pmc.json', 'valu_frac'),
                                'valu_note': 'SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x shader clocks of the launch) of k_logmel_fft from the separate --pmc passes (' + lm_pmc + ': sq_reading) -- the vector ALUs '
                                             'are busy this share of the time a CU is occupied; the kernel is issue-bound on its FFT arithmetic, not on HBM',
                                'bound_note': 'read this fraction as a VALU-bound kernel, not as an HBM kernel at 17 %: its HBM traffic is 1.04 x algorithmic and it moves them at '
                                              'the rate its arithmetic allows.  Per thread and 32-frame tile the compiled kernel issues ~1 900 vector instructions (windowing + DFT-25 ~390, '
                                              'DFT-8 + power ~500, sparse mel + log10 ~700, staging ~300, packed f32 throughout: the 5-point butterflies are already the symmetric '
                                              '(Winograd-form) ones); at a 100 % busy VALU that is ~160 us per 128 windows against 46 us of HBM time, i.e. <= 29 % of the HBM roof for '
                                              'this formulation; measured VALU busy 59 %.  An fp32-accurate DFT-as-GEMM needs >= 163 us of matrix time (3-way bf16 split) -- no better',
                                'seconds': t_mel},
        }
        if lat is not None:
            out.update({'p50_tick_latency_ms': round(float(np.percentile(lat, 50)), 4),
                        'p99_tick_latency_ms': round(float(np.percentile(lat, 99)), 4),
                        'worst_tick_ms': round(probe.worst[0], 3), 'worst_tick_split': probe.worst[1],
                        'tick_latency_note': '%d ticks of [%d,160] paced at 20 ms INSIDE the timed region (H2D -> ingest_tick -> VAD '
                                             'window+decision on %d of them -> mux_encode of real TTS rows -> D2H, host to host), '
                                             'while the TTS/STT lanes were running; the hand-back is the product\'s frontend.TickEgress (D2H copy + one marker packet behind it); the probe\'s VAD model is the one of the throughput path (config.vad_model: '
                                             '%s), so the tick figure and `value` contain the same detector; Silero v3.1 itself '
                                             '(Core/VAD/SileroVAD.py:44) is not available offline' % (len(lat), n_local, probe.windows, args.vad_model)})
    # ---- the other single-GPU configurations, briefly (N = 1 only: extra keys, not the headline)
    if world == 1 and not args.no_extra_configs and not args.calls_per_gpu:
        pipe.close()
        del pipe, frames_all, probe
        torch.cuda.empty_cache()
        extra = {}
        runs = [(c, c, None, args.vad_model) for c in ('C2', 'C4', 'C3') if c != args.config]
        if args.stt_beam > 1:            # SURVEY.md 8(d)'s workload as written (greedy, 32 tokens): keeps rounds comparable
            runs.append((args.config + '_greedy', args.config, 1, args.vad_model))
        if args.vad_model != 'energy':   # the headline's configuration with the round 1-4 energy rule in the VAD step: keeps rounds comparable
            runs.append((args.config + '_energy_vad', args.config, None, 'energy'))
        k2 = max(4, min(args.steps, 12))
        # every extra configuration in a fresh child process (started, not exec'ed: this process keeps the GPU): a pipeline built
        # in a process that has already run another one measured 15-30 % slower than the same pipeline alone
        import subprocess
        common = [sys.executable, os.path.abspath(__file__), '--no-extra-configs', '--no-cpu-baseline', '--no-tick-probe',
                  '--tts-mode', args.tts_mode, '--tts-lanes', str(args.tts_lanes), '--front-lanes', str(args.front_lanes),
                  '--tts-group', str(args.tts_group)]

        def child(argv):
            r = subprocess.run(common + argv, capture_output=True, text=True, timeout=1200)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not lines:
                return {'error': (r.stderr or r.stdout)[-400:]}
            return json.loads(lines[-1])
        for name, cfg, beam, vadm in runs:
            n2 = CONFIGS[cfg][0]
            d2 = child(['--config', cfg, '--steps', str(k2), '--warmup', '2', '--stt-beam', str(beam or args.stt_beam), '--vad-model', vadm])
            if 'error' in d2:
                extra[name] = d2
                continue
            extra[name] = {'workload': CONFIGS[cfg][2] % n2, 'value': d2['value'], 'steps': k2, 'ms_per_step': d2['ms_per_step'],
                           'tts_rows_per_batch': n2 * args.tts_group,
                           'tts_rows_per_decode_step': d2['config'].get('tts_rows_per_decode_step'),
                           'stt_decode': 'greedy' if (beam or args.stt_beam) == 1 else '%d beams' % (beam or args.stt_beam),
                           'vad_model': vadm}
        d5 = child(['--c5-only', '--steps', '5', '--stt-beam', str(args.stt_beam)])
        extra['C5_share'] = d5.get('C5_share', d5)
        dh = child(['--tts-harness-only'])
        extra['tts_harness'] = dh.get('tts_harness', dh)
        out['other_configs'] = extra
    elif rank == 0:
        pipe.close()
    if rank == 0:
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(CONFIGS[args.config][1], beams=args.stt_beam)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
