#!/usr/bin/env python3
"""bench.py -- real-time-factor x concurrent calls of the Infernos speech hot path on MI355X.

Workload (BASELINE.json configs[1]): 64 concurrent synthetic calls per GPU, each a 10 s
G.711 mu-law utterance in 20 ms / 160 B frames, carried through
    ingest (decode + 8k->16k + VAD windows) -> Whisper-tiny STT (log-mel, encoder, 32 greedy
    tokens) -> T2T stub -> SpeechT5 + HiFi-GAN + Amendment TTS (10 infer() calls = 5.12 s of
    speech at T_text = 64) -> 16k->8k -> mu-law encode,
bf16 models with seeded random weights (no checkpoints offline), synthetic audio.

One "step" = one such utterance cycle for every call of every rank.  value = call-seconds of
inbound audio fully processed per wall second = (N calls x 10 s) / step time: the aggregate
real-time factor (how many calls' worth of real time the node sustains).  Inputs are resident
in HBM when the timed region starts.  With N>1 GPUs calls are sharded (64 per GPU, weak
scaling); rank 0 scatters the frame matrix and gathers the encoded output over RCCL inside the
timed region.

Consecutive cycles are pipelined as a serving loop would: --front-lanes ingest+STT lanes run ahead of
--tts-lanes synthesis lanes, each of which synthesises the utterances of --tts-group consecutive cycles
as one batch (the path is bound by the dispatch rate of small dependent kernels, DESIGN.md 5, so fewer
and fatter launches and several independent launch chains in flight are what fill the GPU).  Every
cycle's whole work -- and the fill and drain of this pipeline -- is inside the timed K steps; outputs
are byte-identical to the sequential schedule (tests/test_pipeline_gpu.py).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--calls-per-gpu 64] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
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


def make_frames(ncalls, first_call, codec_encode):
    """[TICKS, ncalls, 160] u8 of the SURVEY.md 8(d) synthetic utterances (seed 1000+call)."""
    from infernos_amd.synth import synth_utterance
    x = np.stack([synth_utterance(1000 + first_call + i, UTT_SECONDS) for i in range(ncalls)])
    ulaw = codec_encode(x)                                     # [ncalls, 80000] u8
    return np.ascontiguousarray(ulaw.reshape(ncalls, TICKS, 160).transpose(1, 0, 2))


def cpu_baseline(ncalls=16, threads=32):
    """The oracle (CPU restatement, kind "port") timed on this host, bounded: one 10 s cycle for `ncalls`
    calls batched (the reference's own TTS cap is 8): ingest + STT + all 10 TTS infer() calls measured, nothing
    scaled (about 10 s of CPU work).  Threads capped (a 1-call fp32 graph on 128 threads
    is slower than on 32)."""
    from oracle import dsp as odsp, nn as onn
    from infernos_amd.synth import synth_utterance
    from infernos_amd.weights import synth_state_dict
    torch.manual_seed(0)
    nthreads = max(1, min(threads, os.cpu_count() or 1))
    prev = torch.get_num_threads()
    torch.set_num_threads(nthreads)
    try:
        x = np.stack([synth_utterance(1000 + i, UTT_SECONDS) for i in range(ncalls)])
        sd_w = synth_state_dict('whisper_tiny', 0)
        sd_t = synth_state_dict('speecht5_tts', 0, stop_bias=-20.0)
        sd_v, sd_a = synth_state_dict('hifigan', 0), synth_state_dict('amendment', 0)
        t0 = time.perf_counter()
        pcm = odsp.g711_decode(odsp.g711_encode(x))
        x16 = odsp.resample(pcm[:, 8000:72000], 8000, 16000)          # ~8 s of speech per call, as the VAD emits
        mel = torch.from_numpy(odsp.logmel(x16))
        with torch.no_grad():
            onn.whisper_greedy(sd_w, mel, torch.tensor([[50258, 50259, 50359, 50363]] * ncalls), 32, 6)
        t_stt = time.perf_counter() - t0
        g = torch.Generator().manual_seed(2000)
        ids = torch.randint(4, 80, (ncalls, 64), generator=g)
        spk = torch.randn(ncalls, 512, generator=g)
        t1 = time.perf_counter()
        with torch.no_grad():
            st = onn.TTSState(sd_t, ids, torch.ones_like(ids).int(), spk)
            t_enc = time.perf_counter() - t1
            masks = (torch.rand(16, 2, 256, generator=g) < 0.5).to(torch.uint8)
            t2 = time.perf_counter()
            nmeas = 10
            for _ in range(nmeas):
                a = onn.tts_infer(sd_t, sd_v, sd_a, st, masks)
                odsp.g711_encode(odsp.resample(a.numpy(), 16000, 8000))
            t_inf = (time.perf_counter() - t2) / nmeas
    finally:
        torch.set_num_threads(prev)
    total = t_stt + t_enc + nmeas * t_inf
    return {'value': ncalls * UTT_SECONDS / total, 'unit': 'x real-time (call-seconds/s)', 'cores': nthreads, 'kind': 'port',
            'sample': '%d calls, one 10 s cycle on the fp32 oracle (oracle/): ingest + log-mel + Whisper-tiny 32 tokens '
                      '(%.2f s), SpeechT5 encoder (%.2f s), 10 TTS infer() calls + resample + mu-law encode (%.2f s each), all '
                      'measured; torch threads=%d of os.cpu_count()=%s' % (ncalls, t_stt, t_enc, t_inf, nthreads, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=24)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--calls-per-gpu', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--latency-ticks', type=int, default=200)
    ap.add_argument('--breakdown', action='store_true', help='print per-stage wall times to stderr')
    ap.add_argument('--tts-lanes', type=int, default=4, help='TTS engine instances whose utterance cycles may overlap')
    ap.add_argument('--front-lanes', type=int, default=2, help='ingest+STT lanes (cycles k, k+1 in flight together)')
    ap.add_argument('--tts-group', type=int, default=4, help='utterance cycles synthesised as one TTS batch')
    ap.add_argument('--no-tts-overlap', action='store_true', help='render on the lane stream instead of a second stream per lane')
    ap.add_argument('--no-pipeline', action='store_true', help='run the stages of consecutive cycles strictly one after another')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    # IFH_DRYRUN_ONE_GPU=1: every rank uses cuda:0 and the collectives run over gloo (host-staged) -- lets the
    # N>1 code path (sharding, collective order, stage threads) be exercised on a single-GPU box
    dry = os.environ.get('IFH_DRYRUN_ONE_GPU') == '1'
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        dist.barrier()
        # ingress (scatter, front-end thread) and egress (gather, main thread) get their own communicators
        g_in, g_out = dist.new_group(), dist.new_group()
    from infernos_amd import _lib
    from infernos_amd.pipeline import SpeechPipeline
    from infernos_amd.shard import gather_rows, scatter_frames, shard_bounds
    from infernos_amd.codecs import G711Codec

    n_local = args.calls_per_gpu
    n_total = n_local * world
    pipe = SpeechPipeline(n_local, dev, tts_lanes=args.tts_lanes, tts_overlap=not args.no_tts_overlap, tts_group=args.tts_group, front_lanes=args.front_lanes)
    codec = G711Codec().to(dev)

    def enc(x):
        return np.frombuffer(codec.encode(torch.from_numpy(x)), dtype=np.uint8).reshape(x.shape)
    # every rank builds its own rows for the N=1 path; with N>1 rank 0 holds all rows and scatters
    if world == 1:
        frames_all = torch.from_numpy(make_frames(n_local, 0, enc)).to(dev)
    else:
        frames_all = torch.from_numpy(make_frames(n_total, 0, enc)).to(dev) if rank == 0 else None

    def reset_state():
        pipe.reset_calls()

    def frames_for(k):
        # with N>1 the ingress rank scatters this cycle's frame block over RCCL (inside the timed region)
        return scatter_frames(frames_all, n_total, TICKS, dev, group=g_in) if world > 1 else frames_all

    def run(nsteps):
        egress = (lambda r: gather_rows(r['ulaw'], n_total, group=g_out)) if world > 1 else None
        return pipe.run_steps(frames_for, nsteps, pipelined=not args.no_pipeline, on_cycle=egress)

    # priming (untimed, not part of the W warm-up steps): two sequential cycles load every kernel and
    # capture the hipGraphs of the decode loops, so the timed steps replay them
    pipe.run_steps(frames_for, 2, pipelined=False)
    pipe.prime(frames_for(0))
    if args.warmup:
        res = run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device='cpu' if dry else dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_total * UTT_SECONDS / (dt / args.steps)

    if args.breakdown and rank == 0 and getattr(pipe, 'stage_wall', None):
        sw = pipe.stage_wall
        print('pipelined job wall ms: front %.1f (n=%d)  tts %.1f (n=%d)' % (
            1e3 * sum(sw['front']) / max(1, len(sw['front'])), len(sw['front']),
            1e3 * sum(sw['tts']) / max(1, len(sw['tts'])), len(sw['tts'])), file=sys.stderr)
    if args.breakdown and rank == 0 and world == 1:
        reset_state()
        fr = frames_for(0)
        torch.cuda.synchronize(); a = time.perf_counter()
        ch = pipe.ingest(fr); torch.cuda.synchronize(); b = time.perf_counter()
        pipe.stt(ch); torch.cuda.synchronize(); c = time.perf_counter()
        pipe.synthesize(); torch.cuda.synchronize(); d = time.perf_counter()
        print('breakdown ms: ingest %.1f stt %.1f tts %.1f' % ((b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3), file=sys.stderr)
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
        nchunks = 4 * n_local * max(1, args.tts_group)        # the vocoder launch group of the timed region: 4 chunks x (calls x grouped cycles)
        voc_in = torch.randn(nchunks, 12, 80, device=dev).to(torch.bfloat16)
        t_voc = ev_time(lambda: pipe.tts.vocoder(voc_in), n=5)
        ach_tf = nchunks * VOCODER_GFLOP_PER_CHUNK / t_voc / 1e3
        x16 = torch.randn(n_local, 480000, device=dev) * 0.1
        lens = torch.full((n_local,), 480000, dtype=torch.int32, device=dev)
        mel_out = torch.empty(n_local, 80, 3000, device=dev)
        t_mel = ev_time(lambda: pipe.logmel(x16, lens=lens, out=mel_out), n=5)
        ach_gbs = n_local * LOGMEL_BYTES_PER_WINDOW / t_mel / 1e9
        # ---- per-tick latency: host frame matrix in, ingest + VAD + encode of one outgoing frame, host bytes out
        reset_state()
        host_frames = frames_all[:, :n_local].cpu().pin_memory() if world > 1 else frames_all.cpu().pin_memory()
        dfr = torch.empty((n_local, 160), dtype=torch.uint8, device=dev)
        outpcm = torch.zeros((n_local, 160), dtype=torch.float32, device=dev)
        outenc = torch.empty((n_local, 160), dtype=torch.uint8, device=dev)
        host_out = torch.empty((n_local, 160), dtype=torch.uint8).pin_memory()
        lat = []
        nb = 0
        L = _lib.lib()
        for t in range(min(args.latency_ticks, TICKS)):
            torch.cuda.synchronize()
            a = time.perf_counter()
            dfr.copy_(host_frames[t], non_blocking=True)
            pipe.calls.tick(dfr, pipe.slots, pipe.pcm8k, pipe.pcm16k)
            nb += 160
            if nb >= 768:
                nb -= 768
                pipe.vad.step(pipe.calls.win)
            L.ifh_g711_encode_f32_u8(_lib.ptr(outpcm), _lib.ptr(outenc), outpcm.numel(), _lib.stream_ptr(dev))
            host_out.copy_(outenc, non_blocking=True)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - a) * 1e3)
        lat = np.array(lat)
        traffic = None
        pmc = os.path.join(ROOT, 'profiles', 'r01_vocoder_pmc.json')
        if os.path.exists(pmc) and n_local == 64:       # PMC counters need their own rocprofv3 run: measured offline
            pj = json.load(open(pmc))
            traffic = pj['hbm_bytes_per_pass'] * nchunks / pj.get('chunks_per_pass', 256)
        traffic_lm = None
        pmc_lm = os.path.join(ROOT, 'profiles', 'r01_logmel_pmc.json')
        if os.path.exists(pmc_lm):
            pj = json.load(open(pmc_lm))
            traffic_lm = pj['hbm_bytes_per_pass'] * n_local / pj.get('chunks_per_pass', 64)
        out = {
            'metric': 'real-time-factor x concurrent calls (STT+TTS on 20 ms G.711 frames)',
            'value': round(value, 2), 'unit': 'x real-time (call-seconds/s)', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'C2: %d concurrent synthetic 10 s G.711 calls per GPU through ingest+VAD -> Whisper-tiny '
                                   'STT (32 tokens) -> T2T stub -> SpeechT5+HiFi-GAN TTS (10 infer calls, T_text 64) -> '
                                   'mu-law' % n_local, 'calls_per_gpu': n_local, 'calls_total': n_total,
                       'utterance_seconds': UTT_SECONDS, 'weights': 'seeded random (HF shapes)',
                       'parallelism': 'calls sharded %d/GPU, models replicated; RCCL scatter/gather of frames/output' % n_local,
                       'stage_pipelining': not args.no_pipeline, 'front_lanes': args.front_lanes, 'tts_lanes': args.tts_lanes,
                       'tts_group_cycles': args.tts_group},
            'p50_tick_latency_ms': round(float(np.percentile(lat, 50)), 4),
            'p99_tick_latency_ms': round(float(np.percentile(lat, 99)), 4),
            'stt_audio_seconds_per_call': round(float(res['stt_seconds'].mean()), 3),
            'tts_samples_per_call': int(res['tts_samples'].float().mean()),
            'roofline': {'kernel': 'HiFi-GAN vocoder pass = k_resblock_pair<*> + k_igemm<*> (%d chunks x 12 frames per launch group)' % nchunks,
                         'bound': 'mfma', 'achieved': round(ach_tf, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': round(ach_tf / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                         'traffic_note': 'HBM bytes per pass from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, profiles/r01_vocoder_pmc.json (scaled by chunks if the pass sizes differ)',
                         'seconds_per_vocoder_pass': t_voc},
            'roofline_logmel': {'kernel': 'k_logmel_fft+k_logmel_finish (%d x 30 s windows)' % n_local, 'bound': 'hbm',
                                'achieved': round(ach_gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                'frac': round(ach_gbs / PEAK_HBM_GBS, 4), 'traffic': traffic_lm,
                                'traffic_note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE(x2)/WRITE_SIZE, profiles/r01_logmel_pmc.json',
                                'seconds': t_mel},
        }
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
