"""Batched per-GPU serving loop for N concurrent calls -- the MI355X counterpart of the
reference's per-call Python threads (RTP/InfernRTPIngest.py:135-150 -> SileroVADWorker ->
STTSession/InfernSTTWorker -> app (T2T stub) -> TTSSession/InfernTTSWorker -> G711 encode,
SURVEY.md section 3).  One instance owns one GPU and a fixed set of calls (sticky sharding,
SURVEY.md 8e); every stage processes all of its calls in one batch of HIP kernels.

Stages of one utterance cycle (`step`):
  ingest : T ticks of [N,160] mu-law frames -> ifh_ingest_tick; each completed 768-sample
           window -> VAD probability (pluggable; stand-in) -> ifh_vad_step -> VadAudioChunks
  stt    : per call merge chunks (VadAudioChunk.append) -> 8k->16k resample -> log-mel ->
           Whisper encoder -> 5-beam search (the reference default engine's decode) or greedy (fixed token budget)
  t2t    : identity stub on token ids (BASELINE config 3 calls it "T2T stub")
  tts    : SpeechT5 encoder once, then n_infer x HelloSippyRTPipe.infer (16 decoder steps,
           postnet, HiFi-GAN, amendment, 16k->8k), trimmed per the reference's dispatch offsets,
           mu-law encoded -> [N, bytes] for the RTP side.
"""
import ctypes
import os
import threading
from typing import List

import torch

from . import _lib, ops
from .audio import VadAudioChunk, get_resampler
from .codecs import G711Codec
from .frontend import CallTable
from .vad import ABUF_CAP, EMIT_CAP, WINDOW


LOG_CAP = 4096


class BatchedVAD:
    """SileroVADWorker's device tables for a fixed set of N calls, stepped for all calls at once."""

    def __init__(self, ncalls, device, model=None, input_sr=8000, threshold=0.5):
        self.device = dev = _lib.require_device(device)
        self.n, self.input_sr, self.threshold = ncalls, input_sr, threshold
        self.model = model
        self.slot = torch.arange(ncalls, dtype=torch.int32, device=dev)
        self.st = torch.zeros((ncalls, 4), dtype=torch.int64, device=dev)
        self.st[:, 3] = -1
        self.blen = torch.zeros(ncalls, dtype=torch.int32, device=dev)
        self.abuf = torch.zeros((ncalls, ABUF_CAP), dtype=torch.float32, device=dev)
        self.emit = torch.empty((ncalls, EMIT_CAP), dtype=torch.float32, device=dev)
        self.ev = torch.empty((ncalls, 8), dtype=torch.int64, device=dev)
        self.prob = torch.empty(ncalls, dtype=torch.float32, device=dev)
        # recurrent state of a stateful probability model (SileroVADUtils.py:11,99,131): every step covers all N
        # calls in slot order, so the [2,N,64] tables are handed to the model as they are
        self.mh = torch.zeros((2, ncalls, 64), dtype=torch.float32, device=dev)
        self.mc = torch.zeros((2, ncalls, 64), dtype=torch.float32, device=dev)
        self.arena = None                      # ifh_ingest_block: emitted chunk audio of one block, device f32
        self.log = torch.zeros((LOG_CAP, 4), dtype=torch.int64)

    def step(self, win: torch.Tensor) -> List[VadAudioChunk]:
        """win f32 [N,768] (CallTable.win).  Returns [(call, VadAudioChunk)] emitted by this window."""
        dev, L = self.device, _lib.lib()
        if self.model is None:
            _lib.check(L.ifh_vad_energy_prob(_lib.ptr(win), _lib.ptr(self.slot), self.n, _lib.ptr(self.prob),
                                             _lib.stream_ptr(dev)), 'ifh_vad_energy_prob')
            prob = self.prob
        else:
            from .vad import _inject_state, _stateful
            if _stateful(self.model):
                _inject_state(self.model, self.mh, self.mc, self.input_sr, self.n)
            prob = self.model(win, self.input_sr).to(dev, torch.float32).contiguous()
            if _stateful(self.model):
                self.mh, self.mc = self.model._c._h, self.model._c._c
        _lib.check(L.ifh_vad_step(_lib.ptr(win), _lib.ptr(prob), _lib.ptr(self.slot), self.n, self.input_sr,
                                  float(self.threshold), _lib.ptr(self.st), _lib.ptr(self.blen), _lib.ptr(self.abuf),
                                  _lib.ptr(self.ev), _lib.ptr(self.emit), _lib.stream_ptr(dev)), 'ifh_vad_step')
        ev = self.ev.cpu()                       # one host sync per window batch, like the reference's .tolist()
        if bool(ev[:, 6].any()):
            raise AssertionError('VAD buffer invariant violated (SileroVAD.py:89/95-98)')
        out = []
        for i in torch.nonzero(ev[:, 3]).flatten().tolist():
            out.append((i, VadAudioChunk(self.emit[i, :int(ev[i, 5])].clone(), self.input_sr, int(ev[i, 4]))))
        return out


# ifh_set_cu_budget is process-wide: the pipelines that reserved CUs, in construction order.  The budget in force is the newest
# live pipeline's; closing one brings back the one before it (or "all CUs" when none is left).
_budget_lock = threading.Lock()
_budget_holders = []


def _budget_push(owner, ncus):
    with _budget_lock:
        _budget_holders.append((id(owner), int(ncus)))
        _lib.check(_lib.lib().ifh_set_cu_budget(int(ncus)), 'ifh_set_cu_budget')


def _budget_pop(owner):
    with _budget_lock:
        for i, (oid, _) in enumerate(_budget_holders):
            if oid == id(owner):
                del _budget_holders[i]
                break
        _lib.check(_lib.lib().ifh_set_cu_budget(_budget_holders[-1][1] if _budget_holders else 0), 'ifh_set_cu_budget')


class SpeechPipeline:
    def __init__(self, ncalls: int, device=None, whisper_family='whisper_tiny', seed=0, n_text=64, n_infer=10,
                 n_new_tokens=32, tts_output_sr=8000, weights=None, tts_lanes=2, tts_overlap=True, tts_group=1,
                 front_lanes=1, stt_beam=1, tts_mode='lanes', cu_reserve=None, vad_model=None):
        from .engines.whisper import Whisper
        from .features import WhisperLogMel
        from .tts import HelloSippyRTPipe
        from .weights import synth_state_dict
        self.device = dev = _lib.require_device(device)
        self.n, self.n_text, self.n_infer, self.n_new = ncalls, n_text, n_infer, n_new_tokens
        # The persistent vocoder kernels (one workgroup per CU holding all of its LDS for 0.4-1.4 ms: chain.hip, level.hip) leave
        # `cu_reserve` CUs alone: the decode chains (a TTS step, a Whisper token) are latency-bound sequences of small launches that
        # otherwise wait for a CU with free LDS until a vocoder workgroup ends.  The renderer only has to keep up with the decoder
        # (2.7 ms of vocoder per 12 ms of decoder steps), so it does not need the whole chip: C3 +3-4 %, TTS stage alone -8 %
        # (IFH_CU_RESERVE; 0 = off).  Process-wide (ifh_set_cu_budget).
        if cu_reserve is None:
            cu_reserve = int(os.environ.get('IFH_CU_RESERVE', '96'))
        self.cu_reserve = max(0, cu_reserve)
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        self._budget_set = False
        if int(os.environ.get('IFH_BIG_CUS', '0')) <= 0:          # (that switch masks the throughput streams and sets the budget itself)
            if self.cu_reserve:                                   # process-wide: close() gives the CUs back
                _budget_push(self, max(32, ncu - self.cu_reserve))
                self._budget_set = True
        self.tts_overlap = tts_overlap      # render of chunk c on a second stream while chunk c+1 decodes (within a lane)
        self.tts_group = max(1, tts_group)  # utterance cycles synthesised as one TTS batch (rows = group * ncalls)
        self.block_ingest = True           # False: one ifh_ingest_tick per tick (what the block form is tested against)
        # The probability model of the VAD step (Core/VAD/SileroVAD.py:78-80 runs its network on every 768-sample window of every call):
        # None / 'energy' = the stateless energy rule (ifh_vad_energy_prob); 'recurrent' = the conv + 2 x LSTM(64) network of
        # csrc/vadnet.hip with the distilled weights and the per-call [2,N,64] x 2 state, driven window by window from
        # ifh_ingest_block_net; or a factory `f(device) -> model(x, sr)` (any other model takes the per-tick path).
        self.vad_model = vad_model
        assert self.vad_model in (None, 'energy', 'recurrent') or callable(self.vad_model)
        self.stt_beam = int(stt_beam)      # 1: greedy (the reference's torch engine); 5: its default engine's beam search
        self.stt_dec_prio = False
        with torch.cuda.device(dev):
            self.calls = CallTable(ncalls, dev)
            self.vad = BatchedVAD(ncalls, dev, model=self._new_vad_model())
            self.codec = G711Codec().to(dev)
            self.up = get_resampler(8000, 16000, str(dev))
            w = weights or {}
            self.whisper = Whisper(w.get(whisper_family) or synth_state_dict(whisper_family, seed), dev)
            self.logmel = WhisperLogMel(self.whisper.n_mel, dev)
            tw = {k: (w.get(k) or synth_state_dict(k, seed, **({'stop_bias': -20.0} if k == 'speecht5_tts' else {})))
                  for k in ('speecht5_tts', 'hifigan', 'amendment')}
            # TTS lanes: independent engine instances (own batch states, frame buffers, hipGraphs and streams)
            # so that the synthesis of consecutive utterance cycles can be in flight together (run_steps)
            self.tts = HelloSippyRTPipe(dev, weights=tw, processor=_NoProcessor(), speaker_embeddings=[],
                                        output_sr=tts_output_sr)
            self.tts_lanes = [self.tts] + [self.tts.clone_for_lane() for _ in range(max(1, tts_lanes) - 1)]
        # tts_mode 'continuous': ONE ragged decode batch over every utterance batch in flight (tts.ContinuousTTS) instead
        # of one engine clone + launch chain per lane; tts_lanes then bounds the utterance batches in flight
        assert tts_mode in ('lanes', 'continuous')
        self.tts_mode = tts_mode
        self.ctts = None
        if tts_mode == 'continuous':
            from .tts import ContinuousTTS
            bucket = -(-ncalls * self.tts_group // 16) * 16
            # (+1 batch of head-room: a finished batch frees its rows one engine call after its last audio was queued)
            # ne ragged batches, each with its own state, graphs, streams and thread; utterance cycle c joins engine c % ne.  Rows
            # are independent, so the audio does not depend on the engine (one engine: two measured 4 % slower, profiles/NOTES.md)
            ne = 1
            per_engine = -(-(max(1, tts_lanes) + 1) // ne) + (1 if ne > 1 else 0)
            self.ctts_all = [ContinuousTTS(self.tts if e == 0 else self.tts.clone_for_lane(),
                                           max_rows=min(1024, bucket * per_engine), max_text=n_text, row_bucket=bucket).start()
                             for e in range(ne)]
            self.ctts = self.ctts_all[0]
            import itertools
            self._tts_rr = itertools.count()
            self.tts_lanes = [self.tts] * max(1, tts_lanes)
        self.slots = torch.arange(ncalls, dtype=torch.int32, device=dev)
        self.prompt = torch.tensor([[50258, 50259, 50359, 50363]] * ncalls, dtype=torch.int32)
        g = torch.Generator().manual_seed(2000 + seed)
        self.speakers = torch.randn(ncalls, 512, generator=g)
        self.text_ids = torch.randint(4, 80, (ncalls, n_text), generator=g, dtype=torch.int32)
        self._lane_streams = [torch.cuda.Stream(device=dev) for _ in self.tts_lanes]
        # (render streams of the lane schedule only: every stream created takes a slot in the round-robin over the hardware queues)
        self._side_streams = [torch.cuda.Stream(device=dev) for _ in self.tts_lanes] if tts_mode == 'lanes' else None
        self._side_stream = self._side_streams[0] if self._side_streams else None
        self.pcm8k = torch.empty((ncalls, 160), dtype=torch.float32, device=dev)
        self.pcm16k = torch.empty((ncalls, 320), dtype=torch.float32, device=dev)
        # front-end lanes: lane 0 is this object's own call table / VAD / Whisper buffers; further lanes are private
        # copies over the same weights, so that ingest+STT of consecutive cycles can overlap too (run_steps)
        self.front_lanes = [_FrontLane(self, first=True)] + [_FrontLane(self) for _ in range(max(1, front_lanes) - 1)]
        # In continuous mode the lanes' submit work (SpeechT5 text encoder, cross K|V: ~1 ms per batch) shares ONE stream.
        if tts_mode == 'continuous':
            self._lane_streams = [self._lane_streams[0]] * len(self._lane_streams)
        # The process's streams are dealt over four hardware queues when they are first used; first use from several threads made
        # the dealing -- and with it the cycle time, 109 or 118 ms -- a matter of thread timing.  Every stream is used once here, in
        # a fixed order (F = front lanes, M / S = TTS decode / render, L = submit), from this thread.
        order = 'S,F,L,M'
        named = {'F': [fl.stream for fl in self.front_lanes], 'L': list(dict.fromkeys(self._lane_streams)),
                 'M': [e.main for e in getattr(self, 'ctts_all', [])], 'S': [e.side for e in getattr(self, 'ctts_all', [])]}
        scratch = torch.zeros(64, device=dev)
        for key in [k.strip() for k in order.split(',') if k.strip()]:
            for st_ in named.get(key, []):
                with torch.cuda.stream(st_):
                    scratch.add_(1.0)
                st_.synchronize()

    def _new_vad_model(self):
        if self.vad_model in (None, 'energy'):
            return None
        if self.vad_model == 'recurrent':
            from .vad import RecurrentVADModel
            return RecurrentVADModel(self.device, weights='distilled')
        return self.vad_model(self.device)

    # ---- stage 1 -----------------------------------------------------------------------------
    def ingest(self, frames: torch.Tensor, fl=None, block=None):
        """frames u8 [T,N,160] on the device -> per-call list of VadAudioChunk.  With the built-in probability model the
        T ticks are driven by one ifh_ingest_block call (same launches, no interpreter in between); block=False or a
        pluggable VAD model takes the per-tick path."""
        fl = self.front_lanes[0] if fl is None else fl
        T = frames.size(0)
        chunks = [[] for _ in range(self.n)]
        from .vad import RecurrentVADModel
        if (self.block_ingest if block is None else block) and T and (fl.vad.model is None or isinstance(fl.vad.model, RecurrentVADModel)):
            return self._ingest_block(frames.contiguous(), fl, chunks)
        nbytes = int(fl.calls.fifo_len[0]) if T else 0
        for t in range(T):
            fl.calls.tick(frames[t], self.slots, fl.pcm8k, fl.pcm16k, want_ready=False)
            nbytes += 160
            if nbytes >= WINDOW:                 # every stream completes its window on the same tick
                nbytes -= WINDOW
                for i, ch in fl.vad.step(fl.calls.win):
                    chunks[i].append(ch)
        return chunks

    def _ingest_block(self, frames, fl, chunks):
        dev, v, c = self.device, fl.vad, fl.calls
        if v.arena is None:
            v.arena = torch.empty(self.n * EMIT_CAP, dtype=torch.float32, device=dev)
        nlog, used = ctypes.c_int(0), ctypes.c_int64(0)
        args = [_lib.ptr(frames), frames.size(0), _lib.ptr(self.slots), self.n, _lib.ptr(c.fifo), _lib.ptr(c.fifo_len),
                _lib.ptr(c.win), _lib.ptr(c.win_ready), _lib.ptr(c.hist), _lib.ptr(fl.pcm8k), _lib.ptr(fl.pcm16k),
                c._rs.handle, _lib.ptr(v.prob), v.input_sr, float(v.threshold), _lib.ptr(v.st), _lib.ptr(v.blen),
                _lib.ptr(v.abuf), _lib.ptr(v.ev), _lib.ptr(v.emit), _lib.ptr(v.arena), v.arena.numel(),
                ctypes.c_void_p(v.log.data_ptr()), LOG_CAP, ctypes.byref(nlog), ctypes.byref(used)]
        with torch.cuda.device(dev):
            if v.model is None:
                _lib.check(_lib.lib().ifh_ingest_block(*args, _lib.stream_ptr(dev)), 'ifh_ingest_block')
            else:       # the recurrent network on every window, its per-call state (v.mh, v.mc) updated in place
                assert v.mh.is_contiguous() and v.mc.is_contiguous()
                _lib.check(_lib.lib().ifh_ingest_block_net(*args, _lib.ptr(v.model.blob), _lib.ptr(v.mh), _lib.ptr(v.mc),
                                                           _lib.stream_ptr(dev)), 'ifh_ingest_block_net')
        for i, ipos, ln, off in v.log[:nlog.value].tolist():
            chunks[i].append(VadAudioChunk(v.arena[off:off + ln].clone(), v.input_sr, ipos))
        return chunks

    # ---- stage 2 -----------------------------------------------------------------------------
    def stt(self, chunks, fl=None):
        """-> (tokens int32 [N, n_new] device, no_speech_prob f32 [N] device, audio seconds per call)"""
        fl = self.front_lanes[0] if fl is None else fl
        dev = self.device
        merged = []
        for lst in chunks:
            if not lst:
                merged.append(torch.zeros(0, device=dev))
                continue
            head = lst[0]
            for nxt in lst[1:]:
                if nxt.tpos() + nxt.duration() - head.tpos() < 32.0:
                    head.append(nxt)
            merged.append(head.audio)
        lens8 = torch.tensor([m.numel() for m in merged], dtype=torch.int32)
        L8 = max(int(lens8.max()), 1)
        nrow = len(chunks)                                   # N, or G*N when the STT of G cycles runs as one batch
        x8 = torch.zeros((nrow, L8), dtype=torch.float32, device=dev)
        for i, m in enumerate(merged):
            x8[i, :m.numel()] = m
        x16 = self.up(x8, lens=lens8)
        lens16 = (lens8 * 2).to(dev)
        raw, wmax = fl.logmel.raw(x16, lens=lens16)        # normalisation fused into conv1's layout change
        enc = fl.whisper.encode(raw=(fl.logmel, raw, wmax))
        # bounded queue depth per lane: the encoder (~60 long kernels) and every 8 decode steps are waited for before more is
        # queued -- the process's streams share four hardware queues, and whatever a lane has queued stands in front of a
        # real-time tick whose launches land on the same queue (p99 tick latency 76-136 ms -> 7-36 ms; throughput +3 %)
        cur = torch.cuda.current_stream(dev)
        cur.synchronize()
        prompt = self.prompt if nrow == self.n else self.prompt.repeat(nrow // self.n, 1)
        # the token loop is a latency-bound chain of small launches like the TTS decode: on a high-priority queue its kernels do
        # not wait for CU slots behind the other lanes' encoder GEMMs (tuning switch IFH_STT_DEC_PRIO; the encoder stays where it is)
        dec = getattr(fl, 'dec_stream', None) if self.stt_dec_prio else None
        with torch.cuda.stream(dec if dec is not None else cur):
            if dec is not None:
                enc.record_stream(dec)
            if self.stt_beam > 1:
                toks, _, _, nsp = fl.whisper.generate_beam(enc, prompt, self.n_new, beams=self.stt_beam, eos_id=50257,
                                                           no_speech_id=50362, check_every=8)
            else:
                toks, nsp, _ = fl.whisper.generate(enc, prompt, self.n_new, no_speech_id=50362)
            if dec is not None:
                ev = torch.cuda.Event()
                ev.record(dec)
                cur.wait_event(ev)
                for t in (toks, nsp):
                    t.record_stream(cur)
        return toks, nsp, (lens8.float() / 8000.0)

    # ---- stage 3 -----------------------------------------------------------------------------
    def synthesize(self, text_ids=None, overlap=None, lane=0, group=1):
        """-> (ulaw u8 [N, n_infer*A] device, valid sample count per call, spans) with A = 8192/(16000/output_sr).
        Two-stream schedule: the decoder steps of call c+1 (launch/latency-bound, few CUs busy) run on the
        main stream while postnet + HiFi-GAN + amendment + resample + mu-law of call c run on a second
        stream (the frame buffers are double-buffered by call parity).  This is the 3-stage pipeline of the
        reference's own harness (HelloSippyRTPipeTest.py:126-161) expressed with HIP streams/events."""
        dev, pp = self.device, self.tts_lanes[lane]
        overlap = self.tts_overlap if overlap is None else overlap
        ids = self.text_ids if text_ids is None else text_ids
        spk = self.speakers
        if group > 1:                                  # the utterances of `group` consecutive cycles as ONE batch (rows g*n + call)
            ids, spk = ids.repeat(group, 1), spk.repeat(group, 1)
        nb = ids.size(0)
        if self.ctts is not None:
            # continuous batching: the batch joins the engine's running decode batch at its next infer() boundary
            eng = self.ctts_all[next(self._tts_rr) % len(self.ctts_all)]       # round-robin over the engines (one: always it)
            grp = eng.submit(ids, torch.full((nb,), ids.size(1), dtype=torch.int32), spk, max_calls=self.n_infer,
                             want_ulaw=True).result()
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(grp.done_event)
            grp.ulaw.record_stream(cur)
            return grp.ulaw, grp.valid, grp.spans
        state = _make_state(pp, ids, spk)
        st = state.dev
        rr = pp.model_sr // pp.output_sr
        A = 8192 // rr
        stepsize = 512 // rr
        out = torch.empty((nb, self.n_infer * A), dtype=torch.uint8, device=dev)
        valid = torch.zeros(nb, dtype=torch.int64)
        spans = []
        main = torch.cuda.current_stream(dev)
        side = self._side_streams[lane] if (overlap and self._side_streams) else main
        out.record_stream(side)                 # written on the render stream: the allocator must not recycle it early
        ren_done = [None, None]
        for c in range(self.n_infer):
            par = st.ncalls & 1
            if ren_done[par] is not None:
                main.wait_event(ren_done[par])          # the renderer of call c-2 has released this parity's buffers
            pp.decode_chunk(state)
            dec_done = torch.cuda.Event()
            dec_done.record(main)
            with torch.cuda.stream(side):
                side.wait_event(dec_done)
                bf = pp.render(st, par, use_graphs=pp.model.use_graphs)
                pcm = bf.float()
                if pp.resampler is not None:
                    pcm = pp.resampler(pcm)
                tmp = torch.empty((nb, A), dtype=torch.uint8, device=dev)
                _lib.check(_lib.lib().ifh_g711_encode_f32_u8(_lib.ptr(pcm), _lib.ptr(tmp), pcm.numel(), _lib.stream_ptr(dev)),
                           'ifh_g711_encode_f32_u8')
                out[:, c * A:(c + 1) * A].copy_(tmp)
                ev = torch.cuda.Event()
                ev.record(side)
                ren_done[par] = ev
            idx = st.idx
            ends = st.ends_at.cpu().tolist()                     # the per-call sync the reference also has (.item())
            row = []
            for i in range(nb):
                s_ = max(0, A - (idx - 1) * stepsize)
                e_ = min(A, A - ((idx - ends[i]) * stepsize if ends[i] >= 0 else 0))
                row.append((s_, max(s_, e_)))
                valid[i] += max(0, e_ - s_)
            spans.append(row)
            if all(e >= 0 and e <= idx - 1 for e in ends):
                break
        for ev in ren_done:
            if ev is not None:
                main.wait_event(ev)
        return out, valid, spans

    def reset_calls(self, fl=None):
        """New utterance on every call slot: clear the per-call front-end state."""
        fl = self.front_lanes[0] if fl is None else fl
        fl.calls.fifo_len.zero_()
        fl.calls.hist.zero_()
        fl.vad.st.zero_()
        fl.vad.st[:, 3] = -1
        fl.vad.blen.zero_()
        fl.vad.mh = torch.zeros_like(fl.vad.mh)      # (new tensors: the per-tick path hands the old ones to the model object)
        fl.vad.mc = torch.zeros_like(fl.vad.mc)

    def front_group(self, frames_list, fl=None):
        """ingest of each cycle in turn, then ONE STT batch over all of them (rows g*N + call): the Whisper token loop is
        a chain of small launches whose cost hardly depends on the row count.  -> list of per-cycle result dicts"""
        fl = self.front_lanes[0] if fl is None else fl
        per, allc = [], []
        for fr in frames_list:
            self.reset_calls(fl)
            ch = self.ingest(fr, fl)
            per.append(ch)
            allc.extend(ch)
        toks, nsp, secs = self.stt(allc, fl)
        n = self.n
        return [dict(tokens=toks[j * n:(j + 1) * n], no_speech_prob=nsp[j * n:(j + 1) * n], stt_seconds=secs[j * n:(j + 1) * n],
                     chunks=[[(c.ipos, c.audio.numel()) for c in lst] for lst in ch]) for j, ch in enumerate(per)]

    def front(self, frames: torch.Tensor, fl=None):
        """ingest + STT of one utterance cycle (stages 1-2) on the current stream"""
        fl = self.front_lanes[0] if fl is None else fl
        self.reset_calls(fl)
        chunks = self.ingest(frames, fl)
        toks, nsp, secs = self.stt(chunks, fl)
        return dict(tokens=toks, no_speech_prob=nsp, stt_seconds=secs,
                    chunks=[[(c.ipos, c.audio.numel()) for c in lst] for lst in chunks])

    def prime(self, frames=None):
        """Untimed preparation of every lane: two eager passes (load the kernels, size the buffers) and one that
        captures the hipGraphs (TTS decode steps and renderer per lane and group size; with `frames`, the Whisper
        token loop of every front-end lane)."""
        if frames is not None:
            for fl in self.front_lanes:
                with torch.cuda.stream(fl.stream):
                    for g in sorted({self.tts_group, 1} | set(range(1, self.tts_group))):
                        for _ in range(3):
                            self.front_group([frames] * g, fl)
                fl.stream.synchronize()
        if self.ctts is not None:
            for e, eng in enumerate(self.ctts_all):
                b = eng.row_bucket
                eng.warm([b * k for k in range(1, eng.st.R // b + 1)])
                self.synthesize(lane=e)
            torch.cuda.synchronize(self.device)
            return
        for lane in range(len(self.tts_lanes)):
            for g in sorted({self.tts_group, 1} | set(range(1, self.tts_group))):     # a trailing group may be smaller
                for _ in range(3):
                    self.synthesize(lane=lane, group=g)
        torch.cuda.synchronize(self.device)

    def run_steps(self, frames_fn, nsteps: int, pipelined: bool = True, on_cycle=None):
        """nsteps utterance cycles.  Pipelined: a front-end thread (own HIP stream) runs ingest+STT of cycle
        k+1 while TTS lane k % L (own thread and streams) synthesises cycle k -- and, the decode loop being a
        latency-bound chain of small launches that leaves most CUs idle, lane (k-1) % L may still be finishing
        cycle k-1.  In steady-state serving the stages always work on different utterances at once; every
        cycle still does all of its work inside the call.  frames_fn(k) -> u8 [T,N,160] device tensor for
        cycle k and on_cycle(result) both run on the calling thread, in cycle order (they may be collectives)."""
        if not pipelined or nsteps < 2:
            out = None
            for k in range(nsteps):
                out = self.front(frames_fn(k))
                out.update(zip(('ulaw', 'tts_samples', 'spans'), self.synthesize()))
                if on_cycle is not None:
                    on_cycle(out)
            return out
        from concurrent.futures import ThreadPoolExecutor
        dev = self.device
        L = len(self.tts_lanes)
        F = len(self.front_lanes)
        if not hasattr(self, '_pool'):                              # (a high-priority front stream made every stage slower)
            self._pool = ThreadPoolExecutor(max_workers=F)
            self._tts_pool = ThreadPoolExecutor(max_workers=L)

        main = torch.cuda.current_stream(dev)

        def fetch(k):
            # frames_fn may be a collective (ingress scatter): always issued from THIS thread, in cycle order,
            # so every rank enqueues its collectives in the same order; the front-end stream waits on the event
            fr = frames_fn(k)
            ev = torch.cuda.Event()
            ev.record(main)
            return fr, ev

        import time
        self.stage_wall = {'front': [], 'tts': []}                 # host wall seconds per job (bench --breakdown)

        def job(gi, frs):
            torch.cuda.set_device(dev)
            t0 = time.perf_counter()
            fl = self.front_lanes[gi % F]
            with fl.lock, torch.cuda.stream(fl.stream):
                for fr, fr_ready in frs:
                    fl.stream.wait_event(fr_ready)
                    fr.record_stream(fl.stream)
                rs = self.front_group([fr for fr, _ in frs], fl)
                ev = torch.cuda.Event()
                ev.record(fl.stream)
            self.stage_wall['front'].append(time.perf_counter() - t0)
            return rs, ev

        G = self.tts_group

        def tts_job(lane, front_fut):
            torch.cuda.set_device(dev)
            rs, stt_done = front_fut.result()                      # the G cycles of this group, in order
            t0 = time.perf_counter()
            stream = self._lane_streams[lane]
            with torch.cuda.stream(stream):
                stream.wait_event(stt_done)                        # T2T stub consumes the STT tokens
                ulaw, valid, spans = self.synthesize(lane=lane, group=len(rs))
                n = self.n
                for j, r in enumerate(rs):
                    r.update(ulaw=ulaw[j * n:(j + 1) * n], tts_samples=valid[j * n:(j + 1) * n],
                             spans=[row[j * n:(j + 1) * n] for row in spans])
                ev = torch.cuda.Event()
                ev.record(stream)
            self.stage_wall['tts'].append(time.perf_counter() - t0)
            return rs, ev

        import sys
        swi = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)        # several launch threads: hand the GIL over quickly
        ttss = {}

        def retire(gi):
            rs, ev = ttss.pop(gi).result()
            main.wait_event(ev)
            for r in rs:
                for v in r.values():                               # produced on other streams, consumed on this one
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(main)
                if on_cycle is not None:
                    on_cycle(r)                                    # e.g. egress gather of this cycle's output rows
            return rs[-1]

        def submit_tts(gi, front_fut):
            ttss[gi] = self._tts_pool.submit(tts_job, gi % L, front_fut)
        out = schedule_cycles(nsteps, G, L, fetch, lambda gi, frs: self._pool.submit(job, gi, frs), submit_tts, retire)
        sys.setswitchinterval(swi)
        return out

    def close(self):
        """stop the continuous TTS engine thread (its state holds the KV caches of every row slot) and give back the CUs the
        persistent kernels were kept off (the budget is process-wide: a pipeline that set it restores "all CUs")"""
        if self.ctts is not None:
            for eng in self.ctts_all:
                eng.stop()
            self.ctts, self.ctts_all = None, []
        if getattr(self, '_budget_set', False):        # behind the engines' last launches; another live pipeline's budget comes back
            _budget_pop(self)
            self._budget_set = False
        for name in ('_pool', '_tts_pool'):
            if hasattr(self, name):
                getattr(self, name).shutdown(wait=True)
                delattr(self, name)
        _lib.release_graphs()          # graphs of states already dropped: destroyed here, not wherever the collector finds them

    def step(self, frames: torch.Tensor):
        chunks = self.ingest(frames)
        toks, nsp, secs = self.stt(chunks)
        # T2T stub: identity on token ids; the TTS text is the fixed synthetic utterance
        ulaw, valid, spans = self.synthesize()
        return dict(tokens=toks, no_speech_prob=nsp, stt_seconds=secs, ulaw=ulaw, tts_samples=valid, spans=spans,
                    chunks=[[(c.ipos, c.audio.numel()) for c in lst] for lst in chunks])


def schedule_cycles(nsteps, group, lanes, fetch, submit_front, submit_tts, retire):
    """The order in which SpeechPipeline.run_steps hands work to its stage threads -- pure Python, no device calls, so that
    the multi-GPU collective order can be tested on CPU (tests/test_shard_cpu.py).  Cycles are grouped `group` at a time;
    fetch(k) (the ingress collective) and retire(g) (the egress collective of every cycle of group g) run on the CALLING
    thread, each in increasing order, so every rank issues the collectives of each communicator in the same order;
    the front-end jobs run up to `lanes` groups ahead of the synthesis jobs, which retire `lanes - 1` groups behind.
    submit_front(g, [fetch(k) ...]) -> future; submit_tts(g, front_future); retire(g) -> result of the last cycle."""
    ngroups = (nsteps + group - 1) // group
    fronts, out = {}, None
    nfront = 0                                                 # groups whose front-end job has been submitted
    for gi in range(ngroups):
        while nfront < min(ngroups, gi + 1 + lanes):           # the front end runs up to `lanes` groups ahead
            lo, hi = nfront * group, min(nsteps, (nfront + 1) * group)
            fronts[nfront] = submit_front(nfront, [fetch(k) for k in range(lo, hi)])
            nfront += 1
        submit_tts(gi, fronts.pop(gi))
        if gi >= lanes - 1:
            out = retire(gi - (lanes - 1))
    for gi in range(max(0, ngroups - (lanes - 1)), ngroups):
        out = retire(gi)
    return out


class _FrontLane:
    """Per-lane state of stages 1-2: call table, VAD tables, tick outputs, Whisper/log-mel working buffers, a stream."""

    def __init__(self, pipe: 'SpeechPipeline', first: bool = False):
        import copy
        import threading
        from .features import WhisperLogMel
        dev = pipe.device
        self.lock = threading.Lock()                       # one cycle at a time per lane
        with torch.cuda.device(dev):
            self.stream = _lib.throughput_stream(dev)        # log-mel, encoder, cross K|V, prompt prefill
            # (created only when used: every stream takes a slot in the round-robin over the four hardware queues)
            self.dec_stream = torch.cuda.Stream(device=dev, priority=-1) if pipe.stt_dec_prio else None
            if first:
                self.calls, self.vad, self.whisper, self.logmel = pipe.calls, pipe.vad, pipe.whisper, pipe.logmel
                self.pcm8k, self.pcm16k = pipe.pcm8k, pipe.pcm16k
            else:
                self.calls, self.vad = CallTable(pipe.n, dev), BatchedVAD(pipe.n, dev, model=pipe._new_vad_model())
                self.whisper = copy.copy(pipe.whisper)     # same weight tensors, private activation / KV buffers and graphs
                self.whisper._enc_bufs, self.whisper._dec_bufs = {}, {}
                self.logmel = WhisperLogMel(pipe.whisper.n_mel, dev)
                self.pcm8k = torch.empty_like(pipe.pcm8k)
                self.pcm16k = torch.empty_like(pipe.pcm16k)


class _NoProcessor:
    def __call__(self, text=None, return_tensors='pt'):
        raise RuntimeError('SpeechPipeline feeds token ids directly; no tokenizer is available offline')


def _make_state(pp, ids, speakers):
    """HelloSippyPipeStateBatched from token ids (bypasses the tokenizer)."""
    from .tts import HelloSippyPipeStateBatched

    class _S:
        pass
    states = []
    for i in range(ids.size(0)):
        s = _S()
        s.session, s.dispatch = None, None
        s.inputs = ids[i:i + 1].long()
        s.speaker_embeddings = speakers[i:i + 1]
        states.append(s)
    return HelloSippyPipeStateBatched(states, pp)
