"""Text-to-speech plugin surface: HelloSippyRTPipe (+ request/state types), InfernTTSWorker,
TTSRequest / TTSSndDispatch / TTSSession.

Interface of HelloSippyTTSRT/HelloSippyRTPipe.py:47-272, Cluster/InfernTTSWorker.py:56-105 and
Cluster/TTSSession.py:41-141.  `infer()` runs the whole 512 ms chunk (16 decoder steps,
postnet, carry+chunking, HiFi-GAN, AmendmentNetwork1, optional 16k->8k resample) as HIP
kernels; `unbatch_and_dispatch()` reproduces the reference's offset arithmetic and hands each
live session a 1-D CPU tensor, then None at the end of the utterance.
"""
import uuid
import weakref
from functools import partial
from time import monotonic
from typing import Dict, List, Optional, Tuple, Union
from uuid import UUID, uuid4

import numpy as np
import torch

from . import _lib, ops
from .audio import AudioChunk, get_resampler
from .engines.speecht5 import SpeechT5, TTSBatchState, TTSRaggedState, decoder_steps, postnet, ragged_decoder_steps
from .engines.vocoder import Amendment, HifiGan
from .muxer import ASMarkerGeneric, ASMarkerNewSent, ASMarkerSentDoneCB
from .torcher import InfernGlobals
from .workers import InfernBatchedWorker


class SessCmd:
    pass


class SessSyncCmd(SessCmd):
    def __init__(self, sessions):
        self.live = tuple(sorted(sessions.keys()))


class SessDispatchCmd(SessCmd):
    session: uuid.UUID

    def __init__(self, session_id: uuid.UUID):
        self.session = session_id


class HelloSippyPlayRequest(SessDispatchCmd):
    def __init__(self, session_id: uuid.UUID, text: str, speaker: torch.Tensor, dispatch: callable):
        self.text, self.speaker, self.dispatch = text, speaker, dispatch
        super().__init__(session_id)


class HelloSippyPipeState:
    """Per-utterance inputs (HelloSippyRTPipe.py:59-79): token ids, speaker x-vector."""

    def __init__(self, pp: 'HelloSippyRTPipe', req: HelloSippyPlayRequest):
        self.session, self.dispatch = req.session, req.dispatch
        text = req.text if pp.cleanup_text is None else pp.cleanup_text(req.text)
        self.inputs = pp.processor(text=text, return_tensors='pt')['input_ids']
        self.speaker_embeddings = req.speaker
        self.encoder_attention_mask = torch.ones_like(self.inputs, dtype=torch.int)


class HelloSippyPipeStateBatched:
    """Batch of utterances frozen at process_batch entry (HelloSippyRTPipe.py:81-121)."""

    def __init__(self, states: List[HelloSippyPipeState], pp: 'HelloSippyRTPipe'):
        self.dispatch = [s.dispatch for s in states]
        self.sessions = [s.session for s in states]
        T = max(s.inputs.size(1) for s in states)
        ids = torch.zeros((len(states), T), dtype=torch.int32)        # right-padded with 0 like the reference
        lens = torch.zeros(len(states), dtype=torch.int32)
        for i, s in enumerate(states):
            n = s.inputs.size(1)
            ids[i, :n] = s.inputs[0].to(torch.int32)
            lens[i] = n
        spk = torch.cat([s.speaker_embeddings.reshape(1, 512).float() for s in states])
        with torch.cuda.device(pp.device):
            self.dev = TTSBatchState.acquire(pp.model, ids, lens, spk)
        self.starts_at_host = [pp.post_nframes // 2] * len(states)
        self.audio = None

    # attribute views the reference exposes
    idx = property(lambda self: self.dev.idx)
    maxlen = property(lambda self: self.dev.maxlen)
    minlen = property(lambda self: self.dev.minlen)
    ends_at = property(lambda self: self.dev.ends_at)
    starts_at = property(lambda self: self.dev.starts_at)
    encoder_last_hidden_state = property(lambda self: self.dev.enc)


class DeviceMaskSource:
    """Bernoulli(0.5) keep-masks for the always-on prenet dropout, one [2][256] pair per decoder
    step, drawn on the device from a seeded generator."""

    def __init__(self, device, seed=0):
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.device = device

    def __call__(self, nsteps):
        return torch.randint(0, 2, (nsteps, 2, 256), dtype=torch.uint8, device=self.device, generator=self.gen)


class HelloSippyRTPipe:
    minlenratio: float = 0.0
    maxlenratio: float = 20.0
    threshold: float = 0.5
    chunk_size: int = 8
    pre_nframes: int = 2
    post_nframes: int = 2
    model_sr: int = 16000
    output_sr: int = 16000
    default_model = 'microsoft/speecht5_tts'
    cleanup_text: Optional[callable] = None

    def __init__(self, device, model=default_model, get_processor: Optional[callable] = None, output_sr: int = output_sr,
                 weights: Optional[Dict[str, dict]] = None, processor=None, speaker_embeddings=None, mask_source=None, **kwa):
        self.cuda_lock = InfernGlobals().torcher
        self.cleanup_text = kwa.get('cleanup_text', self.cleanup_text)
        self.device = dev = _lib.require_device(device)
        if weights is None:
            weights = load_pretrained_weights(model)
        if processor is None:
            processor = load_pretrained_processor(model, get_processor, dev)
        self.processor = processor
        with self.cuda_lock, torch.cuda.device(dev):
            self.model = SpeechT5(weights['speecht5_tts'], dev)
            self.vocoder = HifiGan(weights['hifigan'], dev)
            self.chunker = Amendment(weights['amendment'], dev)
        self.speaker_embeddings = speaker_embeddings if speaker_embeddings is not None else load_xvectors()
        self.resampler = get_resampler(self.model_sr, output_sr, str(dev)) if self.model_sr != output_sr else None
        self.output_sr = output_sr
        self.mask_source = mask_source if mask_source is not None else DeviceMaskSource(dev)

    def clone_for_lane(self) -> 'HelloSippyRTPipe':
        """A second engine over the SAME device weights with its own batch states, frame/vocoder buffers, captured
        graphs and dropout-mask stream: lets the synthesis of different utterance batches be in flight together
        (SpeechPipeline.run_steps) without a second copy of the models in HBM."""
        import copy
        c = copy.copy(self)
        c.model = copy.copy(self.model)
        c.model._states = {}
        c.vocoder = copy.copy(self.vocoder)
        c.vocoder._bufs = {}
        c.chunker = copy.copy(self.chunker)
        c.chunker._bufs = {}
        c.mask_source = DeviceMaskSource(self.device)
        return c

    def _render(self, st, par):
        """postnet -> carry + 4 overlapped chunks -> HiFi-GAN -> AmendmentNetwork1 for the call with
        parity `par`; writes st.render_out[par] (bf16 [B,8192]).  Pure kernel launches over buffers
        that persist with the state, so the whole pass is captured into one hipGraph per parity."""
        dev, B = self.device, st.B
        post = postnet(self.model, st, par)
        _lib.check(_lib.lib().ifh_tts_chunks_bf16(ops._addr(st.pre_frames), ops._addr(post), ops._addr(self.vocoder.mean),
                                                  ops._addr(self.vocoder.scale), ops._addr(st.voc_in), ops._addr(st.amd_mel), B,
                                                  _lib.stream_ptr(dev)), 'ifh_tts_chunks_bf16')
        audio = self.vocoder(st.voc_in, cache=st.voc_cache)      # buffers owned by the state, like the graphs over them
        self.chunker(st.amd_mel, audio, st.render_out[par], B, cache=st.amd_cache)

    def render(self, st, par, use_graphs=True):
        dev = self.device
        if not hasattr(st, 'voc_in'):
            st.voc_in = torch.empty((4 * st.B, 12, 80), dtype=torch.bfloat16, device=dev)
            st.amd_mel = torch.empty((4 * st.B, 12, 80), dtype=torch.bfloat16, device=dev)
            st.render_out = [torch.empty((st.B, 8192), dtype=torch.bfloat16, device=dev) for _ in range(2)]
            st.render_graphs, st.render_eager = {}, 0
            st.voc_cache, st.amd_cache = {}, {}
        if not use_graphs or st.render_eager < 2:          # first passes eager: loads kernels, sizes the vocoder buffers
            self._render(st, par)
            st.render_eager += 1
        else:
            g = st.render_graphs.get(par)
            if g is None:
                g = st.render_graphs[par] = _lib.CountedGraph(lambda: self._render(st, par))
            g.replay()
        return st.render_out[par]

    def decode_chunk(self, state: HelloSippyPipeStateBatched) -> int:
        """The 16 decoder steps of one infer() call (HelloSippyRTPipe.py:195-229); returns the frame-buffer
        parity the renderer must be given.  Asynchronous on the current stream."""
        st = state.dev
        par = st.ncalls & 1
        masks = self.mask_source(self.chunk_size * 4 // 2).to(self.device).contiguous()
        decoder_steps(self.model, st, masks, nsteps=self.chunk_size * 4 // 2, threshold=self.threshold)
        st.ncalls += 1
        return par

    def infer(self, state: HelloSippyPipeStateBatched) -> None:
        st = state.dev
        dev = self.device
        with self.cuda_lock, torch.cuda.device(dev):
            par = self.decode_chunk(state)
            out = self.render(st, par, use_graphs=self.model.use_graphs)
            state.stage = dict(post=st.post[par], voc_in=st.voc_in)
            if self.resampler is not None:
                out = self.resampler(out.float()).to(torch.bfloat16)
            st.audio = state.audio = out

    def unbatch_and_dispatch(self, state: HelloSippyPipeStateBatched):
        """HelloSippyRTPipe.py:242-259.  One D2H of the audio batch and of ends_at, then the
        reference's per-row slicing; returns False once every utterance has ended."""
        sr_rr = self.model_sr // self.output_sr
        idx = state.idx
        end_idx = idx - 1
        stepsize = 256 * 2 // sr_rr
        with self.cuda_lock:
            audio = state.audio.cpu()
            ends = state.ends_at.cpu().tolist()
            for i, dispatch in [(i, d) for i, d in enumerate(state.dispatch) if d is not None]:
                asize = audio[i].size(0)
                startoff = max(0, asize - ((idx - state.starts_at_host[i]) * stepsize))
                ends_at = ends[i]
                endoff = min(asize, asize - (((idx - ends_at) * stepsize) if ends_at >= 0 else 0))
                assert startoff <= endoff
                if startoff != endoff:
                    dispatch(audio[i][startoff:endoff].clone())
                if ends_at >= 0 and ends_at <= end_idx:
                    dispatch(None)
                    state.dispatch[i] = None
            if all(e >= 0 and e <= end_idx for e in ends):
                return False
        return True

    def get_rand_voice_id(self):
        return torch.randint(0, len(self.speaker_embeddings), (1,)).item()

    def get_rand_voice(self):
        s_index = self.get_rand_voice_id()
        return (self.speaker_embeddings[s_index], s_index)

    def get_voice(self, s_index: int):
        return self.speaker_embeddings[s_index]


class TTSGroup:
    """Utterances that joined the running batch of a ContinuousTTS together (one `submit`): their row slots, progress in
    decoder steps and output.  `done` is set once the last audio of the group has been produced; `done_event` (a HIP
    event on the render stream) orders consumers of `ulaw` on other streams."""

    def __init__(self, n, t_true, max_calls, dispatch, want_ulaw):
        import threading
        self.n, self.t_true, self.max_calls = n, t_true, max_calls
        self.dispatch = list(dispatch) if dispatch is not None else None
        self.want_ulaw = want_ulaw
        self.slots = None
        self.idx = self.calls = 0               # decoder steps / infer() calls whose results have been taken
        self.q_idx = self.q_calls = 0           # ... that have been queued (ContinuousTTS looks one call ahead)
        self._finished = False
        self.ulaw = self.valid = None
        self.spans = []
        self.done, self.done_event, self.error = threading.Event(), None, None

    def result(self, timeout=None):
        """Waits in slices: a group whose engine thread has died (or was never started and is not being stepped by hand)
        fails instead of waiting forever."""
        import time
        t_end = None if timeout is None else time.monotonic() + timeout
        while not self.done.wait(0.25 if t_end is None else max(0.0, min(0.25, t_end - time.monotonic()))):
            eng = getattr(self, '_engine', None)
            if eng is not None and eng.failed is not None:
                raise RuntimeError('ContinuousTTS: the engine thread has died') from eng.failed
            if t_end is not None and time.monotonic() >= t_end:
                raise TimeoutError('TTS group still running')
        if self.error is not None:
            raise self.error
        return self


class ContinuousTTS:
    """Continuous batching of HelloSippyRTPipe.infer (HelloSippyRTPipe.py:191-259) over ONE ragged decode batch
    (engines/speecht5.py:TTSRaggedState).  The reference's worker freezes a batch and loops it to the end
    (Cluster/InfernTTSWorker.py:83-92: "no continuous batching => tail waste"); here utterances join the
    running batch at the next infer() boundary and leave when they end, so however many utterance batches are in flight
    the GPU sees one launch chain of ~56 kernels per decoder step over all their rows instead of one chain per batch.
    Per row the arithmetic, the dispatch offsets (`unbatch_and_dispatch`, :242-259) and the end-of-utterance rule are
    the reference's.

    submit() may be called from any thread (the text encoder and the cross-attention K|V projection of the new
    utterances run on the CALLER's current stream); step() -- one infer() call over every live row: 16 decoder steps on the
    engine stream, postnet + HiFi-GAN + amendment (+ resample, mu-law) on a second stream overlapped with the next call's
    steps -- runs on the engine thread (start()/stop(), or driven by hand)."""

    def __init__(self, pp: 'HelloSippyRTPipe', max_rows=1024, max_text=64, row_bucket=128):
        import threading
        self.pp, self.device = pp, pp.device
        dev = pp.device
        assert max_text <= 256, 'cross-attention over > 256 keys takes the 4-wave kernel: such texts go through the frozen-batch path'
        self.row_bucket = row_bucket
        self.host_wait = False        # (see step()): measured without effect, off
        self.admit_ready = False      # (see _admit()): measured without effect, off
        self.sync_every = 8           # decoder steps queued at a time (0: all 16); 8: +2 % and a steadier tick p99 at C3
        with torch.cuda.device(dev):
            self.st = TTSRaggedState(pp.model, max_rows, max_text)
            # the decode chain is one latency-bound sequence of small dependent launches for ALL in-flight rows: on an ordinary
            # queue each of them waits for CU slots behind whatever long throughput kernels other stages have resident (the
            # Whisper encoder's GEMMs), serially; a high-priority queue lets them through
            self.main = torch.cuda.Stream(device=dev, priority=-1)
            self.side = _lib.throughput_stream(dev)          # postnet + HiFi-GAN + amendment passes
        R = self.st.R
        self.free = list(range(R))                       # row slots, lowest first
        self.pending, self.live = [], []
        self.cv = threading.Condition()
        self.ren_done = [None, None]
        self.h_active = torch.zeros(R, dtype=torch.uint8).pin_memory()
        self.h_fresh = [torch.zeros(R, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.h_ends = [torch.zeros(R, dtype=torch.int64).pin_memory() for _ in range(3)]      # end flags of the calls in flight
        self.h_audio = {}                                # pinned audio of a call for the dispatch callbacks, by (shape, call % 3)
        # queue call c + 1 before taking call c's results (step()).  Off by default: it closes the ~2.9 ms the decode chain stands empty
        # at every call boundary (per-step time in the pipelined C3 cycle 2.27 -> 2.08 ms) but C3's throughput does not move -- the GPU is
        # bound by the sum of the stages' work, not by this chain's latency (round 6, profiles/NOTES.md) -- and a session would hear
        # every chunk half a call later
        self.lookahead = False
        self._inflight = None
        self.render_bufs = {}
        self.thread, self.halt = None, False
        self.failed = None                               # the exception that killed the engine thread (submit() then raises)
        self.calls_run = self.rows_run = 0               # statistics: engine calls, sum of row slots they covered
        self.prof = {}                                   # statistics: host wall seconds of step() by phase

    # ---- any thread -----------------------------------------------------------------------------------------------
    def submit_each(self, input_ids, lens, speakers, dispatch):
        """submit() for utterances that arrived together but are each a group of their own (one utterance = one group: its maximum
        length follows its OWN text length, as in a batch of one -- what the worker does with a queue of requests): the text
        encoder and the cross-K|V projection run ONCE over all of them (50 sessions speaking at once were 50 encoder passes,
        65 of the 75 ms to their first frame), every row becomes a TTSGroup over its slice.  dispatch: one callable per row."""
        return self.submit(input_ids, lens, speakers, dispatch=dispatch, split_rows=True)

    def submit(self, input_ids, lens, speakers, max_calls=None, dispatch=None, want_ulaw=False, split_rows=False):
        """input_ids int [g,T] right-padded, lens int [g], speakers float [g,512].  max_calls bounds the infer() calls
        the group takes part in (None: until every row has ended).  Returns the TTSGroup (split_rows: one per row, as a list)."""
        from .engines.speecht5 import D, KVP
        pp, st, dev = self.pp, self.st, self.device
        g, t_true = input_ids.shape
        T = -(-t_true // 16) * 16
        if T > st.T or g > st.R:
            raise ValueError('ContinuousTTS: %d rows x %d tokens exceed the state (%d rows x %d tokens)' % (g, t_true, st.R, st.T))
        if self.failed is not None:
            raise RuntimeError('ContinuousTTS: the engine thread has died') from self.failed
        lens_h = [int(v) for v in lens.tolist()] if split_rows else None
        if split_rows:
            assert dispatch is not None and len(dispatch) == g and not want_ulaw
            grps = [TTSGroup(1, lens_h[i], max_calls, [dispatch[i]], False) for i in range(g)]
        else:
            grps = [TTSGroup(g, t_true, max_calls, dispatch, want_ulaw)]
        for grp in grps:
            grp._engine = self
        grp = grps[0]
        with torch.cuda.device(dev):
            ids = torch.nn.functional.pad(input_ids, (0, T - t_true)) if T != t_true else input_ids
            enc = pp.model.encode(ids, lens)                                            # [g, T, 768] on the caller's stream
            kvs = []
            for L in pp.model.dec_layers:
                kv = torch.empty((g, T, KVP), dtype=torch.bfloat16, device=dev)
                ops.linear(enc, L['cwkv'], L['cbkv'], kv, rows=g * T, k=D, n=2 * D, ldc=KVP)
                kvs.append(kv)
            spk = speakers.to(dev, torch.bfloat16).contiguous().view(g, 512)
            spn = torch.empty((g, 512), dtype=torch.bfloat16, device=dev)
            _lib.check(_lib.lib().ifh_l2norm_rows_bf16(ops._addr(spk), 512, g, ops._addr(spn), 512, _lib.stream_ptr(dev)),
                       'ifh_l2norm_rows_bf16')
            lens_d, ready = lens.to(dev, torch.int32), torch.cuda.Event()
            for t in kvs + [spn, lens_d]:
                t.record_stream(self.main)                   # consumed on the engine stream
            if split_rows:
                for i, gi in enumerate(grps):                # row i of the one encoder pass (views of the batch tensors)
                    gi._admit = dict(T=T, kvs=[kv[i:i + 1] for kv in kvs], spn=spn[i:i + 1], lens=lens_d[i:i + 1], ready=ready)
            else:
                grp._admit = dict(T=T, kvs=kvs, spn=spn, lens=lens_d, ready=ready)
            ready.record(torch.cuda.current_stream(dev))
        with self.cv:
            if self.failed is not None:                  # died between the check above and here
                raise RuntimeError('ContinuousTTS: the engine thread has died') from self.failed
            self.pending.extend(grps)
            self.cv.notify_all()
        return grps if split_rows else grp

    # ---- engine thread --------------------------------------------------------------------------------------------
    def _admit(self):
        st, dev = self.st, self.device
        took = []
        while True:
            wait_ev = None
            with self.cv:
                while self.pending and len(self.free) >= self.pending[0].n:
                    # A batch joins only once its text encoder / cross-K|V work (the submitter's stream, itself behind that
                    # cycle's STT stage) HAS finished: waiting for it in the decode stream would stall every live row -- and,
                    # being a barrier packet in the high-priority hardware queue, the real-time tick behind it -- for as long as
                    # that takes.  It joins at a later infer() boundary instead; with nothing live the engine has nothing better
                    # to do than wait (on the host, outside the lock).
                    rdy = self.pending[0]._admit['ready']
                    if self.admit_ready and not rdy.query():
                        if not self.live and not took:
                            wait_ev = rdy
                        break
                    grp = self.pending.pop(0)
                    self.free.sort()
                    grp.slots, self.free = self.free[:grp.n], self.free[grp.n:]
                    took.append(grp)
            if wait_ev is None:
                break
            wait_ev.synchronize()
        if not took:
            return
        R, T = st.R, st.T
        for grp in took:
            a = grp._admit
            self.main.wait_event(a['ready'])
            sl = torch.tensor(grp.slots, dtype=torch.int64).to(dev, non_blocking=True)
            for kv, cross in zip(a['kvs'], st.cross):
                cross.view(R, T, -1)[:, :a['T']].index_copy_(0, sl, kv)
            st.cat[:, 768:].index_copy_(0, sl, a['spn'])
            st.enc_len.index_copy_(0, sl, a['lens'])
            mm = torch.tensor([[0, int(grp.t_true * 20.0 / 2)]] * grp.n, dtype=torch.int32).to(dev, non_blocking=True)
            st.minmax.index_copy_(0, sl, mm)
            st.pos.index_fill_(0, sl, 0)
            st.ends_at.index_fill_(0, sl, -1)
            sl.record_stream(self.side)
            grp._slots_dev, grp._fresh = sl, True
            del grp._admit
            self.live.append(grp)

    def _bucket(self):
        hi = max(max(grp.slots) for grp in self.live) + 1
        return min(self.st.R, -(-hi // self.row_bucket) * self.row_bucket)

    def step(self) -> bool:
        """One infer() call over every live row: its 16 decoder steps and its render pass are QUEUED, then the results of the call
        before it are taken (end flags, dispatch, groups that finished) -- `lookahead`: the host work between two calls (waiting
        for the end flags, bookkeeping, admission, the next call's launches) used to leave the decode chain empty for ~2.9 ms per
        call, 8-10 % of a C3 cycle (tools/trace_gaps.py).  A group whose last call is known when it is queued (max_calls, or the
        KV cache's capacity) gives its row slots back at once; one that ends by the stop rule is computed for one call more than it
        needed (its rows are independent of the others: nothing changes for them).  Returns False when nothing is live, pending or
        in flight."""
        ctx = self._queue_call()
        if not self.lookahead:
            if ctx is None:
                return False
            self._finish_call(ctx)
            return True
        prev, self._inflight = self._inflight, ctx
        if prev is not None:
            self._finish_call(prev)
        return ctx is not None or prev is not None

    def _close(self, grp, par):
        """the group's row slots go back (the engine thread owns them; a later admission writes them behind everything queued)"""
        self.live.remove(grp)
        with self.cv:
            self.free.extend(grp.slots)

    def _queue_call(self):
        pp, st, dev = self.pp, self.st, self.device
        import time as _time
        prof = self.prof
        t_a = _time.perf_counter()
        with torch.cuda.device(dev), torch.cuda.stream(self.main):
            self._admit()
            if not self.live:
                return None
            t_b = _time.perf_counter()
            n = self._bucket()
            par = st.ncalls & 1
            if self.ren_done[par] is not None:
                # the renderer of call c-2 has released this parity's buffers.  Waited for on the HOST: as a stream wait it is a
                # barrier packet in the high-priority hardware queue, and everything mapped to that queue -- the real-time tick's
                # launches too -- stands behind it for as long as a render pass takes (the 40 ms worst ticks of round 4)
                if self.host_wait:
                    self.ren_done[par].synchronize()
                else:
                    self.main.wait_event(self.ren_done[par])
            self.h_active.zero_()
            self.h_fresh[par].zero_()
            for grp in self.live:
                self.h_active[grp.slots] = 1
                if grp._fresh:
                    self.h_fresh[par][grp.slots] = 1
                    grp._fresh = False
            st.active.copy_(self.h_active, non_blocking=True)
            st.fresh[par].copy_(self.h_fresh[par], non_blocking=True)
            masks = pp.mask_source(16).to(dev).contiguous()
            ragged_decoder_steps(pp.model, st, masks, n, nsteps=16, threshold=pp.threshold, sync_every=self.sync_every)
            # this call's end flags, as they stand behind its 16 steps (a later call's admissions write behind this copy)
            k = self.calls_run % len(self.h_ends)
            ends_host, ends_ev = self.h_ends[k], torch.cuda.Event()
            ends_host[:n].copy_(st.ends_at[:n], non_blocking=True)
            ends_ev.record(self.main)
            t_c = _time.perf_counter()
            dec_done = torch.cuda.Event()
            dec_done.record(self.main)
            rr = pp.model_sr // pp.output_sr
            A, stepsize = 8192 // rr, 512 // rr
            with torch.cuda.stream(self.side):
                self.side.wait_event(dec_done)
                audio = self._render(par, n)                                    # bf16 [n, 8192] @ model_sr
                if pp.resampler is not None:
                    audio = pp.resampler(audio.float())
                ul = None
                if any(grp.want_ulaw for grp in self.live):
                    pcm = audio if audio.dtype == torch.float32 else audio.float()
                    ul = torch.empty((n, A), dtype=torch.uint8, device=dev)
                    _lib.check(_lib.lib().ifh_g711_encode_f32_u8(_lib.ptr(pcm), _lib.ptr(ul), pcm.numel(), _lib.stream_ptr(dev)),
                               'ifh_g711_encode_f32_u8')
                for grp in self.live:
                    if grp.want_ulaw:
                        if grp.ulaw is None:
                            grp.ulaw = torch.empty((grp.n, (grp.max_calls or 64) * A), dtype=torch.uint8, device=dev)
                            grp.valid = torch.zeros(grp.n, dtype=torch.int64)
                        c = grp.q_calls
                        if (c + 1) * A > grp.ulaw.size(1):
                            grp.ulaw = torch.cat([grp.ulaw, torch.empty_like(grp.ulaw)], 1)
                        grp.ulaw[:, c * A:(c + 1) * A] = ul.index_select(0, grp._slots_dev)
                host_audio = None
                if any(grp.dispatch is not None for grp in self.live):
                    # as unbatch_and_dispatch: one D2H of the batch -- into pinned memory, waited for when the call's results are taken
                    dev_audio = (audio if audio.dtype == torch.bfloat16 else audio.to(torch.bfloat16))
                    key = (tuple(dev_audio.shape), self.calls_run % 3)
                    host_audio = self.h_audio.get(key)
                    if host_audio is None:
                        host_audio = self.h_audio[key] = torch.empty(dev_audio.shape, dtype=torch.bfloat16).pin_memory()
                    host_audio.copy_(dev_audio, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.ren_done[par] = ev
            t_d = _time.perf_counter()
        self.calls_run += 1
        self.rows_run += n
        groups = []
        for grp in list(self.live):
            grp.q_idx += 16
            grp.q_calls += 1
            # known now: this was the group's last call (its call budget, or the KV cache is full) -> its slots are free for the
            # next admission, which is queued behind everything above
            last = (grp.max_calls is not None and grp.q_calls >= grp.max_calls) or (grp.q_idx + 16 > st.smax)
            groups.append((grp, grp.q_idx, grp.q_calls, last, grp.q_idx + 16 > st.smax))
            if last:
                self._close(grp, par)
        for k_, v in (('admit', t_b - t_a), ('steps', t_c - t_b), ('render', t_d - t_c)):
            prof[k_] = prof.get(k_, 0.0) + v
        return dict(par=par, n=n, groups=groups, ends_host=ends_host, ends_ev=ends_ev, host_audio=host_audio, ren_ev=ev, A=A, stepsize=stepsize)

    def _finish_call(self, ctx):
        """take the results of a queued call: per group the dispatch offsets of HelloSippyRTPipe.py:242-259 and the end of the utterance"""
        import time as _time
        st, dev = self.st, self.device
        A, stepsize, par = ctx['A'], ctx['stepsize'], ctx['par']
        t_d = _time.perf_counter()
        ctx['ends_ev'].synchronize()                 # waits for this call's decoder steps (the reference's .item())
        ends_all = ctx['ends_host'].numpy()
        host_audio = ctx['host_audio']
        if host_audio is not None:
            ctx['ren_ev'].synchronize()
        t_e = _time.perf_counter()
        finished = []
        for grp, idx, calls, last, out_of_cache in ctx['groups']:
            if grp._finished:                        # ended by the stop rule in the call before: this one was computed for nothing
                continue
            grp.idx, grp.calls = idx, calls
            end_idx = idx - 1
            e = ends_all[grp.slots]
            s_off = max(0, A - (idx - 1) * stepsize)                        # starts_at = post_nframes // 2 = 1 for every row
            e_off = np.where(e >= 0, np.minimum(A, A - (idx - e) * stepsize), A)
            if grp.want_ulaw:
                grp.valid += torch.from_numpy(np.maximum(0, e_off - s_off))
                grp.spans.append(list(zip([s_off] * grp.n, np.maximum(s_off, e_off).tolist())))
            if grp.dispatch is not None:
                for i, d in enumerate(grp.dispatch):
                    if d is None:
                        continue
                    so, eo = s_off, int(e_off[i])
                    assert so <= eo
                    if so != eo:
                        d(host_audio[grp.slots[i]][so:eo].clone())
                    if 0 <= e[i] <= end_idx:
                        d(None)
                        grp.dispatch[i] = None
            ended = bool(np.all((e >= 0) & (e <= end_idx)))
            if out_of_cache and not ended and grp.dispatch is not None:
                for i, d in enumerate(grp.dispatch):      # the KV cache is full: the utterance is cut here, the session is told
                    if d is not None:
                        d(None)
                        grp.dispatch[i] = None
            if ended or last:
                finished.append((grp, last))
        for grp, last in finished:
            grp._finished = True
            if grp in self.live:                          # (a group closed when its last call was queued has given its slots back)
                self._close(grp, par)
            grp.done_event = ctx['ren_ev']
            if grp.ulaw is not None:
                # [n, max_calls*A] as the lanes path returns it: calls the group did not take (it ended early) are mu-law
                # silence (0xFF), `valid`/`spans` say what is audio; without max_calls the width is what was produced
                if grp.max_calls is not None and grp.calls < grp.max_calls:
                    with torch.cuda.device(dev), torch.cuda.stream(self.side):
                        grp.ulaw[:, grp.calls * A:grp.max_calls * A] = 0xFF
                        grp.done_event = torch.cuda.Event()
                        grp.done_event.record(self.side)
                grp.ulaw = grp.ulaw[:, :(grp.max_calls or grp.calls) * A]
            grp.done.set()
        t_f = _time.perf_counter()
        # host wall seconds by phase: admission, queueing the 16 decoder steps (incl. the bounded-queue waits), queueing the render
        # pass (_queue_call); waiting for a call's end flags, per-group bookkeeping / dispatch (here)
        for k, v in (('ends_wait', t_e - t_d), ('book', t_f - t_e)):
            self.prof[k] = self.prof.get(k, 0.0) + v

    def _render(self, par, n):
        """postnet -> carry + 4 overlapped chunks -> HiFi-GAN -> AmendmentNetwork1 over the first n row slots; one
        hipGraph per (parity, n) over buffers that live with this engine."""
        pp, st, dev = self.pp, self.st, self.device
        b = self.render_bufs.get(n)
        if b is None:
            b = self.render_bufs[n] = dict(
                voc_in=torch.empty((4 * n, 12, 80), dtype=torch.bfloat16, device=dev),
                amd_mel=torch.empty((4 * n, 12, 80), dtype=torch.bfloat16, device=dev),
                out=[torch.empty((n, 8192), dtype=torch.bfloat16, device=dev) for _ in range(2)],
                graphs={}, eager=0, voc_cache={}, amd_cache={})

        def run():
            post = postnet(pp.model, st, par, B=n)
            _lib.check(_lib.lib().ifh_tts_chunks_rows_bf16(ops._addr(st.pre_frames), ops._addr(post), ops._addr(pp.vocoder.mean),
                                                           ops._addr(pp.vocoder.scale), ops._addr(b['voc_in']), ops._addr(b['amd_mel']),
                                                           ops._addr(st.fresh[par]), n, _lib.stream_ptr(dev)), 'ifh_tts_chunks_rows_bf16')
            audio = pp.vocoder(b['voc_in'], cache=b['voc_cache'])
            pp.chunker(b['amd_mel'], audio, b['out'][par], n, cache=b['amd_cache'])
        if not pp.model.use_graphs or b['eager'] < 2:
            run()
            b['eager'] += 1
        else:
            g = b['graphs'].get(par)
            if g is None:
                g = b['graphs'][par] = _lib.CountedGraph(run)
            g.replay()
        return b['out'][par]

    def warm(self, rows):
        """Untimed preparation for the row counts `rows` (multiples of row_bucket): with no live utterance (every slot
        inactive, positions frozen) run the decoder steps and the renderer often enough to load the kernels, size the
        buffers and capture the hipGraphs of both frame-buffer parities -- so that no capture happens while other threads
        are launching.  Only with the engine idle."""
        pp, st, dev = self.pp, self.st, self.device
        assert not self.live and not self.pending
        with torch.cuda.device(dev), torch.cuda.stream(self.main):
            st.active.zero_()
            for n in sorted(set(min(st.R, -(-r // self.row_bucket) * self.row_bucket) for r in rows)):
                for _ in range(4 + (st.ncalls & 1)):
                    masks = torch.zeros((16, 2, 256), dtype=torch.uint8, device=dev)
                    par = ragged_decoder_steps(pp.model, st, masks, n, nsteps=16, threshold=pp.threshold)
                    self._render(par, n)
                    self.main.synchronize()
            if st.ncalls & 1:                                       # leave the parity where a fresh engine starts
                st.ncalls += 1
            st.fresh[0].zero_()
            st.fresh[1].zero_()
            self.main.synchronize()

    # ---- background engine thread -----------------------------------------------------------------------------------
    def start(self):
        import threading
        assert self.thread is None
        self.halt = False

        def loop():
            torch.cuda.set_device(self.device)
            while True:
                with self.cv:
                    while not self.halt and not self.pending and not self.live and self._inflight is None:
                        self.cv.wait()
                    if self.halt and not self.pending and not self.live and self._inflight is None:
                        return
                try:
                    self.step()
                except BaseException as e:                  # fail every waiting group loudly, mark the engine dead, re-raise
                    with self.cv:
                        self.failed = e
                        for grp in self.live + self.pending:
                            grp.error = e
                            grp.done.set()
                        self.live, self.pending = [], []
                        self.free = list(range(self.st.R))
                    raise
        self.thread = threading.Thread(target=loop, daemon=True, name='ContinuousTTS')
        self.thread.start()
        return self

    def stop(self):
        with self.cv:
            self.halt = True
            self.cv.notify_all()
        if self.thread is not None:
            self.thread.join()
            self.thread = None


def load_pretrained_weights(model_name):
    """Checkpoints the reference loads (HelloSippyRTPipe.py:164-178).  Needs the HF hub."""
    from transformers import SpeechT5ForTextToSpeech, SpeechT5HifiGan
    from huggingface_hub import hf_hub_download
    from safetensors.torch import load_file
    tts = SpeechT5ForTextToSpeech.from_pretrained(model_name).state_dict()
    voc = SpeechT5HifiGan.from_pretrained('microsoft/speecht5_hifigan').state_dict()
    amd = load_file(hf_hub_download('sobomax/speecht5-rt.post_vocoder.v2', 'model.safetensors'))
    return {'speecht5_tts': tts, 'hifigan': voc, 'amendment': amd}


def load_pretrained_processor(model_name, get_processor, device):
    from transformers import SpeechT5Processor
    if get_processor is not None:
        return get_processor(device, model_name)
    return SpeechT5Processor.from_pretrained(model_name)


def load_xvectors():
    from datasets import load_dataset
    ds = load_dataset('Matthijs/cmu-arctic-xvectors', split='validation')
    return [torch.tensor(ed['xvector'], device='cpu').unsqueeze(0) for ed in sorted(ds, key=lambda x: x['filename'])]


class cleanup_text_eu:
    """Transliteration of accented characters before tokenisation (InfernTTSWorker.py:22-35)."""
    pairs = ('ÄE ÆE ÇC ÉE ÍI ÓO ÖE ÜY ßS àa áa ãa äe åa ëe íi ïi ðo ñn òo óo ôo öu úu üy ýy Āa āa ăa ąa ćc ČC čc ďd ĐD ęe ěe '
             'ğg İI ОO ŁL ńn ňn ŌO ōo őo řr ŚS śs ŞS şs ŠS šs ūu źz ŻZ ŽZ ǐi șs țt ùu').replace('Āa', 'ĀA').split()
    table = str.maketrans(''.join(p[0] for p in pairs), ''.join(p[1] for p in pairs))

    def __call__(self, text):
        return text.translate(self.table)


def get_ja_T5Processor(device, model_name):
    """The 'ja' processor hook of Cluster/InfernTTSWorker.py:9-20 (OpenJTalk tokenizer + SpeechT5 feature extractor).  Its
    tokenizer lives in `utils.speecht5_openjtalk_tokenizer`, a module the reference imports but does not ship; it is
    picked up from the host application when that provides it."""
    try:
        from utils.speecht5_openjtalk_tokenizer import SpeechT5OpenjtalkTokenizer
    except ImportError as e:
        raise ImportError("lang 'ja' needs utils.speecht5_openjtalk_tokenizer (imported by the reference at "
                          "Cluster/InfernTTSWorker.py:10 but absent from its tree): provide it on sys.path") from e
    from transformers import SpeechT5FeatureExtractor, SpeechT5Processor
    tok = SpeechT5OpenjtalkTokenizer.from_pretrained(model_name)
    tok._in_target_context_manager, tok.split_special_tokens, tok._added_tokens_encoder, tok._unk_token = False, True, {}, None
    return SpeechT5Processor(SpeechT5FeatureExtractor.from_pretrained(model_name), tok)


lang2model = {'en': {'cleanup_text': cleanup_text_eu()},
              'it': {'model': 'Sandiago21/speecht5_finetuned_voxpopuli_it', 'cleanup_text': cleanup_text_eu()},
              'es': {'model': 'Sandiago21/speecht5_finetuned_facebook_voxpopuli_spanish', 'cleanup_text': cleanup_text_eu()},
              'fr': {'model': 'Sandiago21/speecht5_finetuned_facebook_voxpopuli_french', 'cleanup_text': cleanup_text_eu()},
              'de': {'model': 'JFuellem/speecht5_finetuned_voxpopuli_de', 'cleanup_text': cleanup_text_eu()},
              'pt': {'model': 'evertonaleixo/speecht5_finetuned_fleurs_ptbr', 'cleanup_text': cleanup_text_eu()},
              'ru': {'model': 'zaebee/speecht5_tts_common_ru'},
              'ja': {'model': 'esnya/japanese_speecht5_tts', 'get_processor': get_ja_T5Processor}}


class InfernTTSWorker(InfernBatchedWorker):
    max_batch_size: int = 8
    debug = False
    tts_engine: HelloSippyRTPipe
    output_sr: int

    def __init__(self, lang, output_sr, device=None, continuous=False, max_rows=256, max_text=128, **engine_kwa):   # max_text <= 256
        """continuous=True: requests join ONE running decode batch at the next infer() boundary instead of waiting for the frozen
        batch in front of them to loop to its end (ContinuousTTS; max_rows row slots, texts of up to max_text tokens -- a longer
        text takes the frozen-batch path).  Every request receives the same audio either way."""
        super().__init__()
        kwa = dict(lang2model[lang])
        kwa.update(engine_kwa)
        self.tts_engine = HelloSippyRTPipe(_lib.require_device(device), output_sr=output_sr, **kwa)
        self.output_sr = output_sr
        self.continuous, self._cont_shape = bool(continuous), (max_rows, max_text)

    def run(self):
        if not self.continuous:
            return super().run()
        from queue import Empty
        from .workers import RTPWrkTRun
        self.thread_started()
        torch.cuda.set_device(self.tts_engine.device)
        eng = ContinuousTTS(self.tts_engine, max_rows=self._cont_shape[0], max_text=self._cont_shape[1], row_bucket=16)
        while self.get_state() == RTPWrkTRun:
            wis, stop = [], False
            while len(wis) < self.max_batch_size:           # block for work only while nothing is being synthesised
                try:
                    wi = self.inf_queue.get(block=not (wis or eng.live or eng.pending))
                except Empty:
                    break
                if wi is None:
                    stop = True
                    break
                wis.append(wi)
            if wis:
                for wi in wis:
                    cb = getattr(wi, '_proc_start_cb', None)
                    if cb is not None:
                        cb()
                states = [HelloSippyPipeState(self.tts_engine, r) for r in wis]
                fits = [st for st in states if st.inputs.size(1) <= eng.st.T]
                rest = [(st, r) for st, r in zip(states, wis) if st.inputs.size(1) > eng.st.T]
                if fits:
                    T = max(st.inputs.size(1) for st in fits)
                    ids = torch.zeros((len(fits), T), dtype=torch.int32)
                    lens = torch.zeros(len(fits), dtype=torch.int32)
                    for i, st in enumerate(fits):
                        n = st.inputs.size(1)
                        ids[i, :n] = st.inputs[0].to(torch.int32)
                        lens[i] = n
                    spk = torch.cat([st.speaker_embeddings.reshape(1, 512).float() for st in fits])
                    # one utterance = one group: its maxlen follows its own text length, as in a batch of one; the encoder work of
                    # the requests that arrived together is one pass
                    eng.submit_each(ids, lens, spk, [st.dispatch for st in fits])
                if rest:
                    self.process_batch([r for _, r in rest])
            if stop:
                while eng.step():
                    pass
                break
            eng.step()

    def process_batch(self, wis: List[HelloSippyPlayRequest]):
        new_states = [HelloSippyPipeState(self.tts_engine, r) for r in wis]
        state = HelloSippyPipeStateBatched(new_states, self.tts_engine)
        while True:
            try:
                self.tts_engine.infer(state)
            except RuntimeError as e:
                self.handle_runtime_error(e, wis, state)
                raise
            if not self.tts_engine.unbatch_and_dispatch(state):
                break

    def handle_runtime_error(self, e, wis, state):
        print(f'InfernTTSWorker.handle_runtime_error: {e}')

    def get_voice(self, *args):
        return self.tts_engine.get_voice(*args)

    def get_rand_voice(self):
        return self.tts_engine.get_rand_voice()

    def get_rand_voice_id(self):
        return self.tts_engine.get_rand_voice_id()


class TTSRequest:
    def __init__(self, text: Union[str, List[str], Tuple[str]], speaker_id: Optional[int] = None,
                 done_cb: Optional[callable] = None):
        self.text, self.speaker_id, self.done_cb = text, speaker_id, done_cb


class TTSSndDispatch:
    debug: bool = False
    cancelled: bool = False
    cleanup_cb: Optional[callable] = None

    def __init__(self, soundout: callable, output_sr: int, done_cb: Optional[callable]):
        self.id = uuid4()
        self.soundout, self.output_sr, self.done_cb = soundout, output_sr, done_cb

    def _end_marker(self):
        return ASMarkerNewSent() if self.done_cb is None else ASMarkerSentDoneCB(self.done_cb, sync=True)

    def cancel(self):
        self.cancelled = True
        self.soundout(chunk=self._end_marker())
        if self.cleanup_cb is not None:
            self.cleanup_cb()

    def sound_dispatch(self, chunk):
        if self.cancelled:
            return
        finished = chunk is None
        if finished:
            chunk = self._end_marker()
        elif not isinstance(chunk, ASMarkerGeneric):
            assert chunk.size(0) > 0
            chunk = AudioChunk(chunk, self.output_sr)
        self.soundout(chunk=chunk)
        if finished and self.cleanup_cb is not None:
            self.cleanup_cb()


class TTSSession:
    debug = False

    def __init__(self, tts: InfernTTSWorker, tts_actr):
        self.id = uuid4()
        self.tts, self.tts_actr = tts, tts_actr
        self.active_req: Dict[UUID, TTSSndDispatch] = {}

    def start(self, soundout: callable):
        self.soundout = soundout

    def say(self, req: TTSRequest) -> UUID:
        if req.speaker_id is not None:
            speaker = self.tts.get_voice(req.speaker_id)
        else:
            speaker, req.speaker_id = self.tts.get_rand_voice()
        if isinstance(req.text, str):
            req.text = (req.text,)
        text, done_cb = req.text[0], req.done_cb
        if len(req.text) > 1:          # chain the remaining sentences through the actor (TTSSession.py:113-115)
            req.text = list(req.text)[1:]
            done_cb = partial(self.tts_actr.tts_session_say.remote, rgen_id=self.id, req=req)
        trd = TTSSndDispatch(self.soundout, self.tts.output_sr, done_cb)
        trd.cleanup_cb = partial(self.active_req.pop, trd.id, None)
        self.active_req[trd.id] = trd
        self.tts.infer(HelloSippyPlayRequest(self.id, text, speaker, trd.sound_dispatch))
        return trd.id

    def stop_saying(self, rsay_id: UUID):
        trd = self.active_req.get(rsay_id)
        if trd is None:
            return False
        trd.cancel()
        return True

    def stop(self):
        pass
