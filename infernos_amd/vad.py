"""Voice-activity detection: windowing, hysteresis state machine, chunk assembly.

Interface of Core/VAD/SileroVAD.py:12-112 (VADChannel, SileroVADWorker) and
Core/VAD/SileroVADUtils.py:4-133 (VADChannelState, VADBatchState, VADBatchFromList,
VADIteratorB).  The per-channel state machine and the active-buffer bookkeeping run on
the HIP device (ifh_vad_fsm_step / ifh_vad_step, csrc/dsp.hip), batched over channels;
the host keeps a mirror of the small integer state so callers can read the same
attributes the reference exposes.

The speech-probability network of the reference (Silero v3.1 TorchScript from torch.hub)
is third-party and not obtainable offline.  It is a plug-in here: `model(x[B,W], sr) ->
prob[B]` (device tensors).  A model that carries recurrent state the way the Silero TorchScript
object does (`model._c._h`, `model._c._c`: [2,B,64] each) gets the per-channel state injected
before and saved after every call, exactly as SileroVADUtils.py:21-26,99,131 do -- the state of
all attached channels lives in one device table indexed by slot.  The default,
EnergyVADModel, is a documented stateless stand-in (ifh_vad_energy_prob), not a Silero equivalent.
"""
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from .audio import AudioChunk, VadAudioChunk
from .workers import InfernBatchedWorker

WINDOW = 768
ABUF_CAP = 240000 + 768
EMIT_CAP = 240000


class VADChannelState:
    triggered: bool = False
    temp_end: int = 0
    current_sample: int = 0
    speech: Optional[Dict[str, int]] = None
    model_state: List[torch.Tensor]

    def __init__(self, device: str = 'cpu'):
        # LSTM state slots of the Silero model ([2,64] x 2); kept for plug-in models that use them
        self.model_state = [torch.zeros(2, 64, device=device), torch.zeros(2, 64, device=device)]


class VADBatchState:
    def __init__(self, batch_size, device: str = 'cpu'):
        self.batch_size = batch_size
        self.channels = [VADChannelState(device) for _ in range(batch_size)]

    def get_model_state(self):
        return [torch.stack([s.model_state[r] for s in self.channels], dim=1) for r in range(2)]

    def save_model_state(self, state: List[torch.Tensor]):
        for c, s1, s2 in zip(self.channels, state[0].unbind(1), state[1].unbind(1)):
            c.model_state = [s1, s2]


class VADBatchFromList(VADBatchState):
    def __init__(self, states: List[VADChannelState]):
        self.batch_size = len(states)
        self.channels = states


def _stateful(model) -> bool:
    """True when `model` exposes the Silero JIT object's recurrent-state attributes (SileroVADUtils.py:99,131)."""
    return hasattr(model, '_c')


def _inject_state(model, h: torch.Tensor, c: torch.Tensor, sr: int, n: int):
    mc = model._c
    mc._h, mc._c, mc._last_sr, mc._last_batch_size = h, c, sr, n


class EnergyVADModel:
    """Stand-in speech-probability model: p = sigmoid(0.5*(10*log10(mean(x^2)+1e-10)+30))."""

    def __init__(self, device=None):
        self.device = _lib.require_device(device)

    def reset_states(self):
        pass

    def __call__(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        x = x.to(self.device, torch.float32).contiguous()
        assert x.dim() == 2 and x.size(1) == WINDOW
        n = x.size(0)
        slot = torch.arange(n, dtype=torch.int32, device=self.device)
        prob = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ifh_vad_energy_prob(_lib.ptr(x), _lib.ptr(slot), n, _lib.ptr(prob),
                                                      _lib.stream_ptr(self.device)), 'ifh_vad_energy_prob')
        return prob


class RecurrentVADModel:
    """A stateful speech-probability model with the interface the reference's iterator drives (SileroVADUtils.py:99,131: the
    recurrent state is injected into `model._c._h` / `._c` before a call and read back after it) and the SHAPE of the detector the
    reference loads (SileroVAD.py:44-45: conv front end + 2 x LSTM(64), state [2,B,64] x 2): csrc/vadnet.hip.  The weights are
    seeded, not Silero's (the model file is not obtainable offline): its probabilities mean nothing, its cost and its state
    plumbing are the real thing.  PARITY UNPINNED against Silero; the arithmetic is pinned to the plain-PyTorch restatement the tests
    hold (tests/test_vadnet_gpu.py)."""

    class _State:
        _h = None
        _c = None
        _last_sr = 0
        _last_batch_size = 0

    def __init__(self, device=None, seed: int = 0, weights=None):
        """weights: None -> seeded (cost and state plumbing only); 'distilled' -> the weights fitted to the energy rule
        (weights.load_vadnet_distilled: usable decisions); or a state dict in torch's module layouts."""
        from .weights import load_vadnet_distilled, pack_vadnet, synth_vadnet
        self.device = _lib.require_device(device)
        self.sd = synth_vadnet(seed) if weights is None else (load_vadnet_distilled() if isinstance(weights, str) else weights)
        self.blob = pack_vadnet(self.sd).to(self.device)
        assert self.blob.numel() == _lib.lib().ifh_vadnet_weight_floats()
        self._c = RecurrentVADModel._State()

    def reset_states(self):
        self._c._h = self._c._c = None
        self._c._last_sr = self._c._last_batch_size = 0

    def __call__(self, x: torch.Tensor, sr: int) -> torch.Tensor:
        dev = self.device
        x = x.to(dev, torch.float32).contiguous()
        assert x.dim() == 2 and x.size(1) == WINDOW
        n = x.size(0)
        st = self._c
        if st._h is None or st._h.size(1) != n or st._last_sr != sr:        # as the JIT model resets on a batch / rate change
            st._h = torch.zeros((2, n, 64), dtype=torch.float32, device=dev)
            st._c = torch.zeros((2, n, 64), dtype=torch.float32, device=dev)
        h_in, c_in = st._h.to(dev, torch.float32).contiguous(), st._c.to(dev, torch.float32).contiguous()
        h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
        prob = torch.empty(n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_vadnet_prob(_lib.ptr(x), n, _lib.ptr(self.blob), _lib.ptr(h_in), _lib.ptr(c_in), _lib.ptr(h_out),
                                                  _lib.ptr(c_out), _lib.ptr(prob), _lib.stream_ptr(dev)), 'ifh_vadnet_prob')
        st._h, st._c, st._last_sr, st._last_batch_size = h_out, c_out, sr, n
        return prob


class VADIteratorB:
    """Batched streaming VAD iterator (SileroVADUtils.py:30-133): model call + FSM."""

    def __init__(self, model, threshold: float = 0.5, sampling_rate: int = 16000,
                 min_silence_duration_ms: int = 100, speech_pad_ms: int = 30):
        if sampling_rate not in (8000, 16000):
            raise ValueError('VADIterator does not support sampling rates other than [8000, 16000]')
        if min_silence_duration_ms != 100 or speech_pad_ms != 30:
            raise ValueError('the device FSM implements the reference constants (100 ms / 30 ms) only')
        self.model = model
        self.threshold = threshold
        self.sampling_rate = sampling_rate
        self.min_silence_samples = sampling_rate * min_silence_duration_ms / 1000
        self.speech_pad_samples = sampling_rate * speech_pad_ms / 1000
        self.model.reset_states()

    def __call__(self, x: torch.Tensor, bstate: Optional[VADBatchState] = None, return_seconds=False):
        if not torch.is_tensor(x):
            x = torch.Tensor(x)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        assert x.dim() == 2, f'Audio should be 1D or 2D tensor, but got {x.dim()}'
        dev = _lib.require_device(x.device if x.is_cuda else None)
        n = x.size(0)
        if bstate is None:
            bstate = VADBatchState(n, device=str(dev))
            self.model.reset_states()
        else:
            assert bstate.batch_size == n, f'Batch size should be {n}, but got {bstate.batch_size}'
            if _stateful(self.model):           # SileroVADUtils.py:99
                h, c = bstate.get_model_state()
                _inject_state(self.model, h.to(dev), c.to(dev), self.sampling_rate, n)
        window = x.size(1)
        probs = self.model(x.to(dev), self.sampling_rate).to(dev, torch.float32).contiguous()
        if _stateful(self.model):               # SileroVADUtils.py:131
            bstate.save_model_state([self.model._c._h, self.model._c._c])
        st = torch.tensor([[int(c.triggered), int(c.temp_end), int(c.current_sample), -1] for c in bstate.channels],
                          dtype=torch.int64, device=dev)
        slot = torch.arange(n, dtype=torch.int32, device=dev)
        ev = torch.empty((n, 2), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ifh_vad_fsm_step(_lib.ptr(probs), _lib.ptr(slot), n, window, self.sampling_rate,
                                                   float(self.threshold), _lib.ptr(st), _lib.ptr(ev),
                                                   _lib.stream_ptr(dev)), 'ifh_vad_fsm_step')
        st_h, ev_h = st.cpu().tolist(), ev.cpu().tolist()
        for c, s, e in zip(bstate.channels, st_h, ev_h):
            c.triggered, c.temp_end, c.current_sample = bool(s[0]), s[1], s[2]
            c.speech = _speech_dict(e[0], e[1], self.sampling_rate, return_seconds)
        return bstate


def _speech_dict(kind, pos, sr, return_seconds=False):
    if kind == 0:
        return None
    v = int(pos) if not return_seconds else round(pos / sr, 1)
    return {'start': v} if kind == 1 else {'end': v}


class VADChannel:
    """One call's VAD front end (SileroVAD.py:12-35): byte FIFO -> decoded 768-sample windows."""
    vad_buffer: bytes = b''
    active_start: Optional[int] = None

    def __init__(self, audio_chunk_in: callable, vad_chunk_in: callable, decode: callable, device: str):
        self.audio_chunk_in = audio_chunk_in
        self.vad_chunk_in = vad_chunk_in
        self.decode = decode
        self.state = VADChannelState('cpu')
        self.buf_len = 0          # len(active_buffer) in the reference
        self._slot = None         # row in the owning worker's device tables
        self._owner = None
        self._fin = None

    def detach(self):
        """Give the channel's row of the worker's device tables back (call teardown).  Also happens when the
        channel object is garbage-collected: table size follows the CONCURRENT number of calls."""
        if self._fin is not None:
            self._fin()                       # runs SileroVADWorker._release once
        self._owner = self._slot = self._fin = None

    @property
    def active_buffer(self) -> torch.Tensor:
        """The channel's pending audio (device view; SileroVAD.py:20,26)."""
        if self._owner is None:
            return torch.zeros(0)
        return self._owner._abuf[self._slot, :self.buf_len]

    def ingest(self, svad: 'SileroVADWorker', data: bytes, codec):
        self.vad_buffer += data
        if codec.e2d_frames(len(self.vad_buffer), svad.input_sr) < svad.window_size_samples:
            return None
        nbytes = codec.d2e_frames(svad.window_size_samples, svad.input_sr)
        chunk = codec.decode(self.vad_buffer[:nbytes], sample_rate=svad.input_sr)
        assert chunk.audio.size(0) == svad.window_size_samples, \
            f'{chunk.audio.size(0)=} {svad.window_size_samples=}'
        self.vad_buffer = self.vad_buffer[nbytes:]
        svad.infer((self, chunk))


class SileroVADWorker(InfernBatchedWorker):
    max_batch_size: int = 200
    input_sr: int
    max_vad_frames: int

    def __init__(self, device, input_sr: int = 8000, model=None, max_channels: int = 64):
        super().__init__()
        self.device = _lib.require_device(device)
        self.model = model if model is not None else EnergyVADModel(self.device)
        self.vad_iterator = VADIteratorB(self.model, sampling_rate=input_sr)
        self.window_size_samples = WINDOW
        self.input_sr = input_sr
        self.max_vad_frames = input_sr * 30
        self._cap = 0
        self._nslots = 0          # high-water mark of rows ever handed out
        self._free = []           # released rows, reused before the tables grow
        self._grow(max_channels)

    # -- device tables -------------------------------------------------------------------
    def _grow(self, cap):
        dev = self.device
        new = dict(win=torch.zeros((cap, WINDOW), dtype=torch.float32, device=dev),
                   st=torch.zeros((cap, 4), dtype=torch.int64, device=dev),
                   blen=torch.zeros(cap, dtype=torch.int32, device=dev),
                   abuf=torch.zeros((cap, ABUF_CAP), dtype=torch.float32, device=dev),
                   emit=torch.empty((cap, EMIT_CAP), dtype=torch.float32, device=dev),
                   # recurrent state of a stateful model, [h|c][2, cap, 64] (SileroVADUtils.py:11: two [2,64] per channel)
                   mh=torch.zeros((2, cap, 64), dtype=torch.float32, device=dev),
                   mc=torch.zeros((2, cap, 64), dtype=torch.float32, device=dev))
        new['st'][:, 3] = -1
        if self._cap:
            n = self._cap
            for k, old in (('win', self._win), ('st', self._st), ('blen', self._blen)):
                new[k][:n] = old
            new['mh'][:, :n], new['mc'][:, :n] = self._mh, self._mc
            new['abuf'][:n] = self._abuf
            self._abuf = None                 # (the old rows are released as soon as the copy has been enqueued)
        self._win, self._st, self._blen, self._abuf, self._emit, self._mh, self._mc = (
            new[k] for k in ('win', 'st', 'blen', 'abuf', 'emit', 'mh', 'mc'))
        self._cap = cap

    def _attach(self, ch: VADChannel):
        if ch._owner is self:
            return
        if self._free:
            slot = self._free.pop()
            self._st[slot] = torch.tensor([0, 0, 0, -1], dtype=torch.int64, device=self.device)
            self._blen[slot] = 0
        else:
            if self._nslots == self._cap:
                self._grow(self._cap * 2)
            slot = self._nslots
            self._nslots += 1
        ch._owner, ch._slot = self, slot
        # the channel's own state object is the authority until now (zeros for a new call)
        self._mh[:, slot] = ch.state.model_state[0].to(self.device)
        self._mc[:, slot] = ch.state.model_state[1].to(self.device)
        ch._fin = weakref.finalize(ch, SileroVADWorker._release, weakref.ref(self), slot)

    @staticmethod
    def _release(wref, slot):
        w = wref()
        if w is not None:
            w._free.append(slot)

    @property
    def channels_attached(self) -> int:
        return self._nslots - len(self._free)

    # -- batch processing ------------------------------------------------------------------
    @torch.no_grad()
    def process_batch(self, wis: List[Tuple[VADChannel, AudioChunk]]):
        dev = self.device
        while len(wis) > 0:
            later = []
            chans: List[VADChannel] = []
            chunks: List[AudioChunk] = []
            for wi in wis:          # a channel appears once per sub-batch (SileroVAD.py:70-77)
                if wi[0] not in chans:
                    chans.append(wi[0])
                    chunks.append(wi[1])
                else:
                    later.append(wi)
            wis = later
            for ch in chans:
                self._attach(ch)
            n = len(chans)
            slot = torch.tensor([ch._slot for ch in chans], dtype=torch.int32, device=dev)
            x = torch.stack([p.audio.to(dev, torch.float32) for p in chunks], dim=0)
            sl = slot.long()
            self._win.index_copy_(0, sl, x)
            stateful = _stateful(self.model)
            if stateful:                      # gather [2,B,64] x 2 by slot (VADBatchFromList.get_model_state, :21-22,99)
                _inject_state(self.model, self._mh.index_select(1, sl), self._mc.index_select(1, sl), self.input_sr, n)
            probs = self.model(x, self.input_sr).to(dev, torch.float32).contiguous()
            if stateful:                      # scatter back (save_model_state, :24-26,131)
                self._mh.index_copy_(1, sl, self.model._c._h.to(dev, torch.float32))
                self._mc.index_copy_(1, sl, self.model._c._c.to(dev, torch.float32))
                for ch in chans:              # the reference's per-channel attribute: views of the table rows
                    ch.state.model_state = [self._mh[:, ch._slot], self._mc[:, ch._slot]]
            ev = torch.empty((n, 8), dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().ifh_vad_step(
                    _lib.ptr(self._win), _lib.ptr(probs), _lib.ptr(slot), n, self.input_sr,
                    float(self.vad_iterator.threshold), _lib.ptr(self._st), _lib.ptr(self._blen), _lib.ptr(self._abuf),
                    _lib.ptr(ev), _lib.ptr(self._emit), _lib.stream_ptr(dev)), 'ifh_vad_step')
            ev_h = ev.cpu().tolist()                       # the one host sync per sub-batch
            st_h = self._st.index_select(0, sl).cpu().tolist()
            bl_h = self._blen.index_select(0, sl).cpu().tolist()
            for vc, p, e, s, bl in zip(chans, chunks, ev_h, st_h, bl_h):
                sd = vc.state
                sd.triggered, sd.temp_end, sd.current_sample = bool(s[0]), s[1], s[2]
                sd.speech = _speech_dict(e[0], e[1], self.input_sr)
                vc.active_start = None if s[3] < 0 else s[3]
                vc.buf_len = bl
                if e[6]:
                    raise AssertionError(f'VAD buffer invariant violated: {sd.speech=} {sd.current_sample=} '
                                         f'{vc.active_start=} {vc.buf_len=}')
                if e[3]:
                    audio = self._emit[vc._slot, :e[5]].clone()
                    vc.vad_chunk_in(VadAudioChunk(audio, self.input_sr, e[4]))
                vc.audio_chunk_in(p, vc.active_start is not None)
