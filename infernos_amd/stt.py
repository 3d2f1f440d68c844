"""Speech-to-text plugin surface: STTRequest / STTSentinel / STTResult / STTSession and
InfernSTTWorker.

Interface of Cluster/STTSession.py:10-113 and Cluster/InfernSTTWorker.py:16-134.  The
session logic (one request in flight per session, merging of adjacent VAD chunks while
the span stays under stt.max_chunk_duration, sentinel echo rule) is host code as in the
reference; resampling and everything inside InfernSTTWorker.process_batch (log-mel,
Whisper encoder/decoder, greedy search) runs on the HIP device.
"""
from fractions import Fraction
from functools import partial
from threading import Lock
from time import monotonic
from typing import List, Union
from uuid import uuid4

from .audio import AudioChunk, VadAudioChunk


class STTRequest:
    """one utterance for the recogniser: audio chunk, language, where the STTResult goes; `mode` 'transcribe' or
    'translate', `timestamps`, and the no-speech probability above which the caller discards the text"""
    mode = 'transcribe'
    timestamps = False
    max_ns_prob = 0.5

    def __init__(self, chunk, text_cb, lang):
        self.chunk = chunk
        self.lang = lang
        self.text_cb = text_cb
        self.stime = monotonic()


class STTSentinel:
    """an in-band marker ('flush', ...): echoed to its callback in order with the results around it"""

    def __init__(self, signal, text_cb):
        self.signal = signal
        self.text_cb = text_cb
        self.stime = monotonic()


class STTResult:
    offsets = None

    def __init__(self, text, no_speech_prob, req):
        self.text, self.no_speech_prob = text, no_speech_prob
        self.duration = Fraction(len(req.chunk.audio), req.chunk.samplerate)      # exact seconds of the recognised audio
        self.inf_time = monotonic() - req.stime


class STTSession:
    debug = False
    lang = 'en'
    busy = False

    def __init__(self, stt, keep_context):
        self.id = uuid4()
        self.stt = stt
        self.pending = []                                   # STTRequest / STTSentinel in arrival order
        self.context = [] if keep_context else None         # token history handed to the worker
        self.state_lock = Lock()

    def stop(self):
        with self.state_lock:
            del self.stt, self.pending

    def soundin(self, req: Union[STTRequest, STTSentinel]):
        deliver = []
        with self.state_lock:
            self.pending.append(req)
            if self.busy:
                return
            assert len(self.pending) == 1
            self.busy = True
            self._drain_locked(deliver)
        for cb, r in deliver:          # callbacks run outside the lock (STTSession.py:77-78)
            cb(result=r)

    def _next_request(self):
        for r in self.pending:
            if isinstance(r, STTRequest):
                return r
        return None

    def _drain_locked(self, deliver: List):
        """Submit the next request to the worker, or echo sentinels (STTSession.py:80-102)."""
        while self.pending:
            head = self.pending.pop(0)
            if not isinstance(head, STTRequest):
                # a sentinel is echoed only when no other sentinel is queued behind it
                if all(isinstance(r, STTRequest) for r in self.pending):
                    deliver.append((head.text_cb, head))
                continue
            if isinstance(head.chunk, VadAudioChunk):
                nxt = self._next_request()
                if nxt is not None and isinstance(nxt.chunk, VadAudioChunk):
                    a, b = head.chunk, nxt.chunk
                    if b.tpos() + b.duration() - a.tpos() < self.stt.max_chunk_duration:
                        a.append(b)                 # merged-away request's callback is never called
                        self.pending.remove(nxt)
                        self.pending.insert(0, head)
                        continue
            if head.chunk.samplerate != self.stt.sample_rate:
                head.chunk.resample(self.stt.sample_rate)
            head.chunk.audio = self._to_worker_array(head.chunk.audio)
            self.stt.infer((head, partial(self.stt_out, head.text_cb), self.context))
            return
        self.busy = False

    def _to_worker_array(self, audio):
        """The reference converts to numpy here (STTSession.py:95); the HIP worker takes the
        device tensor as is, a worker that sets `wants_numpy` gets the numpy array."""
        if getattr(self.stt, 'wants_numpy', False) and hasattr(audio, 'numpy'):
            return audio.cpu().numpy()
        return audio

    def stt_out(self, text_cb, result: STTResult):
        deliver = [(text_cb, result)]
        with self.state_lock:
            if not hasattr(self, 'stt'):
                return
            assert self.busy
            self._drain_locked(deliver)
        for cb, r in deliver:
            cb(result=r)


# =========================================================================================
# InfernSTTWorker (Cluster/InfernSTTWorker.py:16-134) on the HIP device
# =========================================================================================
import torch  # noqa: E402

from . import _lib  # noqa: E402
from .workers import InfernBatchedWorker  # noqa: E402


class InfernSTTWorker(InfernBatchedWorker):
    """Whisper STT worker.  `infer((STTRequest, text_cb, context))`; results are delivered from
    the worker thread, in item order, exactly once per item (InfernSTTWorker.py:118-123).

    max_batch_size is the reference's tuning knob (4 there); the MI355X default is 64 because a
    batch is one set of large GEMMs here.  Constructor extras (all optional): `weights` (HF-format
    state dict; default: download `model_name`), `tokenizer` (needs convert_tokens_to_ids / decode;
    default: WhisperTokenizer.from_pretrained), `max_new_tokens`.

    beam_size selects the decode: 5 (default) is the beam width and length penalty the reference's default engine runs with --
    ctranslate2.models.Whisper.generate(features, prompts, return_no_speech_prob=True), that library's defaults
    (beam_size 5, length_penalty 1; InfernSTTWorker.py:61-75).  The SEARCH itself is transformers' formulation
    (GenerationMixin._beam_search: the K best finished hypotheses are kept and replaced, a row stops when its best running
    beam cannot beat its worst finished one), which is what the fixtures pin; ctranslate2's own termination rule (it is not
    in the image) may pick different tokens -- parity with it is UNPINNED.  Every request is decoded, no_speech_prob is reported;
    1 is its torch engine (infer_and_decode_torch, :77-107): greedy, and nothing is generated when every request of
    the batch is above its max_ns_prob (:91-92).  suppress_tokens / begin_suppress_tokens: token ids masked at every /
    at the first generated position (the model's generation config; ctranslate2 applies them by default)."""
    max_batch_size: int = 64
    max_chunk_duration: float = 32.0
    sample_rate: int = 16000
    debug = False

    def __init__(self, device: str, model_name: str = 'openai/whisper-large-v3', weights=None, tokenizer=None,
                 max_new_tokens: int = 224, fixed_new_tokens=None, beam_size: int = 5, length_penalty: float = 1.0,
                 suppress_tokens=None, begin_suppress_tokens=None):
        super().__init__()
        from .engines.whisper import Whisper
        from .features import WhisperLogMel
        self.device = dev = _lib.require_device(device)
        if weights is None:
            from transformers import WhisperForConditionalGeneration
            hf = WhisperForConditionalGeneration.from_pretrained(model_name)
            weights = hf.state_dict()
            # the checkpoint's generation config carries the token lists ctranslate2's converter bakes into its model
            # (suppress_ids / suppress_ids_begin) and applies by default (suppress_tokens=[-1], suppress_blank=True)
            gc = getattr(hf, 'generation_config', None)
            if suppress_tokens is None and gc is not None and getattr(gc, 'suppress_tokens', None):
                suppress_tokens = list(gc.suppress_tokens)
            if begin_suppress_tokens is None and gc is not None and getattr(gc, 'begin_suppress_tokens', None):
                begin_suppress_tokens = list(gc.begin_suppress_tokens)
            del hf
        if tokenizer is None:
            from transformers import WhisperTokenizer
            tokenizer = WhisperTokenizer.from_pretrained(model_name)
        self.tokenizer = tokenizer
        with torch.cuda.device(dev):
            self.model = Whisper(weights, dev)
            self.logmel = WhisperLogMel(self.model.n_mel, dev)
        self.no_speech_token_id = tokenizer.convert_tokens_to_ids('<|nospeech|>')
        self.eos_token_id = getattr(tokenizer, 'eos_token_id', None)
        self.max_new_tokens = max_new_tokens
        self.fixed_new_tokens = fixed_new_tokens
        self.beam_size, self.length_penalty = int(beam_size), float(length_penalty)
        self._suppress = self._begin_suppress = None
        V = self.model.vocab
        if self.beam_size > 1 and self.eos_token_id is None:
            raise ValueError('beam search needs the tokenizer\'s eos_token_id')
        if suppress_tokens or (fixed_new_tokens and self.beam_size > 1):
            self._suppress = torch.zeros(V, dtype=torch.float32)
            self._suppress[list(suppress_tokens or [])] = float('-inf')
            if fixed_new_tokens:                     # fixed-length workloads: the end token can never win
                self._suppress[self.eos_token_id] = float('-inf')
        if begin_suppress_tokens:
            self._begin_suppress = torch.zeros(V, dtype=torch.float32)
            self._begin_suppress[list(begin_suppress_tokens)] = float('-inf')
        self._prompt_cache = {}

    def get_prompt(self, options):
        """[<|startoftranscript|>, <|lang|>, <|mode|>, (<|notimestamps|>)] per request (:125-134)."""
        if options not in self._prompt_cache:
            self._prompt_cache[options] = tuple(
                self.tokenizer.convert_tokens_to_ids(['<|startoftranscript|>', f'<|{lang}|>', f'<|{mode}|>'] +
                                                     ([] if ts else ['<|notimestamps|>'])) for lang, mode, ts in options)
        return self._prompt_cache[options]

    def transcribe_batch(self, audios, prompts, max_nsps):
        """audios: list of 1-D float tensors/arrays @16 kHz -> [(text, no_speech_prob, token_ids)]"""
        dev = self.device
        B = len(audios)
        with torch.cuda.device(dev):
            lens = torch.tensor([min(len(a), 480000) for a in audios], dtype=torch.int32)
            L = max(int(lens.max()), 1)
            x = torch.zeros((B, L), dtype=torch.float32, device=dev)
            for i, a in enumerate(audios):
                a = torch.as_tensor(a)
                x[i, :lens[i]] = a[:lens[i]].to(dev, torch.float32)
            raw, wmax = self.logmel.raw(x, lens=lens.to(dev))        # normalisation fused into conv1's layout change
            enc = self.model.encode(raw=(self.logmel, raw, wmax))
            P = max(len(p) for p in prompts)
            assert all(len(p) == P for p in prompts), 'prompts of one batch must have equal length'
            pr = torch.tensor(prompts, dtype=torch.int32)
            n_new = self.fixed_new_tokens or self.max_new_tokens
            if self.beam_size > 1:
                toks, tlens, _, nsp = self.model.generate_beam(enc, pr, n_new, beams=self.beam_size, eos_id=self.eos_token_id,
                                                               length_penalty=self.length_penalty, suppress=self._suppress,
                                                               begin_suppress=self._begin_suppress,
                                                               no_speech_id=self.no_speech_token_id)
                nsp = nsp.cpu().tolist()
                toks, tlens = toks.cpu().tolist(), tlens.cpu().tolist()
                out = []
                for row, n, p in zip(toks, tlens, nsp):
                    row = row[:n]
                    if row and row[-1] == self.eos_token_id:
                        row = row[:-1]
                    out.append((self.tokenizer.decode(row, skip_special_tokens=True), p, row))
                return out
            toks, nsp, _ = self.model.generate(enc, pr, n_new, no_speech_id=self.no_speech_token_id,
                                               eos_id=None if self.fixed_new_tokens else self.eos_token_id,
                                               early_exit_nsp=max_nsps)
            nsp = nsp.cpu().tolist()
            if toks is None:            # every row above its max_ns_prob (InfernSTTWorker.py:91-92)
                return [('', p, []) for p in nsp]
            toks = toks.cpu().tolist()
        out = []
        for row, p in zip(toks, nsp):
            if self.eos_token_id is not None and not self.fixed_new_tokens and self.eos_token_id in row:
                row = row[:row.index(self.eos_token_id)]
            out.append((self.tokenizer.decode(row, skip_special_tokens=True), p, row))
        return out

    def process_batch(self, wis):
        assert all(wi[0].chunk.samplerate == self.sample_rate for wi in wis)
        audios = [wi[0].chunk.audio for wi in wis]
        prompts = self.get_prompt(tuple((wi[0].lang, wi[0].mode, wi[0].timestamps) for wi in wis))
        max_nsps = [wi[0].max_ns_prob for wi in wis]
        prompts = [list(p) for p in prompts]
        try:
            results = self.transcribe_batch(audios, prompts, max_nsps)
        except RuntimeError as e:
            # the reference's recovery (InfernSTTWorker.py:66-72): out of device memory on a batch -> release the
            # caching allocator's blocks and run the requests one by one; anything else, or a batch of one, re-raises
            if 'out of memory' not in str(e).lower() or len(wis) == 1:
                raise
            self.model._enc_bufs.clear()
            self.model._dec_bufs.clear()
            torch.cuda.empty_cache()
            results = []
            for a, pr, nsp in zip(audios, prompts, max_nsps):
                results.extend(self.transcribe_batch([a], [pr], [nsp]))
        for (req, text_cb, ctx), (text, nsp, toks) in zip(wis, results):
            if len(text) > 0 and text[0] == ' ':
                text = text[1:]
            if ctx is not None:
                ctx[:] = (ctx + toks)[:-224]
            text_cb(result=STTResult(text=text, no_speech_prob=nsp, req=req))
