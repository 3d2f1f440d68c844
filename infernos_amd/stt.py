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
from typing import List, Optional, Union
from uuid import UUID, uuid4

from .audio import AudioChunk, VadAudioChunk


class STTRequest:
    lang: str
    chunk: AudioChunk
    text_cb: callable
    mode: str = 'transcribe'
    timestamps: bool = False
    stime: float
    max_ns_prob: float = 0.5

    def __init__(self, chunk: AudioChunk, text_cb: callable, lang: str):
        self.stime = monotonic()
        self.lang, self.chunk, self.text_cb = lang, chunk, text_cb


class STTSentinel:
    stime: float
    text_cb: callable

    def __init__(self, signal: str, text_cb: callable):
        self.stime = monotonic()
        self.signal, self.text_cb = signal, text_cb


class STTResult:
    text: str
    no_speech_prob: float
    duration: Fraction
    offsets: Optional[List] = None
    inf_time: float

    def __init__(self, text: str, no_speech_prob: float, req: STTRequest):
        self.text = text
        self.no_speech_prob = no_speech_prob
        self.duration = Fraction(len(req.chunk.audio), req.chunk.samplerate)
        self.inf_time = monotonic() - req.stime


class STTSession:
    debug = False
    id: UUID
    lang: str = 'en'
    context: Optional[List[int]]
    state_lock: Lock
    busy: bool = False
    pending: List[Union[STTRequest, STTSentinel]]

    def __init__(self, stt, keep_context: bool):
        self.id = uuid4()
        self.stt = stt
        self.state_lock = Lock()
        self.context = [] if keep_context else None
        self.pending = []

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
