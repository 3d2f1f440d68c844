"""Audio-stream markers and the per-call output muxer.

Interface of Core/AStreamMarkers.py:7-30 and Core/OutputMuxer.py:10-85 (per-track FIFO
re-blocking into qsize-sample pieces, marker firing, multi-track mix = zero-pad, sum, divide by
the number of tracks).  Host bookkeeping over device (or CPU) tensors; the mix of the
already-blocked pieces is a handful of tensor adds per 100 ms and stays in torch.
"""
from time import monotonic
from typing import Dict, List, Union

import torch
import torch.nn.functional as F

from .audio import AudioChunk


class ASMarkerGeneric:
    track_id: int
    debug: bool = False

    def __init__(self, track_id: int = 0):
        self.track_id = track_id


class ASMarkerNewSent(ASMarkerGeneric):
    def on_proc(self, tro_self, *args):      # runs on the RTP output worker thread
        pass


class ASMarkerSentDoneCB(ASMarkerNewSent):
    def __init__(self, done_cb: callable, sync: bool = False, **kwargs):
        super().__init__(**kwargs)
        self.done_cb, self.sync = done_cb, sync

    def on_proc(self, tro_self):
        x = self.done_cb()
        if self.sync and hasattr(x, 'get'):
            x.get()


class OutputMuxer:
    def __init__(self, output_sr: int, qsize: int, device: str):
        self.output_sr, self.qsize, self.device = output_sr, qsize, device
        self.chunks_in: List[Union[AudioChunk, ASMarkerGeneric]] = []

    def chunk_in(self, chunk):
        if isinstance(chunk, AudioChunk):
            if chunk.samplerate != self.output_sr:
                chunk = chunk.resample(self.output_sr)
            if self.chunks_in and isinstance(self.chunks_in[-1], AudioChunk):
                chunk.audio = torch.cat((self.chunks_in.pop().audio, chunk.audio), dim=0)
        self.chunks_in.append(chunk)

    def idle(self, rtp_worker):
        q = self.chunks_in
        if len(q) == 1 and isinstance(q[0], AudioChunk) and q[0].audio.size(0) < self.qsize:
            return None
        out = None
        have = 0
        while q and have < self.qsize:
            head = q[0]
            if isinstance(head, ASMarkerNewSent):
                if have > 0:
                    return out
                q.pop(0)
                head.on_proc(rtp_worker)
                continue
            need = self.qsize - have
            piece = head.audio[:need]
            out = piece if out is None else torch.cat((out, piece), dim=0)
            have = out.size(0)
            if head.audio.size(0) > need:
                head.audio = head.audio[need:]
            else:
                q.pop(0)
        if 0 < have < self.qsize:
            q.insert(0, AudioChunk(out, self.output_sr))
            return None
        return out if have > 0 else None


class OutputMTMuxer:
    def __init__(self, output_sr: int, qsize: int, device: str):
        self.tracks: Dict[int, OutputMuxer] = {}
        self.output_sr, self.qsize, self.device = output_sr, qsize, device

    def chunk_in(self, chunk):
        if chunk.track_id not in self.tracks:
            self.tracks[chunk.track_id] = OutputMuxer(self.output_sr, self.qsize, self.device)
        self.tracks[chunk.track_id].chunk_in(chunk)

    def idle(self, rtp_worker):
        chunks = [c for c in (t.idle(rtp_worker) for t in self.tracks.values()) if c is not None]
        if not chunks:
            return None
        if len(chunks) == 1:
            return chunks[0]
        n = max(c.size(0) for c in chunks)
        chunks = [F.pad(c, (0, n - c.size(0))) if c.size(0) < n else c for c in chunks]
        return torch.sum(torch.stack(chunks), dim=0) / len(self.tracks)
