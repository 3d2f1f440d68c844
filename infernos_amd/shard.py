"""Multi-GPU sharding of concurrent calls: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" for the CPU tests), sticky call -> rank assignment,
models replicated, no collective on the model path.

The only exchange step is the ingress/egress of the batch: the ingress rank holds the
mu-law frame matrix of every call and scatters each rank's rows; ranks return their encoded
output rows by gather.  Messages are small (160 B per call per tick), so whole utterance
blocks are moved in one collective each way rather than one per tick.  Ingress and egress may be issued from different host
threads (the serving loop pipelines its stages): give each its own process group (`group=`) so that
the per-communicator issue order is the same on every rank.
"""
import threading
from typing import Hashable, List, Optional

import torch
import torch.distributed as dist


class SessionRouter:
    """Sticky session -> shard assignment, least loaded at session start: per-session device state (VAD
    model state, resampler carry, pending STT chunk, TTS KV cache and carry frames) must stay where it was created, so a
    session keeps its shard until it ends.  A shard is a GPU of this process (the actor facades, one worker thread per
    device) or a rank of the torch.distributed job (the batched pipeline).  The reference's counterpart is the round-robin
    over Ray actor replicas (Cluster/InfernBenchActor.py:218-221, InfernSTTActor.py:12)."""

    def __init__(self, n_shards: int):
        assert n_shards >= 1
        self.load = [0] * n_shards
        self._owner = {}
        self._lock = threading.Lock()

    def assign(self, session_id: Hashable) -> int:
        with self._lock:
            if session_id in self._owner:
                return self._owner[session_id]
            k = min(range(len(self.load)), key=lambda i: (self.load[i], i))
            self.load[k] += 1
            self._owner[session_id] = k
            return k

    def shard_of(self, session_id: Hashable) -> int:
        with self._lock:
            return self._owner[session_id]

    def release(self, session_id: Hashable) -> None:
        with self._lock:
            k = self._owner.pop(session_id)
            self.load[k] -= 1

    def rows_by_shard(self) -> List[list]:
        """live sessions grouped by shard, in assignment order: the row order of the ingress scatter"""
        with self._lock:
            out = [[] for _ in self.load]
            for sid, k in self._owner.items():
                out[k].append(sid)
            return out


def _stage(group):
    """gloo has no device scatter/gather: stage through host memory (CPU tests, single-GPU dry runs)."""
    return dist.get_backend(group) == 'gloo'


def shard_bounds(n_total: int, world: int) -> List[range]:
    """Contiguous, balanced call ranges per rank (first n_total % world ranks get one more)."""
    q, r = divmod(n_total, world)
    out, lo = [], 0
    for k in range(world):
        hi = lo + q + (1 if k < r else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def scatter_frames(frames_all: Optional[torch.Tensor], n_total: int, t: int, device, src: int = 0, group=None,
                   always_collective: bool = False) -> torch.Tensor:
    """frames_all u8 [T, n_total, 160] on rank `src` (None elsewhere) -> this rank's [T, n_local, 160].
    always_collective: issue the collective even in a world of one (lets the RCCL call path run on a single-GPU box)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    bounds = shard_bounds(n_total, world)
    mine = torch.empty((t, len(bounds[rank]), 160), dtype=torch.uint8, device=device)
    if world == 1 and not always_collective:
        mine.copy_(frames_all)
        return mine
    if _stage(group) and torch.device(device).type != 'cpu':
        host = scatter_frames(None if frames_all is None else frames_all.cpu(), n_total, t, 'cpu', src=src, group=group)
        mine.copy_(host)
        return mine
    if max(len(b) for b in bounds) == min(len(b) for b in bounds):
        parts = None
        if rank == src:
            parts = [frames_all[:, b.start:b.stop].contiguous() for b in bounds]
        dist.scatter(mine, parts, src=src, group=group)
    else:                                   # ragged: point-to-point
        if rank == src:
            reqs = []
            for k, b in enumerate(bounds):
                part = frames_all[:, b.start:b.stop].contiguous()
                if k == src:
                    mine.copy_(part)
                else:
                    reqs.append(dist.isend(part, dst=k, group=group))
            for r in reqs:
                r.wait()
        else:
            dist.recv(mine, src=src, group=group)
    return mine


def gather_rows(local: torch.Tensor, n_total: int, dst: int = 0, group=None, always_collective: bool = False) -> Optional[torch.Tensor]:
    """local [n_local, W] -> [n_total, W] on rank dst (None elsewhere)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1 and not always_collective:
        return local
    if _stage(group) and local.device.type != 'cpu':
        full = gather_rows(local.cpu(), n_total, dst=dst, group=group)
        return None if full is None else full.to(local.device)
    bounds = shard_bounds(n_total, world)
    W = local.size(1)
    if max(len(b) for b in bounds) == min(len(b) for b in bounds):
        outs = [torch.empty((len(b), W), dtype=local.dtype, device=local.device) for b in bounds] if rank == dst else None
        dist.gather(local.contiguous(), outs, dst=dst, group=group)
        return torch.cat(outs) if rank == dst else None
    if rank == dst:
        full = torch.empty((n_total, W), dtype=local.dtype, device=local.device)
        for k, b in enumerate(bounds):
            if k == dst:
                full[b.start:b.stop] = local
            else:
                dist.recv(full[b.start:b.stop], src=k, group=group)
        return full
    dist.send(local.contiguous(), dst=dst, group=group)
    return None
