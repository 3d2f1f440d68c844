"""CPU: the N>1 ingress scatter / egress gather with gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from infernos_amd.shard import shard_bounds


def test_shard_bounds():
    assert [list(r) for r in shard_bounds(5, 2)] == [[0, 1, 2], [3, 4]]
    assert [len(r) for r in shard_bounds(2048, 8)] == [256] * 8
    assert sum(len(r) for r in shard_bounds(7, 3)) == 7


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infernos_amd.shard import gather_rows, scatter_frames
    T = 5
    g = torch.Generator().manual_seed(0)
    full = torch.randint(0, 256, (T, n_total, 160), dtype=torch.uint8, generator=g)
    g_in, g_out = dist.new_group(), dist.new_group()
    mine = scatter_frames(full if rank == 0 else None, n_total, T, 'cpu', group=g_in)
    b = shard_bounds(n_total, world)[rank]
    ok = torch.equal(mine, full[:, b.start:b.stop])
    local = mine.sum(dim=0).to(torch.int32)                 # stand-in for the encoded output rows
    out = gather_rows(local, n_total, group=g_out)
    if rank == 0:
        ok = ok and torch.equal(out, full.sum(dim=0).to(torch.int32))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [6, 7])
def test_scatter_gather_gloo(n_total):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True), (1, True)]
