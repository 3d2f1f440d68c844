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


def test_session_router_least_loaded_and_sticky():
    from infernos_amd.shard import SessionRouter
    r = SessionRouter(3)
    got = [r.assign('s%d' % i) for i in range(7)]
    assert got == [0, 1, 2, 0, 1, 2, 0] and r.load == [3, 2, 2]
    assert r.assign('s3') == 0 and r.load == [3, 2, 2]                 # sticky: asking again changes nothing
    r.release('s0'); r.release('s6')                                   # shard 0 drops to 1 live session
    assert r.load == [1, 2, 2] and r.assign('n0') == 0 and r.assign('n1') == 0 and r.assign('n2') == 1
    assert r.shard_of('s4') == 1 and r.rows_by_shard()[2] == ['s2', 's5']
    with pytest.raises(KeyError):
        r.release('unknown')


def test_actors_pin_sessions_to_least_loaded_device():
    """InfernSTTActor / InfernTTSActor with several devices: one worker per device, every new session pinned to the least
    loaded one and all of its work delivered there (fake workers: no GPU in this test)."""
    from infernos_amd import actors
    from infernos_amd.stt import STTRequest
    from infernos_amd.audio import AudioChunk

    class FakeSTT:
        max_chunk_duration, sample_rate = 32.0, 16000

        def __init__(self, device, **kw):
            self.device, self.items, self.running = device, [], False

        def start(self): self.running = True
        def stop(self): self.running = False
        def infer(self, wi): self.items.append(wi)

    class STTActor(actors.InfernSTTActor):
        worker_cls = FakeSTT
    a = STTActor()
    a.start(device=['cuda:0', 'cuda:1', 'cuda:2'])
    assert [w.device for w in a.workers] == ['cuda:0', 'cuda:1', 'cuda:2'] and all(w.running for w in a.workers)
    sids = [a.new_stt_session() for _ in range(5)]
    assert [a.router.shard_of(s) for s in sids] == [0, 1, 2, 0, 1]
    for k, sid in enumerate(sids):
        for _ in range(2):                           # the second request waits behind the first (one in flight per session)
            a.stt_session_soundin(sid, STTRequest(AudioChunk(torch.zeros(1600), 16000), lambda result: None, 'en'))
    assert [len(w.items) for w in a.workers] == [2, 2, 1]
    a.stt_session_end(sids[0]); a.stt_session_end(sids[3])
    assert a.router.load == [0, 2, 1] and a.router.shard_of(a.new_stt_session()) == 0
    a.stop()
    assert not any(w.running for w in a.workers)

    class FakeTTS(FakeSTT):
        def __init__(self, lang, output_sr, device, **kw):
            super().__init__(device)
            self.output_sr = output_sr

        def get_rand_voice(self): return torch.zeros(1, 512), 0
        def get_voice(self, i): return torch.zeros(1, 512)

    class TTSActor(actors.InfernTTSActor):
        worker_cls = FakeTTS
    t = TTSActor()
    t.start(output_sr=8000, device=['cuda:0', 'cuda:1'])
    ids = [t.new_tts_session() for _ in range(3)]
    for i in ids:
        t.tts_session_start(i, lambda chunk: None)
        t.tts_session_say(i, actors.TTSRequest('hello there'))
    assert [len(w.items) for w in t.workers] == [2, 1]
    t.tts_session_end(ids[0])
    assert t.router.load == [1, 1]
    t.stop()

    # configuration 5's LLM stage: the same routing (one InfernLLMWorker per device, sessions sticky to the least loaded)
    from infernos_amd.llm import LLMRequest, LLMSessionParams

    class FakeLLM(FakeSTT):
        max_batch_size = 4

    class LLMActor(actors.InfernLLMActor):
        worker_cls = FakeLLM
    m = LLMActor()
    m.start(device=['cuda:0', 'cuda:1', 'cuda:2', 'cuda:3'], warmup=False)
    lids = [m.new_llm_session(LLMSessionParams('system %d' % i)) for i in range(6)]
    assert [m.router.shard_of(i) for i in lids] == [0, 1, 2, 3, 0, 1]
    for i in lids:
        m.llm_session_textin(i, LLMRequest('hello', lambda result: None))
        m.llm_session_context_add(i, 'more', 'user')
    assert [len(w.items) for w in m.workers] == [2, 2, 1, 1]
    # the queued request holds a shallow snapshot (tuple of the same dicts), as Cluster/LLMSession.py:31 does: text added to
    # the last message before the worker picks the request up is seen by it
    assert m.workers[0].items[0].context == ({'role': 'system', 'content': 'system 0'}, {'role': 'user', 'content': 'hello more'})
    assert m.sessions[lids[0]].context[-1] == {'role': 'user', 'content': 'hello more'}
    m.llm_session_end(lids[1])
    assert m.router.load == [2, 1, 1, 1]
    m.stop()


def _pipelined_worker(rank, world, port, n_total, nsteps, group, lanes, q):
    """SpeechPipeline.run_steps' schedule (pipeline.schedule_cycles) over two gloo communicators: the ingress scatter is issued
    by fetch() and the egress gather by retire(), both on the main thread; front-end and synthesis jobs run on pool threads
    with rank-dependent delays, so ranks drift apart as they do on real hardware."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from infernos_amd.pipeline import schedule_cycles
    from infernos_amd.shard import gather_rows, scatter_frames
    g_in, g_out = dist.new_group(), dist.new_group()
    T = 3
    gen = torch.Generator().manual_seed(5)
    cycles = [torch.randint(0, 256, (T, n_total, 160), dtype=torch.uint8, generator=gen) for _ in range(nsteps)]
    b = shard_bounds(n_total, world)[rank]
    fpool, tpool = ThreadPoolExecutor(2), ThreadPoolExecutor(lanes)
    ok, gathered, ttss = [True], [], {}

    def fetch(k):
        mine = scatter_frames(cycles[k] if rank == 0 else None, n_total, T, 'cpu', group=g_in)
        ok[0] = ok[0] and torch.equal(mine, cycles[k][:, b.start:b.stop])
        return k, mine

    def front_job(gi, frs):
        time.sleep(0.002 * ((rank * 7 + gi * 3) % 5))
        return [(k, fr.sum(dim=0).to(torch.int32)) for k, fr in frs]

    def tts_job(front_fut):
        rs = front_fut.result()
        time.sleep(0.003 * ((rank * 5 + rs[0][0]) % 4))
        return rs

    def retire(gi):
        for k, rows in ttss.pop(gi).result():
            full = gather_rows(rows, n_total, group=g_out)
            if rank == 0:
                gathered.append((k, full))
        return gi
    schedule_cycles(nsteps, group, lanes, fetch, lambda gi, frs: fpool.submit(front_job, gi, frs),
                    lambda gi, fut: ttss.__setitem__(gi, tpool.submit(tts_job, fut)), retire)
    if rank == 0:
        ok[0] = ok[0] and [k for k, _ in gathered] == list(range(nsteps))
        for k, full in gathered:
            ok[0] = ok[0] and torch.equal(full, cycles[k].sum(dim=0).to(torch.int32))
    q.put((rank, bool(ok[0])))
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_total,nsteps,group,lanes', [
    (2, 6, 9, 1, 3), (3, 7, 10, 2, 2), (4, 8, 7, 1, 4),
    (8, 2051, 4, 1, 3),       # BASELINE config 4's world: 8 ranks, 2 048 calls + 3 (ragged shards of 257 / 256 rows)
])
def test_pipelined_collective_order_gloo(world, n_total, nsteps, group, lanes):
    """The N > 1 bench path: every rank issues the scatters of the ingress communicator and the gathers of the egress
    communicator in the same order although its stage threads run at their own pace (ragged shards included) --
    no deadlock, every cycle's rows arrive intact and in cycle order."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_pipelined_worker, args=(r, world, port, n_total, nsteps, group, lanes, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(r, True) for r in range(world)]
