"""Actor facades over the workers: session tables keyed by UUID.

Interface of Cluster/InfernSTTActor.py:12-53, Cluster/InfernTTSActor.py:12-52, Cluster/InfernLLMActor.py:11-69 and
Cluster/RemoteTTSSession.py:12-30.  The reference wraps these classes with @ray.remote; here they
are plain classes (Ray is not part of the speech path) and `as_ray_actor(cls)` applies the
reference's decorator arguments when ray is importable, so the app layer keeps calling
`actor.method.remote(...)`.
"""
from typing import Dict, Union
from uuid import UUID

from .llm import InfernLLMWorker, LLMInferRequest, LLMRequest, LLMSession, LLMSessionParams
from .shard import SessionRouter
from .stt import InfernSTTWorker, STTRequest, STTSentinel, STTSession
from .tts import InfernTTSWorker, TTSRequest, TTSSession


def _devices(device):
    """'cuda' / 'cuda:3' / None -> one device; a list or tuple -> those; 'cuda:*' -> every visible GPU.  With several
    devices an actor runs one worker (thread, stream, replicated weights) per GPU and pins each new session to the least
    loaded one (SessionRouter): the in-process form of the reference's one-actor-replica-per-GPU-share round robin
    (Cluster/InfernBenchActor.py:218-221)."""
    if isinstance(device, (list, tuple)):
        return list(device)
    if isinstance(device, str) and device.endswith(':*'):
        import torch
        return ['%s:%d' % (device[:-2], i) for i in range(torch.cuda.device_count())]
    return [device]


class InfernSessNotFoundErr(Exception):
    pass


class InfernSTTActor:
    debug = False
    sessions: Dict[UUID, STTSession]
    stt: InfernSTTWorker

    def __init__(self, **worker_kwa):
        self.sessions = {}
        self._worker_kwa = worker_kwa

    worker_cls = InfernSTTWorker

    def start(self, device='cuda'):
        # no CPU fallback: a worker raises without a HIP device
        self.workers = [self.worker_cls(dev, **self._worker_kwa) for dev in _devices(device)]
        self.router = SessionRouter(len(self.workers))
        self.stt = self.workers[0]
        for w in self.workers:
            w.start()

    def stop(self):
        for w in self.workers:
            w.stop()

    def _pool(self):
        if not hasattr(self, 'workers'):          # `.stt` set directly, the reference's attribute: a pool of one
            self.workers, self.router = [self.stt], SessionRouter(1)
        return self.workers, self.router

    def new_stt_session(self, keep_context: bool = False):
        workers, router = self._pool()
        sess = STTSession(None, keep_context)
        sess.stt = workers[router.assign(sess.id)]                  # sticky: the session's chunks all go to this GPU
        self.sessions[sess.id] = sess
        return sess.id

    def stt_session_end(self, sess_id):
        sess = self.sessions[sess_id]
        sess.stop()
        del self.sessions[sess_id]
        self._pool()[1].release(sess_id)

    def stt_session_soundin(self, sess_id, req: Union[STTRequest, STTSentinel]):
        self.sessions[sess_id].soundin(req)


class InfernTTSActor:
    debug = False
    sessions: Dict[UUID, TTSSession]
    tts: InfernTTSWorker

    def __init__(self, **worker_kwa):
        self.sessions = {}
        self._worker_kwa = worker_kwa

    worker_cls = InfernTTSWorker

    def start(self, lang: str = 'en', output_sr: int = 16000, device=None):
        self.workers = [self.worker_cls(lang, output_sr, dev, **self._worker_kwa) for dev in _devices(device)]
        self.router = SessionRouter(len(self.workers))
        self.tts = self.workers[0]
        for w in self.workers:
            w.start()
        self.tts_actr = getattr(self, '_self_handle', self)

    def stop(self):
        for w in self.workers:
            w.stop()

    def get_rand_voice_id(self) -> int:
        return self.tts.get_rand_voice_id()

    def _pool(self):
        if not hasattr(self, 'workers'):          # `.tts` set directly, the reference's attribute: a pool of one
            self.workers, self.router = [self.tts], SessionRouter(1)
        return self.workers, self.router

    def new_tts_session(self):
        workers, router = self._pool()
        rgen = TTSSession(self.tts, self.tts_actr)
        rgen.tts = workers[router.assign(rgen.id)]                  # sticky: KV caches / carry frames live on that GPU
        self.sessions[rgen.id] = rgen
        return rgen.id

    def tts_session_start(self, rgen_id, soundout: callable):
        self.sessions[rgen_id].start(soundout)

    def tts_session_say(self, rgen_id, req: TTSRequest):
        return self.sessions[rgen_id].say(req)

    def tts_session_stop_saying(self, rgen_id, rsay_id: UUID):
        return self.sessions[rgen_id].stop_saying(rsay_id)

    def tts_session_end(self, rgen_id):
        self.sessions[rgen_id].stop()
        del self.sessions[rgen_id]
        self._pool()[1].release(rgen_id)


class InfernLLMActor:
    """Session table over the LLM worker(s) (InfernLLMActor.py:11-69).  start() warms the worker up the way the
    reference does: max_batch_size 'What is your name?' requests pushed straight into the queue, answers awaited."""
    debug = False
    sessions: Dict[UUID, LLMSession]
    llm: InfernLLMWorker

    def __init__(self, **worker_kwa):
        self.sessions = {}
        self._worker_kwa = worker_kwa

    worker_cls = InfernLLMWorker

    def start(self, device='cuda', warmup=True):
        from queue import Queue
        self.workers = [self.worker_cls(dev, **self._worker_kwa) for dev in _devices(device)]      # no CPU fallback
        self.router = SessionRouter(len(self.workers))
        self.llm = self.workers[0]
        for w in self.workers:
            w.start()
        if not warmup:
            return
        for w in self.workers:
            tq = Queue()
            irs = tuple(LLMInferRequest(LLMRequest('What is your name?', None), [{"role": "user", "content": 'What is your name?'}])
                        for _ in range(w.max_batch_size))
            for _i in irs:
                _i.textout_cb = lambda result: tq.put(result)
            for ir in irs:
                w.infer(ir)
            for _ in irs:
                tq.get()

    def stop(self):
        for w in self.workers:
            w.stop()

    def _pool(self):
        if not hasattr(self, 'workers'):
            self.workers, self.router = [self.llm], SessionRouter(1)
        return self.workers, self.router

    def new_llm_session(self, sconf: LLMSessionParams):
        workers, router = self._pool()
        sess = LLMSession(None, sconf)
        sess.llm = workers[router.assign(sess.id)]
        self.sessions[sess.id] = sess
        return sess.id

    def llm_session_end(self, sess_id):
        sess = self.sessions[sess_id]
        sess.stop()
        del self.sessions[sess_id]
        self._pool()[1].release(sess_id)

    def llm_session_textin(self, sess_id, req: LLMRequest):
        self.sessions[sess_id].textin(req)
        return sess_id

    def llm_session_context_add(self, sess_id, content: str, role: str = 'user'):
        self.sessions[sess_id].context_add(content, role)
        return sess_id


class RemoteTTSSession:
    """Client-side handle (RemoteTTSSession.py:12-30): works with a Ray actor handle
    (`.method.remote(...)` + ray.get) or with the plain actor object."""

    def __init__(self, tts_actr):
        self.tts_actr = tts_actr
        self.sess_id = self._call('new_tts_session')

    def _call(self, name, *a, **k):
        m = getattr(self.tts_actr, name)
        if hasattr(m, 'remote'):
            import ray
            return ray.get(m.remote(*a, **k))
        return m(*a, **k)

    def start(self, soundout: callable):
        return self._call('tts_session_start', self.sess_id, soundout)

    def say(self, req: TTSRequest):
        return self._call('tts_session_say', rgen_id=self.sess_id, req=req)

    def stop_saying(self, rsay_id: UUID):
        return self._call('tts_session_stop_saying', self.sess_id, rsay_id)

    def end(self):
        return self._call('tts_session_end', self.sess_id)


def as_ray_actor(cls):
    """@ray.remote(num_gpus=0.25, resources={...}) like the reference, if ray is installed."""
    import ray
    res = {'InfernSTTActor': {'stt': 1}, 'InfernTTSActor': {'tts': 1}, 'InfernLLMActor': {'llm': 1}}[cls.__name__]
    return ray.remote(num_gpus=1.0 if cls.__name__ == 'InfernLLMActor' else 0.25, resources=res)(cls)
