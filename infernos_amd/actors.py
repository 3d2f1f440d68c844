"""Actor facades over the workers: session tables keyed by UUID.

Interface of Cluster/InfernSTTActor.py:12-53, Cluster/InfernTTSActor.py:12-52 and
Cluster/RemoteTTSSession.py:12-30.  The reference wraps these classes with @ray.remote; here they
are plain classes (Ray is not part of the speech path) and `as_ray_actor(cls)` applies the
reference's decorator arguments when ray is importable, so the app layer keeps calling
`actor.method.remote(...)`.
"""
from typing import Dict, Union
from uuid import UUID

from .stt import InfernSTTWorker, STTRequest, STTSentinel, STTSession
from .tts import InfernTTSWorker, TTSRequest, TTSSession


class InfernSessNotFoundErr(Exception):
    pass


class InfernSTTActor:
    debug = False
    sessions: Dict[UUID, STTSession]
    stt: InfernSTTWorker

    def __init__(self, **worker_kwa):
        self.sessions = {}
        self._worker_kwa = worker_kwa

    def start(self, device='cuda'):
        self.stt = InfernSTTWorker(device, **self._worker_kwa)      # no CPU fallback: raises without a HIP device
        self.stt.start()

    def stop(self):
        self.stt.stop()

    def new_stt_session(self, keep_context: bool = False):
        sess = STTSession(self.stt, keep_context)
        self.sessions[sess.id] = sess
        return sess.id

    def stt_session_end(self, sess_id):
        sess = self.sessions[sess_id]
        sess.stop()
        del self.sessions[sess_id]

    def stt_session_soundin(self, sess_id, req: Union[STTRequest, STTSentinel]):
        self.sessions[sess_id].soundin(req)


class InfernTTSActor:
    debug = False
    sessions: Dict[UUID, TTSSession]
    tts: InfernTTSWorker

    def __init__(self, **worker_kwa):
        self.sessions = {}
        self._worker_kwa = worker_kwa

    def start(self, lang: str = 'en', output_sr: int = 16000, device=None):
        self.tts = InfernTTSWorker(lang, output_sr, device, **self._worker_kwa)
        self.tts.start()
        self.tts_actr = getattr(self, '_self_handle', self)

    def stop(self):
        self.tts.stop()

    def get_rand_voice_id(self) -> int:
        return self.tts.get_rand_voice_id()

    def new_tts_session(self):
        rgen = TTSSession(self.tts, self.tts_actr)
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
    res = {'InfernSTTActor': {'stt': 1}, 'InfernTTSActor': {'tts': 1}}[cls.__name__]
    return ray.remote(num_gpus=0.25, resources=res)(cls)
