"""Process-wide device-work mutex with a deadlock timeout and a busy/idle load estimate.

Interface of safetorch/InfernTorcher.py:20-66 and the InfernGlobals singleton that owns it
(config/InfernGlobals.py:10-21).  On the HIP path kernel ordering comes from the stream, so
the lock only serialises host-side state mutation of one engine between the worker thread
(infer) and dispatch; it is kept because callers use `with InfernGlobals().torcher:`.
"""
from functools import lru_cache
import math
from threading import Lock
from time import monotonic


class InfernTorcherDeadlock(Exception):
    pass


class rc_filter:
    """First-order low-pass `y += alpha * (x - y)`, alpha = 1 / (1 + 2*pi*x0) (safetorch/InfernTorcher.py:8-18:
    constructor `(x=10, init_y=0.0)`, applied by calling the object, state in `last_y`)."""

    def __init__(self, x=10, init_y=0.0):
        self.alpha = 1.0 / (1.0 + 2.0 * math.pi * x)
        self.last_y = init_y

    def __call__(self, x):
        self.last_y += self.alpha * (x - self.last_y)
        return self.last_y


class InfernTorcher:
    """lock(timeout=10) / unlock(), acquire() / release(), context manager returning the torcher
    (safetorch/InfernTorcher.py:34-66).  Every 100th unlock reports busy / (busy + free) like the reference does
    (`quiet = True` silences it)."""
    quiet = False

    def __init__(self):
        self._torch_lock = Lock()
        self._t_lock = self._t_unlock = monotonic()
        self._free_time, self._busy_time = rc_filter(), rc_filter()
        self._nlocks = 0

    def lock(self, timeout: int = 10):
        if not self._torch_lock.acquire(timeout=timeout):
            raise InfernTorcherDeadlock(f'Could not acquire lock within {timeout} seconds')
        self._t_lock = monotonic()
        self._free_time(self._t_lock - self._t_unlock)

    def unlock(self):
        self._t_unlock = monotonic()
        self._busy_time(self._t_unlock - self._t_lock)
        self._nlocks += 1
        report = not self.quiet and self._nlocks % 100 == 0
        load = self.load()
        self._torch_lock.release()
        if report:
            print(f'Torch load: {load}')

    def load(self):
        bt, ft = self._busy_time.last_y, self._free_time.last_y
        return bt / (bt + ft) if bt + ft > 0 else 0.0

    acquire = lock
    release = unlock

    @property
    def nlocks(self):
        return self._nlocks

    def __enter__(self):
        self.lock()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.unlock()


class InfernGlobals:
    _instance = None
    _ilock = Lock()
    torcher: InfernTorcher = None

    def __new__(cls):
        with cls._ilock:
            if cls._instance is None:
                cls._instance = super().__new__(cls)
                cls.torcher = InfernTorcher()
        return cls._instance

    @staticmethod
    def get_resampler(from_sr: int, to_sr: int, device='cuda'):
        from .audio import get_resampler
        return get_resampler(from_sr, to_sr, str(device))

    @staticmethod
    @lru_cache(maxsize=8)
    def get_translator(from_lang: str, to_lang: str, **kwa):
        """config/InfernGlobals.py:28-31: one cached Translator per (from, to, options)"""
        from .t2t import Translator
        return Translator(from_lang, to_lang, **kwa)

    @staticmethod
    def stdtss():
        return f'{monotonic():4.3f}'
