"""Process-wide device-work mutex with a deadlock timeout and a busy/idle load estimate.

Interface of safetorch/InfernTorcher.py:20-66 and the InfernGlobals singleton that owns it
(config/InfernGlobals.py:10-21).  On the HIP path kernel ordering comes from the stream, so
the lock only serialises host-side state mutation of one engine between the worker thread
(infer) and dispatch; it is kept because callers use `with InfernGlobals().torcher:`.
"""
from threading import Lock
from time import monotonic


class InfernTorcherDeadlock(Exception):
    pass


class rc_filter:
    def __init__(self, fcoef, initval=0.0):
        self.fcoef, self.val = fcoef, initval

    def apply(self, x):
        self.val = self.fcoef * x + (1.0 - self.fcoef) * self.val
        return self.val


class InfernTorcher:
    timeout = 10.0
    report_every = 100

    def __init__(self):
        self._lock = Lock()
        self._last = monotonic()
        self._busy = rc_filter(0.1)
        self._idle = rc_filter(0.1)
        self.nlocks = 0
        self.verbose = False

    def acquire(self):
        if not self._lock.acquire(timeout=self.timeout):
            raise InfernTorcherDeadlock('device lock not released for %.0f s' % self.timeout)
        now = monotonic()
        self._idle.apply(now - self._last)
        self._last = now

    def release(self):
        now = monotonic()
        self._busy.apply(now - self._last)
        self._last = now
        self.nlocks += 1
        if self.verbose and self.nlocks % self.report_every == 0:
            print('Torch load: %.3f' % self.load())
        self._lock.release()

    def load(self):
        tot = self._busy.val + self._idle.val
        return self._busy.val / tot if tot > 0 else 0.0

    __enter__ = lambda self: self.acquire()

    def __exit__(self, *a):
        self.release()


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
    def stdtss():
        return f'{monotonic():4.3f}'
