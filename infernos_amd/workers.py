"""Worker-thread base classes.

Interface of Core/InfernWrkThread.py:32-69 (three-state daemon thread) and
Cluster/InfernBatchedWorker.py:7-45 (queue -> batches of at most max_batch_size).
Batch formation policy is the reference's: block for the first item, then drain without
blocking; a None item stops the worker.
"""
from abc import ABC, abstractmethod
from queue import Empty, Queue
from threading import Lock, Thread
from typing import List, Optional

RTPWrkTInit = 0
RTPWrkTRun = 1
RTPWrkTStop = 2


class InfernWrkThread(Thread):
    state_lock: Lock = None
    state: int = RTPWrkTInit

    def __init__(self):
        self.state_lock = Lock()
        super().__init__(daemon=True)

    def get_state(self, locked=False):
        if locked:
            return self.state
        with self.state_lock:
            return self.state

    def _set_state(self, newstate, expected_state=None, raise_on_error=True):
        with self.state_lock:
            prev = self.state
            if expected_state is not None and prev != expected_state:
                if raise_on_error:
                    raise AssertionError(f'Unexpected state: {prev}, {expected_state} expected')
                return prev
            self.state = newstate
            return prev

    def thread_started(self):
        self._set_state(RTPWrkTRun, expected_state=RTPWrkTInit)

    def stop(self):
        prev = self._set_state(RTPWrkTStop, expected_state=RTPWrkTRun, raise_on_error=True)
        if prev == RTPWrkTRun:
            self.join()
        self._set_state(RTPWrkTInit, expected_state=RTPWrkTStop)


class InfernBatchedWorker(InfernWrkThread, ABC):
    max_batch_size: int
    inf_queue: 'Queue[Optional[object]]'

    def __init__(self):
        super().__init__()
        self.inf_queue = Queue()

    def infer(self, wi: object):
        self.inf_queue.put(wi)

    def next_batch(self) -> Optional[List[object]]:
        batch: List[object] = []
        while len(batch) < self.max_batch_size:
            try:
                wi = self.inf_queue.get(block=not batch)
            except Empty:
                break
            if wi is None:
                return None
            batch.append(wi)
        return batch

    @abstractmethod
    def process_batch(self, wis: List[object]):
        ...

    def run(self):
        self.thread_started()
        while self.get_state() == RTPWrkTRun:
            wis = self.next_batch()
            if wis is None:
                break
            for wi in wis:
                cb = getattr(wi, '_proc_start_cb', None)
                if cb is not None:
                    cb()
            self.process_batch(wis)

    def stop(self):
        self.inf_queue.put(None)
        super().stop()
