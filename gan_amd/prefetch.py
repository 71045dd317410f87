"""Host input pipeline that runs ahead of the GPU.

The reference prepares every batch inline, between two session runs (dcgan.py:210-221 decodes and
crops 64 image files, text2image.py:324-341 and cyclegan.py:143-151 likewise), which is fine beside a
step of hundreds of milliseconds and not beside one of a few.  Here a producer thread prepares batches
`depth` ahead of the training loop, in order, and hands them over as page-locked tensors so that the
copy to the GPU is asynchronous; the files of one batch are decoded on a small pool of worker threads
(image decoding releases the interpreter lock).

Order is part of the contract: batch i is made by make(i), calls to make() happen one after the other
in index order on the one producer thread (so a random generator used inside it yields the sequence
the inline loop would have drawn), and `pool_map` keeps the order of its items.
"""
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

_END = object()

# Page-locking host memory is an allocation call, and while a stream capture is under way an allocation
# call on ANY thread invalidates it.  Graph capture holds this lock; the producer takes it to pin a batch.
capture_lock = threading.RLock()


class _Failure(object):
    def __init__(self, exc):
        self.exc = exc


def _to_tensor(a, pin):
    if isinstance(a, np.ndarray) and a.dtype != np.object_ and a.dtype.kind in 'fiub':
        t = torch.from_numpy(np.ascontiguousarray(a))
        if not pin:
            return t
        with capture_lock:
            return t.pin_memory()
    if isinstance(a, (tuple, list)):
        return type(a)(_to_tensor(v, pin) for v in a)
    return a


class BatchPrefetcher(object):
    """Iterates over (index, batch) for index in `indices`, with batch = make(index) prepared up to
    `depth` batches ahead on a background thread.

    make(index) returns a numpy array or a tuple / list of them (other members, e.g. file names, pass
    through untouched); arrays arrive as CPU tensors, page-locked when `pin` (default: when a GPU is
    present).  An exception raised by make() is re-raised in the consuming thread at the position of
    its batch.  Leaving the loop early, or close(), stops the producer; the object is also a context
    manager."""

    def __init__(self, make, indices, depth=4, workers=4, pin=None):
        if depth < 1:
            raise ValueError("BatchPrefetcher: depth must be at least 1")
        self._make = make
        self._indices = list(indices)
        self._pin = torch.cuda.is_available() if pin is None else bool(pin)
        self._queue = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self._thread = threading.Thread(target=self._produce, name="gan-batch-prefetch", daemon=True)
        self._started = False

    # ---------------------------------------------------------------- producer side
    def pool_map(self, fn, items):
        """[fn(item) for item in items], evaluated on the worker pool, order kept.  For use inside make()."""
        return list(self._pool.map(fn, items))

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._queue.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _produce(self):
        try:
            for i in self._indices:
                if self._stop.is_set():
                    return
                if not self._put((i, _to_tensor(self._make(i), self._pin))):
                    return
            self._put(_END)
        except BaseException as exc:            # handed to the consumer, which re-raises it
            self._put(_Failure(exc))

    # ---------------------------------------------------------------- consumer side
    def __iter__(self):
        if self._started:
            raise RuntimeError("BatchPrefetcher: one pass only")
        self._started = True
        self._thread.start()                     # not before: make() may refer to this object (pool_map)
        try:
            while True:
                item = self._queue.get()
                if item is _END:
                    return
                if isinstance(item, _Failure):
                    raise item.exc
                yield item
        finally:
            self.close()

    def close(self):
        self._stop.set()
        while True:                              # unblock a producer waiting on a full queue
            try:
                self._queue.get_nowait()
            except queue.Empty:
                break
        if self._started and self._thread is not threading.current_thread():
            self._thread.join(timeout=10.0)
        self._pool.shutdown(wait=False)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def to_device(batch, device):
    """Tensors of a prefetched batch -> `device` (asynchronous for page-locked ones); the rest as is."""
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, (tuple, list)):
        return type(batch)(to_device(v, device) for v in batch)
    return batch
