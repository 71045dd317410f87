"""HIP-graph capture of a whole training iteration.

A DCGAN-64 iteration is a few hundred kernel launches of 5-100 us each; issued one by one from
Python the host cannot keep the GPU fed.  All shapes are static, every buffer is owned by torch's
caching allocator and the library neither allocates nor synchronises, so the iteration is
captured once into a hipGraph (through torch.cuda.CUDAGraph, which records whatever is queued on
the capturing stream, including the ctypes-launched kernels) and replayed with one host call.
The Adam step count lives in device memory, so replays advance the bias correction correctly.
"""
import torch

from .prefetch import capture_lock


class CapturedIteration(object):
    """Capture `fn(*inputs)`; inputs are copied into static buffers before every replay.

    `snapshot` / `restore` are callables that save and put back whatever state the warm-up run
    mutates (weights, Adam slots, BN moving averages), so that capturing leaves the model exactly
    where it was."""

    def __init__(self, fn, example_inputs, snapshot=None, restore=None, warmup=2, share_inputs=False):
        # share_inputs: the caller already owns static buffers (several captured segments reading
        # the same batch) and refreshes them itself
        self.static_inputs = (list(example_inputs) if share_inputs
                              else [None if t is None else t.clone() for t in example_inputs])
        saved = snapshot() if snapshot is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):            # sizes the workspace and the allocator pools eagerly
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if restore is not None:
            restore(saved)
        saved = snapshot() if snapshot is not None else None
        self.graph = torch.cuda.CUDAGraph()
        with capture_lock:                      # no page-locking by the input pipeline's thread meanwhile
            with torch.cuda.graph(self.graph):
                self.outputs = fn(*self.static_inputs)
        torch.cuda.synchronize()
        if restore is not None:
            restore(saved)                      # capture does not execute, but keep the contract explicit

    def __call__(self, *inputs):
        for dst, src in zip(self.static_inputs, inputs):
            if dst is not None and src is not dst:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.outputs


class SegmentedIteration(object):
    """Capture `fn(*inputs)` as a CHAIN of HIP graphs, cut at every point where `fn` has to do
    something a graph cannot hold -- here the gradient all-reduces of data-parallel training.

    While the capture pass runs, code inside `fn` calls `SegmentedIteration.cut(callback)` (the
    optimiser's all_reduce does): the graph under construction is closed, `callback` is remembered
    as the eager action between two graphs, and a new graph is opened in the same memory pool, so
    tensors and the autograd tape made in one segment stay valid in the next.  A replay runs
    graph, callback, graph, callback, ... on the current stream.  Outside a capture pass `cut`
    simply runs the callback, which is what the eager warm-up runs do."""

    active = None

    def __init__(self, fn, example_inputs, snapshot=None, restore=None, warmup=2):
        self.static_inputs = list(example_inputs)          # the caller owns and refreshes them
        saved = snapshot() if snapshot is not None else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn(*self.static_inputs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if restore is not None:
            restore(saved)
        saved = snapshot() if snapshot is not None else None
        self.chain = []             # [(graph, callback or None)]
        self._stream = torch.cuda.Stream()
        self._pool = torch.cuda.graph_pool_handle()
        self._capturing = False
        SegmentedIteration.active = self
        with capture_lock:                      # no page-locking by the input pipeline's thread meanwhile
            try:
                self._open()
                self.outputs = fn(*self.static_inputs)
                self._close(None)
            finally:
                SegmentedIteration.active = None
                if self._capturing:         # fn raised: leave the stream out of capture mode
                    try:
                        self._graph.capture_end()
                    finally:
                        self._ctx.__exit__(None, None, None)
                        self._capturing = False
        torch.cuda.synchronize()
        if restore is not None:
            restore(saved)

    def _open(self):
        import gc
        torch.cuda.synchronize()
        gc.collect()
        self._graph = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        self._graph.capture_begin(pool=self._pool)
        self._capturing = True

    def _close(self, callback):
        self._graph.capture_end()
        self._capturing = False
        self._ctx.__exit__(None, None, None)
        self.chain.append((self._graph, callback))

    @classmethod
    def cut(cls, callback):
        self = cls.active
        if self is None:
            callback()
            return
        self._close(callback)
        self._open()

    def __call__(self):
        for graph, callback in self.chain:
            graph.replay()
            if callback is not None:
                callback()
        return self.outputs


def model_state_snapshot(store, optims):
    """Clone everything a training iteration mutates."""
    snap = {'vars': {n: v.data.clone() for n, v in store.vars.items() if v.group is None},
            'groups': {k: (g.values.clone(), g.m.clone(), g.v.clone()) for k, g in store.groups.items()},
            'steps': [o.step_dev.clone() for o in optims]}
    return snap


def model_state_restore(store, optims, snap):
    for n, t in snap['vars'].items():
        store.vars[n].data.copy_(t)
    for k, (val, m, v) in snap['groups'].items():
        g = store.groups[k]
        g.values.copy_(val)
        g.m.copy_(m)
        g.v.copy_(v)
    for o, s in zip(optims, snap['steps']):
        o.step_dev.copy_(s)


def iteration_runner(store, optims, fn, example_inputs, world_size=1, process_group=None, segmented=None):
    """`fn(*inputs)` (one training iteration of a model whose variables live in `store` and are stepped by
    `optims`) as replayable HIP graphs over static copies of `example_inputs` (None members stay None).

    One GPU: one graph.  Data parallel (or `segmented`): a chain of graphs cut at the gradient exchanges,
    which run between them on the same stream.  Returns run(*inputs) -> what fn returns; the tensors in
    it are static and overwritten by the next replay."""
    snap = lambda: model_state_snapshot(store, optims)
    rest = lambda s: model_state_restore(store, optims, s)
    static = [None if t is None else t.clone() for t in example_inputs]
    if segmented is None:
        segmented = world_size > 1 or process_group is not None
    if segmented:
        replay = SegmentedIteration(fn, static, snap, rest)
    else:
        whole = CapturedIteration(fn, static, snap, rest, share_inputs=True)
        replay = lambda: whole(*static)

    def run(*inputs):
        if len(inputs) > len(static):
            raise ValueError("iteration runner captured with %d inputs, called with %d" % (len(static), len(inputs)))
        for dst, src in zip(static, inputs):
            if dst is not None and src is not dst:
                if src is None:
                    raise ValueError("iteration runner: an input that was a tensor at capture time is None now")
                dst.copy_(src, non_blocking=True)
        return replay()
    return run
