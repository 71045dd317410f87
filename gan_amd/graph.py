"""HIP-graph capture of a whole training iteration.

A DCGAN-64 iteration is a few hundred kernel launches of 5-100 us each; issued one by one from
Python the host cannot keep the GPU fed.  All shapes are static, every buffer is owned by torch's
caching allocator and the library neither allocates nor synchronises, so the iteration is
captured once into a hipGraph (through torch.cuda.CUDAGraph, which records whatever is queued on
the capturing stream, including the ctypes-launched kernels) and replayed with one host call.
The Adam step count lives in device memory, so replays advance the bias correction correctly.
"""
import torch


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
