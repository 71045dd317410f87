#!/usr/bin/env python
"""Replay subsets of the C-ABI calls of one DCGAN-64 iteration as HIP graphs of their own (on the buffers of a
captured iteration) and report the time per call and per kernel launch: separates what a family of kernels
costs by itself from what it costs inside the whole iteration."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from gan_amd import _lib, autotune  # noqa: E402
from gan_amd._lib import stream  # noqa: E402
from gan_amd.dcgan import DCGAN  # noqa: E402

LAUNCHES = {"gan_bn_fwd_train": 3, "gan_bn_bwd": 3, "gan_bn_moving_update": 1, "gan_act_fwd": 1, "gan_act_bwd": 1,
            "gan_adam_tf_step": 2, "gan_add_fwd": 1, "gan_fill": 1}


def main():
    autotune.enable(True)
    model = DCGAN(None, bench.dcgan64_opts(seed=0), load_data=False)
    rng = np.random.default_rng(1)
    im = torch.from_numpy(rng.uniform(-1, 1, (64, 64, 64, 3)).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.uniform(-1, 1, (64, 100)).astype(np.float32)).cuda()
    for _ in range(3):
        model.train_iteration(im, z)
    torch.cuda.synchronize()
    lib = _lib.lib
    names = [n for n in _lib.SIGNATURES if n.startswith("gan_") and "workspace" not in n and "plan" not in n
             and n not in ("gan_last_error", "gan_set_exact_f32", "gan_get_exact_f32", "gan_linear_as_conv_geom", "gan_version")]
    calls, saved = [], {}

    def wrap(name, fn):
        def call(*args):
            calls.append((name, fn, args))
            return fn(*args)
        return call
    for n in names:
        saved[n] = getattr(lib, n)
        setattr(lib, n, wrap(n, saved[n]))
    holder = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(holder):
            model.train_iteration(im, z)
    finally:
        for n, fn in saved.items():
            setattr(lib, n, fn)
    count = collections.Counter(c[0] for c in calls)
    print("calls in one iteration:", dict(count))

    def time_subset(pred, label):
        sub = [c for c in calls if pred(c[0])]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for name, fn, args in sub:
                rc = fn(*(tuple(args[:-1]) + (stream(),)))
                assert rc == 0, (name, lib.gan_last_error())
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        launches = sum(LAUNCHES.get(n, 0) for n, _, _ in sub)
        print("%-34s %4d calls %8.1f us  %6.2f us per call%s" % (label, len(sub), us, us / max(1, len(sub)),
              ("  %5.2f us per launch" % (us / launches)) if launches else ""))
        return us
    time_subset(lambda n: True, "whole iteration")
    time_subset(lambda n: n.startswith(("gan_conv2d", "gan_deconv2d", "gan_linear")), "contractions")
    time_subset(lambda n: not n.startswith(("gan_conv2d", "gan_deconv2d", "gan_linear")), "everything else")
    time_subset(lambda n: n == "gan_bn_fwd_train", "bn forward")
    time_subset(lambda n: n == "gan_bn_bwd", "bn backward")
    time_subset(lambda n: n == "gan_bn_moving_update", "bn moving update")
    time_subset(lambda n: n in ("gan_act_fwd", "gan_act_bwd"), "activations")
    time_subset(lambda n: n == "gan_adam_tf_step", "adam")


if __name__ == "__main__":
    main()
