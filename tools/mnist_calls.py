#!/usr/bin/env python
"""Per-call contraction times of one DCGAN-MNIST iteration (eager, HIP events around every conv / deconv / linear call),
slowest first: which layers of the y-conditioned 28 x 28 model cost what.

    python tools/mnist_calls.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from gan_amd import _lib, autotune  # noqa: E402
from gan_amd.dcgan import DCGAN  # noqa: E402
from gan_amd.main import build_parser  # noqa: E402

autotune.enable(True)
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
o = build_parser().parse_args("--dataset mnist --input_height 28 --output_height 28".split())
o.input_width, o.output_width, o.y_dim, o.c_dim = 28, 28, 10, 1
m = DCGAN(None, o, load_data=False)
inp = [dev(rng.uniform(0, 1, (64, 28, 28, 1))), dev(rng.uniform(-1, 1, (64, 100))), dev(np.eye(10)[rng.integers(0, 10, 64)])]
for _ in range(3):
    m.train_iteration(*inp)
iters = 5
with bench.ContractionTimer(_lib) as tm:
    for _ in range(iters):
        m.train_iteration(*inp)
    ms, flops, calls, per = tm.summary()
print("contraction calls: %d per iteration, %.3f ms per iteration" % (calls // iters, ms / iters))
for k, (n, t_ms, fl) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-64s calls %2d  %7.1f us/call  %7.1f us/iter  %6.2f TFLOP/s" % (k, n // iters, t_ms / n * 1e3, t_ms / iters * 1e3,
                                                                          fl / (t_ms * 1e-3) / 1e12))
