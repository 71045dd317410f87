#!/usr/bin/env python
"""Throughput of the other BASELINE configurations on one MI355X (synthetic inputs of the shapes the
reference's loaders deliver): DCGAN-MNIST y-conditioned, Text2Image GAN-CLS, CycleGAN 256x256.

    python tools/model_bench.py [--model mnist|text2image|cyclegan|all] [--steps N] [--no-graph]
Prints one JSON line per model.  Eager warm-up tunes the contraction plans first."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gan_amd import autotune  # noqa: E402
from gan_amd.main import build_parser  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def marker():
    """Three 3-element fill launches in a row: brackets the timed region in a kernel trace (tools/trace_summary.py)."""
    from gan_amd import _lib
    buf = torch.zeros(4, device='cuda')
    for _ in range(3):
        _lib.check(_lib.lib.gan_fill(buf.data_ptr(), 3, 1.0, _lib.stream()))


def timed(run, inputs, steps, warmup=3):
    for _ in range(warmup):
        out = run(*inputs)
    torch.cuda.synchronize()
    marker()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run(*inputs)
    torch.cuda.synchronize()
    marker()
    dt = (time.perf_counter() - t0) / steps
    vals = {k: float(v) for k, v in out.items() if getattr(v, "numel", lambda: 0)() == 1}
    assert np.isfinite(list(vals.values())).all(), vals
    return dt, vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="all")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="keep the library's heuristic contraction plans")
    ap.add_argument("--plans", default=None,
                    help="JSON file of pinned contraction plans, loaded before a model is built (default: "
                         "profiles/r03_plans_<model>.json where it exists, found by tools/insitu_tune.py; 'none' tunes every "
                         "layer shape at start-up)")
    args = ap.parse_args()
    autotune.enable(not args.no_autotune)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def pin_plans(model):
        plans = args.plans or os.path.join(root, "profiles", "r03_plans_%s.json" % model)
        if not args.no_autotune and plans != "none" and os.path.exists(plans):
            autotune.load(plans)
    rng = np.random.default_rng(0)
    P = build_parser()
    if args.model in ("mnist", "all"):
        from gan_amd.dcgan import DCGAN
        pin_plans("mnist")
        o = P.parse_args("--dataset mnist --input_height 28 --output_height 28".split())
        o.input_width, o.output_width, o.y_dim, o.c_dim = 28, 28, 10, 1
        m = DCGAN(None, o, load_data=False)
        inp = [dev(rng.uniform(0, 1, (64, 28, 28, 1))), dev(rng.uniform(-1, 1, (64, 100))),
               dev(np.eye(10)[rng.integers(0, 10, 64)])]
        run = (lambda a, b, c: m.train_iteration(a, b, c)) if args.no_graph else m.iteration_runner(*inp)
        dt, vals = timed(run, inp, args.steps)
        print(json.dumps({"model": "DCGAN MNIST 28x28 y-conditioned, batch 64", "ms_per_step": round(dt * 1e3, 3),
                          "images_per_s": round(64 / dt, 1), "losses": vals}), flush=True)
    if args.model in ("text2image", "all"):
        from gan_amd.text2image import Text2Image
        pin_plans("text2image")
        o = P.parse_args("--dataset flowers --input_height 64 --model Text2Image --textEncoder DSSJE".split())
        m = Text2Image(None, o)
        inp = [dev(rng.uniform(0, 1, (64, 64, 64, 3))), dev(rng.uniform(0, 1, (64, 64, 64, 3))),
               dev(rng.standard_normal((64, 1024))), dev(rng.uniform(-1, 1, (64, 100)))]
        run = (lambda a, b, c, d: m.train_iteration(a, b, c, d)) if args.no_graph else m.iteration_runner(*inp)
        dt, vals = timed(run, inp, args.steps)
        print(json.dumps({"model": "Text2Image GAN-CLS 64x64, 1024-d captions, batch 64", "ms_per_step": round(dt * 1e3, 3),
                          "images_per_s": round(64 / dt, 1), "losses": vals}), flush=True)
    if args.model in ("cyclegan", "all"):
        from gan_amd.cyclegan import CycleGAN
        pin_plans("cyclegan")
        o = P.parse_args("--model CycleGAN --batch_size 1 --dataset horse2zebra".split())     # image pool of 50
        m = CycleGAN(None, o)
        inp = [dev(rng.uniform(-1, 1, (1, 256, 256, 6)))]
        run = (lambda a: m.train_iteration(a)) if args.no_graph else m.iteration_runner(inp[0])
        dt, vals = timed(run, inp, max(60, args.steps))      # long enough for the pool to fill and be drawn from
        print(json.dumps({"model": "CycleGAN 256x256 ResNet-9, batch 1, image pool 50 (two graphs, the pool asked in between)",
                          "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(1 / dt, 2), "losses": vals}), flush=True)


if __name__ == "__main__":
    main()
