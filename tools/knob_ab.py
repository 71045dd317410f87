#!/usr/bin/env python
"""A/B of library knobs inside ONE process on ONE GPU: the DCGAN-64 iteration (bench.py's step) is captured as a
HIP graph once per variant (a variant = knob settings in force while it is captured: kernel selection happens at
capture time, a replay launches what was captured) and the graphs are replayed in alternating rounds.

    python tools/knob_ab.py --a no_carry=1 --b no_carry=0 [--rounds 7] [--steps 30] [--model dcgan|t2i|cyclegan]
Prints per variant the per-round ms/step, median and minimum."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from gan_amd import _lib, autotune, ops  # noqa: E402


DEFAULTS = {"force_cfg": -1, "force_splits": 0, "no_thin": 0, "slow_gather": 0, "scalar_epilogue": 0, "three_launch_bn": 0,
            "two_launch_bn": 1, "separate_colsum": 0, "tile_order": 0, "rgb_gather": 1, "row_gather_any": 1, "no_carry": 0, "no_bn_ride": 0, "rgb_direct": 1,
            "ride_parts": 15, "taps": 1}     # (host side, gan_amd.ops.RIDE_PARTS: 1 forward statistics + 2 backward sums + 4 column sums
                                              #  + 8 backward sums in the launch that adds the gradients of a fan-out)


def parse(spec):
    out = {}
    for item in filter(None, spec.split(",")):
        k, v = item.split("=")
        out[k] = int(v)
    return out


def build(model_name, rng):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    if model_name == "dcgan":
        from gan_amd.dcgan import DCGAN
        m = DCGAN(None, bench.dcgan64_opts(seed=0), load_data=False)
        batch = [dev(rng.uniform(-1, 1, (64, 64, 64, 3))), dev(rng.uniform(-1, 1, (64, 100)))]
        return m, batch, lambda: m.iteration_runner(batch[0], batch[1])
    from gan_amd.main import build_parser
    P = build_parser()
    if model_name == "t2i":
        from gan_amd.text2image import Text2Image
        o = P.parse_args("--dataset flowers --input_height 64 --model Text2Image --textEncoder DSSJE".split())
        m = Text2Image(None, o)
        batch = [dev(rng.uniform(0, 1, (64, 64, 64, 3))), dev(rng.uniform(0, 1, (64, 64, 64, 3))),
                 dev(rng.standard_normal((64, 1024))), dev(rng.uniform(-1, 1, (64, 100)))]
        return m, batch, lambda: m.iteration_runner(*batch)
    from gan_amd.cyclegan import CycleGAN
    o = P.parse_args("--model CycleGAN --batch_size 1 --dataset horse2zebra".split())
    m = CycleGAN(None, o)
    batch = [dev(rng.uniform(-1, 1, (1, 256, 256, 6)))]
    return m, batch, lambda: m.iteration_runner(batch[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--plans", default=bench.PLANS_FILE)
    ap.add_argument("--no-autotune", action="store_true")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    autotune.enable(not args.no_autotune)
    if not args.no_autotune and args.model == "dcgan" and os.path.exists(args.plans):
        autotune.load(args.plans)
    variants = [("a", parse(args.a)), ("b", parse(args.b))]
    knobs = sorted(set(variants[0][1]) | set(variants[1][1]))
    model, batch, make_runner = build(args.model, np.random.default_rng(1000))
    runs = {}
    for name, setting in variants:
        for k in knobs:                       # a knob one variant names and the other does not: its default there
            value = setting.get(k, DEFAULTS[k])
            if k == "taps":       # (host side, gan_amd.ops.TAPS_PATH: few-channel 7 x 7 convolutions by taps)
                ops.TAPS_PATH[0] = bool(value)
            elif k == "ride_parts":
                ops.RIDE_PARTS.update({'stats': bool(value & 1), 'sums': bool(value & 2), 'colsum': bool(value & 4),
                                       'fan_out': bool(value & 8)})
            else:
                _lib.knob_set(k, value)
        ops._ws_cache.clear()
        model.train_iteration(*batch)          # eager: workspace sizes follow the plans
        torch.cuda.synchronize()
        runs[name] = make_runner()
    times = {name: [] for name, _ in variants}
    for r in range(args.rounds):
        for name, _ in (variants if r % 2 == 0 else variants[::-1]):
            run = runs[name]
            for _ in range(3):
                run(*batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run(*batch)
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / args.steps * 1e3)
    out = {"model": args.model}
    for name, setting in variants:
        t = times[name]
        out[name] = {"knobs": setting, "ms_per_step_rounds": [round(v, 4) for v in t], "median": round(float(np.median(t)), 4),
                     "min": round(min(t), 4)}
    out["b_over_a_median"] = round(out["b"]["median"] / out["a"]["median"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
