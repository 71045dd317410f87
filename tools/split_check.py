#!/usr/bin/env python
"""fp32-MFMA against split-bf16-MFMA contraction kernels on the DCGAN-64 layer shapes: largest
difference of the results (relative to the largest output magnitude) and time per call.
Tuning / validation aid; not part of the product path.

    python tools/split_check.py [--batch 64] [--reps 20] [--splits N]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from gan_amd import ops  # noqa: E402
from kernel_bench import LAYERS, PASSES, run_pass, time_pass  # noqa: E402
from gan_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--splits", default=None)
    ap.add_argument("--cfgs", default="0,4,2,6,1,5")
    args = ap.parse_args()
    ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    wp, wn = ws.data_ptr(), ws.numel()
    if args.splits:
        L.knob_set("force_splits", int(args.splits))
    for lname, (n, h, w_, cb, cs, k, s) in LAYERS[1:]:
        n = args.batch
        g = ops._conv_geom(n, h, w_, cb, cs, k, k, s, s, 'SAME')
        big = torch.randn(n, h, w_, cb, device='cuda')
        small = torch.randn(n, g.hs, g.ws, cs, device='cuda')
        w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
        flops = 2.0 * n * g.hs * g.ws * cs * k * k * cb
        for p in PASSES:
            ref = None
            for cfg in [int(c) for c in args.cfgs.split(",")]:
                L.knob_set("force_cfg", int(str(cfg)))
                bufs = (big.clone(), small.clone(), w, torch.zeros_like(w), torch.empty(cs, device='cuda'))
                L.check(run_pass(g, p, bufs, wp, wn))
                torch.cuda.synchronize()
                out = {"fwd": bufs[1], "dgrad": bufs[0], "wgrad": bufs[3]}[p].double()
                if ref is None:
                    ref = out           # differences are taken against the first configuration listed
                err = float((out - ref).abs().max() / ref.abs().max())
                us = time_pass(g, p, bufs, wp, wn, args.reps)
                print("%-5s %-6s cfg %d  %8.1f us  %6.1f TFLOP/s  max|diff|/max|ref| %.2e"
                      % (lname, p, cfg, us, flops / us / 1e6, err), flush=True)
    L.knob_set("force_cfg", -1)


if __name__ == "__main__":
    main()
