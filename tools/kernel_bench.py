#!/usr/bin/env python
"""Per-layer timing of the contraction kernels on the DCGAN-64 shapes (HIP events), optionally
sweeping the tile configuration / split-K factor.  Tuning aid; not part of the product path.

    python tools/kernel_bench.py                 # library heuristics
    python tools/kernel_bench.py --sweep         # every cfg x splits, best per layer/pass
    python tools/kernel_bench.py --only d_h1:fwd --reps 50
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gan_amd import _lib as L  # noqa: E402
from gan_amd import ops  # noqa: E402

# name, (n, h, w, cb, cs, k, s): conv2d x[n,h,w,cb] -> [n,h/s,w/s,cs]; deconv layers use the same
# geometry with big = output
LAYERS = [
    ("d_h0", (64, 64, 64, 3, 64, 5, 2)),
    ("d_h1", (64, 32, 32, 64, 128, 5, 2)),
    ("d_h2", (64, 16, 16, 128, 256, 5, 2)),
    ("d_h3", (64, 8, 8, 256, 512, 5, 2)),
]
PASSES = ["fwd", "dgrad", "wgrad"]   # FPROP, TPROP, WGRAD of the conv2d naming (deconv: dgrad, fwd, wgrad)


def run_pass(g, name, bufs, wp, wn):
    big, small, w, dw, db = bufs
    st = L.stream()
    if name == "fwd":
        return L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(big), L.ptr(w), None, L.ptr(small), wp, wn, st)
    if name == "dgrad":
        return L.lib.gan_conv2d_dgrad(ctypes.byref(g), L.ptr(small), L.ptr(w), L.ptr(big), wp, wn, st)
    return L.lib.gan_conv2d_wgrad(ctypes.byref(g), L.ptr(big), L.ptr(small), L.ptr(dw), None, 0, wp, wn, st)


def time_pass(g, name, bufs, wp, wn, reps):
    for _ in range(3):
        L.check(run_pass(g, name, bufs, wp, wn))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run_pass(g, name, bufs, wp, wn)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None, help="layer:pass, e.g. d_h1:fwd")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--autotune", action="store_true", help="pin the tuned plan of each layer before timing it")
    ap.add_argument("--cfg", default=None)
    ap.add_argument("--splits", default=None)
    args = ap.parse_args()
    ws_bytes = 1 << 30
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda')
    wp, wn = ws.data_ptr(), ws_bytes
    for lname, (n, h, w_, cb, cs, k, s) in LAYERS:
        n = args.batch
        g = ops._conv_geom(n, h, w_, cb, cs, k, k, s, s, 'SAME')
        if args.autotune:
            from gan_amd import autotune
            autotune.tune_geometry(g)
        big = torch.randn(n, h, w_, cb, device='cuda')
        small = torch.randn(n, g.hs, g.ws, cs, device='cuda')
        w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
        dw = torch.empty_like(w)
        db = torch.empty(cs, device='cuda')
        bufs = (big, small, w, dw, db)
        flops = 2.0 * n * g.hs * g.ws * cs * k * k * cb
        for p in PASSES:
            if args.only and args.only != "%s:%s" % (lname, p):
                continue
            L.knob_set("force_cfg", -1)
            L.knob_set("force_splits", 0)
            if args.cfg is not None:
                L.knob_set("force_cfg", int(args.cfg))
            if args.splits is not None:
                L.knob_set("force_splits", int(args.splits))
            us = time_pass(g, p, bufs, wp, wn, args.reps)
            print("%-5s %-6s heuristic        %8.1f us  %6.1f TFLOP/s" % (lname, p, us, flops / us / 1e6), flush=True)
            if not args.sweep:
                continue
            best = None
            for cfg in range(4):
                for splits in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                    L.knob_set("force_cfg", int(str(cfg)))
                    L.knob_set("force_splits", int(str(splits)))
                    try:
                        us = time_pass(g, p, bufs, wp, wn, max(3, args.reps // 4))
                    except L.GanHipError:
                        continue
                    if best is None or us < best[0]:
                        best = (us, cfg, splits)
            print("%-5s %-6s best cfg %d splits %2d %8.1f us  %6.1f TFLOP/s"
                  % (lname, p, best[1], best[2], best[0], flops / best[0] / 1e6), flush=True)


if __name__ == "__main__":
    main()
