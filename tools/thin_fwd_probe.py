#!/usr/bin/env python
"""Forward and weight-gradient passes with a few output channels (CycleGAN's 7x7 image head 64 -> 3, its
discriminators' 512 -> 1 patch head): the 128x32 tile on the fp32 MFMA (configuration 3) against the same tile on the
bf16 pipe (configuration 9), each at its best split-K factor."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_amd import _lib as L  # noqa: E402
from gan_amd import ops  # noqa: E402

CASES = [("image head 64->3 7x7 @256", (1, 256, 256, 64, 3, 7, 1)), ("patch head 512->1 4x4 @32", (1, 32, 32, 512, 1, 4, 1)),
         ("image head, batch 4", (4, 256, 256, 64, 3, 7, 1))]


def graph_us(body, reps=10):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): body()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / reps * 1e3


ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
wp, wn = ws.data_ptr(), 1 << 30
for name, (n, h, w_, cb, cs, k, s) in CASES:
    g = ops._conv_geom(n, h, w_, cb, cs, k, k, s, s, 'SAME')
    x = torch.randn(n, h, w_, cb, device='cuda')
    y = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    dw = torch.empty_like(w)
    for pname, call in (("fwd", lambda: L.check(L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(x), L.ptr(w), None, L.ptr(y), wp, wn, L.stream()))),
                        ("wgrad", lambda: L.check(L.lib.gan_conv2d_wgrad(ctypes.byref(g), L.ptr(x), L.ptr(y), L.ptr(dw), None, 0, wp, wn, L.stream())))):
        res = []
        for cfg in (3, 9):
            best = None
            for splits in (1, 2, 4, 8, 16, 32, 64):
                L.knob_set("force_cfg", cfg)
                L.knob_set("force_splits", splits)
                try:
                    us = graph_us(call)
                except L.GanHipError:
                    continue
                if best is None or us < best[0]:
                    best = (us, splits)
            res.append("cfg %d: %7.1f us (splits %d)" % (cfg, best[0], best[1]))
        L.knob_set("force_cfg", -1)
        L.knob_set("force_splits", 0)
        print("%-28s %-5s %s   heuristic %7.1f us" % (name, pname, "   ".join(res), graph_us(call)), flush=True)
