#!/usr/bin/env python
"""How much does the unequal tap count of the stride-parity classes cost the transposed contraction?

A 5x5 stride-2 filter has classes of 9 / 6 / 6 / 4 taps, a 4x4 one four classes of 4 and a 6x6 one four classes of 9:
the three are timed on the DCGAN layer shapes (input gradient of conv2d, batch 64) and compared per filter tap."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gan_amd import _lib as L  # noqa: E402
from gan_amd import ops, autotune  # noqa: E402

LAYERS = [("d_h1", (32, 64, 128)), ("d_h2", (16, 128, 256)), ("d_h3", (8, 256, 512))]


def graph_us(body, reps=20):
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
tune = "--autotune" in sys.argv
for name, (h, cb, cs) in LAYERS:
    for n in (64, 128):
        row = []
        for k in (4, 5, 6):
            g = ops._conv_geom(n, h, h, cb, cs, k, k, 2, 2, 'SAME')
            if tune:
                autotune.tune_geometry(g)
            big = torch.empty(n, h, h, cb, device='cuda')
            small = torch.randn(n, g.hs, g.ws, cs, device='cuda')
            w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
            us = graph_us(lambda: L.check(L.lib.gan_conv2d_dgrad(ctypes.byref(g), L.ptr(small), L.ptr(w), L.ptr(big), wp, wn, L.stream())))
            fw = graph_us(lambda: L.check(L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(big), L.ptr(w), None, L.ptr(small), wp, wn, L.stream())))
            row.append("k=%d: dgrad %6.1f us (%5.2f / tap)  fwd %6.1f us (%5.2f / tap)" % (k, us, us / (k * k), fw, fw / (k * k)))
        print("%s batch %3d  " % (name, n) + "   ".join(row), flush=True)
