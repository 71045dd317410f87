#!/usr/bin/env python
"""Time the thin-output transposed contraction (DCGAN g_h4 forward / d_h0 input gradient: 64 -> 3 channels at
64x64, batch 64) with and without the fused epilogues."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib as L  # noqa: E402
from gan_amd import ops  # noqa: E402


def timeit(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n = 64
    g = ops._conv_geom(n, 64, 64, 3, 64, 5, 5, 2, 2, 'SAME')
    gp = ctypes.byref(g)
    small = torch.randn(n, 32, 32, 64, device='cuda')
    wt = torch.randn(5, 5, 3, 64, device='cuda') * 0.05
    b = torch.randn(3, device='cuda')
    big = torch.empty(n, 64, 64, 3, device='cuda')
    gate = torch.tanh(torch.randn(n, 64, 64, 3, device='cuda'))
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    st = L.stream()
    P = L.ptr
    print("deconv fwd plain      %.1f us" % timeit(lambda: L.lib.gan_deconv2d_fwd(gp, P(small), P(wt), P(b), P(big), wp, wn, st)))
    for act, name in ((0, "none*0.5"), (1, "relu"), (2, "lrelu"), (3, "tanh"), (4, "sigmoid")):
        sc = 0.5 if act == 0 else 1.0
        print("deconv fwd act %-8s %.1f us" % (name, timeit(lambda: L.lib.gan_deconv2d_fwd_act(
            gp, P(small), P(wt), P(b), P(big), act, 0.2, sc, 0.0, wp, wn, st))))
    print("conv dgrad plain      %.1f us" % timeit(lambda: L.lib.gan_conv2d_dgrad(gp, P(small), P(wt), P(big), wp, wn, st)))
    for act, name in ((2, "lrelu"), (3, "tanh")):
        print("conv dgrad gated %-6s %.1f us" % (name, timeit(lambda: L.lib.gan_conv2d_dgrad_gated(
            gp, P(small), P(wt), P(big), P(gate), act, 0.2, 1.0, 0.0, wp, wn, st))))
    # d_h0 forward: 3 -> 64 channels, output-bound (16.8 MB)
    img = torch.randn(n, 64, 64, 3, device='cuda')
    h0 = torch.empty(n, 32, 32, 64, device='cuda')
    b64 = torch.randn(64, device='cuda')
    for cfg in (1, 2):
        L.knob_set("force_cfg", cfg)
        L.knob_set("force_splits", 1)
        print("conv fwd 3->64 cfg %d plain  %.1f us" % (cfg, timeit(lambda: L.lib.gan_conv2d_fwd(gp, P(img), P(wt), P(b64), P(h0), wp, wn, st))))
        print("conv fwd 3->64 cfg %d lrelu  %.1f us" % (cfg, timeit(lambda: L.lib.gan_conv2d_fwd_act(
            gp, P(img), P(wt), P(b64), P(h0), 2, 0.2, 1.0, 0.0, wp, wn, st))))
    L.knob_set("force_cfg", -1)
    L.knob_set("force_splits", 0)
    x = torch.randn(n, 64, 64, 3, device='cuda')
    y = torch.empty_like(x)
    print("act_fwd tanh on 3 MB  %.1f us" % timeit(lambda: L.lib.gan_act_fwd(P(x), P(y), x.numel(), 3, 0.0, 1.0, 0.0, st)))
    print("act_fwd lrelu on 3 MB %.1f us" % timeit(lambda: L.lib.gan_act_fwd(P(x), P(y), x.numel(), 2, 0.2, 1.0, 0.0, st)))


if __name__ == "__main__":
    main()
