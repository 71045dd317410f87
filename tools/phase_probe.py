#!/usr/bin/env python
"""Where the time of one contraction launch goes, from clock stamps taken INSIDE the kernel (probe build of the
library: `make -C gan_amd/csrc STAMPS=1`, loaded through GAN_HIP_LIB).  Thread 0 of every workgroup of
igemm_split_kernel leaves the constant-rate 100 MHz clock at: 0 entry, 1 set-up done (offsets, descriptors), 2 pipeline
primed (first slices fetched, first LDS image written), 3 slice loop done, 4 k groups merged, 5 tile stored.

    GAN_HIP_LIB=gan_amd/lib/libgan_hip_stamps.so python tools/phase_probe.py [--batch 64] [--plans profiles/r03_plans.json]
Prints, per DCGAN-64 layer pass, the launch's span (first entry to last store) and the median / maximum over
workgroups of every phase."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gan_amd import _lib as L, autotune  # noqa: E402
from gan_amd.ops import _conv_geom  # noqa: E402

LAYERS = [("d_h1", 32, 64, 128), ("d_h2", 16, 128, 256), ("d_h3", 8, 256, 512)]
PHASES = ["set-up", "prime", "slice loop", "merge k groups", "store tile"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--plans", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_plans.json"))
    ap.add_argument("--cold", action="store_true", help="sweep a 256 MB buffer before the stamped launch (cold caches, as inside an iteration)")
    args = ap.parse_args()
    assert hasattr(L.lib, "gan_debug_set_stamps"), "load the probe build: GAN_HIP_LIB=gan_amd/lib/libgan_hip_stamps.so"
    L.lib.gan_debug_set_stamps.argtypes = [ctypes.c_void_p]
    L.lib.gan_debug_set_stamps.restype = None
    if os.path.exists(args.plans):
        autotune.load(args.plans)
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device='cuda')
    sweep = torch.empty(64 << 20, device='cuda')
    n = args.batch
    for name, hw, cb, cs in LAYERS:
        g = _conv_geom(n, hw, hw, cb, cs, 5, 5, 2, 2, 'SAME')
        gp = ctypes.byref(g)
        x = torch.randn(n, hw, hw, cb, device='cuda')
        w = torch.randn(5, 5, cb, cs, device='cuda') * 0.05
        y = torch.randn(n, g.hs, g.ws, cs, device='cuda')
        dw = torch.empty_like(w)
        wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
        st = L.stream()
        calls = {"fwd": lambda: L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(w), None, L.ptr(y), wp, wn, st),
                 "dgrad": lambda: L.lib.gan_conv2d_dgrad(gp, L.ptr(y), L.ptr(w), L.ptr(x), wp, wn, st),
                 "wgrad": lambda: L.lib.gan_conv2d_wgrad(gp, L.ptr(x), L.ptr(y), L.ptr(dw), None, 0, wp, wn, st)}
        for mode, (what, call) in enumerate(calls.items()):
            c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            L.check(L.lib.gan_conv_plan_get(mode, gp, ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
            L.lib.gan_debug_set_stamps(None)
            for _ in range(3):
                L.check(call())
            if args.cold:
                sweep.fill_(1.0)
            stamps.zero_()
            torch.cuda.synchronize()
            L.lib.gan_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(call())
            e1.record()
            torch.cuda.synchronize()
            L.lib.gan_debug_set_stamps(None)
            raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
            raw = raw[(raw[:, 0] > 0) & (raw[:, 5] > 0)]  # workgroups that ran the whole kernel
            t = raw[:, :6]
            loop_ticks = raw[:, 3] - raw[:, 2]
            ghz = np.median((raw[:, 7] - raw[:, 6])[loop_ticks > 0] / loop_ticks[loop_ticks > 0]) * 0.1
            us = lambda ticks: ticks / 100.0
            span = us(t[:, 5].max() - t[:, 0].min())
            start_skew = us(t[:, 0].max() - t[:, 0].min())
            d = us(np.diff(t, axis=1))
            print("%s %s  cfg %d splits %d  workgroups %d  events %.1f us  span %.1f us  start skew %.1f us  per workgroup %.1f (median) .. %.1f us"
                  % (name, what, c.value % 16, sp.value, len(t), e0.elapsed_time(e1) * 1e3, span, start_skew,
                     np.median(us(t[:, 5] - t[:, 0])), us(t[:, 5] - t[:, 0]).max()))
            print("    shader clock inside the slice loop %.2f GHz" % ghz)
            print("    " + "   ".join("%s %.1f / %.1f" % (PHASES[i], np.median(d[:, i]), d[:, i].max()) for i in range(5)) + "   (median / max us)")


if __name__ == "__main__":
    main()
