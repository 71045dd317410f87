#!/usr/bin/env python
"""Program for the HBM-traffic passes of the contraction family (run under `rocprofv3 --pmc ...`, one counter
set per run; summarised by tools/traffic_summary.py):

  1. calibration: a 512 MiB and a 16 MiB device-to-device copy of known size (16 bytes per lane, the access
     shape of the contraction kernels' loads, slab stores and slab reductions);
  2. the 62 conv / deconv / linear C-ABI calls of one DCGAN-64 iteration (batch 64), replayed alone as a HIP
     graph on the buffers of a captured iteration -- the launches bench.py's `roofline.achieved` is timed on --
     `--replays` times between two marker triples (three 3-element gan_fill launches in a row).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from gan_amd import _lib, autotune  # noqa: E402
from gan_amd._lib import stream  # noqa: E402
from gan_amd.dcgan import DCGAN  # noqa: E402


def marker(buf):
    for _ in range(3):
        _lib.check(_lib.lib.gan_fill(buf.data_ptr(), 3, 1.0, stream()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replays", type=int, default=3)
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--plans", default=None, help="JSON file of tuned plans: written if absent, pinned if present "
                    "(so that the two counter passes run the same kernels)")
    ap.add_argument("--meta", default=None, help="write {plans_sha16, source_sha16} of the launches here (JSON)")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    mark = torch.zeros(4, device='cuda')
    big_a = torch.randn(128 << 20, device='cuda')          # 512 MiB of fp32
    big_b = torch.empty_like(big_a)
    small_a, small_b = torch.randn(4 << 20, device='cuda'), torch.empty(4 << 20, device='cuda')
    torch.cuda.synchronize()
    marker(mark)
    big_b.copy_(big_a)
    small_b.copy_(small_a)
    # the gather shape of the contraction kernels: a 1x1 convolution reads every input value once, 64 bytes per
    # pixel and 16-deep slice (for 64 channels: a quarter of each pixel's 256 bytes per slice)
    import ctypes
    from gan_amd.ops import _conv_geom
    for n, hw, cb in ((1024, 32, 64), (2048, 32, 16)):     # 268.4 MB and 134.2 MB of input
        xg = torch.randn(n, hw, hw, cb, device='cuda')
        wg = torch.randn(1, 1, cb, 64, device='cuda')
        yg = torch.empty(n, hw, hw, 64, device='cuda')
        g = _conv_geom(n, hw, hw, cb, 64, 1, 1, 1, 1, 'SAME')
        wsb = int(_lib.lib.gan_conv_workspace_bytes(ctypes.byref(g)))
        wsp, wsn = _lib.workspace(wsb)
        _lib.check(_lib.lib.gan_conv2d_fwd(ctypes.byref(g), _lib.ptr(xg), _lib.ptr(wg), None, _lib.ptr(yg), wsp, wsn, stream()))
        torch.cuda.synchronize()
        print("gather calibration: x %s = %.1f MB read once" % (tuple(xg.shape), xg.numel() * 4 / 1e6), flush=True)
        del xg, wg, yg
    marker(mark)
    torch.cuda.synchronize()
    del big_a, big_b

    autotune.enable(not args.no_autotune)
    if args.plans and os.path.exists(args.plans):
        autotune.load(args.plans)
    model = DCGAN(None, bench.dcgan64_opts(seed=0), load_data=False)
    rng = np.random.default_rng(1000)
    im = torch.from_numpy(rng.uniform(-1, 1, (64, 64, 64, 3)).astype(np.float32)).cuda()
    z = torch.from_numpy(rng.uniform(-1, 1, (64, 100)).astype(np.float32)).cuda()
    for _ in range(2):
        model.train_iteration(im, z)
    torch.cuda.synchronize()
    if args.plans and not os.path.exists(args.plans):
        autotune.save(args.plans)
    holder = torch.cuda.CUDAGraph()
    with bench.ContractionRecorder(_lib) as rec:
        with torch.cuda.graph(holder):
            model.train_iteration(im, z)
    only = torch.cuda.CUDAGraph()
    with torch.cuda.graph(only):
        for fn, a, *_ in rec.records:
            _lib.check(fn(*(tuple(a[:-1]) + (stream(),))))
    only.replay()
    torch.cuda.synchronize()
    marker(mark)
    for _ in range(args.replays):
        only.replay()
    marker(mark)
    torch.cuda.synchronize()
    flops = sum(r[2] for r in rec.records)
    # which plans and which library sources these launches belong to (bench.py attaches the traffic figure only to runs
    # with the same two hashes)
    calls = [(plain, shape) for _, _, _, _, plain, shape in rec.records]
    meta = {"plans_sha16": bench.plan_set_sha16(_lib, sorted(set(calls))), "source_sha16": bench.source_sha16(),
            "plans_file": args.plans and os.path.relpath(os.path.abspath(args.plans), bench.ROOT), "replays": args.replays}
    if args.meta:
        import json
        json.dump(meta, open(args.meta, "w"))
    print("plan set %s, library sources %s" % (meta["plans_sha16"], meta["source_sha16"]), flush=True)
    print("contraction calls per iteration: %d, nominal GFLOP per iteration: %.2f, replays: %d"
          % (len(rec.records), flops / 1e9, args.replays), flush=True)


if __name__ == "__main__":
    main()
