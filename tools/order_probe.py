#!/usr/bin/env python
"""Tile order of the contraction kernels (knob "tile_order") per DCGAN-64 layer pass: time per launch with HIP
events, or -- under `rocprofv3 --pmc FETCH_SIZE --output-format csv` -- L2-to-fabric read bytes per launch.

    python tools/order_probe.py                       # times, heuristic plans
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/order_probe.py --reps 2
    python tools/order_probe.py --summarise out       # read MB per launch from the counter pass

Every (layer, pass, order) group of launches is preceded by one 3-element gan_fill launch, which the summary uses
as a separator.  NOTE: a layer pass timed alone, launch after launch, keeps its operands in the L2s between repetitions,
which flatters the XCD-grouped orders; the library's choice was made on whole-iteration timings (bench.py with the
tile_order knob)."""
import argparse
import csv
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAYERS = [("d_h1", (64, 32, 32, 64, 128, 5, 2)), ("d_h2", (64, 16, 16, 128, 256, 5, 2)), ("d_h3", (64, 8, 8, 256, 512, 5, 2)),
          ("d_h1x2", (128, 32, 32, 64, 128, 5, 2)), ("d_h2x2", (128, 16, 16, 128, 256, 5, 2)), ("d_h3x2", (128, 8, 8, 256, 512, 5, 2)),
          ("res3x3", (1, 64, 64, 256, 256, 3, 1)), ("cyc_d2", (1, 64, 64, 128, 256, 4, 2))]
PASSES = ("fwd", "dgrad", "wgrad")
ORDERS = (1, 2, 3, 4)
NAMES = {1: "launch order, k range fastest", 2: "by row tile (XCD)", 3: "by k range (XCD)", 4: "launch order, row tile fastest"}


def groups():
    return [(ln, ps, o) for ln, _ in LAYERS for ps in PASSES for o in ORDERS]


def summarise(path):
    files = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    disp = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], int(r["Grid_Size"]), 0.0])
        d[2] += float(r["Counter_Value"])
    rows = [disp[k] for k in sorted(disp)]
    out, cur = [], None
    for name, grid, kib in rows:
        if "fill_kernel" in name and grid <= 256:
            cur = []
            out.append(cur)
        elif cur is not None and "igemm" in name:
            cur.append(kib * 1024 * 2.0 / 1e6)          # gfx950: FETCH_SIZE counts half the bytes of 16-byte-per-lane reads
    out = [g for g in out]
    gs = groups()
    out = out[-len(gs):]
    print("| layer pass | " + " | ".join(NAMES[o] for o in ORDERS) + " |  (read MB per launch, L2 -> fabric)")
    print("|---|" + "---|" * len(ORDERS))
    for i in range(0, len(gs), len(ORDERS)):
        cells = []
        for j in range(len(ORDERS)):
            v = out[i + j]
            cells.append("%.1f" % (sum(v) / max(1, len(v))))
        print("| %s %s | %s |" % (gs[i][0], gs[i][1], " | ".join(cells)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--summarise", default=None)
    ap.add_argument("--cfg", type=int, default=-1, help="force this tile configuration (knob force_cfg)")
    ap.add_argument("--splits", type=int, default=0, help="force this split-K factor (knob force_splits)")
    args = ap.parse_args()
    if args.summarise:
        return summarise(args.summarise)
    import torch
    from gan_amd import _lib as L
    from gan_amd.ops import _conv_geom
    mark = torch.zeros(4, device='cuda')
    L.knob_set("force_cfg", args.cfg)
    L.knob_set("force_splits", args.splits)
    print("| layer pass | " + " | ".join(NAMES[o] for o in ORDERS) + " | chosen |  (us per launch incl. slab reduction)")
    print("|---|" + "---|" * (len(ORDERS) + 1))
    for ln, (n, h, w, cb, cs, k, s) in LAYERS:
        g = _conv_geom(n, h, w, cb, cs, k, k, s, s, 'SAME')
        gp = ctypes.byref(g)
        big = torch.randn(n, h, w, cb, device='cuda')
        small = torch.randn(n, g.hs, g.ws, cs, device='cuda')
        wt = torch.randn(k, k, cb, cs, device='cuda') * 0.05
        dw = torch.empty_like(wt)
        wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
        st = L.stream()

        def run(ps):
            if ps == "fwd":
                return L.lib.gan_conv2d_fwd(gp, L.ptr(big), L.ptr(wt), None, L.ptr(small), wp, wn, st)
            if ps == "dgrad":
                return L.lib.gan_conv2d_dgrad(gp, L.ptr(small), L.ptr(wt), L.ptr(big), wp, wn, st)
            return L.lib.gan_conv2d_wgrad(gp, L.ptr(big), L.ptr(small), L.ptr(dw), None, 0, wp, wn, st)
        for ps in PASSES:
            cells = []
            for o in ORDERS:
                L.knob_set("tile_order", o)
                L.check(L.lib.gan_fill(mark.data_ptr(), 3, 1.0, st))
                for _ in range(2):
                    L.check(run(ps))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run(ps)
                e1.record()
                torch.cuda.synchronize()
                cells.append(e0.elapsed_time(e1) / args.reps * 1e3)
            # the order the library takes by itself (its estimate, restated)
            image = 4.0 * n * (h * w * cb if ps == "fwd" else g.hs * g.ws * cs)
            filt = 4.0 * k * k * cb * cs
            chosen = 3 if ps == "wgrad" else (4 if ps == "dgrad" else (2 if image + 8 * filt <= 8 * image + filt else 3))
            print("| %s %s | %s | %s |" % (ln, ps, " | ".join("%.1f" % c for c in cells), NAMES[chosen]), flush=True)
        L.knob_set("tile_order", 0)


if __name__ == "__main__":
    main()
