"""Fixed overhead of one contraction launch: time vs K at a fixed grid (linear fit intercept)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import _lib as L, ops
ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda'); wp, wn = ws.data_ptr(), 1 << 30
def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cfg, splits in [(0, 8), (0, 16), (4, 4), (4, 8), (4, 16), (5, 4), (5, 8), (6, 2), (6, 4)]:
    L.knob_set("force_cfg", int(str(cfg))); L.knob_set("force_splits", int(str(splits)))
    xs, ys = [], []
    for cb in (32, 64, 128, 256, 512):
        g = ops._conv_geom(64, 16, 16, cb, 256, 5, 5, 2, 2, 'SAME')
        x = torch.randn(64, 16, 16, cb, device='cuda'); w = torch.randn(5, 5, cb, 256, device='cuda') * 0.05
        y = torch.empty(64, 8, 8, 256, device='cuda')
        st = L.stream()
        us = timeit(lambda: L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(x), L.ptr(w), None, L.ptr(y), wp, wn, st))
        slices = 25 * cb / 16 / splits
        xs.append(slices); ys.append(us)
    a, b = np.polyfit(xs, ys, 1)
    bm, bn = {0: (128, 128), 2: (64, 64), 4: (128, 128), 5: (128, 64), 6: (64, 64)}[cfg]
    tiles = (4096 // bm) * (256 // bn)
    print("cfg %d splits %2d blocks %4d: us vs slices/block %s -> %.2f us/slice, intercept %.1f us (incl. reduce pass)" % (
        cfg, splits, tiles * splits, ["%.0f:%.1f" % p for p in zip(xs, ys)], a, b), flush=True)
