"""Per-geometry tuning of the contraction kernels (tile configuration x split-K factor).

The library picks a plan for every conv / deconv / linear pass with a heuristic.  When tuning is
enabled, the first eager use of a geometry times the candidate plans of its three contraction
modes on scratch tensors of the real shapes (HIP events on the current stream) and pins the
fastest through gan_conv_plan_set.  Nothing is tuned during graph capture: the eager warm-up runs
that precede a capture see every shape first.

Results stay deterministic for a fixed plan; two processes may pin different plans (timing noise),
which changes the summation order of split-K partial sums, i.e. the last bits of a result.
Data-parallel replicas stay identical regardless, because every rank applies the same all-reduced
gradient.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvGeom, check, lib

_enabled = os.environ.get("GAN_HIP_AUTOTUNE", "0") not in ("", "0")
_done = {}          # geometry key -> {mode: (cfg, splits, us, heuristic_us)}
_listeners = []     # callables run after plans changed (workspace size caches)

MODES = (0, 1, 2)
SPLITS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 48, 64, 96, 128, 192, 256)
TILE = ((128, 128), (128, 64), (64, 64), (128, 32), (128, 128), (128, 64), (64, 64),    # 4..6: split-bf16 MFMA
        (128, 128), (128, 64),                                                            # 7..8: + two k groups per tile
        (128, 32))                                                                        # 9: split-bf16, thin outputs
FIRST_SPLIT_CFG = 4
MAX_SLAB_BYTES = 256 << 20
MIN_GAIN = 0.03      # a candidate must beat the heuristic by this fraction to replace it
FINALISTS = 5        # candidates of the quick scan that are timed again
FINAL_REPS = 12
ORDER_STRIDE = 16    # plan_set's cfg = tile configuration + 16 * tile order (0: the library's choice).  The tuner does
                     # not time tile orders: a layer pass timed alone, launch after launch, keeps its operands in the
                     # L2s from one repetition to the next, which favours the XCD-grouped orders by 8-15 % where a
                     # whole training iteration measures no gain or a loss.


def enable(flag=True):
    global _enabled
    _enabled = bool(flag)


def enabled():
    return _enabled


def on_change(fn):
    _listeners.append(fn)


def report():
    """{geometry key: {mode: (cfg, splits, tuned us, heuristic us)}} of everything tuned so far."""
    return dict(_done)


def _problem(mode, g):
    """(M, N, K, output elements) of the contraction, as the library counts them."""
    if mode == 0:
        return g.n * g.hs * g.ws, g.cs, g.kh * g.kw * g.cb, g.n * g.hs * g.ws * g.cs
    if mode == 2:
        return g.kh * g.kw * g.cb, g.cs, g.n * g.hs * g.ws, g.kh * g.kw * g.cb * g.cs
    ncls = g.sh * g.sw
    m = g.n * ((g.hb + g.sh - 1) // g.sh) * ((g.wb + g.sw - 1) // g.sw)
    k = ((g.kh + g.sh - 1) // g.sh) * ((g.kw + g.sw - 1) // g.sw) * g.cs
    return m * ncls, g.cb, k, g.n * g.hb * g.wb * g.cb


def _candidates(mode, g):
    m, n, k, out = _problem(mode, g)
    nkt = (k + 15) // 16
    cands = []
    # the library knows which geometries its bf16 kernels take: it refuses to pin one of their configurations otherwise
    gp = ctypes.byref(g)
    split_ok = not lib.gan_get_exact_f32() and lib.gan_conv_plan_set(mode, gp, FIRST_SPLIT_CFG, 1) == 0
    check(lib.gan_conv_plan_set(mode, gp, -1, 1))
    for cfg, (bm, bn) in enumerate(TILE):
        if bn >= 4 * n and bn > 32:
            continue                                  # tile mostly padding in N
        if cfg >= FIRST_SPLIT_CFG and not split_ok:
            continue
        if cfg == 9 and mode == 1:
            continue                                  # the 32-wide bf16 tile has no transposed mode
        tiles = ((m + bm - 1) // bm) * ((n + bn - 1) // bn)
        for s in SPLITS:
            if s > 1 and (nkt // (s * (2 if cfg in (7, 8) else 1)) < 2 or s * out * 4 > MAX_SLAB_BYTES):
                continue
            if tiles * s > 16384 and s > 1:
                continue
            cands.append((cfg, s))
    return cands


def _time(fn, reps):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def _runner(mode, g, bufs):
    big, small, w, dw = bufs
    gp = ctypes.byref(g)
    st = _lib.stream()

    def run():
        wp, wn = _lib.workspace(int(lib.gan_conv_workspace_bytes(gp)))
        if mode == 0:
            check(lib.gan_conv2d_fwd(gp, _lib.ptr(big), _lib.ptr(w), None, _lib.ptr(small), wp, wn, st))
        elif mode == 1:
            check(lib.gan_conv2d_dgrad(gp, _lib.ptr(small), _lib.ptr(w), _lib.ptr(big), wp, wn, st))
        else:
            check(lib.gan_conv2d_wgrad(gp, _lib.ptr(big), _lib.ptr(small), _lib.ptr(dw), None, 0, wp, wn, st))
    return run


def tune_geometry(g, reps=3, verbose=False):
    """Tune the three contraction modes of one geometry (no-op when already done)."""
    key = g.key()
    if key in _done:
        return _done[key]
    if torch.cuda.is_current_stream_capturing():
        return None
    res = {}
    big = torch.randn(g.n, g.hb, g.wb, g.cb, device='cuda')
    small = torch.randn(g.n, g.hs, g.ws, g.cs, device='cuda')
    w = torch.randn(g.kh, g.kw, g.cb, g.cs, device='cuda') * 0.05
    dw = torch.empty_like(w)
    gp = ctypes.byref(g)
    for mode in MODES:
        c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.gan_conv_plan_set(mode, gp, -1, 1))
        check(lib.gan_conv_plan_get(mode, gp, ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
        run = _runner(mode, g, (big, small, w, dw))
        heur = (c.value, s.value)
        base = min(_time(run, reps), _time(run, reps))
        best = (heur[0], heur[1], base)
        if not thin.value:
            # stage 1: a quick scan of every candidate; stage 2: the few fastest and the heuristic plan are
            # timed again with more repetitions (short timings of ~50 us kernels are noisy), best of two
            scan = []
            for cfg, splits in _candidates(mode, g):
                if (cfg, splits) == heur:
                    continue
                check(lib.gan_conv_plan_set(mode, gp, cfg, splits))
                scan.append((_time(run, reps), cfg, splits))
            scan.sort()
            finalists = [(cfg, splits) for _, cfg, splits in scan[:FINALISTS]]
            check(lib.gan_conv_plan_set(mode, gp, -1, 1))
            base = min(base, _time(run, FINAL_REPS), _time(run, FINAL_REPS))
            best = (heur[0], heur[1], base)
            for cfg, splits in finalists:
                check(lib.gan_conv_plan_set(mode, gp, cfg, splits))
                us = min(_time(run, FINAL_REPS), _time(run, FINAL_REPS))
                if us < best[2]:
                    best = (cfg, splits, us)
            if not (best[:2] != heur and best[2] < base * (1.0 - MIN_GAIN)):
                best = (heur[0], heur[1], base)
            if best[:2] != heur:
                check(lib.gan_conv_plan_set(mode, gp, best[0], best[1]))
            else:
                check(lib.gan_conv_plan_set(mode, gp, -1, 1))
        res[mode] = (best[0], best[1], best[2], base)
        if verbose:
            print("autotune %s mode %d: cfg %d splits %d %.1f us (heuristic cfg %d splits %d %.1f us)"
                  % (key, mode, best[0], best[1], best[2], c.value, s.value, base))
    _done[key] = res
    for fn in _listeners:
        fn()
    return res


def tune_linear(rows, in_dim, out_dim, **kw):
    g = ConvGeom()
    check(lib.gan_linear_as_conv_geom(rows, in_dim, out_dim, ctypes.byref(g)))
    return tune_geometry(g, **kw)


def save(path):
    """Write the tuned plans as JSON: a later process pins the same plans with load() instead of timing again
    (and then sums split-K partial results in the same order, i.e. reproduces this process bit for bit)."""
    import json
    with open(path, "w") as f:
        json.dump([{"geom": list(k), "plans": {str(m): list(v) for m, v in res.items()}} for k, res in _done.items()], f)


def load(path):
    """Pin the plans written by save().  Geometries not in the file are tuned on first use as usual."""
    import json
    with open(path) as f:
        rows = json.load(f)
    for row in rows:
        g = ConvGeom(*row["geom"])
        res = {}
        for m, (cfg, splits, us, heur_us) in row["plans"].items():
            c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib.gan_conv_plan_set(int(m), ctypes.byref(g), -1, 1))
            check(lib.gan_conv_plan_get(int(m), ctypes.byref(g), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
            if not thin.value and (c.value, s.value) != (cfg, splits):
                check(lib.gan_conv_plan_set(int(m), ctypes.byref(g), int(cfg), int(splits)))
            res[int(m)] = (int(cfg), int(splits), float(us), float(heur_us))
        _done[g.key()] = res
    for fn in _listeners:
        fn()


def reset():
    """Forget every pinned plan."""
    lib.gan_conv_plan_clear()
    _done.clear()
    for fn in _listeners:
        fn()
