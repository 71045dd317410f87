#!/usr/bin/env python
"""Contraction plans chosen by what the WHOLE iteration takes.

gan_amd.autotune times a layer pass alone, launch after launch, with its operands hot in the caches.  Inside a training
iteration the same pass meets a cold filter (Adam has just rewritten it), competes with the launches around it for the
memory system, and its split-K slabs are read back by a reduction that may carry a batch-norm reduction.  This tool
starts from a pinned plan file (or the tuner's choices) and walks over the (geometry, mode) pairs the iteration really issues, heaviest
first: for each it captures the iteration with a neighbouring plan (other split-K factors, the other bf16 tile
configurations, the four tile orders), replays it, and keeps the plan only if the iteration got faster twice in a row.
The result is written in gan_amd.autotune's format.

    python tools/insitu_tune.py [--model dcgan|t2i|cyclegan] [--plans FILE] --out gpurun_out/plans_insitu.json [--budget-s 420]
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from gan_amd import _lib, autotune, graph, ops  # noqa: E402
from tools.knob_ab import build  # noqa: E402

L = _lib.lib


def plan_of(mode, g):
    c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(L.gan_conv_plan_get(mode, ctypes.byref(g), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
    return c.value, s.value, thin.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plans", default=None, help="plan file to start from (default: profiles/r03_plans.json for dcgan, else "
                                                  "the plans gan_amd.autotune picks at start-up)")
    ap.add_argument("--model", default="dcgan", choices=["dcgan", "t2i", "cyclegan"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--budget-s", type=float, default=420.0)
    ap.add_argument("--min-gain", type=float, default=0.002)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    autotune.enable(True)
    plans = args.plans or (bench.PLANS_FILE if args.model == "dcgan" else None)
    if plans:
        autotune.load(plans)
    model, batch, make_runner = build(args.model, np.random.default_rng(1000))
    optims = [o for o in (getattr(model, n, None) for n in ("d_optim", "g_optim", "da_optim", "db_optim", "g_a2b_optim", "g_b2a_optim"))
              if o is not None]
    model.train_iteration(*batch)
    torch.cuda.synchronize()
    start = graph.model_state_snapshot(model.store, optims)

    # the (mode, geometry) pairs of one iteration with their share of the contraction time (eager, HIP events)
    with bench.ContractionTimer(_lib) as tm:
        model.train_iteration(*batch)
        _, _, _, per = tm.summary()
    pairs = {}
    for name, shape, _, e0, e1 in tm.records:
        g = _lib.ConvGeom()
        if name.startswith("gan_linear"):
            _lib.check(L.gan_linear_as_conv_geom(shape[0], shape[1], shape[2], ctypes.byref(g)))
        else:
            g = _lib.ConvGeom(*shape)
            if name == "gan_conv2d_dgrad_reflect":      # its contraction runs on the geometry of the reflect-padded image
                g.hb, g.wb, g.pt, g.pl = g.hb + 2 * g.pt, g.wb + 2 * g.pl, 0, 0
        key = (bench.ENTRY_MODE[name],) + tuple(g.key())
        pairs.setdefault(key, [g, 0.0])[1] += e0.elapsed_time(e1)
    order = sorted(pairs.items(), key=lambda kv: -kv[1][1])

    def measure():
        graph.model_state_restore(model.store, optims, start)
        ops._ws_cache.clear()
        model.train_iteration(*batch)          # eager: workspace sizes follow the plans
        torch.cuda.synchronize()
        graph.model_state_restore(model.store, optims, start)
        run = make_runner()
        best = 1e9
        for _ in range(3):
            run(*batch)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run(*batch)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        del run
        gc.collect()
        torch.cuda.empty_cache()
        return best

    t_begin = time.time()
    base = min(measure(), measure())
    print("start: %.4f ms/step, %d (mode, geometry) pairs" % (base, len(order)), flush=True)
    log = []
    for key, (g, share_ms) in order:
        if time.time() - t_begin > args.budget_s:
            print("time budget used up", flush=True)
            break
        mode = key[0]
        cfg_o, splits, thin = plan_of(mode, g)
        if thin or share_ms < 0.02:
            continue
        cfg, tile_order = cfg_o % autotune.ORDER_STRIDE, cfg_o // autotune.ORDER_STRIDE
        valid = set(autotune._candidates(mode, g))
        _lib.check(L.gan_conv_plan_set(mode, ctypes.byref(g), cfg_o, splits))     # (_candidates cleared the pin)
        sp = list(autotune.SPLITS)
        near = [s for s in sp if s != splits and abs(sp.index(s) - (sp.index(splits) if splits in sp else 0)) <= 2]
        cands = [(cfg, s, tile_order) for s in near if (cfg, s) in valid]
        for other in (7, 8, 4, 5):
            if other == cfg:
                continue
            # the same number of k ranges on the chip: a two-group tile needs half the slabs of a one-group tile
            k_ways = splits * (2 if cfg in (7, 8) else 1)
            s2 = max(1, k_ways // (2 if other in (7, 8) else 1))
            for s in (s2, s2 * 2 if other in (5, 8) and cfg in (4, 7) else s2):
                if (other, s) in valid and (other, s, tile_order) not in cands:
                    cands.append((other, s, tile_order))
        cands += [(cfg, splits, o) for o in (1, 2, 3, 4) if o != tile_order]
        best_plan, best_t = (cfg, splits, tile_order), base
        print("mode %d %s: %.3f ms of contraction time, plan cfg %d splits %d order %d, %d candidates"
              % (mode, "x".join(str(v) for v in g.key()), share_ms, cfg, splits, tile_order, len(cands)), flush=True)
        for c, s, o in cands:
            if time.time() - t_begin > args.budget_s:
                break
            if L.gan_conv_plan_set(mode, ctypes.byref(g), c + autotune.ORDER_STRIDE * o, s) != 0:
                continue
            t = measure()
            note = ""
            if t < best_t * (1.0 - args.min_gain):
                # confirm: the incumbent and the candidate once more each
                bp = best_plan
                _lib.check(L.gan_conv_plan_set(mode, ctypes.byref(g), bp[0] + autotune.ORDER_STRIDE * bp[2], bp[1]))
                t_inc = measure()
                _lib.check(L.gan_conv_plan_set(mode, ctypes.byref(g), c + autotune.ORDER_STRIDE * o, s))
                t2 = measure()
                if t2 < t_inc * (1.0 - args.min_gain):
                    best_plan, best_t = (c, s, o), min(t, t2)
                    note = "  <- kept (incumbent again %.4f, candidate again %.4f)" % (t_inc, t2)
                else:
                    note = "  (not confirmed: incumbent again %.4f, candidate again %.4f)" % (t_inc, t2)
            print("    cfg %d splits %3d order %d: %.4f ms/step%s" % (c, s, o, t, note), flush=True)
        _lib.check(L.gan_conv_plan_set(mode, ctypes.byref(g), best_plan[0] + autotune.ORDER_STRIDE * best_plan[2], best_plan[1]))
        if best_plan != (cfg, splits, tile_order):
            log.append({"mode": mode, "geom": list(g.key()), "from": [cfg, splits, tile_order], "to": list(best_plan),
                        "ms_before": base, "ms_after": best_t})
            base = best_t
            res = dict(autotune._done.get(g.key(), {}))
            old = res.get(mode, (0, 0, 0.0, 0.0))
            res[mode] = (best_plan[0] + autotune.ORDER_STRIDE * best_plan[2], best_plan[1], old[2], old[3])
            autotune._done[g.key()] = res
    final = min(measure(), measure())
    autotune.save(args.out)
    print(json.dumps({"ms_per_step_final": final, "changes": log}), flush=True)


if __name__ == "__main__":
    main()
