#!/usr/bin/env python
"""DCGAN-64 G+D training throughput on MI355X (the headline metric of BASELINE.json).

One step = one iteration of the reference's loop body (models/dcgan.py:262-275): one D update and
two G updates on a batch of 64 synthetic 64x64x3 images per GPU, fp32, inputs resident in HBM.
`python bench.py --gpus N --steps K --warmup W`; N > 1 runs one rank per GPU with the gradients
all-reduced over RCCL: either under torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the
environment) or, called plainly, by starting that job itself as a child process.  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16)
SPLIT_PRODUCTS = 6                 # bf16 part products per fp32 product in the default contraction arithmetic
BATCH = 64


def dcgan64_opts(seed):
    return argparse.Namespace(model='DCGAN', dataset='celebA', batch_size=BATCH, input_height=108, input_width=108,
                              output_height=64, output_width=64, y_dim=None, z_dim=100, gf_dim=64, df_dim=64,
                              gfc_dim=1024, dfc_dim=1024, c_dim=3, lr=0.0002, beta1=0.5, crop=True, nEpochs=1,
                              train_size=1e8, print_every=1, checkpoint_dir='./checkpoint', seed=seed)


# nominal FLOPs (2*MAC, every filter tap counted) of one contraction call, from its geometry
def conv_flops(g):
    return 2.0 * g.n * g.hs * g.ws * g.cs * g.kh * g.kw * g.cb


class ContractionTimer(object):
    """Brackets every conv / deconv / linear C-ABI call with HIP events on the launching stream
    (torch's current stream is the stream the library is given) and tallies nominal FLOPs."""

    CONV = ["gan_conv2d_fwd", "gan_conv2d_dgrad", "gan_conv2d_wgrad",
            "gan_deconv2d_fwd", "gan_deconv2d_dgrad", "gan_deconv2d_wgrad",
            "gan_conv2d_fwd_act", "gan_deconv2d_fwd_act", "gan_conv2d_dgrad_gated", "gan_deconv2d_dgrad_gated",
            "gan_conv2d_fwd_reflect", "gan_conv2d_wgrad_reflect", "gan_conv2d_dgrad_reflect"]
    LIN = ["gan_linear_fwd", "gan_linear_dgrad", "gan_linear_wgrad"]
    # entry points that run a contraction and let a batch-norm reduction ride on it: (op, geometry, ...); counted under
    # the name of the plain entry point of the same contraction
    RIDE = {"gan_conv_fwd_bn_stats": ("gan_conv2d_fwd", "gan_deconv2d_fwd", "gan_conv2d_fwd_reflect"),
            "gan_conv_dgrad_bn_sums": ("gan_conv2d_dgrad", "gan_deconv2d_dgrad", "gan_conv2d_dgrad_reflect"),
            "gan_conv_wgrad_colsum": ("gan_conv2d_wgrad", "gan_deconv2d_wgrad", "gan_conv2d_wgrad_reflect")}

    @classmethod
    def describe(cls, name, args):
        """(name of the plain entry point, shape tuple, nominal FLOP, algorithmic bytes) of one call."""
        if name in cls.LIN:
            rows, i, o = (int(v) for v in args[:3])
            return name, (rows, i, o), 2.0 * rows * i * o, 4.0 * (rows * i + rows * o + i * o)
        if name in cls.RIDE:
            name, g = cls.RIDE[name][int(args[0])], args[1]._obj
        else:
            g = args[0]._obj
        shape = (g.n, g.hb, g.wb, g.cb, g.hs, g.ws, g.cs, g.kh, g.kw, g.sh, g.sw, g.pt, g.pl)
        nbytes = 4.0 * (g.n * g.hb * g.wb * g.cb + g.n * g.hs * g.ws * g.cs + g.kh * g.kw * g.cb * g.cs)
        return name, shape, conv_flops(g), nbytes

    def __init__(self, L):
        self.L = L
        self.records = []
        self.saved = {}

    def __enter__(self):
        lib = self.L.lib
        for name in self.CONV + self.LIN + list(self.RIDE):
            fn = getattr(lib, name)
            self.saved[name] = fn
            setattr(lib, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.L.lib, name, fn)

    def _wrap(self, name, fn):
        def call(*args):
            plain, shape, flops, _ = self.describe(name, args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.records.append((plain, shape, flops, e0, e1))
            return rc
        return call

    def summary(self):
        torch.cuda.synchronize()
        per = {}
        tot_ms = tot_fl = 0.0
        for name, shape, flops, e0, e1 in self.records:
            ms = e0.elapsed_time(e1)
            k = "%s %s" % (name, "x".join(str(s) for s in shape))
            a = per.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += flops
            tot_ms += ms
            tot_fl += flops
        return tot_ms, tot_fl, len(self.records), per


class ContractionRecorder(ContractionTimer):
    """Records every conv / deconv / linear C-ABI call (function, arguments, nominal FLOPs) while letting
    it through, so that the same launches can be issued again on the same buffers."""

    def _wrap(self, name, fn):
        def call(*args):
            plain, shape, flops, nbytes = self.describe(name, args)
            # What is issued again, alone, is the CONTRACTION: a forward / input-gradient call that lets a batch-norm
            # reduction ride on its slab reduction is re-issued as the plain entry point of the same contraction (same
            # kernels, plan and slabs); the weight gradient that takes its bias gradient from given column sums as it is.
            again_fn, again_args = fn, args
            if name == "gan_conv_fwd_bn_stats":
                again_fn, again_args = self.saved[plain], args[1:6] + args[8:]
            elif name == "gan_conv_dgrad_bn_sums":
                again_fn, again_args = self.saved[plain], args[1:5] + args[7:]
            self.records.append((again_fn, again_args, flops, nbytes, plain, shape))
            return fn(*args)
        return call


def contraction_sequence_time(model, batch, L, replays=20):
    """Duration of the contraction launches of ONE iteration with nothing in between: the iteration is
    captured once (never replayed) with every contraction call recorded; its buffers stay reserved in that
    graph's memory pool, so the recorded calls are issued again, alone, into a second graph, which is
    replayed `replays` times between two HIP events.  Returns (ms per iteration, FLOP per iteration, calls,
    algorithmic bytes per iteration = every call's two operands and its output once)."""
    from gan_amd._lib import stream
    holder = torch.cuda.CUDAGraph()
    with ContractionRecorder(L) as rec:
        with torch.cuda.graph(holder):
            model.train_iteration(*batch)
    only = torch.cuda.CUDAGraph()
    with torch.cuda.graph(only):
        for fn, args, *_ in rec.records:
            rc = fn(*(tuple(args[:-1]) + (stream(),)))         # the stream is every entry point's last argument
            if rc != 0:
                raise SystemExit("re-issuing a recorded contraction call failed: %s" % L.lib.gan_last_error().decode())
    only.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        only.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / replays
    del only, holder
    return ms, sum(r[2] for r in rec.records), len(rec.records), sum(r[3] for r in rec.records)


def cpu_baseline(iters):
    """The numpy oracle's restatement of the same iteration on the host cores ("port", not TensorFlow)."""
    from oracle import autodiff as A
    from oracle import nets
    rng = np.random.default_rng(0)
    st = A.Store(seed=0)
    net = nets.DCGAN(st, batch_size=BATCH)
    im = rng.uniform(-1, 1, (BATCH, 64, 64, 3)).astype(np.float32)
    z = rng.uniform(-1, 1, (BATCH, 100)).astype(np.float32)
    net.build(im, z)
    t0 = time.time()
    for _ in range(iters):
        net.train_iteration(im, z)
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": round(BATCH * iters / dt, 3), "unit": "images/s", "cores": int(threads), "kind": "port",
            "sample": "%d iterations of the numpy oracle's DCGAN-64 step (batch 64, fp32, BLAS threads = cores), "
                      "CPU restatement, not TensorFlow" % iters}


PLANS_FILE = os.path.join(ROOT, "profiles", "r03_plans.json")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")

# contraction mode (0 forward, 1 transposed, 2 weight gradient) behind each C-ABI entry point
ENTRY_MODE = {"gan_conv2d_fwd": 0, "gan_conv2d_dgrad": 1, "gan_conv2d_wgrad": 2,
              "gan_deconv2d_fwd": 1, "gan_deconv2d_dgrad": 0, "gan_deconv2d_wgrad": 2,
              "gan_conv2d_fwd_act": 0, "gan_deconv2d_fwd_act": 1, "gan_conv2d_dgrad_gated": 1, "gan_deconv2d_dgrad_gated": 0,
              "gan_conv2d_fwd_reflect": 0, "gan_conv2d_wgrad_reflect": 2, "gan_conv2d_dgrad_reflect": 1,
              "gan_linear_fwd": 0, "gan_linear_dgrad": 1, "gan_linear_wgrad": 2}


def source_sha16():
    """Hash of the library's sources (gan_amd/csrc/*.hip, *.h, include/gan_hip.h): which kernels a number belongs to."""
    import glob
    import hashlib
    h = hashlib.sha1()
    files = sorted(glob.glob(os.path.join(ROOT, "gan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "gan_amd", "csrc", "*.h")))
    for f in files + [os.path.join(ROOT, "include", "gan_hip.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def plan_set_sha16(L, calls):
    """Hash of the contraction plans (tile configuration, split-K factor, thin path) the library runs for the given
    calls [(entry point name, geometry tuple)], asked from the library itself: two runs with the same hash launch
    the same kernels with the same split-K summation order."""
    import ctypes
    import hashlib
    plans = set()
    for name, shape in calls:
        g = L.ConvGeom()
        if name.startswith("gan_linear"):
            L.check(L.lib.gan_linear_as_conv_geom(shape[0], shape[1], shape[2], ctypes.byref(g)))
        else:
            g = L.ConvGeom(*shape)
        c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.check(L.lib.gan_conv_plan_get(ENTRY_MODE[name], ctypes.byref(g), ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
        plans.add((ENTRY_MODE[name],) + tuple(g.key()) + (c.value, sp.value, thin.value))
    return hashlib.sha1(json.dumps(sorted(plans)).encode()).hexdigest()[:16]


def committed_traffic(skip, plans_sha, src_sha):
    """HBM-side bytes of the contraction launches of one step, from the committed rocprofv3 counter passes
    (FETCH_SIZE with the gfx950 doubling + WRITE_SIZE, summed over the same launches `roofline.achieved` is timed
    on).  Counters cannot be read from inside this process, so the figure is a committed one; it is attached only
    when the committed passes ran the same plans (plan-set hash) on the same library sources (source hash) as this
    run, and is null with the reason otherwise."""
    if skip or not os.path.exists(TRAFFIC_FILE):
        return None, "no committed counter pass for this configuration"
    t = json.load(open(TRAFFIC_FILE))
    if t.get("plans_sha16") != plans_sha or t.get("source_sha16") != src_sha:
        return None, ("the committed counter passes (profiles/r03_traffic.md) ran plan set %s on library sources %s; this run "
                      "has plan set %s on sources %s" % (t.get("plans_sha16"), t.get("source_sha16"), plans_sha, src_sha))
    return (int(t["read_bytes_per_iteration"] + t["written_bytes_per_iteration"]),
            "profiles/r03_traffic.md (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE over tools/traffic_probe.py, same plan set "
            "and library sources as this run): %.2f GB read + %.2f GB written per step"
            % (t["read_bytes_per_iteration"] / 1e9, t["written_bytes_per_iteration"] / 1e9))


def exact_f32_step_ms(model, pool, _lib, steps=10):
    """The same step with every contraction on the fp32 MFMA (bitwise an fmaf chain per output element):
    a second captured iteration, timed like the first.  Workspace sizes follow the plans, so one eager
    iteration runs before the capture."""
    from gan_amd import ops
    _lib.lib.gan_set_exact_f32(1)
    try:
        ops._ws_cache.clear()
        model.train_iteration(*pool[0])
        run = model.iteration_runner(pool[0][0], pool[0][1])
        for i in range(3):
            run(*pool[i % len(pool)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            run(*pool[i % len(pool)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    finally:
        _lib.lib.gan_set_exact_f32(0)
        ops._ws_cache.clear()


def launch_ranks(n, backend):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a CHILD job
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), hand its one JSON line on and
    return its exit code.  Runs before anything in this process has touched the GPU, and replaces no
    process image: the parent only waits."""
    import socket
    import subprocess
    seen = torch.cuda.device_count()        # counts devices without initialising the runtime
    if backend == "nccl" and seen < n:
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this machine (RCCL needs one per rank; "
                         "--backend gloo rehearses several ranks on one GPU)\n" % (n, seen))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in child.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if child.returncode == 0 and not lines:
        sys.stderr.write("bench.py --gpus %d: the ranks exited cleanly but printed no JSON line\n" % n)
        return 1
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return child.returncode


def main():
    global BATCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying HIP graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="keep the library's heuristic contraction plans")
    ap.add_argument("--plans", default=PLANS_FILE,
                    help="JSON file of pinned contraction plans (gan_amd.autotune.save): loaded when it exists, so that every "
                         "run launches the same kernels; geometries it does not hold are tuned at start-up")
    ap.add_argument("--retune", action="store_true", help="ignore --plans and time the candidate plans at start-up")
    ap.add_argument("--save-plans", default=None, help="write the plans this run ended up with to this file")
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--dp-whole-graph", action="store_true",
                    help="data parallel: capture the all-reduces INTO one HIP graph instead of cutting the iteration "
                         "into a chain of graphs around them (RCCL under graph capture: not yet run with more than one rank)")
    ap.add_argument("--rehearse-dp", action="store_true",
                    help="one GPU only: run the data-parallel path (segmented graphs + RCCL all-reduce) on a one-rank group")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo only for rehearsing several ranks on ONE GPU (RCCL refuses that)")
    ap.add_argument("--exact-f32", action="store_true",
                    help="keep every contraction on the fp32 MFMA instead of the bf16 MFMA on split fp32 operands")
    ap.add_argument("--batch", type=int, default=BATCH,
                    help="images per GPU per step; 64 is the benchmark's configuration, other values are for exploration "
                         "and are named in config.workload")
    ap.add_argument("--detail", action="store_true", help="also print the per-call contraction timing table to stderr")
    ap.add_argument("--no-exact-f32-time", action="store_true",
                    help="skip the second captured iteration that times the step on the fp32 MFMA (exact_f32_ms_per_step)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only check the multi-rank launch: the ranks join the group, count each other with an "
                         "all-reduce and rank 0 prints {\"launch_check\": true, \"n_ranks_seen\": N}; no GPU work")
    args = ap.parse_args()
    nominal_batch, BATCH = BATCH, args.batch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, args.backend))

    # Libraries underneath (RCCL's version banner) write to stdout; the contract is ONE JSON line
    # there, so everything before the final print goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    if args.launch_check:
        import torch.distributed as dist
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            os.write(json_fd, (json.dumps({"launch_check": True, "n_gpus": args.gpus, "n_ranks_seen": seen}) + "\n").encode())
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and there is no CPU path")
    if args.backend == "gloo":
        local_rank = 0                      # rehearsal: every rank on the one visible GPU
    torch.cuda.set_device(local_rank)
    dist = None
    pg = None
    if world > 1 or args.rehearse_dp:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
            pg = dist.group.WORLD
        elif args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gan_amd import _lib, autotune
    from gan_amd.dcgan import DCGAN
    if args.exact_f32:
        _lib.lib.gan_set_exact_f32(1)
    autotune.enable(not args.no_autotune)   # plans are tuned during model construction (eager), before capture
    plans_loaded = False
    if not args.no_autotune and not args.retune and not args.exact_f32 and BATCH == nominal_batch and os.path.exists(args.plans):
        autotune.load(args.plans)
        plans_loaded = True

    model = DCGAN(None, dcgan64_opts(seed=0), world_size=world, process_group=pg, load_data=False)
    # synthetic CelebA-shaped data, per-rank seed, resident in HBM before the timed region
    rng = np.random.default_rng(1000 + rank)
    pool = [(torch.from_numpy(rng.uniform(-1, 1, (BATCH, 64, 64, 3)).astype(np.float32)).cuda(),
             torch.from_numpy(rng.uniform(-1, 1, (BATCH, 100)).astype(np.float32)).cuda()) for _ in range(4)]

    if args.no_graph:
        run = lambda im, z, y=None: model.train_iteration(im, z, y)
    else:
        run = model.iteration_runner(pool[0][0], pool[0][1], segmented=False if args.dp_whole_graph else None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run(*pool[i % len(pool)])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = run(*pool[i % len(pool)])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if args.backend == "gloo" else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = {k: float(v) for k, v in out.items()}
    if not all(np.isfinite(list(losses.values()))):
        raise SystemExit("non-finite loss in the timed region: %s" % losses)

    # roofline of the dominant kernel family (the MFMA implicit-GEMM contractions), measured live with
    # HIP events around every contraction call of eager iterations (graph replay hides single launches)
    roof = None
    alg_bytes = None
    timer_iters = 3
    with ContractionTimer(_lib) as tm:
        for i in range(timer_iters):
            model.train_iteration(*pool[i % len(pool)])
        ms, flops, calls, per = tm.summary()
    eager_ms, eager_achieved = ms / timer_iters, flops / (ms * 1e-3) / 1e12
    method = "HIP events around every contraction call of %d eager iterations" % timer_iters
    if world == 1 and pg is None:
        # one GPU: the same launches replayed back to back as a HIP graph (no host dispatch gap between a
        # contraction kernel and its slab reduction), which is how they run in the timed region
        seq_ms, seq_flops, seq_calls, alg_bytes = contraction_sequence_time(model, pool[0], _lib)
        ms, flops, calls, timer_iters = seq_ms, seq_flops, seq_calls, 1
        method = ("HIP events around 20 replays of a HIP graph holding only the %d contraction calls of one iteration, "
                  "on the buffers of a captured iteration" % seq_calls)
    achieved = flops / (ms * 1e-3) / 1e12
    exact = bool(_lib.lib.gan_get_exact_f32())
    plans_sha = plan_set_sha16(_lib, sorted(set((name, shape) for name, shape, _, _, _ in tm.records)))
    src_sha = source_sha16()
    if args.save_plans and rank == 0:
        autotune.save(args.save_plans)
    traffic, traffic_note = committed_traffic(exact or BATCH != nominal_batch or alg_bytes is None, plans_sha, src_sha)
    # `achieved` counts the ALGORITHMIC fp32 FLOP of the convolutions.  The default arithmetic forms every
    # fp32 product from six bf16 x bf16 part products on the bf16 matrix pipe, so the MFMA ceiling for
    # algorithmic FLOP is the bf16 peak / 6; with --exact-f32 it is the fp32 MFMA peak.
    peak = FP32_MFMA_PEAK_TFLOPS if exact else BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes per step, all contraction launches",
            "traffic_source": traffic_note,
            "algorithmic_bytes": alg_bytes,
            "kernel": ("igemm_kernel" if exact else "igemm_split_kernel") +
                      " (+ reduce_slabs_kernel) over the %d contraction calls of one iteration" % (calls // timer_iters),
            "arithmetic": ("fp32 MFMA" if exact else
                           "fp32 operands split exactly into 3 bf16 parts, 6 of the 9 part products on the bf16 MFMA, "
                           "fp32 accumulation (error at the fp32 rounding level); peak = bf16 dense peak / 6"),
            "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "flop_per_step": round(flops / timer_iters / 1e9, 2), "kernel_ms_per_step": round(ms / timer_iters, 3),
            "method": method, "eager_event_ms_per_step": round(eager_ms, 3), "eager_event_achieved": round(eager_achieved, 2)}
    if args.detail and rank == 0:
        for k, (n, t_ms, fl) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            sys.stderr.write("%-70s calls %3d  %8.3f ms/iter  %7.2f TFLOP/s\n"
                             % (k, n // timer_iters, t_ms / timer_iters, fl / (t_ms * 1e-3) / 1e12))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        line = {
            "metric": "DCGAN-64 G+D train images/sec", "value": round(BATCH * world * args.steps / elapsed, 2),
            "unit": "images/s", "n_gpus": world, "n_ranks_seen": (dist.get_world_size() if dist is not None else 1),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DCGAN 64x64x3 CelebA-shaped synthetic images, batch %d per GPU%s, fp32, "
                                   % (BATCH, "" if BATCH == nominal_batch else " (NOT the benchmark's batch of %d)" % nominal_batch) +
                                   "1 D update + 2 G updates per step (models/dcgan.py:262-275)",
                       "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                       "launch": "eager" if args.no_graph else ("hipGraph" if world == 1 and pg is None
                                                               else "one hipGraph holding the %s all-reduces" % args.backend if args.dp_whole_graph
                                                               else "hipGraph segments + %s all-reduce overlapped with the backward pass"
                                                               % ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)")),
                       "schedule": "the first G update reuses the D update's generator pass (same z, same weights; "
                                   "trained weights identical to re-running it, batch-norm moving variances equal to float32 rounding)",
                       "contraction_plans": ("heuristic" if args.no_autotune else
                                             "pinned from %s" % os.path.relpath(args.plans, ROOT) if plans_loaded else
                                             "autotuned at start-up"),
                       "plans_sha16": plans_sha, "library_source_sha16": src_sha},
            "roofline": roof,
            "mfma_util_step": round(roof["flop_per_step"] * 1e9 / (ms_per_step * 1e-3) / 1e12 / peak, 4),
            "losses_last_step": {k: round(v, 5) for k, v in losses.items()},
        }
        if world == 1 and pg is None and not args.exact_f32 and not args.no_graph and not args.no_exact_f32_time:
            line["exact_f32_ms_per_step"] = round(exact_f32_step_ms(model, pool, _lib), 4)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_iters)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
