"""One rank of a data-parallel test job (started by test_a_dp.py; not a test module itself).

Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment like a torch.distributed.run
worker, joins a gloo group (gloo moves CPU tensors here and GPU tensors on the one-GPU test box, where
two ranks share cuda:0 and RCCL would refuse the duplicate device) and writes what it computed to
<out>/rank<r>.npz for the parent to check.

  cpu  : host-side protocol only -- VariableStore / Group arenas on the CPU, broadcast, gradient
         all-reduce through AdamOptimizer.all_reduce.
  gpu  : the real DCGAN data-parallel iteration on cuda:0 (eager, then the segmented HIP-graph
         runner from the same start state).
  train: DCGAN.train over an in-memory data set whose items carry their index: which items each rank
         consumed, who printed / sampled / saved, and the trained variables.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_cpu(out, rank, world):
    from gan_amd import variables as V
    from gan_amd.optim import AdamOptimizer
    st = V.VariableStore(device='cpu', seed=100 + rank)          # ranks start from DIFFERENT weights
    with st.variable_scope("discriminator"):
        with st.variable_scope("d_h0_conv"):
            st.get_variable('w', [5, 5, 3, 8], V.truncated_normal_initializer(0.02))
            st.get_variable('biases', [8], V.constant_initializer(0.0))
        with st.variable_scope("d_bn1"):
            st.get_variable('beta', [8], V.constant_initializer(0.0))
            st.get_variable('gamma', [8], V.random_normal_initializer(1.0))
            st.get_variable('moving_mean', [8], V.constant_initializer(0.0), trainable=False)
    grp = st.pack('d', st.select('d_'))
    before = grp.values.clone()
    grp.broadcast(None)
    opt = AdamOptimizer(2e-4, beta1=0.5).attach(grp)
    rng = np.random.default_rng(7 + rank)
    local = {}
    for v in grp.variables:
        g = rng.standard_normal(v.shape).astype(np.float32)
        v.grad.copy_(torch.from_numpy(g))
        local[v.name] = g
    opt.all_reduce(world, None)
    res = {"values_before": before.numpy(), "values_after": grp.values.numpy(), "grads_flat": grp.grads.numpy(),
           "numel": np.array([grp.numel]), "moving_in_group": np.array([int(st.vars['discriminator/d_bn1/moving_mean'].group is not None)])}
    for n, g in local.items():
        res["local/" + n] = g
    for v in grp.variables:
        res["reduced/" + v.name] = v.grad.numpy().copy()
    # the same exchange in two parts: the tail of the arena (from d_bn1/beta on) first, the rest later
    grp.split_early('discriminator/d_bn1/beta')
    for v in grp.variables:
        v.grad.copy_(torch.from_numpy(local[v.name]))
        v.grad_fresh = False
    opt.all_reduce_early(world, None)
    if opt._early is not True:
        opt._early.wait()                   # the first part is exchanged asynchronously: look at it once it is done
    res["early_flags"] = np.array([int(v.early) for v in grp.variables])
    res["after_early_flat"] = grp.grads.numpy().copy()
    res["early_offset"] = np.array([grp.early_offset])
    opt.all_reduce(world, None)
    for v in grp.variables:
        res["reduced2/" + v.name] = v.grad.numpy().copy()
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)


def mark(msg):
    sys.stdout.write("[rank %s] %s\n" % (os.environ.get("RANK"), msg))
    sys.stdout.flush()


def run_gpu(out, rank, world):
    from test_gpu_dcgan import make_opts
    from gan_amd.dcgan import DCGAN
    # gloo: every rank on the one visible GPU; nccl (RCCL): one GPU per rank
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("DP_BACKEND") == "nccl" else 0)
    data = np.load(os.path.join(out, "inputs.npz"))
    weights = {k[2:]: data[k] for k in data.files if k.startswith("w/")}
    opts = make_opts(seed=rank)                                  # different initial weights per rank
    res = {}

    def fresh():
        m = DCGAN(None, opts, world_size=world, load_data=False)
        return m

    model = fresh()
    mark("model built")
    res["bcast/g_h0_lin"] = model.store.vars['generator/g_h0_lin/Matrix'].data.cpu().numpy()
    model.store.load_state_dict(weights)
    iters = int(data["iters"])
    im = [torch.from_numpy(data["im/%d/%d" % (rank, i)]).cuda() for i in range(iters)]
    z = [torch.from_numpy(data["z/%d/%d" % (rank, i)]).cuda() for i in range(iters)]
    # one D update on its own first: the all-reduced gradient arena and the post-Adam weights
    probe = fresh()
    probe.store.load_state_dict(weights)
    probe.d_update(im[0], z[0])
    torch.cuda.synchronize()
    mark("probe D update done")
    for v in probe.d_group.variables:
        res["dgrad/" + v.name] = v.grad.cpu().numpy()
        res["dpost/" + v.name] = v.data.cpu().numpy()
    losses = []
    for i in range(iters):
        o = model.train_iteration(im[i], z[i])
        losses.append([float(o[k]) for k in ('d_loss_real', 'd_loss_fake', 'g_loss_1', 'g_loss_2')])
    res["losses_eager"] = np.array(losses)
    mark("eager iterations done")
    for n, v in model.store.vars.items():
        res["final/" + n] = v.data.cpu().numpy()
    res["steps"] = np.array([model.d_optim.steps_taken, model.g_optim.steps_taken])
    # the segmented HIP-graph runner from the same start state
    seg = fresh()
    seg.store.load_state_dict(weights)
    run = seg.iteration_runner(im[0], z[0])
    mark("segments captured")
    losses = []
    for i in range(iters):
        o = run(im[i], z[i])
        torch.cuda.synchronize()
        losses.append([float(o[k]) for k in ('d_loss_real', 'd_loss_fake', 'g_loss_1', 'g_loss_2')])
    res["losses_graph"] = np.array(losses)
    mark("graph iterations done")
    for n, v in seg.store.vars.items():
        res["final_graph/" + n] = v.data.cpu().numpy()
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)


def run_train(out, rank, world):
    from test_gpu_dcgan import make_opts
    from gan_amd.dcgan import DCGAN
    torch.cuda.set_device(0)
    opts = make_opts(seed=rank, checkpoint_dir=os.path.join(out, "ck"), sample_dir=os.path.join(out, "smp"),
                     print_every=2, use_graph=True)
    model = DCGAN(None, opts, world_size=world, load_data=False)
    n = 5 * opts.batch_size * world + 3
    data_X = np.random.default_rng(1).uniform(-1, 1, (n, 16, 16, 3)).astype(np.float32)
    data_X[:, 0, 0, 0] = np.arange(n) / float(n)               # item i says so in its first pixel
    seen = []
    make_runner = model.iteration_runner

    def recording_runner(images, z, y=None):
        run = make_runner(images, z, y)

        def wrapped(im, zz, yy=None):
            seen.append(np.round(im[:, 0, 0, 0].cpu().numpy() * n).astype(np.int64))
            return run(im, zz, yy)
        return wrapped
    model.iteration_runner = recording_runner
    model.train(data_X=data_X, sample_every=3)
    res = {"seen": np.array(seen), "counter": np.array([model.counter]),
           "steps": np.array([model.d_optim.steps_taken, model.g_optim.steps_taken])}
    for name, v in model.store.vars.items():
        res["final/" + name] = v.data.cpu().numpy()
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("DP_WORKER_STUCK_AFTER", "240")), exit=True)   # a stuck rank says where
    mark("joining the group")
    backend = os.environ.get("DP_BACKEND", "gloo")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    mark("joined")
    try:
        {"cpu": run_cpu, "gpu": run_gpu, "train": run_train}[mode](out, rank, world)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
