"""DCGAN on the GPU against the numpy oracle: identical seeded weights and inputs, per-update
gradients, the 1 D + 2 G schedule, Adam bookkeeping, loss trajectories, HIP-graph replay."""
import argparse

import numpy as np
import pytest
import torch

from oracle import autodiff as A
from oracle import nets

pytestmark = pytest.mark.gpu


def make_opts(**kw):
    d = dict(model='DCGAN', dataset='synthetic', batch_size=8, input_height=16, input_width=16, output_height=16,
             output_width=16, y_dim=None, z_dim=10, gf_dim=8, df_dim=8, gfc_dim=32, dfc_dim=32, c_dim=3, lr=0.0002,
             beta1=0.5, crop=True, nEpochs=1, train_size=1e8, print_every=1, checkpoint_dir='./checkpoint', seed=0)
    d.update(kw)
    return argparse.Namespace(**d)


def build_pair(opts, seed=11):
    """An oracle model and a GPU model holding bit-identical weights."""
    from gan_amd.dcgan import DCGAN
    rng = np.random.default_rng(seed)
    st = A.Store(seed=seed)
    ref = nets.DCGAN(st, batch_size=opts.batch_size, output_height=opts.output_height, output_width=opts.output_width,
                     c_dim=opts.c_dim, y_dim=opts.y_dim, z_dim=opts.z_dim, gf_dim=opts.gf_dim, df_dim=opts.df_dim,
                     gfc_dim=opts.gfc_dim, dfc_dim=opts.dfc_dim, lr=opts.lr, beta1=opts.beta1)
    shape = (opts.batch_size, opts.output_height, opts.output_width, opts.c_dim)
    y0 = None if not opts.y_dim else np.zeros((opts.batch_size, opts.y_dim), np.float32)
    ref.build(np.zeros(shape, np.float32), np.zeros((opts.batch_size, opts.z_dim), np.float32), y0)
    # perturb the BN affine parameters and biases so that nothing is checked at a trivial value
    for n in st.trainable:
        if n.endswith(('gamma', 'beta', 'biases', 'bias')):
            st.values[n] = (st.values[n] + rng.standard_normal(st.values[n].shape) * 0.1).astype(np.float32)
    gpu = DCGAN(None, opts, load_data=False)
    assert list(gpu.store.vars) == list(st.values), "variable names / creation order differ from the oracle"
    gpu.store.load_state_dict(st.values)
    return ref, gpu, rng


def batch(opts, rng, lo=-1.0):
    images = rng.uniform(lo, 1, (opts.batch_size, opts.output_height, opts.output_width, opts.c_dim)).astype(np.float32)
    z = rng.uniform(-1, 1, (opts.batch_size, opts.z_dim)).astype(np.float32)
    y = None
    if opts.y_dim:
        y = np.eye(opts.y_dim, dtype=np.float32)[rng.integers(0, opts.y_dim, opts.batch_size)]
    return images, z, y


def to_dev(*arrs):
    return [None if a is None else torch.from_numpy(a).cuda() for a in arrs]


def grads_by_name(group):
    return {v.name: v.grad.detach().cpu().numpy() for v in group.variables}


@pytest.mark.parametrize("cond", [False, True])
def test_update_gradients_match_oracle(cond):
    from gpu_util import assert_close, check_group_grads, is_noise_grad
    opts = make_opts(y_dim=4, c_dim=1, output_height=12, output_width=12, input_height=12) if cond else make_opts()
    ref, gpu, rng = build_pair(opts)
    images, z, y = batch(opts, rng, lo=0.0 if cond else -1.0)
    # oracle gradients of d_loss w.r.t. d_vars and g_loss w.r.t. g_vars at the same weights
    before = {n: v.copy() for n, v in ref.st.values.items()}
    tp = A.Tape(ref.st)
    lr_, lf_ = ref.d_losses(tp, images, z, y)
    tp.backward(A.add(tp, lr_, lf_))
    d_ref = tp.grads(ref.d_vars)
    di, dz, dy = to_dev(images, z, y)
    d_real, d_fake = gpu.d_update(di, dz, dy)
    torch.cuda.synchronize()
    assert_close(d_real.cpu().numpy(), [float(lr_.v)], "d_loss_real")
    assert_close(d_fake.cpu().numpy(), [float(lf_.v)], "d_loss_fake")
    check_group_grads(gpu.d_group, d_ref, 2e-4)
    # after the Adam step the weights must equal the oracle's Adam step from the same gradients
    ref.st.values.update(before)
    for n in list(ref.st.values):           # moving averages moved during the oracle forward: reset
        ref.st.values[n] = before[n]
    ref.d_update(images, z, y)
    for v in gpu.d_group.variables:
        if is_noise_grad(v.name, d_ref):
            continue        # Adam normalises pure rounding noise to a +-lr step: not comparable
        # Adam's first step is lr * g / (|g| + eps): where |g| is within a few hundred eps = 1e-8 of zero the step
        # follows the float32 rounding of g, so the 1e-4 bar holds the elements whose gradient is significant
        # (>= 1e-4 of the variable's largest) and every element stays within the two steps a sign flip can cost
        got, want, gref = v.data.cpu().numpy(), ref.st.values[v.name], np.abs(d_ref[v.name])
        sig = (gref >= 1e-4 * gref.max()) | (gref == 0)       # (taps that only ever see padding: no gradient, no step)
        assert sig.mean() > 0.9, "post-Adam %s: only %.0f %% of the elements are compared" % (v.name, 100 * sig.mean())
        assert_close(got[sig], want[sig], "post-Adam " + v.name, tol=1e-4, elementwise=False)
        assert np.abs(got - want).max() <= 2.01 * gpu.d_optim.lr, "post-Adam " + v.name
    for n, v in gpu.store.vars.items():
        if 'moving_' in n:
            # a moving mean near zero is compared on the scale of the matching standard deviation
            assert_close(v.data.cpu().numpy(), ref.st.values[n], n, tol=1e-3, elementwise=False)
    assert gpu.d_optim.steps_taken == 1 and gpu.g_optim.steps_taken == 0
    # G update from the now-identical state
    gpu.store.load_state_dict(ref.st.values)
    tp = A.Tape(ref.st)
    yn = None if y is None else tp.const(y)
    gl = A.sigmoid_ce_mean(tp, ref.discriminator(tp, ref.generator(tp, tp.const(z), yn), yn), 1.0)
    tp.backward(gl)
    g_ref = tp.grads(ref.g_vars)
    g_loss = gpu.g_update(dz, dy)
    torch.cuda.synchronize()
    assert_close(g_loss.cpu().numpy(), [float(gl.v)], "g_loss")
    check_group_grads(gpu.g_group, g_ref, 2e-4)


def test_schedule_and_loss_trajectory_200_steps():
    """200 iterations of the reference schedule from identical weights, fresh batch every step.
    Training amplifies rounding differences (Adam turns near-zero gradients into +-lr steps, relu
    kinks flip), so the yardstick is the float64 oracle and the fp32 numpy oracle's own drift from it."""
    opts = make_opts()
    ref, gpu, rng = build_pair(opts, seed=5)
    # a float64 copy of the oracle from the same weights: the yardstick for how far fp32 rounding
    # alone moves this trajectory
    st64 = A.Store(seed=5, dtype=np.float64)
    ref64 = nets.DCGAN(st64, batch_size=opts.batch_size, output_height=opts.output_height,
                       output_width=opts.output_width, c_dim=opts.c_dim, z_dim=opts.z_dim, gf_dim=opts.gf_dim,
                       df_dim=opts.df_dim, lr=opts.lr, beta1=opts.beta1)
    shape = (opts.batch_size, opts.output_height, opts.output_width, opts.c_dim)
    ref64.build(np.zeros(shape), np.zeros((opts.batch_size, opts.z_dim)))
    for n in st64.values:
        st64.values[n] = ref.st.values[n].astype(np.float64)
    dev_gpu = np.zeros(200)
    dev_f32 = np.zeros(200)
    for it in range(200):
        images, z, y = batch(opts, rng)
        truth = ref64.train_iteration(images.astype(np.float64), z.astype(np.float64))
        r = ref.train_iteration(images, z, y)
        out = gpu.train_iteration(*to_dev(images, z, y))
        got = {'d_loss': float(out['d_loss_real']) + float(out['d_loss_fake']),
               'g_loss_1': float(out['g_loss_1']), 'g_loss_2': float(out['g_loss_2'])}
        assert np.isfinite(list(got.values())).all()
        dev_gpu[it] = max(abs(got[k] - truth[k]) for k in got)
        dev_f32[it] = max(abs(r[k] - truth[k]) for k in got)
    print("max |loss - float64 oracle| over 20 / 200 steps: gpu %.3e / %.3e, float32 numpy oracle %.3e / %.3e"
          % (dev_gpu[:20].max(), dev_gpu.max(), dev_f32[:20].max(), dev_f32.max()))
    assert gpu.d_optim.steps_taken == 200 and gpu.g_optim.steps_taken == 400
    # stated tolerance: absolute loss deviation from the float64 trajectory <= 5e-2 over 200 steps
    # and <= 2e-2 over the first 20, and never worse than 4x what the fp32 numpy oracle itself shows
    assert dev_gpu[:20].max() <= 2e-2
    assert dev_gpu.max() <= 5e-2
    assert dev_gpu.max() <= 4 * dev_f32.max() + 1e-3


def test_logging_evals_and_sampler():
    from gpu_util import assert_close
    opts = make_opts()
    ref, gpu, rng = build_pair(opts, seed=7)
    images, z, y = batch(opts, rng)
    r = ref.train_iteration(images, z, y, log_evals=True)
    out = gpu.train_iteration(*to_dev(images, z, y), log_evals=True)
    for k in ('errD_fake', 'errD_real', 'errG'):
        assert abs(float(out[k]) - r[k]) <= 1e-4 * max(1.0, abs(r[k])), k
    for n, v in gpu.store.vars.items():
        if 'moving_' in n:
            # a moving mean near zero is compared on the scale of the matching standard deviation
            assert_close(v.data.cpu().numpy(), ref.st.values[n], n, tol=5e-3, elementwise=False)   # three Adam steps in
    s_ref = ref.sample(z)
    s = gpu.sampler(torch.from_numpy(z).cuda())
    assert_close(s.cpu().numpy(), s_ref, "sampler output", tol=5e-4, elementwise=False)


def test_hip_graph_replay_equals_eager():
    from gan_amd.graph import CapturedIteration, model_state_restore, model_state_snapshot
    opts = make_opts()
    _, eager, rng = build_pair(opts, seed=9)
    _, graphed, _ = build_pair(opts, seed=9)
    batches = [to_dev(*batch(opts, rng)[:2]) for _ in range(4)]
    optims = [graphed.d_optim, graphed.g_optim]
    step = CapturedIteration(lambda im, z: graphed.train_iteration(im, z), batches[0],
                             snapshot=lambda: model_state_snapshot(graphed.store, optims),
                             restore=lambda s: model_state_restore(graphed.store, optims, s))
    for im, z in batches:
        e = eager.train_iteration(im, z)
        g = step(im, z)
        torch.cuda.synchronize()
        for k in e:
            assert float(e[k]) == float(g[k]), k
    for name, v in eager.store.vars.items():
        assert torch.equal(v.data, graphed.store.vars[name].data), name
    assert graphed.d_optim.steps_taken == 4 and graphed.g_optim.steps_taken == 8


def test_checkpoint_round_trip(tmp_path):
    opts = make_opts(checkpoint_dir=str(tmp_path))
    _, gpu, rng = build_pair(opts, seed=3)
    im, z = to_dev(*batch(opts, rng)[:2])
    gpu.train_iteration(im, z)
    path = gpu.save(str(tmp_path), 7)
    assert path.endswith("DCGAN_synthetic_8_16_16/DCGAN.model-7.npz")
    ref_state = gpu.store.state_dict()
    expect = gpu.train_iteration(im, z)
    _, other, _ = build_pair(opts, seed=4)
    ok, counter = other.load(str(tmp_path))
    assert ok and counter == 7
    for n, a in ref_state.items():
        np.testing.assert_array_equal(other.store.vars[n].data.cpu().numpy(), a)
    got = other.train_iteration(im, z)
    for k in expect:
        assert float(expect[k]) == float(got[k])      # Adam slots and step counts came back too


@pytest.mark.parametrize("cond", [False, True])
def test_shared_generator_pass_equals_run_by_run_schedule(cond):
    """Reusing the D update's generator pass in the first G update changes nothing that is trained
    (bitwise) and moves the batch-norm moving averages like the second evaluation would have."""
    opts = make_opts(y_dim=4, c_dim=1, output_height=12, output_width=12, input_height=12) if cond else make_opts()
    _, a, rng = build_pair(opts, seed=13)
    _, b, _ = build_pair(opts, seed=13)
    for _ in range(3):
        im, z, y = to_dev(*batch(opts, rng, lo=0.0 if cond else -1.0))
        oa = a.train_iteration(im, z, y, share_g_pass=True)
        ob = b.train_iteration(im, z, y, share_g_pass=False)
        torch.cuda.synchronize()
        for k in oa:
            assert float(oa[k]) == float(ob[k]), k
    for name, v in a.store.vars.items():
        w = b.store.vars[name]
        if 'moving_' in name:
            torch.testing.assert_close(v.data, w.data, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(v.data, w.data), name
    assert a.g_optim.steps_taken == b.g_optim.steps_taken == 6


@pytest.mark.parametrize("cond,stacked", [(False, True), (False, False), (True, True)])
def test_staged_backward_with_early_exchange_equals_plain_backward(cond, stacked, tmp_path):
    """With a process group the backward pass stops at the boundary activation, starts the exchange of
    the early part of the gradient arena and continues: on a one-rank group every trained value must
    come out bit-identical to the unstaged single-process iteration, eager and as a chain of graphs."""
    import torch.distributed as dist
    from gan_amd.dcgan import DCGAN
    opts = make_opts(y_dim=4, c_dim=1, output_height=12, output_width=12, input_height=12) if cond else make_opts()
    _, plain, rng = build_pair(opts, seed=21)
    plain.stack_d_passes = stacked
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rendezvous"), rank=0, world_size=1)
    try:
        models = []
        for _ in range(2):
            m = DCGAN(None, opts, process_group=dist.group.WORLD, load_data=False)
            m.stack_d_passes = stacked
            m.store.load_state_dict(plain.store.state_dict())
            models.append(m)
        staged, chained = models
        assert staged.overlap_exchange and staged.d_group.early_offset > 0 and staged.g_group.early_offset > 0
        assert [v.early for v in staged.g_group.variables].count(True) >= 4
        batches = [to_dev(*batch(opts, rng, lo=0.0 if cond else -1.0)) for _ in range(3)]
        run = chained.iteration_runner(*batches[0])
        for im, z, y in batches:
            want = plain.train_iteration(im, z, y)
            got = staged.train_iteration(im, z, y)
            rep = run(im, z, y)
            torch.cuda.synchronize()
            for k in want:
                assert float(want[k]) == float(got[k]) == float(rep[k]), k
        for name, v in plain.store.vars.items():
            assert torch.equal(v.data, staged.store.vars[name].data), name
            assert torch.equal(v.data, chained.store.vars[name].data), name
        from gan_amd.graph import SegmentedIteration
        assert SegmentedIteration.active is None
    finally:
        dist.destroy_process_group()


def test_default_and_exact_f32_contraction_arithmetic_train_alike():
    """The default contraction arithmetic (fp32 operands as three exact bf16 parts on the bf16 matrix pipe)
    against gan_set_exact_f32(1) (fp32 MFMA): the first iteration's losses agree to fp32 rounding, and over
    20 iterations the two runs drift apart no faster than the GPU drifts from the float64 oracle
    (the 2e-2 bound of the trajectory test)."""
    from gan_amd import _lib as L
    opts = make_opts()
    _, a, rng = build_pair(opts, seed=31)
    _, b, _ = build_pair(opts, seed=31)
    batches = [to_dev(*batch(opts, rng)[:2]) for _ in range(20)]
    keys = ('d_loss_real', 'd_loss_fake', 'g_loss_1', 'g_loss_2')
    import ctypes
    from gpu_util import geom_for
    c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    g = geom_for(8, 4, 4, 32, 64, 5, 2, 'SAME')                # d_h3 of this small model
    L.check(L.lib.gan_conv_plan_get(0, ctypes.byref(g), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
    assert c.value >= 4                                         # the default really is the split kernel here
    try:
        la = [[float(o[k]) for k in keys] for o in (a.train_iteration(im, z) for im, z in batches)]
        L.lib.gan_set_exact_f32(1)
        L.check(L.lib.gan_conv_plan_get(0, ctypes.byref(g), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
        assert c.value < 4
        lb = [[float(o[k]) for k in keys] for o in (b.train_iteration(im, z) for im, z in batches)]
    finally:
        L.lib.gan_set_exact_f32(0)
    la, lb = np.array(la), np.array(lb)
    assert np.abs(la[0] - lb[0]).max() <= 2e-5 * max(1.0, np.abs(lb[0]).max()), (la[0], lb[0])
    assert np.abs(la - lb).max() <= 2e-2, np.abs(la - lb).max(axis=0)


@pytest.mark.parametrize("cond", [False, True])
def test_train_loop_with_graph_replay_and_prefetch_equals_eager_loop(cond, tmp_path, capsys):
    """DCGAN.train replays the iteration as a HIP graph and takes its batches from the background input
    pipeline; with use_graph off it issues every launch eagerly.  Same data, same seeds: the same log lines
    and bit-identical variables, Adam step counts included."""
    kw = dict(y_dim=4, c_dim=1, output_height=28, output_width=28, input_height=28, input_width=28, dataset='mnist') if cond else {}
    rng = np.random.default_rng(5)
    n = 5 * 8
    data_X = rng.uniform(0 if cond else -1, 1, (n, 28 if cond else 16, 28 if cond else 16, 1 if cond else 3)).astype(np.float32)
    data_y = np.eye(4, dtype=np.float32)[rng.integers(0, 4, n)] if cond else None
    logs, models = [], []
    for use_graph in (True, False):
        opts = make_opts(checkpoint_dir=str(tmp_path / ("ck%d" % use_graph)), sample_dir=str(tmp_path / "smp"),
                         use_graph=use_graph, print_every=2, **kw)
        _, gpu, _ = build_pair(opts, seed=13)
        capsys.readouterr()
        gpu.train(data_X=data_X, data_y=data_y, sample_every=3)
        logs.append([l.split("time:")[0] + l.split(",", 1)[1] for l in capsys.readouterr().out.splitlines() if l.startswith("Epoch")])
        models.append(gpu)
    assert len(logs[0]) == 3 and logs[0] == logs[1]
    a, b = models
    assert a.counter == b.counter == 6            # the reference starts its counter at 1 (dcgan.py:189)
    assert a.d_optim.steps_taken == b.d_optim.steps_taken == 5 and a.g_optim.steps_taken == b.g_optim.steps_taken == 10
    for name, v in a.store.vars.items():
        assert torch.equal(v.data, b.store.vars[name].data), name
    assert [p.name for p in (tmp_path / "smp").glob("train_00_*.png")] == ["train_00_0002.png"]     # counter 4


def test_train_from_image_files_with_resident_uint8_copy_equals_decoding_every_epoch(tmp_path, capsys):
    """Image files: epoch 1 decodes them on the input pipeline's threads, ships uint8 and scales on the GPU;
    epoch 2 reads the uint8 copy left in GPU memory.  Against the loop that decodes every batch inline,
    scales on the host and launches eagerly: same log, bit-identical variables."""
    from PIL import Image
    from gan_amd.misc import get_image
    rng = np.random.default_rng(8)
    files = []
    for i in range(19):                                  # 2 batches of 8 per epoch, 3 files left over
        f = tmp_path / ("img%03d.png" % i)
        Image.fromarray(rng.integers(0, 256, (30, 26, 3), dtype=np.uint8)).save(f)
        files.append(str(f))
    logs, models = [], []
    for fast in (True, False):
        opts = make_opts(checkpoint_dir=str(tmp_path / ("ck%d" % fast)), sample_dir=str(tmp_path / "smp"), nEpochs=2,
                         input_height=20, input_width=20, use_graph=fast, dataset_cache_gb=32 if fast else 0)
        _, gpu, _ = build_pair(opts, seed=17)
        if not fast:                                     # the reference's inline float path, batch by batch
            gpu._load_files_u8 = None
            gpu._scale_u8 = lambda t: t
            batches = {tuple(files[i:i + 8]): np.array([get_image(f, 20, 20, 16, 16, True, False) for f in files[i:i + 8]]
                                                       ).astype(np.float32) for i in (0, 8)}
            gpu._load_files_u8 = lambda fs, pool_map=None: batches[tuple(fs)]
        capsys.readouterr()
        gpu.train(data_files=files, sample_every=0)
        logs.append([l.split("time:")[0] + l.split(",", 1)[1] for l in capsys.readouterr().out.splitlines() if l.startswith("Epoch")])
        models.append(gpu)
    assert len(logs[0]) == 4 and logs[0] == logs[1]
    for name, v in models[0].store.vars.items():
        assert torch.equal(v.data, models[1].store.vars[name].data), name
