"""Text2Image (GAN-CLS) and CycleGAN on the GPU against the numpy oracle: identical seeded weights
and inputs, per-update gradients, the reference schedules, loss values over a few iterations."""
import argparse

import numpy as np
import pytest
import torch

from oracle import autodiff as A
from oracle import nets

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def check_grads(group, ref_grads, tol=3e-4):
    from gpu_util import check_group_grads
    check_group_grads(group, ref_grads, tol)


def perturb(st, rng):
    for n in st.trainable:
        if n.endswith(('gamma', 'beta', 'biases', 'bias')):
            st.values[n] = (st.values[n] + rng.standard_normal(st.values[n].shape) * 0.1).astype(np.float32)


class kink_margin(object):
    """Records, over the oracle ops run inside the block, the smallest |pre-activation| seen by a relu
    or lrelu.  Their derivative jumps at 0, so an input that puts a unit within rounding distance of
    the kink makes the float32 and float64 gradients legitimately differ by a whole unit's worth;
    gradient comparisons draw inputs that keep every unit clear of it."""

    def __enter__(self):
        self.value = np.inf
        self.saved = (A.relu, A.lrelu)

        def relu(tp, n, *a, **k):
            self.value = min(self.value, float(np.abs(n.v).min()))
            return self.saved[0](tp, n, *a, **k)

        def lrelu(tp, n, *a, **k):
            self.value = min(self.value, float(np.abs(n.v).min()))
            return self.saved[1](tp, n, *a, **k)
        A.relu, A.lrelu = relu, lrelu
        return self

    def __exit__(self, *exc):
        A.relu, A.lrelu = self.saved


# ------------------------------------------------------------------------------------------ text2image
def t2i_opts():
    return argparse.Namespace(model='Text2Image', dataset='flowers', batch_size=4, input_height=64, z_dim=10, t_dim=8,
                              gf_dim=4, df_dim=4, caption_vector_length=12, textEncoder='DSSJE', lr=0.0002, beta1=0.5,
                              nEpochs=1, seed=0)


def t2i_pair(seed=21):
    from gan_amd.text2image import Text2Image
    o = t2i_opts()
    rng = np.random.default_rng(seed)
    st = A.Store(seed=seed)
    ref = nets.Text2Image(st, batch_size=o.batch_size, image_size=64, z_dim=o.z_dim, t_dim=o.t_dim, gf_dim=o.gf_dim,
                          df_dim=o.df_dim, caption_vector_length=o.caption_vector_length, lr=o.lr, beta1=o.beta1)
    z0 = np.zeros((4, o.z_dim), np.float32)
    ref.build(np.zeros((4, 64, 64, 3), np.float32), np.zeros((4, 64, 64, 3), np.float32),
              np.zeros((4, o.caption_vector_length), np.float32), z0)
    perturb(st, rng)
    gpu = Text2Image(None, o)
    assert list(gpu.store.vars) == list(st.values)
    gpu.store.load_state_dict(st.values)
    return ref, gpu, rng, o


def t2i_batch(rng, o):
    real = rng.uniform(0, 1, (o.batch_size, 64, 64, 3)).astype(np.float32)
    wrong = rng.uniform(0, 1, (o.batch_size, 64, 64, 3)).astype(np.float32)
    cap = rng.standard_normal((o.batch_size, o.caption_vector_length)).astype(np.float32)
    z = rng.uniform(-1, 1, (o.batch_size, o.z_dim)).astype(np.float32)
    return real, wrong, cap, z


def test_text2image_update_gradients_match_oracle():
    from gpu_util import assert_close
    ref, gpu, rng, o = t2i_pair()
    real, wrong, cap, z = t2i_batch(rng, o)
    before = {n: v.copy() for n, v in ref.st.values.items()}
    # oracle: d_loss gradient
    tp = A.Tape(ref.st)
    c = tp.const(cap)
    fake = tp.const(ref.generator(A.Tape(ref.st), tp.const(z), c).v)
    ls = [A.sigmoid_ce_mean(tp, ref.discriminator(tp, tp.const(real), c), 1.0),
          A.sigmoid_ce_mean(tp, ref.discriminator(tp, tp.const(wrong), c), 0.0),
          A.sigmoid_ce_mean(tp, ref.discriminator(tp, fake, c), 0.0)]
    tp.backward(A.add(tp, A.add(tp, ls[0], ls[1]), ls[2]))
    d_ref = tp.grads(ref.d_vars)
    d1, d2, d3, _ = gpu.d_grads(dev(real), dev(wrong), dev(cap), dev(z))
    torch.cuda.synchronize()
    for got, want in zip((d1, d2, d3), ls):
        assert_close(got.cpu().numpy(), [float(want.v)], "d_loss term")
    check_grads(gpu.d_group, d_ref)
    # oracle: g_loss gradient from the same weights
    for n in ref.st.values:
        ref.st.values[n] = before[n]
    gpu.store.load_state_dict(before)
    tp = A.Tape(ref.st)
    c = tp.const(cap)
    gl = A.sigmoid_ce_mean(tp, ref.discriminator(tp, ref.generator(tp, tp.const(z), c), c), 1.0)
    tp.backward(gl)
    g_ref = tp.grads(ref.g_vars)
    g_loss, gen = gpu.g_grads(dev(cap), dev(z))
    torch.cuda.synchronize()
    assert_close(g_loss.cpu().numpy(), [float(gl.v)], "g_loss")
    check_grads(gpu.g_group, g_ref)
    assert gen.shape == (4, 64, 64, 3) and float(gen.min()) >= 0.0 and float(gen.max()) <= 1.0


def test_text2image_schedule_losses():
    ref, gpu, rng, o = t2i_pair(seed=23)
    for it in range(3):
        real, wrong, cap, z = t2i_batch(rng, o)
        want = ref.train_iteration(real, wrong, cap, z)
        out = gpu.train_iteration(dev(real), dev(wrong), dev(cap), dev(z))
        for k in ('d_loss1', 'd_loss2', 'd_loss3', 'g_loss_1', 'g_loss_2'):
            assert abs(float(out[k]) - want[k]) <= 2e-3 * max(1.0, abs(want[k])), (it, k, float(out[k]), want[k])
    assert gpu.d_optim.steps_taken == 3 and gpu.g_optim.steps_taken == 6
    s = gpu.sampler(dev(z), dev(cap))
    assert s.shape == (4, 64, 64, 3)


# ------------------------------------------------------------------------------------------ cyclegan
def cyc_opts(**kw):
    d = dict(model='CycleGAN', dataset='horse2zebra', batch_size=1, fine_size=32, input_nc=3, output_nc=3,
             L1_lambda=10.0, use_resnet=True, use_lsgan=True, max_size=0, gf_dim=4, df_dim=4, lr=0.0002, beta1=0.5,
             data_dir='data', seed=0)
    d.update(kw)
    return argparse.Namespace(**d)


def cyc_pair(seed=31, **kw):
    from gan_amd.cyclegan import CycleGAN
    o = cyc_opts(**kw)
    rng = np.random.default_rng(seed)
    st = A.Store(seed=seed)
    ref = nets.CycleGAN(st, batch_size=o.batch_size, image_size=o.fine_size, gf_dim=o.gf_dim, df_dim=o.df_dim,
                        L1_lambda=o.L1_lambda, use_lsgan=o.use_lsgan, lr=o.lr, beta1=o.beta1, max_size=o.max_size)
    ref.build(np.zeros((o.batch_size, o.fine_size, o.fine_size, 6), np.float32))
    perturb(st, rng)
    gpu = CycleGAN(None, o)
    assert list(gpu.store.vars) == list(st.values), "variable names / creation order differ from the oracle"
    gpu.store.load_state_dict(st.values)
    return ref, gpu, rng, o


@pytest.mark.parametrize("use_lsgan", [True, False])
def test_cyclegan_update_gradients_match_oracle(use_lsgan):
    from gpu_util import assert_close
    ref, gpu, rng, o = cyc_pair(use_lsgan=use_lsgan)
    before = {n: v.copy() for n, v in ref.st.values.items()}
    # The generator is ~50 normalised layers deep and, at batch 1 on a 32x32 image, normalises over
    # as few as 64 values per channel: rounding differences are amplified on the way back to the
    # first layers.  The float64 oracle is the truth and the float32 oracle's own deviation from it
    # is the yardstick for what fp32 arithmetic can deliver here.
    st64 = A.Store(seed=0, dtype=np.float64)
    ref64 = nets.CycleGAN(st64, batch_size=o.batch_size, image_size=o.fine_size, gf_dim=o.gf_dim, df_dim=o.df_dim,
                          L1_lambda=o.L1_lambda, use_lsgan=o.use_lsgan, lr=o.lr, beta1=o.beta1, max_size=0)
    ref64.build(np.zeros((o.batch_size, o.fine_size, o.fine_size, 6)))
    for n in st64.values:
        st64.values[n] = before[n].astype(np.float64)
    for attempt in range(50):
        real = rng.uniform(-1, 1, (1, o.fine_size, o.fine_size, 6)).astype(np.float32)
        for n in st64.values:       # the forward moved the moving averages
            st64.values[n] = before[n].astype(np.float64)
        tp64 = A.Tape(st64)
        with kink_margin() as margin:
            gl64 = ref64.g_loss(tp64, real.astype(np.float64), 'a2b')
        if margin.value >= 2e-5:
            break
    assert margin.value >= 2e-5, "no input found that keeps every relu clear of its kink"
    tp64.backward(gl64)
    g_true = tp64.grads(ref64.g_vars_a2b)
    tp = A.Tape(ref.st)
    gl = ref.g_loss(tp, real, 'a2b')
    tp.backward(gl)
    g_ref = tp.grads(ref.g_vars_a2b)
    gan, ca, cb = gpu.g_update('a2b', dev(real))
    torch.cuda.synchronize()
    assert_close([float(gan) + float(ca) + float(cb)], [float(gl64.v)], "g_loss_a2b")
    from gpu_util import is_noise_grad, host
    worst_gpu = worst_f32 = 0.0
    bad = []
    for v in gpu.g_a2b_optim.group.variables:
        if is_noise_grad(v.name, g_true):
            continue
        scale = float(np.abs(g_true[v.name]).max())
        e_gpu = float(np.abs(host(v.grad) - g_true[v.name]).max()) / scale
        e_f32 = float(np.abs(g_ref[v.name] - g_true[v.name]).max()) / scale
        worst_gpu, worst_f32 = max(worst_gpu, e_gpu), max(worst_f32, e_f32)
        if e_gpu > max(5e-4, 4 * e_f32):
            bad.append((v.name, e_gpu, e_f32))
    assert not bad, bad
    print("cyclegan G gradient: worst rel deviation from float64: gpu %.2e, float32 numpy oracle %.2e" % (worst_gpu, worst_f32))
    # D_B update from the original weights with a given fake sample
    for n in ref.st.values:
        ref.st.values[n] = before[n]
    gpu.store.load_state_dict(before)
    a, b = ref.split(real)
    fake_b = rng.uniform(-1, 1, b.shape).astype(np.float32)
    tp = A.Tape(ref.st)
    dl = ref.d_loss(tp, b, fake_b, "discriminatorB")
    tp.backward(dl)
    d_ref = tp.grads(ref.db_vars)
    lr_, lf_ = gpu.d_update('B', dev(b), dev(fake_b))
    torch.cuda.synchronize()
    assert_close([float(lr_) + float(lf_)], [float(dl.v)], "db_loss")
    check_grads(gpu.db_optim.group, d_ref)


def test_cyclegan_schedule_losses():
    from gan_amd.cyclegan import CycleGAN
    ref, gpu, rng, o = cyc_pair(seed=33)
    for it in range(2):
        real = rng.uniform(-1, 1, (1, o.fine_size, o.fine_size, 6)).astype(np.float32)
        want = ref.train_iteration(real)
        got = CycleGAN.totals(gpu.train_iteration(dev(real)))
        for k in want:
            assert abs(got[k] - want[k]) <= 3e-3 * max(1.0, abs(want[k])), (it, k, got[k], want[k])
    for opt in gpu.optims.values():
        assert opt.steps_taken == 2
    out = gpu.test(dev(real[..., :3]), 'AtoB')
    assert out.shape == (1, o.fine_size, o.fine_size, 3) and float(out.abs().max()) <= 1.0


def test_image_pool_matches_reference_semantics():
    from gan_amd.cyclegan import ImagePool
    seq = iter([0.9, 0.5, 0.1])
    pool = ImagePool(2, rand=lambda: next(seq), randint=lambda n: 1)
    a, b, c, d = ([i] for i in range(4))
    assert pool(a) is a and pool(b) is b
    out = pool(c)
    assert out[0] == 1 and pool.images[1] is c
    assert pool(d) is d
    assert ImagePool(0)(a) is a


def test_instance_norm_equals_per_sample_batch_norm():
    from gan_amd import ops, variables as V
    from gpu_util import assert_close
    from oracle import tf_ops as T
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 6, 6, 8)).astype(np.float32)
    st = V.VariableStore(seed=0)
    with V.use_store(st):
        xt = dev(x).requires_grad_(True)
        y = ops.instance_norm(xt, name='in0', act='relu')
        dy = rng.standard_normal(x.shape).astype(np.float32)
        grp = st.pack('g', st.trainable_variables())
        grp.begin_update()
        y.backward(dev(dy))
        grp.end_update()
    torch.cuda.synchronize()
    ys, dxs = [], []
    dg = np.zeros(8)
    for i in range(3):
        z, mean, var, rstd = T.batch_norm_train_forward(x[i:i + 1].astype(np.float64), np.ones(8), np.zeros(8), 1e-5)
        ys.append(T.relu(z))
        dz = dy[i:i + 1] * (z > 0)
        dx, g_, b_ = T.batch_norm_train_backward(x[i:i + 1].astype(np.float64), np.ones(8), mean, rstd, dz)
        dxs.append(dx)
        dg += g_
    assert_close(y.detach().cpu().numpy(), np.concatenate(ys), "instance norm fwd")
    assert_close(xt.grad.cpu().numpy(), np.concatenate(dxs), "instance norm dx", tol=3e-4)
    assert_close(st.vars['in0/gamma'].grad.cpu().numpy(), dg, "instance norm dgamma", tol=3e-4)


def test_batch_norm_stacked_passes_equal_separate_calls():
    """batch_norm(passes=2) on [a; b] == the layer applied to a and then to b: outputs, input
    gradients, accumulated gamma/beta gradients and the twice-moved moving averages."""
    from gan_amd import ops, variables as V
    from gpu_util import assert_close, host
    rng = np.random.default_rng(9)
    a = rng.standard_normal((4, 5, 5, 8)).astype(np.float32) * 2 + 1
    b = rng.standard_normal((4, 5, 5, 8)).astype(np.float32) - 0.5
    dy = rng.standard_normal((8, 5, 5, 8)).astype(np.float32)
    res = []
    for stacked in (False, True):
        st = V.VariableStore(seed=0)
        with V.use_store(st):
            bn = ops.batch_norm(name='bn')
            ta, tb = dev(a).requires_grad_(True), dev(b).requires_grad_(True)
            bn(ta, act='lrelu')                                   # creates the variables
            for n in ('bn/moving_mean',):
                st.vars[n].data.zero_()
            st.vars['bn/moving_variance'].data.fill_(1.0)
            st.vars['bn/gamma'].data.copy_(dev(np.linspace(0.5, 1.5, 8)))
            grp = st.pack('g', st.trainable_variables())
            grp.begin_update()
            if stacked:
                both = ops.stack_batches(ta, tb).requires_grad_(True)
                y = bn(both, act='lrelu', passes=2)
                y.backward(dev(dy))
                dx = host(both.grad)
            else:
                ya, yb = bn(ta, act='lrelu'), bn(tb, act='lrelu')
                y = torch.cat([ya, yb])
                torch.autograd.backward([ya, yb], [dev(dy[:4]), dev(dy[4:])])
                dx = np.concatenate([host(ta.grad), host(tb.grad)])
            grp.end_update()
        torch.cuda.synchronize()
        res.append((host(y), dx, host(st.vars['bn/gamma'].grad), host(st.vars['bn/beta'].grad),
                    host(st.vars['bn/moving_mean'].data), host(st.vars['bn/moving_variance'].data)))
    for got, want, what in zip(res[1], res[0], ("y", "dx", "dgamma", "dbeta", "moving_mean", "moving_variance")):
        assert_close(got, want, "stacked batch_norm " + what, tol=1e-6, elementwise=False)


def test_cyclegan_shared_generator_passes_equal_run_by_run():
    """train_iteration keeps the generator passes of the first run for the generator updates and
    prunes the backward of frozen sub-graphs; the trained values are those of five independent runs."""
    from gan_amd.cyclegan import CycleGAN
    _, a, rng, o = cyc_pair(seed=35)
    _, b, _, _ = cyc_pair(seed=35)
    for _ in range(2):
        real = dev(rng.uniform(-1, 1, (1, o.fine_size, o.fine_size, 6)))
        ta = CycleGAN.totals(a.train_iteration(real))
        rA, rB = b.split(real)
        fA, fB = b.forward_fakes(real)                      # nothing kept: every run recomputes
        out = {}
        out['db_loss_real'], out['db_loss_fake'] = b.d_update('B', rB, fB)
        out['g_a2b_gan'], out['g_a2b_cyc_a'], out['g_a2b_cyc_b'] = b.g_update('a2b', real)
        out['da_loss_real'], out['da_loss_fake'] = b.d_update('A', rA, fA)
        out['g_b2a_gan'], out['g_b2a_cyc_a'], out['g_b2a_cyc_b'] = b.g_update('b2a', real)
        tb = CycleGAN.totals(out)
        torch.cuda.synchronize()
        assert ta == tb
    for name, v in a.store.vars.items():
        w = b.store.vars[name]
        if 'moving_' in name:
            torch.testing.assert_close(v.data, w.data, rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(v.data, w.data), name


def test_text2image_shared_pass_and_stacked_d_equal_separate_runs():
    _, a, rng, o = t2i_pair(seed=25)
    _, b, _, _ = t2i_pair(seed=25)
    b.stack_d_passes = False
    for _ in range(2):
        real, wrong, cap, z = [dev(t) for t in t2i_batch(rng, o)]
        oa = a.train_iteration(real, wrong, cap, z)
        d1, d2, d3, _ = b.d_update(real, wrong, cap, z)
        g1, _ = b.g_update(cap, z)
        g2, _ = b.g_update(cap, z)
        torch.cuda.synchronize()
        for got, want in zip((oa['d_loss1'], oa['d_loss2'], oa['d_loss3'], oa['g_loss_1'], oa['g_loss_2']),
                             (d1, d2, d3, g1, g2)):
            assert abs(float(got) - float(want)) <= 1e-4 * max(1.0, abs(float(want)))   # second iteration: after Adam steps on rounding-noise gradients
    for name, v in a.store.vars.items():      # stacked vs separate D passes differ in summation order only
        w = b.store.vars[name]
        if name.endswith(('/biases', '/bias')):
            continue                           # rounding-noise gradients under Adam
        # Adam moves an element whose gradient is rounding noise by up to lr per update: 6 updates of 2e-4
        torch.testing.assert_close(v.data, w.data, rtol=2e-3, atol=1.5e-3)


def test_text2image_staged_backward_with_early_exchange_equals_plain(tmp_path):
    """Text2Image with a (one-rank) process group: the staged backward pass, whose boundary leaves the
    text-embedding branch of the discriminator on the early side, trains bit-identical values."""
    import torch.distributed as dist
    from gan_amd.text2image import Text2Image
    _, plain, rng, o = t2i_pair(seed=27)
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rendezvous"), rank=0, world_size=1)
    try:
        staged = Text2Image(None, o, process_group=dist.group.WORLD)
        staged.store.load_state_dict(plain.store.state_dict())
        early = [v.name for v in staged.d_group.variables if v.early]
        assert 'discriminator/d_embedding/Matrix' in early and 'discriminator/d_h2_conv/w' not in early
        for _ in range(2):
            b = [dev(a) for a in t2i_batch(rng, o)]
            want, got = plain.train_iteration(*b), staged.train_iteration(*b)
            for k in ('d_loss1', 'd_loss2', 'd_loss3', 'g_loss_1', 'g_loss_2'):
                assert float(want[k]) == float(got[k]), k
        for name, v in plain.store.vars.items():
            assert torch.equal(v.data, staged.store.vars[name].data), name
    finally:
        dist.destroy_process_group()


def test_text2image_train_loop_with_graph_replay_equals_eager_loop(tmp_path, capsys):
    """Text2Image.train (prefetched batches, HIP-graph replay) against the same loop issued eagerly."""
    from gan_amd.text2image import Text2Image
    rng = np.random.default_rng(2)
    images = rng.uniform(0, 1, (12, 64, 64, 3)).astype(np.float32)
    captions = rng.standard_normal((12, 12)).astype(np.float32)
    models, logs = [], []
    for use_graph in (True, False):
        o = t2i_opts()
        o.use_graph, o.checkpoint_dir, o.save_every_batch = use_graph, str(tmp_path / ("ck%d" % use_graph)), 2
        m = Text2Image(None, o)
        if models:
            m.store.load_state_dict(models[0][1])
        models.append((m, m.store.state_dict()))
        capsys.readouterr()
        m.train(images, captions, seed=4)
        logs.append([l for l in capsys.readouterr().out.splitlines() if l.startswith("epoch")])
    assert len(logs[0]) == 3 and logs[0] == logs[1]
    for name, v in models[0][0].store.vars.items():
        assert torch.equal(v.data, models[1][0].store.vars[name].data), name
    assert (tmp_path / "ck1" / models[0][0].model_dir).exists()
    # --resume_model (text2image.py:237-238): a model started from the checkpoint written after batch 2
    # trains on from there -- one more epoch from that state equals one more epoch of the saved model
    ckpt = str(tmp_path / "ck1" / models[0][0].model_dir / "Text2Image.model-2")
    o = t2i_opts()
    o.use_graph, o.checkpoint_dir, o.save_every_batch, o.resume_model = True, str(tmp_path / "ck3"), 100, ckpt
    resumed = Text2Image(None, o)
    assert resumed.resume_model == ckpt and resumed._resume()
    from gan_amd.dcgan import load_checkpoint
    o2 = t2i_opts()
    o2.use_graph, o2.checkpoint_dir, o2.save_every_batch = True, str(tmp_path / "ck4"), 100
    twin = Text2Image(None, o2)
    load_checkpoint(twin.store, {'d': twin.d_optim, 'g': twin.g_optim}, ckpt + ".npz")
    assert twin.d_optim.steps_taken == 2 and resumed.d_optim.steps_taken == 2
    resumed.train(images, captions, seed=9)
    twin.train(images, captions, seed=9)
    for name, v in twin.store.vars.items():
        assert torch.equal(v.data, resumed.store.vars[name].data), name


def test_cyclegan_train_loop_with_graphs_cut_at_the_image_pool_equals_eager_loop(tmp_path):
    """CycleGAN.train replays the iteration as two HIP graphs with the image pool asked in between (its
    history and its coin tosses are host state).  A pool of 2 is full after two iterations, so the other
    four draw from it: same seeds, same in-memory data -> variables bit-identical to the eager loop's."""
    from gan_amd.cyclegan import CycleGAN
    rng = np.random.default_rng(6)
    data_a = rng.uniform(-1, 1, (6, 32, 32, 3)).astype(np.float32)
    data_b = rng.uniform(-1, 1, (6, 32, 32, 3)).astype(np.float32)
    models, pools = [], []
    for use_graph in (True, False):
        o = cyc_opts(max_size=2, nEpochs=1, checkpoint_dir=str(tmp_path / ("ck%d" % use_graph)), use_graph=use_graph,
                     sample_dir=str(tmp_path / "smp"), train_size=1e8)
        m = CycleGAN(None, o)
        if models:
            m.store.load_state_dict(models[0][1])
        models.append((m, m.store.state_dict()))
        draws = np.random.RandomState(12)
        m.pool.rand = draws.rand
        m.pool.randint = lambda n, d=draws: int(d.rand() * n)
        out = m.train(dataA=data_a, dataB=data_b, seed=3, sample_every=0)
        assert all(np.isfinite(float(v)) for v in out.values())
        pools.append(m.pool)
    assert pools[0].num_img == pools[1].num_img == 2
    for kept_g, kept_e in zip(pools[0].images, pools[1].images):
        for a, b in zip(kept_g, kept_e):
            assert torch.equal(a, b)
    for name, v in models[0][0].store.vars.items():
        assert torch.equal(v.data, models[1][0].store.vars[name].data), name
