"""Whole-graph checks of the numpy oracle's models: float64 finite differences of
the DCGAN / Text2Image / CycleGAN losses w.r.t. sampled weights, variable naming
and grouping, and the step schedule's optimiser bookkeeping."""
import numpy as np

from oracle import autodiff as A
from oracle import nets


def _fd_check(store, loss_fn, names, rng, n_probe=3, eps=1e-6, rtol=2e-4):
    tp = A.Tape(store)
    loss = loss_fn(tp)
    tp.backward(loss)
    grads = tp.grads(names)
    mm = {k: v.copy() for k, v in store.values.items() if 'moving_' in k}
    for name in names:
        g = grads[name]
        for _ in range(n_probe):
            idx = tuple(int(rng.integers(0, s)) for s in g.shape)
            old = store.values[name][idx]
            store.values[name][idx] = old + eps
            lp = float(loss_fn(A.Tape(store)).v)
            store.values[name][idx] = old - eps
            lm = float(loss_fn(A.Tape(store)).v)
            store.values[name][idx] = old
            num = (lp - lm) / (2 * eps)
            assert abs(num - g[idx]) <= rtol * max(abs(num), abs(g[idx])) + 1e-9, (name, idx, num, g[idx])
    for k, v in mm.items():
        store.values[k] = v


def test_dcgan_uncond_gradients_and_names():
    rng = np.random.default_rng(0)
    st = A.Store(seed=1, dtype=np.float64)
    net = nets.DCGAN(st, batch_size=3, output_height=16, output_width=16, c_dim=3, z_dim=5, gf_dim=2, df_dim=2)
    images = rng.uniform(-1, 1, (3, 16, 16, 3))
    z = rng.uniform(-1, 1, (3, 5))
    net.build(images, z)
    assert 'generator/g_h0_lin/Matrix' in st.values and 'generator/g_h4/w' in st.values
    assert 'discriminator/d_h0_conv/w' in st.values and 'discriminator/d_bn3/gamma' in st.values
    assert 'generator/g_bn0/moving_mean' in st.values
    assert st.values['generator/g_h1/w'].shape == (5, 5, 8, 16)       # [kh,kw,Cout,Cin]
    assert st.values['discriminator/d_h1_conv/w'].shape == (5, 5, 2, 4)
    assert all('d_' in n for n in net.d_vars) and all('g_' in n for n in net.g_vars)
    assert not any('moving' in n for n in net.d_vars + net.g_vars)
    assert len(net.d_vars) == 4 * 2 + 3 * 2 + 2 and len(net.g_vars) == 2 + 4 * 2 + 4 * 2

    def d_loss(tp):
        r, f = net.d_losses(tp, images, z)
        return A.add(tp, r, f)
    _fd_check(st, d_loss, net.d_vars, rng)

    def g_loss(tp):
        G = net.generator(tp, tp.const(z))
        return A.sigmoid_ce_mean(tp, net.discriminator(tp, G), 1.0)
    _fd_check(st, g_loss, net.g_vars, rng)


def test_dcgan_cond_gradients():
    rng = np.random.default_rng(1)
    st = A.Store(seed=2, dtype=np.float64)
    net = nets.DCGAN(st, batch_size=3, output_height=12, output_width=12, c_dim=1, y_dim=4, z_dim=5,
                     gf_dim=2, df_dim=3, gfc_dim=6, dfc_dim=7)
    images = rng.uniform(0, 1, (3, 12, 12, 1))
    z = rng.uniform(-1, 1, (3, 5))
    y = np.eye(4)[rng.integers(0, 4, 3)]
    g = net.build(images, z, y)
    assert g.v.shape == (3, 12, 12, 1)
    assert st.values['discriminator/d_h0_conv/w'].shape == (5, 5, 5, 5)
    assert st.values['discriminator/d_h2_lin/Matrix'].shape == (3 * 3 * (3 + 4) + 4, 7)

    def d_loss(tp):
        r, f = net.d_losses(tp, images, z, y)
        return A.add(tp, r, f)
    _fd_check(st, d_loss, net.d_vars, rng, n_probe=2)

    def g_loss(tp):
        yn = tp.const(y)
        return A.sigmoid_ce_mean(tp, net.discriminator(tp, net.generator(tp, tp.const(z), yn), yn), 1.0)
    _fd_check(st, g_loss, net.g_vars, rng, n_probe=2)


def test_dcgan_schedule_and_moving_average_side_effects():
    rng = np.random.default_rng(2)
    st = A.Store(seed=3)
    net = nets.DCGAN(st, batch_size=4, output_height=16, output_width=16, c_dim=3, z_dim=6, gf_dim=2, df_dim=2)
    images = rng.uniform(-1, 1, (4, 16, 16, 3)).astype(np.float32)
    z = rng.uniform(-1, 1, (4, 6)).astype(np.float32)
    net.build(images, z)
    assert np.all(st.values['generator/g_bn0/moving_mean'] == 0)
    w_d = st.values['discriminator/d_h1_conv/w'].copy()
    w_g = st.values['generator/g_h1/w'].copy()
    out = net.train_iteration(images, z)
    assert net.d_adam.t == 1 and net.g_adam.t == 2          # G timestep +2 per iteration
    assert not np.array_equal(w_d, st.values['discriminator/d_h1_conv/w'])
    assert not np.array_equal(w_g, st.values['generator/g_h1/w'])
    assert np.isfinite(list(out.values())).all()
    # G BN ran 3 train-mode forwards, D BN 4 (real+fake in D update, fake in each G update)
    mv = st.values['generator/g_bn0/moving_variance']
    assert np.all(mv != 1)
    s = net.sample(z)
    assert s.shape == (4, 16, 16, 3) and np.all(np.abs(s) <= 1)


def test_text2image_gradients():
    rng = np.random.default_rng(3)
    st = A.Store(seed=4, dtype=np.float64)
    net = nets.Text2Image(st, batch_size=2, image_size=64, z_dim=4, t_dim=3, gf_dim=1, df_dim=1,
                          caption_vector_length=6)
    real, wrong = rng.uniform(0, 1, (2, 64, 64, 3)), rng.uniform(0, 1, (2, 64, 64, 3))
    cap, z = rng.standard_normal((2, 6)), rng.uniform(-1, 1, (2, 4))
    net.build(real, wrong, cap, z)
    assert st.values['discriminator/d_h3_conv_new/w'].shape == (1, 1, 8 + 3, 8)

    def d_loss(tp):
        c = tp.const(cap)
        fake = tp.const(net.generator(A.Tape(st), tp.const(z), c).v)
        ls = [A.sigmoid_ce_mean(tp, net.discriminator(tp, tp.const(real), c), 1.0),
              A.sigmoid_ce_mean(tp, net.discriminator(tp, tp.const(wrong), c), 0.0),
              A.sigmoid_ce_mean(tp, net.discriminator(tp, fake, c), 0.0)]
        return A.add(tp, A.add(tp, ls[0], ls[1]), ls[2])
    _fd_check(st, d_loss, net.d_vars, rng, n_probe=2)

    def g_loss(tp):
        c = tp.const(cap)
        return A.sigmoid_ce_mean(tp, net.discriminator(tp, net.generator(tp, tp.const(z), c), c), 1.0)
    _fd_check(st, g_loss, net.g_vars, rng, n_probe=2)
    out = net.train_iteration(real, wrong, cap, z)
    assert abs(out['d_loss'] - (out['d_loss1'] + out['d_loss2'] + out['d_loss3'])) < 1e-12


def test_cyclegan_gradients_and_schedule():
    rng = np.random.default_rng(4)
    st = A.Store(seed=5, dtype=np.float64)
    net = nets.CycleGAN(st, batch_size=1, image_size=16, gf_dim=2, df_dim=2, max_size=2)
    real = rng.uniform(-1, 1, (1, 16, 16, 6))
    net.build(real)
    assert st.values['generatorA2B/g_e1_c/w'].shape == (7, 7, 3, 2)
    assert st.values['generatorA2B/g_r9_c2/w'].shape == (3, 3, 8, 8)
    assert st.values['generatorB2A/g_d1_dc/w'].shape == (3, 3, 4, 8)
    assert st.values['discriminatorA/d_h3_pred/w'].shape == (5, 5, 16, 1)
    assert len(net.g_vars_a2b) == len(net.g_vars_b2a) == 2 * (3 + 18 + 2 + 1) + 2 * (6 + 18)
    probe = [n for n in net.g_vars_a2b if n.endswith('/w')][::6] + ['generatorA2B/g_r3_bn1/gamma']
    _fd_check(st, lambda tp: net.g_loss(tp, real, 'a2b'), probe, rng, n_probe=2, rtol=5e-4)
    a, b = net.split(real)
    fa, fb = net.forward_fakes(real)
    _fd_check(st, lambda tp: net.d_loss(tp, b, fb, "discriminatorB"),
              [n for n in net.db_vars if n.endswith('/w')], rng, n_probe=2)
    out = net.train_iteration(real)
    assert set(out) == {'db_loss', 'g_loss_a2b', 'da_loss', 'g_loss_b2a'}
    assert net.ga2b_adam.t == net.gb2a_adam.t == net.da_adam.t == net.db_adam.t == 1


def test_image_pool_semantics():
    seq = iter([0.9, 0.5, 0.1])
    pool = nets.ImagePool(2, rand=lambda: next(seq), randint=lambda n: 1)
    a, b, c, d = ([np.full(1, i)] for i in range(4))
    assert pool(a) is a and pool(b) is b            # filling: returned unchanged
    out = pool(c)                                   # rand .9 > .5: swap with slot 1
    assert out[0][0] == 1 and pool.images[1] is c
    assert pool(d) is d                             # rand .5 is not > .5
    assert nets.ImagePool(0)(a) is a
