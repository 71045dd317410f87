"""Committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py).

CPU: the oracle still reproduces them (a regression guard for the oracle itself).
GPU: the HIP path, driven through the reference-shaped ops API, reproduces them without any oracle
code in the loop -- per-op values and gradients within rel 1e-4, the 20-iteration DCGAN loss series
within the stated trajectory tolerance."""
import argparse
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REL_TOL = 1e-4


def golden(name):
    return np.load(os.path.join(HERE, "golden", name))


def close(got, ref, what, tol=REL_TOL):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    err = float(np.abs(got - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
    assert err <= tol, "%s: %.3e > %.1e" % (what, err, tol)


# ------------------------------------------------------------------------------------------ CPU
def test_oracle_reproduces_the_golden_op_vectors():
    from oracle import tf_ops as T
    g = golden("ops_small.npz")
    d = np.float64
    close(T.conv2d_forward(g['conv/x'].astype(d), g['conv/w'].astype(d), 2, 2, 'SAME') + g['conv/b'], g['conv/y'], "conv y", 1e-12)
    close(T.conv2d_backward_input(g['conv/dy'].astype(d), g['conv/w'].astype(d), g['conv/x'].shape, 2, 2, 'SAME'),
          g['conv/dx'], "conv dx", 1e-12)
    close(T.conv2d_transpose_forward(g['deconv/x'].astype(d), g['deconv/w'].astype(d), (2, 8, 8, 3), 2, 2) + g['deconv/b'],
          g['deconv/y'], "deconv y", 1e-12)
    # the float32 oracle agrees with the float64 vectors to float32 accuracy
    y32 = T.conv2d_forward(g['conv/x'], g['conv/w'], 2, 2, 'SAME') + g['conv/b']
    assert y32.dtype == np.float32
    close(y32, g['conv/y'], "conv y (float32 oracle)", 1e-5)
    z, mean, var, rstd = T.batch_norm_train_forward(g['bn/x'].astype(d), g['bn/gamma'].astype(d), g['bn/beta'].astype(d), 1e-5)
    close(T.lrelu(z), g['bn/y'], "bn+lrelu y", 1e-12)
    close(var, g['bn/var'], "bn var", 1e-12)
    theta, m, v = g['adam/theta0'].astype(d), np.zeros(37), np.zeros(37)
    for t in range(1, 4):
        theta, m, v = T.adam_tf_step(theta, g['adam/g%d' % t].astype(d), m, v, t, 2e-4, 0.5)
        close(theta, g['adam/theta%d' % t], "adam step %d" % t, 1e-12)


def test_oracle_reproduces_the_golden_dcgan_trajectory():
    from oracle import autodiff as A
    from oracle import nets
    g = golden("dcgan_traj20.npz")
    st = A.Store(seed=0, dtype=np.float64)
    net = nets.DCGAN(st, batch_size=8, output_height=16, output_width=16, c_dim=3, z_dim=10, gf_dim=8, df_dim=8,
                     lr=0.0002, beta1=0.5)
    net.build(np.zeros((8, 16, 16, 3)), np.zeros((8, 10)))
    assert list(st.values) == [str(n) for n in g['names']]
    for n in st.values:
        st.values[n] = g['w/' + n].astype(np.float64)
    for i in range(5):
        r = net.train_iteration(g['images'][i].astype(np.float64), g['z'][i].astype(np.float64))
        np.testing.assert_allclose([r['d_loss'], r['g_loss_1'], r['g_loss_2']], g['losses'][i], rtol=1e-9)


# ------------------------------------------------------------------------------------------ GPU
def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _layer(build, weights, x, dy):
    """Run one reference-shaped layer on the GPU: returns (y, dx, {name: grad})."""
    import torch
    from gan_amd import ops, variables as V
    st = V.VariableStore(seed=0)
    with V.use_store(st):
        xt = _dev(x).requires_grad_(True)
        y = build(xt)
        st.load_state_dict(weights, strict=True)
        grp = st.pack('g', st.trainable_variables())
        y = build(xt)                       # again, now with the golden weights in place
        grp.begin_update()
        y.backward(_dev(dy))
        grp.end_update()
    torch.cuda.synchronize()
    return y.detach().cpu().numpy(), xt.grad.cpu().numpy(), {v.name: v.grad.cpu().numpy() for v in grp.variables}, st


@pytest.mark.gpu
def test_gpu_layers_reproduce_the_golden_op_vectors():
    import torch
    from gan_amd import ops
    g = golden("ops_small.npz")
    y, dx, gr, _ = _layer(lambda x: ops.conv2d(x, 4, name='c'), {'c/w': g['conv/w'], 'c/biases': g['conv/b']},
                          g['conv/x'], g['conv/dy'])
    close(y, g['conv/y'], "conv2d y"); close(dx, g['conv/dx'], "conv2d dx")
    close(gr['c/w'], g['conv/dw'], "conv2d dw"); close(gr['c/biases'], g['conv/db'], "conv2d db")

    y, dx, gr, _ = _layer(lambda x: ops.deconv2d(x, [2, 8, 8, 3], name='d'), {'d/w': g['deconv/w'], 'd/biases': g['deconv/b']},
                          g['deconv/x'], g['deconv/dy'])
    close(y, g['deconv/y'], "deconv2d y"); close(dx, g['deconv/dx'], "deconv2d dx")
    close(gr['d/w'], g['deconv/dw'], "deconv2d dw"); close(gr['d/biases'], g['deconv/db'], "deconv2d db")

    y, dx, gr, _ = _layer(lambda x: ops.linear(x, 7, 'l'), {'l/Matrix': g['linear/m'], 'l/bias': g['linear/b']},
                          g['linear/x'], g['linear/dy'])
    close(y, g['linear/y'], "linear y"); close(dx, g['linear/dx'], "linear dx")
    close(gr['l/Matrix'], g['linear/dm'], "linear dMatrix"); close(gr['l/bias'], g['linear/db'], "linear dbias")

    bn = ops.batch_norm(name='bn')
    y, dx, gr, st = _layer(lambda x: bn(x, act='lrelu'), {'bn/gamma': g['bn/gamma'], 'bn/beta': g['bn/beta']},
                           g['bn/x'], g['bn/dy'])
    close(y, g['bn/y'], "batch_norm+lrelu y"); close(dx, g['bn/dx'], "batch_norm dx")
    close(gr['bn/gamma'], g['bn/dgamma'], "batch_norm dgamma"); close(gr['bn/beta'], g['bn/dbeta'], "batch_norm dbeta")

    x7 = _dev(g['conv7/x'])
    from gan_amd import variables as V
    st = V.VariableStore(seed=0)
    with V.use_store(st), torch.no_grad():
        ops.conv2d(x7, 5, 7, 7, 1, 1, padding='VALID', name='c7')
        st.load_state_dict({'c7/w': g['conv7/w']}, strict=True)
        close(ops.conv2d(x7, 5, 7, 7, 1, 1, padding='VALID', name='c7').cpu().numpy(), g['conv7/y'], "7x7 VALID conv y")
        close(ops.reflect_pad(_dev(g['reflect/x']), 3).cpu().numpy(), g['reflect/y'], "reflect pad y", 1e-7)
    xr = _dev(g['reflect/x']).requires_grad_(True)
    ops.reflect_pad(xr, 3).backward(_dev(g['reflect/dy']))
    close(xr.grad.cpu().numpy(), g['reflect/dx'], "reflect pad dx")

    for label in (0, 1):
        lt = _dev(g['loss/logits']).requires_grad_(True)
        loss = ops.sigmoid_cross_entropy_mean(lt, float(label))
        ops.backward(loss)
        close(loss.detach().cpu().numpy().reshape(()), g['bce%d/loss' % label], "bce loss")
        close(lt.grad.cpu().numpy(), g['bce%d/grad' % label], "bce grad")
    bt = _dev(g['loss/b']).requires_grad_(True)
    l1 = ops.l1_mean(_dev(g['loss/a']), bt)
    ops.backward(l1)
    close(l1.detach().cpu().numpy().reshape(()), g['l1/loss'], "l1 loss"); close(bt.grad.cpu().numpy(), g['l1/grad_b'], "l1 grad")
    at = _dev(g['loss/a']).requires_grad_(True)
    ms = ops.mse_const_mean(at, 1.0)
    ops.backward(ms)
    close(ms.detach().cpu().numpy().reshape(()), g['mse1/loss'], "mse loss"); close(at.grad.cpu().numpy(), g['mse1/grad'], "mse grad")


@pytest.mark.gpu
def test_gpu_adam_reproduces_the_golden_steps():
    import torch
    from gan_amd import variables as V
    from gan_amd.optim import AdamOptimizer
    g = golden("ops_small.npz")
    st = V.VariableStore(seed=0)
    v = st.get_variable('d_theta', [37], V.constant_initializer(0.0))
    grp = st.pack('d', [v])
    v.data.copy_(_dev(g['adam/theta0']))
    opt = AdamOptimizer(2e-4, beta1=0.5).attach(grp)
    for t in range(1, 4):
        v.grad.copy_(_dev(g['adam/g%d' % t]))
        opt.adam()
        torch.cuda.synchronize()
        close(v.data.cpu().numpy(), g['adam/theta%d' % t], "adam theta after step %d" % t, 1e-6)
    assert opt.steps_taken == 3


@pytest.mark.gpu
def test_gpu_dcgan_reproduces_the_golden_trajectory():
    """Stated tolerance: |loss - golden float64 loss| <= 2e-2 over the 20 iterations."""
    import torch
    from gan_amd.dcgan import DCGAN
    g = golden("dcgan_traj20.npz")
    opts = argparse.Namespace(model='DCGAN', dataset='synthetic', batch_size=8, input_height=16, input_width=16,
                              output_height=16, output_width=16, y_dim=None, z_dim=10, gf_dim=8, df_dim=8, gfc_dim=32,
                              dfc_dim=32, c_dim=3, lr=0.0002, beta1=0.5, crop=True, nEpochs=1, train_size=1e8,
                              print_every=1, checkpoint_dir='./checkpoint', seed=0)
    model = DCGAN(None, opts, load_data=False)
    assert list(model.store.vars) == [str(n) for n in g['names']]
    model.store.load_state_dict({str(n): g['w/' + str(n)] for n in g['names']})
    worst = 0.0
    for i in range(len(g['images'])):
        out = model.train_iteration(_dev(g['images'][i]), _dev(g['z'][i]))
        got = [float(out['d_loss_real']) + float(out['d_loss_fake']), float(out['g_loss_1']), float(out['g_loss_2'])]
        worst = max(worst, float(np.abs(np.array(got) - g['losses'][i]).max()))
    print("max |loss - golden| over 20 iterations: %.3e" % worst)
    assert worst <= 2e-2
