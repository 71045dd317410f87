"""Generates the committed golden vectors (tests/golden/*.npz) from the float64 numpy oracle.

The reference ships no fixtures and TensorFlow 1.x cannot run here, so these vectors pin the
ORACLE's behaviour (inputs float32-representable, expected outputs computed in float64): they guard
the oracle against regressions and give the GPU tests a reference that does not depend on the
oracle code being present.  Re-run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import autodiff as A          # noqa: E402
from oracle import nets                   # noqa: E402
from oracle import tf_ops as T            # noqa: E402


def f32(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def ops_vectors():
    rng = np.random.default_rng(20240)
    g = {}
    d = np.float64
    # conv2d 5x5 stride 2 SAME (ops.py:64-74): odd input size exercises the (1,2) pad split
    x, w, b = f32(rng, (2, 9, 9, 3)), f32(rng, (5, 5, 3, 4), 0.2), f32(rng, (4,))
    y = T.conv2d_forward(x.astype(d), w.astype(d), 2, 2, 'SAME') + b
    dy = f32(rng, y.shape)
    g.update({'conv/x': x, 'conv/w': w, 'conv/b': b, 'conv/y': y, 'conv/dy': dy,
              'conv/dx': T.conv2d_backward_input(dy.astype(d), w.astype(d), x.shape, 2, 2, 'SAME'),
              'conv/dw': T.conv2d_backward_filter(x.astype(d), dy.astype(d), 5, 5, 2, 2, 'SAME'),
              'conv/db': T.bias_add_backward(dy.astype(d))})
    # 7x7 stride 1 VALID (cyclegan_module.py:125)
    x, w = f32(rng, (1, 12, 12, 3)), f32(rng, (7, 7, 3, 5), 0.1)
    g.update({'conv7/x': x, 'conv7/w': w, 'conv7/y': T.conv2d_forward(x.astype(d), w.astype(d), 1, 1, 'VALID')})
    # conv2d_transpose 5x5 stride 2 (ops.py:76-99): filter [kh, kw, out, in]
    x, w, b = f32(rng, (2, 4, 4, 4)), f32(rng, (5, 5, 3, 4), 0.2), f32(rng, (3,))
    y = T.conv2d_transpose_forward(x.astype(d), w.astype(d), (2, 8, 8, 3), 2, 2) + b
    dy = f32(rng, y.shape)
    g.update({'deconv/x': x, 'deconv/w': w, 'deconv/b': b, 'deconv/y': y, 'deconv/dy': dy,
              'deconv/dx': T.conv2d_transpose_backward_input(dy.astype(d), w.astype(d), 2, 2),
              'deconv/dw': T.conv2d_transpose_backward_filter(x.astype(d), dy.astype(d), 5, 5, 2, 2),
              'deconv/db': T.bias_add_backward(dy.astype(d))})
    # linear (ops.py:104-115)
    x, m, b = f32(rng, (6, 10)), f32(rng, (10, 7), 0.3), f32(rng, (7,))
    dy = f32(rng, (6, 7))
    dx, dm, db = T.linear_backward(x.astype(d), m.astype(d), dy.astype(d))
    g.update({'linear/x': x, 'linear/m': m, 'linear/b': b, 'linear/y': T.linear_forward(x.astype(d), m.astype(d), b.astype(d)),
              'linear/dy': dy, 'linear/dx': dx, 'linear/dm': dm, 'linear/db': db})
    # batch norm, training mode, followed by lrelu(0.2) (ops.py:30-44, dcgan.py:329)
    x, gamma, beta = f32(rng, (4, 5, 5, 8)) * 2 + 0.5, f32(rng, (8,)) * 0.2 + 1, f32(rng, (8,)) * 0.3
    z, mean, var, rstd = T.batch_norm_train_forward(x.astype(d), gamma.astype(d), beta.astype(d), 1e-5)
    y = T.lrelu(z)
    dy = f32(rng, x.shape)
    dy[np.abs(z) < 1e-3] = 0            # keep the comparison clear of the kink
    dz = T.lrelu_backward(z, dy.astype(d))
    dx, dg, dbt = T.batch_norm_train_backward(x.astype(d), gamma.astype(d), mean, rstd, dz)
    mm, mv = T.batch_norm_moving_update(np.zeros(8), np.ones(8), mean, var, 0.9, 100)
    g.update({'bn/x': x, 'bn/gamma': gamma, 'bn/beta': beta, 'bn/y': y, 'bn/mean': mean, 'bn/var': var,
              'bn/dy': dy, 'bn/dx': dx, 'bn/dgamma': dg, 'bn/dbeta': dbt, 'bn/moving_mean': mm, 'bn/moving_var': mv})
    # losses (dcgan.py:123-134; cyclegan_module.py:149-153)
    lg = f32(rng, (8, 1)) * 3
    for label in (0.0, 1.0):
        loss, grad = T.sigmoid_cross_entropy_mean(lg.astype(d), label)
        g.update({'bce%d/loss' % label: np.array(loss), 'bce%d/grad' % label: grad})
    a, b2 = f32(rng, (2, 4, 4, 3)), f32(rng, (2, 4, 4, 3))
    l1, gl1 = T.l1_mean(a.astype(d), b2.astype(d))
    ms, gms = T.mse_mean(a.astype(d), np.ones_like(a, dtype=d))
    g.update({'loss/logits': lg, 'loss/a': a, 'loss/b': b2, 'l1/loss': np.array(l1), 'l1/grad_b': -gl1,
              'mse1/loss': np.array(ms), 'mse1/grad': gms})
    # reflect pad (cyclegan_module.py:106)
    x = f32(rng, (1, 5, 6, 2))
    dy = f32(rng, (1, 11, 12, 2))
    g.update({'reflect/x': x, 'reflect/y': T.reflect_pad(x.astype(d), 3), 'reflect/dy': dy,
              'reflect/dx': T.reflect_pad_backward(dy.astype(d), 3)})
    # TF Adam, 3 steps (dcgan.py:154-157)
    theta, m, v = f32(rng, (37,)).astype(d), np.zeros(37), np.zeros(37)
    g['adam/theta0'] = theta.astype(np.float32)
    theta = g['adam/theta0'].astype(d)
    for t in range(1, 4):
        gr = f32(rng, (37,))
        g['adam/g%d' % t] = gr
        theta, m, v = T.adam_tf_step(theta, gr.astype(d), m, v, t, 2e-4, 0.5)
        g['adam/theta%d' % t] = theta
    return g


def dcgan_trajectory():
    """20 iterations of the reference schedule (1 D + 2 G updates, dcgan.py:262-275) of a small
    unconditional DCGAN in float64 from stored float32 weights and inputs."""
    rng = np.random.default_rng(777)
    st = A.Store(seed=777, dtype=np.float64)
    net = nets.DCGAN(st, batch_size=8, output_height=16, output_width=16, c_dim=3, z_dim=10, gf_dim=8, df_dim=8,
                     lr=0.0002, beta1=0.5)
    net.build(np.zeros((8, 16, 16, 3)), np.zeros((8, 10)))
    g = {}
    for n in list(st.values):
        v = st.values[n].astype(np.float32)
        if n.endswith(('gamma', 'beta', 'biases', 'bias')):
            v = (v + rng.standard_normal(v.shape) * 0.1).astype(np.float32)
        g['w/' + n] = v
        st.values[n] = v.astype(np.float64)
    iters = 20
    im = rng.uniform(-1, 1, (iters, 8, 16, 16, 3)).astype(np.float32)
    z = rng.uniform(-1, 1, (iters, 8, 10)).astype(np.float32)
    losses = np.zeros((iters, 3))
    for i in range(iters):
        r = net.train_iteration(im[i].astype(np.float64), z[i].astype(np.float64))
        losses[i] = [r['d_loss'], r['g_loss_1'], r['g_loss_2']]
    g.update({'images': im, 'z': z, 'losses': losses, 'names': np.array(list(st.values))})
    return g


if __name__ == '__main__':
    np.savez_compressed(os.path.join(HERE, 'ops_small.npz'), **ops_vectors())
    np.savez_compressed(os.path.join(HERE, 'dcgan_traj20.npz'), **dcgan_trajectory())
    for f in ('ops_small.npz', 'dcgan_traj20.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)), 'bytes')
