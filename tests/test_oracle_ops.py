"""Pins the numpy oracle: hand-computed known answers, float64 finite differences,
an independent cross-check against torch CPU ops, and algebraic identities.
The reference holds no golden vectors for this path, so these are the pins."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf_ops as T

RNG = np.random.default_rng(1234)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def torch_conv2d_tf(x, w, s, padding):
    """TF conv2d through torch: explicit asymmetric pad, then a VALID conv."""
    n, h, wd, ci = x.shape
    kh, kw = w.shape[:2]
    _, _, (pt, pb, pl, pr) = T.conv_geometry(h, wd, kh, kw, s, s, padding)
    xt = F.pad(t(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xt, t(w).permute(3, 2, 0, 1), stride=s)
    return y.permute(0, 2, 3, 1).numpy()


def torch_conv2d_transpose_tf(x, w, out_shape, s):
    """TF conv2d_transpose through torch: full transposed conv, then the TF crop."""
    n, oh, ow, co = out_shape
    kh, kw = w.shape[:2]
    _, pt, _ = T.same_pad(oh, kh, s)
    _, pl, _ = T.same_pad(ow, kw, s)
    # torch weight for conv_transpose2d: [Cin, Cout, kh, kw]; TF filter [kh,kw,Cout,Cin]
    full = F.conv_transpose2d(t(x).permute(0, 3, 1, 2), t(w).permute(3, 2, 0, 1), stride=s)
    need_h, need_w = pt + oh, pl + ow
    full = F.pad(full, (0, max(0, need_w - full.shape[3]), 0, max(0, need_h - full.shape[2])))
    return full[:, :, pt:pt + oh, pl:pl + ow].permute(0, 2, 3, 1).numpy()


# --------------------------------------------------------------------------- known answers
def test_same_pad_rule():
    assert T.same_pad(64, 5, 2) == (32, 1, 2)
    assert T.same_pad(28, 5, 2) == (14, 1, 2)
    assert T.same_pad(7, 5, 2) == (4, 2, 2)
    assert T.same_pad(128, 3, 2) == (64, 0, 1)
    assert T.same_pad(4, 1, 1) == (4, 0, 0)
    assert T.same_pad(32, 5, 1) == (32, 2, 2)


def test_conv2d_known_answer_ones_filter():
    # 1x4x4x1 input 0..15, 3x3 ones filter, stride 2, SAME: pad (0,1); windows at (0,0),(0,2),(2,0),(2,2)
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    w = np.ones((3, 3, 1, 1))
    y = T.conv2d_forward(x, w, 2, 2, 'SAME')
    expect = np.array([[0 + 1 + 2 + 4 + 5 + 6 + 8 + 9 + 10, 2 + 3 + 6 + 7 + 10 + 11],
                       [8 + 9 + 10 + 12 + 13 + 14, 10 + 11 + 14 + 15]], dtype=np.float64)
    np.testing.assert_array_equal(y[0, :, :, 0], expect)


def test_conv2d_transpose_known_answer_impulse_crop():
    # 2x2 -> 4x4, k=5, s=2: pad_before = 1, so x[0,0] lands its filter centre tap (1,1) at out (0,0)
    x = np.zeros((1, 2, 2, 1))
    x[0, 0, 0, 0] = 1.0
    w = np.arange(25, dtype=np.float64).reshape(5, 5, 1, 1)
    y = T.conv2d_transpose_forward(x, w, (1, 4, 4, 1), 2, 2)
    np.testing.assert_array_equal(y[0, :, :, 0], w[1:5, 1:5, 0, 0])
    x[...] = 0
    x[0, 1, 1, 0] = 1.0           # lands taps (a,b) at out (2+a-1, 2+b-1): a,b in 0..2 visible
    y = T.conv2d_transpose_forward(x, w, (1, 4, 4, 1), 2, 2)
    expect = np.zeros((4, 4))
    expect[1:4, 1:4] = w[0:3, 0:3, 0, 0]
    np.testing.assert_array_equal(y[0, :, :, 0], expect)


def test_batch_norm_known_answer():
    x = np.array([[1.0, 10.0], [3.0, 30.0]])
    y, mean, var, rstd = T.batch_norm_train_forward(x, np.ones(2), np.zeros(2), 0.0)
    np.testing.assert_allclose(mean, [2.0, 20.0])
    np.testing.assert_allclose(var, [1.0, 100.0])          # biased
    np.testing.assert_allclose(y, [[-1, -1], [1, 1]])
    mm, mv = T.batch_norm_moving_update(np.zeros(2), np.ones(2), mean, var, 0.9, 2)
    np.testing.assert_allclose(mm, [0.2, 2.0])
    np.testing.assert_allclose(mv, [1.0, 10.9])
    _, mv_u = T.batch_norm_moving_update(np.zeros(2), np.ones(2), mean, var, 0.9, 2, unbiased_moving_var=True)
    np.testing.assert_allclose(mv_u, [0.9 + 0.1 * 2.0, 0.9 + 0.1 * 200.0])


def test_adam_tf_known_answer():
    theta, m, v = T.adam_tf_step(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, 0.1, 0.5)
    # m = .25, v = .00025, lr_t = .1*sqrt(.001)/.5
    lr_t = 0.1 * np.sqrt(1 - 0.999) / 0.5
    np.testing.assert_allclose(m, [0.25])
    np.testing.assert_allclose(v, [0.00025])
    np.testing.assert_allclose(theta, [1.0 - lr_t * 0.25 / (np.sqrt(0.00025) + 1e-8)])


def test_sigmoid_ce_known_answer():
    x = np.array([[0.0], [2.0], [-3.0]])
    loss, grad = T.sigmoid_cross_entropy_mean(x, 1.0)
    expect = np.mean([np.log(2.0), np.log1p(np.exp(-2.0)), 3 + np.log1p(np.exp(-3.0))])
    np.testing.assert_allclose(loss, expect)
    np.testing.assert_allclose(grad[:, 0], (1 / (1 + np.exp(-x[:, 0])) - 1) / 3)


def test_lrelu_relu_tie_rules():
    x = np.array([-1.0, 0.0, 2.0])
    np.testing.assert_array_equal(T.lrelu_backward(x, np.ones(3)), [0.2, 1.0, 1.0])
    np.testing.assert_array_equal(T.relu_backward(x, np.ones(3)), [0.0, 0.0, 1.0])


def test_truncated_normal_bounds():
    w = T.truncated_normal(np.random.default_rng(0), (5, 5, 3, 64), 0.02)
    assert w.dtype == np.float32 and np.abs(w).max() <= 0.04 + 1e-9
    assert 0.015 < w.std() < 0.02          # truncation shrinks the std to ~0.88 sigma


# --------------------------------------------------------------------------- torch cross-checks
CONV_CASES = [
    # n, h, w, ci, co, k, s, padding
    (2, 8, 8, 3, 4, 5, 2, 'SAME'),
    (2, 7, 7, 5, 6, 5, 2, 'SAME'),       # odd size: pad (2,2)
    (1, 28, 28, 11, 11, 5, 2, 'SAME'),   # MNIST d_h0
    (2, 4, 4, 24, 16, 1, 1, 'SAME'),     # text2image 1x1
    (1, 12, 12, 4, 8, 3, 2, 'SAME'),     # cyclegan 3x3 s2: pad (0,1)
    (1, 10, 10, 4, 4, 3, 1, 'VALID'),
    (1, 14, 14, 3, 8, 7, 1, 'VALID'),
    (1, 8, 8, 6, 2, 5, 1, 'SAME'),       # cyclegan D s1
]


@pytest.mark.parametrize("n,h,w,ci,co,k,s,padding", CONV_CASES)
def test_conv2d_matches_torch(n, h, w, ci, co, k, s, padding):
    x = RNG.standard_normal((n, h, w, ci))
    wt = RNG.standard_normal((k, k, ci, co))
    y = T.conv2d_forward(x, wt, s, s, padding)
    np.testing.assert_allclose(y, torch_conv2d_tf(x, wt, s, padding), rtol=1e-10, atol=1e-10)
    # gradients vs torch autograd
    xt = t(x).requires_grad_(True)
    wtt = t(wt).requires_grad_(True)
    _, _, (pt, pb, pl, pr) = T.conv_geometry(h, w, k, k, s, s, padding)
    yt = F.conv2d(F.pad(xt.permute(0, 3, 1, 2), (pl, pr, pt, pb)), wtt.permute(3, 2, 0, 1), stride=s).permute(0, 2, 3, 1)
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    np.testing.assert_allclose(T.conv2d_backward_input(dy, wt, x.shape, s, s, padding), xt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(T.conv2d_backward_filter(x, dy, k, k, s, s, padding), wtt.grad.numpy(), rtol=1e-9, atol=1e-9)


DECONV_CASES = [
    # n, hs, ws, ci, co, k, s, out_h, out_w
    (2, 4, 4, 8, 4, 5, 2, 8, 8),
    (2, 4, 4, 6, 3, 5, 2, 7, 7),        # odd output (MNIST 7 <- 4 is not used, but 7x7 <- ceil)
    (1, 7, 7, 10, 8, 5, 2, 14, 14),     # MNIST g_h2
    (1, 8, 8, 4, 2, 3, 2, 16, 16),      # cyclegan 3x3 s2
]


@pytest.mark.parametrize("n,hs,ws,ci,co,k,s,oh,ow", DECONV_CASES)
def test_conv2d_transpose_matches_torch(n, hs, ws, ci, co, k, s, oh, ow):
    x = RNG.standard_normal((n, hs, ws, ci))
    wt = RNG.standard_normal((k, k, co, ci))
    y = T.conv2d_transpose_forward(x, wt, (n, oh, ow, co), s, s)
    np.testing.assert_allclose(y, torch_conv2d_transpose_tf(x, wt, (n, oh, ow, co), s), rtol=1e-10, atol=1e-10)
    # identity: conv2d_transpose == <.,.>-adjoint of conv2d
    u = RNG.standard_normal((n, oh, ow, co))
    lhs = np.sum(T.conv2d_forward(u, wt, s, s, 'SAME') * x)
    rhs = np.sum(u * y)
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10)
    # gradients by adjointness with the conv2d gradients already checked against torch
    dy = RNG.standard_normal(y.shape)
    np.testing.assert_allclose(T.conv2d_transpose_backward_input(dy, wt, s, s),
                               T.conv2d_forward(dy, wt, s, s, 'SAME'))
    dw = T.conv2d_transpose_backward_filter(x, dy, k, k, s, s)
    eps = 1e-6
    for idx in [(0, 0, 0, 0), (k - 1, k - 1, co - 1, ci - 1), (k // 2, 1, 0, ci - 1)]:
        wp = wt.copy()
        wp[idx] += eps
        num = np.sum((T.conv2d_transpose_forward(x, wp, (n, oh, ow, co), s, s) - y) * dy) / eps
        np.testing.assert_allclose(dw[idx], num, rtol=1e-5, atol=1e-6)


def test_batch_norm_matches_torch_and_fd():
    x = RNG.standard_normal((4, 3, 3, 5)) * 2 + 1
    gamma, beta = RNG.standard_normal(5), RNG.standard_normal(5)
    y, mean, var, rstd = T.batch_norm_train_forward(x, gamma, beta, 1e-5)
    xt = t(x).requires_grad_(True)
    gt, bt = t(gamma).requires_grad_(True), t(beta).requires_grad_(True)
    yt = F.batch_norm(xt.permute(0, 3, 1, 2), None, None, gt, bt, True, 0.1, 1e-5).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-10, atol=1e-10)
    dy = RNG.standard_normal(y.shape)
    yt.backward(t(dy))
    dx, dg, db = T.batch_norm_train_backward(x, gamma, mean, rstd, dy)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dg, gt.grad.numpy(), rtol=1e-9)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9)
    # eval mode
    mm, mv = RNG.standard_normal(5), RNG.random(5) + 0.5
    ye = T.batch_norm_eval_forward(x, gamma, beta, mm, mv, 1e-5)
    yte = F.batch_norm(t(x).permute(0, 3, 1, 2), t(mm), t(mv), t(gamma), t(beta), False, 0.1, 1e-5).permute(0, 2, 3, 1)
    np.testing.assert_allclose(ye, yte.numpy(), rtol=1e-10, atol=1e-10)


def test_batch_norm_n1_is_instance_norm():
    x = RNG.standard_normal((1, 6, 6, 4))
    y, _, _, _ = T.batch_norm_train_forward(x, np.ones(4), np.zeros(4), 1e-5)
    yt = F.instance_norm(t(x).permute(0, 3, 1, 2), eps=1e-5).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-10, atol=1e-10)


def test_linear_losses_reflect_concat_against_torch():
    x, m, b = RNG.standard_normal((4, 7)), RNG.standard_normal((7, 3)), RNG.standard_normal(3)
    xt, mt, bt = (t(a).requires_grad_(True) for a in (x, m, b))
    yt = xt @ mt + bt
    dy = RNG.standard_normal((4, 3))
    yt.backward(t(dy))
    dx, dm, db = T.linear_backward(x, m, dy)
    np.testing.assert_allclose(T.linear_forward(x, m, b), yt.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(dm, mt.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-12)
    # BCE with logits
    lg = RNG.standard_normal((6, 1)) * 3
    for z in (0.0, 1.0):
        lt = t(lg).requires_grad_(True)
        lo = F.binary_cross_entropy_with_logits(lt, torch.full_like(lt, z))
        lo.backward()
        loss, grad = T.sigmoid_cross_entropy_mean(lg, z)
        np.testing.assert_allclose(loss, lo.item(), rtol=1e-12)
        np.testing.assert_allclose(grad, lt.grad.numpy(), rtol=1e-10, atol=1e-12)
    # l1 / mse
    a, c = RNG.standard_normal((2, 3, 3, 2)), RNG.standard_normal((2, 3, 3, 2))
    at = t(a).requires_grad_(True)
    F.l1_loss(at, t(c)).backward()
    np.testing.assert_allclose(T.l1_mean(a, c)[1], at.grad.numpy(), rtol=1e-12)
    at = t(a).requires_grad_(True)
    F.mse_loss(at, t(c)).backward()
    np.testing.assert_allclose(T.mse_mean(a, c)[1], at.grad.numpy(), rtol=1e-12)
    # reflect pad
    for p in (1, 3):
        xr = RNG.standard_normal((1, 6, 5, 2))
        xrt = t(xr).requires_grad_(True)
        yr = F.pad(xrt.permute(0, 3, 1, 2), (p, p, p, p), mode='reflect').permute(0, 2, 3, 1)
        np.testing.assert_allclose(T.reflect_pad(xr, p), yr.detach().numpy())
        dyr = RNG.standard_normal(yr.shape)
        yr.backward(t(dyr))
        np.testing.assert_allclose(T.reflect_pad_backward(dyr, p), xrt.grad.numpy(), rtol=1e-12)
    # conv_cond_concat
    xx, yy = RNG.standard_normal((2, 3, 3, 4)), RNG.standard_normal((2, 1, 1, 5))
    cc = T.conv_cond_concat(xx, yy)
    assert cc.shape == (2, 3, 3, 9)
    np.testing.assert_array_equal(cc[:, 1, 2, 4:], yy[:, 0, 0, :])
    dxx, dyy = T.conv_cond_concat_backward(np.ones_like(cc), 4)
    assert dxx.shape == xx.shape and np.all(dyy == 9)


def test_adam_tf_differs_from_torch_adam_only_by_eps_placement():
    theta = RNG.standard_normal(10)
    g = RNG.standard_normal(10)
    th, m, v = theta.copy(), np.zeros(10), np.zeros(10)
    p = torch.nn.Parameter(t(theta.copy()))
    opt = torch.optim.Adam([p], lr=2e-4, betas=(0.5, 0.999), eps=0.0)
    for step in range(1, 4):
        th, m, v = T.adam_tf_step(th, g, m, v, step, 2e-4, 0.5, eps=0.0)
        p.grad = t(g.copy())
        opt.step()
    np.testing.assert_allclose(th, p.detach().numpy(), rtol=1e-10)
