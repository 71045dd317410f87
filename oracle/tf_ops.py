"""Numpy restatement of the TensorFlow-1.x ops the reference's layer primitives call.

Every function here is a plain forward or backward formula on NHWC arrays; no
state, no autodiff.  The reference call sites are cited per function
(paths relative to the reference root).  Test infrastructure only.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------
# padding rules
# ----------------------------------------------------------------------------
def same_pad(in_size, k, s):
    """TF 'SAME': out = ceil(in/s); total pad = max((out-1)*s + k - in, 0);
    the smaller half goes before (top/left), the rest after."""
    out = int(math.ceil(float(in_size) / float(s)))
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv_geometry(in_h, in_w, k_h, k_w, d_h, d_w, padding):
    """Output size and (top, bottom, left, right) pads of tf.nn.conv2d."""
    if padding == 'SAME':
        oh, pt, pb = same_pad(in_h, k_h, d_h)
        ow, pl, pr = same_pad(in_w, k_w, d_w)
    elif padding == 'VALID':
        oh = (in_h - k_h) // d_h + 1
        ow = (in_w - k_w) // d_w + 1
        pt = pb = pl = pr = 0
    else:
        raise ValueError("padding must be 'SAME' or 'VALID', got %r" % (padding,))
    return oh, ow, (pt, pb, pl, pr)


def _im2col(xp, k_h, k_w, d_h, d_w, oh, ow):
    """Windows of a padded NHWC array as a view [N, oh, ow, k_h, k_w, C]."""
    n, _, _, c = xp.shape
    sn, sh, sw, sc = xp.strides
    return np.lib.stride_tricks.as_strided(
        xp, shape=(n, oh, ow, k_h, k_w, c),
        strides=(sn, sh * d_h, sw * d_w, sh, sw, sc), writeable=False)


# ----------------------------------------------------------------------------
# tf.nn.conv2d  (ops.py:68) and its two gradients
# ----------------------------------------------------------------------------
def conv2d_forward(x, w, d_h, d_w, padding):
    """y[n,i,j,co] = sum_{a,b,ci} x[n, i*d_h+a-pt, j*d_w+b-pl, ci] * w[a,b,ci,co].
    x NHWC, w HWIO ([k_h,k_w,Cin,Cout], ops.py:66)."""
    n, h, wd, ci = x.shape
    k_h, k_w, wci, co = w.shape
    assert wci == ci, "filter in-channels %d != input channels %d" % (wci, ci)
    oh, ow, (pt, pb, pl, pr) = conv_geometry(h, wd, k_h, k_w, d_h, d_w, padding)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = _im2col(xp, k_h, k_w, d_h, d_w, oh, ow).reshape(n * oh * ow, k_h * k_w * ci)
    y = cols @ w.reshape(k_h * k_w * ci, co)
    return y.reshape(n, oh, ow, co)


def conv2d_backward_input(dy, w, in_shape, d_h, d_w, padding):
    """Gradient of conv2d_forward w.r.t. x (TF Conv2DBackpropInput)."""
    n, h, wd, ci = in_shape
    k_h, k_w, wci, co = w.shape
    assert wci == ci
    oh, ow, (pt, pb, pl, pr) = conv_geometry(h, wd, k_h, k_w, d_h, d_w, padding)
    assert dy.shape == (n, oh, ow, co), (dy.shape, (n, oh, ow, co))
    dcols = (dy.reshape(n * oh * ow, co) @ w.reshape(k_h * k_w * ci, co).T)
    dcols = dcols.reshape(n, oh, ow, k_h, k_w, ci)
    dxp = np.zeros((n, h + pt + pb, wd + pl + pr, ci), dtype=dy.dtype)
    for a in range(k_h):
        for b in range(k_w):
            dxp[:, a:a + d_h * oh:d_h, b:b + d_w * ow:d_w, :] += dcols[:, :, :, a, b, :]
    return dxp[:, pt:pt + h, pl:pl + wd, :]


def conv2d_backward_filter(x, dy, k_h, k_w, d_h, d_w, padding):
    """Gradient of conv2d_forward w.r.t. w (TF Conv2DBackpropFilter)."""
    n, h, wd, ci = x.shape
    co = dy.shape[-1]
    oh, ow, (pt, pb, pl, pr) = conv_geometry(h, wd, k_h, k_w, d_h, d_w, padding)
    assert dy.shape == (n, oh, ow, co)
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    cols = _im2col(xp, k_h, k_w, d_h, d_w, oh, ow).reshape(n * oh * ow, k_h * k_w * ci)
    dw = cols.T @ dy.reshape(n * oh * ow, co)
    return dw.reshape(k_h, k_w, ci, co)


# ----------------------------------------------------------------------------
# tf.nn.conv2d_transpose (ops.py:85-86) and its gradients
# ----------------------------------------------------------------------------
def conv2d_transpose_forward(x, w, output_shape, d_h, d_w, padding='SAME'):
    """tf.nn.conv2d_transpose: the input-gradient of a conv2d that maps
    `output_shape` -> x.shape, evaluated at x.  Filter [k_h,k_w,Cout,Cin]
    (ops.py:80-81), i.e. the HWIO filter of that conv2d."""
    n, oh, ow, co = output_shape
    assert w.shape[2] == co and w.shape[3] == x.shape[3], (w.shape, x.shape, output_shape)
    return conv2d_backward_input(x, w, (n, oh, ow, co), d_h, d_w, padding)


def conv2d_transpose_backward_input(dy, w, d_h, d_w, padding='SAME'):
    """d/dx of conv2d_transpose = the forward conv2d applied to dy."""
    return conv2d_forward(dy, w, d_h, d_w, padding)


def conv2d_transpose_backward_filter(x, dy, k_h, k_w, d_h, d_w, padding='SAME'):
    """d/dw of conv2d_transpose: filter gradient of the underlying conv2d with
    dy playing the conv input and x the conv output-gradient."""
    return conv2d_backward_filter(dy, x, k_h, k_w, d_h, d_w, padding)


# ----------------------------------------------------------------------------
# bias (ops.py:70-71, 93-94), linear (ops.py:104-115)
# ----------------------------------------------------------------------------
def bias_add(x, b):
    return x + b


def bias_add_backward(dy):
    return dy.reshape(-1, dy.shape[-1]).sum(axis=0)


def linear_forward(x, matrix, bias):
    return x @ matrix + bias


def linear_backward(x, matrix, dy):
    return dy @ matrix.T, x.T @ dy, dy.sum(axis=0)


# ----------------------------------------------------------------------------
# tf.contrib.layers.batch_norm (ops.py:38-44)
# ----------------------------------------------------------------------------
def batch_norm_train_forward(x, gamma, beta, eps):
    """Training mode: per-channel mean and BIASED variance over every axis but
    the last; y = (x-mean)*rsqrt(var+eps)*gamma + beta."""
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    mean = x2.mean(axis=0)
    var = ((x2 - mean) ** 2).mean(axis=0)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xhat = (x2 - mean) * rstd
    y = xhat * gamma + beta
    return y.reshape(x.shape), mean, var, rstd


def batch_norm_moving_update(moving_mean, moving_var, mean, var, momentum, count,
                             unbiased_moving_var=False):
    """moving <- momentum*moving + (1-momentum)*batch, executed inside every
    training-mode forward (updates_collections=None, ops.py:40).  The fused TF
    kernel feeds a Bessel-corrected variance here, the unfused one the biased
    variance; `unbiased_moving_var` selects between them (default: biased)."""
    dt = moving_mean.dtype.type
    v = var * dt(count / max(count - 1.0, 1.0)) if unbiased_moving_var else var
    new_mean = moving_mean * dt(momentum) + mean * dt(1.0 - momentum)
    new_var = moving_var * dt(momentum) + v * dt(1.0 - momentum)
    return new_mean, new_var


def batch_norm_eval_forward(x, gamma, beta, moving_mean, moving_var, eps):
    rstd = 1.0 / np.sqrt(moving_var + x.dtype.type(eps))
    return (x - moving_mean) * rstd * gamma + beta


def batch_norm_train_backward(x, gamma, mean, rstd, dy):
    c = x.shape[-1]
    x2 = x.reshape(-1, c)
    dy2 = dy.reshape(-1, c)
    m = x2.shape[0]
    xhat = (x2 - mean) * rstd
    dgamma = (dy2 * xhat).sum(axis=0)
    dbeta = dy2.sum(axis=0)
    dx = (gamma * rstd) * (dy2 - dbeta / m - xhat * (dgamma / m))
    return dx.reshape(x.shape).astype(x.dtype), dgamma, dbeta


# ----------------------------------------------------------------------------
# activations (ops.py:101-102; dcgan.py:334,368-385,409; text2image.py:198)
# ----------------------------------------------------------------------------
def lrelu(x, leak=0.2):
    return np.maximum(x, x.dtype.type(leak) * x)


def lrelu_backward(x, dy, leak=0.2):
    # tf.maximum(x, leak*x): the first argument wins ties, so slope 1 at x == 0.
    return np.where(x >= 0, dy, x.dtype.type(leak) * dy)


def relu(x):
    return np.maximum(x, 0)


def relu_backward(x, dy):
    return np.where(x > 0, dy, 0).astype(dy.dtype)


def tanh_backward(y, dy):
    return dy * (1 - y * y)


def sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def sigmoid_backward(y, dy):
    return dy * y * (1 - y)


# ----------------------------------------------------------------------------
# losses (dcgan.py:123-134, text2image.py:86-90, cyclegan_module.py:149-156)
# ----------------------------------------------------------------------------
def sigmoid_cross_entropy_mean(logits, label):
    """reduce_mean(max(x,0) - x*z + log1p(exp(-|x|))) and its gradient."""
    x = logits
    z = x.dtype.type(label)
    per = np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    loss = per.mean(dtype=x.dtype)
    grad = (sigmoid(x) - z) / x.dtype.type(x.size)
    return loss, grad


def l1_mean(a, b):
    d = a - b
    return np.abs(d).mean(dtype=a.dtype), np.sign(d) / a.dtype.type(a.size)


def mse_mean(a, b):
    d = a - b
    return (d * d).mean(dtype=a.dtype), (2 * d) / a.dtype.type(a.size)


# ----------------------------------------------------------------------------
# tf.pad(..., "REFLECT") (cyclegan_module.py:106,108,124,143)
# ----------------------------------------------------------------------------
def reflect_pad(x, p):
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)), mode='reflect')


def reflect_pad_backward(dy, p):
    n, hp, wp, c = dy.shape
    h, w = hp - 2 * p, wp - 2 * p
    idx_h = np.pad(np.arange(h), (p, p), mode='reflect')
    idx_w = np.pad(np.arange(w), (p, p), mode='reflect')
    tmp = np.zeros((n, h, wp, c), dtype=dy.dtype)
    np.add.at(tmp, (slice(None), idx_h), dy)
    dx = np.zeros((n, h, w, c), dtype=dy.dtype)
    np.add.at(dx, (slice(None), slice(None), idx_w), tmp)
    return dx


# ----------------------------------------------------------------------------
# conv_cond_concat (ops.py:57-62)
# ----------------------------------------------------------------------------
def conv_cond_concat(x, y):
    n, h, w, _ = x.shape
    yb = np.broadcast_to(y.reshape(n, 1, 1, -1), (n, h, w, y.shape[-1]))
    return np.concatenate([x, yb], axis=3)


def conv_cond_concat_backward(dy, cx):
    return dy[..., :cx], dy[..., cx:].sum(axis=(1, 2))


# ----------------------------------------------------------------------------
# tf.train.AdamOptimizer (dcgan.py:154-157)
# ----------------------------------------------------------------------------
def adam_tf_step(theta, g, m, v, t, lr, beta1, beta2=0.999, eps=1e-8):
    """TF-1.x Adam, step t (1-based): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; theta -= lr_t*m/(sqrt(v)+eps).
    eps sits OUTSIDE the bias correction (unlike torch.optim.Adam)."""
    dt = theta.dtype.type
    lr_t = dt(lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
    m = dt(beta1) * m + dt(1.0 - beta1) * g
    v = dt(beta2) * v + dt(1.0 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + dt(eps))
    return theta, m, v


# ----------------------------------------------------------------------------
# initialisers (ops.py:66-70, 81-82, 108-111)
# ----------------------------------------------------------------------------
def truncated_normal(rng, shape, stddev):
    """tf.truncated_normal_initializer: N(0, stddev), resampled beyond 2 stddev."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def random_normal(rng, shape, stddev):
    return (rng.standard_normal(shape) * stddev).astype(np.float32)
