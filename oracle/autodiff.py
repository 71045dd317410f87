"""Tiny reverse-mode tape + name-scoped variable store for the numpy oracle.

The store restates the part of TensorFlow's variable_scope / get_variable
contract the reference relies on: variables are created by name under the
current scope and shared when the same name is requested again
(dcgan.py:323-325, 413-414; cyclegan_module.py:9-14), and variable groups are
selected by substring of the full name (dcgan.py:146-147, cyclegan.py:112-115).
Test infrastructure only.
"""
import contextlib

import numpy as np

from . import tf_ops as T


class Store(object):
    """name -> float array.  BN moving statistics live here too (non-trainable)."""

    def __init__(self, seed=0, dtype=np.float32, unbiased_moving_var=False):
        self.rng = np.random.default_rng(seed)
        self.dtype = dtype
        self.values = {}
        self.trainable = []          # creation order, like tf.trainable_variables()
        self._scope = []
        self.unbiased_moving_var = unbiased_moving_var

    @contextlib.contextmanager
    def scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return '/'.join(self._scope + [name])

    def get(self, name, shape, init, trainable=True):
        full = self.full_name(name)
        if full not in self.values:
            self.values[full] = np.asarray(init(tuple(shape))).astype(self.dtype)
            if trainable:
                self.trainable.append(full)
        v = self.values[full]
        assert v.shape == tuple(shape), "variable %s: shape %s requested, %s stored" % (full, shape, v.shape)
        return full

    def select(self, substr):
        return [n for n in self.trainable if substr in n]

    # initialisers -----------------------------------------------------------
    def truncated_normal(self, stddev):
        return lambda shape: T.truncated_normal(self.rng, shape, stddev)

    def random_normal(self, stddev):
        return lambda shape: T.random_normal(self.rng, shape, stddev)

    @staticmethod
    def constant(value):
        return lambda shape: np.full(shape, value, dtype=np.float64)


class Node(object):
    __slots__ = ('v', 'g', 'back', 'needs_grad')

    def __init__(self, v, back=None, needs_grad=False):
        self.v = v
        self.g = None
        self.back = back
        self.needs_grad = needs_grad

    def accumulate(self, g):
        if not self.needs_grad:
            return
        self.g = g if self.g is None else self.g + g


class Tape(object):
    def __init__(self, store):
        self.store = store
        self.nodes = []
        self.var_nodes = {}

    def const(self, arr):
        return Node(np.asarray(arr, dtype=self.store.dtype))

    def var(self, full_name):
        if full_name not in self.var_nodes:
            self.var_nodes[full_name] = Node(self.store.values[full_name], needs_grad=True)
        return self.var_nodes[full_name]

    def op(self, value, parents, back):
        needs = any(p.needs_grad for p in parents)
        n = Node(value, back if needs else None, needs)
        if needs:
            self.nodes.append(n)
        return n

    def backward(self, loss):
        loss.g = np.ones_like(loss.v)
        for n in reversed(self.nodes):
            if n.g is not None and n.back is not None:
                n.back(n.g)

    def grads(self, names):
        out = {}
        for name in names:
            n = self.var_nodes.get(name)
            out[name] = (np.zeros_like(self.store.values[name])
                         if n is None or n.g is None else n.g)
        return out


# ----------------------------------------------------------------------------
# layer primitives, signatures following models/ops.py with (tape, ...) prepended
# ----------------------------------------------------------------------------
def conv2d(tp, input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02,
           padding='SAME', name="conv2d"):
    """ops.py:64-74"""
    st = tp.store
    cin = input_.v.shape[-1]
    with st.scope(name):
        wn = st.get('w', [k_h, k_w, cin, output_dim], st.truncated_normal(stddev))
        bn = st.get('biases', [output_dim], st.constant(0.0))
    w, b = tp.var(wn), tp.var(bn)
    y = T.conv2d_forward(input_.v, w.v, d_h, d_w, padding) + b.v

    def back(g):
        b.accumulate(T.bias_add_backward(g))
        w.accumulate(T.conv2d_backward_filter(input_.v, g, k_h, k_w, d_h, d_w, padding))
        if input_.needs_grad:
            input_.accumulate(T.conv2d_backward_input(g, w.v, input_.v.shape, d_h, d_w, padding))
    return tp.op(y, [input_, w, b], back)


def deconv2d(tp, input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02,
             name="deconv2d"):
    """ops.py:76-99"""
    st = tp.store
    cin = input_.v.shape[-1]
    with st.scope(name):
        wn = st.get('w', [k_h, k_w, output_shape[-1], cin], st.random_normal(stddev))
        bn = st.get('biases', [output_shape[-1]], st.constant(0.0))
    w, b = tp.var(wn), tp.var(bn)
    y = T.conv2d_transpose_forward(input_.v, w.v, tuple(output_shape), d_h, d_w) + b.v

    def back(g):
        b.accumulate(T.bias_add_backward(g))
        w.accumulate(T.conv2d_transpose_backward_filter(input_.v, g, k_h, k_w, d_h, d_w))
        if input_.needs_grad:
            input_.accumulate(T.conv2d_transpose_backward_input(g, w.v, d_h, d_w))
    return tp.op(y, [input_, w, b], back)


def linear(tp, input_, output_size, scope_name=None, stddev=0.02, bias_start=0.0):
    """ops.py:104-115"""
    st = tp.store
    with st.scope(scope_name or "Linear"):
        mn = st.get('Matrix', [input_.v.shape[1], output_size], st.random_normal(stddev))
        bn = st.get('bias', [output_size], st.constant(bias_start))
    m, b = tp.var(mn), tp.var(bn)
    y = T.linear_forward(input_.v, m.v, b.v)

    def back(g):
        dx, dm, db = T.linear_backward(input_.v, m.v, g)
        m.accumulate(dm)
        b.accumulate(db)
        input_.accumulate(dx)
    return tp.op(y, [input_, m, b], back)


class batch_norm(object):
    """ops.py:30-44.  Variables are created under the CALLER's scope + name."""

    def __init__(self, epsilon=1e-5, momentum=0.9, name="batch_norm"):
        self.epsilon = epsilon
        self.momentum = momentum
        self.name = name

    def __call__(self, tp, x, train=True):
        st = tp.store
        c = x.v.shape[-1]
        with st.scope(self.name):
            bn = st.get('beta', [c], st.constant(0.0))
            gn = st.get('gamma', [c], st.constant(1.0))
            mmn = st.get('moving_mean', [c], st.constant(0.0), trainable=False)
            mvn = st.get('moving_variance', [c], st.constant(1.0), trainable=False)
        gamma, beta = tp.var(gn), tp.var(bn)
        if not train:
            y = T.batch_norm_eval_forward(x.v, gamma.v, beta.v, st.values[mmn], st.values[mvn], self.epsilon)
            return tp.op(y, [], None)
        y, mean, var, rstd = T.batch_norm_train_forward(x.v, gamma.v, beta.v, self.epsilon)
        count = x.v.size // c
        st.values[mmn], st.values[mvn] = T.batch_norm_moving_update(
            st.values[mmn], st.values[mvn], mean, var, self.momentum, count, st.unbiased_moving_var)

        def back(g):
            dx, dgamma, dbeta = T.batch_norm_train_backward(x.v, gamma.v, mean, rstd, g)
            gamma.accumulate(dgamma)
            beta.accumulate(dbeta)
            x.accumulate(dx)
        return tp.op(y, [x, gamma, beta], back)


def instance_norm(tp, x, name="instance_norm", epsilon=1e-5):
    """Extension (absent from the reference): batch_norm(train=True) per sample."""
    st = tp.store
    n = x.v.shape[0]
    c = x.v.shape[-1]
    with st.scope(name):
        bn = st.get('beta', [c], st.constant(0.0))
        gn = st.get('gamma', [c], st.constant(1.0))
    gamma, beta = tp.var(gn), tp.var(bn)
    outs, saved = [], []
    for i in range(n):
        y, mean, var, rstd = T.batch_norm_train_forward(x.v[i:i + 1], gamma.v, beta.v, epsilon)
        outs.append(y)
        saved.append((mean, rstd))

    def back(g):
        dxs = []
        for i in range(n):
            dx, dg, db = T.batch_norm_train_backward(x.v[i:i + 1], gamma.v, saved[i][0], saved[i][1], g[i:i + 1])
            gamma.accumulate(dg)
            beta.accumulate(db)
            dxs.append(dx)
        x.accumulate(np.concatenate(dxs, axis=0))
    return tp.op(np.concatenate(outs, axis=0), [x, gamma, beta], back)


def lrelu(tp, x, leak=0.2):
    return tp.op(T.lrelu(x.v, leak), [x], lambda g: x.accumulate(T.lrelu_backward(x.v, g, leak)))


def relu(tp, x):
    return tp.op(T.relu(x.v), [x], lambda g: x.accumulate(T.relu_backward(x.v, g)))


def tanh(tp, x):
    y = np.tanh(x.v)
    return tp.op(y, [x], lambda g: x.accumulate(T.tanh_backward(y, g)))


def sigmoid(tp, x):
    y = T.sigmoid(x.v)
    return tp.op(y, [x], lambda g: x.accumulate(T.sigmoid_backward(y, g)))


def affine(tp, x, scale, shift):
    """x*scale + shift with python scalars (text2image.py:198: tanh(h4)/2. + 0.5)."""
    dt = x.v.dtype.type
    return tp.op(x.v * dt(scale) + dt(shift), [x], lambda g: x.accumulate(g * dt(scale)))


def add(tp, a, b):
    def back(g):
        a.accumulate(g)
        b.accumulate(g)
    return tp.op(a.v + b.v, [a, b], back)


def reshape(tp, x, shape):
    old = x.v.shape
    return tp.op(x.v.reshape(shape), [x], lambda g: x.accumulate(g.reshape(old)))


def concat(tp, nodes, axis):
    """ops.py:23-28"""
    sizes = [n.v.shape[axis] for n in nodes]
    offs = np.cumsum([0] + sizes)

    def back(g):
        for i, n in enumerate(nodes):
            sl = [slice(None)] * g.ndim
            sl[axis] = slice(offs[i], offs[i + 1])
            n.accumulate(g[tuple(sl)])
    return tp.op(np.concatenate([n.v for n in nodes], axis=axis), nodes, back)


def conv_cond_concat(tp, x, y):
    """ops.py:57-62.  y is [N,1,1,Y]."""
    cx = x.v.shape[-1]

    def back(g):
        dx, dy = T.conv_cond_concat_backward(g, cx)
        x.accumulate(dx)
        y.accumulate(dy.reshape(y.v.shape))
    return tp.op(T.conv_cond_concat(x.v, y.v), [x, y], back)


def tile_spatial(tp, v, h, w):
    """tf.expand_dims x2 + tf.tile [1,h,w,1] (text2image.py:213-215)."""
    n, c = v.v.shape

    def back(g):
        v.accumulate(g.sum(axis=(1, 2)))
    return tp.op(np.broadcast_to(v.v.reshape(n, 1, 1, c), (n, h, w, c)).copy(), [v], back)


def reflect_pad(tp, x, p):
    return tp.op(T.reflect_pad(x.v, p), [x], lambda g: x.accumulate(T.reflect_pad_backward(g, p)))


def sigmoid_ce_mean(tp, logits, label):
    loss, grad = T.sigmoid_cross_entropy_mean(logits.v, label)
    return tp.op(np.asarray(loss), [logits], lambda g: logits.accumulate(grad * g))


def l1_mean(tp, a, b):
    loss, grad = T.l1_mean(a.v, b.v)

    def back(g):
        a.accumulate(grad * g)
        b.accumulate(-grad * g)
    return tp.op(np.asarray(loss), [a, b], back)


def mse_mean(tp, a, b_value):
    """reduce_mean((a - const)**2) with a constant target (cyclegan_module.py:152-153)."""
    b = np.full_like(a.v, b_value)
    loss, grad = T.mse_mean(a.v, b)
    return tp.op(np.asarray(loss), [a], lambda g: a.accumulate(grad * g))


def scale(tp, x, s):
    dt = x.v.dtype.type
    return tp.op(x.v * dt(s), [x], lambda g: x.accumulate(g * dt(s)))


class Adam(object):
    """tf.train.AdamOptimizer(lr, beta1).minimize(loss, var_list) state for one group."""

    def __init__(self, store, names, lr, beta1, beta2=0.999, eps=1e-8):
        self.store, self.names = store, list(names)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = {n: np.zeros_like(store.values[n]) for n in self.names}
        self.v = {n: np.zeros_like(store.values[n]) for n in self.names}

    def step(self, grads):
        self.t += 1
        for n in self.names:
            self.store.values[n], self.m[n], self.v[n] = T.adam_tf_step(
                self.store.values[n], grads[n], self.m[n], self.v[n], self.t,
                self.lr, self.beta1, self.beta2, self.eps)
