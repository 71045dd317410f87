"""conv / deconv with the following activation fused into the store, and the gated input-gradient entry points
that hand a layer the gradient in front of its activation: against the float64 oracle through the C ABI, and the
Python-level protocol (ops.conv2d(act=...) -> consumer conv) against the unfused composition."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

ACTS = [(1, 0.0, 1.0, 0.0), (2, 0.2, 1.0, 0.0), (3, 0.0, 1.0, 0.0), (3, 0.0, 0.5, 0.5), (4, 0.0, 1.0, 0.0)]


def _act(z, act, param):
    if act == 1:
        return np.maximum(z, 0)
    if act == 2:
        return np.maximum(z, param * z)
    if act == 3:
        return np.tanh(z)
    return 1 / (1 + np.exp(-z))


def _dact(z, act, param):
    if act == 1:
        return (z > 0).astype(np.float64)
    if act == 2:
        return np.where(z >= 0, 1.0, param)
    if act == 3:
        return 1 - np.tanh(z) ** 2
    s = 1 / (1 + np.exp(-z))
    return s * (1 - s)


# n, h, w, cb, cs, k, s: a split-K case on the bf16 path, a direct-epilogue case, a thin (3-channel) big side, odd channels
CASES = [(8, 16, 16, 64, 128, 5, 2), (2, 9, 13, 8, 12, 5, 2), (2, 12, 12, 3, 16, 5, 2), (3, 14, 14, 21, 10, 3, 1)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("act,param,scale,shift", ACTS)
def test_fused_activation_entry_points(case, act, param, scale, shift):
    from gan_amd import _lib as L
    from gpu_util import assert_close, dev, geom_for, host
    n, h, w_, cb, cs, k, s = case
    g = geom_for(n, h, w_, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w_, cb)).astype(np.float32)
    xs = rng.standard_normal((n, g.hs, g.ws, cs)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cb, cs)) * 0.1).astype(np.float32)
    bc, bd = rng.standard_normal(cs).astype(np.float32), rng.standard_normal(cb).astype(np.float32)
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    st = L.stream()
    d = np.float64
    x_d, xs_d, wt_d, bc_d, bd_d = dev(x), dev(xs), dev(wt), dev(bc), dev(bd)     # kept alive across the launches
    # conv2d forward + activation
    y = torch.empty(n, g.hs, g.ws, cs, device='cuda')
    L.check(L.lib.gan_conv2d_fwd_act(gp, L.ptr(x_d), L.ptr(wt_d), L.ptr(bc_d), L.ptr(y), act, param, scale, shift, wp, wn, st))
    z = T.conv2d_forward(x.astype(d), wt.astype(d), s, s, 'SAME') + bc
    # tanh / sigmoid compress a sum of magnitude ~10 into [-1, 1]: the rounding of the sum, not of the output, sets
    # the absolute error, so the elementwise bound (relative to the output's own scale) applies to relu / lrelu only
    ew = act in (1, 2)
    assert_close(host(y), _act(z, act, param) * scale + shift, "conv2d_fwd_act", elementwise=ew)
    # conv2d_transpose forward + activation
    yt = torch.empty(n, h, w_, cb, device='cuda')
    L.check(L.lib.gan_deconv2d_fwd_act(gp, L.ptr(xs_d), L.ptr(wt_d), L.ptr(bd_d), L.ptr(yt), act, param, scale, shift, wp, wn, st))
    zt = T.conv2d_transpose_forward(xs.astype(d), wt.astype(d), (n, h, w_, cb), s, s) + bd
    assert_close(host(yt), _act(zt, act, param) * scale + shift, "deconv2d_fwd_act", elementwise=ew)
    # gated input gradients: the gate is the activated tensor of the layer below (pre-activations kept clear of the kinks)
    zg_big = rng.standard_normal((n, h, w_, cb))
    zg_big[np.abs(zg_big) < 1e-3] = 0.5
    gate_big = (_act(zg_big, act, param) * scale + shift).astype(np.float32)
    dx = torch.empty(n, h, w_, cb, device='cuda')
    gb_d = dev(gate_big)
    L.check(L.lib.gan_conv2d_dgrad_gated(gp, L.ptr(xs_d), L.ptr(wt_d), L.ptr(dx), L.ptr(gb_d), act, param, scale, shift,
                                         wp, wn, st))
    want = T.conv2d_backward_input(xs.astype(d), wt.astype(d), x.shape, s, s, 'SAME') * _dact(zg_big, act, param) * scale
    # the derivative is rebuilt from the float32 gate: tanh / sigmoid lose relative accuracy where they saturate
    assert_close(host(dx), want, "conv2d_dgrad_gated", tol=1e-4 if act in (1, 2) else 5e-4, elementwise=act in (1, 2))
    zg_small = rng.standard_normal((n, g.hs, g.ws, cs))
    zg_small[np.abs(zg_small) < 1e-3] = 0.5
    gate_small = (_act(zg_small, act, param) * scale + shift).astype(np.float32)
    dxs = torch.empty(n, g.hs, g.ws, cs, device='cuda')
    gs_d = dev(gate_small)
    L.check(L.lib.gan_deconv2d_dgrad_gated(gp, L.ptr(x_d), L.ptr(wt_d), L.ptr(dxs), L.ptr(gs_d), act, param, scale,
                                           shift, wp, wn, st))
    want = T.conv2d_forward(x.astype(d), wt.astype(d), s, s, 'SAME') * _dact(zg_small, act, param) * scale
    assert_close(host(dxs), want, "deconv2d_dgrad_gated", tol=1e-4 if act in (1, 2) else 5e-4, elementwise=act in (1, 2))


@pytest.mark.parametrize("consumer", ["conv", "deconv", "other"])
def test_fused_activation_protocol_equals_the_unfused_composition(consumer):
    """ops.conv2d(act='lrelu') followed by a conv / deconv (gated input gradient) or by another op (fallback:
    activation-backward kernel) gives the gradients of lrelu(conv2d(..)) composed from separate ops."""
    from gan_amd import ops, variables as V
    from gan_amd.optim import AdamOptimizer
    torch.manual_seed(3)
    x = torch.randn(4, 16, 16, 8, device='cuda')
    results = []
    for fused in (True, False):
        store = V.VariableStore(device='cuda', seed=11)
        with V.use_store(store):
            def net(inp):
                with ops.variable_scope("n"):
                    if fused:
                        h = ops.conv2d(inp, 16, name='c0', act='lrelu')
                    else:
                        h = ops.lrelu(ops.conv2d(inp, 16, name='c0'))
                    if consumer == "conv":
                        o = ops.conv2d(h, 8, 3, 3, 1, 1, name='c1')
                    elif consumer == "deconv":
                        o = ops.deconv2d(h, [4, 16, 16, 8], name='c1')
                    else:
                        o = ops.linear(h.reshape(4, -1), 5, 'c1')
                    return ops.mse_const_mean(o, 0.3)
            loss = net(x)
            group = store.pack('n', store.trainable_variables())
            opt = AdamOptimizer(1e-3, beta1=0.5).attach(group)
            opt.begin()
            xin = x.clone().requires_grad_(True)
            loss = net(xin)
            ops.backward(loss)
            opt.finish_grads()
            torch.cuda.synchronize()
            results.append((float(loss.detach()), group.grads.clone(), xin.grad.clone()))
    (l0, g0, dx0), (l1, g1, dx1) = results
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l1))
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    assert float((dx0 - dx1).abs().max()) <= 1e-5 * float(dx1.abs().max())


@pytest.mark.parametrize("case", [(2, 16, 16, 32, 48, 3), (1, 20, 12, 3, 8, 7), (1, 12, 12, 16, 3, 7), (2, 9, 11, 6, 10, 3)])
def test_reflect_padding_folded_into_the_gather(case):
    """gan_conv2d_fwd_reflect / _wgrad_reflect on the unpadded image against the float64 oracle's
    conv2d(pad(x, REFLECT), VALID) and its filter gradient; ops.conv2d(padding='REFLECT') against
    ops.conv2d(ops.reflect_pad(x), padding='VALID') including the input gradient."""
    from gan_amd import _lib as L
    from gan_amd import ops, variables as V
    from gan_amd.optim import AdamOptimizer
    from gpu_util import assert_close, dev, host
    n, h, w_, cb, cs, k = case
    p = (k - 1) // 2
    g = L.conv_geom(n, h, w_, cb, h, w_, cs, k, k, 1, 1, p, p)
    gp = ctypes.byref(g)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, h, w_, cb)).astype(np.float32)
    wt = (rng.standard_normal((k, k, cb, cs)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cs).astype(np.float32)
    dy = rng.standard_normal((n, h, w_, cs)).astype(np.float32)
    d = np.float64
    xp = np.pad(x.astype(d), ((0, 0), (p, p), (p, p), (0, 0)), mode='reflect')
    x_d, w_d, b_d, dy_d = dev(x), dev(wt), dev(b), dev(dy)
    y = torch.empty(n, h, w_, cs, device='cuda')
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    L.check(L.lib.gan_conv2d_fwd_reflect(gp, L.ptr(x_d), L.ptr(w_d), L.ptr(b_d), L.ptr(y), wp, wn, L.stream()))
    assert_close(host(y), T.conv2d_forward(xp, wt.astype(d), 1, 1, 'VALID') + b, "conv2d_fwd_reflect")
    dw, db = torch.empty(k, k, cb, cs, device='cuda'), torch.empty(cs, device='cuda')
    L.check(L.lib.gan_conv2d_wgrad_reflect(gp, L.ptr(x_d), L.ptr(dy_d), L.ptr(dw), L.ptr(db), 0, wp, wn, L.stream()))
    assert_close(host(dw), T.conv2d_backward_filter(xp, dy.astype(d), k, k, 1, 1, 'VALID'), "conv2d_wgrad_reflect")
    assert_close(host(db), dy.astype(d).sum(axis=(0, 1, 2)), "bias gradient")
    # input gradient: the transposed contraction on the padded geometry and the fold back onto the image in one call,
    # against the oracle (gradient of the padded image, mirrored border added onto its sources) and against the two-step
    # form (gan_conv2d_dgrad on the padded geometry + gan_reflect_pad_bwd), with and without split-K slabs
    dxp_ref = T.conv2d_backward_input(dy.astype(d), wt.astype(d), xp.shape, 1, 1, 'VALID')
    dx_ref = np.zeros((n, h, w_, cb))
    src_y = np.pad(np.arange(h), (p, p), mode='reflect')
    src_x = np.pad(np.arange(w_), (p, p), mode='reflect')
    for a_, sy in enumerate(src_y):
        for b_, sx in enumerate(src_x):
            dx_ref[:, sy, sx, :] += dxp_ref[:, a_, b_, :]
    gpad = L.conv_geom(n, h + 2 * p, w_ + 2 * p, cb, h, w_, cs, k, k, 1, 1, 0, 0)
    from gpu_util import forced
    for splits in (None, 1, 3):
        with forced(None, splits):
            dx = torch.empty(n, h, w_, cb, device='cuda')
            wq, wqn = L.workspace(int(L.lib.gan_conv2d_dgrad_reflect_workspace_bytes(gp)))
            L.check(L.lib.gan_conv2d_dgrad_reflect(gp, L.ptr(dy_d), L.ptr(w_d), L.ptr(dx), wq, wqn, L.stream()))
            assert_close(host(dx), dx_ref, "conv2d_dgrad_reflect (splits %s)" % splits)
            dxp = torch.empty(n, h + 2 * p, w_ + 2 * p, cb, device='cuda')
            dx2 = torch.empty_like(dx)
            wq2, wqn2 = L.workspace(int(L.lib.gan_conv_workspace_bytes(ctypes.byref(gpad))))
            L.check(L.lib.gan_conv2d_dgrad(ctypes.byref(gpad), L.ptr(dy_d), L.ptr(w_d), L.ptr(dxp), wq2, wqn2, L.stream()))
            L.check(L.lib.gan_reflect_pad_bwd(L.ptr(dxp), L.ptr(dx2), n, h, w_, cb, p, L.stream()))
            torch.cuda.synchronize()
            assert float((dx - dx2).abs().max()) <= 1e-5 * float(dx2.abs().max())
    assert L.lib.gan_conv2d_dgrad_reflect(gp, L.ptr(dy_d), L.ptr(w_d), L.ptr(dx), wq, 16, L.stream()) != 0   # workspace too small
    # the Python op, both ways
    res = []
    for folded in (True, False):
        store = V.VariableStore(device='cuda', seed=3)
        with V.use_store(store):
            def net(inp):
                if folded:
                    o = ops.conv2d(inp, cs, k, k, 1, 1, padding='REFLECT', name='c')
                else:
                    o = ops.conv2d(ops.reflect_pad(inp, p), cs, k, k, 1, 1, padding='VALID', name='c')
                return ops.mse_const_mean(o, 0.1)
            net(x_d)
            group = store.pack('n', store.trainable_variables())
            opt = AdamOptimizer(1e-3, beta1=0.5).attach(group)
            opt.begin()
            xin = x_d.clone().requires_grad_(True)
            loss = net(xin)
            ops.backward(loss)
            opt.finish_grads()
            torch.cuda.synchronize()
            res.append((float(loss.detach()), group.grads.clone(), xin.grad.clone()))
    (l0, g0, dx0), (l1, g1, dx1) = res
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l1))
    assert float((g0 - g1).abs().max()) <= 1e-5 * float(g1.abs().max())
    assert float((dx0 - dx1).abs().max()) <= 1e-5 * float(dx1.abs().max())


@pytest.mark.parametrize("gated_first", [True, False])
def test_fused_activation_output_with_two_consumers_is_refused(gated_first):
    """The output of conv2d(.., act=) may feed ONE differentiable consumer chain.  With a gated consumer (a conv2d that
    applies the activation's derivative while it stores its input gradient) and a second consumer, the gradient
    that reaches the producing layer is a sum of one term that went through the derivative and one that did not:
    the backward pass must raise, whichever of the two gradients autograd sees first (when the gated one comes first
    and autograd holds its only reference, the second is added into it IN PLACE: same address, moved version)."""
    from gan_amd import ops, variables as V
    from gan_amd.optim import AdamOptimizer
    store = V.VariableStore(device='cuda', seed=4)
    x = torch.randn(2, 8, 8, 8, device='cuda')
    with V.use_store(store):
        def net(inp):
            y = ops.conv2d(inp, 8, 3, 3, 1, 1, name='p', act='lrelu')
            if gated_first:          # the consumer created LAST runs its backward FIRST
                b = ops.mse_const_mean(y, 0.0)
                a = ops.mse_const_mean(ops.conv2d(y, 8, 3, 3, 1, 1, name='c'), 0.1)
            else:
                a = ops.mse_const_mean(ops.conv2d(y, 8, 3, 3, 1, 1, name='c'), 0.1)
                b = ops.mse_const_mean(y, 0.0)
            return [a, b]
        with torch.no_grad():
            net(x)
        group = store.pack('n', store.trainable_variables())
        opt = AdamOptimizer(1e-3, beta1=0.5).attach(group)
        opt.begin()
        losses = net(x.clone().requires_grad_(True))
        with pytest.raises(RuntimeError, match="feeds a gated consumer AND something else"):
            ops.backward(losses)
        torch.cuda.synchronize()


# n, h, w, cin, cout, k, padding: stride-1 convolutions with few channels on one side
TAPS_CASES = [
    (1, 40, 36, 64, 3, 7, 'REFLECT'),     # the CycleGAN image head: forward, filter / bias / input gradient by taps
    (2, 9, 11, 16, 3, 7, 'REFLECT'),      # an image barely larger than the padding
    (2, 20, 24, 32, 2, 5, 'SAME'),        # zero padding
    (1, 40, 36, 3, 64, 7, 'REFLECT'),     # the CycleGAN stem: input gradient by taps
    (2, 18, 16, 1, 32, 5, 'SAME'),
]


@pytest.mark.parametrize("case", TAPS_CASES)
def test_few_channel_convolutions_by_taps(case):
    """ops.conv2d on layers with at most 4 channels on one side and 25 taps or more, computed as a dense product plus a
    sum over taps (gan_taps_*), against the same call on the general kernels (ops.TAPS_PATH off): loss, every
    variable's gradient (two uses of the layer, so the second accumulates) and the input gradient."""
    from gan_amd import ops, variables as V
    from gan_amd.optim import AdamOptimizer
    n, h, w_, cin, cout, k, padding = case
    torch.manual_seed(9)
    x_d = torch.randn(n, h, w_, cin, device='cuda')
    bias_d = torch.rand(cout, device='cuda') * 2 - 1
    res = []
    try:
        for taps in (True, False):
            ops.TAPS_PATH[0] = taps
            store = V.VariableStore(device='cuda', seed=3)
            with V.use_store(store):
                def net(inp):
                    with ops.variable_scope("l") as scope:
                        a = ops.conv2d(inp, cout, k, k, 1, 1, padding=padding, name='c')
                        scope.reuse_variables()
                        b = ops.conv2d(inp * 0.5, cout, k, k, 1, 1, padding=padding, name='c')
                    return [ops.mse_const_mean(a, 0.1), ops.mse_const_mean(b, -0.2)]
                net(x_d)
                store.vars['l/c/biases'].data.copy_(bias_d)
                group = store.pack('n', store.trainable_variables())
                opt = AdamOptimizer(1e-3, beta1=0.5).attach(group)
                opt.begin()
                xin = x_d.clone().requires_grad_(True)
                losses = net(xin)
                ops.backward(losses)
                opt.finish_grads()
                torch.cuda.synchronize()
                res.append(([float(v.detach()) for v in losses], {v.name: v.grad.clone() for v in group.variables}, xin.grad.clone()))
    finally:
        ops.TAPS_PATH[0] = True
    (l0, g0, dx0), (l1, g1, dx1) = res
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    for name in g1:
        assert torch.isfinite(g0[name]).all()
        assert float((g0[name] - g1[name]).abs().max()) <= 2e-5 * float(g1[name].abs().max()), name
    assert float((dx0 - dx1).abs().max()) <= 2e-5 * float(dx1.abs().max())
