"""Batch-norm reductions that ride on a neighbouring contraction's launch (gan_conv_fwd_bn_stats,
gan_conv_dgrad_bn_sums, gan_bn_bwd_tiled, gan_conv_wgrad_colsum) against the separate entry points, through the
C ABI: bit for bit, except the bias gradient that comes from column-sum partials (same terms, other grouping)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

OPS = ("conv2d", "deconv2d", "reflect")


@pytest.fixture(scope="module")
def L():
    from gan_amd import _lib
    return _lib


def _ws(L, g):
    n = max(int(L.lib.gan_conv_workspace_bytes(ctypes.byref(g))), int(L.lib.gan_conv2d_dgrad_reflect_workspace_bytes(ctypes.byref(g))))
    return L.workspace(n)


def _geom(op, n, hw, cb, cs, k, s):
    from gpu_util import geom_for
    return geom_for(n, hw, hw, cb, cs, k, s, 'REFLECT' if op == "reflect" else 'SAME')


def _bn_ws(L, groups, rows, c):
    return L.workspace(int(L.lib.gan_bn_workspace_bytes(groups, rows, c)))


# op, n, big side, cb, cs, k, stride, groups: outputs with c % 64 == 0 and > 1024 rows per group
FWD_CASES = [
    ("conv2d", 8, 32, 16, 64, 5, 2, 1),      # 2048 rows
    ("conv2d", 8, 32, 16, 128, 5, 2, 2),     # two stacked passes of 1024 + ... rows each -> 1024 rows: refused below
    ("conv2d", 12, 32, 32, 128, 5, 2, 3),    # three stacked passes of 1024 rows each: no decomposition either
    ("conv2d", 16, 32, 8, 64, 3, 2, 2),      # 2048 rows per group
    ("deconv2d", 4, 32, 64, 32, 5, 2, 1),    # output = big side: 4096 rows x 64
    ("deconv2d", 6, 32, 128, 16, 3, 2, 3),   # 2048 rows per group x 128
    ("reflect", 2, 40, 32, 64, 3, 1, 1),     # 3200 rows
    ("reflect", 2, 36, 64, 128, 3, 1, 2),    # instance-norm style: one group per image, 1296 rows
    ("conv2d", 64, 32, 16, 128, 5, 2, 1),    # 16384 rows: row blocks of 128 rows
    ("conv2d", 50, 32, 8, 256, 3, 2, 2),     # 6400 rows per group, row blocks of 200 rows
    ("deconv2d", 20, 32, 128, 16, 3, 2, 1),  # 20480 rows x 128, row blocks of 160
]


def _fwd_calls(L, op):
    lib = L.lib
    return {"conv2d": (L.OP_CONV2D, lib.gan_conv2d_fwd), "deconv2d": (L.OP_DECONV2D, lib.gan_deconv2d_fwd),
            "reflect": (L.OP_CONV2D_REFLECT, lib.gan_conv2d_fwd_reflect)}[op]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("splits", [1, 2, 5, 20])
@pytest.mark.parametrize("ride_knob", [0, 2])
def test_forward_statistics_riding_on_the_slab_reduction_are_bit_identical(L, case, splits, ride_knob):
    """conv / deconv / reflect conv + training batch norm (with activation, residual and moving averages): the
    statistics taken by the launch that sums the split-K slabs (or, without slabs / with the wide reduction / with the
    knob, by a launch of their own behind the contraction) give the same y, z, saved statistics and moving averages
    as gan_conv*_fwd followed by gan_bn_fwd_train."""
    op, n, hw, cb, cs, k, s, groups = case
    g = _geom(op, n, hw, cb, cs, k, s)
    gp = ctypes.byref(g)
    out_big = op == "deconv2d"
    xs = (n, g.hs, g.ws, cs) if out_big else (n, hw, hw, cb)
    ys = (n, hw, hw, cb) if out_big else (n, g.hs, g.ws, cs)
    c = ys[-1]
    rows = ys[0] * ys[1] * ys[2] // groups
    nb = int(L.lib.gan_bn_tile_blocks(groups, rows, c, 0))
    torch.manual_seed(11)
    x = torch.randn(xs, device='cuda')
    w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    bias = torch.randn(c, device='cuda')
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda')
    res = torch.randn(ys, device='cuda')
    code, fn = _fwd_calls(L, op)
    st = L.stream()
    L.knob_set("force_splits", splits)
    wp, wn = _ws(L, g)

    def run(ride):
        y, z = torch.empty(ys, device='cuda'), torch.empty(ys, device='cuda')
        mean, rstd = torch.empty(groups * c, device='cuda'), torch.empty(groups * c, device='cuda')
        mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        common = (L.ptr(gamma), L.ptr(beta), L.ptr(mm), L.ptr(mv), L.ptr(res), L.ptr(z), L.ptr(mean), L.ptr(rstd), groups, rows, c,
                  1e-5, 0.9, 1, L.ACT_LRELU, 0.2)
        if ride:
            partial = torch.full((2 * groups * nb * c,), float('nan'), device='cuda')
            L.check(L.lib.gan_conv_fwd_bn_stats(code, gp, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), groups, L.ptr(partial), wp, wn, st))
            L.check(L.lib.gan_bn_fwd_train_from_stats(L.ptr(y), *common, L.ptr(partial), st))
        else:
            L.check(fn(gp, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), wp, wn, st))
            bp, bn_ = _bn_ws(L, groups, rows, c)
            L.check(L.lib.gan_bn_fwd_train(L.ptr(y), *common, bp, bn_, st))
        torch.cuda.synchronize()
        return y, z, mean, rstd, mm, mv

    if nb == 0:       # (a shape the one-launch batch norm serves: the riding entry point refuses it)
        y = torch.empty(ys, device='cuda')
        partial = torch.empty(16, device='cuda')
        assert L.lib.gan_conv_fwd_bn_stats(code, gp, L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(y), groups, L.ptr(partial), wp, wn, st) != 0
        assert b"gan_bn_tile_blocks" in L.lib.gan_last_error()
        return
    plain = run(False)
    L.knob_set("no_bn_ride", ride_knob)
    riding = run(True)
    for name, a, b in zip(("y", "z", "mean", "rstd", "moving_mean", "moving_var"), plain, riding):
        assert torch.isfinite(a).all() and torch.equal(a, b), "%s differs: max |diff| %g at %d elements" % (
            name, float((a - b).abs().max()), int((a != b).sum()))


# op, n, big side, cb, cs, k, stride, groups of the batch norm whose output is the op's INPUT
BWD_CASES = [
    ("conv2d", 8, 32, 64, 16, 5, 2, 1),       # dx = big side: 8192 rows x 64
    ("conv2d", 6, 32, 128, 32, 3, 2, 3),      # 2048 rows per group
    ("deconv2d", 8, 32, 16, 64, 5, 2, 1),     # dx = small side: 2048 rows x 64
    ("deconv2d", 16, 16, 8, 128, 5, 2, 4),    # 256 rows per group: served when groups are stacked
    ("reflect", 2, 40, 64, 32, 3, 1, 1),      # fold of the mirrored border + sums, 3200 rows
    ("reflect", 2, 36, 128, 64, 3, 1, 2),
    ("reflect", 1, 40, 64, 8, 7, 1, 1),       # 7 x 7: three mirrored rows / columns onto one
    ("conv2d", 32, 32, 128, 16, 5, 2, 1),     # 32768 rows: row blocks of 256 rows
    ("deconv2d", 50, 32, 8, 256, 5, 2, 2),    # 6400 rows per group, row blocks of 200
    ("conv2d", 40, 32, 64, 8, 3, 2, 1),       # 40960 rows x 64: row blocks of 160
]


@pytest.mark.parametrize("case", BWD_CASES)
@pytest.mark.parametrize("splits", [1, 3, 20])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("ride_knob", [0, 2])
def test_backward_sums_riding_on_the_input_gradient_are_bit_identical(L, case, splits, act, ride_knob):
    """Input gradient of a conv / deconv / reflect conv whose input was a batch norm's output, then that batch norm's
    backward: the block sums taken by the launch that finishes the input gradient give the same dx, dgamma, dbeta as
    gan_*_dgrad followed by gan_bn_bwd; the column sums gan_bn_bwd_tiled leaves give the bias gradient of the layer
    in front to rounding."""
    op, n, hw, cb, cs, k, s, groups = case
    g = _geom(op, n, hw, cb, cs, k, s)
    gp = ctypes.byref(g)
    in_big = op != "deconv2d"
    xs = (n, hw, hw, cb) if in_big else (n, g.hs, g.ws, cs)        # the op's input = the batch norm's tensors
    ys = (n, g.hs, g.ws, cs) if in_big else (n, hw, hw, cb)
    c = xs[-1]
    rows = xs[0] * xs[1] * xs[2] // groups
    nb = int(L.lib.gan_bn_tile_blocks(groups, rows, c, 1))
    assert nb > 0
    torch.manual_seed(23)
    bn_x = torch.randn(xs, device='cuda')
    dy = torch.randn(ys, device='cuda')
    w = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1
    mean, rstd = torch.randn(groups * c, device='cuda') * 0.1, torch.rand(groups * c, device='cuda') + 0.5
    code = {"conv2d": L.OP_CONV2D, "deconv2d": L.OP_DECONV2D, "reflect": L.OP_CONV2D_REFLECT}[op]
    dgrad = {"conv2d": L.lib.gan_conv2d_dgrad, "deconv2d": L.lib.gan_deconv2d_dgrad, "reflect": L.lib.gan_conv2d_dgrad_reflect}[op]
    st = L.stream()
    L.knob_set("force_splits", splits)
    wp, wn = _ws(L, g)

    def run(ride):
        dx = torch.empty(xs, device='cuda')
        dxn = torch.empty(xs, device='cuda')
        dg, db = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')      # accumulated onto
        tail = (1, groups, rows, c, act, 0.2)
        colsum = None
        if ride:
            partial = torch.full((2 * groups * nb * c,), float('nan'), device='cuda')
            colsum = torch.full((groups * nb * c,), float('nan'), device='cuda')
            saved = L.BnSaved(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), groups, act, 0.2)
            L.check(L.lib.gan_conv_dgrad_bn_sums(code, gp, L.ptr(dy), L.ptr(w), L.ptr(dx), ctypes.byref(saved), L.ptr(partial), wp, wn, st))
            L.check(L.lib.gan_bn_bwd_tiled(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dxn),
                                           L.ptr(dg), L.ptr(db), *tail, L.ptr(partial), 1, L.ptr(colsum), st))
        else:
            L.check(dgrad(gp, L.ptr(dy), L.ptr(w), L.ptr(dx), wp, wn, st))
            bp, bn_ = _bn_ws(L, groups, rows, c)
            L.check(L.lib.gan_bn_bwd(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(dxn),
                                     L.ptr(dg), L.ptr(db), *tail, bp, bn_, st))
        torch.cuda.synchronize()
        return (dx, dxn, dg, db), colsum

    plain, _ = run(False)
    L.knob_set("no_bn_ride", ride_knob)
    riding, colsum = run(True)
    for a, b in zip(plain, riding):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # the column-sum partials add up to the column sums of the batch norm's input gradient
    dxn = plain[1].double().reshape(-1, c)
    got = colsum.double().reshape(groups * nb, c).sum(0)
    torch.testing.assert_close(got, dxn.sum(0), rtol=0, atol=1e-5 * float(dxn.abs().sum(0).max()))


@pytest.mark.parametrize("op", OPS)
@pytest.mark.parametrize("splits", [1, 4])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_bias_gradient_from_column_sum_partials(L, op, splits, accumulate):
    """gan_conv_wgrad_colsum: the weight gradient is bit-identical to gan_*_wgrad's, the bias gradient is the sum of
    the given partials (carried by the contraction launch, its slab reduction, or a launch of its own)."""
    n, hw, cb, cs, k, s = (4, 32, 32, 64, 3, 1) if op == "reflect" else (4, 32, 32, 64, 5, 2)
    g = _geom(op, n, hw, cb, cs, k, s)
    gp = ctypes.byref(g)
    torch.manual_seed(31)
    big = torch.randn(n, hw, hw, cb, device='cuda')
    small = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    x, dy = (small, big) if op == "deconv2d" else (big, small)
    c = dy.shape[-1]
    blocks = 37
    partial = torch.randn(blocks, c, device='cuda')
    code = {"conv2d": L.OP_CONV2D, "deconv2d": L.OP_DECONV2D, "reflect": L.OP_CONV2D_REFLECT}[op]
    wgrad = {"conv2d": L.lib.gan_conv2d_wgrad, "deconv2d": L.lib.gan_deconv2d_wgrad, "reflect": L.lib.gan_conv2d_wgrad_reflect}[op]
    st = L.stream()
    L.knob_set("force_splits", splits)
    wp, wn = _ws(L, g)
    torch.manual_seed(5)
    dw0, db0 = torch.randn(k, k, cb, cs, device='cuda'), torch.randn(c, device='cuda')
    dw1, db1 = dw0.clone(), db0.clone()
    start = db0.clone()
    L.check(wgrad(gp, L.ptr(x), L.ptr(dy), L.ptr(dw0), L.ptr(db0), accumulate, wp, wn, st))
    L.check(L.lib.gan_conv_wgrad_colsum(code, gp, L.ptr(x), L.ptr(dy), L.ptr(dw1), L.ptr(db1), accumulate, L.ptr(partial), blocks,
                                        wp, wn, st))
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    want = partial.double().sum(0) + (start.double() if accumulate else 0.0)
    torch.testing.assert_close(db1.double(), want, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("shape", [(1, 4096, 256), (2, 1296, 128), (1, 16384, 64), (3, 1300, 64)])
@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("layout", ["a_first", "b_first", "in_place"])
def test_fan_out_sum_with_backward_sums_is_bit_identical(L, shape, act, layout):
    """gan_add_bn_sums (the gradient of a batch norm's output with two consumers, and that norm's backward block sums
    in the same launch) followed by gan_bn_bwd_tiled against gan_add_fwd followed by gan_bn_bwd: every tensor bit for
    bit, whichever addend lies first in memory, and with the sum stored over the first addend."""
    groups, rows, c = shape
    nb = int(L.lib.gan_bn_tile_blocks(groups, rows, c, 1))
    assert nb > 0
    n = groups * rows * c
    torch.manual_seed(41)
    pool = torch.randn(2 * n, device='cuda')
    a, b = (pool[:n], pool[n:]) if layout != "b_first" else (pool[n:], pool[:n])
    bn_x = torch.randn(n, device='cuda')
    gamma, beta = torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda') * 0.1
    mean, rstd = torch.randn(groups * c, device='cuda') * 0.1, torch.rand(groups * c, device='cuda') + 0.5
    st = L.stream()
    tail = (1, groups, rows, c, act, 0.2)
    # separate launches
    out0, dx0 = torch.empty(n, device='cuda'), torch.empty(n, device='cuda')
    dg0, db0 = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')
    L.check(L.lib.gan_add_fwd(L.ptr(a), L.ptr(b), L.ptr(out0), n, st))
    bp, bn_ = _bn_ws(L, groups, rows, c)
    L.check(L.lib.gan_bn_bwd(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(out0), L.ptr(dx0),
                             L.ptr(dg0), L.ptr(db0), *tail, bp, bn_, st))
    # one launch for the sum and the block sums
    out1 = a.clone() if layout == "in_place" else torch.full((n,), float('nan'), device='cuda')
    first = out1 if layout == "in_place" else a
    dx1 = torch.empty(n, device='cuda')
    dg1, db1 = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')
    partial = torch.full((2 * groups * nb * c,), float('nan'), device='cuda')
    saved = L.BnSaved(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), groups, act, 0.2)
    L.check(L.lib.gan_add_bn_sums(L.ptr(first), L.ptr(b), L.ptr(out1), ctypes.byref(saved), rows, c, L.ptr(partial), st))
    L.check(L.lib.gan_bn_bwd_tiled(L.ptr(bn_x), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(rstd), L.ptr(out1), L.ptr(dx1),
                                   L.ptr(dg1), L.ptr(db1), *tail, L.ptr(partial), 1, None, st))
    torch.cuda.synchronize()
    for u, v in ((out0, out1), (dx0, dx1), (dg0, dg1), (db0, db1)):
        assert torch.isfinite(u).all() and torch.equal(u, v)
    # shapes without a block decomposition and misaligned addends are refused
    assert L.lib.gan_add_bn_sums(L.ptr(a), L.ptr(b), L.ptr(out1), ctypes.byref(saved), 128, c, L.ptr(partial), st) != 0
    assert L.lib.gan_add_bn_sums(L.ptr(a), L.ptr(pool[1:n + 1]), L.ptr(out1), ctypes.byref(saved), rows, c, L.ptr(partial), st) != 0


def test_riding_entry_points_refuse_what_they_cannot_serve(L):
    from gpu_util import geom_for
    g = geom_for(8, 32, 32, 16, 64, 5, 2, 'SAME')
    gp = ctypes.byref(g)
    x, w = torch.randn(8, 32, 32, 16, device='cuda'), torch.randn(5, 5, 16, 64, device='cuda')
    y, partial = torch.empty(8, 16, 16, 64, device='cuda'), torch.empty(1 << 16, device='cuda')
    wp, wn = _ws(L, g)
    st = L.stream()
    # groups that do not divide the rows, unknown op, no decomposition for the shape
    assert L.lib.gan_conv_fwd_bn_stats(L.OP_CONV2D, gp, L.ptr(x), L.ptr(w), None, L.ptr(y), 3, L.ptr(partial), wp, wn, st) != 0
    assert L.lib.gan_conv_fwd_bn_stats(7, gp, L.ptr(x), L.ptr(w), None, L.ptr(y), 1, L.ptr(partial), wp, wn, st) != 0
    assert L.lib.gan_conv_fwd_bn_stats(L.OP_CONV2D, gp, L.ptr(x), L.ptr(w), None, L.ptr(y), 4, L.ptr(partial), wp, wn, st) != 0
    assert int(L.lib.gan_bn_tile_blocks(1, 2048, 60, 0)) == 0 and int(L.lib.gan_bn_tile_blocks(1, 1024, 64, 0)) == 0
    assert int(L.lib.gan_bn_tile_blocks(1, 2048, 64, 0)) > 0
    # a refused call leaves nothing behind: the next plain forward runs as ever
    L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(w), None, L.ptr(y), wp, wn, st))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    L.knob_set("no_bn_ride", 1)
    assert int(L.lib.gan_bn_tile_blocks(1, 2048, 64, 0)) == 0


def _dcgan_step(L, knob_value):
    import bench
    from gan_amd import ops
    from gan_amd.dcgan import DCGAN
    L.knob_set("no_bn_ride", knob_value)
    ops._ws_cache.clear()
    rng = np.random.default_rng(77)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    m = DCGAN(None, bench.dcgan64_opts(seed=0), load_data=False)
    images, z = dev(rng.uniform(-1, 1, (64, 64, 64, 3))), dev(rng.uniform(-1, 1, (64, 100)))
    d = m.train_iteration(images, z)
    torch.cuda.synchronize()
    losses = [float(d[k]) for k in sorted(d)]
    values = {name: v.data.clone() for name, v in m.store.vars.items()}
    ops._ws_cache.clear()
    return losses, values


def test_dcgan_iteration_with_and_without_riding_reductions(L):
    """One DCGAN-64 iteration (batch 64: D update, two G updates) through ops.py with the reductions riding (default)
    and with every batch norm on its own launches (knob).  The biases of the conv / deconv layers in front of a batch
    norm take their gradient -- a column sum that is zero up to rounding, which Adam turns into a step of up to lr --
    from other partial sums: they differ by at most a step each way.  Everything else in the discriminator (updated
    once, from identical inputs) is bit-identical.  The generator is updated after the discriminator's biases moved:
    those cancel in the batch norms up to the last bit of the conv output, which is enough to move a leaky-relu unit
    across its kink now and then; one such unit changes the image gradient in a patch and with it every generator
    gradient by ~1e-3 of its scale, so the generator's variables agree to a small fraction of an Adam step on
    average and to a step per update at worst."""
    lr = 0.0002
    losses_a, va = _dcgan_step(L, 1)
    losses_b, vb = _dcgan_step(L, 0)
    assert set(va) == set(vb)
    fronts = ('d_h1_conv', 'd_h2_conv', 'd_h3_conv', 'g_h1', 'g_h2', 'g_h3')
    differing = 0
    for name in va:
        diff = (va[name] - vb[name]).abs()
        noisy = name.endswith('/biases') and any(k in name for k in fronts)
        if noisy:
            differing += int(not torch.equal(va[name], vb[name]))
            assert float(diff.max()) <= 2 * 2 * lr + 1e-7, name
        elif name.startswith('discriminator/') and '/moving_' not in name:
            assert torch.equal(va[name], vb[name]), name
        elif '/moving_' in name:       # (they follow the activations: the moved biases, the generator's output)
            assert float(diff.max()) <= 1e-3, name
        else:
            assert float(diff.max()) <= 2 * 2 * lr * 1.05, name
            assert float(diff.mean()) <= 0.1 * lr, name
    assert differing, "no layer took its bias gradient from the batch norm's column sums"
    np.testing.assert_allclose(np.asarray(losses_a, dtype=np.float64), np.asarray(losses_b, dtype=np.float64), rtol=2e-3)
