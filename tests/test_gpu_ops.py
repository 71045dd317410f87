"""GPU parity of every C-ABI entry point against the float64 numpy oracle, on identical seeded
inputs.  Calls go through ctypes into libgan_hip.so (torch only owns the device memory)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(2024)


@pytest.fixture(scope="module")
def L():
    from gan_amd import _lib
    return _lib


def _ws(L, g):
    n = int(L.lib.gan_conv_workspace_bytes(ctypes.byref(g)))
    return L.workspace(n)


# n, h, w, cb, cs, k, s, padding   (conv2d: x is [n,h,w,cb] -> y [n,hs,ws,cs])
SMALL_CONVS = [
    (2, 8, 8, 3, 4, 5, 2, 'SAME'),
    (2, 7, 7, 5, 6, 5, 2, 'SAME'),
    (3, 28, 28, 11, 11, 5, 2, 'SAME'),     # MNIST d_h0 (odd channels -> scalar load path)
    (3, 14, 14, 21, 74, 5, 2, 'SAME'),     # MNIST d_h1
    (4, 4, 4, 24, 16, 1, 1, 'SAME'),       # 1x1
    (1, 12, 12, 4, 8, 3, 2, 'SAME'),       # cyclegan 3x3 s2, pad (0,1)
    (1, 10, 10, 8, 8, 3, 1, 'VALID'),
    (1, 22, 22, 3, 8, 7, 1, 'VALID'),
    (1, 16, 16, 12, 1, 5, 1, 'SAME'),      # cyclegan d_h3_pred: Cout = 1
    (1, 22, 22, 8, 3, 7, 1, 'VALID'),      # odd Cout on the bf16 kernels: the last column is loaded on its own
    (1, 40, 36, 64, 3, 7, 1, 'SAME'),      # cyclegan image head 64 -> 3 (128x32 bf16 tile)
    (2, 9, 13, 8, 5, 5, 2, 'SAME'),
    (2, 9, 13, 8, 12, 5, 2, 'SAME'),       # non-square, odd
    (2, 14, 14, 128, 138, 5, 2, 'SAME'),   # MNIST g_h2 (cs even, not a multiple of 4: pair loads, dword epilogue)
    (2, 9, 13, 8, 10, 5, 2, 'SAME'),
    (1, 12, 12, 4, 6, 3, 2, 'SAME'),
    # thin big side (cb <= 4) with cs a multiple of 16: the vector-ALU input-gradient path
    (1, 22, 22, 3, 16, 7, 1, 'VALID'),
    (2, 9, 13, 2, 32, 5, 2, 'SAME'),
    (3, 28, 28, 1, 16, 5, 2, 'SAME'),
    (2, 12, 12, 4, 16, 3, 2, 'SAME'),
    (2, 70, 41, 3, 48, 5, 2, 'SAME'),      # several 8x32 patches per class, ragged edges
    # thin big side, cs NOT a multiple of 16: the same path, halo staged dword by dword, filter chunk range-checked
    (3, 28, 28, 1, 138, 5, 2, 'SAME'),     # MNIST g_h3 (y-conditioned: 128 + 10 channels -> the image)
    (2, 9, 13, 2, 10, 5, 2, 'SAME'),       # fewer channels than one chunk
    (1, 33, 20, 4, 37, 3, 1, 'SAME'),      # odd channel count, stride 1
    # thin big side with >= 4096 small-side pixels: the vector-ALU weight-gradient path
    (2, 96, 100, 3, 64, 5, 2, 'SAME'),
    (1, 70, 74, 3, 24, 7, 1, 'VALID'),     # 7x7 stem, cs not a multiple of 64
    (2, 90, 90, 1, 130, 5, 2, 'SAME'),     # cb = 1, three channel chunks, ragged last chunk
    (1, 68, 70, 4, 18, 3, 1, 'SAME'),      # cs % 4 != 0: scalar staging and reduction
]

DCGAN_CONVS = [
    (64, 64, 64, 3, 64, 5, 2, 'SAME'),     # d_h0
    (64, 32, 32, 64, 128, 5, 2, 'SAME'),   # d_h1
    (64, 16, 16, 128, 256, 5, 2, 'SAME'),  # d_h2
    (64, 8, 8, 256, 512, 5, 2, 'SAME'),    # d_h3
]


def _conv_case(L, case, cfg=None, splits=None):
    from gpu_util import assert_close, dev, forced, geom_for, host
    n, h, w, cb, cs, k, s, padding = case
    x = RNG.standard_normal((n, h, w, cb)).astype(np.float32)
    wt = (RNG.standard_normal((k, k, cb, cs)) * 0.05).astype(np.float32)
    b = RNG.standard_normal(cs).astype(np.float32)
    g = geom_for(n, h, w, cb, cs, k, s, padding)
    x64, w64 = x.astype(np.float64), wt.astype(np.float64)
    y_ref = T.conv2d_forward(x64, w64, s, s, padding) + b
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = T.conv2d_backward_input(dy.astype(np.float64), w64, x.shape, s, s, padding)
    dw_ref = T.conv2d_backward_filter(x64, dy.astype(np.float64), k, k, s, s, padding)
    db_ref = T.bias_add_backward(dy.astype(np.float64))
    xd, wd, bd, dyd = dev(x), dev(wt), dev(b), dev(dy)
    y = torch.empty(y_ref.shape, device='cuda')
    dx = torch.empty(x.shape, device='cuda')
    dw0 = RNG.standard_normal(wt.shape).astype(np.float32)
    db0 = RNG.standard_normal(cs).astype(np.float32)
    dw, db = dev(dw0), dev(db0)
    wp, wn = _ws(L, g)
    st = L.stream()
    with forced(cfg, splits):
        L.check(L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), wp, wn, st))
        L.check(L.lib.gan_conv2d_dgrad(ctypes.byref(g), L.ptr(dyd), L.ptr(wd), L.ptr(dx), wp, wn, st))
        L.check(L.lib.gan_conv2d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(db), 0, wp, wn, st))
        torch.cuda.synchronize()
        assert_close(host(y), y_ref, "conv2d fwd %s" % (case,))
        assert_close(host(dx), dx_ref, "conv2d dgrad %s" % (case,))
        assert_close(host(dw), dw_ref, "conv2d wgrad %s" % (case,))
        assert_close(host(db), db_ref, "conv2d bias grad %s" % (case,))
        # accumulate: a second call adds onto what is there
        L.check(L.lib.gan_conv2d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(db), 1, wp, wn, st))
        torch.cuda.synchronize()
        assert_close(host(dw), 2 * dw_ref, "conv2d wgrad accumulate %s" % (case,))
        assert_close(host(db), 2 * db_ref, "conv2d bias grad accumulate %s" % (case,))


@pytest.mark.parametrize("case", SMALL_CONVS)
def test_conv2d_small(L, case):
    _conv_case(L, case)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("splits", [1, 3])
def test_conv2d_every_tile_config_and_split(L, cfg, splits):
    _conv_case(L, (2, 12, 12, 8, 40, 5, 2, 'SAME'), cfg, splits)
    _conv_case(L, (2, 9, 9, 5, 7, 3, 1, 'SAME'), cfg, splits)      # odd channels: 4..6 fall back to 0..2


@pytest.mark.parametrize("case", [(3, 28, 28, 11, 11, 5, 2, 'SAME'),      # MNIST d_h0: runs of 55 floats, 4 slices per filter row
                                  (3, 14, 14, 21, 74, 5, 2, 'SAME'),      # MNIST d_h1: runs of 105 floats, 7 slices
                                  (3, 28, 28, 1, 138, 5, 2, 'SAME'),      # MNIST g_h3 (as a convolution): runs of 5 floats
                                  (2, 13, 17, 6, 10, 7, 1, 'SAME'),       # 7 x 7, stride 1, ragged image
                                  (2, 11, 9, 18, 5, 7, 2, 'VALID'),       # runs of 126 floats: the longest run served
                                  (1, 12, 12, 19, 8, 7, 1, 'SAME')])      # runs of 133 floats: stays on the fp32 kernel
@pytest.mark.parametrize("cfg", [4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("splits", [1, 2])
def test_filter_row_gather_for_any_channel_count(L, case, cfg, splits):
    """Forward and weight gradient over a big image whose channel count is no multiple of 4 on the bf16 kernels: a
    filter row's kw x cb values are gathered as one contiguous run, element by element (every bf16 tile
    configuration, with and without split-K slabs), against the float64 oracle."""
    _conv_case(L, case, cfg, splits)


@pytest.mark.parametrize("case,cfg,splits", [((3, 28, 28, 1, 138, 5, 2, 'SAME'), 2, 20),     # dw: 3450 elements
                                             ((3, 28, 28, 11, 11, 5, 2, 'SAME'), 3, 16),    # dw: 3025, dx: 25872
                                             ((2, 14, 14, 128, 138, 5, 2, 'SAME'), 5, 24)])
def test_many_splits_over_outputs_that_are_no_multiple_of_four(L, case, cfg, splits):
    """16 or more split-K slabs are summed by split lanes working side by side; when the output's element count is not
    a multiple of 4 (a slab then starts at any dword) that reduction runs dword by dword."""
    _conv_case(L, case, cfg, splits)


# channel counts that are multiples of 4 (the split-bf16 kernels' condition) with ragged tiles:
# rows, columns and depth that are not multiples of the tile sides / of the 16-deep slice
SPLIT_CONVS = [
    (3, 14, 14, 20, 76, 5, 2, 'SAME'),      # N = 76: one full and one ragged 64-wide column tile
    (1, 17, 11, 12, 136, 3, 1, 'SAME'),     # K = 108 (not a multiple of 16), N = 136, M = 187
    (2, 9, 13, 8, 12, 5, 2, 'SAME'),        # N = 12: every tile mostly padding
    (5, 4, 4, 24, 16, 1, 1, 'SAME'),        # 1x1: K = 24
    (1, 10, 10, 8, 8, 3, 1, 'VALID'),
    (2, 16, 16, 12, 4, 5, 1, 'SAME'),       # transposed contraction with 12 output channels
    # channel counts / widths that take the cheap gather arithmetic (16-channel aligned gathered image, small
    # width a multiple of 4), still with ragged tiles
    (3, 14, 14, 32, 76, 5, 2, 'SAME'),      # FPROP fast (cb = 32), TPROP slow (cs = 76), WGRAD slow (ws = 7)
    (1, 17, 12, 16, 144, 3, 1, 'SAME'),     # all three fast: cb = 16, cs = 144, ws = 12; M = 204
    (2, 9, 16, 48, 32, 5, 2, 'SAME'),       # all three fast, odd height, pad (1, 2)
    (1, 11, 7, 20, 48, 3, 1, 'SAME'),       # only TPROP fast (cs = 48)
]


@pytest.mark.parametrize("cfg", [4, 5, 6])
@pytest.mark.parametrize("case", SPLIT_CONVS)
def test_conv2d_split_bf16_kernels_on_ragged_shapes(L, cfg, case):
    _conv_case(L, case, cfg, 1)
    _conv_case(L, case, cfg, 2)


@pytest.mark.parametrize("case", DCGAN_CONVS)
def test_conv2d_dcgan64_shapes(L, case):
    _conv_case(L, case)


# 3-channel images (the filter-row gather of the split kernels) with 32 / 64 / 128 output channels and 3x3, 4x4, 5x5,
# 7x7 filters; pixel counts that are no multiple of a tile, borders on every side, both strides
RGB_CONVS = [
    (2, 13, 11, 3, 32, 5, 2, 'SAME'),
    (1, 33, 31, 3, 128, 5, 2, 'SAME'),
    (3, 16, 16, 3, 64, 5, 1, 'SAME'),
    (1, 9, 9, 3, 128, 3, 1, 'SAME'),
    (2, 12, 12, 3, 64, 4, 2, 'SAME'),
    (1, 20, 18, 3, 64, 7, 1, 'VALID'),
    (3, 7, 9, 3, 32, 7, 1, 'SAME'),
    (5, 66, 64, 3, 64, 5, 2, 'SAME'),
]


@pytest.mark.parametrize("case", RGB_CONVS)
def test_conv2d_rgb_layers(L, case):
    _conv_case(L, case)
    _conv_case(L, case, 5, 1)
    _conv_case(L, case, 8, 2)


# thin OUTPUT of a stride-1 forward pass (<= 4 output channels: CycleGAN's 7x7 head 64 -> 3, the discriminators'
# prediction layer 512 -> 1); these stay on the 128x32 MFMA tile
THIN_FWD_CONVS = [
    (1, 20, 18, 32, 3, 7, 1, 'VALID'),
    (2, 16, 16, 16, 1, 4, 1, 'SAME'),
    (1, 40, 70, 64, 3, 7, 1, 'SAME'),
    (2, 9, 11, 48, 4, 3, 1, 'SAME'),
    (3, 8, 8, 512, 1, 4, 1, 'SAME'),
    (1, 33, 5, 16, 2, 5, 1, 'SAME'),
]


@pytest.mark.parametrize("case", THIN_FWD_CONVS)
def test_conv2d_thin_output_shapes(L, case):
    _conv_case(L, case)


# n, hs, ws, cs(in), cb(out), k, s, out_h, out_w   (deconv2d: x [n,hs,ws,cs] -> y [n,oh,ow,cb])
SMALL_DECONVS = [
    (2, 4, 4, 8, 4, 5, 2, 8, 8),
    (2, 4, 4, 6, 3, 5, 2, 7, 7),
    (3, 7, 7, 138, 128, 5, 2, 14, 14),   # MNIST g_h2
    (3, 14, 14, 138, 1, 5, 2, 28, 28),   # MNIST g_h3: Cout = 1
    (1, 8, 8, 8, 4, 3, 2, 16, 16),       # cyclegan 3x3 s2
    (2, 5, 6, 4, 12, 5, 2, 9, 11),       # odd, non-square
]

DCGAN_DECONVS = [
    (64, 4, 4, 512, 256, 5, 2, 8, 8),      # g_h1
    (64, 8, 8, 256, 128, 5, 2, 16, 16),    # g_h2
    (64, 16, 16, 128, 64, 5, 2, 32, 32),   # g_h3
    (64, 32, 32, 64, 3, 5, 2, 64, 64),     # g_h4
]


def _deconv_case(L, case, cfg=None, splits=None):
    from gpu_util import assert_close, dev, forced, geom_for, host
    n, hs, ws_, cs, cb, k, s, oh, ow = case
    x = RNG.standard_normal((n, hs, ws_, cs)).astype(np.float32)
    wt = (RNG.standard_normal((k, k, cb, cs)) * 0.05).astype(np.float32)
    b = RNG.standard_normal(cb).astype(np.float32)
    g = geom_for(n, oh, ow, cb, cs, k, s, 'SAME')
    assert (g.hs, g.ws) == (hs, ws_)
    x64, w64 = x.astype(np.float64), wt.astype(np.float64)
    y_ref = T.conv2d_transpose_forward(x64, w64, (n, oh, ow, cb), s, s) + b
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = T.conv2d_transpose_backward_input(dy.astype(np.float64), w64, s, s)
    dw_ref = T.conv2d_transpose_backward_filter(x64, dy.astype(np.float64), k, k, s, s)
    db_ref = T.bias_add_backward(dy.astype(np.float64))
    xd, wd, bd, dyd = dev(x), dev(wt), dev(b), dev(dy)
    y = torch.empty(y_ref.shape, device='cuda')
    dx = torch.empty(x.shape, device='cuda')
    dw = torch.empty(wt.shape, device='cuda')
    db = torch.empty(cb, device='cuda')
    wp, wn = _ws(L, g)
    st = L.stream()
    with forced(cfg, splits):
        L.check(L.lib.gan_deconv2d_fwd(ctypes.byref(g), L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y), wp, wn, st))
        L.check(L.lib.gan_deconv2d_dgrad(ctypes.byref(g), L.ptr(dyd), L.ptr(wd), L.ptr(dx), wp, wn, st))
        L.check(L.lib.gan_deconv2d_wgrad(ctypes.byref(g), L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(db), 0, wp, wn, st))
        torch.cuda.synchronize()
    assert_close(host(y), y_ref, "deconv2d fwd %s" % (case,))
    assert_close(host(dx), dx_ref, "deconv2d dgrad %s" % (case,))
    assert_close(host(dw), dw_ref, "deconv2d wgrad %s" % (case,))
    assert_close(host(db), db_ref, "deconv2d bias grad %s" % (case,))


@pytest.mark.parametrize("case", SMALL_DECONVS)
def test_deconv2d_small(L, case):
    _deconv_case(L, case)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("splits", [1, 2])
def test_deconv2d_every_tile_config_and_split(L, cfg, splits):
    _deconv_case(L, (2, 6, 6, 8, 36, 5, 2, 12, 12), cfg, splits)
    _deconv_case(L, (2, 5, 5, 3, 5, 3, 2, 9, 9), cfg, splits)


@pytest.mark.parametrize("case", DCGAN_DECONVS)
def test_deconv2d_dcgan64_shapes(L, case):
    _deconv_case(L, case)


def test_deconv2d_is_adjoint_of_conv2d_on_device(L):
    """Size-independent property at the full DCGAN g_h2 size: <conv2d(u, w), x> == <u, deconv2d(x, w)>."""
    from gpu_util import dev, geom_for
    n, hs, cs, cb, oh = 64, 8, 256, 128, 16
    g = geom_for(n, oh, oh, cb, cs, 5, 2, 'SAME')
    x = dev(RNG.standard_normal((n, hs, hs, cs)))
    u = dev(RNG.standard_normal((n, oh, oh, cb)))
    wd = dev(RNG.standard_normal((5, 5, cb, cs)) * 0.05)
    y = torch.empty((n, oh, oh, cb), device='cuda')
    cu = torch.empty((n, hs, hs, cs), device='cuda')
    wp, wn = _ws(L, g)
    L.check(L.lib.gan_deconv2d_fwd(ctypes.byref(g), L.ptr(x), L.ptr(wd), None, L.ptr(y), wp, wn, L.stream()))
    L.check(L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(u), L.ptr(wd), None, L.ptr(cu), wp, wn, L.stream()))
    lhs = float((cu.double() * x.double()).sum())
    rhs = float((u.double() * y.double()).sum())
    # both sides are sums of ~2M cancelling products: the bar is relative to the mass that cancels, not to the rest
    mass = float((cu.double() * x.double()).abs().sum())
    assert abs(lhs - rhs) <= 1e-6 * mass


@pytest.mark.parametrize("rows,i,o", [(64, 100, 8192), (64, 8192, 1), (64, 110, 1024), (64, 1034, 6272),
                                      (64, 3636, 1024), (5, 7, 3), (64, 1024, 256),
                                      # a handful of outputs: the matrix-vector kernels
                                      (64, 1034, 1), (128, 8192, 1), (9, 300, 3), (7, 513, 4), (8, 256, 2), (2, 260, 1)])
def test_linear(L, rows, i, o):
    from gpu_util import assert_close, dev, host
    x = RNG.standard_normal((rows, i)).astype(np.float32)
    m = (RNG.standard_normal((i, o)) * 0.05).astype(np.float32)
    b = RNG.standard_normal(o).astype(np.float32)
    dy = RNG.standard_normal((rows, o)).astype(np.float32)
    y_ref = T.linear_forward(x.astype(np.float64), m.astype(np.float64), b.astype(np.float64))
    dx_ref, dm_ref, db_ref = T.linear_backward(x.astype(np.float64), m.astype(np.float64), dy.astype(np.float64))
    xd, md, bd, dyd = dev(x), dev(m), dev(b), dev(dy)
    y, dx = torch.empty((rows, o), device='cuda'), torch.empty((rows, i), device='cuda')
    dm, db = torch.empty((i, o), device='cuda'), torch.empty(o, device='cuda')
    wp, wn = L.workspace(int(L.lib.gan_linear_workspace_bytes(rows, i, o)))
    st = L.stream()
    L.check(L.lib.gan_linear_fwd(rows, i, o, L.ptr(xd), L.ptr(md), L.ptr(bd), L.ptr(y), wp, wn, st))
    L.check(L.lib.gan_linear_dgrad(rows, i, o, L.ptr(dyd), L.ptr(md), L.ptr(dx), wp, wn, st))
    L.check(L.lib.gan_linear_wgrad(rows, i, o, L.ptr(xd), L.ptr(dyd), L.ptr(dm), L.ptr(db), 0, wp, wn, st))
    torch.cuda.synchronize()
    assert_close(host(y), y_ref, "linear fwd")
    assert_close(host(dx), dx_ref, "linear dgrad")
    assert_close(host(dm), dm_ref, "linear wgrad")
    assert_close(host(db), db_ref, "linear bias grad")
    # accumulate: a second call adds onto what is there
    L.check(L.lib.gan_linear_wgrad(rows, i, o, L.ptr(xd), L.ptr(dyd), L.ptr(dm), L.ptr(db), 1, wp, wn, st))
    torch.cuda.synchronize()
    assert_close(host(dm), 2 * dm_ref, "linear wgrad accumulate")
    assert_close(host(db), 2 * db_ref, "linear bias grad accumulate")


def test_thin_linear_forward_takes_rows_and_vectors_at_any_dword_address(L):
    """The single-output forward moves 16 bytes per lane only when the rows and the vector allow it."""
    from gpu_util import assert_close, dev, host
    rows, i = 6, 512
    x = RNG.standard_normal((rows, i)).astype(np.float32)
    m = (RNG.standard_normal((i, 1)) * 0.05).astype(np.float32)
    y_ref = T.linear_forward(x.astype(np.float64), m.astype(np.float64), np.zeros(1))
    xbuf, mbuf = torch.empty(rows * i + 1, device='cuda'), torch.empty(i + 1, device='cuda')
    xbuf[1:].copy_(dev(x).reshape(-1))
    mbuf[1:].copy_(dev(m).reshape(-1))
    xd, md = dev(x), dev(m)
    y = torch.empty((rows, 1), device='cuda')
    for xp, mp in [(xbuf.data_ptr() + 4, L.ptr(md)), (L.ptr(xd), mbuf.data_ptr() + 4), (xbuf.data_ptr() + 4, mbuf.data_ptr() + 4)]:
        y.fill_(7.0)
        L.check(L.lib.gan_linear_fwd(rows, i, 1, xp, mp, None, L.ptr(y), None, 0, L.stream()))
        torch.cuda.synchronize()
        assert_close(host(y), y_ref, "thin linear fwd, unaligned")


ACTS = [(0, 0.0), (1, 0.0), (2, 0.2), (3, 0.0), (4, 0.0)]


def _act_ref(z, act, p):
    return [z, T.relu(z), T.lrelu(z, p), np.tanh(z), T.sigmoid(z)][act]


def _act_grad_ref(z, a, act, p):
    return [np.ones_like(z), (z > 0).astype(z.dtype), np.where(z >= 0, 1.0, p), 1 - a * a, a * (1 - a)][act]


@pytest.mark.parametrize("shape,groups", [((64, 16, 16, 128), 1), ((64, 4, 4, 512), 1), ((64, 1024), 1),
                                          ((64, 6272), 1), ((5, 7, 7, 74), 1), ((1, 32, 32, 3), 1),
                                          ((3, 8, 8, 12), 3), ((64, 32, 32, 64), 1)])
@pytest.mark.parametrize("act,param", ACTS)
def test_batch_norm_train(L, shape, groups, act, param):
    from gpu_util import assert_close, dev, host
    c = shape[-1]
    rows = int(np.prod(shape)) // c // groups
    x = (RNG.standard_normal(shape) * 1.7 + 0.8).astype(np.float32)
    gamma = (RNG.standard_normal(c) * 0.3 + 1).astype(np.float32)
    beta = (RNG.standard_normal(c) * 0.3).astype(np.float32)
    res = RNG.standard_normal(shape).astype(np.float32)
    dy = RNG.standard_normal(shape).astype(np.float32)
    mm0, mv0 = RNG.standard_normal(c).astype(np.float32), (RNG.random(c) + 0.5).astype(np.float32)
    x64 = x.astype(np.float64).reshape(groups, rows, c)
    if act in (1, 2):
        # relu / lrelu have a kink at z == 0: where the fp32 and fp64 pre-activations could land on
        # different sides of it, take the element out of the gradient comparison (dy = 0 there)
        for gi in range(groups):
            z, _, _, _ = T.batch_norm_train_forward(x64[gi], gamma.astype(np.float64), beta.astype(np.float64), 1e-5)
            dy.reshape(groups, rows, c)[gi][np.abs(z) < 1e-3] = 0.0
    z_ref = np.empty_like(x64)
    dx_ref = np.empty_like(x64)
    dg_ref, db_ref = np.zeros(c), np.zeros(c)
    means, variances = [], []
    for gi in range(groups):
        z, mean, var, rstd = T.batch_norm_train_forward(x64[gi], gamma.astype(np.float64), beta.astype(np.float64), 1e-5)
        z_ref[gi] = z
        a = _act_ref(z, act, param)
        dz = dy.astype(np.float64).reshape(groups, rows, c)[gi] * _act_grad_ref(z, a, act, param)
        dx_ref[gi], dg, db = T.batch_norm_train_backward(x64[gi], gamma.astype(np.float64), mean, rstd, dz)
        dg_ref += dg
        db_ref += db
        means.append(mean)
        variances.append(var)
    y_ref = _act_ref(z_ref, act, param).reshape(shape) + res
    xd, gd, bd, rd, dyd = dev(x), dev(gamma), dev(beta), dev(res), dev(dy)
    mm, mv = (dev(mm0), dev(mv0)) if groups == 1 else (None, None)
    y, dx = torch.empty(shape, device='cuda'), torch.empty(shape, device='cuda')
    sm, sr = torch.empty(groups * c, device='cuda'), torch.empty(groups * c, device='cuda')
    dgam, dbet = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    wp, wn = L.workspace(int(L.lib.gan_bn_workspace_bytes(groups, rows, c)))
    st = L.stream()
    L.check(L.lib.gan_bn_fwd_train(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(mm), L.ptr(mv), L.ptr(rd), L.ptr(y), L.ptr(sm),
                                   L.ptr(sr), groups, rows, c, 1e-5, 0.9, 0, act, param, wp, wn, st))
    L.check(L.lib.gan_bn_bwd(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(sm), L.ptr(sr), L.ptr(dyd), L.ptr(dx), L.ptr(dgam),
                             L.ptr(dbet), 0, groups, rows, c, act, param, wp, wn, st))
    torch.cuda.synchronize()
    assert_close(host(y), y_ref, "bn fwd")
    assert_close(host(sm), np.concatenate(means), "bn saved mean")
    assert_close(host(dx).reshape(groups, rows, c), dx_ref, "bn dx", tol=2e-4 if rows < 8 else 1e-4)
    assert_close(host(dgam), dg_ref, "bn dgamma")
    assert_close(host(dbet), db_ref, "bn dbeta")
    if groups == 1:
        mm_ref, mv_ref = T.batch_norm_moving_update(mm0.astype(np.float64), mv0.astype(np.float64), means[0],
                                                    variances[0], 0.9, rows)
        assert_close(host(mm), mm_ref, "moving mean")
        assert_close(host(mv), mv_ref, "moving variance")
        # inference path on the updated moving statistics
        ye = torch.empty(shape, device='cuda')
        L.check(L.lib.gan_bn_fwd_eval(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(mm), L.ptr(mv), L.ptr(ye), rows, c, 1e-5,
                                      act, param, st))
        torch.cuda.synchronize()
        ze = T.batch_norm_eval_forward(x.astype(np.float64), gamma, beta, mm_ref, mv_ref, 1e-5)
        assert_close(host(ye), _act_ref(ze, act, param), "bn eval")


@pytest.mark.parametrize("groups,rows,c", [(1, 1024, 512), (2, 1024, 512), (1, 64, 1024), (3, 300, 8), (1, 513, 132)])
def test_one_launch_batch_norm_agrees_with_the_three_launch_path(L, groups, rows, c, monkeypatch):
    """Small batch norms (<= 1024 rows per group) run as ONE launch, a workgroup per channel quad with the
    rows in registers; the general three-launch path (test knob) gives the same results to rounding,
    moving averages of stacked passes included."""
    torch.manual_seed(groups * 1000 + rows)
    x = torch.randn(groups * rows, c, device='cuda') * 1.5 + 0.3
    res, dy = torch.randn_like(x), torch.randn_like(x)
    ga, be = torch.randn(c, device='cuda') * 0.3 + 1, torch.randn(c, device='cuda') * 0.3
    outs = []
    for general in (False, True):
        if general:
            L.knob_set("three_launch_bn", 1)
        mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
        y, dx = torch.empty_like(x), torch.empty_like(x)
        sm, sr = torch.empty(groups * c, device='cuda'), torch.empty(groups * c, device='cuda')
        dg, db = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')
        wp, wn = L.workspace(int(L.lib.gan_bn_workspace_bytes(groups, rows, c)))
        st = L.stream()
        L.check(L.lib.gan_bn_fwd_train(L.ptr(x), L.ptr(ga), L.ptr(be), L.ptr(mm), L.ptr(mv), L.ptr(res), L.ptr(y), L.ptr(sm),
                                       L.ptr(sr), groups, rows, c, 1e-5, 0.9, 1, 3, 0.0, wp, wn, st))
        L.check(L.lib.gan_bn_bwd(L.ptr(x), L.ptr(ga), L.ptr(be), L.ptr(sm), L.ptr(sr), L.ptr(dy), L.ptr(dx), L.ptr(dg),
                                 L.ptr(db), 1, groups, rows, c, 3, 0.0, wp, wn, st))
        torch.cuda.synchronize()
        outs.append((y, sm, sr, mm, mv, dx, dg, db))
    for name, a, b in zip(("y", "mean", "rstd", "moving mean", "moving variance", "dx", "dgamma", "dbeta"), *outs):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()), msg=lambda m, n=name: n + ": " + m)


@pytest.mark.parametrize("groups,rows,c", [(1, 16384, 128), (2, 16384, 128), (1, 65536, 64), (1, 4096, 256), (3, 16384, 128),
                                           (1, 2048, 512), (1, 5001, 64), (2, 1025, 64), (5, 3000, 256), (2, 1024, 512), (3, 300, 64)])
def test_two_launch_batch_norm_agrees_with_the_three_launch_path(L, groups, rows, c):
    """Batch norms above 1024 rows per group (channel count a multiple of 64) run as TWO launches: statistics per
    (row block, 64-channel tile), then an apply kernel whose workgroups merge the block statistics of their tile
    themselves.  The three-launch path (knob) gives the same results to rounding: outputs, saved statistics, moving
    averages of stacked passes (moved once per group, in order), input and parameter gradients with accumulation."""
    torch.manual_seed(groups * 1000 + rows)
    ga, be = torch.randn(c, device='cuda') * 0.3 + 1, torch.randn(c, device='cuda') * 0.3
    wp, wn = L.workspace(int(L.lib.gan_bn_workspace_bytes(groups, rows, c)))
    names = ("y", "mean", "rstd", "moving mean", "moving variance", "dx", "dgamma", "dbeta")
    for rnd in range(2):
        x = torch.randn(groups * rows, c, device='cuda') * (1.5 + rnd) + 0.3 * (rnd - 1)
        res, dy = torch.randn_like(x), torch.randn_like(x)
        outs = []
        for general in (False, True):
            L.knob_set("two_launch_bn", 0 if general else 1)
            mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
            y, dx = torch.empty_like(x), torch.empty_like(x)
            sm, sr = torch.empty(groups * c, device='cuda'), torch.empty(groups * c, device='cuda')
            dg, db = torch.ones(c, device='cuda'), torch.ones(c, device='cuda')
            st = L.stream()
            L.check(L.lib.gan_bn_fwd_train(L.ptr(x), L.ptr(ga), L.ptr(be), L.ptr(mm), L.ptr(mv), L.ptr(res), L.ptr(y),
                                           L.ptr(sm), L.ptr(sr), groups, rows, c, 1e-5, 0.9, 1, 2, 0.2, wp, wn, st))
            L.check(L.lib.gan_bn_bwd(L.ptr(x), L.ptr(ga), L.ptr(be), L.ptr(sm), L.ptr(sr), L.ptr(dy), L.ptr(dx), L.ptr(dg),
                                     L.ptr(db), 1, groups, rows, c, 2, 0.2, wp, wn, st))
            torch.cuda.synchronize()
            outs.append((y, sm, sr, mm, mv, dx, dg, db))
        L.knob_set("two_launch_bn", 1)
        for name, a, b in zip(names, outs[0], outs[1]):
            torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max()),
                                       msg=lambda m, n=name, r=rnd: "round %d, %s: %s" % (r, n, m))


def test_moving_average_replay_of_many_layers_in_one_call(L):
    """gan_bn_moving_update_multi moves the averages of many layers with one launch per 64 of them; bit-identical to
    gan_bn_moving_update layer by layer."""
    from gan_amd._lib import BnMovingJob
    torch.manual_seed(11)
    cs = [int(c) for c in np.random.default_rng(3).integers(1, 700, size=70)]
    layers = [(torch.randn(c, device='cuda'), torch.rand(c, device='cuda') + 0.5, torch.randn(c, device='cuda'),
               torch.rand(c, device='cuda') + 0.1, 64 + i) for i, c in enumerate(cs)]
    single = [(mm.clone(), mv.clone()) for mm, mv, _, _, _ in layers]
    for (mm, mv), (_, _, mean, rstd, rows), c in zip(single, layers, cs):
        L.check(L.lib.gan_bn_moving_update(L.ptr(mm), L.ptr(mv), L.ptr(mean), L.ptr(rstd), rows, c, 1e-5, 0.9, 1, L.stream()))
    arr = (BnMovingJob * len(layers))()
    for j, (mm, mv, mean, rstd, rows), c in zip(arr, layers, cs):
        j.moving_mean, j.moving_var, j.save_mean, j.save_rstd, j.rows, j.c = L.ptr(mm), L.ptr(mv), L.ptr(mean), L.ptr(rstd), rows, c
    L.check(L.lib.gan_bn_moving_update_multi(arr, len(layers), 1e-5, 0.9, 1, L.stream()))
    torch.cuda.synchronize()
    for (mm1, mv1), (mm, mv, _, _, _) in zip(single, layers):
        assert torch.equal(mm1, mm) and torch.equal(mv1, mv)
    assert L.lib.gan_bn_moving_update_multi(arr, 0, 1e-5, 0.9, 1, L.stream()) != 0


def test_batch_norm_unbiased_moving_variance_flag(L):
    from gpu_util import assert_close, dev, host
    shape, c = (8, 4, 4, 16), 16
    rows = 128
    x = RNG.standard_normal(shape).astype(np.float32)
    xd, gd, bd = dev(x), dev(np.ones(c)), dev(np.zeros(c))
    mm, mv = dev(np.zeros(c)), dev(np.ones(c))
    y, sm, sr = torch.empty(shape, device='cuda'), torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    wp, wn = L.workspace(int(L.lib.gan_bn_workspace_bytes(1, rows, c)))
    L.check(L.lib.gan_bn_fwd_train(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(mm), L.ptr(mv), None, L.ptr(y), L.ptr(sm),
                                   L.ptr(sr), 1, rows, c, 1e-5, 0.9, 1, 0, 0.0, wp, wn, L.stream()))
    torch.cuda.synchronize()
    _, mean, var, _ = T.batch_norm_train_forward(x.astype(np.float64), np.ones(c), np.zeros(c), 1e-5)
    _, mv_ref = T.batch_norm_moving_update(np.zeros(c), np.ones(c), mean, var, 0.9, rows, unbiased_moving_var=True)
    assert_close(host(mv), mv_ref, "unbiased moving variance")


@pytest.mark.parametrize("act,param", ACTS[1:])
@pytest.mark.parametrize("scale,shift", [(1.0, 0.0), (0.5, 0.5)])
def test_activations(L, act, param, scale, shift):
    from gpu_util import assert_close, dev, host
    x = (RNG.standard_normal(10007) * 2).astype(np.float32)
    x[:3] = [0.0, -0.0, 1e-30]
    if (scale, shift) != (1.0, 0.0) and act in (1, 2):
        xd = dev(x)
        assert L.lib.gan_act_fwd(L.ptr(xd), L.ptr(xd), x.size, act, param, scale, shift, L.stream()) != 0
        assert b"not supported" in L.lib.gan_last_error()
        return
    dy = RNG.standard_normal(10007).astype(np.float32)
    a = _act_ref(x.astype(np.float64), act, param)
    xd, dyd = dev(x), dev(dy)
    y, dx = torch.empty_like(xd), torch.empty_like(xd)
    L.check(L.lib.gan_act_fwd(L.ptr(xd), L.ptr(y), x.size, act, param, scale, shift, L.stream()))
    L.check(L.lib.gan_act_bwd(L.ptr(y), L.ptr(dyd), L.ptr(dx), x.size, act, param, scale, shift, L.stream()))
    torch.cuda.synchronize()
    assert_close(host(y), a * scale + shift, "act fwd")
    assert_close(host(dx), dy * scale * _act_grad_ref(x.astype(np.float64), a, act, param), "act bwd", tol=2e-4)


def test_add_fill_concat_reflect(L):
    from gpu_util import assert_close, dev, host
    a, b = RNG.standard_normal(1001).astype(np.float32), RNG.standard_normal(1001).astype(np.float32)
    ad, bd = dev(a), dev(b)
    y = torch.empty_like(ad)
    L.check(L.lib.gan_add_fwd(L.ptr(ad), L.ptr(bd), L.ptr(y), 1001, L.stream()))
    L.check(L.lib.gan_fill(L.ptr(ad), 1001, 3.5, L.stream()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(y), a + b)
    assert np.all(host(ad) == 3.5)
    # conv_cond_concat and plain concat
    for shape_x, cy, rpb in [((3, 7, 7, 5), 10, 49), ((4, 4, 4, 512), 256, 16), ((6, 37), 10, 1)]:
        x = RNG.standard_normal(shape_x).astype(np.float32)
        n = shape_x[0]
        yv = RNG.standard_normal((n, cy)).astype(np.float32)
        ref = (T.conv_cond_concat(x, yv.reshape(n, 1, 1, cy)) if len(shape_x) == 4
               else np.concatenate([x, yv], axis=1))
        out = torch.empty(ref.shape, device='cuda')
        rows = int(np.prod(shape_x[:-1]))
        xd, yd = dev(x), dev(yv)         # named: a temporary would be freed (and its block reused) at once
        L.check(L.lib.gan_concat_fwd(L.ptr(xd), L.ptr(yd), L.ptr(out), rows, shape_x[-1], cy, rpb, L.stream()))
        dout = RNG.standard_normal(ref.shape).astype(np.float32)
        doutd = dev(dout)
        da, db = torch.empty(shape_x, device='cuda'), torch.empty((n, cy), device='cuda')
        L.check(L.lib.gan_concat_bwd(L.ptr(doutd), L.ptr(da), L.ptr(db), rows, shape_x[-1], cy, rpb, L.stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host(out), ref)
        np.testing.assert_array_equal(host(da), dout[..., :shape_x[-1]])
        db_ref = dout[..., shape_x[-1]:].astype(np.float64).reshape(n, -1, cy).sum(axis=1)
        assert_close(host(db), db_ref, "concat db")
    for p, shape in [(1, (2, 6, 5, 8)), (3, (1, 9, 11, 3)), (3, (1, 4, 4, 2))]:
        x = RNG.standard_normal(shape).astype(np.float32)
        ref = T.reflect_pad(x, p)
        out = torch.empty(ref.shape, device='cuda')
        n, h, w, c = shape
        xd = dev(x)
        L.check(L.lib.gan_reflect_pad_fwd(L.ptr(xd), L.ptr(out), n, h, w, c, p, L.stream()))
        dyv = RNG.standard_normal(ref.shape).astype(np.float32)
        dyd = dev(dyv)
        dx = torch.empty(shape, device='cuda')
        L.check(L.lib.gan_reflect_pad_bwd(L.ptr(dyd), L.ptr(dx), n, h, w, c, p, L.stream()))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host(out), ref)
        assert_close(host(dx), T.reflect_pad_backward(dyv.astype(np.float64), p), "reflect pad bwd")


def test_losses(L):
    from gpu_util import assert_close, dev, host
    for n in (64, 1024, 3 * 32 * 32, 50000, 50001, 3 * 256 * 256):   # >= 4096 and a multiple of 4: 16-byte path
        x = (RNG.standard_normal(n) * 3).astype(np.float32)
        xd = dev(x)
        loss = torch.zeros(1, device='cuda')
        g = torch.empty(n, device='cuda')
        for label in (0.0, 1.0):
            L.check(L.lib.gan_sigmoid_bce_mean(L.ptr(xd), n, label, 1.0, 0, L.ptr(loss), L.ptr(g), L.stream()))
            torch.cuda.synchronize()
            lref, gref = T.sigmoid_cross_entropy_mean(x.astype(np.float64), label)
            assert_close(host(loss), [lref], "bce loss")
            assert_close(host(g), gref, "bce grad")
        L.check(L.lib.gan_mse_const_mean(L.ptr(xd), n, 1.0, 0.5, 1, L.ptr(loss), L.ptr(g), L.stream()))
        torch.cuda.synchronize()
        l2, g2 = T.mse_mean(x.astype(np.float64), np.ones(n))
        assert_close(host(loss), [lref + 0.5 * l2], "mse accumulate")
        assert_close(host(g), 0.5 * g2, "mse grad")
        b = RNG.standard_normal(n).astype(np.float32)
        bd = dev(b)
        L.check(L.lib.gan_l1_mean(L.ptr(xd), L.ptr(bd), n, 10.0, 0, L.ptr(loss), L.ptr(g), L.stream()))
        torch.cuda.synchronize()
        l1, g1 = T.l1_mean(x.astype(np.float64), b.astype(np.float64))
        assert_close(host(loss), [10 * l1], "l1 loss")
        assert_close(host(g), -10 * g1, "l1 grad w.r.t. b")
        # the multi-block form: block sums through a workspace, any n and alignment
        wp, wn = L.workspace(int(L.lib.gan_l1_mean_workspace_bytes(n)))
        for acc in (0, 1):
            g.fill_(7.0)
            L.check(L.lib.gan_l1_mean_ws(L.ptr(xd), L.ptr(bd), n, 10.0, acc, L.ptr(loss), L.ptr(g), wp, wn, L.stream()))
            torch.cuda.synchronize()
            assert_close(host(loss), [10 * l1 * (1 + acc)], "l1 loss, multi-block")
            assert_close(host(g), -10 * g1, "l1 grad w.r.t. b, multi-block")
        L.check(L.lib.gan_l1_mean_ws(L.ptr(xd), L.ptr(bd), n, 10.0, 0, L.ptr(loss), None, wp, wn, L.stream()))
        torch.cuda.synchronize()
        assert_close(host(loss), [10 * l1], "l1 loss, multi-block, no gradient")
        assert L.lib.gan_l1_mean_ws(L.ptr(xd), L.ptr(bd), n, 10.0, 0, L.ptr(loss), None, wp, 8, L.stream()) != 0   # workspace too small


def test_adam_tf_steps(L):
    from gpu_util import assert_close, dev, host
    n = 100003
    theta = RNG.standard_normal(n).astype(np.float32)
    th, m, v = dev(theta), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    lr_s = torch.zeros(1, device='cuda')
    rt, rm, rv = theta.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 6):
        g = (RNG.standard_normal(n) * 0.01).astype(np.float32)
        gd = dev(g * 4)
        L.check(L.lib.gan_adam_tf_step(L.ptr(th), L.ptr(gd), L.ptr(m), L.ptr(v), n, 2e-4, 0.5, 0.999, 1e-8, 0.25,
                                       L.ptr(step), L.ptr(lr_s), L.stream()))
        rt, rm, rv = T.adam_tf_step(rt, g.astype(np.float64), rm, rv, t, 2e-4, 0.5)
    torch.cuda.synchronize()
    assert int(step.item()) == 5
    assert_close(host(th), rt, "adam theta", tol=1e-6)
    assert_close(host(m), rm, "adam m")
    assert_close(host(v), rv, "adam v")


def test_argument_errors_are_reported(L):
    from gpu_util import dev, geom_for
    g = geom_for(1, 8, 8, 4, 4, 5, 2, 'SAME')
    x = dev(np.zeros((1, 8, 8, 4)))
    rc = L.lib.gan_conv2d_fwd(ctypes.byref(g), L.ptr(x), None, None, L.ptr(x), None, 0, L.stream())
    assert rc != 0 and b"NULL" in L.lib.gan_last_error()
    bad = geom_for(1, 8, 8, 4, 4, 5, 2, 'SAME')
    bad.sh = 3
    assert L.lib.gan_conv2d_fwd(ctypes.byref(bad), L.ptr(x), L.ptr(x), None, L.ptr(x), None, 0, L.stream()) != 0
    assert b"stride" in L.lib.gan_last_error()
    with pytest.raises(L.GanHipError):
        L.ptr(torch.zeros(4))          # host tensor: no CPU path


def test_autotuned_plans_keep_parity(L):
    """Tuning pins some plan for each mode of the geometry; results stay within the parity bar and
    the pins disappear on reset."""
    from gan_amd import autotune
    case = (4, 16, 16, 32, 64, 5, 2, 'SAME')
    from gpu_util import geom_for
    g = geom_for(*case)
    try:
        res = autotune.tune_geometry(g, reps=2)
        assert set(res) == {0, 1, 2} and all(us > 0 for _, _, us, _ in res.values())
        assert autotune.tune_geometry(g) is res          # cached
        _conv_case(L, case)
        lin = autotune.tune_linear(8, 40, 24, reps=2)
        assert set(lin) == {0, 1, 2}
        # saved plans are pinned again by load(), without timing anything
        import tempfile
        path = os.path.join(tempfile.mkdtemp(), "plans.json")
        autotune.save(path)
        before = autotune.report()
        autotune.reset()
        autotune.load(path)
        assert autotune.report() == before
        c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for mode in (0, 1, 2):
            L.check(L.lib.gan_conv_plan_get(mode, ctypes.byref(g), ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
            want = before[g.key()][mode][:2]      # the library rounds a split request to equal k ranges
            assert thin.value or (c.value == want[0] and 0 < sp.value <= want[1])
        _conv_case(L, case)
    finally:
        autotune.reset()
    assert autotune.report() == {}


@pytest.mark.parametrize("case", [(3, 14, 16, 32, 80, 5, 2), (1, 17, 12, 16, 144, 3, 1), (2, 16, 16, 64, 128, 5, 2)])
def test_split_kernel_gather_variants_agree_bitwise(L, case, monkeypatch):
    """The cheap gather arithmetic of the split kernels (scalar tap decode, one pixel decode per four) only
    changes how addresses are formed: results are bit-identical to the general path."""
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = case
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    x = torch.randn(n, h, w, cb, device='cuda')
    wt = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    dy = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    outs = []
    for slow in (False, True):
        if slow:
            L.knob_set("slow_gather", 1)
        y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(wt)
        wp, wn = _ws(L, g)
        st = L.stream()
        c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.check(L.lib.gan_conv_plan_get(0, gp, ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
        assert c.value >= 4
        L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(wt), None, L.ptr(y), wp, wn, st))
        L.check(L.lib.gan_conv2d_dgrad(gp, L.ptr(dy), L.ptr(wt), L.ptr(dx), wp, wn, st))
        L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(x), L.ptr(dy), L.ptr(dw), None, 0, wp, wn, st))
        torch.cuda.synchronize()
        outs.append((y, dx, dw))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(3, 14, 16, 32, 80, 5, 2), (1, 17, 13, 16, 144, 3, 1), (5, 15, 15, 3, 24, 5, 2),
                                  (2, 9, 11, 21, 74, 5, 2), (2, 16, 16, 64, 128, 5, 2)])
@pytest.mark.parametrize("cfg", [None, 2, 6, 8])
def test_xcd_grouped_tile_order_is_a_permutation_of_the_launch_order(L, case, cfg):
    """The contraction kernels map launch positions to tiles so that tiles sharing operand bytes run on one XCD.
    The map must be a bijection for every grid size (also when the tile count is no multiple of 8, and with
    parity classes of unequal size): each tile computed exactly once gives bit-identical outputs to the plain
    launch order; a tile computed twice or never would leave poison or a stale value in the output."""
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = case
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    x = torch.randn(n, h, w, cb, device='cuda')
    wt = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    dy = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    if cfg is not None and not (cfg >= 4 and (cb % 4 or cs % 4)):
        L.knob_set("force_cfg", cfg)
    L.knob_set("no_thin", 1)
    outs = []
    for order in (1, 0, 2, 3, 4, 1234, -2):
        if order >= 0:
            L.knob_set("tile_order", order)
        else:           # the same through a pinned plan (cfg + 16 x order), as the tuner sets it
            L.knob_set("tile_order", 0)
            c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            plans = []
            for mode in (0, 1, 2):      # the plans of the runs above (a forced configuration included)
                L.check(L.lib.gan_conv_plan_get(mode, gp, ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
                plans.append((c.value, sp.value))
            L.knob_set("force_cfg", -1)
            for mode, (pc, ps) in enumerate(plans):
                L.check(L.lib.gan_conv_plan_set(mode, gp, pc % 16 + 16 * (-order), ps))
        y, dx, dw = (torch.full_like(t, float('nan')) for t in (dy, x, wt))
        wp, wn = _ws(L, g)
        st = L.stream()
        L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(wt), None, L.ptr(y), wp, wn, st))
        L.check(L.lib.gan_conv2d_dgrad(gp, L.ptr(dy), L.ptr(wt), L.ptr(dx), wp, wn, st))
        L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(x), L.ptr(dy), L.ptr(dw), None, 0, wp, wn, st))
        torch.cuda.synchronize()
        outs.append((y, dx, dw))
    L.lib.gan_conv_plan_clear()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("case,accumulate", [((4, 16, 16, 64, 128, 5, 2), 0), ((2, 12, 12, 3, 16, 5, 2), 1),
                                             ((64, 8, 8, 16, 10, 3, 1), 0)])
def test_bias_gradient_carried_by_the_slab_reduction_is_bit_identical(L, case, accumulate, monkeypatch):
    """Weight gradient + bias gradient: the column sum's first stage runs ahead of the contraction and its last
    stage rides on the slab reduction's launch; as launches of their own (the test knob) the same numbers come
    out, for conv, transposed conv and linear, overwriting or accumulating."""
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = case
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    x = torch.randn(n, h, w, cb, device='cuda')
    dy = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    xl, dyl = torch.randn(n * 8, 96, device='cuda'), torch.randn(n * 8, 40, device='cuda')
    outs = []
    for separate in (False, True):
        if separate:
            L.knob_set("separate_colsum", 1)
        torch.manual_seed(3)
        dw, db = torch.randn(k, k, cb, cs, device='cuda'), torch.randn(cs, device='cuda')       # accumulate onto these
        dwt, dbt = torch.randn(k, k, cb, cs, device='cuda'), torch.randn(cb, device='cuda')
        dm, dbl = torch.randn(96, 40, device='cuda'), torch.randn(40, device='cuda')
        wp, wn = _ws(L, g)
        st = L.stream()
        L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), accumulate, wp, wn, st))
        # transposed conv: x is the small tensor, dy the big one, the bias has cb entries
        L.check(L.lib.gan_deconv2d_wgrad(gp, L.ptr(dy), L.ptr(x), L.ptr(dwt), L.ptr(dbt), accumulate, wp, wn, st))
        nb = int(L.lib.gan_linear_workspace_bytes(n * 8, 96, 40))
        lp, ln = L.workspace(nb)
        L.check(L.lib.gan_linear_wgrad(n * 8, 96, 40, L.ptr(xl), L.ptr(dyl), L.ptr(dm), L.ptr(dbl), accumulate, lp, ln, st))
        torch.cuda.synchronize()
        outs.append((dw, db, dwt, dbt, dm, dbl))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # and the bias gradients are the column sums
    if not accumulate:
        torch.testing.assert_close(outs[0][1].double(), dy.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(outs[0][3].double(), x.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("case", [(4, 16, 16, 64, 128, 5, 2), (3, 14, 16, 32, 80, 5, 2), (2, 17, 13, 16, 144, 3, 1),
                                  (8, 8, 8, 256, 512, 5, 2)])
@pytest.mark.parametrize("cfg", [4, 5, 6, 7, 8])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_bias_gradient_carried_by_the_contraction_launch_is_bit_identical(L, case, cfg, accumulate):
    """A weight gradient without split-K slabs has no reduction launch: the last stage of the bias gradient's column
    sum then rides on extra workgroups of the contraction launch itself (256- and 512-thread workgroups alike)
    instead of being a launch of its own (knob no_carry).  Same numbers either way, for conv and transposed conv,
    overwriting or accumulating."""
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = case
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    torch.manual_seed(17)
    x = torch.randn(n, h, w, cb, device='cuda')
    dy = torch.randn(n, g.hs, g.ws, cs, device='cuda')
    L.knob_set("force_cfg", cfg)
    L.knob_set("force_splits", 1)
    outs = []
    for no_carry in (0, 1, 0):
        L.knob_set("no_carry", no_carry)
        torch.manual_seed(5)
        dw, db = torch.randn(k, k, cb, cs, device='cuda'), torch.randn(cs, device='cuda')      # accumulated onto
        dwt, dbt = torch.randn(k, k, cb, cs, device='cuda'), torch.randn(cb, device='cuda')
        wp, wn = _ws(L, g)
        st = L.stream()
        L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(x), L.ptr(dy), L.ptr(dw), L.ptr(db), accumulate, wp, wn, st))
        # (transposed conv: the SMALL tensor is its input, the big one its output gradient)
        L.check(L.lib.gan_deconv2d_wgrad(gp, L.ptr(dy), L.ptr(x), L.ptr(dwt), L.ptr(dbt), accumulate, wp, wn, st))
        torch.cuda.synchronize()
        outs.append((dw, db, dwt, dbt))
    for other in outs[1:]:
        for a, b_ in zip(outs[0], other):
            assert torch.isfinite(a).all() and torch.equal(a, b_)
    if not accumulate:
        torch.testing.assert_close(outs[0][1].double(), dy.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(outs[0][3].double(), x.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-4)


def test_split_bf16_arithmetic_is_as_accurate_as_the_fp32_mfma(L):
    """The default contraction arithmetic (fp32 operands split exactly into three bf16 parts, six
    part products on the bf16 matrix pipe, fp32 accumulation) against the float64 oracle, next to
    the exact fp32 MFMA (gan_set_exact_f32): both far inside the 1e-4 bar and within a small factor of
    each other -- the difference is summation order, not operand precision."""
    from gpu_util import dev, geom_for, host
    n, h, w, cb, cs, k, s, padding = DCGAN_CONVS[2]
    x = RNG.standard_normal((n, h, w, cb)).astype(np.float32)
    wt = (RNG.standard_normal((k, k, cb, cs)) * 0.05).astype(np.float32)
    g = geom_for(n, h, w, cb, cs, k, s, padding)
    y_ref = T.conv2d_forward(x.astype(np.float64), wt.astype(np.float64), s, s, padding)
    dy = RNG.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = T.conv2d_backward_input(dy.astype(np.float64), wt.astype(np.float64), x.shape, s, s, padding)
    dw_ref = T.conv2d_backward_filter(x.astype(np.float64), dy.astype(np.float64), k, k, s, s, padding)
    xd, wd, dyd = dev(x), dev(wt), dev(dy)
    y, dx, dw = torch.empty(y_ref.shape, device='cuda'), torch.empty(x.shape, device='cuda'), torch.empty(wt.shape, device='cuda')
    gp = ctypes.byref(g)
    c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    errs = {}
    try:
        for exact in (1, 0):
            L.lib.gan_set_exact_f32(exact)
            wp, wn = _ws(L, g)
            st = L.stream()
            L.check(L.lib.gan_conv_plan_get(0, gp, ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
            assert (c.value >= 4) == (exact == 0)
            L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(xd), L.ptr(wd), None, L.ptr(y), wp, wn, st))
            L.check(L.lib.gan_conv2d_dgrad(gp, L.ptr(dyd), L.ptr(wd), L.ptr(dx), wp, wn, st))
            L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(xd), L.ptr(dyd), L.ptr(dw), None, 0, wp, wn, st))
            torch.cuda.synchronize()
            errs[exact] = [float(np.abs(host(t) - r).max() / np.abs(r).max())
                           for t, r in ((y, y_ref), (dx, dx_ref), (dw, dw_ref))]
    finally:
        L.lib.gan_set_exact_f32(0)
    for e_exact, e_split in zip(errs[1], errs[0]):
        assert e_exact < 5e-6 and e_split < 5e-6, errs
        assert e_split < 4 * e_exact + 2e-7, errs


def test_fork_sums_the_gradients_of_its_consumers(L):
    """ops.fork hands one tensor to several consumers; the backward pass adds their gradients with the
    library's add kernel.  Without a backward pass to run it is the identity."""
    from gan_amd import ops
    x = torch.randn(3, 4, 4, 8, device='cuda', requires_grad=True)
    a, b, c = ops.fork(x, 3)
    assert a.data_ptr() == x.data_ptr() and torch.equal(b, x)
    wa, wb, wc = (torch.randn_like(x) for _ in range(3))
    torch.autograd.backward([a, b, c], [wa, wb, wc])
    torch.cuda.synchronize()
    assert torch.equal(x.grad, (wa + wb) + wc)
    with torch.no_grad():
        assert ops.fork(x)[1] is x
    assert ops.fork(x.detach())[0] is not None and ops.fork(x.detach())[0].requires_grad is False


def test_weight_and_bias_gradient_of_unaligned_views(L):
    """Operand pointers that are not 16-byte aligned (views one float into a buffer) switch the weight-gradient
    contraction from the split kernel to the fp32 kernel, whose plan takes more split-K slabs: the workspace
    query covers both plans, and the bias-gradient partial sums that share the workspace stay intact."""
    from gpu_util import geom_for, assert_close, host
    g = geom_for(2, 16, 16, 128, 256, 5, 2, 'SAME')          # filter gradient of 3200 x 256: few tiles, many slabs
    gp = ctypes.byref(g)
    torch.manual_seed(11)
    xb = torch.randn(2 * 16 * 16 * 128 + 1, device='cuda')
    dyb = torch.randn(2 * 8 * 8 * 256 + 1, device='cuda')
    outs = []
    for shift in (0, 1):
        x = xb[shift:shift + 2 * 16 * 16 * 128]
        dy = dyb[shift:shift + 2 * 8 * 8 * 256]
        if shift == 0:
            x0, dy0 = x.clone(), dy.clone()
        else:
            x.copy_(x0)
            dy.copy_(dy0)
            assert x.data_ptr() % 16 == 4 and dy.data_ptr() % 16 == 4
        dw, db = torch.empty(5, 5, 128, 256, device='cuda'), torch.empty(256, device='cuda')
        wp, wn = _ws(L, g)
        L.check(L.lib.gan_conv2d_wgrad(gp, x.data_ptr(), dy.data_ptr(), L.ptr(dw), L.ptr(db), 0, wp, wn, L.stream()))
        torch.cuda.synchronize()
        outs.append((host(dw), host(db)))
    ref_db = host(dy0.view(-1, 256).double().sum(0))
    for dw, db in outs:
        assert_close(db, ref_db, "bias gradient", tol=1e-5)
    assert_close(outs[1][0], outs[0][0], "weight gradient of the unaligned views")


# n, h, w, cs, k, stride, padding: forward passes over 3-channel images; the one-pass kernel serves the 5 x 5 ones
RGB_FWD_CASES = [
    (64, 64, 64, 64, 5, 2, 'SAME'),        # DCGAN-64 / Text2Image d_h0 (benchmark size)
    (3, 37, 29, 64, 5, 2, 'SAME'),         # ragged: 285 pixels, odd image sides, pad on every border
    (2, 33, 33, 128, 5, 2, 'SAME'),        # two column tiles
    (1, 40, 36, 64, 7, 1, 'REFLECT'),      # CycleGAN stem: 7 x 7 over the mirrored border (general kernel either way)
    (2, 21, 23, 64, 7, 1, 'SAME'),         # 7 x 7 with zero padding
    (2, 30, 30, 64, 5, 1, 'VALID'),
    (1, 16, 16, 64, 5, 2, 'SAME'),         # fewer pixels than one workgroup
]


@pytest.mark.parametrize("case", RGB_FWD_CASES)
@pytest.mark.parametrize("act", [None, 'lrelu'])
def test_forward_over_rgb_images_without_slice_pipeline(L, case, act):
    """The one-pass kernel for forward convolutions over 3-channel images (bias and fused activation on store) against
    the float64 oracle, and against the general kernels it replaces (knob rgb_direct = 0): same arithmetic -- three
    bf16 parts, six part products -- in another summation order."""
    from gpu_util import assert_close, dev, geom_for, host
    n, h, w, cs, k, s, padding = case
    x = RNG.standard_normal((n, h, w, 3)).astype(np.float32)
    wt = (RNG.standard_normal((k, k, 3, cs)) * 0.05).astype(np.float32)
    b = RNG.standard_normal(cs).astype(np.float32)
    g = geom_for(n, h, w, 3, cs, k, s, padding)
    x64 = x.astype(np.float64)
    if padding == 'REFLECT':
        y_ref = T.conv2d_forward(T.reflect_pad(x64, (k - 1) // 2), wt.astype(np.float64), 1, 1, 'VALID') + b
    else:
        y_ref = T.conv2d_forward(x64, wt.astype(np.float64), s, s, padding) + b
    if act == 'lrelu':
        y_ref = T.lrelu(y_ref, 0.2)
    x_d, w_d, b_d = dev(x), dev(wt), dev(b)
    wp, wn = _ws(L, g)
    outs = []
    for direct in (1, 0):
        L.knob_set("rgb_direct", direct)
        y = torch.full(y_ref.shape, float('nan'), device='cuda')
        gp = ctypes.byref(g)
        if padding == 'REFLECT':
            if act is not None:
                continue
            L.check(L.lib.gan_conv2d_fwd_reflect(gp, L.ptr(x_d), L.ptr(w_d), L.ptr(b_d), L.ptr(y), wp, wn, L.stream()))
        elif act is None:
            L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x_d), L.ptr(w_d), L.ptr(b_d), L.ptr(y), wp, wn, L.stream()))
        else:
            L.check(L.lib.gan_conv2d_fwd_act(gp, L.ptr(x_d), L.ptr(w_d), L.ptr(b_d), L.ptr(y), L.ACT_LRELU, 0.2, 1.0, 0.0, wp, wn,
                                             L.stream()))
        torch.cuda.synchronize()
        assert_close(host(y), y_ref, "rgb forward %s (direct=%d)" % (case, direct))
        outs.append(y)
    if len(outs) == 2:
        assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[1].abs().max())
