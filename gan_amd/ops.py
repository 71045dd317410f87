"""Layer primitives with the reference's `models/ops.py` signatures, backed by libgan_hip.

Drop-in surface (names, positional order and defaults as in the reference, ops.py:30-115):
    batch_norm(epsilon=1e-5, momentum=0.9, name="batch_norm") / __call__(x, train=True)
    conv_cond_concat(x, y)
    conv2d(input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, padding='SAME', name="conv2d")
    deconv2d(input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, name="deconv2d", with_w=False)
    lrelu(x, leak=0.2, name="lrelu")
    linear(input_, output_size, scope_name=None, stddev=0.02, bias_start=0.0, with_w=False, reuse=None)
    concat(tensors, axis)
plus what the model files take from TensorFlow directly (relu, tanh, sigmoid, reflect pad, the
losses) and `instance_norm`, which the reference does not have.

Tensors are dense fp32 NHWC torch tensors in GPU memory.  torch.autograd only sequences the
backward; every forward and backward computation is a call into the HIP library.  Weight
gradients are not returned through autograd: the backward of each layer writes them straight into
the owning optimiser group's gradient arena when that variable is being trained in the current
update (`Variable.wants_grad`), which is how `minimize(loss, var_list=...)` restricts the work.
"""
import ctypes
import math

import torch

from . import _lib
from . import autotune
from . import variables as V
from ._lib import ACT_CODES, check, lib, ptr, stream

variable_scope = V.variable_scope


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
_token = {}


_force_tape = [False]


class taping(object):
    """Inside the block every layer records its backward, also layers whose weights are not being
    updated right now and whose input needs no gradient (a pass that a later update will reuse)."""

    def __enter__(self):
        self.previous, _force_tape[0] = _force_tape[0], True

    def __exit__(self, *exc):
        _force_tape[0] = self.previous


def _grad_token(device, var=None):
    """A leaf that requires grad, threaded through a layer so that its backward runs (and can emit
    weight gradients) even when its data input needs no gradient.  Layers whose weights are frozen
    in the current update get none: with a no-gradient input they drop out of the backward pass
    (tf.gradients prunes the same sub-graph, e.g. G_B2A(real_B) in cyclegan.py's g_loss_a2b)."""
    if var is not None and not var.wants_grad and not _force_tape[0]:
        return None
    key = (device, bool(var is not None and var.early))
    t = _token.get(key)
    if t is None:
        t = torch.zeros(1, device=device, requires_grad=True)
        _token[key] = t
    return t


def same_pad(in_size, k, s):
    """TensorFlow 'SAME': out = ceil(in/s), total = max((out-1)*s+k-in, 0), before = total//2."""
    out = int(math.ceil(float(in_size) / float(s)))
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


def _conv_geom(n, h, w, c_big, c_small, k_h, k_w, d_h, d_w, padding, small_hw=None):
    if padding == 'SAME':
        hs, pt = same_pad(h, k_h, d_h)
        ws, pl = same_pad(w, k_w, d_w)
    elif padding == 'REFLECT':
        if k_h % 2 == 0 or k_w % 2 == 0 or d_h != 1 or d_w != 1:
            raise ValueError("REFLECT padding needs odd filter sizes and stride 1 (got %dx%d, stride %dx%d)" % (k_h, k_w, d_h, d_w))
        hs, ws, pt, pl = h, w, (k_h - 1) // 2, (k_w - 1) // 2
        if pt >= h or pl >= w:
            raise ValueError("REFLECT padding of %dx%d does not fit a %dx%d image" % (pt, pl, h, w))
    elif padding == 'VALID':
        if h < k_h or w < k_w:
            raise ValueError("VALID convolution: %dx%d filter larger than %dx%d input" % (k_h, k_w, h, w))
        hs, ws, pt, pl = (h - k_h) // d_h + 1, (w - k_w) // d_w + 1, 0, 0
    else:
        raise ValueError("padding must be 'SAME', 'VALID' or 'REFLECT', got %r" % (padding,))
    if small_hw is not None and tuple(small_hw) != (hs, ws):
        raise ValueError("conv2d_transpose: output_shape %s does not match input spatial size %s "
                         "(a stride-%dx%d SAME conv of the output gives %s)"
                         % ((h, w), tuple(small_hw), d_h, d_w, (hs, ws)))
    return _lib.conv_geom(n, h, w, c_big, hs, ws, c_small, k_h, k_w, d_h, d_w, pt, pl)


_ws_cache = {}
autotune.on_change(_ws_cache.clear)     # pinned plans change the split-K workspace sizes


def _conv_ws(g):
    key = ('c',) + g.key()
    n = _ws_cache.get(key)
    if n is None:
        n = _ws_cache[key] = int(lib.gan_conv_workspace_bytes(ctypes.byref(g)))
    return _lib.workspace(n)


def _reflect_dgrad_ws(g):
    key = ('r',) + g.key()
    n = _ws_cache.get(key)
    if n is None:
        n = _ws_cache[key] = int(lib.gan_conv2d_dgrad_reflect_workspace_bytes(ctypes.byref(g)))
    return _lib.workspace(n)


def _lin_ws(rows, i, o):
    key = ('l', rows, i, o)
    n = _ws_cache.get(key)
    if n is None:
        n = _ws_cache[key] = int(lib.gan_linear_workspace_bytes(rows, i, o))
    return _lib.workspace(n)


def _bn_ws(groups, rows, c):
    key = ('b', groups, rows, c)
    n = _ws_cache.get(key)
    if n is None:
        n = _ws_cache[key] = int(lib.gan_bn_workspace_bytes(groups, rows, c))
    return _lib.workspace(n)


# which batch-norm reductions ride on neighbouring contraction launches (all by default; tests and measurements switch
# them one by one): forward statistics, backward sums, column sums for the bias gradient of the layer in front
RIDE_PARTS = {'stats': True, 'sums': True, 'colsum': True, 'fan_out': True}


def _tile_blocks(groups, rows, c, backward):
    """Row blocks per group of the two-launch batch norm for this shape (0: the shape is served by other kernels and no
    reduction can ride on a neighbouring contraction)."""
    key = ('t', groups, rows, c, backward)
    n = _ws_cache.get(key)
    if n is None:
        n = _ws_cache[key] = int(lib.gan_bn_tile_blocks(groups, rows, c, backward))
    return n


def _dense(x):
    if x.dtype != torch.float32:
        raise TypeError("expected a float32 tensor, got %s" % x.dtype)
    return x if x.is_contiguous() else x.contiguous()


# --------------------------------------------------------------------------------------------
# conv2d / deconv2d / linear
# --------------------------------------------------------------------------------------------
class _ActToken(object):
    """Travels with the output of a conv / deconv whose activation is fused into its epilogue (`act=`).  A conv /
    deconv that takes that output as its input multiplies the input gradient it produces by the activation's
    derivative on store (the gated input-gradient entry points) and notes the gradient tensor here; the
    producing layer's backward then finds its incoming gradient already taken through the activation."""
    __slots__ = ('spec', 'gated')

    def __init__(self, spec):
        self.spec = spec          # (activation code, parameter, scale, shift)
        self.gated = None         # (gradient tensor a gated consumer produced, its version counter at that time)

    def note_gated(self, dx):
        self.gated = (dx, dx._version)


class _NormToken(object):
    """Travels with the output of a conv / deconv that feeds a training-mode batch norm (`norm_groups=`): forward, the
    contraction leaves the norm's block statistics here; backward, the norm leaves the column sums of its input
    gradient here for the contraction's bias gradient.  Either side falls back to its own launches when the other
    did not deliver (shape not served, tensor replaced)."""
    __slots__ = ('groups', 'stats', 'colsum')

    def __init__(self, groups):
        self.groups = int(groups)
        self.stats = None         # (partial statistics, address of the tensor they describe, its version counter)
        self.colsum = None        # (gradient tensor, its version counter, partial column sums, number of row blocks)


class _BnToken(object):
    """Travels with the output of a training-mode batch norm.  A conv / deconv that takes that output as its input
    lets the launch that finishes its input gradient also take the block sums of the norm's backward pass over it."""
    __slots__ = ('saved', 'sums')

    def __init__(self):
        self.saved = None         # (x, save_mean, save_rstd, gamma, beta, groups, rows, c, act code, act parameter)
        self.sums = None          # (gradient tensor, its version counter, partial sums)


def _same_tensor(t, ref, version):
    return t.data_ptr() == ref.data_ptr() and t._version == version


def _dgrad_with_bn_sums(op, g, dy, w, dx, in_bn, wp, wn):
    """The input gradient dx of a conv / deconv whose input was the output of the batch norm `in_bn`, leaving that
    norm's backward block sums with the token; False when the norm's shape has no such decomposition."""
    bx, mean, rstd, gamma, beta, groups, rows, c, act, act_param = in_bn.saved
    nb = _tile_blocks(groups, rows, c, 1) if RIDE_PARTS['sums'] else 0
    if not nb or dx.numel() != groups * rows * c:
        return False
    partial = torch.empty(2 * groups * nb * c, dtype=torch.float32, device=dx.device)
    saved = _lib.BnSaved(ptr(bx), ptr(gamma.data), ptr(beta.data), ptr(mean), ptr(rstd), groups, act, act_param)
    check(lib.gan_conv_dgrad_bn_sums(op, ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), ctypes.byref(saved), ptr(partial),
                                     wp, wn, stream()))
    in_bn.sums = (dx, dx._version, partial)
    return True


def _add_with_bn_sums(a, b, out, in_bn):
    """out = a + b, the gradient of a batch norm's output that had two consumers, leaving that norm's backward block
    sums with the token; False when the norm's shape has no such decomposition."""
    bx, mean, rstd, gamma, beta, groups, rows, c, act, act_param = in_bn.saved
    nb = _tile_blocks(groups, rows, c, 1) if RIDE_PARTS['fan_out'] else 0
    if not nb or out.numel() != groups * rows * c or (a.data_ptr() | b.data_ptr() | out.data_ptr()) & 15:
        return False
    partial = torch.empty(2 * groups * nb * c, dtype=torch.float32, device=out.device)
    saved = _lib.BnSaved(ptr(bx), ptr(gamma.data), ptr(beta.data), ptr(mean), ptr(rstd), groups, act, act_param)
    check(lib.gan_add_bn_sums(ptr(a), ptr(b), ptr(out), ctypes.byref(saved), rows, c, ptr(partial), stream()))
    in_bn.sums = (out, out._version, partial)
    return True


def _wgrad(op, plain, g, x, dy, w, b, norm_tok, wp, wn):
    """Weight and bias gradient; the bias gradient from the column sums the batch norm behind this layer left."""
    dw, acc = w.grad_target()
    db, accb = b.grad_target()
    assert acc == accb
    cs = None
    if norm_tok is not None:
        cs, norm_tok.colsum = norm_tok.colsum, None
    if cs is not None and _same_tensor(dy, cs[0], cs[1]):
        check(lib.gan_conv_wgrad_colsum(op, ctypes.byref(g), ptr(x), ptr(dy), ptr(dw), ptr(db), acc, ptr(cs[2]), cs[3], wp, wn,
                                        stream()))
    else:
        check(plain(ctypes.byref(g), ptr(x), ptr(dy), ptr(dw), ptr(db), acc, wp, wn, stream()))


def _fwd_with_bn_stats(op, g, x, w, b, y, norm_tok, wp, wn):
    """y = conv / deconv of x, leaving the block statistics of the batch norm that follows with the token; False when
    that norm's shape has no such decomposition."""
    c = y.shape[-1]
    rows_all = y.numel() // c
    if rows_all % norm_tok.groups:
        return False
    nb = _tile_blocks(norm_tok.groups, rows_all // norm_tok.groups, c, 0) if RIDE_PARTS['stats'] else 0
    if not nb:
        return False
    partial = torch.empty(2 * norm_tok.groups * nb * c, dtype=torch.float32, device=y.device)
    check(lib.gan_conv_fwd_bn_stats(op, ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), norm_tok.groups, ptr(partial),
                                    wp, wn, stream()))
    norm_tok.stats = (partial, y.data_ptr(), y._version)
    return True


# Stride-1 convolutions with at most 4 channels on one side and a filter of 25 taps or more (the 7 x 7 image head and
# stem of the CycleGAN generators) run as a dense product plus a sum over taps (include/gan_hip.h, gan_taps_*).
TAPS_PATH = [True]


def _taps_kind(g, reflect):
    """'out': few OUTPUT channels (forward, weight and input gradient by taps); 'in': few INPUT channels (input gradient
    by taps); None: the general kernels."""
    if not TAPS_PATH[0] or g.sh != 1 or g.sw != 1 or g.kh != g.kw or not (g.kh & 1) or g.kh * g.kw < 25:
        return None
    if g.hs != g.hb or g.ws != g.wb or g.pt != (g.kh - 1) // 2 or g.pl != (g.kw - 1) // 2:
        return None
    if g.cs <= 4 and g.cb >= 16 and g.cb % 4 == 0:
        return 'out'
    if g.cb <= 4 and g.cs >= 16 and g.cs % 4 == 0:
        return 'in'
    return None


def _taps_ld(g, thin):
    return (g.kh * g.kw * thin + 3) & ~3


def _taps_linear(kind, rows, i, o, a, b, out, extra=None):
    """One of the three dense products of the tap decomposition, on the contraction kernels."""
    if autotune.enabled():
        autotune.tune_linear(rows, i, o)
    wp, wn = _lin_ws(rows, i, o)
    if kind == 'fwd':
        check(lib.gan_linear_fwd(rows, i, o, ptr(a), ptr(b), None, ptr(out), wp, wn, stream()))
    elif kind == 'dgrad':
        check(lib.gan_linear_dgrad(rows, i, o, ptr(a), ptr(b), ptr(out), wp, wn, stream()))
    else:
        check(lib.gan_linear_wgrad(rows, i, o, ptr(a), ptr(b), ptr(out), ptr(extra), 0, wp, wn, stream()))


def _taps_conv_fwd(g, x, w, b, y, reflect):
    """Forward of a few-output-channel convolution; returns the repacked filter (the backward pass multiplies by it)."""
    taps, ld, rows = g.kh * g.kw, _taps_ld(g, g.cs), g.n * g.hb * g.wb
    wpk = torch.empty((g.cb, ld), dtype=torch.float32, device=x.device)
    check(lib.gan_taps_pack_filter(ptr(w.data), ptr(wpk), taps, g.cb, g.cs, ld, 0, stream()))
    z = torch.empty((rows, ld), dtype=torch.float32, device=x.device)
    _taps_linear('fwd', rows, g.cb, ld, x, wpk, z)
    check(lib.gan_taps_sum_fwd(ptr(z), ptr(b.data), ptr(y), g.n, g.hb, g.wb, g.cs, g.kh, g.kw, ld, int(reflect), stream()))
    return wpk


def _taps_conv_bwd(g, x, dy, w, b, wpk, reflect, want_dx):
    """Weight, bias and input gradient of a few-output-channel convolution."""
    taps, ld, rows = g.kh * g.kw, _taps_ld(g, g.cs), g.n * g.hb * g.wb
    e = torch.empty((rows, ld), dtype=torch.float32, device=x.device)
    check(lib.gan_taps_spread_bwd(ptr(dy), ptr(e), g.n, g.hb, g.wb, g.cs, g.kh, g.kw, ld, int(reflect), stream()))
    if w.wants_grad:
        dw, acc = w.grad_target()
        db, accb = b.grad_target()
        assert acc == accb
        dwp = torch.empty((g.cb, ld), dtype=torch.float32, device=x.device)
        dbp = torch.empty((ld,), dtype=torch.float32, device=x.device)
        _taps_linear('wgrad', rows, g.cb, ld, x, e, dwp, dbp)
        check(lib.gan_taps_unpack_filter_grad(ptr(dwp), ptr(dbp), ptr(dw), ptr(db), taps, g.cb, g.cs, ld, 0, acc, stream()))
    if not want_dx:
        return None
    dx = torch.empty_like(x)
    _taps_linear('dgrad', rows, g.cb, ld, e, wpk, dx)
    return dx


def _taps_stem_dgrad(g, dy, w, x, reflect):
    """Input gradient of a few-INPUT-channel convolution (filter [t][o][c])."""
    taps, ld, rows = g.kh * g.kw, _taps_ld(g, g.cb), g.n * g.hb * g.wb
    wpk = torch.empty((g.cs, ld), dtype=torch.float32, device=dy.device)
    check(lib.gan_taps_pack_filter(ptr(w.data), ptr(wpk), taps, g.cs, g.cb, ld, 1, stream()))
    z = torch.empty((rows, ld), dtype=torch.float32, device=dy.device)
    _taps_linear('fwd', rows, g.cs, ld, dy, wpk, z)
    dx = torch.empty_like(x)
    check(lib.gan_taps_sum_bwd(ptr(z), ptr(dx), g.n, g.hb, g.wb, g.cb, g.kh, g.kw, ld, int(reflect), stream()))
    return dx


def _act_spec(act, act_param, act_scale, act_shift):
    code = ACT_CODES[act]
    if code == _lib.ACT_NONE and act_scale == 1.0 and act_shift == 0.0:
        return None
    return (code, float(act_param), float(act_scale), float(act_shift))


def _through_own_activation(ctx, y, dy):
    """Gradient w.r.t. the contraction output in front of a fused activation, from the gradient w.r.t. the
    activated output y."""
    tok = ctx.act_tok
    if tok.gated is not None:
        (dx, version), tok.gated = tok.gated, None
        # The incoming gradient must be the gated consumer's tensor, untouched.  A second consumer's gradient is either
        # summed into a new tensor (another address) or, when autograd owns the only reference, added IN PLACE into dx
        # (same address, version counter moved): both are refused.
        if dy.data_ptr() != dx.data_ptr() or dy._version != version:
            raise RuntimeError("the output of a conv / deconv with a fused activation feeds a gated consumer AND something "
                               "else: their gradients were summed after one of them had been taken through the "
                               "activation; apply the activation as an op of its own (ops.lrelu / tanh / ...) here")
        return dy                 # the consumer's input-gradient epilogue has applied the derivative
    code, param, scale, shift = tok.spec
    dy = _dense(dy)
    dz = torch.empty_like(y)
    check(lib.gan_act_bwd(ptr(y), ptr(dy), ptr(dz), y.numel(), code, param, scale, shift, stream()))
    return dz


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, w, b, g, act_tok, in_tok, reflect=False, norm_tok=None, in_bn=None):
        x = _dense(x)
        y = torch.empty((g.n, g.hs, g.ws, g.cs), dtype=torch.float32, device=x.device)
        wp, wn = _conv_ws(g)
        ctx.reflect = reflect
        ctx.op = _lib.OP_CONV2D_REFLECT if reflect else _lib.OP_CONV2D
        ctx.norm_tok, ctx.in_bn = norm_tok, in_bn
        if reflect and (act_tok is not None or in_tok is not None):
            raise ValueError("conv2d: REFLECT padding does not combine with a fused activation")
        ctx.taps = _taps_kind(g, reflect) if (act_tok is None and in_tok is None) else None
        if ctx.taps == 'out':
            ctx.save_for_backward(x, _taps_conv_fwd(g, x, w, b, y, reflect))
        elif norm_tok is not None and act_tok is None and _fwd_with_bn_stats(ctx.op, g, x, w, b, y, norm_tok, wp, wn):
            ctx.save_for_backward(x)
        elif reflect:
            check(lib.gan_conv2d_fwd_reflect(ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), wp, wn, stream()))
            ctx.save_for_backward(x)
        elif act_tok is None:
            check(lib.gan_conv2d_fwd(ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), wp, wn, stream()))
            ctx.save_for_backward(x)
        else:
            code, param, scale, shift = act_tok.spec
            check(lib.gan_conv2d_fwd_act(ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), code, param, scale, shift,
                                         wp, wn, stream()))
            ctx.save_for_backward(x, y)
        ctx.w, ctx.b, ctx.g, ctx.act_tok, ctx.in_tok = w, b, g, act_tok, in_tok
        return y

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        w, b, g = ctx.w, ctx.b, ctx.g
        if ctx.act_tok is not None:
            dy = _through_own_activation(ctx, ctx.saved_tensors[1], dy)
        dy = _dense(dy)
        if ctx.taps == 'out':
            return (_taps_conv_bwd(g, x, dy, w, b, ctx.saved_tensors[1], ctx.reflect, ctx.needs_input_grad[0]),) + (None,) * 9
        wp, wn = _conv_ws(g)
        if w.wants_grad:
            _wgrad(ctx.op, lib.gan_conv2d_wgrad_reflect if ctx.reflect else lib.gan_conv2d_wgrad, g, x, dy, w, b, ctx.norm_tok,
                   wp, wn)
        dx = None
        if ctx.needs_input_grad[0] and ctx.taps == 'in':
            dx = _taps_stem_dgrad(g, dy, w, x, ctx.reflect)
        elif ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.in_bn is not None and ctx.in_tok is None and ctx.reflect and g.pt == g.pl and \
                    _dgrad_with_bn_sums(ctx.op, g, dy, w, dx, ctx.in_bn, *_reflect_dgrad_ws(g)):
                pass
            elif ctx.in_bn is not None and ctx.in_tok is None and not ctx.reflect and \
                    _dgrad_with_bn_sums(ctx.op, g, dy, w, dx, ctx.in_bn, wp, wn):
                pass
            elif ctx.reflect:
                # input gradient of the VALID convolution on the reflect-padded image, folded back onto the image by the
                # kernel that sums the split-K slabs
                if g.pt != g.pl:
                    raise ValueError("REFLECT padding: square filters only")
                wpp, wnp = _reflect_dgrad_ws(g)
                check(lib.gan_conv2d_dgrad_reflect(ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), wpp, wnp, stream()))
            elif ctx.in_tok is None:
                check(lib.gan_conv2d_dgrad(ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), wp, wn, stream()))
            else:       # x is the activated output of the layer below: hand it the gradient in front of its activation
                code, param, scale, shift = ctx.in_tok.spec
                check(lib.gan_conv2d_dgrad_gated(ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), ptr(x), code, param, scale,
                                                 shift, wp, wn, stream()))
                ctx.in_tok.note_gated(dx)
        return dx, None, None, None, None, None, None, None, None, None


class _Deconv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, w, b, g, act_tok, in_tok, norm_tok=None, in_bn=None):
        x = _dense(x)
        y = torch.empty((g.n, g.hb, g.wb, g.cb), dtype=torch.float32, device=x.device)
        wp, wn = _conv_ws(g)
        ctx.norm_tok, ctx.in_bn = norm_tok, in_bn
        if norm_tok is not None and act_tok is None and _fwd_with_bn_stats(_lib.OP_DECONV2D, g, x, w, b, y, norm_tok, wp, wn):
            ctx.save_for_backward(x)
        elif act_tok is None:
            check(lib.gan_deconv2d_fwd(ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), wp, wn, stream()))
            ctx.save_for_backward(x)
        else:
            code, param, scale, shift = act_tok.spec
            check(lib.gan_deconv2d_fwd_act(ctypes.byref(g), ptr(x), ptr(w.data), ptr(b.data), ptr(y), code, param, scale,
                                           shift, wp, wn, stream()))
            ctx.save_for_backward(x, y)
        ctx.w, ctx.b, ctx.g, ctx.act_tok, ctx.in_tok = w, b, g, act_tok, in_tok
        return y

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        w, b, g = ctx.w, ctx.b, ctx.g
        if ctx.act_tok is not None:
            dy = _through_own_activation(ctx, ctx.saved_tensors[1], dy)
        dy = _dense(dy)
        wp, wn = _conv_ws(g)
        if w.wants_grad:
            _wgrad(_lib.OP_DECONV2D, lib.gan_deconv2d_wgrad, g, x, dy, w, b, ctx.norm_tok, wp, wn)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.in_bn is not None and ctx.in_tok is None and \
                    _dgrad_with_bn_sums(_lib.OP_DECONV2D, g, dy, w, dx, ctx.in_bn, wp, wn):
                pass
            elif ctx.in_tok is None:
                check(lib.gan_deconv2d_dgrad(ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), wp, wn, stream()))
            else:
                code, param, scale, shift = ctx.in_tok.spec
                check(lib.gan_deconv2d_dgrad_gated(ctypes.byref(g), ptr(dy), ptr(w.data), ptr(dx), ptr(x), code, param,
                                                   scale, shift, wp, wn, stream()))
                ctx.in_tok.note_gated(dx)
        return dx, None, None, None, None, None, None, None, None


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, m, b):
        x = _dense(x)
        rows, i = x.shape
        o = m.shape[1]
        y = torch.empty((rows, o), dtype=torch.float32, device=x.device)
        wp, wn = _lin_ws(rows, i, o)
        check(lib.gan_linear_fwd(rows, i, o, ptr(x), ptr(m.data), ptr(b.data), ptr(y), wp, wn, stream()))
        ctx.save_for_backward(x)
        ctx.m, ctx.b = m, b
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        m, b = ctx.m, ctx.b
        dy = _dense(dy)
        rows, i = x.shape
        o = m.shape[1]
        wp, wn = _lin_ws(rows, i, o)
        if m.wants_grad:
            dm, acc = m.grad_target()
            db, accb = b.grad_target()
            assert acc == accb
            check(lib.gan_linear_wgrad(rows, i, o, ptr(x), ptr(dy), ptr(dm), ptr(db), acc, wp, wn, stream()))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.gan_linear_dgrad(rows, i, o, ptr(dy), ptr(m.data), ptr(dx), wp, wn, stream()))
        return dx, None, None, None


def _fused_act_of(t):
    """The token of a conv / deconv output with a fused activation (None for any other tensor)."""
    return getattr(t, '_gan_act', None) if torch.is_grad_enabled() and t.requires_grad else None


def _bn_of(t):
    """The token of a training-mode batch norm's output (None for any other tensor)."""
    return getattr(t, '_gan_bn', None) if torch.is_grad_enabled() and t.requires_grad else None


def _norm_token(norm_groups, spec):
    if norm_groups is None:
        return None
    if spec is not None:
        raise ValueError("norm_groups= (a batch norm follows) does not combine with a fused activation")
    return _NormToken(norm_groups)


def _hand_over(y, act_tok, norm_tok):
    if act_tok is not None:
        y._gan_act = act_tok
    if norm_tok is not None:
        y._gan_norm = norm_tok
        if norm_tok.stats is not None:      # (the version counter as the consumer will see it)
            norm_tok.stats = (norm_tok.stats[0], y.data_ptr(), y._version)
    return y


def conv2d(input_, output_dim, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, padding='SAME', name="conv2d", *,
           act=None, act_param=0.2, act_scale=1.0, act_shift=0.0, norm_groups=None):
    """ops.py:64-74: w [k_h,k_w,Cin,Cout] truncated-normal, biases zeros, tf.nn.conv2d + bias_add.

    Extension (keyword-only; the default is the reference's op): `act` fuses the activation the caller would
    apply to the result -- lrelu(conv2d(..)) of dcgan.py:328 is conv2d(.., act='lrelu') -- into the store, as
    act(conv + bias) * act_scale + act_shift.  Such an output must feed ONE consumer; when that consumer is a
    conv2d / deconv2d, its input-gradient pass also applies this activation's derivative.
    `norm_groups` = k says that the result goes straight into a training-mode batch_norm(.., passes=k) (k = batch size
    for instance_norm): the launch that finishes the convolution then takes that norm's statistics too.  Values are
    unchanged, and a batch norm that does not follow after all is harmless."""
    st = V.default_store()
    n, h, w_, cin = input_.shape
    with st.variable_scope(name):
        w = st.get_variable('w', [k_h, k_w, cin, output_dim], V.truncated_normal_initializer(stddev))
        biases = st.get_variable('biases', [output_dim], V.constant_initializer(0.0))
    g = _conv_geom(n, h, w_, cin, output_dim, k_h, k_w, d_h, d_w, padding)
    if autotune.enabled():
        autotune.tune_geometry(g)
    spec = _act_spec(act, act_param, act_scale, act_shift)
    tok = _ActToken(spec) if spec is not None else None
    if padding == 'REFLECT' and autotune.enabled():
        autotune.tune_geometry(_lib.conv_geom(g.n, g.hb + 2 * g.pt, g.wb + 2 * g.pl, g.cb, g.hs, g.ws, g.cs, g.kh, g.kw, 1, 1, 0, 0))
    ntok = _norm_token(norm_groups, spec)
    y = _Conv2dFn.apply(input_, _grad_token(input_.device, w), w, biases, g, tok, _fused_act_of(input_),
                        padding == 'REFLECT', ntok, _bn_of(input_))
    return _hand_over(y, tok, ntok)


def deconv2d(input_, output_shape, k_h=5, k_w=5, d_h=2, d_w=2, stddev=0.02, name="deconv2d", with_w=False, *,
             act=None, act_param=0.2, act_scale=1.0, act_shift=0.0, norm_groups=None):
    """ops.py:76-99: w [k_h,k_w,Cout,Cin] normal, tf.nn.conv2d_transpose (SAME) + bias_add.
    `output_shape` carries the static batch size, as in the reference.  `act` ...: as for conv2d (the returned
    tensor is then the ACTIVATED one: tanh(deconv2d(..)) of dcgan.py:385 is deconv2d(.., act='tanh')); `norm_groups`: as
    for conv2d."""
    st = V.default_store()
    n, hs, ws, cin = input_.shape
    on, oh, ow, oc = [int(v) for v in output_shape]
    if on != n:
        raise ValueError("deconv2d: output_shape batch %d != input batch %d" % (on, n))
    with st.variable_scope(name):
        w = st.get_variable('w', [k_h, k_w, oc, cin], V.random_normal_initializer(stddev))
        biases = st.get_variable('biases', [oc], V.constant_initializer(0.0))
    g = _conv_geom(n, oh, ow, oc, cin, k_h, k_w, d_h, d_w, 'SAME', small_hw=(hs, ws))
    if autotune.enabled():
        autotune.tune_geometry(g)
    spec = _act_spec(act, act_param, act_scale, act_shift)
    tok = _ActToken(spec) if spec is not None else None
    ntok = _norm_token(norm_groups, spec)
    deconv = _hand_over(_Deconv2dFn.apply(input_, _grad_token(input_.device, w), w, biases, g, tok, _fused_act_of(input_), ntok,
                                          _bn_of(input_)), tok, ntok)
    if with_w:
        return deconv, w, biases
    return deconv


def linear(input_, output_size, scope_name=None, stddev=0.02, bias_start=0.0, with_w=False, reuse=None):
    """ops.py:104-115: Matrix [in,out] normal, bias constant, tf.matmul + bias."""
    st = V.default_store()
    if input_.dim() != 2:
        raise ValueError("linear expects a rank-2 input, got shape %s" % (tuple(input_.shape),))
    with st.variable_scope(scope_name or "Linear", reuse=reuse):
        matrix = st.get_variable("Matrix", [input_.shape[1], output_size], V.random_normal_initializer(stddev))
        bias = st.get_variable("bias", [output_size], V.constant_initializer(bias_start))
    if autotune.enabled():
        autotune.tune_linear(int(input_.shape[0]), int(input_.shape[1]), int(output_size))
    y = _LinearFn.apply(input_, _grad_token(input_.device, matrix), matrix, bias)
    if with_w:
        return y, matrix, bias
    return y


# --------------------------------------------------------------------------------------------
# batch norm / instance norm (fused with the activation that follows)
# --------------------------------------------------------------------------------------------
class moving_update_log(object):
    """Collects, for every training-mode batch norm run inside the block, what is needed to apply
    its moving-average update once more (`replay`).  A model uses it to reuse one forward pass for
    two of the reference's session runs that evaluate the same sub-graph on the same inputs and
    weights: the activations are shared, the moving averages still move twice."""

    current = None

    def __init__(self):
        self.entries = []

    def __enter__(self):
        self.previous, moving_update_log.current = moving_update_log.current, self
        return self

    def __exit__(self, *exc):
        moving_update_log.current = self.previous

    def replay(self):
        """The logged updates again, in log order.  Layers that share eps / momentum / variance flavour go into one
        launch (a whole generator); the jobs of a launch run concurrently, so a layer that is logged twice (its
        moving averages must move twice, one after the other) starts a new launch."""
        batches = []
        for mm, mv, mean, rstd, rows, c, eps, momentum, unbiased in self.entries:
            key = (float(eps), float(momentum), int(unbiased))
            targets = (mm.data.data_ptr(), mv.data.data_ptr())
            if not batches or batches[-1][0] != key or any(t in batches[-1][2] for t in targets):
                batches.append((key, [], set()))
            batches[-1][1].append((mm, mv, mean, rstd, rows, c))
            batches[-1][2].update(targets)
        for (eps, momentum, unbiased), jobs, _ in batches:
            arr = (_lib.BnMovingJob * len(jobs))()
            for j, (mm, mv, mean, rstd, rows, c) in zip(arr, jobs):
                j.moving_mean, j.moving_var, j.save_mean, j.save_rstd = ptr(mm.data), ptr(mv.data), ptr(mean), ptr(rstd)
                j.rows, j.c = int(rows), int(c)
            check(lib.gan_bn_moving_update_multi(arr, len(jobs), eps, momentum, unbiased, stream()))


class _NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, token, gamma, beta, mm, mv, groups, eps, momentum, unbiased, act, act_param, norm_tok=None,
                bn_tok=None):
        stats = None
        if norm_tok is not None:
            stats, norm_tok.stats = norm_tok.stats, None
            if norm_tok.groups != groups or stats is None or stats[1] != x.data_ptr() or stats[2] != x._version or \
                    not x.is_contiguous():
                stats = None
        x = _dense(x)
        c = x.shape[-1]
        rows = x.numel() // c // groups
        y = torch.empty_like(x)
        save_mean = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        save_rstd = torch.empty(groups * c, dtype=torch.float32, device=x.device)
        res = None if residual is None else _dense(residual)
        pmm, pmv = (None if mm is None else ptr(mm.data)), (None if mv is None else ptr(mv.data))
        if stats is not None:      # the convolution in front left the block statistics
            check(lib.gan_bn_fwd_train_from_stats(ptr(x), ptr(gamma.data), ptr(beta.data), pmm, pmv, ptr(res), ptr(y),
                                                  ptr(save_mean), ptr(save_rstd), groups, rows, c, eps, momentum, int(unbiased),
                                                  act, act_param, ptr(stats[0]), stream()))
        else:
            wp, wn = _bn_ws(groups, rows, c)
            check(lib.gan_bn_fwd_train(ptr(x), ptr(gamma.data), ptr(beta.data), pmm, pmv, ptr(res), ptr(y), ptr(save_mean),
                                       ptr(save_rstd), groups, rows, c, eps, momentum, int(unbiased), act, act_param, wp, wn,
                                       stream()))
        ctx.norm_tok, ctx.bn_tok = norm_tok, bn_tok
        if bn_tok is not None:
            bn_tok.saved = (x, save_mean, save_rstd, gamma, beta, groups, rows, c, act, act_param)
        if moving_update_log.current is not None and mm is not None and groups == 1:
            moving_update_log.current.entries.append((mm, mv, save_mean, save_rstd, rows, c, eps, momentum, unbiased))
        ctx.save_for_backward(x, save_mean, save_rstd)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.cfg = (groups, rows, c, act, act_param)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, save_mean, save_rstd = ctx.saved_tensors
        gamma, beta = ctx.gamma, ctx.beta
        groups, rows, c, act, act_param = ctx.cfg
        dy = _dense(dy)
        dg = db = None
        acc = 0
        if gamma.wants_grad:
            dg, acc = gamma.grad_target()
            db, accb = beta.grad_target()
            assert acc == accb
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        sums = None
        if ctx.bn_tok is not None:
            sums, ctx.bn_tok.sums = ctx.bn_tok.sums, None
        nb = _tile_blocks(groups, rows, c, 1) if (dx is not None or dg is not None) else 0
        if nb:
            # block sums left by the consumer's input-gradient launch (only if dy is that very tensor, untouched), and
            # column sums of dx for the bias gradient of the conv / deconv in front
            ready = sums is not None and _same_tensor(dy, sums[0], sums[1])
            partial = sums[2] if ready else torch.empty(2 * groups * nb * c, dtype=torch.float32, device=x.device)
            colsum = None
            if dx is not None and ctx.norm_tok is not None and RIDE_PARTS['colsum']:
                colsum = torch.empty(groups * nb * c, dtype=torch.float32, device=x.device)
            check(lib.gan_bn_bwd_tiled(ptr(x), ptr(gamma.data), ptr(beta.data), ptr(save_mean), ptr(save_rstd), ptr(dy), ptr(dx),
                                       ptr(dg), ptr(db), acc, groups, rows, c, act, act_param, ptr(partial), int(ready),
                                       ptr(colsum), stream()))
            if colsum is not None:
                ctx.norm_tok.colsum = (dx, dx._version, colsum, groups * nb)
        elif dx is not None or dg is not None:
            wp, wn = _bn_ws(groups, rows, c)
            check(lib.gan_bn_bwd(ptr(x), ptr(gamma.data), ptr(beta.data), ptr(save_mean), ptr(save_rstd), ptr(dy),
                                 ptr(dx), ptr(dg), ptr(db), acc, groups, rows, c, act, act_param, wp, wn, stream()))
        dres = dy if ctx.needs_input_grad[1] else None
        return (dx, dres) + (None,) * 13


class batch_norm(object):
    """ops.py:30-44 (tf.contrib.layers.batch_norm with decay=momentum, updates_collections=None,
    scale=True).  Variables beta / gamma / moving_mean / moving_variance are created under the
    CALLER's current scope + name on first use.  Training mode normalises with the biased batch
    variance and updates the moving averages inside every forward.

    Extensions (keyword-only, default to the reference behaviour): `act` / `act_param` fuse the
    activation that follows; `residual` is added after it (the ResNet block's `y + x`).
    `unbiased_moving_var` picks the Bessel-corrected batch variance for the moving average, which is
    what TensorFlow's fused kernel feeds it (the unfused one feeds the biased value: default).
    `passes` = k treats x as k equal batches stacked along axis 0, each normalised with its own
    statistics and each moving the averages in turn: k applications of the layer in one call."""

    unbiased_moving_var = False

    def __init__(self, epsilon=1e-5, momentum=0.9, name="batch_norm"):
        self.epsilon = epsilon
        self.momentum = momentum
        self.name = name

    def __call__(self, x, train=True, act=None, act_param=0.2, residual=None, passes=1):
        st = V.default_store()
        c = x.shape[-1]
        with st.variable_scope(self.name):
            beta = st.get_variable('beta', [c], V.constant_initializer(0.0))
            gamma = st.get_variable('gamma', [c], V.constant_initializer(1.0))
            mm = st.get_variable('moving_mean', [c], V.constant_initializer(0.0), trainable=False)
            mv = st.get_variable('moving_variance', [c], V.constant_initializer(1.0), trainable=False)
        code = ACT_CODES[act]
        if train:
            if x.shape[0] % passes:
                raise ValueError("batch_norm: batch of %d does not split into %d passes" % (x.shape[0], passes))
            btok = _BnToken()
            y = _NormFn.apply(x, residual, _grad_token(x.device, gamma), gamma, beta, mm, mv, int(passes), float(self.epsilon),
                              float(self.momentum), self.unbiased_moving_var, code, float(act_param),
                              getattr(x, '_gan_norm', None), btok)
            y._gan_bn = btok
            return y
        if residual is not None:
            raise ValueError("batch_norm(train=False) does not take a residual")
        x = _dense(x.detach())
        y = torch.empty_like(x)
        check(lib.gan_bn_fwd_eval(ptr(x), ptr(gamma.data), ptr(beta.data), ptr(mm.data), ptr(mv.data), ptr(y),
                                  x.numel() // c, c, float(self.epsilon), code, float(act_param), stream()))
        return y


def instance_norm(x, name="instance_norm", epsilon=1e-5, act=None, act_param=0.2):
    """Extension: per-sample, per-channel normalisation with learned gamma / beta.  Defined as
    batch_norm(train=True) applied to each sample on its own; the reference's CycleGAN reaches the
    same arithmetic through batch_norm at batch size 1 (cyclegan_module.py:16-18, scripts/CycleGAN.sh:2)."""
    st = V.default_store()
    if x.dim() != 4:
        raise ValueError("instance_norm expects NHWC input")
    c = x.shape[-1]
    with st.variable_scope(name):
        beta = st.get_variable('beta', [c], V.constant_initializer(0.0))
        gamma = st.get_variable('gamma', [c], V.constant_initializer(1.0))
    btok = _BnToken()
    y = _NormFn.apply(x, None, _grad_token(x.device, gamma), gamma, beta, None, None, int(x.shape[0]), float(epsilon),
                      0.0, False, ACT_CODES[act], float(act_param), getattr(x, '_gan_norm', None), btok)
    y._gan_bn = btok
    return y


# --------------------------------------------------------------------------------------------
# activations, add, concat, reflect pad
# --------------------------------------------------------------------------------------------
class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, param, scale, shift):
        x = _dense(x)
        y = torch.empty_like(x)
        check(lib.gan_act_fwd(ptr(x), ptr(y), x.numel(), act, param, scale, shift, stream()))
        ctx.save_for_backward(y)
        ctx.cfg = (act, param, scale, shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, param, scale, shift = ctx.cfg
        dy = _dense(dy)
        dx = torch.empty_like(y)
        check(lib.gan_act_bwd(ptr(y), ptr(dy), ptr(dx), y.numel(), act, param, scale, shift, stream()))
        return dx, None, None, None, None


def lrelu(x, leak=0.2, name="lrelu"):
    """ops.py:101-102: tf.maximum(x, leak*x)."""
    return _ActFn.apply(x, _lib.ACT_LRELU, float(leak), 1.0, 0.0)


def relu(x):
    return _ActFn.apply(x, _lib.ACT_RELU, 0.0, 1.0, 0.0)


def tanh(x, scale=1.0, shift=0.0):
    """tf.nn.tanh; scale/shift fold the `tanh(h4)/2. + 0.5` of text2image.py:198 into the same pass."""
    return _ActFn.apply(x, _lib.ACT_TANH, 0.0, float(scale), float(shift))


def sigmoid(x):
    return _ActFn.apply(x, _lib.ACT_SIGMOID, 0.0, 1.0, 0.0)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _dense(a), _dense(b)
        y = torch.empty_like(a)
        check(lib.gan_add_fwd(ptr(a), ptr(b), ptr(y), a.numel(), stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    if a.shape != b.shape:
        raise ValueError("add: shapes %s and %s differ" % (tuple(a.shape), tuple(b.shape)))
    return _AddFn.apply(a, b)


class _ForkFn(torch.autograd.Function):
    """One tensor handed to several consumers: the backward pass sums their gradients with the library's
    add kernel (left to itself, autograd would accumulate them with a torch operator)."""

    @staticmethod
    def forward(ctx, x, n, in_bn=None):
        ctx.in_bn = in_bn
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        grads = [_dense(g) for g in grads if g is not None]
        if not grads:
            return None, None, None
        total = grads[0]
        for i, g in enumerate(grads[1:]):
            out = torch.empty_like(total)
            # x came out of a batch norm: the launch that finishes its gradient also takes that norm's backward sums
            if not (i == len(grads) - 2 and ctx.in_bn is not None and ctx.in_bn.saved is not None and
                    _add_with_bn_sums(total, g, out, ctx.in_bn)):
                check(lib.gan_add_fwd(ptr(total), ptr(g), ptr(out), total.numel(), stream()))
            total = out
        return total, None, None


def fork(x, n=2):
    """`n` aliases of `x`, one per consumer (a residual connection, a generated image that feeds both a
    discriminator and the other generator).  Without a backward pass to run it returns `x` n times."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    return _ForkFn.apply(x, n, _bn_of(x))


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, rows_per_b):
        a, b = _dense(a), _dense(b)
        ca, cb = a.shape[-1], b.shape[-1]
        rows = a.numel() // ca
        out = torch.empty(tuple(a.shape[:-1]) + (ca + cb,), dtype=torch.float32, device=a.device)
        check(lib.gan_concat_fwd(ptr(a), ptr(b), ptr(out), rows, ca, cb, rows_per_b, stream()))
        ctx.cfg = (tuple(a.shape), tuple(b.shape), rows, ca, cb, rows_per_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        ashape, bshape, rows, ca, cb, rpb = ctx.cfg
        dout = _dense(dout)
        da = torch.empty(ashape, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(bshape, dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            check(lib.gan_concat_bwd(ptr(dout), ptr(da), ptr(db), rows, ca, cb, rpb, stream()))
        return da, db, None


def concat(tensors, axis, *args, **kwargs):
    """ops.py:23-28.  Last-axis concatenation (the only kind the hot path uses)."""
    tensors = list(tensors)
    nd = tensors[0].dim()
    if axis not in (nd - 1, -1):
        raise ValueError("concat: only the last axis is supported (got axis=%d for rank %d)" % (axis, nd))
    out = tensors[0]
    for t in tensors[1:]:
        if tuple(t.shape[:-1]) != tuple(out.shape[:-1]):
            raise ValueError("concat: leading shapes %s and %s differ" % (tuple(out.shape), tuple(t.shape)))
        out = _ConcatFn.apply(out, t, 1)
    return out


def conv_cond_concat(x, y):
    """ops.py:57-62: tile y [N,1,1,Y] over the feature map of x [N,H,W,C] and concat on channels."""
    n, h, w, _ = x.shape
    if y.dim() != 4 or y.shape[0] != n or y.shape[1] != 1 or y.shape[2] != 1:
        raise ValueError("conv_cond_concat: y must be [N,1,1,Y], got %s" % (tuple(y.shape),))
    return _ConcatFn.apply(x, y, int(h * w))


class _ReflectPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _dense(x)
        n, h, w, c = x.shape
        y = torch.empty((n, h + 2 * p, w + 2 * p, c), dtype=torch.float32, device=x.device)
        check(lib.gan_reflect_pad_fwd(ptr(x), ptr(y), n, h, w, c, p, stream()))
        ctx.cfg = (n, h, w, c, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, p = ctx.cfg
        dy = _dense(dy)
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
        check(lib.gan_reflect_pad_bwd(ptr(dy), ptr(dx), n, h, w, c, p, stream()))
        return dx, None


def reflect_pad(x, p):
    """tf.pad(x, [[0,0],[p,p],[p,p],[0,0]], "REFLECT") (cyclegan_module.py:106,108,124,143)."""
    if p == 0:
        return x
    return _ReflectPadFn.apply(x, int(p))


# --------------------------------------------------------------------------------------------
# losses.  Each is fused with its gradient and is a ROOT of the backward pass: the value handed
# to backward() is taken to be 1 and the loss's weight in the objective is the `weight` argument.
# --------------------------------------------------------------------------------------------
_L1_MANY_BLOCKS = 32768   # values from which the L1 term takes the multi-block kernels


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, x, other, target, weight):
        x = _dense(x)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        need = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        grad = torch.empty_like(x) if need else None
        n = x.numel()
        if kind == 'bce':
            check(lib.gan_sigmoid_bce_mean(ptr(x), n, target, weight, 0, ptr(loss), ptr(grad), stream()))
        elif kind == 'mse':
            check(lib.gan_mse_const_mean(ptr(x), n, target, weight, 0, ptr(loss), ptr(grad), stream()))
        else:   # l1: x = a (reference side), other = b; the kernel emits d/db
            other = _dense(other)
            if n >= _L1_MANY_BLOCKS:   # image-sized: spread over the chip, block sums through the workspace
                wp, wn = _lib.workspace(int(lib.gan_l1_mean_workspace_bytes(n)))
                check(lib.gan_l1_mean_ws(ptr(x), ptr(other), n, weight, 0, ptr(loss), ptr(grad), wp, wn, stream()))
            else:
                check(lib.gan_l1_mean(ptr(x), ptr(other), n, weight, 0, ptr(loss), ptr(grad), stream()))
        ctx.kind = kind
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, _one):
        g = ctx.grad
        if ctx.kind == 'l1':
            ga = None
            if ctx.needs_input_grad[1]:
                ga = torch.empty_like(g)
                check(lib.gan_act_fwd(ptr(g), ptr(ga), g.numel(), _lib.ACT_NONE, 0.0, -1.0, 0.0, stream()))
            return None, ga, (g if ctx.needs_input_grad[2] else None), None, None
        return None, g, None, None, None


def sigmoid_cross_entropy_mean(logits, label, weight=1.0):
    """tf.reduce_mean(tf.nn.sigmoid_cross_entropy_with_logits(logits, labels=const)) * weight
    (dcgan.py:123-134, text2image.py:86-90, cyclegan_module.py:155-156)."""
    return _LossFn.apply('bce', logits, None, float(label), float(weight))


class _StackedBceFn(torch.autograd.Function):
    """One sigmoid cross-entropy mean per part of a stack of equally sized batches of logits."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        x = _dense(logits)
        k = len(labels)
        n = x.numel() // k
        losses = [torch.empty(1, dtype=torch.float32, device=x.device) for _ in range(k)]
        grad = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gp = ptr(grad)
        for i, label in enumerate(labels):
            check(lib.gan_sigmoid_bce_mean(ptr(x) + 4 * n * i, n, label, weight, 0, ptr(losses[i]),
                                           None if gp is None else gp + 4 * n * i, stream()))
        ctx.grad = grad
        return tuple(losses)

    @staticmethod
    def backward(ctx, *_ones):
        return ctx.grad, None, None


def stacked_sigmoid_cross_entropy_means(logits, labels, weight=1.0):
    """Mean cross entropy of each of the len(labels) equal parts of `logits` (batches stacked along
    axis 0) against its constant label; every returned loss is a root for backward()."""
    labels = tuple(float(v) for v in labels)
    if logits.shape[0] % len(labels):
        raise ValueError("stacked logits: batch of %d does not split into %d parts" % (logits.shape[0], len(labels)))
    return _StackedBceFn.apply(logits, labels, float(weight))


def stack_batches(*parts):
    """[a; b; ...] along axis 0 in one new buffer (one device copy per part)."""
    a = parts[0]
    for b in parts[1:]:
        if a.shape != b.shape:
            raise ValueError("stack_batches: shapes %s and %s differ" % (tuple(a.shape), tuple(b.shape)))
    n = a.shape[0]
    out = torch.empty((len(parts) * n,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    for i, t in enumerate(parts):
        out[i * n:(i + 1) * n].copy_(t.detach())
    return out


def mse_const_mean(x, target, weight=1.0):
    """tf.reduce_mean((x - const)**2) * weight: the reference's `mae_criterion` against
    ones_like / zeros_like (cyclegan_module.py:152-153)."""
    return _LossFn.apply('mse', x, None, float(target), float(weight))


def l1_mean(a, b, weight=1.0):
    """tf.reduce_mean(tf.abs(a - b)) * weight (abs_criterion, cyclegan_module.py:149-150)."""
    if a.shape != b.shape:
        raise ValueError("l1_mean: shapes %s and %s differ" % (tuple(a.shape), tuple(b.shape)))
    return _LossFn.apply('l1', a, b, 0.0, float(weight))


_ones = {}


def backward(losses, boundary=None, at_boundary=None):
    """Run the backward pass from one or several loss roots (their weights are already applied).

    Staged form (data parallel overlap): `boundary` lists the activations that separate the layers
    owning the EARLY variables of the group being trained (Group.split_early) from the rest.  The
    pass first runs every layer that leads to those activations or owns an early variable, then
    calls at_boundary() -- on the calling thread, so it may cut a graph capture and start the
    exchange of the finished gradients -- and then continues from the boundary downwards.  Every
    layer that owns a late variable must lie below the boundary."""
    if torch.is_tensor(losses):
        losses = [losses]
    dev = losses[0].device
    one = _ones.get(dev)
    if one is None:
        one = _ones[dev] = torch.ones(1, dtype=torch.float32, device=dev)
    roots, seeds = list(losses), [one] * len(losses)
    if boundary:
        boundary = list(boundary)
        for t in boundary:
            if not t.requires_grad:
                raise ValueError("backward: a boundary activation does not take part in the backward pass")
        early_token = _token.get((dev, True))
        wanted = boundary + ([early_token] if early_token is not None else [])
        grads = torch.autograd.grad(roots, wanted, seeds, allow_unused=True)[:len(boundary)]
        if any(g is None for g in grads):
            raise ValueError("backward: a boundary activation is not reachable from the losses")
        if at_boundary is not None:
            at_boundary()
        roots, seeds = boundary, list(grads)
    torch.autograd.backward(roots, seeds)
