"""Helpers shared by the GPU parity tests: device round trips and the tolerance rule."""
import ctypes
import os

import numpy as np
import torch

# Parity bar for floating point (north_star: "per-op fp32 outputs and gradients within rel 1e-4"), two
# conditions on the deviation from the float64 oracle:
#   max-norm:     max |gpu - ref| <= REL_TOL * max |ref|
#   elementwise:  |gpu_i - ref_i| <= REL_TOL * |ref_i| + ELEM_FLOOR * REL_TOL * max |ref|   for every i
# The second one holds every element to 1e-4 of ITSELF; its absolute term (1e-6 of the tensor's largest
# magnitude at the default tolerance) is what float32 accumulation of a long, cancelling sum leaves on an
# element whose exact value is near zero.
REL_TOL = 1e-4
ELEM_FLOOR = 1e-2


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_close(got, ref, what="", tol=REL_TOL, elementwise=True):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs oracle %s" % (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    scale = max(float(np.abs(ref).max()), 1e-30)
    diff = np.abs(got - ref)
    err = float(diff.max()) / scale
    assert err <= tol, "%s: max |gpu - oracle| / max |oracle| = %.3e > %.1e" % (what, err, tol)
    if elementwise:
        excess = diff - (tol * np.abs(ref) + ELEM_FLOOR * tol * scale)
        worst = int(np.argmax(excess))
        assert excess.flat[worst] <= 0, (
            "%s: element %d: |gpu - oracle| = %.3e exceeds %.1e * |oracle| (%.3e) + %.1e * max |oracle| (%.3e)"
            % (what, worst, diff.flat[worst], tol, abs(ref.flat[worst]), ELEM_FLOOR * tol, scale))
    return err


def check_group_grads(group, ref_grads, tol=3e-4):
    """Compare every gradient of an optimiser group with the oracle's.

    A bias that feeds a batch norm has an exactly-zero true gradient (the normalisation removes any
    per-channel shift), so both sides hold only the rounding noise of a long sum.  Such a gradient is
    recognised by being tiny next to the gradient of the weight it belongs to, and is then only
    required to be tiny on the GPU side too.

    Gradients that have passed through a whole network are held to the max-norm bar only: an element of,
    say, a first-layer filter gradient is a sum over 10^5 .. 10^6 cancelling terms, so its own magnitude
    says nothing about the rounding error it carries."""
    got_all = {v.name: host(v.grad) for v in group.variables}
    for name, got in got_all.items():
        ref = ref_grads[name]
        layer, leaf = name.rsplit('/', 1)
        sibling = layer + ('/w' if leaf == 'biases' else '/Matrix')
        if leaf in ('biases', 'bias') and sibling in ref_grads:
            wscale = float(np.abs(ref_grads[sibling]).max())
            if float(np.abs(ref).max()) < 2e-3 * wscale:
                assert float(np.abs(got).max()) < 2e-3 * wscale, "grad %s should vanish" % name
                continue
        assert_close(got, ref, "grad " + name, tol=tol, elementwise=False)


def is_noise_grad(name, ref_grads):
    layer, leaf = name.rsplit('/', 1)
    sibling = layer + ('/w' if leaf == 'biases' else '/Matrix')
    if leaf in ('biases', 'bias') and sibling in ref_grads:
        return float(np.abs(ref_grads[name]).max()) < 2e-3 * float(np.abs(ref_grads[sibling]).max())
    return False


class forced(object):
    """Force a tile configuration / split-K factor of the contraction kernels for one block."""

    def __init__(self, cfg=None, splits=None):
        self.cfg, self.splits = cfg, splits

    def __enter__(self):
        from gan_amd._lib import knob_set
        knob_set("force_cfg", -1 if self.cfg is None else int(self.cfg))
        knob_set("force_splits", 0 if self.splits is None else int(self.splits))

    def __exit__(self, *exc):
        from gan_amd._lib import knob_set
        knob_set("force_cfg", -1)
        knob_set("force_splits", 0)


class knob(object):
    """Turn one boolean library knob on for one block."""

    def __init__(self, name, on=True):
        self.name, self.on = name, on

    def __enter__(self):
        from gan_amd._lib import knob_set
        knob_set(self.name, 1 if self.on else 0)

    def __exit__(self, *exc):
        from gan_amd._lib import knob_set
        knob_set(self.name, 0)


def geom_for(n, h, w, cb, cs, k, s, padding):
    from gan_amd import ops
    return ops._conv_geom(n, h, w, cb, cs, k, k, s, s, padding)


def byref(g):
    return ctypes.byref(g)
