"""Helpers shared by the GPU parity tests: device round trips and the tolerance rule."""
import ctypes
import os

import numpy as np
import torch

# Parity bar for floating point (north_star: "per-op fp32 outputs and gradients within rel 1e-4"):
# the largest absolute deviation from the float64 oracle, relative to the largest magnitude of the
# oracle tensor, must not exceed REL_TOL.
REL_TOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_close(got, ref, what="", tol=REL_TOL):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs oracle %s" % (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % what
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(got - ref).max()) / scale
    assert err <= tol, "%s: max |gpu - oracle| / max |oracle| = %.3e > %.1e" % (what, err, tol)
    return err


class forced(object):
    """Force a tile configuration / split-K factor of the contraction kernels for one block."""

    def __init__(self, cfg=None, splits=None):
        self.env = {"GAN_HIP_FORCE_CFG": cfg, "GAN_HIP_FORCE_SPLITS": splits}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        for k, v in self.env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def geom_for(n, h, w, cb, cs, k, s, padding):
    from gan_amd import ops
    return ops._conv_geom(n, h, w, cb, cs, k, k, s, s, padding)


def byref(g):
    return ctypes.byref(g)
