"""gan_conv2d_fwd_static_ws: the conv2d forward whose split-K slabs live in a device array built into the
library, for callers that cannot allocate.  Checked through the C ABI against gan_conv2d_fwd (same kernels,
same plan: bit-identical) on two calls in a row, the second on a smaller input: nothing clears the array, so the
second call finds the first call's slabs in it (and, in the first test, those of a call forced to more slabs than
any later one reads).  A last test captures the entry point into a HIP graph after an eager warm-up call, which is
the use it exists for, and replays it on new inputs."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, case, seed):
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = case
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    torch.manual_seed(seed)
    x = torch.randn(n, h, w, cb, device='cuda')
    wt = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    b = torch.randn(cs, device='cuda')
    c, sp, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.check(L.lib.gan_conv_plan_get(0, gp, ctypes.byref(c), ctypes.byref(sp), ctypes.byref(thin)))
    assert sp.value > 1, "the case must use split-K slabs, or the scratch array is never touched"
    y_ref = torch.empty(n, g.hs, g.ws, cs, device='cuda')
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(y_ref), wp, wn, L.stream()))
    y = torch.full_like(y_ref, float('nan'))
    L.check(L.lib.gan_conv2d_fwd_static_ws(gp, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(y), L.stream()))
    torch.cuda.synchronize()
    return y, y_ref


def test_static_workspace_conv_matches_the_workspace_entry_point_twice_in_a_row():
    from gan_amd import _lib as L
    L.knob_set("force_splits", 12)                              # dirties 12 slabs x 4 MB of the array first
    try:
        y, y_ref = _both(L, (64, 16, 16, 128, 256, 5, 2), 7)
    finally:
        L.knob_set("force_splits", 0)
    assert torch.equal(y, y_ref)
    y, y_ref = _both(L, (64, 16, 16, 128, 256, 5, 2), 1)        # DCGAN-64 d_h2: 4096 x 256 output, several slabs
    assert torch.isfinite(y).all() and torch.equal(y, y_ref)
    y, y_ref = _both(L, (8, 16, 16, 128, 256, 5, 2), 2)         # smaller input: the array still holds the first call's slabs
    assert torch.isfinite(y).all() and torch.equal(y, y_ref)
    y, y_ref = _both(L, (3, 14, 16, 32, 80, 5, 2), 3)           # ragged tiles
    assert torch.equal(y, y_ref)


def test_static_workspace_conv_refuses_what_does_not_fit():
    from gan_amd import _lib as L
    from gpu_util import geom_for
    g = geom_for(64, 32, 32, 64, 128, 5, 2, 'SAME')             # 8.4 MB output
    L.knob_set("force_splits", 16)                              # 16 slabs = 134 MB > 64 MiB
    try:
        x = torch.zeros(64, 32, 32, 64, device='cuda')
        wt = torch.zeros(5, 5, 64, 128, device='cuda')
        y = torch.zeros(64, 16, 16, 128, device='cuda')
        rc = L.lib.gan_conv2d_fwd_static_ws(ctypes.byref(g), L.ptr(x), L.ptr(wt), None, L.ptr(y), L.stream())
        assert rc != 0 and b"built-in scratch array" in L.lib.gan_last_error()
    finally:
        L.knob_set("force_splits", 0)


def test_static_workspace_conv_under_graph_capture():
    """The stated use: a caller under HIP graph capture cannot allocate.  One eager call first (it looks the array's
    address up), then the call is captured on torch's capture stream and the graph replayed on new input values."""
    from gan_amd import _lib as L
    from gpu_util import geom_for
    n, h, w, cb, cs, k, s = 64, 16, 16, 128, 256, 5, 2
    g = geom_for(n, h, w, cb, cs, k, s, 'SAME')
    gp = ctypes.byref(g)
    torch.manual_seed(11)
    x = torch.randn(n, h, w, cb, device='cuda')
    wt = torch.randn(k, k, cb, cs, device='cuda') * 0.05
    b = torch.randn(cs, device='cuda')
    y = torch.empty(n, g.hs, g.ws, cs, device='cuda')
    L.check(L.lib.gan_conv2d_fwd_static_ws(gp, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(y), L.stream()))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        L.check(L.lib.gan_conv2d_fwd_static_ws(gp, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(y), L.stream()))
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    for seed in (12, 13):
        torch.manual_seed(seed)
        x.copy_(torch.randn_like(x))
        y.fill_(float('nan'))
        graph.replay()
        y_ref = torch.empty_like(y)
        L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(x), L.ptr(wt), L.ptr(b), L.ptr(y_ref), wp, wn, L.stream()))
        torch.cuda.synchronize()
        assert torch.isfinite(y).all() and torch.equal(y, y_ref)
