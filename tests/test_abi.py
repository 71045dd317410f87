"""The C-ABI library loads on a CPU-only host and exports every symbol include/gan_hip.h declares
(no compute calls are made here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gan_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gan_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), "libgan_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes table has no signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib.gan_abi_version() == 1


def test_host_side_argument_checks_need_no_gpu():
    import ctypes
    from gan_amd import _lib
    g = _lib.conv_geom(1, 8, 8, 4, 4, 4, 4, 5, 5, 3, 2, 1, 1)     # stride 3 is rejected before any launch
    assert _lib.lib.gan_conv_workspace_bytes(ctypes.byref(g)) == 0
    assert b"stride" in _lib.lib.gan_last_error()
    assert _lib.lib.gan_bn_workspace_bytes(1, 64 * 16 * 16, 128) > 0
    assert _lib.lib.gan_linear_workspace_bytes(64, 100, 8192) >= 0


def test_reflect_input_gradient_refuses_geometries_its_fold_is_not_the_adjoint_of():
    """gan_conv2d_dgrad_reflect folds about a border of pt pixels on every side: only a filter that covers the
    reflect-padded image exactly (tf.pad by p + VALID, cyclegan_module.py:102-110) makes that the adjoint of
    gan_conv2d_fwd_reflect.  Other geometries are refused before any launch (fake non-null pointers, no GPU)."""
    import ctypes
    from gan_amd import _lib as L
    fake = ctypes.c_void_p(256)
    ok = L.conv_geom(1, 16, 16, 8, 16, 16, 8, 3, 3, 1, 1, 1, 1)       # 3x3 stride 1, pad 1 each side: 14 + 3 == 16 + 2 ... exact
    assert (ok.hs - 1) * ok.sh + ok.kh == ok.hb + 2 * ok.pt
    assert L.lib.gan_conv2d_dgrad_reflect_workspace_bytes(ctypes.byref(ok)) > 0
    for bad in (L.conv_geom(1, 16, 16, 8, 8, 8, 8, 5, 5, 2, 2, 1, 1),     # TF SAME split (1, 2): one more pixel behind
                L.conv_geom(1, 16, 16, 8, 8, 8, 8, 3, 3, 2, 2, 0, 0),     # TF SAME split (0, 1)
                L.conv_geom(1, 16, 16, 8, 15, 15, 8, 3, 3, 1, 1, 1, 1)):  # small side one short of covering the padding
        rc = L.lib.gan_conv2d_dgrad_reflect(ctypes.byref(bad), fake, fake, fake, fake, 1 << 30, None)
        assert rc != 0 and b"cover the reflect-padded image exactly" in L.lib.gan_last_error(), L.lib.gan_last_error()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_plan_pinning_is_host_side_bookkeeping():
    """gan_conv_plan_set / _get / _clear need no GPU: a pinned plan is what the planner reports, the
    workspace query follows it, and removing the pin restores the heuristic."""
    import ctypes
    from gan_amd import _lib as L
    g = L.conv_geom(64, 16, 16, 128, 8, 8, 256, 5, 5, 2, 2, 1, 1)
    gp = ctypes.byref(g)
    c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()

    def get(mode):
        L.check(L.lib.gan_conv_plan_get(mode, gp, ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
        return c.value, s.value, thin.value
    L.lib.gan_conv_plan_clear()
    heur = get(0)
    ws0 = L.lib.gan_conv_workspace_bytes(gp)
    L.check(L.lib.gan_conv_plan_set(0, gp, 2, 1))
    L.check(L.lib.gan_conv_plan_set(1, gp, 2, 1))
    L.check(L.lib.gan_conv_plan_set(2, gp, 2, 1))
    assert get(0) == (2, 1, 0) and get(1)[:2] == (2, 1)
    assert L.lib.gan_conv_workspace_bytes(gp) < ws0 or ws0 == 0     # splits == 1 needs no slabs
    L.check(L.lib.gan_conv_plan_set(0, gp, 0, 7))
    assert get(0)[0] == 0 and 1 < get(0)[1] <= 7       # the planner may round the split count down
    L.check(L.lib.gan_conv_plan_set(0, gp, -1, 0))
    assert get(0) == heur
    assert L.lib.gan_conv_plan_set(3, gp, 0, 1) != 0 and b"mode" in L.lib.gan_last_error()
    assert L.lib.gan_conv_plan_set(0, gp, 10, 1) != 0
    L.check(L.lib.gan_conv_plan_set(0, gp, -1, 0))
    # configurations 4..9 are the split-bf16 kernels (7, 8: split-K inside the workgroup; 9: 128x32 tile): the heuristic takes them for channel counts that are
    # multiples of 4, a pin is honoured, and the exact-fp32 switch (or an odd channel count) maps them back
    assert heur[0] >= 4 and L.lib.gan_get_exact_f32() == 0
    L.check(L.lib.gan_conv_plan_set(0, gp, 5, 2))
    assert get(0)[:2] == (5, 2)
    L.lib.gan_set_exact_f32(1)
    assert get(0)[:2] == (1, 2) and L.lib.gan_get_exact_f32() == 1
    L.lib.gan_set_exact_f32(0)
    L.check(L.lib.gan_conv_plan_set(0, gp, 8, 2))
    assert get(0)[:2] == (8, 2)
    # cfg may carry a tile order (16 x order, 1..3); it is reported back in the same encoding
    L.check(L.lib.gan_conv_plan_set(0, gp, 8 + 16 * 2, 2))
    assert get(0)[:2] == (8 + 32, 2)
    assert L.lib.gan_conv_plan_set(0, gp, 10 + 16, 1) != 0 and L.lib.gan_conv_plan_set(0, gp, 5 + 80, 1) != 0
    assert L.lib.gan_conv_plan_set(1, gp, 9, 1) != 0 and b"modes 0 and 2" in L.lib.gan_last_error()
    L.check(L.lib.gan_conv_plan_set(2, gp, 9, 2))
    L.check(L.lib.gan_conv_plan_set(2, gp, -1, 0))
    L.check(L.lib.gan_conv_plan_set(0, gp, 8, 2))
    L.lib.gan_set_exact_f32(1)
    assert get(0)[:2] == (1, 4)        # two k groups per workgroup fall back to one: twice the slabs
    L.lib.gan_set_exact_f32(0)
    L.check(L.lib.gan_conv_plan_set(0, gp, -1, 0))
    godd = L.conv_geom(8, 16, 16, 27, 8, 8, 10, 5, 5, 2, 2, 1, 1)
    # a geometry that cannot take the split kernels (gathered-image channel count no multiple of 4 and filter rows of
    # 5 x 27 = 135 floats, longer than the filter-row gather serves): the pin is refused, so that what is pinned is
    # exactly what runs
    assert L.lib.gan_conv_plan_set(0, ctypes.byref(godd), 6, 1) != 0 and b"multiple of 4" in L.lib.gan_last_error()
    # ... while 6 channels (rows of 30 floats) take them through the filter-row gather; the transposed mode needs an even count
    gsix = L.conv_geom(8, 16, 16, 6, 8, 8, 9, 5, 5, 2, 2, 1, 1)
    L.check(L.lib.gan_conv_plan_set(0, ctypes.byref(gsix), 6, 1))
    L.check(L.lib.gan_conv_plan_set(0, ctypes.byref(gsix), -1, 0))
    assert L.lib.gan_conv_plan_set(1, ctypes.byref(gsix), 6, 1) != 0
    L.check(L.lib.gan_conv_plan_set(0, ctypes.byref(godd), 2, 1))
    L.check(L.lib.gan_conv_plan_get(0, ctypes.byref(godd), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
    assert c.value == 2
    # thin outputs of the transposed contraction have no plan
    g3 = L.conv_geom(64, 64, 64, 3, 32, 32, 64, 5, 5, 2, 2, 1, 1)
    L.check(L.lib.gan_conv_plan_get(1, ctypes.byref(g3), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
    assert thin.value == 1
    lg = L.ConvGeom()
    L.check(L.lib.gan_linear_as_conv_geom(64, 100, 8192, ctypes.byref(lg)))
    assert (lg.n, lg.cb, lg.cs, lg.kh, lg.hb) == (64, 100, 8192, 1, 1)
    L.lib.gan_conv_plan_clear()
    # knobs: initial values from the environment once, changed through the setter only
    assert L.lib.gan_knob_set(b"no_such_knob", 1) != 0 and b"unknown knob" in L.lib.gan_last_error()
    L.knob_set("force_cfg", 2)
    assert get(0)[0] == 2
    L.knob_set("force_cfg", -1)
    assert get(0) == heur
