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


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
