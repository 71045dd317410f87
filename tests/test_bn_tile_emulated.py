"""The two-launch batch-norm kernels of gan_amd/csrc/norm.hip -- and their variants that sum a contraction's split-K
slabs on the way -- compiled from the library's own source for the HOST (tests/hip_emul: one OS thread per work-item,
one workgroup at a time) and run under AddressSanitizer on buffers of exactly the documented sizes.  Checks what no GPU
run can show safely: that no work-item of these kernels reads or writes outside its tensors, for ragged row blocks,
several channel tiles and stacked groups.  No GPU, no product code path: the harness only launches kernels."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL = os.path.join(HERE, "hip_emul")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_batch_norm_tile_kernels_stay_inside_their_buffers(tmp_path):
    exe = str(tmp_path / "bn_tile_harness")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-x", "c++",
                            "-I", EMUL, os.path.join(EMUL, "bn_tile_harness.cpp"), "-o", exe, "-lpthread"],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-4000:]
    # (a preloaded library in front of the sanitizer's run time is fine: only its link-order check objects)
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, (run.stdout[-2000:] + run.stderr[-4000:])
    assert "ok: 0 failure(s)" in run.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_tap_kernels_against_direct_convolution(tmp_path):
    """gan_amd/csrc/taps.hip (few-channel convolutions as a dense product plus a sum over taps) compiled for the host
    and run under AddressSanitizer: forward, filter / bias / input gradient of a thin-output layer and the input
    gradient of a thin-input layer against direct convolution loops in double precision, mirrored and zero padding,
    images down to 4 x 4 under a 7 x 7 filter."""
    exe = str(tmp_path / "taps_harness")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-x", "c++",
                            "-I", EMUL, os.path.join(EMUL, "taps_harness.cpp"), "-o", exe, "-lpthread"],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-4000:]
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0, (run.stdout[-2000:] + run.stderr[-4000:])
    assert "ok: 0 failure(s)" in run.stdout
