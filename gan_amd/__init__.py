"""gan_amd: the GAN training hot path of Youngjung/GAN on MI355X (gfx950).

`gan_amd.ops` mirrors the reference's models/ops.py; the model modules mirror models/dcgan.py,
models/text2image.py, models/cyclegan_module.py and models/cyclegan.py.  All arithmetic runs in
libgan_hip.so (hand-written HIP kernels); importing this package without that library fails.
"""
from . import _lib  # noqa: F401  (raises ImportError when libgan_hip.so is missing)

__all__ = ["ops", "variables", "optim", "dcgan", "text2image", "cyclegan_module", "cyclegan"]
