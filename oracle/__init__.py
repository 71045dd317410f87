"""CPU oracle for the GAN training hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain numpy, the arithmetic that the reference
(Youngjung/GAN) delegates to TensorFlow 1.x: the layer primitives of
``models/ops.py:30-115`` and the generator / discriminator steps composed from
them in ``models/dcgan.py``, ``models/text2image.py`` and
``models/cyclegan_module.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  Nothing under
``gan_amd/`` imports it: the product path is the HIP library and fails loudly
when that library is missing.

PARITY UNPINNED at the TensorFlow boundary: the reference ships no tests, no
golden vectors and no fixtures, and TensorFlow 1.x (un-vendored, un-pinned) is
not installable here, so no output of the reference itself exists to pin this
restatement against.  It is pinned instead by (i) hand-computed known-answer
cases, (ii) float64 finite-difference checks of every backward, (iii) an
independent cross-check against torch CPU ops, and (iv) algebraic identities
(conv2d_transpose == input-gradient of conv2d; batch norm at N=1 == instance
norm).  See tests/test_oracle_*.py.
"""
