"""CPU oracle for the CycleGAN -> MultiResUNet training hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the reported CPU
baseline.  The product path (``automatic-sem-image-segmentation_amd``) never
imports this package and fails loudly when the HIP library is missing.

What it is: a plain-torch (CPU, fp32 or fp64) restatement of the arithmetic the
reference delegates to Keras 3.5 on the torch backend, following

* topology: ``Releases/Version 1.2.0/CycleGAN.py:323-451,482-506`` and
  ``UNet_Segmentation.py:401-562``;
* train steps: ``CycleGAN.py:615-710`` and Keras' default ``train_step`` as
  invoked at ``UNet_Segmentation.py:277-283``;
* losses: ``CycleGAN.py:301-308``, ``UNet_Segmentation.py:379-384``;
* image buffer: ``CycleGAN.py:908-964``;
* Keras-internal semantics (conv padding, transposed-conv alignment,
  GroupNormalization/BatchNormalization formulas, Adam): SURVEY.md section 8
  "K-list".  keras~=3.5.0 / torch~=2.3.0 are pinned in
  ``Releases/Version 1.2.0/requirements.txt:1-4`` but are NOT vendored in
  /root/reference and are not installed here.

Pinning status
--------------
* Control flow of the CycleGAN step and of ``ImagePool`` is PINNED: the
  reference's own ``CycleGanModel.train_step_torch`` / ``ImagePool`` were
  executed in the build container under a test-only ``keras`` stub
  (``tests/golden/make_goldens.py``) and the resulting vectors are committed
  under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks the oracle
  against them.
* The Keras-internal layer arithmetic is "parity unpinned": the reference has
  no tests / golden vectors / shipped weights, and Keras cannot be imported
  here.  The formulas are restated from the published Keras 3.5 behaviour and
  isolated in ``oracle/ops.py`` so they can be corrected in one place.
"""

from . import ops, nets, steps  # noqa: F401
