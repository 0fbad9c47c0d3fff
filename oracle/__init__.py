"""CPU oracle for the CropNeRF fruit_nerf volumetric ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The shipped path (``cropnerf_amd``) never imports this package and
fails loudly when the HIP library is missing.

What it is: an op-for-op PyTorch fp32 restatement (materialised ``[R,S,.]``
tensors, unfused ops, same operation order) of the arithmetic the reference
executes on the path

    ray generation -> sampling -> hash-grid + tiny-MLP field -> alpha compositing
    -> point-cloud / semantic-projection export masks

The reference (``/root/reference/crop_nerf``) holds only the *wiring* of that
path; the arithmetic lives in ``nerfstudio==1.1.3`` (pinned only through
``crop_nerf/Dockerfile:1``), which is absent from this image.  Every function
cites the reference call site (file:line under ``crop_nerf/``) it follows and
names the upstream routine it restates.

PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this
path and cannot be imported here (``nerfstudio``,
``jaxtyping``, ``open3d`` ... are ordinary ``ModuleNotFoundError``s; no
permission denial occurred).  The oracle is therefore pinned only by the
analytic known-answer tests of SURVEY.md section 8(c) (``tests/test_oracle_kat.py``).
Exception: ``oracle/zbuffer.py`` (the depth-based projection, a "next" row) follows numpy code that IS in the
reference (``scripts/depth_based_semantic_projection.py:31-105``) statement by statement.
``oracle/outliers.py`` and ``oracle/clustering.py`` (the segmenter's super-cluster stage) restate open3d's published
algorithms with scipy / scikit-learn -- open3d is absent, so they too are unpinned.
The merger's label propagation needs no oracle: the reference's own ``segmentation/lpa.py`` imports here and produced
``tests/golden/merger_small.npz`` (``tests/golden/make_golden_merger.py``).
"""

from . import field, model, rays, render, samplers  # noqa: F401
