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

PARITY UNPINNED for the ray-marching arithmetic: the reference has no tests, golden vectors or fixtures for this path
 and cannot be imported here (``nerfstudio``, ``jaxtyping``, ``open3d`` ... are ordinary
``ModuleNotFoundError``s; no permission denial occurred).  That arithmetic lives in third-party packages (nerfstudio 1.1.3,
tiny-cuda-nn) that are absent; its restatements -- ``field.py`` / ``samplers.py`` / ``render.py`` / ``rays.py`` for nerfstudio's
torch modules, ``tcnn.py`` for tiny-cuda-nn's published grid / MLP / SH semantics, the reference's DEFAULT implementation --
are pinned only by analytic known-answer tests (``tests/test_oracle_kat.py``, ``tests/test_oracle_tcnn.py``).

PINNED by the reference's own code where that code can run: ``tests/golden/make_golden_reference.py`` extracts (``ast``) the
definitions of the reference's pure numpy / torch / networkx functions next to the path, executes them unchanged in the
build container and stores inputs / outputs in ``tests/golden/reference_functions.npz``: ``get_projection_mat``,
``get_projection``, ``update_buffer`` (``scripts/depth_based_semantic_projection.py:31-49,84-105``) pin ``zbuffer.py``;
``get_corners_of_aabb``, ``sample_surface_points`` (``data/fruit_datamanager.py:42-121``) pin ``rays.corners_of_aabb`` /
``rays.surface_points``; ``calc_affinity`` / ``get_component`` (``segmentation/merger.py:26-74,335-355``) and
``segmentation/lpa.py`` (imports as is; ``tests/golden/merger_small.npz``) pin the merger's graph stage
(``tests/test_reference_golden.py``, ``tests/test_merger.py``).  ``outliers.py`` and ``clustering.py`` restate open3d's
published algorithms with scipy / scikit-learn, ``contours.py`` OpenCV's ``findContours`` / ``contourArea`` / ``boundingRect`` /
``drawContours`` for the merger's image stage -- open3d and OpenCV are absent, so they are unpinned (known-answer tests).
"""

from . import field, model, rays, render, samplers  # noqa: F401
