"""Oracle: open3d ``PointCloud.estimate_normals()`` and the re-orientation that follows it in the point-cloud exporter
(``fruit_nerf/export/exporter_utils_nerfacto.py:203-225``; ``README.md:125``: ``--normal-method open3d``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  open3d is not installed here, so this restates its published
algorithm (``geometry/EstimateNormals.cpp``, open3d 0.17/0.18) at the defaults the reference uses --
``KDTreeSearchParamKNN(knn=30)``: per point the 30 nearest points (itself included, scipy's exact search), their
covariance from the nine cumulants in float64, the unit eigenvector of the smallest eigenvalue; fewer than three
neighbours or a zero covariance -> (0, 0, 1).  The eigenvector comes from ``numpy.linalg.eigh`` -- an independent
method from the closed-form solver the kernel (and open3d) uses, which is the point of a checker: the two agree up to
SIGN, which neither fixes (open3d's own sign is whatever its solver leaves; the exporter settles it afterwards against the
view directions).  PARITY UNPINNED like the rest of the oracle's open3d restatements (no open3d to run).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np
from scipy.spatial import cKDTree


def estimate_normals(points: np.ndarray, knn: int = 30) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(normals [N,3] float64, degenerate [N] bool, gap [N]): ``gap`` = (second smallest - smallest eigenvalue) / largest,
    the conditioning of the normal (0 for a line-like neighbourhood, where any vector across the line is as good)."""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    normals = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    degenerate = np.ones(n, dtype=bool)
    gap = np.zeros(n)
    k = min(knn, n)
    if k < 3:
        return normals, degenerate, gap
    _, idx = cKDTree(pts).query(pts, k=k)
    nb = pts[idx]  # [N,k,3]
    mean = nb.mean(axis=1)
    second = np.einsum("nki,nkj->nij", nb, nb) / k
    cov = second - mean[:, :, None] * mean[:, None, :]  # the cumulant form open3d uses
    w, v = np.linalg.eigh(cov)  # ascending
    ok = np.abs(cov).max(axis=(1, 2)) > 0
    normals[ok] = v[ok, :, 0]
    degenerate = ~ok
    gap[ok] = (w[ok, 1] - w[ok, 0]) / np.maximum(w[ok, 2], 1e-300)
    return normals, degenerate, gap


def reorient_normals(normals: np.ndarray, view_directions: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """``:219-225``: float32 normals, flipped where ``sum(view_direction * normal) > 0``; back to float64."""
    nf = np.asarray(normals, dtype=np.float32).copy()
    mask = np.sum(np.asarray(view_directions, dtype=np.float32) * nf, axis=-1) > 0
    nf[mask] *= -1
    return nf.astype(np.float64), mask
