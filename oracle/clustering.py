"""Oracle: the super-cluster stage of the segmenter (``segmentation/segmenter.py:69-86``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  open3d is not installed, so its two routines are restated from their
published behaviour (PARITY UNPINNED): ``voxel_down_sample`` (voxel index = floor((p - (min_bound - voxel/2)) / voxel),
mean of the points of a voxel) and ``cluster_dbscan`` -- for which scikit-learn's ``DBSCAN`` (same definitions: a core
point has at least ``min_points`` points, itself included, within ``eps``) supplies an independent implementation.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    pts = np.asarray(points, dtype=np.float64)
    vmin = pts.min(axis=0) - voxel_size * 0.5
    idx = np.floor((pts - vmin) / voxel_size).astype(np.int64)
    _, inv = np.unique(idx, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    out = np.zeros((inv.max() + 1, 3))
    np.add.at(out, inv, pts)
    return out / np.bincount(inv)[:, None]


def dbscan(points: np.ndarray, eps: float, min_points: int) -> Tuple[np.ndarray, np.ndarray]:
    """labels [N] (-1 noise) and core mask [N] from scikit-learn."""
    from sklearn.cluster import DBSCAN

    m = DBSCAN(eps=eps, min_samples=min_points).fit(np.asarray(points, dtype=np.float64))
    core = np.zeros(len(points), dtype=bool)
    core[m.core_sample_indices_] = True
    return m.labels_, core
