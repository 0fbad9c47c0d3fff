"""Oracle: open3d ``remove_statistical_outlier`` as the point-cloud exporter calls it
(``fruit_nerf/export/exporter_utils_nerfacto.py:194-199``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  open3d is not installed here, so this restates its published
algorithm (``PointCloud::RemoveStatisticalOutliers``, open3d 0.17/0.18) with an exact k-nearest-neighbour search from
scipy: per point the mean of the distances to its ``nb_neighbors`` nearest points -- the KD-tree query returns the point
itself first, at distance 0, and open3d averages over all returned distances --, then keep the points whose mean is
positive and below cloud mean + std_ratio * sample std.  PARITY UNPINNED like the rest of the oracle (no open3d to
run; the exactness of the neighbour search itself is pinned by scipy).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np
from scipy.spatial import cKDTree


def knn_mean_distance(points: np.ndarray, nb_neighbors: int = 20) -> np.ndarray:
    pts = np.asarray(points, dtype=np.float64)
    k = min(nb_neighbors, len(pts))
    dist, _ = cKDTree(pts).query(pts, k=k)
    dist = dist.reshape(len(pts), -1)
    return dist.mean(axis=1)


def statistical_outlier_mask(points: np.ndarray, nb_neighbors: int = 20, std_ratio: float = 2.0) -> Tuple[np.ndarray, np.ndarray]:
    avg = knn_mean_distance(points, nb_neighbors)
    valid = avg > 0
    nv = int(valid.sum())
    if nv < 2:
        return valid, avg
    mean = avg[valid].sum() / nv
    std = np.sqrt(((avg[valid] - mean) ** 2).sum() / (nv - 1))
    return valid & (avg < mean + std_ratio * std), avg
