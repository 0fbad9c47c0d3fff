"""Mirror of ``crop_nerf/segmentation/segmenter.py`` -- the stage between the point-cloud export and the semantic
projections: it turns the exported semantic point cloud into the
``all_super_cluster_info*.npy`` list that ``FruitModel.get_outputs_for_projections`` and the depth-based projection read.

``get_super_clusters`` (``:69-86``: voxel down-sampling, DBSCAN, noise removal, statistical outlier removal) runs on the HIP
kernels (``csrc/cluster.hip``, ``csrc/knn.hip``) with the cloud resident on the device.  ``cluster_kmeans`` (``:28-45``) runs
its Lloyd iterations on the device too (``cn_kmeans_step``, float64); only the k-means++ seeding -- a few weighted draws --
stays with scikit-learn's own routine and random stream on the host, so the labels are scikit-learn's.  open3d point-cloud objects are replaced by [N,3] arrays / tensors; the visualisation helpers
are not mirrored.
"""

from __future__ import annotations

from collections import Counter
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import ops


def cluster_kmeans(points, k: int = 10, device="cuda") -> np.ndarray:
    """``:28-45`` without the normals option: KMeans(init="k-means++", n_clusters=k, n_init="auto", random_state=0) --
    ``ops.kmeans`` (Lloyd iterations on the device, scikit-learn's seeding on the host)."""
    pts = points if isinstance(points, Tensor) else torch.as_tensor(np.asarray(points))
    return ops.kmeans(pts.to(device), k).cpu().numpy()


def get_super_clusters(points, vx_size: float = 10e-5, device="cuda") -> Tuple[Tensor, Tensor]:
    """``:69-86``: returns (points [M,3] on the device, labels [M] int64)."""
    pts = points if isinstance(points, Tensor) else torch.as_tensor(np.asarray(points))
    return ops.get_super_clusters(pts.to(device=device, dtype=torch.float32).contiguous(), vx_size)


def _by_size(labels: Tensor) -> List[Tuple[int, int]]:
    return sorted([(v, k) for k, v in Counter(labels.tolist()).items()], reverse=True)


def get_nth_largest_super_point(points: Tensor, labels: Tensor, n: int) -> Tuple[Tensor, Tensor]:
    """``:88-92``."""
    label = _by_size(labels)[n][1]
    return points[labels == label], labels


def bounds_as_sorted_list(points: Tensor, labels: Tensor) -> List[np.ndarray]:
    """``:102-112``: [min_bound, max_bound] of every super-cluster, largest first."""
    out = []
    for _, label in _by_size(labels):
        p = points[labels == label]
        out.append(np.stack([p.min(dim=0).values.cpu().numpy(), p.max(dim=0).values.cpu().numpy()]).astype(np.float64))
    return out


def process_and_save_all(points, k: int, save_path=None, vx_size: float = 10e-5, device="cuda") -> List[Dict]:
    """``:153-181``: super-clusters (largest first) -> k-means sub-clusters -> {'aabb': [k,2,3], 'pcd': {i: points}} each;
    super-clusters with at most ``k`` points are skipped.  Written with ``np.save`` when ``save_path`` is given."""
    pts, labels = get_super_clusters(points, vx_size, device)
    res = []
    for _, label in _by_size(labels):
        poi = pts[labels == label].double().cpu().numpy()
        if len(poi) <= k:
            continue
        sub = cluster_kmeans(poi, k=k)
        pc_aabb, pc_list = [], []
        for i in range(k):
            p = poi[sub == i]
            pc_aabb.append(np.stack([p.min(axis=0), p.max(axis=0)]))
            pc_list.append(p)
        res.append({"aabb": np.stack(pc_aabb), "pcd": {i: pc for i, pc in enumerate(pc_list)}})
    if save_path is not None:
        np.save(save_path, np.asarray(res, dtype=object), allow_pickle=True)
    return res


def process_for_pipeline(input_path, dataname: str = "semantics_pc.ply", k: int = 2, vx_size: float = 10e-5) -> str:
    """``:183-185``: ``<input_path>/<dataname>`` (a PLY written by the exporters) -> ``<input_path>/all_super_cluster_info_nsub_2.npy``."""
    import os

    from ..fruit_nerf.ply import read_ply

    save_path = os.path.join(input_path, "all_super_cluster_info_nsub_2.npy")
    points, _ = read_ply(os.path.join(input_path, dataname))
    res = process_and_save_all(points, k=k, save_path=save_path, vx_size=vx_size)
    print(f"{len(res)} super-clusters -> {save_path}")
    return save_path


if __name__ == "__main__":  # python segmenter.py <pcd dir> [ply name] [k] [voxel size]  (the reference hard-codes these, :206-218)
    import sys

    a = sys.argv[1:]
    process_for_pipeline(a[0], a[1] if len(a) > 1 else "semantics_pc.ply", int(a[2]) if len(a) > 2 else 2,
                         float(a[3]) if len(a) > 3 else 10e-5)
