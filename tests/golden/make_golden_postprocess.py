#!/usr/bin/env python
"""Golden vectors of the stages either side of the path: inputs + expected outputs.

    python tests/golden/make_golden_postprocess.py        ->  tests/golden/postprocess_small.npz

* ``zbuffer/*``: the depth-based projection -- produced by ``oracle/zbuffer.py``, which restates the reference's numpy
  functions (``scripts/depth_based_semantic_projection.py:31-105``) statement by statement, so these are the reference's
  results on this input;
* ``cluster/*`` / ``outlier/*``: voxel down-sampling, DBSCAN and the statistical-outlier pass from ``oracle/clustering.py``
  / ``oracle/outliers.py`` (scikit-learn / scipy evaluations of open3d's published algorithms; open3d is absent here).
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import clustering as OC  # noqa: E402
from oracle import outliers as OO  # noqa: E402
from oracle import zbuffer as OZ  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # ---- z-buffer: an occluding "plant" cloud (large=True, label 0) then a "fruit" cloud (label 1) ------------------------
    H, W = 60, 80
    c2w = np.eye(4)
    c2w[:3, 3] = [0.1, -0.05, 2.2]
    intr = (90.0, 90.0, 40.0, 30.0)
    P = OZ.get_projection_mat(*intr, c2w)
    tree = rng.normal(size=(4000, 3)) * 0.35
    fruit = rng.normal(size=(900, 3)) * 0.12 + [0.1, 0.0, 0.2]
    z = np.full((H, W), np.inf, dtype=np.float32)
    img = np.zeros((H, W), dtype=np.uint8)
    z, img, _ = OZ.update_buffer(z, OZ.get_projection(P, tree), img, 0, large=True)
    z, img, (vx, vy) = OZ.update_buffer(z, OZ.get_projection(P, fruit), img, 1)
    vis = np.zeros((H, W), dtype=np.uint8)
    vis[vx, vy] = 255
    out.update({"zbuffer/c2w": c2w, "zbuffer/intr": np.array(intr), "zbuffer/hw": np.array([H, W]), "zbuffer/tree": tree,
                "zbuffer/fruit": fruit, "zbuffer/out/z": z, "zbuffer/out/img": img, "zbuffer/out/visible": vis})
    # ---- clustering: five blobs + clutter ----------------------------------------------------------------------------------
    centres = rng.uniform(-0.4, 0.4, size=(5, 3))
    pts = np.concatenate([rng.normal(size=(700, 3)) * 0.012 + c for c in centres] + [rng.uniform(-0.6, 0.6, size=(500, 3))])
    pts = pts[rng.permutation(len(pts))].astype(np.float32)
    vx_size, eps, mp = 0.004, 0.02, 12
    down = OC.voxel_down_sample(pts.astype(np.float64), vx_size)
    down = down[np.lexsort(np.round(down, 6).T[::-1])].astype(np.float32)
    labels, core = OC.dbscan(down, eps, mp)
    out.update({"cluster/points": pts, "cluster/voxel_size": np.array(vx_size), "cluster/eps": np.array(eps),
                "cluster/min_points": np.array(mp), "cluster/out/down": down, "cluster/out/labels": labels,
                "cluster/out/core": core})
    # ---- statistical outlier removal (20 neighbours, std_ratio 2) on the down-sampled cloud --------------------------------
    mean_d = OO.knn_mean_distance(down.astype(np.float64), 20)
    mask, _ = OO.statistical_outlier_mask(down.astype(np.float64), 20, 2.0)
    out.update({"outlier/out/mean_distance": mean_d, "outlier/out/mask": mask})
    path = os.path.join(HERE, "postprocess_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", int(labels.max()) + 1, "clusters,", int((labels < 0).sum()), "noise,",
          int((~mask).sum()), "outliers")


if __name__ == "__main__":
    main()
