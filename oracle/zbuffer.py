"""Oracle: depth-based semantic projection (``fruit_nerf/scripts/depth_based_semantic_projection.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Unlike the rest of the oracle this path is PINNED by the reference's
own source: the functions below restate ``get_projection_mat`` (``:31-43``), ``get_projection`` (``:45-49``) and
``update_buffer`` (``:84-105``) statement by statement in numpy (the module itself cannot be imported here: it imports
open3d and cv2 at the top and runs a hard-coded job at import time).
"""

from __future__ import annotations

from typing import Tuple

import numpy as np


def get_projection_mat(fx: float, fy: float, cx: float, cy: float, c2w: np.ndarray) -> np.ndarray:
    orig = c2w[:3, 3]
    rot_inv = c2w[:3, :3].T
    t = -rot_inv @ orig
    extrinsic = np.eye(4)
    extrinsic[:3, :3] = rot_inv
    extrinsic[:3, 3] = t
    K = np.asarray([[fx, 0, -cx, 0], [0, -fy, -cy, 0], [0, 0, 1, 0]], dtype=np.float64)
    return K @ extrinsic


def get_projection(P: np.ndarray, points: np.ndarray) -> np.ndarray:
    points_h = np.hstack((points, np.ones((points.shape[0], 1))))
    return (P @ points_h.T).T


def update_buffer(z_buffer: np.ndarray, pc: np.ndarray, img: np.ndarray, label: int, large: bool = False
                  ) -> Tuple[np.ndarray, np.ndarray, Tuple[np.ndarray, np.ndarray]]:
    """In place, like the reference; the clip bounds come from the buffer shape (1440 x 1920 in the reference)."""
    H, W = z_buffer.shape
    yx = pc[:, :2] / -pc[:, 2:3]
    yx = np.round(yx).astype(int)
    ys = np.clip(yx[:, 0], 0, W - 1)
    xs = np.clip(yx[:, 1], 0, H - 1)
    zs = -pc[:, 2]
    if large:
        img[xs, ys] = label
        z_buffer[xs, ys] = zs
        return z_buffer, img, (xs, ys)
    visible_xs, visible_ys = [], []
    for y, x, z in zip(ys, xs, zs):
        if z <= z_buffer[x, y]:
            z_buffer[x, y] = z
            img[x, y] = label
            visible_xs.append(x)
            visible_ys.append(y)
    return z_buffer, img, (np.asarray(visible_xs, dtype=np.int32), np.asarray(visible_ys, dtype=np.int32))
