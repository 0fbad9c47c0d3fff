"""Mirror of ``crop_nerf/fruit_nerf/scripts/depth_based_semantic_projection.py`` on the HIP z-buffer kernels: the
depth-based alternative to the NeRF projection.

Same function names and argument meaning -- ``get_projection_mat`` (``:31-43``), ``get_projection`` (``:45-49``),
``update_buffer`` (``:84-105``), ``remove_part`` (``:108-117``, an AABB crop), ``project_and_save_super_clusters``
(``:120-181``) -- with the point clouds, the z-buffer and the label image resident on the device.  The reference
hard-codes 1920 x 1440 images and the intrinsics of its recording; here they are arguments with those defaults.

Two documented differences: ``update_buffer(large=False)`` returns the accepted PIXELS (a mask / their coordinates) instead
of one entry per accepted point -- the only use of that return value is ``img[xs, ys] = 255`` (``:159-161``); and a point
whose depth falls between the float32 rounding of the current buffer value and the float64 value it came from can be
decided differently (one float32 ulp).
"""

from __future__ import annotations

import os
import shutil
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from ... import ops

# intrinsics of the reference's recording (depth_based_semantic_projection.py:17-20)
FX, FY, CX, CY = 1442.4805, 1442.4805, 960.0, 720.0
IMG_H, IMG_W = 1440, 1920


def get_projection_mat(fx: float, fy: float, cx: float, cy: float, c2w) -> np.ndarray:
    """``:31-43``: P = K @ [R^T | -R^T o] with the reference's sign convention (float64)."""
    c2w = np.asarray(c2w, dtype=np.float64)
    orig = c2w[:3, 3]
    rot_inv = c2w[:3, :3].T
    t = -rot_inv @ orig
    extrinsic = np.eye(4)
    extrinsic[:3, :3] = rot_inv
    extrinsic[:3, 3] = t
    K = np.asarray([[fx, 0, -cx, 0], [0, -fy, -cy, 0], [0, 0, 1, 0]], dtype=np.float64)
    return K @ extrinsic


def _dev_points(points, device) -> Tensor:
    t = points if isinstance(points, Tensor) else torch.as_tensor(np.asarray(points))
    return t.to(device=device, dtype=torch.float64).contiguous()


def get_projection(P, points, height: int = IMG_H, width: int = IMG_W, device="cuda") -> Tuple[Tensor, Tensor, Tensor]:
    """``:45-49`` fused with the pixel arithmetic of ``update_buffer``: returns (xs rows, ys columns, zs) on the device."""
    Pd = torch.as_tensor(np.asarray(P, dtype=np.float64)).to(device).contiguous()
    return ops.depth_project(Pd, _dev_points(points, device), height, width)


def update_buffer(z_buffer: Tensor, pc: Tuple[Tensor, Tensor, Tensor], img: Tensor, label: int, large: bool = False):
    """``:84-105`` in place; ``pc`` = the (xs, ys, zs) of ``get_projection``.  Returns (z_buffer, img, visible mask or
    None)."""
    xs, ys, zs = pc
    vis = ops.zbuffer_update(z_buffer, img, xs, ys, zs, label, large=large)
    return z_buffer, img, vis


def remove_part(points: Tensor, aabb) -> Tensor:
    """``:108-117``: the points inside the axis-aligned box (open3d ``crop``: bounds inclusive)."""
    lo = torch.as_tensor(np.asarray(aabb[0], dtype=np.float64), device=points.device)
    hi = torch.as_tensor(np.asarray(aabb[1], dtype=np.float64), device=points.device)
    keep = ((points >= lo) & (points <= hi)).all(dim=-1)
    return points[keep].contiguous()


def project_and_save_super_clusters(c2w, cluster_data: Sequence[Dict], full_tree_pc, full_semantic_pc, save_dir: Optional[str],
                                    cam_idx: int = 0, intrinsics=(FX, FY, CX, CY), height: int = IMG_H, width: int = IMG_W,
                                    instance_mask_img: Optional[str] = None, device="cuda"):
    """``:120-181`` for one camera: splat the whole plant as the occluder (label 0, ``large=True``), then per super-cluster
    every sub-cluster's points (label = sub index + 1) against it.  Writes ``occ_free_{i}.png``, ``visible_label.png``
    and ``visible.png`` per super-cluster when ``save_dir`` is given; returns {super index: (visible_label [H,W] uint8,
    {sub index: occlusion-free mask [H,W] uint8})} as device tensors."""
    from ..fruit_nerf import save_image

    P = get_projection_mat(*intrinsics, c2w)
    tree = _dev_points(full_tree_pc, device)
    sem = _dev_points(full_semantic_pc, device)
    z_init = torch.full((height, width), float("inf"), dtype=torch.float32, device=device)
    img_init = torch.zeros(height, width, dtype=torch.uint8, device=device)
    update_buffer(z_init, get_projection(P, tree, height, width, device), img_init, label=0, large=True)
    results = {}
    for sup_idx, sup in enumerate(cluster_data):
        cam_dir = None
        if save_dir is not None:
            cam_dir = os.path.join(save_dir, f"super_cluster_{sup_idx}", f"cam_{cam_idx}")
            os.makedirs(cam_dir, exist_ok=True)
        z_buffer = z_init.clone()
        visible_label = img_init.clone()
        occ = {}
        for sub_idx in sup["pcd"].keys():
            pc = remove_part(sem, sup["aabb"][sub_idx])
            _, _, vis = update_buffer(z_buffer, get_projection(P, pc, height, width, device), visible_label, sub_idx + 1)
            occ[sub_idx] = vis
            if cam_dir is not None:
                save_image((vis.float() / 255.0)[..., None].expand(-1, -1, 3), os.path.join(cam_dir, f"occ_free_{sub_idx}.png"))
                if instance_mask_img and os.path.exists(instance_mask_img):
                    shutil.copy(instance_mask_img, cam_dir)
        if cam_dir is not None:
            save_image((visible_label.float() / 255.0)[..., None].expand(-1, -1, 3), os.path.join(cam_dir, "visible_label.png"))
            save_image(((visible_label > 0).float())[..., None].expand(-1, -1, 3), os.path.join(cam_dir, "visible.png"))
        results[sup_idx] = (visible_label, occ)
    return results
