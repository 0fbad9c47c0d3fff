"""Oracle: ray generation, AABB intersection, collider, pose refinement.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates, in plain PyTorch fp32:

* pinhole ray generation -- reference call sites ``fruit_nerf/data/fruit_datamanager.py:188-197``
  (``train_ray_generator(ray_indices)``), ``fruit_nerf/fruit_nerf.py:283``
  (``cam.generate_rays(camera_indices=0, keep_shape=True, aabb_box=aabb)``),
  ``fruit_nerf/export/exporter_utils_nerfacto.py:266-268``; arithmetic = upstream
  nerfstudio 1.1.3 ``Cameras._generate_rays_from_coords``.
* ``intersect_aabb`` -- upstream ``nerfstudio.utils.math.intersect_aabb`` as used by
  ``generate_rays(aabb_box=...)`` (``fruit_nerf/fruit_nerf.py:283-286``).
* orthographic surface rays -- ``fruit_nerf/data/fruit_datamanager.py:42-121,157-172,199-204``
  and ``fruit_nerf/components/ray_generators.py:46-66``.
* ``NearFarCollider`` -- ``fruit_nerf/fruit_nerf.py:167,625-626`` (upstream scene_colliders).
* SO3xR3 camera pose refinement -- ``fruit_nerf/fruit_nerf.py:114-116,547`` (upstream
  ``CameraOptimizer.apply_to_raybundle`` / ``exp_map_SO3xR3``).
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional, Tuple

import torch
from torch import Tensor

_EPS_NORM = 1e-7  # upstream camera_utils.normalize_with_norm floor


@dataclass
class RayBundle:
    """SoA ray bundle, same field names as nerfstudio ``RayBundle``."""

    origins: Tensor  # [R,3]
    directions: Tensor  # [R,3] unit
    pixel_area: Tensor  # [R,1]
    camera_indices: Optional[Tensor] = None  # [R,1] int64
    nears: Optional[Tensor] = None  # [R,1]
    fars: Optional[Tensor] = None  # [R,1]
    directions_norm: Optional[Tensor] = None  # [R,1]

    def __len__(self) -> int:
        return self.origins.shape[0]

    def slice(self, start: int, end: int) -> "RayBundle":
        """``get_row_major_sliced_ray_bundle`` on an already flattened bundle."""

        def s(x):
            return None if x is None else x[start:end]

        return RayBundle(
            s(self.origins), s(self.directions), s(self.pixel_area), s(self.camera_indices),
            s(self.nears), s(self.fars), s(self.directions_norm),
        )

    def mask(self, m: Tensor) -> "RayBundle":
        def s(x):
            return None if x is None else x[m]

        return RayBundle(
            s(self.origins), s(self.directions), s(self.pixel_area), s(self.camera_indices),
            s(self.nears), s(self.fars), s(self.directions_norm),
        )

    def clone(self) -> "RayBundle":
        def c(x):
            return None if x is None else x.clone()

        return RayBundle(
            c(self.origins), c(self.directions), c(self.pixel_area), c(self.camera_indices),
            c(self.nears), c(self.fars), c(self.directions_norm),
        )


def pinhole_rays(
    c2w: Tensor,  # [C,3,4] camera-to-world (OpenGL: -z forward, +y up)
    intrinsics: Tensor,  # [C,4] fx, fy, cx, cy
    cam_idx: Tensor,  # [R] int64
    rows: Tensor,  # [R] int64 pixel row (y)
    cols: Tensor,  # [R] int64 pixel col (x)
) -> RayBundle:
    """Upstream ``Cameras._generate_rays_from_coords`` for undistorted perspective cameras.

    coords = (row+0.5, col+0.5); cam dir ((x-cx)/fx, -(y-cy)/fy, -1); two more with x+1 / y+1
    for ``pixel_area``; rotate with c2w[:3,:3] (sum over last axis of dir[...,None,:]*R);
    normalise with a 1e-7 floor; origins = c2w[:3,3].
    """
    cam_idx = cam_idx.long()
    y = rows.to(torch.float32) + 0.5
    x = cols.to(torch.float32) + 0.5
    fx, fy, cx, cy = (intrinsics[cam_idx, i] for i in range(4))
    coord = torch.stack([(x - cx) / fx, -(y - cy) / fy], -1)
    coord_x = torch.stack([(x - cx + 1) / fx, -(y - cy) / fy], -1)
    coord_y = torch.stack([(x - cx) / fx, -(y - cy + 1) / fy], -1)
    stack = torch.stack([coord, coord_x, coord_y], dim=0)  # [3,R,2]
    dirs = torch.cat([stack, -torch.ones_like(stack[..., :1])], dim=-1)  # [3,R,3]
    rot = c2w[cam_idx][:, :3, :3]  # [R,3,3]
    dirs = torch.sum(dirs[..., None, :] * rot[None], dim=-1)  # [3,R,3]
    norm = torch.clamp_min(torch.linalg.vector_norm(dirs, dim=-1, keepdim=True), _EPS_NORM)
    dirs = dirs / norm
    d = dirs[0]
    dx = torch.sqrt(torch.sum((d - dirs[1]) ** 2, dim=-1))
    dy = torch.sqrt(torch.sum((d - dirs[2]) ** 2, dim=-1))
    return RayBundle(
        origins=c2w[cam_idx][:, :3, 3].contiguous(),
        directions=d.contiguous(),
        pixel_area=(dx * dy)[:, None],
        camera_indices=cam_idx[:, None].clone(),
        directions_norm=norm[0],
    )


def image_rays(c2w: Tensor, intrinsics: Tensor, cam: int, height: int, width: int,
               start: int = 0, end: Optional[int] = None, camera_index_value: Optional[int] = None) -> RayBundle:
    """Full-image rays of one camera, row-major flattened, optionally the slice [start,end).

    ``camera_index_value`` reproduces the reference's ``generate_rays(camera_indices=0, ...)`` on a
    single-camera ``Cameras`` slice (``fruit_nerf/fruit_nerf.py:283``): every ray carries index 0.
    """
    end = height * width if end is None else min(end, height * width)
    pix = torch.arange(start, end, dtype=torch.int64)
    rows, cols = pix // width, pix % width
    rb = pinhole_rays(c2w, intrinsics, torch.full_like(pix, cam), rows, cols)
    if camera_index_value is not None:
        rb.camera_indices = torch.full_like(rb.camera_indices, camera_index_value)
    return rb


def intersect_aabb(origins: Tensor, directions: Tensor, aabb: Tensor,
                   max_bound: float = 1e10, invalid_value: float = 1e10) -> Tuple[Tensor, Tensor]:
    """Upstream ``nerfstudio.utils.math.intersect_aabb`` (slab test). aabb = [6] (min xyz, max xyz)."""
    aabb = aabb.reshape(-1)
    tx_min = (aabb[:3] - origins) / directions
    tx_max = (aabb[3:] - origins) / directions
    t_min = torch.stack((tx_min, tx_max)).amin(dim=0)
    t_max = torch.stack((tx_min, tx_max)).amax(dim=0)
    t_min = t_min.amax(dim=-1)
    t_max = t_max.amin(dim=-1)
    t_min = torch.clamp(t_min, min=0, max=max_bound)
    t_max = torch.clamp(t_max, min=0, max=max_bound)
    cond = t_max <= t_min
    t_min = torch.where(cond, torch.full_like(t_min, invalid_value), t_min)
    t_max = torch.where(cond, torch.full_like(t_max, invalid_value), t_max)
    return t_min, t_max


def with_aabb_near_far(rb: RayBundle, aabb: Tensor) -> RayBundle:
    """``generate_rays(aabb_box=...)``: nears/fars <- ray/AABB hit, misses get 1e10 (``fruit_nerf.py:283-286``)."""
    t_min, t_max = intersect_aabb(rb.origins, rb.directions, aabb)
    return replace(rb, nears=t_min[:, None], fars=t_max[:, None])


def near_far_collider(rb: RayBundle, training: bool, near_plane: float = 0.05, far_plane: float = 1000.0) -> RayBundle:
    """Upstream ``NearFarCollider`` (``fruit_nerf.py:167,625-626``): untouched when nears/fars are set;
    near plane is reset to 0 outside training."""
    if rb.nears is not None and rb.fars is not None:
        return rb
    ones = torch.ones_like(rb.origins[..., 0:1])
    near = near_plane if training else 0.0
    return replace(rb, nears=ones * near, fars=ones * far_plane)


# ----------------------------------------------------------------------------------------------
# Orthographic surface rays (dense volume export)
# ----------------------------------------------------------------------------------------------

def corners_of_aabb(aabb: Tensor) -> Tensor:
    """``get_corners_of_aabb`` (``data/fruit_datamanager.py:42-69``): 8 corners, x fastest then y then z."""
    mn, mx = aabb[0], aabb[1]
    return torch.stack([
        torch.stack([mn[0], mn[1], mn[2]]), torch.stack([mx[0], mn[1], mn[2]]),
        torch.stack([mn[0], mx[1], mn[2]]), torch.stack([mx[0], mx[1], mn[2]]),
        torch.stack([mn[0], mn[1], mx[2]]), torch.stack([mx[0], mn[1], mx[2]]),
        torch.stack([mn[0], mx[1], mx[2]]), torch.stack([mx[0], mx[1], mx[2]]),
    ]).to(torch.float32)


def surface_points(corners: Tensor, n: int) -> Tuple[Tensor, Tensor]:
    """``sample_surface_points`` (``data/fruit_datamanager.py:71-121``).

    Grid on the face through corners 0,1,2; counts int(dx/dconst*n) x int(dy/dconst*n) with ``dconst``
    the extent along the constant axis; the constant coordinate is written into column 2 regardless of
    which axis is constant (reference quirk, ``:108-111``); plane vector (0,0,sign(c4)*|c1|+|c4|) on
    that axis.  ``torch.meshgrid`` default indexing is "ij".
    """
    c1, c2, c3 = corners[0], corners[1], corners[2]
    ext = torch.abs(corners.max(dim=0).values - corners.min(dim=0).values)
    const = int(torch.argmax(torch.logical_and(c1 == c2, c2 == c3).to(torch.int64)))
    ax = int(torch.argmax(torch.abs(c1 - c2)))
    ay = int(torch.argmax(torch.abs(c1 - c3)))
    nx = int(ext[0] / ext[const] * n)
    ny = int(ext[1] / ext[const] * n)
    x = torch.linspace(float(c1[ax]), float(c2[ax]), nx, dtype=torch.float32)
    y = torch.linspace(float(c1[ay]), float(c3[ay]), ny, dtype=torch.float32)
    xx, yy = torch.meshgrid(x, y, indexing="ij")
    pts = torch.column_stack((xx.flatten(), yy.flatten(), torch.full_like(xx.flatten(), float(c3[const]))))
    c4 = corners[-1]
    plane = torch.tensor([[0.0, 0.0, float(torch.sign(c4[const]) * torch.abs(c1[const]) + torch.abs(c4[const]))]],
                         dtype=torch.float32)
    return pts, plane


def ortho_rays(points: Tensor, plane_vector: Tensor, batch: int, count: int) -> RayBundle:
    """``OrthographicRayGenerator.forward(count)`` (``components/ray_generators.py:46-66``); ``count`` is 1-based
    (``next_sample_volume`` increments ``train_count`` first, ``data/fruit_datamanager.py:199-204``)."""
    start = batch * (count - 1)
    end = batch * count
    if batch * count >= points.shape[0]:
        end = points.shape[0]
    o = points[start:end]
    n = o.shape[0]
    normal = torch.nn.functional.normalize(plane_vector)  # [1,3]
    length = torch.linalg.norm(plane_vector)
    return RayBundle(
        origins=o.contiguous(),
        directions=normal.repeat(n, 1),
        pixel_area=torch.zeros(n, 1),
        nears=torch.zeros(n, 1),
        fars=torch.ones(n, 1) * length,
    )


# ----------------------------------------------------------------------------------------------
# Camera pose refinement (SO3xR3)
# ----------------------------------------------------------------------------------------------

def exp_map_so3xr3(tangent: Tensor) -> Tensor:
    """Upstream ``exp_map_SO3xR3``: [N,6] (t, log-rot) -> [N,3,4]."""
    log_rot = tangent[:, 3:]
    nrms = (log_rot * log_rot).sum(1)
    ang = torch.clamp(nrms, 1e-4).sqrt()
    inv = 1.0 / ang
    fac1 = inv * ang.sin()
    fac2 = inv * inv * (1.0 - ang.cos())
    sk = torch.zeros(tangent.shape[0], 3, 3, dtype=tangent.dtype)
    sk[:, 0, 1] = -log_rot[:, 2]
    sk[:, 0, 2] = log_rot[:, 1]
    sk[:, 1, 0] = log_rot[:, 2]
    sk[:, 1, 2] = -log_rot[:, 0]
    sk[:, 2, 0] = -log_rot[:, 1]
    sk[:, 2, 1] = log_rot[:, 0]
    sk2 = torch.bmm(sk, sk)
    ret = torch.zeros(tangent.shape[0], 3, 4, dtype=tangent.dtype)
    ret[:, :3, :3] = fac1[:, None, None] * sk + fac2[:, None, None] * sk2 + torch.eye(3, dtype=tangent.dtype)[None]
    ret[:, :3, 3] = tangent[:, :3]
    return ret


def apply_pose_adjustment(rb: RayBundle, pose_adjustment: Tensor) -> RayBundle:
    """``camera_optimizer.apply_to_raybundle`` (``fruit_nerf.py:547``): o += t, d = R d, also in eval."""
    assert rb.camera_indices is not None
    corr = exp_map_so3xr3(pose_adjustment[rb.camera_indices[:, 0]])
    o = rb.origins + corr[:, :3, 3]
    d = torch.bmm(corr[:, :3, :3], rb.directions[..., None])[..., 0]
    return replace(rb, origins=o, directions=d)
