"""Device-side ray containers with nerfstudio's field names (``RayBundle``, ``RaySamples``/``Frustums``, ``Cameras``,
``SceneBox``): what the reference's call sites pass around (``fruit_nerf/fruit_nerf.py:283-310,617-637``).  SoA tensors
on the ROCm device; no nerfstudio import."""

from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, Optional, Sequence, Union

import torch
from torch import Tensor


@dataclass
class SceneBox:
    aabb: Tensor  # [2,3]

    def flat(self):
        return [float(v) for v in self.aabb.reshape(-1).tolist()]


@dataclass
class OrientedBox:
    """nerfstudio ``OrientedBox`` (the ``--obb_center / --obb_rotation / --obb_scale`` crop of ``ns-export pointcloud``,
    ``debug/exporter_nerfacto.py:119-121``): rotation R [3,3], centre T [3], edge lengths S [3]."""

    R: Tensor
    T: Tensor
    S: Tensor

    @staticmethod
    def from_params(pos, rpy, scale) -> "OrientedBox":
        """Centre, roll-pitch-yaw in radians (R = Rz(yaw) Ry(pitch) Rx(roll), viser's ``SO3.from_rpy_radians``), scale."""
        import math

        r, p, y = (float(v) for v in rpy)
        rx = torch.tensor([[1.0, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
        ry = torch.tensor([[math.cos(p), 0, math.sin(p)], [0, 1.0, 0], [-math.sin(p), 0, math.cos(p)]])
        rz = torch.tensor([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1.0]])
        return OrientedBox(rz @ ry @ rx, torch.tensor([float(v) for v in pos]), torch.tensor([float(v) for v in scale]))

    def within(self, pts: Tensor) -> Tensor:
        """[N,3] -> [N] bool: strictly inside (both comparisons are strict upstream)."""
        R, T, S = self.R.to(pts), self.T.to(pts), self.S.to(pts)
        local = (pts - T) @ R  # R^T (p - T), row-vector form
        return ((local > -S / 2) & (local < S / 2)).all(dim=-1)


@dataclass
class RayBundle:
    origins: Tensor  # [R,3] or [H,W,3]
    directions: Tensor
    pixel_area: Optional[Tensor] = None
    camera_indices: Optional[Tensor] = None  # [R,1] int64
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)

    _FIELDS = ("origins", "directions", "pixel_area", "camera_indices", "nears", "fars")

    def __len__(self) -> int:
        n = 1
        for s in self.origins.shape[:-1]:
            n *= s
        return n

    @property
    def shape(self):
        return tuple(self.origins.shape[:-1])

    def _map(self, fn) -> "RayBundle":
        kw = {k: (None if getattr(self, k) is None else fn(getattr(self, k))) for k in self._FIELDS}
        return RayBundle(**kw, metadata={k: fn(v) for k, v in self.metadata.items()})

    def flatten(self) -> "RayBundle":
        return self._map(lambda t: t.reshape(-1, t.shape[-1]))

    def to(self, device) -> "RayBundle":
        return self._map(lambda t: t.to(device))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        return self.flatten()._map(lambda t: t[start_idx:end_idx])

    def __getitem__(self, idx) -> "RayBundle":
        """Boolean-mask / index selection over the leading (ray) dimensions, like nerfstudio's TensorDataclass."""
        if isinstance(idx, Tensor) and idx.dtype == torch.bool:
            n = idx.dim()
            return self._map(lambda t: t[idx.to(t.device)] if t.dim() - 1 == n else t.reshape(-1, t.shape[-1])[idx.reshape(-1).to(t.device)])
        return self._map(lambda t: t[idx])

    def clone(self) -> "RayBundle":
        return self._map(lambda t: t.clone())


@dataclass
class RaySamples:
    """Materialised samples ([R,S,1] tensors), only built on the unfused / training-list paths."""

    origins: Tensor  # [R,3]
    directions: Tensor  # [R,3]
    starts: Tensor  # [R,S,1]
    ends: Tensor
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    camera_indices: Optional[Tensor] = None
    density: Optional[Tensor] = None  # [R,S] proposal density at these samples (training lists only)

    @property
    def deltas(self) -> Tensor:
        return self.ends - self.starts

    @property
    def shape(self):
        return tuple(self.starts.shape[:-1])


@dataclass
class Cameras:
    """Undistorted perspective cameras (nerfstudio ``Cameras`` subset used by the reference's hot path)."""

    camera_to_worlds: Tensor  # [N,3,4]
    fx: Tensor  # [N]
    fy: Tensor
    cx: Tensor
    cy: Tensor
    height: int
    width: int
    metadata: Optional[dict] = None  # nerfstudio's Cameras.metadata ("cam_idx" for the pose optimiser's apply_to_camera)

    def __len__(self) -> int:
        return self.camera_to_worlds.shape[0]

    @property
    def size(self) -> int:
        return len(self)

    @property
    def device(self):
        return self.camera_to_worlds.device

    def __getitem__(self, i) -> "Cameras":
        """An int or a slice -> a batch of cameras (``cameras[idx : idx + 1]`` at ``exporter_utils_nerfacto.py:321``); every
        selection owns a fresh ``metadata`` dict, like nerfstudio's."""
        if not isinstance(i, slice):
            i = int(i)
            if i < 0:
                i += len(self)
            i = slice(i, i + 1)
        return Cameras(self.camera_to_worlds[i], self.fx[i], self.fy[i], self.cx[i], self.cy[i], self.height, self.width,
                       dict(self.metadata or {}))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def rescale_output_resolution(self, scaling_factor: float) -> None:
        """nerfstudio ``Cameras.rescale_output_resolution`` (in place, default ``floor`` rounding), as ``render_trajectory``
        calls it (``exporter_utils_nerfacto.py:254``)."""
        if float(scaling_factor) == 1.0:
            return
        f = float(scaling_factor)
        self.fx, self.fy, self.cx, self.cy = self.fx * f, self.fy * f, self.cx * f, self.cy * f
        self.height, self.width = int(self.height * f), int(self.width * f)

    @property
    def image_height(self):
        return self.height

    @property
    def image_width(self):
        return self.width

    def intrinsics(self) -> Tensor:
        return torch.stack([self.fx, self.fy, self.cx, self.cy], dim=-1).to(torch.float32).contiguous()

    def to(self, device) -> "Cameras":
        return Cameras(self.camera_to_worlds.to(device), self.fx.to(device), self.fy.to(device), self.cx.to(device),
                       self.cy.to(device), self.height, self.width, self.metadata)

    def generate_rays(self, camera_indices: Union[int, Tensor], keep_shape: Optional[bool] = None,
                      aabb_box: Optional[SceneBox] = None, coords: Optional[Tensor] = None) -> RayBundle:
        """``Cameras.generate_rays`` for (a) one camera's full image (``camera_indices`` int, ``keep_shape=True`` gives
        [H,W,.] tensors as at ``fruit_nerf.py:283``) and (b) a [R,3] tensor of (camera,row,col) ray indices
        (``train_ray_generator``).  ``aabb_box`` fills nears/fars with the slab test (misses: 1e10)."""
        from . import ops

        dev = self.camera_to_worlds.device
        c2w = self.camera_to_worlds.to(torch.float32).contiguous()
        intr = self.intrinsics()
        if isinstance(camera_indices, int):
            out = ops.raygen_pinhole(c2w, intr, cam=camera_indices, height=self.height, width=self.width,
                                     camera_index_value=camera_indices)
        else:
            out = ops.raygen_pinhole(c2w, intr, ray_indices=camera_indices.to(dev).to(torch.int64).contiguous())
        rb = RayBundle(out["origins"], out["directions"], out["pixel_area"], out["camera_indices"],
                       metadata={"directions_norm": out["directions_norm"]})
        if aabb_box is not None:
            rb.nears, rb.fars = ops.intersect_aabb(rb.origins, rb.directions, aabb_box.flat())
        if isinstance(camera_indices, int) and keep_shape:
            rb = rb._map(lambda t: t.reshape(self.height, self.width, t.shape[-1]))
        return rb
