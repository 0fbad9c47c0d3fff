"""``OrthographicRayGenerator`` -- mirror of ``crop_nerf/fruit_nerf/components/ray_generators.py:24-66`` on
``cn_raygen_ortho``: parallel rays from a surface grid, used for the dense volume export."""

from __future__ import annotations

import torch

from ... import ops
from ...rays import RayBundle


class OrthographicRayGenerator:
    def __init__(self, surface_points, plane_normal, ray_batch_size, device, aabb) -> None:
        self.surface_points = surface_points.to(device).contiguous()
        self.plane_vector = [float(v) for v in plane_normal.reshape(-1).tolist()]
        self.surface_vector_norm = float(torch.linalg.norm(plane_normal))
        self.ray_batch_size = ray_batch_size
        self.device = device
        self.aabb = aabb

    def forward(self, count) -> RayBundle:
        """Batch ``count`` (1-based): points [bs*(count-1), bs*count), clipped at the end (:52-56)."""
        start = self.ray_batch_size * (count - 1)
        end = self.ray_batch_size * count
        if self.ray_batch_size * count >= self.surface_points.shape[0]:
            end = self.surface_points.shape[0]
        n = max(end - start, 0)
        out = ops.raygen_ortho(self.surface_points, self.plane_vector, start, n)
        return RayBundle(out["origins"], out["directions"], out["pixel_area"], None, out["nears"], out["fars"])

    __call__ = forward
