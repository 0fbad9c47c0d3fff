"""``UniformSamplerWithNoise`` -- mirror of ``crop_nerf/fruit_nerf/components/ray_samplers.py:31-104`` on
``cn_sample_spaced``: linear-in-t stratified sampler used by the export / inference wiring (``fruit_nerf.py:188``)."""

from __future__ import annotations

from typing import Optional

import torch

from ... import _lib as L
from ... import ops
from ...rays import RayBundle, RaySamples


class UniformSamplerWithNoise:
    def __init__(self, num_samples: Optional[int] = None, train_stratified: bool = True, single_jitter: bool = False):
        self.num_samples = num_samples
        self.train_stratified = train_stratified
        self.single_jitter = single_jitter
        self.training = False

    def generate_ray_samples(self, ray_bundle: Optional[RayBundle] = None, num_samples: Optional[int] = None) -> RaySamples:
        assert ray_bundle is not None
        assert ray_bundle.nears is not None
        assert ray_bundle.fars is not None
        num_samples = num_samples or self.num_samples
        assert num_samples is not None
        num_rays = ray_bundle.origins.shape[0]
        t_rand = None
        if self.train_stratified and self.training:
            # same shapes as the reference draws (:80-83); torch.rand supplies the random numbers, the kernel the bins
            cols = 1 if self.single_jitter else num_samples + 1
            t_rand = torch.rand((num_rays, cols), dtype=torch.float32, device=ray_bundle.origins.device)
        out = ops.sample_spaced(ray_bundle.nears.contiguous(), ray_bundle.fars.contiguous(), num_samples,
                                L.SPACING_UNIFORM, t_rand)
        return RaySamples(ray_bundle.origins, ray_bundle.directions, out["starts"][..., None], out["ends"][..., None],
                          out["spacing_starts"][..., None], out["spacing_ends"][..., None], ray_bundle.camera_indices)

    __call__ = generate_ray_samples
