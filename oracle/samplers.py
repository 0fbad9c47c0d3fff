"""Oracle: samplers along rays.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates:

* ``UniformSamplerWithNoise.generate_ray_samples`` -- ``fruit_nerf/components/ray_samplers.py:54-104``
  (the arithmetic is fully in the reference; ``spacing_fn`` = identity).
* the proposal sampler wired at ``fruit_nerf/fruit_nerf.py:157-164`` and called at ``:549,501,429,337``:
  upstream nerfstudio 1.1.3 ``ProposalNetworkSampler`` = ``UniformLinDispPiecewiseSampler`` + 2x ``PDFSampler``
  (SURVEY.md A.2/A.3).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .rays import RayBundle


@dataclass
class RaySamples:
    """Materialised [R,S,1] sample tensors (nerfstudio ``RaySamples``/``Frustums`` fields we need)."""

    origins: Tensor  # [R,3]
    directions: Tensor  # [R,3]
    starts: Tensor  # [R,S,1] euclidean bin starts
    ends: Tensor  # [R,S,1]
    spacing_starts: Tensor  # [R,S,1] in [0,1]
    spacing_ends: Tensor  # [R,S,1]
    camera_indices: Optional[Tensor]  # [R,1]
    nears: Tensor  # [R,1]
    fars: Tensor  # [R,1]
    spacing: str  # "uniform" | "piecewise" -- which spacing_to_euclidean_fn rides along

    @property
    def deltas(self) -> Tensor:
        return self.ends - self.starts

    def positions(self) -> Tensor:
        """Upstream ``Frustums.get_positions``: o + d * (start + end) / 2 -> [R,S,3]."""
        return self.origins[:, None, :] + self.directions[:, None, :] * (self.starts + self.ends) / 2

    def spacing_to_euclidean(self, x: Tensor) -> Tensor:
        fn, fn_inv = SPACING[self.spacing]
        s_near, s_far = fn(self.nears), fn(self.fars)
        return fn_inv(x * s_far + (1 - x) * s_near)


def _id(x: Tensor) -> Tensor:
    return x


def _piecewise(x: Tensor) -> Tensor:
    return torch.where(x < 1, x / 2, 1 - 1 / (2 * x))


def _piecewise_inv(x: Tensor) -> Tensor:
    return torch.where(x < 0.5, 2 * x, 1 / (2 - 2 * x))


SPACING = {"uniform": (_id, _id), "piecewise": (_piecewise, _piecewise_inv)}


def spaced_sampler(rb: RayBundle, num_samples: int, spacing: str = "uniform",
                   t_rand: Optional[Tensor] = None) -> RaySamples:
    """``components/ray_samplers.py:54-104`` (and upstream ``SpacedSampler``).

    ``t_rand`` = the stratified jitter (``[R,1]`` single-jitter or ``[R,S+1]``); ``None`` = eval / no jitter.
    Reference quirk kept: the export sampler draws S+1 randoms (``:82-83``) even though each sample uses one.
    """
    assert rb.nears is not None and rb.fars is not None
    bins = torch.linspace(0.0, 1.0, num_samples + 1)[None, ...]  # [1,S+1]
    if t_rand is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * t_rand
    fn, fn_inv = SPACING[spacing]
    s_near, s_far = fn(rb.nears), fn(rb.fars)
    eu = fn_inv(bins * s_far + (1 - bins) * s_near)  # [R,S+1]
    bins = bins.expand(eu.shape)
    return RaySamples(
        origins=rb.origins, directions=rb.directions,
        starts=eu[..., :-1, None], ends=eu[..., 1:, None],
        spacing_starts=bins[..., :-1, None], spacing_ends=bins[..., 1:, None],
        camera_indices=rb.camera_indices, nears=rb.nears, fars=rb.fars, spacing=spacing,
    )


def pdf_sampler(prev: RaySamples, weights: Tensor, num_samples: int, u_rand: Optional[Tensor] = None,
                histogram_padding: float = 0.01, eps: float = 1e-5) -> RaySamples:
    """Upstream ``PDFSampler.generate_ray_samples`` (include_original=False), SURVEY.md A.3.

    weights [R,S_in,1]; ``u_rand`` = training jitter already divided by nothing ([R,1] or [R,S+1] in [0,1));
    ``None`` = eval (bin centres).
    """
    num_bins = num_samples + 1
    w = weights[..., 0] + histogram_padding
    w_sum = torch.sum(w, dim=-1, keepdim=True)
    padding = torch.relu(eps - w_sum)
    w = w + padding / w.shape[-1]
    w_sum = w_sum + padding
    pdf = w / w_sum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)  # [R,S_in+1]

    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins)
    if u_rand is not None:
        u = u.expand(size=(*cdf.shape[:-1], num_bins)) + u_rand / num_bins
    else:
        u = u + 1.0 / (2 * num_bins)
        u = u.expand(size=(*cdf.shape[:-1], num_bins))
    u = u.contiguous()

    existing = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[..., -1:, 0]], dim=-1)  # [R,S_in+1]
    inds = torch.searchsorted(cdf, u, side="right")
    below = torch.clamp(inds - 1, 0, existing.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing.shape[-1] - 1)
    cdf_g0 = torch.gather(cdf, -1, below)
    bins_g0 = torch.gather(existing, -1, below)
    cdf_g1 = torch.gather(cdf, -1, above)
    bins_g1 = torch.gather(existing, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    bins = bins_g0 + t * (bins_g1 - bins_g0)  # [R,S+1] spacing domain
    bins = bins.detach()  # upstream: "Stop gradients"

    eu = prev.spacing_to_euclidean(bins)
    return RaySamples(
        origins=prev.origins, directions=prev.directions,
        starts=eu[..., :-1, None], ends=eu[..., 1:, None],
        spacing_starts=bins[..., :-1, None], spacing_ends=bins[..., 1:, None],
        camera_indices=prev.camera_indices, nears=prev.nears, fars=prev.fars, spacing=prev.spacing,
    )


def get_weights(deltas: Tensor, densities: Tensor) -> Tensor:
    """Upstream ``RaySamples.get_weights`` (``fruit_nerf.py:556``), SURVEY.md A.4."""
    dd = deltas * densities
    alphas = 1 - torch.exp(-dd)
    tr = torch.cumsum(dd[..., :-1, :], dim=-2)
    tr = torch.cat([torch.zeros((*tr.shape[:1], 1, 1)), tr], dim=-2)
    tr = torch.exp(-tr)
    return torch.nan_to_num(alphas * tr)


def proposal_sampler(
    rb: RayBundle,
    density_fns: Sequence[Callable[[Tensor], Tensor]],
    num_proposal_samples: Sequence[int] = (256, 96),
    num_nerf_samples: int = 48,
    anneal: float = 1.0,
    initial_sampler: str = "piecewise",
    jitter: Optional[Sequence[Optional[Tensor]]] = None,
) -> Tuple[RaySamples, List[Tensor], List[RaySamples]]:
    """Upstream ``ProposalNetworkSampler.generate_ray_samples`` as configured at ``fruit_nerf.py:157-164``.

    Level 0: piecewise-linear-in-disparity spaced sampler; level i>0: PDF sampling of ``weights**anneal``.
    ``jitter`` = per-level random tensors for training (None = eval).
    """
    n = len(num_proposal_samples)
    weights_list: List[Tensor] = []
    samples_list: List[RaySamples] = []
    weights = None
    rs = None
    for level in range(n + 1):
        is_prop = level < n
        ns = num_proposal_samples[level] if is_prop else num_nerf_samples
        jit = None if jitter is None else jitter[level]
        if level == 0:
            rs = spaced_sampler(rb, ns, spacing=initial_sampler, t_rand=jit)
        else:
            assert weights is not None and rs is not None
            rs = pdf_sampler(rs, torch.pow(weights, anneal), ns, u_rand=jit)
        if is_prop:
            density = density_fns[level](rs.positions())
            weights = get_weights(rs.deltas, density)
            weights_list.append(weights)
            samples_list.append(rs)
    assert rs is not None
    return rs, weights_list, samples_list
