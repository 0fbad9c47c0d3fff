"""Oracle: training losses and the training-mode forward (autograd gives the reference gradients).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates ``FruitModel.get_loss_dict``
(``fruit_nerf/fruit_nerf.py:601-615``): ``MSELoss(image[:, :3], rgb)``, ``semantic_loss_weight *
BCEWithLogitsLoss(semantics, fruit_mask)`` and, in training, ``interlevel_loss_mult * interlevel_loss(weights_list,
ray_samples_list)`` (upstream nerfstudio ``losses.interlevel_loss`` / ``lossfun_outer`` / ``outer``);
``get_metrics_dict`` (``:639-645``): PSNR and the distortion metric; ``camera_opt_regularizer`` below.  Works for
both implementations of the field (``FieldSpec.implementation``: parameters under the torch or the tcnn names).
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import field as F
from . import rays as RY
from . import render as RD
from . import samplers as SM

EPS = 1.0e-7  # nerfstudio.model_components.losses.EPS


def sdist(rs: SM.RaySamples) -> Tensor:
    """``ray_samples_to_sdist``: cat(spacing_starts, spacing_ends[-1]) -> [R,S+1]."""
    return torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[..., -1:, 0]], dim=-1)


def outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1
    idx_lo = torch.clamp(idx_lo, min=0, max=y1.shape[-1] - 1)
    idx_hi = torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right")
    idx_hi = torch.clamp(idx_hi, min=0, max=y1.shape[-1] - 1)
    cy1_lo = torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)
    cy1_hi = torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1)
    return cy1_hi - cy1_lo


def lossfun_outer(t, w, t_env, w_env):
    w_outer = outer(t[..., :-1], t[..., 1:], t_env[..., :-1], t_env[..., 1:], w_env)
    return torch.clip(w - w_outer, min=0) ** 2 / (w + EPS)


def interlevel_loss(weights_list: Sequence[Tensor], samples_list: Sequence[SM.RaySamples]) -> Tensor:
    c = sdist(samples_list[-1]).detach()
    w = weights_list[-1][..., 0].detach()
    loss = 0.0
    for rs, weights in zip(samples_list[:-1], weights_list[:-1]):
        loss = loss + torch.mean(lossfun_outer(c, w, sdist(rs), weights[..., 0]))
    return loss


def distortion_loss(weights_list, samples_list) -> Tensor:
    """nerfstudio ``distortion_loss`` on the final level (metric only in the reference)."""
    c = sdist(samples_list[-1])
    w = weights_list[-1][..., 0]
    ut = (c[..., 1:] + c[..., :-1]) / 2
    dut = torch.abs(ut[..., :, None] - ut[..., None, :])
    inter = torch.sum(w * torch.sum(w[..., None, :] * dut, dim=-1), dim=-1)
    intra = torch.sum(w ** 2 * (c[..., 1:] - c[..., :-1]), dim=-1) / 3
    return torch.mean(inter + intra)


def train_forward(
    rb: RY.RayBundle, params: Dict[str, Tensor], fspec: F.FieldSpec, pspecs: List[F.ProposalSpec], aabb: Tensor,
    num_proposal_samples: Sequence[int], num_nerf_samples: int, jitter: Sequence[Optional[Tensor]],
    anneal: float = 1.0, near_plane: float = 0.05, far_plane: float = 1000.0, apply_pose: bool = True,
    update_proposals: bool = True,
) -> Dict[str, Tensor]:
    """``FruitModel.get_outputs`` with ``self.training`` (``fruit_nerf.py:543-599``): collider near 0.05, pose tweak,
    proposal sampler with single-jitter randoms ``jitter[level]`` ([R,1] each), per-camera appearance, no clamp."""
    rb = RY.near_far_collider(rb, training=True, near_plane=near_plane, far_plane=far_plane)
    if apply_pose:
        rb = RY.apply_pose_adjustment(rb, params["camera_optimizer.pose_adjustment"])
    # update_proposals False = the sampler's ``with torch.no_grad()`` branch between scheduled proposal updates
    def _fn(i, ps):
        def fn(pos):
            den = F.proposal_density(pos, params, i, ps, aabb, True)
            return den if update_proposals else den.detach()
        return fn

    fns = [_fn(i, ps) for i, ps in enumerate(pspecs)]
    rs, weights_list, samples_list = SM.proposal_sampler(rb, fns, num_proposal_samples, num_nerf_samples,
                                                         anneal=anneal, jitter=jitter)
    fo = F.field_forward(rs.positions(), rs.directions, rs.camera_indices, params, fspec, aabb, True, "val",
                         training=True)
    weights = SM.get_weights(rs.deltas, fo["density"])
    weights_list = list(weights_list) + [weights]
    samples_list = list(samples_list) + [rs]
    rgb = RD.render_rgb(fo["rgb"], weights, "last_sample", training=True)
    # the semantic MLP sees detached geo features and the renderer detached weights (fruit_field.py:264-266,
    # fruit_nerf.py:586-591): gradients of the semantic loss reach only mlp_semantics and its head
    geo = F.field_density(rs.positions(), params, fspec, aabb, True)[1].detach()
    sem_s = F.semantics_from_geo(geo.reshape(-1, fspec.geo_feat_dim), params, fspec).view(*weights.shape[:2], 1)
    sem = RD.render_semantics(sem_s, weights.detach())
    return {"rgb": rgb, "semantics": sem, "accumulation": RD.render_accumulation(weights),
            "weights_list": weights_list, "ray_samples_list": samples_list, "_field": fo}


def data_losses(outputs: Dict[str, Tensor], image: Tensor, fruit_mask: Tensor, semantic_loss_weight: float = 1.0
                ) -> Dict[str, Tensor]:
    """The two data terms of ``get_loss_dict`` (``fruit_nerf.py:603-608``): ``MSELoss()(image[:, :3], rgb)`` and
    ``semantic_loss_weight * BCEWithLogitsLoss(reduction="mean")(semantics, fruit_mask)`` (``:177-178``).  Pinned by the
    reference's own statements (``tests/golden/make_golden_reference.py: loss_cases``)."""
    return {
        "rgb_loss": torch.nn.functional.mse_loss(image[:, :3], outputs["rgb"]),
        "semantics_loss": semantic_loss_weight * torch.nn.functional.binary_cross_entropy_with_logits(
            outputs["semantics"], fruit_mask),
    }


def loss_dict(outputs: Dict[str, Tensor], image: Tensor, fruit_mask: Tensor, semantic_loss_weight: float = 1.0,
              interlevel_loss_mult: float = 1.0) -> Dict[str, Tensor]:
    """``get_loss_dict`` (``fruit_nerf.py:601-615``)."""
    ld = data_losses(outputs, image, fruit_mask, semantic_loss_weight)
    ld["interlevel_loss"] = interlevel_loss_mult * interlevel_loss(outputs["weights_list"], outputs["ray_samples_list"])
    return ld


def camera_opt_regularizer(pose_adjustment: Tensor, trans_l2_penalty: float = 1e-2,
                           rot_l2_penalty: float = 1e-3) -> Tensor:
    """nerfstudio ``CameraOptimizer.get_loss_dict`` (called at ``fruit_nerf.py:614``), mode SO3xR3."""
    return (pose_adjustment[:, :3].norm(dim=-1).mean() * trans_l2_penalty
            + pose_adjustment[:, 3:].norm(dim=-1).mean() * rot_l2_penalty)


def proposal_update_due(step: int, steps_since_update: int, proposal_warmup: int = 5000,
                        proposal_update_every: int = 5) -> bool:
    """``ProposalNetworkSampler.generate_ray_samples``: ``updated`` (schedule at ``fruit_nerf.py:144-149``)."""
    import numpy as np

    sched = np.clip(np.interp(step, [0, proposal_warmup], [0, proposal_update_every]), 1, proposal_update_every)
    return bool(steps_since_update > sched or step < 10)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-15):
    """torch.optim.Adam (no weight decay, no amsgrad), in place; step is 1-based."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def exponential_decay_lr(step: int, lr_init: float, lr_final: float, max_steps: int) -> float:
    """nerfstudio ``ExponentialDecayScheduler`` without warm-up (``fruit_nerf_config.py:47,51``)."""
    t = min(max(step / max_steps, 0.0), 1.0)
    import math

    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
