"""Oracle: FruitModel forward variants, chunked image render, exporter masks, projection passes.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows ``fruit_nerf/fruit_nerf.py``:
``forward :617-637``, ``get_outputs :543-599``, ``get_inference_outputs :497-541``,
``get_export_outputs :476-494``, ``setup_inference :185-189``, ``get_outputs_for_camera_ray_bundle :377-404``,
``get_density_for_camera_ray_bundle :320-344``, ``get_outputs_for_projections :254-318``; and the exporters'
masking logic ``fruit_nerf/export/exporter_utils.py:100-153`` and
``fruit_nerf/export/exporter_utils_nerfacto.py:156-166``.
"""

from __future__ import annotations

from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import field as F
from . import rays as RY
from . import render as RD
from . import samplers as SM


@dataclass
class ModelConfig:
    """``FruitNerfModelConfig`` (``fruit_nerf.py:59-68``) + the nerfacto defaults it inherits."""

    near_plane: float = 0.05
    far_plane: float = 1000.0
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    disable_scene_contraction: bool = False
    eval_num_rays_per_chunk: int = 1 << 15
    background_color: Union[str, Tuple[float, float, float]] = "last_sample"
    field: F.FieldSpec = dc_field(default_factory=F.FieldSpec)
    proposals: List[F.ProposalSpec] = dc_field(default_factory=F.default_proposal_specs)


class OracleModel:
    """Eval-mode FruitModel on CPU tensors."""

    def __init__(self, params: Dict[str, Tensor], config: ModelConfig, aabb: Tensor, test_mode: str = "test"):
        self.params = params
        self.config = config
        self.aabb = aabb.to(torch.float32)  # scene box [2,3]
        self.test_mode = test_mode
        self.contraction = not config.disable_scene_contraction
        self.uniform_samples: Optional[int] = None  # set by setup_inference
        self.anneal = 1.0
        self.background_override: Optional[Tensor] = None

    # fruit_nerf.py:185-189
    def setup_inference(self, render_rgb: bool, num_inference_samples: int) -> None:
        self.uniform_samples = int(num_inference_samples)
        self.contraction = False  # self.field.spatial_distortion = None

    # -- helpers ---------------------------------------------------------------------------------
    def _density_fns(self):
        # proposal nets keep their own contraction even after setup_inference (only field's is cleared)
        prop_contraction = not self.config.disable_scene_contraction
        return [
            (lambda pos, i=i, ps=ps: F.proposal_density(pos, self.params, i, ps, self.aabb, prop_contraction))
            for i, ps in enumerate(self.config.proposals)
        ]

    def _sample(self, rb: RY.RayBundle):
        if self.uniform_samples is not None:
            return SM.spaced_sampler(rb, self.uniform_samples, "uniform"), [], []
        return SM.proposal_sampler(
            rb, self._density_fns(), self.config.num_proposal_samples_per_ray,
            self.config.num_nerf_samples_per_ray, anneal=self.anneal,
        )

    def _field(self, rs: SM.RaySamples) -> Dict[str, Tensor]:
        return F.field_forward(
            rs.positions(), rs.directions, rs.camera_indices, self.params, self.config.field, self.aabb,
            self.contraction, self.test_mode, training=False,
        )

    def _background(self):
        if self.background_override is not None:
            return self.background_override
        bg = self.config.background_color
        return bg if isinstance(bg, str) else torch.tensor(bg, dtype=torch.float32)

    # -- forward variants ----------------------------------------------------------------------------
    def forward(self, rb: RY.RayBundle) -> Dict[str, Tensor]:
        """``FruitModel.forward`` (``fruit_nerf.py:617-637``), eval mode."""
        rb = RY.near_far_collider(rb, training=False, near_plane=self.config.near_plane,
                                  far_plane=self.config.far_plane)
        if self.test_mode == "export":
            return self.get_export_outputs(rb)
        if self.test_mode != "inference":
            # get_outputs applies the pose tweak also in eval (fruit_nerf.py:545-547)
            rb = RY.apply_pose_adjustment(rb, self.params["camera_optimizer.pose_adjustment"])
        return self._render(rb)

    def _render(self, rb: RY.RayBundle) -> Dict[str, Tensor]:
        rs, weights_list, samples_list = self._sample(rb)
        fo = self._field(rs)
        weights = SM.get_weights(rs.deltas, fo["density"])
        out = {
            "rgb": RD.render_rgb(fo["rgb"], weights, self._background()),
            "accumulation": RD.render_accumulation(weights),
            "depth": RD.render_depth_median(weights, rs.starts, rs.ends),
        }
        for i, (w, s) in enumerate(zip(weights_list, samples_list)):
            out[f"prop_depth_{i}"] = RD.render_depth_median(w, s.starts, s.ends)
        out["semantics"] = RD.render_semantics(fo["semantics"], weights)
        out["semantics_colormap"] = RD.semantics_colormap(out["semantics"])
        out["_weights"] = weights  # oracle-only extras for tests
        out["_starts"] = rs.starts
        out["_ends"] = rs.ends
        return out

    def get_export_outputs(self, rb: RY.RayBundle) -> Dict[str, Tensor]:
        """``get_export_outputs`` (``fruit_nerf.py:476-494``): per-sample outputs, no compositing."""
        assert self.uniform_samples is not None, "export mode needs setup_inference()"
        rs = SM.spaced_sampler(rb, self.uniform_samples, "uniform")
        fo = self._field(rs)
        sem = fo["semantics"][..., 0]
        labels = torch.heaviside(torch.sigmoid(sem) - 0.9, torch.tensor(0.0)).to(torch.long)
        return {
            "rgb": fo["rgb"],
            "point_location": rs.positions(),
            "semantics": sem,
            "density": fo["density"][..., 0],
            "semantics_colormap": labels,
        }

    # -- chunked renders -----------------------------------------------------------------------------
    def render_rays(self, rb: RY.RayBundle) -> Dict[str, Tensor]:
        """``get_outputs_for_camera(_jagged)_ray_bundle`` (``fruit_nerf.py:346-404``): chunk, forward, cat."""
        chunk = self.config.eval_num_rays_per_chunk
        outs: Dict[str, List[Tensor]] = {}
        for i in range(0, len(rb), chunk):
            o = self.forward(rb.slice(i, i + chunk))
            for k, v in o.items():
                if k.startswith("_"):
                    continue
                outs.setdefault(k, []).append(v)
        return {k: torch.cat(v) for k, v in outs.items()}

    def density_for_rays(self, rb: RY.RayBundle) -> Tensor:
        """``get_density_for_camera_ray_bundle`` (``fruit_nerf.py:320-344``): sum_s w per ray, no pose tweak,
        no collider (nears/fars must be set)."""
        chunk = self.config.eval_num_rays_per_chunk
        acc = []
        for i in range(0, len(rb), chunk):
            sub = rb.slice(i, i + chunk)
            rs, _, _ = self._sample(sub)
            fo = self._field(rs)
            w = SM.get_weights(rs.deltas, fo["density"])
            acc.append(w.squeeze(-1).sum(-1))
        return torch.cat(acc)

    # -- projection (fruit_nerf.py:281-315) ----------------------------------------------------------------
    def project_cluster(self, rb_full: RY.RayBundle, aabb: Tensor, height: int, width: int
                        ) -> Tuple[Tensor, Tensor]:
        """One (camera, sub-cluster AABB) job of ``get_outputs_for_projections``.

        ``rb_full``: all H*W rays of the camera (camera_indices = 0, as the reference builds them).
        Returns (wo_occ, visible) float images [H,W,3] *before* ``save_image``'s clamp/quantise.
        """
        rays = RY.with_aabb_near_far(rb_full, aabb.reshape(-1))
        valid = (rays.nears < 1e10)[:, 0]
        img = torch.zeros(height * width, 3)
        if int(valid.sum()) < 10:
            z = img.reshape(height, width, 3)
            return z, z.clone()
        out = self.render_rays(rays.mask(valid))
        img[valid] = out["semantics"]
        img = img.reshape(height, width, 3)
        wo_occ = img.clone()
        occ = rays.clone()
        occ.fars[valid] = rays.nears[valid]
        occ.nears[valid] = 0.0
        w = torch.zeros(height * width)
        w[valid] = self.density_for_rays(occ.mask(valid))
        mark = (w >= 0.5).reshape(height, width)
        img[mark] = 0.0
        return wo_occ, img


# ----------------------------------------------------------------------------------------------
# exporter masks
# ----------------------------------------------------------------------------------------------

def sample_volume_masks(outputs: Dict[str, Tensor]) -> Dict[str, Dict[str, Tensor]]:
    """``sample_volume`` per-call masking (``export/exporter_utils.py:100-153``).

    Three point sets, colour = [rgb, sigmoid(.)]:
      semantic_colormap: label >= 0.999 and density >= 70     (4th colour = sigmoid(sem logit))
      semantic:          sem logit >= 3 and density >= 70     (4th colour = sigmoid(sem logit))
      density:           density >= 70                        (4th colour = sigmoid(density))
    """
    pts = outputs["point_location"].reshape(-1, 3)
    sem = outputs["semantics"].reshape(-1)
    lab = outputs["semantics_colormap"].reshape(-1).to(torch.float32)
    den = outputs["density"].reshape(-1)
    rgb = outputs["rgb"].reshape(-1, 3)
    m_sem, m_den, m_lab = sem >= 3, den >= 70, lab >= 0.999
    res = {}
    for name, m, fourth in (
        ("semantic_colormap", m_lab & m_den, sem),
        ("semantic", m_sem & m_den, sem),
        ("density", m_den, den),
    ):
        res[name] = {
            "points": pts[m],
            "colors": torch.hstack([rgb[m], torch.sigmoid(fourth[m]).unsqueeze(-1)]),
        }
    return res


def pointcloud_from_outputs(rb: RY.RayBundle, outputs: Dict[str, Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """``generate_point_cloud`` inner step (``export/exporter_utils_nerfacto.py:156-166``):
    point = o + d*depth, kept where semantics_colormap[:,0] > 0.  Returns (points, rgb, view_dirs)."""
    point = rb.origins + rb.directions * outputs["depth"]
    mask = outputs["semantics_colormap"][:, 0] > 0
    return point[mask], outputs["rgb"][mask], rb.directions[mask]
