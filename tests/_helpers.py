"""Shared test scaffolding: one seeded scene (parameters + cameras + rays) for the oracle and the HIP path."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from oracle import field as OF
from oracle import model as OM
from oracle import rays as ORY


@dataclass
class Scene:
    params: Dict[str, torch.Tensor]  # CPU fp32
    fspec: OF.FieldSpec
    pspecs: List[OF.ProposalSpec]
    aabb: torch.Tensor  # [2,3]
    c2w: torch.Tensor  # [N,3,4]
    intr: torch.Tensor  # [N,4]
    height: int
    width: int


def make_scene(seed: int = 0, log2_T: int = 19, num_images: int = 6, height: int = 40, width: int = 40,
               focal: float = 55.0, grid_scale: float = 0.1, prop_log2_T: int = 17) -> Scene:
    from cropnerf_amd import synthetic

    fspec = OF.FieldSpec(grid=OF.GridSpec(log2_hashmap_size=log2_T), num_images=num_images)
    pspecs = [OF.ProposalSpec(OF.GridSpec(5, 16, 128, prop_log2_T)), OF.ProposalSpec(OF.GridSpec(5, 16, 256, prop_log2_T))]
    params = OF.random_params(fspec, pspecs, seed=seed, grid_scale=grid_scale)
    # a non-trivial pose refinement so the SO3xR3 path is exercised
    g = torch.Generator().manual_seed(seed + 99)
    params["camera_optimizer.pose_adjustment"] = (torch.rand(num_images, 6, generator=g) - 0.5) * 0.02
    c2w, intr = synthetic.orbit_cameras(num_images, height=height, width=width, focal=focal)
    aabb = torch.tensor(synthetic.SCENE_AABB, dtype=torch.float32)
    return Scene(params, fspec, pspecs, aabb, c2w, intr, height, width)


def make_tcnn_scene(seed: int = 0, log2_T: int = 19, num_images: int = 6, height: int = 40, width: int = 40,
                    focal: float = 55.0, grid_scale: float = 0.1, prop_log2_T: int = 17,
                    second_prop=(5, 256)) -> Scene:
    """The same scene shape with the reference's default implementation: tcnn modules, parameters under nerfstudio's
    tcnn state-dict names (fp32 master values, as a checkpoint holds them)."""
    from cropnerf_amd import synthetic
    from oracle import tcnn as TC

    fspec = OF.FieldSpec(grid=OF.GridSpec(log2_hashmap_size=log2_T), num_images=num_images, implementation="tcnn")
    pspecs = [OF.ProposalSpec(OF.GridSpec(5, 16, 128, prop_log2_T), implementation="tcnn"),
              OF.ProposalSpec(OF.GridSpec(second_prop[0], 16, second_prop[1], prop_log2_T), implementation="tcnn")]
    params = TC.random_params(fspec, pspecs, seed=seed, grid_scale=grid_scale)
    g = torch.Generator().manual_seed(seed + 99)
    params["camera_optimizer.pose_adjustment"] = (torch.rand(num_images, 6, generator=g) - 0.5) * 0.02
    c2w, intr = synthetic.orbit_cameras(num_images, height=height, width=width, focal=focal)
    aabb = torch.tensor(synthetic.SCENE_AABB, dtype=torch.float32)
    return Scene(params, fspec, pspecs, aabb, c2w, intr, height, width)


def oracle_model(sc: Scene, test_mode: str = "test", **cfg) -> OM.OracleModel:
    config = OM.ModelConfig(field=sc.fspec, proposals=sc.pspecs, **cfg)
    return OM.OracleModel(sc.params, config, sc.aabb, test_mode=test_mode)


def to_dev(t, device="cuda"):
    if t is None:
        return None
    return t.to(device).contiguous()


def dev_params(sc: Scene, device="cuda", table_dtype=None):
    """Device parameters of the product.  A tcnn scene goes through the importer (tcnn state dict -> nn.Linear-shaped
    weights + packed hash tables, fp16 by default: the values tcnn computes with)."""
    if sc.fspec.implementation == "tcnn":
        from cropnerf_amd.fruit_nerf import tcnn_params

        fspec, pspecs = product_specs(sc)
        return tcnn_params.from_tcnn_state_dict(sc.params, fspec, pspecs, device,
                                                table_dtype or torch.float16)
    return {k: v.to(device).contiguous() for k, v in sc.params.items()}


def product_specs(sc: Scene):
    """The product-side spec objects equivalent to the oracle's."""
    from cropnerf_amd import config as PC

    g = sc.fspec.grid
    layout = sc.fspec.implementation
    fspec = PC.FieldSpec(grid=PC.GridSpec(g.num_levels, g.min_res, g.max_res, g.log2_hashmap_size, 2, layout),
                         geo_feat_dim=sc.fspec.geo_feat_dim, num_layers_semantic=sc.fspec.num_layers_semantic,
                         hidden_dim_semantics=sc.fspec.hidden_dim_semantics, num_images=sc.fspec.num_images,
                         sh_input=sc.fspec.sh_input)
    pspecs = [PC.ProposalSpec(PC.GridSpec(p.grid.num_levels, p.grid.min_res, p.grid.max_res,
                                          p.grid.log2_hashmap_size, 2, layout), p.hidden_dim) for p in sc.pspecs]
    return fspec, pspecs


def rays_with_box(sc: Scene, cam: int = 0, n: Optional[int] = None) -> ORY.RayBundle:
    """Full-image rays of camera `cam` with near/far from the scene box (all samples inside the grid)."""
    rb = ORY.image_rays(sc.c2w, sc.intr, cam, sc.height, sc.width)
    rb = ORY.with_aabb_near_far(rb, sc.aabb.reshape(-1))
    if n is not None:
        rb = rb.slice(0, n)
    return rb


def assert_close(a: torch.Tensor, b: torch.Tensor, rtol: float, atol: float, name: str, frac_ok: float = 1.0):
    a = a.detach().cpu().float()
    b = b.detach().cpu().float()
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    frac = 1.0 - bad.float().mean().item()
    assert frac >= frac_ok, (f"{name}: {bad.sum().item()} / {bad.numel()} outside rtol={rtol} atol={atol}; "
                             f"max abs err {err.max().item():.3e}, worst ref {b.flatten()[err.flatten().argmax()].item():.6g}")
