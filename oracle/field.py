"""Oracle: the FruitField (hash grid + tiny MLPs) and the proposal density fields.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Wiring follows ``fruit_nerf/fruit_field.py:71-302``
(module graph ``:109-167``, ``get_density :169-194``, ``get_outputs :235-282``,
``get_inference_outputs :196-233``, ``forward :284-302``) and ``fruit_nerf/fruit_nerf.py:118-142``
(proposal networks).  Arithmetic restates the nerfstudio 1.1.3 *torch* implementations
(SURVEY.md A.5-A.7): ``HashEncoding.pytorch_fwd``, ``MLP``, ``SHEncoding`` /
``components_from_spherical_harmonics``, ``SceneContraction(order=inf)``, ``trunc_exp``, ``Embedding.mean``.

Parameters live in a flat ``dict[str, Tensor]`` whose keys are the logical state-dict names
(``field.mlp_base_grid.hash_table``, ``field.mlp_base_mlp.layers.0.weight`` ...), ``nn.Linear`` layout
``[out, in]``.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field as dc_field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

HASH_PRIMES = (1, 2654435761, 805459861)


@dataclass
class GridSpec:
    """``HashEncoding`` hyper-parameters (``fruit_field.py:125-132``)."""

    num_levels: int = 16
    min_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2

    @property
    def table_size(self) -> int:
        return 2 ** self.log2_hashmap_size

    def scalings(self) -> Tensor:
        """``floor(min_res * growth**l)``, growth = exp((ln max - ln min)/(L-1)) (numpy float64 then floor)."""
        levels = torch.arange(self.num_levels)
        growth = (
            np.exp((np.log(self.max_res) - np.log(self.min_res)) / (self.num_levels - 1))
            if self.num_levels > 1 else 1.0
        )
        return torch.floor(self.min_res * growth ** levels).to(torch.float32)


@dataclass
class FieldSpec:
    """``FruitField.__init__`` arguments that ``FruitModel.populate_modules`` can reach
    (``fruit_nerf.py:97-112``); the rest stay at ``FruitField`` defaults (``fruit_field.py:75-95``)."""

    grid: GridSpec = dc_field(default_factory=GridSpec)
    hidden_dim: int = 64
    geo_feat_dim: int = 15
    num_layers_semantic: int = 2
    hidden_dim_semantics: int = 64
    hidden_dim_transient: int = 64  # out_dim of mlp_semantics
    hidden_dim_color: int = 64
    num_layers_color: int = 3
    appearance_embedding_dim: int = 32
    num_images: int = 100
    use_average_appearance_embedding: bool = True
    sh_input: str = "unit"  # "unit": SH of the unit direction (tcnn semantics); "shifted": SH of (d+1)/2
    # which of FruitField's two implementations (fruit_field.py:95): "torch" = nerfstudio's torch modules (this file),
    # "tcnn" = tiny-cuda-nn's grid / fully-fused MLPs / SH (oracle/tcnn.py; parameter names *.tcnn_encoding.params)
    implementation: str = "torch"
    tcnn_half_activations: bool = False  # diagnostic: round activations to fp16 as tcnn's kernels do


@dataclass
class ProposalSpec:
    """``HashMLPDensityField`` args from nerfacto ``proposal_net_args_list``."""

    grid: GridSpec
    hidden_dim: int = 16
    implementation: str = "torch"
    tcnn_half_activations: bool = False


def default_proposal_specs() -> List[ProposalSpec]:
    return [
        ProposalSpec(GridSpec(num_levels=5, min_res=16, max_res=128, log2_hashmap_size=17)),
        ProposalSpec(GridSpec(num_levels=5, min_res=16, max_res=256, log2_hashmap_size=17)),
    ]


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------

def hash_fn(corner: Tensor, spec: GridSpec) -> Tensor:
    """``HashEncoding.hash_fn``: int64 products with (1, 2654435761, 805459861), xor, mod T, + l*T.
    corner: [..., L, 3] int32."""
    t = corner.to(torch.int64) * torch.tensor(HASH_PRIMES, dtype=torch.int64)
    x = torch.bitwise_xor(t[..., 0], t[..., 1])
    x = torch.bitwise_xor(x, t[..., 2])
    x = x % spec.table_size
    x = x + torch.arange(spec.num_levels, dtype=torch.int64) * spec.table_size
    return x


def hash_grid(x: Tensor, table: Tensor, spec: GridSpec) -> Tensor:
    """``HashEncoding.pytorch_fwd`` (SURVEY.md A.6 variant (i)). x [...,3] in [0,1] -> [..., L*F]."""
    x = x[..., None, :]
    scaled = x * spec.scalings().view(-1, 1)
    c = torch.ceil(scaled).to(torch.int32)
    f = torch.floor(scaled).to(torch.int32)
    off = scaled - f

    def pick(ax, ay, az):
        return torch.cat([ax[..., 0:1], ay[..., 1:2], az[..., 2:3]], dim=-1)

    h0 = hash_fn(c, spec)
    h1 = hash_fn(pick(c, f, c), spec)
    h2 = hash_fn(pick(f, f, c), spec)
    h3 = hash_fn(pick(f, c, c), spec)
    h4 = hash_fn(pick(c, c, f), spec)
    h5 = hash_fn(pick(c, f, f), spec)
    h6 = hash_fn(f, spec)
    h7 = hash_fn(pick(f, c, f), spec)
    f0, f1, f2, f3, f4, f5, f6, f7 = (table[h] for h in (h0, h1, h2, h3, h4, h5, h6, h7))
    ox, oy, oz = off[..., 0:1], off[..., 1:2], off[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return torch.flatten(enc, start_dim=-2, end_dim=-1)


def mlp(x: Tensor, params: Dict[str, Tensor], prefix: str, num_layers: int, out_activation: Optional[str] = None) -> Tensor:
    """nerfstudio torch ``MLP``: Linear+ReLU hidden layers, last Linear, optional out activation."""
    for i in range(num_layers):
        x = torch.nn.functional.linear(x, params[f"{prefix}.layers.{i}.weight"], params[f"{prefix}.layers.{i}.bias"])
        if i < num_layers - 1:
            x = torch.relu(x)
    if out_activation == "sigmoid":
        x = torch.sigmoid(x)
    return x


def contract_inf(x: Tensor) -> Tensor:
    """``SceneContraction(order=inf)`` (``fruit_nerf.py:94``): x if |x|inf < 1 else (2 - 1/m) x/m."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def sh_deg4(d: Tensor) -> Tensor:
    """nerfstudio ``components_from_spherical_harmonics(levels=4)`` -> [...,16]."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x * x, y * y, z * z
    c = torch.zeros((*d.shape[:-1], 16), dtype=d.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = 0.4886025119029199 * y
    c[..., 2] = 0.4886025119029199 * z
    c[..., 3] = 0.4886025119029199 * x
    c[..., 4] = 1.0925484305920792 * x * y
    c[..., 5] = 1.0925484305920792 * y * z
    c[..., 6] = 0.9461746957575601 * zz - 0.31539156525251999
    c[..., 7] = 1.0925484305920792 * x * z
    c[..., 8] = 0.5462742152960396 * (xx - yy)
    c[..., 9] = 0.5900435899266435 * y * (3 * xx - yy)
    c[..., 10] = 2.890611442640554 * x * y * z
    c[..., 11] = 0.4570457994644658 * y * (5 * zz - 1)
    c[..., 12] = 0.3731763325901154 * z * (5 * zz - 3)
    c[..., 13] = 0.4570457994644658 * x * (5 * zz - 1)
    c[..., 14] = 1.445305721320277 * z * (xx - yy)
    c[..., 15] = 0.5900435899266435 * x * (xx - 3 * yy)
    return c


def normalized_positions(positions: Tensor, aabb: Tensor, contraction: bool) -> Tuple[Tensor, Tensor]:
    """``get_density`` head (``fruit_field.py:171-180``): contraction then (p+2)/4, or AABB-normalise; selector
    = all(0<p<1); positions zeroed where deselected."""
    if contraction:
        p = (contract_inf(positions) + 2.0) / 4.0
    else:
        p = (positions - aabb[0]) / (aabb[1] - aabb[0])
    selector = ((p > 0.0) & (p < 1.0)).all(dim=-1)
    return p * selector[..., None], selector


# ----------------------------------------------------------------------------------------------
# FruitField
# ----------------------------------------------------------------------------------------------

def semantics_from_geo(geo_flat: Tensor, params: Dict[str, Tensor], spec: "FieldSpec") -> Tensor:
    """``mlp_semantics`` -> ``field_head_semantics`` (``fruit_field.py:146-157,264-269``) on [N, geo] -> [N, 1]."""
    if spec.implementation == "tcnn":
        from . import tcnn as TC

        x = TC.network(geo_flat, params["field.mlp_semantics.tcnn_encoding.params"], spec.geo_feat_dim,
                       spec.hidden_dim_transient, spec.hidden_dim_semantics, spec.num_layers_semantic - 1,
                       half_activations=spec.tcnn_half_activations)
    else:
        x = mlp(geo_flat, params, "field.mlp_semantics", spec.num_layers_semantic)
    return torch.nn.functional.linear(x, params["field.field_head_semantics.net.weight"],
                                      params["field.field_head_semantics.net.bias"])


def field_density(positions: Tensor, params: Dict[str, Tensor], spec: FieldSpec, aabb: Tensor,
                  contraction: bool) -> Tuple[Tensor, Tensor]:
    """``FruitField.get_density`` (``fruit_field.py:169-194``) -> density [...,1], geo features [...,geo]."""
    p, selector = normalized_positions(positions, aabb, contraction)
    if spec.implementation == "tcnn":
        from . import tcnn as TC

        g = TC.grid_spec_of(spec.grid)
        ha = spec.tcnn_half_activations
        enc = TC.hash_grid(p.reshape(-1, 3), params["field.mlp_base_grid.tcnn_encoding.params"], g, half_activations=ha)
        h = TC.network(enc, params["field.mlp_base_mlp.tcnn_encoding.params"], g.out_dim, 1 + spec.geo_feat_dim,
                       spec.hidden_dim, 1, half_activations=ha).view(*positions.shape[:-1], -1)
    else:
        enc = hash_grid(p.reshape(-1, 3), params["field.mlp_base_grid.hash_table"], spec.grid)
        h = mlp(enc, params, "field.mlp_base_mlp", 2).view(*positions.shape[:-1], -1)
    dba, geo = torch.split(h, [1, spec.geo_feat_dim], dim=-1)
    density = torch.exp(dba) * selector[..., None]  # trunc_exp forward = exp
    return density, geo


def field_forward(
    positions: Tensor,  # [R,S,3]
    directions: Tensor,  # [R,3]
    camera_indices: Optional[Tensor],  # [R,1]
    params: Dict[str, Tensor],
    spec: FieldSpec,
    aabb: Tensor,
    contraction: bool,
    test_mode: str,  # "val" | "test" | "inference" | "export"
    training: bool = False,
) -> Dict[str, Tensor]:
    """``FruitField.forward`` (``fruit_field.py:284-302``): density, rgb [R,S,3], semantics logit [R,S,1]."""
    R, S = positions.shape[:2]
    density, geo = field_density(positions, params, spec, aabb, contraction)

    d = directions[:, None, :].expand(R, S, 3)
    shifted = (d + 1.0) / 2.0  # shift_directions_for_tcnn (fruit_field.py:209,244)
    tcnn_impl = spec.implementation == "tcnn"
    if tcnn_impl:
        from . import tcnn as TC

        sh = TC.sh_deg4(shifted.reshape(-1, 3))  # tcnn maps its [0,1] input back to [-1,1] itself
    else:
        sh_in = shifted * 2.0 - 1.0 if spec.sh_input == "unit" else shifted
        sh = sh_deg4(sh_in.reshape(-1, 3))

    emb = params["field.embedding_appearance.embedding.weight"]
    if test_mode in ("inference", "export"):
        app = torch.ones(R * S, spec.appearance_embedding_dim) * emb.mean(dim=0)  # fruit_field.py:218-220
    elif training:
        if camera_indices is None:
            raise AttributeError("Camera indices are not provided.")  # fruit_field.py:241-242
        app = emb[camera_indices[:, 0]][:, None, :].expand(R, S, -1).reshape(R * S, -1)
    elif spec.use_average_appearance_embedding:
        if camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        app = torch.ones(R * S, spec.appearance_embedding_dim) * emb.mean(dim=0)  # fruit_field.py:254-257
    else:
        app = torch.zeros(R * S, spec.appearance_embedding_dim)

    geo_flat = geo.reshape(-1, spec.geo_feat_dim)
    if tcnn_impl:
        ha = spec.tcnn_half_activations
        sem = semantics_from_geo(geo_flat, params, spec).view(R, S, -1)
        h = torch.cat([sh, geo_flat, app], dim=-1)
        rgb = TC.network(h, params["field.mlp_head.tcnn_encoding.params"], h.shape[-1], 3, spec.hidden_dim_color,
                         spec.num_layers_color - 1, "sigmoid", half_activations=ha).view(R, S, 3)
        return {"density": density, "rgb": rgb, "semantics": sem}
    sem = semantics_from_geo(geo_flat, params, spec).view(R, S, -1)

    h = torch.cat([sh, geo_flat, app], dim=-1)
    rgb = mlp(h, params, "field.mlp_head", spec.num_layers_color, out_activation="sigmoid").view(R, S, 3)
    return {"density": density, "rgb": rgb, "semantics": sem}


def proposal_density(positions: Tensor, params: Dict[str, Tensor], level: int, spec: ProposalSpec, aabb: Tensor,
                     contraction: bool = True) -> Tensor:
    """Upstream ``HashMLPDensityField.density_fn`` as built at ``fruit_nerf.py:133-142`` -> [R,S,1]."""
    p, selector = normalized_positions(positions, aabb, contraction)
    if spec.implementation == "tcnn":
        from . import tcnn as TC

        dba = TC.network_with_grid(p.reshape(-1, 3), params[f"proposal_networks.{level}.mlp_base.tcnn_encoding.params"],
                                   TC.grid_spec_of(spec.grid), 1, spec.hidden_dim, 1,
                                   half_activations=spec.tcnn_half_activations).view(*positions.shape[:-1], -1)
        return torch.exp(dba) * selector[..., None]
    pre = f"proposal_networks.{level}"
    enc = hash_grid(p.reshape(-1, 3), params[f"{pre}.encoding.hash_table"], spec.grid)
    dba = mlp(enc, params, f"{pre}.mlp", 2).view(*positions.shape[:-1], -1)
    return torch.exp(dba) * selector[..., None]


# ----------------------------------------------------------------------------------------------
# parameter construction (shapes only; values are the caller's business)
# ----------------------------------------------------------------------------------------------

def param_shapes(spec: FieldSpec, prop_specs: List[ProposalSpec]) -> Dict[str, Tuple[int, ...]]:
    g = spec.grid
    shapes: Dict[str, Tuple[int, ...]] = {}
    shapes["field.mlp_base_grid.hash_table"] = (g.table_size * g.num_levels, g.features_per_level)
    enc_dim = g.num_levels * g.features_per_level

    def add_mlp(prefix, in_dim, num_layers, width, out_dim):
        dims = [in_dim] + [width] * (num_layers - 1) + [out_dim]
        for i in range(num_layers):
            shapes[f"{prefix}.layers.{i}.weight"] = (dims[i + 1], dims[i])
            shapes[f"{prefix}.layers.{i}.bias"] = (dims[i + 1],)

    add_mlp("field.mlp_base_mlp", enc_dim, 2, spec.hidden_dim, 1 + spec.geo_feat_dim)
    add_mlp("field.mlp_semantics", spec.geo_feat_dim, spec.num_layers_semantic, spec.hidden_dim_semantics,
            spec.hidden_dim_transient)
    shapes["field.field_head_semantics.net.weight"] = (1, spec.hidden_dim_transient)
    shapes["field.field_head_semantics.net.bias"] = (1,)
    add_mlp("field.mlp_head", 16 + spec.geo_feat_dim + spec.appearance_embedding_dim, spec.num_layers_color,
            spec.hidden_dim_color, 3)
    shapes["field.embedding_appearance.embedding.weight"] = (spec.num_images, spec.appearance_embedding_dim)
    for i, ps in enumerate(prop_specs):
        pg = ps.grid
        shapes[f"proposal_networks.{i}.encoding.hash_table"] = (pg.table_size * pg.num_levels, pg.features_per_level)
        add_mlp(f"proposal_networks.{i}.mlp", pg.num_levels * pg.features_per_level, 2, ps.hidden_dim, 1)
    shapes["camera_optimizer.pose_adjustment"] = (spec.num_images, 6)
    return shapes


def random_params(spec: FieldSpec, prop_specs: List[ProposalSpec], seed: int = 0, grid_scale: float = 0.1,
                  reference_init: bool = False) -> Dict[str, Tensor]:
    """Parameter set *P-rand* of SURVEY.md 8(d): grid ~ U(-1,1)*grid_scale, Linear = Kaiming-uniform(a=sqrt 5)
    (``nn.Linear`` default), embedding N(0,1), pose adjustment 0.  ``reference_init`` uses the reference's grid
    init scale 1e-3 instead."""
    gen = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    scale = 1e-3 if reference_init else grid_scale
    for name, shape in param_shapes(spec, prop_specs).items():
        if name.endswith("hash_table"):
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) * scale
        elif name.endswith("embedding.weight"):
            out[name] = torch.randn(shape, generator=gen)
        elif name.endswith("pose_adjustment"):
            out[name] = torch.zeros(shape)
        elif name.endswith(".weight"):
            bound = 1.0 / math.sqrt(shape[1])
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        elif name.endswith(".bias"):
            fan_in = param_shapes(spec, prop_specs)[name[:-4] + "weight"][1]
            bound = 1.0 / math.sqrt(fan_in)
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        else:
            raise KeyError(name)
    return out
