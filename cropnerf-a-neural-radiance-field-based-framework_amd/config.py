"""Hyper-parameters of the fruit_nerf hot path (field / proposal nets / model), product side.

Defaults restate ``FruitNerfModelConfig`` (``fruit_nerf/fruit_nerf.py:59-68``), the nerfacto defaults it inherits
(SURVEY.md A.0), ``FruitField.__init__`` (``fruit_nerf/fruit_field.py:71-96``) and the three method specs in
``fruit_nerf/fruit_nerf_config.py:29-172``.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from .fruit_nerf.components.field_heads import SemanticFieldHead


@dataclass
class GridSpec:
    """nerfstudio ``HashEncoding`` hyper-parameters (``fruit_field.py:125-132``).

    ``layout`` selects which of the reference's two implementations the table follows (``FruitField``'s
    ``implementation`` argument, ``fruit_field.py:95``): ``"torch"`` = nerfstudio's torch ``HashEncoding`` (every level
    hashed, ``[L * 2^k, 2]``), ``"tcnn"`` = tiny-cuda-nn's ``GridEncoding`` geometry (dense coarse levels, +0.5 cell
    offset) in this library's table layout (``include/cropnerf_hip.h``: ``cn_tcnn_grid_plan``)."""

    num_levels: int = 16
    min_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    layout: str = "torch"

    @property
    def table_size(self) -> int:
        return 1 << self.log2_hashmap_size

    def growth(self) -> float:
        """nerfstudio ``HashEncoding.__init__``: exp((ln max_res - ln min_res) / (L - 1)) in float64."""
        if self.num_levels <= 1:
            return 1.0
        return float(np.exp((np.log(self.max_res) - np.log(self.min_res)) / (self.num_levels - 1)))

    def plan(self):
        """``cn_tcnn_grid_plan`` of this grid (tcnn layout only); computed by the library (host code, no GPU needed)."""
        if self.layout != "tcnn":
            raise ValueError("GridSpec.plan() is defined for layout='tcnn'")
        key = (self.num_levels, self.min_res, self.max_res, self.log2_hashmap_size)
        plan = _PLAN_CACHE.get(key)
        if plan is None:
            import ctypes as C

            from . import _lib as L

            plan = L.TcnnGridPlan()
            # tcnn reads "per_level_scale" from its JSON config as a float
            L.check(L.load().cn_tcnn_grid_plan_init(self.num_levels, self.log2_hashmap_size, self.min_res,
                                                    float(np.float32(self.growth())), C.byref(plan)))
            _PLAN_CACHE[key] = plan
        return plan

    @property
    def num_entries(self) -> int:
        """Rows of the hash table."""
        if self.layout == "tcnn":
            return int(self.plan().level_offset[self.num_levels])
        return self.table_size * self.num_levels

    @property
    def num_packed_entries(self) -> int:
        """Entries of tcnn's own parameter vector for this grid (``n_params / features_per_level``)."""
        return int(self.plan().packed_offset[self.num_levels])

    def scalings(self) -> List[float]:
        """torch layout: floor(min_res * growth**l).  Upstream evaluates ``np.float64 ** int64 Tensor`` through
        ``Tensor.__rpow__``, i.e. in float32 (default 16..2048 grid: last level 2047, not 2048); reproduced with
        the same expression.  tcnn layout: grid.h ``grid_scale`` (from the plan)."""
        if self.layout == "tcnn":
            p = self.plan()
            return [float(p.scalings[i]) for i in range(self.num_levels)]
        levels = torch.arange(self.num_levels)
        growth = self.growth() if self.num_levels > 1 else 1.0
        return torch.floor(self.min_res * np.float64(growth) ** levels).to(torch.float32).tolist()


_PLAN_CACHE: Dict[tuple, object] = {}


@dataclass
class FieldSpec:
    grid: GridSpec = field(default_factory=GridSpec)
    hidden_dim: int = 64
    geo_feat_dim: int = 15
    num_layers_semantic: int = 2
    hidden_dim_semantics: int = 64
    hidden_dim_transient: int = 64
    hidden_dim_color: int = 64
    num_layers_color: int = 3
    appearance_embedding_dim: int = 32
    num_images: int = 100
    use_average_appearance_embedding: bool = True
    sh_input: str = "unit"  # "unit" (tcnn semantics) or "shifted" ((d+1)/2, torch fallback as written)


@dataclass
class ProposalSpec:
    grid: GridSpec
    hidden_dim: int = 16


def default_proposal_specs() -> List[ProposalSpec]:
    return [
        ProposalSpec(GridSpec(num_levels=5, min_res=16, max_res=128, log2_hashmap_size=17)),
        ProposalSpec(GridSpec(num_levels=5, min_res=16, max_res=256, log2_hashmap_size=17)),
    ]


@dataclass
class FruitNerfModelConfig:
    """Same field names as the reference's ``FruitNerfModelConfig(NerfactoModelConfig)``."""

    semantic_loss_weight: float = 1.0
    pass_semantic_gradients: bool = False
    num_layers_semantic: int = 2
    hidden_dim_semantics: int = 64
    geo_feat_dim: int = 15
    # nerfacto
    near_plane: float = 0.05
    far_plane: float = 1000.0
    background_color: Union[str, Tuple[float, float, float]] = "last_sample"
    num_levels: int = 16
    base_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_nerf_samples_per_ray: int = 48
    num_proposal_iterations: int = 2
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256, "use_linear": False},
    ])
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    proposal_update_every: int = 5
    proposal_warmup: int = 5000
    disable_scene_contraction: bool = False
    use_average_appearance_embedding: bool = True
    interlevel_loss_mult: float = 1.0
    distortion_loss_mult: float = 0.002
    eval_num_rays_per_chunk: int = 1 << 15
    sh_input: str = "unit"
    # extension (not in the reference, which composites every sample): > 0 stops a ray in eval renders once its
    # transmittance falls below this value; 0 keeps the reference's behaviour
    early_stop_transmittance: float = 0.0
    # extension: the matrix arithmetic of the eval / export renders that fill the device (cn_render_opts.matrix_precision).
    # "split_bf16" (default since round 5): every operand of the MLP products as bf16 hi + bf16 lo (16 mantissa bits) on the
    # bf16 matrix pipe, fp32 accumulation -- held to the SAME parity bars against the fp32 oracle as the exact kernels
    # (tests/test_gpu_parity.py) and 1.5x faster; "fp32" = exact fp32 matrix products (v_mfma_f32_16x16x4_f32; what small
    # batches and TRAINING always use); "f16" = the reference's own arithmetic class (tcnn FullyFusedMLP under
    # mixed_precision=True, fruit_field.py:95, fruit_nerf_config.py:35): fp16 weights and layer inputs, fp32 accumulation, in
    # every eval render and -- with fp32 masters -- in the training iteration
    matrix_precision: str = "split_bf16"
    # Which of the reference's two implementations the parameters follow (nerfacto's ``implementation``; FruitField's own
    # default is "tcnn", fruit_field.py:95).  "torch": nerfstudio's torch HashEncoding / MLP (all levels hashed, biases).
    # "tcnn": tiny-cuda-nn's grid geometry (dense coarse levels, +0.5 offset), bias-free MLPs (tcnn_params.py) -- the
    # layout a reference-trained checkpoint has.  hash_table_dtype "float16" stores the tables as tcnn computes with
    # them (inference only; training keeps float32 masters).
    implementation: str = "torch"
    hash_table_dtype: str = "float32"

    def field_spec(self, num_images: int) -> FieldSpec:
        # FruitModel.populate_modules forwards only these (fruit_nerf.py:97-112); the rest stay FruitField defaults.
        return FieldSpec(
            grid=GridSpec(self.num_levels, 16, self.max_res, self.log2_hashmap_size, 2, self.implementation),
            geo_feat_dim=self.geo_feat_dim, num_layers_semantic=self.num_layers_semantic,
            hidden_dim_semantics=self.hidden_dim_semantics, num_images=num_images,
            use_average_appearance_embedding=self.use_average_appearance_embedding, sh_input=self.sh_input,
        )

    def proposal_specs(self) -> List[ProposalSpec]:
        out = []
        for i in range(self.num_proposal_iterations):
            a = self.proposal_net_args_list[min(i, len(self.proposal_net_args_list) - 1)]
            out.append(ProposalSpec(GridSpec(a["num_levels"], a.get("base_res", 16), a["max_res"],
                                             a["log2_hashmap_size"], a.get("features_per_level", 2),
                                             self.implementation),
                                    hidden_dim=a["hidden_dim"]))
        return out


def param_shapes(spec: FieldSpec, prop_specs: List[ProposalSpec]) -> Dict[str, Tuple[int, ...]]:
    """Logical state-dict names -> shapes (``nn.Linear`` layout [out,in]; hash tables [L*T, F])."""
    g = spec.grid
    shapes: Dict[str, Tuple[int, ...]] = {}
    shapes["field.mlp_base_grid.hash_table"] = (g.num_entries, g.features_per_level)

    def add_mlp(prefix, in_dim, num_layers, width, out_dim):
        dims = [in_dim] + [width] * (num_layers - 1) + [out_dim]
        for i in range(num_layers):
            shapes[f"{prefix}.layers.{i}.weight"] = (dims[i + 1], dims[i])
            shapes[f"{prefix}.layers.{i}.bias"] = (dims[i + 1],)

    add_mlp("field.mlp_base_mlp", g.num_levels * g.features_per_level, 2, spec.hidden_dim, 1 + spec.geo_feat_dim)
    add_mlp("field.mlp_semantics", spec.geo_feat_dim, spec.num_layers_semantic, spec.hidden_dim_semantics,
            spec.hidden_dim_transient)
    shapes.update(SemanticFieldHead(spec.hidden_dim_transient, 1).shapes())
    add_mlp("field.mlp_head", 16 + spec.geo_feat_dim + spec.appearance_embedding_dim, spec.num_layers_color,
            spec.hidden_dim_color, 3)
    shapes["field.embedding_appearance.embedding.weight"] = (spec.num_images, spec.appearance_embedding_dim)
    for i, ps in enumerate(prop_specs):
        pg = ps.grid
        shapes[f"proposal_networks.{i}.encoding.hash_table"] = (pg.num_entries, pg.features_per_level)
        add_mlp(f"proposal_networks.{i}.mlp", pg.num_levels * pg.features_per_level, 2, ps.hidden_dim, 1)
    shapes["camera_optimizer.pose_adjustment"] = (spec.num_images, 6)
    return shapes


def init_params(spec: FieldSpec, prop_specs: List[ProposalSpec], seed: int = 0, grid_scale: float = 1e-3,
                device: Union[str, torch.device] = "cpu") -> Dict[str, torch.Tensor]:
    """Random initialisation: hash tables U(-1,1)*grid_scale (reference: 1e-3), Linear layers the ``nn.Linear``
    default (Kaiming-uniform a=sqrt 5 == U(+-1/sqrt(fan_in)) for weight and bias), embedding N(0,1), pose 0.
    tcnn layout: the biases a tcnn module cannot hold start (and are kept) at zero, and the caller ties the alias
    entries of the tables (``ops.tcnn_grid_tie_parameters``; ``FruitModel`` does)."""
    gen = torch.Generator().manual_seed(seed)
    shapes = param_shapes(spec, prop_specs)
    frozen = set()
    if spec.grid.layout == "tcnn":
        from .fruit_nerf.tcnn_params import frozen_parameter_names

        frozen = set(frozen_parameter_names(spec, prop_specs))
    out: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        if name in frozen:
            out[name] = torch.zeros(shape, device=device)
            continue
        if name.endswith("hash_table"):
            t = (torch.rand(shape, generator=gen) * 2 - 1) * grid_scale
        elif name.endswith("embedding.weight"):
            t = torch.randn(shape, generator=gen)
        elif name.endswith("pose_adjustment"):
            t = torch.zeros(shape)
        else:
            fan_in = shape[1] if name.endswith(".weight") else shapes[name[:-4] + "weight"][1]
            t = (torch.rand(shape, generator=gen) * 2 - 1) / math.sqrt(fan_in)
        out[name] = t.to(device)
    return out
