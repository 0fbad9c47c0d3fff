"""tcnn-packed parameters <-> this library's parameters.

The reference builds ``FruitField`` and the proposal networks with ``implementation="tcnn"``
(``crop_nerf/fruit_nerf/fruit_field.py:95,116-167``; ``fruit_nerf.py:118-142`` through nerfstudio's
``HashMLPDensityField``), so the state dict of a reference-trained model holds

    field.mlp_base_grid.tcnn_encoding.params        tcnn GridEncoding parameter vector
    field.mlp_base_mlp.tcnn_encoding.params         tcnn.Network (FullyFusedMLP) 32 -> 64 -> 16
    field.mlp_semantics.tcnn_encoding.params        tcnn.Network 15 -> 64 -> 64
    field.mlp_head.tcnn_encoding.params             tcnn.Network 63 -> 64 -> 64 -> 3 (sigmoid)
    field.field_head_semantics.net.{weight,bias}    nn.Linear(64, 1)
    field.embedding_appearance.embedding.weight     nn.Embedding
    proposal_networks.N.mlp_base.tcnn_encoding.params   tcnn.NetworkWithInputEncoding: [MLP | grid]
    camera_optimizer.pose_adjustment

as float32 *master* copies that tcnn casts to fp16 for every forward pass.  The kernels here take ``nn.Linear``-shaped
weights with biases and a hash table in the layout of ``cn_tcnn_grid_plan``; this module converts, both ways:

* grid: ``ops.tcnn_grid_pack`` / ``ops.tcnn_grid_unpack`` (device kernels of the C ABI);
* FullyFusedMLP: matrices ``[width, pad16(in)]``, ``[width, width]`` ..., ``[pad16(out), width]`` row-major, no biases.
  ``tcnn.Network`` feeds the MLP through an ``Identity`` encoding that pads the input to a multiple of 16 with ONES, so
  the padded columns of the first matrix act as a bias: bias_0 = sum of those columns.  A grid encoding pads with ZEROS
  (the proposal networks' 10 -> 16): those columns are dead.  Padded output rows are dropped.
* SH: tcnn's real spherical harmonics carry the Condon-Shortley phase (components 1, 3, 5, ... have the opposite sign of
  nerfstudio's torch ``components_from_spherical_harmonics``, which is what the kernels evaluate): the sign is folded
  into the matching input columns of ``mlp_head``'s first matrix.

Algorithm source: tiny-cuda-nn (``encodings/grid.h``, ``encodings/identity.h``, ``encodings/spherical_harmonics.h``,
``networks/fully_fused_mlp.cu``, ``network_with_input_encoding.h``), restated in ``oracle/tcnn.py`` -- tcnn itself is
neither in ``/root/reference`` nor in this image.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from ..config import FieldSpec, ProposalSpec
from .components.field_heads import SemanticFieldHead

MLP_ALIGN = 16
SH_FLIPPED = (1, 3, 5, 7, 9, 11, 13, 15)  # components whose sign differs between tcnn and nerfstudio's torch SH

TCNN_FIELD_KEYS = ("field.mlp_base_grid.tcnn_encoding.params", "field.mlp_base_mlp.tcnn_encoding.params",
                   "field.mlp_semantics.tcnn_encoding.params", "field.mlp_head.tcnn_encoding.params")


def _pad(n: int) -> int:
    return (n + MLP_ALIGN - 1) // MLP_ALIGN * MLP_ALIGN


def mlp_param_count(n_in: int, n_out: int, width: int, n_hidden: int) -> int:
    return width * _pad(n_in) + (n_hidden - 1) * width * width + _pad(n_out) * width


def _matrices(params: Tensor, n_in: int, n_out: int, width: int, n_hidden: int) -> List[Tensor]:
    shapes = [(width, _pad(n_in))] + [(width, width)] * (n_hidden - 1) + [(_pad(n_out), width)]
    want = sum(r * c for r, c in shapes)
    if params.numel() != want:
        raise ValueError(f"FullyFusedMLP {n_in}->{width}x{n_hidden}->{n_out}: {params.numel()} parameters, expected {want}")
    out, o = [], 0
    for r, c in shapes:
        out.append(params[o:o + r * c].reshape(r, c))
        o += r * c
    return out


def mlp_to_linear(params: Tensor, n_in: int, n_out: int, width: int, n_hidden: int, input_pad_value: float = 1.0,
                  half: bool = True) -> List[Tuple[Tensor, Tensor]]:
    """tcnn FullyFusedMLP parameter vector -> [(weight [out,in], bias [out])] of the equivalent ``nn.Linear`` stack.
    ``half``: round to fp16 first (the values tcnn computes with)."""
    p = params.detach().to(torch.float32)
    if half:
        p = p.to(torch.float16).to(torch.float32)
    mats = _matrices(p, n_in, n_out, width, n_hidden)
    layers: List[Tuple[Tensor, Tensor]] = []
    for i, m in enumerate(mats):
        first, last = i == 0, i == len(mats) - 1
        w = m[:, :n_in] if first else m
        b = m[:, n_in:].sum(dim=1) * input_pad_value if first else torch.zeros(m.shape[0])
        if last:
            w, b = w[:n_out], b[:n_out]
        layers.append((w.contiguous(), b.contiguous()))
    return layers


def linear_to_mlp(layers: Sequence[Tuple[Tensor, Tensor]], n_in: int, n_out: int, width: int, n_hidden: int,
                  input_pad_value: float = 1.0) -> Tensor:
    """The inverse.  Biases other than a first-layer bias behind a padded input cannot be represented (tcnn's MLPs have
    none) and raise."""
    assert len(layers) == n_hidden + 1
    out = []
    for i, (w, b) in enumerate(layers):
        first, last = i == 0, i == len(layers) - 1
        rows = _pad(n_out) if last else width
        cols = _pad(n_in) if first else width
        m = torch.zeros(rows, cols, dtype=torch.float32)
        m[:w.shape[0], :w.shape[1]] = w.detach().cpu().to(torch.float32)
        b = b.detach().cpu().to(torch.float32)
        if first and cols > n_in and input_pad_value != 0.0:
            m[:w.shape[0], n_in] = b / input_pad_value
        elif bool((b != 0).any()):
            raise ValueError(f"layer {i}: a non-zero bias has no place in a tcnn FullyFusedMLP "
                             f"(input width {n_in if first else width} is not padded)")
        out.append(m.reshape(-1))
    return torch.cat(out)


def is_tcnn_state_dict(state: Dict[str, Tensor]) -> bool:
    return any(k.endswith("tcnn_encoding.params") for k in state)


def _field_dims(fs: FieldSpec):
    enc = fs.grid.num_levels * fs.grid.features_per_level
    head_in = 16 + fs.geo_feat_dim + fs.appearance_embedding_dim
    return enc, head_in


def from_tcnn_state_dict(state: Dict[str, Tensor], field_spec: FieldSpec, prop_specs: Sequence[ProposalSpec],
                         device, table_dtype: torch.dtype = torch.float16, round_to_half: Optional[bool] = None
                         ) -> Dict[str, Tensor]:
    """nerfstudio state dict of a tcnn-built ``FruitModel`` (keys without the ``_model.`` prefix) -> the parameter dict
    ``FruitModel`` / ``ops.FieldHandle`` take (``config.param_shapes`` names, specs with ``grid.layout == "tcnn"``).
    Hash tables are packed on the device.  ``round_to_half`` (default: exactly when the tables are float16, i.e. for
    inference): every tcnn parameter is rounded to fp16 first -- the values tcnn's kernels compute with; False keeps the
    float32 master values (resuming training)."""
    from .. import ops

    if round_to_half is None:
        round_to_half = table_dtype == torch.float16
    half = bool(round_to_half)

    if field_spec.grid.layout != "tcnn" or any(p.grid.layout != "tcnn" for p in prop_specs):
        raise ValueError("from_tcnn_state_dict needs specs with grid.layout == 'tcnn'")
    fs = field_spec
    enc, head_in = _field_dims(fs)
    out: Dict[str, Tensor] = {}

    def put_mlp(prefix: str, layers):
        for i, (w, b) in enumerate(layers):
            out[f"{prefix}.layers.{i}.weight"] = w.to(device).contiguous()
            out[f"{prefix}.layers.{i}.bias"] = b.to(device).contiguous()

    def packed(t: Tensor) -> Tensor:
        t = t.detach().to(device=device, dtype=torch.float32)
        if half:
            t = t.to(torch.float16).to(torch.float32)
        return t.contiguous()

    out["field.mlp_base_grid.hash_table"] = ops.tcnn_grid_pack(
        fs.grid, packed(state["field.mlp_base_grid.tcnn_encoding.params"]), table_dtype)
    put_mlp("field.mlp_base_mlp", mlp_to_linear(state["field.mlp_base_mlp.tcnn_encoding.params"], enc,
                                                1 + fs.geo_feat_dim, fs.hidden_dim, 1, half=half))
    put_mlp("field.mlp_semantics", mlp_to_linear(state["field.mlp_semantics.tcnn_encoding.params"], fs.geo_feat_dim,
                                                 fs.hidden_dim_transient, fs.hidden_dim_semantics,
                                                 fs.num_layers_semantic - 1, half=half))
    head = mlp_to_linear(state["field.mlp_head.tcnn_encoding.params"], head_in, 3, fs.hidden_dim_color,
                         fs.num_layers_color - 1, half=half)
    w0 = head[0][0].clone()
    w0[:, list(SH_FLIPPED)] *= -1.0  # tcnn SH sign convention -> the kernels' (nerfstudio torch) convention
    head[0] = (w0, head[0][1])
    put_mlp("field.mlp_head", head)
    for k in SemanticFieldHead.keys + ("field.embedding_appearance.embedding.weight",):
        out[k] = state[k].detach().to(device=device, dtype=torch.float32).contiguous()
    for i, ps in enumerate(prop_specs):
        p = state[f"proposal_networks.{i}.mlp_base.tcnn_encoding.params"]
        pin = ps.grid.num_levels * ps.grid.features_per_level
        n_mlp = mlp_param_count(pin, 1, ps.hidden_dim, 1)
        put_mlp(f"proposal_networks.{i}.mlp", mlp_to_linear(p[:n_mlp], pin, 1, ps.hidden_dim, 1, input_pad_value=0.0,
                                                            half=half))
        out[f"proposal_networks.{i}.encoding.hash_table"] = ops.tcnn_grid_pack(ps.grid, packed(p[n_mlp:]), table_dtype)
    pose = state.get("camera_optimizer.pose_adjustment")
    out["camera_optimizer.pose_adjustment"] = (
        pose.detach().to(device=device, dtype=torch.float32).contiguous() if pose is not None
        else torch.zeros(fs.num_images, 6, device=device))
    return out


def to_tcnn_state_dict(params: Dict[str, Tensor], field_spec: FieldSpec, prop_specs: Sequence[ProposalSpec]
                       ) -> Dict[str, Tensor]:
    """The inverse of ``from_tcnn_state_dict``: float32 CPU tensors under nerfstudio's tcnn names (what
    ``ns-export`` / ``eval_setup`` of the reference would load).  Alias entries of the tables must be tied
    (``ops.tcnn_grid_tie_parameters``) -- the trainer does that after every step."""
    from .. import ops

    fs = field_spec
    enc, head_in = _field_dims(fs)
    out: Dict[str, Tensor] = {}

    def get_mlp(prefix: str, n: int):
        return [(params[f"{prefix}.layers.{i}.weight"], params[f"{prefix}.layers.{i}.bias"]) for i in range(n)]

    out["field.mlp_base_grid.tcnn_encoding.params"] = ops.tcnn_grid_unpack(
        fs.grid, params["field.mlp_base_grid.hash_table"]).cpu()
    out["field.mlp_base_mlp.tcnn_encoding.params"] = linear_to_mlp(
        get_mlp("field.mlp_base_mlp", 2), enc, 1 + fs.geo_feat_dim, fs.hidden_dim, 1)
    out["field.mlp_semantics.tcnn_encoding.params"] = linear_to_mlp(
        get_mlp("field.mlp_semantics", fs.num_layers_semantic), fs.geo_feat_dim, fs.hidden_dim_transient,
        fs.hidden_dim_semantics, fs.num_layers_semantic - 1)
    head = [(w.detach().cpu().clone(), b) for w, b in get_mlp("field.mlp_head", fs.num_layers_color)]
    head[0][0][:, list(SH_FLIPPED)] *= -1.0
    out["field.mlp_head.tcnn_encoding.params"] = linear_to_mlp(head, head_in, 3, fs.hidden_dim_color,
                                                               fs.num_layers_color - 1)
    for k in SemanticFieldHead.keys + ("field.embedding_appearance.embedding.weight",
                                       "camera_optimizer.pose_adjustment"):
        out[k] = params[k].detach().cpu().to(torch.float32)
    for i, ps in enumerate(prop_specs):
        pin = ps.grid.num_levels * ps.grid.features_per_level
        net = linear_to_mlp(get_mlp(f"proposal_networks.{i}.mlp", 2), pin, 1, ps.hidden_dim, 1, input_pad_value=0.0)
        grid = ops.tcnn_grid_unpack(ps.grid, params[f"proposal_networks.{i}.encoding.hash_table"]).cpu()
        out[f"proposal_networks.{i}.mlp_base.tcnn_encoding.params"] = torch.cat([net, grid])
    return out


def frozen_parameter_names(field_spec: FieldSpec, prop_specs: Sequence[ProposalSpec]) -> List[str]:
    """Parameters that must stay zero for a model to remain expressible as tcnn modules: every bias except the first-layer
    bias of an MLP whose input width is not a multiple of 16 (that one lives in the padded column)."""
    fs = field_spec
    enc, head_in = _field_dims(fs)
    names: List[str] = []

    def add(prefix: str, n_layers: int, n_in: int, ones_padded: bool):
        for i in range(n_layers):
            if i == 0 and ones_padded and _pad(n_in) > n_in:
                continue
            names.append(f"{prefix}.layers.{i}.bias")

    add("field.mlp_base_mlp", 2, enc, True)
    add("field.mlp_semantics", fs.num_layers_semantic, fs.geo_feat_dim, True)
    add("field.mlp_head", fs.num_layers_color, head_in, True)
    for i, ps in enumerate(prop_specs):
        add(f"proposal_networks.{i}.mlp", 2, ps.grid.num_levels * ps.grid.features_per_level, False)
    return names
