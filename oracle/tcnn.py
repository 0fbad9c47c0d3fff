"""Oracle: tiny-cuda-nn semantics of the modules ``FruitField`` builds with ``implementation="tcnn"``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The reference constructs its field with the default
``implementation="tcnn"`` (``fruit_nerf/fruit_field.py:95``, never overridden at ``fruit_nerf/fruit_nerf.py:97-112``),
i.e. ``HashEncoding`` -> ``tcnn.Encoding(HashGrid)``, ``MLP`` -> ``tcnn.Network(FullyFusedMLP)``, ``SHEncoding`` ->
``tcnn.Encoding(SphericalHarmonics)`` (``fruit_field.py:116-167``) and the proposal ``HashMLPDensityField``s ->
``tcnn.NetworkWithInputEncoding`` (``fruit_nerf.py:118-142``).  tiny-cuda-nn is a third-party dependency that is
absent from ``/root/reference`` and from this image (un-vendored, version unpinned by the reference --
it arrives through the nerfstudio 1.1.3 Docker image); this module restates its PUBLISHED algorithm from
``include/tiny-cuda-nn/encodings/grid.h`` (``grid_scale``, ``grid_resolution``, ``grid_index``, ``pos_fract``,
``kernel_grid``, the offset table of ``GridEncodingTemplated``), ``encodings/spherical_harmonics.h``,
``encodings/identity.h``, ``networks/fully_fused_mlp.cu`` (weight-matrix order and padding) and
``network_with_input_encoding.h`` (parameter order: network, then encoding).  PARITY UNPINNED: no tcnn here to
run, no reference checkpoint to load; pinned only by the known-answer tests in ``tests/test_oracle_tcnn.py``.

Arithmetic: tcnn stores fp32 master parameters in the torch module and casts them to fp16 for every forward
(``tinycudann/modules.py``); activations between layers are fp16 with fp32 accumulation.  The oracle evaluates with
the SAME fp16-rounded parameter values but in fp32 arithmetic (``half_params=True``, the default) -- the HIP path does
the same -- so it is *more* precise than the reference's own kernels; ``half_activations=True`` additionally rounds the
encoding outputs and every layer output to fp16 to show how far tcnn's own rounding sits from that (a diagnostic, not
a parity target: the tensor-core summation order cannot be restated).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

PRIMES = (1, 2654435761, 805459861)  # grid.h: coherent_prime_hash factors for 3 dimensions
MLP_ALIGN = 16  # FullyFusedMLP: input / output widths padded to multiples of 16 (tensor-core tile)


# ----------------------------------------------------------------------------------------------
# grid.h
# ----------------------------------------------------------------------------------------------

def grid_scale(level: int, log2_per_level_scale: np.float32, base_resolution: int) -> np.float32:
    """``grid_scale``: ``exp2f(level * log2_per_level_scale) * base_resolution - 1.0f`` (float32 throughout).
    The -1: ``base_resolution`` counts grid *vertices*."""
    return np.float32(np.exp2(np.float32(np.float32(level) * log2_per_level_scale)) * np.float32(base_resolution)
                      - np.float32(1.0))


def grid_resolution(scale: np.float32) -> int:
    """``grid_resolution``: ``(uint32_t)ceilf(scale) + 1``."""
    return int(np.ceil(scale)) + 1


@dataclass
class TcnnGridSpec:
    """What nerfstudio's ``HashEncoding(implementation="tcnn")`` passes to ``tcnn.Encoding`` (``otype: HashGrid``,
    ``interpolation: Linear``): ``base_resolution = min_res``, ``per_level_scale = growth_factor``."""

    num_levels: int = 16
    min_res: int = 16
    max_res: int = 2048
    log2_hashmap_size: int = 19
    features_per_level: int = 2

    def per_level_scale(self) -> np.float32:
        """nerfstudio: ``np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1))`` (float64), read by tcnn's
        JSON config as a float."""
        if self.num_levels <= 1:
            return np.float32(1.0)
        return np.float32(np.exp((np.log(self.max_res) - np.log(self.min_res)) / (self.num_levels - 1)))

    def scales(self) -> List[np.float32]:
        l2 = np.float32(np.log2(self.per_level_scale()))  # std::log2(float)
        return [grid_scale(l, l2, self.min_res) for l in range(self.num_levels)]

    def resolutions(self) -> List[int]:
        return [grid_resolution(s) for s in self.scales()]

    def offset_table(self) -> List[int]:
        """``GridEncodingTemplated`` constructor: entries (not floats) per level = min(next_multiple(res^3, 8), 2^log2_T),
        prefix-summed."""
        offs = [0]
        for res in self.resolutions():
            dense = res ** 3
            max_params = (2 ** 32 - 1) // 2
            n = max_params if float(dense) > float(max_params) else dense
            n = (n + 7) // 8 * 8
            n = min(n, 1 << self.log2_hashmap_size)
            offs.append(offs[-1] + n)
        return offs

    @property
    def n_params(self) -> int:
        return self.offset_table()[-1] * self.features_per_level

    @property
    def out_dim(self) -> int:
        return self.num_levels * self.features_per_level


def grid_index(pos_grid: np.ndarray, resolution: int, hashmap_size: int) -> np.ndarray:
    """``grid_index<3, CoherentPrime>`` for ``GridType::Hash``.  pos_grid: uint32 [...,3].

    The dense index is accumulated while ``stride <= hashmap_size``; the level is hashed iff the final stride
    (res^3, or the first partial product beyond the table) exceeds the table; ``index % hashmap_size`` either way."""
    pg = pos_grid.astype(np.uint32)
    stride = 1
    index = np.zeros(pg.shape[:-1], dtype=np.uint32)
    for dim in range(3):
        if stride > hashmap_size:
            break
        index = (index + pg[..., dim] * np.uint32(stride)).astype(np.uint32)
        stride *= resolution
    if hashmap_size < stride:
        index = np.zeros(pg.shape[:-1], dtype=np.uint32)
        for dim in range(3):
            index = index ^ (pg[..., dim] * np.uint32(PRIMES[dim])).astype(np.uint32)  # uint32 wrap-around
    return (index % np.uint32(hashmap_size)).astype(np.int64)


def round_f16(t: Tensor) -> Tensor:
    """Round to fp16 values (kept as float32).  Under autograd the rounding is a straight-through step: the gradient passes
    unchanged, in float32 -- a plain ``.to(float16)`` would also round the GRADIENT to fp16 on the way back and flush the
    ~1e-9 deltas of a 65 536-ray batch to zero (tiny-cuda-nn avoids that with its loss scale; the checker has no need to)."""
    t = t.to(torch.float32)
    r = t.detach().to(torch.float16).to(torch.float32)
    return t + (r - t.detach()) if t.requires_grad else r


def _as_compute(params: Tensor, half_params: bool) -> Tensor:
    return round_f16(params) if half_params else params.to(torch.float32)


def hash_grid(x: Tensor, params: Tensor, spec: TcnnGridSpec, half_params: bool = True,
              half_activations: bool = False) -> Tensor:
    """``kernel_grid`` with linear interpolation.  x [N,3] in [0,1]; params: the flat tcnn parameter vector
    (``n_params`` values, level after level, ``[entry][feature]`` within a level).  -> [N, L*F], level-major.

    Per level: ``pos = fmaf(scale, x, 0.5)``; ``pos_grid = floor(pos)``; ``pos -= pos_grid``; the 8 corners
    ``pos_grid + {0,1}^3`` with weights ``prod(pos or 1-pos)``, accumulated in corner order idx = 0..7 (bit d of idx
    selects the upper corner along dimension d)."""
    F = spec.features_per_level
    table = _as_compute(params, half_params).reshape(-1, F)
    offs = spec.offset_table()
    outs = []
    for l, (scale, res) in enumerate(zip(spec.scales(), spec.resolutions())):
        size = offs[l + 1] - offs[l]
        pos = x.to(torch.float32) * float(scale) + 0.5  # fmaf(scale, x, 0.5); differentiable w.r.t. x
        fl = torch.floor(pos.detach())
        frac = pos - fl
        pg = fl.to(torch.int64).numpy().astype(np.uint32)  # (uint32_t)(int)floorf(pos)
        acc = torch.zeros(x.shape[0], F, dtype=torch.float32)
        for idx in range(8):
            w = torch.ones(x.shape[0], dtype=torch.float32)
            corner = pg.copy()
            for d in range(3):
                if idx & (1 << d):
                    w = w * frac[:, d]
                    corner[:, d] = corner[:, d] + np.uint32(1)
                else:
                    w = w * (1.0 - frac[:, d])
            index = torch.from_numpy(grid_index(corner, res, size) + offs[l])
            acc = acc + table[index] * w[:, None]
            if half_activations:  # tcnn accumulates `result` in the parameter type
                acc = round_f16(acc)
        outs.append(acc)
    return torch.cat(outs, dim=-1)


# ----------------------------------------------------------------------------------------------
# spherical_harmonics.h
# ----------------------------------------------------------------------------------------------

def sh_deg4(u: Tensor) -> Tensor:
    """``kernel_sh`` degree 4: input in [0,1]^3 (the reference shifts its unit directions with
    ``shift_directions_for_tcnn``, ``fruit_field.py:209,244``), mapped back with ``x*2-1``; real SH WITH the
    Condon-Shortley phase, i.e. components 1,3,5,7,9,11,13,15 have the opposite sign of nerfstudio's torch
    ``components_from_spherical_harmonics``."""
    d = u * 2.0 - 1.0
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    c = torch.zeros((*u.shape[:-1], 16), dtype=u.dtype)
    c[..., 0] = 0.28209479177387814
    c[..., 1] = -0.48860251190291987 * y
    c[..., 2] = 0.48860251190291987 * z
    c[..., 3] = -0.48860251190291987 * x
    c[..., 4] = 1.0925484305920792 * xy
    c[..., 5] = -1.0925484305920792 * yz
    c[..., 6] = 0.94617469575755997 * z2 - 0.31539156525251999
    c[..., 7] = -1.0925484305920792 * xz
    c[..., 8] = 0.54627421529603959 * x2 - 0.54627421529603959 * y2
    c[..., 9] = 0.59004358992664352 * y * (-3.0 * x2 + y2)
    c[..., 10] = 2.8906114426405538 * xy * z
    c[..., 11] = 0.45704579946446572 * y * (1.0 - 5.0 * z2)
    c[..., 12] = 0.3731763325901154 * z * (5.0 * z2 - 3.0)
    c[..., 13] = 0.45704579946446572 * x * (1.0 - 5.0 * z2)
    c[..., 14] = 1.4453057213202769 * z * (x2 - y2)
    c[..., 15] = 0.59004358992664352 * x * (-x2 + 3.0 * y2)
    return c


# ----------------------------------------------------------------------------------------------
# fully_fused_mlp.cu / identity.h / network_with_input_encoding.h
# ----------------------------------------------------------------------------------------------

def _pad(n: int, a: int = MLP_ALIGN) -> int:
    return (n + a - 1) // a * a


def mlp_param_count(n_in: int, n_out: int, width: int, n_hidden: int) -> int:
    """``FullyFusedMLP``: [width, pad16(in)], (n_hidden-1) x [width, width], [pad16(out), width]; no biases."""
    return width * _pad(n_in) + (n_hidden - 1) * width * width + _pad(n_out) * width


def mlp_matrices(params: Tensor, n_in: int, n_out: int, width: int, n_hidden: int) -> List[Tensor]:
    """Split a FullyFusedMLP parameter vector into its row-major weight matrices ``[out, in]`` (``y = W x``)."""
    shapes = [(width, _pad(n_in))] + [(width, width)] * (n_hidden - 1) + [(_pad(n_out), width)]
    mats, o = [], 0
    for r, c in shapes:
        mats.append(params[o:o + r * c].reshape(r, c))
        o += r * c
    assert o == params.numel(), f"FullyFusedMLP parameter count {params.numel()} != {o}"
    return mats


def fully_fused_mlp(x_padded: Tensor, params: Tensor, n_in: int, n_out: int, width: int, n_hidden: int,
                    out_activation: Optional[str] = None, half_params: bool = True,
                    half_activations: bool = False) -> Tensor:
    """ReLU hidden layers, ``out_activation`` in {None, "sigmoid"}; returns the first ``n_out`` outputs."""
    mats = mlp_matrices(_as_compute(params, half_params), n_in, n_out, width, n_hidden)
    h = x_padded
    for i, w in enumerate(mats):
        if half_activations:
            h = round_f16(h)
        h = h @ w.t()
        if i < len(mats) - 1:
            h = torch.relu(h)
    if out_activation == "sigmoid":
        h = torch.sigmoid(h)
    if half_activations:
        h = round_f16(h)
    return h[:, :n_out]


def network(x: Tensor, params: Tensor, n_in: int, n_out: int, width: int, n_hidden: int,
            out_activation: Optional[str] = None, **kw) -> Tensor:
    """``tcnn.Network`` = ``NetworkWithInputEncoding`` with an ``Identity`` encoding whose output is padded to the
    network's input alignment; identity.h writes **1** into the padded columns (they act as a bias)."""
    pad = _pad(n_in) - n_in
    xp = torch.cat([x, torch.ones(x.shape[0], pad, dtype=x.dtype)], dim=-1) if pad else x
    return fully_fused_mlp(xp, params, n_in, n_out, width, n_hidden, out_activation, **kw)


def network_with_grid(x: Tensor, params: Tensor, spec: TcnnGridSpec, n_out: int, width: int, n_hidden: int,
                      out_activation: Optional[str] = None, half_params: bool = True,
                      half_activations: bool = False) -> Tensor:
    """``tcnn.NetworkWithInputEncoding(HashGrid, FullyFusedMLP)`` (nerfstudio ``MLPWithHashEncoding``, the proposal
    networks): parameters = [network | encoding]; grid.h pads the encoding output with **0**."""
    n_mlp = mlp_param_count(spec.out_dim, n_out, width, n_hidden)
    enc = hash_grid(x, params[n_mlp:], spec, half_params, half_activations)
    pad = _pad(spec.out_dim) - spec.out_dim
    if pad:
        enc = torch.cat([enc, torch.zeros(enc.shape[0], pad, dtype=enc.dtype)], dim=-1)
    return fully_fused_mlp(enc, params[:n_mlp], spec.out_dim, n_out, width, n_hidden, out_activation, half_params,
                           half_activations)


# ----------------------------------------------------------------------------------------------
# parameter vectors of a whole fruit_nerf model in nerfstudio's tcnn state-dict names
# ----------------------------------------------------------------------------------------------

def grid_spec_of(g) -> TcnnGridSpec:
    return TcnnGridSpec(g.num_levels, g.min_res, g.max_res, g.log2_hashmap_size, g.features_per_level)


def param_shapes(spec, prop_specs) -> Dict[str, Tuple[int, ...]]:
    """State-dict names / shapes of a tcnn-built ``FruitModel`` (``field.*`` per ``fruit_field.py:109-167``,
    ``proposal_networks.N.mlp_base`` per upstream ``HashMLPDensityField``)."""
    g = grid_spec_of(spec.grid)
    s: Dict[str, Tuple[int, ...]] = {}
    s["field.mlp_base_grid.tcnn_encoding.params"] = (g.n_params,)
    s["field.mlp_base_mlp.tcnn_encoding.params"] = (mlp_param_count(g.out_dim, 1 + spec.geo_feat_dim, spec.hidden_dim, 1),)
    s["field.mlp_semantics.tcnn_encoding.params"] = (mlp_param_count(
        spec.geo_feat_dim, spec.hidden_dim_transient, spec.hidden_dim_semantics, spec.num_layers_semantic - 1),)
    s["field.field_head_semantics.net.weight"] = (1, spec.hidden_dim_transient)
    s["field.field_head_semantics.net.bias"] = (1,)
    s["field.mlp_head.tcnn_encoding.params"] = (mlp_param_count(
        16 + spec.geo_feat_dim + spec.appearance_embedding_dim, 3, spec.hidden_dim_color, spec.num_layers_color - 1),)
    s["field.embedding_appearance.embedding.weight"] = (spec.num_images, spec.appearance_embedding_dim)
    for i, ps in enumerate(prop_specs):
        pg = grid_spec_of(ps.grid)
        s[f"proposal_networks.{i}.mlp_base.tcnn_encoding.params"] = (
            mlp_param_count(pg.out_dim, 1, ps.hidden_dim, 1) + pg.n_params,)
    s["camera_optimizer.pose_adjustment"] = (spec.num_images, 6)
    return s


def random_params(spec, prop_specs, seed: int = 0, grid_scale: float = 0.1) -> Dict[str, Tensor]:
    """Seeded test parameters (fp32 master values, as a checkpoint holds them): grid U(-1,1)*grid_scale (tcnn's own
    init is U(-1e-4,1e-4): a near-empty volume), MLP weights U(+-1/sqrt(fan_in)), embedding N(0,1)."""
    gen = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for name, shape in param_shapes(spec, prop_specs).items():
        if name.endswith("pose_adjustment"):
            out[name] = torch.zeros(shape)
        elif name.endswith("embedding.weight"):
            out[name] = torch.randn(shape, generator=gen)
        elif name.endswith("net.weight") or name.endswith("net.bias"):
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) / 8.0
        elif name.endswith("mlp_base_grid.tcnn_encoding.params"):
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) * grid_scale
        elif ".mlp_base.tcnn_encoding.params" in name:  # proposal: [network | grid]
            i = int(name.split(".")[1])
            pg = grid_spec_of(prop_specs[i].grid)
            n_mlp = shape[0] - pg.n_params
            net = (torch.rand(n_mlp, generator=gen) * 2 - 1) / 4.0
            grid = (torch.rand(pg.n_params, generator=gen) * 2 - 1) * grid_scale
            out[name] = torch.cat([net, grid])
        else:
            out[name] = (torch.rand(shape, generator=gen) * 2 - 1) / 8.0
    return out
