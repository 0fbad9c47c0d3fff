"""Tensor-level wrappers over the C ABI (``include/cropnerf_hip.h``).

PyTorch is plumbing here: device memory, the current HIP stream and nothing else -- every value these functions
return was computed by a kernel of libcropnerf_hip.so.  Inputs must be contiguous float32 (int64 for indices)
tensors on the ROCm device; a missing library or a failed call raises.
"""

from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib as L
from .config import FieldSpec, GridSpec, ProposalSpec
from .fruit_nerf.components.field_heads import SemanticFieldHead


def _stream(t: Tensor):
    if not t.is_cuda:
        raise RuntimeError("cropnerf_amd ops need tensors on the ROCm device (no CPU fallback exists)")
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name}: expected a contiguous float32 device tensor, got {t.dtype} {t.device} "
                        f"contiguous={t.is_contiguous()}")
    return t


def _i64(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.int64 or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name}: expected a contiguous int64 device tensor")
    return t


def _p(t: Optional[Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _farr(vals: Sequence[float]):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


# --------------------------------------------------------------------------------------------------------------
# parameter handles
# --------------------------------------------------------------------------------------------------------------

def _table_dtype(table: Tensor, name: str = "hash_table") -> int:
    if not table.is_cuda or not table.is_contiguous() or table.dtype not in (torch.float32, torch.float16):
        raise TypeError(f"{name}: expected a contiguous float32 / float16 device tensor, got {table.dtype} {table.device}")
    return L.TABLE_F32 if table.dtype == torch.float32 else L.TABLE_F16


def _grid_struct(table: Tensor, spec: GridSpec) -> L.Grid:
    dtype = _table_dtype(table)
    if tuple(table.shape) != (spec.num_entries, 2):
        raise ValueError(f"hash table shape {tuple(table.shape)} != {(spec.num_entries, 2)} (layout {spec.layout})")
    g = L.Grid()
    if spec.layout == "tcnn":
        plan = spec.plan()
        L.check(L.load().cn_tcnn_grid_describe(C.byref(plan), C.c_void_p(table.data_ptr()), dtype, C.byref(g)))
        return g
    g.table = table.data_ptr()
    g.num_levels = spec.num_levels
    g.log2_table_size = spec.log2_hashmap_size
    g.layout = L.GRID_TORCH
    g.table_dtype = dtype
    sc = spec.scalings()
    for i in range(L.CN_MAX_LEVELS):
        g.scalings[i] = sc[i] if i < len(sc) else 0.0
    return g


def tcnn_grid_pack(spec: GridSpec, packed: Tensor, table_dtype: torch.dtype = torch.float16) -> Tensor:
    """tcnn's parameter vector of one grid (``[n_params]``, float32 master copy or float16) -> this library's table."""
    plan = spec.plan()
    pd = _table_dtype(packed, "packed tcnn parameters")
    if packed.numel() != 2 * spec.num_packed_entries:
        raise ValueError(f"tcnn grid parameter count {packed.numel()} != {2 * spec.num_packed_entries}")
    table = torch.empty(spec.num_entries, 2, dtype=table_dtype, device=packed.device)
    L.check(L.load().cn_tcnn_grid_pack(C.byref(plan), _p(packed), pd, _p(table), _table_dtype(table), _stream(packed)))
    return table


def tcnn_grid_unpack(spec: GridSpec, table: Tensor, packed_dtype: torch.dtype = torch.float32) -> Tensor:
    plan = spec.plan()
    if tuple(table.shape) != (spec.num_entries, 2):
        raise ValueError(f"hash table shape {tuple(table.shape)} != {(spec.num_entries, 2)}")
    packed = torch.empty(2 * spec.num_packed_entries, dtype=packed_dtype, device=table.device)
    L.check(L.load().cn_tcnn_grid_unpack(C.byref(plan), _p(table), _table_dtype(table), _p(packed),
                                         _table_dtype(packed, "packed"), _stream(table)))
    return packed


def tcnn_grid_tie_gradients(spec: GridSpec, grad_table: Tensor) -> None:
    """Fold the gradients of alias entries into the entries that own the tcnn parameter (in place)."""
    L.check(L.load().cn_tcnn_grid_tie_gradients(C.byref(spec.plan()), _p(_f32(grad_table, "grad_table")),
                                                _stream(grad_table)))


def tcnn_grid_tie_parameters(spec: GridSpec, table: Tensor) -> None:
    """Copy every tcnn parameter of the dense levels into its alias entries (in place, after an optimiser step)."""
    L.check(L.load().cn_tcnn_grid_tie_parameters(C.byref(spec.plan()), _p(_f32(table, "table")), _stream(table)))


def _mlp_struct(params: Dict[str, Tensor], prefix: str, num_layers: int) -> L.Mlp:
    m = L.Mlp()
    m.num_layers = num_layers
    for i in range(num_layers):
        w = _f32(params[f"{prefix}.layers.{i}.weight"], f"{prefix}.layers.{i}.weight")
        b = _f32(params[f"{prefix}.layers.{i}.bias"], f"{prefix}.layers.{i}.bias")
        m.dims[i] = w.shape[1]
        m.dims[i + 1] = w.shape[0]
        m.weight[i] = w.data_ptr()
        m.bias[i] = b.data_ptr()
    return m


def _attach_scatter_scratch(grid: L.Grid, device, max_samples: Optional[int] = None) -> Optional[Tensor]:
    """``cn_grid.scatter_scratch`` for a GRADIENT grid: a zeroed buffer of ``cn_grid_scatter_scratch_bytes_for`` (the backward
    kernels accumulate the coarse levels' gradients in private copies / cell-major records there and leave it zeroed).
    ``max_samples``: the largest backward call it has to serve (None: any size -- 160-180 MB per grid of the default method)."""
    n = int(L.load().cn_grid_scatter_scratch_bytes_for(C.byref(grid), int(max_samples or 0)))
    if n <= 0:
        return None
    buf = torch.zeros(n // 4, dtype=torch.float32, device=device)
    grid.scatter_scratch = buf.data_ptr()
    grid.scatter_scratch_bytes = n
    return buf


def _scratch_too_small(handle, max_samples: Optional[int]) -> bool:
    """Whether a gradient handle's scatter scratch has to be (re-)allocated for backward calls of up to ``max_samples`` samples.
    A scratch sized for ANY batch (``_scratch_samples`` None) is the largest there is and is kept for every later request."""
    if os.environ.get("CN_SCATTER_SCRATCH", "1") == "0":
        return False
    if handle._scatter_scratch is None:
        return True
    have = getattr(handle, "_scratch_samples", None)
    if have is None:
        return False
    return max_samples is None or have < max_samples


class FieldHandle:
    """cn_field_params for a parameter dict (keeps the tensors alive)."""

    def __init__(self, params: Dict[str, Tensor], spec: FieldSpec):
        self.params = params
        self.spec = spec
        p = L.FieldParams()
        p.grid = _grid_struct(params["field.mlp_base_grid.hash_table"], spec.grid)
        p.base = _mlp_struct(params, "field.mlp_base_mlp", 2)
        p.semantics = _mlp_struct(params, "field.mlp_semantics", spec.num_layers_semantic)
        head_w, head_b = SemanticFieldHead(spec.hidden_dim_transient, 1).check(params)
        p.sem_head_weight = _f32(head_w, "sem head w").data_ptr()
        p.sem_head_bias = _f32(head_b, "sem head b").data_ptr()
        p.color = _mlp_struct(params, "field.mlp_head", spec.num_layers_color)
        emb = _f32(params["field.embedding_appearance.embedding.weight"], "appearance embedding")
        p.appearance = emb.data_ptr()
        p.num_images = emb.shape[0]
        p.app_dim = emb.shape[1]
        p.geo_feat_dim = spec.geo_feat_dim
        self.struct = p
        self.device = emb.device
        self._workspace: Optional[Tensor] = None
        self._scatter_scratch: Optional[Tensor] = None

    def enable_scatter_scratch(self, max_samples: Optional[int] = None) -> "FieldHandle":
        """For a handle over GRADIENT buffers: ``cn_grid.scatter_scratch``, sized for backward calls of at most ``max_samples``
        samples (None: any size).  Called again with a larger size it re-allocates."""
        if _scratch_too_small(self, max_samples):
            self._scatter_scratch = _attach_scatter_scratch(self.struct.grid, self.device, max_samples)
            self._scratch_samples = max_samples
        return self

    def workspace(self) -> Tensor:
        n = int(L.load().cn_render_workspace_bytes(C.byref(self.struct)))
        if self._workspace is None or self._workspace.numel() < n:
            self._workspace = torch.empty(n + 16, dtype=torch.uint8, device=self.device)
        return self._workspace


class DensityHandle:
    def __init__(self, params: Dict[str, Tensor], level: int, spec: ProposalSpec):
        self.params = params
        self.spec = spec
        p = L.DensityParams()
        p.grid = _grid_struct(params[f"proposal_networks.{level}.encoding.hash_table"], spec.grid)
        p.mlp = _mlp_struct(params, f"proposal_networks.{level}.mlp", 2)
        self.struct = p
        self._scatter_scratch: Optional[Tensor] = None

    def enable_scatter_scratch(self, max_samples: Optional[int] = None) -> "DensityHandle":
        """For a handle over GRADIENT buffers (see ``FieldHandle.enable_scatter_scratch``)."""
        if _scratch_too_small(self, max_samples):
            self._scatter_scratch = _attach_scatter_scratch(
                self.struct.grid, self.params[next(k for k in self.params if k.endswith("hash_table"))].device, max_samples)
            self._scratch_samples = max_samples
        return self


# ---- deterministic accumulation (tests only: the CN_DETERMINISTIC_SCATTER=1 library, include/cropnerf_hip.h) --------------------
_det_state: Dict[str, object] = {"key": None, "owner": None, "shadows": [], "misses": None}


def deterministic_register(tensors: Sequence[Tensor], owner: object = None) -> None:
    """Route the float atomics that land in ``tensors`` (float32, on the device) through 64-bit integer shadows.  The registry is
    one per process: registering replaces what was registered before (the buffers of another trainer, a re-allocated scratch).
    A no-op when the same buffers of the same ``owner`` are registered already; must not be called during a stream capture when
    something changed."""
    if not L.deterministic():
        raise RuntimeError("deterministic_register: set CN_DETERMINISTIC_SCATTER=1 (the test build) first")
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    key = tuple((t.data_ptr(), t.numel()) for t in ts)
    if _det_state["key"] == key and _det_state["owner"] is owner:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("deterministic_register: the set of accumulation buffers changed inside a graph capture")
    lib = L.load()
    torch.cuda.synchronize()
    L.check(lib.cn_deterministic_clear())
    if _det_state["misses"] is None or _det_state["misses"].device != ts[0].device:
        _det_state["misses"] = torch.zeros(1, dtype=torch.int64, device=ts[0].device)
    shadows = []
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("deterministic_register: contiguous float32 buffers only")
        sh = torch.zeros(t.numel(), dtype=torch.int64, device=t.device)
        L.check(lib.cn_deterministic_register(_p(t), t.numel(), _p(sh), _p(_det_state["misses"])))
        shadows.append(sh)
    _det_state.update(key=key, owner=owner, shadows=shadows, tensors=ts)


def deterministic_misses() -> int:
    """Float atomics of the training kernels whose destination was in no registered range (they were summed the default,
    order-dependent way) since the counter was created.  Tests assert 0."""
    m = _det_state["misses"]
    return 0 if m is None else int(m.item())


def scene_struct(aabb: Tensor, contraction: bool) -> L.Scene:
    s = L.Scene()
    flat = [float(v) for v in aabb.reshape(-1).tolist()]
    for i in range(6):
        s.aabb[i] = flat[i]
    s.contraction = 1 if contraction else 0
    return s


def render_opts(num_samples: int, spacing: int = L.SPACING_UNIFORM, bg_mode: int = L.BG_LAST_SAMPLE,
                bg_color: Sequence[float] = (0.0, 0.0, 0.0), app_mode: int = L.APP_MEAN, sh_unit_dir: bool = True,
                eval_clamp: bool = True, density_only: bool = False, image_width: int = 0,
                pixel_start: int = 0, early_stop_transmittance: float = 0.0,
                matrix_precision: int = L.MATRIX_FP32) -> L.RenderOpts:
    o = L.RenderOpts()
    o.num_samples = int(num_samples)
    o.spacing = spacing
    o.bg_mode = bg_mode
    for i in range(3):
        o.bg_color[i] = float(bg_color[i])
    o.app_mode = app_mode
    o.sh_unit_dir = 1 if sh_unit_dir else 0
    o.eval_clamp = 1 if eval_clamp else 0
    o.density_only = 1 if density_only else 0
    o.image_width = int(image_width)
    o.pixel_start = int(pixel_start)
    o.early_stop_transmittance = float(early_stop_transmittance)
    o.matrix_precision = int(matrix_precision)
    return o


# --------------------------------------------------------------------------------------------------------------
# ray generation
# --------------------------------------------------------------------------------------------------------------

def raygen_pinhole(c2w: Tensor, intrinsics: Tensor, *, ray_indices: Optional[Tensor] = None, cam: int = 0,
                   height: int = 0, width: int = 0, pixel_start: int = 0, num_rays: Optional[int] = None,
                   camera_index_value: int = -1) -> Dict[str, Tensor]:
    lib = L.load()
    c2w = _f32(c2w, "c2w")
    intrinsics = _f32(intrinsics, "intrinsics")
    if ray_indices is not None:
        ray_indices = _i64(ray_indices, "ray_indices")
        R = ray_indices.shape[0]
    else:
        R = height * width - pixel_start if num_rays is None else num_rays
    dev = c2w.device
    out = {
        "origins": torch.empty(R, 3, device=dev), "directions": torch.empty(R, 3, device=dev),
        "pixel_area": torch.empty(R, 1, device=dev), "camera_indices": torch.empty(R, 1, dtype=torch.int64, device=dev),
        "directions_norm": torch.empty(R, 1, device=dev),
    }
    L.check(lib.cn_raygen_pinhole(_p(c2w), _p(intrinsics), _p(ray_indices), cam, height, width, pixel_start, R,
                                  camera_index_value, _p(out["origins"]), _p(out["directions"]),
                                  _p(out["pixel_area"]), _p(out["camera_indices"]), _p(out["directions_norm"]),
                                  _stream(c2w)))
    return out


def intersect_aabb(origins: Tensor, directions: Tensor, aabb6: Sequence[float]) -> Tuple[Tensor, Tensor]:
    lib = L.load()
    R = origins.shape[0]
    nears = torch.empty(R, 1, device=origins.device)
    fars = torch.empty(R, 1, device=origins.device)
    L.check(lib.cn_intersect_aabb(_p(_f32(origins, "origins")), _p(_f32(directions, "directions")), _farr(aabb6), R,
                                  _p(nears), _p(fars), _stream(origins)))
    return nears, fars


def surface_grid(x0: float, x1: float, nx: int, y0: float, y1: float, ny: int, z: float, device) -> Tensor:
    lib = L.load()
    pts = torch.empty(nx * ny, 3, device=device)
    L.check(lib.cn_surface_grid(x0, x1, nx, y0, y1, ny, z, _p(pts), _stream(pts)))
    return pts


def raygen_ortho(surface_points: Tensor, plane_vector: Sequence[float], start: int, num_rays: int) -> Dict[str, Tensor]:
    lib = L.load()
    dev = surface_points.device
    out = {"origins": torch.empty(num_rays, 3, device=dev), "directions": torch.empty(num_rays, 3, device=dev),
           "pixel_area": torch.empty(num_rays, 1, device=dev), "nears": torch.empty(num_rays, 1, device=dev),
           "fars": torch.empty(num_rays, 1, device=dev)}
    if num_rays == 0:  # past the end of the surface grid: an empty bundle, as slicing gives the reference
        return out
    L.check(lib.cn_raygen_ortho(_p(_f32(surface_points, "surface_points")), _farr(plane_vector), start, num_rays,
                                _p(out["origins"]), _p(out["directions"]), _p(out["pixel_area"]), _p(out["nears"]),
                                _p(out["fars"]), _stream(surface_points)))
    return out


def apply_pose_adjustment(pose_adjustment: Tensor, camera_indices: Tensor, origins: Tensor, directions: Tensor) -> None:
    """In place, like ``camera_optimizer.apply_to_raybundle``."""
    lib = L.load()
    L.check(lib.cn_apply_pose_adjustment(_p(_f32(pose_adjustment, "pose_adjustment")),
                                         _p(_i64(camera_indices, "camera_indices")), origins.shape[0],
                                         _p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                         _stream(origins)))


# --------------------------------------------------------------------------------------------------------------
# batched semantic projection (fruit_nerf.py:254-318): the ray side of many (camera, box) jobs in one launch sequence
# --------------------------------------------------------------------------------------------------------------

def _u8(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.uint8 or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name}: expected a contiguous uint8 device tensor")
    return t


def _i32(t: Optional[Tensor], name: str) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.int32 or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name}: expected a contiguous int32 device tensor")
    return t


def _jobs(jobs: Tensor) -> Tuple[Tensor, int]:
    size = C.sizeof(L.ProjectionJob)
    jobs = _u8(jobs, "jobs")
    if jobs.numel() % size or jobs.data_ptr() % 8:
        raise ValueError(f"jobs: a byte image of cn_projection_job records ({size} B each, 8-byte aligned)")
    return jobs, jobs.numel() // size


def projection_test(jobs: Tensor, num_slots: int, image_width: int, min_rays: int = 10) -> Dict[str, Tensor]:
    """``cn_projection_test``: ``jobs`` = the device copy of a ``cn_projection_job`` array as bytes.  Returns ``flags`` [P] uint8
    (1 = the slot's ray hits its job's box and the job has at least ``min_rays`` such rays), ``job_of_slot`` [P] int32 and
    ``hit_count`` [J] int32 (before the ``min_rays`` rule)."""
    lib = L.load()
    jobs, J = _jobs(jobs)
    dev = jobs.device
    out = {"flags": torch.empty(num_slots, dtype=torch.uint8, device=dev),
           "job_of_slot": torch.empty(num_slots, dtype=torch.int32, device=dev),
           "hit_count": torch.empty(J, dtype=torch.int32, device=dev)}
    L.check(lib.cn_projection_test(_p(jobs), J, num_slots, image_width, min_rays, _p(out["flags"]), _p(out["job_of_slot"]),
                                   _p(out["hit_count"]), _stream(jobs)))
    return out


def projection_gather(jobs: Tensor, job_of_slot: Tensor, hit_slots: Tensor, image_width: int,
                      want_job_pixel: bool = False) -> Dict[str, Tensor]:
    """``cn_projection_gather``: the jagged ray bundle of the listed slots."""
    lib = L.load()
    jobs, _ = _jobs(jobs)
    dev = jobs.device
    N = hit_slots.numel()
    out = {"origins": torch.empty(N, 3, device=dev), "directions": torch.empty(N, 3, device=dev),
           "nears": torch.empty(N, 1, device=dev), "fars": torch.empty(N, 1, device=dev),
           "camera_indices": torch.empty(N, 1, dtype=torch.int64, device=dev)}
    if want_job_pixel:
        out["ray_job"] = torch.empty(N, dtype=torch.int32, device=dev)
        out["ray_pixel"] = torch.empty(N, dtype=torch.int32, device=dev)
    L.check(lib.cn_projection_gather(_p(jobs), _p(_i32(job_of_slot, "job_of_slot")), _p(_i64(hit_slots, "hit_slots")), N,
                                     image_width, _p(out["origins"]), _p(out["directions"]), _p(out["nears"]),
                                     _p(out["fars"]), _p(out["camera_indices"]), _p(out.get("ray_job")),
                                     _p(out.get("ray_pixel")), _stream(jobs)))
    return out


def projection_scatter(semantics: Tensor, occlusion: Tensor, hit_slots: Tensor, num_slots: int,
                       occlusion_threshold: float = 0.5, want_float: bool = False, want_u8: bool = True) -> Dict[str, Tensor]:
    """``cn_projection_scatter``: per-slot ``wo_occ`` / ``visible`` values (zero where no ray hit), float and / or uint8."""
    lib = L.load()
    dev = semantics.device
    out: Dict[str, Tensor] = {}
    if want_float:
        out["wo_occ_f32"], out["visible_f32"] = torch.zeros(num_slots, device=dev), torch.zeros(num_slots, device=dev)
    if want_u8:
        out["wo_occ_u8"] = torch.zeros(num_slots, dtype=torch.uint8, device=dev)
        out["visible_u8"] = torch.zeros(num_slots, dtype=torch.uint8, device=dev)
    sem = _f32(semantics.reshape(-1), "semantics")
    occ = _f32(occlusion.reshape(-1), "occlusion")
    N = hit_slots.numel()
    if sem.numel() != N or occ.numel() != N:
        raise ValueError(f"semantics / occlusion hold {sem.numel()} / {occ.numel()} rays for {N} hit slots")
    L.check(lib.cn_projection_scatter(_p(sem), _p(occ), _p(_i64(hit_slots, "hit_slots")), N, occlusion_threshold,
                                      _p(out.get("wo_occ_f32")), _p(out.get("visible_f32")), _p(out.get("wo_occ_u8")),
                                      _p(out.get("visible_u8")), _stream(semantics)))
    return out


def projection_paste(jobs: Tensor, job_of_slot: Tensor, slot_values: Tensor, image_of_job: Tensor, images: Tensor) -> Tensor:
    """``cn_projection_paste``: job rectangles of per-slot uint8 values into ``images`` [num_images, H, W] (in place)."""
    lib = L.load()
    jobs, J = _jobs(jobs)
    if image_of_job.numel() != J or images.dim() != 3:
        raise ValueError("image_of_job: one image index per job; images: [num_images, H, W]")
    L.check(lib.cn_projection_paste(_p(jobs), _p(_i32(job_of_slot, "job_of_slot")), _p(_u8(slot_values, "slot_values")),
                                    slot_values.numel(), _p(_i32(image_of_job, "image_of_job")), images.shape[0],
                                    images.shape[1], images.shape[2], _p(_u8(images, "images")), _stream(jobs)))
    return images


def apply_pose_adjustment_to(pose_adjustment: Tensor, camera_indices: Tensor, origins: Tensor, directions: Tensor,
                             out_origins: Tensor, out_directions: Tensor) -> None:
    """``cn_apply_pose_adjustment_to``: the tweak written to separate outputs (the inputs stay as they are)."""
    lib = L.load()
    L.check(lib.cn_apply_pose_adjustment_to(_p(_f32(pose_adjustment, "pose_adjustment")),
                                            _p(_i64(camera_indices, "camera_indices")), origins.shape[0],
                                            _p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                            _p(_f32(out_origins, "out_origins")), _p(_f32(out_directions, "out_directions")),
                                            _stream(origins)))


def train_epilogue(loss_sums: Tensor, num_rays: int, num_samples: int, semantic_loss_weight: float,
                   interlevel_loss_mult: float, pose_adjustment: Optional[Tensor], out: Optional[Tensor] = None) -> Tensor:
    """``cn_train_epilogue``: [rgb_loss, semantics_loss, interlevel_loss, camera_opt_regularizer, psnr, |t|, |w|, distortion]
    (``loss_sums`` of five elements: the fifth is the sum ``distortion_metric(..., acc=)`` left there; of four: distortion = 0)."""
    lib = L.load()
    if out is None:
        out = torch.empty(8, device=loss_sums.device)
    C_ = 0 if pose_adjustment is None else pose_adjustment.shape[0]
    L.check(lib.cn_train_epilogue(_p(_f32(loss_sums, "loss_sums")), loss_sums.numel(), num_rays, num_samples, semantic_loss_weight,
                                  interlevel_loss_mult, _p(_f32(pose_adjustment, "pose_adjustment")), C_, _p(_f32(out, "out")),
                                  _stream(loss_sums)))
    return out


def embedding_mean(embedding: Tensor) -> Tensor:
    lib = L.load()
    out = torch.empty(embedding.shape[1], device=embedding.device)
    L.check(lib.cn_embedding_mean(_p(_f32(embedding, "embedding")), embedding.shape[0], embedding.shape[1], _p(out),
                                  _stream(embedding)))
    return out


# --------------------------------------------------------------------------------------------------------------
# samplers
# --------------------------------------------------------------------------------------------------------------

def sample_spaced(nears: Tensor, fars: Tensor, num_samples: int, spacing: int = L.SPACING_UNIFORM,
                  t_rand: Optional[Tensor] = None) -> Dict[str, Tensor]:
    lib = L.load()
    R = nears.shape[0]
    dev = nears.device
    out = {k: torch.empty(R, num_samples, device=dev) for k in ("starts", "ends", "spacing_starts", "spacing_ends")}
    stride = 0 if t_rand is None else t_rand.shape[1]
    L.check(lib.cn_sample_spaced(_p(_f32(nears, "nears")), _p(_f32(fars, "fars")), R, num_samples, spacing,
                                 _p(_f32(t_rand, "t_rand")), stride, _p(out["starts"]), _p(out["ends"]),
                                 _p(out["spacing_starts"]), _p(out["spacing_ends"]), _stream(nears)))
    return out


def sample_pdf(prev_spacing_bins: Tensor, weights: Tensor, nears: Tensor, fars: Tensor, num_samples: int,
               anneal: float = 1.0, spacing: int = L.SPACING_PIECEWISE, u_rand: Optional[Tensor] = None
               ) -> Tuple[Tensor, Tensor]:
    lib = L.load()
    R, s_in = weights.shape
    dev = weights.device
    sp = torch.empty(R, num_samples + 1, device=dev)
    eu = torch.empty(R, num_samples + 1, device=dev)
    stride = 0 if u_rand is None else u_rand.shape[1]
    L.check(lib.cn_sample_pdf(_p(_f32(prev_spacing_bins, "prev_spacing_bins")), _p(_f32(weights, "weights")),
                              _p(_f32(nears, "nears")), _p(_f32(fars, "fars")), R, s_in, num_samples, anneal, spacing,
                              _p(_f32(u_rand, "u_rand")), stride, _p(sp), _p(eu), _stream(weights)))
    return sp, eu


def proposal_sample(props: Sequence[DensityHandle], scene: L.Scene, origins: Tensor, directions: Tensor,
                    nears: Tensor, fars: Tensor, s_prop: Sequence[int], s_final: int, anneal: float = 1.0,
                    matrix_precision: int = L.MATRIX_FP32) -> Dict[str, Tensor]:
    """``cn_proposal_sample`` / ``cn_proposal_sample_mp``: ``matrix_precision=MATRIX_F16`` evaluates half-table proposal networks in
    tiny-cuda-nn's arithmetic class (the sampler's side of ``FruitNerfModelConfig.matrix_precision = "f16"``)."""
    lib = L.load()
    R = origins.shape[0]
    dev = origins.device
    n = len(props)
    arr = (C.POINTER(L.DensityParams) * n)(*[C.pointer(p.struct) for p in props])
    sp_arr = (C.c_int32 * n)(*[int(s) for s in s_prop])
    eu = torch.empty(R, s_final + 1, device=dev)
    sp = torch.empty(R, s_final + 1, device=dev)
    depth = torch.empty(n, R, device=dev)
    if matrix_precision != L.MATRIX_FP32:
        L.check(lib.cn_proposal_sample_mp(arr, n, C.byref(scene), _p(_f32(origins, "origins")),
                                          _p(_f32(directions, "directions")), _p(_f32(nears, "nears")), _p(_f32(fars, "fars")),
                                          R, sp_arr, s_final, anneal, _p(eu), _p(sp), _p(depth), int(matrix_precision),
                                          _stream(origins)))
    else:
        L.check(lib.cn_proposal_sample(arr, n, C.byref(scene), _p(_f32(origins, "origins")),
                                       _p(_f32(directions, "directions")), _p(_f32(nears, "nears")), _p(_f32(fars, "fars")),
                                       R, sp_arr, s_final, anneal, _p(eu), _p(sp), _p(depth), C.c_void_p(0), 0,
                                       _stream(origins)))
    return {"euclidean_bins": eu, "spacing_bins": sp, "prop_depth": depth}


def proposal_sample_fused_supported(props: Sequence[DensityHandle], s_prop: Sequence[int], s_final: int) -> bool:
    """The shapes ``cn_proposal_sample`` / ``cn_proposal_sample_train`` are built for (others compose the unfused calls)."""
    return (1 <= len(props) <= 3 and max(list(s_prop) + [s_final]) <= 512 and
            all(p.spec.grid.num_levels in (5, 7) and p.spec.hidden_dim == 16 for p in props))


def proposal_sample_train(props: Sequence[DensityHandle], scene: L.Scene, origins: Tensor, directions: Tensor,
                          nears: Tensor, fars: Tensor, s_prop: Sequence[int], s_final: int, anneal,
                          jitter: Tensor) -> Dict[str, object]:
    """``cn_proposal_sample_train``: the proposal sampler of the training forward in one launch.  ``jitter`` [n + 1, R].
    Returns the final bins and, per level, what the interlevel loss and the proposal backward read.  ``anneal``: a float, or a
    one-element device tensor (``cn_proposal_sample_train_dev``: the exponent is read at run time -- captured HIP graphs)."""
    lib = L.load()
    R = origins.shape[0]
    dev = origins.device
    n = len(props)
    if tuple(jitter.shape) != (n + 1, R):
        raise ValueError(f"jitter must be [{n + 1}, {R}], got {tuple(jitter.shape)}")
    arr = (C.POINTER(L.DensityParams) * n)(*[C.pointer(p.struct) for p in props])
    sp_arr = (C.c_int32 * n)(*[int(s) for s in s_prop])
    eu = torch.empty(R, s_final + 1, device=dev)
    sp = torch.empty(R, s_final + 1, device=dev)
    levels = [{"bins": torch.empty(R, int(s) + 1, device=dev), "starts": torch.empty(R, int(s), device=dev),
               "ends": torch.empty(R, int(s), device=dev), "density": torch.empty(R, int(s), device=dev)} for s in s_prop]
    outs = (L.ProposalLevelOut * n)(*[L.ProposalLevelOut(lv["bins"].data_ptr(), lv["starts"].data_ptr(),
                                                         lv["ends"].data_ptr(), lv["density"].data_ptr())
                                      for lv in levels])
    starts, ends = torch.empty(R, s_final, device=dev), torch.empty(R, s_final, device=dev)
    if isinstance(anneal, Tensor):
        L.check(lib.cn_proposal_sample_train_dev(arr, n, C.byref(scene), _p(_f32(origins, "origins")),
                                                 _p(_f32(directions, "directions")), _p(_f32(nears, "nears")),
                                                 _p(_f32(fars, "fars")), R, sp_arr, s_final, _p(_f32(anneal, "anneal")),
                                                 _p(_f32(jitter, "jitter")), outs, _p(eu), _p(sp), _p(starts), _p(ends),
                                                 _stream(origins)))
    else:
        L.check(lib.cn_proposal_sample_train(arr, n, C.byref(scene), _p(_f32(origins, "origins")),
                                             _p(_f32(directions, "directions")), _p(_f32(nears, "nears")),
                                             _p(_f32(fars, "fars")), R, sp_arr, s_final, anneal, _p(_f32(jitter, "jitter")),
                                             outs, _p(eu), _p(sp), _p(starts), _p(ends), _stream(origins)))
    return {"euclidean_bins": eu, "spacing_bins": sp, "starts": starts, "ends": ends, "levels": levels}


# --------------------------------------------------------------------------------------------------------------
# field / compositing
# --------------------------------------------------------------------------------------------------------------

def proposal_density(prop: DensityHandle, scene: L.Scene, origins: Tensor, directions: Tensor, starts: Tensor,
                     ends: Tensor) -> Tensor:
    lib = L.load()
    R, S = starts.shape
    den = torch.empty(R, S, device=starts.device)
    L.check(lib.cn_proposal_density(C.byref(prop.struct), C.byref(scene), _p(_f32(origins, "origins")),
                                    _p(_f32(directions, "directions")), _p(_f32(starts, "starts")),
                                    _p(_f32(ends, "ends")), R, S, _p(den), _stream(starts)))
    return den


def field_eval(fh: FieldHandle, scene: L.Scene, origins: Tensor, directions: Tensor, camera_indices: Optional[Tensor],
               starts: Tensor, ends: Tensor, app_mode: int = L.APP_MEAN, sh_unit_dir: bool = True,
               want_positions: bool = False, matrix_precision: int = L.MATRIX_FP32) -> Dict[str, Tensor]:
    """``cn_field_eval_mp``: ``matrix_precision`` as ``cn_render_opts.matrix_precision`` (the two field shapes of the
    reference's configs have a split-bf16 form; other shapes are evaluated in exact fp32 whatever it says)."""
    lib = L.load()
    R, S = starts.shape
    dev = starts.device
    out = {"density": torch.empty(R, S, device=dev), "rgb": torch.empty(R, S, 3, device=dev),
           "semantics": torch.empty(R, S, device=dev)}
    pos = torch.empty(R, S, 3, device=dev) if want_positions else None
    L.check(lib.cn_field_eval_mp(C.byref(fh.struct), C.byref(scene), app_mode, 1 if sh_unit_dir else 0,
                                 _p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                 _p(_i64(camera_indices, "camera_indices")), _p(_f32(starts, "starts")),
                                 _p(_f32(ends, "ends")), R, S, _p(out["density"]), _p(out["rgb"]), _p(out["semantics"]),
                                 _p(pos), int(matrix_precision), _stream(starts)))
    if pos is not None:
        out["positions"] = pos
    return out


def composite(starts: Tensor, ends: Tensor, density: Tensor, rgb: Optional[Tensor] = None,
              semantics: Optional[Tensor] = None, bg_mode: int = L.BG_LAST_SAMPLE,
              bg_color: Sequence[float] = (0.0, 0.0, 0.0), eval_clamp: bool = True, want_weights: bool = False
              ) -> Dict[str, Tensor]:
    lib = L.load()
    R, S = starts.shape
    dev = starts.device
    out = {"accumulation": torch.empty(R, 1, device=dev), "depth": torch.empty(R, 1, device=dev)}
    if rgb is not None:
        out["rgb"] = torch.empty(R, 3, device=dev)
    if semantics is not None:
        out["semantics"] = torch.empty(R, 1, device=dev)
        out["semantics_colormap"] = torch.empty(R, 3, device=dev)
    if want_weights:
        out["weights"] = torch.empty(R, S, device=dev)
    L.check(lib.cn_composite(_p(_f32(starts, "starts")), _p(_f32(ends, "ends")), _p(_f32(density, "density")),
                             _p(_f32(rgb, "rgb")), _p(_f32(semantics, "semantics")), R, S, bg_mode, _farr(bg_color),
                             1 if eval_clamp else 0, _p(out.get("rgb")), _p(out["accumulation"]), _p(out["depth"]),
                             _p(out.get("semantics")), _p(out.get("semantics_colormap")), _p(out.get("weights")),
                             _stream(starts)))
    return out


def render_rays(fh: FieldHandle, scene: L.Scene, opts: L.RenderOpts, origins: Tensor, directions: Tensor,
                nears: Tensor, fars: Tensor, camera_indices: Optional[Tensor] = None, bins: Optional[Tensor] = None,
                want_weights: bool = False) -> Dict[str, Tensor]:
    """Fused sampler + field + compositor (cn_render_rays)."""
    lib = L.load()
    R = origins.shape[0]
    dev = origins.device
    S = opts.num_samples
    if bins is not None and tuple(bins.shape) != (R, S + 1):
        raise ValueError(f"bins shape {tuple(bins.shape)} != {(R, S + 1)}")
    out: Dict[str, Tensor] = {"accumulation": torch.empty(R, 1, device=dev)}
    if not opts.density_only:
        out.update({"rgb": torch.empty(R, 3, device=dev), "depth": torch.empty(R, 1, device=dev),
                    "semantics": torch.empty(R, 1, device=dev), "semantics_colormap": torch.empty(R, 3, device=dev)})
    if want_weights:
        out["weights"] = torch.empty(R, S, device=dev)
    ws = fh.workspace()
    L.check(lib.cn_render_rays(C.byref(fh.struct), C.byref(scene), C.byref(opts), _p(_f32(origins, "origins")),
                               _p(_f32(directions, "directions")), _p(_f32(nears, "nears")), _p(_f32(fars, "fars")),
                               _p(_i64(camera_indices, "camera_indices")), _p(_f32(bins, "bins")), R,
                               _p(out.get("rgb")), _p(out["accumulation"]), _p(out.get("depth")),
                               _p(out.get("semantics")), _p(out.get("semantics_colormap")), _p(out.get("weights")),
                               C.c_void_p(ws.data_ptr()), ws.numel(), _stream(origins)))
    return out


def render_samples(fh: FieldHandle, scene: L.Scene, opts: L.RenderOpts, origins: Tensor, directions: Tensor,
                   nears: Tensor, fars: Tensor, camera_indices: Optional[Tensor] = None,
                   bins: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """Fused sampler + field with per-sample outputs (cn_render_samples): the export-mode forward."""
    lib = L.load()
    R = origins.shape[0]
    dev = origins.device
    S = opts.num_samples
    out = {"density": torch.empty(R, S, device=dev), "rgb": torch.empty(R, S, 3, device=dev),
           "semantics": torch.empty(R, S, device=dev), "positions": torch.empty(R, S, 3, device=dev),
           "semantics_colormap": torch.empty(R, S, dtype=torch.int64, device=dev)}
    ws = fh.workspace()
    L.check(lib.cn_render_samples(C.byref(fh.struct), C.byref(scene), C.byref(opts), _p(_f32(origins, "origins")),
                                  _p(_f32(directions, "directions")), _p(_f32(nears, "nears")), _p(_f32(fars, "fars")),
                                  _p(_i64(camera_indices, "camera_indices")), _p(_f32(bins, "bins")), R,
                                  _p(out["density"]), _p(out["rgb"]), _p(out["semantics"]), _p(out["positions"]),
                                  _p(out["semantics_colormap"]), C.c_void_p(ws.data_ptr()), ws.numel(),
                                  _stream(origins)))
    return out


# --------------------------------------------------------------------------------------------------------------
# exporters
# --------------------------------------------------------------------------------------------------------------

def export_compact(positions: Tensor, rgb: Tensor, semantics: Tensor, density: Tensor, capacity: int,
                   sem_thresh: float = 3.0, den_thresh: float = 70.0,
                   buffers: Optional[Tuple[List[Tensor], List[Tensor], Tensor]] = None
                   ) -> Tuple[List[Tensor], List[Tensor], Tensor]:
    """Appends to three (points [cap,3], colors [cap,4]) sets; returns (points, colors, counts[3] int64 device)."""
    lib = L.load()
    dev = positions.device
    N = semantics.numel()
    if buffers is None:
        pts = [torch.empty(capacity, 3, device=dev) for _ in range(3)]
        cols = [torch.empty(capacity, 4, device=dev) for _ in range(3)]
        counts = torch.zeros(3, dtype=torch.int64, device=dev)
    else:
        pts, cols, counts = buffers
    parr = (C.c_void_p * 3)(*[t.data_ptr() for t in pts])
    carr = (C.c_void_p * 3)(*[t.data_ptr() for t in cols])
    L.check(lib.cn_export_compact(_p(_f32(positions, "positions")), _p(_f32(rgb, "rgb")),
                                  _p(_f32(semantics, "semantics")), _p(_f32(density, "density")), N, sem_thresh,
                                  den_thresh, capacity, parr, carr, _p(counts), _stream(positions)))
    return pts, cols, counts


def pointcloud_compact(origins: Tensor, directions: Tensor, depth: Tensor, rgb: Tensor, semantics_colormap: Tensor,
                       capacity: int, buffers: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None
                       ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    lib = L.load()
    dev = origins.device
    R = origins.shape[0]
    if buffers is None:
        pts = torch.empty(capacity, 3, device=dev)
        cols = torch.empty(capacity, 3, device=dev)
        dirs = torch.empty(capacity, 3, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
    else:
        pts, cols, dirs, count = buffers
    L.check(lib.cn_pointcloud_compact(_p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                      _p(_f32(depth, "depth")), _p(_f32(rgb, "rgb")),
                                      _p(_f32(semantics_colormap, "semantics_colormap")), R, capacity, _p(pts),
                                      _p(cols), _p(dirs), _p(count), _stream(origins)))
    return pts, cols, dirs, count


def pixel_sample(seed: int, first_call: Tensor, num_calls: int, rays_per_call: int, num_cameras: int, height: int,
                 width: int) -> Tensor:
    """``cn_pixel_sample``: the (camera, row, col) draws of calls ``first_call`` .. ``first_call + num_calls - 1`` of the point-cloud
    exporter's pixel stream, [num_calls * rays_per_call, 3] int64.  ``first_call``: device int64 tensor of one element."""
    lib = L.load()
    first_call = _i64(first_call, "first_call")
    out = torch.empty(num_calls * rays_per_call, 3, dtype=torch.int64, device=first_call.device)
    L.check(lib.cn_pixel_sample(int(seed) & 0xFFFFFFFFFFFFFFFF, _p(first_call), num_calls, rays_per_call, num_cameras, height,
                                width, _p(out), _stream(first_call)))
    return out


def ray_sort_permutation(ray_indices: Tensor, height: int, width: int) -> Tensor:
    """Permutation that orders (camera, row, col) ray indices by camera and, inside a camera, along the Morton curve of the pixel
    (torch plumbing: bit spreads and one sort).  Random pixels of random cameras read the hash tables at random: the field pass
    of 65 536 such rays takes 1.77 ms against 0.73 ms for an image's coherent rays (its gathers miss the L2).  Sorted, the rays
    of a large launch are neighbours again -- 0.93 ms per 65 536 rays inside a 2^20-ray launch, 0.82 ms inside a 2^22-ray one
    (``tools/sorted_ray_probe.py``) -- and a ray's result does not depend on its neighbours, so un-permuting the outputs gives
    the same numbers in the same order."""
    idx = _i64(ray_indices, "ray_indices")

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555

    if height > (1 << 16) or width > (1 << 16):  # (host integers: nothing here waits for the device, so a graph can capture it)
        raise ValueError("ray_sort_permutation: images of at most 65 536 pixels a side")
    key = (idx[:, 0] << 32) | spread(idx[:, 1]) | (spread(idx[:, 2]) << 1)
    return torch.argsort(key)


def pointcloud_compact_calls(origins: Tensor, directions: Tensor, depth: Tensor, rgb: Tensor, semantics_colormap: Tensor,
                             rays_per_call: int, target_points: int, capacity: int,
                             buffers: Optional[Tuple[Tensor, ...]] = None) -> Tuple[Tensor, ...]:
    """``cn_pointcloud_compact_calls``: buffers = (points, colors, dirs, count, call_counts, ray_limit)."""
    lib = L.load()
    dev = origins.device
    R = origins.shape[0]
    calls = -(-R // rays_per_call)
    if buffers is None:
        buffers = (torch.empty(capacity, 3, device=dev), torch.empty(capacity, 3, device=dev), torch.empty(capacity, 3, device=dev),
                   torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(max(calls, 1), dtype=torch.int64, device=dev),
                   torch.zeros(1, dtype=torch.int64, device=dev))
    pts, cols, dirs, count, call_counts, ray_limit = buffers
    if call_counts.numel() < calls:
        raise ValueError(f"call_counts holds {call_counts.numel()} calls, the launch has {calls}")
    L.check(lib.cn_pointcloud_compact_calls(_p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
                                            _p(_f32(depth, "depth")), _p(_f32(rgb, "rgb")),
                                            _p(_f32(semantics_colormap, "semantics_colormap")), R, rays_per_call,
                                            target_points, capacity, _p(pts), _p(cols), _p(dirs), _p(count), _p(call_counts),
                                            _p(ray_limit), _stream(origins)))
    return buffers


# --------------------------------------------------------------------------------------------------------------
# training
# --------------------------------------------------------------------------------------------------------------

def train_render_backward(starts: Tensor, ends: Tensor, density: Tensor, rgb: Tensor, semantics: Tensor,
                          image: Tensor, fruit_mask: Tensor, semantic_loss_weight: float, loss_sums: Tensor,
                          spacing_bins: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """cn_train_render_backward: rendered values, per-sample gradients of rgb_loss + semantics_loss.  ``spacing_bins`` [R, S+1]:
    the distortion metric's sum over rays is added to ``loss_sums[4]`` as well (``loss_sums`` has five slots then)."""
    lib = L.load()
    R, S = starts.shape
    dev = starts.device
    if spacing_bins is not None and (tuple(spacing_bins.shape) != (R, S + 1) or loss_sums.numel() < 5):
        raise ValueError(f"train_render_backward: spacing_bins must be [{R},{S + 1}] and loss_sums hold five sums")
    out = {"rgb": torch.empty(R, 3, device=dev), "semantics": torch.empty(R, 1, device=dev),
           "accumulation": torch.empty(R, 1, device=dev), "weights": torch.empty(R, S, device=dev),
           "d_density": torch.empty(R, S, device=dev), "d_rgb": torch.empty(R, S, 3, device=dev),
           "d_semantics": torch.empty(R, S, device=dev)}
    L.check(lib.cn_train_render_backward(
        _p(_f32(starts, "starts")), _p(_f32(ends, "ends")), _p(_f32(density, "density")), _p(_f32(rgb, "rgb")),
        _p(_f32(semantics, "semantics")), _p(_f32(image, "image")), _p(_f32(fruit_mask, "fruit_mask")), R, S,
        float(semantic_loss_weight), _p(out["rgb"]), _p(out["semantics"]), _p(out["accumulation"]), _p(out["weights"]),
        _p(out["d_density"]), _p(out["d_rgb"]), _p(out["d_semantics"]), _p(_f32(loss_sums, "loss_sums")),
        _p(_f32(spacing_bins, "spacing_bins")), _stream(starts)))
    return out


def interlevel_backward(final_spacing_bins: Tensor, final_weights: Tensor, prop_spacing_bins: Tensor,
                        prop_starts: Tensor, prop_ends: Tensor, prop_density: Tensor, loss_mult: float,
                        loss_sum: Tensor) -> Tensor:
    lib = L.load()
    R, Sp = prop_density.shape
    Sf = final_weights.shape[1]
    d = torch.empty(R, Sp, device=prop_density.device)
    L.check(lib.cn_interlevel_backward(
        _p(_f32(final_spacing_bins, "final_spacing_bins")), _p(_f32(final_weights, "final_weights")),
        _p(_f32(prop_spacing_bins, "prop_spacing_bins")), _p(_f32(prop_starts, "prop_starts")),
        _p(_f32(prop_ends, "prop_ends")), _p(_f32(prop_density, "prop_density")), R, Sf, Sp, float(loss_mult), _p(d),
        _p(_f32(loss_sum, "loss_sum")), _stream(prop_density)))
    return d


def interlevel_backward_levels(final_spacing_bins: Tensor, final_weights: Tensor, levels: Sequence[Dict[str, Tensor]],
                               loss_mult: float, loss_sum: Tensor) -> List[Tensor]:
    """``cn_interlevel_backward_levels``: ``interlevel_backward`` for every proposal level (dicts with "bins", "starts", "ends",
    "density") in one launch; returns the levels' d loss / d density."""
    lib = L.load()
    R, Sf = final_weights.shape
    arr = (L.InterlevelLevel * len(levels))()
    outs, keep = [], []
    for k, lv in enumerate(levels):
        Sp = lv["density"].shape[1]
        if tuple(lv["bins"].shape) != (R, Sp + 1) or tuple(lv["starts"].shape) != (R, Sp) or tuple(lv["ends"].shape) != (R, Sp):
            raise ValueError(f"interlevel_backward_levels: level {k} has inconsistent shapes")
        d = torch.empty(R, Sp, device=final_weights.device)
        ts = [_f32(lv["bins"], "bins"), _f32(lv["starts"], "starts"), _f32(lv["ends"], "ends"), _f32(lv["density"], "density")]
        keep.append(ts)
        arr[k].spacing_bins, arr[k].starts, arr[k].ends, arr[k].density = (t.data_ptr() for t in ts)
        arr[k].d_density, arr[k].num_samples, arr[k].reserved = d.data_ptr(), Sp, 0
        outs.append(d)
    L.check(lib.cn_interlevel_backward_levels(_p(_f32(final_spacing_bins, "final_spacing_bins")),
                                              _p(_f32(final_weights, "final_weights")), arr, len(levels), R, Sf,
                                              float(loss_mult), _p(_f32(loss_sum, "loss_sum")), _stream(final_weights)))
    return outs


def field_backward(fh: FieldHandle, gh: FieldHandle, scene: L.Scene, origins: Tensor, directions: Tensor,
                   camera_indices: Optional[Tensor], starts: Tensor, ends: Tensor, d_density: Tensor, d_rgb: Tensor,
                   d_semantics: Tensor, app_mode: int = L.APP_PER_CAMERA, sh_unit_dir: bool = True,
                   app_mean: Optional[Tensor] = None, d_positions: Optional[Tensor] = None,
                   d_directions: Optional[Tensor] = None, matrix_precision: int = L.MATRIX_FP32) -> None:
    """Accumulates parameter gradients into the tensors behind ``gh`` (a FieldHandle over the gradient dict).
    ``d_positions`` / ``d_directions`` [R,S,3] (optional) are overwritten with the per-sample position / SH-direction
    gradients that feed the camera pose refinement.  ``matrix_precision``: ``MATRIX_F16`` = the reference's mixed-precision
    class (fp16 forward recompute, bf16 gradient products, fp32 sums: ``cn_field_backward_mp``)."""
    lib = L.load()
    R, S = starts.shape
    for t, nm in ((d_positions, "d_positions"), (d_directions, "d_directions")):
        if t is not None and tuple(t.shape) != (R, S, 3):
            raise ValueError(f"{nm} must be [{R},{S},3]")
    L.check(lib.cn_field_backward_mp(
        C.byref(fh.struct), C.byref(gh.struct), C.byref(scene), app_mode, 1 if sh_unit_dir else 0,
        _p(_f32(app_mean, "app_mean")), _p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
        _p(_i64(camera_indices, "camera_indices")), _p(_f32(starts, "starts")), _p(_f32(ends, "ends")),
        _p(_f32(d_density, "d_density")), _p(_f32(d_rgb, "d_rgb")), _p(_f32(d_semantics, "d_semantics")), R, S,
        _p(_f32(d_positions, "d_positions")), _p(_f32(d_directions, "d_directions")), int(matrix_precision), _stream(starts)))


def field_backward_general(fh: FieldHandle, gh: FieldHandle, scene: L.Scene, origins: Tensor, directions: Tensor,
                           camera_indices: Optional[Tensor], starts: Tensor, ends: Tensor, d_density: Tensor,
                           d_rgb: Tensor, d_semantics: Tensor, app_mode: int = L.APP_PER_CAMERA,
                           sh_unit_dir: bool = True, app_mean: Optional[Tensor] = None,
                           workspace: Optional[Tensor] = None, d_positions: Optional[Tensor] = None,
                           d_directions: Optional[Tensor] = None) -> Tensor:
    """``cn_field_backward_general``: parameter gradients for any field shape of the reference's configs (accumulated
    into the tensors behind ``gh``).  Returns the workspace so the caller can reuse it."""
    lib = L.load()
    R, S = starts.shape
    need = lib.cn_field_backward_general_workspace_bytes(C.byref(fh.struct))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=starts.device)
    L.check(lib.cn_field_backward_general(
        C.byref(fh.struct), C.byref(gh.struct), C.byref(scene), app_mode, 1 if sh_unit_dir else 0,
        _p(_f32(app_mean, "app_mean")), _p(_f32(origins, "origins")), _p(_f32(directions, "directions")),
        _p(_i64(camera_indices, "camera_indices")), _p(_f32(starts, "starts")), _p(_f32(ends, "ends")),
        _p(_f32(d_density, "d_density")), _p(_f32(d_rgb, "d_rgb")), _p(_f32(d_semantics, "d_semantics")), R, S,
        _p(_f32(d_positions, "d_positions")), _p(_f32(d_directions, "d_directions")),
        C.c_void_p(workspace.data_ptr()), workspace.numel(), _stream(starts)))
    return workspace


def proposal_backward(dh: DensityHandle, gh: DensityHandle, scene: L.Scene, origins: Tensor, directions: Tensor,
                      starts: Tensor, ends: Tensor, d_density: Tensor, d_positions: Optional[Tensor] = None) -> None:
    lib = L.load()
    R, S = starts.shape
    if d_positions is not None and tuple(d_positions.shape) != (R, S, 3):
        raise ValueError(f"d_positions must be [{R},{S},3]")
    L.check(lib.cn_proposal_backward(
        C.byref(dh.struct), C.byref(gh.struct), C.byref(scene), _p(_f32(origins, "origins")),
        _p(_f32(directions, "directions")), _p(_f32(starts, "starts")), _p(_f32(ends, "ends")),
        _p(_f32(d_density, "d_density")), R, S, _p(_f32(d_positions, "d_positions")), _stream(starts)))


def ray_backward(d_positions: Tensor, d_dir_samples: Optional[Tensor], starts: Tensor, ends: Tensor,
                 d_origins: Tensor, d_directions: Tensor) -> None:
    """Accumulate per-ray origin / direction gradients from per-sample position gradients (``cn_ray_backward``)."""
    lib = L.load()
    R, S = starts.shape
    if tuple(d_positions.shape) != (R, S, 3) or tuple(d_origins.shape) != (R, 3) or tuple(d_directions.shape) != (R, 3):
        raise ValueError("ray_backward: shape mismatch")
    L.check(lib.cn_ray_backward(_p(_f32(d_positions, "d_positions")), _p(_f32(d_dir_samples, "d_dir_samples")),
                                _p(_f32(starts, "starts")), _p(_f32(ends, "ends")), R, S,
                                _p(_f32(d_origins, "d_origins")), _p(_f32(d_directions, "d_directions")),
                                _stream(starts)))


def pose_adjustment_backward(pose_adjustment: Tensor, camera_indices: Tensor, directions_raw: Tensor,
                             d_origins: Tensor, d_directions: Tensor, grad_pose: Tensor) -> None:
    """Chain per-ray gradients through exp_map_SO3xR3 into ``grad_pose`` [C,6] (accumulated)."""
    lib = L.load()
    R = directions_raw.shape[0]
    if grad_pose.shape != pose_adjustment.shape:
        raise ValueError("grad_pose must have the shape of pose_adjustment")
    L.check(lib.cn_pose_adjustment_backward(
        _p(_f32(pose_adjustment, "pose_adjustment")), _p(_i64(camera_indices, "camera_indices")),
        _p(_f32(directions_raw, "directions_raw")), _p(_f32(d_origins, "d_origins")),
        _p(_f32(d_directions, "d_directions")), R, int(pose_adjustment.shape[0]), _p(_f32(grad_pose, "grad_pose")),
        _stream(directions_raw)))


def pose_regularizer(pose_adjustment: Tensor, grad_pose: Optional[Tensor], loss_out: Tensor,
                     trans_l2_penalty: float = 1e-2, rot_l2_penalty: float = 1e-3) -> None:
    """``camera_opt_regularizer``: adds the loss to ``loss_out`` [1] and its gradient to ``grad_pose``."""
    lib = L.load()
    L.check(lib.cn_pose_regularizer(_p(_f32(pose_adjustment, "pose_adjustment")), pose_adjustment.shape[0],
                                    float(trans_l2_penalty), float(rot_l2_penalty), _p(_f32(grad_pose, "grad_pose")),
                                    _p(_f32(loss_out, "loss_out")), _stream(pose_adjustment)))


def adam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15, zero_grad: bool = True) -> None:
    lib = L.load()
    L.check(lib.cn_adam_step(_p(_f32(param, "param")), _p(_f32(grad, "grad")), _p(_f32(exp_avg, "exp_avg")),
                             _p(_f32(exp_avg_sq, "exp_avg_sq")), param.numel(), int(step), float(lr), beta1, beta2,
                             eps, 1 if zero_grad else 0, _stream(param)))


def adam_hyper(step: int, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15, out: Optional[Tensor] = None
               ) -> Tensor:
    """``cn_adam_hyper``: the eight per-step floats ``cn_adam_step_dev`` reads, into a HOST tensor (pinned by the caller)."""
    lib = L.load()
    if out is None:
        out = torch.empty(8)
    if out.is_cuda or out.dtype != torch.float32 or out.numel() < 8 or not out.is_contiguous():
        raise TypeError("adam_hyper: out must be a contiguous float32 host tensor of 8 elements")
    L.check(lib.cn_adam_hyper(int(step), float(lr), beta1, beta2, eps, C.c_void_p(out.data_ptr())))
    return out


def adam_step_dev(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, hyper: Tensor, zero_grad: bool = True) -> None:
    """``cn_adam_step_dev``: ``adam_step`` with its per-step scalars in device memory (``hyper`` [8], see ``adam_hyper``)."""
    lib = L.load()
    L.check(lib.cn_adam_step_dev(_p(_f32(param, "param")), _p(_f32(grad, "grad")), _p(_f32(exp_avg, "exp_avg")),
                                 _p(_f32(exp_avg_sq, "exp_avg_sq")), param.numel(), _p(_f32(hyper, "hyper")),
                                 1 if zero_grad else 0, _stream(param)))


def adam_step_groups_dev(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, bounds: Sequence[int],
                         hyper: Tensor) -> None:
    """``cn_adam_step_groups_dev``: several optimiser groups of a flat buffer in one launch (``bounds``: ascending element offsets
    from 0, one more than groups; ``hyper`` [groups, 8] on the device, ``hyper[k][7] != 0``: group k takes no step, its gradients
    are only zeroed)."""
    lib = L.load()
    n = len(bounds) - 1
    if n < 1 or param.numel() < int(bounds[-1]) or tuple(hyper.shape) != (n, 8) or not hyper.is_contiguous():
        raise ValueError("adam_step_groups_dev: bounds exceed the buffer, or hyper is not a contiguous [groups, 8]")
    for t in (grad, exp_avg, exp_avg_sq):
        if t.numel() != param.numel():
            raise ValueError("adam_step_groups_dev: param / grad / moments differ in length")
    arr = (C.c_int64 * (n + 1))(*[int(b) for b in bounds])
    L.check(lib.cn_adam_step_groups_dev(_p(_f32(param, "param")), _p(_f32(grad, "grad")), _p(_f32(exp_avg, "exp_avg")),
                                        _p(_f32(exp_avg_sq, "exp_avg_sq")), C.cast(arr, C.c_void_p), n, _p(_f32(hyper, "hyper")),
                                        _stream(param)))


def radam_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-15, zero_grad: bool = True) -> None:
    """``torch.optim.RAdam`` (the ``_big`` / ``_huge`` methods' optimiser)."""
    lib = L.load()
    L.check(lib.cn_radam_step(_p(_f32(param, "param")), _p(_f32(grad, "grad")), _p(_f32(exp_avg, "exp_avg")),
                              _p(_f32(exp_avg_sq, "exp_avg_sq")), param.numel(), int(step), float(lr), beta1, beta2,
                              eps, 1 if zero_grad else 0, _stream(param)))


def distortion_metric(spacing_bins: Tensor, weights: Tensor, acc: Optional[Tensor] = None) -> Optional[Tensor]:
    """mean over rays of nerfstudio's distortion loss (get_metrics_dict "distortion").  With ``acc`` (one zeroed device float)
    only the SUM over rays is added there and None is returned: the training step divides it in ``cn_train_epilogue``."""
    lib = L.load()
    R, S = weights.shape
    own = acc is None
    if own:
        acc = torch.zeros(1, device=weights.device)
    elif acc.numel() != 1:
        raise ValueError("distortion_metric: acc is one float")
    L.check(lib.cn_distortion_metric(_p(_f32(spacing_bins, "spacing_bins")), _p(_f32(weights, "weights")), R, S,
                                     _p(_f32(acc, "acc")), _stream(weights)))
    return acc[0] / R if own else None


# --------------------------------------------------------------------------------------------------------------
# depth-based semantic projection (scripts/depth_based_semantic_projection.py)
# --------------------------------------------------------------------------------------------------------------

def _dev(t: Tensor, dtype, name: str) -> Tensor:
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name}: expected a contiguous {dtype} device tensor, got {t.dtype} {t.device}")
    return t


def depth_project(P: Tensor, points: Tensor, height: int, width: int) -> Tuple[Tensor, Tensor, Tensor]:
    """``get_projection`` + pixel rounding / clipping of ``update_buffer``: (xs rows, ys columns, zs), float64 in."""
    lib = L.load()
    n = points.shape[0]
    dev = points.device
    xs = torch.empty(n, dtype=torch.int32, device=dev)
    ys = torch.empty(n, dtype=torch.int32, device=dev)
    zs = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.cn_depth_project(_p(_dev(P, torch.float64, "P")), _p(_dev(points, torch.float64, "points")), n, height,
                                 width, _p(xs), _p(ys), _p(zs), _stream(points)))
    return xs, ys, zs


def zbuffer_update(z_buffer: Tensor, img: Tensor, xs: Tensor, ys: Tensor, zs: Tensor, label: int, large: bool = False,
                   want_visible: bool = True) -> Optional[Tensor]:
    """``update_buffer`` in place on ``z_buffer`` [H,W] float32 and ``img`` [H,W] uint8; returns the visible mask
    [H,W] uint8 (255 where a point was accepted) for ``large=False``."""
    lib = L.load()
    H, W = z_buffer.shape
    _dev(z_buffer, torch.float32, "z_buffer")
    _dev(img, torch.uint8, "img")
    ws = torch.empty(lib.cn_zbuffer_workspace_bytes(H, W), dtype=torch.uint8, device=z_buffer.device)
    n = xs.shape[0]
    if large:
        L.check(lib.cn_zbuffer_update_large(_p(_dev(xs, torch.int32, "xs")), _p(_dev(ys, torch.int32, "ys")),
                                            _p(_dev(zs, torch.float64, "zs")), n, int(label), H, W, _p(z_buffer),
                                            _p(img), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(z_buffer)))
        return None
    vis = torch.empty(H, W, dtype=torch.uint8, device=z_buffer.device) if want_visible else None
    L.check(lib.cn_zbuffer_update(_p(_dev(xs, torch.int32, "xs")), _p(_dev(ys, torch.int32, "ys")),
                                  _p(_dev(zs, torch.float64, "zs")), n, int(label), H, W, _p(z_buffer), _p(img), _p(vis),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), _stream(z_buffer)))
    return vis


# --------------------------------------------------------------------------------------------------------------
# statistical outlier removal (open3d remove_statistical_outlier as used by the point-cloud exporter)
# --------------------------------------------------------------------------------------------------------------

def _knn_grid(pts: Tensor, points_per_cell: float = 8.0, top_cells: int = 128):
    """The two-level grid of the k-nearest passes (``cn_point_grid``; torch plumbing: cell keys, one sort, offsets).  A dense TOP
    grid of at most ``top_cells`` cells per axis over the cloud's bounding box, its occupied cells numbered, and ``sub``^3 fine
    cells inside each occupied one, ``sub`` chosen so that an occupied FINE cell holds about ``points_per_cell`` points: an
    exported cloud is surfaces or clumps, not a volume -- the 10^7 kept points of the C4 bench sit in a few hundred cells of any
    dense grid that fits memory, 27 000 to a cell.  Returns (sorted points, the ``L.PointGrid`` struct, order, keep-alive tensors)."""
    n = pts.shape[0]
    lo, hi = pts.min(dim=0).values, pts.max(dim=0).values
    ext = (hi - lo).clamp(min=1e-12).double()
    # top cell: sized for points_per_cell points per cell if the cloud filled its box, at most top_cells per axis
    H = float((ext.prod() * points_per_cell / n) ** (1.0 / 3.0))
    H = max(H, float(ext.max()) / top_cells)
    top = [max(1, min(top_cells, int(float(e) / H) + 1)) for e in ext]
    tcell = ((pts - lo) / H).floor().to(torch.int64)
    for a_ in range(3):
        tcell[:, a_].clamp_(0, top[a_] - 1)
    tkey = (tcell[:, 2] * top[1] + tcell[:, 1]) * top[0] + tcell[:, 0]
    occ_keys, counts = torch.unique(tkey, return_counts=True)
    occupied = int(occ_keys.numel())
    # fine cells: points on a surface thin out with the SQUARE of the cell size; never more than 2^26 fine cells in all
    occupancy = n / occupied
    sub = 1
    while sub < 64 and occupancy / (sub * sub) > 1.5 * points_per_cell and occupied * (2 * sub) ** 3 <= (1 << 26):
        sub *= 2
    top_rank = torch.full((top[0] * top[1] * top[2],), -1, dtype=torch.int32, device=pts.device)
    top_rank[occ_keys] = torch.arange(occupied, dtype=torch.int32, device=pts.device)
    h = H / sub
    fcell = ((pts - lo) / h).floor().to(torch.int64)
    for a_ in range(3):
        fcell[:, a_].clamp_(0, top[a_] * sub - 1)
    sub_idx = ((fcell[:, 2] % sub) * sub + (fcell[:, 1] % sub)) * sub + (fcell[:, 0] % sub)
    # (the top cell of the FINE coordinates, so that both levels agree on points that sit on a cell face)
    tkey2 = ((fcell[:, 2] // sub) * top[1] + (fcell[:, 1] // sub)) * top[0] + (fcell[:, 0] // sub)
    rank = top_rank[tkey2].to(torch.int64)
    if bool((rank < 0).any()):  # a point whose fine cell lies in a top cell the coarse binning found empty (rounding at a face)
        extra = torch.unique(tkey2[rank < 0])
        top_rank[extra] = torch.arange(occupied, occupied + extra.numel(), dtype=torch.int32, device=pts.device)
        occupied += int(extra.numel())
        rank = top_rank[tkey2].to(torch.int64)
    key = rank * (sub ** 3) + sub_idx
    key_sorted, order = torch.sort(key)
    cell_start = torch.searchsorted(key_sorted, torch.arange(occupied * sub ** 3 + 1, device=pts.device)).to(torch.int32).contiguous()
    pts_sorted = pts[order].contiguous()
    g = L.PointGrid()
    for a_ in range(3):
        g.top[a_] = top[a_]
        g.origin[a_] = float(lo[a_])
    g.sub = sub
    g.fine_rings = max(2, sub)
    g.top_cell_size = H
    g.top_rank = top_rank.data_ptr()
    g.cell_start = cell_start.data_ptr()
    return pts_sorted, g, order, (top_rank, cell_start)


def estimate_normals(points: Tensor, knn: int = 30, points_per_cell: float = 8.0) -> Tuple[Tensor, Tensor]:
    """open3d ``PointCloud.estimate_normals()`` at its defaults (``KDTreeSearchParamKNN(30)``, fast 3 x 3 eigen-solver) on the
    device: (normals [N,3] float64 -- unit vectors, sign as the solver leaves it --, degenerate [N] bool: fewer than three
    neighbours or a zero covariance, normal (0, 0, 1) there).  ``cn_estimate_normals_grid`` on the two-level grid of the outlier
    pass."""
    lib = L.load()
    pts = _f32(points.contiguous(), "points")
    n = pts.shape[0]
    if n == 0:
        return torch.empty(0, 3, dtype=torch.float64, device=pts.device), torch.empty(0, dtype=torch.bool, device=pts.device)
    pts_sorted, grid, order, keep = _knn_grid(pts, points_per_cell)
    nrm_sorted = torch.empty(n, 3, dtype=torch.float64, device=pts.device)
    deg_sorted = torch.empty(n, dtype=torch.int32, device=pts.device)
    L.check(lib.cn_estimate_normals_grid(_p(pts_sorted), C.byref(grid), n, int(knn), _p(nrm_sorted), _p(deg_sorted), _stream(pts)))
    del keep
    normals = torch.empty_like(nrm_sorted)
    normals[order] = nrm_sorted
    degenerate = torch.empty(n, dtype=torch.bool, device=pts.device)
    degenerate[order] = deg_sorted != 0
    return normals, degenerate


def reorient_normals(normals: Tensor, view_directions: Tensor) -> Tuple[Tensor, Tensor]:
    """``exporter_utils_nerfacto.py:219-225``: in float32, flip every normal with ``sum(view_direction * normal) > 0`` (it
    would point away from the camera that saw the point); back to float64.  Returns (normals, flipped mask)."""
    nf = normals.to(torch.float32)
    mask = torch.sum(view_directions.to(torch.float32) * nf, dim=-1) > 0
    nf[mask] *= -1
    return nf.double(), mask


def knn_mean_distance(points: Tensor, nb_neighbors: int = 20, points_per_cell: float = 6.0) -> Tensor:
    """Mean distance of every point to its ``nb_neighbors`` nearest points (itself included), [N] float32, on a uniform
    two-level grid with about ``points_per_cell`` points per occupied fine cell (``_knn_grid``).  The binning (cell keys, sort,
    offsets) is torch plumbing; the search is ``cn_knn_mean_distance_grid``."""
    lib = L.load()
    pts = _f32(points.contiguous(), "points")
    n = pts.shape[0]
    if n == 0:
        return torch.empty(0, device=pts.device)
    pts_sorted, grid, order, keep = _knn_grid(pts, points_per_cell)
    mean_sorted = torch.empty(n, device=pts.device)
    L.check(lib.cn_knn_mean_distance_grid(_p(pts_sorted), C.byref(grid), n, int(nb_neighbors), _p(mean_sorted), _stream(pts)))
    del keep
    out = torch.empty_like(mean_sorted)
    out[order] = mean_sorted
    return out


def statistical_outlier_mask(points: Tensor, nb_neighbors: int = 20, std_ratio: float = 2.0) -> Tensor:
    """open3d ``PointCloud::RemoveStatisticalOutliers``: keep a point when its mean neighbour distance is below
    cloud mean + std_ratio * cloud std (sample std, N - 1).  Returns the boolean inlier mask [N]."""
    avg = knn_mean_distance(points, nb_neighbors).double()
    valid = avg > 0
    nv = int(valid.sum())
    if nv < 2:
        return valid
    mean = avg[valid].sum() / nv
    std = torch.sqrt(((avg[valid] - mean) ** 2).sum() / (nv - 1))
    return valid & (avg < mean + std_ratio * std)


# --------------------------------------------------------------------------------------------------------------
# super-cluster stage of the segmenter (segmentation/segmenter.py:69-86)
# --------------------------------------------------------------------------------------------------------------

def _bin_points(pts: Tensor, h: float, lo: Optional[Tensor] = None):
    """Uniform-grid binning (torch plumbing): returns (sorted points, cell_start int32, dims, origin, order)."""
    lo = pts.min(dim=0).values if lo is None else lo
    hi = pts.max(dim=0).values
    dims = [max(1, int(float((hi[a] - lo[a]) / h)) + 1) for a in range(3)]
    if dims[0] * dims[1] * dims[2] > (1 << 28):
        raise ValueError(f"grid of {dims} cells is too fine for this cloud")
    cell = ((pts - lo) / h).floor().to(torch.int64)
    for a in range(3):
        cell[:, a].clamp_(0, dims[a] - 1)
    key = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    key_sorted, order = torch.sort(key)
    cell_start = torch.searchsorted(key_sorted, torch.arange(dims[0] * dims[1] * dims[2] + 1, device=pts.device))
    return pts[order].contiguous(), cell_start.to(torch.int32).contiguous(), dims, lo, order.contiguous()


def voxel_down_sample(points: Tensor, voxel_size: float, colors: Optional[Tensor] = None):
    """open3d ``voxel_down_sample``: voxel index = floor((p - (min_bound - voxel_size / 2)) / voxel_size); the points (and
    colours) of a voxel are averaged.  Output order: ascending voxel key (open3d's is unspecified)."""
    lib = L.load()
    pts = _f32(points.contiguous(), "points")
    if pts.shape[0] == 0:
        return pts, colors
    vmin = pts.min(dim=0).values - voxel_size * 0.5
    cell = ((pts - vmin) / voxel_size).floor().to(torch.int64)
    dims = cell.max(dim=0).values + 1
    key = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    key_sorted, order = torch.sort(key)
    _, counts = torch.unique_consecutive(key_sorted, return_counts=True)
    seg = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=pts.device)
    seg[1:] = counts.cumsum(0).to(torch.int32)
    vals = pts[order] if colors is None else torch.cat([pts[order], _f32(colors.contiguous(), "colors")[order]], dim=1)
    vals = vals.contiguous()
    out = torch.empty(counts.numel(), vals.shape[1], device=pts.device)
    L.check(lib.cn_segment_mean(_p(vals), _p(seg), counts.numel(), vals.shape[1], _p(out), _stream(pts)))
    return (out[:, :3].contiguous(), None) if colors is None else (out[:, :3].contiguous(), out[:, 3:].contiguous())


def dbscan(points: Tensor, eps: float, min_points: int) -> Tuple[Tensor, Tensor]:
    """``cluster_dbscan(eps, min_points)``: labels [N] int64 (-1 = noise; clusters numbered by their smallest core point
    index, as a scan in point order discovers them) and the core mask [N]."""
    lib = L.load()
    pts = _f32(points.contiguous(), "points")
    n = pts.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=pts.device), torch.empty(0, dtype=torch.bool, device=pts.device)
    # sparse grid with cell diagonal <= eps (torch plumbing: keys, sort, run lengths)
    h = 0.99 * float(eps) / math.sqrt(3.0)
    lo = pts.min(dim=0).values
    cell = ((pts - lo) / h).floor().to(torch.int64)
    dims = [int(v) + 1 for v in cell.max(dim=0).values.tolist()]
    if max(dims) >= (1 << 20):
        raise ValueError(f"eps = {eps} is too small for a cloud of this extent (grid {dims})")
    key = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    key_sorted, order = torch.sort(key)
    cell_keys, point_cell, counts = torch.unique_consecutive(key_sorted, return_inverse=True, return_counts=True)
    m = cell_keys.numel()
    cell_start = torch.zeros(m + 1, dtype=torch.int32, device=pts.device)
    cell_start[1:] = torch.cumsum(counts, 0)
    ps = pts[order].contiguous()
    point_cell = point_cell.to(torch.int32)
    count = torch.empty(n, dtype=torch.int32, device=pts.device)
    parent = torch.empty(n, dtype=torch.int32, device=pts.device)
    root = torch.empty(n, dtype=torch.int32, device=pts.device)
    ws = torch.empty(lib.cn_dbscan_workspace_bytes(m), dtype=torch.uint8, device=pts.device)
    L.check(lib.cn_dbscan(_p(ps), _p(cell_keys), _p(cell_start), _p(point_cell), m, dims[0], dims[1], dims[2], h, float(eps),
                          int(min_points), _p(order), n, _p(count), _p(parent), _p(root), _p(ws), ws.numel(), _stream(pts)))
    core_sorted = count >= min_points
    # cluster ids in order of the smallest ORIGINAL index among each cluster's core points
    big = torch.iinfo(torch.int64).max
    first = torch.full((n,), big, dtype=torch.int64, device=pts.device)
    cs = core_sorted.nonzero().squeeze(1)
    first.scatter_reduce_(0, root[cs].to(torch.int64), order[cs], reduce="amin")
    roots = (first < big).nonzero().squeeze(1)
    rank = torch.empty_like(first)
    rank[roots[torch.argsort(first[roots])]] = torch.arange(roots.numel(), device=pts.device)
    lab_sorted = torch.where(root >= 0, rank[root.clamp(min=0).to(torch.int64)], torch.full_like(first, -1))
    labels = torch.empty(n, dtype=torch.int64, device=pts.device)
    labels[order] = lab_sorted
    core = torch.empty(n, dtype=torch.bool, device=pts.device)
    core[order] = core_sorted
    return labels, core


def contour_largest(gray: Tensor, thresh: int, roi: Optional[Tensor] = None, labels: Optional[Tensor] = None,
                    label_index: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """``cn_contour_largest`` on a stack of gray images [J,H,W] uint8 (device): per image the OpenCV contour of largest
    area inside ``roi`` [J,4] int32 (x0, y0, x1, y1) -> ``area`` [J], ``bbox`` [J,4] (x, y, w, h), ``start`` [J]; with
    ``labels`` [C,H,W] uint8 (+ ``label_index`` [J] int32) also ``vertex_count``, ``label``, ``label_count`` [J]."""
    lib = L.load()
    if gray.dtype != torch.uint8 or not gray.is_cuda or not gray.is_contiguous() or gray.dim() != 3:
        raise TypeError("contour_largest: expected a contiguous uint8 device tensor [J,H,W]")
    J, H, W = gray.shape
    dev = gray.device
    out = {"area": torch.empty(J, device=dev), "bbox": torch.empty(J, 4, dtype=torch.int32, device=dev),
           "start": torch.empty(J, dtype=torch.int32, device=dev)}
    if roi is not None:
        roi = _dev(roi, torch.int32, "roi")
        if tuple(roi.shape) != (J, 4):
            raise ValueError(f"roi shape {tuple(roi.shape)} != {(J, 4)}")
    if labels is not None:
        labels = _dev(labels, torch.uint8, "labels")
        if labels.dim() != 3 or tuple(labels.shape[1:]) != (H, W):
            raise ValueError("labels: expected [C,H,W] uint8 frames of the image size")
        if label_index is None:
            if labels.shape[0] != J:
                raise ValueError("label_index is required when the label stack is not one frame per image")
        else:
            label_index = _dev(label_index, torch.int32, "label_index")
            if J and (int(label_index.min()) < 0 or int(label_index.max()) >= labels.shape[0]):
                raise ValueError("label_index out of range")
        for k_ in ("vertex_count", "label", "label_count"):
            out[k_] = torch.empty(J, dtype=torch.int32, device=dev)
    ws = torch.empty(int(lib.cn_contour_workspace_bytes(J, H, W)) + 8, dtype=torch.uint8, device=dev)
    L.check(lib.cn_contour_largest(_p(gray), _p(roi), J, H, W, int(thresh), _p(labels), _p(label_index), _p(out["area"]),
                                   _p(out["bbox"]), _p(out["start"]), _p(out.get("vertex_count")), _p(out.get("label")),
                                   _p(out.get("label_count")), _p(ws), ws.numel(), _stream(gray)))
    return out


def kmeans(points: Tensor, k: int, max_iter: int = 300, tol: float = 1e-4, random_state: int = 0) -> Tensor:
    """``sklearn.cluster.KMeans(init="k-means++", n_clusters=k, n_init="auto", random_state=0).fit(points).labels_``
    (``segmentation/segmenter.py:41-43``) with the Lloyd iterations on the device (``cn_kmeans_step``, float64 as sklearn).
    Seeding is scikit-learn's own ``kmeans_plusplus`` on the host with the same random stream (a few weighted draws over
    the points); centring, the scaled tolerance and the two stopping rules follow ``_kmeans_single_lloyd``.  Returns
    labels [N] int64 on the device."""
    import numpy as np
    from sklearn.cluster import kmeans_plusplus

    lib = L.load()
    if points.dtype not in (torch.float32, torch.float64) or not points.is_cuda:
        raise TypeError("kmeans: expected a float device tensor [N,3]")
    n = points.shape[0]
    if n < k:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")  # sklearn's message
    x = points.to(torch.float64)
    x = (x - x.mean(dim=0)).contiguous()  # KMeans.fit centres the data before anything else
    xh = x.cpu().numpy()
    tol_ = float(np.mean(np.var(xh, axis=0)) * tol)  # _tolerance
    centers_h, _ = kmeans_plusplus(xh, k, random_state=np.random.RandomState(random_state))
    centers = torch.from_numpy(np.ascontiguousarray(centers_h)).to(x.device)
    labels = torch.full((n,), -1, dtype=torch.int32, device=x.device)
    sums = torch.empty(k, 3, dtype=torch.float64, device=x.device)
    counts = torch.empty(k, dtype=torch.int64, device=x.device)
    changed = torch.empty(1, dtype=torch.int32, device=x.device)
    strict = False
    for it in range(max_iter):
        sums.zero_(), counts.zero_(), changed.zero_()
        L.check(lib.cn_kmeans_step(_p(x), n, _p(centers), k, _p(labels), _p(sums), _p(counts), _p(changed), 1, _stream(x)))
        if bool((counts == 0).any()):
            # _relocate_empty_clusters: an empty cluster takes the point farthest from its own centre (rare)
            lab = labels.to(torch.int64)
            d = ((x - centers[lab]) ** 2).sum(dim=1)
            empty = (counts == 0).nonzero().squeeze(1)
            far = torch.argsort(d, descending=True)[: empty.numel()]
            for e, f in zip(empty.tolist(), far.tolist()):
                old = int(lab[f])
                sums[old] -= x[f]
                counts[old] -= 1
                sums[e] = x[f]
                counts[e] = 1
        new = sums / counts.to(torch.float64)[:, None]
        shift = float(((new - centers) ** 2).sum())
        centers = new.contiguous()
        if it > 0 and int(changed.item()) == 0:
            strict = True  # labels equal to the previous iteration's: sklearn's strict convergence
            break
        if shift <= tol_:
            break
    if not strict:  # rerun the E-step so that the labels match the final centres
        changed.zero_()
        L.check(lib.cn_kmeans_step(_p(x), n, _p(centers), k, _p(labels), None, None, _p(changed), 0, _stream(x)))
    return labels.to(torch.int64)


def get_super_clusters(points: Tensor, vx_size: float = 10e-5, colors: Optional[Tensor] = None):
    """``segmentation/segmenter.py:69-86``: voxel down-sample, DBSCAN(eps = 20 voxels, min_points = 30), drop the noise,
    statistical outlier removal (20 neighbours, std_ratio 2).  Returns (points, labels)."""
    pts, _ = voxel_down_sample(points, vx_size, colors)
    labels, _ = dbscan(pts, 20 * vx_size, 30)
    keep = labels >= 0
    pts, labels = pts[keep].contiguous(), labels[keep]
    inl = statistical_outlier_mask(pts, 20, 2.0)
    return pts[inl].contiguous(), labels[inl]
