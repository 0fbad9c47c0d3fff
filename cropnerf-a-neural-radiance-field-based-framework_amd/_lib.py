"""ctypes binding of libcropnerf_hip.so (declarations mirror include/cropnerf_hip.h one to one).

The library is the product: there is no Python/CPU fallback.  If it is missing or a call fails, an exception is
raised (``CropNerfHipError``) -- nothing silently degrades to PyTorch ops.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

CN_MAX_LEVELS = 16
CN_MAX_LAYERS = 4

SPACING_UNIFORM, SPACING_PIECEWISE = 0, 1
APP_ZEROS, APP_MEAN, APP_PER_CAMERA = 0, 1, 2
BG_LAST_SAMPLE, BG_COLOR = 0, 1

CN_ERR_INVALID, CN_ERR_UNSUPPORTED, CN_ERR_LAUNCH, CN_ERR_WORKSPACE = -1, -2, -3, -4

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libcropnerf_hip.so"


class CropNerfHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libcropnerf_hip error {code}: {message}")
        self.code = code
        self.message = message


GRID_TORCH, GRID_TCNN = 0, 1  # cn_grid.layout
TABLE_F32, TABLE_F16 = 0, 1  # cn_grid.table_dtype


class Grid(C.Structure):
    _fields_ = [("table", C.c_void_p), ("num_levels", C.c_int32), ("log2_table_size", C.c_int32),
                ("scalings", C.c_float * CN_MAX_LEVELS), ("layout", C.c_int32), ("table_dtype", C.c_int32),
                ("level_offset", C.c_uint32 * CN_MAX_LEVELS), ("level_bits", C.c_uint8 * CN_MAX_LEVELS),
                ("scatter_scratch", C.c_void_p), ("scatter_scratch_bytes", C.c_uint64)]


class TcnnGridPlan(C.Structure):
    _fields_ = [("num_levels", C.c_int32), ("log2_table_size", C.c_int32), ("base_resolution", C.c_int32),
                ("per_level_scale", C.c_float), ("scalings", C.c_float * CN_MAX_LEVELS),
                ("resolution", C.c_uint32 * CN_MAX_LEVELS), ("packed_offset", C.c_uint32 * (CN_MAX_LEVELS + 1)),
                ("level_offset", C.c_uint32 * (CN_MAX_LEVELS + 1)), ("level_bits", C.c_uint8 * CN_MAX_LEVELS)]


class Mlp(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("dims", C.c_int32 * (CN_MAX_LAYERS + 1)),
                ("weight", C.c_void_p * CN_MAX_LAYERS), ("bias", C.c_void_p * CN_MAX_LAYERS)]


class FieldParams(C.Structure):
    _fields_ = [("grid", Grid), ("base", Mlp), ("semantics", Mlp), ("sem_head_weight", C.c_void_p),
                ("sem_head_bias", C.c_void_p), ("color", Mlp), ("appearance", C.c_void_p),
                ("num_images", C.c_int32), ("app_dim", C.c_int32), ("geo_feat_dim", C.c_int32)]


class DensityParams(C.Structure):
    _fields_ = [("grid", Grid), ("mlp", Mlp)]


class ProposalLevelOut(C.Structure):
    """``cn_proposal_level_out``"""
    _fields_ = [("spacing_bins", C.c_void_p), ("starts", C.c_void_p), ("ends", C.c_void_p), ("density", C.c_void_p)]


class Scene(C.Structure):
    _fields_ = [("aabb", C.c_float * 6), ("contraction", C.c_int32)]


class RenderOpts(C.Structure):
    _fields_ = [("num_samples", C.c_int32), ("spacing", C.c_int32), ("bg_mode", C.c_int32),
                ("bg_color", C.c_float * 3), ("app_mode", C.c_int32), ("sh_unit_dir", C.c_int32),
                ("eval_clamp", C.c_int32), ("density_only", C.c_int32), ("image_width", C.c_int32),
                ("pixel_start", C.c_int64), ("early_stop_transmittance", C.c_float), ("matrix_precision", C.c_int32)]


MATRIX_FP32, MATRIX_SPLIT_BF16, MATRIX_F16 = 0, 1, 2  # cn_render_opts.matrix_precision


class ProjectionJob(C.Structure):
    """``cn_projection_job`` (a device array: filled on the host, copied over as bytes)"""
    _fields_ = [("c2w", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("aabb", C.c_float * 6), ("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("camera_index", C.c_int32), ("reserved", C.c_int32), ("slot_offset", C.c_int64)]


class PointGrid(C.Structure):
    """``cn_point_grid``: the two-level grid of the k-nearest passes"""
    _fields_ = [("top", C.c_int32 * 3), ("sub", C.c_int32), ("fine_rings", C.c_int32), ("origin", C.c_float * 3),
                ("top_cell_size", C.c_float), ("top_rank", C.c_void_p), ("cell_start", C.c_void_p)]


class InterlevelLevel(C.Structure):
    """``cn_interlevel_level`` (a host array read by ``cn_interlevel_backward_levels`` at call time)"""
    _fields_ = [("spacing_bins", C.c_void_p), ("starts", C.c_void_p), ("ends", C.c_void_p), ("density", C.c_void_p),
                ("d_density", C.c_void_p), ("num_samples", C.c_int32), ("reserved", C.c_int32)]


_P = C.c_void_p
_I32, _I64, _F = C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); the not-gpu test checks every symbol of the header is exported and listed here
SIGNATURES = {
    "cn_last_error": (C.c_char_p, []),
    "cn_version": (C.c_int, []),
    "cn_tcnn_grid_plan_init": (C.c_int, [_I32, _I32, _I32, _F, C.POINTER(TcnnGridPlan)]),
    "cn_tcnn_grid_describe": (C.c_int, [C.POINTER(TcnnGridPlan), _P, _I32, C.POINTER(Grid)]),
    "cn_tcnn_grid_pack": (C.c_int, [C.POINTER(TcnnGridPlan), _P, _I32, _P, _I32, _P]),
    "cn_tcnn_grid_unpack": (C.c_int, [C.POINTER(TcnnGridPlan), _P, _I32, _P, _I32, _P]),
    "cn_tcnn_grid_tie_gradients": (C.c_int, [C.POINTER(TcnnGridPlan), _P, _P]),
    "cn_tcnn_grid_tie_parameters": (C.c_int, [C.POINTER(TcnnGridPlan), _P, _P]),
    "cn_raygen_pinhole": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I64, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "cn_intersect_aabb": (C.c_int, [_P, _P, C.POINTER(_F), _I64, _P, _P, _P]),
    "cn_raygen_ortho": (C.c_int, [_P, C.POINTER(_F), _I64, _I64, _P, _P, _P, _P, _P, _P]),
    "cn_surface_grid": (C.c_int, [_F, _F, _I32, _F, _F, _I32, _F, _P, _P]),
    "cn_apply_pose_adjustment": (C.c_int, [_P, _P, _I64, _P, _P, _P]),
    "cn_apply_pose_adjustment_to": (C.c_int, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    "cn_train_epilogue": (C.c_int, [_P, _I32, _I64, _I32, _F, _F, _P, _I32, _P, _P]),
    "cn_projection_test": (C.c_int, [_P, _I32, _I64, _I32, _I32, _P, _P, _P, _P]),
    "cn_projection_gather": (C.c_int, [_P, _P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cn_projection_scatter": (C.c_int, [_P, _P, _P, _I64, _F, _P, _P, _P, _P, _P]),
    "cn_png_write_gray_rects": (C.c_int, [_I32, C.POINTER(C.c_char_p), _P, _P, _P, _I32, _I32, _I32]),
    "cn_projection_paste": (C.c_int, [_P, _P, _P, _I64, _P, _I32, _I32, _I32, _P, _P]),
    "cn_sample_spaced": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _I32, _P, _P, _P, _P, _P]),
    "cn_sample_pdf": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _I32, _F, _I32, _P, _I32, _P, _P, _P]),
    "cn_proposal_density": (C.c_int, [C.POINTER(DensityParams), C.POINTER(Scene), _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "cn_field_eval": (C.c_int, [C.POINTER(FieldParams), C.POINTER(Scene), _I32, _I32, _P, _P, _P, _P, _P, _I64, _I32,
                                _P, _P, _P, _P, _P]),
    "cn_field_eval_mp": (C.c_int, [C.POINTER(FieldParams), C.POINTER(Scene), _I32, _I32, _P, _P, _P, _P, _P, _I64, _I32,
                                   _P, _P, _P, _P, _I32, _P]),
    "cn_composite": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, C.POINTER(_F), _I32, _P, _P, _P, _P, _P, _P, _P]),
    "cn_render_workspace_bytes": (C.c_size_t, [C.POINTER(FieldParams)]),
    "cn_render_rays": (C.c_int, [C.POINTER(FieldParams), C.POINTER(Scene), C.POINTER(RenderOpts), _P, _P, _P, _P, _P,
                                 _P, _I64, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_render_samples": (C.c_int, [C.POINTER(FieldParams), C.POINTER(Scene), C.POINTER(RenderOpts), _P, _P, _P, _P,
                                    _P, _P, _I64, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_proposal_sample_workspace_bytes": (C.c_size_t, [_I64, C.POINTER(_I32), _I32, _I32]),
    "cn_proposal_sample": (C.c_int, [C.POINTER(C.POINTER(DensityParams)), _I32, C.POINTER(Scene), _P, _P, _P, _P, _I64,
                                     C.POINTER(_I32), _I32, _F, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_proposal_sample_mp": (C.c_int, [C.POINTER(C.POINTER(DensityParams)), _I32, C.POINTER(Scene), _P, _P, _P, _P, _I64,
                                        C.POINTER(_I32), _I32, _F, _P, _P, _P, _I32, _P]),
    "cn_grid_scatter_scratch_bytes": (C.c_size_t, [C.POINTER(Grid)]),
    "cn_grid_scatter_scratch_bytes_for": (C.c_size_t, [C.POINTER(Grid), _I64]),
    "cn_proposal_sample_train": (C.c_int, [C.POINTER(C.POINTER(DensityParams)), _I32, C.POINTER(Scene), _P, _P, _P, _P,
                                           _I64, C.POINTER(_I32), _I32, _F, _P, C.POINTER(ProposalLevelOut), _P, _P, _P, _P,
                                           _P]),
    "cn_proposal_sample_train_dev": (C.c_int, [C.POINTER(C.POINTER(DensityParams)), _I32, C.POINTER(Scene), _P, _P, _P, _P,
                                               _I64, C.POINTER(_I32), _I32, _P, _P, C.POINTER(ProposalLevelOut), _P, _P, _P, _P,
                                               _P]),
    "cn_adam_hyper": (C.c_int, [_I32, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "cn_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _I64, _P, _I32, _P]),
    "cn_adam_step_groups_dev": (C.c_int, [_P, _P, _P, _P, _P, _I32, _P, _P]),
    "cn_export_compact": (C.c_int, [_P, _P, _P, _P, _I64, _F, _F, _I64, C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "cn_pointcloud_compact": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "cn_pixel_sample": (C.c_int, [C.c_uint64, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "cn_pointcloud_compact_calls": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "cn_embedding_mean": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "cn_train_render_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F, _P, _P, _P, _P, _P, _P, _P, _P,
                                           _P, _P]),
    "cn_interlevel_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _F, _P, _P, _P]),
    "cn_interlevel_backward_levels": (C.c_int, [_P, _P, C.POINTER(InterlevelLevel), _I32, _I64, _I32, _F, _P, _P]),
    "cn_field_backward": (C.c_int, [C.POINTER(FieldParams), C.POINTER(FieldParams), C.POINTER(Scene), _I32, _I32, _P,
                                    _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "cn_field_backward_mp": (C.c_int, [C.POINTER(FieldParams), C.POINTER(FieldParams), C.POINTER(Scene), _I32, _I32, _P,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _I32, _P]),
    "cn_proposal_backward": (C.c_int, [C.POINTER(DensityParams), C.POINTER(DensityParams), C.POINTER(Scene), _P, _P,
                                       _P, _P, _P, _I64, _I32, _P, _P]),
    "cn_field_backward_general_workspace_bytes": (C.c_size_t, [C.POINTER(FieldParams)]),
    "cn_field_backward_general": (C.c_int, [C.POINTER(FieldParams), C.POINTER(FieldParams), C.POINTER(Scene), _I32, _I32,
                                            _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P, _P, _P, C.c_size_t, _P]),
    "cn_ray_backward": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "cn_pose_adjustment_backward": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _P, _P]),
    "cn_pose_regularizer": (C.c_int, [_P, _I32, _F, _F, _P, _P, _P]),
    "cn_distortion_metric": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "cn_depth_project": (C.c_int, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "cn_zbuffer_workspace_bytes": (C.c_size_t, [_I32, _I32]),
    "cn_zbuffer_update_large": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, C.c_size_t, _P]),
    "cn_zbuffer_update": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_knn_mean_distance": (C.c_int, [_P, _P, _I32, _I32, _I32, _F, _F, _F, _F, _I64, _I32, _P, _P]),
    "cn_estimate_normals": (C.c_int, [_P, _P, _I32, _I32, _I32, _F, _F, _F, _F, _I64, _I32, _P, _P, _P]),
    "cn_knn_mean_distance_grid": (C.c_int, [_P, C.POINTER(PointGrid), _I64, _I32, _P, _P]),
    "cn_estimate_normals_grid": (C.c_int, [_P, C.POINTER(PointGrid), _I64, _I32, _P, _P, _P]),
    "cn_segment_mean": (C.c_int, [_P, _P, _I64, _I32, _P, _P]),
    "cn_dbscan_workspace_bytes": (C.c_size_t, [_I64]),
    "cn_dbscan": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _F, _I32, _P, _I64, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_contour_workspace_bytes": (C.c_size_t, [_I32, _I32, _I32]),
    "cn_contour_largest": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "cn_kmeans_step": (C.c_int, [_P, _I64, _P, _I32, _P, _P, _P, _P, _I32, _P]),
    "cn_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _I32, C.c_double, C.c_double, C.c_double, C.c_double, _I32, _P]),
    "cn_radam_step": (C.c_int, [_P, _P, _P, _P, _I64, _I32, C.c_double, C.c_double, C.c_double, C.c_double, _I32, _P]),
    "cn_deterministic_build": (C.c_int, []),
    "cn_deterministic_register": (C.c_int, [_P, _I64, _P, _P]),
    "cn_deterministic_clear": (C.c_int, []),
    "cn_deterministic_flush": (C.c_int, [_P]),
}

DET_LIB_PATH = _HERE / "libcropnerf_hip_det.so"
_libs: dict = {}


def deterministic() -> bool:
    """``CN_DETERMINISTIC_SCATTER=1`` (read per call, so one test process can run both modes): calls go to the test build whose
    training kernels accumulate order-independently (``csrc/cn_det.hpp``, ``include/cropnerf_hip.h``).  Never set by the
    product path."""
    return os.environ.get("CN_DETERMINISTIC_SCATTER", "0") not in ("", "0")


def load() -> C.CDLL:
    """Load the shared library (built in-tree by build.py).  Raises if it is absent: no fallback path exists."""
    det = deterministic()
    lib = _libs.get(det)
    if lib is not None:
        return lib
    path = DET_LIB_PATH if det else Path(os.environ.get("CROPNERF_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found: build it with `python {(_HERE / 'build.py')}` (hipcc --offload-arch=gfx950). "
            "The HIP library is required; there is no CPU or PyTorch fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if bool(lib.cn_deterministic_build()) != det:
        raise RuntimeError(f"{path}: cn_deterministic_build() says {lib.cn_deterministic_build()}, expected {int(det)}")
    _libs[det] = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().cn_last_error()
        raise CropNerfHipError(rc, msg.decode() if msg else "")
