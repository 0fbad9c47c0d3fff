"""``FruitField`` -- mirror of ``crop_nerf/fruit_nerf/fruit_field.py:71-302`` on ``cn_field_eval`` (materialised
per-sample outputs; the renderers use the fused kernel instead and never build RaySamples)."""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .. import _lib as L
from .. import ops
from ..config import FieldSpec
from ..rays import RaySamples, SceneBox


class FruitField:
    def __init__(self, aabb: Tensor, params: Dict[str, Tensor], spec: FieldSpec, spatial_distortion: bool = True,
                 test_mode: Optional[str] = None, training: bool = False) -> None:
        self.aabb = aabb
        self.spec = spec
        self.handle = ops.FieldHandle(params, spec)
        self.spatial_distortion = spatial_distortion  # SceneContraction(inf) on / off (None in the reference)
        self.test_mode = test_mode
        self.training = training

    def _app_mode(self) -> int:
        if self.test_mode in ("inference", "export"):
            return L.APP_MEAN
        if self.training:
            return L.APP_PER_CAMERA
        return L.APP_MEAN if self.spec.use_average_appearance_embedding else L.APP_ZEROS

    def forward(self, ray_samples: RaySamples) -> Dict[str, Tensor]:
        """``fruit_field.py:284-302``: density [R,S,1], rgb [R,S,3], semantics [R,S,1]."""
        mode = self._app_mode()
        if mode != L.APP_MEAN or self.test_mode not in ("inference", "export"):
            if ray_samples.camera_indices is None:
                raise AttributeError("Camera indices are not provided.")
        cam = None if ray_samples.camera_indices is None else ray_samples.camera_indices.reshape(-1).contiguous()
        scene = ops.scene_struct(self.aabb, bool(self.spatial_distortion))
        out = ops.field_eval(self.handle, scene, ray_samples.origins, ray_samples.directions, cam,
                             ray_samples.starts[..., 0].contiguous(), ray_samples.ends[..., 0].contiguous(),
                             app_mode=mode, sh_unit_dir=self.spec.sh_input == "unit")
        return {"density": out["density"][..., None], "rgb": out["rgb"], "semantics": out["semantics"][..., None]}

    __call__ = forward

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, None]:
        return self.forward(ray_samples)["density"], None
