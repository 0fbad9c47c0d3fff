"""FruitModel -- host-side mirror of the reference's model for the hot path, running on libcropnerf_hip.

Same names, argument meaning, output-dict keys and error behaviour as ``crop_nerf/fruit_nerf/fruit_nerf.py``
(``FruitModel``: ``forward :617-637``, ``get_outputs :543-599``, ``get_inference_outputs :497-541``,
``get_export_outputs :476-494``, ``setup_inference :185-189``, ``get_param_groups :191-196``,
``get_outputs_for_camera_ray_bundle :377-404``, ``get_outputs_for_camera_jagged_ray_bundle :346-374``,
``get_density_for_camera_ray_bundle :320-344``, ``get_outputs_for_projections :254-318``,
``get_metrics_dict :639-645``).  Every number comes from a HIP kernel; this file is control flow only.

Differences kept deliberately: outputs stay on the model's device until the caller asks for
them (one D2H per image instead of one per chunk); ``depth`` is not forced to ``"cuda"``; artefact paths of the
projection pass are arguments with the reference's values as defaults; ``compat_projection_cam0`` reproduces the
reference's use of camera index 0 for every projected camera.
"""

from __future__ import annotations

import contextlib
import os
import shutil
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import _lib as L
from .. import ops
from ..config import FruitNerfModelConfig, init_params, param_shapes
from ..rays import Cameras, RayBundle, SceneBox

__all__ = ["FruitNerfModelConfig", "FruitModel", "Semantics", "background_color_override_context"]

_BACKGROUND_COLOR_OVERRIDE: Optional[Tuple[float, float, float]] = None


@contextlib.contextmanager
def background_color_override_context(color):
    """nerfstudio ``renderers.background_color_override_context`` (used at
    ``fruit_nerf/scripts/semantic_projection.py:169``)."""
    global _BACKGROUND_COLOR_OVERRIDE
    old = _BACKGROUND_COLOR_OVERRIDE
    _BACKGROUND_COLOR_OVERRIDE = tuple(float(c) for c in (color.tolist() if isinstance(color, Tensor) else color))
    try:
        yield
    finally:
        _BACKGROUND_COLOR_OVERRIDE = old


class Semantics:
    """``nerfstudio.data.utils.dataparsers...Semantics`` as built at ``data/cotton_nerf_dataparser.py:244-254``."""

    def __init__(self, filenames: Sequence[str] = (), classes: Sequence[str] = ("apple", "stuff"),
                 colors: Optional[Tensor] = None, mask_classes: Sequence[str] = ()):
        self.filenames = list(filenames)
        self.classes = list(classes)
        self.colors = torch.tensor([0.0, 1.0]) if colors is None else colors
        self.mask_classes = list(mask_classes)


class TrainingCallback:
    """nerfstudio ``TrainingCallback``: ``func(step)`` at the named locations, every ``update_every_num_iters`` steps."""

    def __init__(self, where_to_run: Sequence[str], update_every_num_iters: Optional[int], func, iters=None):
        self.where_to_run, self.update_every_num_iters, self.func, self.iters = list(where_to_run), update_every_num_iters, func, iters

    def run_callback(self, step: int) -> None:
        if self.update_every_num_iters is not None:
            if step % self.update_every_num_iters == 0:
                self.func(step)
        elif self.iters is not None and step in self.iters:
            self.func(step)

    def run_callback_at_location(self, step: int, location: str) -> None:
        if location in self.where_to_run:
            self.run_callback(step)


class FruitModel:
    """The reference's ``FruitModel`` (``fruit_nerf.py:73-700``) on the HIP kernels: every forward variant, the chunked
    image / projection renders and the loss / metric dictionaries; the backward pass lives in ``trainer.FruitTrainer``."""

    def __init__(self, config: FruitNerfModelConfig, scene_box: SceneBox, num_train_data: int, metadata: Dict,
                 device: Union[str, torch.device] = "cuda", grad_scaler=None, test_mode: str = "val",
                 render_rgb_inference: bool = True, params: Optional[Dict[str, Tensor]] = None, seed: int = 0,
                 **kwargs) -> None:
        assert "semantics" in metadata.keys() and isinstance(metadata["semantics"], Semantics)
        self.semantics = metadata["semantics"]
        self.test_mode = test_mode
        self.config = config
        self.scene_box = scene_box
        self.num_train_data = num_train_data
        self.device = torch.device(device)
        self.training = False
        self.compat_projection_cam0 = False
        self.colormap = self.semantics.colors.clone().detach().to(self.device)
        self._given_params = params
        self._seed = seed
        self.populate_modules()

    # ------------------------------------------------------------------------------------------ modules / params
    def populate_modules(self) -> None:
        cfg = self.config
        self.field_spec = cfg.field_spec(self.num_train_data)
        self.proposal_specs = cfg.proposal_specs()
        shapes = param_shapes(self.field_spec, self.proposal_specs)
        table_dtype = {"float32": torch.float32, "float16": torch.float16}[cfg.hash_table_dtype]
        if self._given_params is not None:
            # hash tables keep a half dtype (the values tcnn computes with); everything else is float32
            params = {k: v.to(self.device, v.dtype if k.endswith("hash_table") and v.dtype == torch.float16
                              else torch.float32).contiguous() for k, v in self._given_params.items()}
            for k, shp in shapes.items():
                if tuple(params[k].shape) != tuple(shp):
                    raise ValueError(f"parameter {k}: shape {tuple(params[k].shape)} != {tuple(shp)}")
        else:
            params = init_params(self.field_spec, self.proposal_specs, seed=self._seed, device=self.device)
            if cfg.implementation == "tcnn":  # several table entries stand for one tcnn parameter: make them agree
                ops.tcnn_grid_tie_parameters(self.field_spec.grid, params["field.mlp_base_grid.hash_table"])
                for i, ps in enumerate(self.proposal_specs):
                    ops.tcnn_grid_tie_parameters(ps.grid, params[f"proposal_networks.{i}.encoding.hash_table"])
            if table_dtype != torch.float32:
                for k in list(params):
                    if k.endswith("hash_table"):
                        params[k] = params[k].to(table_dtype)
        self.params = params
        self.field = ops.FieldHandle(params, self.field_spec)
        self.proposal_networks = [ops.DensityHandle(params, i, ps) for i, ps in enumerate(self.proposal_specs)]
        self._field_contraction = not cfg.disable_scene_contraction
        self._prop_contraction = not cfg.disable_scene_contraction
        self._uniform_samples: Optional[int] = None
        self._anneal = 1.0
        self.render_rgb = True
        self._image_hint: Tuple[int, int] = (0, 0)  # (image width, first pixel) of the chunk being rendered
        # The fused kernels are specialised for the default fruit_nerf_method field; fruit_nerf_method_big / _huge
        # (geo_feat_dim 30, 3 x 128 semantic layers: fruit_nerf_config.py:66-172) run the same path through the
        # shape-generic kernels (sampler -> cn_field_eval -> cn_composite, [R,S,.] tensors materialised per sub-chunk).
        fs = self.field_spec
        self._fused_shape = (fs.grid.num_levels == 16 and fs.grid.features_per_level == 2 and fs.hidden_dim == 64
                             and fs.geo_feat_dim == 15 and fs.num_layers_semantic == 2
                             and fs.hidden_dim_semantics == 64 and fs.hidden_dim_transient == 64
                             and fs.hidden_dim_color == 64 and fs.num_layers_color == 3
                             and fs.appearance_embedding_dim == 32)
        self.general_rays_per_call = 32768

    @property
    def camera_optimizer(self):
        """``self.camera_optimizer`` of the reference (``fruit_nerf.py:114-116``) as the exporters use it: whole poses
        (``apply_to_camera``).  Rays are corrected by ``cn_apply_pose_adjustment`` (``get_outputs``)."""
        from .camera_optimizer import CameraOptimizer

        return CameraOptimizer(self.params["camera_optimizer.pose_adjustment"], mode="SO3xR3")

    def state_dict(self) -> Dict[str, Tensor]:
        return dict(self.params)

    def load_state_dict(self, state: Dict[str, Tensor]) -> None:
        for k in self.params:
            self.params[k].copy_(state[k].to(self.device))

    def get_param_groups(self) -> Dict[str, List[Tensor]]:
        """``fruit_nerf.py:191-196``: groups ``proposal_networks``, ``fields``, ``camera_opt``."""
        return {
            "proposal_networks": [v for k, v in self.params.items() if k.startswith("proposal_networks.")],
            "fields": [v for k, v in self.params.items() if k.startswith("field.")],
            "camera_opt": [self.params["camera_optimizer.pose_adjustment"]],
        }

    def get_training_callbacks(self, training_callback_attributes=None) -> List["TrainingCallback"]:
        """``fruit_nerf.py:198-232``: the two callbacks a nerfstudio ``Trainer`` runs around every iteration -- the
        proposal-weight annealing (arXiv 2111.12077 eq. 18) before it and the proposal sampler's ``step_cb`` after it.
        ``FruitTrainer.train_iteration`` performs the same two updates itself; these objects are for a foreign loop."""
        callbacks: List[TrainingCallback] = []
        cfg = self.config
        if cfg.use_proposal_weight_anneal:
            N = cfg.proposal_weights_anneal_max_num_iters

            def set_anneal(step):
                from .schedules import proposal_weights_anneal

                self.step = step
                self.set_anneal(proposal_weights_anneal(step, N, cfg.proposal_weights_anneal_slope))

            def step_cb(step):  # ProposalNetworkSampler.step_cb
                self._sampler_step = step
                self._steps_since_update = getattr(self, "_steps_since_update", 0) + 1

            callbacks.append(TrainingCallback(["before_train_iteration"], 1, set_anneal))
            callbacks.append(TrainingCallback(["after_train_iteration"], 1, step_cb))
        return callbacks

    def eval(self):
        self.training = False
        return self

    def setup_inference(self, render_rgb: bool, num_inference_samples: int) -> None:
        """``fruit_nerf.py:185-189``: uniform sampler with ``num_inference_samples`` and no field contraction."""
        self.render_rgb = render_rgb
        self.num_inference_samples = int(num_inference_samples)
        self._uniform_samples = int(num_inference_samples)
        self._field_contraction = False

    def set_anneal(self, anneal: float) -> None:
        self._anneal = float(anneal)

    # ------------------------------------------------------------------------------------------ helpers
    def _scene(self, contraction: bool) -> L.Scene:
        return ops.scene_struct(self.scene_box.aabb, contraction)

    def _background(self):
        if _BACKGROUND_COLOR_OVERRIDE is not None:
            return L.BG_COLOR, _BACKGROUND_COLOR_OVERRIDE
        bg = self.config.background_color
        if isinstance(bg, str):
            if bg == "last_sample":
                return L.BG_LAST_SAMPLE, (0.0, 0.0, 0.0)
            # NerfactoModelConfig.background_color (inherited at fruit_nerf.py:60) also admits the named colours of
            # nerfstudio's RGBRenderer.  "random": combine_rgb returns the composited colour WITHOUT blending a background,
            # "as if the background color was black" -- and the reference's get_loss_dict (:601-615) does not blend the target
            # either, so for this model it IS black, in training and in eval.
            named = {"black": (0.0, 0.0, 0.0), "white": (1.0, 1.0, 1.0), "random": (0.0, 0.0, 0.0)}
            if bg not in named:
                raise ValueError(f"background_color={bg!r}: 'last_sample', 'black', 'white', 'random' or an RGB triple")
            return L.BG_COLOR, named[bg]
        return L.BG_COLOR, tuple(float(c) for c in bg)

    def _app_mode(self) -> int:
        if self.test_mode in ("inference", "export"):
            return L.APP_MEAN
        if self.training:
            return L.APP_PER_CAMERA
        return L.APP_MEAN if self.config.use_average_appearance_embedding else L.APP_ZEROS

    def _collide(self, rb: RayBundle) -> RayBundle:
        """NearFarCollider (``fruit_nerf.py:167,625-626``): only fills missing nears/fars; near plane 0 in eval."""
        if rb.nears is not None and rb.fars is not None:
            return rb
        R = rb.origins.shape[0]
        near = self.config.near_plane if self.training else 0.0
        rb.nears = torch.full((R, 1), float(near), device=self.device)
        rb.fars = torch.full((R, 1), float(self.config.far_plane), device=self.device)
        return rb

    def _prepared(self, ray_bundle: RayBundle) -> RayBundle:
        rb = ray_bundle.flatten().to(self.device)
        rb = rb._map(lambda t: t.contiguous())
        return self._collide(rb)

    def _cam_idx(self, rb: RayBundle) -> Optional[Tensor]:
        if rb.camera_indices is None:
            return None
        return rb.camera_indices.reshape(-1).to(torch.int64).contiguous()

    def _matrix_mode(self) -> str:
        mode = getattr(self.config, "matrix_precision", "split_bf16")
        mode = {"fp16": "f16", "bf16": "split_bf16"}.get(mode, mode)  # the spellings a user of 'fp32' tries first
        if mode not in ("fp32", "split_bf16", "f16"):
            raise ValueError(f"matrix_precision {mode!r}: 'fp32', 'split_bf16' or 'f16' (alias 'fp16')")
        return mode

    def _matrix_precision(self) -> int:
        """The matrix mode of the eval / export renders (a model in training mode that is rendered through these paths, e.g.
        an eval image during a run, stays exact fp32)."""
        mode = self._matrix_mode()
        if self.training or mode == "fp32":
            return L.MATRIX_FP32
        return L.MATRIX_SPLIT_BF16 if mode == "split_bf16" else L.MATRIX_F16

    def train_matrix_precision(self) -> int:
        """The matrix mode of the TRAINING iteration (``FruitTrainer.forward_backward``): ``matrix_precision="f16"`` is the
        reference's own training arithmetic -- ``mixed_precision=True`` on tiny-cuda-nn's fp16 modules
        (``fruit_nerf_config.py:35``, ``fruit_field.py:95,125-167``): the field's forward and its backward recompute with fp16
        operands, gradient products in bf16, fp32 sums, fp32 master parameters and Adam (``cn_field_backward_mp``).  The field
        shapes of the ``_big`` / ``_huge`` methods (shape-generic kernels) and the proposal networks train in fp32 whatever the
        setting."""
        mode = self._matrix_mode()
        if mode != "f16" or not self._fused_shape:
            return L.MATRIX_FP32  # "split_bf16" is an arithmetic of the eval renders: training stays exact fp32
        return L.MATRIX_F16

    def _opts(self, num_samples: int, density_only: bool = False) -> L.RenderOpts:
        bg_mode, bg = self._background()
        return ops.render_opts(num_samples, spacing=L.SPACING_UNIFORM, bg_mode=bg_mode, bg_color=bg,
                               app_mode=self._app_mode(), sh_unit_dir=self.config.sh_input == "unit",
                               eval_clamp=not self.training, density_only=density_only,
                               image_width=self._image_hint[0], pixel_start=self._image_hint[1],
                               early_stop_transmittance=0.0 if self.training else
                               getattr(self.config, "early_stop_transmittance", 0.0),
                               matrix_precision=self._matrix_precision())

    def _sample_and_render(self, rb: RayBundle, density_only: bool = False) -> Dict[str, Tensor]:
        """proposal (or uniform) sampler -> field -> renderers, all on device."""
        o, d, n, f = rb.origins, rb.directions, rb.nears, rb.fars
        cam = self._cam_idx(rb)
        if self._app_mode() == L.APP_PER_CAMERA and cam is None:
            raise AttributeError("Camera indices are not provided.")  # fruit_field.py:241-242
        if self._uniform_samples is not None:
            S = self._uniform_samples
            if not self._fused_shape:
                return self._render_general(o, d, n, f, cam, S, None, density_only)
            return ops.render_rays(self.field, self._scene(self._field_contraction), self._opts(S, density_only),
                                   o, d, n, f, camera_indices=cam)
        cfg = self.config
        S = cfg.num_nerf_samples_per_ray
        ps = ops.proposal_sample(self.proposal_networks, self._scene(self._prop_contraction), o, d, n, f,
                                 cfg.num_proposal_samples_per_ray, S, anneal=self._anneal,
                                 matrix_precision=self._matrix_precision())
        if not self._fused_shape:
            out = self._render_general(o, d, n, f, cam, S, ps["euclidean_bins"], density_only)
        else:
            out = ops.render_rays(self.field, self._scene(self._field_contraction), self._opts(S, density_only), o, d,
                                  n, f, camera_indices=cam, bins=ps["euclidean_bins"])
        for i in range(len(self.proposal_networks)):
            out[f"prop_depth_{i}"] = ps["prop_depth"][i][:, None]
        return out

    def _general_samples(self, n: Tensor, f: Tensor, S: int, bins: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
        if bins is not None:
            return bins[:, :-1].contiguous(), bins[:, 1:].contiguous()
        sm = ops.sample_spaced(n, f, S, L.SPACING_UNIFORM)
        return sm["starts"], sm["ends"]

    def _render_general(self, o: Tensor, d: Tensor, n: Tensor, f: Tensor, cam: Optional[Tensor], S: int,
                        bins: Optional[Tensor], density_only: bool) -> Dict[str, Tensor]:
        """Any field shape: cn_field_eval + cn_composite on sub-chunks (the [R,S,.] tensors exist only per sub-chunk)."""
        scene = self._scene(self._field_contraction)
        bg_mode, bg = self._background()
        outs: Dict[str, List[Tensor]] = {}
        step = self.general_rays_per_call
        for i in range(0, o.shape[0], step):
            sl = slice(i, i + step)
            starts, ends = self._general_samples(n[sl], f[sl], S, None if bins is None else bins[sl])
            fo = ops.field_eval(self.field, scene, o[sl].contiguous(), d[sl].contiguous(),
                                None if cam is None else cam[sl].contiguous(), starts, ends,
                                app_mode=self._app_mode(), sh_unit_dir=self.config.sh_input == "unit",
                                matrix_precision=self._matrix_precision())
            if density_only:
                comp = ops.composite(starts, ends, fo["density"], None, None, bg_mode, bg, eval_clamp=not self.training)
                comp = {"accumulation": comp["accumulation"]}
            else:
                comp = ops.composite(starts, ends, fo["density"], fo["rgb"], fo["semantics"], bg_mode, bg,
                                     eval_clamp=not self.training)
            for k, v in comp.items():
                outs.setdefault(k, []).append(v)
        return {k: torch.cat(v) for k, v in outs.items()}

    # ------------------------------------------------------------------------------------------ forward variants
    def forward(self, ray_bundle: RayBundle) -> Dict[str, Union[Tensor, List]]:
        """``fruit_nerf.py:617-637``."""
        rb = self._prepared(ray_bundle)
        if self.test_mode == "inference":
            return self.get_inference_outputs(rb)
        if self.test_mode == "export":
            return self.get_export_outputs(rb)
        return self.get_outputs(rb)

    __call__ = forward

    def get_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """``fruit_nerf.py:543-599`` (eval): the camera-optimizer tweak is applied also outside training."""
        if self.training:
            return self._training_outputs(ray_bundle)
        rb = ray_bundle
        if rb.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        rb = RayBundle(rb.origins.clone(), rb.directions.clone(), rb.pixel_area, rb.camera_indices, rb.nears, rb.fars)
        ops.apply_pose_adjustment(self.params["camera_optimizer.pose_adjustment"], self._cam_idx(rb), rb.origins,
                                  rb.directions)
        return self._finish(self._sample_and_render(rb))

    def _training_outputs(self, ray_bundle: RayBundle, jitter: Optional[List[Tensor]] = None) -> Dict:
        """``get_outputs`` with ``self.training`` (``fruit_nerf.py:543-599``): stratified single-jitter proposal sampling,
        per-camera appearance, no clamp, plus ``weights_list`` / ``ray_samples_list`` for ``get_loss_dict``.  Forward
        only -- gradients come from ``FruitTrainer.forward_backward``, which runs the same kernels."""
        from ..rays import RaySamples

        cfg, dev = self.config, self.device
        rb = ray_bundle
        if rb.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        R = rb.origins.shape[0]
        cam = self._cam_idx(rb)
        o, d = rb.origins.clone(), rb.directions.clone()
        ops.apply_pose_adjustment(self.params["camera_optimizer.pose_adjustment"], cam, o, d)
        n, f = rb.nears, rb.fars
        n_lvl = len(self.proposal_networks)
        if jitter is None:
            jitter = [torch.rand(R, 1, device=dev) for _ in range(n_lvl + 1)]
        jitter = [j.to(dev).contiguous() for j in jitter]
        scene = self._scene(True)
        weights_list, samples_list = [], []
        sm = ops.sample_spaced(n, f, cfg.num_proposal_samples_per_ray[0], L.SPACING_PIECEWISE, jitter[0])
        bins = torch.cat([sm["spacing_starts"], sm["spacing_ends"][:, -1:]], -1).contiguous()
        starts, ends = sm["starts"], sm["ends"]

        def samples(starts, ends, bins):
            return RaySamples(o, d, starts[..., None], ends[..., None], bins[:, :-1, None], bins[:, 1:, None],
                              rb.camera_indices)

        out: Dict = {}
        for lvl in range(n_lvl):
            den = ops.proposal_density(self.proposal_networks[lvl], scene, o, d, starts, ends)
            comp = ops.composite(starts, ends, den, want_weights=True, eval_clamp=False)
            weights_list.append(comp["weights"][..., None])
            rs_l = samples(starts, ends, bins)
            rs_l.density = den  # kept for the interlevel term of get_loss_dict
            samples_list.append(rs_l)
            out[f"prop_depth_{lvl}"] = comp["depth"]
            s_next = cfg.num_proposal_samples_per_ray[lvl + 1] if lvl + 1 < n_lvl else cfg.num_nerf_samples_per_ray
            bins, eu = ops.sample_pdf(bins, comp["weights"], n, f, s_next, anneal=self._anneal, u_rand=jitter[lvl + 1])
            starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
        if self._fused_shape:
            opts = ops.render_opts(cfg.num_nerf_samples_per_ray, app_mode=L.APP_PER_CAMERA,
                                   sh_unit_dir=cfg.sh_input == "unit", eval_clamp=False)
            fo = ops.render_samples(self.field, scene, opts, o, d, n, f, camera_indices=cam, bins=eu.contiguous())
        else:
            fo = ops.field_eval(self.field, scene, o, d, cam, starts, ends, app_mode=L.APP_PER_CAMERA,
                                sh_unit_dir=cfg.sh_input == "unit")
        bg_mode, bg = self._background()
        comp = ops.composite(starts, ends, fo["density"], fo["rgb"], fo["semantics"], bg_mode, bg, eval_clamp=False,
                             want_weights=True)
        weights_list.append(comp["weights"][..., None])
        samples_list.append(samples(starts, ends, bins))
        out.update({"rgb": comp["rgb"], "accumulation": comp["accumulation"], "depth": comp["depth"],
                    "semantics": comp["semantics"], "semantics_colormap": comp["semantics_colormap"],
                    "weights_list": weights_list, "ray_samples_list": samples_list})
        return out

    def get_loss_dict(self, outputs: Dict, batch: Dict[str, Tensor], metrics_dict=None) -> Dict[str, Tensor]:
        """``fruit_nerf.py:601-615``: values only (the optimisation path is ``FruitTrainer``, which fuses these losses
        with their backward)."""
        cfg = self.config
        image = batch["image"].to(self.device)[:, :3].to(torch.float32)
        loss = {"rgb_loss": torch.mean((image - outputs["rgb"]) ** 2)}
        x, y = outputs["semantics"], batch["fruit_mask"].to(self.device).to(torch.float32).reshape(-1, 1)
        bce = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
        loss["semantics_loss"] = cfg.semantic_loss_weight * bce.mean()
        if self.training and "weights_list" in outputs:
            final = outputs["ray_samples_list"][-1]
            fbins = torch.cat([final.spacing_starts[..., 0], final.spacing_ends[:, -1:, 0]], -1).contiguous()
            fw = outputs["weights_list"][-1][..., 0].contiguous()
            total = torch.zeros(1, device=self.device)
            for w, rs in zip(outputs["weights_list"][:-1], outputs["ray_samples_list"][:-1]):
                pb = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1).contiguous()
                # cn_interlevel_backward evaluates the term from the level's density (kept on the sample list); only its
                # loss sum is used here
                ops.interlevel_backward(fbins, fw, pb, rs.starts[..., 0].contiguous(), rs.ends[..., 0].contiguous(),
                                        rs.density, 1.0, total)
            loss["interlevel_loss"] = cfg.interlevel_loss_mult * total[0] / fw.numel()
        pose = self.params["camera_optimizer.pose_adjustment"]
        loss["camera_opt_regularizer"] = (pose[:, :3].norm(dim=-1).mean() * 1e-2 + pose[:, 3:].norm(dim=-1).mean() * 1e-3)
        return loss

    def get_inference_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """``fruit_nerf.py:497-541``: no pose tweak, mean appearance."""
        return self._finish(self._sample_and_render(ray_bundle))

    def _finish(self, out: Dict[str, Tensor]) -> Dict[str, Tensor]:
        keys = ["rgb", "accumulation", "depth"] + [k for k in out if k.startswith("prop_depth_")] + \
               ["semantics", "semantics_colormap"]
        return {k: out[k] for k in keys if k in out}

    def get_export_outputs(self, ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """``fruit_nerf.py:476-494``: per-sample rgb / position / semantic logit / density / label, no compositing."""
        if self._uniform_samples is None:
            raise RuntimeError("export mode needs setup_inference(render_rgb, num_inference_samples) first")
        rb = ray_bundle
        if not self._fused_shape:
            starts, ends = self._general_samples(rb.nears, rb.fars, self._uniform_samples, None)
            fo = ops.field_eval(self.field, self._scene(self._field_contraction), rb.origins, rb.directions,
                                self._cam_idx(rb), starts, ends, app_mode=self._app_mode(),
                                sh_unit_dir=self.config.sh_input == "unit", want_positions=True,
                                matrix_precision=self._matrix_precision())
            label = (torch.sigmoid(fo["semantics"]) - 0.9 > 0).to(torch.int64)  # fruit_nerf.py:488-492
            return {"rgb": fo["rgb"], "point_location": fo["positions"], "semantics": fo["semantics"],
                    "density": fo["density"], "semantics_colormap": label}
        out = ops.render_samples(self.field, self._scene(self._field_contraction), self._opts(self._uniform_samples),
                                 rb.origins, rb.directions, rb.nears, rb.fars, camera_indices=self._cam_idx(rb))
        return {"rgb": out["rgb"], "point_location": out["positions"], "semantics": out["semantics"],
                "density": out["density"], "semantics_colormap": out["semantics_colormap"]}

    # ------------------------------------------------------------------------------------------ chunked renders
    # The reference's chunk size (eval_num_rays_per_chunk = 32 768; the projection CLI asks for 4 096) bounds ITS memory --
    # materialised [R, S, .] tensors.  Rays are independent, so any size gives the same image here, and small chunks cost twice:
    # two launches each, and a 32 768-ray launch of the sampler fills half of the 256 CUs.  800 x 800: 13.95 ms per image in
    # chunks of 2^15 rays, 13.22 in 2^16, 12.68 in 2^18 (tools/image_probe.py) -- every render works in chunks of at least 2^18.
    EVAL_CHUNK = 1 << 18

    def _chunked(self, camera_ray_bundle: RayBundle, fn, image_width: int = 0, keys: Optional[Sequence[str]] = None,
                 min_chunk: Optional[int] = None) -> Dict[str, Tensor]:
        """``image_width`` > 0: the bundle is a whole row-major image, so every chunk is a pixel run -- passed to the
        renderer as a scheduling hint (cn_render_opts.image_width / pixel_start).  ``keys``: keep only these outputs."""
        chunk = max(int(self.config.eval_num_rays_per_chunk), int(self.EVAL_CHUNK if min_chunk is None else min_chunk))
        flat = camera_ray_bundle.flatten()
        n = len(flat)
        if n > chunk:  # equal chunks instead of full ones and a short tail (640 000 rays: 3 x 213 376, not 2 x 262 144 + 115 712)
            pieces = -(-n // chunk)
            chunk = min(chunk, -(-(-(-n // pieces)) // 64) * 64)
        lists: Dict[str, List[Tensor]] = {}
        for i in range(0, n, chunk):
            self._image_hint = (image_width, i)
            try:
                out = fn(flat if n <= chunk else flat.get_row_major_sliced_ray_bundle(i, i + chunk))
            finally:
                self._image_hint = (0, 0)
            for k, v in out.items():
                if isinstance(v, Tensor) and (keys is None or k in keys):
                    lists.setdefault(k, []).append(v)
        return {k: (v[0] if len(v) == 1 else torch.cat(v)) for k, v in lists.items()}

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Dict[str, Tensor]:
        """``fruit_nerf.py:377-404``: [H,W,C] outputs of a full-image bundle."""
        image_height, image_width = camera_ray_bundle.origins.shape[:2]
        out = self._chunked(camera_ray_bundle, self.forward, image_width=image_width)
        return {k: v.view(image_height, image_width, -1) for k, v in out.items()}

    JAGGED_CHUNK = EVAL_CHUNK  # (a jagged bundle has no image to stripe; same reasoning)

    @torch.no_grad()
    def get_outputs_for_camera_jagged_ray_bundle(self, camera_ray_bundle: RayBundle,
                                                 keys: Optional[Sequence[str]] = None) -> Dict[str, Tensor]:
        """``fruit_nerf.py:346-374``: same for an arbitrary list of rays ([N,C] outputs).  ``keys`` (extension): only these
        outputs are collected."""
        return self._chunked(camera_ray_bundle, self.forward, keys=keys, min_chunk=self.JAGGED_CHUNK)

    @torch.no_grad()
    def get_density_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle) -> Tensor:
        """``fruit_nerf.py:320-344``: sum_s w per ray (no pose tweak, no collider); density-only kernel variant."""
        out = self._chunked(camera_ray_bundle,
                            lambda rb: self._sample_and_render(rb.flatten().to(self.device)._map(lambda t: t.contiguous()),
                                                               density_only=True), min_chunk=self.JAGGED_CHUNK)
        return out["accumulation"][:, 0]

    # ------------------------------------------------------------------------------------------ projection
    @torch.no_grad()
    def get_outputs_for_projections(self, train_dataset, camera_optimizer=None,
                                    pcd_path: str = "/opt/data/artifacts/pear/pcd/all_super_cluster_info_nsub_2.npy",
                                    output_root: str = "/opt/data/artifacts/pear/projection",
                                    pcd_data=None, save: bool = True, batched: bool = True, return_run: bool = False):
        """``fruit_nerf.py:254-318``: per (super-cluster, camera, sub-cluster AABB) an unoccluded semantic render and an
        occlusion-masked one, written as PNGs (``save``), else returned as {(i_sc, cam_idx, i): (wo_occ, visible)} float
        images.  ``batched`` (default): the jobs run many at a time through ``projection.project_all`` -- rays only for the
        pixels a box can cover, one host synchronisation per batch, PNG files written by worker threads behind the GPU; the
        per-job loop of the reference (``batched=False`` -> ``project_cluster``) computes the same values.  ``return_run``:
        return the compact ``ProjectionRun`` instead (what the in-process merger reads; nothing is expanded to frames)."""
        cameras: Cameras = train_dataset.cameras
        segmentation_files = train_dataset.metadata["semantics"].filenames
        if pcd_data is None:
            pcd_data = np.load(pcd_path, allow_pickle=True)
        from ..distributed import world as _world

        rank, world_size = _world()  # (camera x sub-cluster) jobs are independent: dealt round-robin over the ranks,
        if batched:                  # no communication (every rank writes its own PNGs / returns its own results)
            from .projection import project_all

            run = project_all(self, cameras.to(self.device), pcd_data, output_root=output_root if save else None,
                              segmentation_files=segmentation_files, want_float=not save and not return_run,
                              keep=return_run or not save, rank=rank, world_size=world_size)
            if return_run:
                return run
            return None if save else run.float_results()
        results = {}
        job = -1
        for i_sc in range(len(pcd_data)):
            cluster_aabb = np.asarray(pcd_data[i_sc]["aabb"])
            save_dir = os.path.join(output_root, f"super_cluster_{i_sc}")
            if save:
                os.makedirs(save_dir, exist_ok=True)
            for cam_idx, cam in enumerate(cameras):
                cam_dir = os.path.join(save_dir, f"cam_{cam_idx}")
                if save:
                    os.makedirs(cam_dir, exist_ok=True)
                for i in range(cluster_aabb.shape[0]):
                    job += 1
                    if job % world_size != rank:
                        continue
                    aabb = SceneBox(torch.tensor(cluster_aabb[i], dtype=torch.float32))
                    wo_occ, visible = self.project_cluster(cam, aabb, cam_idx)
                    if save:
                        save_image(visible, os.path.join(cam_dir, f"visible_cluster_{i}.png"))
                        save_image(wo_occ, os.path.join(cam_dir, f"wo_occ_cluster_{i}.png"))
                    else:
                        results[(i_sc, cam_idx, i)] = (wo_occ, visible)
                if save and cam_idx < len(segmentation_files) and os.path.exists(segmentation_files[cam_idx]):
                    shutil.copy(segmentation_files[cam_idx], cam_dir)
        return results if not save else None

    def project_cluster(self, cam: Cameras, aabb: SceneBox, cam_idx: int = 0) -> Tuple[Tensor, Tensor]:
        """One job of the loop above: returns float images [H,W,3] (wo_occ, visible) on the device."""
        cam = cam.to(self.device)
        rays = cam.generate_rays(camera_indices=0, keep_shape=True, aabb_box=aabb)  # fruit_nerf.py:283
        if not self.compat_projection_cam0:
            rays.camera_indices = torch.full_like(rays.camera_indices, cam_idx)
        H, W = rays.origins.shape[:2]
        # The reference indexes with the boolean mask six times (:289-315); every such indexing is a nonzero() with a host
        # round trip.  Here the pixel indices of the rays that hit the box are formed ONCE and reused -- same values.
        idx = (rays.nears.reshape(-1) < 1e10).nonzero(as_tuple=False).squeeze(1)  # the job's one synchronisation
        img = torch.zeros(H * W, 3, device=self.device)
        if idx.numel() < 10:  # fruit_nerf.py:293
            z = img.reshape(H, W, 3)
            return z, z.clone()
        sub = rays.flatten()[idx]
        out = self.get_outputs_for_camera_jagged_ray_bundle(sub)
        img[idx] = out["semantics"]  # logit sum, un-sigmoided (reference quirk, :302); [N,1] broadcast over the channels
        wo_occ = img.reshape(H, W, 3).clone()
        occ = sub.clone()  # the occlusion pass: from the camera to the box's near side (:305-308)
        occ.fars = sub.nears.clone()
        occ.nears = torch.zeros_like(sub.nears)
        hidden = self.get_density_for_camera_ray_bundle(occ) >= 0.5  # [N]
        img[idx] = torch.where(hidden[:, None], torch.zeros((), device=self.device), img[idx])
        return wo_occ, img.reshape(H, W, 3)

    # ------------------------------------------------------------------------------------------ metrics
    def get_metrics_dict(self, outputs, batch) -> Dict[str, Tensor]:
        """``fruit_nerf.py:639-645`` (PSNR only; the distortion metric needs the training lists)."""
        image = batch["image"].to(self.device)
        mse = torch.mean((outputs["rgb"] - image[:, :3]) ** 2)
        return {"psnr": -10.0 * torch.log10(mse)}

    def get_image_metrics_and_images(self, outputs: Dict[str, Tensor], batch: Dict[str, Tensor], lpips_fn=None):
        """``fruit_nerf.py:647-700``: (psnr, ssim, lpips, iou) and the logged images for one rendered view."""
        from .image_metrics import get_image_metrics_and_images

        return get_image_metrics_and_images(self, outputs, batch, lpips_fn)


def save_image(img_hw3: Tensor, path: str) -> None:
    """``torchvision.utils.save_image`` semantics for one [H,W,3] image: clamp to [0,1], x255 + 0.5, uint8 PNG."""
    from PIL import Image

    arr = img_hw3.detach().clamp(0, 1).mul(255).add_(0.5).clamp_(0, 255).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(path)
