"""The fruit_nerf plugin as REAL nerfstudio objects (imported only when ``import nerfstudio`` succeeds).

``fruit_nerf_config.py`` then exports ``fruit_nerf_method`` / ``_big`` / ``_huge`` as
``nerfstudio.plugins.types.MethodSpecification`` instances whose ``TrainerConfig`` has the reference's shape
(``crop_nerf/fruit_nerf/fruit_nerf_config.py:29-172``): ``pipeline=FruitPipelineConfig(datamanager=
FruitDataManagerConfig(dataparser=CottonNerfDataParserConfig(), ...), model=FruitNerfModelConfig(...))``,
``optimizers={"proposal_networks", "fields", "camera_opt"}`` of ``AdamOptimizerConfig`` / ``RAdamOptimizerConfig`` +
``ExponentialDecaySchedulerConfig``, ``viewer=ViewerConfig(num_rays_per_chunk=1 << 15)``, ``vis="viewer"`` -- with
``_target`` classes that subclass nerfstudio's ``Model`` / ``VanillaPipeline`` / ``VanillaDataManager`` and run on
libcropnerf_hip:

* ``FruitModel(Model)`` (reference ``fruit_nerf.py:73-700``) owns the HIP model (``fruit_nerf.FruitModel`` of this package)
  and a ``FruitTrainer`` for its flat parameter / gradient buffers.  Every parameter tensor is exposed as an
  ``nn.Parameter`` VIEW of the flat buffer (same storage: ``torch.optim`` steps taken by nerfstudio's ``Optimizers``
  update what the kernels read), grouped by ``get_param_groups`` exactly as the reference groups them
  (``fruit_nerf.py:191-196``).
* training: the loss and all gradients come out of the fused HIP forward + backward.  ``get_loss_dict`` runs it and
  returns the loss terms through ``_GradientBridge`` -- an ``autograd.Function`` whose backward hands the kernels'
  gradients (times the incoming scale: nerfstudio's ``GradScaler``) to the parameters' ``.grad``; on the iterations where
  the proposal networks run without gradient (``fruit_nerf.py:144-149``) their parameters get no gradient at all, so
  ``torch.optim.Adam`` skips them, as in the reference.
* ``FruitPipeline(VanillaPipeline)`` (reference ``fruit_pipeline.py:88-121``) forwards ``test_mode`` into the model;
  ``FruitDataManager(VanillaDataManager)`` (``data/fruit_datamanager.py:124-215``) adds ``setup_inference`` /
  ``next_sample_volume`` on the orthographic ray kernels.

None of nerfstudio ships in this image: the wiring is tested against ``tests/fakes/nerfstudio`` (types and field names
only, ``tests/test_plugin.py``); INTEGRATION.md describes what a maintainer with a real installation checks first.
"""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Type

import numpy as np
import torch
from torch import nn

from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
from nerfstudio.configs.base_config import ViewerConfig
from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager, VanillaDataManagerConfig
from nerfstudio.data.dataparsers.base_dataparser import DataParser, DataParserConfig
from nerfstudio.engine.optimizers import AdamOptimizerConfig, RAdamOptimizerConfig
from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
from nerfstudio.engine.trainer import TrainerConfig
from nerfstudio.models.base_model import Model, ModelConfig
from nerfstudio.pipelines.base_pipeline import VanillaPipeline, VanillaPipelineConfig
from nerfstudio.plugins.types import MethodSpecification

from .. import config as native_config
from ..rays import RayBundle as NativeRayBundle
from ..rays import SceneBox as NativeSceneBox

_NATIVE_MODEL_FIELDS = [f for f in dataclasses.fields(native_config.FruitNerfModelConfig)]


def _native_model_config(cfg) -> native_config.FruitNerfModelConfig:
    return native_config.FruitNerfModelConfig(**{f.name: getattr(cfg, f.name) for f in _NATIVE_MODEL_FIELDS})


def _to_native_rays(rb) -> NativeRayBundle:
    """nerfstudio ``RayBundle`` -> this package's (same field names; plain tensors)."""
    if isinstance(rb, NativeRayBundle):
        return rb
    return NativeRayBundle(origins=rb.origins, directions=rb.directions, pixel_area=rb.pixel_area,
                           camera_indices=rb.camera_indices, nears=rb.nears, fars=rb.fars)


# ------------------------------------------------------------------------------------------------------------- model


class _GradientBridge(torch.autograd.Function):
    """loss value (computed by the kernels) -> a tensor whose backward delivers the kernels' parameter gradients."""

    @staticmethod
    def forward(ctx, value: torch.Tensor, owner, names, *params):
        ctx.owner, ctx.names = owner, names
        return value.detach().clone()

    @staticmethod
    def backward(ctx, grad_output):
        grads = tuple(ctx.owner.trainer.grads[k] * grad_output for k in ctx.names)
        return (None, None, None) + grads


class FruitModel(Model):
    """nerfstudio ``Model`` over the HIP model.  Constructor keywords as the reference's (``fruit_nerf.py:80-85``,
    ``fruit_pipeline.py:107-115``): ``scene_box, num_train_data, metadata{"semantics"}, device, grad_scaler, test_mode,
    render_rgb_inference``."""

    def __init__(self, config, metadata: Dict, test_mode: str = "val", render_rgb_inference: bool = True, **kwargs) -> None:
        assert "semantics" in metadata.keys()  # fruit_nerf.py:81
        self.semantics = metadata["semantics"]
        self.test_mode = test_mode
        self.render_rgb_inference = render_rgb_inference
        self._hip_device = kwargs.get("device", "cuda")
        super().__init__(config=config, **kwargs)
        self.colormap = self.hip.colormap

    def populate_modules(self):
        from .fruit_nerf import FruitModel as HipFruitModel
        from .fruit_nerf import Semantics
        from .trainer import FruitTrainer

        sem = self.semantics
        native_sem = sem if isinstance(sem, Semantics) else Semantics(
            getattr(sem, "filenames", ()), getattr(sem, "classes", ("apple", "stuff")), getattr(sem, "colors", None),
            getattr(sem, "mask_classes", ()))
        aabb = self.scene_box.aabb if hasattr(self.scene_box, "aabb") else self.scene_box
        self.hip = HipFruitModel(_native_model_config(self.config), scene_box=NativeSceneBox(torch.as_tensor(aabb).float()),
                                 num_train_data=self.num_train_data, metadata={"semantics": native_sem},
                                 device=self._hip_device, test_mode=self.test_mode,
                                 render_rgb_inference=self.render_rgb_inference)
        # flat parameter / gradient buffers; optimiser steps are nerfstudio's (torch.optim on the views below)
        self.trainer = FruitTrainer(self.hip)
        self._names: Dict[str, str] = {}
        for k, v in self.hip.params.items():
            attr = "p__" + k.replace(".", "__")
            self.register_parameter(attr, nn.Parameter(v, requires_grad=True))  # a view: same storage as the flat buffer
            self._names[k] = attr
        self._train_rays = None
        self.step = 0

    # -- the reference's surface ----------------------------------------------------------------------------------
    def param(self, name: str) -> nn.Parameter:
        return getattr(self, self._names[name])

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        """``fruit_nerf.py:191-196``."""
        return {
            "proposal_networks": [self.param(k) for k in self._names if k.startswith("proposal_networks.")],
            "fields": [self.param(k) for k in self._names if k.startswith("field.")],
            "camera_opt": [self.param("camera_optimizer.pose_adjustment")],
        }

    def get_training_callbacks(self, training_callback_attributes) -> List:
        """``fruit_nerf.py:198-232``: annealing before, the proposal sampler's step callback after every iteration."""
        from nerfstudio.engine.callbacks import TrainingCallback, TrainingCallbackLocation

        tr = self.trainer

        def set_anneal(step):
            self.step = step
            tr.set_anneal(step)

        def step_cb(step):
            # AFTER_TRAIN_ITERATION runs behind nerfstudio's optimizer_scaler_step_all: the optimiser has moved the entries
            # that own a tcnn parameter -- copy them to the alias entries before the next forward reads the table
            self._tie_tcnn_parameters()
            tr._sampler_step = step
            tr._steps_since_update += 1

        return [TrainingCallback([TrainingCallbackLocation.BEFORE_TRAIN_ITERATION], set_anneal, update_every_num_iters=1),
                TrainingCallback([TrainingCallbackLocation.AFTER_TRAIN_ITERATION], step_cb, update_every_num_iters=1)]

    def _tie_tcnn_parameters(self) -> None:
        """tcnn layout: several table entries of a dense level may stand for ONE tcnn parameter (``tcnn_grid.hip``).  Their
        gradients are folded into the owning entry before the optimiser step (``_run_training_step``); after any write to
        the tables -- an optimiser step of nerfstudio's own ``torch.optim``, a loaded state dict -- the owners' values are
        copied back to the aliases, as ``FruitTrainer.optimizer_step`` does on this package's own training path."""
        tr = self.trainer
        if not tr.tcnn:
            return
        from .. import ops

        for spec, key in tr._tcnn_tables:
            ops.tcnn_grid_tie_parameters(spec, self.hip.params[key])

    def setup_inference(self, render_rgb, num_inference_samples):
        self.hip.setup_inference(render_rgb, num_inference_samples)

    def train(self, mode: bool = True):
        super().train(mode)
        self.hip.training = bool(mode)
        return self

    def forward(self, ray_bundle):
        """``fruit_nerf.py:617-637``.  In training only the rays are recorded here: the fused forward + backward needs the
        batch and runs in ``get_loss_dict``; its rendered outputs are what ``get_metrics_dict`` reads."""
        rb = _to_native_rays(ray_bundle)
        if self.training:
            self._train_rays = rb
            return {"_pending": True}
        return self.hip.forward(rb)

    def get_outputs(self, ray_bundle):
        return self.forward(ray_bundle)

    def _run_training_step(self, batch) -> Dict[str, Any]:
        tr = self.trainer
        tr.flat_grads.zero_()
        updated = tr.proposal_update_due(tr._sampler_step)
        out = tr.forward_backward(self._train_rays, batch, update_proposals=updated)
        if updated:
            tr._steps_since_update = 0
        if tr.tcnn:
            from .. import ops

            for spec, key in tr._tcnn_tables:
                ops.tcnn_grid_tie_gradients(spec, tr.grads[key])
            for k in tr._frozen:
                tr.grads[k].zero_()
        out["_proposals_updated"] = updated
        return out

    def get_metrics_dict(self, outputs, batch):
        if outputs.get("_pending"):
            outputs.update(self._run_training_step(batch))
            outputs["_pending"] = False
        if "loss_dict" in outputs:
            return self.trainer.get_metrics_dict(outputs)
        return self.hip.get_metrics_dict(outputs, batch)

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        """``fruit_nerf.py:601-615``.  Training: the kernels' loss terms, wired to the parameters through the bridge."""
        if not self.training:
            return self.hip.get_loss_dict(outputs, batch, metrics_dict)
        if outputs.get("_pending"):
            outputs.update(self._run_training_step(batch))
            outputs["_pending"] = False
        names = [k for k in self._names
                 if outputs["_proposals_updated"] or not k.startswith("proposal_networks.")]
        if not self.trainer.train_pose:
            names = [k for k in names if not k.startswith("camera_optimizer.")]
        ld = dict(outputs["loss_dict"])
        total = sum(ld.values())
        bridged = _GradientBridge.apply(total, self, tuple(names), *[self.param(k) for k in names])
        # the sum of the returned entries carries the gradient of the total loss exactly once
        first = next(iter(ld))
        ld[first] = ld[first].detach() + (bridged - bridged.detach())
        return ld

    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle):
        return self.hip.get_outputs_for_camera_ray_bundle(_to_native_rays(camera_ray_bundle))

    def get_outputs_for_projections(self, *args, **kwargs):
        return self.hip.get_outputs_for_projections(*args, **kwargs)

    def get_image_metrics_and_images(self, outputs, batch):
        return self.hip.get_image_metrics_and_images(outputs, batch)

    # -- nerfstudio checkpoints carry this module's state dict under "_model." --------------------------------------------
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        from . import nerfstudio_io as NIO
        from .tcnn_params import to_tcnn_state_dict

        if self.hip.config.implementation == "tcnn":
            state = to_tcnn_state_dict(self.hip.params, self.hip.field_spec, self.hip.proposal_specs)
        else:
            state = NIO.nerfstudio_names({k: v.detach() for k, v in self.hip.params.items()})
        state.update(NIO.field_buffers(self.hip.config, self.hip.scene_box.aabb))
        out = destination if destination is not None else {}
        for k, v in state.items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        from . import nerfstudio_io as NIO
        from .tcnn_params import from_tcnn_state_dict, is_tcnn_state_dict

        state = NIO.model_state_from_pipeline({"_model." + k: v for k, v in state_dict.items()})
        if is_tcnn_state_dict(state):
            state = from_tcnn_state_dict(state, self.hip.field_spec, self.hip.proposal_specs, self.hip.device, torch.float32)
        for k in self.hip.params:
            if k in state:
                self.hip.params[k].copy_(state[k].to(self.hip.device))
            elif strict:
                raise KeyError(k)
        self._tie_tcnn_parameters()  # a torch-named state dict written into a tcnn-layout model: aliases follow their owners


def _make_model_config() -> Type:
    """``FruitNerfModelConfig(NerfactoModelConfig)`` (``fruit_nerf.py:59-68``) as a nerfstudio ``ModelConfig`` with this
    package's fields (same names and defaults as the reference's config tree) at the top level."""
    base_names = {f.name for f in dataclasses.fields(ModelConfig)}
    extra = []
    for f in _NATIVE_MODEL_FIELDS:
        if f.name in base_names:
            continue
        d = (field(default_factory=f.default_factory) if f.default_factory is not dataclasses.MISSING
             else field(default=f.default))
        extra.append((f.name, f.type, d))
    extra.append(("_target", Type, field(default_factory=lambda: FruitModel)))
    extra.append(("eval_num_rays_per_chunk", int, field(default=1 << 15)))
    extra.append(("camera_optimizer", Any, field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3"))))
    return dataclasses.make_dataclass("FruitNerfModelConfig", extra, bases=(ModelConfig,))


FruitNerfModelConfig = _make_model_config()


# ---------------------------------------------------------------------------------------------------------- dataparser


class CottonNerf(DataParser):
    """nerfstudio ``DataParser`` over this package's parser (``data/cotton_nerf_dataparser.py``)."""

    native_config_cls = None

    def __init__(self, config):
        super().__init__(config)
        kw = {f.name: getattr(config, f.name) for f in dataclasses.fields(self.native_config_cls) if hasattr(config, f.name)}
        self.native = self.native_config_cls(**kw).setup()

    def _generate_dataparser_outputs(self, split: str = "train", **kwargs):
        return self.native.get_dataparser_outputs(split)


def _make_dataparser_config(name: str, native_cls, target_name: str) -> Type:
    target = type(target_name, (CottonNerf,), {"native_config_cls": native_cls})
    base_names = {f.name for f in dataclasses.fields(DataParserConfig)}
    extra = []
    for f in dataclasses.fields(native_cls):
        if f.name in base_names:
            continue
        d = (field(default_factory=f.default_factory) if f.default_factory is not dataclasses.MISSING
             else field(default=f.default))
        extra.append((f.name, f.type, d))
    extra.append(("_target", Type, field(default_factory=lambda: target)))
    return dataclasses.make_dataclass(name, extra, bases=(DataParserConfig,))


from .data.cotton_nerf_dataparser import CottonNerfDataParserConfig as _NativeCotton  # noqa: E402
from .data.fruitnerf_dataparser import FruitNerfDataParserConfig as _NativeFruit  # noqa: E402

CottonNerfDataParserConfig = _make_dataparser_config("CottonNerfDataParserConfig", _NativeCotton, "CottonNerf")
FruitNerfDataParserConfig = _make_dataparser_config("FruitNerfDataParserConfig", _NativeFruit, "FruitNerf")


# --------------------------------------------------------------------------------------------------- datamanager / pipeline


class FruitDataManager(VanillaDataManager):
    """``data/fruit_datamanager.py:124-215``: nerfstudio's stored-data manager plus the orthographic ray source of the
    dense export (``setup_inference`` ``:157-172``, ``next_sample_volume`` ``:199-204``) on ``cn_surface_grid`` /
    ``cn_raygen_ortho``."""

    def setup_inference(self, aabb, num_points) -> int:
        from .data.fruit_datamanager import get_corners_of_aabb, sample_surface_points

        corners = get_corners_of_aabb(aabb)
        self._surface_points, self._plane_vector = sample_surface_points(corners, num_points, self.device)
        return int(self._surface_points.shape[0])

    def next_sample_volume(self, step: int):
        from .. import ops

        self.eval_count += 1
        bs = self.config.eval_num_rays_per_batch
        start = bs * (self.eval_count - 1)
        n = max(0, min(bs, int(self._surface_points.shape[0]) - start))
        r = ops.raygen_ortho(self._surface_points, self._plane_vector.reshape(-1).tolist(), start, n)
        return NativeRayBundle(origins=r["origins"], directions=r["directions"], pixel_area=r["pixel_area"],
                               camera_indices=None, nears=r["nears"], fars=r["fars"]), None


@dataclass
class FruitDataManagerConfig(VanillaDataManagerConfig):
    _target: Type = field(default_factory=lambda: FruitDataManager)


class FruitPipeline(VanillaPipeline):
    """``fruit_pipeline.py:88-121``: as ``VanillaPipeline.__init__`` but ``test_mode`` (and ``render_rgb_inference``) reach
    the model.  Multi-GPU: the reference wraps the model in DDP (``:119-121``); here every rank keeps its own ray batches
    and the flat gradient buffer is averaged with one all-reduce before the optimiser step
    (``FruitTrainer.all_reduce_gradients``), which ``get_train_loss_dict`` does."""

    def __init__(self, config, device: str, test_mode: str = "val", world_size: int = 1, local_rank: int = 0,
                 grad_scaler=None, render_rgb_inference: bool = True):
        nn.Module.__init__(self)
        self.config, self.test_mode = config, test_mode
        self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size,
                                                    local_rank=local_rank)
        assert self.datamanager.train_dataset is not None, "Missing input dataset"
        self._model = config.model.setup(
            scene_box=self.datamanager.train_dataset.scene_box, num_train_data=len(self.datamanager.train_dataset),
            metadata=self.datamanager.train_dataset.metadata, device=device, grad_scaler=grad_scaler,
            test_mode=test_mode, render_rgb_inference=render_rgb_inference)
        self.world_size = world_size

    def get_train_loss_dict(self, step: int):
        ray_bundle, batch = self.datamanager.next_train(step)
        model_outputs = self._model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)  # runs the fused forward + backward
        if self.world_size > 1:
            self.model.trainer.all_reduce_gradients()
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict


@dataclass
class FruitPipelineConfig(VanillaPipelineConfig):
    _target: Type = field(default_factory=lambda: FruitPipeline)
    datamanager: Any = field(default_factory=lambda: FruitDataManagerConfig())
    model: Any = field(default_factory=lambda: FruitNerfModelConfig())


# ------------------------------------------------------------------------------------------------- method specifications


def _optimizers(native: Dict[str, Any]) -> Dict[str, Dict[str, Any]]:
    out = {}
    for name, o in native.items():
        opt = (RAdamOptimizerConfig if o.optimizer == "radam" else AdamOptimizerConfig)(lr=o.lr, eps=o.eps)
        sched = (ExponentialDecaySchedulerConfig(lr_final=o.lr_final, max_steps=o.max_steps)
                 if o.lr_final is not None and o.max_steps is not None else None)
        out[name] = {"optimizer": opt, "scheduler": sched}
    return out


def method_specification(native_spec) -> MethodSpecification:
    """A native method specification (``fruit_nerf_config.py`` of this package) as nerfstudio's own types."""
    tc = native_spec.config
    ndp = tc.pipeline.datamanager.dataparser
    dp_cls = FruitNerfDataParserConfig if type(ndp).__name__.startswith("FruitNerf") else CottonNerfDataParserConfig
    dp = dp_cls(**{f.name: getattr(ndp, f.name) for f in dataclasses.fields(ndp) if f.name != "data"})
    dm = FruitDataManagerConfig(dataparser=dp, train_num_rays_per_batch=tc.pipeline.datamanager.train_num_rays_per_batch,
                                eval_num_rays_per_batch=tc.pipeline.datamanager.eval_num_rays_per_batch)
    model = FruitNerfModelConfig(**{f.name: getattr(tc.pipeline.model, f.name) for f in _NATIVE_MODEL_FIELDS})
    return MethodSpecification(
        config=TrainerConfig(
            method_name=tc.method_name, steps_per_eval_batch=tc.steps_per_eval_batch, steps_per_save=tc.steps_per_save,
            max_num_iterations=tc.max_num_iterations, mixed_precision=tc.mixed_precision,
            pipeline=FruitPipelineConfig(datamanager=dm, model=model), optimizers=_optimizers(tc.optimizers),
            viewer=ViewerConfig(num_rays_per_chunk=tc.viewer_num_rays_per_chunk), vis=tc.vis),
        description=native_spec.description)
