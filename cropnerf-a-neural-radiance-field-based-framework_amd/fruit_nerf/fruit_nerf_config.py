"""Plugin surface -- mirror of ``crop_nerf/fruit_nerf/fruit_nerf_config.py:29-172``: the three method specs
``fruit_nerf_method``, ``fruit_nerf_method_big``, ``fruit_nerf_method_huge`` under the same attribute names, so
``NERFSTUDIO_METHOD_CONFIGS=fruit_nerf=fruit_nerf.fruit_nerf_config:fruit_nerf_method`` (``README.md:79``) resolves
here when this directory is on PYTHONPATH.

When ``import nerfstudio`` succeeds the three attributes ARE ``nerfstudio.plugins.types.MethodSpecification`` objects
built from nerfstudio's own config types, with ``Model`` / ``VanillaPipeline`` / ``VanillaDataManager`` subclasses around
the HIP model as their targets (``nerfstudio_adapter.py``; wiring tested against ``tests/fakes/nerfstudio``).  Without it
(this image has no nerfstudio) they are the same-shaped dataclasses below, which this repo's own CLIs consume in either
case (``native_method``)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional

from ..config import FruitNerfModelConfig
from .data.cotton_nerf_dataparser import CottonNerfDataParserConfig
from .data.fruit_datamanager import FruitDataManagerConfig
from .data.fruitnerf_dataparser import FruitNerfDataParserConfig
from .fruit_pipeline import FruitPipelineConfig


@dataclass
class OptimizerSpec:
    optimizer: str  # "adam" | "radam"
    lr: float
    eps: float
    lr_final: Optional[float] = None
    max_steps: Optional[int] = None
    weight_decay: float = 0.0


@dataclass
class TrainerConfig:
    method_name: str
    steps_per_eval_batch: int = 500
    steps_per_save: int = 2000
    max_num_iterations: int = 40000
    mixed_precision: bool = True
    pipeline: FruitPipelineConfig = field(default_factory=FruitPipelineConfig)
    optimizers: Dict[str, OptimizerSpec] = field(default_factory=dict)
    viewer_num_rays_per_chunk: int = 1 << 15
    vis: str = "viewer"


@dataclass
class MethodSpecification:
    config: TrainerConfig
    description: str


def _optim(field_opt: str, lr_final=1e-4, max_steps=200000, prop_sched=True):
    return {
        "proposal_networks": OptimizerSpec(field_opt, 1e-2, 1e-15, lr_final if prop_sched else None,
                                           max_steps if prop_sched else None),
        "fields": OptimizerSpec(field_opt, 1e-2, 1e-15, lr_final, max_steps),
        "camera_opt": OptimizerSpec("adam", 1e-3, 1e-15, 1e-4, 5000),
    }


_native_fruit_nerf_method = MethodSpecification(
    config=TrainerConfig(
        method_name="fruit_nerf", steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=40000,
        mixed_precision=True,
        pipeline=FruitPipelineConfig(
            datamanager=FruitDataManagerConfig(train_num_rays_per_batch=4096, eval_num_rays_per_batch=4096,
                                               dataparser=CottonNerfDataParserConfig()),
            model=FruitNerfModelConfig(eval_num_rays_per_chunk=1 << 15),
        ),
        optimizers=_optim("adam"),
    ),
    description="Base config for LERF",  # sic (fruit_nerf_config.py:64)
)

_native_fruit_nerf_method_big = MethodSpecification(
    config=TrainerConfig(
        method_name="fruit_nerf_big", max_num_iterations=100000,
        pipeline=FruitPipelineConfig(
            datamanager=FruitDataManagerConfig(train_num_rays_per_batch=4096 * 2, eval_num_rays_per_batch=4096,
                                               dataparser=CottonNerfDataParserConfig(train_split_fraction=0.99)),
            model=FruitNerfModelConfig(
                eval_num_rays_per_chunk=1 << 15, num_nerf_samples_per_ray=128,
                num_proposal_samples_per_ray=(512, 256), geo_feat_dim=30, hidden_dim_semantics=128,
                num_layers_semantic=3, max_res=4096, proposal_weights_anneal_max_num_iters=5000,
                log2_hashmap_size=21),
        ),
        optimizers=_optim("radam", 1e-4, 50000, prop_sched=False),
    ),
    description="Base config for FruitNeRF-Big",
)

_native_fruit_nerf_method_huge = MethodSpecification(
    config=TrainerConfig(
        method_name="fruit_nerf_huge", max_num_iterations=100000,
        pipeline=FruitPipelineConfig(
            datamanager=FruitDataManagerConfig(train_num_rays_per_batch=4096 * 4, eval_num_rays_per_batch=4096,
                                               dataparser=FruitNerfDataParserConfig()),
            model=FruitNerfModelConfig(
                eval_num_rays_per_chunk=1 << 15, num_nerf_samples_per_ray=64,
                num_proposal_samples_per_ray=(512, 512),
                proposal_net_args_list=[
                    {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 512, "use_linear": False},
                    {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 7, "max_res": 2048, "use_linear": False},
                ],
                geo_feat_dim=30, hidden_dim_semantics=128, num_layers_semantic=3, max_res=8192,
                proposal_weights_anneal_max_num_iters=5000, log2_hashmap_size=21),
        ),
        optimizers={k: v for k, v in _optim("radam", 1e-4, 50000, prop_sched=False).items() if k != "camera_opt"},
    ),
    description="Base config for FruitNeRF-Huge",
)


NATIVE_METHODS = {"fruit_nerf": _native_fruit_nerf_method, "fruit_nerf_big": _native_fruit_nerf_method_big,
                  "fruit_nerf_huge": _native_fruit_nerf_method_huge}


def native_method(name: str) -> MethodSpecification:
    """The method specification in this package's own dataclasses (what ``scripts/train.py`` runs), whatever the public
    attributes below are."""
    return NATIVE_METHODS[name]


def _nerfstudio_available() -> bool:
    try:
        import nerfstudio.plugins.types  # noqa: F401
    except ImportError:
        return False
    return True


HAVE_NERFSTUDIO = _nerfstudio_available()
if HAVE_NERFSTUDIO:
    from . import nerfstudio_adapter as _adapter

    fruit_nerf_method = _adapter.method_specification(_native_fruit_nerf_method)
    fruit_nerf_method_big = _adapter.method_specification(_native_fruit_nerf_method_big)
    fruit_nerf_method_huge = _adapter.method_specification(_native_fruit_nerf_method_huge)
else:
    fruit_nerf_method = _native_fruit_nerf_method
    fruit_nerf_method_big = _native_fruit_nerf_method_big
    fruit_nerf_method_huge = _native_fruit_nerf_method_huge
