"""``FruitPipeline`` -- mirror of ``crop_nerf/fruit_nerf/fruit_pipeline.py:57-121``: builds the datamanager and the
model (forwarding ``test_mode``, ``:107-115``); with ``world_size > 1`` the rays are sharded by batch across ranks
(``cropnerf_amd.distributed``) where the reference wraps the model in DDP (``:119-121``)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from ..config import FruitNerfModelConfig
from ..rays import Cameras, SceneBox
from .data.fruit_datamanager import FruitDataManager, FruitDataManagerConfig
from .fruit_nerf import FruitModel, Semantics


@dataclass
class FruitPipelineConfig:
    datamanager: FruitDataManagerConfig = field(default_factory=FruitDataManagerConfig)
    model: FruitNerfModelConfig = field(default_factory=FruitNerfModelConfig)


class FruitPipeline:
    def __init__(self, config: FruitPipelineConfig, device: str, cameras: Cameras, scene_box: SceneBox,
                 test_mode: str = "val", world_size: int = 1, local_rank: int = 0, grad_scaler=None,
                 render_rgb_inference: bool = True, params: Optional[Dict[str, torch.Tensor]] = None,
                 semantics: Optional[Semantics] = None, **dm_kwargs):
        self.config = config
        self.test_mode = test_mode
        self.world_size = world_size
        self.local_rank = local_rank
        self.datamanager = FruitDataManager(config.datamanager, cameras, device=device, test_mode=test_mode,
                                            world_size=world_size, local_rank=local_rank, **dm_kwargs)
        self.model = FruitModel(config.model, scene_box=scene_box, num_train_data=len(cameras),
                                metadata={"semantics": semantics or Semantics()}, device=device,
                                grad_scaler=grad_scaler, test_mode=test_mode,
                                render_rgb_inference=render_rgb_inference, params=params)

    def eval(self):
        self.model.eval()
        return self
