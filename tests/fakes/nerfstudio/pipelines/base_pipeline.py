from dataclasses import dataclass, field
from typing import Any, Optional, Type

import torch
from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig
from nerfstudio.models.base_model import ModelConfig


class Pipeline(nn.Module):
    datamanager: Any
    _model: Any
    world_size: int = 1

    @property
    def model(self):
        return self._model

    @property
    def device(self):
        return self.model.device


@dataclass
class VanillaPipelineConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: VanillaPipeline)
    datamanager: Any = None
    model: ModelConfig = field(default_factory=ModelConfig)


class VanillaPipeline(Pipeline):
    def __init__(self, config, device: str, test_mode: str = "val", world_size: int = 1, local_rank: int = 0, grad_scaler=None):
        super().__init__()
        self.config, self.test_mode = config, test_mode
        self.datamanager = config.datamanager.setup(device=device, test_mode=test_mode, world_size=world_size, local_rank=local_rank)
        self._model = config.model.setup(scene_box=self.datamanager.train_dataset.scene_box,
                                         num_train_data=len(self.datamanager.train_dataset),
                                         metadata=self.datamanager.train_dataset.metadata, device=device, grad_scaler=grad_scaler)
        self.model.to(device)
        self.world_size = world_size

    def get_train_loss_dict(self, step: int):
        ray_bundle, batch = self.datamanager.next_train(step)
        model_outputs = self._model(ray_bundle)
        metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
        loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        return model_outputs, loss_dict, metrics_dict

    def get_param_groups(self):
        return {**self.datamanager.get_param_groups(), **self.model.get_param_groups()}

    def get_training_callbacks(self, attrs):
        return self.datamanager.get_training_callbacks(attrs) + self.model.get_training_callbacks(attrs)
