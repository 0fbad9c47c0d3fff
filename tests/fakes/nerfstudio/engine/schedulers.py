from dataclasses import dataclass, field
from typing import Optional, Type

import numpy as np
from torch.optim import lr_scheduler

from nerfstudio.configs.base_config import InstantiateConfig


class Scheduler:
    def __init__(self, config):
        self.config = config


class ExponentialDecayScheduler(Scheduler):
    def get_scheduler(self, optimizer, lr_init: float):
        lr_final = self.config.lr_final if self.config.lr_final is not None else lr_init

        def func(step):
            t = np.clip(step / self.config.max_steps, 0, 1)
            return np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t) / lr_init

        return lr_scheduler.LambdaLR(optimizer, lr_lambda=func)


@dataclass
class ExponentialDecaySchedulerConfig(InstantiateConfig):
    _target: Type = ExponentialDecayScheduler
    lr_pre_warmup: float = 1e-8
    lr_final: Optional[float] = None
    warmup_steps: int = 0
    max_steps: int = 100000
    ramp: str = "cosine"

    def setup(self, **kwargs):
        return self._target(self)
