from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Type

import torch

from nerfstudio.configs.base_config import PrintableConfig


@dataclass
class OptimizerConfig(PrintableConfig):
    _target: Type = torch.optim.Adam
    lr: float = 0.0005
    eps: float = 1e-08
    max_norm: Optional[float] = None

    def setup(self, params) -> torch.optim.Optimizer:
        kwargs = {k: v for k, v in vars(self).items() if k not in ("_target", "max_norm")}
        return self._target(params, **kwargs)


@dataclass
class AdamOptimizerConfig(OptimizerConfig):
    _target: Type = torch.optim.Adam
    weight_decay: float = 0


@dataclass
class RAdamOptimizerConfig(OptimizerConfig):
    _target: Type = torch.optim.RAdam
    weight_decay: float = 0


class Optimizers:
    """config: {group: {"optimizer": OptimizerConfig, "scheduler": SchedulerConfig | None}}"""

    def __init__(self, config: Dict[str, Any], param_groups: Dict[str, List[torch.nn.Parameter]]):
        self.config, self.optimizers, self.schedulers, self.parameters = config, {}, {}, {}
        for name, params in param_groups.items():
            self.optimizers[name] = config[name]["optimizer"].setup(params=params)
            self.parameters[name] = params
            if config[name].get("scheduler"):
                self.schedulers[name] = config[name]["scheduler"].setup().get_scheduler(
                    optimizer=self.optimizers[name], lr_init=config[name]["optimizer"].lr)

    def zero_grad_all(self):
        for o in self.optimizers.values():
            o.zero_grad()

    def optimizer_step_all(self):
        for o in self.optimizers.values():
            o.step()

    def scheduler_step_all(self, step: int):
        for s in self.schedulers.values():
            s.step()
