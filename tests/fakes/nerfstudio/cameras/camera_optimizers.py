from dataclasses import dataclass, field
from typing import Type

from nerfstudio.configs.base_config import InstantiateConfig


class CameraOptimizer:
    pass


@dataclass
class CameraOptimizerConfig(InstantiateConfig):
    _target: Type = CameraOptimizer
    mode: str = "off"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3
