from dataclasses import dataclass

from nerfstudio.engine.trainer import TrainerConfig


@dataclass
class MethodSpecification:
    config: TrainerConfig
    description: str
