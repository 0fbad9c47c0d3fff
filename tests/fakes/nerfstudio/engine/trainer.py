from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, Optional, Type

from nerfstudio.configs.base_config import InstantiateConfig, MachineConfig, ViewerConfig


class Trainer:
    pass


@dataclass
class TrainerConfig(InstantiateConfig):
    _target: Type = Trainer
    method_name: Optional[str] = None
    experiment_name: Optional[str] = None
    output_dir: Path = Path("outputs")
    timestamp: str = "{timestamp}"
    machine: MachineConfig = field(default_factory=MachineConfig)
    viewer: ViewerConfig = field(default_factory=ViewerConfig)
    pipeline: Any = None
    optimizers: Dict[str, Any] = field(default_factory=dict)
    vis: str = "wandb"
    data: Optional[Path] = None
    relative_model_dir: Path = Path("nerfstudio_models/")
    steps_per_save: int = 1000
    steps_per_eval_batch: int = 500
    steps_per_eval_image: int = 500
    steps_per_eval_all_images: int = 25000
    max_num_iterations: int = 1000000
    mixed_precision: bool = False
    use_grad_scaler: bool = False
    save_only_latest_checkpoint: bool = True
    load_dir: Optional[Path] = None
    load_step: Optional[int] = None
    load_config: Optional[Path] = None
    gradient_accumulation_steps: Dict[str, int] = field(default_factory=dict)
