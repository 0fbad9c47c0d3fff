from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Type

from torch import nn

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class DataManagerConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: DataManager)
    data: Optional[Path] = None
    masks_on_gpu: bool = False
    images_on_gpu: bool = False


class DataManager(nn.Module):
    train_dataset: Any = None
    eval_dataset: Any = None

    def __init__(self):
        super().__init__()
        self.train_count = 0
        self.eval_count = 0

    def get_param_groups(self) -> Dict[str, List]:
        return {}

    def get_training_callbacks(self, attrs) -> List:
        return []


@dataclass
class VanillaDataManagerConfig(DataManagerConfig):
    _target: Type = field(default_factory=lambda: VanillaDataManager)
    dataparser: Any = None
    train_num_rays_per_batch: int = 1024
    train_num_images_to_sample_from: int = -1
    train_num_times_to_repeat_images: int = -1
    eval_num_rays_per_batch: int = 1024
    eval_num_images_to_sample_from: int = -1
    eval_num_times_to_repeat_images: int = -1
    eval_image_indices: Optional[tuple] = (0,)
    camera_res_scale_factor: float = 1.0
    patch_size: int = 1


class VanillaDataManager(DataManager):
    def __init__(self, config, device="cpu", test_mode="val", world_size=1, local_rank=0, **kwargs):
        super().__init__()
        self.config, self.device, self.test_mode = config, device, test_mode
        self.world_size, self.local_rank = world_size, local_rank
        self.dataparser = config.dataparser.setup() if config.dataparser is not None else None
