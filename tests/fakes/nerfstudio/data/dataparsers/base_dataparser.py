from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional, Type

from nerfstudio.configs.base_config import InstantiateConfig


@dataclass
class DataparserOutputs:
    image_filenames: List[Path]
    cameras: Any
    alpha_color: Any = None
    scene_box: Any = None
    mask_filenames: Optional[List[Path]] = None
    metadata: Dict[str, Any] = field(default_factory=dict)
    dataparser_transform: Any = None
    dataparser_scale: float = 1.0


@dataclass
class DataParserConfig(InstantiateConfig):
    _target: Type = field(default_factory=lambda: DataParser)
    data: Path = Path()


class DataParser:
    def __init__(self, config):
        self.config = config

    def get_dataparser_outputs(self, split: str = "train", **kwargs):
        return self._generate_dataparser_outputs(split, **kwargs)
