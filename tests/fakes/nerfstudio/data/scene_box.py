from dataclasses import dataclass

from torch import Tensor


@dataclass
class SceneBox:
    aabb: Tensor
