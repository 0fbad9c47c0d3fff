from dataclasses import dataclass, field
from typing import Dict, Optional

from torch import Tensor


@dataclass
class RayBundle:
    origins: Tensor
    directions: Tensor
    pixel_area: Tensor
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def __len__(self):
        return self.origins.shape[0]
