from dataclasses import dataclass, field
from typing import Any, Type


class PrintableConfig:
    pass


@dataclass
class InstantiateConfig(PrintableConfig):
    _target: Type = object

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


@dataclass
class ViewerConfig(PrintableConfig):
    num_rays_per_chunk: int = 32768
    websocket_port_default: int = 7007


@dataclass
class MachineConfig(PrintableConfig):
    seed: int = 42
    num_devices: int = 1
    device_type: str = "cuda"
