"""``FruitNerf`` dataparser -- mirror of ``crop_nerf/fruit_nerf/data/fruitnerf_dataparser.py`` (the dataparser of
``fruit_nerf_method_huge``, ``fruit_nerf_config.py:130``).  It is the CottonNerf parser with two differences: the mask of
a frame is named by the frame's ``"semantic_path"`` entry (down-scaled copies live in ``semantics_<k>`` folders,
``:141-148``) and the default train split is 0.9 (``:62``)."""

from __future__ import annotations

from dataclasses import dataclass

from .cotton_nerf_dataparser import CottonNerf, CottonNerfDataParserConfig, DataparserOutputs  # noqa: F401


@dataclass
class FruitNerfDataParserConfig(CottonNerfDataParserConfig):
    train_split_fraction: float = 0.9
    semantics_from_frames: bool = True


FruitNerf = CottonNerf
