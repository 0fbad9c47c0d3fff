"""Oracle: volume-rendering reductions.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates the upstream nerfstudio 1.1.3 renderers wired at
``fruit_nerf/fruit_nerf.py:170-174`` and used at ``:560-564,583,589-597``:
``RGBRenderer`` ("last_sample" background by default, fixed colour when overridden as in
``fruit_nerf/scripts/semantic_projection.py:158,169``), ``AccumulationRenderer``, ``DepthRenderer("median")``,
``SemanticRenderer`` and the semantic colormap threshold (``fruit_nerf.py:593-597``).
"""

from __future__ import annotations

from typing import Optional, Union

import torch
from torch import Tensor


def render_rgb(rgb: Tensor, weights: Tensor, background: Union[str, Tensor] = "last_sample",
               training: bool = False) -> Tensor:
    """``RGBRenderer.forward``: eval -> nan_to_num before, clamp [0,1] after."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = torch.sum(weights * rgb, dim=-2)
    acc = torch.sum(weights, dim=-2)
    if isinstance(background, str):
        assert background == "last_sample"
        bg = rgb[..., -1, :]
    else:
        bg = background.to(comp).expand(comp.shape)
    comp = comp + bg * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, min=0.0, max=1.0)
    return comp


def render_accumulation(weights: Tensor) -> Tensor:
    return torch.sum(weights, dim=-2)


def render_depth_median(weights: Tensor, starts: Tensor, ends: Tensor) -> Tensor:
    """``DepthRenderer(method="median")``: mid-t at the first index with cumsum(w) >= 0.5 (clamped to S-1)."""
    steps = (starts + ends) / 2
    cum = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.ones((*weights.shape[:-2], 1)) * 0.5
    idx = torch.searchsorted(cum, split, side="left")
    idx = torch.clamp(idx, 0, steps.shape[-2] - 1)
    return torch.gather(steps[..., 0], dim=-1, index=idx)


def render_semantics(semantics: Tensor, weights: Tensor) -> Tensor:
    """``SemanticRenderer``: sum_s w * logit (weights detached in the reference, ``fruit_nerf.py:586-591``)."""
    return torch.sum(weights * semantics, dim=-2)


def semantics_colormap(sem: Tensor, colormap: Optional[Tensor] = None, repeat3: bool = True) -> Tensor:
    """``fruit_nerf.py:593-597``: heaviside(sigmoid(sem) - 0.9, 0) -> long -> colormap[label] -> repeat(1,3).
    colormap = (0, 1) from ``data/cotton_nerf_dataparser.py:248-254``."""
    if colormap is None:
        colormap = torch.tensor([0.0, 1.0])
    labels = torch.heaviside(torch.sigmoid(sem) - 0.9, torch.tensor(0.0)).to(torch.long)
    out = colormap[labels]
    return out.repeat(1, 3) if repeat3 else out
