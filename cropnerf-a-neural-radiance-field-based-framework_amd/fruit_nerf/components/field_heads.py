"""``SemanticFieldHead`` -- ``crop_nerf/fruit_nerf/components/field_heads.py:29-40``: a bias-ful ``Linear(in_dim,
num_classes)`` without activation behind ``mlp_semantics`` (``fruit_field.py:153-166``).

On the HIP path the head is never a separate launch: ``prep_kernel`` folds it into the last layer of ``mlp_semantics``
when it builds the weight image (two Linears without an activation between them are one), and the training kernels carry
it as the ``sem_head_weight / sem_head_bias`` pair of ``cn_field_params``.  This class is the one place that knows the
head's state-dict names, shapes and nerfstudio initialisation; ``config.param_shapes``, ``ops.FieldHandle`` and the tcnn
checkpoint converters (``tcnn_params.py``) take them from here.  The kernels composite ONE logit per sample
(``cn_field_params`` has no class stride), so ``num_classes`` other than the reference's 1 (``fruit_field.py:164``) is
refused here rather than rendered wrongly."""

from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import Tensor


class SemanticFieldHead:
    prefix = "field.field_head_semantics.net"
    weight_key = prefix + ".weight"
    bias_key = prefix + ".bias"
    keys = (weight_key, bias_key)

    def __init__(self, in_dim: int, num_classes: int = 1, activation=None) -> None:
        if activation is not None:
            raise ValueError("SemanticFieldHead has no activation (field_heads.py:37)")
        if num_classes != 1:
            raise NotImplementedError(f"the HIP field composites one semantic logit per sample; num_classes={num_classes}")
        self.in_dim = int(in_dim)
        self.num_classes = int(num_classes)

    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        """``nn.Linear`` layout: weight [out, in], bias [out]."""
        return {self.weight_key: (self.num_classes, self.in_dim), self.bias_key: (self.num_classes,)}

    def init(self, generator: torch.Generator) -> Dict[str, Tensor]:
        """``nn.Linear.reset_parameters``: both uniform in +-1/sqrt(in_dim)."""
        bound = 1.0 / math.sqrt(self.in_dim)
        return {k: (torch.rand(s, generator=generator) * 2.0 - 1.0) * bound for k, s in self.shapes().items()}

    def check(self, params: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
        """The (weight, bias) pair of a parameter dict, shape-checked against this head."""
        for k, s in self.shapes().items():
            if k not in params:
                raise KeyError(f"parameter dict has no {k}")
            if tuple(params[k].shape) != s:
                raise ValueError(f"{k}: shape {tuple(params[k].shape)}, expected {s}")
        return params[self.weight_key], params[self.bias_key]

    def __call__(self, features: Tensor, params: Dict[str, Tensor]) -> Tensor:
        """The head as a separate op on materialised ``mlp_semantics`` outputs [..., in_dim] (what the reference's
        ``get_outputs`` does, ``fruit_field.py:262-267``); the renderers never call this -- the kernels apply the folded
        head."""
        w, b = self.check(params)
        return torch.nn.functional.linear(features, w.to(features.dtype), b.to(features.dtype))
