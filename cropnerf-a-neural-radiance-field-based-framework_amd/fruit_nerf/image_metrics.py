"""``FruitModel.get_image_metrics_and_images`` (``crop_nerf/fruit_nerf/fruit_nerf.py:647-700``) and the upstream pieces it
calls, restated on plain torch ops (reductions and one small convolution over a single image: plumbing, not a hot path):

* nerfstudio ``colormaps.apply_colormap`` / ``apply_depth_colormap`` with their default options (single-channel float ->
  matplotlib's "turbo" table indexed by ``(x * 255).long()``; depth normalised by its own min / max and faded to white by
  ``1 - accumulation``);
* torchmetrics ``PeakSignalNoiseRatio(data_range=1.0)`` and ``structural_similarity_index_measure`` (11 x 11 Gaussian
  window, sigma 1.5, k1 = 0.01, k2 = 0.03, mean over the windows that lie inside the image);
* ``BinaryJaccardIndex`` as the reference applies it: to ``softmax(semantics)`` over the LAST axis of an [H,W,1] tensor,
  i.e. to a constant 1 -- so "iou" is the fruit fraction of the mask (reference quirk, kept; ``:694-698``).

LPIPS needs the pretrained AlexNet weights torchmetrics downloads; there is no network here, so ``lpips`` is reported as
NaN unless the caller passes an ``lpips_fn``."""

from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

_TABLES: Dict[str, Tensor] = {}


def _table(name: str, device) -> Tensor:
    if name not in _TABLES:
        import matplotlib

        _TABLES[name] = torch.tensor(matplotlib.colormaps[name].colors, dtype=torch.float32)
    return _TABLES[name].to(device)


def apply_float_colormap(image: Tensor, colormap: str = "turbo") -> Tensor:
    """[...,1] in [0,1] -> [...,3]."""
    image = torch.nan_to_num(image, 0)
    idx = (image * 255).long()
    assert int(idx.min()) >= 0 and int(idx.max()) <= 255, "the colormap input must lie in [0, 1]"
    return _table(colormap, image.device)[idx[..., 0]]


def apply_colormap(image: Tensor) -> Tensor:
    """``colormaps.apply_colormap`` with default options for what the reference passes (float [H,W,1])."""
    return apply_float_colormap(image, "turbo")


def apply_depth_colormap(depth: Tensor, accumulation: Optional[Tensor] = None, near_plane: Optional[float] = None,
                         far_plane: Optional[float] = None) -> Tensor:
    near = float(torch.min(depth)) if near_plane is None else near_plane
    far = float(torch.max(depth)) if far_plane is None else far_plane
    depth = torch.clip((depth - near) / (far - near + 1e-10), 0, 1)
    colored = apply_float_colormap(depth, "turbo")
    if accumulation is not None:
        colored = colored * accumulation + (1 - accumulation)
    return colored


def psnr(preds: Tensor, target: Tensor, data_range: float = 1.0) -> Tensor:
    return 10.0 * torch.log10(data_range ** 2 / torch.mean((preds - target) ** 2))


def ssim(preds: Tensor, target: Tensor, data_range: float = 1.0, kernel_size: int = 11, sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> Tensor:
    """[1,C,H,W] images -> scalar (torchmetrics defaults: reflect-pad, Gaussian window, crop the padded rim, mean)."""
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    x = torch.arange(kernel_size, dtype=preds.dtype, device=preds.device) - (kernel_size - 1) / 2.0
    g = torch.exp(-(x / sigma) ** 2 / 2)
    g = (g / g.sum()).unsqueeze(0)
    C = preds.shape[1]
    kernel = (g.T @ g).expand(C, 1, kernel_size, kernel_size)
    pad = (kernel_size - 1) // 2
    p = F.pad(preds, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(target, (pad, pad, pad, pad), mode="reflect")
    stack = torch.cat([p, t, p * p, t * t, p * t])
    out = F.conv2d(stack, kernel, groups=C)
    mu_p, mu_t, pp, tt, pt = out.split(preds.shape[0])
    s_pp, s_tt, s_pt = pp - mu_p * mu_p, tt - mu_t * mu_t, pt - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p * mu_p + mu_t * mu_t + c1) * (s_pp + s_tt + c2))
    return full[..., pad:-pad, pad:-pad].mean()


def binary_jaccard_index(preds: Tensor, target: Tensor, threshold: float = 0.5) -> Tensor:
    """torchmetrics ``BinaryJaccardIndex``: float predictions are thresholded at 0.5 (probabilities) or sigmoid-ed first
    when they fall outside [0, 1]; IoU of the positive class, 0 when the union is empty."""
    if preds.is_floating_point():
        if not bool(((preds >= 0) & (preds <= 1)).all()):
            preds = preds.sigmoid()
        preds = preds > threshold
    p, t = preds.reshape(-1).bool(), target.reshape(-1) > 0.5
    inter, union = (p & t).sum(), (p | t).sum()
    return torch.where(union > 0, inter.float() / union.float().clamp_min(1), torch.zeros((), device=preds.device))


def get_image_metrics_and_images(model, outputs: Dict[str, Tensor], batch: Dict[str, Tensor],
                                 lpips_fn: Optional[Callable[[Tensor, Tensor], Tensor]] = None
                                 ) -> Tuple[Dict[str, float], Dict[str, Tensor]]:
    dev = model.device
    image = batch["image"].to(dev).to(torch.float32)[..., :3]
    out = {k: v.to(dev) for k, v in outputs.items() if isinstance(v, Tensor)}
    rgb = torch.clamp(out["rgb"], min=0, max=1)
    acc = apply_colormap(out["accumulation"])
    depth = apply_depth_colormap(out["depth"], accumulation=out["accumulation"])
    combined_rgb = torch.cat([image, rgb], dim=1)
    image_c = torch.moveaxis(image, -1, 0)[None, ...]
    rgb_c = torch.moveaxis(rgb, -1, 0)[None, ...]
    metrics_dict = {"psnr": float(psnr(image_c, rgb_c)), "ssim": float(ssim(image_c, rgb_c)),
                    "lpips": float(lpips_fn(image_c, rgb_c)) if lpips_fn is not None else math.nan}
    images_dict = {"img": combined_rgb, "accumulation": acc, "depth": depth}
    for i in range(len(model.proposal_networks)):
        key = f"prop_depth_{i}"
        if key in out:
            images_dict[key] = apply_depth_colormap(out[key], accumulation=out["accumulation"])
    images_dict["semantics_colormap"] = torch.sigmoid(out["semantics"])
    mask = batch["fruit_mask"].to(dev)
    images_dict["fruit_mask"] = mask.repeat(1, 1, 3)
    semantic_labels = torch.softmax(out["semantics"], dim=-1)  # over a size-1 axis: all ones (reference quirk)
    metrics_dict["iou"] = float(binary_jaccard_index(semantic_labels[..., 0], mask[..., 0]))
    return metrics_dict, images_dict
