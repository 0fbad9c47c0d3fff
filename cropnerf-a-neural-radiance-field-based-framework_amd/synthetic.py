"""Synthetic inputs for tests and bench: the 3DCotton data is not available here or on the
GPU box, so scenes are parameter sets and cameras are an analytic orbit.

* cameras: N poses on a radius-0.8 orbit at two elevations (+-20 deg) looking at the origin, OpenGL convention
  (-z forward, +y up in camera space; world z up), H=W=800, fx=fy=1111.1, cx=cy=400 (C1: 400x400, 555.6).
* parameters: *P-rand* -- hash tables U(-1,1)*0.1 (the reference's 1e-3 init gives an empty volume), Linear layers
  the nn.Linear default init, embeddings N(0,1).
"""

from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

from .config import FieldSpec, ProposalSpec, init_params


def orbit_cameras(num: int = 100, radius: float = 0.8, elevation_deg: float = 20.0, height: int = 800,
                  width: int = 800, focal: float = 1111.1) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (c2w [N,3,4], intrinsics [N,4] = fx,fy,cx,cy), float32 CPU tensors."""
    c2w = torch.zeros(num, 3, 4, dtype=torch.float64)
    for i in range(num):
        az = 2.0 * math.pi * i / num
        el = math.radians(elevation_deg if i % 2 == 0 else -elevation_deg)
        pos = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az),
                            radius * math.sin(el)], dtype=torch.float64)
        fwd = -pos / pos.norm()  # look at the origin
        up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        cam_up = torch.linalg.cross(right, fwd)
        # columns: camera x (right), y (up), z (backward = -forward)
        c2w[i, :, 0] = right
        c2w[i, :, 1] = cam_up
        c2w[i, :, 2] = -fwd
        c2w[i, :, 3] = pos
    intr = torch.tensor([[focal, focal, width / 2.0, height / 2.0]], dtype=torch.float32).repeat(num, 1)
    return c2w.to(torch.float32), intr


def p_rand(spec: FieldSpec, prop_specs: List[ProposalSpec], seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Parameter set P-rand."""
    return init_params(spec, prop_specs, seed=seed, grid_scale=0.1, device=device)


SCENE_AABB = ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))  # SceneBox(+-1), data/cotton_nerf_dataparser.py


# ------------------------------------------------------------------------------------------------------------------
# Analytic "cotton plant" (parameter set P-fit): three coloured spheres ("bolls", semantic class 1)
# on a vertical cylinder ("stem", class 0) in front of a constant background, rendered in closed form.  It is the
# stand-in for the 3DCotton images: ground-truth pixels and fruit masks for any camera, so that the training rows can
# be exercised end to end and PSNR has a meaning.
# ------------------------------------------------------------------------------------------------------------------
BOLLS = (  # centre xyz, radius, rgb
    ((0.10, 0.00, 0.12), 0.085, (0.92, 0.90, 0.84)),
    ((-0.06, 0.09, -0.05), 0.075, (0.90, 0.62, 0.58)),
    ((-0.04, -0.10, 0.22), 0.065, (0.62, 0.80, 0.92)),
)
STEM = (0.035, -0.35, 0.35, (0.36, 0.25, 0.12))  # radius, z0, z1, rgb
BACKGROUND = (0.08, 0.09, 0.12)
LIGHT = (0.3, 0.5, 0.81)


def analytic_render(origins: torch.Tensor, directions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Closed-form image of the analytic plant for rays [..., 3] (unit directions): rgb [...,3], fruit mask [...,1]
    (1 where a boll is the first hit), depth [...,1] (1e10 on a miss).  Lambert-like view-independent shading."""
    rgb, mask, depth, _ = _analytic_hits(origins, directions)
    return rgb, mask, depth


def analytic_instance_labels(origins: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
    """Instance label per ray [...] uint8: b + 1 where boll b is the first hit, 0 elsewhere -- the analytic plant's stand-in
    for the instance-segmentation frames (``label_frame*.png``) the merger reads beside the projections
    (``segmentation/merger.py:221,240-248``)."""
    return _analytic_hits(origins, directions)[3]


def _analytic_hits(origins: torch.Tensor, directions: torch.Tensor):
    o, d = origins.to(torch.float32), directions.to(torch.float32)
    dev = o.device
    light = torch.tensor(LIGHT, device=dev)
    light = light / light.norm()
    best_t = torch.full(o.shape[:-1], 1e10, device=dev)
    rgb = torch.tensor(BACKGROUND, device=dev).expand(*o.shape[:-1], 3).clone()
    mask = torch.zeros(*o.shape[:-1], device=dev)
    inst = torch.zeros(o.shape[:-1], dtype=torch.uint8, device=dev)

    def shade(base, normal):
        k = 0.55 + 0.45 * (normal * light).sum(-1).clamp(min=0.0)
        return torch.tensor(base, device=dev) * k[..., None]

    for b_id, (centre, radius, colour) in enumerate(BOLLS):
        c = torch.tensor(centre, device=dev)
        oc = o - c
        b = (oc * d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - radius * radius)
        t = -b - torch.sqrt(disc.clamp(min=0.0))
        hit = (disc > 0) & (t > 0) & (t < best_t)
        n = (oc + d * t[..., None]) / radius
        rgb = torch.where(hit[..., None], shade(colour, n), rgb)
        mask = torch.where(hit, torch.ones_like(mask), mask)
        inst = torch.where(hit, torch.full_like(inst, b_id + 1), inst)
        best_t = torch.where(hit, t, best_t)
    radius, z0, z1, colour = STEM
    a = (d[..., :2] ** 2).sum(-1).clamp(min=1e-12)
    b = (o[..., :2] * d[..., :2]).sum(-1)
    cc = (o[..., :2] ** 2).sum(-1) - radius * radius
    disc = b * b - a * cc
    t = (-b - torch.sqrt(disc.clamp(min=0.0))) / a
    z = o[..., 2] + d[..., 2] * t
    hit = (disc > 0) & (t > 0) & (t < best_t) & (z > z0) & (z < z1)
    p = o + d * t[..., None]
    n = torch.cat([p[..., :2] / radius, torch.zeros_like(p[..., :1])], -1)
    rgb = torch.where(hit[..., None], shade(colour, n), rgb)
    mask = torch.where(hit, torch.zeros_like(mask), mask)
    inst = torch.where(hit, torch.zeros_like(inst), inst)
    best_t = torch.where(hit, t, best_t)
    return rgb, mask[..., None], best_t[..., None], inst


def analytic_dataset(cameras, device="cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """Ground-truth images [N,H,W,3] and fruit masks [N,H,W,1] of the analytic plant for ``cameras`` (rays from the
    product ray generator, so pixel centres follow the reference's convention)."""
    cams = cameras.to(device)
    images, masks = [], []
    for i in range(len(cams)):
        rb = cams.generate_rays(i, keep_shape=True)
        rgb, m, _ = analytic_render(rb.origins, rb.directions)
        images.append(rgb)
        masks.append(m)
    return torch.stack(images), torch.stack(masks)


def write_capture(path, num: int = 24, res: int = 64, radius: float = 0.8, world_shift=(0.3, -0.2, 0.1),
                  world_scale: float = 2.5) -> str:
    """Writes the analytic plant as a nerfstudio-format capture (what ``CottonNerf`` parses, ``data/cotton_nerf_dataparser.py``):
    ``transforms.json`` (shared intrinsics, one 4x4 ``transform_matrix`` per frame), ``images/frame_%05d.png`` and
    ``semantics/frame_%05d.png`` (fruit = 255).  The poses are written in a shifted, scaled world frame so that the
    dataparser's centring and auto-scaling have something to undo.  CPU only (closed-form renderer + PIL)."""
    import json
    import os

    import numpy as np
    from PIL import Image

    focal = 1111.1 * res / 800.0
    c2w, _ = orbit_cameras(num, radius=radius, height=res, width=res, focal=focal)
    os.makedirs(os.path.join(path, "images"), exist_ok=True)
    os.makedirs(os.path.join(path, "semantics"), exist_ok=True)
    ys, xs = torch.meshgrid(torch.arange(res, dtype=torch.float32), torch.arange(res, dtype=torch.float32), indexing="ij")
    cam_dirs = torch.stack([(xs + 0.5 - res / 2.0) / focal, -(ys + 0.5 - res / 2.0) / focal, -torch.ones_like(xs)], -1)
    frames = []
    shift = torch.tensor(world_shift, dtype=torch.float32)
    for i in range(num):
        d = cam_dirs @ c2w[i, :, :3].T
        d = d / d.norm(dim=-1, keepdim=True)
        o = c2w[i, :, 3].expand_as(d)
        rgb, mask, _ = analytic_render(o.reshape(-1, 3), d.reshape(-1, 3))
        img = (rgb.reshape(res, res, 3).clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).numpy()
        Image.fromarray(img, "RGB").save(os.path.join(path, "images", f"frame_{i + 1:05d}.png"))
        m = (mask.reshape(res, res) > 0.5).to(torch.uint8).numpy() * 255
        Image.fromarray(m, "L").save(os.path.join(path, "semantics", f"frame_{i + 1:05d}.png"))
        pose = torch.eye(4)
        pose[:3, :3] = c2w[i, :, :3]
        pose[:3, 3] = c2w[i, :, 3] * world_scale + shift
        frames.append({"file_path": f"images/frame_{i + 1:05d}.png", "semantic_path": f"semantics/frame_{i + 1:05d}.png",
                       "transform_matrix": pose.tolist()})
    meta = {"fl_x": focal, "fl_y": focal, "cx": res / 2.0, "cy": res / 2.0, "w": res, "h": res, "k1": 0, "k2": 0, "p1": 0,
            "p2": 0, "frames": frames}
    with open(os.path.join(path, "transforms.json"), "w", encoding="UTF-8") as f:
        json.dump(meta, f)
    return str(path)


def write_transforms_json(path, frame_numbers, transform_matrices, intrinsics, size_hw, orientation_override=None,
                          auto_scale_poses_override=None) -> str:
    """A nerfstudio-format ``transforms.json`` (shared intrinsics, ``images/frame_%05d.jpg`` names, no distortion) from
    arrays -- e.g. the camera data of the reference's real capture file kept in ``tests/golden/capture_3dcotton.npz``
    (``fruit_nerf/utils/transforms.json``: 147 frames, 1920 x 1440).  Only the description is written, no images: parse it
    with ``downscale_factor=1`` (the automatic choice opens the first image)."""
    import json
    import os

    fx, fy, cx, cy = (float(v) for v in intrinsics)
    meta = {"fl_x": fx, "fl_y": fy, "cx": cx, "cy": cy, "h": int(size_hw[0]), "w": int(size_hw[1]),
            "k1": 0, "k2": 0, "p1": 0, "p2": 0}
    if orientation_override is not None:
        meta["orientation_override"] = orientation_override
    if auto_scale_poses_override is not None:
        meta["auto_scale_poses_override"] = bool(auto_scale_poses_override)
    meta["frames"] = [{"file_path": f"images/frame_{int(n):05d}.jpg", "transform_matrix": [[float(v) for v in row] for row in m]}
                      for n, m in zip(frame_numbers, transform_matrices)]
    os.makedirs(path, exist_ok=True)
    out = os.path.join(path, "transforms.json")
    with open(out, "w", encoding="UTF-8") as f:
        json.dump(meta, f)
    return out
