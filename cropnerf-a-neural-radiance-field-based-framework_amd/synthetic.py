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
