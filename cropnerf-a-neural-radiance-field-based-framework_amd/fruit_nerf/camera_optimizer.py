"""Host-side view of the camera pose refinement (nerfstudio's ``CameraOptimizer`` in its ``SO3xR3`` mode, built at
``crop_nerf/fruit_nerf/fruit_nerf.py:114-116``), for the places where the reference asks it for whole POSES instead of
correcting rays: ``collect_camera_poses_for_dataset`` (``export/exporter_utils_nerfacto.py:318-324``) writes
``camera_optimizer.apply_to_camera(camera)`` of every training camera into ``transforms_train.json``.

The per-ray correction (``apply_to_raybundle``: o += t, d = R d) is the HIP kernel ``cn_apply_pose_adjustment``; this
module only turns the [num_cameras, 6] parameter into 3 x 4 matrices for at most a few hundred cameras -- host
arithmetic in float32, the closed form of the same exponential map (angle from the squared norm clamped at 1e-4, as
upstream ``exp_map_SO3xR3`` and the kernel do).

Note the two compositions differ upstream and are mirrored as they are: rays get ``R d`` and ``o + t`` (the correction
acts in the WORLD frame), ``apply_to_camera`` returns ``c2w @ [[R, t], [0, 1]]`` (the correction acts in the CAMERA
frame)."""

from __future__ import annotations

from typing import Optional, Sequence, Union

import torch
from torch import Tensor


def pose_correction_matrices(pose_adjustment: Tensor) -> Tensor:
    """[N,6] tangent vectors (translation, rotation vector w) -> [N,3,4] ``[R | t]`` with
    R = I + (sin a / a) K + ((1 - cos a) / a^2) K^2,  a = sqrt(max(|w|^2, 1e-4)),  K = [w]_x,
    written out per element (K^2 = w w^T - |w|^2 I)."""
    p = pose_adjustment.detach().to(torch.float32).cpu()
    t, w = p[:, :3], p[:, 3:]
    sq = (w * w).sum(dim=1)
    a = torch.clamp(sq, min=1e-4).sqrt()
    f1 = (a.sin() / a)[:, None]
    f2 = ((1.0 - a.cos()) / (a * a))[:, None]
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    one = torch.ones_like(wx)
    rows = [
        torch.cat([one - f2 * (wy * wy + wz * wz), f2 * wx * wy - f1 * wz, f2 * wx * wz + f1 * wy, t[:, 0:1]], dim=1),
        torch.cat([f2 * wx * wy + f1 * wz, one - f2 * (wx * wx + wz * wz), f2 * wy * wz - f1 * wx, t[:, 1:2]], dim=1),
        torch.cat([f2 * wx * wz - f1 * wy, f2 * wy * wz + f1 * wx, one - f2 * (wx * wx + wy * wy), t[:, 2:3]], dim=1),
    ]
    return torch.stack(rows, dim=1)


class CameraOptimizer:
    """The subset of nerfstudio's ``CameraOptimizer`` the exporters call.  ``pose_adjustment`` is the model's own
    parameter tensor (shared, not copied), so poses exported after training are the trained ones."""

    def __init__(self, pose_adjustment: Tensor, mode: str = "SO3xR3",
                 non_trainable_camera_indices: Optional[Sequence[int]] = None):
        if mode not in ("off", "SO3xR3"):
            raise ValueError(f"camera optimizer mode {mode!r}: the reference configures 'SO3xR3' (fruit_nerf.py:114)")
        self.mode = mode
        self.pose_adjustment = pose_adjustment
        self.num_cameras = int(pose_adjustment.shape[0])
        self.non_trainable_camera_indices = None if non_trainable_camera_indices is None else \
            torch.as_tensor(list(non_trainable_camera_indices), dtype=torch.long)

    def forward(self, indices: Union[Tensor, Sequence[int]]) -> Tensor:
        """[n] camera indices -> [n,3,4] corrections (identity when the optimiser is off, and for cameras listed as
        non-trainable)."""
        idx = torch.as_tensor(indices, dtype=torch.long).reshape(-1).cpu()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= self.num_cameras):
            raise IndexError(f"camera index out of range for {self.num_cameras} cameras")
        if self.mode == "off":
            return torch.eye(4)[None, :3, :4].repeat(idx.shape[0], 1, 1)
        adj = self.pose_adjustment.detach().to(torch.float32).cpu()
        if self.non_trainable_camera_indices is not None:
            adj = adj.clone()
            adj[self.non_trainable_camera_indices] = 0.0
        return pose_correction_matrices(adj[idx])

    __call__ = forward

    def apply_to_camera(self, camera) -> Tensor:
        """``c2w @ [[R, t], [0, 1]]`` for the batch of cameras in ``camera`` ([n,3,4]); the camera's row in the parameter comes
        from ``camera.metadata["cam_idx"]``.  Without that key (eval cameras) or with the optimiser off: the stored pose."""
        c2w = camera.camera_to_worlds.detach().to(torch.float32).cpu()
        if self.mode == "off":
            return c2w
        if camera.metadata is None:
            raise AssertionError("Must provide id of camera in its metadata")
        if "cam_idx" not in camera.metadata:
            return c2w
        adj = self.forward([int(camera.metadata["cam_idx"])])
        bottom = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]])
        return torch.bmm(c2w, torch.cat([adj, bottom], dim=1).expand(c2w.shape[0], 4, 4))
