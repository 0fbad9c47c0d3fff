"""``CottonNerf`` dataparser -- mirror of ``crop_nerf/fruit_nerf/data/cotton_nerf_dataparser.py:43-331`` (SURVEY.md
section 8(f) row 4, the wire format in front of the path): a nerfstudio-format ``transforms.json`` + image folder +
``semantics/`` mask folder -> cameras, file lists, scene box, and the dataparser transform / scale that the exporters
undo (``dataparser_transforms.json``).  Host code (json, file names, a handful of 4x4 matrices); no device work.

The pose normalisation is upstream arithmetic (nerfstudio 1.1.3 ``camera_utils.auto_orient_and_center_poses`` and
``rotation_matrix``), restated here for the methods the reference's configs and data use
(``"up"`` -- the config default, ``:58`` -- and ``"none"`` -- what the 3DCotton ``transforms.json`` overrides it to,
``fruit_nerf/utils/transforms.json``); ``"pca"`` / ``"vertical"`` and the ``"focus"`` centring raise.  Cameras are
undistorted pinholes (the 3DCotton captures have k1 = k2 = p1 = p2 = 0); a non-zero distortion raises."""

from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from ...rays import Cameras, SceneBox
from ..fruit_nerf import Semantics

MAX_AUTO_RESOLUTION = 1200  # :40


def rotation_matrix(a: Tensor, b: Tensor) -> Tensor:
    """nerfstudio ``camera_utils.rotation_matrix``: the rotation that takes direction ``a`` to direction ``b``."""
    a = a / torch.linalg.norm(a)
    b = b / torch.linalg.norm(b)
    v = torch.linalg.cross(a, b)
    c = torch.dot(a, b)
    if c < -1 + 1e-8:  # opposite vectors: perturb (upstream uses a random epsilon; a fixed one keeps this deterministic)
        return rotation_matrix(a + torch.tensor([0.003, -0.002, 0.001]), b)
    s = torch.linalg.norm(v)
    skew = torch.tensor([[0.0, -float(v[2]), float(v[1])], [float(v[2]), 0.0, -float(v[0])], [-float(v[1]), float(v[0]), 0.0]])
    return torch.eye(3) + skew + (skew @ skew) * ((1 - c) / (s ** 2 + 1e-8))


def auto_orient_and_center_poses(poses: Tensor, method: str = "up", center_method: str = "poses"):
    """nerfstudio ``camera_utils.auto_orient_and_center_poses`` for [N,4,4] poses: returns ([N,3,4], transform [3,4])."""
    origins = poses[..., :3, 3]
    mean_origin = torch.mean(origins, dim=0)
    if center_method == "poses":
        translation = mean_origin
    elif center_method == "none":
        translation = torch.zeros_like(mean_origin)
    else:
        raise NotImplementedError(f"center_method {center_method!r} (upstream focus-of-attention search) is not mirrored")
    if method == "up":
        up = torch.mean(poses[:, :3, 1], dim=0)
        up = up / torch.linalg.norm(up)
        rotation = rotation_matrix(up, torch.tensor([0.0, 0.0, 1.0]))
        transform = torch.cat([rotation, rotation @ -translation[..., None]], dim=-1)
    elif method == "none":
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
    else:
        raise NotImplementedError(f"orientation_method {method!r} is not mirrored (use 'up' or 'none')")
    return transform @ poses, transform


@dataclass
class DataparserOutputs:
    image_filenames: List[Path]
    cameras: Cameras
    scene_box: SceneBox
    dataparser_scale: float
    dataparser_transform: Tensor  # [3,4]
    metadata: Dict[str, Any] = field(default_factory=dict)

    def save_dataparser_transform(self, path) -> None:
        """``dataparser_transforms.json`` as nerfstudio writes it next to ``config.yml`` (read back by the exporters,
        ``scripts/exporter.py:100-101``)."""
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        with open(path, "w", encoding="UTF-8") as f:
            json.dump({"transform": self.dataparser_transform.tolist(), "scale": float(self.dataparser_scale)}, f, indent=4)


@dataclass
class CottonNerfDataParserConfig:
    """Fields as ``:43-66``."""

    data: Path = Path()
    scale_factor: float = 1.0
    downscale_factor: Optional[int] = None
    scene_scale: float = 1.0
    orientation_method: str = "up"
    center_method: str = "poses"
    auto_scale_poses: bool = True
    train_split_fraction: float = 0.95
    semantic_dir: Optional[str] = "semantics"
    semantic_img_ext: Optional[str] = "png"
    # False: <semantic_dir>/<image stem>.<ext> (this file).  True: every frame names its mask in "semantic_path", with
    # ``semantics_<k>`` down-scale folders -- the one difference of ``data/fruitnerf_dataparser.py`` (``:141-148``).
    semantics_from_frames: bool = False

    def setup(self) -> "CottonNerf":
        return CottonNerf(self)


class CottonNerf:
    def __init__(self, config: CottonNerfDataParserConfig):
        self.config = config
        self.config.data = Path(self.config.data)
        self.downscale_factor: Optional[int] = None

    def get_dataparser_outputs(self, split: str = "train") -> DataparserOutputs:
        return self._generate_dataparser_outputs(split)

    def _generate_dataparser_outputs(self, split="train") -> DataparserOutputs:
        cfg = self.config
        assert cfg.data.exists(), f"Data directory {cfg.data} does not exist."
        if cfg.data.suffix == ".json":
            meta, data_dir = json.loads(cfg.data.read_text()), cfg.data.parent
        else:
            meta, data_dir = json.loads((cfg.data / "transforms.json").read_text()), cfg.data
        fixed = {k: k in meta for k in ("fl_x", "fl_y", "cx", "cy", "h", "w")}
        distort_keys = ("k1", "k2", "k3", "k4", "p1", "p2")
        per_frame: Dict[str, list] = {k: [] for k in fixed}
        image_filenames, semantic_filenames, poses = [], [], []
        for frame in meta["frames"]:
            fname = self._get_fname(Path(str(frame["file_path"]).replace("\\", "/")), data_dir)
            for k, is_fixed in fixed.items():
                if not is_fixed:
                    assert k in frame, f"{k} not specified in frame"
                    per_frame[k].append(float(frame[k]))
            if any(float(frame.get(k, 0.0)) != 0.0 for k in distort_keys):
                raise NotImplementedError("lens distortion is not supported (undistort the images first)")
            image_filenames.append(fname)
            poses.append(np.array(frame["transform_matrix"]))
            if not cfg.semantics_from_frames:
                semantic_filenames.append(self._get_semantic_filepath(fname, data_dir))
            elif "semantic_path" in frame:
                semantic_filenames.append(self._get_fname(Path(str(frame["semantic_path"]).replace("\\", "/")), data_dir,
                                                          downsample_folder_prefix="semantics_"))
        assert len(semantic_filenames) == 0 or len(semantic_filenames) == len(image_filenames), \
            "Different number of image and semantic filenames."  # :147-152
        if any(float(meta.get(k, 0.0)) != 0.0 for k in distort_keys):
            raise NotImplementedError("lens distortion is not supported (undistort the images first)")
        if meta.get("camera_model", "OPENCV") not in ("OPENCV", "PINHOLE", "SIMPLE_PINHOLE"):
            raise NotImplementedError(f"camera model {meta['camera_model']!r}: only perspective cameras are supported")

        # train / eval split (:154-185)
        has_split_files_spec = any(f"{s}_filenames" in meta for s in ("train", "val", "test"))
        if f"{split}_filenames" in meta:
            split_filenames = set(self._get_fname(Path(x), data_dir) for x in meta[f"{split}_filenames"])
            unmatched = split_filenames.difference(image_filenames)
            if unmatched:
                raise RuntimeError(f"Some filenames for split {split} were not found: {unmatched}.")
            indices = np.array([i for i, p in enumerate(image_filenames) if p in split_filenames], dtype=np.int32)
        elif has_split_files_spec:
            raise RuntimeError(f"The dataset's list of filenames for split {split} is missing.")
        else:
            num_images = len(image_filenames)
            num_train_images = math.ceil(num_images * cfg.train_split_fraction)
            i_train = np.linspace(0, num_images - 1, num_train_images, dtype=int)
            i_eval = np.setdiff1d(np.arange(num_images), i_train)
            assert len(i_eval) == num_images - num_train_images
            if split == "train":
                indices = i_train
            elif split in ("val", "test"):
                indices = i_eval
            else:
                raise ValueError(f"Unknown dataparser split {split}")

        orientation_method = meta.get("orientation_override", cfg.orientation_method)
        poses_t = torch.from_numpy(np.array(poses).astype(np.float32))
        poses_t, transform_matrix = auto_orient_and_center_poses(poses_t, method=orientation_method,
                                                                 center_method=cfg.center_method)
        scale_factor = 1.0
        if cfg.auto_scale_poses:
            scale_factor /= float(torch.max(torch.abs(poses_t[:, :3, 3])))
        scale_factor *= cfg.scale_factor
        poses_t[:, :3, 3] *= scale_factor

        image_filenames = [image_filenames[i] for i in indices]
        semantic_filenames = [semantic_filenames[i] for i in indices] if len(semantic_filenames) > 0 else []
        idx = torch.tensor(np.asarray(indices), dtype=torch.long)
        poses_t = poses_t[idx]
        s = float(cfg.scene_scale)
        scene_box = SceneBox(torch.tensor([[-s, -s, -s], [s, s, s]], dtype=torch.float32))

        def intr(k: str) -> Tensor:
            if fixed[k]:
                return torch.full((len(indices),), float(meta[k]), dtype=torch.float32)
            return torch.tensor(per_frame[k], dtype=torch.float32)[idx]

        hs, ws = intr("h"), intr("w")
        if len(indices) and (bool((hs != hs[0]).any()) or bool((ws != ws[0]).any())):
            raise NotImplementedError("all images must have one size")
        assert self.downscale_factor is not None
        f = 1.0 / self.downscale_factor  # Cameras.rescale_output_resolution (floor rounding)
        height = int(math.floor(float(hs[0]) * f)) if len(indices) else 0
        width = int(math.floor(float(ws[0]) * f)) if len(indices) else 0
        cameras = Cameras(poses_t[:, :3, :4].contiguous(), intr("fl_x") * f, intr("fl_y") * f, intr("cx") * f,
                          intr("cy") * f, height, width)

        if "applied_transform" in meta:
            applied = torch.tensor(meta["applied_transform"], dtype=transform_matrix.dtype)
            transform_matrix = transform_matrix @ torch.cat(
                [applied, torch.tensor([[0, 0, 0, 1]], dtype=transform_matrix.dtype)], 0)
        if "applied_scale" in meta:
            scale_factor *= float(meta["applied_scale"])

        semantics = Semantics(filenames=semantic_filenames, classes=["apple", "stuff"],
                              colors=torch.tensor([0.0, 255.0]) / 255.0, mask_classes=["apple", "stuff"])  # :244-254
        return DataparserOutputs(image_filenames=image_filenames, cameras=cameras, scene_box=scene_box,
                                 dataparser_scale=scale_factor, dataparser_transform=transform_matrix,
                                 metadata={"semantics": semantics} if len(semantic_filenames) > 0 else {})

    def _get_semantic_filepath(self, img_filename, data_dir) -> Path:
        """``:300-305``: ``<semantic_dir>/<image stem>.<semantic_img_ext>`` (same down-scale folder rule)."""
        stem = os.path.splitext(os.path.split(str(img_filename))[1])[0]
        return self._get_fname(Path(os.path.join(self.config.semantic_dir, f"{stem}.{self.config.semantic_img_ext}")), data_dir)

    def _get_fname(self, filepath: Path, data_dir: Path, downsample_folder_prefix="images_") -> Path:
        """``:307-331``: the image path, in the ``images_<2^k>`` folder when a down-scale factor applies (chosen
        automatically on first use: halve while the longer side is >= 1200 px and the folder exists)."""
        filepath = Path(filepath)
        if self.downscale_factor is None:
            if self.config.downscale_factor is None:
                from PIL import Image

                h, w = Image.open(data_dir / filepath).size
                max_res, df = max(h, w), 0
                while True:
                    if (max_res / 2 ** df) < MAX_AUTO_RESOLUTION:
                        break
                    if not (data_dir / f"{downsample_folder_prefix}{2 ** (df + 1)}" / filepath.name).exists():
                        break
                    df += 1
                self.downscale_factor = 2 ** df
            else:
                self.downscale_factor = self.config.downscale_factor
        if self.downscale_factor > 1:
            return data_dir / f"{downsample_folder_prefix}{self.downscale_factor}" / filepath.name
        return data_dir / filepath
