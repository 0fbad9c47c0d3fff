"""``FruitDataset`` -- mirror of ``crop_nerf/fruit_nerf/data/cotton_dataset.py:34-151``: images as float16-rounded
[H,W,3] values in [0,1] and the binarised fruit mask of each image (``fruit_mask``).  Host code (PIL + numpy; the
reference's two cv2 calls are restated: ``cvtColor(RGB2GRAY)`` in its 14-bit fixed point, ``threshold(3, 255, BINARY)``)."""

from __future__ import annotations

from pathlib import Path
from typing import Dict

import numpy as np
import torch
from PIL import Image
from torch import Tensor

from ..fruit_nerf import Semantics
from .cotton_nerf_dataparser import DataparserOutputs


def get_object_semantics(pil_img) -> Tensor:
    """``:34-39``: grey level > 3 -> 255 (float16)."""
    arr = np.array(pil_img)
    if pil_img.mode == "RGB":  # cv2.cvtColor(..., COLOR_RGB2GRAY): (4899 R + 9617 G + 1868 B + 2^13) >> 14
        a = arr.astype(np.int64)
        arr = ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)
    semantics = np.where(arr > 3, 255, 0).astype(arr.dtype)
    return torch.from_numpy(np.float16(semantics))


def get_semantics_and_mask_tensors_from_path(filepath: Path, mask_indices=None, scale_factor: float = 1.0) -> Tensor:
    """``:60-80``: the mask image -> {0, 1} float16 [H,W]; an all-background mask raises like the reference."""
    pil_image = Image.open(filepath)
    if scale_factor != 1.0:
        width, height = pil_image.size
        pil_image = pil_image.resize((int(width * scale_factor), int(height * scale_factor)), resample=Image.NEAREST)
    semantics = get_object_semantics(pil_image)
    if semantics.max() > 1.0:
        semantics = semantics / 255
    else:
        raise ValueError("Please look at mask file manually! How to normalize")
    return semantics


class FruitDataset:
    def __init__(self, dataparser_outputs: DataparserOutputs, scale_factor: float = 1.0):
        self._dataparser_outputs = dataparser_outputs
        self.scale_factor = scale_factor
        self.cameras = dataparser_outputs.cameras
        self.scene_box = dataparser_outputs.scene_box
        self.metadata = dataparser_outputs.metadata
        assert "semantics" in self.metadata.keys() and isinstance(self.metadata["semantics"], Semantics), \
            "No semantic instance could be found! Is a semantic folder included in the input folder and transform.json file?"
        self.semantics = self.metadata["semantics"]
        self.mask_indices = torch.tensor(
            [self.semantics.classes.index(c) for c in self.semantics.mask_classes]).view(1, 1, -1)

    def __len__(self) -> int:
        return len(self._dataparser_outputs.image_filenames)

    @property
    def image_filenames(self):
        """nerfstudio ``InputDataset.image_filenames`` (read by ``collect_camera_poses_for_dataset``)."""
        return self._dataparser_outputs.image_filenames

    def get_numpy_image(self, image_idx: int) -> np.ndarray:
        """nerfstudio ``InputDataset.get_numpy_image``: uint8 [H,W,3|4]."""
        pil_image = Image.open(self._dataparser_outputs.image_filenames[image_idx])
        if self.scale_factor != 1.0:
            width, height = pil_image.size
            pil_image = pil_image.resize((int(width * self.scale_factor), int(height * self.scale_factor)),
                                         resample=Image.BILINEAR)
        image = np.array(pil_image, dtype="uint8")
        if image.ndim == 2:
            image = image[:, :, None].repeat(3, axis=2)
        assert image.ndim == 3 and image.shape[2] in (3, 4), f"Image shape of {image.shape} is incorrect."
        return image

    def get_image_float16(self, image_idx: int) -> Tensor:
        """``:109-121``: uint8 -> float16 / 255 (alpha, if any, dropped: the reference composites onto ``alpha_color``
        only when one is configured, and its dataparser never sets one)."""
        image = torch.from_numpy(self.get_numpy_image(image_idx).astype("float16") / 255.0)
        return image[:, :, :3]

    def get_metadata(self, data: Dict) -> Dict:
        """``:100-107``."""
        filepath = self.semantics.filenames[data["image_idx"]]
        label = get_semantics_and_mask_tensors_from_path(filepath, self.mask_indices, self.scale_factor)
        return {"fruit_mask": label[..., None]}

    def get_data(self, image_idx: int, image_type: str = "float16") -> Dict:
        """``:123-151``."""
        if image_type != "float16":
            raise NotImplementedError(f"image_type (={image_type}) getter was not implemented, use float16")
        data = {"image_idx": image_idx, "image": self.get_image_float16(image_idx)}
        data.update(self.get_metadata(data))
        return data

    def __getitem__(self, image_idx: int) -> Dict:
        return self.get_data(image_idx)
