"""``FruitDataManager`` -- the ray-source part of ``crop_nerf/fruit_nerf/data/fruit_datamanager.py`` on the HIP ray
generators: ``next_train`` (pixel sample -> ``train_ray_generator``, ``:188-197``), ``setup_inference`` /
``next_sample_volume`` (orthographic surface rays for the dense export, ``:157-172,199-204``) and the AABB corner /
surface-grid helpers (``:42-121``).  Images and masks are tensors resident in HBM ([N,H,W,3] / [N,H,W,1]): ``from_dataset``
decodes a ``FruitDataset`` (``cotton_dataset.py``) once and uploads it, where the reference's dataloader re-collates image
batches on the host (147 images of 1920 x 1440 are 4.9 GB of fp32, 1.7 % of one MI355X).  The reference's ``create_train_dataset`` / ``create_eval_dataset`` (``:174-186``)."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from ... import ops
from ...rays import Cameras, RayBundle
from ..components.ray_generators import OrthographicRayGenerator


@dataclass
class FruitDataManagerConfig:
    train_num_rays_per_batch: int = 4096
    eval_num_rays_per_batch: int = 4096
    camera_res_scale_factor: float = 1.0
    dataparser: Optional[object] = None  # a CottonNerfDataParserConfig / FruitNerfDataParserConfig (None: CottonNerf defaults)


def get_corners_of_aabb(aabb, device=None) -> Tensor:
    """``:42-69``: 8 corners, x fastest."""
    mn, mx = [float(v) for v in aabb[0]], [float(v) for v in aabb[1]]
    return torch.tensor([
        [mn[0], mn[1], mn[2]], [mx[0], mn[1], mn[2]], [mn[0], mx[1], mn[2]], [mx[0], mx[1], mn[2]],
        [mn[0], mn[1], mx[2]], [mx[0], mn[1], mx[2]], [mn[0], mx[1], mx[2]], [mx[0], mx[1], mx[2]],
    ], dtype=torch.float32)


def sample_surface_points(corners: Tensor, n: int, device, noise: bool = False) -> Tuple[Tensor, Tensor]:
    """``:71-121``: n x n grid on the face through corners 0,1,2; the constant coordinate goes to column 2
    (reference quirk); plane vector along that column.  Grid parameters on the host, points by ``cn_surface_grid``."""
    c1, c2, c3 = corners[0], corners[1], corners[2]
    ext = torch.abs(corners.max(dim=0).values - corners.min(dim=0).values)
    const = int(torch.argmax(torch.logical_and(c1 == c2, c2 == c3).to(torch.int64)))
    ax = int(torch.argmax(torch.abs(c1 - c2)))
    ay = int(torch.argmax(torch.abs(c1 - c3)))
    nx = int(ext[0] / ext[const] * n)
    ny = int(ext[1] / ext[const] * n)
    pts = ops.surface_grid(float(c1[ax]), float(c2[ax]), nx, float(c1[ay]), float(c3[ay]), ny, float(c3[const]), device)
    c4 = corners[-1]
    plane = torch.tensor([[0.0, 0.0, float(torch.sign(c4[const]) * torch.abs(c1[const]) + torch.abs(c4[const]))]],
                         dtype=torch.float32)
    return pts, plane


class CameraDataset:
    """What the exporters read from a nerfstudio ``InputDataset`` (``exporter_utils_nerfacto.py:308-309``,
    ``fruit_nerf.py:263-264``): the split's cameras, its image file names and the metadata.  Stands in for the training /
    eval dataset of a run that is opened for export only (no images decoded)."""

    def __init__(self, cameras: Cameras, image_filenames=None, metadata: Optional[dict] = None):
        names = list(image_filenames) if image_filenames is not None else [f"frame_{i:05d}" for i in range(len(cameras))]
        if len(names) != len(cameras):
            raise ValueError(f"{len(names)} image file names for {len(cameras)} cameras")
        self.cameras = cameras
        self.image_filenames = names
        self.metadata = metadata if metadata is not None else {}

    def __len__(self) -> int:
        return len(self.image_filenames)


class FruitDataManager:
    SORT_BATCHES_FROM = 16384  # rays: smaller batches keep their draw order (see _sample)

    def __init__(self, config: FruitDataManagerConfig, cameras: Cameras, device="cuda", test_mode: str = "val",
                 world_size: int = 1, local_rank: int = 0, images: Optional[Tensor] = None,
                 fruit_masks: Optional[Tensor] = None, seed: int = 0, **kwargs):
        self.config = config
        self.device = torch.device(device)
        self.cameras = cameras.to(self.device)
        self.test_mode = test_mode
        self.world_size = world_size
        self.local_rank = local_rank
        self.images = images  # [N,H,W,3] float (optional)
        self.fruit_masks = fruit_masks  # [N,H,W,1] float (optional)
        self.train_count = 0
        self.eval_count = 0
        # the datasets behind the rays (``:174-186``): what ``collect_camera_poses`` walks.  ``from_dataset`` / ``eval_setup``
        # replace them with the capture's own (file names, eval split)
        self.train_dataset = CameraDataset(self.cameras)
        self.eval_dataset: Optional[CameraDataset] = None
        self._gen = torch.Generator(device="cpu").manual_seed(seed + 1000 * local_rank)
        self._device_generator: Optional[torch.Generator] = None
        self._seed = seed + 1000 * local_rank
        self.orthographic_ray_generator: Optional[OrthographicRayGenerator] = None
        self._idx_ring: Dict[int, dict] = {}  # pinned staging of next_train / next_eval's pixel draws, per batch shape
        self._dims_dev: Optional[Tensor] = None
        self.sort_batches = True  # batches of SORT_BATCHES_FROM rays and more are handed out sorted by (camera, pixel Morton): see _sample

    @classmethod
    def from_dataset(cls, config: FruitDataManagerConfig, dataset, device="cuda", **kwargs) -> "FruitDataManager":
        """``create_train_dataset`` (``:174-179``) + the image cache: every image and fruit mask of ``dataset`` (a
        ``FruitDataset``) decoded once and kept on the device as fp32 (values are the reference's float16-rounded ones)."""
        n = len(dataset)
        cams = dataset.cameras
        images = torch.empty(n, cams.height, cams.width, 3, device=device)
        masks = torch.empty(n, cams.height, cams.width, 1, device=device)
        for i in range(n):
            d = dataset.get_data(i)
            if tuple(d["image"].shape[:2]) != (cams.height, cams.width):
                raise ValueError(f"image {i} is {tuple(d['image'].shape[:2])}, the cameras say {(cams.height, cams.width)}")
            assert d["fruit_mask"].shape[:2] == d["image"].shape[:2], "Mask and image have different shapes."
            images[i] = d["image"].to(device=device, dtype=torch.float32)
            masks[i] = d["fruit_mask"].to(device=device, dtype=torch.float32)
        dm = cls(config, cams, device=device, images=images, fruit_masks=masks, **kwargs)
        dm.train_dataset = dataset
        return dm

    # PixelSampler.sample + train_ray_generator (:188-197)
    def _uniforms_to_device(self, num_rays: int) -> Tensor:
        """[num_rays, 3] uniforms of the data manager's host stream, on the device: drawn straight into a pinned staging slot
        (a ring of four, each guarded by an event: the host may run four batches ahead of the GPU) and sent with ONE
        asynchronous copy.  Indexing resident images with host index tensors made three synchronous pageable copies per image
        tensor -- at 4 096 rays per batch the host side of ``next_train`` was longer than half a training iteration."""
        if self.device.type != "cuda":
            return torch.rand(num_rays, 3, generator=self._gen).to(self.device)
        ring = self._idx_ring.get(num_rays)
        if ring is None:
            ring = self._idx_ring[num_rays] = {"slots": [torch.empty(num_rays, 3).pin_memory() for _ in range(4)],
                                               "events": [None] * 4, "next": 0}
        k = ring["next"] % 4
        ring["next"] += 1
        if ring["events"][k] is not None:
            ring["events"][k].synchronize()
        torch.rand(num_rays, 3, generator=self._gen, out=ring["slots"][k])
        dev = ring["slots"][k].to(self.device, non_blocking=True)
        ring["events"][k] = torch.cuda.Event()
        ring["events"][k].record()
        return dev

    def _sample(self, num_rays: int) -> Tuple[RayBundle, Dict]:
        """nerfstudio's ``PixelSampler.sample_method``: ``floor(rand(B, 3) * (images, height, width))``.  The uniforms are drawn
        on the host (the data manager's seeded stream), staged through the pinned ring and turned into indices ON THE DEVICE:
        at 65 536 rays three ``torch.randint`` calls on a CPU generator took 13-48 ms per batch -- their OpenMP regions wake
        the worker threads for 65 536 elements each time -- which is longer than the iteration they feed (15.8 ms); the
        arithmetic on 196 608 floats is two small launches instead."""
        n, h, w = len(self.cameras), self.cameras.height, self.cameras.width
        u_d = self._uniforms_to_device(num_rays)
        if u_d.is_cuda:
            dims = self._dims_dev
            if dims is None or dims.device != u_d.device:
                dims = self._dims_dev = torch.tensor([n, h, w], dtype=torch.float32, device=u_d.device)
            idx_d = torch.floor(u_d * dims).to(torch.int64)
        else:
            idx_d = torch.floor(u_d * torch.tensor([n, h, w], dtype=torch.float32)).to(torch.int64)
        if self.sort_batches and u_d.is_cuda and idx_d.shape[0] >= self.SORT_BATCHES_FROM:
            # The ORDER of a batch's rays means nothing to the losses (means over the batch).  Sorted by camera and, inside a
            # camera, along the pixel's Morton curve, neighbouring rays share table lines: the forward field pass of a 65 536-ray
            # batch 1.77 -> 1.28 ms (tools/sorted_ray_probe.py), the backward's recompute gathers likewise.  With the backward
            # kernels' OLD tile order (workgroup w took tiles w, w + 256, ...) this lost far more than it gained -- neighbouring
            # workgroups added to the same lines at the same moment and the memory side serialised them: 15.6 -> 23.0 ms; since
            # a workgroup takes one contiguous run of tiles (train_field_mfma.hpp) the iteration gains: 15.6 -> 14.9 ms at
            # 65 536 rays.  At 4 096 rays (41 rays to a camera) the argsort costs what the order gains: batches below
            # SORT_BATCHES_FROM rays keep their draw order.
            idx_d = idx_d[ops.ray_sort_permutation(idx_d, h, w)]
        batch: Dict[str, Tensor] = {"indices": idx_d}
        if self.images is not None:
            ii = idx_d.to(self.images.device)
            batch["image"] = self.images[ii[:, 0], ii[:, 1], ii[:, 2]]
        if self.fruit_masks is not None:
            ii = idx_d.to(self.fruit_masks.device)
            batch["fruit_mask"] = self.fruit_masks[ii[:, 0], ii[:, 1], ii[:, 2]]
        return self.cameras.generate_rays(idx_d), batch

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        self.train_count += 1
        return self._sample(self.config.train_num_rays_per_batch)

    @property
    def device_generator(self) -> torch.Generator:
        """The device-side random stream of ``next_train_device`` (seeded per rank; a HIP graph that captures the call must
        register it: ``graph.register_generator_state``)."""
        if self._device_generator is None:
            self._device_generator = torch.Generator(device=self.device)
            self._device_generator.manual_seed(self._seed)
        return self._device_generator

    def next_train_device(self, step: int) -> Tuple[RayBundle, Dict]:
        """``next_train`` with the pixel indices drawn on the device (torch's device generator, as nerfstudio's
        ``PixelSampler`` does): no host-to-device copy, so the call can be captured in a HIP graph.  Ray bundle only --
        the exporters do not read the pixels."""
        self.train_count += 1
        n, h, w = len(self.cameras), self.cameras.height, self.cameras.width
        num_rays, dev = self.config.train_num_rays_per_batch, self.device
        g = self.device_generator
        idx = torch.stack([torch.randint(0, n, (num_rays,), device=dev, generator=g),
                           torch.randint(0, h, (num_rays,), device=dev, generator=g),
                           torch.randint(0, w, (num_rays,), device=dev, generator=g)], dim=-1)
        if self.sort_batches and num_rays >= self.SORT_BATCHES_FROM:
            idx = idx[ops.ray_sort_permutation(idx, h, w)]
        return self.cameras.generate_rays(idx), {"indices": idx}

    @property
    def export_seed(self) -> int:
        """Seed of the point-cloud exporter's counter-based pixel stream (``cn_pixel_sample``), one stream per rank."""
        return self._seed

    def next_eval(self, step: int) -> Tuple[RayBundle, Dict]:
        self.eval_count += 1
        return self._sample(self.config.eval_num_rays_per_batch)

    def setup_inference(self, aabb, num_points) -> int:
        """``:157-172``: surface grid on the bottom face + orthographic generator; returns the ray count."""
        corners = get_corners_of_aabb(aabb)
        pts, plane = sample_surface_points(corners, n=num_points, device=self.device, noise=False)
        self.orthographic_ray_generator = OrthographicRayGenerator(
            surface_points=pts, plane_normal=plane, ray_batch_size=self.config.eval_num_rays_per_batch,
            device=self.device, aabb=aabb)
        return pts.shape[0]

    def next_sample_volume(self, step: int) -> Tuple[RayBundle, None]:
        """``:199-204``."""
        self.train_count += 1
        return self.orthographic_ray_generator(count=self.train_count), None
