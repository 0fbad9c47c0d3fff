"""Batched semantic projection -- ``FruitModel.get_outputs_for_projections`` (``crop_nerf/fruit_nerf/fruit_nerf.py:254-318``,
driven by ``scripts/semantic_projection.py:132-170``) as the reference RUNS it: #super-clusters x #cameras x k sub-cluster
boxes, two PNGs per job.

The reference pays, per job, for a whole frame of rays, six boolean-mask indexings (each a host round trip), two full-frame
float images and two synchronous PNG encodes -- for a box that covers a few thousand of the frame's pixels.  Here

* the host projects the 8 corners of every box into its camera and keeps the bounding screen rectangle (``box_screen_rects``;
  a box with a corner behind the camera keeps the whole frame), so only the pixels that can hit are ever turned into rays;
* the jobs of many (camera, box) pairs form ONE batch: one slab-test launch, one list of the hitting pixels (the batch's only
  host synchronisation), one jagged ray bundle, one sampler + render, one density-only occlusion pass, one scatter into
  per-slot values (``cn_projection_*``, ``csrc/projection.hip``).  The arithmetic per ray is the per-job path's, bit for bit;
* results stay compact (one byte per rectangle pixel): the PNG tree is written by worker threads from pinned copies while
  the GPU works on the next batch (``PngWriter``; same file tree, same pixel values), and ``ProjectionRun.images_u8`` hands
  whole frames to the in-process merger (``segmentation/merger.py: process_super_cluster``) without any file.
"""

from __future__ import annotations

import ctypes as C
import os
import shutil
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _lib as L
from .. import ops
from ..rays import Cameras, RayBundle

JOB_DTYPE = np.dtype(L.ProjectionJob)
MIN_VALID_RAYS = 10  # fruit_nerf.py:293


# ---------------------------------------------------------------------------------------------------------------------
# host geometry: which pixels can see a box
# ---------------------------------------------------------------------------------------------------------------------

def box_screen_rects(c2w: np.ndarray, fx: float, fy: float, cx: float, cy: float, height: int, width: int,
                     aabbs: np.ndarray, margin: int = 2) -> np.ndarray:
    """[J,2,3] boxes seen by one pinhole camera (``c2w`` 3x4, OpenGL frame: -z forward, +y up, pixel centres at +0.5 as
    ``Cameras.generate_rays`` has them) -> [J,4] int32 (x0, y0, w, h): a rectangle, clipped to the frame, that contains every
    pixel whose centre ray can hit the box -- the bounding box of the 8 projected corners, ``margin`` pixels wider on every
    side (the rays are float32, this is float64).  A box that straddles the camera plane gets the whole frame; a box that
    projects outside the frame, or lies wholly behind the camera, gets an empty rectangle (w and h are 0 together)."""
    aabbs = np.asarray(aabbs, dtype=np.float64).reshape(-1, 2, 3)
    J = aabbs.shape[0]
    m = np.asarray(c2w, dtype=np.float64).reshape(3, 4)
    sel = np.array([[(i >> 2) & 1, (i >> 1) & 1, i & 1] for i in range(8)])
    corners = aabbs[:, sel, np.arange(3)]  # [J,8,3]
    try:
        rinv = np.linalg.inv(m[:, :3])
    except np.linalg.LinAlgError:
        return np.tile(np.array([[0, 0, width, height]], np.int32), (J, 1))
    v = (corners - m[:, 3]) @ rinv.T  # camera-frame coordinates
    depth = -v[..., 2]
    scale = max(1.0, float(np.abs(corners - m[:, 3]).max(initial=0.0)))
    front = (depth > 1e-6 * scale).all(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        px = cx + fx * v[..., 0] / depth
        py = cy - fy * v[..., 1] / depth
    rects = np.empty((J, 4), np.int32)
    x0 = np.floor(np.where(front, px.min(axis=1), 0.0) - 0.5) - margin
    x1 = np.ceil(np.where(front, px.max(axis=1), 0.0) - 0.5) + margin  # inclusive
    y0 = np.floor(np.where(front, py.min(axis=1), 0.0) - 0.5) - margin
    y1 = np.ceil(np.where(front, py.max(axis=1), 0.0) - 0.5) + margin
    x0, x1 = np.clip(x0, 0, width), np.clip(x1 + 1, 0, width)
    y0, y1 = np.clip(y0, 0, height), np.clip(y1 + 1, 0, height)
    w, h = np.maximum(x1 - x0, 0), np.maximum(y1 - y0, 0)
    empty = (w == 0) | (h == 0)
    rects[:, 0], rects[:, 1] = np.where(empty, 0, x0), np.where(empty, 0, y0)
    rects[:, 2], rects[:, 3] = np.where(empty, 0, w), np.where(empty, 0, h)
    rects[~front] = (0, 0, width, height)
    # every corner behind the camera plane: the (convex) box is behind it, t > 0 never reaches it -- no pixel can hit
    rects[(depth < -1e-6 * scale).all(axis=1)] = (0, 0, 0, 0)
    return rects


# ---------------------------------------------------------------------------------------------------------------------
# PNG files off the critical path
# ---------------------------------------------------------------------------------------------------------------------

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_zero_bands: Dict[Tuple[int, int], bytes] = {}


def _deflate_piece(data, level: int = 1) -> bytes:
    """Raw deflate of ``data`` ended by a full flush: byte-aligned and history-free, so pieces concatenate."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush(zlib.Z_FULL_FLUSH)


def _zero_band(row_bytes: int, rows: int) -> bytes:
    """``rows`` all-zero rows as cached pieces of 2^k rows, largest first (log2(height) distinct pieces per row length)."""
    out = []
    for k in range(30, -1, -1):
        n = 1 << k
        if rows & n:
            piece = _zero_bands.get((row_bytes, n))
            if piece is None:
                piece = _zero_bands[(row_bytes, n)] = _deflate_piece(bytes(row_bytes * n))
            out.append(piece)
    return b"".join(out)


def _adler32_zeros(adler: int, n: int) -> int:
    a, b = adler & 0xFFFF, adler >> 16
    return (((b + n * a) % 65521) << 16) | a


def _png_chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png_gray_rect(crop: np.ndarray, x0: int, y0: int, height: int, width: int) -> bytes:
    """An 8-bit RGB PNG of a ``height`` x ``width`` frame that is black except for the gray rectangle ``crop`` [h,w] uint8 at
    (x0, y0), all three channels equal -- the file ``torchvision.utils.save_image`` writes for such an image
    (``fruit_nerf.py:304,315``) up to the compressed byte stream, which decodes to the same pixels.  The rows above and
    below the rectangle are all zero: their deflate streams are assembled from cached pieces of 2^k rows and spliced in (pieces
    end with a full flush), so a file costs the rectangle's rows, not the frame's (0.6 ms instead of 9 ms at 800 x 800)."""
    h, w = (int(crop.shape[0]), int(crop.shape[1])) if crop.size else (0, 0)
    if h == 0:
        x0 = y0 = 0  # an empty rectangle: one band of `height` zero rows
    row_bytes = 1 + 3 * width  # filter type 0 + RGB
    below = height - y0 - h
    mid = np.zeros((h, row_bytes), np.uint8)
    if h:
        mid[:, 1 + 3 * x0:1 + 3 * (x0 + w)].reshape(h, w, 3)[:] = crop[:, :, None]
    adler = _adler32_zeros(1, row_bytes * y0)
    adler = zlib.adler32(mid, adler)
    adler = _adler32_zeros(adler, row_bytes * below)
    parts = [b"\x78\x01"]
    if y0:
        parts.append(_zero_band(row_bytes, y0))
    if h:
        parts.append(_deflate_piece(mid))
    if below:
        parts.append(_zero_band(row_bytes, below))
    parts.append(b"\x01\x00\x00\xff\xff")  # final, empty stored block
    parts.append(struct.pack(">I", adler))
    return (_PNG_MAGIC + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 2, 0, 0, 0))
            + _png_chunk(b"IDAT", b"".join(parts)) + _png_chunk(b"IEND", b""))


def _worker_count() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


class PngWriter:
    """Worker threads that turn rectangle crops into the reference's PNG files through ``cn_png_write_gray_rects`` (native:
    the GIL is released for the whole call, so the threads really run side by side; the Python encoder above is the readable
    statement of the same stream and the test's cross-check).  ``submit_rects`` never blocks on the GPU: it is handed host
    arrays.  ``close`` waits for every file and re-raises the first failure."""

    FILES_PER_TASK = 32

    def __init__(self, workers: Optional[int] = None):
        self.pool = ThreadPoolExecutor(max_workers=workers or _worker_count(), thread_name_prefix="cn-png")
        self.futures: List = []
        self.files = 0
        self._lib = L.load()

    def _write(self, paths: List[str], values: np.ndarray, offsets: np.ndarray, rects: np.ndarray, height: int, width: int) -> None:
        n = len(paths)
        arr = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        offsets = np.ascontiguousarray(offsets, np.int64)
        rects = np.ascontiguousarray(rects, np.int32)
        L.check(self._lib.cn_png_write_gray_rects(n, arr, C.c_void_p(values.ctypes.data), C.c_void_p(offsets.ctypes.data),
                                                  C.c_void_p(rects.ctypes.data), height, width, 1))

    def _copy(self, src: str, dst_dir: str) -> None:
        os.makedirs(dst_dir, exist_ok=True)
        shutil.copy(src, dst_dir)

    def submit_rects(self, paths: Sequence[str], values: np.ndarray, offsets: np.ndarray, rects: np.ndarray, height: int,
                     width: int) -> None:
        """One file per path: the frame that is black but for ``rects[i]`` = (x0, y0, w, h), filled row-major from the uint8
        array ``values`` starting at ``offsets[i]`` (``values`` must stay alive and unchanged until ``close``)."""
        if values.dtype != np.uint8 or not values.flags["C_CONTIGUOUS"]:
            raise TypeError("values: a contiguous uint8 array")
        for lo in range(0, len(paths), self.FILES_PER_TASK):
            hi = lo + self.FILES_PER_TASK
            self.futures.append(self.pool.submit(self._write, list(paths[lo:hi]), values, offsets[lo:hi], rects[lo:hi],
                                                 height, width))
        self.files += len(paths)

    def submit(self, path: str, crop: np.ndarray, x0: int, y0: int, height: int, width: int) -> None:
        crop = np.ascontiguousarray(crop, np.uint8)
        h, w = (crop.shape if crop.size else (0, 0))
        self.submit_rects([path], crop.reshape(-1), np.zeros(1, np.int64), np.array([[x0, y0, w, h]], np.int32), height, width)

    def submit_copy(self, src: str, dst_dir: str) -> None:
        self.futures.append(self.pool.submit(self._copy, src, dst_dir))

    def close(self) -> None:
        try:
            for f in self.futures:
                f.result()
        finally:
            self.pool.shutdown(wait=True)
            self.futures = []


# ---------------------------------------------------------------------------------------------------------------------
# batches
# ---------------------------------------------------------------------------------------------------------------------

@dataclass
class ProjectionBatch:
    """Compact results of one batch: per job its key (i_sc, cam_idx, i) and screen rectangle, per rectangle pixel ("slot") one
    byte (or float) of ``wo_occ`` and ``visible``."""

    keys: List[Tuple[int, int, int]]
    table: np.ndarray  # [J] JOB_DTYPE, host
    height: int
    width: int
    num_slots: int
    num_rays: int
    device: Optional[torch.device] = None
    jobs_dev: Optional[Tensor] = None
    job_of_slot: Optional[Tensor] = None
    hit_count: Optional[Tensor] = None
    wo_occ_u8: Optional[Tensor] = None
    visible_u8: Optional[Tensor] = None
    wo_occ_f32: Optional[Tensor] = None
    visible_f32: Optional[Tensor] = None

    def float_images(self, j: int) -> Tuple[Tensor, Tensor]:
        """(wo_occ, visible) of job ``j`` as the per-job path returns them: float [H,W,3] on the device."""
        if self.num_slots == 0:  # nothing of this batch is inside any frame: black images (fruit_nerf.py:293-297)
            z = torch.zeros(self.height, self.width, 3, device=self.device)
            return z, z.clone()
        if self.wo_occ_f32 is None:
            raise RuntimeError("this batch was run without float outputs")
        t = self.table[j]
        x0, y0, w, h, off = int(t["x0"]), int(t["y0"]), int(t["w"]), int(t["h"]), int(t["slot_offset"])
        out = []
        for vals in (self.wo_occ_f32, self.visible_f32):
            img = torch.zeros(self.height, self.width, device=vals.device)
            if w and h:
                img[y0:y0 + h, x0:x0 + w] = vals[off:off + w * h].view(h, w)
            out.append(img[..., None].repeat(1, 1, 3))
        return out[0], out[1]


@dataclass
class ProjectionRun:
    """All batches of one ``get_outputs_for_projections`` call (compact: a byte per rectangle pixel and image kind)."""

    batches: List[ProjectionBatch] = field(default_factory=list)
    num_cameras: int = 0
    height: int = 0
    width: int = 0
    stats: Dict[str, float] = field(default_factory=dict)

    def __iter__(self) -> Iterator[ProjectionBatch]:
        return iter(self.batches)

    def float_results(self) -> Dict[Tuple[int, int, int], Tuple[Tensor, Tensor]]:
        return {key: b.float_images(j) for b in self.batches for j, key in enumerate(b.keys)}

    def images_u8(self, i_sc: int, num_sub_clusters: int) -> Tuple[Tensor, Tensor]:
        """(wo_occ, visible) [n_cams, k, H, W] uint8 on the device for one super-cluster: what
        ``segmentation.merger.process_super_cluster`` takes (the PNG round trip's pixel values, no file involved)."""
        dev = self.batches[0].device
        k = int(num_sub_clusters)
        wo = torch.zeros(self.num_cameras * k, self.height, self.width, dtype=torch.uint8, device=dev)
        vis = torch.zeros_like(wo)
        for b in self.batches:
            if b.num_slots == 0:
                continue
            idx = np.array([cam * k + i if sc == i_sc else -1 for sc, cam, i in b.keys], np.int32)
            if (idx < 0).all():
                continue
            image_of_job = torch.from_numpy(idx).to(dev)
            ops.projection_paste(b.jobs_dev, b.job_of_slot, b.wo_occ_u8, image_of_job, wo)
            ops.projection_paste(b.jobs_dev, b.job_of_slot, b.visible_u8, image_of_job, vis)
        shape = (self.num_cameras, k, self.height, self.width)
        return wo.view(shape), vis.view(shape)


def _camera_host(cameras: Cameras):
    c2w = cameras.camera_to_worlds.detach().to("cpu", torch.float32).numpy().reshape(-1, 12)
    intr = cameras.intrinsics().detach().cpu().numpy()
    return c2w, intr


def plan_jobs(cameras: Cameras, pcd_data, rank: int = 0, world_size: int = 1, compat_cam0: bool = False
              ) -> Tuple[List[Tuple[int, int, int]], np.ndarray]:
    """Every (super-cluster, camera, sub-cluster) job of ``fruit_nerf.py:267-281`` that belongs to this rank (jobs are numbered
    in the reference's loop order and dealt round-robin), camera-major, with its screen rectangle.  Returns (keys, table)."""
    c2w, intr = _camera_host(cameras)
    H, W = int(cameras.height), int(cameras.width)
    boxes = [np.asarray(pcd_data[i]["aabb"], dtype=np.float32).reshape(-1, 2, 3) for i in range(len(pcd_data))]
    n_cams = len(cameras)
    if not boxes or n_cams == 0:
        return [], np.zeros(0, JOB_DTYPE)
    ks = np.array([b.shape[0] for b in boxes], np.int64)
    all_boxes = np.concatenate(boxes)  # [B,2,3]
    sc_of = np.repeat(np.arange(len(boxes)), ks)
    i_of = np.concatenate([np.arange(k) for k in ks])
    # job number in the reference's order: super-cluster-major, then camera, then sub-cluster
    first = np.concatenate([[0], np.cumsum(ks * n_cams)])[:-1]
    number = first[sc_of][None, :] + np.arange(n_cams)[:, None] * ks[sc_of][None, :] + i_of[None, :]  # [C,B]
    mine = (number % world_size) == rank
    rects = np.stack([box_screen_rects(c2w[cam], *[float(v) for v in intr[cam]], H, W, all_boxes) for cam in range(n_cams)])
    cam_idx, box_idx = np.nonzero(mine)  # camera-major
    t = np.zeros(len(cam_idx), JOB_DTYPE)
    t["c2w"] = c2w[cam_idx]
    t["fx"], t["fy"], t["cx"], t["cy"] = intr[cam_idx].T
    t["aabb"] = all_boxes[box_idx].reshape(-1, 6)
    t["x0"], t["y0"], t["w"], t["h"] = rects[cam_idx, box_idx].T
    t["camera_index"] = 0 if compat_cam0 else cam_idx
    keys = list(zip(sc_of[box_idx].tolist(), cam_idx.tolist(), i_of[box_idx].tolist()))
    return keys, t


def run_batch(model, keys: Sequence[Tuple[int, int, int]], table: np.ndarray, height: int, width: int,
              want_float: bool = False, want_u8: bool = True) -> ProjectionBatch:
    """One batch through the kernels.  ``table``'s ``slot_offset`` column is filled here."""
    dev = model.device
    table = table.copy()
    sizes = table["w"].astype(np.int64) * table["h"].astype(np.int64)
    table["slot_offset"] = np.concatenate([[0], np.cumsum(sizes)[:-1]]) if len(table) else 0
    P = int(sizes.sum())
    batch = ProjectionBatch(list(keys), table, height, width, P, 0, dev)
    if P == 0:
        return batch
    raw = torch.from_numpy(table.view(np.uint8).reshape(-1))
    batch.jobs_dev = raw.pin_memory().to(dev, non_blocking=True)
    t = ops.projection_test(batch.jobs_dev, P, width, MIN_VALID_RAYS)
    batch.job_of_slot, batch.hit_count = t["job_of_slot"], t["hit_count"]
    hit_slots = t["flags"].nonzero(as_tuple=False).squeeze(1)  # the batch's one synchronisation
    N = batch.num_rays = int(hit_slots.numel())
    if N:
        g = ops.projection_gather(batch.jobs_dev, batch.job_of_slot, hit_slots, width)
        sub = RayBundle(g["origins"], g["directions"], None, g["camera_indices"], g["nears"], g["fars"])
        sem = model.get_outputs_for_camera_jagged_ray_bundle(sub, keys=("semantics",))["semantics"]  # fruit_nerf.py:301
        occ = RayBundle(g["origins"], g["directions"], None, g["camera_indices"], torch.zeros_like(g["nears"]), g["nears"])
        weight = model.get_density_for_camera_ray_bundle(occ)  # :305-310
    else:
        sem = torch.empty(0, device=dev)
        weight = torch.empty(0, device=dev)
    s = ops.projection_scatter(sem, weight, hit_slots, P, 0.5, want_float=want_float, want_u8=want_u8)
    batch.wo_occ_u8, batch.visible_u8 = s.get("wo_occ_u8"), s.get("visible_u8")
    batch.wo_occ_f32, batch.visible_f32 = s.get("wo_occ_f32"), s.get("visible_f32")
    return batch


class _PngStage:
    """Device results of a batch -> pinned host copy (asynchronous) -> PNG tasks, one batch behind the GPU."""

    def __init__(self, writer: PngWriter, output_root: str, height: int, width: int):
        self.writer, self.root, self.H, self.W = writer, output_root, height, width
        self.pending: List[Tuple[ProjectionBatch, Tensor, torch.cuda.Event]] = []

    def push(self, batch: ProjectionBatch) -> None:
        self.flush(wait=False)
        if batch.num_slots == 0:
            self._emit(batch, None)
            return
        host = torch.empty(2, batch.num_slots, dtype=torch.uint8, pin_memory=True)
        host[0].copy_(batch.wo_occ_u8, non_blocking=True)
        host[1].copy_(batch.visible_u8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((batch, host, ev))

    def flush(self, wait: bool) -> None:
        keep = []
        for batch, host, ev in self.pending:
            if wait:
                ev.synchronize()
            if wait or ev.query():
                self._emit(batch, host.numpy())
            else:
                keep.append((batch, host, ev))
        self.pending = keep

    def _emit(self, batch: ProjectionBatch, host: Optional[np.ndarray]) -> None:
        t = batch.table
        rects = np.stack([t["x0"], t["y0"], t["w"], t["h"]], axis=1).astype(np.int32)
        offsets = t["slot_offset"].astype(np.int64)
        dirs = [os.path.join(self.root, f"super_cluster_{i_sc}", f"cam_{cam}") for i_sc, cam, _ in batch.keys]
        empty = np.zeros(0, np.uint8)
        for kind, name in ((0, "wo_occ_cluster"), (1, "visible_cluster")):
            paths = [os.path.join(d, f"{name}_{i}.png") for d, (_, _, i) in zip(dirs, batch.keys)]
            self.writer.submit_rects(paths, host[kind] if host is not None else empty, offsets, rects, self.H, self.W)


def project_all(model, cameras: Cameras, pcd_data, output_root: Optional[str] = None, segmentation_files: Sequence[str] = (),
                want_float: bool = False, keep: bool = True, max_slots: int = 1 << 20, rank: int = 0, world_size: int = 1,
                png_workers: Optional[int] = None) -> ProjectionRun:
    """The whole job list of ``fruit_nerf.py:267-316`` in batches of at most ``max_slots`` rectangle pixels.  With
    ``output_root`` the reference's file tree is written (``super_cluster_<i>/cam_<j>/{wo_occ,visible}_cluster_<c>.png`` and
    the camera's mask copied beside them, ``:316``) by worker threads; ``keep`` retains the compact results in the returned
    ``ProjectionRun`` (for the in-process merger / the float images of ``save=False``)."""
    import time

    H, W = int(cameras.height), int(cameras.width)
    t0 = time.perf_counter()
    keys, table = plan_jobs(cameras, pcd_data, rank, world_size, getattr(model, "compat_projection_cam0", False))
    t_plan = time.perf_counter() - t0
    run = ProjectionRun([], len(cameras), H, W)
    writer = PngWriter(png_workers) if output_root is not None else None
    stage = _PngStage(writer, output_root, H, W) if writer else None
    sizes = table["w"].astype(np.int64) * table["h"].astype(np.int64) if len(table) else np.zeros(0, np.int64)
    rays = 0
    try:
        if writer:
            seen = set()
            for i_sc, cam, _ in keys:  # the mask of every (super-cluster, camera) directory this rank writes into (:316)
                if (i_sc, cam) not in seen and cam < len(segmentation_files) and os.path.exists(segmentation_files[cam]):
                    writer.submit_copy(segmentation_files[cam], os.path.join(output_root, f"super_cluster_{i_sc}", f"cam_{cam}"))
                seen.add((i_sc, cam))
        lo = 0
        while lo < len(keys):
            hi, acc = lo, 0
            while hi < len(keys) and (hi == lo or acc + int(sizes[hi]) <= max_slots):
                acc += int(sizes[hi])
                hi += 1
            batch = run_batch(model, keys[lo:hi], table[lo:hi], H, W, want_float=want_float, want_u8=True)
            rays += batch.num_rays
            if stage:
                stage.push(batch)
            if keep:
                run.batches.append(batch)
            lo = hi
        if stage:
            stage.flush(wait=True)
    finally:
        if writer:
            writer.close()
    run.stats = {"jobs": len(keys), "slots": int(sizes.sum()), "rays": rays, "plan_seconds": t_plan,
                 "png_files": writer.files if writer else 0}
    return run
