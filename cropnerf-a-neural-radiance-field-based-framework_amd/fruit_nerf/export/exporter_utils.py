"""Dense volume export -- mirror of ``crop_nerf/fruit_nerf/export/exporter_utils.py:47-258`` (``sample_volume``).

Per call: orthographic rays x N uniform samples -> ``FruitModel.get_export_outputs`` (``cn_render_samples``) ->
threshold masks + stream compaction on the device (``cn_export_compact``: semantic logit >= 3, density >= 70,
label) -> three point sets.  Only kept points ever cross PCIe, once, at the end (the reference boolean-indexes and
``.cpu()``s three sets per 512-ray call and empties the allocator cache each time, ``:127-171``)."""

from __future__ import annotations

import pathlib
from typing import Dict, Optional

import numpy as np
import torch

from ... import ops


def sample_volume(pipeline, num_points: int, output_dir: Optional[pathlib.Path] = None, config=None,
                  transform_json: Optional[dict] = None, capacity: Optional[int] = None,
                  sem_thresh: float = 3.0, den_thresh: float = 70.0) -> Dict[str, Dict]:
    """Returns {'semantic_colormap' | 'semantic' | 'density': {'points' [n,3] f64, 'colors' [n,3] f64, 'path'}}.

    ``num_points`` = number of rays of the surface grid (what ``datamanager.setup_inference`` returned).
    """
    model, dm = pipeline.model, pipeline.datamanager
    device = model.device
    S = model.num_inference_samples
    if capacity is None:
        capacity = max(1 << 20, min(num_points * S // 8, 1 << 27))
    from ...distributed import all_gather_points, world as _world

    rank, world_size = _world()
    buffers = None
    done = 0

    def one_batch(ray_bundle):
        out = model(ray_bundle)
        return ops.export_compact(out["point_location"].reshape(-1, 3), out["rgb"].reshape(-1, 3),
                                  out["semantics"].reshape(-1), out["density"].reshape(-1), capacity, sem_thresh,
                                  den_thresh, buffers), out["point_location"].shape[0]

    with torch.no_grad():
        if world_size == 1:
            while done < num_points:
                ray_bundle, _ = dm.next_sample_volume(0)
                if len(ray_bundle) == 0:
                    break
                buffers, rays = one_batch(ray_bundle)
                done += rays  # progress.advance(task, rays) at :172
        else:  # several ranks: the batches of the surface grid are dealt round-robin, the kept points gathered at the end
            per_batch = dm.config.eval_num_rays_per_batch
            for b in range(-(-num_points // per_batch)):
                if b % world_size != rank:
                    continue
                dm.train_count = b  # next_sample_volume serves batch train_count + 1 (ray_generators.py:52-56)
                ray_bundle, _ = dm.next_sample_volume(0)
                if len(ray_bundle) == 0:
                    break
                buffers, _ = one_batch(ray_bundle)
    if buffers is None:  # a rank without a batch
        dev = model.device
        buffers = ([torch.empty(0, 3, device=dev)] * 3, [torch.empty(0, 4, device=dev)] * 3,
                   torch.zeros(3, dtype=torch.int64, device=dev))
    pts, cols, counts = buffers
    counts = [int(c) for c in counts.cpu().tolist()]
    if max(counts) > capacity:
        raise RuntimeError(f"export capacity {capacity} exceeded (kept {counts}); pass a larger capacity")
    if world_size > 1:
        pts, cols = list(pts), list(cols)
        for k in range(3):
            w = cols[k].shape[1]
            rows = all_gather_points(torch.cat([pts[k][: counts[k]], cols[k][: counts[k]]], dim=-1).contiguous())
            pts[k], cols[k], counts[k] = rows[:, :3].contiguous(), rows[:, 3:3 + w].contiguous(), rows.shape[0]
    scale = 1.0
    if transform_json is not None:
        scale = (1.0 / float(transform_json["scale"])) * 2.0  # pcd.scale(1/scale) then pcd.scale(2), :190-191
    base = None
    if output_dir is not None and config is not None:
        base = pathlib.Path(output_dir) / config.load_dir.parts[-3]
    res = {}
    for k, name in enumerate(("semantic_colormap", "semantic", "density")):
        p = pts[k][: counts[k]].double().cpu().numpy() * scale
        c = cols[k][: counts[k]].double().cpu().numpy()
        if name != "semantic_colormap" and c.shape[0] != 0:
            c = c / c.max()  # "Normalize to visualize as point cloud", :205-206,229-230
        res[name] = {"points": p, "colors": c[:, :3], "path": None if base is None else str(base / f"{name}.ply")}
    return res
