#!/usr/bin/env python
"""``exporter.py semantic-pointcloud | pointcloud`` -- mirror of ``crop_nerf/fruit_nerf/scripts/exporter.py:64-136``
and of the reference's copy of ``ns-export pointcloud`` (``crop_nerf/debug/exporter_nerfacto.py:66-146``).

    python exporter.py semantic-pointcloud --load-config RUN/config.json --output-dir OUT \
        [--num-rays-per-batch 512] [--num-points-per-side 3000] [--bounding-box-min ...] [--bounding-box-max ...]
    python exporter.py pointcloud --load-config RUN/config.json --output-dir OUT --num-points 10000000 \
        [--remove-outliers True] [--num-rays-per-batch 2048] [--std-ratio 10.0]
"""

from __future__ import annotations

import argparse
import json
import os
import sys
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple


@dataclass
class ExportSemanticPointCloud:
    """Fields as ``scripts/exporter.py:64-78``."""

    load_config: Path
    output_dir: Path
    use_bounding_box: bool = True
    bounding_box_min: Tuple[float, float, float] = (-1, -1, -1 + 0.318)
    bounding_box_max: Tuple[float, float, float] = (1, 1, 1 + 0.318)
    num_rays_per_batch: int = 512
    num_points_per_side: int = 3000
    # extension: None = what the run's config says (default "split_bf16": bf16 hi + lo matrix products, the fp32 parity bars);
    # "fp32" = exact fp32 products; "f16" = tiny-cuda-nn's class
    matrix_precision: Optional[str] = None

    def main(self) -> None:
        from cropnerf_amd.fruit_nerf.checkpoint import eval_setup
        from cropnerf_amd.fruit_nerf.export.exporter_utils import sample_volume
        from cropnerf_amd.fruit_nerf.ply import write_ply

        if not self.output_dir.exists():
            self.output_dir.mkdir(parents=True, exist_ok=True)  # (several ranks get here at once)
        config, pipeline, _, _ = eval_setup(self.load_config, test_mode="export", matrix_precision=self.matrix_precision)
        pipeline.datamanager.config.eval_num_rays_per_batch = self.num_rays_per_batch
        pipeline.model.setup_inference(render_rgb=True, num_inference_samples=self.num_points_per_side)
        num_points = pipeline.datamanager.setup_inference(num_points=self.num_points_per_side,
                                                          aabb=(self.bounding_box_min, self.bounding_box_max))
        with open(self.load_config.parent / "dataparser_transforms.json", "r") as fp:
            transform_json = json.load(fp)
        pcds = sample_volume(pipeline=pipeline, num_points=num_points, output_dir=self.output_dir, config=config,
                             transform_json=transform_json)
        if pipeline.local_rank != 0:  # several ranks: everyone holds the gathered cloud, rank 0 writes it
            return
        os.makedirs(str(self.output_dir / config.load_dir.parts[-3]), exist_ok=True)
        print("Saving Point Cloud...")
        for name, pcd in pcds.items():
            write_ply(pcd["path"], pcd["points"], pcd["colors"])
        print("Saving Point Cloud: done")


@dataclass
class ExportPointCloud:
    """Fields as ``debug/exporter_nerfacto.py:66-97``."""

    load_config: Path
    output_dir: Path
    num_points: int = 1000000
    remove_outliers: bool = True
    num_rays_per_batch: int = 2048
    std_ratio: float = 10.0
    save_world_frame: bool = False
    obb_center: Optional[Tuple[float, float, float]] = None
    obb_rotation: Optional[Tuple[float, float, float]] = None
    obb_scale: Optional[Tuple[float, float, float]] = None
    matrix_precision: Optional[str] = None
    reorient_normals: bool = True
    normal_method: str = "open3d"  # "open3d" | "model_output" (debug/exporter_nerfacto.py:74-77)
    normal_output_name: str = "normals"

    def main(self) -> None:
        from cropnerf_amd.fruit_nerf.checkpoint import eval_setup
        from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import generate_point_cloud
        from cropnerf_amd.rays import OrientedBox
        from cropnerf_amd.fruit_nerf.ply import write_ply

        if not self.output_dir.exists():
            self.output_dir.mkdir(parents=True, exist_ok=True)  # (several ranks get here at once)
        _, pipeline, _, _ = eval_setup(self.load_config, test_mode="test", matrix_precision=self.matrix_precision)
        pipeline.datamanager.config.train_num_rays_per_batch = self.num_rays_per_batch
        crop_obb = None  # debug/exporter_nerfacto.py:119-121
        if self.obb_center is not None and self.obb_rotation is not None and self.obb_scale is not None:
            crop_obb = OrientedBox.from_params(self.obb_center, self.obb_rotation, self.obb_scale)
        estimate_normals = self.normal_method == "open3d"  # debug/exporter_nerfacto.py:117 (the name is the reference's flag
        # value: the estimate itself is cn_estimate_normals, open3d's algorithm on the device)
        pcd = generate_point_cloud(pipeline=pipeline, num_points=self.num_points, remove_outliers=self.remove_outliers,
                                   reorient_normals=self.reorient_normals, estimate_normals=estimate_normals,
                                   normal_output_name=self.normal_output_name if self.normal_method == "model_output" else None,
                                   std_ratio=self.std_ratio, crop_obb=crop_obb)
        if pipeline.local_rank != 0:
            return
        print("Saving Point Cloud...")
        write_ply(str(self.output_dir / "semantics_pc.ply"), pcd["points"], pcd["colors"], pcd.get("normals"))
        print("Saving Point Cloud: done")


def _floats3(s):
    v = tuple(float(x) for x in s.replace(",", " ").split())
    assert len(v) == 3
    return v


def entrypoint(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    sp = sub.add_parser("semantic-pointcloud")
    pc = sub.add_parser("pointcloud")
    for p in (sp, pc):
        p.add_argument("--load-config", type=Path, required=True)
        p.add_argument("--output-dir", type=Path, required=True)
        p.add_argument("--matrix-precision", choices=["fp32", "split_bf16", "f16"], default=None)
    sp.add_argument("--use-bounding-box", type=lambda s: s.lower() == "true", default=True)
    sp.add_argument("--bounding-box-min", type=_floats3, default=ExportSemanticPointCloud.bounding_box_min)
    sp.add_argument("--bounding-box-max", type=_floats3, default=ExportSemanticPointCloud.bounding_box_max)
    sp.add_argument("--num-rays-per-batch", type=int, default=512)
    sp.add_argument("--num-points-per-side", type=int, default=3000)
    pc.add_argument("--num-points", type=int, default=1000000)
    pc.add_argument("--remove-outliers", type=lambda s: s.lower() == "true", default=True)
    pc.add_argument("--normal-method", choices=["open3d", "model_output"], default="open3d")
    pc.add_argument("--reorient-normals", type=lambda s: s.lower() == "true", default=True)
    pc.add_argument("--num-rays-per-batch", type=int, default=2048)
    pc.add_argument("--std-ratio", type=float, default=10.0)
    pc.add_argument("--save-world-frame", type=lambda s: s.lower() == "true", default=False)
    for flag in ("--obb_center", "--obb_rotation", "--obb_scale"):  # ns-export pointcloud's spelling (README.md:125)
        pc.add_argument(flag, type=float, nargs=3, default=None)
    a = ap.parse_args(argv)
    if a.cmd == "semantic-pointcloud":
        ExportSemanticPointCloud(a.load_config, a.output_dir, a.use_bounding_box, a.bounding_box_min,
                                 a.bounding_box_max, a.num_rays_per_batch, a.num_points_per_side, a.matrix_precision).main()
    else:
        ExportPointCloud(a.load_config, a.output_dir, a.num_points, a.remove_outliers, a.num_rays_per_batch,
                         a.std_ratio, a.save_world_frame, a.obb_center, a.obb_rotation, a.obb_scale, a.matrix_precision,
                         a.reorient_normals, a.normal_method).main()


if __name__ == "__main__":
    sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
    entrypoint()
