"""Run directory I/O standing in for nerfstudio's ``eval_setup(load_config, eval_num_rays_per_chunk, test_mode)``
(used at ``scripts/semantic_projection.py:139-143``, ``scripts/exporter.py:87``, ``debug/exporter_nerfacto.py:105``).

Layout of a run directory (what ``--load-config`` points into):
    config.json                 method name, model-config overrides, scene box, camera intrinsics / poses
    dataparser_transforms.json  {"transform": 3x4, "scale": s}   (read by the dense exporter, scripts/exporter.py:100)
    nerfstudio_models/step-XXXXXXXXX.pt   torch.save({"step": n, "params": {logical state-dict name: tensor}})
Parameter names follow the reference's module names (checkpoint row); importing tcnn-packed
checkpoints written by the reference is a "next" item."""

from __future__ import annotations

import json
import pathlib
from dataclasses import asdict
from typing import Dict, Optional, Tuple

import torch

from ..config import FruitNerfModelConfig
from ..rays import Cameras, SceneBox
from .fruit_pipeline import FruitPipeline, FruitPipelineConfig
from .data.fruit_datamanager import FruitDataManagerConfig


class RunConfig:
    def __init__(self, path: pathlib.Path, raw: dict):
        self.path = path
        self.raw = raw
        self.load_dir = path.parent / "nerfstudio_models"
        self.eval_num_rays_per_chunk: Optional[int] = None


def save_run(run_dir, model_config: FruitNerfModelConfig, cameras: Cameras, scene_box: SceneBox,
             params: Dict[str, torch.Tensor], step: int = 0, transform=None, scale: float = 1.0,
             method_name: str = "fruit_nerf", optimizers: Optional[dict] = None) -> pathlib.Path:
    run_dir = pathlib.Path(run_dir)
    (run_dir / "nerfstudio_models").mkdir(parents=True, exist_ok=True)
    mc = asdict(model_config)
    mc["num_proposal_samples_per_ray"] = list(mc["num_proposal_samples_per_ray"])
    raw = {
        "method_name": method_name, "model": mc, "scene_box": scene_box.aabb.tolist(),
        "cameras": {"camera_to_worlds": cameras.camera_to_worlds.cpu().tolist(), "fx": cameras.fx.cpu().tolist(),
                    "fy": cameras.fy.cpu().tolist(), "cx": cameras.cx.cpu().tolist(), "cy": cameras.cy.cpu().tolist(),
                    "height": cameras.height, "width": cameras.width},
    }
    (run_dir / "config.json").write_text(json.dumps(raw))
    t = transform if transform is not None else [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]
    (run_dir / "dataparser_transforms.json").write_text(json.dumps({"transform": t, "scale": scale}))
    ckpt = {"step": step, "params": {k: v.detach().cpu() for k, v in params.items()}}
    if optimizers is not None:  # FruitTrainer.state_dict() (+ the datamanager's sampling state): what --load-dir resumes from
        ckpt["optimizers"] = optimizers
    torch.save(ckpt, run_dir / "nerfstudio_models" / f"step-{step:09d}.pt")
    return run_dir / "config.json"


def eval_setup(load_config, eval_num_rays_per_chunk: Optional[int] = None, test_mode: str = "test",
               device: str = "cuda") -> Tuple[RunConfig, FruitPipeline, pathlib.Path, int]:
    from ..distributed import init_from_env

    rank, world_size, dist_device = init_from_env()  # under torch.distributed.run: one rank per GPU
    if world_size > 1:
        device = dist_device
    load_config = pathlib.Path(load_config)
    raw = json.loads(load_config.read_text())
    cfg = RunConfig(load_config, raw)
    m = dict(raw["model"])
    m["num_proposal_samples_per_ray"] = tuple(m["num_proposal_samples_per_ray"])
    if isinstance(m.get("background_color"), list):
        m["background_color"] = tuple(m["background_color"])
    model_cfg = FruitNerfModelConfig(**m)
    if eval_num_rays_per_chunk is not None:
        model_cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
        cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
    c = raw["cameras"]
    cams = Cameras(torch.tensor(c["camera_to_worlds"], dtype=torch.float32), torch.tensor(c["fx"]), torch.tensor(c["fy"]),
                   torch.tensor(c["cx"]), torch.tensor(c["cy"]), int(c["height"]), int(c["width"]))
    ckpts = sorted(cfg.load_dir.glob("step-*.pt"))
    if not ckpts:
        raise FileNotFoundError(f"no checkpoint under {cfg.load_dir}")
    state = torch.load(ckpts[-1], map_location="cpu")
    pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(), model_cfg), device=device, cameras=cams,
                         scene_box=SceneBox(torch.tensor(raw["scene_box"], dtype=torch.float32)), test_mode=test_mode,
                         params=state["params"], world_size=world_size, local_rank=rank)
    pipe.eval()
    return cfg, pipe, ckpts[-1], int(state["step"])
