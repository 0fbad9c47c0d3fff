"""Run directory I/O standing in for nerfstudio's ``eval_setup(load_config, eval_num_rays_per_chunk, test_mode)``
(used at ``scripts/semantic_projection.py:139-143``, ``scripts/exporter.py:87``, ``debug/exporter_nerfacto.py:105``).

Two layouts are read; ``--load-config`` points at the config file of either:

nerfstudio's own (what ``ns-train fruit_nerf`` of the reference writes, and what ``save_run`` writes by default):
    config.yml                  PyYAML dump of the TrainerConfig tree (``nerfstudio_io.py``)
    dataparser_transforms.json  {"transform": 3x4, "scale": s}   (read by the dense exporter, scripts/exporter.py:100)
    nerfstudio_models/step-XXXXXXXXX.ckpt   {"step", "pipeline": {"_model.<name>": tensor}, "optimizers", ...}
  The model's tensors may be tcnn-packed (the reference's default ``implementation="tcnn"``: converted by
  ``tcnn_params.from_tcnn_state_dict``, hash tables as float16 -- the values tcnn computes with) or nerfstudio's torch
  modules.  Cameras come from ``cameras.json`` next to the config when this package wrote the run, otherwise from the
  capture the config's ``data`` entry names, through the recorded dataparser settings.

this package's first format (round 1; still read):
    config.json                 method name, model-config overrides, scene box, camera intrinsics / poses
    nerfstudio_models/step-XXXXXXXXX.pt   torch.save({"step": n, "params": {logical state-dict name: tensor}})
"""

from __future__ import annotations

import json
import os
import pathlib
import sys
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


def _cameras_dict(cameras: Cameras) -> dict:
    return {"camera_to_worlds": cameras.camera_to_worlds.cpu().tolist(), "fx": cameras.fx.cpu().tolist(),
            "fy": cameras.fy.cpu().tolist(), "cx": cameras.cx.cpu().tolist(), "cy": cameras.cy.cpu().tolist(),
            "height": cameras.height, "width": cameras.width}


def _attach_datasets(pipe, side: dict) -> None:
    """The train / eval datasets the exporters walk (``collect_camera_poses``), from a run's ``cameras.json``."""
    from .data.fruit_datamanager import CameraDataset

    dm = pipe.datamanager
    if side.get("image_filenames"):
        dm.train_dataset = CameraDataset(dm.cameras, side["image_filenames"])
    if side.get("eval"):
        dm.eval_dataset = CameraDataset(_cameras_from_dict(side["eval"]), side["eval"].get("image_filenames"))


def _cameras_from_dict(c: dict) -> Cameras:
    return Cameras(torch.tensor(c["camera_to_worlds"], dtype=torch.float32), torch.tensor(c["fx"]), torch.tensor(c["fy"]),
                   torch.tensor(c["cx"]), torch.tensor(c["cy"]), int(c["height"]), int(c["width"]))


def save_run(run_dir, model_config: FruitNerfModelConfig, cameras: Cameras, scene_box: SceneBox,
             params: Dict[str, torch.Tensor], step: int = 0, transform=None, scale: float = 1.0,
             method_name: str = "fruit_nerf", optimizers: Optional[dict] = None, format: str = "nerfstudio",
             data: Optional[str] = None, dataparser: Optional[dict] = None, trainer_config=None,
             schedulers: Optional[dict] = None, image_filenames=None, eval_cameras: Optional[Cameras] = None,
             eval_image_filenames=None) -> pathlib.Path:
    """Write a run directory; returns the path of its config file (what ``--load-config`` takes).
    ``format="nerfstudio"``: config.yml + step-*.ckpt under nerfstudio's state-dict names (tcnn-packed when the model
    is a tcnn-layout one); ``format="json"``: this package's round-1 layout."""
    run_dir = pathlib.Path(run_dir)
    (run_dir / "nerfstudio_models").mkdir(parents=True, exist_ok=True)
    t = transform if transform is not None else [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]]
    if format == "nerfstudio":
        from . import nerfstudio_io as NIO

        fs = model_config.field_spec(len(cameras))
        ps = model_config.proposal_specs()
        if model_config.implementation == "tcnn":
            from .tcnn_params import to_tcnn_state_dict

            state = to_tcnn_state_dict({k: (v.float() if v.dtype == torch.float16 else v) for k, v in params.items()},
                                       fs, ps)
        else:
            state = NIO.nerfstudio_names({k: v.detach().cpu() for k, v in params.items()})
        parts = run_dir.resolve().parts
        tc = trainer_config
        NIO.write_config_yml(
            run_dir / "config.yml", method_name=method_name, model_config=model_config, data=data,
            output_dir=str(pathlib.Path(*parts[:-3])) if len(parts) > 3 else ".",
            experiment_name=parts[-3] if len(parts) > 3 else "unnamed", timestamp=parts[-1],
            max_num_iterations=getattr(tc, "max_num_iterations", step + 1),
            steps_per_save=getattr(tc, "steps_per_save", 2000), mixed_precision=getattr(tc, "mixed_precision", False),
            train_num_rays_per_batch=getattr(getattr(getattr(tc, "pipeline", None), "datamanager", None),
                                             "train_num_rays_per_batch", 4096),
            eval_num_rays_per_batch=getattr(getattr(getattr(tc, "pipeline", None), "datamanager", None),
                                            "eval_num_rays_per_batch", 4096),
            dataparser=dataparser)
        side = _cameras_dict(cameras)
        if image_filenames is not None:  # what ``semantic_projection.py cameras`` names the frames by
            side["image_filenames"] = [str(f) for f in image_filenames]
        if eval_cameras is not None and len(eval_cameras) > 0:
            side["eval"] = _cameras_dict(eval_cameras)
            if eval_image_filenames is not None:
                side["eval"]["image_filenames"] = [str(f) for f in eval_image_filenames]
        (run_dir / "cameras.json").write_text(json.dumps(side))
        (run_dir / "dataparser_transforms.json").write_text(json.dumps({"transform": t, "scale": scale}))
        NIO.save_checkpoint(run_dir / "nerfstudio_models" / f"step-{step:09d}.ckpt", step, state, optimizers=optimizers,
                            schedulers=schedulers, buffers=NIO.field_buffers(model_config, scene_box.aabb))
        return run_dir / "config.yml"
    mc = asdict(model_config)
    mc["num_proposal_samples_per_ray"] = list(mc["num_proposal_samples_per_ray"])
    raw = {
        "method_name": method_name, "model": mc, "scene_box": scene_box.aabb.tolist(),
        "cameras": _cameras_dict(cameras),
    }
    if image_filenames is not None:
        raw["cameras"]["image_filenames"] = [str(f) for f in image_filenames]
    if eval_cameras is not None and len(eval_cameras) > 0:
        raw["cameras"]["eval"] = _cameras_dict(eval_cameras)
        if eval_image_filenames is not None:
            raw["cameras"]["eval"]["image_filenames"] = [str(f) for f in eval_image_filenames]
    (run_dir / "config.json").write_text(json.dumps(raw))
    (run_dir / "dataparser_transforms.json").write_text(json.dumps({"transform": t, "scale": scale}))
    ckpt = {"step": step, "params": {k: v.detach().cpu() for k, v in params.items()}}
    if optimizers is not None:  # FruitTrainer.state_dict() (+ the datamanager's sampling state): what --load-dir resumes from
        ckpt["optimizers"] = optimizers
    torch.save(ckpt, run_dir / "nerfstudio_models" / f"step-{step:09d}.pt")
    return run_dir / "config.json"


def eval_setup(load_config, eval_num_rays_per_chunk: Optional[int] = None, test_mode: str = "test",
               device: str = "cuda", matrix_precision: Optional[str] = None
               ) -> Tuple[RunConfig, FruitPipeline, pathlib.Path, int]:
    """``matrix_precision`` (extension): ``None`` = automatic -- a run whose checkpoint is tcnn-packed and whose config says
    ``mixed_precision: true`` (every reference run: ``fruit_field.py:95``, ``fruit_nerf_config.py:35``) is evaluated in the
    arithmetic it was trained in, ``"f16"`` (fp16 weights and layer inputs, fp32 accumulation; 1.9x the exact-fp32 mode on
    such a table); pass ``"fp32"`` (or set ``CROPNERF_MATRIX_PRECISION=fp32``) for exact fp32 on the same fp16 values."""
    from ..distributed import init_from_env

    rank, world_size, dist_device = init_from_env()  # under torch.distributed.run: one rank per GPU
    if world_size > 1:
        device = dist_device
    load_config = pathlib.Path(load_config)
    if load_config.suffix in (".yml", ".yaml"):
        return _eval_setup_nerfstudio(load_config, eval_num_rays_per_chunk, test_mode, device, rank, world_size,
                                      matrix_precision)
    raw = json.loads(load_config.read_text())
    cfg = RunConfig(load_config, raw)
    m = dict(raw["model"])
    m["num_proposal_samples_per_ray"] = tuple(m["num_proposal_samples_per_ray"])
    if isinstance(m.get("background_color"), list):
        m["background_color"] = tuple(m["background_color"])
    model_cfg = FruitNerfModelConfig(**m)
    if matrix_precision or os.environ.get("CROPNERF_MATRIX_PRECISION"):
        model_cfg.matrix_precision = matrix_precision or os.environ["CROPNERF_MATRIX_PRECISION"]
    if eval_num_rays_per_chunk is not None:
        model_cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
        cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
    cams = _cameras_from_dict(raw["cameras"])
    ckpts = sorted(cfg.load_dir.glob("step-*.pt"))
    if not ckpts:
        raise FileNotFoundError(f"no checkpoint under {cfg.load_dir}")
    state = torch.load(ckpts[-1], map_location="cpu")
    pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(), model_cfg), device=device, cameras=cams,
                         scene_box=SceneBox(torch.tensor(raw["scene_box"], dtype=torch.float32)), test_mode=test_mode,
                         params=state["params"], world_size=world_size, local_rank=rank)
    _attach_datasets(pipe, raw["cameras"])
    pipe.eval()
    return cfg, pipe, ckpts[-1], int(state["step"])


def _eval_setup_nerfstudio(load_config: pathlib.Path, eval_num_rays_per_chunk, test_mode, device, rank, world_size,
                           matrix_precision: Optional[str] = None):
    """``eval_setup`` on a nerfstudio run directory (config.yml + step-*.ckpt)."""
    from . import nerfstudio_io as NIO
    from .tcnn_params import from_tcnn_state_dict, is_tcnn_state_dict

    tree = NIO.load_config_yml(load_config)
    cfg = RunConfig(load_config, tree)
    cfg.load_dir = NIO.checkpoint_dir(load_config, tree)
    model_cfg = NIO.model_config_from_tree(tree)
    if eval_num_rays_per_chunk is not None:
        model_cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
        cfg.eval_num_rays_per_chunk = eval_num_rays_per_chunk
    ckpt = NIO.latest_checkpoint(cfg.load_dir)
    step, state, _ = NIO.load_checkpoint(ckpt)
    num_images = int(state["field.embedding_appearance.embedding.weight"].shape[0])
    # cameras / scene box: the side file this package writes, else the capture named by the config
    side = load_config.parent / "cameras.json"
    semantics = None
    side_dict, eval_out, train_names = {}, None, None
    if side.exists():
        side_dict = json.loads(side.read_text())
        cams = _cameras_from_dict(side_dict)
        aabb = state.get("field.aabb")
        scene_box = SceneBox(aabb.to(torch.float32) if aabb is not None else torch.tensor([[-1.0] * 3, [1.0] * 3]))
    else:
        from .data.cotton_nerf_dataparser import CottonNerfDataParserConfig

        dp = dict(tree["pipeline"]["datamanager"].get("dataparser") or {})
        data = dp.get("data") or tree["pipeline"]["datamanager"].get("data") or tree.get("data")
        if not data:
            raise FileNotFoundError(f"{load_config}: no cameras.json next to the config and no 'data' entry in it")
        pc = CottonNerfDataParserConfig()
        for k, v in dp.items():
            if hasattr(pc, k) and k not in ("data", "_target"):
                setattr(pc, k, v)
        pc.data = pathlib.Path(str(data))
        parser = pc.setup()
        out = parser.get_dataparser_outputs("train")
        cams, scene_box, semantics = out.cameras, out.scene_box, out.metadata.get("semantics")
        train_names = out.image_filenames
        eval_out = parser.get_dataparser_outputs("val" if test_mode == "val" else "test")
    if len(cams) != num_images:
        raise ValueError(f"{load_config}: {len(cams)} training cameras but the appearance embedding has {num_images} rows")
    if is_tcnn_state_dict(state):
        # hash tables as half2 entries: exactly the values tcnn's kernels read (it casts its fp32 masters per forward)
        model_cfg.implementation, model_cfg.hash_table_dtype = "tcnn", "float16"
        params = from_tcnn_state_dict(state, model_cfg.field_spec(num_images), model_cfg.proposal_specs(), device,
                                      torch.float16)
        # ... and in the arithmetic tcnn runs under mixed precision, unless the caller / the run's own config says otherwise
        explicit = matrix_precision or os.environ.get("CROPNERF_MATRIX_PRECISION")
        if not explicit and "matrix_precision" not in tree["pipeline"]["model"] and bool(tree.get("mixed_precision", False)):
            model_cfg.matrix_precision = "f16"
            print(f"[cropnerf_amd] {load_config}: tcnn-packed checkpoint of a mixed-precision run -> matrix_precision='f16' "
                  "(tiny-cuda-nn's own arithmetic class); eval_setup(..., matrix_precision='fp32') or "
                  "CROPNERF_MATRIX_PRECISION=fp32 selects exact fp32", file=sys.stderr)
    else:
        model_cfg.implementation = "torch"
        from ..config import param_shapes

        want = param_shapes(model_cfg.field_spec(num_images), model_cfg.proposal_specs())
        params = {k: state[k] for k in want if k in state}
        if "camera_optimizer.pose_adjustment" not in params:
            params["camera_optimizer.pose_adjustment"] = torch.zeros(num_images, 6)
        missing = sorted(set(want) - set(params))
        if missing:
            raise KeyError(f"{ckpt}: parameters missing from the checkpoint: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
    if matrix_precision or os.environ.get("CROPNERF_MATRIX_PRECISION"):
        model_cfg.matrix_precision = matrix_precision or os.environ["CROPNERF_MATRIX_PRECISION"]
    pipe = FruitPipeline(FruitPipelineConfig(FruitDataManagerConfig(), model_cfg), device=device, cameras=cams,
                         scene_box=scene_box, test_mode=test_mode, params=params, world_size=world_size,
                         local_rank=rank, semantics=semantics)
    _attach_datasets(pipe, side_dict)
    if train_names is not None:
        from .data.fruit_datamanager import CameraDataset

        pipe.datamanager.train_dataset = CameraDataset(pipe.datamanager.cameras, train_names, out.metadata)
        if eval_out is not None and len(eval_out.image_filenames) > 0:
            pipe.datamanager.eval_dataset = CameraDataset(eval_out.cameras, eval_out.image_filenames, eval_out.metadata)
    pipe.eval()
    return cfg, pipe, ckpt, step
