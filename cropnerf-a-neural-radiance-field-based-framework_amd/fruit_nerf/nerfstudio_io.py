"""nerfstudio run-directory format: ``config.yml`` + ``nerfstudio_models/step-XXXXXXXXX.ckpt``.

What the reference's CLIs load through nerfstudio's ``eval_setup(load_config, ...)``
(``crop_nerf/fruit_nerf/scripts/semantic_projection.py:139-143``, ``scripts/exporter.py:87,100-101``,
``debug/exporter_nerfacto.py:105``) and what ``ns-train fruit_nerf`` writes:

    <output_dir>/<experiment>/<method>/<timestamp>/config.yml
                                                    dataparser_transforms.json      {"transform": 3x4, "scale": s}
                                                    nerfstudio_models/step-000029999.ckpt

* ``config.yml`` is a PyYAML dump of the ``TrainerConfig`` dataclass tree with ``!!python/object:...`` tags.  nerfstudio
  is not installed here, so the tags are read as plain mappings (class path kept under ``"__class__"``) and written back
  as the same tags -- no class is imported either way.
* the checkpoint is ``torch.save({"step", "pipeline", "optimizers", "schedulers", "scalers"})``; ``pipeline`` is the
  state dict of the ``VanillaPipeline`` module, i.e. the model's tensors under ``_model.`` (``module._model.`` when it
  was saved from a DDP run, ``fruit_pipeline.py:119-121``).

State-dict names of the model (``fruit_nerf.py:97-142``, ``fruit_field.py:99-167``): with ``implementation="tcnn"`` (the
default) ``field.mlp_base_grid.tcnn_encoding.params`` etc. (see ``tcnn_params.py``); with the torch fallback
``field.mlp_base_grid.hash_table``, ``field.mlp_base_mlp.layers.N.{weight,bias}``, and for the proposal networks
``proposal_networks.N.mlp_base.model.0.hash_table`` / ``...mlp_base.model.1.layers.M.*`` (nerfstudio's
``MLPWithHashEncoding`` is a ``Sequential(HashEncoding, MLP)``; this package's own short names
``proposal_networks.N.encoding.hash_table`` / ``proposal_networks.N.mlp.layers.M.*`` are accepted as well).
Buffers (``field.aabb``, ``field.max_res`` ...) and modules without a counterpart here (``lpips.*``) are ignored on load
and the buffers are written on save.
"""

from __future__ import annotations

import os
import pathlib
import re
from dataclasses import asdict, fields
from typing import Any, Dict, Optional, Tuple

import torch
import yaml

from ..config import FruitNerfModelConfig

PY = "tag:yaml.org,2002:python/"

# ----------------------------------------------------------------------------------------------------------- config.yml


class _Loader(yaml.SafeLoader):
    """SafeLoader + the ``python/`` tags of a nerfstudio config, constructed as plain data (nothing is imported)."""


def _object(loader, suffix, node):
    d = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
    d["__class__"] = suffix
    return d


def _apply(loader, suffix, node):
    args = loader.construct_sequence(node, deep=True) if isinstance(node, yaml.SequenceNode) else []
    if suffix in ("pathlib.PosixPath", "pathlib.WindowsPath", "pathlib.Path"):
        return str(pathlib.PurePosixPath(*[str(a) for a in args])) if args else "."
    return {"__apply__": suffix, "args": args}


_Loader.add_multi_constructor(PY + "object:", _object)
_Loader.add_multi_constructor(PY + "object/apply:", _apply)
_Loader.add_multi_constructor(PY + "name:", lambda loader, suffix, node: suffix)
_Loader.add_constructor(PY + "tuple", lambda loader, node: tuple(loader.construct_sequence(node, deep=True)))


def load_config_yml(path) -> Dict[str, Any]:
    with open(path, encoding="utf-8") as f:
        tree = yaml.load(f, Loader=_Loader)
    if not isinstance(tree, dict) or "pipeline" not in tree:
        raise ValueError(f"{path}: not a nerfstudio TrainerConfig dump (no 'pipeline' entry)")
    return tree


class _Obj(dict):
    def __init__(self, cls: str, **kw):
        super().__init__(**kw)
        self.cls = cls


class _Name(str):
    pass


class _Path(str):
    pass


class _Dumper(yaml.SafeDumper):
    pass


_Dumper.add_representer(_Obj, lambda d, o: d.represent_mapping(PY + "object:" + o.cls, dict(o)))
_Dumper.add_representer(_Name, lambda d, o: d.represent_scalar(PY + "name:" + str(o), ""))
_Dumper.add_representer(_Path, lambda d, o: d.represent_sequence(
    PY + "object/apply:pathlib.PosixPath", list(pathlib.PurePosixPath(str(o)).parts)))
_Dumper.add_representer(tuple, lambda d, o: d.represent_sequence(PY + "tuple", list(o)))


def write_config_yml(path, *, method_name: str, model_config: FruitNerfModelConfig, data: Optional[str],
                     output_dir: str, experiment_name: str, timestamp: str, max_num_iterations: int,
                     steps_per_save: int, mixed_precision: bool, train_num_rays_per_batch: int,
                     eval_num_rays_per_batch: int, dataparser: Optional[Dict[str, Any]] = None,
                     optimizers: Optional[Dict[str, Any]] = None) -> None:
    """A ``TrainerConfig`` dump with the reference's class paths (``fruit_nerf_config.py:29-65``)."""
    mc = asdict(model_config)
    if mc.get("matrix_precision") == FruitNerfModelConfig().matrix_precision:
        # this package's extension field at its default makes no statement (a reference config has no such key): eval_setup
        # then picks the arithmetic from the checkpoint -- fp16 products for a tcnn-packed mixed-precision run
        del mc["matrix_precision"]
    mc["num_proposal_samples_per_ray"] = tuple(mc["num_proposal_samples_per_ray"])
    if isinstance(mc.get("background_color"), (list, tuple)):
        mc["background_color"] = tuple(mc["background_color"])
    dp = dict(dataparser or {})
    dp_cls = dp.pop("__class__", "fruit_nerf.data.cotton_nerf_dataparser.CottonNerfDataParserConfig")
    for k, v in list(dp.items()):
        if isinstance(v, (pathlib.PurePath,)):
            dp[k] = _Path(str(v))
    tree = _Obj(
        "nerfstudio.engine.trainer.TrainerConfig",
        _target=_Name("nerfstudio.engine.trainer.Trainer"),
        method_name=method_name, experiment_name=experiment_name, timestamp=timestamp,
        output_dir=_Path(output_dir), relative_model_dir=_Path("nerfstudio_models"),
        data=_Path(data) if data else None,
        max_num_iterations=int(max_num_iterations), steps_per_save=int(steps_per_save),
        mixed_precision=bool(mixed_precision), load_dir=None, load_step=None, vis="viewer",
        pipeline=_Obj(
            "fruit_nerf.fruit_pipeline.FruitPipelineConfig",
            _target=_Name("fruit_nerf.fruit_pipeline.FruitPipeline"),
            datamanager=_Obj(
                "fruit_nerf.data.fruit_datamanager.FruitDataManagerConfig",
                _target=_Name("fruit_nerf.data.fruit_datamanager.FruitDataManager"),
                data=_Path(data) if data else None,
                dataparser=_Obj(dp_cls, **dp),
                train_num_rays_per_batch=int(train_num_rays_per_batch),
                eval_num_rays_per_batch=int(eval_num_rays_per_batch)),
            model=_Obj("fruit_nerf.fruit_nerf.FruitNerfModelConfig",
                       _target=_Name("fruit_nerf.fruit_nerf.FruitModel"), **mc)),
        optimizers=optimizers or {},
    )
    with open(path, "w", encoding="utf-8") as f:
        yaml.dump(tree, f, Dumper=_Dumper, default_flow_style=False, sort_keys=False)


def model_config_from_tree(tree: Dict[str, Any]) -> FruitNerfModelConfig:
    """``pipeline.model`` of a config tree -> ``FruitNerfModelConfig`` (fields this package does not know are ignored:
    a nerfacto config carries many -- loss multipliers of unused heads, viewer options ...)."""
    m = tree["pipeline"]["model"]
    known = {f.name for f in fields(FruitNerfModelConfig)}
    kw = {k: v for k, v in m.items() if k in known}
    if "num_proposal_samples_per_ray" in kw:
        kw["num_proposal_samples_per_ray"] = tuple(kw["num_proposal_samples_per_ray"])
    if isinstance(kw.get("background_color"), list):
        kw["background_color"] = tuple(kw["background_color"])
    if "proposal_net_args_list" in kw:
        kw["proposal_net_args_list"] = [dict(a) for a in kw["proposal_net_args_list"]]
    return FruitNerfModelConfig(**kw)


def checkpoint_dir(config_path, tree: Dict[str, Any]) -> pathlib.Path:
    """nerfstudio's ``config.get_checkpoint_dir()`` = output_dir/experiment/method/timestamp/relative_model_dir; the
    exporters are normally pointed at the config inside that directory, which is tried first (runs get moved)."""
    here = pathlib.Path(config_path).parent / str(tree.get("relative_model_dir") or "nerfstudio_models")
    if here.is_dir():
        return here
    return (pathlib.Path(str(tree["output_dir"])) / str(tree["experiment_name"]) / str(tree["method_name"]) /
            str(tree["timestamp"]) / str(tree.get("relative_model_dir") or "nerfstudio_models"))


# ----------------------------------------------------------------------------------------------------------- checkpoints

_PREFIXES = ("module._model.", "_model.module.", "_model.")
_PROP_ALIASES = (
    (re.compile(r"^proposal_networks\.(\d+)\.mlp_base\.model\.0\.hash_table$"), r"proposal_networks.\1.encoding.hash_table"),
    (re.compile(r"^proposal_networks\.(\d+)\.mlp_base\.model\.1\.layers\.(\d+)\.(weight|bias)$"),
     r"proposal_networks.\1.mlp.layers.\2.\3"),
    (re.compile(r"^proposal_networks\.(\d+)\.mlp_base\.encoder\.hash_table$"), r"proposal_networks.\1.encoding.hash_table"),
)


def model_state_from_pipeline(pipeline_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Strip the pipeline prefixes; keep the model's entries only."""
    out = {}
    for k, v in pipeline_state.items():
        for p in _PREFIXES:
            if k.startswith(p):
                name = k[len(p):]
                for rx, rep in _PROP_ALIASES:
                    name = rx.sub(rep, name)
                out[name] = v
                break
    if not out:
        raise ValueError("checkpoint has no '_model.' entries under 'pipeline'")
    return out


def latest_checkpoint(load_dir) -> pathlib.Path:
    ckpts = sorted(pathlib.Path(load_dir).glob("step-*.ckpt"))
    if not ckpts:
        raise FileNotFoundError(f"no step-*.ckpt under {load_dir}")
    return ckpts[-1]


def load_checkpoint(path, trusted: Optional[bool] = None) -> Tuple[int, Dict[str, torch.Tensor], Dict[str, Any]]:
    """-> (step, model state dict without prefixes, the whole loaded dict).

    A step-*.ckpt is a pickle, and run directories now come from outside this package (reference-trained runs handed to
    ``eval_setup`` / ``--load-dir``): the file is read with ``weights_only=True`` first -- tensors, numbers, strings and
    containers, which is all a nerfstudio checkpoint's ``pipeline`` entry holds and all the exporters need.  Only if that
    fails AND the caller vouches for the file (``trusted=True``, or ``CROPNERF_TRUST_CHECKPOINTS=1`` in the environment: this
    package's own checkpoints carry generator states and per-rank stream objects for ``--load-dir`` resumes) is it
    unpickled without restrictions."""
    import os
    import pickle

    try:
        loaded = torch.load(path, map_location="cpu", weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, TypeError) as e:
        if trusted is None:
            trusted = os.environ.get("CROPNERF_TRUST_CHECKPOINTS") == "1"
        if not trusted:
            raise ValueError(
                f"{path}: holds objects beyond tensors and plain containers ({type(e).__name__}: {str(e)[:200]}); loading it "
                "runs arbitrary pickle code -- pass trusted=True / set CROPNERF_TRUST_CHECKPOINTS=1 only for files you wrote") from e
        loaded = torch.load(path, map_location="cpu", weights_only=False)
    if "pipeline" not in loaded:
        raise ValueError(f"{path}: not a nerfstudio checkpoint (no 'pipeline' entry)")
    return int(loaded["step"]), model_state_from_pipeline(loaded["pipeline"]), loaded


def nerfstudio_names(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """This package's parameter names -> the names nerfstudio's modules give them (torch implementation)."""
    out = {}
    for k, v in params.items():
        m = re.match(r"^proposal_networks\.(\d+)\.encoding\.hash_table$", k)
        if m:
            k = f"proposal_networks.{m.group(1)}.mlp_base.model.0.hash_table"
        m = re.match(r"^proposal_networks\.(\d+)\.mlp\.layers\.(\d+)\.(weight|bias)$", k)
        if m:
            k = f"proposal_networks.{m.group(1)}.mlp_base.model.1.layers.{m.group(2)}.{m.group(3)}"
        out[k] = v
    return out


def save_checkpoint(path, step: int, model_state: Dict[str, torch.Tensor], optimizers: Optional[dict] = None,
                    schedulers: Optional[dict] = None, buffers: Optional[Dict[str, torch.Tensor]] = None,
                    extra: Optional[dict] = None) -> None:
    pipeline = {"_model." + k: v.detach().cpu() for k, v in model_state.items()}
    for k, v in (buffers or {}).items():
        pipeline["_model." + k] = v.detach().cpu()
    ckpt = {"step": int(step), "pipeline": pipeline, "optimizers": optimizers or {}, "schedulers": schedulers or {},
            "scalers": {}}
    if extra:
        ckpt.update(extra)
    pathlib.Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(ckpt, path)


def field_buffers(model_config: FruitNerfModelConfig, aabb: torch.Tensor) -> Dict[str, torch.Tensor]:
    """The buffers ``FruitField`` registers (``fruit_field.py:99-104``)."""
    return {"field.aabb": aabb.detach().cpu().to(torch.float32), "field.max_res": torch.tensor(model_config.max_res),
            "field.num_levels": torch.tensor(model_config.num_levels),
            "field.log2_hashmap_size": torch.tensor(model_config.log2_hashmap_size)}
