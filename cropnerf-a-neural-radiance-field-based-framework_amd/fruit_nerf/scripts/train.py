"""``ns-train fruit_nerf --data <dir>`` for this package: trains one of the three method specifications of
``fruit_nerf_config.py`` (``fruit_nerf``, ``fruit_nerf_big``, ``fruit_nerf_huge``) on a nerfstudio-format capture with a
``semantics/`` mask folder (``data/cotton_nerf_dataparser.py``) and writes the run directory the exporters read:

    outputs/<experiment>/<method>/<timestamp>/config.yml                    (nerfstudio's TrainerConfig dump)
                                              dataparser_transforms.json
                                              cameras.json
                                              nerfstudio_models/step-000001999.ckpt
                                                  {"step", "pipeline": {"_model.<name>": tensor}, "optimizers", ...}

    python cropnerf-a-neural-radiance-field-based-framework_amd/fruit_nerf/scripts/train.py fruit_nerf --data plant_1 \\
        [--output-dir outputs] [--max-num-iterations 40000] [--steps-per-save 2000] [--downscale-factor 2]

The loop is nerfstudio's ``Trainer.train`` reduced to what the method needs: ``next_train`` -> ``FruitTrainer.train_iteration``
(forward, losses, backward, one optimiser step per parameter group with its exponential-decay schedule), a checkpoint
every ``steps_per_save`` iterations and at the end, PSNR of the eval split at the end.  Multi-GPU: launch with
``torch.distributed.run`` (one rank per GPU): every rank draws its own ray batch and the gradients are averaged with one
all-reduce (``FruitTrainer.all_reduce_gradients``)."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from datetime import datetime
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))


def train(method: str, data: Path, output_dir: Path = Path("outputs"), max_num_iterations=None, steps_per_save=None,
          downscale_factor=None, experiment_name=None, timestamp=None, seed: int = 0, log_every: int = 100,
          train_split_fraction=None, device: str = "cuda", quiet: bool = False, load_dir=None, implementation=None,
          matrix_precision=None):
    from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC
    from cropnerf_amd.fruit_nerf.checkpoint import save_run
    from cropnerf_amd.fruit_nerf.data.cotton_dataset import FruitDataset
    from cropnerf_amd.fruit_nerf.data.cotton_nerf_dataparser import CottonNerfDataParserConfig
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import FruitDataManager
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer, groups_from_spec

    specs = FC.NATIVE_METHODS  # this package's own dataclasses (FC.fruit_nerf_method may be nerfstudio's type)
    if method not in specs:
        raise SystemExit(f"unknown method {method!r}; choose from {sorted(specs)}")
    import copy as _copy

    tc = _copy.deepcopy(specs[method].config)
    if implementation is not None:  # "tcnn" (the reference's default module implementation) or "torch"
        tc.pipeline.model.implementation = implementation
    if matrix_precision is not None:
        # "f16": the arithmetic the reference trains in (TrainerConfig.mixed_precision=True on tiny-cuda-nn's fp16 modules,
        # fruit_nerf_config.py:35) -- fp16 operands in the field's forward / backward recompute, bf16 gradient products, fp32
        # sums and master parameters (cn_field_backward_mp); default: exact fp32
        tc.pipeline.model.matrix_precision = matrix_precision
    iters = max_num_iterations if max_num_iterations is not None else tc.max_num_iterations
    save_every = steps_per_save if steps_per_save is not None else tc.steps_per_save
    from cropnerf_amd.distributed import init_from_env

    rank, world, dist_device = init_from_env()
    if world > 1:
        import torch.distributed as dist

        device = dist_device
    say = (lambda *a: None) if (quiet or rank != 0) else (lambda *a: print(*a, flush=True))

    import copy

    pc = copy.deepcopy(tc.pipeline.datamanager.dataparser) or CottonNerfDataParserConfig()  # the method's dataparser
    pc.data, pc.downscale_factor = Path(data), downscale_factor
    if train_split_fraction is not None:
        pc.train_split_fraction = train_split_fraction
    parser = pc.setup()
    train_out = parser.get_dataparser_outputs("train")
    eval_out = parser.get_dataparser_outputs("val")
    train_set = FruitDataset(train_out, tc.pipeline.datamanager.camera_res_scale_factor)
    say(f"[data] {len(train_set)} training / {len(eval_out.image_filenames)} eval images of "
        f"{train_out.cameras.height} x {train_out.cameras.width}, dataparser scale {train_out.dataparser_scale:.5f}")
    dm = FruitDataManager.from_dataset(tc.pipeline.datamanager, train_set, device=device, seed=seed, world_size=world,
                                       local_rank=rank)
    model = FruitModel(tc.pipeline.model, scene_box=train_out.scene_box, num_train_data=len(train_set),
                       metadata=train_out.metadata, device=device, test_mode="val", seed=seed)
    model.training = True
    trainer = FruitTrainer(model, groups_from_spec(tc.optimizers), seed=seed + rank)

    start = 0
    if load_dir is not None:  # ns-train --load-dir: parameters, optimiser moments, schedules and sampling state
        from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO
        from cropnerf_amd.fruit_nerf.tcnn_params import from_tcnn_state_dict, is_tcnn_state_dict

        ck = NIO.latest_checkpoint(load_dir)
        step0, state, loaded = NIO.load_checkpoint(ck)
        if is_tcnn_state_dict(state) != (model.config.implementation == "tcnn"):
            raise ValueError(f"{ck} holds a {'tcnn' if is_tcnn_state_dict(state) else 'torch'}-implementation model, the "
                             f"method is configured for {model.config.implementation!r}")
        if is_tcnn_state_dict(state):  # fp32 tables: the master values, not their half cast
            state = from_tcnn_state_dict(state, model.field_spec, model.proposal_specs, device, torch.float32)  # unrounded
        for k in model.params:
            model.params[k].copy_(state[k].to(device))
        opt = loaded.get("optimizers") or {}
        if "exp_avg" in opt:
            # every rank has its own random streams (jitter, pixel sampler): restore THIS rank's; a checkpoint written
            # by a different number of ranks gives new streams derived from (seed, rank, step) instead of duplicates
            ranks = opt.get("rank_states") or []
            mine = ranks[rank] if len(ranks) == world else None
            if len(ranks) != world:
                say(f"[resume] WARNING: {ck} was written by {len(ranks)} rank(s), this run has {world}: the per-rank random "
                    "streams (ray jitter, pixel sampler) cannot be continued and are re-derived from (seed, rank, step)")
            trainer.load_state_dict(opt, load_generator=False)
            if mine is not None:
                trainer._gen.set_state(mine["generator"])
                dm._gen.set_state(mine["datamanager_generator"])
                dm.train_count = int(mine["train_count"])
            else:
                trainer._gen.manual_seed(seed + rank + 7919 * (step0 + 1))
                dm._gen.manual_seed(seed + 1000 * rank + 104729 * (step0 + 1))
                dm.train_count = step0 + 1
        start = step0 + 1
        say(f"[resume] {ck} -> continuing at step {start}")

    run_dir = Path(output_dir) / (experiment_name or Path(data).name) / tc.method_name / (
        timestamp or datetime.now().strftime("%Y-%m-%d_%H%M%S"))

    def checkpoint(step: int) -> Path:
        # the random streams of EVERY rank go into the checkpoint (rank 0 writes it).  The gather below is a collective: every
        # rank must reach it for the same step, or the job hangs -- so rank 0's step is broadcast first and a rank that
        # disagrees fails loudly instead
        mine = {"generator": trainer._gen.get_state(), "datamanager_generator": dm._gen.get_state(),
                "train_count": dm.train_count}
        rank_states = [mine]
        if world > 1:
            agreed = [step]
            dist.broadcast_object_list(agreed, src=0)
            if agreed[0] != step:
                raise RuntimeError(f"rank {rank} reached checkpoint({step}) while rank 0 is at checkpoint({agreed[0]}): the ranks "
                                   "must save at the same steps")
            rank_states = [None] * world if rank == 0 else None
            dist.gather_object(mine, rank_states, dst=0)
        if rank != 0:
            return run_dir / "config.yml"
        dp = {k: (str(v) if isinstance(v, Path) else v) for k, v in vars(pc).items()
              if isinstance(v, (int, float, str, bool, Path, type(None)))}
        dp["__class__"] = f"fruit_nerf.data.{type(pc).__module__.rsplit('.', 1)[-1]}.{type(pc).__name__}"
        cfg_path = save_run(run_dir, model.config, dm.cameras.to("cpu"), train_out.scene_box, model.params, step=step,
                            transform=train_out.dataparser_transform.tolist(), scale=train_out.dataparser_scale,
                            method_name=tc.method_name, data=str(Path(data).resolve()), dataparser=dp, trainer_config=tc,
                            optimizers=dict(trainer.state_dict(), rank_states=rank_states),
                            schedulers={g: {"last_epoch": trainer.step, "lr": grp.lr_at(trainer.step)}
                                        for g, grp in trainer.groups.items()},
                            image_filenames=train_out.image_filenames, eval_cameras=eval_out.cameras,
                            eval_image_filenames=eval_out.image_filenames)
        for old in sorted((run_dir / "nerfstudio_models").glob("step-*.ckpt"))[:-1]:
            old.unlink()  # nerfstudio's save_only_latest_checkpoint
        return cfg_path

    t0 = time.perf_counter()
    t_log, cfg_path = t0, run_dir / "config.yml"
    for step in range(start, iters):
        ray_bundle, batch = dm.next_train(step)
        out = trainer.train_iteration(ray_bundle, batch)  # averages the gradients over the ranks when there are several
        if step % log_every == 0 or step == iters - 1:
            ld = {k: float(v) for k, v in out["loss_dict"].items()}
            now = time.perf_counter()
            rate = log_every * dm.config.train_num_rays_per_batch * world / max(now - t_log, 1e-9) if step else 0.0
            t_log = now
            say(f"[{step:6d}] " + " ".join(f"{k} {v:.5f}" for k, v in ld.items()) +
                f" psnr {float(out['metrics_dict']['psnr']):.2f} rays/s {rate:.3g}")
        if save_every and step and step % save_every == 0:
            cfg_path = checkpoint(step)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    cfg_path = checkpoint(max(iters - 1, 0))

    # eval split: get_image_metrics_and_images (fruit_nerf.py:647-700) averaged over the eval images
    done = max(iters - start, 0)
    result = {"config": str(cfg_path), "iterations": iters, "resumed_at": start, "train_seconds": round(seconds, 2),
              "rays_per_sec": done * dm.config.train_num_rays_per_batch * world / max(seconds, 1e-9)}
    if rank == 0 and len(eval_out.image_filenames) > 0:
        model.eval()
        eval_set = FruitDataset(eval_out)
        cams = eval_out.cameras.to(device)
        sums: dict = {}
        for i in range(len(eval_set)):
            rb = cams.generate_rays(i, keep_shape=True)
            rb.camera_indices = torch.zeros_like(rb.camera_indices)  # unseen view: pose tweak / embedding of camera 0
            outputs = model.get_outputs_for_camera_ray_bundle(rb)
            metrics, _ = model.get_image_metrics_and_images(outputs, eval_set.get_data(i))
            for k, v in metrics.items():
                sums[k] = sums.get(k, 0.0) + v
        for k, v in sums.items():
            if v == v:  # lpips is NaN without its pretrained network
                result[f"eval_{k}"] = round(v / len(eval_set), 4)
    if world > 1:
        # data-parallel invariant: every rank applied the same averaged gradients to the same initial parameters
        from cropnerf_amd.distributed import all_reduce_mean

        chk = trainer.flat_params.double().sum().reshape(1)
        mean = all_reduce_mean(chk)
        if abs(float(mean) - float(chk)) > 1e-6 * max(1.0, abs(float(chk))):
            raise RuntimeError(f"rank {rank}: parameters diverged between ranks ({float(chk)!r} vs mean {float(mean)!r})")
        result["ranks"] = world
        dist.barrier()
        dist.destroy_process_group()
    say(json.dumps(result))
    return result


def entrypoint(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("method", choices=["fruit_nerf", "fruit_nerf_big", "fruit_nerf_huge"])
    ap.add_argument("--data", type=Path, required=True)
    ap.add_argument("--output-dir", type=Path, default=Path("outputs"))
    ap.add_argument("--max-num-iterations", type=int, default=None)
    ap.add_argument("--steps-per-save", type=int, default=None)
    ap.add_argument("--downscale-factor", type=int, default=None)
    ap.add_argument("--experiment-name", default=None)
    ap.add_argument("--timestamp", default=None)
    ap.add_argument("--train-split-fraction", type=float, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--load-dir", type=Path, default=None, help="a run's nerfstudio_models directory to resume from")
    ap.add_argument("--implementation", choices=["tcnn", "torch"], default=None,
                    help="module implementation of the field / proposal networks (nerfacto's `implementation`): tcnn = "
                         "tiny-cuda-nn's grid geometry and bias-free MLPs, the reference's default; torch = nerfstudio's "
                         "torch modules.  Default: the method specification's")
    ap.add_argument("--matrix-precision", choices=["fp32", "f16", "split_bf16"], default=None,
                    help="matrix arithmetic of the field in training: fp32 (default, exact), f16 = the reference's mixed-precision "
                         "class (fp16 forward operands, bf16 gradient products, fp32 sums and masters), split_bf16 (forward only)")
    a = ap.parse_args(argv)
    return train(a.method, a.data, a.output_dir, a.max_num_iterations, a.steps_per_save, a.downscale_factor,
                 a.experiment_name, a.timestamp, a.seed, a.log_every, a.train_split_fraction, load_dir=a.load_dir,
                 implementation=a.implementation, matrix_precision=a.matrix_precision)


if __name__ == "__main__":
    entrypoint()
