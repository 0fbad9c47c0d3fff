"""Training step of the fruit_nerf method on the HIP kernels.

What the nerfstudio Trainer does around the reference's model per iteration:
``datamanager.next_train`` -> ``FruitModel.forward`` (training: near 0.05, stratified single-jitter proposal sampling,
per-camera appearance) -> ``get_loss_dict`` (``fruit_nerf/fruit_nerf.py:601-615``: rgb MSE + semantic BCE + interlevel)
-> backward -> Adam with exponential LR decay (``fruit_nerf/fruit_nerf_config.py:45-60``) -> anneal callback
(``fruit_nerf.py:198-232``).

``fruit_nerf_method_big`` / ``_huge`` (other field shapes) train through the shape-generic kernels.  The camera pose refinement (``camera_opt`` group, ``fruit_nerf.py:195``) is trained too: the field / proposal backward
kernels return d loss / d sample position (and the SH-input gradient of the colour branch), ``cn_ray_backward`` reduces
them per ray and ``cn_pose_adjustment_backward`` chains through exp_map_SO3xR3; ``camera_opt_regularizer``
(``fruit_nerf.py:614``) is added by ``cn_pose_regularizer``.  The proposal networks follow the reference's update
schedule (``fruit_nerf.py:144-149``): evaluated without gradient unless ``steps_since_update > update_sched(step) or
step < 10``.  No GradScaler: exact fp32 by default; ``config.matrix_precision = "f16"`` selects the reference's
mixed-precision class for the field (fp16 forward operands, bf16 gradient products, fp32 sums and master parameters:
``FruitModel.train_matrix_precision``), whose bf16 deltas need no loss scale.

An optimiser group is stepped only when the iteration produced a gradient for it: on the iterations where the proposal
networks are evaluated under ``no_grad`` their parameters have ``grad is None`` in the reference, ``torch.optim.Adam``
skips them (moments, per-parameter step count and weights untouched) -- the learning-rate schedules advance every
iteration regardless (nerfstudio steps all schedulers).  ``group_steps`` holds the per-group Adam step counts.

tcnn-layout models (``FruitNerfModelConfig.implementation == "tcnn"``, fp32 tables): several table entries of a dense
level can stand for one tcnn parameter (``cn_tcnn_grid_plan``), so their gradients are folded into the owning entry before
the step and the parameter is copied back into its aliases after it; the biases a tcnn module cannot hold stay zero
(``tcnn_params.frozen_parameter_names``).  The trained model exports to tcnn's own parameter vectors.
"""

from __future__ import annotations

import os

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _lib as L
from .. import ops
from ..rays import RayBundle
from .fruit_nerf import FruitModel
from .schedules import proposal_update_schedule, proposal_weights_anneal


@dataclass
class OptimGroup:
    lr: float = 1e-2
    eps: float = 1e-15
    lr_final: Optional[float] = 1e-4
    max_steps: Optional[int] = 200000
    optimizer: str = "adam"  # "adam" | "radam" (fruit_nerf_config.py: RAdam for the _big / _huge methods)

    def lr_at(self, step: int) -> float:
        """nerfstudio ExponentialDecayScheduler (no warm-up)."""
        if self.lr_final is None or self.max_steps is None:
            return self.lr
        t = min(max(step / self.max_steps, 0.0), 1.0)
        return math.exp(math.log(self.lr) * (1 - t) + math.log(self.lr_final) * t)


def groups_from_spec(optimizers) -> Dict[str, "OptimGroup"]:
    """Optimiser groups of a method specification (``fruit_nerf_config.py``: name -> OptimizerSpec)."""
    return {k: OptimGroup(o.lr, o.eps, o.lr_final, o.max_steps, o.optimizer) for k, o in optimizers.items()}


class FruitTrainer:
    def __init__(self, model: FruitModel, groups: Optional[Dict[str, OptimGroup]] = None, seed: int = 0):
        self.model = model
        if model.config.background_color != "last_sample":
            # cn_train_render_backward is RGBRenderer("last_sample") -- the nerfacto default every reference config keeps
            # (fruit_nerf_config.py never sets background_color); the eval renders take any named colour or triple
            raise NotImplementedError(f"training with background_color={model.config.background_color!r}: the training kernels "
                                      "implement 'last_sample' (the reference's setting); other backgrounds are eval-only")
        # fruit_nerf_method_big / _huge field shapes train through the shape-generic kernels (cn_field_eval +
        # cn_field_backward_general)
        self.general = not model._fused_shape
        # the proposal sampler of the training forward as ONE launch (cn_proposal_sample_train) when the proposal networks
        # have the shapes it is built for; CN_TRAIN_FUSED_SAMPLER=0 composes the materialising calls instead (A/B, tests)
        self.fused_sampler = os.environ.get("CN_TRAIN_FUSED_SAMPLER", "1") != "0"
        # CN_TRAIN_CONCURRENT_BACKWARD=1: field / proposal backward passes on three streams (forward_backward).  Off by default:
        # measured 39.7 vs 40.4 ms at 65 536 rays and 3.06 vs 3.04 ms at 4 096 -- the field backward's workgroup owns a CU's whole
        # LDS, so the kernels share the device CU by CU instead of overlapping on a CU
        self.concurrent_backward = os.environ.get("CN_TRAIN_CONCURRENT_BACKWARD", "0") != "0" and model.device.type == "cuda"
        self._side_streams = [torch.cuda.Stream(device=model.device) for _ in range(2)] if self.concurrent_backward else []
        self.groups = groups or {"proposal_networks": OptimGroup(), "fields": OptimGroup(),
                                 "camera_opt": OptimGroup(1e-3, 1e-15, 1e-4, 5000)}
        self._general_ws = None
        self._exchange = None
        self.force_exchange = False  # run the data-parallel exchange also in a one-rank process group (tests)
        # CN_TRAIN_GRAPH=0: every iteration as eager launches (A/B; default: replay a captured HIP graph where it applies)
        self.use_graph = os.environ.get("CN_TRAIN_GRAPH", "1") != "0"
        self._graphs: Dict[tuple, dict] = {}
        self._g_dev: Dict[int, Tensor] = {}  # per batch size R: Adam scalars of every group | the sampler's jitter, one buffer
        self._g_ring: Dict[int, List[dict]] = {}  # its pinned staging slots
        self._g_ring_next = 0
        self._planes: Dict[tuple, Tensor] = {}  # constant [R, 1] near / far columns of bundles that carry none
        dev = model.device
        # a group that is not listed is frozen (its gradients are still computed for "fields"/"proposal_networks")
        self.train_pose = "camera_opt" in self.groups
        self.trans_l2_penalty, self.rot_l2_penalty = 1e-2, 1e-3  # CameraOptimizerConfig defaults
        self.trainable = [k for k in model.params if self.train_pose or not k.startswith("camera_optimizer.")]
        for k, v in model.params.items():
            if v.dtype != torch.float32:
                raise TypeError(f"{k} is {v.dtype}: training needs float32 parameters (half hash tables are inference-only; "
                                "load the run with hash_table_dtype='float32')")
        self.tcnn = model.config.implementation == "tcnn"
        self._tcnn_tables = []
        self._frozen: List[str] = []
        if self.tcnn:
            from .tcnn_params import frozen_parameter_names

            self._tcnn_tables = [(model.field_spec.grid, "field.mlp_base_grid.hash_table")] + [
                (ps.grid, f"proposal_networks.{i}.encoding.hash_table") for i, ps in enumerate(model.proposal_specs)]
            self._frozen = frozen_parameter_names(model.field_spec, model.proposal_specs)
        # Parameters, gradients and both Adam moments live in four flat buffers with per-tensor views, ordered by
        # optimiser group: data-parallel training all-reduces the gradients in ONE collective (the reference's DDP,
        # fruit_pipeline.py:119-121, made explicit; 78 MB per step for the default field) and the optimiser step is one
        # launch per group instead of one per tensor (29 -> 3).  The model's parameter dict is re-homed onto the flat
        # buffer (same values), so its kernel handles are rebuilt.
        def group_of(k: str) -> str:
            return ("proposal_networks" if k.startswith("proposal_networks.") else
                    "camera_opt" if k.startswith("camera_optimizer.") else "fields")

        self.group_of = group_of
        rank = {"fields": 0, "proposal_networks": 1, "camera_opt": 2}
        keys = sorted(model.params, key=lambda k: rank[group_of(k)])  # stable: keeps the dict order inside a group
        sizes = {k: model.params[k].numel() for k in keys}
        pad4 = lambda n: (n + 3) // 4 * 4  # every tensor starts on a 16-byte boundary (pad elements stay zero)
        total = sum(pad4(n) for n in sizes.values())
        self.flat_params = torch.zeros(total, device=dev)
        self.flat_grads = torch.zeros(total, device=dev)
        self.flat_exp_avg = torch.zeros(total, device=dev)
        self.flat_exp_avg_sq = torch.zeros(total, device=dev)
        self.grads, self.exp_avg, self.exp_avg_sq, self.group_range, off = {}, {}, {}, {}, 0
        for k in keys:
            v = model.params[k]
            view = self.flat_params[off:off + sizes[k]].view_as(v)
            view.copy_(v)
            model.params[k] = view
            self.grads[k] = self.flat_grads[off:off + sizes[k]].view_as(v)
            self.exp_avg[k] = self.flat_exp_avg[off:off + sizes[k]].view_as(v)
            self.exp_avg_sq[k] = self.flat_exp_avg_sq[off:off + sizes[k]].view_as(v)
            g = group_of(k)
            lo, hi = self.group_range.get(g, (off, off))
            self.group_range[g] = (lo, off + pad4(sizes[k]))
            off += pad4(sizes[k])
        model.field = ops.FieldHandle(model.params, model.field_spec)
        model.proposal_networks = [ops.DensityHandle(model.params, i, ps) for i, ps in enumerate(model.proposal_specs)]
        # gradient handles; their scatter scratch (cn_grid.scatter_scratch) is attached by forward_backward, sized for the
        # batch it sees (160-180 MB per grid when sized for any batch, a few MB at the reference's 4 096 rays)
        self.grad_field = ops.FieldHandle(self.grads, model.field_spec)
        self.grad_props = [ops.DensityHandle(self.grads, i, ps) for i, ps in enumerate(model.proposal_specs)]
        self.step = 0
        self.group_steps = {g: 0 for g in self.group_range}  # Adam's per-parameter step count, per group
        self._steps_since_update = 0  # ProposalNetworkSampler state (_steps_since_update, _step)
        self._sampler_step = 0
        self._gen = torch.Generator(device="cpu").manual_seed(seed)
        self.loss_sums = torch.zeros(5, device=dev)
        self._zeroed: Dict[int, Tensor] = {}  # per batch size R, [6 R + 8]: ray-gradient accumulators | loss sums, one fill per iteration
        self._epilogue = None  # cn_train_epilogue's output of the last forward_backward

    # ------------------------------------------------------------------------------------------------------
    def set_anneal(self, step: int) -> None:
        """``set_anneal`` callback (``fruit_nerf.py:206-216``), arXiv 2111.12077 eq. 18."""
        cfg = self.model.config
        if not cfg.use_proposal_weight_anneal:
            return
        self.model.set_anneal(proposal_weights_anneal(step, cfg.proposal_weights_anneal_max_num_iters,
                                                      cfg.proposal_weights_anneal_slope))

    def proposal_update_due(self, step: int) -> bool:
        """``ProposalNetworkSampler``: proposal densities carry gradient only when this is true."""
        cfg = self.model.config
        sched = proposal_update_schedule(step, cfg.proposal_warmup, cfg.proposal_update_every)
        return self._steps_since_update > sched or step < 10

    def forward_backward(self, ray_bundle: RayBundle, batch: Dict[str, Tensor],
                         jitter: Optional[List[Tensor]] = None, update_proposals: bool = True,
                         on_group_ready=None, anneal_dev: Optional[Tensor] = None,
                         jitter_rows: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """One training forward + backward; gradients are ACCUMULATED into ``self.grads``.  ``jitter`` = the three
        [R,1] uniform randoms of the proposal sampler (drawn here when None).  ``update_proposals`` False = the
        reference's ``no_grad`` proposal evaluation (interlevel loss reported, no proposal gradients).
        ``on_group_ready(name)`` is called as soon as every kernel that writes the gradients of an optimiser group has been
        enqueued (``fields`` after the field backward, ``proposal_networks`` after both proposal passes, ``camera_opt`` at the
        end): the data-parallel exchange of that group starts there, under the kernels that follow."""
        ready = on_group_ready or (lambda name: None)
        m, cfg = self.model, self.model.config
        dev = m.device
        rb = ray_bundle.flatten().to(dev)._map(lambda t: t.contiguous())
        R = rb.origins.shape[0]
        if rb.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        cam = rb.camera_indices.reshape(-1).to(torch.int64).contiguous()
        d_raw = rb.directions
        pose = m.params["camera_optimizer.pose_adjustment"]
        o, d = torch.empty_like(rb.origins), torch.empty_like(rb.directions)
        ops.apply_pose_adjustment_to(pose, cam, rb.origins, rb.directions, o, d)  # (out of place: no clones of the raw rays)
        # one zeroed buffer per iteration: the ray-gradient accumulators, the four loss sums and the distortion metric's sum (one
        # fill instead of four)
        zeroed = self._zeroed.get(R)  # (kept per batch size: a captured iteration of another size still reads its own)
        if zeroed is None:
            zeroed = self._zeroed[R] = torch.empty(6 * R + 8, device=dev)
        self.loss_sums = zeroed[6 * R:6 * R + 5]
        zeroed.zero_()
        if self.train_pose:
            d_o, d_d = zeroed[:3 * R].view(R, 3), zeroed[3 * R:6 * R].view(R, 3)
        nears = rb.nears if rb.nears is not None else self._plane(R, cfg.near_plane)
        fars = rb.fars if rb.fars is not None else self._plane(R, cfg.far_plane)
        n_lvl = len(m.proposal_networks)
        handles = [self.grad_field] + list(self.grad_props)
        before = [h._scatter_scratch for h in handles]
        self.grad_field.enable_scatter_scratch(R * int(cfg.num_nerf_samples_per_ray))
        for i, gp in enumerate(self.grad_props):
            gp.enable_scatter_scratch(R * int(cfg.num_proposal_samples_per_ray[i]))
        if any(h._scatter_scratch is not b for h, b in zip(handles, before)):
            # a larger batch re-allocated a scratch: captured iterations of smaller batches hold the OLD buffer's address and
            # layout as kernel arguments -- drop them (they are captured again on their next occurrence, against the new one)
            self._drop_graphs()
        if L.deterministic():
            # the test build: every buffer the training kernels add to with float atomics accumulates through an integer shadow
            # (the gradients, the scatter scratches, this iteration's accumulators and loss sums)
            ops.deterministic_register([self.flat_grads] + list(self._zeroed.values()) + [h._scatter_scratch for h in handles],
                                       owner=self)
        if jitter_rows is not None:  # [levels + 1, R] on the device already (the graph-replayed iteration's static buffer)
            jitter = [jitter_rows[i].reshape(R, 1) for i in range(n_lvl + 1)]
        elif jitter is None:
            # one draw, one host-to-device copy: rows = the samplers' per-ray uniforms, level by level
            jitter_rows = torch.rand(n_lvl + 1, R, generator=self._gen).to(dev)
            jitter = [jitter_rows[i].reshape(R, 1) for i in range(n_lvl + 1)]
        else:
            jitter = [j.to(dev).contiguous() for j in jitter]
        scene = m._scene(True)
        # ---- proposal sampler (bins, intervals and densities of every level are kept for the interlevel loss) --------
        s_prop = [int(v) for v in cfg.num_proposal_samples_per_ray[:n_lvl]]
        if self.fused_sampler and ops.proposal_sample_fused_supported(m.proposal_networks, s_prop, cfg.num_nerf_samples_per_ray):
            # one launch: cn_proposal_sample_train
            ps = ops.proposal_sample_train(m.proposal_networks, scene, o, d, nears, fars, s_prop,
                                           cfg.num_nerf_samples_per_ray, m._anneal if anneal_dev is None else anneal_dev,
                                           jitter_rows if jitter_rows is not None else
                                           torch.cat([j.reshape(1, R) for j in jitter], 0).contiguous())
            levels = ps["levels"]
            bins, eu = ps["spacing_bins"], ps["euclidean_bins"]
            starts, ends = ps["starts"], ps["ends"]
        else:
            # the same, level by level through the materialising calls (any proposal shape)
            levels = []
            sm = ops.sample_spaced(nears, fars, s_prop[0], L.SPACING_PIECEWISE, jitter[0])
            bins = torch.cat([sm["spacing_starts"], sm["spacing_ends"][:, -1:]], -1).contiguous()
            starts, ends = sm["starts"], sm["ends"]
            for lvl in range(n_lvl):
                den = ops.proposal_density(m.proposal_networks[lvl], scene, o, d, starts, ends)
                w = ops.composite(starts, ends, den, want_weights=True, eval_clamp=False)["weights"]
                levels.append({"bins": bins, "starts": starts, "ends": ends, "density": den})
                s_next = s_prop[lvl + 1] if lvl + 1 < n_lvl else cfg.num_nerf_samples_per_ray
                bins, eu = ops.sample_pdf(bins, w, nears, fars, s_next, anneal=m._anneal, u_rand=jitter[lvl + 1])
                starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
        # ---- field forward (fused kernel, per-sample outputs) --------------------------------------------------------
        S = cfg.num_nerf_samples_per_ray
        mp = m.train_matrix_precision()  # fp32, or the reference's mixed-precision class (config.matrix_precision = "f16")
        if self.general:
            fo = ops.field_eval(m.field, scene, o, d, cam, starts, ends, app_mode=L.APP_PER_CAMERA,
                                sh_unit_dir=cfg.sh_input == "unit")
        else:
            opts = ops.render_opts(S, app_mode=L.APP_PER_CAMERA, sh_unit_dir=cfg.sh_input == "unit", eval_clamp=False,
                                   matrix_precision=mp)
            fo = ops.render_samples(m.field, scene, opts, o, d, nears, fars, camera_indices=cam, bins=eu.contiguous())
        # ---- renderer + losses + their backward ------------------------------------------------------------------------
        image = batch["image"].to(dev)[:, :3].to(torch.float32).contiguous()
        mask = batch["fruit_mask"].to(dev).to(torch.float32).reshape(R, 1).contiguous()
        # (with the final level's spacing bins: get_metrics_dict's distortion, fruit_nerf.py:643, rides in the same kernel -- its
        #  sum over rays goes to loss_sums[4], the epilogue divides)
        rb_out = ops.train_render_backward(starts, ends, fo["density"], fo["rgb"], fo["semantics"], image, mask,
                                           cfg.semantic_loss_weight, self.loss_sums, spacing_bins=bins)
        # The three backward passes (field, proposal network 0, proposal network 1) only share read-only inputs, so they run
        # on three streams when self.concurrent_backward: each is bound by the float-atomic request rate of its scatter for
        # part of its time and by matrix / gather work for the rest, and the parts of different kernels overlap on a CU.
        def field_pass(d_o_acc, d_d_acc):
            dpos = torch.empty(R, S, 3, device=dev) if self.train_pose else None
            ddir = torch.empty(R, S, 3, device=dev) if self.train_pose else None
            if self.general:
                self._general_ws = ops.field_backward_general(
                    m.field, self.grad_field, scene, o, d, cam, starts, ends, rb_out["d_density"], rb_out["d_rgb"],
                    rb_out["d_semantics"], app_mode=L.APP_PER_CAMERA, sh_unit_dir=cfg.sh_input == "unit",
                    workspace=self._general_ws, d_positions=dpos, d_directions=ddir)
            else:
                ops.field_backward(m.field, self.grad_field, scene, o, d, cam, starts, ends, rb_out["d_density"],
                                   rb_out["d_rgb"], rb_out["d_semantics"], app_mode=L.APP_PER_CAMERA,
                                   sh_unit_dir=cfg.sh_input == "unit", d_positions=dpos, d_directions=ddir,
                                   matrix_precision=mp)
            if self.train_pose:
                ops.ray_backward(dpos, ddir, starts, ends, d_o_acc, d_d_acc)

        dd_levels = None  # every level's interlevel term in one launch (they share the final bins / weights and one loss sum)

        def proposal_pass(lvl, d_o_acc, d_d_acc):
            lv = levels[lvl]
            dd = dd_levels[lvl] if dd_levels is not None else ops.interlevel_backward(
                bins, rb_out["weights"], lv["bins"], lv["starts"], lv["ends"], lv["density"], cfg.interlevel_loss_mult,
                self.loss_sums[2:3])
            if not update_proposals:
                return
            dpos = torch.empty(R, lv["starts"].shape[1], 3, device=dev) if self.train_pose else None
            ops.proposal_backward(m.proposal_networks[lvl], self.grad_props[lvl], scene, o, d, lv["starts"], lv["ends"],
                                  dd, d_positions=dpos)
            if self.train_pose:
                ops.ray_backward(dpos, None, lv["starts"], lv["ends"], d_o_acc, d_d_acc)

        if self.concurrent_backward and len(levels) <= len(self._side_streams):
            main = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(main)
            partial = []
            for lvl in range(len(levels)):
                side = self._side_streams[lvl]
                side.wait_event(fork)
                with torch.cuda.stream(side):
                    acc = (torch.zeros(R, 3, device=dev), torch.zeros(R, 3, device=dev)) if self.train_pose else (None, None)
                    proposal_pass(lvl, *acc)
                    done = torch.cuda.Event()
                    done.record(side)
                    partial.append((acc, done))
            field_pass(d_o if self.train_pose else None, d_d if self.train_pose else None)
            for acc, done in partial:
                main.wait_event(done)
                if self.train_pose:
                    d_o += acc[0]
                    d_d += acc[1]
            ready("fields")
            if update_proposals:
                ready("proposal_networks")
        else:
            if 1 <= len(levels) <= 4:
                dd_levels = ops.interlevel_backward_levels(bins, rb_out["weights"], levels, cfg.interlevel_loss_mult,
                                                           self.loss_sums[2:3])
            field_pass(d_o if self.train_pose else None, d_d if self.train_pose else None)
            ready("fields")
            for lvl in range(len(levels)):
                proposal_pass(lvl, d_o if self.train_pose else None, d_d if self.train_pose else None)
            if update_proposals:
                ready("proposal_networks")
        if self.train_pose:
            gp = self.grads["camera_optimizer.pose_adjustment"]
            ops.pose_adjustment_backward(pose, cam, d_raw, d_o, d_d, gp)
            ops.pose_regularizer(pose, gp, self.loss_sums[3:4], self.trans_l2_penalty, self.rot_l2_penalty)
            ready("camera_opt")
        self._last_bins, self._last_weights = bins, rb_out["weights"]
        # get_loss_dict + the scalar metrics in one launch (cn_train_epilogue); the dictionary holds views of its output
        ep = self._epilogue = ops.train_epilogue(self.loss_sums, R, S, cfg.semantic_loss_weight, cfg.interlevel_loss_mult,
                                                 pose if self.train_pose else None)
        loss_dict = {"rgb_loss": ep[0], "semantics_loss": ep[1], "interlevel_loss": ep[2]}
        if self.train_pose:
            loss_dict["camera_opt_regularizer"] = ep[3]
        return {"loss_dict": loss_dict, "rgb": rb_out["rgb"], "semantics": rb_out["semantics"],
                "accumulation": rb_out["accumulation"]}

    def _drop_graphs(self) -> None:
        captured = [k for k, st in self._graphs.items() if "graph" in st]
        if not captured:
            return
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the gradient scatter scratch was re-allocated inside a graph capture")
        torch.cuda.synchronize()  # replays in flight still write the old scratch
        for k in captured:
            self._graphs[k] = {"seen": 1}

    def _plane(self, R: int, value: float) -> Tensor:
        key = (R, float(value))
        t = self._planes.get(key)
        if t is None:
            t = self._planes[key] = torch.full((R, 1), float(value), device=self.model.device)
        return t

    def all_reduce_gradients(self, group=None) -> None:
        """Data-parallel step, blocking form: average the whole flat gradient buffer over the ranks in place (each rank
        trained on its own ray batch).  ``train_iteration`` uses the overlapped per-group form (``gradient_exchange``)."""
        from ..distributed import all_reduce_mean_

        all_reduce_mean_(self.flat_grads, group).wait()

    def gradient_exchange(self, group=None, force: bool = False):
        """The per-group, overlapped exchange over this trainer's flat gradient buffer (created on first use)."""
        from ..distributed import GroupedGradientExchange

        if self._exchange is None or self._exchange.group is not group or self._exchange.force != force:
            self._exchange = GroupedGradientExchange(self.flat_grads, self.group_range, group, force)
        return self._exchange

    def _group_stepped(self, g: str, proposals_updated: bool) -> bool:
        return g in self.groups and not (g == "proposal_networks" and not proposals_updated)

    def _adam_group_dev(self, g: str, proposals_updated: bool, hyper_row: Tensor) -> None:
        lo, hi = self.group_range[g]
        if not self._group_stepped(g, proposals_updated):
            self.flat_grads[lo:hi].zero_()
            return
        ops.adam_step_dev(self.flat_params[lo:hi], self.flat_grads[lo:hi], self.flat_exp_avg[lo:hi],
                          self.flat_exp_avg_sq[lo:hi], hyper_row, zero_grad=True)

    def _optimizer_step_dev(self, proposals_updated: bool, hyper: Tensor) -> None:
        """``optimizer_step`` for the captured iteration: every per-step scalar is read from ``hyper`` ([groups, 8] on the
        device, one row per optimiser group in ``group_range`` order, ``hyper[k][7] != 0`` = no step for group k); counters are
        kept by the caller.  The first group (the field, 67 of the 78 MB) keeps its own launch with its scalars in registers;
        the small ones after it (proposal networks, camera poses) share ONE launch."""
        if self.tcnn:
            for spec, key in self._tcnn_tables:
                ops.tcnn_grid_tie_gradients(spec, self.grads[key])
            for k in self._frozen:
                self.grads[k].zero_()
        names = list(self.group_range)
        self._adam_group_dev(names[0], proposals_updated, hyper[0])
        rest = names[1:]
        if len(rest) == 1:
            self._adam_group_dev(rest[0], proposals_updated, hyper[len(names) - 1])
        elif rest:
            lo = self.group_range[rest[0]][0]
            bounds = [0] + [self.group_range[g][1] - lo for g in rest]
            hi = lo + bounds[-1]
            ops.adam_step_groups_dev(self.flat_params[lo:hi], self.flat_grads[lo:hi], self.flat_exp_avg[lo:hi],
                                     self.flat_exp_avg_sq[lo:hi], bounds, hyper[len(names) - len(rest):])
        if self.tcnn:
            for spec, key in self._tcnn_tables:
                ops.tcnn_grid_tie_parameters(spec, self.model.params[key])

    def optimizer_step(self, proposals_updated: bool = True, exchange=None) -> None:
        """One optimiser step of every group that has a gradient.  ``proposals_updated`` False: the proposal networks
        were evaluated without gradient this iteration (``grad is None`` in the reference): their group is not stepped.
        ``exchange``: a ``GroupedGradientExchange`` with collectives in flight -- every group is stepped behind its own."""
        self.step += 1
        if self.tcnn and exchange is not None:
            exchange.wait_all()  # the alias folds below read whole tables of several groups
        if self.tcnn:
            for spec, key in self._tcnn_tables:
                ops.tcnn_grid_tie_gradients(spec, self.grads[key])
            for k in self._frozen:
                self.grads[k].zero_()
        for g, (lo, hi) in self.group_range.items():
            if g not in self.groups or (g == "proposal_networks" and not proposals_updated):
                self.flat_grads[lo:hi].zero_()  # frozen group / no gradient this iteration: nothing moves
                continue
            grp = self.groups[g]
            self.group_steps[g] += 1
            if exchange is not None:
                exchange.wait(g)
            step_fn = {"adam": ops.adam_step, "radam": ops.radam_step}[grp.optimizer]
            step_fn(self.flat_params[lo:hi], self.flat_grads[lo:hi], self.flat_exp_avg[lo:hi],
                    self.flat_exp_avg_sq[lo:hi], self.group_steps[g], grp.lr_at(self.step - 1), eps=grp.eps,
                    zero_grad=True)
        if exchange is not None:
            exchange.wait_all()  # a group that was exchanged but is frozen here
        if self.tcnn:
            for spec, key in self._tcnn_tables:
                ops.tcnn_grid_tie_parameters(spec, self.model.params[key])

    def state_dict(self) -> Dict[str, object]:
        """What a resumed run needs beside the parameters (nerfstudio checkpoints carry "optimizers" too): the Adam
        moments in flat-buffer order, the global and per-group step counts, the proposal sampler's schedule state and the
        jitter generator (this rank's; ``scripts/train.py`` stores one per rank)."""
        return {"step": self.step, "group_steps": dict(self.group_steps),
                "exp_avg": self.flat_exp_avg.detach().cpu(), "exp_avg_sq": self.flat_exp_avg_sq.detach().cpu(),
                "steps_since_update": self._steps_since_update, "sampler_step": self._sampler_step,
                "generator": self._gen.get_state()}

    def load_state_dict(self, state: Dict[str, object], load_generator: bool = True) -> None:
        if tuple(state["exp_avg"].shape) != tuple(self.flat_exp_avg.shape):
            raise ValueError("optimizer state of a different model / trainer configuration")
        self.step = int(state["step"])
        gs = state.get("group_steps") or {}
        self.group_steps = {g: int(gs.get(g, self.step)) for g in self.group_range}
        self.flat_exp_avg.copy_(state["exp_avg"])
        self.flat_exp_avg_sq.copy_(state["exp_avg_sq"])
        self._steps_since_update, self._sampler_step = int(state["steps_since_update"]), int(state["sampler_step"])
        if load_generator:
            self._gen.set_state(state["generator"])

    # ---- the iteration as a replayed HIP graph -----------------------------------------------------------------------------------
    def _graph_eligible(self) -> bool:
        import torch.distributed as dist

        m, cfg = self.model, self.model.config
        n_lvl = len(m.proposal_networks)
        # the captured iteration reads the annealing exponent from device memory, which only the one-launch sampler does: the
        # materialising sampler calls take it as a host float that a capture would freeze at its value of iteration ~2
        fused = self.fused_sampler and ops.proposal_sample_fused_supported(
            m.proposal_networks, [int(v) for v in cfg.num_proposal_samples_per_ray[:n_lvl]], cfg.num_nerf_samples_per_ray)
        return (fused and self.use_graph and not self.general and not self.concurrent_backward and self.model.device.type == "cuda"
                and all(g.optimizer == "adam" for g in self.groups.values())
                and not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_exchange)))

    def _train_iteration_graph(self, ray_bundle: RayBundle, batch: Dict[str, Tensor]) -> Optional[Dict[str, Tensor]]:
        """At the reference's batch size (4 096 rays) an iteration is ~45 launches of 5-700 us: the gaps between them are a tenth
        of it.  The whole iteration -- pose tweak, sampler, field forward, losses, the three backward passes, folds, epilogue,
        the optimiser steps, the distortion metric -- is captured ONCE per (batch geometry, proposal-update yes / no) into a
        HIP graph and replayed; what changes from step to step enters through device memory: the annealing exponent, the Adam
        scalars of every group and the sampler's jitter (ONE host-to-device copy from a pinned slot), and the batch itself
        (read in place when the caller hands over the same tensors as at capture time, e.g. a resident dataset's views;
        otherwise copied into the captured buffers).  The graph is one chain: a branch for the work nothing downstream waits
        for (the distortion metric beside the field backward, the field group's HBM-bound Adam step beside the atomic-bound
        proposal backward) was built and measured -- 1.84-1.87 ms with it, 1.86-1.87 without at 4 096 rays, and the same on two
        streams without a graph -- and removed.  Returns None when the iteration has to run eagerly (first two occurrences of
        a variant: warm-up, then capture).

        The losses and metrics of the returned dictionary are a per-iteration copy; its per-ray tensors (``rgb``, ``semantics``,
        ``accumulation``) are the graph's STATIC outputs, valid until the next ``train_iteration`` of the same variant overwrites
        them (an eager iteration returns fresh tensors) -- ``.clone()`` what has to outlive the call."""
        m, cfg, dev = self.model, self.model.config, self.model.device
        rb = ray_bundle.flatten()
        R = rb.origins.shape[0]
        if rb.camera_indices is None or not rb.origins.is_cuda:
            return None
        n_lvl = len(m.proposal_networks)
        updated = self.proposal_update_due(self._sampler_step)
        key = (R, bool(updated), rb.nears is None)
        st = self._graphs.get(key)
        if st is None:
            self._graphs[key] = {"seen": 1}
            return None  # first occurrence: eager (runs every first-call initialisation)
        # ---- per-step scalars and randoms -> device --------------------------------------------------------------------------------
        # (pinned staging in a ring of four slots, each guarded by an event: the host runs ahead of the GPU, and a slot must not
        #  be rewritten before the copy that reads it has executed)
        ngrp = len(self.group_range)
        nsc = (1 + ngrp) * 8
        if R not in self._g_dev:
            self._g_dev[R] = torch.zeros(nsc + (n_lvl + 1) * R, device=dev)
            self._g_ring[R] = []
            for _ in range(4):
                buf = torch.zeros(nsc + (n_lvl + 1) * R).pin_memory()
                self._g_ring[R].append({"buf": buf, "scalars": buf[:nsc].view(1 + ngrp, 8), "jitter": buf[nsc:].view(n_lvl + 1, R),
                                        "event": None})
        g_dev = self._g_dev[R]
        g_scalars, g_jitter = g_dev[:nsc].view(1 + ngrp, 8), g_dev[nsc:].view(n_lvl + 1, R)
        slot = self._g_ring[R][self._g_ring_next % 4]
        self._g_ring_next += 1
        if slot["event"] is not None:
            slot["event"].synchronize()
        self.set_anneal(self.step)
        slot["scalars"][0, 0] = float(m._anneal)
        steps = dict(self.group_steps)
        for gi, g in enumerate(self.group_range):
            if self._group_stepped(g, updated):
                steps[g] += 1
                grp = self.groups[g]
                ops.adam_hyper(steps[g], grp.lr_at(self.step), eps=grp.eps, out=slot["scalars"][1 + gi])
            else:
                slot["scalars"][1 + gi, 7] = 1.0  # no step for this group: cn_adam_step_groups_dev only zeroes its gradients
        gen_state = self._gen.get_state() if "graph" not in st else None  # (a failed capture hands the draw back)
        torch.rand(n_lvl + 1, R, generator=self._gen, out=slot["jitter"])
        given = {"origins": rb.origins, "directions": rb.directions, "cam": rb.camera_indices, "nears": rb.nears, "fars": rb.fars,
                 "image": batch["image"], "mask": batch["fruit_mask"]}

        def stage_inputs():
            # the captured launches read the trainer's own copies of the batch; a tensor the caller hands over again unchanged
            # (same storage, same version counter: a resident batch) is not copied again.  The STORAGE object is part of the
            # tag and thereby kept alive: a freshly allocated batch can land on the address of a freed one with the same shape
            # and version 0, and an address alone would then pass stale rays for new ones.
            for name, t in given.items():
                if t is None:
                    continue
                tag = (t.untyped_storage(), t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype)
                old = st["seen_inputs"].get(name)
                if old is None or old[0] is not tag[0] or old[1:] != tag[1:]:
                    st["inputs"][name].copy_(t.to(dev).reshape(st["inputs"][name].shape), non_blocking=True)
                    st["seen_inputs"][name] = tag
            g_dev.copy_(slot["buf"], non_blocking=True)
            slot["event"] = torch.cuda.Event()
            slot["event"].record()

        if "graph" not in st:
            # second occurrence: capture
            st["inputs"] = {k: (None if t is None else torch.empty_like(t.to(dev))) for k, t in given.items()}
            st["seen_inputs"] = {}
            stage_inputs()
            ins = st["inputs"]
            srb = RayBundle(ins["origins"], ins["directions"], None, ins["cam"], ins["nears"], ins["fars"])
            sbatch = {"image": ins["image"], "fruit_mask": ins["mask"]}

            def body():
                out = self.forward_backward(srb, sbatch, update_proposals=updated, anneal_dev=g_scalars[0, 0:1],
                                            jitter_rows=g_jitter)
                self._optimizer_step_dev(updated, g_scalars[1:])
                out["metrics_dict"] = self.get_metrics_dict(out)
                return out

            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    st["out"] = body()
                    st["ep"] = self._epilogue  # this variant's static epilogue tensor (another variant captures its own)
            except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly, for good
                import sys

                print(f"[FruitTrainer] HIP graph capture of the training iteration failed ({type(e).__name__}: {e}); running eagerly",
                      file=sys.stderr)
                self.use_graph = False
                torch.cuda.synchronize()
                self._gen.set_state(gen_state)  # the eager iteration that follows draws the same jitter an eager run would
                self._g_ring_next -= 1
                self._graphs[key] = {"seen": 1}
                return None
            st["graph"] = graph
        else:
            stage_inputs()
        st["graph"].replay()
        # the scalars a caller is most likely to keep (losses, metrics: views of the 8-float epilogue) are handed out as a COPY per
        # iteration -- one 32-byte device copy -- so that a list of past iterations' losses stays what it was; the per-ray outputs
        # (rgb, semantics, accumulation) are the graph's static tensors, valid until the next replay of this variant
        ep = st["ep"].clone()
        out = dict(st["out"])
        out["loss_dict"] = {k: ep[i] for i, k in enumerate(("rgb_loss", "semantics_loss", "interlevel_loss")) if k in st["out"]["loss_dict"]}
        if "camera_opt_regularizer" in st["out"]["loss_dict"]:
            out["loss_dict"]["camera_opt_regularizer"] = ep[3]
        md = {"psnr": ep[4], "distortion": ep[7]}
        if self.train_pose:
            md["camera_opt_translation"], md["camera_opt_rotation"] = ep[5], ep[6]
        out["metrics_dict"] = md
        # ---- the host-side counters of optimizer_step / train_iteration ------------------------------------------------------------
        it = self.step
        self.step += 1
        self.group_steps = steps
        if updated:
            self._steps_since_update = 0
        self._sampler_step = it
        self._steps_since_update += 1
        return out

    def train_iteration(self, ray_bundle: RayBundle, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        if self._graph_eligible():
            out = self._train_iteration_graph(ray_bundle, batch)
            if out is not None:
                return out
        self.set_anneal(self.step)
        it = self.step
        updated = self.proposal_update_due(self._sampler_step)
        import torch.distributed as dist

        exchange = None
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_exchange):
            # data-parallel training: each group's gradients are averaged under the kernels that follow their last writer
            # (CN_DP_EXCHANGE=blocking: one all-reduce of the whole buffer after the backward, the round-3 form, for A/B runs)
            if os.environ.get("CN_DP_EXCHANGE", "overlap") == "blocking":
                out = self.forward_backward(ray_bundle, batch, update_proposals=updated)
                self.all_reduce_gradients()
            else:
                exchange = self.gradient_exchange(force=self.force_exchange)
                exchange.begin_iteration()
                # (only the groups this iteration steps: a frozen group's gradients are zeroed, not averaged -- an exchange
                #  nobody waits for would race with that fill)
                out = self.forward_backward(ray_bundle, batch, update_proposals=updated, on_group_ready=lambda g: (
                    exchange.start(g) if self._group_stepped(g, updated) else None))
        else:
            out = self.forward_backward(ray_bundle, batch, update_proposals=updated)
        if updated:
            self._steps_since_update = 0
        self.optimizer_step(proposals_updated=updated, exchange=exchange)
        self._sampler_step = it  # step_cb, an AFTER_TRAIN_ITERATION callback
        self._steps_since_update += 1
        out["metrics_dict"] = self.get_metrics_dict(out)
        return out

    def get_metrics_dict(self, out) -> Dict[str, Tensor]:
        """``get_metrics_dict`` (``fruit_nerf.py:639-645``): PSNR and the distortion metric of the last batch."""
        ep = self._epilogue  # of the forward_backward that produced `out` (the pose norms are those BEFORE the optimiser step,
        md = {"psnr": ep[4],  # as nerfstudio's get_train_loss_dict computes its metrics before the step)
              "distortion": ep[7]}
        if self.train_pose:  # CameraOptimizer.get_metrics_dict (fruit_nerf.py:644)
            md["camera_opt_translation"] = ep[5]
            md["camera_opt_rotation"] = ep[6]
        return md
