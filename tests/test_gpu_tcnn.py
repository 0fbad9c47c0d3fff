"""GPU parity tests of the tcnn-compatible field: the reference's default ``implementation="tcnn"``
(``fruit_nerf/fruit_field.py:95,116-167``) -- tcnn grid geometry (dense coarse levels, +0.5 offset), fp16 parameter
values, bias-free padded MLPs, tcnn's SH sign convention -- imported from a nerfstudio-style state dict
(``cropnerf_amd.fruit_nerf.tcnn_params``) and evaluated by the same HIP kernels, against ``oracle/tcnn.py``.

Tolerance: the oracle and the kernels compute in fp32 on the SAME fp16-rounded parameter values, so the fp32 bars of
``test_gpu_parity.py`` apply (rtol 2e-4, atol 2e-5).
"""

import math

import pytest
import torch

from _helpers import assert_close, dev_params, make_tcnn_scene, oracle_model, product_specs, rays_with_box, to_dev
from oracle import field as OF
from oracle import rays as ORY
from oracle import samplers as OSM
from oracle import tcnn as TC

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from cropnerf_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def scene():
    return make_tcnn_scene(seed=3)


# ------------------------------------------------------------------------------------------------ layout conversion
def test_pack_unpack_round_trip_and_alias_entries(scene, ops):
    fspec, pspecs = product_specs(scene)
    for spec in (fspec.grid, pspecs[0].grid, pspecs[1].grid):
        plan = spec.plan()
        n = 2 * spec.num_packed_entries
        g = torch.Generator().manual_seed(11)
        packed = torch.randn(n, generator=g).cuda()
        table = ops.tcnn_grid_pack(spec, packed, torch.float32)
        assert table.shape == (spec.num_entries, 2)
        back = ops.tcnn_grid_unpack(spec, table)
        assert torch.equal(back, packed)  # every tcnn parameter (padding entries included) survives bit for bit
        # half tables: exactly the fp16 cast of the master copy
        th = ops.tcnn_grid_pack(spec, packed, torch.float16)
        assert torch.equal(th, table.to(torch.float16))
        assert torch.equal(ops.tcnn_grid_pack(spec, packed.to(torch.float16), torch.float16), th)
        # entry by entry against the oracle's index arithmetic on the dense levels (x | y << b | z << 2b), corner
        # index `res` included: the entry holds the parameter tcnn's grid_index gives for that corner
        import numpy as np

        o = TC.TcnnGridSpec(spec.num_levels, spec.min_res, spec.max_res, spec.log2_hashmap_size)
        offs, ress = o.offset_table(), o.resolutions()
        tcpu, pcpu = table.cpu(), packed.cpu().view(-1, 2)
        for l in range(spec.num_levels):
            b = int(plan.level_bits[l])
            res, size = ress[l], offs[l + 1] - offs[l]
            lo = int(plan.level_offset[l])
            if b == 0:
                assert torch.equal(tcpu[lo:lo + size], pcpu[offs[l]:offs[l + 1]])
                continue
            rng = np.random.default_rng(l)
            c = rng.integers(0, res + 1, size=(4000, 3)).astype(np.uint32)
            c[:200] = res  # corners on the upper faces / edges / the far corner
            c[:200, rng.integers(0, 3, 200)] = rng.integers(0, res + 1, 200).astype(np.uint32)
            idx = TC.grid_index(c, res, size) + offs[l]
            mine = lo + (c[:, 0].astype(np.int64) | (c[:, 1].astype(np.int64) << b) | (c[:, 2].astype(np.int64) << (2 * b)))
            assert torch.equal(tcpu[torch.from_numpy(mine)], pcpu[torch.from_numpy(idx)]), f"level {l}"
        # tying: parameters already agree after a pack; a gradient placed on an alias moves to the owner
        t2 = table.clone()
        ops.tcnn_grid_tie_parameters(spec, t2)
        assert torch.equal(t2, table)
        b0, res0 = int(plan.level_bits[0]), int(plan.resolution[0])
        grad = torch.zeros_like(table)
        alias = res0 | (3 << b0) | (5 << (2 * b0))  # (res, 3, 5) stands for (0, 4, 5)
        owner = 0 | (4 << b0) | (5 << (2 * b0))
        grad[alias] = torch.tensor([1.5, -2.0]).cuda()
        grad[owner] = torch.tensor([0.25, 0.5]).cuda()
        ops.tcnn_grid_tie_gradients(spec, grad)
        assert grad[alias].abs().sum().item() == 0
        assert torch.equal(grad[owner].cpu(), torch.tensor([1.75, -1.5]))
        assert abs(grad.sum().item() - 0.25) < 1e-6


# ------------------------------------------------------------------------------------------------ field kernels
@pytest.fixture(scope="module")
def rounded_scene():
    """A tcnn scene whose grid parameters are fp16-representable: fp32 and fp16 device tables then hold the same values
    and one oracle evaluation serves both."""
    sc = make_tcnn_scene(seed=3)
    for k, v in sc.params.items():
        if k.endswith("tcnn_encoding.params"):
            sc.params[k] = v.to(torch.float16).to(torch.float32)
    return sc


@pytest.fixture(scope="module", params=["float16", "float32"])
def rhandles(request, rounded_scene, ops):
    dtype = getattr(torch, request.param)
    fspec, pspecs = product_specs(rounded_scene)
    dp = dev_params(rounded_scene, table_dtype=dtype)
    fh = ops.FieldHandle(dp, fspec)
    dh = [ops.DensityHandle(dp, i, ps) for i, ps in enumerate(pspecs)]
    assert dp["field.mlp_base_grid.hash_table"].dtype == dtype
    return dp, fh, dh


@pytest.mark.parametrize("contraction", [True, False])
def test_proposal_density(rounded_scene, ops, rhandles, contraction):
    scene = rounded_scene
    dp, fh, dh = rhandles
    rb = rays_with_box(scene, 1, 300)
    rs = OSM.spaced_sampler(rb, 40, "uniform")
    sc = ops.scene_struct(scene.aabb, contraction)
    for lvl in range(2):
        ref = OF.proposal_density(rs.positions(), scene.params, lvl, scene.pspecs[lvl], scene.aabb, contraction)
        out = ops.proposal_density(dh[lvl], sc, to_dev(rb.origins), to_dev(rb.directions), to_dev(rs.starts[..., 0]),
                                   to_dev(rs.ends[..., 0]))
        assert_close(out, ref[..., 0], RTOL, ATOL, f"proposal density {lvl}")


@pytest.mark.parametrize("contraction,mode", [(True, "test"), (False, "inference"), (True, "train_app")])
@pytest.mark.parametrize("impl", ["regw", "mfma", "scalar"])
def test_field_eval(rounded_scene, ops, rhandles, contraction, mode, impl, monkeypatch):
    from cropnerf_amd import _lib as L

    monkeypatch.setenv("CN_FIELD_EVAL_IMPL", impl)
    scene = rounded_scene
    dp, fh, dh = rhandles
    rb = rays_with_box(scene, 2, 150)
    g = torch.Generator().manual_seed(7)
    rb.camera_indices = torch.randint(0, scene.c2w.shape[0], (len(rb), 1), generator=g)
    rs = OSM.spaced_sampler(rb, 33, "uniform")
    training = mode == "train_app"
    ref = OF.field_forward(rs.positions(), rb.directions, rb.camera_indices, scene.params, scene.fspec, scene.aabb,
                           contraction, "inference" if mode == "inference" else "test", training=training)
    sc = ops.scene_struct(scene.aabb, contraction)
    out = ops.field_eval(fh, sc, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.camera_indices[:, 0]),
                         to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]),
                         app_mode=L.APP_PER_CAMERA if training else L.APP_MEAN)
    assert_close(out["density"], ref["density"][..., 0], RTOL, ATOL, "density")
    assert_close(out["semantics"], ref["semantics"][..., 0], RTOL, ATOL, "semantics")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")


def test_samples_in_the_outer_half_cell_read_the_wrapped_entries(rounded_scene, ops, rhandles):
    """Positions within half a coarse cell of the upper box faces have the corner index `res` on the dense levels --
    tcnn wraps into the next row of its linear array; the packed table reproduces that (alias entries)."""
    scene = rounded_scene
    dp, fh, dh = rhandles
    g = torch.Generator().manual_seed(21)
    R, S = 256, 8
    # AABB-normalised coordinates in (0.97, 1): level 0 (scale 15) has pos in (15.05, 15.5) -> corner 16 = res
    target = 0.97 + 0.02 * torch.rand(R, S, 3, generator=g)
    lo, hi = scene.aabb[0], scene.aabb[1]
    world = lo + target * (hi - lo)
    # one ray per row through its first sample; samples are placed by (start + end) / 2 = t
    origins = world[:, 0, :].clone()
    directions = torch.nn.functional.normalize(torch.tensor([[1e-3, 2e-3, 1e-3]]).expand(R, 3), dim=-1).contiguous()
    starts = (torch.arange(S, dtype=torch.float32) * 1e-4).expand(R, S).contiguous()
    ends = starts + 1e-4
    pos = origins[:, None, :] + directions[:, None, :] * ((starts + ends) / 2)[..., None]
    q, sel = OF.normalized_positions(pos, scene.aabb, False)
    assert bool(sel.all()) and float(q.min()) > 0.96
    ref = OF.field_forward(pos, directions, None, scene.params, scene.fspec, scene.aabb, False, "inference")
    sc = ops.scene_struct(scene.aabb, False)
    out = ops.field_eval(fh, sc, to_dev(origins), to_dev(directions), None, to_dev(starts), to_dev(ends))
    assert_close(out["density"], ref["density"][..., 0], RTOL, ATOL, "density at the upper faces")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb at the upper faces")
    for lvl in range(2):
        refd = OF.proposal_density(pos, scene.params, lvl, scene.pspecs[lvl], scene.aabb, False)
        outd = ops.proposal_density(dh[lvl], sc, to_dev(origins), to_dev(directions), to_dev(starts), to_dev(ends))
        assert_close(outd, refd[..., 0], RTOL, ATOL, f"proposal density {lvl} at the upper faces")


# ------------------------------------------------------------------------------------------------ fused renderers
def _fused_vs_oracle(scene, ops, fh, S, contraction, n_rays, cam, **opt_kw):
    rb = rays_with_box(scene, cam, n_rays)
    m = oracle_model(scene, "inference", disable_scene_contraction=not contraction)
    m.uniform_samples = S
    ref = m.forward(rb)
    sc = ops.scene_struct(scene.aabb, contraction)
    opts = ops.render_opts(S, **opt_kw)
    out = ops.render_rays(fh, sc, opts, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars),
                          want_weights=True)
    return ref, out


def _depth_match(dev_depth, ref_depth, frac=0.995):
    ok = (dev_depth.cpu() - ref_depth).abs() <= 1e-5 + 1e-5 * ref_depth.abs()
    assert ok.float().mean().item() >= frac


@pytest.mark.parametrize("split", ["0", "2"])  # single-wave kernel / producer-consumer kernel
@pytest.mark.parametrize("S,contraction", [(192, False), (64, True), (100, False)])
def test_render_rays(rounded_scene, ops, rhandles, S, contraction, split, monkeypatch):
    monkeypatch.setenv("CN_FUSED_SPLIT", split)
    dp, fh, dh = rhandles
    ref, out = _fused_vs_oracle(rounded_scene, ops, fh, S, contraction, 600, 0)
    assert_close(out["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    assert_close(out["semantics"], ref["semantics"], RTOL, 5e-5, "semantics")
    _depth_match(out["depth"], ref["depth"])


def test_render_rays_split_bf16_option(rounded_scene, ops, rhandles, monkeypatch):
    from cropnerf_amd import _lib as L

    monkeypatch.setenv("CN_FUSED_SPLIT", "2")
    dp, fh, dh = rhandles
    ref, out = _fused_vs_oracle(rounded_scene, ops, fh, 96, False, 600, 1, matrix_precision=L.MATRIX_SPLIT_BF16)
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb (split bf16)")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation (split bf16)")


@pytest.mark.parametrize("split", ["0", "2"])
def test_render_samples_export_mode(rounded_scene, ops, rhandles, split, monkeypatch):
    monkeypatch.setenv("CN_FUSED_SPLIT", split)
    scene = rounded_scene
    dp, fh, dh = rhandles
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = ORY.surface_points(ORY.corners_of_aabb(aabb), 12)
    rb = ORY.ortho_rays(pts, plane, 100, 1)
    S = 150
    m = oracle_model(scene, "export")
    m.setup_inference(True, S)
    ref = m.forward(rb)
    sc = ops.scene_struct(scene.aabb, False)
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars))
    out = ops.render_samples(fh, sc, ops.render_opts(S), o, d, n, f)
    assert_close(out["density"], ref["density"], RTOL, ATOL, "density")
    assert_close(out["semantics"], ref["semantics"], RTOL, ATOL, "semantics")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    clear = (ref["semantics"] - math.log(9.0)).abs() > 1e-3
    assert torch.equal(out["semantics_colormap"].cpu()[clear], ref["semantics_colormap"][clear])


def test_proposal_sample_and_full_forward(rounded_scene, ops, rhandles):
    scene = rounded_scene
    dp, fh, dh = rhandles
    rb = ORY.image_rays(scene.c2w, scene.intr, 5, scene.height, scene.width).slice(0, 500)
    m = oracle_model(scene, "test")
    ref = m.forward(rb)
    o, d = to_dev(rb.origins).clone(), to_dev(rb.directions).clone()
    ops.apply_pose_adjustment(dp["camera_optimizer.pose_adjustment"], to_dev(rb.camera_indices[:, 0]), o, d)
    R = len(rb)
    nears = torch.zeros(R, 1, device="cuda")
    fars = torch.full((R, 1), 1000.0, device="cuda")
    sc = ops.scene_struct(scene.aabb, True)
    ps = ops.proposal_sample(dh, sc, o, d, nears, fars, (256, 96), 48)
    ref_bins = torch.cat([ref["_starts"][..., 0], ref["_ends"][:, -1:, 0]], -1)
    assert_close(ps["euclidean_bins"], ref_bins, 2e-3, 1e-4, "final euclidean bins", frac_ok=0.999)
    opts = ops.render_opts(48)
    out2 = ops.render_rays(fh, sc, opts, o, d, nears, fars, bins=to_dev(ref_bins), want_weights=True)
    assert_close(out2["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights (oracle bins)")
    assert_close(out2["rgb"], ref["rgb"], RTOL, ATOL, "rgb (oracle bins)")
    assert_close(out2["semantics"], ref["semantics"], RTOL, 5e-5, "semantics (oracle bins)")


def test_unrounded_master_copy_in_fp16_tables_matches_the_oracle(scene, ops):
    """The real import path: fp32 master values (not fp16-representable) -> fp16 tables; the oracle rounds the same
    way (tcnn casts its parameters to half for every forward)."""
    fspec, pspecs = product_specs(scene)
    dp = dev_params(scene)  # fp16 tables by default
    assert dp["field.mlp_base_grid.hash_table"].dtype == torch.float16
    fh = ops.FieldHandle(dp, fspec)
    ref, out = _fused_vs_oracle(scene, ops, fh, 128, False, 500, 2)
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation")
    assert_close(out["semantics"], ref["semantics"], RTOL, 5e-5, "semantics")


def test_torch_layout_with_half_table(ops):
    """The table dtype is independent of the layout: a torch-layout grid stored as half2 entries."""
    from _helpers import make_scene

    sc_ = make_scene(seed=5, log2_T=15, prop_log2_T=12)
    sc_.params["field.mlp_base_grid.hash_table"] = sc_.params["field.mlp_base_grid.hash_table"].to(torch.float16).to(torch.float32)
    fspec, _ = product_specs(sc_)
    dp = dev_params(sc_)
    dp["field.mlp_base_grid.hash_table"] = dp["field.mlp_base_grid.hash_table"].to(torch.float16)
    fh = ops.FieldHandle(dp, fspec)
    rb = rays_with_box(sc_, 0, 400)
    m = oracle_model(sc_, "inference", disable_scene_contraction=True)
    m.uniform_samples = 96
    ref = m.forward(rb)
    out = ops.render_rays(fh, ops.scene_struct(sc_.aabb, False), ops.render_opts(96), to_dev(rb.origins),
                          to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars))
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb (torch layout, half table)")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation")


def test_backward_rejects_half_tables(rounded_scene, ops):
    from cropnerf_amd._lib import CropNerfHipError

    scene = rounded_scene
    fspec, pspecs = product_specs(scene)
    dp = dev_params(scene, table_dtype=torch.float16)
    dh = ops.DensityHandle(dp, 0, pspecs[0])
    grads = {k: torch.zeros_like(v) for k, v in dp.items()}
    gh = ops.DensityHandle(grads, 0, pspecs[0])
    rb = rays_with_box(scene, 0, 64)
    rs = OSM.spaced_sampler(rb, 16, "uniform")
    with pytest.raises(CropNerfHipError, match="fp32 hash table"):
        ops.proposal_backward(dh, gh, ops.scene_struct(scene.aabb, True), to_dev(rb.origins), to_dev(rb.directions),
                              to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]),
                              torch.ones(64, 16, device="cuda"))


# ------------------------------------------------------------------------------------------------ run directories
def _write_reference_style_run(tmp_path, sc):
    """A run directory as the reference's ``ns-train fruit_nerf`` leaves it: config.yml (TrainerConfig dump) and a
    checkpoint whose ``pipeline`` holds the tcnn-packed tensors under ``_model.``; plus the camera side file (there is no
    capture on disk to re-parse)."""
    import json

    from cropnerf_amd.config import FruitNerfModelConfig
    from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO
    from cropnerf_amd.fruit_nerf.checkpoint import _cameras_dict
    from cropnerf_amd.rays import Cameras

    run = tmp_path / "outputs" / "plant_1" / "fruit_nerf" / "2024-05-01_120000"
    (run / "nerfstudio_models").mkdir(parents=True)
    pl = [{"hidden_dim": 16, "log2_hashmap_size": p.grid.log2_hashmap_size, "num_levels": 5, "max_res": p.grid.max_res,
           "use_linear": False} for p in sc.pspecs]
    mc = FruitNerfModelConfig(log2_hashmap_size=sc.fspec.grid.log2_hashmap_size, proposal_net_args_list=pl)
    NIO.write_config_yml(run / "config.yml", method_name="fruit_nerf", model_config=mc, data=None,
                         output_dir=str(tmp_path / "outputs"), experiment_name="plant_1", timestamp=run.name,
                         max_num_iterations=30000, steps_per_save=2000, mixed_precision=True,
                         train_num_rays_per_batch=4096, eval_num_rays_per_batch=4096)
    state = dict(sc.params)
    state["lpips.net.scaling_layer.shift"] = torch.zeros(1, 3, 1, 1)  # a module without a counterpart here: ignored
    NIO.save_checkpoint(run / "nerfstudio_models" / "step-000029999.ckpt", 29999, state,
                        buffers=NIO.field_buffers(mc, sc.aabb))
    cams = Cameras(sc.c2w, sc.intr[:, 0], sc.intr[:, 1], sc.intr[:, 2], sc.intr[:, 3], sc.height, sc.width)
    (run / "cameras.json").write_text(json.dumps(_cameras_dict(cams)))
    (run / "dataparser_transforms.json").write_text(json.dumps(
        {"transform": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], "scale": 0.5}))
    return run


def test_eval_setup_and_exporter_cli_load_a_tcnn_run_directory(tmp_path, ops):
    """scripts/exporter.py:87 / scripts/semantic_projection.py:139-143: ``eval_setup`` on a reference-style run whose
    checkpoint is tcnn-packed -> the model renders what the tcnn oracle renders; the exporter CLI runs on it; and
    ``save_run`` writes the same tcnn vectors back."""
    from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO
    from cropnerf_amd.fruit_nerf.checkpoint import eval_setup, save_run
    from cropnerf_amd.fruit_nerf.scripts import exporter

    sc = make_tcnn_scene(seed=4, log2_T=16, num_images=3, height=20, width=20, focal=28.0, prop_log2_T=13, grid_scale=0.5)
    run = _write_reference_style_run(tmp_path, sc)
    cfg, pipe, ck, step = eval_setup(run / "config.yml", test_mode="inference")
    m = pipe.model
    assert step == 29999 and ck.name == "step-000029999.ckpt"
    assert m.config.implementation == "tcnn" and m.params["field.mlp_base_grid.hash_table"].dtype == torch.float16
    assert m.field_spec.grid.layout == "tcnn" and m.num_train_data == 3
    # a tcnn-packed checkpoint of a mixed_precision run renders in tiny-cuda-nn's arithmetic class unless told otherwise
    assert m.config.matrix_precision == "f16"
    _, pipe32, _, _ = eval_setup(run / "config.yml", test_mode="inference", matrix_precision="fp32")
    assert pipe32.model.config.matrix_precision == "fp32"
    # full-image render of camera 1 against the tcnn oracle (inference mode: proposal sampler + 48 field samples)
    rb = ORY.image_rays(sc.c2w, sc.intr, 1, sc.height, sc.width)
    ref = oracle_model(sc, "inference").render_rays(rb)
    from cropnerf_amd.rays import RayBundle

    out = m.get_outputs_for_camera_ray_bundle(pipe.datamanager.cameras.to("cuda").generate_rays(1, keep_shape=True))
    assert_close(out["rgb"].reshape(-1, 3), ref["rgb"], 2e-3, 2e-3, "rgb of a loaded tcnn run", frac_ok=0.99)
    assert_close(out["accumulation"].reshape(-1, 1), ref["accumulation"], 2e-3, 2e-3, "accumulation", frac_ok=0.99)
    out32 = pipe32.model.get_outputs_for_camera_ray_bundle(pipe32.datamanager.cameras.to("cuda").generate_rays(1, keep_shape=True))
    assert_close(out32["rgb"].reshape(-1, 3), ref["rgb"], 2e-3, 2e-3, "rgb of the same run in exact fp32", frac_ok=0.99)
    assert not torch.equal(out32["rgb"], out["rgb"])
    # the dense exporter CLI on the same run
    outdir = tmp_path / "export"
    exporter.entrypoint(["semantic-pointcloud", "--load-config", str(run / "config.yml"), "--output-dir", str(outdir),
                         "--num-points-per-side", "12", "--num-rays-per-batch", "50"])
    assert sorted(p.name for p in outdir.rglob("*.ply")) == ["density.ply", "semantic.ply", "semantic_colormap.ply"]
    # and back: a run written from this model holds the (fp16-rounded) tcnn vectors the checkpoint had
    cfg2 = save_run(tmp_path / "o2" / "plant_1" / "fruit_nerf" / "t", m.config, pipe.datamanager.cameras.to("cpu"),
                    m.scene_box, m.params, step=5)
    _, state2, loaded2 = NIO.load_checkpoint(NIO.latest_checkpoint(cfg2.parent / "nerfstudio_models"))
    assert set(loaded2) >= {"step", "pipeline", "optimizers", "schedulers", "scalers"}
    for k in ("field.mlp_base_grid.tcnn_encoding.params", "field.mlp_base_mlp.tcnn_encoding.params",
              "proposal_networks.1.mlp_base.tcnn_encoding.params"):
        want = sc.params[k].to(torch.float16).to(torch.float32)
        assert state2[k].shape == want.shape
        if "grid" in k:
            assert torch.equal(state2[k], want), k
    # MLP vectors: every value tcnn reads is back (padded output rows are written as zeros)
    base = TC.mlp_matrices(state2["field.mlp_base_mlp.tcnn_encoding.params"], 32, 16, 64, 1)
    base0 = TC.mlp_matrices(sc.params["field.mlp_base_mlp.tcnn_encoding.params"].to(torch.float16).to(torch.float32), 32, 16, 64, 1)
    assert torch.equal(base[0], base0[0]) and torch.equal(base[1], base0[1])
    head = TC.mlp_matrices(state2["field.mlp_head.tcnn_encoding.params"], 63, 3, 64, 2)
    head0 = TC.mlp_matrices(sc.params["field.mlp_head.tcnn_encoding.params"].to(torch.float16).to(torch.float32), 63, 3, 64, 2)
    assert torch.equal(head[0], head0[0]) and torch.equal(head[2][:3], head0[2][:3])


# ------------------------------------------------------------------------------------------------ training
def _tcnn_train_setup(seed=5, R=96):
    sc = make_tcnn_scene(seed=seed, log2_T=12, num_images=4, height=20, width=20, focal=28.0, prop_log2_T=10)
    # colours that vary along a ray: with near-constant colours the density gradient is a small difference of large terms
    # and fp32 round-off alone moves it by ~5e-3 (measured: 4.9e-3 at x1, 3.4e-4 at x4 -- conditioning, not arithmetic)
    sc.params["field.mlp_head.tcnn_encoding.params"] = sc.params["field.mlp_head.tcnn_encoding.params"] * 4.0
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack([torch.randint(0, 4, (R,), generator=g), torch.randint(0, 20, (R,), generator=g),
                       torch.randint(0, 20, (R,), generator=g)], -1)
    jitter = [torch.rand(R, 1, generator=g) for _ in range(3)]
    image = torch.rand(R, 3, generator=g)
    mask = (torch.rand(R, 1, generator=g) > 0.5).float()
    return sc, idx, jitter, image, mask


def _tcnn_hip_model(sc, S_PROP=(64, 32), S_FINAL=16):
    from cropnerf_amd.config import FruitNerfModelConfig
    from cropnerf_amd.fruit_nerf import tcnn_params
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
    from cropnerf_amd.rays import SceneBox

    pl = [{"hidden_dim": 16, "log2_hashmap_size": p.grid.log2_hashmap_size, "num_levels": 5, "max_res": p.grid.max_res}
          for p in sc.pspecs]
    cfg = FruitNerfModelConfig(log2_hashmap_size=sc.fspec.grid.log2_hashmap_size, proposal_net_args_list=pl,
                               num_proposal_samples_per_ray=S_PROP, num_nerf_samples_per_ray=S_FINAL, implementation="tcnn")
    fspec, pspecs = product_specs(sc)
    params = tcnn_params.from_tcnn_state_dict(sc.params, fspec, pspecs, "cuda", torch.float32)  # fp32 masters
    return FruitModel(cfg, SceneBox(sc.aabb), num_train_data=sc.c2w.shape[0], metadata={"semantics": Semantics()},
                      device="cuda", test_mode="val", params=params)


def test_tcnn_training_gradients_match_autograd_on_the_tcnn_oracle():
    """Backward kernels on a tcnn-layout model (fp32 master tables): the gradient of every tcnn parameter vector --
    this library's table gradient with the alias entries folded (cn_tcnn_grid_tie_gradients) and unpacked, the MLP
    gradients mapped back into tcnn's matrices -- against torch autograd through ``oracle/tcnn.py``."""
    from cropnerf_amd.fruit_nerf import tcnn_params as TP
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
    from cropnerf_amd.rays import Cameras
    from oracle import losses as OL

    S_PROP, S_FINAL = (64, 32), 16
    sc, idx, jitter, image, mask = _tcnn_train_setup()
    # the oracle evaluates at the fp32 master values too (no fp16 rounding: training keeps masters)
    import oracle.tcnn as TCM

    params = {k: v.clone().requires_grad_(True) for k, v in sc.params.items()}
    rb = ORY.pinhole_rays(sc.c2w, sc.intr, idx[:, 0], idx[:, 1], idx[:, 2])
    real = TCM._as_compute
    TCM._as_compute = lambda p, half: p.to(torch.float32)
    try:
        out_ref = OL.train_forward(rb, params, sc.fspec, sc.pspecs, sc.aabb, S_PROP, S_FINAL, jitter)
        ld = OL.loss_dict(out_ref, image, mask)
        ld["camera_opt_regularizer"] = OL.camera_opt_regularizer(params["camera_optimizer.pose_adjustment"])
        sum(ld.values()).backward()
    finally:
        TCM._as_compute = real
    model = _tcnn_hip_model(sc, S_PROP, S_FINAL)
    model.training = True
    tr = FruitTrainer(model)
    assert tr.tcnn
    cams = Cameras(sc.c2w, sc.intr[:, 0], sc.intr[:, 1], sc.intr[:, 2], sc.intr[:, 3], sc.height, sc.width).to("cuda")
    out = tr.forward_backward(cams.generate_rays(idx.cuda()), {"image": image, "fruit_mask": mask}, jitter=jitter)
    for k, v in ld.items():
        assert abs(float(out["loss_dict"][k]) - float(v)) <= 2e-4 * abs(float(v)) + 1e-7, k
    assert_close(out["rgb"], out_ref["rgb"].detach(), RTOL, ATOL, "train rgb")
    # fold the alias gradients, then express everything as gradients of tcnn's own vectors
    from cropnerf_amd import ops as O

    for spec, key in tr._tcnn_tables:
        O.tcnn_grid_tie_gradients(spec, tr.grads[key])
    for name in tr._frozen:  # biases a tcnn module cannot hold: their gradient is discarded by the optimiser step
        tr.grads[name].zero_()
    got = TP.to_tcnn_state_dict({k: v for k, v in tr.grads.items()}, model.field_spec, model.proposal_specs)
    worst = {}
    for k, p in params.items():
        if p.grad is None:
            continue
        g_ref, g = p.grad, got[k]
        if k.endswith("tcnn_encoding.params") and "grid" not in k and "mlp_base." not in k:
            # compare what tcnn reads: the padded output rows never receive gradient on either side; padded input
            # columns beyond the first hold the same gradient as the first in tcnn, zero here
            if "mlp_semantics" in k:
                a, b = TC.mlp_matrices(g, 15, 64, 64, 1), TC.mlp_matrices(g_ref, 15, 64, 64, 1)
            elif "mlp_head" in k:
                a, b = TC.mlp_matrices(g, 63, 3, 64, 2), TC.mlp_matrices(g_ref, 63, 3, 64, 2)
            else:
                a, b = TC.mlp_matrices(g, 32, 16, 64, 1), TC.mlp_matrices(g_ref, 32, 16, 64, 1)
            g, g_ref = torch.cat([x.reshape(-1) for x in a]), torch.cat([x.reshape(-1) for x in b])
        worst[k] = float((g - g_ref).norm() / (g_ref.norm() + 1e-12))
    bad = {k: v for k, v in worst.items() if v > (1e-2 if k.startswith("camera_optimizer") else 3e-3)}
    assert not bad, f"relative gradient error too large: {bad} (all: {worst})"
    assert len(worst) >= 8


def test_mixed_precision_training_gradients_against_the_half_activation_oracle():
    """``matrix_precision="f16"`` in TRAINING -- the reference's own class (``mixed_precision=True`` on tiny-cuda-nn's fp16
    modules, ``fruit_nerf_config.py:35``, ``fruit_field.py:95,125-167``): the field's forward with fp16 operands
    (``cn_render_samples``), its backward with the fp16 forward recompute and bf16 gradient products
    (``cn_field_backward_mp``), fp32 sums and fp32 master gradients.  Checked against autograd through ``oracle/tcnn.py`` with
    ``tcnn_half_activations=True`` on the FIELD (fp16 parameters, every encoding accumulation and layer input rounded to fp16,
    straight-through gradients in float32); the proposal networks stay fp32 on both sides.

    Stated bars (relative L2 per parameter tensor): losses 2e-3; field gradients 2e-2 -- bf16 operands carry 8 mantissa bits
    (4e-3 per product, averaged down by the 16-64-term sums but compounded over up to four layers), and the kernel rounds the
    interpolated feature once where tcnn rounds after every corner; the same kernels in exact fp32 sit at 3e-3 against the
    unrounded oracle (the test above).  Measured on the MI355X: hash table 1.1e-2, base MLP 3.9e-3, the rest <= 2.4e-3.  The
    exact-fp32 kernels differentiate a DIFFERENT function (the unrounded forward): the mixed-mode table gradient is 2.4e-2 from
    theirs, i.e. closer to the half-activation oracle than to the fp32 run -- asserted, so that a mixed mode which only
    perturbed the fp32 gradient would fail."""
    import dataclasses

    from cropnerf_amd import _lib as L
    from cropnerf_amd.fruit_nerf import tcnn_params as TP
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
    from cropnerf_amd.rays import Cameras
    from oracle import losses as OL

    S_PROP, S_FINAL = (64, 32), 16
    sc, idx, jitter, image, mask = _tcnn_train_setup(seed=7, R=128)
    # masters that are fp16 values already (a checkpoint of the reference holds such): the oracle's fp16 parameter copy is then
    # the master itself, and what is compared is the arithmetic, not a second rounding of the parameters
    for k in list(sc.params):
        if k.endswith("tcnn_encoding.params"):
            sc.params[k] = sc.params[k].to(torch.float16).to(torch.float32)
    half_f = dataclasses.replace(sc.fspec, tcnn_half_activations=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sc.params.items()}
    rb = ORY.pinhole_rays(sc.c2w, sc.intr, idx[:, 0], idx[:, 1], idx[:, 2])
    out_ref = OL.train_forward(rb, params, half_f, sc.pspecs, sc.aabb, S_PROP, S_FINAL, jitter)
    ld = OL.loss_dict(out_ref, image, mask)
    ld["camera_opt_regularizer"] = OL.camera_opt_regularizer(params["camera_optimizer.pose_adjustment"])
    sum(ld.values()).backward()
    cams = Cameras(sc.c2w, sc.intr[:, 0], sc.intr[:, 1], sc.intr[:, 2], sc.intr[:, 3], sc.height, sc.width).to("cuda")
    from cropnerf_amd import ops as O

    def run(mode):
        model = _tcnn_hip_model(sc, S_PROP, S_FINAL)
        model.config.matrix_precision = mode
        model.training = True
        assert model.train_matrix_precision() == (L.MATRIX_F16 if mode == "f16" else L.MATRIX_FP32)
        tr = FruitTrainer(model)
        out = tr.forward_backward(cams.generate_rays(idx.cuda()), {"image": image, "fruit_mask": mask}, jitter=jitter)
        for spec, key in tr._tcnn_tables:
            O.tcnn_grid_tie_gradients(spec, tr.grads[key])
        for name in tr._frozen:
            tr.grads[name].zero_()
        got = TP.to_tcnn_state_dict({k: v for k, v in tr.grads.items()}, model.field_spec, model.proposal_specs)
        return out, {k: v.detach().cpu() for k, v in got.items()}

    out16, g16 = run("f16")
    out32, g32 = run("fp32")
    for k, v in ld.items():
        assert abs(float(out16["loss_dict"][k]) - float(v)) <= 2e-3 * abs(float(v)) + 1e-7, (k, float(out16["loss_dict"][k]), float(v))
    assert_close(out16["rgb"], out_ref["rgb"].detach(), 2e-3, 1e-3, "mixed-precision train rgb")

    def rel(a, b, k):
        if k.endswith("tcnn_encoding.params") and "grid" not in k and "mlp_base." not in k:
            shp = (15, 64, 64, 1) if "mlp_semantics" in k else (63, 3, 64, 2) if "mlp_head" in k else (32, 16, 64, 1)
            a = torch.cat([x.reshape(-1) for x in TC.mlp_matrices(a, *shp)])
            b = torch.cat([x.reshape(-1) for x in TC.mlp_matrices(b, *shp)])
        return float((a - b).norm() / (b.norm() + 1e-12))

    vs_oracle, vs_fp32 = {}, {}
    for k, p in params.items():
        if p.grad is None:
            continue
        vs_oracle[k] = rel(g16[k], p.grad, k)
        vs_fp32[k] = rel(g16[k], g32[k], k)
    print("mixed precision vs half-activation oracle:", {k: f"{v:.2e}" for k, v in vs_oracle.items()})
    print("mixed precision vs the fp32 kernels:      ", {k: f"{v:.2e}" for k, v in vs_fp32.items()})
    field = [k for k in vs_oracle if k.startswith("field.")]
    assert len(field) >= 5
    bad = {k: v for k, v in vs_oracle.items() if v > (2e-2 if k.startswith("field.") else 3e-2)}
    assert not bad, f"mixed-precision gradients vs the half-activation oracle: {bad}"
    assert max(vs_fp32[k] for k in field) < 5e-2, vs_fp32
    grid = "field.mlp_base_grid.tcnn_encoding.params"
    assert vs_oracle[grid] < vs_fp32[grid], "the mixed-mode table gradient is no closer to the rounded function's than to fp32's"


def test_mixed_precision_training_reduces_the_loss_like_fp32():
    """Twelve optimiser steps on one batch in both matrix modes: the mixed-precision run's loss goes down and stays within 3 %
    of the exact-fp32 run's at every step (``test_training_reduces_loss_like_the_oracle`` holds the fp32 run to the oracle's)."""
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
    from cropnerf_amd.rays import Cameras

    sc, idx, jitter, image, mask = _tcnn_train_setup(seed=3, R=128)
    cams = Cameras(sc.c2w, sc.intr[:, 0], sc.intr[:, 1], sc.intr[:, 2], sc.intr[:, 3], sc.height, sc.width).to("cuda")
    rays = cams.generate_rays(idx.cuda())
    losses = {}
    for mode in ("fp32", "f16"):
        model = _tcnn_hip_model(sc)
        model.config.matrix_precision = mode
        model.training = True
        tr = FruitTrainer(model)
        hist = []
        for it in range(12):
            out = tr.forward_backward(rays, {"image": image, "fruit_mask": mask}, jitter=jitter)
            hist.append(sum(float(v) for v in out["loss_dict"].values()))
            tr.optimizer_step()
        losses[mode] = hist
    assert losses["f16"][-1] < 0.9 * losses["f16"][0], losses
    for a, b in zip(losses["f16"], losses["fp32"]):
        assert abs(a - b) <= 0.03 * abs(b) + 1e-4, losses


def test_tcnn_training_keeps_the_model_expressible_as_tcnn_modules(tmp_path):
    """A few optimiser steps on a tcnn-layout model: the loss falls, alias entries stay tied, frozen biases stay zero, and
    the model survives a save (tcnn vectors) -> load round trip bit for bit in its tables."""
    from cropnerf_amd import ops as O
    from cropnerf_amd.fruit_nerf import tcnn_params as TP
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
    from cropnerf_amd.rays import Cameras

    sc, idx, jitter, image, mask = _tcnn_train_setup(seed=9, R=256)
    model = _tcnn_hip_model(sc)
    model.training = True
    tr = FruitTrainer(model)
    cams = Cameras(sc.c2w, sc.intr[:, 0], sc.intr[:, 1], sc.intr[:, 2], sc.intr[:, 3], sc.height, sc.width).to("cuda")
    rays = cams.generate_rays(idx.cuda())
    first = last = None
    for it in range(12):
        out = tr.forward_backward(rays, {"image": image, "fruit_mask": mask}, jitter=jitter)
        tr.optimizer_step()
        loss = float(out["loss_dict"]["rgb_loss"])
        first = loss if first is None else first
        last = loss
    assert last < first
    for name in TP.frozen_parameter_names(model.field_spec, model.proposal_specs):
        assert float(model.params[name].abs().sum()) == 0.0, name
    assert float(model.params["field.mlp_head.layers.0.bias"].abs().sum()) > 0  # the padded-column bias trains
    for spec, key in tr._tcnn_tables:
        t = model.params[key].clone()
        O.tcnn_grid_tie_parameters(spec, t)
        assert torch.equal(t, model.params[key]), key  # already consistent
    state = TP.to_tcnn_state_dict(model.params, model.field_spec, model.proposal_specs)
    back = TP.from_tcnn_state_dict(state, model.field_spec, model.proposal_specs, "cuda", torch.float32)
    for k, v in model.params.items():
        assert torch.equal(back[k], v), k


# ------------------------------------------------------------------------------------------------ other field shapes
@pytest.mark.parametrize("max_res", [4096, 8192])
def test_big_shapes_in_the_tcnn_layout_forward_and_backward(max_res, monkeypatch):
    """fruit_nerf_method_big / _huge field shapes (geo 30, 3 x 128 semantic layers, max_res 4096 / 8192:
    fruit_nerf_config.py:66-172) with tcnn modules: the three implementations of cn_field_eval and
    cn_field_backward_general against ``oracle/tcnn.py`` (forward values, autograd gradients of every tcnn vector)."""
    from cropnerf_amd import config as PC
    from cropnerf_amd import ops as O
    from cropnerf_amd.fruit_nerf import tcnn_params as TP
    from _helpers import make_scene
    import oracle.tcnn as TCM

    n_img, R, S = 5, 41, 13
    ospec = OF.FieldSpec(grid=OF.GridSpec(16, 16, max_res, 12, 2), geo_feat_dim=30, num_layers_semantic=3,
                         hidden_dim_semantics=128, num_images=n_img, implementation="tcnn")
    state = {k: v for k, v in TC.random_params(ospec, [], seed=22, grid_scale=0.1).items() if k.startswith("field.")}
    state["field.mlp_head.tcnn_encoding.params"] = state["field.mlp_head.tcnn_encoding.params"] * 4.0  # see _tcnn_train_setup
    pspec = PC.FieldSpec(grid=PC.GridSpec(16, 16, max_res, 12, 2, "tcnn"), geo_feat_dim=30, num_layers_semantic=3,
                         hidden_dim_semantics=128, num_images=n_img)
    sc = make_scene(seed=2, log2_T=12, num_images=n_img, height=12, width=12, focal=16.0, prop_log2_T=10)
    rb = ORY.with_aabb_near_far(ORY.image_rays(sc.c2w, sc.intr, 1, 12, 12), sc.aabb.reshape(-1)).slice(0, R)
    g = torch.Generator().manual_seed(4)
    cam = torch.randint(0, n_img, (R, 1), generator=g)
    rs = OSM.spaced_sampler(rb, S, "uniform")
    gd, grgb, gsem = (torch.randn(R, S, generator=g), torch.randn(R, S, 3, generator=g), torch.randn(R, S, generator=g))
    # ---- oracle at the fp32 master values + autograd ----------------------------------------------------------------------
    pos = rs.positions()
    real = TCM._as_compute
    TCM._as_compute = lambda q, half: q.to(torch.float32)
    try:
        # Conditioning.  At max_res 4096+ one ulp of a position moves the finest levels' interpolation weights by 2e-4 and the
        # base MLP's hidden pre-activations by ~1e-5; a hidden unit that close to zero has its ReLU gate decided by the order
        # of the position arithmetic, and ONE flipped gate among the 34k moves the base MLP's and the grid's gradient by
        # 1e-3..1e-2 of its norm (measured per sample; the oracle shows the same on itself when its positions are moved by one
        # ulp: 7e-3).  So the tight comparison gives the samples holding such a unit no upstream gradient (a few per cent of
        # them); EVERY sample is still exercised by the second, un-masked comparison below at the tolerance that one flipped
        # gate needs.
        x01, _ = OF.normalized_positions(pos, sc.aabb, True)
        enc = TC.hash_grid(x01.reshape(-1, 3), state["field.mlp_base_grid.tcnn_encoding.params"], TC.grid_spec_of(ospec.grid),
                           half_params=False)
        pre = enc @ TC.mlp_matrices(state["field.mlp_base_mlp.tcnn_encoding.params"], 32, 31, 64, 1)[0][:, :32].t()
        keep = (pre.abs().min(dim=1).values > 5e-4 * pre.abs().max()).view(R, S)
        kept = float(keep.float().mean())
        print(f"samples kept for the tight comparison: {100 * kept:.1f} %")
        assert 0.88 < kept < 1.0, kept  # measured: 90.6 % (max_res 4096), the rest hold a hidden unit within 5e-4 of a ReLU edge

        def oracle_grads(gd_, grgb_, gsem_):
            p = {k: v.clone().requires_grad_(True) for k, v in state.items()}
            fo = OF.field_forward(pos, rb.directions, cam, p, ospec, sc.aabb, True, "val", training=True)
            geo = OF.field_density(pos, p, ospec, sc.aabb, True)[1].detach()
            sem = OF.semantics_from_geo(geo.reshape(-1, 30), p, ospec).view(R, S)
            ((fo["density"][..., 0] * gd_).sum() + (fo["rgb"] * grgb_).sum() + (sem * gsem_).sum()).backward()
            return p, fo, sem

        p_masked, fo, sem = oracle_grads(gd * keep, grgb * keep[..., None], gsem * keep)
        p_all, _, _ = oracle_grads(gd, grgb, gsem)
    finally:
        TCM._as_compute = real
    # ---- HIP: forward through every implementation that takes this shape ------------------------------------------------------
    full = dict(state)
    full["camera_optimizer.pose_adjustment"] = torch.zeros(n_img, 6)
    # (no proposal networks in this test: convert the field part only)
    dp = TP.from_tcnn_state_dict({**full}, pspec, [], "cuda", torch.float32)
    fh = O.FieldHandle(dp, pspec)
    scene = O.scene_struct(sc.aabb, True)
    from cropnerf_amd import _lib as L

    for impl in ("regw", "mfma", "scalar"):
        monkeypatch.setenv("CN_FIELD_EVAL_IMPL", impl)
        out = O.field_eval(fh, scene, to_dev(rb.origins), to_dev(rb.directions), to_dev(cam[:, 0]), to_dev(rs.starts[..., 0]),
                           to_dev(rs.ends[..., 0]), app_mode=L.APP_PER_CAMERA)
        assert_close(out["density"], fo["density"][..., 0].detach(), RTOL, ATOL, f"density ({impl})")
        assert_close(out["rgb"], fo["rgb"].detach(), RTOL, ATOL, f"rgb ({impl})")
        assert_close(out["semantics"], sem.detach(), RTOL, 5e-5, f"semantics ({impl})")
    monkeypatch.delenv("CN_FIELD_EVAL_IMPL")
    # ---- HIP: backward, with the masked upstream gradients (tight) and with all of them (every sample exercised) ---------------
    dims = {"field.mlp_base_mlp": (32, 31, 64, 1), "field.mlp_semantics": (30, 64, 128, 2), "field.mlp_head": (16 + 30 + 32, 3, 64, 2)}

    def compare(p, gd_, grgb_, gsem_, tol, what):
        grads = {k: torch.zeros_like(v) for k, v in dp.items()}
        O.field_backward_general(fh, O.FieldHandle(grads, pspec), scene, to_dev(rb.origins), to_dev(rb.directions), to_dev(cam[:, 0]),
                                 to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]), to_dev(gd_), to_dev(grgb_), to_dev(gsem_))
        O.tcnn_grid_tie_gradients(pspec.grid, grads["field.mlp_base_grid.hash_table"])
        for name in TP.frozen_parameter_names(pspec, []):
            grads[name].zero_()
        got = TP.to_tcnn_state_dict(grads, pspec, [])
        worst = {}
        for k, v in p.items():
            assert v.grad is not None and v.grad.abs().sum() > 0, k
            a, b = got[k], v.grad
            if k[: -len(".tcnn_encoding.params")] in dims:
                # compare what is a free parameter on both sides: tcnn's gradient is the same in EVERY padded (constant one)
                # input column, here the first padded column is the bias and the others are frozen at zero
                n_in, n_out, width, n_hidden = dims[k[: -len(".tcnn_encoding.params")]]
                ma, mb = TC.mlp_matrices(a, n_in, n_out, width, n_hidden), TC.mlp_matrices(b, n_in, n_out, width, n_hidden)
                ma[0], mb[0] = ma[0][:, : n_in + 1], mb[0][:, : n_in + 1]
                a, b = torch.cat([m.reshape(-1) for m in ma]), torch.cat([m.reshape(-1) for m in mb])
            worst[k] = float((a - b).norm() / (b.norm() + 1e-12))
        bad = {k: e for k, e in worst.items() if e > tol}
        assert not bad, f"{what}: {bad} (all: {worst})"

    compare(p_masked, gd * keep, grgb * keep[..., None], gsem * keep, 1e-3, "ill-conditioned samples masked")
    compare(p_all, gd, grgb, gsem, 2e-2, "every sample")


def test_seven_level_proposal_net_in_the_tcnn_layout(ops):
    """The _huge method's second proposal network has 7 levels up to 2048 (fruit_nerf_config.py:143-146): fused sampler and
    density kernel in the tcnn layout (dense levels 0-2, hashed above)."""
    from cropnerf_amd import config as PC
    from cropnerf_amd.fruit_nerf import tcnn_params as TP
    from _helpers import make_scene

    sc = make_scene(seed=6, log2_T=12, num_images=3, height=16, width=16, focal=20.0, prop_log2_T=12)
    og = [OF.ProposalSpec(OF.GridSpec(5, 16, 512, 12), implementation="tcnn"),
          OF.ProposalSpec(OF.GridSpec(7, 16, 2048, 12), implementation="tcnn")]
    ofs = OF.FieldSpec(grid=OF.GridSpec(log2_hashmap_size=12), num_images=3, implementation="tcnn")
    state = TC.random_params(ofs, og, seed=8, grid_scale=0.3)
    for k in list(state):
        if k.endswith("tcnn_encoding.params"):
            state[k] = state[k].to(torch.float16).to(torch.float32)
    fs = PC.FieldSpec(grid=PC.GridSpec(16, 16, 2048, 12, 2, "tcnn"), num_images=3)
    ps = [PC.ProposalSpec(PC.GridSpec(5, 16, 512, 12, 2, "tcnn")), PC.ProposalSpec(PC.GridSpec(7, 16, 2048, 12, 2, "tcnn"))]
    dp = TP.from_tcnn_state_dict(state, fs, ps, "cuda", torch.float16)
    dh = [ops.DensityHandle(dp, i, s) for i, s in enumerate(ps)]
    rb = rays_with_box(sc, 1, 200)
    rs = OSM.spaced_sampler(rb, 24, "uniform")
    scene = ops.scene_struct(sc.aabb, True)
    for lvl in range(2):
        ref = OF.proposal_density(rs.positions(), state, lvl, og[lvl], sc.aabb, True)
        out = ops.proposal_density(dh[lvl], scene, to_dev(rb.origins), to_dev(rb.directions), to_dev(rs.starts[..., 0]),
                                   to_dev(rs.ends[..., 0]))
        assert_close(out, ref[..., 0], RTOL, ATOL, f"{og[lvl].grid.num_levels}-level proposal density")
    # fused sampler (5- and 7-level nets) against the unfused composition on the device
    from cropnerf_amd import _lib as L

    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears + 0.01, rb.fars))
    fused = ops.proposal_sample(dh, scene, o, d, n, f, (64, 32), 16)
    sm = ops.sample_spaced(n, f, 64, L.SPACING_PIECEWISE)
    bins = torch.cat([sm["spacing_starts"], sm["spacing_ends"][:, -1:]], -1).contiguous()
    starts, ends = sm["starts"], sm["ends"]
    for lvl, s_next in ((0, 32), (1, 16)):
        den = ops.proposal_density(dh[lvl], scene, o, d, starts, ends)
        w = ops.composite(starts, ends, den, want_weights=True)["weights"]
        bins, eu = ops.sample_pdf(bins, w, n, f, s_next)
        starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
    assert_close(eu, fused["euclidean_bins"].cpu(), 1e-5, 1e-6, "fused vs composed bins (7-level net, tcnn layout)")
