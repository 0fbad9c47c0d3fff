"""CPU-side checks of host logic that needs no GPU: plugin config objects, PLY I/O, run-directory config."""

import math
import os

import numpy as np
import pytest
import torch


def test_plugin_surface_names_and_values():
    from cropnerf_amd.fruit_nerf import fruit_nerf_config as c

    m = c.fruit_nerf_method.config  # fruit_nerf_config.py:29-65
    assert m.method_name == "fruit_nerf" and m.steps_per_save == 2000 and m.max_num_iterations == 40000
    assert m.pipeline.datamanager.train_num_rays_per_batch == 4096
    assert m.pipeline.model.eval_num_rays_per_chunk == 1 << 15
    assert set(m.optimizers) == {"proposal_networks", "fields", "camera_opt"}
    assert m.optimizers["fields"].lr == 1e-2 and m.optimizers["fields"].max_steps == 200000
    big = c.fruit_nerf_method_big.config.pipeline.model  # :68-119
    assert big.num_nerf_samples_per_ray == 128 and big.num_proposal_samples_per_ray == (512, 256)
    assert big.geo_feat_dim == 30 and big.log2_hashmap_size == 21 and big.max_res == 4096
    huge = c.fruit_nerf_method_huge.config  # :121-172
    assert "camera_opt" not in huge.optimizers
    specs = huge.pipeline.model.proposal_specs()
    assert [s.grid.num_levels for s in specs] == [5, 7] and specs[1].grid.max_res == 2048


def test_model_config_defaults_follow_reference():
    from cropnerf_amd.config import FruitNerfModelConfig, param_shapes

    cfg = FruitNerfModelConfig()  # fruit_nerf.py:59-68 + nerfacto defaults
    assert (cfg.semantic_loss_weight, cfg.pass_semantic_gradients, cfg.num_layers_semantic) == (1.0, False, 2)
    assert (cfg.hidden_dim_semantics, cfg.geo_feat_dim) == (64, 15)
    assert cfg.num_proposal_samples_per_ray == (256, 96) and cfg.num_nerf_samples_per_ray == 48
    fs = cfg.field_spec(10)
    shapes = param_shapes(fs, cfg.proposal_specs())
    assert shapes["field.mlp_base_grid.hash_table"] == (16 * 2 ** 19, 2)
    assert shapes["field.mlp_base_mlp.layers.1.weight"] == (16, 64)
    assert shapes["field.mlp_semantics.layers.0.weight"] == (64, 15)
    assert shapes["field.mlp_head.layers.0.weight"] == (64, 63)
    assert shapes["proposal_networks.1.mlp.layers.0.weight"] == (16, 10)


def test_ply_roundtrip(tmp_path):
    from cropnerf_amd.fruit_nerf.ply import read_ply, write_ply

    g = np.random.default_rng(0)
    pts, cols = g.normal(size=(1000, 3)), g.uniform(size=(1000, 3))
    write_ply(str(tmp_path / "a.ply"), pts, cols)
    p2, c2 = read_ply(str(tmp_path / "a.ply"))
    np.testing.assert_array_equal(p2, pts)
    assert np.abs(c2 - cols).max() <= 1 / 255 + 1e-9
    write_ply(str(tmp_path / "e.ply"), np.zeros((0, 3)), np.zeros((0, 3)))
    p3, _ = read_ply(str(tmp_path / "e.ply"))
    assert p3.shape == (0, 3)


def test_raybundle_container_semantics():
    from cropnerf_amd.rays import RayBundle

    rb = RayBundle(torch.arange(24.).reshape(2, 4, 3), torch.ones(2, 4, 3), nears=torch.zeros(2, 4, 1), fars=torch.ones(2, 4, 1))
    assert len(rb) == 8 and rb.shape == (2, 4)
    sl = rb.get_row_major_sliced_ray_bundle(3, 6)
    assert sl.origins.shape == (3, 3) and sl.origins[0, 0] == 9.0
    m = torch.tensor([[True, False, False, True], [False, True, False, False]])
    sub = rb[m]
    assert sub.origins.shape == (3, 3) and sub.nears.shape == (3, 1)


def test_image_metrics_restatements():
    """get_image_metrics_and_images' upstream pieces (fruit_nerf.py:647-700): SSIM against a direct float64 evaluation of
    its definition, PSNR, the colormaps' end points and the reference's softmax-over-one-channel IoU."""
    import numpy as np

    from cropnerf_amd.fruit_nerf import image_metrics as IM

    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 24, 20, generator=g)
    b = (a + 0.1 * torch.randn(1, 3, 24, 20, generator=g)).clamp(0, 1)
    assert abs(float(IM.ssim(a, a)) - 1.0) < 1e-6
    # direct evaluation: Gaussian-weighted statistics of every 11 x 11 window inside the image
    x = np.arange(11) - 5.0
    w = np.exp(-(x / 1.5) ** 2 / 2)
    w = np.outer(w / w.sum(), w / w.sum())
    A, B = a[0].double().numpy(), b[0].double().numpy()
    vals = []
    for c in range(3):
        for i in range(24 - 10):
            for j in range(20 - 10):
                p, t = A[c, i:i + 11, j:j + 11], B[c, i:i + 11, j:j + 11]
                mp, mt = (w * p).sum(), (w * t).sum()
                spp, stt, spt = (w * p * p).sum() - mp * mp, (w * t * t).sum() - mt * mt, (w * p * t).sum() - mp * mt
                vals.append(((2 * mp * mt + 1e-4) * (2 * spt + 9e-4)) / ((mp * mp + mt * mt + 1e-4) * (spp + stt + 9e-4)))
    assert abs(float(IM.ssim(a, b)) - float(np.mean(vals))) < 2e-5
    assert abs(float(IM.psnr(a, b)) + 10 * math.log10(float(((a - b) ** 2).mean()))) < 1e-5
    lo, hi = IM.apply_colormap(torch.zeros(2, 2, 1)), IM.apply_colormap(torch.ones(2, 2, 1))
    assert lo.shape == (2, 2, 3) and torch.allclose(lo[0, 0], torch.tensor([0.18995, 0.07176, 0.23217]), atol=1e-5)
    assert torch.allclose(hi[0, 0], torch.tensor([0.4796, 0.01583, 0.01055]), atol=1e-5)
    depth = torch.tensor([[[1.0], [3.0]]])
    faded = IM.apply_depth_colormap(depth, accumulation=torch.tensor([[[1.0], [0.0]]]))
    assert torch.allclose(faded[0, 0], lo[0, 0], atol=1e-5) and torch.allclose(faded[0, 1], torch.ones(3))

    class _M:
        device = torch.device("cpu")
        proposal_networks = [None, None]

    H, W = 16, 16
    mask = (torch.rand(H, W, 1, generator=g) > 0.7).float()
    outs = {"rgb": torch.rand(H, W, 3, generator=g) * 1.2, "accumulation": torch.rand(H, W, 1, generator=g),
            "depth": torch.rand(H, W, 1, generator=g) * 4, "prop_depth_0": torch.rand(H, W, 1, generator=g),
            "prop_depth_1": torch.rand(H, W, 1, generator=g), "semantics": torch.randn(H, W, 1, generator=g) * 5}
    metrics, images = IM.get_image_metrics_and_images(_M(), outs, {"image": torch.rand(H, W, 3, generator=g), "fruit_mask": mask})
    assert set(metrics) == {"psnr", "ssim", "lpips", "iou"} and math.isnan(metrics["lpips"])
    assert metrics["iou"] == pytest.approx(float(mask.mean()))     # softmax over one channel == 1 everywhere
    assert images["img"].shape == (H, 2 * W, 3) and images["fruit_mask"].shape == (H, W, 3)
    assert set(images) == {"img", "accumulation", "depth", "prop_depth_0", "prop_depth_1", "semantics_colormap", "fruit_mask"}


def test_training_callbacks_surface():
    """get_training_callbacks (fruit_nerf.py:198-232): annealing before, sampler step after every iteration."""
    from cropnerf_amd.config import FruitNerfModelConfig
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel

    class _M:  # the method only touches config, set_anneal and two counters
        config = FruitNerfModelConfig()
        anneal = None

        def set_anneal(self, a):
            self.anneal = a

    m = _M()
    cbs = FruitModel.get_training_callbacks(m)
    assert [c.where_to_run for c in cbs] == [["before_train_iteration"], ["after_train_iteration"]]
    for step in (0, 500, 5000):
        for c in cbs:
            c.run_callback_at_location(step, "before_train_iteration")
        frac = min(step / m.config.proposal_weights_anneal_max_num_iters, 1.0)
        b = m.config.proposal_weights_anneal_slope
        assert m.anneal == pytest.approx(b * frac / ((b - 1) * frac + 1)) and m.step == step
        for c in cbs:
            c.run_callback_at_location(step, "after_train_iteration")
    assert m._steps_since_update == 3 and m._sampler_step == 5000
    m.config.use_proposal_weight_anneal = False
    assert FruitModel.get_training_callbacks(m) == []


def test_oriented_box():
    """nerfstudio OrientedBox.from_params / within (the ns-export pointcloud crop)."""
    from cropnerf_amd.rays import OrientedBox

    box = OrientedBox.from_params((0.5, 0.0, 0.0), (0.0, 0.0, math.pi / 2), (2.0, 1.0, 4.0))
    # yaw 90 deg: the box's x edge (length 2) lies along world y, its y edge (length 1) along world -x
    pts = torch.tensor([[0.5, 0.0, 0.0], [0.5, 0.9, 0.0], [0.5, 1.1, 0.0], [0.9, 0.0, 0.0], [1.1, 0.0, 0.0],
                        [0.5, 0.0, 1.9], [0.5, 0.0, 2.0]])
    assert box.within(pts).tolist() == [True, True, False, True, False, True, False]
    assert torch.allclose(box.R @ box.R.T, torch.eye(3), atol=1e-6)
    rpy = OrientedBox.from_params((0, 0, 0), (0.3, -0.2, 0.7), (1, 1, 1)).R
    assert torch.allclose(rpy @ torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.04521531, -0.3482963, 0.93629336]), atol=1e-5)


def test_semantic_field_head_names_shapes_and_the_separate_op():
    """``components/field_heads.py:29-40``: Linear(in_dim, num_classes), no activation.  The class is where the package
    takes the head's state-dict names and shapes from; as a separate op it equals the oracle's head on the same features."""
    from cropnerf_amd.config import FieldSpec, GridSpec, init_params, param_shapes
    from cropnerf_amd.fruit_nerf.components.field_heads import SemanticFieldHead

    spec = FieldSpec(grid=GridSpec(4, 16, 64, 10, 2), num_images=3)
    head = SemanticFieldHead(spec.hidden_dim_transient, 1)
    shapes = param_shapes(spec, [])
    assert {k: shapes[k] for k in head.keys} == head.shapes() == {
        "field.field_head_semantics.net.weight": (1, 64), "field.field_head_semantics.net.bias": (1,)}
    params = init_params(spec, [], seed=1)
    w, b = head.check(params)
    assert float(w.abs().max()) <= 1.0 / math.sqrt(64) and float(b.abs().max()) <= 1.0 / math.sqrt(64)
    x = torch.randn(5, 7, 64, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(head(x, params), x @ w.T + b, atol=1e-6)
    g = torch.Generator().manual_seed(2)
    fresh = head.init(g)
    assert {k: tuple(v.shape) for k, v in fresh.items()} == head.shapes()
    assert all(float(v.abs().max()) <= 0.125 for v in fresh.values())
    with pytest.raises(ValueError, match="shape"):
        head.check({**params, head.weight_key: torch.zeros(1, 32)})
    with pytest.raises(KeyError):
        head.check({})
    with pytest.raises(NotImplementedError, match="one semantic logit"):
        SemanticFieldHead(64, 2)
    with pytest.raises(ValueError, match="activation"):
        SemanticFieldHead(64, 1, activation=torch.nn.ReLU())


def test_bench_compact_line_fits_the_driver():
    """bench.py's stdout line built from a FULL record (round 4's, 20.6 KB, which the driver could not read) stays under 4 KB,
    keeps the contract keys, `roofline` and `cpu_baseline`, and carries the secondaries as bare numbers."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench_for_test", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(root, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 16000
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "rays_per_sec", "psnr_vs_oracle_db", "roofline", "cpu_baseline")
    line = {k: full[k] for k in contract}
    line["cpu_baseline"]["sample_short"] = "C2: 10 chunks x 4096 rays x 192 spp, median 2894 ms"
    line["cpu_baseline"]["c1"]["sample_short"] = "C1: 400x400, 64 spp, 20 chunks x 1024 rays, median 72 ms"
    extra = {k: v for k, v in full.items() if k not in contract}
    c = b.compact_line(line, extra)
    text = json.dumps(c)
    assert len(text) < b.MAX_LINE_BYTES == 4096
    assert json.loads(text) == c
    for k in contract:
        assert k in c
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"]
    assert c["roofline"]["traffic"] == full["roofline"]["traffic"] and c["cpu_baseline"]["cores"] == 16
    assert all(isinstance(v, (int, float)) for v in c["secondary"].values())
    assert c["secondary"]["train_65536_ms"] == 15.56 and c["secondary"]["c4_seconds"] == 12.77
