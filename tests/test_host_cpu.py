"""CPU-side checks of host logic that needs no GPU: plugin config objects, PLY I/O, run-directory config."""

import numpy as np
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
