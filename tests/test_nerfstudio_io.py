"""nerfstudio run-directory format (``cropnerf_amd/fruit_nerf/nerfstudio_io.py``): config.yml with python tags read
without nerfstudio installed, checkpoint name handling.  CPU only."""

import textwrap

import pytest
import torch

from cropnerf_amd.config import FruitNerfModelConfig
from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO

# the shape of a config.yml as ``ns-train fruit_nerf`` dumps it (abridged; tags and nesting as PyYAML writes dataclasses,
# paths, tuples and class references)
SAMPLE = textwrap.dedent("""\
    !!python/object:nerfstudio.engine.trainer.TrainerConfig
    _target: !!python/name:nerfstudio.engine.trainer.Trainer ''
    data: null
    experiment_name: plant_1
    gradient_accumulation_steps: {}
    load_dir: null
    machine: !!python/object:nerfstudio.configs.base_config.MachineConfig
      device_type: cuda
      num_devices: 1
      seed: 42
    max_num_iterations: 40000
    method_name: fruit_nerf
    mixed_precision: true
    optimizers:
      fields:
        optimizer: !!python/object:nerfstudio.engine.optimizers.AdamOptimizerConfig
          _target: &id001 !!python/name:torch.optim.adam.Adam ''
          eps: 1.0e-15
          lr: 0.01
          max_norm: null
          weight_decay: 0
        scheduler: !!python/object:nerfstudio.engine.schedulers.ExponentialDecaySchedulerConfig
          _target: !!python/name:nerfstudio.engine.schedulers.ExponentialDecayScheduler ''
          lr_final: 0.0001
          max_steps: 200000
      proposal_networks:
        optimizer: !!python/object:nerfstudio.engine.optimizers.AdamOptimizerConfig
          _target: *id001
          eps: 1.0e-15
          lr: 0.01
        scheduler: null
    output_dir: !!python/object/apply:pathlib.PosixPath
    - outputs
    pipeline: !!python/object:fruit_nerf.fruit_pipeline.FruitPipelineConfig
      _target: !!python/name:fruit_nerf.fruit_pipeline.FruitPipeline ''
      datamanager: !!python/object:fruit_nerf.data.fruit_datamanager.FruitDataManagerConfig
        _target: !!python/name:fruit_nerf.data.fruit_datamanager.FruitDataManager ''
        data: &id002 !!python/object/apply:pathlib.PosixPath
        - /
        - workspace
        - data
        - plant_1
        dataparser: !!python/object:fruit_nerf.data.cotton_nerf_dataparser.CottonNerfDataParserConfig
          _target: !!python/name:fruit_nerf.data.cotton_nerf_dataparser.CottonNerf ''
          auto_scale_poses: true
          data: *id002
          downscale_factor: null
          orientation_method: up
          train_split_fraction: 0.95
        eval_num_rays_per_batch: 4096
        train_num_rays_per_batch: 4096
      model: !!python/object:fruit_nerf.fruit_nerf.FruitNerfModelConfig
        _target: !!python/name:fruit_nerf.fruit_nerf.FruitModel ''
        appearance_embed_dim: 32
        background_color: last_sample
        camera_optimizer: !!python/object:nerfstudio.cameras.camera_optimizers.CameraOptimizerConfig
          _target: !!python/name:nerfstudio.cameras.camera_optimizers.CameraOptimizer ''
          mode: SO3xR3
        collider_params:
          far_plane: 6.0
          near_plane: 2.0
        eval_num_rays_per_chunk: 32768
        far_plane: 1000.0
        geo_feat_dim: 15
        hidden_dim_semantics: 64
        implementation: tcnn
        log2_hashmap_size: 19
        loss_coefficients:
          rgb_loss_coarse: 1.0
          rgb_loss_fine: 1.0
        max_res: 2048
        near_plane: 0.05
        num_layers_semantic: 2
        num_levels: 16
        num_nerf_samples_per_ray: 48
        num_proposal_iterations: 2
        num_proposal_samples_per_ray: !!python/tuple
        - 256
        - 96
        proposal_net_args_list:
        - hidden_dim: 16
          log2_hashmap_size: 17
          max_res: 128
          num_levels: 5
          use_linear: false
        - hidden_dim: 16
          log2_hashmap_size: 17
          max_res: 256
          num_levels: 5
          use_linear: false
        semantic_loss_weight: 1.0
        use_average_appearance_embedding: true
    relative_model_dir: !!python/object/apply:pathlib.PosixPath
    - nerfstudio_models
    steps_per_save: 2000
    timestamp: 2024-05-01_120000
    vis: viewer
    """)


def test_reads_a_nerfstudio_config_without_importing_its_classes(tmp_path):
    p = tmp_path / "config.yml"
    p.write_text(SAMPLE)
    tree = NIO.load_config_yml(p)
    assert tree["__class__"] == "nerfstudio.engine.trainer.TrainerConfig"
    assert tree["_target"] == "nerfstudio.engine.trainer.Trainer"
    assert tree["output_dir"] == "outputs" and tree["relative_model_dir"] == "nerfstudio_models"
    dm = tree["pipeline"]["datamanager"]
    assert dm["data"] == "/workspace/data/plant_1" and dm["dataparser"]["data"] == "/workspace/data/plant_1"  # anchors
    assert dm["dataparser"]["__class__"].endswith("CottonNerfDataParserConfig")
    mc = NIO.model_config_from_tree(tree)
    assert isinstance(mc, FruitNerfModelConfig)
    assert mc.num_proposal_samples_per_ray == (256, 96) and mc.implementation == "tcnn" and mc.max_res == 2048
    assert mc.proposal_net_args_list[1]["max_res"] == 256
    # nerfstudio's config.get_checkpoint_dir(): output_dir / experiment / method / timestamp / relative_model_dir,
    # unless the models sit next to the config (a run that was moved)
    assert str(NIO.checkpoint_dir(p, tree)) == "outputs/plant_1/fruit_nerf/2024-05-01_120000/nerfstudio_models"
    (tmp_path / "nerfstudio_models").mkdir()
    assert NIO.checkpoint_dir(p, tree) == tmp_path / "nerfstudio_models"
    with pytest.raises(ValueError):
        q = tmp_path / "other.yml"
        q.write_text("a: 1\n")
        NIO.load_config_yml(q)


def test_written_config_has_the_reference_class_paths_and_round_trips(tmp_path):
    mc = FruitNerfModelConfig(num_nerf_samples_per_ray=192, background_color=(0.0, 0.0, 0.0), implementation="tcnn")
    p = tmp_path / "config.yml"
    NIO.write_config_yml(p, method_name="fruit_nerf", model_config=mc, data="/d/plant_1", output_dir="outputs",
                         experiment_name="plant_1", timestamp="t0", max_num_iterations=40000, steps_per_save=2000,
                         mixed_precision=True, train_num_rays_per_batch=4096, eval_num_rays_per_batch=4096,
                         dataparser={"data": "/d/plant_1", "downscale_factor": 2})
    text = p.read_text()
    for tag in ("!!python/object:nerfstudio.engine.trainer.TrainerConfig",
                "!!python/object:fruit_nerf.fruit_pipeline.FruitPipelineConfig",
                "!!python/object:fruit_nerf.data.fruit_datamanager.FruitDataManagerConfig",
                "!!python/object:fruit_nerf.fruit_nerf.FruitNerfModelConfig",
                "!!python/name:fruit_nerf.fruit_nerf.FruitModel ''", "!!python/object/apply:pathlib.PosixPath",
                "!!python/tuple"):
        assert tag in text, tag
    tree = NIO.load_config_yml(p)
    back = NIO.model_config_from_tree(tree)
    assert back == mc
    assert tree["pipeline"]["datamanager"]["dataparser"]["downscale_factor"] == 2 and tree["data"] == "/d/plant_1"


def test_checkpoint_names(tmp_path):
    w = torch.arange(6.0).reshape(3, 2)
    pipeline = {
        "_model.field.mlp_base_grid.tcnn_encoding.params": w,
        "_model.proposal_networks.0.mlp_base.model.0.hash_table": w + 1,
        "_model.proposal_networks.1.mlp_base.model.1.layers.0.weight": w + 2,
        "_model.lpips.net.x": w,
        "datamanager.train_ray_generator.image_coords": w,  # not part of the model
    }
    st = NIO.model_state_from_pipeline(pipeline)
    assert "datamanager.train_ray_generator.image_coords" not in st
    assert torch.equal(st["proposal_networks.0.encoding.hash_table"], w + 1)
    assert torch.equal(st["proposal_networks.1.mlp.layers.0.weight"], w + 2)
    # a checkpoint written from a DDP run (fruit_pipeline.py:119-121) carries module. in front
    st = NIO.model_state_from_pipeline({"module._model.field.x": w})
    assert list(st) == ["field.x"]
    with pytest.raises(ValueError):
        NIO.model_state_from_pipeline({"something.else": w})
    # names on the way out, and the file layout
    names = NIO.nerfstudio_names({"proposal_networks.0.encoding.hash_table": w, "proposal_networks.0.mlp.layers.1.bias": w,
                                  "field.mlp_head.layers.0.weight": w})
    assert set(names) == {"proposal_networks.0.mlp_base.model.0.hash_table",
                          "proposal_networks.0.mlp_base.model.1.layers.1.bias", "field.mlp_head.layers.0.weight"}
    ck = tmp_path / "nerfstudio_models" / "step-000000012.ckpt"
    NIO.save_checkpoint(ck, 12, names, optimizers={"step": 13}, buffers={"field.aabb": torch.zeros(2, 3)})
    step, state, loaded = NIO.load_checkpoint(NIO.latest_checkpoint(ck.parent))
    assert step == 12 and set(loaded) == {"step", "pipeline", "optimizers", "schedulers", "scalers"}
    assert "_model.field.aabb" in loaded["pipeline"] and "proposal_networks.0.encoding.hash_table" in state
    with pytest.raises(FileNotFoundError):
        NIO.latest_checkpoint(tmp_path)


def test_checkpoint_with_arbitrary_objects_is_refused_unless_trusted(tmp_path, monkeypatch):
    """step-*.ckpt files may come from outside (reference-trained runs): they are read with ``weights_only=True``; a file that
    needs unrestricted unpickling is refused unless the caller vouches for it."""
    import torch

    from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO

    import fractions

    def Payload():  # noqa: N802 -- any class off torch's allow-list makes the restricted unpickler stop
        return fractions.Fraction(1, 3)

    plain = tmp_path / "step-000000001.ckpt"
    torch.save({"step": 1, "pipeline": {"_model.field.w": torch.ones(2)}, "optimizers": {"lr": 0.01, "m": [torch.zeros(1)]}}, plain)
    step, state, loaded = NIO.load_checkpoint(plain)
    assert step == 1 and torch.equal(state["field.w"], torch.ones(2))
    odd = tmp_path / "step-000000002.ckpt"
    torch.save({"step": 2, "pipeline": {"_model.field.w": torch.ones(2)}, "extra": Payload()}, odd)
    monkeypatch.delenv("CROPNERF_TRUST_CHECKPOINTS", raising=False)
    with pytest.raises(ValueError, match="arbitrary pickle code"):
        NIO.load_checkpoint(odd)
    assert NIO.load_checkpoint(odd, trusted=True)[0] == 2
    monkeypatch.setenv("CROPNERF_TRUST_CHECKPOINTS", "1")
    assert NIO.load_checkpoint(odd)[0] == 2
