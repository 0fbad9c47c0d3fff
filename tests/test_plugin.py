"""The nerfstudio plugin objects (``fruit_nerf_config.py`` + ``nerfstudio_adapter.py``) against the minimal fake
``nerfstudio`` package in ``tests/fakes`` (types and field names only).  Each check runs in a fresh interpreter with
``tests/fakes`` on ``PYTHONPATH`` -- the main test process has already imported the package WITHOUT nerfstudio, and must
keep seeing it that way.  Reference: ``crop_nerf/fruit_nerf/fruit_nerf_config.py:29-172``,
``fruit_pipeline.py:88-121``, ``fruit_nerf.py:80-85,191-232``."""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKES = os.path.join(ROOT, "tests", "fakes")


def _run(code: str, timeout: int = 600) -> str:
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([FAKES, ROOT, os.path.join(ROOT, "tests")]))
    p = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_without_nerfstudio_the_native_specifications_are_exported():
    from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC

    assert not FC.HAVE_NERFSTUDIO
    assert FC.fruit_nerf_method is FC.native_method("fruit_nerf")
    assert FC.fruit_nerf_method.config.method_name == "fruit_nerf"


def test_method_specifications_are_nerfstudio_types():
    out = _run("""
        import dataclasses
        import nerfstudio
        from nerfstudio.plugins.types import MethodSpecification
        from nerfstudio.engine.trainer import TrainerConfig
        from nerfstudio.engine.optimizers import AdamOptimizerConfig, RAdamOptimizerConfig
        from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
        from nerfstudio.models.base_model import Model, ModelConfig
        from nerfstudio.pipelines.base_pipeline import VanillaPipeline, VanillaPipelineConfig
        from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager, VanillaDataManagerConfig
        from nerfstudio.data.dataparsers.base_dataparser import DataParserConfig, DataParser
        from nerfstudio.configs.base_config import ViewerConfig
        from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC
        from cropnerf_amd.fruit_nerf import nerfstudio_adapter as A

        assert FC.HAVE_NERFSTUDIO
        # what NERFSTUDIO_METHOD_CONFIGS=fruit_nerf=fruit_nerf.fruit_nerf_config:fruit_nerf_method resolves to
        for name, spec in (("fruit_nerf", FC.fruit_nerf_method), ("fruit_nerf_big", FC.fruit_nerf_method_big),
                           ("fruit_nerf_huge", FC.fruit_nerf_method_huge)):
            assert type(spec) is MethodSpecification and type(spec.config) is TrainerConfig
            c = spec.config
            assert c.method_name == name and c.vis == "viewer" and c.mixed_precision is True
            assert type(c.viewer) is ViewerConfig and c.viewer.num_rays_per_chunk == 1 << 15
            p = c.pipeline
            assert isinstance(p, VanillaPipelineConfig) and type(p).__name__ == "FruitPipelineConfig"
            assert issubclass(p._target, VanillaPipeline) and p._target.__name__ == "FruitPipeline"
            assert isinstance(p.datamanager, VanillaDataManagerConfig) and issubclass(p.datamanager._target, VanillaDataManager)
            assert isinstance(p.datamanager.dataparser, DataParserConfig) and issubclass(p.datamanager.dataparser._target, DataParser)
            assert isinstance(p.model, ModelConfig) and type(p.model).__name__ == "FruitNerfModelConfig"
            assert issubclass(p.model._target, Model) and p.model._target.__name__ == "FruitModel"
            assert p.model.eval_num_rays_per_chunk == 1 << 15
        c = FC.fruit_nerf_method.config
        assert (c.steps_per_eval_batch, c.steps_per_save, c.max_num_iterations) == (500, 2000, 40000)
        assert c.pipeline.datamanager.train_num_rays_per_batch == 4096 and c.pipeline.datamanager.eval_num_rays_per_batch == 4096
        assert type(c.pipeline.datamanager.dataparser).__name__ == "CottonNerfDataParserConfig"
        assert set(c.optimizers) == {"proposal_networks", "fields", "camera_opt"}
        for g, lr, fin, steps in (("proposal_networks", 1e-2, 1e-4, 200000), ("fields", 1e-2, 1e-4, 200000), ("camera_opt", 1e-3, 1e-4, 5000)):
            o, s = c.optimizers[g]["optimizer"], c.optimizers[g]["scheduler"]
            assert type(o) is AdamOptimizerConfig and o.lr == lr and o.eps == 1e-15
            assert type(s) is ExponentialDecaySchedulerConfig and s.lr_final == fin and s.max_steps == steps
        m = c.pipeline.model  # FruitNerfModelConfig(NerfactoModelConfig) fields, fruit_nerf.py:59-68 + nerfacto defaults
        for k, v in dict(semantic_loss_weight=1.0, pass_semantic_gradients=False, num_layers_semantic=2, hidden_dim_semantics=64,
                         geo_feat_dim=15, num_nerf_samples_per_ray=48, num_proposal_samples_per_ray=(256, 96), max_res=2048,
                         log2_hashmap_size=19, near_plane=0.05, far_plane=1000.0, background_color="last_sample").items():
            assert getattr(m, k) == v, k
        big, huge = FC.fruit_nerf_method_big.config, FC.fruit_nerf_method_huge.config
        assert type(big.optimizers["fields"]["optimizer"]) is RAdamOptimizerConfig and big.optimizers["proposal_networks"]["scheduler"] is None
        assert big.pipeline.model.geo_feat_dim == 30 and big.pipeline.model.num_proposal_samples_per_ray == (512, 256)
        assert big.pipeline.datamanager.train_num_rays_per_batch == 8192 and big.max_num_iterations == 100000
        assert "camera_opt" not in huge.optimizers and type(huge.pipeline.datamanager.dataparser).__name__ == "FruitNerfDataParserConfig"
        assert huge.pipeline.model.proposal_net_args_list[1]["num_levels"] == 7
        # the reference's constructor surface
        import inspect
        sig = inspect.signature(A.FruitPipeline.__init__)
        assert list(sig.parameters)[1:] == ["config", "device", "test_mode", "world_size", "local_rank", "grad_scaler", "render_rgb_inference"]
        for meth in ("get_param_groups", "get_training_callbacks", "setup_inference", "get_outputs", "get_loss_dict", "get_metrics_dict",
                     "get_image_metrics_and_images", "get_outputs_for_camera_ray_bundle", "get_outputs_for_projections", "forward"):
            assert callable(getattr(A.FruitModel, meth)), meth
        for meth in ("setup_inference", "next_sample_volume"):
            assert callable(getattr(A.FruitDataManager, meth)), meth
        # the native specifications stay available to this repo's own CLIs
        assert FC.native_method("fruit_nerf").config.optimizers["fields"].lr == 1e-2
        print("plugin-types-ok")
    """)
    assert "plugin-types-ok" in out


# a small fruit_nerf pipeline configuration with a synthetic datamanager (shared by the GPU tests below)
_PIPE_PRELUDE = """
        import functools, torch
        from nerfstudio.engine.optimizers import Optimizers
        from nerfstudio.engine.callbacks import TrainingCallbackAttributes, TrainingCallbackLocation
        from cropnerf_amd.fruit_nerf import fruit_nerf_config as FC
        from cropnerf_amd.fruit_nerf import nerfstudio_adapter as A
        from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel as Hip, Semantics
        from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
        from cropnerf_amd.rays import RayBundle, SceneBox
        from cropnerf_amd import synthetic
        import copy, dataclasses

        torch.manual_seed(0)
        N, H, W = 6, 32, 32
        c2w, intr = synthetic.orbit_cameras(N, height=H, width=W, focal=40.0)
        aabb = torch.tensor(synthetic.SCENE_AABB, dtype=torch.float32)

        class Dataset:  # what the pipeline reads from datamanager.train_dataset
            scene_box = SceneBox(aabb); metadata = {"semantics": Semantics()}
            def __len__(self): return N

        from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManager
        class DM(A.FruitDataManager):
            def __init__(self, config, device="cpu", **kw):
                VanillaDataManager.__init__(self, dataclasses.replace(config, dataparser=None), device=device, **kw)
                self.train_dataset = Dataset()
                self.gen = torch.Generator().manual_seed(5)
            def next_train(self, step):
                from cropnerf_amd import ops
                R = 512
                idx = torch.stack([torch.randint(0, N, (R,), generator=self.gen), torch.randint(0, H, (R,), generator=self.gen),
                                   torch.randint(0, W, (R,), generator=self.gen)], -1).cuda()
                r = ops.raygen_pinhole(c2w.cuda(), intr.cuda(), ray_indices=idx)
                rb = RayBundle(r["origins"], r["directions"], r["pixel_area"], r["camera_indices"])
                batch = {"image": torch.rand(R, 3, generator=self.gen), "fruit_mask": (torch.rand(R, 1, generator=self.gen) > 0.5).float()}
                return rb, batch

        spec = copy.deepcopy(FC.fruit_nerf_method)
        cfg = spec.config
        cfg.pipeline.datamanager._target = DM
        small = dict(log2_hashmap_size=14, proposal_net_args_list=[
            {"hidden_dim": 16, "log2_hashmap_size": 12, "num_levels": 5, "max_res": 128, "use_linear": False},
            {"hidden_dim": 16, "log2_hashmap_size": 12, "num_levels": 5, "max_res": 256, "use_linear": False}],
            num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24)
        for k, v in small.items():
            setattr(cfg.pipeline.model, k, v)
"""


@pytest.mark.gpu
def test_plugin_model_trains_through_nerfstudio_style_optimizers():
    """Pipeline -> model construction through the config ``setup()`` chain, ``get_param_groups`` as ``nn.Parameter`` views of
    the flat buffers, and train iterations driven the way nerfstudio's Trainer drives them (``Optimizers`` of
    ``torch.optim.Adam`` + ``loss.backward()``), against this package's own ``FruitTrainer`` on the same batches."""
    out = _run(_PIPE_PRELUDE + """
        pipe = cfg.pipeline.setup(device="cuda", test_mode="val")
        assert type(pipe).__name__ == "FruitPipeline" and type(pipe.model).__name__ == "FruitModel"
        groups = pipe.get_param_groups()
        assert set(groups) == {"proposal_networks", "fields", "camera_opt"}
        assert all(isinstance(p, torch.nn.Parameter) and p.is_cuda for g in groups.values() for p in g)
        # the parameters ARE the kernels' memory
        tr = pipe.model.trainer
        p0 = pipe.model.param("field.mlp_head.layers.0.weight")
        assert p0.data_ptr() == pipe.model.hip.params["field.mlp_head.layers.0.weight"].data_ptr()

        # the same model / batches through this package's own trainer
        from cropnerf_amd import config as NC
        ref_model = Hip(NC.FruitNerfModelConfig(**small), SceneBox(aabb), N, {"semantics": Semantics()}, device="cuda",
                        params={k: v.detach().clone() for k, v in pipe.model.hip.params.items()})
        ref_model.training = True
        ref = FruitTrainer(ref_model)
        ref._gen.manual_seed(123); tr._gen.manual_seed(123)
        ref_dm = DM(cfg.pipeline.datamanager, device="cuda")

        opt = Optimizers(cfg.optimizers, groups)
        callbacks = pipe.get_training_callbacks(TrainingCallbackAttributes(optimizers=opt, pipeline=pipe))
        pipe.train()
        skipped = 0
        for step in range(14):
            for cb in callbacks: cb.run_callback_at_location(step, TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
            opt.zero_grad_all()
            _, loss_dict, metrics = pipe.get_train_loss_dict(step)
            loss = functools.reduce(torch.add, loss_dict.values())
            loss.backward()
            prop_grad = groups["proposal_networks"][0].grad
            skipped += prop_grad is None
            opt.optimizer_step_all(); opt.scheduler_step_all(step)
            for cb in callbacks: cb.run_callback_at_location(step, TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
            rb, batch = ref_dm.next_train(step)
            out = ref.train_iteration(rb, batch)
            for k in loss_dict:
                a, b = float(loss_dict[k]), float(out["loss_dict"][k])
                assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (step, k, a, b)
        assert skipped >= 1, "after step 10 the proposal networks must sit out some iterations"
        # same trajectory: parameters agree (atomics make the two runs differ in the last bits)
        for k, v in ref_model.params.items():
            w = pipe.model.hip.params[k]
            rel = float((w - v).norm() / (v.norm() + 1e-12))
            assert rel < (0.3 if k.startswith("camera_optimizer") else 5e-2), (k, rel)
        # eval goes straight to the HIP forward
        pipe.model.eval()
        rb, _ = ref_dm.next_train(0)
        o = pipe.model(rb)
        assert o["rgb"].shape == (512, 3) and "semantics_colormap" in o
        sd = pipe.model.state_dict()
        assert "field.mlp_base_grid.hash_table" in sd and "field.aabb" in sd and "proposal_networks.0.mlp_base.model.0.hash_table" in sd
        print("plugin-train-ok", skipped)
    """, timeout=900)
    assert "plugin-train-ok" in out


@pytest.mark.gpu
def test_tcnn_alias_entries_follow_their_owners_under_nerfstudio_optimizers():
    """implementation="tcnn" through the plugin: nerfstudio's own ``torch.optim.Adam`` steps the ``nn.Parameter`` views, so
    nothing inside the optimiser knows that several entries of a dense level stand for one tcnn parameter.  After every
    iteration (AFTER_TRAIN_ITERATION callback) the alias entries must equal their owners -- otherwise samples in the upper
    half-cell of a dense level read stale values and the exported tcnn vector is not what was rendered."""
    out = _run(_PIPE_PRELUDE + """
        from cropnerf_amd import ops
        cfg.pipeline.model.implementation = "tcnn"
        pipe = cfg.pipeline.setup(device="cuda", test_mode="val")
        model = pipe.model
        tr = model.trainer
        assert tr.tcnn and len(tr._tcnn_tables) == 3
        groups = pipe.get_param_groups()
        opt = Optimizers(cfg.optimizers, groups)
        callbacks = pipe.get_training_callbacks(TrainingCallbackAttributes(optimizers=opt, pipeline=pipe))
        pipe.train()

        def tied(spec, table):
            # copying every owner's value to its aliases changes nothing iff the aliases already agree with their owners
            # (entries no position can reach stand for no parameter and are left alone)
            t2 = table.clone()
            ops.tcnn_grid_tie_parameters(spec, t2)
            return torch.equal(t2, table)

        before = {key: model.hip.params[key].clone() for _, key in tr._tcnn_tables}
        assert all(tied(spec, model.hip.params[key]) for spec, key in tr._tcnn_tables)
        for step in range(3):
            for cb in callbacks: cb.run_callback_at_location(step, TrainingCallbackLocation.BEFORE_TRAIN_ITERATION)
            opt.zero_grad_all()
            _, loss_dict, metrics = pipe.get_train_loss_dict(step)
            functools.reduce(torch.add, loss_dict.values()).backward()
            opt.optimizer_step_all(); opt.scheduler_step_all(step)
            if step == 0:  # the hazard itself: Adam has moved the owners, the aliases still hold the old values
                spec, key = tr._tcnn_tables[0]
                assert not tied(spec, model.hip.params[key])
            for cb in callbacks: cb.run_callback_at_location(step, TrainingCallbackLocation.AFTER_TRAIN_ITERATION)
            for spec, key in tr._tcnn_tables:
                assert tied(spec, model.hip.params[key]), (step, key)
        for _, key in tr._tcnn_tables:
            assert not torch.equal(before[key], model.hip.params[key]), key  # training did move the tables
        # state dict -> a fresh model: what is rendered is what was exported, and a torch-named load re-ties as well
        sd = model.state_dict()
        assert "field.mlp_base_grid.tcnn_encoding.params" in sd
        spec, key = tr._tcnn_tables[0]
        packed = ops.tcnn_grid_unpack(spec, model.hip.params[key])
        assert torch.equal(sd["field.mlp_base_grid.tcnn_encoding.params"].float().cuda(), packed)
        torch_named = {k: v.detach().clone() for k, v in model.hip.params.items()}
        plan = spec.plan()
        b0, res0 = int(plan.level_bits[0]), int(plan.resolution[0])
        alias = res0 | (3 << b0) | (5 << (2 * b0))     # (res, 3, 5) stands for the parameter of (0, 4, 5)
        torch_named[key][alias] += 1.0                  # a stale alias in the incoming state
        from cropnerf_amd.fruit_nerf import nerfstudio_io as NIO
        model.load_state_dict(NIO.nerfstudio_names(torch_named), strict=False)
        assert tied(spec, model.hip.params[key])
        print("tcnn-tie-ok")
    """, timeout=900)
    assert "tcnn-tie-ok" in out
