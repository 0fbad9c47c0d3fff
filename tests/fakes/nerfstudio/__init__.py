"""A minimal stand-in for the ``nerfstudio`` package -- TEST FIXTURE ONLY (``tests/fakes`` is put on ``sys.path`` by the
plugin-wiring tests).  It holds the types and field names the ``fruit_nerf`` plugin touches (nerfstudio 1.1.3:
``plugins.types``, ``engine.trainer / optimizers / schedulers``, ``configs.base_config``, ``models.base_model``,
``pipelines.base_pipeline``, ``data.datamanagers.base_datamanager``, ``data.dataparsers.base_dataparser``,
``cameras.rays / cameras / camera_optimizers``, ``data.scene_box``) with just enough behaviour to construct a method
specification, build a pipeline and run a train step.  It is not nerfstudio and nothing in the product imports it."""
__version__ = "1.1.3+fake"
