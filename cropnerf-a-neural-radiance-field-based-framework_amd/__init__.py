"""cropnerf_amd -- MI355X-native hot path of CropNeRF's ``fruit_nerf`` method.

Import as ``cropnerf_amd`` (the directory name carries the reference's full name and is not a valid Python
identifier; ``cropnerf_amd/__init__.py`` at the repo root maps the import name onto this directory).

Layout
    csrc/       hand-written HIP kernels for gfx950 + the C ABI (include/cropnerf_hip.h)
    _lib.py     ctypes binding (fails loudly when libcropnerf_hip.so is missing; there is no fallback)
    ops.py      tensor-level wrappers over the C ABI
    config.py   hyper-parameters / parameter shapes
    synthetic.py  synthetic cameras + parameters for tests and bench
    fruit_nerf/ host-side mirror of the reference's plugin surface (FruitModel, exporters, CLIs)
"""

from . import _lib, config  # noqa: F401

__all__ = ["_lib", "config", "ops", "synthetic", "build_library"]


def build_library(force: bool = False):
    """Compile libcropnerf_hip.so for gfx950 in-tree (hipcc)."""
    from . import build as _build

    return _build.build(force=force)


def __getattr__(name):
    if name in ("ops", "synthetic", "rays", "distributed"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
