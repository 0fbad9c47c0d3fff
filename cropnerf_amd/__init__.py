"""Import alias: ``cropnerf_amd`` -> ``cropnerf-a-neural-radiance-field-based-framework_amd/``.

The package directory keeps the reference's full (hyphenated) name; this shim makes it importable.
"""

import pathlib as _pathlib

_REAL = _pathlib.Path(__file__).resolve().parent.parent / "cropnerf-a-neural-radiance-field-based-framework_amd"
__path__ = [str(_REAL)]
__file__ = str(_REAL / "__init__.py")
exec(compile((_REAL / "__init__.py").read_text(), __file__, "exec"))
