"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly the symbols
``include/cropnerf_hip.h`` declares (no compute calls here -- there is no GPU in this tier)."""

import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "cropnerf_hip.h"


@pytest.fixture(scope="module")
def lib():
    import cropnerf_amd
    from cropnerf_amd import _lib

    if not _lib.LIB_PATH.exists():
        cropnerf_amd.build_library()
    return _lib.load()


def _declared():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("cn_raygen_pinhole", "cn_intersect_aabb", "cn_raygen_ortho", "cn_sample_spaced", "cn_sample_pdf",
                 "cn_proposal_density", "cn_field_eval", "cn_composite", "cn_render_rays", "cn_render_samples",
                 "cn_proposal_sample", "cn_proposal_sample_train", "cn_export_compact", "cn_pointcloud_compact", "cn_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_binding_covers_every_declared_symbol(lib):
    from cropnerf_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()


# ctypes mirror -> C struct name, for every struct that crosses the boundary by pointer
BY_POINTER_STRUCTS = {"Grid": "cn_grid", "TcnnGridPlan": "cn_tcnn_grid_plan", "Mlp": "cn_mlp", "FieldParams": "cn_field_params",
                      "DensityParams": "cn_density_params", "ProposalLevelOut": "cn_proposal_level_out",
                      "Scene": "cn_scene", "RenderOpts": "cn_render_opts", "ProjectionJob": "cn_projection_job",
                      "InterlevelLevel": "cn_interlevel_level", "PointGrid": "cn_point_grid"}


def test_struct_layout_matches_header():
    """sizeof AND the offset of every field of all eleven by-pointer structs, as the C compiler lays out the header, against
    the ctypes mirrors of ``_lib.py`` -- the C program is generated from the mirrors' ``_fields_``, so a field that is
    missing, renamed or reordered on either side fails to compile or to compare."""
    import subprocess
    import tempfile

    from cropnerf_amd import _lib

    lines, expect = [], []
    for py_name, c_name in BY_POINTER_STRUCTS.items():
        cls = getattr(_lib, py_name)
        lines.append(f'printf("%zu\\n", sizeof({c_name}));')
        expect.append((f"sizeof({c_name})", ctypes.sizeof(cls)))
        for fname, _ in cls._fields_:
            lines.append(f'printf("%zu\\n", offsetof({c_name}, {fname}));')
            expect.append((f"offsetof({c_name}, {fname})", getattr(cls, fname).offset))
            lines.append(f'printf("%zu\\n", sizeof((({c_name}*)0)->{fname}));')
            expect.append((f"sizeof({c_name}.{fname})", getattr(cls, fname).size))
    src = f'#include "{HEADER}"\n#include <stddef.h>\n#include <stdio.h>\nint main(void){{\n' + "\n".join(lines) + "\nreturn 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        c = Path(d) / "s.c"
        c.write_text(src)
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", str(c), "-o", str(Path(d) / "s")], check=True)
        out = subprocess.run([str(Path(d) / "s")], check=True, capture_output=True, text=True).stdout.split()
    assert len(out) == len(expect)
    wrong = [(what, int(got), want) for (what, want), got in zip(expect, out) if int(got) != want]
    assert not wrong, f"C layout differs from the ctypes mirror: {wrong}"
    # and the header declares no field the mirrors lack: count the members of each struct body
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    for py_name, c_name in BY_POINTER_STRUCTS.items():
        body = re.search(r"typedef struct " + c_name + r"\s*\{(.*?)\}\s*" + c_name + r"\s*;", text, flags=re.S)
        assert body, f"{c_name} not found in the header"
        members = [m for m in body.group(1).split(";") if m.strip()]
        assert len(members) == len(getattr(_lib, py_name)._fields_), f"{c_name}: {len(members)} members in the header"


def test_error_path_without_gpu(lib):
    """Argument validation happens on the host before any HIP call: a null pointer is CN_ERR_INVALID + message."""
    rc = lib.cn_intersect_aabb(None, None, None, 4, None, None, None)
    assert rc == -1
    assert b"cn_intersect_aabb" in lib.cn_last_error()
    assert lib.cn_version() >= 100


def test_ops_refuse_cpu_tensors():
    import torch

    from cropnerf_amd import ops

    with pytest.raises((RuntimeError, TypeError)):
        ops.intersect_aabb(torch.zeros(4, 3), torch.ones(4, 3), [0, 0, 0, 1, 1, 1])


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cropnerf_amd import _lib

    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setenv("CROPNERF_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(FileNotFoundError, match="no CPU or PyTorch fallback"):
        _lib.load()
    monkeypatch.delenv("CROPNERF_HIP_LIB")
    monkeypatch.setattr(_lib, "_libs", {})
    _lib.load()


def test_deterministic_test_library_exports_the_same_abi(monkeypatch):
    """libcropnerf_hip_det.so (the deterministic-accumulation test build, csrc/cn_det.hpp): same symbols, says what it is;
    the default library refuses a shadow registration instead of pretending."""
    from cropnerf_amd import _lib

    default = _lib.load()
    assert default.cn_deterministic_build() == 0
    assert default.cn_deterministic_register(None, 0, None, None) == _lib.CN_ERR_UNSUPPORTED
    assert b"default build" in default.cn_last_error()
    assert default.cn_deterministic_clear() == 0 and default.cn_deterministic_flush(None) == 0
    monkeypatch.setenv("CN_DETERMINISTIC_SCATTER", "1")
    det = _lib.load()
    assert det is not default and det.cn_deterministic_build() == 1
    for name in _lib.SIGNATURES:
        assert hasattr(det, name), name
    monkeypatch.setenv("CN_DETERMINISTIC_SCATTER", "0")
    assert _lib.load() is default


def test_scatter_scratch_size_is_a_host_computation(lib):
    """cn_grid_scatter_scratch_bytes: 64 private dense copies of level 0 (n + 1 = floor(scale_0 + offset) + 2 vertices per axis)
    + cell-major records (16 floats per cell) of the consecutive coarse levels (at most ten) with at most 1.75e7 cells, the
    small ones replicated (16 copies up to 8192 cells, 4 up to 65536)."""
    import ctypes as C
    import math

    from cropnerf_amd import _lib
    from cropnerf_amd.config import GridSpec

    def expected(scalings, offset):
        head = 64 * (math.floor(scalings[0] + offset) + 2) ** 3 * 2 * 4
        floats = 0
        for sc in scalings[:10]:
            cells = (math.floor(sc + offset) + 1) ** 3
            if cells > 17_500_000:
                break
            floats += (16 if cells <= 8192 else 4 if cells <= 65536 else 1) * cells * 16
        return head + 4 * floats

    g = _lib.Grid()
    g.num_levels = 16
    g.log2_table_size = 19
    sc = GridSpec().scalings()
    for i, v in enumerate(sc):
        g.scalings[i] = v
    assert g.scalings[0] == 16.0
    assert lib.cn_grid_scatter_scratch_bytes(C.byref(g)) == expected(sc, 0.0)
    g.layout = _lib.GRID_TCNN  # tcnn: scale_0 = 15, positions shifted by half a cell -> cells 0..15, vertices 0..16
    g.scalings[0] = 15.0
    assert lib.cn_grid_scatter_scratch_bytes(C.byref(g)) == expected([15.0] + list(sc[1:]), 0.5)
    # sized for a maximum batch: only the consecutive levels with at most 8 x max_samples cells (a prefix of the layout)
    def expected_for(scalings, offset, max_samples):
        head = 64 * (math.floor(scalings[0] + offset) + 2) ** 3 * 2 * 4
        floats = 0
        for s_ in scalings[:10]:
            cells = (math.floor(s_ + offset) + 1) ** 3
            if cells > 17_500_000 or cells > 8 * max_samples:
                break
            floats += (16 if cells <= 8192 else 4 if cells <= 65536 else 1) * cells * 16
        return head + 4 * floats

    tc = [15.0] + list(sc[1:])
    for ms in (4096 * 48, 6144, 65536 * 192, 1):
        assert lib.cn_grid_scatter_scratch_bytes_for(C.byref(g), ms) == expected_for(tc, 0.5, ms), ms
    assert lib.cn_grid_scatter_scratch_bytes_for(C.byref(g), 0) == lib.cn_grid_scatter_scratch_bytes(C.byref(g))
    assert lib.cn_grid_scatter_scratch_bytes_for(C.byref(g), 4096 * 48) < lib.cn_grid_scatter_scratch_bytes(C.byref(g)) // 2
    g.scalings[0] = 1000.0  # a fine "coarsest level": not worth private copies
    assert lib.cn_grid_scatter_scratch_bytes(C.byref(g)) == 0
    assert lib.cn_grid_scatter_scratch_bytes(None) == 0
