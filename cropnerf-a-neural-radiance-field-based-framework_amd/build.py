"""Build libcropnerf_hip.so (gfx950) in-tree with hipcc.

    python cropnerf-a-neural-radiance-field-based-framework_amd/build.py [--force]

The shared library is written next to this file (``libcropnerf_hip.so``); it is git-ignored but travels with
the tree.  hipcc cross-compiles for gfx950 without a GPU.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libcropnerf_hip.so"
BUILD = HERE / "build"
ARCH = "gfx950"

OUT_DET = HERE / "libcropnerf_hip_det.so"  # the deterministic-accumulation test build (csrc/cn_det.hpp)
DET_SOURCES = ["train_render.hip", "train_field.hip", "tcnn_grid.hip", "deterministic.hip"]  # compiled again with the macro

SOURCES = ["api_common.cpp", "deterministic.hip", "raygen.hip", "sampler.hip", "field_simple.hip", "composite.hip", "render_fused.hip",
           "proposal.hip", "export.hip", "train_render.hip", "train_field.hip", "zbuffer.hip", "knn.hip", "cluster.hip", "tcnn_grid.hip", "contour.hip", "projection.hip", "png_writer.cpp"]


def _headers():
    """Every header a translation unit may include: all of csrc/*.hpp (globbed, so a new header cannot be forgotten and
    leave a stale object behind) and the public C header."""
    return sorted(CSRC.glob("*.hpp")) + [HERE.parent / "include" / "cropnerf_hip.h"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")
    return exe


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    BUILD.mkdir(exist_ok=True)
    headers = _headers()
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-x", "hip", "-Wno-unused-result"]
    flags += os.environ.get("CN_EXTRA_HIPCC_FLAGS", "").split()  # e.g. -DCN_FUSED_PIPELINE=0 for A/B builds
    jobs = []
    objs = []
    det_objs = []  # the test library: the training units compiled with the macro, every other object shared
    for src in SOURCES:
        s = CSRC / src
        o = BUILD / (src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s, *headers]):
            jobs.append([hipcc, *flags, "-c", str(s), "-o", str(o)])
        if src in DET_SOURCES:
            od = BUILD / (src.rsplit(".", 1)[0] + ".det.o")
            det_objs.append(od)
            if force or _stale(od, [s, *headers]):
                jobs.append([hipcc, *flags, "-DCN_DETERMINISTIC_SCATTER=1", "-c", str(s), "-o", str(od)])
        else:
            det_objs.append(o)
    jobs.sort(key=lambda c: -Path(c[-3]).stat().st_size)  # the long compiles first

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{p.stdout}\n{p.stderr}")

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(OUT, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(OUT), *map(str, objs), "-lz"])  # zlib: png_writer.cpp
    if force or jobs or _stale(OUT_DET, det_objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(OUT_DET), *map(str, det_objs), "-lz"])
    return OUT


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
