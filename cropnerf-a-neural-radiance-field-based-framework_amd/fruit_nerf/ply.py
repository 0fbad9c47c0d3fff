"""Minimal binary-little-endian PLY writer/reader for point clouds (xyz float64, rgb uchar, optional normals):
the files ``open3d.io.write_point_cloud`` produces for the reference's exporters (``semantics_pc.ply``,
``semantic_colormap.ply`` / ``semantic.ply`` / ``density.ply``)."""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def write_ply(path: str, points: np.ndarray, colors: Optional[np.ndarray] = None,
              normals: Optional[np.ndarray] = None) -> None:
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    n = points.shape[0]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    header = ["ply", "format binary_little_endian 1.0", "comment Created by cropnerf_amd", f"element vertex {n}",
              "property double x", "property double y", "property double z"]
    if normals is not None:
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
        header += ["property double nx", "property double ny", "property double nz"]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += ["property uchar red", "property uchar green", "property uchar blue"]
    header.append("end_header")
    rec = np.empty(n, dtype=np.dtype(fields))
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    if normals is not None:
        normals = np.asarray(normals, dtype=np.float64).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = normals[:, 0], normals[:, 1], normals[:, 2]
    if colors is not None:
        c = np.clip(np.asarray(colors, dtype=np.float64).reshape(-1, 3), 0.0, 1.0)
        c8 = (c * 255.0).astype(np.uint8)  # open3d truncates
        rec["red"], rec["green"], rec["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply(path: str, with_normals: bool = False):
    """(points, colours or None); with ``with_normals`` a third entry: the normals [N,3] or None."""
    with open(path, "rb") as f:
        fields = []
        n = 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                _, typ, name = line.split()
                fields.append((name, {"double": "<f8", "float": "<f4", "uchar": "u1"}[typ]))
            elif line == "end_header":
                break
        rec = np.frombuffer(f.read(), dtype=np.dtype(fields), count=n)
    pts = np.stack([rec["x"], rec["y"], rec["z"]], -1).astype(np.float64)
    cols = None
    if "red" in rec.dtype.names:
        cols = np.stack([rec["red"], rec["green"], rec["blue"]], -1).astype(np.float64) / 255.0
    if not with_normals:
        return pts, cols
    nrm = None
    if "nx" in rec.dtype.names:
        nrm = np.stack([rec["nx"], rec["ny"], rec["nz"]], -1).astype(np.float64)
    return pts, cols, nrm
