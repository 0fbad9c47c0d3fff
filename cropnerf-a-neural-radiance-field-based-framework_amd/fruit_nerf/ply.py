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
    # The vertex records as a byte matrix [rows, record size], filled by block copies (xyz, normals, colours) one slab of 2^20
    # points at a time into ONE reused buffer and written without another copy.  A structured array of the whole cloud filled
    # field by field took 6.3 s for 10^7 points with normals (510 MB, most of it first-touch page faults of the two 510 MB
    # temporaries and strided per-field copies) -- half of what rendering those points takes on the device.
    rec_bytes = sum(np.dtype(t).itemsize for _, t in fields)
    points = np.asarray(points, dtype="<f8").reshape(-1, 3)
    if normals is not None:
        normals = np.asarray(normals, dtype="<f8").reshape(-1, 3)
    if colors is not None:
        colors = np.asarray(colors).reshape(-1, 3)
    slab = 1 << 20
    buf = np.empty((min(n, slab), rec_bytes), dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        for i in range(0, n, slab):
            m = min(slab, n - i)
            out = buf[:m]
            out[:, 0:24] = np.ascontiguousarray(points[i:i + m]).view(np.uint8).reshape(m, 24)
            col = 24
            if normals is not None:
                out[:, col:col + 24] = np.ascontiguousarray(normals[i:i + m]).view(np.uint8).reshape(m, 24)
                col += 24
            if colors is not None:
                c = np.clip(colors[i:i + m].astype(np.float64, copy=False), 0.0, 1.0)
                c *= 255.0
                out[:, col:col + 3] = c.astype(np.uint8)  # open3d truncates
            f.write(memoryview(out))


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
