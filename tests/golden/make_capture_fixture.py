#!/usr/bin/env python
"""The camera data of the reference's one real capture file (build container only).

    python tests/golden/make_capture_fixture.py      ->  tests/golden/capture_3dcotton.npz

``/root/reference/crop_nerf/fruit_nerf/utils/transforms.json`` is a nerfstudio-format capture description of a 3DCotton
plant: 147 frames at 1920 x 1440 from one pinhole camera (f = 1442.48), poses as 4x4 camera-to-world matrices,
``orientation_override: "none"``, ``auto_scale_poses_override: false``, no lens distortion.  Poses and intrinsics are DATA:
the fixture keeps them as arrays -- the frame numbers of the ``images/frame_%05d.jpg`` names, the 147 matrices, the shared
intrinsics -- and nothing else of the file (no velocities, blur scores, depth paths).  ``tests/test_dataparser.py`` rebuilds a
``transforms.json`` from it for the dataparser mirror; ``bench.py`` renders at the capture's resolution from these poses."""

import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/crop_nerf/fruit_nerf/utils/transforms.json"


def main():
    with open(SRC, encoding="utf-8") as f:
        meta = json.load(f)
    frames = meta["frames"]
    numbers, poses = [], []
    for fr in frames:
        m = re.fullmatch(r"images/frame_(\d{5})\.jpg", fr["file_path"])
        assert m, fr["file_path"]
        numbers.append(int(m.group(1)))
        poses.append(np.asarray(fr["transform_matrix"], dtype=np.float64))
    out = {
        "frame_number": np.asarray(numbers, np.int32),
        "transform_matrix": np.stack(poses),
        "intrinsics": np.asarray([meta["fl_x"], meta["fl_y"], meta["cx"], meta["cy"]], np.float64),
        "size_hw": np.asarray([meta["h"], meta["w"]], np.int32),
        "distortion_k1_k2_p1_p2": np.asarray([meta["k1"], meta["k2"], meta["p1"], meta["p2"]], np.float64),
        # orientation_override: 0 = "none"; auto_scale_poses_override: 0 = false
        "orientation_override_is_none": np.asarray(int(meta["orientation_override"] == "none"), np.int32),
        "auto_scale_poses_override": np.asarray(int(bool(meta["auto_scale_poses_override"])), np.int32),
        "aabb_scale": np.asarray(meta["aabb_scale"], np.int32),
    }
    path = os.path.join(HERE, "capture_3dcotton.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(frames), "frames")


if __name__ == "__main__":
    main()
