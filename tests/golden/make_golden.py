#!/usr/bin/env python
"""Generate the committed golden vectors (inputs + expected outputs) from the CPU oracle.

    python tests/golden/make_golden.py

The reference itself cannot run in this image (nerfstudio absent) and has no fixtures of its own, so these vectors
come from ``oracle/`` (whose arithmetic is pinned by the analytic KATs in tests/test_oracle_kat.py).  They freeze the
oracle against drift and give the HIP path a data-only target that travels to the GPU box.
Files: fruit_nerf_small.npz (a 2^11-entry-per-level field, 96 rays).
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from _helpers import make_scene, oracle_model, rays_with_box  # noqa: E402
from oracle import model as OM  # noqa: E402
from oracle import rays as ORY  # noqa: E402


def main():
    torch.set_num_threads(1)
    sc = make_scene(seed=7, log2_T=11, num_images=4, height=12, width=8, focal=14.0, prop_log2_T=9)
    out = {f"param/{k}": v.numpy() for k, v in sc.params.items()}
    out["c2w"], out["intr"], out["aabb"] = sc.c2w.numpy(), sc.intr.numpy(), sc.aabb.numpy()
    out["hw"] = np.array([sc.height, sc.width])

    # (1) M-uniform, 64 samples, no contraction, rays clipped to the scene box, mean appearance
    rb = rays_with_box(sc, 1)
    m = oracle_model(sc, "inference", disable_scene_contraction=True)
    m.uniform_samples = 64
    ref = m.forward(rb)
    for k in ("origins", "directions", "nears", "fars"):
        out[f"uniform/in/{k}"] = getattr(rb, k).numpy()
    for k in ("rgb", "accumulation", "depth", "semantics", "semantics_colormap"):
        out[f"uniform/out/{k}"] = ref[k].numpy()
    out["uniform/out/weights"] = ref["_weights"][..., 0].numpy()

    # (2) test-mode forward: collider + SO3xR3 tweak + proposal sampler (256, 96) + 48 samples, contraction on
    rb2 = ORY.image_rays(sc.c2w, sc.intr, 2, sc.height, sc.width)
    ref2 = oracle_model(sc, "test").forward(rb2)
    out["proposal/in/camera_indices"] = rb2.camera_indices.numpy()
    for k in ("origins", "directions"):
        out[f"proposal/in/{k}"] = getattr(rb2, k).numpy()
    for k in ("rgb", "accumulation", "depth", "prop_depth_0", "prop_depth_1", "semantics"):
        out[f"proposal/out/{k}"] = ref2[k].numpy()
    out["proposal/out/bins"] = torch.cat([ref2["_starts"][..., 0], ref2["_ends"][:, -1:, 0]], -1).numpy()

    # (3) export mode: orthographic rays x 40 samples, per-sample outputs + the three kept-point counts
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = ORY.surface_points(ORY.corners_of_aabb(aabb), 6)
    rb3 = ORY.ortho_rays(pts, plane, 36, 1)
    m3 = oracle_model(sc, "export")
    m3.setup_inference(True, 40)
    ref3 = m3.forward(rb3)
    out["export/in/origins"], out["export/in/directions"] = rb3.origins.numpy(), rb3.directions.numpy()
    out["export/in/nears"], out["export/in/fars"] = rb3.nears.numpy(), rb3.fars.numpy()
    for k in ("rgb", "point_location", "semantics", "density"):
        out[f"export/out/{k}"] = ref3[k].numpy()
    np.savez_compressed(os.path.join(HERE, "fruit_nerf_small.npz"), **out)
    print("wrote", os.path.join(HERE, "fruit_nerf_small.npz"), sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
