"""Golden vectors (tests/golden/fruit_nerf_small.npz, made by tests/golden/make_golden.py from the oracle):
 * CPU: the oracle still reproduces them (guards the checker against drift);
 * GPU: the HIP path, through the C ABI, reproduces them -- data only, nothing of the reference or oracle needed."""

import os

import numpy as np
import pytest
import torch

from _helpers import assert_close

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fruit_nerf_small.npz")
RTOL, ATOL = 2e-4, 2e-5


@pytest.fixture(scope="module")
def gold():
    z = np.load(GOLD)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _scene_from_gold(gold):
    from _helpers import Scene
    from oracle import field as OF

    params = {k[len("param/"):]: v for k, v in gold.items() if k.startswith("param/")}
    fspec = OF.FieldSpec(grid=OF.GridSpec(log2_hashmap_size=11), num_images=4)
    pspecs = [OF.ProposalSpec(OF.GridSpec(5, 16, 128, 9)), OF.ProposalSpec(OF.GridSpec(5, 16, 256, 9))]
    h, w = (int(v) for v in gold["hw"])
    return Scene(params, fspec, pspecs, gold["aabb"], gold["c2w"], gold["intr"], h, w)


def test_oracle_reproduces_golden(gold):
    from _helpers import oracle_model, rays_with_box
    from oracle import rays as ORY

    torch.set_num_threads(1)
    sc = _scene_from_gold(gold)
    m = oracle_model(sc, "inference", disable_scene_contraction=True)
    m.uniform_samples = 64
    ref = m.forward(rays_with_box(sc, 1))
    for k in ("rgb", "accumulation", "depth", "semantics"):
        assert_close(ref[k], gold[f"uniform/out/{k}"], 1e-5, 1e-6, f"uniform {k}")
    ref2 = oracle_model(sc, "test").forward(ORY.image_rays(sc.c2w, sc.intr, 2, sc.height, sc.width))
    for k in ("rgb", "accumulation", "semantics", "prop_depth_0"):
        assert_close(ref2[k], gold[f"proposal/out/{k}"], 1e-4, 1e-5, f"proposal {k}")


@pytest.mark.gpu
def test_hip_reproduces_golden(gold):
    from _helpers import dev_params, product_specs, to_dev
    from cropnerf_amd import ops

    sc = _scene_from_gold(gold)
    fspec, pspecs = product_specs(sc)
    dp = dev_params(sc)
    fh = ops.FieldHandle(dp, fspec)
    dh = [ops.DensityHandle(dp, i, ps) for i, ps in enumerate(pspecs)]
    # (1) uniform render
    i = {k: to_dev(gold[f"uniform/in/{k}"]) for k in ("origins", "directions", "nears", "fars")}
    out = ops.render_rays(fh, ops.scene_struct(sc.aabb, False), ops.render_opts(64), i["origins"], i["directions"],
                          i["nears"], i["fars"], want_weights=True)
    for k in ("rgb", "accumulation", "semantics", "weights"):
        assert_close(out[k], gold[f"uniform/out/{k}"], RTOL, 5e-5 if k == "semantics" else ATOL, f"uniform {k}")
    ok = (out["depth"].cpu() - gold["uniform/out/depth"]).abs() < 1e-5
    assert ok.float().mean() > 0.98
    # (2) test-mode forward with the proposal sampler
    o, d = to_dev(gold["proposal/in/origins"]).clone(), to_dev(gold["proposal/in/directions"]).clone()
    cam = to_dev(gold["proposal/in/camera_indices"][:, 0])
    ops.apply_pose_adjustment(dp["camera_optimizer.pose_adjustment"], cam, o, d)
    R = o.shape[0]
    n, f = torch.zeros(R, 1, device="cuda"), torch.full((R, 1), 1000.0, device="cuda")
    scn = ops.scene_struct(sc.aabb, True)
    ps = ops.proposal_sample(dh, scn, o, d, n, f, (256, 96), 48)
    assert_close(ps["euclidean_bins"], gold["proposal/out/bins"], 2e-3, 1e-4, "bins", frac_ok=0.995)
    out = ops.render_rays(fh, scn, ops.render_opts(48), o, d, n, f, camera_indices=cam, bins=to_dev(gold["proposal/out/bins"]))
    for k in ("rgb", "accumulation", "semantics"):
        assert_close(out[k], gold[f"proposal/out/{k}"], RTOL, 5e-5 if k == "semantics" else ATOL, f"proposal {k}")
    # (3) export-mode per-sample outputs
    e = {k: to_dev(gold[f"export/in/{k}"]) for k in ("origins", "directions", "nears", "fars")}
    out = ops.render_samples(fh, ops.scene_struct(sc.aabb, False), ops.render_opts(40), e["origins"], e["directions"],
                             e["nears"], e["fars"])
    assert_close(out["positions"], gold["export/out/point_location"], 1e-6, 1e-6, "export positions")
    for k in ("rgb", "semantics", "density"):
        assert_close(out[k], gold[f"export/out/{k}"], RTOL, ATOL, f"export {k}")
