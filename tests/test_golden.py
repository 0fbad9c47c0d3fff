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


# ---------------------------------------------------------------------------------------------------------------------
# tests/golden/postprocess_small.npz (tests/golden/make_golden_postprocess.py): the stages either side of the path
# ---------------------------------------------------------------------------------------------------------------------
POST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess_small.npz")


@pytest.fixture(scope="module")
def post():
    return dict(np.load(POST))


def test_oracle_reproduces_postprocess_golden(post):
    from oracle import clustering as OC
    from oracle import outliers as OO
    from oracle import zbuffer as OZ

    H, W = (int(v) for v in post["zbuffer/hw"])
    P = OZ.get_projection_mat(*post["zbuffer/intr"], post["zbuffer/c2w"])
    z = np.full((H, W), np.inf, dtype=np.float32)
    img = np.zeros((H, W), dtype=np.uint8)
    z, img, _ = OZ.update_buffer(z, OZ.get_projection(P, post["zbuffer/tree"]), img, 0, large=True)
    z, img, (vx, vy) = OZ.update_buffer(z, OZ.get_projection(P, post["zbuffer/fruit"]), img, 1)
    vis = np.zeros((H, W), dtype=np.uint8)
    vis[vx, vy] = 255
    assert np.array_equal(z, post["zbuffer/out/z"]) and np.array_equal(img, post["zbuffer/out/img"])
    assert np.array_equal(vis, post["zbuffer/out/visible"])
    down = OC.voxel_down_sample(post["cluster/points"].astype(np.float64), float(post["cluster/voxel_size"]))
    down = down[np.lexsort(np.round(down, 6).T[::-1])].astype(np.float32)
    assert np.allclose(down, post["cluster/out/down"], atol=1e-7)
    labels, core = OC.dbscan(post["cluster/out/down"], float(post["cluster/eps"]), int(post["cluster/min_points"]))
    assert np.array_equal(labels, post["cluster/out/labels"]) and np.array_equal(core, post["cluster/out/core"])
    assert np.allclose(OO.knn_mean_distance(post["cluster/out/down"].astype(np.float64), 20), post["outlier/out/mean_distance"])


@pytest.mark.gpu
def test_hip_reproduces_postprocess_golden(post):
    from cropnerf_amd import ops
    from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M
    from oracle import zbuffer as OZ

    # depth-based projection: exact (labels, visibility) / fp32 (depths)
    H, W = (int(v) for v in post["zbuffer/hw"])
    P = OZ.get_projection_mat(*post["zbuffer/intr"], post["zbuffer/c2w"])  # 12 numbers of host arithmetic
    z = torch.full((H, W), float("inf"), dtype=torch.float32, device="cuda")
    img = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    M.update_buffer(z, M.get_projection(P, post["zbuffer/tree"], H, W), img, label=0, large=True)
    _, _, vis = M.update_buffer(z, M.get_projection(P, post["zbuffer/fruit"], H, W), img, label=1)
    assert np.array_equal(img.cpu().numpy(), post["zbuffer/out/img"])
    assert np.array_equal(vis.cpu().numpy(), post["zbuffer/out/visible"])
    assert np.allclose(z.cpu().numpy(), post["zbuffer/out/z"], rtol=1e-6, equal_nan=True)
    # voxel down-sampling, DBSCAN, statistical outliers
    pts = torch.from_numpy(post["cluster/points"]).cuda()
    down, _ = ops.voxel_down_sample(pts, float(post["cluster/voxel_size"]))
    d = down.cpu().numpy()
    d = d[np.lexsort(np.round(d, 6).T[::-1])]
    assert d.shape == post["cluster/out/down"].shape and np.allclose(d, post["cluster/out/down"], atol=2e-6)
    gd = torch.from_numpy(post["cluster/out/down"]).cuda()
    labels, core = ops.dbscan(gd, float(post["cluster/eps"]), int(post["cluster/min_points"]))
    labels, core = labels.cpu().numpy(), core.cpu().numpy()
    ref_l, ref_c = post["cluster/out/labels"], post["cluster/out/core"]
    assert np.array_equal(core, ref_c) and np.array_equal(labels == -1, ref_l == -1)
    assert np.array_equal(labels[core], ref_l[core])   # border points may belong to either neighbouring cluster
    mean_d = ops.knn_mean_distance(gd, 20).cpu().numpy()
    assert np.allclose(mean_d, post["outlier/out/mean_distance"], rtol=2e-5, atol=1e-7)
    mask = ops.statistical_outlier_mask(gd, 20, 2.0).cpu().numpy()
    ref_m = post["outlier/out/mask"]
    assert (mask != ref_m).sum() <= 2                   # only points sitting on the threshold can flip in fp32
