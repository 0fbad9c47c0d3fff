"""Depth-based semantic projection: the numpy oracle restates the reference's functions statement by
statement (``oracle/zbuffer.py``); known-answer tests on CPU, HIP parity on the GPU."""

import numpy as np
import pytest
import torch

from oracle import zbuffer as OZ


def _camera(seed=0):
    rng = np.random.default_rng(seed)
    # a camera 2 units from the origin looking at it (OpenGL convention: -z forward)
    pos = np.array([0.3, -0.2, 2.0])
    fwd = -pos / np.linalg.norm(pos)
    right = np.cross(fwd, [0.0, 1.0, 0.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, pos
    return c2w, rng


def test_projection_matrix_and_center_pixel():
    c2w, _ = _camera()
    P = OZ.get_projection_mat(100.0, 100.0, 32.0, 24.0, c2w)
    im = OZ.get_projection(P, np.zeros((1, 3)))  # the origin is on the optical axis
    yx = im[:, :2] / -im[:, 2:3]
    assert np.allclose(yx, [[32.0, 24.0]], atol=1e-9)        # (cx, cy)
    assert np.isclose(-im[0, 2], np.linalg.norm(c2w[:3, 3]))  # depth = distance along the axis


def test_update_buffer_known_answers():
    H, W = 4, 6
    z = np.full((H, W), np.inf, dtype=np.float32)
    img = np.zeros((H, W), dtype=np.uint8)
    # three points on pixel (row 1, col 2) at depths 5, 3, 4 (pc[:, 2] = -depth; pc[:, :2] / depth = (col, row))
    pc = np.array([[2 * 5.0, 1 * 5.0, -5.0], [2 * 3.0, 1 * 3.0, -3.0], [2 * 4.0, 1 * 4.0, -4.0]])
    z, img, (vx, vy) = OZ.update_buffer(z, pc, img, label=7)
    assert z[1, 2] == 3.0 and img[1, 2] == 7
    assert list(zip(vx, vy)) == [(1, 2), (1, 2)]  # 5 accepted, 3 accepted, 4 rejected
    # large=True: the last point wins, whatever its depth
    z2 = np.full((H, W), np.inf, dtype=np.float32)
    img2 = np.ones((H, W), dtype=np.uint8)
    OZ.update_buffer(z2, pc, img2, label=0, large=True)
    assert z2[1, 2] == 4.0 and img2[1, 2] == 0
    # clipping: a point far outside lands on the border pixel; rounding is half-to-even
    pc3 = np.array([[1000.0, -1000.0, -1.0], [0.5, 1.5, -1.0], [1.5, 2.5, -1.0]])
    z3 = np.full((H, W), np.inf, dtype=np.float32)
    _, _, (vx, vy) = OZ.update_buffer(z3, pc3, np.zeros((H, W), np.uint8), 1)
    assert list(zip(vx, vy)) == [(0, W - 1), (2, 0), (2, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("n,hw", [(20000, (90, 120)), (777, (1440, 1920)), (0, (8, 8))])
def test_hip_zbuffer_matches_the_sequential_reference(n, hw):
    from cropnerf_amd import ops
    from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M

    H, W = hw
    c2w, rng = _camera(n)
    P = OZ.get_projection_mat(0.06 * W * 10, 0.06 * W * 10, W / 2.0, H / 2.0, c2w)
    tree = rng.normal(size=(max(n, 1), 3)) * 0.5
    tree = tree[:n]
    part = tree[: n // 3] + rng.normal(size=(n // 3, 3)) * 0.05
    # reference
    z_ref = np.full((H, W), np.inf, dtype=np.float32)
    img_ref = np.zeros((H, W), dtype=np.uint8)
    OZ.update_buffer(z_ref, OZ.get_projection(P, tree), img_ref, label=0, large=True)
    z_ref2, img_ref2 = z_ref.copy(), img_ref.copy()
    _, _, (vx, vy) = OZ.update_buffer(z_ref2, OZ.get_projection(P, part), img_ref2, label=3)
    vis_ref = np.zeros((H, W), dtype=np.uint8)
    vis_ref[vx, vy] = 255
    # HIP
    z = torch.full((H, W), float("inf"), dtype=torch.float32, device="cuda")
    img = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    proj = M.get_projection(P, tree, H, W)
    ref_proj = OZ.get_projection(P, tree)
    if n:
        assert np.allclose(proj[2].cpu().numpy(), -ref_proj[:, 2], rtol=1e-13, atol=0)
    M.update_buffer(z, proj, img, label=0, large=True)
    assert np.array_equal(img.cpu().numpy(), img_ref)
    assert np.allclose(z.cpu().numpy(), z_ref, rtol=1e-6, equal_nan=True)
    _, _, vis = M.update_buffer(z, M.get_projection(P, part, H, W), img, label=3)
    assert np.array_equal(vis.cpu().numpy(), vis_ref)
    assert np.array_equal(img.cpu().numpy(), img_ref2)
    assert np.allclose(z.cpu().numpy(), z_ref2, rtol=1e-6, equal_nan=True)


@pytest.mark.gpu
def test_project_super_clusters_end_to_end(tmp_path):
    from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M

    c2w, rng = _camera(5)
    H, W = 120, 160
    tree = rng.normal(size=(5000, 3)) * 0.4
    sem = np.concatenate([rng.normal(size=(400, 3)) * 0.03 + c for c in ([0.2, 0.1, 0.0], [-0.2, 0.0, 0.1], [0.0, -0.2, 0.3])])
    boxes = {i: (np.array(c) - 0.1, np.array(c) + 0.1) for i, c in enumerate(([0.2, 0.1, 0.0], [-0.2, 0.0, 0.1], [0.0, -0.2, 0.3]))}
    clusters = [{"pcd": {0: None, 1: None}, "aabb": {0: boxes[0], 1: boxes[1]}}, {"pcd": {0: None}, "aabb": {0: boxes[2]}}]
    res = M.project_and_save_super_clusters(c2w, clusters, tree, sem, str(tmp_path), cam_idx=4,
                                            intrinsics=(900.0, 900.0, W / 2.0, H / 2.0), height=H, width=W)
    assert set(res) == {0, 1}
    label0, occ0 = res[0]
    assert set(torch.unique(label0).tolist()) <= {0, 1, 2} and int((label0 > 0).sum()) > 0
    assert set(occ0) == {0, 1}
    for f in ("occ_free_0.png", "occ_free_1.png", "visible_label.png", "visible.png"):
        assert (tmp_path / "super_cluster_0" / "cam_4" / f).exists()
    # the reference's loop on the same inputs
    P = OZ.get_projection_mat(900.0, 900.0, W / 2.0, H / 2.0, c2w)
    z0 = np.full((H, W), np.inf, dtype=np.float32)
    i0 = np.zeros((H, W), dtype=np.uint8)
    OZ.update_buffer(z0, OZ.get_projection(P, tree), i0, 0, large=True)
    zb, vl = z0.copy(), i0.copy()
    for sub in (0, 1):
        lo, hi = boxes[sub]
        pc = sem[((sem >= lo) & (sem <= hi)).all(1)]
        OZ.update_buffer(zb, OZ.get_projection(P, pc), vl, sub + 1)
    assert np.array_equal(label0.cpu().numpy(), vl)
