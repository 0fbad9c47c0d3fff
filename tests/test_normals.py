"""Normals of the exported cloud (``generate_point_cloud``'s ``pcd.estimate_normals()`` + re-orientation,
``fruit_nerf/export/exporter_utils_nerfacto.py:203-225``): ``cn_estimate_normals`` against the oracle's restatement of open3d
(exact k-nearest search + ``numpy.linalg.eigh``), and analytic cases for the oracle itself."""

import numpy as np
import pytest
import torch

from oracle import normals as ON


def test_oracle_on_analytic_surfaces():
    g = np.random.default_rng(0)
    # a plane z = 0.3 x - 0.2 y + 1: every neighbourhood's normal is the plane's
    xy = g.uniform(-1, 1, size=(4000, 2))
    pts = np.concatenate([xy, (0.3 * xy[:, :1] - 0.2 * xy[:, 1:] + 1.0)], axis=1)
    n, deg, gap = ON.estimate_normals(pts, 30)
    want = np.array([-0.3, 0.2, 1.0]) / np.linalg.norm([-0.3, 0.2, 1.0])
    assert not deg.any() and np.abs(np.abs(n @ want) - 1).max() < 1e-9 and gap.min() > 1e-3
    # a sphere: normals are radial (up to the curvature inside a 30-point patch)
    p = g.normal(size=(20000, 3))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    n, _, _ = ON.estimate_normals(p, 30)
    assert np.median(np.abs(np.sum(n * p, axis=1))) > 0.999
    # fewer than three points, coincident points: (0, 0, 1), flagged
    n, deg, _ = ON.estimate_normals(np.zeros((2, 3)), 30)
    assert deg.all() and (n == [0, 0, 1]).all()
    n, deg, _ = ON.estimate_normals(np.ones((40, 3)), 30)
    assert deg.all() and (n == [0, 0, 1]).all()
    # re-orientation: a normal pointing along the view direction is flipped, float32 round trip
    nn, mask = ON.reorient_normals(np.array([[0, 0, 1.0], [0, 0, 1.0], [1.0, 0, 0]]),
                                   np.array([[0, 0, 1.0], [0, 0, -1.0], [0, 1.0, 0]]))
    assert mask.tolist() == [True, False, False] and nn[0].tolist() == [0, 0, -1] and nn[1].tolist() == [0, 0, 1]


def _cloud(n, seed):
    """Bumpy shells and a sheet, fruit-sized: what a kept-point cloud looks like (curved surfaces, varying density)."""
    g = np.random.default_rng(seed)
    a = g.normal(size=(n // 2, 3))
    a = a / np.linalg.norm(a, axis=1, keepdims=True) * (0.3 + 0.02 * np.sin(9 * a[:, :1])) + [0.2, -0.1, 0.05]
    uv = g.uniform(-0.8, 0.8, size=(n - n // 2, 2))
    b = np.concatenate([uv, 0.1 * np.sin(4 * uv[:, :1]) * np.cos(3 * uv[:, 1:]) - 0.5], axis=1)
    pts = np.concatenate([a, b]) + g.normal(scale=2e-4, size=(n, 3))
    return pts.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [20000, 3000, 50])
def test_estimate_normals_matches_the_oracle(n):
    from cropnerf_amd import ops

    pts = _cloud(n, seed=n)
    ref, ref_deg, gap = ON.estimate_normals(pts.astype(np.float64), 30)
    got, deg = ops.estimate_normals(torch.from_numpy(pts).cuda(), 30)
    got, deg = got.cpu().numpy(), deg.cpu().numpy()
    assert got.dtype == np.float64 and (deg == ref_deg).all() and not deg.any()
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-12
    # neighbourhoods whose two smallest eigenvalues nearly coincide have no defined normal: excluded, and counted
    well = gap > 1e-6
    assert well.mean() > 0.999, f"{(~well).sum()} ill-conditioned neighbourhoods of {n}"
    dot = np.abs(np.sum(got * ref, axis=1))
    frac = float((dot[well] >= 1 - 1e-5).mean())
    assert frac >= 0.999, f"|n . n_ref| >= 1 - 1e-5 on {frac:.5f} of {well.sum()} points (worst {dot[well].min():.3e})"
    # re-orientation against the view directions (the camera looks along +d; the normal must face it): flags exact wherever the
    # test is not decided by the last bits of a float32 dot product
    g = np.random.default_rng(1)
    views = g.normal(size=(n, 3)).astype(np.float32)
    views /= np.linalg.norm(views, axis=1, keepdims=True)
    signed = np.where(np.sum(got * ref, axis=1, keepdims=True) < 0, -ref, ref)  # the oracle's normals with the kernel's signs
    want, want_mask = ON.reorient_normals(signed, views)
    out, mask = ops.reorient_normals(torch.from_numpy(got).cuda(), torch.from_numpy(views).cuda())
    out, mask = out.cpu().numpy(), mask.cpu().numpy()
    decided = well & (np.abs(np.sum(views * signed.astype(np.float32), axis=1)) > 1e-4)
    assert decided.mean() > 0.99 and (mask[decided] == want_mask[decided]).all()
    assert (np.sum(out * views, axis=1) <= 1e-6).all()  # no normal points away from its camera afterwards
    assert np.abs(out[decided] - want[decided]).max() < 2e-3 and out.dtype == np.float64


@pytest.mark.gpu
def test_estimate_normals_degenerate_and_empty_inputs():
    from cropnerf_amd import ops

    n, d = ops.estimate_normals(torch.zeros(0, 3, device="cuda"))
    assert n.shape == (0, 3) and d.shape == (0,)
    n, d = ops.estimate_normals(torch.tensor([[0.0, 0, 0], [1.0, 0, 0]], device="cuda"))  # two points: fewer than three neighbours
    assert d.all() and (n.cpu() == torch.tensor([0.0, 0, 1.0], dtype=torch.float64)).all()
    n, d = ops.estimate_normals(torch.ones(64, 3, device="cuda"))  # coincident points: zero covariance
    assert d.all() and (n.cpu() == torch.tensor([0.0, 0, 1.0], dtype=torch.float64)).all()
    # an exact plane aligned with the axes (diagonal covariance: the solver's third branch)
    g = torch.Generator().manual_seed(0)
    p = torch.rand(500, 3, generator=g)
    p[:, 1] = 0.25
    n, d = ops.estimate_normals(p.cuda())
    assert not d.any() and (n.cpu().abs() - torch.tensor([0.0, 1.0, 0.0], dtype=torch.float64)).abs().max() < 1e-9
