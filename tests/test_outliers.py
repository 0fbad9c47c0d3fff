"""Statistical outlier removal of the point-cloud exporter (open3d semantics): oracle known answers on CPU, HIP grid search
against the exact KD-tree search on the GPU."""

import numpy as np
import pytest
import torch

from oracle import outliers as OO


def test_oracle_known_answers():
    # a 1-D lattice with spacing 1: interior points have neighbours at 0,1,1,2,2 -> mean of k=5 is 6/5
    pts = np.stack([np.arange(50.0), np.zeros(50), np.zeros(50)], 1)
    avg = OO.knn_mean_distance(pts, 5)
    assert np.allclose(avg[10:40], 6.0 / 5.0)
    assert np.isclose(avg[0], (0 + 1 + 2 + 3 + 4) / 5.0)
    # one far point is the only outlier
    cloud = np.concatenate([np.random.default_rng(0).normal(size=(500, 3)) * 0.1, [[5.0, 5.0, 5.0]]])
    mask, _ = OO.statistical_outlier_mask(cloud, 20, 2.0)
    assert not mask[-1] and mask[:-1].mean() > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(20000, 20), (3000, 8), (50, 20), (15, 20)])
def test_hip_knn_matches_kdtree(n, k):
    from cropnerf_amd import ops

    rng = np.random.default_rng(n)
    # clustered cloud with very uneven density (dense blobs + sparse halo + a few far strays)
    pts = np.concatenate([rng.normal(size=(n // 2, 3)) * 0.02 + [0.2, 0.1, 0.0], rng.normal(size=(n // 3, 3)) * 0.3,
                          rng.uniform(-2, 2, size=(n - n // 2 - n // 3, 3))]).astype(np.float32)
    ref = OO.knn_mean_distance(pts.astype(np.float64), k)
    got = ops.knn_mean_distance(torch.from_numpy(pts).cuda(), k).cpu().numpy()
    assert np.allclose(got, ref, rtol=2e-5, atol=1e-7), float(np.abs(got - ref).max())
    mask_ref, avg = OO.statistical_outlier_mask(pts.astype(np.float64), k, 2.0)
    mask = ops.statistical_outlier_mask(torch.from_numpy(pts).cuda(), k, 2.0).cpu().numpy()
    # identical except for points whose mean distance sits within rounding of the threshold
    assert (mask != mask_ref).sum() <= max(1, n // 5000)


@pytest.mark.gpu
def test_exporter_removes_outliers_without_open3d():
    """generate_point_cloud(remove_outliers=True) -- the reference's default -- uses the HIP path when open3d is absent."""
    from cropnerf_amd import ops

    rng = np.random.default_rng(1)
    pts = torch.from_numpy(np.concatenate([rng.normal(size=(5000, 3)) * 0.05, rng.uniform(-3, 3, size=(20, 3))]).astype(np.float32)).cuda()
    keep = ops.statistical_outlier_mask(pts, 20, 10.0)  # std_ratio 10: the exporter's default
    assert keep[:5000].float().mean() > 0.99 and keep[5000:].float().mean() < 0.5
