"""Super-cluster stage of the segmenter (voxel down-sampling, DBSCAN): oracle known answers on CPU, HIP against the oracle
on the GPU."""

import numpy as np
import pytest
import torch

from oracle import clustering as OC


def _blobs(seed, n):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1, 1, size=(6, 3))
    pts = np.concatenate([rng.normal(size=(n // 8, 3)) * 0.03 + c for c in centres] +
                         [rng.uniform(-1.5, 1.5, size=(n - 6 * (n // 8), 3))])
    return pts.astype(np.float32)


def test_oracle_known_answers():
    # two points in one voxel, one in another
    pts = np.array([[0.0, 0.0, 0.0], [0.01, 0.01, 0.0], [1.0, 1.0, 1.0]])
    out = OC.voxel_down_sample(pts, 0.1)
    out = out[np.lexsort(out.T[::-1])]
    assert np.allclose(out, [[0.005, 0.005, 0.0], [1.0, 1.0, 1.0]])
    # a chain of points 0.09 apart is one cluster at eps 0.1 / min_points 2; a far point is noise
    chain = np.stack([np.arange(10) * 0.09, np.zeros(10), np.zeros(10)], 1)
    labels, core = OC.dbscan(np.concatenate([chain, [[5.0, 5.0, 5.0]]]), 0.1, 2)
    assert (labels[:10] == 0).all() and labels[10] == -1 and core[:10].all() and not core[10]


@pytest.mark.gpu
def test_hip_voxel_down_sample():
    from cropnerf_amd import ops

    pts = _blobs(0, 20000)
    cols = np.random.default_rng(1).uniform(size=(len(pts), 4)).astype(np.float32)
    ref = OC.voxel_down_sample(pts, 0.05)
    out, oc = ops.voxel_down_sample(torch.from_numpy(pts).cuda(), 0.05, torch.from_numpy(cols).cuda())
    got = out.cpu().numpy().astype(np.float64)
    assert got.shape == ref.shape and oc.shape == (len(ref), 4)
    key = lambda a: a[np.lexsort(np.round(a, 4).T[::-1])]
    assert np.allclose(key(got), key(ref), atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,eps,mp", [(30000, 0.03, 30), (5000, 0.05, 8), (40, 0.5, 3),
                                      (16000, 0.12, 30),       # a whole blob within eps: thousands of neighbours per point
                                      (-20000, 0.3, 30), (-20001, 0.3, 30)])  # dense slabs a gap of 0.9 / 1.1 eps apart
def test_hip_dbscan_matches_sklearn(n, eps, mp):
    from scipy.spatial import cKDTree

    from cropnerf_amd import ops

    if n < 0:  # two uniformly filled unit cubes (cells of >= min_points points) that do / do not link up
        rng = np.random.default_rng(-n)
        gap = eps * (0.9 if n % 2 == 0 else 1.1)
        a = rng.uniform(0, 1, size=(-n // 2, 3))
        b = rng.uniform(0, 1, size=(-n // 2, 3)) + [1.0 + gap, 0.0, 0.0]
        pts = np.concatenate([a, b, rng.uniform(-3, 4, size=(200, 3))]).astype(np.float32)
        pts = pts[rng.permutation(len(pts))]
    else:
        pts = _blobs(n, n)
    ref_labels, ref_core = OC.dbscan(pts, eps, mp)
    labels, core = ops.dbscan(torch.from_numpy(pts).cuda(), eps, mp)
    labels, core = labels.cpu().numpy(), core.cpu().numpy()
    assert np.array_equal(core, ref_core)
    assert np.array_equal(labels == -1, ref_labels == -1)          # noise is decided by the definitions alone
    # the core points are partitioned identically, and numbered in the same (scan) order
    assert np.array_equal(labels[core], ref_labels[core])
    # a border point may legitimately belong to any cluster that has a core point within eps of it
    border = (~core) & (labels >= 0)
    tree = cKDTree(pts[core].astype(np.float64))
    core_labels = labels[core]
    for i in np.nonzero(border)[0]:
        near = tree.query_ball_point(pts[i].astype(np.float64), eps * (1 + 1e-6))
        assert labels[i] in set(core_labels[near])


@pytest.mark.gpu
def test_get_super_clusters_pipeline():
    from cropnerf_amd import ops

    pts = _blobs(3, 60000)
    out, labels = ops.get_super_clusters(torch.from_numpy(pts).cuda(), vx_size=0.004)
    assert out.shape[0] == labels.shape[0] > 0 and int(labels.min()) >= 0
    assert 4 <= int(labels.max()) + 1 <= 8   # the six dense blobs (sparse background is noise)


@pytest.mark.gpu
def test_segmenter_mirror_feeds_the_projections(tmp_path):
    """export -> segmenter -> projection: process_and_save_all builds the all_super_cluster_info list that the depth-based
    projection (and FruitModel.get_outputs_for_projections) consume."""
    from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M
    from cropnerf_amd.segmentation import segmenter as SG

    pts = _blobs(5, 40000) * 0.4
    path = tmp_path / "all_super_cluster_info_nsub_2.npy"
    info = SG.process_and_save_all(pts, k=2, save_path=str(path), vx_size=0.002)
    assert len(info) >= 4 and path.exists()
    loaded = np.load(str(path), allow_pickle=True)
    assert len(loaded) == len(info)
    sizes = [sum(len(p) for p in sc["pcd"].values()) for sc in info]
    assert sizes == sorted(sizes, reverse=True)            # largest super-cluster first
    for sc in info:
        assert sc["aabb"].shape == (2, 2, 3)
        for i, p in sc["pcd"].items():
            assert (p >= sc["aabb"][i][0] - 1e-9).all() and (p <= sc["aabb"][i][1] + 1e-9).all()
    c2w = np.eye(4)
    c2w[:3, 3] = [0.0, 0.0, 2.0]
    res = M.project_and_save_super_clusters(c2w, list(loaded), pts, pts, None, intrinsics=(300.0, 300.0, 80.0, 60.0),
                                            height=120, width=160)
    assert len(res) == len(info) and all(int((lab > 0).sum()) > 0 for lab, _ in res.values())


@pytest.mark.gpu
def test_clustering_edge_cases():
    """Empty clouds, all-noise clouds, a single point, duplicates, and a cloud that is one cluster."""
    from cropnerf_amd import ops
    from cropnerf_amd.segmentation import segmenter as SG

    empty = torch.empty(0, 3, device="cuda")
    labels, core = ops.dbscan(empty, 0.1, 5)
    assert labels.shape == (0,) and core.shape == (0,)
    down, cols = ops.voxel_down_sample(empty, 0.1)
    assert down.shape == (0, 3) and cols is None
    # one point: noise unless min_points == 1
    one = torch.zeros(1, 3, device="cuda")
    assert ops.dbscan(one, 0.1, 2)[0].tolist() == [-1]
    assert ops.dbscan(one, 0.1, 1)[0].tolist() == [0]
    # widely spaced points: all noise
    rng = np.random.default_rng(0)
    sparse = torch.from_numpy(rng.uniform(-10, 10, size=(500, 3)).astype(np.float32)).cuda()
    labels, core = ops.dbscan(sparse, 0.01, 3)
    assert int((labels >= 0).sum()) == 0 and not bool(core.any())
    # 100 copies of the same point: one cluster, all core (distance 0 <= eps)
    dup = torch.ones(100, 3, device="cuda") * 0.25
    labels, core = ops.dbscan(dup, 1e-3, 50)
    assert bool(core.all()) and labels.unique().tolist() == [0]
    assert ops.voxel_down_sample(dup, 0.1)[0].shape == (1, 3)
    # a filled ball: one cluster, identical to the oracle's labelling
    ball = rng.normal(size=(4000, 3)).astype(np.float32) * 0.05
    ref_labels, ref_core = OC.dbscan(ball, 0.05, 10)
    labels, core = ops.dbscan(torch.from_numpy(ball).cuda(), 0.05, 10)
    assert np.array_equal(core.cpu().numpy(), ref_core) and np.array_equal(labels.cpu().numpy()[ref_core], ref_labels[ref_core])
    # the segmenter skips super-clusters that have no more points than sub-clusters (segmenter.py:163-164)
    few = np.concatenate([ball[:60] * 0.01, ball[:60] * 0.01 + 5.0])
    assert SG.process_and_save_all(few, k=100, vx_size=1e-4) == []


@pytest.mark.gpu
def test_segmenter_process_for_pipeline(tmp_path):
    """segmenter.py's entry point: exported PLY in, all_super_cluster_info_nsub_2.npy out (segmenter.py:183-185)."""
    from cropnerf_amd.fruit_nerf.ply import write_ply
    from cropnerf_amd.segmentation import segmenter as SG

    pts = (_blobs(7, 30000) * 0.4).astype(np.float64)
    write_ply(str(tmp_path / "semantics_pc.ply"), pts, np.full_like(pts, 0.5))
    out = SG.process_for_pipeline(str(tmp_path), "semantics_pc.ply", k=2, vx_size=0.002)
    info = np.load(out, allow_pickle=True)
    assert out.endswith("all_super_cluster_info_nsub_2.npy") and len(info) >= 4 and info[0]["aabb"].shape == (2, 2, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 3, 5, 10])
def test_device_kmeans_labels_are_scikit_learns(k):
    """segmentation/segmenter.py:28-45: the sub-cluster labels of ``ops.kmeans`` (Lloyd iterations in ``cn_kmeans_step``,
    scikit-learn's k-means++ seeding with random_state=0 on the host) against
    ``KMeans(init="k-means++", n_clusters=k, n_init="auto", random_state=0)`` itself -- identical, point for point."""
    from sklearn.cluster import KMeans

    from cropnerf_amd import ops
    from cropnerf_amd.segmentation import segmenter

    rng = np.random.default_rng(100 + k)
    for n, spread in ((400, 0.02), (6000, 0.05), (50000, 0.03)):
        # a fruit-sized super-cluster: a few overlapping lumps (float32 coordinates, as the exporter writes them)
        centres = rng.normal(size=(max(k, 3), 3)) * 0.05
        pts = (centres[rng.integers(0, len(centres), n)] + rng.normal(size=(n, 3)) * spread).astype(np.float32)
        ref = KMeans(init="k-means++", n_clusters=k, n_init="auto", random_state=0).fit(pts.astype(np.float64)).labels_
        got = ops.kmeans(torch.from_numpy(pts).cuda(), k).cpu().numpy()
        assert got.shape == ref.shape and set(got.tolist()) == set(range(k))
        same = (got == ref).mean()
        assert same == 1.0, f"k={k} n={n}: {100 * same:.3f} % of the labels agree"
        # through the mirror's entry point (what process_and_save_all calls), from float64 input as there
        assert np.array_equal(segmenter.cluster_kmeans(pts.astype(np.float64), k), ref)
    with pytest.raises(ValueError):
        ops.kmeans(torch.zeros(3, 3).cuda(), 5)
