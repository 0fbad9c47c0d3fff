"""GPU tests of the batched semantic projection (``cn_projection_*`` + ``fruit_nerf/projection.py``) against the per-job
path, which mirrors the reference's loop (``fruit_nerf.py:254-318``) call for call and is itself checked against the oracle
in ``tests/test_gpu_model.py``."""

import os

import numpy as np
import pytest
import torch

from _helpers import make_scene
from test_gpu_model import _cameras, _model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    return make_scene(seed=7, log2_T=16, num_images=4, height=96, width=112, focal=140.0, prop_log2_T=13)


def _clusters(scene):
    """Three super-clusters of two sub-cluster boxes: boll-sized boxes inside the frame, one cut by the frame's border, one
    nobody sees, one so small that fewer than 10 rays hit it (``fruit_nerf.py:293``), one that contains camera 2."""
    o = scene.c2w[2, :, 3].numpy()
    return [
        {"aabb": np.array([[[-0.12, -0.10, -0.08], [0.05, 0.09, 0.11]], [[0.10, -0.22, -0.15], [0.26, -0.05, 0.02]]], np.float32)},
        {"aabb": np.array([[[0.35, 0.30, -0.45], [0.60, 0.55, -0.20]], [[5.0, 5.0, 5.0], [6.0, 6.0, 6.0]]], np.float32)},
        {"aabb": np.array([[[-0.004, -0.004, -0.004], [0.004, 0.004, 0.004]], [o - 0.06, o + 0.06]], np.float32)},
    ]


class _DS:
    def __init__(self, cams, files=()):
        from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics

        self.cameras = cams
        self.metadata = {"semantics": Semantics(filenames=list(files))}


@pytest.fixture()
def single_kernel(monkeypatch):
    """The per-job path renders a small box with the single-wave kernel and a batch with the producer/consumer kernel; the two
    agree to an ulp (tested elsewhere).  Bit-for-bit comparisons pin both to one kernel."""
    monkeypatch.setenv("CN_FUSED_SPLIT", "0")


def test_batched_projection_is_the_per_job_path_bit_for_bit(scene, single_kernel):
    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context

    m = _model(scene)
    ds = _DS(_cameras(scene))
    pcd = _clusters(scene)
    with background_color_override_context(torch.zeros(3)):  # scripts/semantic_projection.py:169
        ref = m.get_outputs_for_projections(ds, None, pcd_data=pcd, save=False, batched=False)
        got = m.get_outputs_for_projections(ds, None, pcd_data=pcd, save=False)
    assert sorted(got) == sorted(ref) and len(ref) == 3 * 4 * 2
    lit = 0
    for key in ref:
        for a, b, name in zip(got[key], ref[key], ("wo_occ", "visible")):
            assert a.shape == b.shape == (scene.height, scene.width, 3)
            assert torch.equal(a, b), f"{name} of job {key}: max difference {float((a - b).abs().max()):.3e}"
        lit += int(bool((ref[key][0] != 0).any()))
    assert lit >= 10  # the boxes are seen by most cameras
    # the invisible box and the box with fewer than 10 rays are black in both passes, the camera's own box is not
    for cam in range(4):
        assert float(ref[(1, cam, 1)][0].abs().sum()) == 0 and float(ref[(2, cam, 0)][0].abs().sum()) == 0
    assert float(ref[(2, 2, 1)][0].abs().sum()) > 0
    # the reference's camera index 0 for every camera (fruit_nerf.py:283) through the batched path
    m.compat_projection_cam0 = True
    with background_color_override_context(torch.zeros(3)):
        ref0 = m.get_outputs_for_projections(ds, None, pcd_data=pcd[:1], save=False, batched=False)
        got0 = m.get_outputs_for_projections(ds, None, pcd_data=pcd[:1], save=False)
    assert all(torch.equal(got0[k][0], ref0[k][0]) and torch.equal(got0[k][1], ref0[k][1]) for k in ref0)
    assert not torch.equal(ref0[(0, 3, 0)][0], ref[(0, 3, 0)][0])  # the pose tweak of camera 0 is not camera 3's


def test_batched_projection_in_small_batches_and_default_kernels(scene):
    """Batches cut at any job boundary give the same images; with the default kernel choice the batched values agree with the
    per-job ones to the ulp by which the two render kernels differ."""
    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context
    from cropnerf_amd.fruit_nerf.projection import project_all

    m = _model(scene)
    cams = _cameras(scene)
    pcd = _clusters(scene)
    with background_color_override_context(torch.zeros(3)):
        ref = m.get_outputs_for_projections(_DS(cams), None, pcd_data=pcd, save=False, batched=False)
        one = project_all(m, cams, pcd, want_float=True)
        many = project_all(m, cams, pcd, want_float=True, max_slots=700)
    assert len(one.batches) == 1 and len(many.batches) > 3
    assert one.stats["jobs"] == many.stats["jobs"] == 24 and one.stats["rays"] == many.stats["rays"] > 1000
    a, b = one.float_results(), many.float_results()
    for key in ref:
        for got in (a, b):  # (a small batch renders with the single-wave kernel, like a small job of the per-job loop)
            torch.testing.assert_close(got[key][0], ref[key][0], rtol=2e-5, atol=2e-6)
            hidden_g, hidden_r = (got[key][1] == 0) & (got[key][0] != 0), (ref[key][1] == 0) & (ref[key][0] != 0)
            assert (hidden_g != hidden_r).float().mean() < 1e-3  # an occlusion weight within an ulp of 0.5


def test_batch_boundaries_do_not_change_a_bit(scene, single_kernel):
    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context
    from cropnerf_amd.fruit_nerf.projection import project_all

    m = _model(scene)
    cams = _cameras(scene)
    pcd = _clusters(scene)
    with background_color_override_context(torch.zeros(3)):
        a = project_all(m, cams, pcd, want_float=True).float_results()
        many = project_all(m, cams, pcd, want_float=True, max_slots=700)
    b = many.float_results()
    assert len(many.batches) > 3
    for key in a:
        assert torch.equal(a[key][0], b[key][0]) and torch.equal(a[key][1], b[key][1])


def test_projection_png_tree_matches_the_per_job_tree(scene, single_kernel, tmp_path):
    from PIL import Image

    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context

    masks = []
    for c in range(4):
        p = tmp_path / f"mask_{c}.png"
        Image.fromarray(np.full((scene.height, scene.width), 40 * c, np.uint8)).save(p)
        masks.append(str(p))
    m = _model(scene)
    ds = _DS(_cameras(scene), masks)
    pcd = _clusters(scene)
    with background_color_override_context(torch.zeros(3)):
        m.get_outputs_for_projections(ds, None, pcd_data=pcd, output_root=str(tmp_path / "per_job"), batched=False)
        m.get_outputs_for_projections(ds, None, pcd_data=pcd, output_root=str(tmp_path / "batched"))

    def tree(root):
        return sorted(os.path.relpath(os.path.join(d, f), root) for d, _, fs in os.walk(root) for f in fs)

    files = tree(tmp_path / "per_job")
    assert files == tree(tmp_path / "batched") and len(files) == 3 * 4 * (2 * 2 + 1)
    lit = 0
    for f in files:
        a, b = np.asarray(Image.open(tmp_path / "per_job" / f)), np.asarray(Image.open(tmp_path / "batched" / f))
        assert a.shape == b.shape and a.dtype == b.dtype and (a == b).all(), f
        lit += int(a.any())
    assert lit > 20


def test_projection_run_feeds_the_merger_without_files(scene):
    """``ProjectionRun.images_u8`` = what the PNG round trip would hand the merger (``segmentation/merger.py:219-271``),
    straight from device memory; the merger's image stage gives the same cluster properties either way."""
    from cropnerf_amd.fruit_nerf.fruit_nerf import background_color_override_context
    from cropnerf_amd.segmentation.merger import process_super_cluster, quantise_projection

    m = _model(scene)
    ds = _DS(_cameras(scene))
    pcd = _clusters(scene)
    with background_color_override_context(torch.zeros(3)):
        run = m.get_outputs_for_projections(ds, None, pcd_data=pcd, save=False, return_run=True)
        flt = m.get_outputs_for_projections(ds, None, pcd_data=pcd, save=False)
    assert run.batches[0].wo_occ_f32 is None  # compact results only
    labels = torch.zeros(4, scene.height, scene.width, dtype=torch.uint8)
    labels[:, : scene.height // 2] = 1
    labels[:, scene.height // 2:] = 2
    for i_sc in range(3):
        wo, vis = run.images_u8(i_sc, 2)
        assert wo.shape == vis.shape == (4, 2, scene.height, scene.width) and wo.dtype == torch.uint8
        wo_ref = torch.stack([torch.stack([quantise_projection(flt[(i_sc, c, i)][0]) for i in range(2)]) for c in range(4)])
        vis_ref = torch.stack([torch.stack([quantise_projection(flt[(i_sc, c, i)][1]) for i in range(2)]) for c in range(4)])
        assert torch.equal(wo, wo_ref) and torch.equal(vis, vis_ref)
        a = process_super_cluster(wo, vis, labels, binary_thresh=100, frame_sampling_interval=1)
        b = process_super_cluster(wo_ref, vis_ref, labels, binary_thresh=100, frame_sampling_interval=1)
        for cid in a:
            for k in a[cid]:
                assert np.array_equal(a[cid][k], b[cid][k])


def test_projection_kernels_reproduce_raygen_and_slab_test(scene):
    """``cn_projection_test`` / ``cn_projection_gather`` against ``cn_raygen_pinhole`` + ``cn_intersect_aabb`` for the same
    pixels: the same bits, the same hit set, hit counts, the fewer-than-10 rule, empty rectangles."""
    from cropnerf_amd import ops
    from cropnerf_amd.fruit_nerf.projection import plan_jobs
    from cropnerf_amd.rays import SceneBox

    cams = _cameras(scene)
    pcd = _clusters(scene)
    keys, table = plan_jobs(cams, pcd)
    sizes = table["w"].astype(np.int64) * table["h"]
    table["slot_offset"] = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    P = int(sizes.sum())
    jobs = torch.from_numpy(table.view(np.uint8).reshape(-1).copy()).cuda()
    t = ops.projection_test(jobs, P, scene.width, min_rays=10)
    raw = ops.projection_test(jobs, P, scene.width, min_rays=0)
    hit_slots = t["flags"].nonzero().squeeze(1)
    g = ops.projection_gather(jobs, t["job_of_slot"], hit_slots, scene.width, want_job_pixel=True)
    counts = t["hit_count"].cpu().numpy()
    assert (counts == raw["hit_count"].cpu().numpy()).all()
    seen_small = False
    for j, (i_sc, cam, i) in enumerate(keys):
        rays = cams[cam].generate_rays(camera_indices=0, keep_shape=False,
                                       aabb_box=SceneBox(torch.tensor(pcd[i_sc]["aabb"][i])))
        hit = (rays.nears[:, 0] < 1e10)
        assert int(hit.sum()) == counts[j], (keys[j], int(hit.sum()), counts[j])
        mine = (g["ray_job"] == j).nonzero().squeeze(1)
        if counts[j] < 10:
            seen_small = seen_small or counts[j] > 0
            assert mine.numel() == 0
            continue
        pix = hit.nonzero().squeeze(1)
        assert torch.equal(g["ray_pixel"][mine].long(), pix)  # ascending slots = row-major pixels
        for name in ("origins", "directions", "nears", "fars"):
            assert torch.equal(g[name][mine], getattr(rays, name)[pix]), (keys[j], name)
        assert (g["camera_indices"][mine] == cam).all()
    assert seen_small and (table["w"] == 0).any()
