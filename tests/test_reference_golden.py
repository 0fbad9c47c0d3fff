"""Tests against ``tests/golden/reference_functions.npz``: outputs of the REFERENCE'S OWN functions, produced by
``tests/golden/make_golden_reference.py`` (which executes their definitions, extracted with ``ast`` from
``/root/reference``, in the build container).  These vectors pin

* ``oracle/zbuffer.py`` and the HIP depth projection / z-buffer kernels
  (``fruit_nerf/scripts/depth_based_semantic_projection.py:31-49,84-105``),
* ``oracle/rays.py`` ``corners_of_aabb`` / ``surface_points`` and the product's surface grid (``cn_surface_grid``;
  ``fruit_nerf/data/fruit_datamanager.py:42-121``),
* the merger's ``calc_affinity`` / ``get_component`` host mirrors (``segmentation/merger.py:26-74,335-355``),
* (round 3) ``oracle/samplers.py: spaced_sampler`` and ``cn_sample_spaced`` -- the reference's own
  ``UniformSamplerWithNoise.generate_ray_samples`` (``fruit_nerf/components/ray_samplers.py:54-104``) in eval and with both
  kinds of training jitter -- and ``oracle/rays.py: ortho_rays`` / ``cn_raygen_ortho`` against the reference's
  ``OrthographicRayGenerator`` (``fruit_nerf/components/ray_generators.py:22-66``).
* (round 5) STATEMENT BLOCKS of the hot path's own functions, executed from the reference's AST on seeded tensors: the
  semantics colormap of ``get_outputs`` / ``get_export_outputs`` and the dataparser's colour table (a13:
  ``fruit_nerf.py:594-597,488-492``, ``data/cotton_nerf_dataparser.py:248-254``), ``get_loss_dict``'s two data terms (a18:
  ``:178,603-608``), ``sample_volume``'s masks and point sets (a16: ``export/exporter_utils.py:96-153``),
  ``generate_point_cloud``'s point / mask step and the re-orientation of the normals (a17:
  ``export/exporter_utils_nerfacto.py:156-180,221-225``), and ``cluster_kmeans`` (f2: ``segmentation/segmenter.py:28-45``) --
  against the oracle on the CPU and against the HIP kernels on the device.
"""

import os
import random

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "reference_functions.npz"))


def _dense(idx, val, shape, fill, dtype):
    a = np.full(shape, fill, dtype=dtype)
    a[idx[:, 0], idx[:, 1]] = val
    return a


# ------------------------------------------------------------------------------------------ oracle vs reference (CPU)
def test_oracle_projection_and_zbuffer_reproduce_the_reference(gold):
    from oracle import zbuffer as OZ

    for c in range(int(gold["num_proj"])):
        k = f"proj{c}"
        fx, fy, cx, cy = gold[f"{k}/intrinsics"]
        P = OZ.get_projection_mat(fx, fy, cx, cy, gold[f"{k}/c2w"])
        assert np.array_equal(P, gold[f"{k}/P"])
        z = np.ones((1440, 1920), dtype=np.float32) * 1e10
        img = np.zeros((1440, 1920), dtype=np.uint8)
        for name in ("tree", "c1", "c2"):
            im = OZ.get_projection(P, gold[f"{k}/{name}/points"])
            assert np.array_equal(im, gold[f"{k}/{name}/im"])
            z, img, (vx, vy) = OZ.update_buffer(z, im, img, int(gold[f"{k}/{name}/label"]), bool(gold[f"{k}/{name}/large"]))
            assert np.array_equal(np.asarray(vx), gold[f"{k}/{name}/visible_xs"])
            assert np.array_equal(np.asarray(vy), gold[f"{k}/{name}/visible_ys"])
            assert np.array_equal(img, _dense(gold[f"{k}/{name}/img_idx"], gold[f"{k}/{name}/img_val"], img.shape, 0, np.uint8))
            assert np.array_equal(z, _dense(gold[f"{k}/{name}/z_idx"], gold[f"{k}/{name}/z_val"], z.shape, 1e10, np.float32))


def test_oracle_surface_grid_reproduces_the_reference(gold):
    from oracle import rays as ORY

    for i in range(int(gold["num_dm"])):
        aabb = torch.from_numpy(gold[f"dm{i}/aabb"])
        corners = ORY.corners_of_aabb(aabb)
        assert np.array_equal(corners.numpy(), gold[f"dm{i}/corners"])
        pts, plane = ORY.surface_points(corners, int(gold[f"dm{i}/n"]))
        assert np.array_equal(pts.numpy(), gold[f"dm{i}/points"])
        assert np.array_equal(plane.numpy(), gold[f"dm{i}/plane"])


def _sampler_case(gold, c):
    k = f"us{c}"
    S = int(gold[f"{k}/num_samples"])
    t = torch.from_numpy(gold[f"{k}/t_rand"]) if bool(gold[f"{k}/training"]) else None
    return k, S, torch.from_numpy(gold[f"{k}/nears"]), torch.from_numpy(gold[f"{k}/fars"]), t


def test_oracle_uniform_sampler_reproduces_the_reference(gold):
    """``UniformSamplerWithNoise.generate_ray_samples`` executed from the reference's source: bins, stratified jitter (single
    and per-bin) and the spacing -> euclidean map, bit for bit."""
    from oracle import rays as ORY
    from oracle import samplers as OSM

    assert int(gold["num_us"]) == 6
    for c in range(int(gold["num_us"])):
        k, S, nears, fars, t = _sampler_case(gold, c)
        R = nears.shape[0]
        rb = ORY.RayBundle(torch.zeros(R, 3), torch.zeros(R, 3), torch.zeros(R, 1), None, nears, fars)
        rs = OSM.spaced_sampler(rb, S, "uniform", t_rand=t)
        assert np.array_equal(rs.starts.numpy(), gold[f"{k}/bin_starts"]), k
        assert np.array_equal(rs.ends.numpy(), gold[f"{k}/bin_ends"]), k
        assert np.array_equal(rs.spacing_starts.expand(R, S, 1).numpy(), gold[f"{k}/spacing_starts"]), k
        assert np.array_equal(rs.spacing_ends.expand(R, S, 1).numpy(), gold[f"{k}/spacing_ends"]), k
        x = torch.linspace(0, 1, 5)[None, :].expand(R, 5)
        assert np.array_equal((x * fars + (1 - x) * nears).numpy(), gold[f"{k}/s2e_at_quarters"]), k


def test_oracle_ortho_rays_reproduce_the_reference(gold):
    from oracle import rays as ORY

    assert int(gold["num_og"]) >= 6
    for c in range(int(gold["num_og"])):
        k = f"og{c}"
        rb = ORY.ortho_rays(torch.from_numpy(gold[f"{k}/points"]), torch.from_numpy(gold[f"{k}/plane"]), int(gold[f"{k}/batch"]),
                            int(gold[f"{k}/count"]))
        for name in ("origins", "directions", "pixel_area", "nears", "fars"):
            assert np.array_equal(getattr(rb, name).numpy(), gold[f"{k}/{name}"]), (k, name)


def test_host_corners_mirror_reproduces_the_reference(gold):
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import get_corners_of_aabb

    for i in range(int(gold["num_dm"])):
        assert np.array_equal(get_corners_of_aabb(torch.from_numpy(gold[f"dm{i}/aabb"])).numpy(), gold[f"dm{i}/corners"])


def test_merger_mirrors_reproduce_the_reference(gold):
    from cropnerf_amd.segmentation import merger

    for c in range(int(gold["num_mg"])):
        lab, rel = gold[f"mg{c}/labels"], gold[f"mg{c}/reliability"]
        prop = {i: {"label": lab[i], "reliability": rel[i]} for i in range(lab.shape[0])}
        aff = merger.calc_affinity(prop)
        assert np.array_equal(aff, gold[f"mg{c}/affinity"])
        for algo in ("clique", "bridge", "community"):
            random.seed(35)
            count, labels = merger.get_component(gold[f"mg{c}/affinity"].copy(), algo)
            assert count == int(gold[f"mg{c}/{algo}/count"]), (c, algo)
            assert np.array_equal(np.asarray(labels), gold[f"mg{c}/{algo}/labels"]), (c, algo)


# ------------------------------------------------------------------------------------------ HIP kernels vs reference
@pytest.mark.gpu
def test_hip_projection_and_zbuffer_reproduce_the_reference(gold):
    """``cn_depth_project`` + ``cn_zbuffer_update_large`` / ``cn_zbuffer_update`` through the host mirror against the
    reference's ``get_projection`` / ``update_buffer`` at its own 1440 x 1920 size: label images and the set of touched
    pixels exact; depths exact (float32 of the same float64 products)."""
    from cropnerf_amd.fruit_nerf.scripts import depth_based_semantic_projection as M

    for c in range(int(gold["num_proj"])):
        k = f"proj{c}"
        fx, fy, cx, cy = gold[f"{k}/intrinsics"]
        P = M.get_projection_mat(fx, fy, cx, cy, gold[f"{k}/c2w"])
        assert np.array_equal(P, gold[f"{k}/P"])
        z = torch.full((1440, 1920), 1e10, dtype=torch.float32, device="cuda")
        img = torch.zeros((1440, 1920), dtype=torch.uint8, device="cuda")
        for name in ("tree", "c1", "c2"):
            pc = M.get_projection(P, gold[f"{k}/{name}/points"])
            large = bool(gold[f"{k}/{name}/large"])
            z, img, vis = M.update_buffer(z, pc, img, int(gold[f"{k}/{name}/label"]), large)
            want_img = _dense(gold[f"{k}/{name}/img_idx"], gold[f"{k}/{name}/img_val"], (1440, 1920), 0, np.uint8)
            want_z = _dense(gold[f"{k}/{name}/z_idx"], gold[f"{k}/{name}/z_val"], (1440, 1920), 1e10, np.float32)
            assert np.array_equal(img.cpu().numpy(), want_img), (k, name)
            got_z = z.cpu().numpy()
            assert np.array_equal(got_z != np.float32(1e10), want_z != np.float32(1e10))
            assert np.allclose(got_z, want_z, rtol=2e-7, atol=0), (k, name)
            if not large:  # the pixels the sequential loop accepted (as a set: it lists a pixel once per accepted point)
                want_vis = np.zeros((1440, 1920), dtype=bool)
                want_vis[gold[f"{k}/{name}/visible_xs"], gold[f"{k}/{name}/visible_ys"]] = True
                assert np.array_equal(vis.cpu().numpy() != 0, want_vis), (k, name)


@pytest.mark.gpu
def test_hip_surface_grid_reproduces_the_reference(gold):
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import get_corners_of_aabb, sample_surface_points

    for i in range(int(gold["num_dm"])):
        corners = get_corners_of_aabb(torch.from_numpy(gold[f"dm{i}/aabb"]))
        pts, plane = sample_surface_points(corners, int(gold[f"dm{i}/n"]), "cuda")
        want = gold[f"dm{i}/points"]
        assert tuple(pts.shape) == want.shape
        # torch.linspace on the host vs the kernel's lerp: the last bit of an interior grid coordinate may differ
        assert np.allclose(pts.cpu().numpy(), want, rtol=0, atol=2.4e-7)
        assert np.array_equal(plane.numpy(), gold[f"dm{i}/plane"])


@pytest.mark.gpu
def test_hip_uniform_sampler_reproduces_the_reference(gold):
    """``cn_sample_spaced`` against the reference's own ``generate_ray_samples``: eval bins, single and per-bin jitter.  The
    kernel evaluates linspace and the lerp in its own order: 1e-6 absolute on values in [0, 3.5] (the sampler bar of
    DESIGN.md section 2 is 1e-5 relative)."""
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops

    for c in range(int(gold["num_us"])):
        k, S, nears, fars, t = _sampler_case(gold, c)
        out = ops.sample_spaced(nears.cuda(), fars.cuda(), S, L.SPACING_UNIFORM, None if t is None else t.cuda().contiguous())
        for mine, ref in (("starts", "bin_starts"), ("ends", "bin_ends"), ("spacing_starts", "spacing_starts"),
                          ("spacing_ends", "spacing_ends")):
            got, want = out[mine].cpu().numpy(), gold[f"{k}/{ref}"][..., 0]
            assert got.shape == want.shape
            err = np.abs(got - want).max()
            assert err <= 1e-6, f"{k} {mine}: max abs err {err:.3e}"


@pytest.mark.gpu
def test_hip_ortho_raygen_reproduces_the_reference(gold):
    from cropnerf_amd import ops

    for c in range(int(gold["num_og"])):
        k = f"og{c}"
        pts = torch.from_numpy(gold[f"{k}/points"]).cuda()
        batch, count = int(gold[f"{k}/batch"]), int(gold[f"{k}/count"])
        start = batch * (count - 1)
        n = min(batch * count, pts.shape[0]) - start
        out = ops.raygen_ortho(pts, [float(v) for v in gold[f"{k}/plane"][0]], start, n)
        assert np.array_equal(out["origins"].cpu().numpy(), gold[f"{k}/origins"])
        assert np.array_equal(out["pixel_area"].cpu().numpy(), gold[f"{k}/pixel_area"])
        assert np.array_equal(out["nears"].cpu().numpy(), gold[f"{k}/nears"])
        for name in ("directions", "fars"):
            err = np.abs(out[name].cpu().numpy() - gold[f"{k}/{name}"]).max()
            assert err <= 2.4e-7, f"{k} {name}: {err:.3e}"  # normalize / norm: one fp32 ulp of a value <= 2


def test_schedule_mirrors_reproduce_the_references_closures(gold):
    """``update_schedule`` (``fruit_nerf.py:144-149``) and ``set_anneal`` / ``bias`` (``:206-216``), the two closures of the hot
    path that are the reference's own code, executed from source by ``make_golden_reference.py``: the mirrors in
    ``fruit_nerf/schedules.py`` (used by ``FruitTrainer`` and ``FruitModel.get_training_callbacks``) give the same numbers."""
    from cropnerf_amd.fruit_nerf.schedules import proposal_update_schedule, proposal_weights_anneal

    warmup, every = (int(v) for v in gold["sched_config"])
    got = [proposal_update_schedule(int(s), warmup, every) for s in gold["sched_steps"]]
    assert np.array_equal(np.array(got), gold["sched_values"])
    slope, n_iters = float(gold["anneal_config"][0]), int(gold["anneal_config"][1])
    got = [proposal_weights_anneal(int(s), n_iters, slope) for s in gold["anneal_steps"]]
    assert np.array_equal(np.array(got), gold["anneal_values"])
    # the defaults the vectors were made with are the model config's
    from cropnerf_amd.config import FruitNerfModelConfig

    c = FruitNerfModelConfig()
    assert (c.proposal_warmup, c.proposal_update_every) == (warmup, every)
    assert (c.proposal_weights_anneal_slope, c.proposal_weights_anneal_max_num_iters) == (slope, n_iters)


@pytest.mark.gpu
def test_trainer_and_sampler_follow_the_references_schedules(gold):
    """The same vectors through the product: ``FruitTrainer.set_anneal`` / ``FruitModel.get_training_callbacks`` set the
    reference's annealing exponents, ``proposal_update_due`` follows ``update_schedule``, and ``cn_proposal_sample``'s ``anneal``
    input resamples on ``weights ** anneal`` as the oracle does for each of those exponents."""
    from _helpers import assert_close, dev_params, make_scene, product_specs, rays_with_box, to_dev
    from cropnerf_amd import ops
    from cropnerf_amd.config import FruitNerfModelConfig
    from cropnerf_amd.fruit_nerf.fruit_nerf import FruitModel, Semantics
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer
    from cropnerf_amd.rays import SceneBox
    from oracle import samplers as OSM

    sc = make_scene(seed=2, log2_T=14, num_images=3, height=16, width=16, focal=22.0, prop_log2_T=12)
    pl = [{"hidden_dim": 16, "log2_hashmap_size": p.grid.log2_hashmap_size, "num_levels": 5, "max_res": p.grid.max_res}
          for p in sc.pspecs]
    model = FruitModel(FruitNerfModelConfig(log2_hashmap_size=14, proposal_net_args_list=pl), SceneBox(sc.aabb), 3,
                       {"semantics": Semantics()}, device="cuda", params=sc.params)
    tr = FruitTrainer(model)
    for s, a in zip(gold["anneal_steps"], gold["anneal_values"]):
        tr.set_anneal(int(s))
        assert model._anneal == float(a)
    cb = model.get_training_callbacks()[0]
    for s, a in zip(gold["anneal_steps"], gold["anneal_values"]):
        cb.run_callback(int(s))
        assert model._anneal == float(a)
    for s, v in zip(gold["sched_steps"], gold["sched_values"]):
        for since in (1, 2, 3, 5, 6):
            tr._steps_since_update = since
            assert tr.proposal_update_due(int(s)) == (since > float(v) or int(s) < 10), (int(s), since)
    # the sampler's anneal input, at three of the reference's exponents
    from oracle import field as OF

    fspec, pspecs = product_specs(sc)
    dp = dev_params(sc)
    dh = [ops.DensityHandle(dp, i, ps) for i, ps in enumerate(pspecs)]
    rb = rays_with_box(sc, 1, 200)
    scene = ops.scene_struct(sc.aabb, True)
    dens = [lambda pos, i=i: OF.proposal_density(pos, sc.params, i, sc.pspecs[i], sc.aabb, True) for i in range(2)]
    for a in (float(gold["anneal_values"][1]), float(gold["anneal_values"][4]), 1.0):
        ref = OSM.proposal_sampler(rb, dens, (64, 32), 24, anneal=a)
        got = ops.proposal_sample(dh, scene, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars),
                                  (64, 32), 24, anneal=a)
        ref_eu = torch.cat([ref[0].starts[..., 0], ref[0].ends[:, -1:, 0]], -1)
        assert_close(got["euclidean_bins"], ref_eu, 2e-3, 2e-4, f"final bins at anneal {a:.4f}", frac_ok=0.99)


# ------------------------------------------------------------------------------------------ round 5: statement blocks
def _rows_sorted(a):
    a = np.asarray(a)
    return a[np.lexsort(a.T[::-1])]


def test_oracle_colormap_reproduces_the_reference(gold):
    from oracle import render as OR

    colors = torch.from_numpy(gold["cm_colors"])
    assert list(gold["cm_classes"]) == ["apple", "stuff"] and colors.tolist() == [0.0, 1.0]
    got = OR.semantics_colormap(torch.from_numpy(gold["cm_logits"]), colors)
    assert np.array_equal(got.numpy(), gold["cm_colormap"])  # every logit, the threshold ln 9 and its neighbours included
    assert set(np.unique(gold["cm_colormap"])) == {0.0, 1.0}
    lab = OR.semantics_colormap(torch.from_numpy(gold["cm_export_logits"])[..., None], repeat3=False)[..., 0]
    assert np.array_equal(lab.numpy().astype(np.int64), gold["cm_export_labels"])
    # the product's Semantics carries the same table (data/cotton_nerf_dataparser.py:248-254)
    from cropnerf_amd.fruit_nerf.fruit_nerf import Semantics

    sm = Semantics()
    assert list(sm.classes) == ["apple", "stuff"] and sm.colors.tolist() == colors.tolist()


def test_oracle_data_losses_reproduce_the_reference(gold):
    from oracle import losses as OL

    ld = OL.data_losses({"rgb": torch.from_numpy(gold["loss_rgb"]), "semantics": torch.from_numpy(gold["loss_sem"])},
                        torch.from_numpy(gold["loss_image"]), torch.from_numpy(gold["loss_mask"]), 1.0)
    assert float(ld["rgb_loss"]) == gold["loss_values"][0] and float(ld["semantics_loss"]) == gold["loss_values"][1]


def test_oracle_export_masks_reproduce_the_reference(gold):
    from oracle import model as OM

    outputs = {k: torch.from_numpy(gold[f"sv_in_{k}"]) for k in ("point_location", "semantics", "semantics_colormap", "density", "rgb")}
    res = OM.sample_volume_masks(outputs)
    for name in ("semantic_colormap", "semantic", "density"):
        assert np.array_equal(res[name]["points"].numpy(), gold[f"sv_{name}_points"]), name
        assert np.array_equal(res[name]["colors"].numpy(), gold[f"sv_{name}_colors"]), name
    assert gold["sv_semantic_points"].shape[0] > 0 and gold["sv_density_points"].shape[0] > gold["sv_semantic_points"].shape[0]


def test_oracle_pointcloud_step_and_reorientation_reproduce_the_reference(gold):
    from oracle import model as OM
    from oracle import normals as ON
    from oracle import rays as ORY

    R = gold["pc_origins"].shape[0]
    rb = ORY.RayBundle(torch.from_numpy(gold["pc_origins"]), torch.from_numpy(gold["pc_directions"]), torch.zeros(R, 1), None)
    pts, rgb, dirs = OM.pointcloud_from_outputs(rb, {"depth": torch.from_numpy(gold["pc_depth"]),
                                                    "semantics_colormap": torch.from_numpy(gold["pc_colormap"]),
                                                    "rgb": torch.from_numpy(gold["pc_rgba"])[:, :3]})
    assert np.array_equal(pts.numpy(), gold["pc_points"]) and np.array_equal(rgb.numpy(), gold["pc_rgbs"])
    assert np.array_equal(dirs.numpy(), gold["pc_view_directions"])
    out, flipped = ON.reorient_normals(gold["pc_normals_in"], gold["pc_view_directions"])
    assert np.array_equal(out, gold["pc_normals_out"])
    assert flipped[0] and not flipped[1] and not flipped[2]  # along the view: flipped; against it, or zero: kept


@pytest.mark.gpu
def test_hip_colormap_reproduces_the_reference(gold):
    """The composite epilogue's colormap (``cn_composite``) on the reference's own logits: one sample per ray with weight exactly 1
    (density 1000 over a unit interval), so the rendered semantic logit IS the input logit."""
    from cropnerf_amd import ops

    logits = torch.from_numpy(gold["cm_logits"]).cuda()
    R = logits.shape[0]
    out = ops.composite(torch.zeros(R, 1, device="cuda"), torch.ones(R, 1, device="cuda"), torch.full((R, 1), 1000.0, device="cuda"),
                        rgb=torch.zeros(R, 1, 3, device="cuda"), semantics=logits.reshape(R, 1, 1).contiguous())
    assert torch.equal(out["semantics"].cpu(), torch.from_numpy(gold["cm_logits"]))
    got, want = out["semantics_colormap"].cpu().numpy(), gold["cm_colormap"]
    # the device's exp is not torch's to the last bit: a logit within 2 ulp of the threshold may fall on the other side
    ln9 = float(np.log(9.0))
    clear = np.abs(gold["cm_logits"][:, 0] - ln9) > 1e-6
    assert clear.sum() >= 250 and np.array_equal(got[clear], want[clear])
    assert set(np.unique(got)) <= {0.0, 1.0}


@pytest.mark.gpu
def test_hip_export_compaction_reproduces_the_reference(gold):
    """``cn_export_compact`` on the reference's own seeded model outputs: the three point sets of ``sample_volume``
    (``exporter_utils.py:96-153``), values ON the thresholds included, as sets (the device appends in its own order)."""
    from cropnerf_amd import ops

    t = {k: torch.from_numpy(gold[f"sv_in_{k}"]).cuda() for k in ("point_location", "semantics", "density", "rgb")}
    n = t["semantics"].numel()
    pts, cols, counts = ops.export_compact(t["point_location"].reshape(n, 3).contiguous(), t["rgb"].reshape(n, 3).contiguous(),
                                           t["semantics"].reshape(n).contiguous(), t["density"].reshape(n).contiguous(), capacity=n)
    counts = counts.cpu().tolist()
    for i, name in enumerate(("semantic_colormap", "semantic", "density")):
        want_p, want_c = gold[f"sv_{name}_points"], gold[f"sv_{name}_colors"]
        assert counts[i] == want_p.shape[0], (name, counts[i], want_p.shape[0])
        got = np.concatenate([pts[i][:counts[i]].cpu().numpy(), cols[i][:counts[i]].cpu().numpy()], axis=1)
        want = np.concatenate([want_p, want_c], axis=1)
        got, want = _rows_sorted(got), _rows_sorted(want)
        assert np.array_equal(got[:, :6], want[:, :6]), name          # positions and rgb: copies
        assert np.abs(got[:, 6] - want[:, 6]).max() <= 2e-7, name      # the sigmoid channel: the device's exp


@pytest.mark.gpu
def test_hip_pointcloud_step_and_reorientation_reproduce_the_reference(gold):
    from cropnerf_amd import ops

    o, d, depth = (torch.from_numpy(gold[k]).cuda() for k in ("pc_origins", "pc_directions", "pc_depth"))
    rgb = torch.from_numpy(gold["pc_rgba"])[:, :3].contiguous().cuda()
    cmap = torch.from_numpy(gold["pc_colormap"]).cuda()
    pts, cols, dirs, count = ops.pointcloud_compact(o, d, depth, rgb, cmap, capacity=o.shape[0])
    n = int(count.item())
    assert n == gold["pc_points"].shape[0]
    got = _rows_sorted(np.concatenate([pts[:n].cpu().numpy(), cols[:n].cpu().numpy(), dirs[:n].cpu().numpy()], axis=1))
    want = _rows_sorted(np.concatenate([gold["pc_points"], gold["pc_rgbs"], gold["pc_view_directions"]], axis=1))
    assert np.abs(got[:, :3] - want[:, :3]).max() <= 2.4e-7  # o + d * depth: an fma on the device, mul + add in torch
    assert np.array_equal(got[:, 3:], want[:, 3:])
    out, flipped = ops.reorient_normals(torch.from_numpy(gold["pc_normals_in"]).cuda(), torch.from_numpy(gold["pc_view_directions"]).cuda())
    assert np.array_equal(out.cpu().numpy(), gold["pc_normals_out"])


@pytest.mark.gpu
def test_hip_data_losses_reproduce_the_reference(gold):
    """``cn_train_render_backward`` + ``cn_train_epilogue`` on the reference's own loss inputs: one sample per ray with weight
    exactly 1, so the rendered colour and logit are the inputs and the two loss values are ``get_loss_dict``'s."""
    from cropnerf_amd import ops

    rgb, sem = torch.from_numpy(gold["loss_rgb"]).cuda(), torch.from_numpy(gold["loss_sem"]).cuda()
    image = torch.from_numpy(gold["loss_image"])[:, :3].contiguous().cuda()
    mask = torch.from_numpy(gold["loss_mask"]).cuda()
    R = rgb.shape[0]
    sums = torch.zeros(5, device="cuda")
    out = ops.train_render_backward(torch.zeros(R, 1, device="cuda"), torch.ones(R, 1, device="cuda"),
                                    torch.full((R, 1), 1000.0, device="cuda"), rgb.reshape(R, 1, 3).contiguous(),
                                    sem.reshape(R, 1, 1).contiguous(), image, mask, 1.0, sums)
    assert torch.equal(out["rgb"], rgb) and torch.equal(out["semantics"], sem)
    ep = ops.train_epilogue(sums[:4], R, 1, 1.0, 1.0, None).cpu()
    assert abs(float(ep[0]) - gold["loss_values"][0]) <= 2e-6 * gold["loss_values"][0]
    assert abs(float(ep[1]) - gold["loss_values"][1]) <= 2e-6 * gold["loss_values"][1]


@pytest.mark.gpu
def test_device_kmeans_reproduces_the_references_cluster_kmeans(gold):
    """``segmentation/segmenter.py:28-45`` executed from source (scikit-learn's KMeans with the reference's arguments) against
    the product's ``cluster_kmeans`` (scikit-learn's seeding on the host, Lloyd iterations on the device): the same labels."""
    from cropnerf_amd.segmentation.segmenter import cluster_kmeans

    for c in range(int(gold["num_km"])):
        labels = cluster_kmeans(gold[f"km_{c}_points"], k=int(gold[f"km_{c}_k"]))
        assert np.array_equal(np.asarray(labels), gold[f"km_{c}_labels"]), c
