"""GPU parity tests: every HIP entry point, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Stated fp32 tolerance for the field / renderer: rtol 2e-4, atol 2e-5 (fp32 accumulation-order and
exp/sigmoid differences; the MLPs run on exact-fp32 MFMA chains).  Index-like outputs (median depth, labels)
may flip on exact ties and are checked on >= 99.5 % of the rays.
"""

import math

import pytest
import torch

from _helpers import assert_close, dev_params, make_scene, oracle_model, product_specs, rays_with_box, to_dev
from oracle import field as OF
from oracle import model as OM
from oracle import rays as ORY
from oracle import render as ORD
from oracle import samplers as OSM

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-5


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from cropnerf_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def scene():
    return make_scene(seed=0)


@pytest.fixture(scope="module")
def handles(scene, ops):
    fspec, pspecs = product_specs(scene)
    dp = dev_params(scene)
    fh = ops.FieldHandle(dp, fspec)
    dh = [ops.DensityHandle(dp, i, ps) for i, ps in enumerate(pspecs)]
    return dp, fh, dh


# ------------------------------------------------------------------------------------------------ ray generation
def test_raygen_pinhole_indices_and_image(scene, ops):
    g = torch.Generator().manual_seed(1)
    R = 777
    idx = torch.stack([torch.randint(0, scene.c2w.shape[0], (R,), generator=g),
                       torch.randint(0, scene.height, (R,), generator=g),
                       torch.randint(0, scene.width, (R,), generator=g)], -1)
    ref = ORY.pinhole_rays(scene.c2w, scene.intr, idx[:, 0], idx[:, 1], idx[:, 2])
    out = ops.raygen_pinhole(to_dev(scene.c2w), to_dev(scene.intr), ray_indices=to_dev(idx))
    assert_close(out["origins"], ref.origins, 0, 1e-7, "origins")
    assert_close(out["directions"], ref.directions, 1e-6, 1e-6, "directions")
    assert_close(out["pixel_area"], ref.pixel_area, 2e-3, 1e-12, "pixel_area")
    assert torch.equal(out["camera_indices"].cpu(), ref.camera_indices)
    assert_close(out["directions_norm"], ref.directions_norm, 1e-6, 0, "directions_norm")
    # full image slice, camera index forced to 0 like fruit_nerf.py:283
    ref = ORY.image_rays(scene.c2w, scene.intr, 3, scene.height, scene.width, start=100, end=1300, camera_index_value=0)
    out = ops.raygen_pinhole(to_dev(scene.c2w), to_dev(scene.intr), cam=3, height=scene.height, width=scene.width,
                             pixel_start=100, num_rays=1200, camera_index_value=0)
    assert_close(out["directions"], ref.directions, 1e-6, 1e-6, "image directions")
    assert int(out["camera_indices"].abs().sum()) == 0


def test_raygen_empty_and_bad_args(scene, ops):
    from cropnerf_amd._lib import CropNerfHipError

    out = ops.raygen_pinhole(to_dev(scene.c2w), to_dev(scene.intr), cam=0, height=4, width=4, pixel_start=16, num_rays=0)
    assert out["origins"].shape == (0, 3)
    with pytest.raises(CropNerfHipError):
        ops.raygen_pinhole(to_dev(scene.c2w), to_dev(scene.intr), cam=0, height=4, width=4, pixel_start=10, num_rays=10)


def test_intersect_aabb(scene, ops):
    rb = ORY.image_rays(scene.c2w, scene.intr, 1, scene.height, scene.width)
    for box in ([-1, -1, -1, 1, 1, 1], [-0.2, -0.1, -0.3, 0.1, 0.2, 0.0], [3, 3, 3, 4, 4, 4]):
        tmin, tmax = ORY.intersect_aabb(rb.origins, rb.directions, torch.tensor(box, dtype=torch.float32))
        n, f = ops.intersect_aabb(to_dev(rb.origins), to_dev(rb.directions), box)
        assert_close(n[:, 0], tmin, 1e-6, 1e-6, "nears")
        assert_close(f[:, 0], tmax, 1e-6, 1e-6, "fars")


def test_ortho_rays_and_surface_grid(ops):
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = ORY.surface_points(ORY.corners_of_aabb(aabb), 37)
    dpts = ops.surface_grid(-1.0, 1.0, 37, -1.0, 1.0, 37, -0.682, "cuda")
    assert_close(dpts, pts, 0, 1e-7, "surface grid")
    ref = ORY.ortho_rays(pts, plane, batch=512, count=2)
    out = ops.raygen_ortho(dpts, plane[0].tolist(), 512, len(ref))
    for k in ("origins", "directions", "nears", "fars", "pixel_area"):
        assert_close(out[k], getattr(ref, k), 1e-6, 1e-7, k)


def test_pose_adjustment(scene, ops):
    rb = ORY.image_rays(scene.c2w, scene.intr, 2, scene.height, scene.width)
    g = torch.Generator().manual_seed(5)
    rb.camera_indices = torch.randint(0, scene.c2w.shape[0], (len(rb), 1), generator=g)
    adj = (torch.rand(scene.c2w.shape[0], 6, generator=g) - 0.5) * 0.3
    ref = ORY.apply_pose_adjustment(rb, adj)
    o, d = to_dev(rb.origins).clone(), to_dev(rb.directions).clone()
    ops.apply_pose_adjustment(to_dev(adj), to_dev(rb.camera_indices[:, 0]), o, d)
    assert_close(o, ref.origins, 1e-6, 1e-7, "origins")
    assert_close(d, ref.directions, 1e-5, 1e-6, "directions")


def test_embedding_mean(scene, ops):
    emb = scene.params["field.embedding_appearance.embedding.weight"]
    assert_close(ops.embedding_mean(to_dev(emb)), emb.mean(0), 1e-5, 1e-6, "embedding mean")


# ------------------------------------------------------------------------------------------------ samplers
@pytest.mark.parametrize("spacing", ["uniform", "piecewise"])
@pytest.mark.parametrize("jitter", [None, "single", "full"])
def test_sample_spaced(scene, ops, spacing, jitter):
    from cropnerf_amd import _lib as L

    rb = rays_with_box(scene, 0, 300)
    rb.nears = rb.nears + 0.05
    S = 50
    g = torch.Generator().manual_seed(2)
    t_rand = None if jitter is None else torch.rand(len(rb), 1 if jitter == "single" else S + 1, generator=g)
    ref = OSM.spaced_sampler(rb, S, spacing, t_rand=t_rand)
    out = ops.sample_spaced(to_dev(rb.nears), to_dev(rb.fars), S,
                            L.SPACING_UNIFORM if spacing == "uniform" else L.SPACING_PIECEWISE, to_dev(t_rand))
    assert_close(out["starts"], ref.starts[..., 0], 2e-6, 1e-6, "starts")
    assert_close(out["ends"], ref.ends[..., 0], 2e-6, 1e-6, "ends")
    assert_close(out["spacing_starts"], ref.spacing_starts[..., 0].expand(len(rb), S), 1e-6, 1e-7, "spacing_starts")
    assert_close(out["spacing_ends"], ref.spacing_ends[..., 0].expand(len(rb), S), 1e-6, 1e-7, "spacing_ends")


@pytest.mark.parametrize("anneal", [1.0, 0.35])
@pytest.mark.parametrize("train", [False, True])
def test_sample_pdf(scene, ops, anneal, train):
    rb = rays_with_box(scene, 0, 257)
    rb.nears = rb.nears + 0.05
    S_in, S_out = 96, 48
    prev = OSM.spaced_sampler(rb, S_in, "piecewise")
    g = torch.Generator().manual_seed(3)
    w = torch.rand(len(rb), S_in, 1, generator=g) ** 4
    w[5] = 0.0  # all-zero weights: the eps padding branch
    w[6, :] = 0.0
    w[6, 17] = 1.0  # one-hot
    u_rand = torch.rand(len(rb), 1, generator=g) if train else None
    ref = OSM.pdf_sampler(prev, torch.pow(w, anneal), S_out, u_rand=u_rand)
    prev_bins = torch.cat([prev.spacing_starts[..., 0], prev.spacing_ends[:, -1:, 0]], -1).expand(len(rb), S_in + 1)
    sp, eu = ops.sample_pdf(to_dev(prev_bins.contiguous()), to_dev(w[..., 0]), to_dev(rb.nears), to_dev(rb.fars), S_out,
                            anneal=anneal, u_rand=to_dev(u_rand))
    ref_sp = torch.cat([ref.spacing_starts[..., 0], ref.spacing_ends[:, -1:, 0]], -1)
    ref_eu = torch.cat([ref.starts[..., 0], ref.ends[:, -1:, 0]], -1)
    assert_close(sp, ref_sp, 1e-4, 2e-6, "spacing bins")
    assert_close(eu, ref_eu, 2e-4, 2e-6, "euclidean bins")


# ------------------------------------------------------------------------------------------------ field
@pytest.mark.parametrize("contraction", [True, False])
def test_proposal_density(scene, ops, handles, contraction):
    dp, fh, dh = handles
    rb = rays_with_box(scene, 1, 200)
    rs = OSM.spaced_sampler(rb, 40, "uniform")
    sc = ops.scene_struct(scene.aabb, contraction)
    for lvl in range(2):
        ref = OF.proposal_density(rs.positions(), scene.params, lvl, scene.pspecs[lvl], scene.aabb, contraction)
        out = ops.proposal_density(dh[lvl], sc, to_dev(rb.origins), to_dev(rb.directions), to_dev(rs.starts[..., 0]),
                                   to_dev(rs.ends[..., 0]))
        assert_close(out, ref[..., 0], RTOL, ATOL, f"proposal density {lvl}")


@pytest.mark.parametrize("contraction,mode", [(True, "test"), (False, "inference"), (True, "train_app")])
def test_field_eval_general_kernel(scene, ops, handles, contraction, mode):
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = rays_with_box(scene, 2, 150)
    g = torch.Generator().manual_seed(7)
    rb.camera_indices = torch.randint(0, scene.c2w.shape[0], (len(rb), 1), generator=g)
    rs = OSM.spaced_sampler(rb, 33, "uniform")
    training = mode == "train_app"
    ref = OF.field_forward(rs.positions(), rb.directions, rb.camera_indices, scene.params, scene.fspec, scene.aabb,
                           contraction, "inference" if mode == "inference" else "test", training=training)
    sc = ops.scene_struct(scene.aabb, contraction)
    out = ops.field_eval(fh, sc, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.camera_indices[:, 0]),
                         to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]),
                         app_mode=L.APP_PER_CAMERA if training else L.APP_MEAN, want_positions=True)
    assert_close(out["positions"], rs.positions(), 1e-6, 1e-6, "positions")
    assert_close(out["density"], ref["density"][..., 0], RTOL, ATOL, "density")
    assert_close(out["semantics"], ref["semantics"][..., 0], RTOL, ATOL, "semantics")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")


# ------------------------------------------------------------------------------------------------ compositing
def _depth_match(dev_depth, ref_depth, frac=0.995):
    ok = (dev_depth.cpu() - ref_depth).abs() <= 1e-5 + 1e-5 * ref_depth.abs()
    assert ok.float().mean().item() >= frac, f"median depth agrees on {ok.float().mean().item():.4f} of rays"


@pytest.mark.parametrize("S", [48, 64, 192, 333])
def test_composite(ops, S):
    from cropnerf_amd import _lib as L

    g = torch.Generator().manual_seed(S)
    R = 500
    nears = torch.rand(R, 1, generator=g)
    rb = ORY.RayBundle(torch.zeros(R, 3), torch.zeros(R, 3), torch.zeros(R, 1), None, nears, nears + 1 + torch.rand(R, 1, generator=g))
    rs = OSM.spaced_sampler(rb, S, "uniform")
    den = torch.rand(R, S, 1, generator=g) ** 3 * 40
    den[::7] *= 0.01  # rays that never reach 0.5 accumulated weight
    rgb = torch.rand(R, S, 3, generator=g)
    sem = torch.randn(R, S, 1, generator=g) * 4
    w = OSM.get_weights(rs.deltas, den)
    for bg_mode, bg in ((L.BG_LAST_SAMPLE, (0, 0, 0)), (L.BG_COLOR, (0.0, 0.0, 0.0)), (L.BG_COLOR, (0.2, 0.5, 1.0))):
        ref_rgb = ORD.render_rgb(rgb, w, "last_sample" if bg_mode == L.BG_LAST_SAMPLE else torch.tensor(bg, dtype=torch.float32))
        out = ops.composite(to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]), to_dev(den[..., 0]), to_dev(rgb),
                            to_dev(sem[..., 0]), bg_mode=bg_mode, bg_color=bg, want_weights=True)
        assert_close(out["rgb"], ref_rgb, RTOL, ATOL, "rgb")
    assert_close(out["weights"], w[..., 0], RTOL, 1e-7, "weights")
    assert_close(out["accumulation"], ORD.render_accumulation(w), RTOL, ATOL, "accumulation")
    ref_sem = ORD.render_semantics(sem, w)
    assert_close(out["semantics"], ref_sem, RTOL, 5e-5, "semantics")
    _depth_match(out["depth"], ORD.render_depth_median(w, rs.starts, rs.ends))
    cm = ORD.semantics_colormap(ref_sem)
    far_from_tie = (ref_sem[:, 0] - math.log(9.0)).abs() > 1e-3
    assert torch.equal(out["semantics_colormap"].cpu()[far_from_tie], cm[far_from_tie])


def test_composite_homogeneous_known_answer(ops):
    S, near, far, sigma = 192, 0.5, 2.5, 3.0
    R = 8
    rb = ORY.RayBundle(torch.zeros(R, 3), torch.zeros(R, 3), torch.zeros(R, 1), None, torch.full((R, 1), near), torch.full((R, 1), far))
    rs = OSM.spaced_sampler(rb, S, "uniform")
    out = ops.composite(to_dev(rs.starts[..., 0]), to_dev(rs.ends[..., 0]), torch.full((R, S), sigma, device="cuda"))
    assert abs(out["accumulation"][0, 0].item() - (1 - math.exp(-sigma * (far - near)))) < 1e-5
    delta = (far - near) / S
    k = math.ceil(math.log(2) / (sigma * delta)) - 1
    assert abs(out["depth"][0, 0].item() - (near + (k + 0.5) * delta)) < 1e-5


# ------------------------------------------------------------------------------------------------ fused renderer
def _fused_vs_oracle(scene, ops, handles, S, contraction, n_rays, cam, bg_override=None, density_only=False,
                     matrix_precision=0):
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = rays_with_box(scene, cam, n_rays)
    m = oracle_model(scene, "inference", disable_scene_contraction=not contraction)
    m.uniform_samples = S  # uniform sampler, contraction as configured
    if bg_override is not None:
        m.background_override = torch.tensor(bg_override, dtype=torch.float32)
    ref = m.forward(rb)
    sc = ops.scene_struct(scene.aabb, contraction)
    opts = ops.render_opts(S, bg_mode=L.BG_COLOR if bg_override is not None else L.BG_LAST_SAMPLE,
                           bg_color=bg_override or (0, 0, 0), density_only=density_only, matrix_precision=matrix_precision)
    out = ops.render_rays(fh, sc, opts, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars),
                          want_weights=True)
    return ref, out


@pytest.mark.parametrize("S,contraction", [(192, False), (64, True), (48, True), (100, False), (333, False)])
def test_render_rays_uniform(scene, ops, handles, S, contraction):
    ref, out = _fused_vs_oracle(scene, ops, handles, S, contraction, 600, 0)
    assert_close(out["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    assert_close(out["semantics"], ref["semantics"], RTOL, 5e-5, "semantics")
    _depth_match(out["depth"], ref["depth"])


def test_render_rays_black_background_and_density_only(scene, ops, handles):
    ref, out = _fused_vs_oracle(scene, ops, handles, 96, True, 300, 3, bg_override=(0.0, 0.0, 0.0))
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb (black bg)")
    ref, out = _fused_vs_oracle(scene, ops, handles, 96, True, 300, 3, density_only=True)
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation (density only)")
    assert "rgb" not in out


def test_render_samples_matches_general_kernel_and_oracle(scene, ops, handles):
    """Export-mode forward (fruit_nerf.py:476-494): per-sample outputs of the fused kernel vs the oracle AND the
    shape-generic HIP kernel (two independent device implementations)."""
    dp, fh, dh = handles
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = ORY.surface_points(ORY.corners_of_aabb(aabb), 12)
    rb = ORY.ortho_rays(pts, plane, 100, 1)
    S = 150
    m = oracle_model(scene, "export")
    m.setup_inference(True, S)
    ref = m.forward(rb)
    sc = ops.scene_struct(scene.aabb, False)
    opts = ops.render_opts(S)
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars))
    out = ops.render_samples(fh, sc, opts, o, d, n, f)
    assert_close(out["positions"], ref["point_location"], 1e-6, 1e-6, "positions")
    assert_close(out["density"], ref["density"], RTOL, ATOL, "density")
    assert_close(out["semantics"], ref["semantics"], RTOL, ATOL, "semantics")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    clear = (ref["semantics"] - math.log(9.0)).abs() > 1e-3
    assert torch.equal(out["semantics_colormap"].cpu()[clear], ref["semantics_colormap"][clear])
    sm = ops.sample_spaced(n, f, S)
    gen = ops.field_eval(fh, sc, o, d, None, sm["starts"], sm["ends"])
    assert_close(out["density"], gen["density"].cpu(), RTOL, ATOL, "fused vs general: density")
    assert_close(out["rgb"], gen["rgb"].cpu(), RTOL, ATOL, "fused vs general: rgb")
    assert_close(out["semantics"], gen["semantics"].cpu(), RTOL, ATOL, "fused vs general: semantics")


def test_render_rays_per_camera_appearance(scene, ops, handles):
    """Training-style appearance (embedding[camera_idx], fruit_field.py:251-252) through the fused kernel."""
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = rays_with_box(scene, 4, 256)
    g = torch.Generator().manual_seed(11)
    rb.camera_indices = torch.randint(0, scene.c2w.shape[0], (len(rb), 1), generator=g)
    S = 64
    rs = OSM.spaced_sampler(rb, S, "uniform")
    fo = OF.field_forward(rs.positions(), rb.directions, rb.camera_indices, scene.params, scene.fspec, scene.aabb,
                          True, "test", training=True)
    w = OSM.get_weights(rs.deltas, fo["density"])
    ref_rgb = ORD.render_rgb(fo["rgb"], w, "last_sample")
    sc = ops.scene_struct(scene.aabb, True)
    opts = ops.render_opts(S, app_mode=L.APP_PER_CAMERA)
    out = ops.render_rays(fh, sc, opts, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars),
                          camera_indices=to_dev(rb.camera_indices[:, 0]))
    assert_close(out["rgb"], ref_rgb, RTOL, ATOL, "rgb per-camera appearance")
    from cropnerf_amd._lib import CropNerfHipError

    with pytest.raises(CropNerfHipError, match="Camera indices are not provided"):
        ops.render_rays(fh, sc, opts, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars))


# ------------------------------------------------------------------------------------------------ proposal path
def test_proposal_sample_and_full_forward(scene, ops, handles):
    """get_outputs in eval (fruit_nerf.py:543-599): pose tweak -> proposal sampler (256, 96) -> 48 field samples."""
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = ORY.image_rays(scene.c2w, scene.intr, 5, scene.height, scene.width).slice(0, 500)
    m = oracle_model(scene, "test")
    ref = m.forward(rb)
    o, d = to_dev(rb.origins).clone(), to_dev(rb.directions).clone()
    ops.apply_pose_adjustment(dp["camera_optimizer.pose_adjustment"], to_dev(rb.camera_indices[:, 0]), o, d)
    R = len(rb)
    nears = torch.zeros(R, 1, device="cuda")
    fars = torch.full((R, 1), 1000.0, device="cuda")
    sc = ops.scene_struct(scene.aabb, True)
    ps = ops.proposal_sample(dh, sc, o, d, nears, fars, (256, 96), 48)
    ref_bins = torch.cat([ref["_starts"][..., 0], ref["_ends"][:, -1:, 0]], -1)
    # bins are a piecewise-linear inverse-cdf of fp32 weights; far-field bins are huge numbers in metric space
    assert_close(ps["euclidean_bins"], ref_bins, 2e-3, 1e-4, "final euclidean bins", frac_ok=0.999)
    _depth_match(ps["prop_depth"][0][:, None], ref["prop_depth_0"], 0.99)
    _depth_match(ps["prop_depth"][1][:, None], ref["prop_depth_1"], 0.99)
    opts = ops.render_opts(48)
    out = ops.render_rays(fh, sc, opts, o, d, nears, fars, camera_indices=to_dev(rb.camera_indices[:, 0]),
                          bins=ps["euclidean_bins"])
    # end to end (sampler differences feed through, see test_proposal_sampler_stage_by_stage_on_the_oracles_inputs for the
    # stages on identical inputs): most rays agree to the fp32 bars, the rest are rays where a bin edge moved
    err = (out["rgb"].cpu() - ref["rgb"]).abs().max(dim=-1).values
    tight = (err <= ATOL + RTOL).float().mean().item()
    print(f"proposal path end to end: rgb max err {err.max().item():.3e}, median {err.median().item():.3e}, "
          f"{100 * tight:.1f} % of rays inside the fp32 bars")
    assert tight >= 0.90, f"only {100 * tight:.1f} % of rays inside rtol {RTOL} / atol {ATOL}; worst {err.max().item():.3e}"
    assert_close(out["rgb"], ref["rgb"], 2e-3, 2e-3, "rgb (proposal path)", frac_ok=0.995)
    assert_close(out["accumulation"], ref["accumulation"], 2e-3, 2e-3, "accumulation (proposal path)", frac_ok=0.995)
    # and exactly, given the oracle's own bins
    out2 = ops.render_rays(fh, sc, opts, o, d, nears, fars, bins=to_dev(ref_bins), want_weights=True)
    assert_close(out2["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights (oracle bins)")
    assert_close(out2["rgb"], ref["rgb"], RTOL, ATOL, "rgb (oracle bins)")
    assert_close(out2["semantics"], ref["semantics"], RTOL, 5e-5, "semantics (oracle bins)")


def test_proposal_sampler_stage_by_stage_on_the_oracles_inputs(scene, ops, handles):
    """The end-to-end comparison above is loose by nature: level i + 1's bins are an inverse cdf of level i's weights, so a
    1e-6 difference in one weight moves a bin edge, the next level's samples and finally the rendered pixel.  Here every
    stage is fed the ORACLE's inputs and held to the tight bars: the two proposal networks' weights on the oracle's own
    sample intervals, and each PDF resampling step on the oracle's own bins and weights."""
    from cropnerf_amd import _lib as L
    from oracle import samplers as OSM

    dp, fh, dh = handles
    rb = ORY.image_rays(scene.c2w, scene.intr, 5, scene.height, scene.width).slice(0, 500)
    m = oracle_model(scene, "test")
    rbo = ORY.apply_pose_adjustment(ORY.near_far_collider(rb, training=False), scene.params["camera_optimizer.pose_adjustment"])
    rs, weights_list, samples_list = OSM.proposal_sampler(rbo, m._density_fns(), (256, 96), 48)
    o, d = to_dev(rbo.origins), to_dev(rbo.directions)
    nears, fars = to_dev(rbo.nears), to_dev(rbo.fars)
    sc = ops.scene_struct(scene.aabb, True)
    levels = list(samples_list) + [rs]
    for lvl in range(2):
        smp = samples_list[lvl]
        st, en = to_dev(smp.starts[..., 0]), to_dev(smp.ends[..., 0])
        den = ops.proposal_density(dh[lvl], sc, o, d, st, en)
        w = ops.composite(st, en, den, want_weights=True)["weights"]
        assert_close(w, weights_list[lvl][..., 0], RTOL, 1e-6, f"proposal level {lvl}: weights on the oracle's intervals")
        # the resampling step that follows, on the oracle's bins and weights
        nxt = levels[lvl + 1]
        prev_bins = torch.cat([smp.spacing_starts[..., 0], smp.spacing_ends[:, -1:, 0]], -1)
        sp, eu = ops.sample_pdf(to_dev(prev_bins), to_dev(weights_list[lvl][..., 0]), nears, fars, nxt.starts.shape[1])
        ref_sp = torch.cat([nxt.spacing_starts[..., 0], nxt.spacing_ends[:, -1:, 0]], -1)
        ref_eu = torch.cat([nxt.starts[..., 0], nxt.ends[:, -1:, 0]], -1)
        assert_close(sp, ref_sp, 1e-5, 1e-6, f"PDF resampling after level {lvl}: spacing bins")
        # euclidean bins = s^-1(spacing) with s^-1(x) = 1 / (2 - 2x) in the far field: d(eu)/dx = 2 eu^2, so a spacing bin that
        # is right to a few fp32 ulps (previous line) gives a euclidean bin right to that many ulps x 2 eu^2
        err = (eu.cpu() - ref_eu).abs()
        tol = 1e-6 + 2e-5 * ref_eu.abs() + 1e-6 * ref_eu ** 2
        assert bool((err <= tol).all()), (f"PDF resampling after level {lvl}: euclidean bins, worst excess "
                                          f"{(err - tol).max().item():.3e} at {ref_eu.flatten()[(err - tol).flatten().argmax()].item():.4g}")


def test_unfused_proposal_chain_matches_fused(scene, ops, handles):
    """cn_sample_spaced + cn_proposal_density + cn_composite + cn_sample_pdf composed == cn_proposal_sample."""
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = rays_with_box(scene, 1, 200)
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears + 0.01, rb.fars))
    sc = ops.scene_struct(scene.aabb, True)
    fused = ops.proposal_sample(dh, sc, o, d, n, f, (128, 64), 32)
    sm = ops.sample_spaced(n, f, 128, L.SPACING_PIECEWISE)
    bins = torch.cat([sm["spacing_starts"], sm["spacing_ends"][:, -1:]], -1).contiguous()
    starts, ends = sm["starts"], sm["ends"]
    for lvl, s_next in ((0, 64), (1, 32)):
        den = ops.proposal_density(dh[lvl], sc, o, d, starts, ends)
        w = ops.composite(starts, ends, den, want_weights=True)["weights"]
        bins, eu = ops.sample_pdf(bins, w, n, f, s_next)
        starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
    assert_close(eu, fused["euclidean_bins"].cpu(), 1e-5, 1e-6, "fused vs composed bins")


def test_training_sampler_in_one_launch_matches_the_composed_chain(scene, ops, handles):
    """cn_proposal_sample_train (stratified single-jitter bins, PDF resampling at u + rand / nb, every level's bins /
    intervals / densities written out) == cn_sample_spaced + cn_proposal_density + cn_composite + cn_sample_pdf with the same
    randoms (fruit_nerf.py:549 under model.train()).  Against the oracle: tests/test_gpu_train.py, whose training
    iterations run through this launch and are compared with oracle/losses.py loss by loss and gradient by gradient."""
    from cropnerf_amd import _lib as L

    dp, fh, dh = handles
    rb = rays_with_box(scene, 1, 300)
    R = rb.origins.shape[0]
    g = torch.Generator().manual_seed(31)
    jitter = [torch.rand(R, 1, generator=g) for _ in range(3)]
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears + 0.01, rb.fars))
    sc = ops.scene_struct(scene.aabb, True)
    s_prop, s_final, anneal = (128, 70), 33, 0.7  # ragged against the 64-lane chunks on purpose
    assert ops.proposal_sample_fused_supported(dh, s_prop, s_final)
    fused = ops.proposal_sample_train(dh, sc, o, d, n, f, s_prop, s_final, anneal,
                                      to_dev(torch.cat([j.reshape(1, R) for j in jitter], 0)))
    sm = ops.sample_spaced(n, f, s_prop[0], L.SPACING_PIECEWISE, to_dev(jitter[0]))
    bins = torch.cat([sm["spacing_starts"], sm["spacing_ends"][:, -1:]], -1).contiguous()
    starts, ends = sm["starts"], sm["ends"]
    for lvl, s_next in ((0, s_prop[1]), (1, s_final)):
        den = ops.proposal_density(dh[lvl], sc, o, d, starts, ends)
        lv = fused["levels"][lvl]
        assert_close(lv["bins"].cpu(), bins.cpu(), 1e-5, 1e-6, f"level {lvl} spacing bins")
        assert_close(lv["starts"].cpu(), starts.cpu(), 1e-5, 1e-6, f"level {lvl} starts")
        assert_close(lv["ends"].cpu(), ends.cpu(), 1e-5, 1e-6, f"level {lvl} ends")
        assert_close(lv["density"].cpu(), den.cpu(), 2e-4, 1e-6, f"level {lvl} density")
        w = ops.composite(starts, ends, den, want_weights=True, eval_clamp=False)["weights"]
        bins, eu = ops.sample_pdf(bins, w, n, f, s_next, anneal=anneal, u_rand=to_dev(jitter[lvl + 1]))
        starts, ends = eu[:, :-1].contiguous(), eu[:, 1:].contiguous()
    assert_close(fused["euclidean_bins"].cpu(), eu.cpu(), 1e-5, 2e-6, "final euclidean bins")
    assert torch.equal(fused["starts"], fused["euclidean_bins"][:, :-1]) and torch.equal(fused["ends"], fused["euclidean_bins"][:, 1:])
    assert_close(fused["spacing_bins"].cpu(), bins.cpu(), 1e-5, 2e-6, "final spacing bins")


# ------------------------------------------------------------------------------------------------ exporters
def _sort_rows(t):
    """Lexicographic row order on the three position columns (positions are copied bit-exactly)."""
    for c in (2, 1, 0):
        t = t[torch.sort(t[:, c], stable=True).indices]
    return t


def test_export_compact_sets(scene, ops):
    g = torch.Generator().manual_seed(21)
    N = 100_003
    out = {
        "point_location": torch.rand(N, 3, generator=g),
        "semantics": torch.randn(N, generator=g) * 3 + 1,
        "density": torch.rand(N, generator=g) * 140,
        "rgb": torch.rand(N, 3, generator=g),
    }
    out["semantics"][:3] = 3.0
    out["density"][:3] = 70.0  # inclusive thresholds (exporter_utils.py:111-112)
    out["semantics_colormap"] = torch.heaviside(torch.sigmoid(out["semantics"]) - 0.9, torch.tensor(0.0)).long()
    ref = OM.sample_volume_masks(out)
    pts, cols, counts = ops.export_compact(to_dev(out["point_location"]), to_dev(out["rgb"]), to_dev(out["semantics"]),
                                           to_dev(out["density"]), capacity=N)
    counts = counts.cpu().tolist()
    for k, name in enumerate(("semantic_colormap", "semantic", "density")):
        assert counts[k] == ref[name]["points"].shape[0], name
        got = torch.cat([pts[k][: counts[k]].cpu(), cols[k][: counts[k]].cpu()], -1)
        exp = torch.cat([ref[name]["points"], ref[name]["colors"]], -1)
        assert_close(_sort_rows(got), _sort_rows(exp), 1e-5, 1e-6, name)
    # capacity overflow is counted, not written
    pts, cols, counts2 = ops.export_compact(to_dev(out["point_location"]), to_dev(out["rgb"]), to_dev(out["semantics"]),
                                            to_dev(out["density"]), capacity=10)
    assert counts2.cpu().tolist() == counts


def test_pointcloud_compact(scene, ops):
    g = torch.Generator().manual_seed(22)
    R = 5000
    o, d = torch.rand(R, 3, generator=g), torch.randn(R, 3, generator=g)
    rb = ORY.RayBundle(o, d, torch.zeros(R, 1))
    outputs = {"depth": torch.rand(R, 1, generator=g), "rgb": torch.rand(R, 3, generator=g),
               "semantics_colormap": (torch.rand(R, 1, generator=g) > 0.7).float().repeat(1, 3)}
    rp, rc, rd = OM.pointcloud_from_outputs(rb, outputs)
    pts, cols, dirs, count = ops.pointcloud_compact(to_dev(o), to_dev(d), to_dev(outputs["depth"]), to_dev(outputs["rgb"]),
                                                    to_dev(outputs["semantics_colormap"]), capacity=R)
    n = int(count.item())
    assert n == rp.shape[0]
    got = torch.cat([pts[:n], cols[:n], dirs[:n]], -1).cpu()
    exp = torch.cat([rp, rc, rd], -1)
    assert_close(_sort_rows(got), _sort_rows(exp), 1e-6, 1e-6, "pointcloud rows")


# ------------------------------------------------------------------------------------------------ full size
def test_full_size_chunk_properties(ops):
    """BASELINE config C2 chunk (65 536 rays x 192 samples): size-independent properties instead of the oracle."""
    sc_ = make_scene(seed=3, height=800, width=800, focal=1111.1, num_images=4)
    fspec, pspecs = product_specs(sc_)
    dp = dev_params(sc_)
    fh = ops.FieldHandle(dp, fspec)
    R, S = 65536, 192
    rays = ops.raygen_pinhole(to_dev(sc_.c2w), to_dev(sc_.intr), cam=1, height=800, width=800, pixel_start=800 * 300,
                              num_rays=R)
    n, f = ops.intersect_aabb(rays["origins"], rays["directions"], [-1, -1, -1, 1, 1, 1])
    sc = ops.scene_struct(sc_.aabb, False)
    opts = ops.render_opts(S)
    out = ops.render_rays(fh, sc, opts, rays["origins"], rays["directions"], n, f, want_weights=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out["rgb"]).all() and out["rgb"].min() >= 0 and out["rgb"].max() <= 1
    assert out["accumulation"].min() >= 0 and out["accumulation"].max() <= 1 + 1e-5
    assert_close(out["weights"].sum(-1, keepdim=True), out["accumulation"].cpu(), 1e-5, 1e-6, "sum w == acc")
    assert (out["depth"] >= n - 1e-6).all() and (out["depth"] <= f + 1e-6).all()
    # chunk invariance: any sub-range rendered alone gives bit-identical rows
    lo, hi = 12345, 23456
    sub = ops.render_rays(fh, sc, opts, rays["origins"][lo:hi].contiguous(), rays["directions"][lo:hi].contiguous(),
                          n[lo:hi].contiguous(), f[lo:hi].contiguous())
    assert torch.equal(sub["rgb"], out["rgb"][lo:hi])
    assert torch.equal(sub["semantics"], out["semantics"][lo:hi])
    # and a spot check of 256 rays against the oracle
    idx = torch.arange(0, R, R // 256)[:256]
    rb = ORY.RayBundle(rays["origins"].cpu()[idx], rays["directions"].cpu()[idx], torch.zeros(256, 1), None,
                       n.cpu()[idx], f.cpu()[idx])
    m = oracle_model(sc_, "inference", disable_scene_contraction=True)
    m.uniform_samples = S
    ref = m.forward(rb)
    assert_close(out["rgb"][idx.cuda()], ref["rgb"], RTOL, ATOL, "rgb spot check")
    assert_close(out["accumulation"][idx.cuda()], ref["accumulation"], RTOL, ATOL, "acc spot check")


def test_full_size_proposal_sampler_properties(ops):
    """The proposal sampler on a C2-size batch of the default method (65 536 rays, (256, 96) proposal + 48 field samples,
    2^17-entry proposal tables, scene contraction): properties that do not depend on the size -- bins ordered and inside
    [near, far], every ray independent of the call it is part of and of its place in it -- and a 256-ray spot check of the
    whole chain (sampler, then a render on its bins) against the oracle."""
    sc_ = make_scene(seed=3, height=800, width=800, focal=1111.1, num_images=4, grid_scale=1.0)
    fspec, pspecs = product_specs(sc_)
    dp = dev_params(sc_)
    fh = ops.FieldHandle(dp, fspec)
    dh = [ops.DensityHandle(dp, i, ps) for i, ps in enumerate(pspecs)]
    R = 65536
    rays = ops.raygen_pinhole(to_dev(sc_.c2w), to_dev(sc_.intr), cam=1, height=800, width=800, pixel_start=800 * 300,
                              num_rays=R)
    o, d = rays["origins"], rays["directions"]
    n, f = torch.full((R, 1), 0.05, device="cuda"), torch.full((R, 1), 1000.0, device="cuda")  # the method's collider
    sc = ops.scene_struct(sc_.aabb, True)
    ps = ops.proposal_sample(dh, sc, o, d, n, f, (256, 96), 48)
    eu, sp = ps["euclidean_bins"], ps["spacing_bins"]
    assert torch.isfinite(eu).all() and torch.isfinite(sp).all()
    assert (eu[:, 1:] >= eu[:, :-1]).all() and (sp[:, 1:] >= sp[:, :-1]).all(), "bins out of order"
    assert (eu[:, :1] >= n * (1 - 1e-6)).all() and (eu[:, -1:] <= f * (1 + 1e-6)).all()
    assert (sp >= 0).all() and (sp <= 1).all()
    assert (ps["prop_depth"] >= 0.05 * (1 - 1e-6)).all() and (ps["prop_depth"] <= 1000.0 * (1 + 1e-6)).all()
    # any sub-range sampled alone gives bit-identical rows (one wave owns one ray; nothing is shared between rays)
    lo, hi = 12345, 23456
    sub = ops.proposal_sample(dh, sc, o[lo:hi].contiguous(), d[lo:hi].contiguous(), n[lo:hi].contiguous(),
                              f[lo:hi].contiguous(), (256, 96), 48)
    assert torch.equal(sub["euclidean_bins"], eu[lo:hi]) and torch.equal(sub["prop_depth"], ps["prop_depth"][:, lo:hi])
    # and so does any permutation of the rays
    perm = torch.randperm(R, generator=torch.Generator().manual_seed(5)).cuda()
    pp = ops.proposal_sample(dh, sc, o[perm].contiguous(), d[perm].contiguous(), n[perm].contiguous(), f[perm].contiguous(),
                             (256, 96), 48)
    assert torch.equal(pp["euclidean_bins"], eu[perm])
    # 256 rays of it against the oracle: final bins, and the render on them
    idx = torch.arange(0, R, R // 256)[:256]
    rb = ORY.RayBundle(o.cpu()[idx], d.cpu()[idx], torch.zeros(256, 1), None, n.cpu()[idx], f.cpu()[idx])
    m = oracle_model(sc_, "inference")
    ref = m.forward(rb)
    ref_bins = torch.cat([ref["_starts"][..., 0], ref["_ends"][:, -1:, 0]], -1)
    assert_close(eu[idx.cuda()], ref_bins, 2e-3, 1e-4, "final euclidean bins (spot check)", frac_ok=0.999)
    out = ops.render_rays(fh, sc, ops.render_opts(48), o[idx.cuda()].contiguous(), d[idx.cuda()].contiguous(),
                          n[idx.cuda()].contiguous(), f[idx.cuda()].contiguous(), bins=eu[idx.cuda()].contiguous())
    assert_close(out["rgb"], ref["rgb"], 0.0, 2e-3, "rgb on the sampler's own bins (spot check)", frac_ok=0.99)


# ------------------------------------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("R,S", [(1, 1), (3, 2), (5, 65), (130, 1)])
def test_render_rays_tiny_shapes(scene, ops, handles, R, S):
    """Fewer rays than waves, a single sample, a chunk boundary + 1."""
    ref, out = _fused_vs_oracle(scene, ops, handles, S, False, R, 2)
    assert_close(out["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    assert_close(out["semantics"], ref["semantics"], RTOL, 5e-5, "semantics")
    _depth_match(out["depth"], ref["depth"], 0.99 if R > 50 else 1.0)


def test_render_rays_empty_and_missing_rays(scene, ops, handles):
    """R = 0 is a no-op; rays that miss the box (near = far = 1e10, fruit_nerf.py:286) composite to zero weight and
    take the last-sample colour, exactly like the oracle."""
    dp, fh, dh = handles
    sc = ops.scene_struct(scene.aabb, False)
    opts = ops.render_opts(32)
    z3, z1 = torch.empty(0, 3, device="cuda"), torch.empty(0, 1, device="cuda")
    out = ops.render_rays(fh, sc, opts, z3, z3, z1, z1)
    assert out["rgb"].shape == (0, 3)
    rb = ORY.image_rays(scene.c2w, scene.intr, 0, scene.height, scene.width).slice(0, 64)
    rb = ORY.with_aabb_near_far(rb, torch.tensor([5.0, 5, 5, 6, 6, 6]))
    assert float(rb.nears.min()) == 1e10
    m = oracle_model(scene, "inference", disable_scene_contraction=True)
    m.uniform_samples = 32
    ref = m.forward(rb)
    out = ops.render_rays(fh, sc, opts, to_dev(rb.origins), to_dev(rb.directions), to_dev(rb.nears), to_dev(rb.fars))
    assert float(out["accumulation"].abs().max()) == 0.0 and float(ref["accumulation"].abs().max()) == 0.0
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb of missing rays")
    assert_close(out["depth"], ref["depth"], 1e-6, 0, "depth of missing rays")


def test_unsupported_shape_is_reported_not_computed(scene, ops):
    """The fused kernel refuses field shapes it is not built for (CN_ERR_UNSUPPORTED) instead of guessing."""
    from cropnerf_amd import config as PC
    from cropnerf_amd._lib import CropNerfHipError

    spec = PC.FieldSpec(grid=PC.GridSpec(log2_hashmap_size=10), geo_feat_dim=30, num_layers_semantic=3,
                        hidden_dim_semantics=128, num_images=2)
    params = PC.init_params(spec, [], device="cuda")
    fh = ops.FieldHandle(params, spec)
    o = torch.zeros(4, 3, device="cuda")
    d = torch.ones(4, 3, device="cuda")
    n, f = torch.zeros(4, 1, device="cuda"), torch.ones(4, 1, device="cuda")
    with pytest.raises(CropNerfHipError) as ei:
        ops.render_rays(fh, ops.scene_struct(scene.aabb, False), ops.render_opts(8), o, d, n, f)
    assert ei.value.code == -2
    # ... while the shape-generic kernel evaluates it (fruit_nerf_method_big field: geo 30, 3-layer 128-wide semantics)
    sm = ops.sample_spaced(n, f, 8)
    out = ops.field_eval(fh, ops.scene_struct(scene.aabb, False), o, d, None, sm["starts"], sm["ends"])
    assert torch.isfinite(out["rgb"]).all() and out["rgb"].shape == (4, 8, 3)


@pytest.mark.parametrize("width,start,R", [(40, 0, 1600), (40, 13, 1000), (37, 5, 700), (800, 777, 3000), (3, 1, 50)])
def test_image_width_hint_changes_schedule_not_results(scene, ops, handles, width, start, R):
    """cn_render_opts.image_width / pixel_start only re-map rays to workgroups (XCD column stripes): every ray must
    still be rendered exactly once, bit-identically, for any width / offset / partial rows."""
    dp, fh, dh = handles
    g = torch.Generator().manual_seed(width * 1000 + start)
    rb = rays_with_box(scene, 3)
    idx = torch.randint(0, len(rb), (R,), generator=g)
    o, d, n, f = (to_dev(t[idx]) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    sc = ops.scene_struct(scene.aabb, True)
    base = ops.render_rays(fh, sc, ops.render_opts(48), o, d, n, f)
    hinted = ops.render_rays(fh, sc, ops.render_opts(48, image_width=width, pixel_start=start), o, d, n, f)
    for k in base:
        assert torch.equal(base[k], hinted[k]), k
    # "every ray exactly once": call the C ABI on output buffers pre-filled with NaN -- a ray the striped schedule
    # skipped would keep its poison, and since the launch only ever writes (no accumulation into the outputs), a ray
    # rendered twice could not differ from the baseline either; no NaN left + bit-equality = each row written
    import ctypes as C

    from cropnerf_amd import _lib as L

    poison = {k: torch.full_like(base[k], float("nan")) for k in ("rgb", "accumulation", "depth", "semantics", "semantics_colormap")}
    opts = ops.render_opts(48, image_width=width, pixel_start=start)
    ws = fh.workspace()
    L.check(L.load().cn_render_rays(C.byref(fh.struct), C.byref(sc), C.byref(opts), o.data_ptr(), d.data_ptr(), n.data_ptr(),
                                    f.data_ptr(), None, None, R, poison["rgb"].data_ptr(), poison["accumulation"].data_ptr(),
                                    poison["depth"].data_ptr(), poison["semantics"].data_ptr(),
                                    poison["semantics_colormap"].data_ptr(), None, ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for k, v in poison.items():
        assert not torch.isnan(v).any(), f"{k}: rows never written under the striped schedule"
        assert torch.equal(v, base[k]), k


@pytest.mark.parametrize("split", ["0", "2"])  # single-wave kernel / producer-consumer kernel (forced for a small batch)
@pytest.mark.parametrize("eps", [1e-2, 1e-4])
def test_early_stop_is_off_by_default_and_bounded_when_on(scene, ops, eps, split, monkeypatch):
    """cn_render_opts.early_stop_transmittance (an extension; the reference composites every sample): 0 is bit-identical
    to the plain render, > 0 on an opaque medium changes every output by less than the threshold and really stops
    (the weights behind the cut are exactly 0)."""
    monkeypatch.setenv("CN_FUSED_SPLIT", split)
    fspec, _ = product_specs(scene)
    dp = dev_params(scene)
    dp = {k: v.clone() for k, v in dp.items()}
    dp["field.mlp_base_mlp.layers.1.bias"][0] += 4.0  # density x e^4: rays go opaque within the first chunks
    fh = ops.FieldHandle(dp, fspec)
    rb = rays_with_box(scene, 2)
    o, d, n, f = (to_dev(t[:1500]) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    sc = ops.scene_struct(scene.aabb, True)
    S = 256
    full = ops.render_rays(fh, sc, ops.render_opts(S), o, d, n, f, want_weights=True)
    off = ops.render_rays(fh, sc, ops.render_opts(S, early_stop_transmittance=0.0), o, d, n, f, want_weights=True)
    for k in full:
        assert torch.equal(full[k], off[k]), k
    cut = ops.render_rays(fh, sc, ops.render_opts(S, early_stop_transmittance=eps), o, d, n, f, want_weights=True)
    stopped = (cut["weights"][:, 192:] == 0).all(dim=1) & (full["weights"][:, 192:] != 0).any(dim=1)
    assert stopped.float().mean() > 0.5, "the medium was meant to be opaque enough to stop most rays"
    assert_close(cut["rgb"], full["rgb"], 0, 1.01 * eps, "early-stop rgb")
    assert_close(cut["accumulation"], full["accumulation"], 0, 1.01 * eps, "early-stop accumulation")
    sem_scale = float(full["semantics"].abs().max()) + 1.0
    assert_close(cut["semantics"], full["semantics"], 0, 4 * eps * sem_scale, "early-stop semantics")
    assert torch.equal(cut["depth"], full["depth"])  # the median is always inside the evaluated part (T < 0.5)
    # where a ray was not stopped nothing changes at all
    kept = ~(cut["weights"] == 0).all(dim=1) & (cut["weights"][:, -1] != 0)
    assert torch.equal(cut["rgb"][kept], full["rgb"][kept])


@pytest.mark.parametrize("S,width", [(48, 0), (192, 40), (100, 37), (7, 0)])
def test_split_kernel_matches_fused(scene, ops, handles, S, width, monkeypatch):
    """render_split_kernel (gather waves feeding matrix waves through LDS) is a re-scheduling of render_fused_kernel:
    same arithmetic in the same order; the two binaries differ only in where hipcc contracts multiply-adds, i.e. by an
    ulp here and there (measured <= 3e-7 on O(1) outputs) -- with and without the image hint, with per-camera
    appearance, proposal bins, ragged S and weights requested.  Median depth and the label image are exact."""
    dp, fh, dh = handles
    rb = rays_with_box(scene, 1)
    g = torch.Generator().manual_seed(S)
    idx = torch.randint(0, len(rb), (1237,), generator=g)
    o, d, n, f = (to_dev(t[idx]) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    cam = torch.randint(0, scene.c2w.shape[0], (1237,), generator=g).cuda()
    sc = ops.scene_struct(scene.aabb, True)
    bins = None
    if S == 48:
        ps = ops.proposal_sample(dh, sc, o, d, n, f, (64, 32), S)
        bins = ps["euclidean_bins"]
    opts = ops.render_opts(S, app_mode=2, image_width=width, pixel_start=3 if width else 0, eval_clamp=S != 100)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CN_FUSED_SPLIT", "2" if mode == "1" else "0")  # 2 = split even for small batches
        outs[mode] = ops.render_rays(fh, sc, opts, o, d, n, f, camera_indices=cam, bins=bins, want_weights=True)
    for k in outs["0"]:
        if k in ("depth", "semantics_colormap"):
            assert torch.equal(outs["0"][k], outs["1"][k]), k
        else:
            assert_close(outs["1"][k], outs["0"][k], 2e-6, 1e-6, f"split vs fused {k}")
    assert torch.isfinite(outs["1"]["rgb"]).all()


@pytest.mark.parametrize("S", [48, 47, 33])
@pytest.mark.parametrize("n,width", [(1600, 40), (1237, 0), (1, 0), (2, 0), (333, 37)])
def test_packed_schedule_of_48_sample_rays_changes_no_bit(scene, ops, handles, S, n, width, monkeypatch):
    """32 < S <= 48 in exact fp32 (the default method's 48 field samples per ray): render_split_kernel<..., PACK> walks a
    pair's rays two at a time in three half-steps -- (A0, A1), (A2, B0), (B1, B2) -- instead of leaving every fourth 16-sample
    tile empty.  Same arithmetic per ray: every output of the composited render (weights included, per-camera appearance,
    proposal bins, with and without the image hint, whose striped schedule has empty slots in the middle of a pair's list) and
    of the per-sample render is bit-identical to the one-ray-per-two-half-steps schedule (CN_SPLIT_PACK=0)."""
    dp, fh, dh = handles
    monkeypatch.setenv("CN_FUSED_SPLIT", "2")  # the producer/consumer kernel also for these small batches
    rb = rays_with_box(scene, 1)
    g = torch.Generator().manual_seed(S + n)
    idx = torch.arange(n) if width else torch.randint(0, len(rb), (n,), generator=g)
    o, d, ne, f = (to_dev(t[idx]) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    cam = torch.randint(0, scene.c2w.shape[0], (n,), generator=g).cuda()
    sc = ops.scene_struct(scene.aabb, True)
    bins = ops.proposal_sample(dh, sc, o, d, ne, f, (64, 32), S)["euclidean_bins"] if n % 2 else None
    opts = ops.render_opts(S, app_mode=2, image_width=width, pixel_start=0)
    outs, samples = {}, {}
    for pack in ("0", "1"):
        monkeypatch.setenv("CN_SPLIT_PACK", pack)
        outs[pack] = ops.render_rays(fh, sc, opts, o, d, ne, f, camera_indices=cam, bins=bins, want_weights=True)
        samples[pack] = ops.render_samples(fh, sc, ops.render_opts(S, app_mode=2, eval_clamp=False), o, d, ne, f,
                                           camera_indices=cam, bins=bins)
    torch.cuda.synchronize()
    for k in outs["0"]:
        assert torch.equal(outs["0"][k], outs["1"][k]), f"composited {k}"
    for k in samples["0"]:
        assert torch.equal(samples["0"][k], samples["1"][k]), f"per-sample {k}"
    assert torch.isfinite(outs["1"]["rgb"]).all() and float(outs["1"]["weights"].abs().sum()) > 0


@pytest.mark.parametrize("S,contraction", [(192, False), (64, True), (100, False)])
def test_split_bf16_matrix_option_meets_the_parity_bar(scene, ops, handles, S, contraction, monkeypatch):
    """cn_render_opts.matrix_precision = CN_MATRIX_SPLIT_BF16: the MLP products on v_mfma_f32_16x16x32_bf16 with operands
    split into bf16 hi + lo (the lo x lo term dropped, fp32 accumulation).  Held to the SAME bars against the oracle as the
    exact-fp32 kernels (2e-4 relative + 2e-5); against the fp32 render of the same kernel it is within 5e-5.  The option is
    honoured by the producer/consumer kernel only (forced here for the small batch): elsewhere it changes nothing."""
    from cropnerf_amd import _lib as L

    monkeypatch.setenv("CN_FUSED_SPLIT", "2")
    ref, out = _fused_vs_oracle(scene, ops, handles, S, contraction, 600, 0, matrix_precision=L.MATRIX_SPLIT_BF16)
    _, exact = _fused_vs_oracle(scene, ops, handles, S, contraction, 600, 0)
    assert_close(out["weights"], ref["_weights"][..., 0], RTOL, 1e-6, "weights")
    assert_close(out["accumulation"], ref["accumulation"], RTOL, ATOL, "accumulation")
    assert_close(out["rgb"], ref["rgb"], RTOL, ATOL, "rgb")
    assert_close(out["semantics"], ref["semantics"], RTOL, 5e-5, "semantics")
    _depth_match(out["depth"], ref["depth"])
    for k in ("rgb", "accumulation", "semantics"):
        assert_close(out[k], exact[k], 5e-5, 5e-5, f"split-bf16 vs fp32 {k}")
    assert not torch.equal(out["rgb"], exact["rgb"])       # it really is a different arithmetic
    monkeypatch.setenv("CN_FUSED_SPLIT", "0")              # single-wave kernel: the option is ignored, products stay fp32
    _, fused_opt = _fused_vs_oracle(scene, ops, handles, S, contraction, 600, 0, matrix_precision=L.MATRIX_SPLIT_BF16)
    _, fused = _fused_vs_oracle(scene, ops, handles, S, contraction, 600, 0)
    assert torch.equal(fused_opt["rgb"], fused["rgb"])


@pytest.mark.parametrize("S,contraction", [(70, False), (128, True)])
def test_split_kernel_per_sample_outputs_match_fused(scene, ops, handles, S, contraction, monkeypatch):
    """The export-mode forward (cn_render_samples) in its producer/consumer form against the single-wave form."""
    dp, fh, dh = handles
    rb = rays_with_box(scene, 2)
    o, d, n, f = (to_dev(t[100:611]) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    sc = ops.scene_struct(scene.aabb, contraction)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("CN_FUSED_SPLIT", "2" if mode == "1" else "0")  # 2 = split even for small batches
        outs[mode] = ops.render_samples(fh, sc, ops.render_opts(S), o, d, n, f)
    assert torch.equal(outs["0"]["semantics_colormap"], outs["1"]["semantics_colormap"])
    for k in ("density", "rgb", "semantics", "positions"):
        assert_close(outs["1"][k], outs["0"][k], 2e-6, 1e-6, f"split vs fused per-sample {k}")
    # the split-bf16 matrix option of the per-sample form (densities are exp(logit): relative bar)
    from cropnerf_amd import _lib as L

    fast = ops.render_samples(fh, sc, ops.render_opts(S, matrix_precision=L.MATRIX_SPLIT_BF16), o, d, n, f)
    assert torch.equal(fast["positions"], outs["1"]["positions"])
    assert_close(fast["density"], outs["1"]["density"], 2e-4, 1e-6, "split-bf16 per-sample density")
    assert_close(fast["rgb"], outs["1"]["rgb"], 5e-5, 5e-5, "split-bf16 per-sample rgb")
    assert_close(fast["semantics"], outs["1"]["semantics"], 2e-4, 5e-5, "split-bf16 per-sample semantics")


def test_generic_field_kernels_agree(scene, ops, monkeypatch):
    """cn_field_eval has three device implementations (register-resident weights, LDS-staged weights, scalar): same
    results on the default shape and on the fruit_nerf_method_big shape, per-camera appearance, ragged sample counts."""
    from cropnerf_amd import config as PC

    rb = rays_with_box(scene, 0, 333)
    o, d, n, f = (to_dev(t) for t in (rb.origins, rb.directions, rb.nears, rb.fars))
    cam = (torch.arange(333) % scene.c2w.shape[0]).cuda()
    sm = ops.sample_spaced(n, f, 37)
    for big in (False, True):
        spec = PC.FieldSpec(grid=PC.GridSpec(16, 16, 4096 if big else 2048, 12, 2), geo_feat_dim=30 if big else 15,
                            num_layers_semantic=3 if big else 2, hidden_dim_semantics=128 if big else 64,
                            num_images=scene.c2w.shape[0])
        params = PC.init_params(spec, [], seed=3, grid_scale=0.1, device="cuda")
        fh = ops.FieldHandle(params, spec)
        outs = {}
        for impl in ("regw", "mfma", "scalar"):
            monkeypatch.setenv("CN_FIELD_EVAL_IMPL", impl)
            outs[impl] = ops.field_eval(fh, ops.scene_struct(scene.aabb, True), o, d, cam, sm["starts"], sm["ends"],
                                        app_mode=2, want_positions=True)
        for impl in ("regw", "mfma"):
            for k in outs[impl]:
                assert_close(outs[impl][k], outs["scalar"][k], 2e-5, 2e-6, f"{impl} {'big' if big else 'default'} {k}")
        # the split-bf16 form of the register-resident kernel (cn_field_eval_mp, field_regw_split.hpp): bf16 hi + lo operands,
        # fp32 sums -- the bars the split-bf16 mode of the fused renderer is held to per sample (densities are exp(logit):
        # relative); all three app modes, the density-only and semantics-less calls, a ragged tail (333 x 37 is not a
        # multiple of the 64-sample tile)
        from cropnerf_amd import _lib as L

        monkeypatch.delenv("CN_FIELD_EVAL_IMPL")
        name = "big" if big else "default"
        for app_mode in (0, 1, 2):
            exact = ops.field_eval(fh, ops.scene_struct(scene.aabb, True), o, d, cam, sm["starts"], sm["ends"],
                                   app_mode=app_mode, want_positions=True)
            fast = ops.field_eval(fh, ops.scene_struct(scene.aabb, True), o, d, cam, sm["starts"], sm["ends"],
                                  app_mode=app_mode, want_positions=True, matrix_precision=L.MATRIX_SPLIT_BF16)
            assert torch.equal(fast["positions"], exact["positions"])
            assert_close(fast["density"], exact["density"], 2e-4, 1e-6, f"split-bf16 {name} density, app mode {app_mode}")
            assert_close(fast["rgb"], exact["rgb"], 5e-5, 5e-5, f"split-bf16 {name} rgb, app mode {app_mode}")
            assert_close(fast["semantics"], exact["semantics"], 2e-4, 5e-5, f"split-bf16 {name} semantics, app mode {app_mode}")
            assert not torch.equal(fast["rgb"], exact["rgb"])  # it really is the other arithmetic
        f16 = ops.field_eval(fh, ops.scene_struct(scene.aabb, True), o, d, cam, sm["starts"], sm["ends"], app_mode=2,
                             want_positions=True, matrix_precision=L.MATRIX_F16)  # no fp16 form: the split-bf16 kernel
        assert torch.equal(f16["rgb"], fast["rgb"])
        with pytest.raises(L.CropNerfHipError, match="matrix_precision"):
            ops.field_eval(fh, ops.scene_struct(scene.aabb, True), o, d, cam, sm["starts"], sm["ends"], matrix_precision=7)
