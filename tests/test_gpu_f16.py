"""GPU parity tests of the fp16 matrix mode (``cn_render_opts.matrix_precision = CN_MATRIX_F16``): the reference's OWN
arithmetic class.  The reference builds every module of ``FruitField`` with ``implementation="tcnn"``
(``fruit_nerf/fruit_field.py:95,125-167``) and trains / renders under ``mixed_precision=True``
(``fruit_nerf/fruit_nerf_config.py:35``): tiny-cuda-nn casts parameters and layer inputs to fp16.  The oracle of that
arithmetic is ``oracle/tcnn.py`` with ``tcnn_half_activations=True`` (encoding accumulated in fp16, every layer input and
the network output rounded to fp16, products summed in fp32).

Stated tolerances (measured worst cases in brackets, ``tools/f16_error_probe.py``, 800 rays, 64 / 192 samples):
  * rgb against the half-activation oracle: 2.5e-4 absolute -- half an fp16 ulp of a colour in [0.5, 1), which is the
    rounding that oracle applies to its OWN output (tcnn networks return fp16); the kernel keeps the fp32 accumulators
    [7.6e-5];  rgb against the fp32-arithmetic oracle on the same fp16 parameters: 5e-5 [5.2e-6];
  * accumulation and weights: the fp32 bars of test_gpu_parity.py (rtol 2e-4, atol 2e-5) [5e-6];
  * semantic logit sums: 5e-5 absolute [2.4e-6];  median depth: the same sample as the oracle away from ties;
  * PSNR of the fp16-mode render against the exact-fp32 render of the same model: >= 90 dB [107-115 dB]
    (BASELINE.json asks for 0.1 dB against the reference).
"""

import dataclasses
import math

import pytest
import torch

from _helpers import assert_close, dev_params, make_scene, make_tcnn_scene, oracle_model, product_specs, rays_with_box, to_dev
from oracle import rays as ORY

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from cropnerf_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def L():
    from cropnerf_amd import _lib

    return _lib


def _half_oracle_scene(scene):
    return dataclasses.replace(scene, fspec=dataclasses.replace(scene.fspec, tcnn_half_activations=True))


def _psnr(a, b):
    mse = float(((a.detach().cpu().float() - b.detach().cpu().float()) ** 2).mean())
    return -10.0 * math.log10(max(mse, 1e-30))


@pytest.mark.parametrize("table_dtype", ["float16", "float32"])
@pytest.mark.parametrize("S,contraction,grid_scale", [(192, False, 0.1), (64, True, 1.0), (100, False, 1.0)])
def test_render_matches_the_half_activation_oracle(ops, L, S, contraction, grid_scale, table_dtype):
    scene = make_tcnn_scene(seed=3, grid_scale=grid_scale)
    fspec, _ = product_specs(scene)
    fh = ops.FieldHandle(dev_params(scene, table_dtype=getattr(torch, table_dtype)), fspec)
    rb = rays_with_box(scene, 0, 700)
    m16 = oracle_model(_half_oracle_scene(scene), "inference", disable_scene_contraction=not contraction)
    m16.uniform_samples = S
    ref16 = m16.forward(rb)
    m32 = oracle_model(scene, "inference", disable_scene_contraction=not contraction)
    m32.uniform_samples = S
    ref32 = m32.forward(rb)
    sc = ops.scene_struct(scene.aabb, contraction)
    args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
    out = ops.render_rays(fh, sc, ops.render_opts(S, matrix_precision=L.MATRIX_F16), *args, want_weights=True)
    exact = ops.render_rays(fh, sc, ops.render_opts(S), *args)
    assert_close(out["rgb"], ref16["rgb"], 0.0, 2.5e-4, "rgb vs the half-activation oracle")
    assert_close(out["rgb"], ref32["rgb"], 0.0, 5e-5, "rgb vs the fp32-arithmetic oracle")
    assert_close(out["accumulation"], ref16["accumulation"], 2e-4, 2e-5, "accumulation")
    assert_close(out["weights"], ref16["_weights"][..., 0], 2e-4, 2e-5, "weights")
    assert_close(out["semantics"], ref16["semantics"], 0.0, 5e-5, "semantics")
    ok = (out["depth"].cpu() - ref16["depth"]).abs() <= 1e-5 + 1e-5 * ref16["depth"].abs()
    assert ok.float().mean().item() >= 0.995
    psnr = _psnr(out["rgb"], exact["rgb"])
    assert psnr >= 90.0, f"PSNR of the fp16-mode render against the exact-fp32 render: {psnr:.1f} dB"
    # it IS a different arithmetic: the outputs are not the fp32 ones
    assert not torch.equal(out["rgb"], exact["rgb"])


def test_result_does_not_depend_on_the_call_a_ray_is_part_of(ops, L):
    """fp16 mode always runs the kernel that implements it, whatever the batch size: one call of 1 600 rays and four calls
    of 400 give the same bits (the fp32 mode switches kernels with the batch size, which moves results by an ulp)."""
    scene = make_tcnn_scene(seed=4)
    fspec, _ = product_specs(scene)
    fh = ops.FieldHandle(dev_params(scene), fspec)
    rb = rays_with_box(scene, 1)
    sc = ops.scene_struct(scene.aabb, False)
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars))
    opts = ops.render_opts(96, matrix_precision=L.MATRIX_F16)
    whole = ops.render_rays(fh, sc, opts, o, d, n, f)
    for k in ("rgb", "accumulation", "depth", "semantics"):
        parts = torch.cat([ops.render_rays(fh, sc, opts, o[i:i + 400].contiguous(), d[i:i + 400].contiguous(),
                                           n[i:i + 400].contiguous(), f[i:i + 400].contiguous())[k]
                           for i in range(0, o.shape[0], 400)])
        assert torch.equal(parts, whole[k]), k


def test_per_sample_outputs_in_export_mode(ops, L):
    scene = make_tcnn_scene(seed=3)
    fspec, _ = product_specs(scene)
    fh = ops.FieldHandle(dev_params(scene), fspec)
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = ORY.surface_points(ORY.corners_of_aabb(aabb), 12)
    rb = ORY.ortho_rays(pts, plane, 100, 1)
    S = 150
    m = oracle_model(_half_oracle_scene(scene), "export")
    m.setup_inference(True, S)
    ref = m.forward(rb)
    sc = ops.scene_struct(scene.aabb, False)
    o, d, n, f = (to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars))
    out = ops.render_samples(fh, sc, ops.render_opts(S, matrix_precision=L.MATRIX_F16), o, d, n, f)
    # per-sample values carry the fp16 rounding of single activations (no averaging along a ray): the oracle's own outputs
    # are fp16 values, i.e. exact to 2^-11 relative -- density is exp() of such a value
    assert_close(out["density"], ref["density"], 4e-3, 1e-5, "density")
    assert_close(out["semantics"], ref["semantics"], 2e-3, 2e-4, "semantics")
    assert_close(out["rgb"], ref["rgb"], 0.0, 6e-4, "rgb")
    clear = (ref["semantics"] - math.log(9.0)).abs() > 5e-3
    assert torch.equal(out["semantics_colormap"].cpu()[clear], ref["semantics_colormap"][clear])


def test_torch_layout_fp32_model_through_the_fp16_mode(ops, L):
    """Any model may be rendered in fp16 mode: the weights are rounded to fp16 on the way into the weight image."""
    sc_ = make_scene(seed=5, log2_T=15, prop_log2_T=12)
    fspec, _ = product_specs(sc_)
    fh = ops.FieldHandle(dev_params(sc_), fspec)
    rb = rays_with_box(sc_, 0, 900)
    args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
    scn = ops.scene_struct(sc_.aabb, False)
    exact = ops.render_rays(fh, scn, ops.render_opts(96), *args)
    out = ops.render_rays(fh, scn, ops.render_opts(96, matrix_precision=L.MATRIX_F16), *args)
    assert_close(out["rgb"], exact["rgb"], 0.0, 1e-4, "rgb vs the fp32 render")
    assert_close(out["accumulation"], exact["accumulation"], 2e-4, 5e-5, "accumulation vs the fp32 render")
    assert _psnr(out["rgb"], exact["rgb"]) >= 90.0


def test_density_only_pass(ops, L):
    """``get_density_for_camera_ray_bundle`` (the occlusion pass of the projection, fruit_nerf.py:320-344) in fp16 mode: the
    base MLP alone, same arithmetic as the full render -- the accumulation of the two agrees to the last bit or two (the two
    kernels contract multiply-adds differently), and with the half-activation oracle within the fp32 bars."""
    scene = make_tcnn_scene(seed=3)
    fspec, _ = product_specs(scene)
    fh = ops.FieldHandle(dev_params(scene), fspec)
    rb = rays_with_box(scene, 2, 500)
    args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
    scn = ops.scene_struct(scene.aabb, False)
    m16 = oracle_model(_half_oracle_scene(scene), "inference", disable_scene_contraction=True)
    m16.uniform_samples = 64
    ref = m16.forward(rb)
    full = ops.render_rays(fh, scn, ops.render_opts(64, matrix_precision=L.MATRIX_F16), *args)
    b = ops.render_rays(fh, scn, ops.render_opts(64, density_only=True, matrix_precision=L.MATRIX_F16), *args)
    assert set(b) == {"accumulation"}
    assert_close(b["accumulation"], ref["accumulation"], 2e-4, 2e-5, "accumulation (density only) vs the oracle")
    assert_close(b["accumulation"], full["accumulation"], 0.0, 2e-6, "accumulation: density-only vs full render")


def test_unknown_matrix_precision_is_rejected(ops, L):
    scene = make_tcnn_scene(seed=3)
    fspec, _ = product_specs(scene)
    fh = ops.FieldHandle(dev_params(scene), fspec)
    rb = rays_with_box(scene, 2, 64)
    args = [to_dev(x) for x in (rb.origins, rb.directions, rb.nears, rb.fars)]
    with pytest.raises(L.CropNerfHipError, match="matrix_precision"):
        ops.render_rays(fh, ops.scene_struct(scene.aabb, False), ops.render_opts(64, matrix_precision=7), *args)


@pytest.mark.parametrize("second_prop", [(5, 256), (7, 2048)])  # the default method's second network / fruit_nerf_method_big's
def test_proposal_sampler_in_fp16_mode(ops, L, second_prop):
    """``cn_proposal_sample_mp`` with ``CN_MATRIX_F16`` on half tables: the proposal networks in tiny-cuda-nn's arithmetic class
    (packed-fp16 grid interpolation, fp16 weights / layer inputs / network output, fp32 accumulation) against the oracle with
    ``tcnn_half_activations=True`` on the proposal networks too.  The kernel rounds where tcnn rounds but not in tcnn's ORDER (the
    oracle accumulates the eight corners one by one in fp16, the kernel blends pairs), so densities agree to fp16 noise and the
    resampled bins to a small fraction of a bin: stated bar 2 % of the bin's own width for 99.9 % of the edges [measured
    worst case in the message], and everything downstream -- a render on those bins -- inside the fp16 render bars.  A float
    table keeps the fp32 sampler (same bits as ``cn_proposal_sample``)."""
    scene = make_tcnn_scene(seed=3, grid_scale=1.0, second_prop=second_prop)
    fspec, pspecs = product_specs(scene)
    dp16 = dev_params(scene, table_dtype=torch.float16)
    fh = ops.FieldHandle(dp16, fspec)
    dh = [ops.DensityHandle(dp16, i, ps) for i, ps in enumerate(pspecs)]
    assert all(dp16[f"proposal_networks.{i}.encoding.hash_table"].dtype == torch.float16 for i in range(len(pspecs)))
    assert pspecs[1].grid.num_levels == second_prop[0]
    rb = ORY.image_rays(scene.c2w, scene.intr, 5, scene.height, scene.width).slice(0, 600)
    half = dataclasses.replace(scene, fspec=dataclasses.replace(scene.fspec, tcnn_half_activations=True),
                               pspecs=[dataclasses.replace(p, tcnn_half_activations=True) for p in scene.pspecs])
    ref = oracle_model(half, "test").forward(rb)
    o, d = to_dev(rb.origins).clone(), to_dev(rb.directions).clone()
    ops.apply_pose_adjustment(dp16["camera_optimizer.pose_adjustment"], to_dev(rb.camera_indices[:, 0]), o, d)
    R = len(rb)
    nears, fars = torch.zeros(R, 1, device="cuda"), torch.full((R, 1), 1000.0, device="cuda")
    sc = ops.scene_struct(scene.aabb, True)
    ps16 = ops.proposal_sample(dh, sc, o, d, nears, fars, (256, 96), 48, matrix_precision=L.MATRIX_F16)
    ps32 = ops.proposal_sample(dh, sc, o, d, nears, fars, (256, 96), 48)
    ref_bins = torch.cat([ref["_starts"][..., 0], ref["_ends"][:, -1:, 0]], -1)
    got = ps16["euclidean_bins"].cpu()
    width = (ref_bins[:, 1:] - ref_bins[:, :-1]).clamp_min(1e-9)
    width = torch.cat([width, width[:, -1:]], -1)
    dev_frac = (got - ref_bins).abs() / width
    assert float((dev_frac <= 0.02).float().mean()) >= 0.999, f"bin edges: worst {float(dev_frac.max()):.3g} of a bin width"
    assert not torch.equal(ps16["euclidean_bins"], ps32["euclidean_bins"])  # it IS another arithmetic
    assert float((ps16["euclidean_bins"] - ps32["euclidean_bins"]).abs().max()) < 1e-2
    out = ops.render_rays(fh, sc, ops.render_opts(48, matrix_precision=L.MATRIX_F16), o, d, nears, fars,
                          bins=ps16["euclidean_bins"], camera_indices=None)
    assert_close(out["rgb"], ref["rgb"], 0.0, 5e-4, "rgb: fp16 sampler + fp16 render vs the half-activation oracle")
    assert_close(out["accumulation"], ref["accumulation"], 2e-4, 5e-5, "accumulation")
    # float tables: the mode has no fp16 sampler to offer and must not change a bit
    dp32 = dev_params(scene, table_dtype=torch.float32)
    dh32 = [ops.DensityHandle(dp32, i, ps) for i, ps in enumerate(pspecs)]
    a = ops.proposal_sample(dh32, sc, o, d, nears, fars, (256, 96), 48, matrix_precision=L.MATRIX_F16)
    b = ops.proposal_sample(dh32, sc, o, d, nears, fars, (256, 96), 48)
    assert torch.equal(a["euclidean_bins"], b["euclidean_bins"])
    with pytest.raises(L.CropNerfHipError, match="matrix_precision"):
        ops.proposal_sample(dh, sc, o, d, nears, fars, (256, 96), 48, matrix_precision=9)


@pytest.mark.parametrize("mode", ["f16", "split_bf16"])
def test_early_stop_in_the_team_gather_path_is_bounded(mode, monkeypatch):
    """Early termination (an extension, off by default) in the producer/consumer kernel's TEAM gather -- the path of the fp16
    and split-bf16 matrix modes, where two or four gather waves share a half-step and read each other's stop flags: on a
    batch that fills the device (8 rays per workgroup x 256 CUs and more) every output stays within the threshold of the same
    mode without early stop, and most rays really stop."""
    from cropnerf_amd import _lib as L
    from cropnerf_amd import ops

    sc = make_tcnn_scene(seed=3) if mode == "f16" else make_scene(seed=3)  # half tcnn table / float torch table
    monkeypatch.setenv("CN_FUSED_SPLIT", "2")  # the split kernel also for split-bf16 with early stop (default: single-wave kernel)
    fspec, _ = product_specs(sc)
    dp = {k: v.clone() for k, v in dev_params(sc).items()}
    dp["field.mlp_base_mlp.layers.1.bias"][0] += 4.0  # an opaque medium: rays stop within the first chunks
    fh = ops.FieldHandle(dp, fspec)
    rbs = [rays_with_box(sc, c) for c in range(3)]
    o, d, n, f = (to_dev(torch.cat([getattr(rb, k) for rb in rbs])) for k in ("origins", "directions", "nears", "fars"))
    assert o.shape[0] >= 4096
    scene = ops.scene_struct(sc.aabb, True)
    mp = {"f16": L.MATRIX_F16, "split_bf16": L.MATRIX_SPLIT_BF16}[mode]
    S, eps = 256, 1e-3
    full = ops.render_rays(fh, scene, ops.render_opts(S, matrix_precision=mp), o, d, n, f, want_weights=True)
    again = ops.render_rays(fh, scene, ops.render_opts(S, matrix_precision=mp, early_stop_transmittance=0.0), o, d, n, f,
                            want_weights=True)
    for k in full:
        assert torch.equal(full[k], again[k]), k
    cut = ops.render_rays(fh, scene, ops.render_opts(S, matrix_precision=mp, early_stop_transmittance=eps), o, d, n, f,
                          want_weights=True)
    torch.cuda.synchronize()
    stopped = (cut["weights"][:, 192:] == 0).all(dim=1) & (full["weights"][:, 192:] != 0).any(dim=1)
    assert stopped.float().mean() > 0.5, "the medium was meant to be opaque enough to stop most rays"
    assert_close(cut["rgb"], full["rgb"], 0, 1.01 * eps, f"{mode} early-stop rgb")
    assert_close(cut["accumulation"], full["accumulation"], 0, 1.01 * eps, f"{mode} early-stop accumulation")
    sem_scale = float(full["semantics"].abs().max()) + 1.0
    assert_close(cut["semantics"], full["semantics"], 0, 4 * eps * sem_scale, f"{mode} early-stop semantics")
    assert torch.equal(cut["depth"], full["depth"])
    # the evaluated part of a stopped ray is the same arithmetic: weights in front of the cut are bit-identical
    front = cut["weights"] != 0
    assert torch.equal(cut["weights"][front], full["weights"][front])
