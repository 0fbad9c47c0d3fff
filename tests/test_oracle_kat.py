"""Known-answer tests that pin the CPU oracle.

The reference has no tests or fixtures for this path, so these analytic cases are the only pin.
"""

import math

import numpy as np
import pytest
import torch

from oracle import field as F
from oracle import model as M
from oracle import rays as RY
from oracle import render as RD
from oracle import samplers as SM


def _bundle(n, near, far, o=None, d=None):
    o = torch.zeros(n, 3) if o is None else o
    d = torch.tensor([[0.0, 0.0, 1.0]]).repeat(n, 1) if d is None else d
    return RY.RayBundle(o, d, torch.zeros(n, 1), torch.zeros(n, 1, dtype=torch.long),
                        torch.full((n, 1), float(near)), torch.full((n, 1), float(far)))


# (1) homogeneous medium ------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma", [0.5, 3.0, 40.0])
def test_homogeneous_medium(sigma):
    S, near, far = 64, 0.5, 2.5
    rs = SM.spaced_sampler(_bundle(2, near, far), S)
    dens = torch.full((2, S, 1), sigma)
    w = SM.get_weights(rs.deltas, dens)
    delta = (far - near) / S
    t = near + delta * torch.arange(S)
    expect = torch.exp(-sigma * (t - near)) * (1 - math.exp(-sigma * delta))
    assert torch.allclose(w[0, :, 0], expect, rtol=2e-4, atol=1e-7)
    acc = RD.render_accumulation(w)
    assert torch.allclose(acc, torch.full((2, 1), 1 - math.exp(-sigma * (far - near))), rtol=1e-4, atol=1e-6)
    depth = RD.render_depth_median(w, rs.starts, rs.ends)
    if 1 - math.exp(-sigma * (far - near)) >= 0.5:
        k = math.ceil(math.log(2) / (sigma * delta)) - 1  # first bin whose inclusive cum >= .5
        assert abs(float(depth[0, 0]) - (near + (k + 0.5) * delta)) < 1e-5
    else:
        assert abs(float(depth[0, 0]) - (far - delta / 2)) < 1e-5  # clamp to last sample


# (2) intersect_aabb ----------------------------------------------------------------------------------
def test_intersect_aabb():
    aabb = torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 1.0])
    o = torch.tensor([[-1.0, 0.5, 0.5], [0.5, 0.5, 0.5], [-1.0, 2.0, 0.5], [3.0, 0.5, 0.5]])
    d = torch.tensor([[1.0, 1e-9, 1e-9], [1.0, 1e-9, 1e-9], [1.0, 1e-9, 1e-9], [1.0, 1e-9, 1e-9]])
    tmin, tmax = RY.intersect_aabb(o, d, aabb)
    assert torch.allclose(tmin[:2], torch.tensor([1.0, 0.0]))
    assert torch.allclose(tmax[:2], torch.tensor([2.0, 0.5]))
    assert float(tmin[2]) == 1e10 and float(tmax[2]) == 1e10  # miss
    assert float(tmin[3]) == 1e10 and float(tmax[3]) == 1e10  # box behind the ray


# (3) contraction -------------------------------------------------------------------------------------
def test_contraction():
    x = torch.tensor([[0.3, -0.9, 0.2], [4.0, 0.0, 0.0], [0.0, -2.0, 1.0]])
    c = F.contract_inf(x)
    assert torch.equal(c[0], x[0])
    assert torch.allclose(c[1], torch.tensor([1.75, 0.0, 0.0]))
    assert torch.allclose(c[2], torch.tensor([0.0, -1.5, 0.75]))


# (4) hash grid ---------------------------------------------------------------------------------------
def test_hash_values_and_grid():
    spec = F.GridSpec(num_levels=4, min_res=4, max_res=32, log2_hashmap_size=10)
    T = spec.table_size
    corners = torch.tensor([[1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.int32)[:, None, :].repeat(1, 4, 1)
    h = F.hash_fn(corners, spec)
    base = torch.arange(4) * T
    assert torch.equal(h[0], base + 1)
    assert torch.equal(h[1], base + 2654435761 % T)
    assert torch.equal(h[2], base + 805459861 % T)

    # constant table -> constant output; trilinear weights sum to 1
    table = torch.full((T * 4, 2), 0.37)
    x = torch.rand(50, 3, generator=torch.Generator().manual_seed(1))
    assert torch.allclose(F.hash_grid(x, table, spec), torch.full((50, 8), 0.37), atol=1e-6)

    # integer-lattice input returns the table row of that corner (ceil == floor there)
    table = torch.randn(T * 4, 2, generator=torch.Generator().manual_seed(2))
    s0 = float(spec.scalings()[0])  # = 4
    p = torch.tensor([[2.0 / s0, 1.0 / s0, 3.0 / s0]])
    enc = F.hash_grid(p, table, spec)
    idx = int((2 * 1) ^ (1 * 2654435761) ^ (3 * 805459861)) % T
    assert torch.allclose(enc[0, 0:2], table[idx], atol=1e-6)


def test_grid_scalings_default():
    # upstream evaluates ``np.float64 ** int64 Tensor`` through Tensor.__rpow__, i.e. in float32:
    # floor(16 * 127.9999) = 2047, not 2048.  The literal expression is kept so the oracle follows it.
    s = F.GridSpec().scalings()
    assert s.tolist() == [16, 22, 30, 42, 58, 80, 111, 153, 212, 294, 406, 561, 776, 1072, 1482, 2047]
    assert torch.all(s[1:] > s[:-1])


# (5) uniform sampler (eval) --------------------------------------------------------------------------
def test_uniform_sampler_eval():
    S, near, far = 10, 0.2, 4.2
    rs = SM.spaced_sampler(_bundle(3, near, far), S)
    i = torch.arange(S, dtype=torch.float32)
    assert torch.allclose(rs.starts[1, :, 0], near + (far - near) * i / S, atol=1e-6)
    assert torch.allclose(rs.ends[1, :, 0], near + (far - near) * (i + 1) / S, atol=1e-6)
    assert torch.allclose(rs.spacing_starts[0, :, 0], i / S, atol=1e-7)


def test_uniform_sampler_jitter_bounds():
    S = 16
    g = torch.Generator().manual_seed(0)
    t_rand = torch.rand(5, S + 1, generator=g)
    rs = SM.spaced_sampler(_bundle(5, 0.0, 1.0), S, t_rand=t_rand)
    b = torch.cat([rs.spacing_starts[..., 0], rs.spacing_ends[:, -1:, 0]], -1)
    assert torch.all(b[:, 1:] >= b[:, :-1]) and b.min() >= 0 and b.max() <= 1


# (6) PDF sampler -------------------------------------------------------------------------------------
def test_pdf_equal_weights_and_onehot():
    S_in, S_out = 8, 8
    prev = SM.spaced_sampler(_bundle(1, 0.0, 1.0), S_in)
    rs = SM.pdf_sampler(prev, torch.ones(1, S_in, 1), S_out)
    nb = S_out + 1
    u = torch.linspace(0, 1 - 1 / nb, nb) + 1 / (2 * nb)
    bins = torch.cat([rs.spacing_starts[0, :, 0], rs.spacing_ends[0, -1:, 0]])
    assert torch.allclose(bins, u, atol=1e-6)  # uniform cdf is the identity on [0,1]

    w = torch.zeros(1, S_in, 1)
    w[0, 3] = 1.0
    rs = SM.pdf_sampler(prev, w, 32)
    bins = torch.cat([rs.spacing_starts[0, :, 0], rs.spacing_ends[0, -1:, 0]])
    inside = ((bins >= 3 / 8 - 1e-6) & (bins <= 4 / 8 + 1e-6)).float().mean()
    assert inside > 0.9  # all mass but the 0.01 histogram padding
    assert torch.all(bins[1:] >= bins[:-1])


# (7) piecewise spacing -------------------------------------------------------------------------------
def test_piecewise_spacing():
    fn, inv = SM.SPACING["piecewise"]
    x = torch.tensor([0.5, 1.0, 2.0, 7.0])
    assert torch.allclose(fn(x)[:3], torch.tensor([0.25, 0.5, 0.75]))
    assert torch.allclose(inv(fn(x)), x, atol=1e-5)


# (8) SH ----------------------------------------------------------------------------------------------
def test_sh_at_z():
    c = F.sh_deg4(torch.tensor([[0.0, 0.0, 1.0]]))[0]
    expect = torch.zeros(16)
    expect[0], expect[2], expect[6], expect[12] = 0.28209479, 0.48860251, 0.63078313, 0.74635267
    assert torch.allclose(c, expect, atol=1e-6)


# (9) colormap threshold ------------------------------------------------------------------------------
def test_semantics_colormap_threshold():
    ln9 = math.log(9.0)
    sem = torch.tensor([[ln9 + 1e-3], [ln9 - 1e-3], [-4.0], [50.0]])
    cm = RD.semantics_colormap(sem)
    assert cm.shape == (4, 3)
    assert torch.equal(cm[:, 0], torch.tensor([1.0, 0.0, 0.0, 1.0]))
    # heaviside(0, 0) = 0: sigmoid(x) == 0.9 exactly maps to 0
    assert float(torch.heaviside(torch.tensor(0.0), torch.tensor(0.0))) == 0.0


# (10) export masks inclusive ----------------------------------------------------------------------------
def test_export_masks_inclusive():
    out = {
        "point_location": torch.arange(12, dtype=torch.float32).reshape(1, 4, 3),
        "semantics": torch.tensor([[3.0, 2.999, 3.0, 10.0]]),
        "density": torch.tensor([[70.0, 70.0, 69.999, 1e3]]),
        "rgb": torch.rand(1, 4, 3),
    }
    out["semantics_colormap"] = torch.heaviside(torch.sigmoid(out["semantics"]) - 0.9, torch.tensor(0.0)).long()
    m = M.sample_volume_masks(out)
    assert m["semantic"]["points"].shape[0] == 2  # samples 0 and 3
    assert m["density"]["points"].shape[0] == 3  # 0, 1, 3
    assert m["semantic_colormap"]["points"].shape[0] == 3  # sigmoid(2.999) > .9 too
    assert m["density"]["colors"].shape[1] == 4


# (11) orthographic grid -------------------------------------------------------------------------------
def test_orthographic_grid():
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    corners = RY.corners_of_aabb(aabb)
    pts, plane = RY.surface_points(corners, 4)
    assert pts.shape == (16, 3)
    assert torch.allclose(pts[:, 2], torch.full((16,), -0.682))
    assert torch.allclose(plane, torch.tensor([[0.0, 0.0, 2.0]]), atol=1e-6)
    rb = RY.ortho_rays(pts, plane, batch=6, count=3)  # rays [12,16)
    assert len(rb) == 4
    assert torch.allclose(rb.directions, torch.tensor([[0.0, 0.0, 1.0]]).repeat(4, 1))
    assert torch.allclose(rb.fars, torch.full((4, 1), 2.0), atol=1e-6) and torch.all(rb.nears == 0)
    # grid is "ij": x varies slowest
    assert torch.allclose(pts[:4, 0], torch.full((4,), -1.0)) and torch.allclose(pts[:4, 1], torch.linspace(-1, 1, 4))


# (12) last_sample background ----------------------------------------------------------------------------
def test_last_sample_background():
    g = torch.Generator().manual_seed(3)
    rgb = torch.rand(4, 7, 3, generator=g)
    w = torch.rand(4, 7, 1, generator=g) * 0.1
    out = RD.render_rgb(rgb, w, "last_sample")
    expect = (w * rgb).sum(1) + rgb[:, -1] * (1 - w.sum(1))
    assert torch.allclose(out, expect.clamp(0, 1), atol=1e-6)
    black = RD.render_rgb(rgb, w, torch.zeros(3))
    assert torch.allclose(black, (w * rgb).sum(1), atol=1e-6)


# pinhole geometry ------------------------------------------------------------------------------------
def test_pinhole_center_ray_and_pixel_area():
    c2w = torch.eye(4)[:3][None]
    intr = torch.tensor([[100.0, 100.0, 8.0, 8.0]])
    # pixel (row 7, col 7): centre 7.5 -> (-0.5/100, +0.5/100, -1)
    rb = RY.pinhole_rays(c2w, intr, torch.tensor([0]), torch.tensor([7]), torch.tensor([7]))
    v = torch.tensor([-0.005, 0.005, -1.0])
    assert torch.allclose(rb.directions[0], v / v.norm(), atol=1e-6)
    assert torch.allclose(rb.origins[0], torch.zeros(3))
    assert abs(float(rb.pixel_area[0, 0]) - 1e-4) < 2e-6
    full = RY.image_rays(c2w, intr, 0, 16, 16)
    assert len(full) == 256 and torch.allclose(full.directions[7 * 16 + 7], rb.directions[0])


def test_pose_adjustment_identity_and_translation():
    rb = _bundle(3, 0, 1)
    rb.camera_indices = torch.tensor([[0], [1], [1]])
    adj = torch.zeros(2, 6)
    adj[1, :3] = torch.tensor([0.1, -0.2, 0.3])
    out = RY.apply_pose_adjustment(rb, adj)
    assert torch.allclose(out.directions, rb.directions, atol=1e-6)
    assert torch.allclose(out.origins[1], torch.tensor([0.1, -0.2, 0.3]))
    adj[0, 5] = math.pi / 2  # rotate about z
    rb.directions = torch.tensor([[1.0, 0.0, 0.0]]).repeat(3, 1)
    out = RY.apply_pose_adjustment(rb, adj)
    assert torch.allclose(out.directions[0], torch.tensor([0.0, 1.0, 0.0]), atol=1e-5)


# collider --------------------------------------------------------------------------------------------
def test_collider():
    rb = _bundle(2, 0, 1)
    rb.nears = rb.fars = None
    ev = RY.near_far_collider(rb, training=False)
    tr = RY.near_far_collider(rb, training=True)
    assert float(ev.nears[0]) == 0.0 and float(tr.nears[0]) == pytest.approx(0.05) and float(ev.fars[0]) == 1000.0
    keep = RY.near_far_collider(_bundle(2, 0.3, 0.7), training=True)
    assert float(keep.nears[0]) == pytest.approx(0.3)


# end-to-end shape / dict-key check on a tiny model -----------------------------------------------------
def _tiny_model(test_mode="test"):
    fs = F.FieldSpec(grid=F.GridSpec(num_levels=16, min_res=16, max_res=256, log2_hashmap_size=12), num_images=3)
    ps = [F.ProposalSpec(F.GridSpec(5, 16, 64, 10)), F.ProposalSpec(F.GridSpec(5, 16, 128, 10))]
    cfg = M.ModelConfig(field=fs, proposals=ps, num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8,
                        eval_num_rays_per_chunk=50)
    params = F.random_params(fs, ps, seed=0)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    return M.OracleModel(params, cfg, aabb, test_mode=test_mode)


def test_forward_keys_and_chunking():
    m = _tiny_model()
    c2w = torch.tensor([[[1.0, 0, 0, 0.1], [0, 1, 0, 0.0], [0, 0, 1, 0.9]]])
    rb = RY.image_rays(c2w, torch.tensor([[20.0, 20.0, 6.0, 6.0]]), 0, 12, 12)
    out = m.render_rays(rb)
    for k, c in (("rgb", 3), ("accumulation", 1), ("depth", 1), ("prop_depth_0", 1), ("prop_depth_1", 1),
                 ("semantics", 1), ("semantics_colormap", 3)):
        assert out[k].shape == (144, c), k
    assert torch.isfinite(out["rgb"]).all() and out["rgb"].min() >= 0 and out["rgb"].max() <= 1
    one = m.forward(rb.slice(10, 30))
    assert torch.allclose(one["rgb"], out["rgb"][10:30], atol=1e-6)


def test_export_mode_outputs():
    m = _tiny_model("export")
    m.setup_inference(True, 20)
    aabb = torch.tensor([[-1.0, -1.0, -0.682], [1.0, 1.0, 1.318]])
    pts, plane = RY.surface_points(RY.corners_of_aabb(aabb), 5)
    out = m.forward(RY.ortho_rays(pts, plane, 8, 1))
    assert out["rgb"].shape == (8, 20, 3) and out["point_location"].shape == (8, 20, 3)
    assert out["semantics"].shape == (8, 20) and out["density"].shape == (8, 20)
    assert out["semantics_colormap"].dtype == torch.long
    z = out["point_location"][0, :, 2]
    assert torch.allclose(z, -0.682 + 2.0 * (torch.arange(20) + 0.5) / 20, atol=1e-5)


def test_projection_two_pass_small():
    m = _tiny_model()
    c2w = torch.tensor([[[1.0, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 0.9]]])
    rb = RY.image_rays(c2w, torch.tensor([[20.0, 20.0, 8.0, 8.0]]), 0, 16, 16, camera_index_value=0)
    box = torch.tensor([[-0.2, -0.2, -0.2], [0.2, 0.2, 0.2]])
    wo, vis = m.project_cluster(rb, box, 16, 16)
    assert wo.shape == (16, 16, 3) and vis.shape == (16, 16, 3)
    hit = RY.with_aabb_near_far(rb, box.reshape(-1)).nears[:, 0] < 1e10
    assert torch.all(wo.reshape(-1, 3)[~hit] == 0)
    assert int(hit.sum()) >= 10
    far_box = torch.tensor([[5.0, 5.0, 5.0], [6.0, 6.0, 6.0]])
    wo2, vis2 = m.project_cluster(rb, far_box, 16, 16)
    assert float(wo2.abs().sum()) == 0 and float(vis2.abs().sum()) == 0
