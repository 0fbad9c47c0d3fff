"""Known-answer tests of ``oracle/tcnn.py`` (tiny-cuda-nn semantics restated from its published source) and CPU checks
of the product-side converter ``cropnerf_amd.fruit_nerf.tcnn_params`` against it.  No GPU."""

import math

import numpy as np
import pytest
import torch

from oracle import field as OF
from oracle import tcnn as TC

from cropnerf_amd.config import FieldSpec, GridSpec, ProposalSpec
from cropnerf_amd.fruit_nerf import tcnn_params as TP


def test_grid_geometry_default_field():
    g = TC.TcnnGridSpec()  # 16 levels, 16 -> 2048, T = 2^19
    assert float(g.scales()[0]) == 15.0  # exp2(0) * 16 - 1: base_resolution counts vertices
    res = g.resolutions()
    assert res[:6] == [16, 23, 31, 43, 59, 81]
    offs = g.offset_table()
    sizes = [offs[i + 1] - offs[i] for i in range(16)]
    # dense levels: res^3 rounded up to a multiple of 8; 81^3 = 531441 > 2^19 -> hashed from level 5 on
    assert sizes[:5] == [4096, 12168, 29792, 79512, 205384]
    assert all(s == 1 << 19 for s in sizes[5:])
    assert g.n_params == 2 * offs[-1] == 12196240


def test_grid_geometry_matches_the_library_plan():
    for args in [(16, 16, 2048, 19), (5, 16, 128, 17), (5, 16, 256, 17), (7, 16, 512, 19), (16, 16, 4096, 21)]:
        o = TC.TcnnGridSpec(*args)
        p = GridSpec(*args, 2, "tcnn").plan()
        L = args[0]
        assert [int(p.resolution[i]) for i in range(L)] == o.resolutions()
        assert [int(p.packed_offset[i]) for i in range(L + 1)] == o.offset_table()
        a = np.array([p.scalings[i] for i in range(L)], dtype=np.float32)
        b = np.array(o.scales(), dtype=np.float32)
        assert np.all(np.abs(a - b) <= np.spacing(np.maximum(a, b)))  # exp2f of two libms: one ulp
        for l in range(L):
            dense = o.resolutions()[l] ** 3 <= (1 << args[3])
            bits = int(p.level_bits[l])
            assert (bits > 0) == dense
            if dense:
                assert (1 << bits) >= o.resolutions()[l] + 1 > (1 << (bits - 1))


def test_grid_index_dense_hashed_and_wrap():
    # dense: x + y*res + z*res^2
    pg = np.array([[1, 2, 3]], dtype=np.uint32)
    assert TC.grid_index(pg, 16, 4096)[0] == 1 + 2 * 16 + 3 * 256
    # the corner index `res` is not clamped: (16, 0, 0) is the first element of the next row
    assert TC.grid_index(np.array([[16, 0, 0]], dtype=np.uint32), 16, 4096)[0] == TC.grid_index(
        np.array([[0, 1, 0]], dtype=np.uint32), 16, 4096)[0]
    # (res, res-1, res-1) wraps around the level: res^3 mod size
    assert TC.grid_index(np.array([[23, 22, 22]], dtype=np.uint32), 23, 12168)[0] == 23 ** 3 % 12168 == 12167
    assert TC.grid_index(np.array([[16, 15, 15]], dtype=np.uint32), 16, 4096)[0] == 0
    # hashed: xor of coordinate * prime in uint32, mod T
    T = 1 << 19
    for c, want in (((1, 0, 0), 1), ((0, 1, 0), 2654435761 % T), ((0, 0, 1), 805459861 % T),
                    ((3, 5, 7), (3 ^ ((5 * 2654435761) & 0xFFFFFFFF) ^ ((7 * 805459861) & 0xFFFFFFFF)) % T)):
        assert TC.grid_index(np.array([c], dtype=np.uint32), 81, T)[0] == want


def test_hash_grid_partition_of_unity_and_lattice_points():
    g = TC.TcnnGridSpec(num_levels=4, min_res=16, max_res=64, log2_hashmap_size=13)
    gen = torch.Generator().manual_seed(0)
    x = torch.rand(64, 3, generator=gen)
    const = torch.full((g.n_params,), 0.375)
    enc = TC.hash_grid(x, const, g)
    assert enc.shape == (64, 8)
    assert torch.allclose(enc, torch.full_like(enc, 0.375), atol=1e-6)
    # x = (k - 0.5) / scale puts pos on the lattice point k: the encoding is that entry
    params = torch.rand(g.n_params, generator=gen)
    k = torch.tensor([[3.0, 4.0, 5.0]])
    xs = (k - 0.5) / float(g.scales()[0])
    enc = TC.hash_grid(xs, params, g, half_params=False)
    row = 3 + 4 * 16 + 5 * 256
    assert torch.allclose(enc[0, :2], params[2 * row:2 * row + 2], atol=1e-6)


def test_half_params_are_the_fp16_cast_of_the_master_copy():
    g = TC.TcnnGridSpec(num_levels=2, min_res=16, max_res=32, log2_hashmap_size=12)
    gen = torch.Generator().manual_seed(1)
    p = torch.rand(g.n_params, generator=gen)
    x = torch.rand(16, 3, generator=gen)
    a = TC.hash_grid(x, p, g, half_params=True)
    b = TC.hash_grid(x, p.to(torch.float16).to(torch.float32), g, half_params=False)
    assert torch.equal(a, b)


def test_sh_degree4_values_and_sign_convention():
    u = torch.tensor([[0.5, 0.5, 1.0]])  # d = (0, 0, 1) after the 2u - 1 map
    c = TC.sh_deg4(u)[0]
    want = {0: 0.28209479, 2: 0.48860251, 6: 0.63078313, 12: 0.74635267}
    for i in range(16):
        assert abs(float(c[i]) - want.get(i, 0.0)) < 1e-6
    # against nerfstudio's torch components: the odd-numbered components flip sign, the rest agree
    gen = torch.Generator().manual_seed(2)
    d = torch.nn.functional.normalize(torch.randn(32, 3, generator=gen), dim=-1)
    t = TC.sh_deg4((d + 1) / 2)
    n = OF.sh_deg4(d)
    sign = torch.ones(16)
    sign[list(TP.SH_FLIPPED)] = -1
    assert torch.allclose(t, n * sign, atol=1e-5)


def test_fully_fused_mlp_layout_and_identity_padding():
    assert TC.mlp_param_count(32, 16, 64, 1) == 64 * 32 + 16 * 64
    assert TC.mlp_param_count(15, 64, 64, 1) == 64 * 16 + 64 * 64
    assert TC.mlp_param_count(63, 3, 64, 2) == 64 * 64 + 64 * 64 + 16 * 64
    # 15 inputs are padded to 16 with a ONE: a network whose only non-zero first-layer column is the padded one
    # ignores its input, i.e. the column is a bias
    n = TC.mlp_param_count(15, 64, 64, 1)
    p = torch.zeros(n)
    w0 = p[:64 * 16].view(64, 16)
    w0[:, 15] = torch.arange(64, dtype=torch.float32) / 64 - 0.25
    p[64 * 16:].view(64, 64).copy_(torch.eye(64))
    y = TC.network(torch.randn(5, 15), p, 15, 64, 64, 1, half_params=False)
    assert torch.allclose(y, torch.relu(w0[:, 15]).expand(5, 64))
    # matrices are [out, in] row-major: y_j = sum_k W[j, k] x_k
    p = torch.zeros(TC.mlp_param_count(16, 16, 16, 1))
    p[:256].view(16, 16)[3, 5] = 2.0  # hidden neuron 3 <- input 5
    p[256:].view(16, 16)[7, 3] = 0.5  # output 7 <- hidden 3
    x = torch.zeros(1, 16)
    x[0, 5] = 1.5
    y = TC.network(x, p, 16, 16, 16, 1, half_params=False)
    assert float(y[0, 7]) == 1.5 and float(y.abs().sum()) == 1.5


def test_network_with_grid_parameter_order_and_zero_padding():
    g = TC.TcnnGridSpec(num_levels=5, min_res=16, max_res=128, log2_hashmap_size=12)  # 10 outputs -> padded to 16
    n_mlp = TC.mlp_param_count(10, 1, 16, 1)
    assert n_mlp == 16 * 16 + 16 * 16
    gen = torch.Generator().manual_seed(3)
    p = torch.rand(n_mlp + g.n_params, generator=gen) - 0.5
    x = torch.rand(9, 3, generator=gen)
    y = TC.network_with_grid(x, p, g, 1, 16, 1, half_params=False)
    enc = TC.hash_grid(x, p[n_mlp:], g, half_params=False)
    w0 = p[:256].view(16, 16)
    w1 = p[256:512].view(16, 16)
    want = torch.relu(enc @ w0[:, :10].t()) @ w1[0]  # the padded columns 10..15 see zeros
    assert torch.allclose(y[:, 0], want, atol=1e-6)


def _specs(log2_T=12, prop_log2_T=10, num_images=5):
    fs = FieldSpec(grid=GridSpec(16, 16, 2048, log2_T, 2, "tcnn"), num_images=num_images)
    ps = [ProposalSpec(GridSpec(5, 16, 128, prop_log2_T, 2, "tcnn")), ProposalSpec(GridSpec(5, 16, 256, prop_log2_T, 2, "tcnn"))]
    ofs = OF.FieldSpec(grid=OF.GridSpec(16, 16, 2048, log2_T), num_images=num_images, implementation="tcnn")
    ops_ = [OF.ProposalSpec(OF.GridSpec(5, 16, 128, prop_log2_T), implementation="tcnn"),
            OF.ProposalSpec(OF.GridSpec(5, 16, 256, prop_log2_T), implementation="tcnn")]
    return fs, ps, ofs, ops_


def test_converted_mlps_reproduce_the_tcnn_field_on_cpu():
    """mlp_to_linear + the SH sign fold: the nn.Linear stacks the kernels take give the oracle's tcnn outputs."""
    fs, ps, ofs, ops_ = _specs()
    state = TC.random_params(ofs, ops_, seed=4)
    gen = torch.Generator().manual_seed(5)
    N = 40
    enc = torch.randn(N, 32, generator=gen)
    base = TP.mlp_to_linear(state["field.mlp_base_mlp.tcnn_encoding.params"], 32, 16, 64, 1)
    h = torch.relu(enc @ base[0][0].t() + base[0][1]) @ base[1][0].t() + base[1][1]
    want = TC.network(enc, state["field.mlp_base_mlp.tcnn_encoding.params"], 32, 16, 64, 1)
    assert torch.allclose(h, want, atol=1e-5)
    assert not base[0][1].any() and not base[1][1].any()  # 32 inputs: nothing padded, no bias at all
    geo = h[:, 1:]
    sem = TP.mlp_to_linear(state["field.mlp_semantics.tcnn_encoding.params"], 15, 64, 64, 1)
    assert sem[0][1].abs().sum() > 0  # 15 -> 16: the padded column became a bias
    x = torch.relu(geo @ sem[0][0].t() + sem[0][1]) @ sem[1][0].t() + sem[1][1]
    assert torch.allclose(x, TC.network(geo, state["field.mlp_semantics.tcnn_encoding.params"], 15, 64, 64, 1), atol=1e-5)
    # colour head: tcnn SH on the shifted direction vs the kernels' (torch) SH on the unit direction
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=-1)
    app = torch.randn(N, 32, generator=gen)
    want = TC.network(torch.cat([TC.sh_deg4((d + 1) / 2), geo, app], -1), state["field.mlp_head.tcnn_encoding.params"],
                      63, 3, 64, 2, "sigmoid")
    head = TP.mlp_to_linear(state["field.mlp_head.tcnn_encoding.params"], 63, 3, 64, 2)
    w0 = head[0][0].clone()
    w0[:, list(TP.SH_FLIPPED)] *= -1
    y = torch.cat([OF.sh_deg4(d), geo, app], -1)
    y = torch.relu(y @ w0.t() + head[0][1])
    y = torch.relu(y @ head[1][0].t() + head[1][1])
    y = torch.sigmoid(y @ head[2][0].t() + head[2][1])
    assert torch.allclose(y, want, atol=1e-5)


def test_mlp_round_trip_and_unrepresentable_bias():
    gen = torch.Generator().manual_seed(6)
    p = (torch.rand(TC.mlp_param_count(63, 3, 64, 2), generator=gen) - 0.5)
    layers = TP.mlp_to_linear(p, 63, 3, 64, 2, half=False)
    q = TP.linear_to_mlp(layers, 63, 3, 64, 2)
    # everything tcnn reads survives: the real columns / rows and the (single) padded input column
    mp, mq = TC.mlp_matrices(p, 63, 3, 64, 2), TC.mlp_matrices(q, 63, 3, 64, 2)
    assert torch.equal(mp[0], mq[0]) and torch.equal(mp[1], mq[1]) and torch.equal(mp[2][:3], mq[2][:3])
    assert not mq[2][3:].any()  # padded output rows are written as zeros
    layers[1] = (layers[1][0], torch.ones(64))
    with pytest.raises(ValueError):
        TP.linear_to_mlp(layers, 63, 3, 64, 2)
    names = TP.frozen_parameter_names(*_specs()[:2])
    assert "field.mlp_base_mlp.layers.0.bias" in names and "field.mlp_semantics.layers.0.bias" not in names
    assert "field.mlp_head.layers.0.bias" not in names and "field.mlp_head.layers.2.bias" in names
    assert "proposal_networks.0.mlp.layers.0.bias" in names  # grid encodings pad with zeros


def test_tcnn_field_oracle_runs_and_half_activation_diagnostic_is_close():
    fs, ps, ofs, ops_ = _specs()
    state = TC.random_params(ofs, ops_, seed=7)
    gen = torch.Generator().manual_seed(8)
    pos = torch.rand(6, 7, 3, generator=gen) * 1.8 - 0.9
    d = torch.nn.functional.normalize(torch.randn(6, 3, generator=gen), dim=-1)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    out = OF.field_forward(pos, d, None, state, ofs, aabb, False, "inference")
    assert out["rgb"].shape == (6, 7, 3) and out["density"].shape == (6, 7, 1) and out["semantics"].shape == (6, 7, 1)
    import dataclasses

    ha = dataclasses.replace(ofs, tcnn_half_activations=True)
    out_h = OF.field_forward(pos, d, None, state, ha, aabb, False, "inference")
    # fp16 activations (what tcnn's kernels round to) stay within ~1e-2 of the fp32-arithmetic evaluation
    assert (out_h["rgb"] - out["rgb"]).abs().max() < 2e-2
    den = OF.proposal_density(pos, state, 0, ops_[0], aabb, True)
    assert den.shape == (6, 7, 1) and bool((den >= 0).all())
