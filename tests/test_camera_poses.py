"""``semantic_projection.py cameras`` and ``collect_camera_poses*`` (CPU, host logic): the frames the reference writes
(``scripts/semantic_projection.py:174-199``, ``export/exporter_utils_nerfacto.py:290-357``) -- optimised poses for the
training split, stored poses for the eval split, the datasets' own image file names, both files, an empty split skipped."""

import json
import math
import types

import numpy as np
import torch


def _cameras(n, seed=0):
    from cropnerf_amd.rays import Cameras

    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
    c2w = torch.cat([q, torch.randn(n, 3, 1, generator=g)], dim=2)
    one = torch.ones(n)
    return Cameras(c2w, one * 500.0, one * 500.0, one * 10.0, one * 10.0, 20, 20)


def test_pose_correction_against_hand_computed_matrices():
    from cropnerf_amd.fruit_nerf.camera_optimizer import CameraOptimizer, pose_correction_matrices

    # a quarter turn about z with translation (1, 2, 3); a half turn about x; the zero vector
    adj = torch.tensor([[1.0, 2.0, 3.0, 0.0, 0.0, math.pi / 2], [0.0, 0.0, 0.0, math.pi, 0.0, 0.0], [0.0] * 6])
    m = pose_correction_matrices(adj)
    want0 = torch.tensor([[0.0, -1.0, 0.0, 1.0], [1.0, 0.0, 0.0, 2.0], [0.0, 0.0, 1.0, 3.0]])
    want1 = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 0.0]])
    assert torch.allclose(m[0], want0, atol=2e-7) and torch.allclose(m[1], want1, atol=2e-7)
    assert torch.equal(m[2], torch.eye(4)[:3])
    # apply_to_camera = c2w @ [[R, t], [0, 1]] (the correction acts in the camera frame): camera looking down -y at (5, 6, 7)
    from cropnerf_amd.rays import Cameras

    c2w = torch.tensor([[[1.0, 0.0, 0.0, 5.0], [0.0, 0.0, 1.0, 6.0], [0.0, -1.0, 0.0, 7.0]]])
    cam = Cameras(c2w, torch.ones(1), torch.ones(1), torch.ones(1), torch.ones(1), 2, 2, {"cam_idx": 0})
    got = CameraOptimizer(adj).apply_to_camera(cam)
    # rotation: columns of c2w[:, :3] @ Rz(90): (col1, -col0, col2); translation: c2w R part @ (1,2,3) + (5,6,7)
    want = torch.tensor([[[0.0, -1.0, 0.0, 6.0], [0.0, 0.0, 1.0, 9.0], [-1.0, 0.0, 0.0, 5.0]]])
    assert torch.allclose(got, want, atol=1e-6)
    # no cam_idx (eval cameras) / optimiser off: the stored pose
    cam.metadata = {}
    assert torch.equal(CameraOptimizer(adj).apply_to_camera(cam), c2w)
    cam.metadata = {"cam_idx": 0}
    assert torch.equal(CameraOptimizer(adj, mode="off").apply_to_camera(cam), c2w)


def test_pose_correction_matches_the_oracles_exponential_map():
    """Same numbers as the oracle's restatement of upstream ``exp_map_SO3xR3`` (``oracle/rays.py``), small angles (below the
    1e-4 clamp on the squared norm) included."""
    from cropnerf_amd.fruit_nerf.camera_optimizer import pose_correction_matrices
    from oracle.rays import exp_map_so3xr3

    g = torch.Generator().manual_seed(3)
    adj = torch.randn(64, 6, generator=g) * torch.logspace(-6, 0.3, 64)[:, None]
    got, ref = pose_correction_matrices(adj), exp_map_so3xr3(adj.double())
    assert (got.double() - ref).abs().max() < 5e-7


def test_collect_camera_poses_train_optimised_eval_original():
    from cropnerf_amd.fruit_nerf.camera_optimizer import CameraOptimizer
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import CameraDataset
    from cropnerf_amd.fruit_nerf.export.exporter_utils_nerfacto import (collect_camera_poses,
                                                                         collect_camera_poses_for_dataset)
    from oracle.rays import exp_map_so3xr3

    train, evalc = _cameras(5, 1), _cameras(2, 2)
    adj = torch.randn(5, 6, generator=torch.Generator().manual_seed(4)) * 0.05
    names = [f"images/frame_{i:04d}.JPG" for i in range(5)]
    pipe = types.SimpleNamespace(
        datamanager=types.SimpleNamespace(train_dataset=CameraDataset(train, names),
                                          eval_dataset=CameraDataset(evalc, ["images/e0.JPG", "images/e1.JPG"])),
        model=types.SimpleNamespace(camera_optimizer=CameraOptimizer(adj)))
    tf, ef = collect_camera_poses(pipe)
    assert [f["file_path"] for f in tf] == names and [f["file_path"] for f in ef] == ["images/e0.JPG", "images/e1.JPG"]
    assert all(set(f) == {"file_path", "transform"} for f in tf + ef)
    corr = exp_map_so3xr3(adj.double())
    for i, f in enumerate(tf):
        m = torch.cat([corr[i], torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)])
        want = train.camera_to_worlds[i].double() @ m
        got = np.asarray(f["transform"])
        assert got.shape == (3, 4) and np.abs(got - want.numpy()).max() < 2e-6
        assert np.abs(got - train.camera_to_worlds[i].numpy()).max() > 1e-3  # the correction is in there
    for i, f in enumerate(ef):  # "returning original poses" (:353-354)
        assert f["transform"] == evalc.camera_to_worlds[i].tolist()
    assert collect_camera_poses_for_dataset(None) == []
    # without an optimiser: the stored poses
    raw = collect_camera_poses_for_dataset(pipe.datamanager.train_dataset)
    assert raw[3]["transform"] == train.camera_to_worlds[3].tolist()


def test_cameras_cli_writes_both_files_and_skips_an_empty_split(tmp_path, monkeypatch, capsys):
    from cropnerf_amd.fruit_nerf import checkpoint
    from cropnerf_amd.fruit_nerf.camera_optimizer import CameraOptimizer
    from cropnerf_amd.fruit_nerf.data.fruit_datamanager import CameraDataset
    from cropnerf_amd.fruit_nerf.scripts import semantic_projection

    train = _cameras(3, 5)
    adj = torch.full((3, 6), 0.01)
    state = {"eval": CameraDataset(_cameras(1, 6), ["images/held_out.png"])}

    def fake_eval_setup(load_config, *a, **k):
        pipe = types.SimpleNamespace(
            datamanager=types.SimpleNamespace(train_dataset=CameraDataset(train, ["a.png", "b.png", "c.png"]),
                                              eval_dataset=state["eval"]),
            model=types.SimpleNamespace(camera_optimizer=CameraOptimizer(adj)))
        return None, pipe, None, 0

    monkeypatch.setattr(checkpoint, "eval_setup", fake_eval_setup)
    out = tmp_path / "poses"
    semantic_projection.entrypoint(["cameras", "--load-config", "unused.yml", "--output-dir", str(out)])
    tr = json.loads((out / "transforms_train.json").read_text())
    ev = json.loads((out / "transforms_eval.json").read_text())
    assert [f["file_path"] for f in tr] == ["a.png", "b.png", "c.png"] and len(ev) == 1
    assert ev[0] == {"file_path": "images/held_out.png", "transform": state["eval"].cameras.camera_to_worlds[0].tolist()}
    # no eval frames: that file is skipped, with a message
    state["eval"] = None
    out2 = tmp_path / "poses2"
    capsys.readouterr()
    semantic_projection.entrypoint(["cameras", "--load-config", "unused.yml", "--output-dir", str(out2)])
    assert (out2 / "transforms_train.json").exists() and not (out2 / "transforms_eval.json").exists()
    assert "No frames found for transforms_eval.json" in capsys.readouterr().out


def test_cameras_slicing_and_rescale():
    cams = _cameras(4, 7)
    one = cams[2:3]
    assert len(one) == 1 and torch.equal(one.camera_to_worlds[0], cams.camera_to_worlds[2])
    one.metadata["cam_idx"] = 2
    assert cams.metadata is None and cams[2].metadata == {}  # a selection owns its metadata
    assert len(cams[1:]) == 3 and cams.size == 4
    cams.rescale_output_resolution(0.5)
    assert (cams.height, cams.width) == (10, 10) and float(cams.fx[0]) == 250.0 and float(cams.cx[0]) == 5.0
