"""World-size-2 (and 3) gloo tests of the ray-batch sharding layer, on CPU.  The render function is injected, so the
CPU oracle stands in for the HIP renderer here (tests may use the oracle; the product path cannot)."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn_name, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return dict(ret)


def _case_render(rank, world):
    from _helpers import make_scene, oracle_model, rays_with_box
    from cropnerf_amd import distributed as D

    sc = make_scene(seed=2, log2_T=10, num_images=3, height=9, width=7, focal=10.0, prop_log2_T=8)
    m = oracle_model(sc, "inference", disable_scene_contraction=True)
    m.uniform_samples = 16
    rb = rays_with_box(sc, 0)
    calls = []

    def render(lo, hi):
        calls.append((lo, hi))
        return m.forward(rb.slice(lo, hi))

    full = D.render_rays_sharded(render, len(rb))
    ref = m.forward(rb)
    ok = all(torch.allclose(full[k], ref[k], atol=1e-6) for k in ("rgb", "accumulation", "depth", "semantics"))
    return {"ok": ok, "calls": calls, "n": len(rb)}


def _case_points(rank, world):
    from cropnerf_amd import distributed as D

    g = torch.Generator().manual_seed(rank)
    n = [5, 0, 11][rank % 3]  # ragged, including an empty rank
    rows = torch.rand(n, 7, generator=g) + rank
    out = D.all_gather_points(rows)
    mean = D.all_reduce_mean(torch.tensor([float(rank)]))
    return {"shape": tuple(out.shape), "first": [float(out[:, 0].min()), float(out[:, 0].max())], "mean": float(mean)}


def _case_grad_allreduce(rank, world):
    """Data-parallel gradient averaging (what FruitTrainer.all_reduce_gradients does with its flat buffer)."""
    from cropnerf_amd import distributed as D

    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    out = D.all_reduce_mean(flat)
    return out.tolist()


def _case_grouped_exchange(rank, world):
    """Three data-parallel optimiser steps on a flat buffer with the trainer's group layout: the overlapped per-group exchange
    (collectives started group by group in backward order, each group stepped behind its own) against the blocking whole-buffer
    all-reduce -- same parameters, bit for bit; a group without a gradient (the proposal networks, two iterations in three here)
    exchanges nothing and keeps its values."""
    from cropnerf_amd import distributed as D

    ranges = {"fields": (0, 4000), "proposal_networks": (4000, 5200), "camera_opt": (5200, 5236)}
    lr = {"fields": 1e-2, "proposal_networks": 1e-2, "camera_opt": 1e-3}

    def grads(step):
        g = torch.Generator().manual_seed(1000 * step + rank)
        return torch.randn(5236, generator=g)

    def run(overlap):
        params = torch.linspace(-1, 1, 5236)
        flat = torch.zeros(5236)
        ex = D.GroupedGradientExchange(flat, ranges)
        order = []
        for step in range(3):
            updated = step % 3 == 0
            flat.copy_(grads(step))
            if overlap:
                ex.begin_iteration()
                ex.start("fields")  # ... proposal backward kernels would run here ...
                if updated:
                    ex.start("proposal_networks")
                ex.start("camera_opt")
                order.append(list(ex.started))
            else:
                flat.copy_(D.all_reduce_mean(flat))
            for name, (lo, hi) in ranges.items():
                if name == "proposal_networks" and not updated:
                    continue
                if overlap:
                    ex.wait(name)
                params[lo:hi] -= lr[name] * flat[lo:hi]
            if overlap:
                ex.wait_all()
        return params, order

    a, order = run(True)
    b, _ = run(False)
    # an exchange that was started and never waited for is an error at the next iteration
    ex = D.GroupedGradientExchange(torch.zeros(8), {"fields": (0, 8)})
    ex.begin_iteration()
    ex.start("fields")
    try:
        ex.begin_iteration()
        leak = False
    except RuntimeError:
        leak = True
    ex.wait_all()
    return {"equal": bool(torch.equal(a, b)), "order": order, "sum": float(a.double().sum()), "leak_detected": leak}


def test_grouped_gradient_exchange_equals_the_blocking_all_reduce():
    res = _run("_case_grouped_exchange", 2)
    assert all(r["equal"] and r["leak_detected"] for r in res.values())
    assert res[0]["sum"] == res[1]["sum"]  # both ranks hold the same parameters
    assert res[0]["order"] == [["fields", "proposal_networks", "camera_opt"], ["fields", "camera_opt"], ["fields", "camera_opt"]]
    assert res[0]["order"] == res[1]["order"]  # every rank issues the same collectives in the same order


@pytest.mark.gpu
def test_trainer_overlapped_exchange_on_one_rank_rccl():
    """``FruitTrainer.train_iteration`` with the data-parallel exchange forced on in a ONE-rank RCCL group: the per-group
    in-place ``ReduceOp.AVG`` collectives on RCCL's stream between the backward kernels and the optimiser step -- the call
    pattern (async issue behind the group's last writer, stream-side wait before the group's Adam step) is valid on this box,
    a mean over one rank changes nothing (losses fall as without it), and the blocking form (``CN_DP_EXCHANGE=blocking``) runs
    too.  The multi-rank arithmetic is the gloo test above."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_train import _hip_model, _hip_rays, _setup
    from cropnerf_amd.fruit_nerf.trainer import FruitTrainer

    def _trainer_setup():
        sc, idx, _, image, mask = _setup(seed=6, R=128)
        model = _hip_model(sc)
        model.training = True
        return FruitTrainer(model, seed=3), _hip_rays(sc, idx), {"image": image, "fruit_mask": mask}

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        losses = {}
        for mode in ("overlap", "blocking", "none"):
            os.environ["CN_DP_EXCHANGE"] = mode if mode != "none" else "overlap"
            tr, rb, batch = _trainer_setup()
            tr.force_exchange = mode != "none"
            hist = []
            for _ in range(6):
                out = tr.train_iteration(rb, batch)
                hist.append(float(out["loss_dict"]["rgb_loss"]))
            torch.cuda.synchronize()
            if mode == "overlap":
                assert tr._exchange is not None and tr._exchange.started == ["fields", "proposal_networks", "camera_opt"]
                assert not tr._exchange.pending
            losses[mode] = hist
        for mode in ("overlap", "blocking"):
            assert losses[mode][-1] < losses[mode][0]
            for a, b in zip(losses[mode], losses["none"]):  # the same trajectory up to the order of the atomic sums
                assert abs(a - b) <= 2e-2 * abs(b) + 1e-5, (mode, losses)
    finally:
        os.environ.pop("CN_DP_EXCHANGE", None)
        dist.destroy_process_group()


def test_gradient_all_reduce_mean():
    res = _run("_case_grad_allreduce", 2)
    expect = [1.5 * i for i in range(10)]
    assert res[0] == expect and res[1] == expect


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_render_equals_single_rank(world):
    res = _run("_case_render", world)
    n = res[0]["n"]
    assert all(r["ok"] for r in res.values())
    ranges = sorted(r["calls"][0] for r in res.values())
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))  # contiguous, disjoint
    assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 1  # balanced


def test_ragged_point_all_gather():
    res = _run("_case_points", 2)
    assert all(r["shape"] == (5, 7) for r in res.values())  # 5 + 0 rows
    assert abs(res[0]["mean"] - 0.5) < 1e-6
    res = _run("_case_points", 3)
    assert all(r["shape"] == (16, 7) for r in res.values())
    assert all(r["first"][0] < 1.0 and r["first"][1] >= 2.0 for r in res.values())


def test_shard_helpers_single_process():
    from cropnerf_amd import distributed as D

    assert [D.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert D.shard_range(0, 0, 2) == (0, 0)
    assert D.shard_jobs(list(range(7)), 1, 3) == [1, 4]
    t = torch.arange(6.0).reshape(3, 2)
    assert D.all_gather_points(t) is t  # world size 1: no communication


@pytest.mark.gpu
def test_rccl_async_all_gather_single_rank():
    """The collective pattern of bench.py's N > 1 path (double-buffered ``all_gather_into_tensor(async_op=True)`` over
    RCCL, stream-side waits) on a one-rank RCCL group: checks that RCCL loads and the pattern is valid on this box.  The
    multi-rank arithmetic is covered by the gloo tests above."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bufs = [torch.empty(1024, 6, device="cuda") for _ in range(2)]
        pending, sent = [], []
        for i in range(5):
            packed = torch.full((1024, 6), float(i), device="cuda")
            if len(pending) == 2:
                pending.pop(0).wait()
            pending.append(dist.all_gather_into_tensor(bufs[i % 2], packed, async_op=True))
            sent.append(packed)
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        dist.barrier()
        assert float(bufs[0][0, 0]) == 4.0 and float(bufs[1][0, 0]) == 3.0
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_exporters_two_ranks_rehearsal():
    """The exporters' N > 1 paths on the product code: point-cloud export (every rank collects its share from its own random
    stream -- also inside the captured HIP graph --, one variable-length all-gather) and projection jobs (dealt
    round-robin, no communication).  Two ranks on this box's single GPU over gloo (tests/_dist_export_worker.py); on a
    multi-GPU node the same code runs over RCCL."""
    import subprocess

    worker = os.path.join(ROOT, "tests", "_dist_export_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), worker]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sharded export ok" in p.stdout, p.stdout[-1500:] + p.stderr[-2500:]


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """``bench.py --gpus 2`` launched exactly as the driver launches it (``torch.distributed.run``, fresh child processes),
    rehearsed on this box's single GPU (``BENCH_REHEARSE_ON_ONE_GPU=1``: both ranks on cuda:0, exchange over gloo; on a
    multi-GPU node the same code path runs one rank per GPU over RCCL): one JSON line from rank 0, n_gpus 2, the sharding
    string, weak scaling (twice the work per step), and the gathered buffer holds both ranks' rows."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert "all-gather" in d["config"]["sharding"] and d["cpu_baseline"] is None  # the CPU baseline is an N = 1 item
    g = d["gather_check"]
    assert g["ranks_in_buffer"] == 2 and g["rows_per_rank"] == 65536 and g["distinct_batches"] is True
    # value = the samples BOTH ranks rendered per step / the slowest rank's time
    assert abs(d["ms_per_step"] * 1e-3 * d["value"] - 2 * 65536 * 192) / (2 * 65536 * 192) < 1e-6
    assert d["roofline"]["kernel"].startswith("render_")


@pytest.mark.gpu
def test_bench_plain_invocation_launches_its_own_ranks():
    """``python3 bench.py --gpus 2`` with no launcher around it (WORLD_SIZE unset): bench.py starts the two ranks itself as
    fresh child processes under ``torch.distributed.run`` before it touches the GPU, relays rank 0's single JSON line and
    the children's exit code (rehearsal environment as above: both ranks on this box's one GPU over gloo)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(BENCH_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["gather_check"]["ranks_in_buffer"] == 2
    assert "launching" in p.stderr


def test_bench_self_launch_relays_the_exit_code_without_touching_the_gpu(monkeypatch):
    """CPU tier: the parent of a plain ``bench.py --gpus 2`` never imports a GPU runtime before it spawns its ranks, passes
    its own arguments through and returns the launcher's exit code (here the ranks fail at once: this tier has no GPU)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():  # before anything is started: on a GPU box two real ranks would come up and be torn down
        pytest.skip("CPU-tier check of the failure path")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode != 0
    assert "launching -m torch.distributed.run --nnodes=1 --nproc-per-node=2" in p.stderr
    assert "needs an MI355X" in p.stderr  # every rank refused for the same reason the N = 1 run does
