"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on MI355X; "gloo" in the CPU tests).

Rays are independent, so the path shards with no collective inside the render.  The reference's only collective
call site is DDP around the model (``crop_nerf/fruit_nerf/fruit_pipeline.py:119-121``); for render / export the
equivalent is:

* ``render_rays_sharded``  -- contiguous ray ranges per rank, ONE all-gather of the fixed-size per-ray outputs
  (rgb 3 + accumulation 1 + depth 1 + semantics 1 = 24 B/ray);
* ``all_gather_points``    -- variable-length exporter output: all-gather of the counts, then one padded all-gather
  of the rows (xyz + colour), trimmed on arrival;
* ``shard_jobs``           -- projection jobs (camera x sub-cluster) dealt round-robin, no communication at all.

xGMI is point to point (7 links per GPU); these payloads are 1-2 MB per rank, i.e. latency-bound single-shot
collectives -- no bucketing is needed and none is done.
"""

from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

PACK_KEYS = (("rgb", 3), ("accumulation", 1), ("depth", 1), ("semantics", 1))


def init_from_env() -> Tuple[int, int, str]:
    """What the CLIs call first: under ``torch.distributed.run`` (WORLD_SIZE > 1) join the job -- one rank per GPU over
    RCCL, device = LOCAL_RANK; ``CROPNERF_REHEARSE_ON_ONE_GPU=1`` puts every rank on cuda:0 over gloo (tests) -- and
    return (rank, world size, device string).  A plain ``python script.py`` returns (0, 1, "cuda")."""
    import os

    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return 0, 1, "cuda"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        if os.environ.get("CROPNERF_REHEARSE_ON_ONE_GPU") == "1":
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            return dist.get_rank(), ws, "cuda:0"
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return dist.get_rank(), ws, f"cuda:{torch.cuda.current_device()}"


def world(group=None) -> Tuple[int, int]:
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_range(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous balanced split of [0, n): the first n % world ranks get one extra row."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_jobs(jobs: Sequence, rank: int, world_size: int) -> List:
    return list(jobs[rank::world_size])


def _gather_into(out: Tensor, src: Tensor, group=None) -> None:
    """``all_gather_into_tensor``; device tensors over gloo (rehearsals on a one-GPU box, CPU tests) go through the host."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, src.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, src, group=group)


def _all_gather_rows(local: Tensor, counts: List[int], group=None) -> Tensor:
    """All-gather of [n_r, k] row blocks of different lengths: pad to the longest, gather once, trim."""
    rank, ws = world(group)
    mx = max(counts) if counts else 0
    k = local.shape[1]
    padded = torch.zeros(mx, k, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty(ws * mx, k, dtype=local.dtype, device=local.device)
    _gather_into(out, padded, group)
    out = out.view(ws, mx, k)
    return torch.cat([out[r, : counts[r]] for r in range(ws)], dim=0)


def render_rays_sharded(render_fn: Callable[[int, int], Dict[str, Tensor]], num_rays: int, group=None
                        ) -> Dict[str, Tensor]:
    """``render_fn(lo, hi)`` renders rays [lo, hi) on this rank and returns at least the PACK_KEYS tensors
    ([n,3],[n,1],[n,1],[n,1]).  Returns the full-size tensors on every rank."""
    rank, ws = world(group)
    lo, hi = shard_range(num_rays, rank, ws)
    out = render_fn(lo, hi)
    if ws == 1:
        return {k: out[k] for k, _ in PACK_KEYS}
    packed = torch.cat([out[k].reshape(hi - lo, w) for k, w in PACK_KEYS], dim=-1).contiguous()
    counts = [shard_range(num_rays, r, ws)[1] - shard_range(num_rays, r, ws)[0] for r in range(ws)]
    full = _all_gather_rows(packed, counts, group)
    res, c = {}, 0
    for k, w in PACK_KEYS:
        res[k] = full[:, c:c + w].contiguous()
        c += w
    return res


def all_gather_points(rows: Tensor, group=None) -> Tensor:
    """Variable-length all-gather of exporter rows ([n_r, k], e.g. xyz+rgb+prob): counts first, then payload."""
    rank, ws = world(group)
    if ws == 1:
        return rows
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    all_n = torch.empty(ws, dtype=torch.int64, device=rows.device)
    _gather_into(all_n, n, group)
    return _all_gather_rows(rows.contiguous(), [int(v) for v in all_n.tolist()], group)


def all_reduce_mean(value: Tensor, group=None) -> Tensor:
    """Per-view scalar metrics (loss / PSNR terms): sum-reduce then divide."""
    rank, ws = world(group)
    if ws == 1:
        return value
    v = value.clone()
    if v.is_cuda and dist.get_backend(group) == "gloo":  # rehearsal on a one-GPU box: gloo moves host memory
        h = v.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        v.copy_(h)
    else:
        dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v / ws


class _Finished:
    """Handle of an exchange that needed no communication."""

    def wait(self) -> None:
        return None


class _Then:
    """A collective's work handle plus what has to follow it on the host (gloo: the division, the copy back to the device)."""

    def __init__(self, work, after):
        self.work, self.after = work, after

    def wait(self) -> None:
        self.work.wait()
        if self.after is not None:
            self.after()
            self.after = None


def all_reduce_mean_(buf: Tensor, group=None, force: bool = False):
    """IN-PLACE mean of ``buf`` over the ranks, asynchronous: returns a handle whose ``wait()`` orders the caller after the
    result (RCCL: the collective runs on the process group's own stream behind everything enqueued on the current stream so
    far, and ``wait()`` is a stream-side dependency -- the host never blocks; gloo: a host wait).  No clone, no separate
    divide pass, no copy back on RCCL (``ReduceOp.AVG``); gloo has no AVG: SUM, then one in-place division -- the same
    arithmetic as ``all_reduce_mean``.  ``force``: run the collective also in a one-rank group (tests of the call pattern)."""
    rank, ws = world(group)
    if not (dist.is_available() and dist.is_initialized()) or (ws == 1 and not force):
        return _Finished()
    if dist.get_backend(group) == "gloo":
        if buf.is_cuda:  # rehearsal on a one-GPU box: gloo moves host memory
            host = buf.cpu()
            work = dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group, async_op=True)
            return _Then(work, lambda: buf.copy_(host.div_(ws)))
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
        return _Then(work, lambda: buf.div_(ws))
    return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=True)


class GroupedGradientExchange:
    """Data-parallel gradient averaging per OPTIMISER GROUP, overlapped with the rest of the backward pass -- the reference's
    DDP (``crop_nerf/fruit_nerf/fruit_pipeline.py:119-121``: bucketed all-reduce hooked into autograd) for a backward that is
    a fixed sequence of kernels: the ``fields`` slice of the flat gradient buffer (67 of its 78 MB) is final when the field
    backward has been enqueued, ~0.7 ms of proposal-backward / pose kernels before the optimiser needs it, so its all-reduce
    is issued right there (``start``) and each group is stepped behind its own reduce (``wait``).  One in-place collective
    per group; nothing is exchanged for a group that has no gradient this iteration."""

    def __init__(self, flat_grads: Tensor, group_range: Dict[str, Tuple[int, int]], group=None, force: bool = False):
        self.flat, self.ranges, self.group, self.force = flat_grads, dict(group_range), group, force
        self.pending: Dict[str, object] = {}
        self.started: List[str] = []  # order of the last iteration's collectives (every rank must agree on it)

    def begin_iteration(self) -> None:
        if self.pending:
            raise RuntimeError(f"gradient exchange of {sorted(self.pending)} was started and never waited for")
        self.started = []

    def start(self, name: str) -> None:
        lo, hi = self.ranges[name]
        if hi > lo and name not in self.pending:
            self.pending[name] = all_reduce_mean_(self.flat[lo:hi], self.group, self.force)
            self.started.append(name)

    def wait(self, name: str) -> None:
        h = self.pending.pop(name, None)
        if h is not None:
            h.wait()

    def wait_all(self) -> None:
        for name in list(self.pending):
            self.wait(name)
