"""Multi-GPU sharding of the env axis: one process per GPU, contiguous env blocks, no
collective during the rollout, one all-gather (RCCL over xGMI on GPUs, gloo on CPU tensors) for the
returned trajectories."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialises torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank); a single process needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(nenv_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the env axis owned by `rank`; blocks differ by at most one env."""
    base, rem = divmod(nenv_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_trajectories(xs_local: torch.Tensor, nenv_total: int | None = None, equal_blocks: bool = False) -> torch.Tensor:
    """All-gathers per-rank trajectory blocks [nenv_local, ...] into [nenv_total, ...] on every rank.
    Equal blocks use one all_gather_into_tensor (a single RCCL all-gather); ragged blocks are padded.
    `equal_blocks=True`: the caller guarantees every rank passes the same block shape (shard_range of a multiple of the
    world size, bench.py's layout) - no size exchange and no host synchronisation, just the one collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return xs_local
    world = dist.get_world_size()
    if equal_blocks:
        out = torch.empty((world * xs_local.shape[0],) + tuple(xs_local.shape[1:]), dtype=xs_local.dtype, device=xs_local.device)
        dist.all_gather_into_tensor(out, xs_local.contiguous())
        return out
    n_local = torch.tensor([xs_local.shape[0]], dtype=torch.int64, device=xs_local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = torch.empty((world * sizes[0],) + tuple(xs_local.shape[1:]), dtype=xs_local.dtype, device=xs_local.device)
        dist.all_gather_into_tensor(out, xs_local.contiguous())
        return out
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(xs_local.shape[1:]), dtype=xs_local.dtype, device=xs_local.device)
    pad[: xs_local.shape[0]] = xs_local
    out = torch.empty((world * nmax,) + tuple(xs_local.shape[1:]), dtype=xs_local.dtype, device=xs_local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * nmax: r * nmax + sizes[r]] for r in range(world)], dim=0)


class PendingGather:
    """An all-gather in flight.  Holds the gathered tensor, the work handle AND the rank's own block: the collective reads
    `src` on the communication stream after the caller has moved on, so the block must stay allocated (and must not be
    handed back to the caching allocator for reuse by the compute stream) until `wait()` - the caller may rebind its own
    name for it right away (bench.py does, on the next rollout)."""
    __slots__ = ("out", "work", "src")

    def __init__(self, out, work, src):
        self.out, self.work, self.src = out, work, src

    def wait(self) -> torch.Tensor:
        """Orders the current stream after the collective, releases the source block, returns the gathered tensor."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        self.src = None
        return self.out

    def __iter__(self):  # (out, work) unpacking, as the first version of this function returned
        return iter((self.out, self))


def gather_trajectories_async(xs_local: torch.Tensor) -> PendingGather:
    """Equal-block all-gather started asynchronously.  The collective is ordered after the kernels already queued on the
    current stream (the rollout that produced `xs_local`) and runs on the communication stream, so the next rollout
    overlaps with it; `PendingGather.wait()` orders the current stream after it and returns the gathered tensor.
    Every rank must pass the same block shape (bench.py's weak-scaling layout)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return PendingGather(xs_local, None, None)
    world = dist.get_world_size()
    xs_local = xs_local.contiguous()
    out = torch.empty((world * xs_local.shape[0],) + tuple(xs_local.shape[1:]), dtype=xs_local.dtype, device=xs_local.device)
    work = dist.all_gather_into_tensor(out, xs_local, async_op=True)
    return PendingGather(out, work, xs_local)


def backend_report() -> dict:
    """What the communication backend says about itself (bench.py's scaling_report): backend name, world size as the process
    group reports it, RCCL's version when the backend is nccl (= RCCL on ROCm)."""
    rep = {"backend": None, "world_size_reported": 1}
    if dist.is_available() and dist.is_initialized():
        rep["backend"] = dist.get_backend()
        rep["world_size_reported"] = dist.get_world_size()
        if rep["backend"] == "nccl":
            try:
                rep["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # a report field must not take the run down
                rep["rccl_version"] = f"unavailable ({type(e).__name__})"
    return rep
