"""Multi-GPU sharding of the env axis: one process per GPU, contiguous env blocks, no
collective during the rollout, one all-gather for the returned trajectories - as an RCCL collective over xGMI
(gloo on CPU tensors), or as peer-to-peer copies on the copy engines (PeerCopyGather), which unlike a collective's
kernels can proceed while the next rollout occupies every CU."""
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


def _open_shared_cpu(reduced_storage, shape, dtype):
    """A tensor over another process's shared-memory storage (PeerCopyGather's CPU rehearsal)."""
    fn, args = reduced_storage
    return torch.empty(0, dtype=dtype).set_(fn(*args), 0, shape)


class PeerCopyGather:
    """Equal-block all-gather of trajectories by peer-to-peer copies instead of a collective kernel (r03).

    Why: the rollout kernel leaves nothing on a CU for another kernel to live on - two waves per SIMD hold 464 of its 512 vector
    registers and the workgroup all but 0.9 KB of the LDS - so a collective's kernels are never resident next to it: whichever is
    launched first runs first (profiles/r03_coresidency.txt: 33.9 ms + 20.9 ms = 55.4 ms either way).  A device-to-device copy
    between two GPUs is carried out by the copy (SDMA) engines over xGMI and needs no compute unit, so it does run under the
    next rollout.

    Every rank owns `nbuf` gather buffers [world, *block]; their IPC handles are exchanged once (torch's CUDA IPC, pickled through
    a gloo group), and a gather is `world` asynchronous copies of the rank's block into slot `rank` of every peer's buffer on a
    side stream, ordered after the kernels already queued on the current stream.  `wait()` synchronises the side stream and meets
    the other ranks at a CPU (gloo) barrier: after it every rank's copies have landed, the buffer is complete.  Buffers rotate,
    so at most `nbuf` gathers may be in flight.  OWNERSHIP (r04): the tensor `wait()` returns is a view of one of the rotating
    buffers, which the PEERS overwrite `nbuf` gathers later; it belongs to the caller until the handle's `release()` (or
    `wait(copy=True)`, which returns a private copy and releases at once), and `start()` refuses - on every rank, through one gloo
    reduction, so nobody hangs - to reuse a buffer that some rank still holds.  Raises (at construction) if the IPC exchange or the
    probe transfer fails: the caller falls back to the RCCL all-gather."""

    def __init__(self, block_shape, dtype, device, nbuf: int = 2):
        from torch.multiprocessing.reductions import reduce_tensor
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device, self.nbuf, self.k = device, nbuf, 0
        # CPU rehearsal (r05; bench.py --dry-run, tests/test_dist_cpu.py at world size 8): the same exchange, ownership protocol and
        # failure handling with the buffers in shared host memory (torch's shared-memory handles instead of CUDA IPC) and no streams
        self.on_cpu = torch.device(device).type == "cpu"
        self.held = [False] * nbuf  # buffer b handed out by wait() and not yet released by this rank
        self.cpu_group = dist.new_group(backend="gloo")
        self.block_shape = tuple(block_shape)

        # Every phase ends in a collective that all ranks reach whether or not their own part failed: a rank that raised early
        # would otherwise leave its peers waiting in the next exchange.
        def all_ok(err, what):
            msgs = [None] * self.world
            dist.all_gather_object(msgs, None if err is None else f"rank {self.rank}: {type(err).__name__}: {err}", group=self.cpu_group)
            bad = [m for m in msgs if m is not None]
            if bad:
                raise RuntimeError(f"peer-copy gather, {what}: " + "; ".join(bad)[:400])

        err, mine = None, None
        try:
            self.bufs = [torch.empty((self.world,) + self.block_shape, dtype=dtype, device=device) for _ in range(nbuf)]
            self.probe = torch.full((self.world, 4), -1.0, device=device)
            if self.on_cpu:
                import torch.multiprocessing as tmp_
                tmp_.set_sharing_strategy("file_system")  # handles that are names in /dev/shm: they survive a plain pickle (gloo's object exchange)
                for t_ in self.bufs + [self.probe]:
                    t_.share_memory_()
            if self.on_cpu:
                # (reduce_tensor of a CPU tensor carries the storage OBJECT, which a plain pickle copies by value: export the storage's
                #  shared-memory name instead and rebuild the tensor over it on the other side)
                from torch.multiprocessing.reductions import reduce_storage
                mine = [(_open_shared_cpu, (reduce_storage(t.untyped_storage()), tuple(t.shape), t.dtype)) for t in self.bufs + [self.probe]]
            else:
                mine = [reduce_tensor(t) for t in self.bufs + [self.probe]]
        except Exception as e:
            err = e
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.cpu_group)
        all_ok(err, "allocating and exporting the buffers")
        try:
            if os.environ.get("AMX_P2P_FAIL_RANK") == str(self.rank):  # test hook: one rank fails, nobody may hang
                raise OSError("injected failure (AMX_P2P_FAIL_RANK)")
            self.peer = []  # peer[p] = [buffers of rank p ..., probe of rank p] as tensors of this process
            for p in range(self.world):
                self.peer.append(self.bufs + [self.probe] if p == self.rank else [fn(*a) for fn, a in everyone[p]])
        except Exception as e:
            err = e
        all_ok(err, "opening the peers' buffers")
        # probe: every rank writes its number into its slot of every peer's probe; all slots must arrive
        ok = 0
        try:
            self.stream = None if self.on_cpu else torch.cuda.Stream(device)
            src = torch.full((4,), float(self.rank), device=device)
            if self.on_cpu:
                for p in range(self.world):
                    self.peer[p][nbuf][self.rank].copy_(src)
            else:
                with torch.cuda.stream(self.stream):
                    for p in range(self.world):
                        self.peer[p][nbuf][self.rank].copy_(src, non_blocking=True)
                self.stream.synchronize()
        except Exception as e:
            err = e
        dist.barrier(group=self.cpu_group)
        try:
            if err is None:
                want = torch.arange(self.world, dtype=torch.float32, device=device)[:, None].expand(self.world, 4)
                ok = int(torch.equal(self.probe, want))
                if not ok:
                    err = RuntimeError("the probe transfer did not arrive")
        except Exception as e:
            err = e
        all_ok(err, "probe transfer")

    def start(self, xs_local: torch.Tensor) -> "PendingPeerCopy":
        """Starts the gather of `xs_local` (ordered after the current stream's queued work); returns the handle to wait on."""
        assert tuple(xs_local.shape) == self.block_shape, (tuple(xs_local.shape), self.block_shape)
        b = self.k % self.nbuf
        if self.k >= self.nbuf:
            # the peers are about to overwrite buffer b on EVERY rank: all of them must have let go of it.  One small CPU reduction
            # (the ranks arrive here in step anyway: the previous wait() ended in a barrier); every rank sees the same verdict.
            free = torch.tensor([0 if self.held[b] else 1], dtype=torch.int32)
            dist.all_reduce(free, op=dist.ReduceOp.MIN, group=self.cpu_group)
            if int(free.item()) == 0:
                raise RuntimeError(f"peer-copy gather: buffer {b} is still held by a rank (a tensor returned by wait() {self.nbuf} gathers ago "
                                   "was not released): call release() on that handle, or use wait(copy=True)")
        self.k += 1
        xs_local = xs_local.contiguous()
        if self.on_cpu:
            for d in range(self.world):
                self.peer[(self.rank + d) % self.world][b][self.rank].copy_(xs_local)
            return PendingPeerCopy(self, b, None)
        ready = torch.cuda.Event()
        ready.record()
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            for d in range(self.world):
                p = (self.rank + d) % self.world  # every rank starts with a different peer: the links are used evenly
                self.peer[p][b][self.rank].copy_(xs_local, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)  # this gather's copies only: wait() must not wait for a later gather queued behind a later rollout
        xs_local.record_stream(self.stream)
        return PendingPeerCopy(self, b, done)


    def close(self) -> None:
        """Drops this process's mappings of the peers' buffers, then meets the peers: a rank must not free buffers its peers still map."""
        if self.stream is not None:
            self.stream.synchronize()
        self.peer = None
        dist.barrier(group=self.cpu_group)


class PendingPeerCopy:
    __slots__ = ("owner", "b", "done")

    def __init__(self, owner, b, done):
        self.owner, self.b, self.done = owner, b, done

    def wait(self, copy: bool = False) -> torch.Tensor:
        """Blocks the host until every rank's block has landed in this rank's buffer; returns it as [world * nenv_local, ...].
        The result is a view of the rotating buffer and stays valid until `release()`; with copy=True it is a private copy and
        the buffer is released here."""
        o = self.owner
        if self.done is not None:
            self.done.synchronize()         # my outgoing copies of THIS gather are done
        dist.barrier(group=o.cpu_group)     # ... and so are everybody else's (a CPU barrier: no kernel behind the rollout)
        buf = o.bufs[self.b].view((o.world * o.block_shape[0],) + o.block_shape[1:])
        if copy:
            out = buf.clone()
            if not o.on_cpu:
                torch.cuda.current_stream(o.device).synchronize()  # the copy has read the buffer before a peer may write it again
            return out
        o.held[self.b] = True
        return buf

    def release(self) -> None:
        """The caller is done with the tensor wait() returned: its buffer may be overwritten by the peers' next gather into it.
        Work the caller queued on that tensor must have finished (the peers write from other devices, unordered with this
        device's streams): the current stream is synchronised here."""
        o = self.owner
        if o.held[self.b]:
            if not o.on_cpu:
                torch.cuda.current_stream(o.device).synchronize()
            o.held[self.b] = False


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
