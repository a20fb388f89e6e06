"""N>1 path on CPU: two gloo ranks shard the env axis and all-gather trajectories."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nenv_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ambersim_amd.dist import gather_trajectories, init_from_env, shard_range
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_range(nenv_total, rank, world)
    # each env's "trajectory" encodes its global env id, so the gathered order is checkable
    xs_local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1) * torch.ones((1, 4, 3))
    full = gather_trajectories(xs_local)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    if nenv_total % world == 0:  # the overlapped path of bench.py: two equal-block gathers in flight, then waited
        from ambersim_amd.dist import gather_trajectories_async
        pend = [gather_trajectories_async(xs_local + k) for k in range(2)]  # the temporaries xs_local + k live on in the handles
        for k, pg in enumerate(pend):
            assert pg.src is not None  # the rank's own block is held until the collective has been waited for
            out = pg.wait()
            assert pg.src is None and torch.equal(out, full + k)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nenv_total", [8, 7])
def test_gloo_shard_and_gather(tmp_path, nenv_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), nenv_total, str(tmp_path)), nprocs=world, join=True)
    expect = np.arange(nenv_total, dtype=np.float32).reshape(-1, 1, 1) * np.ones((1, 4, 3), dtype=np.float32)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert got.shape == (nenv_total, 4, 3) and np.array_equal(got, expect)


def test_equal_blocks_gather_skips_the_size_exchange(tmp_path):
    world = 2
    mp.spawn(_worker_equal, args=(world, _free_port()), nprocs=world, join=True)


def _worker_equal(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ambersim_amd.dist import gather_trajectories, init_from_env
    init_from_env("gloo")
    xs = torch.full((3, 2, 5), float(rank))
    full = gather_trajectories(xs, equal_blocks=True)
    assert full.shape == (6, 2, 5) and torch.equal(full[:, 0, 0], torch.tensor([0., 0., 0., 1., 1., 1.]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("extra", [[], ["--no-overlap"], ["--gather", "none"]])
def test_bench_self_launch_dry_run(extra):
    """`python bench.py --gpus 2` from a plain shell (WORLD_SIZE unset): the parent starts torch.distributed.run as a child
    before importing torch, both ranks run, rank 0 prints ONE JSON line, the exit code is the child's.  --dry-run swaps
    the kernel for zero trajectories on the CPU over gloo, everything else (sharding, barriers, gathers in rank order,
    overlap bookkeeping, scaling report) is the real code path."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--nenv", "6",
           "--horizon", "4"] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert "biped_stand.xml" in out["config"]["workload"] and out["config"]["nenv_per_gpu"] == 6  # one workload at every N
    assert out["scaling_report"]["backend"] == "gloo" and out["scaling_report"]["world_size_reported"] == 2
    want = {"": "traj+overlapped", "--no-overlap": "traj", "--gather": "none"}[extra[0] if extra else ""]
    assert out["config"]["gather"] == want
    rep = out["scaling_report"]
    assert rep["gather_alone_ms"] > 0 and rep["rollout_only_env_steps_per_s"] > 0


def test_bench_self_launch_propagates_failure():
    """A rank that dies makes `python bench.py --gpus N` exit non-zero (here: --gpus disagrees with the forced model path)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--model", "models/does_not_exist.xml", "--nenv", "2", "--horizon", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_dry_run_four_ranks():
    """The same launcher with four ranks (rehearsal of a wider node on the CPU: gloo, block order 0..3 in the gathered tensor)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--dry-run", "--nenv", "3", "--horizon", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 4 and out["config"]["gather"] == "traj+overlapped" and out["scaling_report"]["gathered_GB_per_rank"] > 0


def test_bench_workload_is_the_same_at_every_n():
    """The headline is north_star's target (the ~30-DoF biped, 4096 envs per GPU x 1000 steps) at N = 1 and at N > 1: a 1 -> 8
    scaling curve compares one workload with itself."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    loads = []
    for n in (1, 2):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--dry-run", "--no-scaling-report"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
        assert out["n_gpus"] == n and out["config"]["nenv_per_gpu"] == 4096 and out["config"]["horizon"] == 1000
        loads.append(out["config"]["workload"])
    assert loads[0] == loads[1] and "biped_stand.xml" in loads[0]


def _worker_peer_copy(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ambersim_amd.dist import PeerCopyGather, init_from_env
    init_from_env("gloo")
    nenv, T, nx = 3, 2, 5
    pg = PeerCopyGather((nenv, T, nx), torch.float32, torch.device("cpu"), nbuf=2)
    want = lambda k: torch.arange(world, dtype=torch.float32).repeat_interleave(nenv)[:, None, None].expand(world * nenv, T, nx) + 100.0 * k
    held = []
    for k in range(2):  # two gathers fill both rotating buffers; their results are kept (not released)
        h = pg.start(torch.full((nenv, T, nx), float(rank) + 100.0 * k))
        full = h.wait()
        assert full.shape == (world * nenv, T, nx) and torch.equal(full, want(k)), (rank, k)  # blocks in rank order 0 .. world - 1
        held.append((h, full))
    # a third gather would let the peers overwrite buffer 0, which THIS job still holds on every rank: every rank is refused, nobody hangs
    try:
        pg.start(torch.zeros(nenv, T, nx))
        raise AssertionError("a held buffer was reused")
    except RuntimeError as e:
        assert "still held" in str(e)
    pg.k -= 0  # (the refused start() did not advance the rotation)
    # one rank alone keeping its view is enough to refuse everybody
    for h, _ in held[1:]:
        h.release()
    if rank != world - 1:
        held[0][0].release()
    try:
        pg.start(torch.zeros(nenv, T, nx))
        raise AssertionError("a buffer held by the last rank was reused")
    except RuntimeError as e:
        assert "still held" in str(e)
    held[0][0].release()
    assert torch.equal(held[0][1], want(0))  # untouched while it was held
    h = pg.start(torch.full((nenv, T, nx), float(rank) + 200.0))
    full = h.wait(copy=True)  # a private copy: released at once
    assert torch.equal(full, want(2))
    h2 = pg.start(torch.full((nenv, T, nx), float(rank) + 300.0))
    assert torch.equal(h2.wait(copy=True), want(3))
    pg.close()
    torch.save(full, os.path.join(out_dir, f"peer{rank}.pt"))


def test_peer_copy_gather_protocol_at_world_size_eight(tmp_path):
    """The peer-copy gather (ambersim_amd.dist.PeerCopyGather: what `bench.py --gather p2p` runs over xGMI) rehearsed at the width of
    the node the driver benches on - eight ranks, buffers in shared host memory instead of IPC-mapped device memory: blocks land in rank
    order 0 .. 7 on every rank, a buffer still held by ANY rank refuses the gather that would overwrite it on EVERY rank (no hang), a
    released one is reused, wait(copy=True) releases at once.  No hardware run is claimed: the copy engines and xGMI are not involved."""
    world = 8
    mp.spawn(_worker_peer_copy, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = torch.load(tmp_path / "peer0.pt")
    for r in range(1, world):
        assert torch.equal(torch.load(tmp_path / f"peer{r}.pt"), ref)


def test_bench_self_launch_dry_run_eight_ranks():
    """`python bench.py --gpus 8 --dry-run` (the driver's widest launch, on the CPU over gloo): eight ranks, one JSON line, the collective
    as the headline's gather and the peer-copy gather measured beside it (shared-memory transport), per-rank blocks in order."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--nenv", "3", "--horizon", "2"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["config"]["gather"] == "traj+overlapped" and out["scaling_report"]["gathered_GB_per_rank"] > 0
    assert out["scaling_report"].get("block_order_ok") is True
    assert "peer_copy_alone_ms" not in out["scaling_report"]  # the default run (the driver's) depends on the collective alone
    p = subprocess.run(cmd + ["--p2p-report"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["gather"] == "traj+overlapped" and out["scaling_report"]["peer_copy_alone_ms"] > 0  # the alternative beside it, on request
    p = subprocess.run(cmd + ["--gather", "p2p"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 8 and "peer-to-peer" in out["config"]["gather"] and out["scaling_report"].get("block_order_ok") is True
