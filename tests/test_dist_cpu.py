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
        pend = [gather_trajectories_async(xs_local + k) for k in range(2)]
        for k, (out, work) in enumerate(pend):
            work.wait()
            assert torch.equal(out, full + k)
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
