"""Mechanics of ambersim_amd.dist.PeerCopyGather with two processes on ONE GPU (the 1-GPU box: both ranks use cuda:0, process group
over gloo - RCCL refuses two ranks on one device): IPC exchange, probe, two gathers in flight, contents and order.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dbg/p2p_gather_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from ambersim_amd.dist import PeerCopyGather
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("AMX_P2P_FAIL_RANK") is not None:  # one rank fails during the set-up: every rank must get the error, none may hang
    try:
        PeerCopyGather((64, 11, 7), torch.float32, dev)
        raise SystemExit("the injected failure was not reported")
    except RuntimeError as e:
        assert "injected failure" in str(e), e
    dist.barrier()
    if rank == 0: print("peer-copy gather: a failing rank is reported on every rank")
    dist.destroy_process_group()
    raise SystemExit(0)
g = PeerCopyGather((64, 11, 7), torch.float32, dev)
pend = []
for k in range(5):
    xs = torch.full((64, 11, 7), float(10 * k + rank), device=dev)
    pend.append((k, g.start(xs)))
    if len(pend) > 1:
        kk, pg = pend.pop(0)
        out = pg.wait()
        want = torch.cat([torch.full((64, 11, 7), float(10 * kk + r)) for r in range(world)]).to(dev)
        assert torch.equal(out, want), (rank, kk)
        if kk == 1:  # holding on to a result: the gather that would overwrite its buffer is refused, on every rank
            pend_k, pend_h = pend[0]
            try:
                g.start(xs)
                raise SystemExit("start() reused a buffer that is still held")
            except RuntimeError as e:
                assert "still held" in str(e), e
        pg.release()
for kk, pg in pend:
    out = pg.wait()
    want = torch.cat([torch.full((64, 11, 7), float(10 * kk + r)) for r in range(world)]).to(dev)
    assert torch.equal(out, want), (rank, kk)
    pg.release()
g.close()
dist.barrier()
if rank == 0: print("peer-copy gather: 5 gathers, two in flight, contents and order correct on", world, "ranks")
dist.destroy_process_group()
