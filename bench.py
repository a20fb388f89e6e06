#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused rollout (BASELINE.json metric).

One "step" = one amx_rollout launch over one batch: NENV envs x HORIZON physics steps of the
build-authored quadruped (4096 envs, random ctrl, 1000-step rollout), inputs
resident in HBM.  With --gpus N every rank runs its own NENV envs (weak scaling, no collective in the
data path) and the returned trajectories are all-gathered over RCCL once per step, overlapped with the next rollout.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured copy


def make_inputs(mj, nenv, horizon, seed_state, seed_ctrl, device):
    """Standing keyframe + N(0, 0.01^2) joint-velocity noise (seed 1); ctrl ~ U(ctrlrange) i.i.d. (seed 2)."""
    import torch
    q0 = mj.key_qpos[0] if mj.nkey else mj.qpos0
    rng = np.random.default_rng(seed_state)
    x0 = np.tile(np.concatenate([q0, np.zeros(mj.nv)]), (nenv, 1)).astype(np.float32)
    x0[:, mj.nq:] += (0.01 * rng.standard_normal((nenv, mj.nv))).astype(np.float32)
    g = torch.Generator(device=device).manual_seed(seed_ctrl)
    lo = torch.tensor(mj.actuator_ctrlrange[:, 0], dtype=torch.float32, device=device)
    hi = torch.tensor(mj.actuator_ctrlrange[:, 1], dtype=torch.float32, device=device)
    us = lo + (hi - lo) * torch.rand((nenv, horizon, mj.nu), generator=g, device=device, dtype=torch.float32)
    return torch.tensor(x0, device=device), us


def pmc_traffic(default_workload):
    """HBM bytes per launch of the rollout kernel from the committed rocprofv3 PMC passes (profiles/, collected with
    tools/pmc_collect.sh: FETCH_SIZE and WRITE_SIZE in separate passes, KiB).  gfx950 correction from the guide: FETCH_SIZE
    tallies 128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is exact.  Only valid for the default workload."""
    path = os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_amx_rollout.csv")
    if not default_workload or not os.path.exists(path):
        return None
    vals = {}
    for line in open(path):
        p = line.strip().split(",")
        if len(p) >= 2 and p[0] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[p[0]] = float(p[1])
    if len(vals) != 2:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def cpu_baseline(mj, x0_np, us_np, target_s=12.0):
    """Times the CPU oracle (oracle/amx_oracle.c, fp32, OpenMP over envs, all host threads) on a bounded sample of the
    same workload: whole-horizon rollouts of the first envs, repeated until about `target_s` seconds of wall time."""
    from oracle.oracle import Oracle
    o = Oracle(mj, np.float32)
    cores = o.max_threads()
    nenv = min(us_np.shape[0], 8 * cores)
    horizon = us_np.shape[1]
    o.rollout(x0_np[:nenv], us_np[:nenv, : min(20, horizon)], nthreads=cores)  # touch code and threads
    done, t0 = 0, time.perf_counter()
    while True:
        o.rollout(x0_np[:nenv], us_np[:nenv], nthreads=cores)
        done += nenv * horizon
        dt = time.perf_counter() - t0
        if dt >= target_s:
            break
    return {"value": done / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{done // (nenv * horizon)} x ({nenv} envs x {horizon} steps) of the same workload, fp32 C oracle "
                      f"(oracle/amx_oracle.c), OpenMP over envs, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nenv", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--model", default="models/quadruped/quadruped.xml")
    ap.add_argument("--gather", choices=["traj", "none"], default="traj", help="N>1: all-gather returned trajectories")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: wait for each all-gather before the next rollout")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ambersim_amd import mjx
    from ambersim_amd.dist import gather_trajectories, gather_trajectories_async, init_from_env
    from ambersim_amd.utils.io_utils import load_mj_model_from_file

    rank, world, local = init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    mj = load_mj_model_from_file(args.model)
    model = mjx.put_model(mj, dev)
    x0, us = make_inputs(mj, args.nenv, args.horizon, 1 + 1000 * rank, 2 + 1000 * rank, dev)
    nx = mj.nq + mj.nv

    overlap = world > 1 and args.gather == "traj" and not args.no_overlap

    def run_steps(nsteps, events=None):
        """`nsteps` rollouts; with N > 1 each trajectory block is all-gathered.  The gather of step k runs on the
        communication stream while rollout k+1 runs on the compute stream (at most two gathers in flight); all of them
        are complete when this returns (the caller synchronises)."""
        pending, xs = [], None
        for k in range(nsteps):
            if events is not None:
                events[k][0].record()  # HIP events on the launch stream (torch's current stream is the one handed to amx_rollout)
            xs = mjx.rollout(model, x0, us)
            if events is not None:
                events[k][1].record()
            if world > 1 and args.gather == "traj":
                if overlap:
                    pending.append(gather_trajectories_async(xs))
                    if len(pending) > 2:
                        pending.pop(0)[1].wait()
                else:
                    xs = gather_trajectories(xs)
        for out, work in pending:
            work.wait()
            xs = out
        return xs

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    xs = run_steps(args.steps, ev)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    finite = bool(torch.isfinite(xs).all().item())

    if rank == 0:
        env_steps = args.nenv * args.horizon * args.steps * world
        bytes_per_launch = 4.0 * (mj.nu + nx) * args.nenv * args.horizon  # ctrl row in, state row out per env-step
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec, 4096 envs x 1000-step rollout",
            "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{os.path.basename(args.model)} (build-authored, BASELINE.md C2), {args.nenv} envs/GPU x "
                                   f"{args.horizon}-step fused rollout, random ctrl ~U(ctrlrange), Newton {mj.opt.iterations} it / "
                                   f"{mj.opt.ls_iterations} ls, dt={mj.opt.timestep}",
                       "nenv_per_gpu": args.nenv, "horizon": args.horizon, "nq": mj.nq, "nv": mj.nv, "nu": mj.nu,
                       "gather": (args.gather + ("+overlapped" if overlap else "")) if world > 1 else "none", "outputs_finite": finite},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.nenv == 4096 and args.horizon == 1000 and args.model.endswith("quadruped.xml")),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_rocprofv3_pmc_amx_rollout.csv)",
                         "kernel": "amx_rollout_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mj, x0.cpu().numpy(), us.cpu().numpy())
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
