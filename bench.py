#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the fused rollout (BASELINE.json metric).

One "step" = one amx_rollout launch over one batch: NENV envs x HORIZON physics steps, inputs resident in HBM.

The workload is the SAME at every N: north_star's target, BASELINE.md C3 per GPU / C5 sharded - the build-authored ~30-DoF
biped on its two convex-mesh feet on flat ground, 4096 envs per GPU, ctrl ~ U(ctrlrange), 1000-step fused rollout.  Since r04 the
headline model is models/biped/biped_stand.xml: biped.xml's bodies, feet and solver options with joint position servos (as C2's
quadruped has) whose ctrlrange is the standing pose +- 0.02 rad - the robots stay on their feet for the whole horizon, which
`config.activity` reports (base height, contacts in touch, constraint rows).  The two other readings of the same robot are
`secondary` lines with the same statistics: biped_rest.xml (random torques; eight extra contact spheres, the fallen robot rests on the
floor; 16 contact slots -> one environment per wavefront) and biped.xml (r01-r03's headline: random torques, feet-only contact -
after ~150 steps the robots hang below the floor from their feet; kept for continuity).
  N = 1   the JSON line also carries `secondary`: those two, C2 (the quadruped, 4096 x 1000, with its own roofline and CPU baseline),
          C4 (1024 x 64 forward-difference gradient through amx_fd_grad), the headline at other solver options (MuJoCo's defaults
          100 / 50 with the tolerance exit; the reference pendulum's 5 / 10) so that the headline is not tied to the model
          file's Newton 3 / 5, the headline model on the generic (not per-model) kernels, A8 (trajectory cost) and the FP32 figure
          (recorded flop census, profiles/rNN_flop_census.json, / 157.3 TFLOP/s; VALU-busy from the committed PMC pass);
          `cpu_baseline`; the MJX-CPU probe.
  N > 1   every rank runs its own 4096 envs (weak scaling, no collective in the data path) and the returned trajectories
          are all-gathered once per step by ONE RCCL all-gather (north_star's exchange; `--gather traj`, the default), started
          asynchronously behind each rollout.  The rollout's workgroups fill every CU, so a collective's KERNELS cannot run
          next to it (profiles/r03_coresidency.txt); `--gather p2p` does the same exchange as peer-to-peer copies on the copy
          engines (ambersim_amd.dist.PeerCopyGather), which can - a named alternative; `--p2p-report` times it beside the
          collective in `scaling_report` (rollout alone, the gather alone / serialised / overlapped, RCCL version, world size as
          the backend reports it).  The default run sets up RCCL only.

`python bench.py --gpus N` is self-contained: with WORLD_SIZE unset and N > 1 the parent - which has not imported torch
or touched the GPU - starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process, relays
its output and returns its exit code.  Under an external torchrun (WORLD_SIZE set) it is a plain rank.
`--dry-run` replaces the kernel by a zero-filled trajectory on the CPU (gloo): it exercises the launcher, the sharding and
the gather path without a GPU (tests/test_dist_cpu.py).

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6290 GB/s measured copy
FP32_PEAK_TFLOPS = 157.3  # MI355X FP32 vector peak (same guide)
QUADRUPED = "models/quadruped/quadruped.xml"
BIPED = "models/biped/biped.xml"              # legacy: random torques, feet-only contact (r01-r03 headline)
BIPED_STAND = "models/biped/biped_stand.xml"  # headline since r04: the same robot held standing by joint position servos
BIPED_REST = "models/biped/biped_rest.xml"    # random torques, eight extra contact spheres: falls and rests on the floor
HEADLINE = BIPED_STAND


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nenv", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--model", default=None, help=f"default: {HEADLINE} (C3 / C5) at every N")
    ap.add_argument("--gather", choices=["traj", "p2p", "none"], default="traj",
                    help="N>1: how the returned trajectories are all-gathered: traj (default) = one RCCL all-gather, north_star's exchange (its "
                         "kernels cannot be resident next to the rollout's); p2p = the same exchange as peer-to-peer copies on the copy engines "
                         "(they run under the next rollout; ambersim_amd.dist.PeerCopyGather) - reported beside the collective in "
                         "scaling_report either way; none = no gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the C2 / C4 / solver-option / FP32 secondary measurements")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: wait for each all-gather before the next rollout")
    ap.add_argument("--p2p-report", action="store_true",
                    help="N>1 with --gather traj: also set up the peer-copy gather and time it beside the collective in scaling_report.  Opt-in "
                         "since r05: cross-device IPC mappings have never run on hardware (no multi-GPU node in any round), and the default "
                         "run - the driver's scaling bench - should depend on RCCL alone")
    ap.add_argument("--no-scaling-report", action="store_true", help="N>1: skip the extra rollout-only / gather-only timings")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: zero trajectories on the CPU, gloo collectives")
    ap.add_argument("--master-port", type=int, default=0, help="self-launch only: rendezvous port (0 = pick a free one)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------- self-launch (no torch here)
def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """Parent of an N > 1 run started from a plain shell.  Nothing in this process has imported torch or initialised
    HIP, and it never replaces itself: the ranks are children of a torch.distributed.run child."""
    port = args.master_port or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# --------------------------------------------------------------------------------------------- inputs
def make_inputs(mj, nenv, horizon, seed_state, seed_ctrl, device):
    """Standing keyframe + N(0, 0.01^2) joint-velocity noise (seed 1); ctrl ~ U(ctrlrange) i.i.d. (seed 2)."""
    import numpy as np
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


def activity_stats(model, mj, xs, us, nenv_sample=512, every=25):
    """What the workload physically did - outside any timed region.  From the returned trajectories: the lowest and the mean final
    height of the floating base (qpos[2]) and the fraction of env-steps with the base below the floor; from one amx_forward launch
    over a sample of the visited states (the first `nenv_sample` envs at every `every`-th step, with the control applied there): mean
    contacts in touch (dist < 0) per env-step of the model's ncon slots, and mean rows in the constraint set (D > 0: violated
    limits + pyramid rows of contacts inside the margin) of its nefc_max."""
    import torch
    from ambersim_amd import mjx
    n, T = xs.shape[0], xs.shape[1] - 1
    out = {}
    if mj.nq >= 7 and mj.jnt_type[0] == 0:
        z = xs[:, :, 2]
        out.update({"base_z_min": float(z.min().item()), "base_z_final_mean": float(z[:, -1].mean().item()),
                    "frac_env_steps_base_below_floor": float((z < 0).float().mean().item())})
    ns = min(n, nenv_sample)
    ts = torch.arange(0, T, every, device=xs.device)
    st = xs[:ns][:, ts].reshape(-1, xs.shape[2])
    ct = us[:ns][:, ts].reshape(-1, us.shape[2])
    dist, nefc = mjx.constraint_activity(model, st[:, :mj.nq].contiguous(), st[:, mj.nq:].contiguous(), ct.contiguous())
    out.update({"contacts_touching_mean": float((dist < 0).sum(dim=1).float().mean().item()), "contact_slots": int(mj.ncon),
                "constraint_rows_mean": float(nefc.float().mean().item()), "constraint_rows_max": int(model.nefc_max),
                "sampled_states": int(st.shape[0])})
    return out


def pmc_record(model_path, nenv, horizon):
    """Recorded (NOT measured by this run) rocprofv3 PMC figures of the rollout kernel for the default workloads, from the
    newest committed profiles/rNN_rocprofv3_pmc_<model>.csv (tools/pmc_collect.sh: one counter per pass).  gfx950
    correction from the guide: FETCH_SIZE tallies 128-byte read requests at 64 bytes, so it is doubled; WRITE_SIZE is exact."""
    import glob
    if nenv != 4096 or horizon != 1000:
        return None
    tag = os.path.basename(model_path).split(".")[0]
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_rocprofv3_pmc_{tag}.csv")))
    if not cands and tag == "quadruped":
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_rocprofv3_pmc_amx_rollout.csv")))
    if not cands:
        return None
    vals = {}
    for line in open(cands[-1]):
        p = line.strip().split(",")
        if len(p) >= 2:
            try:
                vals[p[0]] = float(p[1])
            except ValueError:
                pass
    out = {"file": os.path.relpath(cands[-1], ROOT)}
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out["hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    if "valu_busy_frac" in vals:
        out["valu_busy_frac"] = vals["valu_busy_frac"]
    return out


def cpu_baseline(mj, x0_np, us_np, target_s):
    """Times the CPU oracle (oracle/amx_oracle.c, fp32, OpenMP over envs, all host threads) on a bounded sample of the
    same workload: whole-horizon rollouts of the first envs, repeated until about `target_s` seconds of wall time.  The
    timed library is built here, on the host that runs it, with -O3 -march=native (`make -C oracle native`); if that
    build fails the portable -O2 build is timed and the sample string says so."""
    from oracle.oracle import Oracle
    flags = "-O3 -march=native"
    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], check=True, capture_output=True, timeout=300)
        o = Oracle(mj, variant="native")
    except Exception as e:  # no compiler / read-only tree: the portable build still gives a baseline
        o = Oracle(mj, "float32")
        flags = f"-O2 (native build failed: {type(e).__name__})"
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
            "sample": f"{done // (nenv * horizon)} x ({nenv} envs x {horizon} steps) of the same workload, fp32 C restatement of the MJX algorithm "
                      f"(oracle/amx_oracle.c: dense M, dense Cholesky, scalar loops; gcc {flags}), OpenMP over envs, {dt:.1f} s"}


def probe_mjx_cpu():
    """is MJX-on-CPU (jax[cpu] + mujoco.mjx) available on this box?  Import probe only."""
    missing = []
    for mod in ("jax", "mujoco", "mujoco.mjx"):
        try:
            __import__(mod)
        except Exception as e:
            missing.append(f"{mod}: {type(e).__name__}")
    return "available (not timed here)" if not missing else "unavailable on this box (" + "; ".join(missing) + ") - cpu_baseline is the build's own CPU restatement"


def flop_record(model_path):
    """Recorded (NOT computed by this run) arithmetic operations per env-step of the restated algorithm for the shipped models:
    profiles/rNN_flop_census.json, written by tools/flop_census.py from the counting build of the CPU restatement."""
    import glob
    import json
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_flop_census.json")))
    if not cands:
        return None
    rec = json.load(open(cands[-1]))
    tag = os.path.basename(model_path).split(".")[0]
    if tag not in rec.get("flops_per_env_step", {}):
        return None
    return {"flops": float(rec["flops_per_env_step"][tag]), "file": os.path.relpath(cands[-1], ROOT), "how": rec.get("how", "")}


def roofline(mj, nenv, horizon, kern_ms, model_path):
    nx = mj.nq + mj.nv
    bytes_per_launch = 4.0 * (mj.nu + nx) * nenv * horizon  # ctrl row in, state row out per env-step
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    rec = pmc_record(model_path, nenv, horizon)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": rec.get("hbm_bytes_per_launch") if rec else None,
            "traffic_source": (f"RECORDED, not measured by this run: rocprofv3 PMC passes in {rec['file']} (2*FETCH_SIZE + WRITE_SIZE, KiB)" if rec else None),
            "kernel": "amx_rollout_kernel", "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
            "binding": "the named bound (north_star) is HBM; the counters say the kernel is VALU-issue / dependent-latency bound "
                       "(DESIGN.md §4): algorithmic intensity ~1e3 flop/B against a machine balance of ~25"}


# --------------------------------------------------------------------------------------------- one rank
def worker(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from ambersim_amd.dist import PeerCopyGather, backend_report, gather_trajectories, gather_trajectories_async, init_from_env
    from ambersim_amd.utils.io_utils import load_mj_model_from_file

    # AMX_BENCH_SHARE_DEVICE=1: every rank on cuda:0 with a gloo process group - the only way to run more than one rank on a
    # one-GPU box (RCCL refuses two ranks on one device); exercises the launcher and the peer-copy gather, not RCCL
    share = os.environ.get("AMX_BENCH_SHARE_DEVICE") == "1"
    rank, world, local = init_from_env("gloo" if (args.dry_run or share) else None)
    if share:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    sync = (lambda: None) if dry else torch.cuda.synchronize

    model_path = args.model or HEADLINE  # one workload at every N (north_star's target)
    mj = load_mj_model_from_file(model_path)
    nx = mj.nq + mj.nv
    if dry:
        model = None
        x0 = torch.zeros(args.nenv, nx)
        us = torch.zeros(args.nenv, args.horizon, mj.nu)
        rollout = lambda: torch.full((args.nenv, args.horizon + 1, nx), float(rank))
    else:
        from ambersim_amd import mjx
        model = mjx.put_model(mj, dev)
        x0, us = make_inputs(mj, args.nenv, args.horizon, 1 + 1000 * rank, 2 + 1000 * rank, dev)
        rollout = lambda: mjx.rollout(model, x0, us)

    red_dev = "cpu" if (dry or share) else dev  # small host-side reductions: gloo takes CPU tensors
    # the gather method: peer-to-peer copies (copy engines) or the RCCL collective (kernels)
    p2p, p2p_note = None, None
    # the alternative is measured beside the collective when asked for (--p2p-report); ranks sharing one device have no RCCL at all
    want_p2p = args.gather == "p2p" or (args.gather == "traj" and not args.no_scaling_report and (args.p2p_report or share))
    if world > 1 and want_p2p:  # (dry run: the same class on shared host memory - the protocol without the copy engines)
        try:
            p2p = PeerCopyGather((args.nenv, args.horizon + 1, nx), torch.float32, dev, nbuf=2)
        except Exception as e:  # IPC not available, peer access refused, probe lost: the collective still works
            if args.gather == "p2p":
                raise
            p2p_note = f"peer-copy set-up failed ({type(e).__name__}: {str(e)[:160]}): not measured"
        # every rank must take the same path
        ok = torch.tensor([1 if p2p is not None else 0], device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            p2p = None
    # the headline's exchange is the one asked for: RCCL unless `--gather p2p` (ranks sharing one device have no RCCL: peer copies)
    method = "none" if (world == 1 or args.gather == "none") else ("p2p" if (args.gather == "p2p" or share) and p2p is not None else "rccl")

    def run_steps(nsteps, mode, events=None, how=None):
        """`nsteps` rollouts.  mode "overlap": the gather of step k proceeds while rollout k+1 runs (at most two gathers in
        flight); "serial": each gather is waited for; "none": no gather.  how: "p2p" (peer copies) or "rccl" (the collective);
        default: the run's method.  Everything is complete when this returns (the caller synchronises)."""
        how = how or method
        pending, xs = [], None
        for k in range(nsteps):
            if events is not None:
                events[k][0].record()  # HIP events on the launch stream (torch's current stream is the one handed to amx_rollout)
            xs = rollout()
            if events is not None:
                events[k][1].record()
            if world > 1 and mode == "overlap":
                # (the pending handle holds xs until waited for: the name is rebound by the next rollout)
                pending.append(p2p.start(xs) if how == "p2p" else gather_trajectories_async(xs))
                if len(pending) > 1:
                    done_pg = pending.pop(0)
                    done_pg.wait()
                    if how == "p2p":
                        done_pg.release()  # the gathered view belongs to us until released (PeerCopyGather: rotating buffers)
            elif world > 1 and mode == "serial":
                if how == "p2p":
                    pg = p2p.start(xs)
                    xs = pg.wait()
                    pg.release()  # (the contents stay readable until the peers' next gather into this buffer, two gathers on)
                else:
                    xs = gather_trajectories(xs, equal_blocks=True)
        for pg in pending:
            xs = pg.wait()
            if how == "p2p":
                pg.release()
        return xs

    def timed(nsteps, mode, with_events=False, how=None):
        sync()
        if world > 1:
            dist.barrier()
        sync()
        ev = None
        if with_events and not dry:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
        t0 = time.perf_counter()
        xs = run_steps(nsteps, mode, ev, how)
        sync()
        if world > 1:
            dist.barrier()
        sync()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float("nan")
        return elapsed, kern_ms, xs

    mode = "none" if (world == 1 or args.gather == "none") else ("serial" if args.no_overlap else "overlap")
    run_steps(args.warmup, mode)
    elapsed, kern_ms, xs = timed(args.steps, mode, with_events=True)
    finite = bool(torch.isfinite(xs).all().item())
    activity = None
    if rank == 0 and not dry:
        # what the robots did (rank 0's own block; outside the timed region): base height, contacts in touch, constraint rows
        try:
            activity = activity_stats(model, mj, xs[:args.nenv], us)
        except Exception as e:  # a diagnostic must never take the headline down
            activity = {"error": f"{type(e).__name__}: {e}"}
    block_order_ok = None
    if dry:
        # gathered blocks are in rank order: block r is filled with r
        want = torch.arange(world, dtype=torch.float32).repeat_interleave(args.nenv) if mode != "none" else torch.full((args.nenv,), float(rank))
        assert torch.equal(xs[:, 0, 0], want), "gathered trajectory blocks are out of order"
        block_order_ok = True

    report = None
    if world > 1 and not args.no_scaling_report:
        k = args.steps
        es = args.nenv * args.horizon * k * world
        gb = args.nenv * (args.horizon + 1) * nx * 4 * world / 1e9

        def alone(gather_once):  # k gathers of one resident block, nothing else on the GPU
            xs_local = rollout()
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                gather_once(xs_local)
            sync()
            dist.barrier()
            tg = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            return float(tg.item())

        t_roll, _, _ = timed(k, "none")
        report = {"rollout_only_env_steps_per_s": es / t_roll, "gathered_GB_per_rank": gb,
                  "note": "single-GPU throughput of this workload = `value` of the N=1 line (same model, same envs per GPU); the rollout's "
                          "workgroups fill every CU (LDS and registers), so a collective's kernels run before or after a rollout, never "
                          "next to it (profiles/r03_coresidency.txt) - copies on the copy engines do"}
        if not share:  # the RCCL collective (skipped when the ranks share one device: no RCCL there)
            t_serial, _, _ = timed(k, "serial", how="rccl")
            t_over, _, _ = timed(k, "overlap", how="rccl")
            t_gather = alone(lambda x: gather_trajectories(x, equal_blocks=True))
            report.update({"with_gather_serial_env_steps_per_s": es / t_serial, "with_gather_overlapped_env_steps_per_s": es / t_over,
                           "gather_alone_ms": 1e3 * t_gather / k, "gather_GBps_per_rank_received": gb * (world - 1) / world / (t_gather / k)})
        if p2p is not None:
            t_pser, _, _ = timed(k, "serial", how="p2p")
            t_pover, _, _ = timed(k, "overlap", how="p2p")
            t_pg = alone(lambda x: p2p.start(x).wait(copy=True))
            report.update({"with_peer_copy_serial_env_steps_per_s": es / t_pser, "with_peer_copy_overlapped_env_steps_per_s": es / t_pover,
                           "peer_copy_alone_ms": 1e3 * t_pg / k, "peer_copy_GBps_per_rank_received": gb * (world - 1) / world / (t_pg / k)})
        if p2p_note:
            report["peer_copy"] = p2p_note
        if block_order_ok is not None:
            report["block_order_ok"] = block_order_ok  # dry run: every rank found block r of the gathered tensor filled by rank r
        report.update(backend_report())

    if rank == 0:
        env_steps = args.nenv * args.horizon * args.steps * world
        tag = {QUADRUPED: "BASELINE.md C2", BIPED: "BASELINE.md C3 per GPU / C5 sharded; legacy: random torques, feet-only contact",
               BIPED_STAND: "BASELINE.md C3 per GPU / C5 sharded; held standing on its convex-mesh feet by joint position servos; since r05 the soles are an irregular hexagon on which MJX's 4-point manifold heuristic has no ties (DESIGN.md 6) - same vertex count and launch plan",
               BIPED_REST: "BASELINE.md C3 per GPU / C5 sharded; random torques, falls and rests on eight extra contact spheres"}.get(model_path, "custom")
        out = {
            "metric": "env-steps/sec, 4096 envs x 1000-step rollout",
            "value": env_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" + (" (dry run: no kernel, zero trajectories on the CPU)" if dry else ""),
            "config": {"workload": f"{os.path.basename(model_path)} (build-authored, {tag}), {args.nenv} envs/GPU x "
                                   f"{args.horizon}-step fused rollout, ctrl ~U(ctrlrange) i.i.d. per env per step, Newton {mj.opt.iterations} it / "
                                   f"{mj.opt.ls_iterations} ls, dt={mj.opt.timestep}",
                       "nenv_per_gpu": args.nenv, "horizon": args.horizon, "nq": mj.nq, "nv": mj.nv, "nu": mj.nu,
                       "gather": {"none": "none", "serial": "traj", "overlap": "traj+overlapped"}[mode] + ("" if method != "p2p" else " (peer-to-peer copies on the copy engines)"),
                       "outputs_finite": finite},
        }
        if not dry:
            out["config"]["activity"] = activity
            out["roofline"] = roofline(mj, args.nenv, args.horizon, kern_ms, model_path)
        if report:
            out["scaling_report"] = report
        if world == 1 and not dry:
            x0_np, us_np = x0.cpu().numpy(), us.cpu().numpy()
            if not args.no_secondary:
                out["secondary"] = secondary(args, model, mj, model_path, x0_np, us_np, kern_ms, dev)
            out["mjx_cpu"] = probe_mjx_cpu()
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(mj, x0_np, us_np, target_s=10.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        if p2p is not None:
            p2p.close()
        dist.barrier()
        dist.destroy_process_group()


def solver_variant(mj, iterations, ls_iterations):
    """The host model with other solver iteration caps (everything else, tolerances included, as loaded)."""
    import copy
    from ambersim_amd.mjcf import recompile_options
    mv = copy.copy(mj)
    mv.opt = mj.opt.replace(iterations=int(iterations), ls_iterations=int(ls_iterations))
    return recompile_options(mv)


SOLVER_VARIANTS = ((100, 50), (5, 10))  # MuJoCo's defaults; the reference pendulum's (models/pendulum/pendulum.xml:3)


def secondary(args, model, mj, model_path, x0_np, us_np, kern_ms, dev):
    """N = 1 extras: C2 (quadruped rollout, its own roofline + CPU baseline), the biped at other solver options, C4 (FD gradient),
    A8 (trajectory cost), the FP32 figure."""
    import numpy as np
    import torch
    from ambersim_amd import mjx
    from ambersim_amd.trajopt.fd import fd_eps, fd_grad_terminal_cost
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    out = []

    def fp32(mj_, x0n, usn, steps_per_s, path):
        fr = flop_record(path)
        if fr is None:
            return {"note": "no recorded flop census for this model (tools/flop_census.py)"}
        fl = fr["flops"]
        rec = pmc_record(path, args.nenv, args.horizon)
        return {"algorithmic_flops_per_env_step": fl, "achieved_tflops": fl * steps_per_s / 1e12, "peak_tflops": FP32_PEAK_TFLOPS,
                "frac": fl * steps_per_s / 1e12 / FP32_PEAK_TFLOPS,
                "flops_source": f"RECORDED: {fr['file']} ({fr['how']})",
                "valu_busy_frac": rec.get("valu_busy_frac") if rec else None,
                "valu_busy_source": (f"RECORDED: {rec['file']}" if rec and "valu_busy_frac" in rec else None)}

    steps_per_s = args.nenv * args.horizon / (kern_ms * 1e-3)
    # A8: the batched trajectory cost over the headline's trajectories - the one HBM-shaped kernel of the path
    try:
        from ambersim_amd.trajopt.cost import StaticGoalQuadraticCost
        nx = mj.nq + mj.nv
        xs_t = mjx.rollout(model, torch.tensor(x0_np, device=dev), torch.tensor(us_np, device=dev))
        us_t = torch.tensor(us_np, device=dev)
        cost = StaticGoalQuadraticCost(torch.eye(nx, device=dev), 2 * torch.eye(nx, device=dev), 0.1 * torch.eye(mj.nu, device=dev), xs_t[0, 0].clone())
        cost.cost(xs_t, us_t)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            cost.cost(xs_t, us_t)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        nbytes = 4.0 * (xs_t.numel() + us_t.numel())
        out.append({"name": "A8 quadcost", "metric": "trajectory bytes/s of amx_quadcost (StaticGoalQuadraticCost over the headline's rollouts)",
                    "ms_per_call": ms, "roofline": {"bound": "hbm", "achieved": nbytes / ms * 1e-6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                    "frac": nbytes / ms * 1e-6 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes,
                                                    "kernel": "amx_quadcost_mfma_kernel",
                                                    "note": "bound by LDS operand staging and the fp32 MFMA rate in equal parts (DESIGN.md §4); 1-2 % of a trajectory-optimisation iteration"}})
        del xs_t, us_t
    except Exception as e:  # a secondary figure must never take the headline down
        out.append({"name": "A8 quadcost", "error": f"{type(e).__name__}: {e}"})
    out.append({"name": "fp32_valu", "workload": os.path.basename(model_path), **fp32(mj, x0_np, us_np, steps_per_s, model_path)})

    def timed_rollouts(mdl, x0_, us_, k):
        mjx.rollout(mdl, x0_, us_)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record()
            xs_ = mjx.rollout(mdl, x0_, us_)
            b.record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el, float(np.mean([a.elapsed_time(b) for a, b in ev])), bool(torch.isfinite(xs_).all().item())

    if model_path != QUADRUPED:
        # C2: the quadruped, 4096 x 1000 (BASELINE.json configs[1])
        try:
            mjq = load_mj_model_from_file(QUADRUPED)
            mq = mjx.put_model(mjq, dev)
            x0q, usq = make_inputs(mjq, args.nenv, args.horizon, 1, 2, dev)
            k = 5
            el, kms, fin = timed_rollouts(mq, x0q, usq, k)
            try:
                act_q = activity_stats(mq, mjq, mjx.rollout(mq, x0q, usq), usq)
            except Exception as e:
                act_q = {"error": f"{type(e).__name__}: {e}"}
            x0qn, usqn = x0q.cpu().numpy(), usq.cpu().numpy()
            entry = {"name": "C2", "metric": "env-steps/sec, 4096 envs x 1000-step rollout", "value": args.nenv * args.horizon * k / el, "unit": "env-steps/s",
                     "steps": k, "ms_per_step": 1e3 * el / k,
                     "config": {"workload": f"quadruped.xml (build-authored, BASELINE.md C2: free base + 12 hinges, sphere feet), {args.nenv} envs x {args.horizon}-step fused rollout, "
                                            f"random ctrl, Newton {mjq.opt.iterations} it / {mjq.opt.ls_iterations} ls; only the feet collide: under random servo targets the base ends below the floor "
                                            "(activity)", "nq": mjq.nq, "nv": mjq.nv, "nu": mjq.nu, "outputs_finite": fin, "activity": act_q},
                     "roofline": roofline(mjq, args.nenv, args.horizon, kms, QUADRUPED),
                     "fp32_valu": fp32(mjq, x0qn, usqn, args.nenv * args.horizon / (kms * 1e-3), QUADRUPED)}
            if not args.no_cpu_baseline:
                entry["cpu_baseline"] = cpu_baseline(mjq, x0qn, usqn, target_s=6.0)
            out.append(entry)
            del usq, mq
        except Exception as e:
            out.append({"name": "C2", "error": f"{type(e).__name__}: {e}"})

    if model_path == HEADLINE:
        # the same robot, the two other workloads: what they cost and what the robots do in them
        for name, path, what in (("C3 falls-and-rests", BIPED_REST, "random torques ~U(ctrlrange); two convex-mesh feet + eight contact spheres (pelvis, chest, shoulders, head, knees, "
                                  "hands): the robot falls within ~150 steps and rests on the floor; 16 contact slots = 64 pyramid rows, two to a lane of a 32-lane group in the planner's dense form "
                                  "(r05: compact M, three-row Jacobian blocks, sixteen environments per CU; one environment per wavefront and eight per CU before)"),
                                 ("C3 legacy feet-only", BIPED, "random torques ~U(ctrlrange); only the two convex-mesh feet collide (r01-r03 headline): after the fall the robot hangs "
                                  "below the floor from its feet")):
            try:
                mjv = load_mj_model_from_file(path)
                mdl = mjx.put_model(mjv, dev)
                x0v, usv = make_inputs(mjv, args.nenv, args.horizon, 1, 2, dev)
                k = 5
                el, kms, fin = timed_rollouts(mdl, x0v, usv, k)
                xsv = mjx.rollout(mdl, x0v, usv)
                out.append({"name": name, "metric": "env-steps/sec, 4096 envs x 1000-step rollout", "value": args.nenv * args.horizon * k / el, "unit": "env-steps/s",
                            "steps": k, "ms_per_step": 1e3 * el / k, "kernel_ms": kms,
                            "config": {"workload": f"{os.path.basename(path)}: {what}; {args.nenv} envs x {args.horizon} steps, Newton {mjv.opt.iterations} it / {mjv.opt.ls_iterations} ls",
                                       "lanes_per_env": int(mdl._h.query("lanes_per_env")), "envs_per_workgroup": int(mdl._h.query("envs_per_block")), "specialised_build": bool(mdl._h.query("specialised")),
                                       "activity": activity_stats(mdl, mjv, xsv, usv), "outputs_finite": fin},
                            "roofline": roofline(mjv, args.nenv, args.horizon, kms, path),
                            "fp32_valu": fp32(mjv, None, None, args.nenv * args.horizon / (kms * 1e-3), path)})
                del mdl, xsv, usv
            except Exception as e:
                out.append({"name": name, "error": f"{type(e).__name__}: {e}"})

        # the headline's cost is set by the model file's solver options (Newton 3 iterations / 5 line-search iterations).  The same
        # workload at MuJoCo's defaults (100 / 50: the tolerance exit decides) and at the reference pendulum's 5 / 10
        # (reference ambersim/models/pendulum/pendulum.xml:3).  Each variant runs its own per-model build of the kernels
        # (mjx.put_model(specialise=True): prebuilt by __graft_entry__.build(), compiled here if absent).
        x0_t, us_t = torch.tensor(x0_np, device=dev), torch.tensor(us_np, device=dev)
        for it, ls in SOLVER_VARIANTS:
            name = f"C3 solver options {it}/{ls}"
            try:
                t0 = time.perf_counter()
                mv = solver_variant(mj, it, ls)
                mdl = mjx.put_model(mv, dev, specialise=True)
                t_build = time.perf_counter() - t0
                k = 3
                el, kms, fin = timed_rollouts(mdl, x0_t, us_t, k)
                out.append({"name": name, "metric": "env-steps/sec, 4096 envs x 1000-step rollout", "value": args.nenv * args.horizon * k / el, "unit": "env-steps/s",
                            "steps": k, "ms_per_step": 1e3 * el / k, "kernel_ms": kms,
                            "config": {"workload": f"{os.path.basename(model_path)} with <option iterations={it} ls_iterations={ls}> (tolerance {mv.opt.tolerance:g}: the solver leaves its loop when the "
                                                   f"scaled improvement or gradient falls below it), {args.nenv} envs x {args.horizon} steps, same x0 / ctrl as the headline",
                                       "specialised_build": bool(mdl._h.query("specialised")), "model_setup_s": t_build, "outputs_finite": fin}})
                del mdl
            except Exception as e:
                out.append({"name": name, "error": f"{type(e).__name__}: {e}"})
        # the headline model on the GENERIC kernels (run-time descriptor): what a model that is neither shipped nor given a per-model
        # build gets
        try:
            os.environ["AMX_NO_SPEC"] = "1"
            try:
                mdl = mjx.put_model(mj, dev, specialise=False)
            finally:
                del os.environ["AMX_NO_SPEC"]
            k = 3
            el, kms, fin = timed_rollouts(mdl, x0_t, us_t, k)
            out.append({"name": "C3 generic kernels", "metric": "env-steps/sec, 4096 envs x 1000-step rollout", "value": args.nenv * args.horizon * k / el, "unit": "env-steps/s",
                        "steps": k, "ms_per_step": 1e3 * el / k, "kernel_ms": kms,
                        "config": {"workload": f"{os.path.basename(model_path)}, same inputs as the headline, on amx_rollout_kernel<G, DynSpec> (no per-model build)",
                                   "specialised_build": bool(mdl._h.query("specialised")), "outputs_finite": fin}})
            del mdl
        except Exception as e:
            out.append({"name": "C3 generic kernels", "error": f"{type(e).__name__}: {e}"})
        del x0_t, us_t

    if True:
        mjb = mj if model_path == BIPED else load_mj_model_from_file(BIPED)
        mb = model if model_path == BIPED else mjx.put_model(mjb, dev)
        x0b = torch.tensor(x0_np[:1], device=dev) if model_path == BIPED else make_inputs(mjb, 1, 1, 1, 2, dev)[0]
        # C4: 1024 rollouts x 64 knots from one x0, forward-difference gradient of the terminal cost (amx_fd_grad)
        S, N = 1024, 64
        nxb = mjb.nq + mjb.nv
        lo = torch.tensor(mjb.actuator_ctrlrange[:, 0], dtype=torch.float32, device=dev)
        hi = torch.tensor(mjb.actuator_ctrlrange[:, 1], dtype=torch.float32, device=dev)
        g = torch.Generator(device=dev).manual_seed(4)
        usn = 0.5 * (lo + hi) + 0.25 * (hi - lo) * (2 * torch.rand((N, mjb.nu), generator=g, device=dev) - 1)
        Qf = torch.eye(nxb, device=dev)
        call = lambda: fd_grad_terminal_cost(mb, x0b[0], usn, Qf, x0b[0], fd_eps(mb), nsamples=S)
        call()
        torch.cuda.synchronize()
        k = 10
        t0 = time.perf_counter()
        for _ in range(k):
            r = call()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out.append({"name": "C4", "metric": "env-steps/sec, 1024 rollouts x 64 knots, forward-difference gradient of the terminal cost wrt ctrl",
                    "value": S * N * k / el, "unit": "env-steps/s", "steps": k, "ms_per_gradient_launch": 1e3 * el / k,
                    "config": {"workload": f"biped.xml, {S} rollouts x {N} steps from one shared x0 (rollout 0 nominal, rollout k perturbs control coordinate k-1 "
                                           "by 1e-3*range), terminal cost + difference on device (amx_fd_grad: 3 kernels, final states only)",
                               "coordinates_per_launch": S - 1, "grad_finite": bool(torch.isfinite(r.grad).all().item())}})
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
