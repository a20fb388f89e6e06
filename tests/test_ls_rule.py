"""The HIP kernel's line-search acceptance rule against the oracle's (MJX's), both ways.

MJX accepts a line-search step on `cost(alpha) < cost(0)` with both sides carrying the whole constant term; the kernel accepts on
the sign of `alpha^2 q2 + alpha q1 + sum_rows (q0_row(alpha) - q0_row(0))` and hands that decrease to the solver's stopping test
(ambersim_amd/csrc/amx_kernels.hip: Env::linesearch, Env::solve).  oracle/amx_oracle.c carries both: `ls_rule` 0 = MJX's
formulation, 1 = the kernel's rule, 3 = the kernel's rule with the kernel's arithmetic around it (reciprocal trial points,
d1 = max(2 q2, MINVAL), multiplied-out stopping tests, the quadratic part of the decrease from the bracket ends' derivatives).
With `ls_stats` every line search evaluates BOTH acceptance tests at the same bracket and counts what each decides.

What is shown here, on the CPU:
  * fp64: on the bench models and the Barrett hand the two rules never take a different accept / reject decision; over the fuzz
    models they differ only on steps whose decrease is below ONE ULP of cost(0) (the steps MJX's comparison cannot see in any
    precision - measured: <= 1e-17 of the cost); where both accept but name different bracket ends the two ends' decreases tie
    to rounding (<= 1e-9 relative); the iterates of the three builds agree to rounding before the first discontinuity of the
    contact dynamics;
  * fp32: the kernel's rule accepts every step MJX's rule accepts (the same bracket, a sharper comparison), and MJX's rule
    drops steps the kernel's takes - the measured fraction is printed by tools/ls_rule_report.py and quoted in DESIGN.md §5.
Reference call sites of the solver: /root/reference/ambersim/rl/base.py:93, ambersim/trajopt/shooting.py:41 (mjx.step)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_models import random_inputs, random_model  # noqa: E402

from ambersim_amd.utils.io_utils import load_mj_model_from_file  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BENCH_MODELS = ("models/quadruped/quadruped.xml", "models/biped/biped.xml", os.path.join(GOLD, "bh280_nomesh.xml"))


def bench_inputs(mj, nenv, N, seed=7):
    rng = np.random.default_rng(seed)
    q0 = mj.key_qpos[0] if mj.nkey else mj.qpos0
    x0 = np.tile(np.concatenate([q0, np.zeros(mj.nv)]), (nenv, 1))
    x0[:, mj.nq:] += 0.05 * rng.standard_normal((nenv, mj.nv))
    us = rng.uniform(mj.actuator_ctrlrange[:, 0], mj.actuator_ctrlrange[:, 1], (nenv, N, mj.nu))
    return x0, us


def run(mj, x0, us, dtype, rule):
    o = Oracle(mj, dtype, ls_rule=rule, ls_stats=True)
    xs = o.rollout(x0, us).astype(np.float64)
    return xs, o.ls_stats()


def rel(a, b):
    return np.abs(a - b).max(axis=-1) / np.maximum(1.0, np.abs(b).max(axis=-1))


def fuzz_cases(seeds):
    for seed in seeds:
        mj = random_model(seed)
        if mj is None or mj.neq + int(np.sum(mj.jnt_limited)) + mj.ncon == 0:
            continue  # no constraint rows: the solver never runs
        o = Oracle(mj, np.float64)
        o.forward()
        if np.linalg.cond(o.get("qM", (mj.nv, mj.nv))) > 1e10:
            # e.g. two ball joints on one body: M is singular, `qacc_smooth` does not solve M a = qfrc_smooth, and the line
            # search's quadratic (which assumes it does: its linear Gauss term is search . (M a - qfrc_smooth)) is not the cost
            continue
        yield seed, mj


def test_fp64_decisions_identical_on_fuzz_models():
    total = dict(linesearches=0, end_pick_differ=0)
    ncases = 0
    for seed, mj in fuzz_cases(range(150)):
        x0, us = random_inputs(mj, seed, 8, 12)
        ref, s0 = run(mj, x0, us, np.float64, 0)
        ok = np.isfinite(ref).all(axis=(1, 2))  # a few draws are unstable at their random time step: their finite part counts
        if not ok.any():
            continue
        EPS64 = np.finfo(np.float64).eps
        # a differing decision is a step whose decrease the full-cost comparison cannot resolve: below one ulp of cost(0)
        assert s0["accept_differ_decrease_max"] <= EPS64, (seed, s0)
        assert s0["accept_kernel_rule"] - s0["accept_mjx_rule"] == s0["accept_differ"], (seed, s0)  # one direction only
        assert s0["end_gap_max"] <= 1e-9, (seed, s0)
        # the solver's stopping test `improvement < tolerance * scale`: MJX takes the improvement from two full costs, the kernel
        # from the line search; the two tests may come out differently only inside the cancellation noise of the full costs
        assert s0["stop_differ_gap_max"] <= 1e-9, (seed, s0)
        for k in ("accept_differ", "stop_differ"):
            total[k] = total.get(k, 0) + s0[k]
        for rule in (1, 3):
            xs, s = run(mj, x0, us, np.float64, rule)
            assert s["accept_differ_decrease_max"] <= EPS64 and s["end_gap_max"] <= 1e-9 and s["stop_differ_gap_max"] <= 1e-9, (seed, rule, s)
            if s0["accept_differ"] + s0["stop_differ"] + s["accept_differ"] + s["stop_differ"] == 0:
                # every decision the same: the same iterates.  One step is exact to rounding; over the horizon contacts may
                # amplify a last-bit difference (manifold selection, row activation), so the horizon is held at the median
                assert rel(xs[ok, 1], ref[ok, 1]).max() <= 1e-10, (seed, rule)
                assert np.median(rel(xs[ok, -1], ref[ok, -1])) <= 1e-9, (seed, rule)
                total["clean"] = total.get("clean", 0) + 1
        total["linesearches"] += s0["linesearches"]
        total["end_pick_differ"] += s0["end_pick_differ"]
        ncases += 1
    assert ncases >= 60 and total["linesearches"] > 8000, (ncases, total)
    assert total["accept_differ"] + total["stop_differ"] <= 0.01 * total["linesearches"], total
    assert total["clean"] >= 1.6 * ncases, total  # two rule variants per case: at least 80 % of the runs take identical decisions throughout


@pytest.mark.parametrize("path", BENCH_MODELS)
def test_fp64_decisions_identical_on_bench_models(path):
    mj = load_mj_model_from_file(path)
    x0, us = bench_inputs(mj, 32, 120)
    ref, s0 = run(mj, x0, us, np.float64, 0)
    assert s0["linesearches"] > 1000
    assert s0["accept_differ"] == 0 and s0["solves_affected"] == 0 and s0["stop_differ"] == 0, s0
    assert s0["accept_mjx_rule"] == s0["accept_kernel_rule"]
    assert s0["end_gap_max"] <= 1e-9, s0
    for rule in (1, 3):
        xs, s = run(mj, x0, us, np.float64, rule)
        assert s["accept_differ"] == 0 and s["stop_differ"] == 0 and s["end_gap_max"] <= 1e-9, (rule, s)
        assert rel(xs[:, 1], ref[:, 1]).max() <= 1e-11, rule
        assert rel(xs[:, 10], ref[:, 10]).max() <= 1e-9, rule  # before the first foot impact of any environment
        assert np.median(rel(xs[:, -1], ref[:, -1])) <= 1e-9, rule


@pytest.mark.parametrize("path", BENCH_MODELS[:2])
def test_fp32_kernel_rule_only_adds_steps(path):
    """In single precision the two rules part ways in one direction only: every step MJX's comparison accepts, the decrease-form
    accepts too (same bracket; the difference of two 1e3..1e5 sums against the difference itself), and MJX's drops steps whose
    improvement is below half an ulp of the constant term."""
    mj = load_mj_model_from_file(path)
    x0, us = bench_inputs(mj, 32, 120)
    for rule in (0, 3):
        _, s = run(mj, x0, us, np.float32, rule)
        assert s["accept_kernel_rule"] >= s["accept_mjx_rule"], s
        assert s["accept_kernel_rule"] - s["accept_mjx_rule"] == s["accept_differ"], s  # never the other way round
        assert s["accept_differ"] > 0, s  # the effect is real on these models: DESIGN.md §5 quotes the fraction
        assert s["accept_differ_decrease_max"] <= np.finfo(np.float32).eps, s  # and it is exactly the sub-ulp steps
    # and with the kernel's rule the fp32 oracle is no farther from fp64 than with MJX's (median over environments, 10 steps)
    ref, _ = run(mj, x0, us, np.float64, 0)
    e0 = np.median(rel(run(mj, x0, us, np.float32, 0)[0][:, 10], ref[:, 10]))
    e3 = np.median(rel(run(mj, x0, us, np.float32, 3)[0][:, 10], ref[:, 10]))
    assert e3 <= 1.5 * e0 + 1e-7, (e0, e3)
