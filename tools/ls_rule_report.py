#!/usr/bin/env python3
"""Where MJX's line-search acceptance rule and the kernel's part ways, measured.  CPU only.

    python tools/ls_rule_report.py [--nenv 256] [--steps 1000] [--out profiles/r04_ls_rule_report.txt]

For C2 (quadruped) and C3 (biped) on the bench inputs, the CPU oracle (oracle/amx_oracle.c) runs in fp64 and fp32 with
MJX's formulation (ls_rule 0) and with the kernel's (ls_rule 3); every line search evaluates BOTH acceptance tests at
its final bracket and counts what each decides.  Printed per run: line searches, steps accepted by either rule, the
env-steps (solver calls) in which MJX's rule dropped a step the kernel's rule takes, the stopping tests on `improvement`
that come out differently, and the largest dropped decrease relative to cost(0) (sub-ulp by construction)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ambersim_amd.utils.io_utils import load_mj_model_from_file
    from oracle.oracle import Oracle
    from tools.parity_horizons import inputs
    ap = argparse.ArgumentParser()
    ap.add_argument("--nenv", type=int, default=256)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    lines = [f"# line-search rule report: {a.nenv} envs x {a.steps} steps per config, bench inputs (standing keyframe + noise, ctrl ~ U(ctrlrange))",
             "# run = precision / rule the trajectory follows (0 = MJX's, 3 = the kernel's); both acceptance tests are evaluated at every line search of that trajectory",
             f"{'config':<14}{'run':<12}{'linesearches':>13}{'acc MJX':>10}{'acc kernel':>11}{'dropped by MJX':>15}{'env-steps affected':>19}{'frac':>8}{'stop tests differ':>18}{'max dropped decrease/cost':>27}"]
    for name, path in (("C2 quadruped", "models/quadruped/quadruped.xml"), ("C3 biped", "models/biped/biped.xml")):
        mj = load_mj_model_from_file(path)
        x0, us = inputs(mj, a.nenv, a.steps, 7)
        for dt in (np.float64, np.float32):
            for rule in (0, 3):
                o = Oracle(mj, dt, ls_rule=rule, ls_stats=True)
                o.rollout(x0, us, keep_traj=False)
                s = o.ls_stats()
                lines.append(f"{name:<14}{np.dtype(dt).name + ' / ' + str(rule):<12}{s['linesearches']:>13}{s['accept_mjx_rule']:>10}{s['accept_kernel_rule']:>11}"
                             f"{s['accept_differ']:>15}{s['solves_affected']:>19}{s['solves_affected'] / max(1, s['solves']):>8.3f}{s['stop_differ']:>18}"
                             f"{s['accept_differ_decrease_max']:>27.2e}")
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
