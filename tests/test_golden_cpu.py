"""The committed golden rollouts (tests/golden/rollouts.npz) are reproduced by the CPU oracle."""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle

from conftest import GOLDEN


def golden_cases():
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    return make_golden.cases()


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "rollouts.npz"))


@pytest.mark.parametrize("name", ["pendulum", "pendulum_ctrl", "quadruped", "biped", "biped_stand", "biped_rest", "bh280"])
def test_oracle_reproduces_golden(golden, name):
    mj, x0, us = golden_cases()[name]
    assert np.array_equal(x0, golden[name + "_x0"]) and np.array_equal(us, golden[name + "_us"])
    xs = Oracle(mj, np.float64).rollout(x0, us, nthreads=2)
    ref = golden[name + "_xs"]
    assert np.abs(xs - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    # fp32 restatement of the same algorithm: one step from each golden state stays within fp32 rounding
    o32 = Oracle(mj, np.float32)
    nq = mj.nq
    worst = 0.0
    for e in range(min(4, x0.shape[0])):
        for t in range(0, us.shape[1], max(1, us.shape[1] // 5)):
            # identical inputs for both: the golden state rounded to fp32 (a flat foot's 4-point manifold selection is discontinuous in the
            # state - fed the unrounded fp64 state, the fp64 oracle can pick other corners than the fp32 one does from the rounded state)
            q32, v32 = ref[e, t, :nq].astype(np.float32), ref[e, t, nq:].astype(np.float32)
            o32.set_state(q32, v32, us[e, t], np.zeros(mj.nv))
            o64 = Oracle(mj, np.float64)
            o64.set_state(q32, v32, us[e, t], np.zeros(mj.nv))
            o32.step(); o64.step()
            a = np.concatenate([o32.get("qpos"), o32.get("qvel")]).astype(np.float64)
            b = np.concatenate([o64.get("qpos"), o64.get("qvel")])
            worst = max(worst, np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    assert worst < 2e-4, worst
