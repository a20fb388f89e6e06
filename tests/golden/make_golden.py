"""Generates tests/golden/rollouts.npz: seeded inputs and the fp64 CPU-oracle trajectories for them.

The reference itself cannot produce vectors here (its physics is the absent mujoco-mjx package, SURVEY.md
§8c), so these fixtures pin the *oracle* (regression) and give the GPU tests committed expected outputs.
The oracle in turn is pinned by closed-form known-answer tests (tests/test_oracle_kat.py).

Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from ambersim_amd.mjcf import DisableBit, MjModel, recompile_options  # noqa: E402
from ambersim_amd.utils.io_utils import load_mj_model_from_file  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def bh280_fixture_model():
    """bh280 with the options of the reference's VPS fixture (tests/trajopt/test_predictive_sampler.py:22-31)."""
    mj = MjModel.from_xml_path(os.path.join(HERE, "bh280_nomesh.xml"))
    mj.opt = mj.opt.replace(timestep=0.002, iterations=1, ls_iterations=4, integrator=0, solver=2,
                            disableflags=DisableBit.CONTACT)
    return recompile_options(mj)


def cases():
    out = {}
    # C1: pendulum, 64 envs x 100 steps, qpos~U(-pi,pi), qvel~U(-2,2), default_rng(0), zero ctrl
    mj = load_mj_model_from_file("models/pendulum/pendulum.xml")
    rng = np.random.default_rng(0)
    x0 = np.stack([rng.uniform(-np.pi, np.pi, 64), rng.uniform(-2, 2, 64)], axis=1)
    out["pendulum"] = (mj, x0, np.zeros((64, 100, 1)))
    # pendulum with random torque beyond the +-2 control limit (exercises the clamp)
    rng = np.random.default_rng(10)
    out["pendulum_ctrl"] = (mj, x0[:16], rng.uniform(-3, 3, (16, 60, 1)))
    for name, path, nenv, N, seed in (("quadruped", "models/quadruped/quadruped.xml", 16, 50, 1),
                                      ("biped", "models/biped/biped.xml", 8, 30, 2)):
        mj = load_mj_model_from_file(path)
        rng = np.random.default_rng(seed)
        x0 = np.tile(np.concatenate([mj.key_qpos[0], np.zeros(mj.nv)]), (nenv, 1))
        x0[:, mj.nq:] += 0.05 * rng.standard_normal((nenv, mj.nv))
        us = rng.uniform(mj.actuator_ctrlrange[:, 0], mj.actuator_ctrlrange[:, 1], (nenv, N, mj.nu))
        out[name] = (mj, x0, us)
    # the bench's headline models (r05): the servo-held standing biped, and the one that falls and comes to rest on its contact
    # spheres - long enough to hold the fall and its impacts (BASELINE.json configs[2]; bench.make_inputs' distribution: standing
    # keyframe + N(0, 0.01^2) velocity noise, ctrl ~ U(ctrlrange))
    for name, path, nenv, N, seed in (("biped_stand", "models/biped/biped_stand.xml", 8, 100, 4),
                                      ("biped_rest", "models/biped/biped_rest.xml", 8, 240, 5)):
        mj = load_mj_model_from_file(path)
        rng = np.random.default_rng(seed)
        x0 = np.tile(np.concatenate([mj.key_qpos[0], np.zeros(mj.nv)]), (nenv, 1))
        x0[:, mj.nq:] += 0.01 * rng.standard_normal((nenv, mj.nv))
        us = rng.uniform(mj.actuator_ctrlrange[:, 0], mj.actuator_ctrlrange[:, 1], (nenv, N, mj.nu))
        out[name] = (mj, x0.astype(np.float32).astype(np.float64), us.astype(np.float32).astype(np.float64))  # exactly representable in fp32
    mj = bh280_fixture_model()
    rng = np.random.default_rng(3)
    x0 = 0.3 * rng.standard_normal((10, mj.nq + mj.nv))
    x0[0] = 0.0  # the reference's smoke test starts at x0 = 0, on every lower joint limit
    out["bh280"] = (mj, x0, rng.standard_normal((10, 10, mj.nu)))
    return out


def main():
    blob = {}
    for name, (mj, x0, us) in cases().items():
        xs = Oracle(mj, np.float64).rollout(x0, us)
        assert np.isfinite(xs).all(), name
        blob[name + "_x0"], blob[name + "_us"], blob[name + "_xs"] = x0, us, xs
        print(name, xs.shape)
    np.savez_compressed(os.path.join(HERE, "rollouts.npz"), **blob)


if __name__ == "__main__":
    main()
