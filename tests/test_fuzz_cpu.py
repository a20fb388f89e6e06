"""The fuzz generator is deterministic and every draw compiles and steps finitely through the fp64 oracle (the GPU side of
the same draws is tests/test_gpu_parity.py::test_random_models_vs_oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_models import random_inputs, random_mjcf, random_model  # noqa: E402


def test_generator_is_deterministic_and_models_step():
    from oracle.oracle import Oracle
    assert random_mjcf(7) == random_mjcf(7) and random_mjcf(7) != random_mjcf(8)
    kinds, usable = set(), 0
    for seed in range(40):
        m = random_model(seed)
        if m is None:
            continue
        kinds |= set(int(t) for t in m.jnt_type)
        x0, us = random_inputs(m, seed, 2, 3)
        xs = Oracle(m, np.float64).rollout(x0, us)
        assert xs.shape == (2, 4, m.nq + m.nv)
        usable += bool(np.isfinite(xs).all())
    assert kinds == {0, 1, 2, 3} and usable >= 35  # free, ball, slide and hinge joints all occur


def test_convex_family_is_deterministic_and_covers_the_convex_pairs():
    """The convex family (r04): boxes and small mesh hulls beside spheres and capsules, roots low over the floor.  Leaves the plain
    family's draws as they were; every convex pair function occurs; most draws start with active contacts and step finitely."""
    from oracle.oracle import Oracle
    assert random_mjcf(7, convex=True) == random_mjcf(7, convex=True) and random_mjcf(7, convex=True) != random_mjcf(7)
    fns, usable, touching = set(), 0, 0
    for seed in range(60):
        m = random_model(seed, convex=True)
        if m is None:
            continue
        fns |= set(int(f) for f in m.pair_fn)
        x0, us = random_inputs(m, seed, 2, 3)
        o = Oracle(m, np.float64)
        xs = o.rollout(x0, us)
        usable += bool(np.isfinite(xs).all())
        o.set_state(x0[0, :m.nq], x0[0, m.nq:], us[0, 0]); o.forward()
        touching += bool(m.ncon and (o.get("con_dist") < 0).any())
    assert {2, 6, 7, 11} <= fns and usable >= 35 and touching >= 15, (fns, usable, touching)
