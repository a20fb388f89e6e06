"""Convex-convex one-step error against the solver's iteration count (box-box draw of tests/test_convex.py): where along the
Newton iterations the HIP path leaves the fp64 oracle's, next to the fp32 oracle's own distance."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import TWO
from ambersim_amd import mjx
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
xml = TWO.format(pb="0 0 0.7", qb="", gb='type="box" size="0.3 0.2 0.25"', ja="<freejoint/>", g=-9.81, asset="")
rng = np.random.default_rng(11)
nenv = 32
mj0 = MjModel.from_xml_string(xml)
q = np.tile(mj0.qpos0, (nenv, 1))
for k in range(nenv):
    for a in (3, 10):
        quat = np.array([1.0, 0, 0, 0]) + 0.5 * rng.uniform(-1, 1, 4)
        q[k, a:a + 4] = quat / np.linalg.norm(quat)
    q[k, 7:10] = [0.3 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), rng.uniform(0.6, 0.95)]
v = 0.3 * rng.standard_normal((nenv, mj0.nv))
x0 = np.concatenate([q, v], axis=1).astype(np.float32)
us = np.zeros((nenv, 1, 0), dtype=np.float32)
sel = [3, 12, 13, 16, 18, 19]
for it, ls in ((1, 5), (2, 5), (3, 5), (4, 5), (3, 10), (3, 20), (3, 50), (8, 50)):
    mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', f'iterations="{it}" ls_iterations="{ls}"'))
    model = mjx.put_model(mj, dev)
    xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
    r64 = Oracle(mj, np.float64).rollout(x0, us); r32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
    dt = mj.opt.timestep
    a_g, a_64, a_32 = [(r[:, 1, mj.nq:] - x0[:, mj.nq:]) / dt for r in (xs, r64, r32)]
    s = np.maximum(1.0, np.abs(a_64).max(axis=1))
    eg, e32 = np.abs(a_g - a_64).max(axis=1) / s, np.abs(a_32 - a_64).max(axis=1) / s
    print(f"iterations {it} ls {ls}: qacc rel err gpu max {eg.max():.1e} median {np.median(eg):.1e} | cpu32 max {e32.max():.1e} median {np.median(e32):.1e} | envs {sel}: gpu",
          " ".join(f"{eg[k]:.0e}" for k in sel), "cpu32", " ".join(f"{e32[k]:.0e}" for k in sel))

# per-env fp32 floor: how far the fp64 answer moves when the inputs move by one fp32 ulp (a backward-stable fp32 algorithm
# cannot be expected to do better than a small multiple of this)
mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', 'iterations="3" ls_iterations="5"'))
model = mjx.put_model(mj, dev)
xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
o64 = Oracle(mj, np.float64)
r64 = o64.rollout(x0, us); r32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
prng = np.random.default_rng(5)
floor = np.zeros(nenv)
for rep in range(8):
    xp = x0.astype(np.float64) * (1 + 6e-8 * prng.choice([-1.0, 1.0], size=x0.shape))
    rp = o64.rollout(xp, us)
    floor = np.maximum(floor, np.abs(rp[:, 1] - r64[:, 1]).max(axis=1) / np.maximum(1.0, np.abs(r64[:, 1]).max(axis=1)))
eg = np.abs(xs[:, 1] - r64[:, 1]).max(axis=1) / np.maximum(1.0, np.abs(r64[:, 1]).max(axis=1))
e32 = np.abs(r32[:, 1] - r64[:, 1]).max(axis=1) / np.maximum(1.0, np.abs(r64[:, 1]).max(axis=1))
for k in range(nenv):
    print(f"env {k}: one-step state err gpu {eg[k]:.1e} cpu32 {e32[k]:.1e} | one-ulp input sensitivity {floor[k]:.1e} | gpu/floor {eg[k] / max(floor[k], 1e-12):.1f}")
