import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_closed_forms import EQ_BOXES, _integrate
from ambersim_amd.mjcf import MjModel
from ambersim_amd import mjx
from oracle.oracle import Oracle
mj = MjModel.from_xml_string(EQ_BOXES)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4)
n = 16
PERT = float(sys.argv[1]) if len(sys.argv) > 1 else 0.04
x0 = np.zeros((n, mj.nq + mj.nv))
for e in range(n):
    x0[e, : mj.nq] = _integrate(mj, mj.qpos0.copy(), PERT * rng.standard_normal(mj.nv), 1.0)
    x0[e, mj.nq:] = 7.5 * PERT * rng.standard_normal(mj.nv)
us = rng.uniform(-1, 1, (n, 10, mj.nu))
ref = Oracle(mj, np.float64).rollout(x0, us); r32 = Oracle(mj, np.float32).rollout(x0, us)
model = mjx.put_model(mj, dev)
xs = mjx.rollout(model, torch.tensor(x0, dtype=torch.float32), torch.tensor(us, dtype=torch.float32)).cpu().numpy()
for t in (1, 2, 10):
    e = np.abs(xs[:, t] - ref[:, t]).max(axis=1); e32 = np.abs(r32[:, t] - ref[:, t]).max(axis=1)
    print("step", t, "gpu err per env", e.round(6), "\n   f32 oracle", e32.round(6))
k = int(np.argmax(np.abs(xs[:, 1] - ref[:, 1]).max(axis=1)))
print("worst env", k, "\n gpu", xs[k, 1].round(5), "\n ref", ref[k, 1].round(5))
from test_gpu_kat import Gpu
g = Gpu(mj, dev)
o, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
for e in (0, 3, 5):
    f = g.forward(x0[e, :mj.nq], x0[e, mj.nq:], us[e, 0])
    for oo in (o, o32):
        oo.set_state(x0[e, :mj.nq], x0[e, mj.nq:], us[e, 0]); oo.forward()
    a, a32 = o.get("qacc"), o32.get("qacc")
    print("env", e, "qacc err gpu", np.abs(f["qacc"] - a).max(), "f32 oracle", np.abs(a32 - a).max(), "max|qacc|", np.abs(a).max(), "niter f64/f32", o.get("solver_niter")[0], o32.get("solver_niter")[0])
    print("   fcon err gpu", np.abs(f["qfrc_constraint"] - o.get("qfrc_constraint")).max(), "f32", np.abs(o32.get("qfrc_constraint") - o.get("qfrc_constraint")).max(), "max", np.abs(o.get("qfrc_constraint")).max())
print("---- step vs rollout")
q = torch.tensor(x0[:, :mj.nq], dtype=torch.float32); v = torch.tensor(x0[:, mj.nq:], dtype=torch.float32)
d = mjx.make_data(model, n).replace(qpos=q, qvel=v)
s1 = mjx.step(model, d)
g1 = np.concatenate([s1.qpos.cpu().numpy(), s1.qvel.cpu().numpy()], axis=1)
print("mjx.step (cold warmstart) err vs f64 rollout step 1 per env:", np.abs(g1 - ref[:, 1]).max(axis=1).round(6))
fw = mjx.forward(model, d)
d2 = d.replace(qacc_warmstart=fw.qacc)
s2 = mjx.step(model, d2)
g2 = np.concatenate([s2.qpos.cpu().numpy(), s2.qvel.cpu().numpy()], axis=1)
print("mjx.step (warm = forward qacc) err:", np.abs(g2 - ref[:, 1]).max(axis=1).round(6))
print("rollout step 1 err:", np.abs(xs[:, 1] - ref[:, 1]).max(axis=1).round(6))
