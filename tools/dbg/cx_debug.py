import sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_convex import TWO
from ambersim_amd import mjx, _lib
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
xml = TWO.format(pb="0 0 0.7", qb="", gb='type="box" size="0.3 0.2 0.25"', ja="<freejoint/>", g=-9.81, asset="")
mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', 'iterations="3" ls_iterations="5"'))
model = mjx.put_model(mj, dev)
print("G", model._h.query("lanes_per_env"), "ncon", model.ncon)
rng = np.random.default_rng(11)
nenv = 32
q = np.tile(mj.qpos0, (nenv, 1))
for k in range(nenv):
    for a in (3, 10):
        quat = np.array([1.0, 0, 0, 0]) + 0.5 * rng.uniform(-1, 1, 4)
        q[k, a:a + 4] = quat / np.linalg.norm(quat)
    q[k, 7:10] = [0.3 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), rng.uniform(0.45, 0.95)]
v = 0.3 * rng.standard_normal((nenv, mj.nv))
tq, tv = torch.tensor(q, dtype=torch.float32, device=dev), torch.tensor(v, dtype=torch.float32, device=dev)
tu = torch.zeros((nenv, 1), device=dev)
cd = torch.zeros((nenv, 4), device=dev)
out = _lib.AmxForwardOut(None, None, None, None, None, None, None, cd.data_ptr(), None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
o = Oracle(mj, np.float64)
for k in range(nenv):
    o.set_state(q[k].astype(np.float32), v[k], np.zeros(0)); o.forward()
    ref = o.get("con_dist")
    g = cd[k].cpu().numpy()
    o32 = Oracle(mj, np.float32); o32.set_state(q[k].astype(np.float32), v[k], np.zeros(0)); o32.forward()
    flag = "" if np.allclose(g, ref, atol=1e-5) else ("   <<<<< 32ok" if np.allclose(g, o32.get("con_dist"), atol=1e-5) else "   <<<<< BAD")
    print(k, np.round(g, 5), np.round(ref, 5), np.round(o32.get("con_dist"), 5), flag)

x0 = np.concatenate([q, v], axis=1).astype(np.float32)
us = np.zeros((nenv, 1, 0), dtype=np.float32)
xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
r64 = Oracle(mj, np.float64).rollout(x0, us); r32 = Oracle(mj, np.float32).rollout(x0, us).astype(np.float64)
e64 = np.abs(xs[:, 1] - r64[:, 1]).max(axis=1); e32 = np.abs(xs[:, 1] - r32[:, 1]).max(axis=1)
for k in range(nenv):
    if min(e64[k], e32[k]) > 1e-5: print("env", k, "e64", e64[k], "e32", e32[k])
