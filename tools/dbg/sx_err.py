"""Sphere-mesh draw of tests/test_convex.py: per-env contact distance and one-step error of the HIP path and of the fp32 oracle
against the fp64 oracle."""
import sys, os, numpy as np, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import SPHERE_BOX, ulp_sensitivity
from ambersim_amd import mjx, _lib
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
octa = "0.3 0 0  -0.3 0 0  0 0.3 0  0 -0.3 0  0 0 0.35  0 0 -0.25"
xml = SPHERE_BOX.format(pb="0 0 0.6", qa="", ja="<freejoint/>", g=-9.81, it=3).replace('type="box" size="0.5 0.4 0.3"', 'type="mesh" mesh="o"')
xml = xml.replace("<worldbody>", f'<asset><mesh name="o" vertex="{octa}"/></asset><worldbody>')
mj = MjModel.from_xml_string(xml)
model = mjx.put_model(mj, dev)
rng = np.random.default_rng(32)
nenv = 64
q = np.tile(mj.qpos0, (nenv, 1))
for k in range(nenv):
    quat = np.array([1.0, 0, 0, 0]) + 0.6 * rng.uniform(-1, 1, 4)
    q[k, 3:7] = quat / np.linalg.norm(quat)
    u = rng.standard_normal(3)
    q[k, 7:10] = u / np.linalg.norm(u) * rng.uniform(0.1, 0.45)
v = 0.3 * rng.standard_normal((nenv, mj.nv))
x0 = np.concatenate([q, v], axis=1).astype(np.float32)
us = np.zeros((nenv, 1, 0), dtype=np.float32)
tq, tv = torch.tensor(x0[:, :mj.nq], device=dev), torch.tensor(x0[:, mj.nq:], device=dev)
cd = torch.zeros((nenv, 1), device=dev)
fc = torch.zeros((nenv, mj.nv), device=dev)
out = _lib.AmxForwardOut(None, None, None, fc.data_ptr(), None, None, None, cd.data_ptr(), None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), torch.zeros((nenv, 1), device=dev).data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
o64, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
r64 = o64.rollout(x0, us); r32 = o32.rollout(x0, us).astype(np.float64)
sens = ulp_sensitivity(mj, x0, us)
s = np.maximum(1.0, np.abs(r64[:, 1]).max(axis=1))
eg, e32 = np.abs(xs[:, 1] - r64[:, 1]).max(axis=1) / s, np.abs(r32[:, 1] - r64[:, 1]).max(axis=1) / s
for k in range(nenv):
    o64.set_state(x0[k, :mj.nq], x0[k, mj.nq:], np.zeros(0), np.zeros(mj.nv)); o64.forward()
    o32.set_state(x0[k, :mj.nq], x0[k, mj.nq:], np.zeros(0), np.zeros(mj.nv)); o32.forward()
    d64, d32, dg = o64.get("con_dist")[0], float(o32.get("con_dist")[0]), float(cd[k, 0])
    f64 = o64.get("qfrc_constraint"); fg = fc[k].cpu().numpy().astype(np.float64); f32 = o32.get("qfrc_constraint").astype(np.float64)
    fs = max(1.0, np.abs(f64).max())
    if d64 < 0:
        print(f"env {k}: dist {d64:+.5f} gpu-64 {(dg - d64) * 1e7:+.1f}e-7 cpu32-64 {(d32 - d64) * 1e7:+.1f}e-7 | qfrc_constraint rel gpu {np.abs(fg - f64).max() / fs:.1e} cpu32 {np.abs(f32 - f64).max() / fs:.1e} (|f| {np.abs(f64).max():.0f}) | step err gpu {eg[k]:.1e} cpu32 {e32[k]:.1e} sens {sens[k]:.1e}")
