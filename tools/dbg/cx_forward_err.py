"""Where the convex-convex one-step error comes from: qacc_smooth, qfrc_constraint and qacc of amx_forward against the fp64 and
the fp32 oracle on the box-box draw of tests/test_convex.py, with the test's solver options (3/5) and with a converged solve (50/50)."""
import sys, os, numpy as np, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import TWO
from ambersim_amd import mjx, _lib
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
q = q.astype(np.float32); v = v.astype(np.float32)
for its in ("3/5", "50/50"):
    it, ls = its.split("/")
    mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', f'iterations="{it}" ls_iterations="{ls}"'))
    model = mjx.put_model(mj, dev)
    tq, tv = torch.tensor(q, device=dev), torch.tensor(v, device=dev)
    tu = torch.zeros((nenv, 1), device=dev)
    outs = {n: torch.zeros((nenv, mj.nv), device=dev) for n in ("qacc", "qacc_smooth", "qfrc_bias", "qfrc_constraint")}
    out = _lib.AmxForwardOut(outs["qacc"].data_ptr(), outs["qacc_smooth"].data_ptr(), outs["qfrc_bias"].data_ptr(), outs["qfrc_constraint"].data_ptr(), None, None, None, None, None)
    _lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
    torch.cuda.synchronize()
    print("solver", its)
    o64, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
    for k in range(nenv):
        o64.set_state(q[k], v[k], np.zeros(0)); o64.forward()
        o32.set_state(q[k], v[k], np.zeros(0)); o32.forward()
        line = []
        for n in ("qacc_smooth", "qfrc_constraint", "qacc"):
            r = o64.get(n); g = outs[n][k].cpu().numpy().astype(np.float64); r32 = o32.get(n).astype(np.float64)
            s = max(1.0, np.abs(r).max())
            line.append(f"{n} gpu {np.abs(g - r).max() / s:.1e} cpu32 {np.abs(r32 - r).max() / s:.1e} (|ref| {np.abs(r).max():.0f})")
        print(k, " | ".join(line))

# forward against one rollout step of the same library (3/5): (v1 - v0) / dt is qacc
mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', 'iterations="3" ls_iterations="5"'))
model = mjx.put_model(mj, dev)
x0 = np.concatenate([q, v], axis=1)
xs = mjx.rollout(model, torch.tensor(x0), torch.zeros((nenv, 1, 0))).cpu().numpy().astype(np.float64)
qa_roll = (xs[:, 1, mj.nq:] - x0[:, mj.nq:].astype(np.float64)) / mj.opt.timestep
out = _lib.AmxForwardOut(outs["qacc"].data_ptr(), None, None, None, None, None, None, None, None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
qa_fwd = outs["qacc"].cpu().numpy().astype(np.float64)
o64 = Oracle(mj, np.float64)
r = o64.rollout(x0, np.zeros((nenv, 1, 0)))
qa_or = (r[:, 1, mj.nq:] - x0[:, mj.nq:].astype(np.float64)) / mj.opt.timestep
for k in range(nenv):
    o64.set_state(q[k], v[k], np.zeros(0), np.zeros(mj.nv)); o64.forward()
    print(k, "fwd-vs-roll (gpu)", f"{np.abs(qa_fwd[k] - qa_roll[k]).max():.2e}", "oracle fwd-vs-roll", f"{np.abs(o64.get('qacc') - qa_or[k]).max():.2e}", "gpu roll vs oracle roll", f"{np.abs(qa_roll[k] - qa_or[k]).max():.2e}")
