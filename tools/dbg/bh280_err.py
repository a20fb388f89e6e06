import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden")); sys.path.insert(0, os.path.join(R, "tests"))
import make_golden
from ambersim_amd import mjx
from oracle.oracle import Oracle
name = sys.argv[1] if len(sys.argv) > 1 else "bh280"
mj, x0, us = make_golden.cases()[name]
golden = np.load(os.path.join(R, "tests", "golden", "rollouts.npz"))
ref = golden[name + "_xs"]
dev = torch.device("cuda", 0)
model = mjx.put_model(mj, dev)
nq, nv, N = mj.nq, mj.nv, us.shape[1]
o, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
rel = lambda a, b: np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(1.0, np.abs(b).max())
W = {}
for t in sorted({0, 1, N // 3, N // 2, N - 1}):
    q, v, u = ref[:, t, :nq], ref[:, t, nq:], us[:, t]
    d = mjx.make_data(model, q.shape[0]).replace(qpos=torch.tensor(q), qvel=torch.tensor(v), ctrl=torch.tensor(u))
    f = mjx.forward(model, d)
    s = mjx.step(model, d)
    for e in range(q.shape[0]):
        for oo in (o, o32):
            oo.set_state(q[e].astype(np.float32), v[e].astype(np.float32), u[e].astype(np.float32), np.zeros(nv)); oo.forward()
        for k, gv in (("qacc_smooth", f.qacc_smooth), ("qfrc_bias", f.qfrc_bias), ("qacc", f.qacc), ("qfrc_constraint", f.qfrc_constraint)):
            a = W.setdefault(k, [0.0, 0.0])
            a[0] = max(a[0], rel(gv[e].cpu().numpy(), o.get(k))); a[1] = max(a[1], rel(o32.get(k), o.get(k)))
        for oo in (o, o32):
            oo.set_state(q[e].astype(np.float32), v[e].astype(np.float32), u[e].astype(np.float32), np.zeros(nv)); oo.step()
        for k, gv in (("qpos", s.qpos), ("qvel", s.qvel)):
            a = W.setdefault("step " + k, [0.0, 0.0])
            a[0] = max(a[0], rel(gv[e].cpu().numpy(), o.get(k))); a[1] = max(a[1], rel(o32.get(k), o.get(k)))
print(name, "worst relative error vs fp64 oracle: GPU | fp32 CPU oracle")
for k, (a, b) in W.items(): print(f"  {k:18s} {a:.2e} | {b:.2e}")
