import os, sys, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_closed_forms import EQ_LOOP, _integrate
from test_gpu_kat import Gpu
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
mj = MjModel.from_xml_string(EQ_LOOP)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(4)
g = Gpu(mj, dev); o = Oracle(mj, np.float64); o32 = Oracle(mj, np.float32)
for e in range(4):
    q = _integrate(mj, mj.qpos0.copy(), 0.01 * rng.standard_normal(mj.nv), 1.0); v = 0.075 * rng.standard_normal(mj.nv); u = rng.uniform(-1, 1, mj.nu)
    for oo in (o, o32):
        oo.set_state(q, v, u); oo.forward()
    f = g.forward(q, v, u)
    ne = int(o.get("nefc")[0]); nv = mj.nv
    M, J, D, aref, a0 = o.get("qM", (nv, nv)), o.get("efc_J")[:ne*nv].reshape(ne, nv), o.get("efc_D")[:ne], o.get("efc_aref")[:ne], o.get("qacc_smooth")
    cost = lambda a: 0.5 * (a - a0) @ M @ (a - a0) + 0.5 * np.sum(D * (J @ a - aref) ** 2)
    grad = lambda a: M @ (a - a0) + J.T @ (D * (J @ a - aref))
    a_d = np.linalg.solve(M + J.T @ (D[:, None] * J), M @ a0 + J.T @ (D * aref))
    print(f"env {e}: nefc gpu {f['nefc']} oracle {ne}; cost dense {cost(a_d):.6f} f64 {cost(o.get('qacc')):.6f} f32 {cost(o32.get('qacc').astype(float)):.6f} gpu {cost(f['qacc']):.6f}; |grad| f32 {np.linalg.norm(grad(o32.get('qacc').astype(float))):.2e} gpu {np.linalg.norm(grad(f['qacc'])):.2e}; qacc_smooth err gpu {np.abs(f['qacc_smooth']-a0).max():.2e}; qacc err f32 {np.abs(o32.get('qacc')-a_d).max():.2e} gpu {np.abs(f['qacc']-a_d).max():.2e}; cond(H) {np.linalg.cond(M + J.T @ (D[:, None] * J)):.1e}")
