import sys, os, numpy as np, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["AMX_LIBRARY"] = os.path.join(ROOT, "ambersim_amd/csrc/libamx_cxdbg.so")
from test_convex import TWO
from ambersim_amd import mjx, _lib
from ambersim_amd.mjcf import MjModel, _quat_to_mat
dev = torch.device("cuda", 0)
xml = TWO.format(pb="0 0 0.7", qb="", gb='type="box" size="0.3 0.2 0.25"', ja="<freejoint/>", g=-9.81, asset="")
mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', 'iterations="3" ls_iterations="5"'))
model = mjx.put_model(mj, dev)
rng = np.random.default_rng(11); nenv = 32
q = np.tile(mj.qpos0, (nenv, 1))
for k in range(nenv):
    for a in (3, 10):
        quat = np.array([1.0, 0, 0, 0]) + 0.5 * rng.uniform(-1, 1, 4); q[k, a:a + 4] = quat / np.linalg.norm(quat)
    q[k, 7:10] = [0.3 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), rng.uniform(0.45, 0.95)]
q = q.astype(np.float32).astype(np.float64)
tq = torch.tensor(q, dtype=torch.float32, device=dev); tv = torch.zeros((nenv, mj.nv), device=dev); tu = torch.zeros((nenv, 1), device=dev)
cd = torch.zeros((nenv, 4), device=dev)
out = _lib.AmxForwardOut(None, None, None, None, None, None, None, cd.data_ptr(), None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
g = cd.cpu().numpy()

def hull(h):
    va, vn, fa, fn_, ea, en = mj.hull_vertadr[h], mj.hull_vertnum[h], mj.hull_faceadr[h], mj.hull_facenum[h], mj.hull_edgeadr[h], mj.hull_edgenum[h]
    faces = [mj.hull_facevert[mj.hull_face_vertadr[fa + f]: mj.hull_face_vertadr[fa + f] + mj.hull_face_vertnum[fa + f]] for f in range(fn_)]
    return mj.hull_vert[va:va + vn], faces, mj.hull_face_normal[fa:fa + fn_], mj.hull_edge.reshape(-1, 2)[ea:ea + en]

def sat(qk):
    g1, g2 = mj.pair_geom1[0], mj.pair_geom2[0]
    def pose(gm):
        b = mj.geom_bodyid[gm]; j = mj.body_jntadr[b]; a = mj.jnt_qposadr[j]
        return qk[a:a + 3], _quat_to_mat(qk[a + 3:a + 7] / np.linalg.norm(qk[a + 3:a + 7]))
    (p1, R1), (p2, R2) = pose(g1), pose(g2)
    vA, fA, nA, eA = hull(mj.pair_hull1[0]); vB, fB, nB, eB = hull(mj.pair_hull2[0])
    VA, VB = vA @ R1.T + p1, vB @ R2.T + p2
    best, bt, bs = 1e30, 0, 1
    axes = [R1 @ n for n in nA] + [R2 @ n for n in nB]
    for t, n in enumerate(axes):
        d1 = (VA @ n).max() - (VB @ n).min(); d2 = (VB @ n).max() - (VA @ n).min()
        pen = min(d1, d2)
        if pen < best: best, bt, bs = pen, t, (-1 if d1 > d2 else 1)
    t = len(axes)
    for ea in eA:
        for eb in eB:
            da = VA[ea[1]] - VA[ea[0]]; db = VB[eb[1]] - VB[eb[0]]; da /= np.linalg.norm(da); db /= np.linalg.norm(db)
            n = np.cross(da, db); c2 = n @ n
            if c2 >= 1e-6:
                n = n / np.sqrt(c2)
                hiA, loA, hiB, loB = (VA @ n).max(), (VA @ n).min(), (VB @ n).max(), (VB @ n).min()
                d1, d2 = hiA - loB, hiB - loA; pen = min(d1, d2); sg = -1 if d1 > d2 else 1
                supA, supB = (hiA if sg > 0 else loA), (loB if sg > 0 else hiB)
                if abs(VA[ea[0]] @ n - supA) <= 1e-5 and abs(VB[eb[0]] @ n - supB) <= 1e-5 and pen < best: best, bt, bs = pen, t, sg
            t += 1
    return bt, best

for k in range(nenv):
    bt, best = sat(q[k])
    flag = "" if int(g[k, 0]) == bt else "  <<<< axis differs"
    print(k, "gpu bt", int(g[k, 0]), "pen", g[k, 3], "ref/rf", g[k, 1], "fi*100+np", g[k, 2], "| ref bt", bt, "pen", round(best, 5), flag)
