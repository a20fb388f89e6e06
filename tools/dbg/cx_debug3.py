import sys, os, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, os, numpy as np, torch, ctypes as C
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import TWO
from ambersim_amd import mjx, _lib
from ambersim_amd.mjcf import MjModel
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
tq = torch.tensor(q, dtype=torch.float32, device=dev); tv = torch.zeros((nenv, mj.nv), device=dev); tu = torch.zeros((nenv, 1), device=dev)
cd = torch.zeros((nenv, 4), device=dev)
out = _lib.AmxForwardOut(None, None, None, None, None, None, None, cd.data_ptr(), None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
np.save(sys.argv[1], cd.cpu().numpy()); np.save(sys.argv[1] + ".q.npy", q)
''' % ROOT
res = {}
for v in (2, 3, 4):
    env = dict(os.environ, AMX_LIBRARY=os.path.join(ROOT, f"ambersim_amd/csrc/libamx_cxdbg{v}.so"))
    subprocess.run([sys.executable, "-c", CODE, f"/tmp/cx{v}.npy"], env=env, check=True)
    res[v] = np.load(f"/tmp/cx{v}.npy")
q = np.load("/tmp/cx2.npy.q.npy")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import TWO
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
xml = TWO.format(pb="0 0 0.7", qb="", gb='type="box" size="0.3 0.2 0.25"', ja="<freejoint/>", g=-9.81, asset="")
mj = MjModel.from_xml_string(xml)
o = Oracle(mj, np.float64)
for k in (7, 10, 19, 26, 29, 0):
    o.set_state(q[k].astype(np.float32), np.zeros(mj.nv), np.zeros(0)); o.forward()
    dbg = o.get("cx_dbg")
    n = int(dbg[0])
    print(f"env {k}: oracle np {n} idx {dbg[1:5].astype(int)} depth {np.round(dbg[5:5+n], 5)}")
    print(f"        gpu idx {res[4][k].astype(int)} depth {np.round(np.concatenate([res[2][k], res[3][k]]), 5)}")
