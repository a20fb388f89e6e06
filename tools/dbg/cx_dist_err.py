"""Contact distances of the convex-convex path against the fp64 oracle in units of 1e-7 (box-box draw of tests/test_convex.py),
next to the fp32 oracle's."""
import sys, os, numpy as np, torch, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_convex import TWO
from ambersim_amd import mjx, _lib
from ambersim_amd.mjcf import MjModel
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
xml = TWO.format(pb="0 0 0.7", qb="", gb='type="box" size="0.3 0.2 0.25"', ja="<freejoint/>", g=-9.81, asset="")
mj = MjModel.from_xml_string(xml.replace('iterations="50" ls_iterations="50"', 'iterations="3" ls_iterations="5"'))
model = mjx.put_model(mj, dev)
rng = np.random.default_rng(11)
nenv = 32
q = np.tile(mj.qpos0, (nenv, 1))
for k in range(nenv):
    for a in (3, 10):
        quat = np.array([1.0, 0, 0, 0]) + 0.5 * rng.uniform(-1, 1, 4)
        q[k, a:a + 4] = quat / np.linalg.norm(quat)
    q[k, 7:10] = [0.3 * rng.uniform(-1, 1), 0.3 * rng.uniform(-1, 1), rng.uniform(0.6, 0.95)]
v = 0.3 * rng.standard_normal((nenv, mj.nv))
q = q.astype(np.float32); v = v.astype(np.float32)
tq, tv = torch.tensor(q, device=dev), torch.tensor(v, device=dev)
tu = torch.zeros((nenv, 1), device=dev)
cd = torch.zeros((nenv, 4), device=dev)
out = _lib.AmxForwardOut(None, None, None, None, None, None, None, cd.data_ptr(), None)
_lib.check(_lib.lib().amx_forward(model._h.ptr, nenv, tq.data_ptr(), tv.data_ptr(), tu.data_ptr(), None, C.byref(out), None))
torch.cuda.synchronize()
o64, o32 = Oracle(mj, np.float64), Oracle(mj, np.float32)
for k in range(nenv):
    o64.set_state(q[k], v[k], np.zeros(0)); o64.forward()
    o32.set_state(q[k], v[k], np.zeros(0)); o32.forward()
    ref = o64.get("con_dist"); g = cd[k].cpu().numpy().astype(np.float64); r32 = o32.get("con_dist").astype(np.float64)
    act = ref < 0
    print(k, "gpu-ref64 [1e-7]:", np.round((g - ref)[act] * 1e7, 1), " cpu32-ref64:", np.round((r32 - ref)[act] * 1e7, 1))
