"""Can a collective's kernels run next to amx_rollout_kernel on the same GPU?  The biped rollout (4096 envs x 1000 steps: 256
workgroups, one per CU, each with ~159 KB of LDS and 8 waves of 226 VGPRs) is timed alone and with a stand-in "communication" kernel
(tools/probes/fake_comm.hip: 32 workgroups x 256 threads spinning for ~20 ms) on a second stream, launched just before or just
after it, for three register footprints of the stand-in."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ambersim_amd import mjx
from ambersim_amd.utils.io_utils import load_mj_model_from_file
from bench import make_inputs
dev = torch.device("cuda", 0)
C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)  # the probe is linked -no-hip-rt
lib = C.CDLL(os.path.join(ROOT, "tools", "probes", "libfake_comm.so"))
lib.fake_comm.argtypes = [C.c_void_p, C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
mj = load_mj_model_from_file("models/biped/biped.xml"); model = mjx.put_model(mj, dev)
x0, us = make_inputs(mj, 4096, 1000, 1, 2, dev)
sink = torch.zeros(16, device=dev)
side = torch.cuda.Stream()
mjx.rollout(model, x0, us); torch.cuda.synchronize()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0)
t_roll = min(timed(lambda: mjx.rollout(model, x0, us)) for _ in range(3))
cyc = 50_000_000  # s_memtime ticks (~2.4 GHz here): ~20 ms
def comm(vg, nb=32): lib.fake_comm(C.c_void_p(side.cuda_stream), nb, cyc, vg, C.c_void_p(sink.data_ptr()))
print(f"rollout alone {t_roll:.1f} ms")
for vg in (8, 24, 56, 112):
    t_comm = min(timed(lambda: comm(vg)) for _ in range(2))
    def comm_first(): comm(vg); mjx.rollout(model, x0, us)
    def roll_first(): mjx.rollout(model, x0, us); comm(vg)
    a = min(timed(comm_first) for _ in range(2)); b = min(timed(roll_first) for _ in range(2))
    print(f"stand-in with ~{vg} VGPRs x 256 threads x 32 workgroups: alone {t_comm:.1f} ms | launched before the rollout: both done after {a:.1f} ms | "
          f"after it: {b:.1f} ms   (serial would be {t_roll + t_comm:.1f}, perfect overlap {max(t_roll, t_comm):.1f})")
