"""One-off long fuzz: seeded random models (tests/fuzz_models.py) on the GPU against the fp64 oracle, generic kernels at every
lane-group width.  python tools/dbg/fuzz_long.py [first seed] [count]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_models import random_inputs, random_model
from ambersim_amd import mjx
from oracle.oracle import Oracle
dev = torch.device("cuda", 0)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
bad, tested, skipped = [], 0, 0
for seed in range(first, first + count):
    mj = random_model(seed)
    if mj is None:
        skipped += 1; continue
    x0, us = random_inputs(mj, seed, 8, 5)
    x0, us = x0.astype(np.float32), us.astype(np.float32)
    ref = Oracle(mj, np.float64).rollout(x0, us); ref32 = Oracle(mj, np.float32).rollout(x0, us)
    if not (np.isfinite(ref).all() and np.isfinite(ref32).all()) or np.abs(ref).max() > 100:
        skipped += 1; continue
    for force in (None, "32", "64"):
        if force: os.environ["AMX_FORCE_G"] = force
        try:
            model = mjx.put_model(mj, dev)
        except NotImplementedError as e:
            continue
        finally:
            os.environ.pop("AMX_FORCE_G", None)
        xs = mjx.rollout(model, torch.tensor(x0), torch.tensor(us)).cpu().numpy().astype(np.float64)
        e1, f1 = rel(xs[:, 1], ref[:, 1]), rel(ref32[:, 1], ref[:, 1])
        e5, f5 = rel(xs[:, 5], ref[:, 5]), rel(ref32[:, 5], ref[:, 5])
        ok = np.isfinite(xs).all() and e1 <= max(1e-5, 3 * f1) and e5 <= max(5e-5, 5 * f5)
        tested += 1
        if not ok:
            bad.append((seed, force, model._h.query("lanes_per_env"), mj.nv, e1, f1, e5, f5))
print(f"tested {tested} (model, width) cases, skipped {skipped} seeds, failures {len(bad)}")
for b in bad: print("  seed %d force %s G %d nv %d  e1 %.2e (floor %.2e) e5 %.2e (floor %.2e)" % b)
