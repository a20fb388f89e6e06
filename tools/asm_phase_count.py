#!/usr/bin/env python3
"""Static instruction census per phase of a rollout kernel: the -DAMX_PROFILE build brackets every phase with s_memtime, so
the instructions between two of them (in program order; the specialised kernels are straight-line apart from the solver
loops) are that phase's.  python tools/asm_phase_count.py /tmp/amx_prof.s 'amx_rollout_kernelILi32E14Spec_quadruped'"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (l.startswith("_Z") and key in l and ":" in l.split(";")[0]))
funcs = {}
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        funcs[cur].append(l.strip())
def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op in ("s_waitcnt", "s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
def census(body):
    c = {}
    for ins in body:
        k = classify(ins); c[k] = c.get(k, 0) + 1
    return c
name = next(f for f in funcs if key in f)
body = funcs[name]
seg, segs = [], []
for ins in body:
    seg.append(ins)
    if ins.startswith("s_memtime"):
        segs.append(seg); seg = []
segs.append(seg)
print(f"{name}: {len(body)} instructions, {len(segs)} segments (program order)")
print(f"{'seg':>4} {'valu':>6} {'salu':>6} {'lds':>5} {'mfma':>5} {'vmem':>5} {'wait':>5} {'calls':>6}  swizzle/bperm dpp readlane")
tot = {}
for k, sg in enumerate(segs):
    c = census(sg)
    calls = sum(1 for i in sg if i.startswith("s_swappc"))
    sw = sum(1 for i in sg if i.startswith(("ds_swizzle", "ds_bpermute")))
    dpp = sum(1 for i in sg if "dpp" in i or "row_" in i or "quad_perm" in i)
    rl = sum(1 for i in sg if i.startswith(("v_readlane", "v_readfirstlane", "v_writelane")))
    for kk, v in c.items(): tot[kk] = tot.get(kk, 0) + v
    print(f"{k:>4} {c.get('valu',0):>6} {c.get('salu',0):>6} {c.get('lds',0):>5} {c.get('mfma',0):>5} {c.get('vmem',0):>5} {c.get('wait',0):>5} {calls:>6}  {sw:>6} {dpp:>6} {rl:>6}")
print("total", tot)
print("\nout-of-line functions:")
for f, b in funcs.items():
    if f != name and any(t in f for t in ("spd_solve", "newton_solve", "build_H", "plane_convex", "chol_solve")) and (("Spec_" not in f) or key.split("E")[-1][2:] in f or True):
        c = census(b)
        sw = sum(1 for i in b if i.startswith(("ds_swizzle", "ds_bpermute")))
        if len(b) > 50 and ("quadruped" in f or "biped" in f): print(f"  {f[:70]:70s} valu {c.get('valu',0):>5} salu {c.get('salu',0):>5} lds {c.get('lds',0):>5} (swz {sw}) mfma {c.get('mfma',0)} wait {c.get('wait',0)}")
