#!/usr/bin/env python3
"""Static instruction census per phase of a rollout kernel: the -DAMX_PROFILE build brackets every phase with s_memtime, so
the instructions between two of them (in program order; the specialised kernels are straight-line apart from the solver
loops) are that phase's; the phase id is the offset of the counter the marker adds to (global_atomic_add_f32 ... offset:4*id).
Segments of one id are summed per pass through them; loops (Newton iterations, line-search passes) multiply at run time.
  ASM=1 SPECS=biped bash tools/devbuild.sh prof -DAMX_PROFILE
  python tools/asm_phase_count.py ambersim_amd/csrc/ab/amx_prof.s 'amx_rollout_kernelILi32E10Spec_bipedLi2'"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
funcs, cur = {}, None
for l in open(path):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1); funcs[cur] = []
    elif cur and l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        funcs[cur].append(l.strip())
def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_mov", "v_cndmask", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_swap")): return "vmove"
    if op.startswith(("v_cmp", "v_and", "v_or", "v_xor", "v_lshl", "v_lshr", "v_ashr", "v_bfe", "v_add_u", "v_sub_u", "v_add_co", "v_mul_lo", "v_mul_u", "v_mad_u", "v_mad_i", "v_add_nc", "v_sub_nc", "v_add3", "v_lshl_add", "v_min_u", "v_max_u", "v_min_i", "v_max_i", "v_not", "v_bfi", "v_cvt")): return "vint"
    if op.startswith("v_"): return "vfp"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")): return "vmem"
    if op in ("s_waitcnt", "s_nop"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
name = next(f for f in funcs if key in f)
body = funcs[name]
segs, seg = [], []
for ins in body:
    seg.append(ins)
    if ins.startswith("s_memtime"):
        segs.append(seg); seg = []
segs.append(seg)
# the id of segment k is the offset of the first atomic add of segment k + 1 (the marker that closes it)
ids = []
for k in range(len(segs)):
    nxt = segs[k + 1] if k + 1 < len(segs) else []
    a = next((i for i in nxt if i.startswith("global_atomic_add_f32")), None)
    m = re.search(r"offset:(\d+)", a) if a else None
    ids.append(int(m.group(1)) // 4 if m else (0 if a else -1))
cols = ("vfp", "vmove", "vint", "mfma", "lds", "salu", "wait", "vmem")
print(f"{name}: {len(body)} instructions, {len(segs)} segments (program order)")
print(f"{'seg':>4} {'id':>3} " + " ".join(f"{c:>6}" for c in cols) + "   dpp")
tot = {}
for k, sg in enumerate(segs):
    c = {}
    for ins in sg:
        t = classify(ins); c[t] = c.get(t, 0) + 1
    dpp = sum(1 for i in sg if "_dpp" in i or "row_" in i or "quad_perm" in i)
    print(f"{k:>4} {ids[k]:>3} " + " ".join(f"{c.get(x, 0):>6}" for x in cols) + f" {dpp:>5}")
    t = tot.setdefault(ids[k], {})
    for x in cols: t[x] = t.get(x, 0) + c.get(x, 0)
print("\nper phase id (one pass through each of its segments):")
for i in sorted(tot):
    t = tot[i]
    print(f"{i:>4}     " + " ".join(f"{t.get(x, 0):>6}" for x in cols) + f"   valu {t.get('vfp', 0) + t.get('vmove', 0) + t.get('vint', 0)}")
