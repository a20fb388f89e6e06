#!/usr/bin/env python3
"""Checks the emitted gfx950 code for the one hazard the compiler cannot see for us: a DPP instruction written as inline
assembly (v_fmac_f32_dpp in the static eliminations) whose DPP source register (src0) was written by a VALU instruction
fewer than 2 wait states earlier.  Wait states: every instruction issued in between counts 1, `s_nop N` counts N + 1.
python tools/check_dpp_hazard.py [ambersim_amd/csrc/amx_gfx950.s]   (make -C ambersim_amd/csrc asm writes that file)"""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "ambersim_amd", "csrc", "amx_gfx950.s")
if len(sys.argv) <= 1:
    src = [os.path.join(root, "ambersim_amd", "csrc", f) for f in ("amx_kernels.hip", "amx_api.cpp", "amx_dev.h", "amx_specs_gen.h")]
    stale = [f for f in src if os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(path)] if os.path.exists(path) else src
    if stale:
        sys.exit(f"{os.path.relpath(path, root)} is older than {', '.join(os.path.basename(f) for f in stale)}: run `make -C ambersim_amd/csrc asm` first")
reg = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
def regs(tok):
    out = set()
    for m in reg.finditer(tok):
        if m.group(1): out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
bad = checked = 0
func = None
hist = []  # (written vgprs, wait states this instruction accounts for)
# (r05) second rule, same reason - inline assembly that consumes a compiler-emitted transcendental's result: on gfx940 / gfx950 a VALU read
# of a VGPR written by v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos needs 1 wait state (the compiler pads its own uses only)
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
last_trans = set()  # VGPRs written by the immediately preceding instruction, if it was a transcendental
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m: func, hist = m.group(1), []; continue
    t = line.strip()
    if not line.startswith("\t") or t.startswith((".", ";")) or not t: continue
    op, _, rest = t.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith("v_") and last_trans and not op.startswith(TRANS):
        srcs = set()
        for o in ops[1:]: srcs |= regs(o)
        if srcs & last_trans:
            bad += 1
            print(f"TRANS HAZARD in {func}: `{t}` reads {sorted(srcs & last_trans)} written by a transcendental 0 wait states earlier")
    last_trans = regs(ops[0]) if op.startswith(TRANS) else set()
    if "_dpp" in op and op.startswith("v_"):  # every DPP instruction: the inline-assembly ones (v_fmac, v_add, v_max) are the ones the compiler cannot pad
        checked += 1
        src0 = regs(ops[1].split()[0])
        ws = 0
        for wr, n in reversed(hist):
            if ws >= 2: break
            if wr & src0:
                bad += 1
                print(f"HAZARD in {func}: `{t}` reads {sorted(src0)} written {ws} wait states earlier")
                break
            ws += n
    if op == "s_nop": hist.append((set(), int(ops[0], 0) + 1))
    elif op.startswith("v_") and not op.startswith("v_cmp"):
        w = regs(ops[0])
        if op.startswith("v_permlane") and len(ops) > 1: w |= regs(ops[1])  # the swaps write both operands
        hist.append((w, 1))
    elif op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")): hist = []  # label targets: start over (conservative enough: producers are in-block)
    else: hist.append((set(), 1))
    hist = hist[-4:]
print(f"{checked} DPP instructions checked (inline assembly and compiler-emitted), {bad} hazards")
sys.exit(1 if bad else 0)
