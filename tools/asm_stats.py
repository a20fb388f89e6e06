#!/usr/bin/env python3
"""Kernel resource usage of every amx kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage through
`make -C ambersim_amd/csrc asm`): VGPRs, SGPR / VGPR spills, scratch.  python tools/asm_stats.py [filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-B", "-C", os.path.join(root, "ambersim_amd", "csrc"), "asm"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':<58}{'VGPR':>6}{'AGPR':>6}{'SGPRspill':>10}{'VGPRspill':>10}{'scratch':>8}{'occ':>4}")
for k, r in rows.items():
    if flt in k:
        print(f"{k:<58}{r.get('VGPRs', 0):>6}{r.get('AGPRs', 0):>6}{r.get('SGPRs Spill', 0):>10}{r.get('VGPRs Spill', 0):>10}{r.get('ScratchSize', 0):>8}{r.get('Occupancy', 0):>4}")
