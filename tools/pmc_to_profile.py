#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>_summary.csv (tools/pmc_collect.sh) -> profiles/rNN_rocprofv3_pmc_<tag>.csv: the per-dispatch counter
means plus the derived figures bench.py and DESIGN.md quote.  python tools/pmc_to_profile.py quadruped|biped [round]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r05"
G, waves, passes = 32, 2048, 1001  # 4096 envs x 1000 steps, two environments per wave
if tag == "biped_rest" and rnd < "r05":
    G, waves = 64, 4096  # sixteen contact slots: one environment per wave (since r05: two contact rows per lane, two environments per wave - the dense form)
model_arg = "" if tag == "biped_stand" else f" --model models/{tag.split('_')[0]}/{tag}.xml"  # biped_stand is bench.py's default workload since r04
v = {}
for line in open(os.path.join(root, "gpurun_out", f"pmc_{tag}_summary.csv")):
    p = line.strip().split(",")
    try: v[p[0]] = float(p[1])
    except (ValueError, IndexError): pass
out = [f"# rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary{model_arg}; one pass per group (tools/pmc_collect.sh {tag})",
       f"# per dispatch of amx_rollout_kernel<{G}, Spec_{'biped' if tag.startswith('biped') else tag}> ({tag}, 4096 envs x 1000 steps, {waves} waves x {passes} forward passes); FETCH_SIZE/WRITE_SIZE in KiB;",
       "# SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_INSTS_* are wave instructions",
       "# derived: valu_busy_frac = 4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE/8); lds_busy_frac = SQ_LDS_IDX_ACTIVE / (256 CUs * GRBM_GUI_ACTIVE/8);",
       f"#          executed_fp32_flops = 64 lanes * (2 FMA + ADD + MUL + TRANS) (idle lanes count as work); *_per_wave_step = SQ_INSTS_* / ({waves} * {passes})",
       "counter,mean_per_dispatch"]
for k in sorted(v):
    out.append(f"{k},{v[k]:g}")
cyc = v["GRBM_GUI_ACTIVE"] / 8
out.append(f"valu_busy_frac,{4 * v['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.4f}")
out.append(f"lds_busy_frac,{v['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):.4f}")
out.append(f"executed_fp32_flops,{64 * (2 * v['SQ_INSTS_VALU_FMA_F32'] + v['SQ_INSTS_VALU_ADD_F32'] + v['SQ_INSTS_VALU_MUL_F32'] + v['SQ_INSTS_VALU_TRANS_F32']):g}")
if "SQ_LDS_BANK_CONFLICT" in v:
    out.append(f"lds_bank_conflict_frac,{v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.4f}")
if "SQC_ICACHE_MISSES" in v and v.get("SQC_ICACHE_REQ"):
    out.append(f"icache_miss_frac,{v['SQC_ICACHE_MISSES'] / v['SQC_ICACHE_REQ']:.2e}")
for k in ("VALU", "SALU", "LDS", "VMEM"):
    out.append(f"insts_{k.lower()}_per_wave_step,{v['SQ_INSTS_' + k] / (waves * passes):.1f}")
path = os.path.join(root, "profiles", f"{rnd}_rocprofv3_pmc_{tag}.csv")
open(path, "w").write("\n".join(out) + "\n")
print(path); print("\n".join(out[-7:]))
