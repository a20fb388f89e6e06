"""Copies the summaries produced by tools/round_measure.sh (gpurun_out/) into profiles/ under round-1 names."""
import csv
import os
import shutil

os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
R = "r01"
shutil.copy("gpurun_out/bench_n1.json", f"profiles/{R}_bench_quadruped_n1.json")
rows = list(csv.reader(open("gpurun_out/stats/stats_kernel_stats.csv")))
with open(f"profiles/{R}_rocprofv3_kernel_stats_quadruped_4096x1000.csv", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline  (quadruped, 4096 envs x 1000 steps per launch; kernel names cut at 120 chars)\n")
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)
with open(f"profiles/{R}_rocprofv3_kernel_trace_amx_rollout.csv", "w") as f:
    f.write("# per-dispatch rows of the rollout kernel from the same rocprofv3 run (stats_kernel_trace.csv)\n")
    f.write("Kernel_Name,Start_Timestamp,End_Timestamp,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Workgroup_Size_X,Grid_Size_X,Duration_ns\n")
    for r in csv.DictReader(open("gpurun_out/stats/stats_kernel_trace.csv")):
        if "amx_rollout_kernel" in r["Kernel_Name"]:
            f.write(",".join(["amx_rollout_kernel<32>", r["Start_Timestamp"], r["End_Timestamp"], r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"],
                              r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"], str(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))]) + "\n")
hdr = """# rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; one pass per group (tools/pmc_collect.sh):
# {FETCH_SIZE} {WRITE_SIZE} {SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU}
# {SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES}
# {SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32} {GRBM_GUI_ACTIVE}
# per dispatch of amx_rollout_kernel<32> (quadruped, 4096 envs x 1000 steps = 4,100,096 forward passes, 2048 waves); FETCH_SIZE/WRITE_SIZE in KiB;
# SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs
"""
open(f"profiles/{R}_rocprofv3_pmc_amx_rollout.csv", "w").write(hdr + open("gpurun_out/pmc_summary.csv").read())
for a, b in (("phase_quadruped.txt", f"{R}_phase_profile_v3_quadruped.txt"), ("phase_biped.txt", f"{R}_phase_profile_v3_biped.txt")):
    t = [l for l in open("gpurun_out/" + a) if "amdgpu.ids" not in l]
    open("profiles/" + b, "w").write("# tools/phase_profile.py with libamx_prof.so (-DAMX_PROFILE, s_memtime stamps; the instrumented build is ~5% slower)\n" + "".join(t))
print("profiles refreshed")
