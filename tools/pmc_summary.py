"""Summarise rocprofv3 --pmc passes: mean counter value per dispatch of one kernel.
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv (searched recursively)> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "amx_rollout_kernel"
tot, cnt = defaultdict(float), defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if kern not in r.get("Kernel_Name", ""):
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]].add((f, r.get("Dispatch_Id")))
print("counter,mean_per_dispatch,dispatches")
for k in sorted(tot):
    print(f"{k},{tot[k] / len(cnt[k]):.6g},{len(cnt[k])}")
