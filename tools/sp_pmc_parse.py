"""Summarise the rocprofv3 counter CSVs of tools/sp_pmc_workload.py: per group of 5 dispatches of gemm_sp_nt_kernel."""
import csv
import glob
import json
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "gemm_sp_nt_kernel" in r["Kernel_Name"]]
by_disp = defaultdict(dict)
for r in rows:
    by_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    by_disp[int(r["Dispatch_Id"])]["_dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) if "End_Timestamp" in r else 0.0
ids = sorted(by_disp)
names = ["full random", "full zeros", "mfma-only random", "mfma-only zeros"]
out = {}
for gi, name in enumerate(names):
    grp = [by_disp[i] for i in ids[gi * 5 + 1:gi * 5 + 5]]
    if not grp:
        continue
    avg = {k: sum(g.get(k, 0.0) for g in grp) / len(grp) for k in grp[0]}
    out[name] = avg
print(json.dumps(out, indent=1))
json.dump(out, open(sys.argv[2], "w"), indent=1)
