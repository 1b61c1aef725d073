#!/bin/bash
# matrix-pipe utilisation of the headline product kernels from SQ counters (two rocprofv3 --pmc passes over
# tools/pmc_workload.py rmat30k):  tools/pmc_mfma.sh r03  ->  gpurun_out/<tag>/<tag>_pmc_mfma.json
set -u
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/$TAG/pmc_mfma; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/a -- python tools/pmc_workload.py rmat30k > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/b -- python tools/pmc_workload.py rmat30k > $O/b.log 2>&1
python - <<PY
import csv, glob, json
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for d in ("$O/a", "$O/b"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in vals.items():
    if not any(n in k for n in ("gemm_sp_nt_kernel", "gemm_sp_tn_kernel", "csr_gather_reduce_kernel")):
        continue
    e = {n: sum(v) / len(v) for n, v in c.items()}
    if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        cycles = e["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        e["kernel_cycles"] = cycles
        e["mfma_busy_fraction_all_simds"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)
    out[k[:110]] = e
json.dump(out, open("gpurun_out/$TAG/${TAG}_pmc_mfma.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
