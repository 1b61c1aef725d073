#!/bin/bash
# matrix-pipe utilisation of the product kernels from SQ counters (two rocprofv3 --pmc passes over tools/pmc_workload.py per
# workload; never combined with the sys / hip / hsa trace domains):
#   tools/pmc_mfma.sh r05 rmat30k arxiv-rgin qm9-ggnn  ->  gpurun_out/<tag>/<tag>_pmc_mfma.json
set -u
TAG=${1:-r05}; shift
WLS=${@:-rmat30k}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for WL in $WLS; do
  O=gpurun_out/$TAG/pmc_mfma_$WL; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/a -- python tools/pmc_workload.py $WL > $O/a.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/b -- python tools/pmc_workload.py $WL > $O/b.log 2>&1
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
done
python - <<PY
import csv, glob, json
from collections import defaultdict
out = {}
for wl in "$WLS".split():
    vals = defaultdict(lambda: defaultdict(list))
    for d in (f"gpurun_out/$TAG/pmc_mfma_{wl}/a", f"gpurun_out/$TAG/pmc_mfma_{wl}/b"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, c in vals.items():
        if not any(n in k for n in ("gemm_sp_nt_kernel", "gemm_sp_tn_kernel", "gemm_x3", "gemm_mfma_kernel", "csr_gather_reduce_")):
            continue
        e = {n: sum(v) / len(v) for n, v in c.items()}
        if e.get("GRBM_GUI_ACTIVE") and e.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            cycles = e["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            e["kernel_cycles"] = cycles
            e["mfma_busy_fraction_all_simds"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024.0)  # 256 CUs x 4 SIMDs
        res[k[:130]] = e
    out[wl] = res
json.dump(out, open("gpurun_out/$TAG/${TAG}_pmc_mfma.json", "w"), indent=1)
for wl, res in out.items():
    for k, e in res.items():
        print(wl, k[:90], "mfma busy %.3f" % e.get("mfma_busy_fraction_all_simds", float("nan")))
PY
