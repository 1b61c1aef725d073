#!/bin/bash
# per-kernel time of the headline step: bench.py under rocprofv3 (no roofline probes), summary -> gpurun_out/<tag>/
set -u
TAG=${1:-r02c}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs > $O/bench.json 2> $O/err.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
tn=[x for x in rows if "gemm_sp_tn_kernel" in x["Name"]]
steps=int(tn[0]["Calls"])/4 if tn else 1
print("ms_per_step", r["ms_per_step"], "steps in trace", steps)
tot=0
for x in rows[:32]:
    us=float(x["TotalDurationNs"])/1e3/steps; tot+=us
    print(f"{x['Name'][:100]:100s} {float(x['Calls'])/steps:6.1f}/step {float(x['AverageNs'])/1e3:8.1f} us  {us:8.1f} us/step")
print("sum of listed", tot)
PY
