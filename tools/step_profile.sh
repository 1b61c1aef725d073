#!/bin/bash
# per-kernel time of a step: bench.py under rocprofv3 (no roofline probes), summary -> gpurun_out/<tag>/
#   tools/step_profile.sh <tag> [workload] [steps]      (default: the headline workload, 20 steps)
set -u
TAG=${1:-r02c}
WL=${2:-}
STEPS=${3:-20}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py ${WL:+--workload $WL} --steps $STEPS --warmup 2 --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs --no-settle > $O/bench.json 2> $O/err.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
r=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
steps=float(r["steps"]) + float(r["warmup"])
print("ms_per_step", r["ms_per_step"], "steps in trace", steps)
tot=0
for x in rows[:32]:
    us=float(x["TotalDurationNs"])/1e3/steps; tot+=us
    print(f"{x['Name'][:100]:100s} {float(x['Calls'])/steps:6.1f}/step {float(x['AverageNs'])/1e3:8.1f} us  {us:8.1f} us/step")
print("sum of listed", tot)
PY
