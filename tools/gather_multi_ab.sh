#!/bin/bash
# A/B of the short-row gather shapes (TFGNN_GATHER_MULTI: 1 = one row per lane group, 0 = automatic, 10 R + U = forced):
# step time of the workloads whose gathers walk short rows.   tools/gather_multi_ab.sh [tag] [workloads...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=${1:-r06n}; shift
WLS=${@:-qm9-edgemlp qm9-ggnn arxiv-rgin rmat30k rgat}
O=gpurun_out/$TAG; mkdir -p $O
for W in $WLS; do
  ST=4; [ $W = rmat30k ] && ST=30; [ $W = rgat ] && ST=15
  for M in 1 0 21 1 0; do
    R=$(TFGNN_GATHER_MULTI=$M timeout 300 python bench.py --workload $W --steps $ST --warmup 2 --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs 2>$O/err_${W}_$M.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$W multi=$M ms_per_step=$R" | tee -a $O/gather_multi_ab.txt
  done
done
