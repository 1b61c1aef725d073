#!/bin/bash
# Round-6 evidence in one gpurun call (final sources): default bench, per-kernel stats of all five workloads, FETCH / WRITE and
# MFMA-busy counters (separate rocprofv3 --pmc passes, never combined with the sys / hip / hsa trace domains), kernel resources.
#   tools/r06_profile.sh [stage ...]     stages: bench stats pmc mfma res   (default: all)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
STAGES=${@:-bench stats pmc mfma res}
WLS="rmat30k rgat qm9-ggnn qm9-edgemlp arxiv-rgin"
for S in $STAGES; do case $S in
  bench)
    python bench.py > $O/bench_default.out 2> $O/bench_default.err
    grep "^BENCH_DETAIL " $O/bench_default.out | sed 's/^BENCH_DETAIL //' > $O/r06_bench_default_detail.json
    tail -1 $O/bench_default.out > $O/r06_bench_default.json
    wc -c $O/r06_bench_default.json ;;
  stats)
    tools/step_profile.sh r06/stats_rmat30k "" 20 > $O/stats_rmat30k.txt 2>&1
    cp $O/stats_rmat30k/kernel_stats.csv $O/r06_kernel_stats.csv 2>/dev/null
    cp $O/stats_rmat30k/bench.json $O/r06_bench_under_rocprof.json 2>/dev/null
    for W in rgat qm9-ggnn qm9-edgemlp arxiv-rgin ppi; do
      ST=5; [ $W = rgat ] && ST=10; [ $W = ppi ] && ST=20
      tools/step_profile.sh r06/stats_$W $W $ST > $O/stats_$W.txt 2>&1
      cp $O/stats_$W/kernel_stats.csv $O/r06_kernel_stats_$W.csv 2>/dev/null
    done
    head -14 $O/stats_rmat30k.txt ;;
  pmc) tools/pmc_collect.sh r06 $WLS > $O/pmc_collect.log 2>&1; grep -A3 '"gather_sp"\|"gemm_sp_nt"' $O/r06_pmc_traffic_rmat30k.json | head -12 ;;
  mfma) tools/pmc_mfma.sh r06 $WLS > $O/pmc_mfma.log 2>&1; tail -16 $O/pmc_mfma.log ;;
  res) python tools/kernel_resources.py > $O/r06_kernel_resources.txt 2>&1; tail -3 $O/r06_kernel_resources.txt ;;
esac; done
