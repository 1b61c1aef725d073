#!/bin/bash
# HBM-side traffic (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) of the roofline kernels of the given workloads:
#   tools/pmc_collect.sh r03 rmat30k rgat ...   ->  profiles/<tag>_pmc_traffic_<workload>.json + the counter CSVs behind it
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for WL in "$@"; do
  O=gpurun_out/$TAG/pmc_$WL; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python tools/pmc_workload.py $WL > $O/fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python tools/pmc_workload.py $WL > $O/write.log 2>&1
  python tools/parse_pmc.py $O/fetch $O/write $O/traffic.json > $O/parse.log 2>&1 || tail -3 $O/parse.log
  for side in fetch write; do
    f=$(find $O/$side -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" gpurun_out/$TAG/${TAG}_pmc_${side}_${WL}_counter_collection.csv
  done
  cp $O/traffic.json gpurun_out/$TAG/${TAG}_pmc_traffic_$WL.json 2>/dev/null
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
  echo "== $WL"; cat $O/traffic.json 2>/dev/null | head -60
done
