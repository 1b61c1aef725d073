#!/bin/bash
# all BASELINE configs through bench.py + rocprof kernel stats + PMC traffic of the headline kernels (run on the GPU box)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02
mkdir -p $O
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_rmat30k.json 2> $O/bench_rmat30k.err
for w in rgat qm9-ggnn qm9-edgemlp arxiv-rgin; do
  timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-mode > $O/bench_under_rocprof.json 2> $O/prof.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python tools/pmc_probe.py > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python tools/pmc_probe.py > $O/pmc_write.log 2>&1
python tools/parse_pmc.py $O/pmc_fetch $O/pmc_write $O/r02_pmc_traffic_rmat30k.json > $O/pmc_parse.log 2>&1
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
tail -c 600 $O/bench_rmat30k.json; for w in rgat qm9-ggnn qm9-edgemlp arxiv-rgin; do echo; head -c 400 $O/bench_$w.json; tail -3 $O/bench_$w.err; done
