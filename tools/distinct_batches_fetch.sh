#!/bin/bash
# the timed step re-reads the same batch every step - are its source rows warm in the Infinity Cache "by
# construction"?  FETCH_SIZE of the gather kernel over the real timed steps with the same batch every step and with 8 different
# batches taking turns (bench.py --distinct-batches), plus the step times of both runs:
#   tools/distinct_batches_fetch.sh r05  ->  gpurun_out/<tag>/<tag>_distinct_batches.json
set -u
TAG=${1:-r05}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
for K in 1 8; do
  O=gpurun_out/$TAG/distinct_$K; rm -rf $O; mkdir -p $O
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p -- python bench.py --distinct-batches $K --steps 10 --warmup 2 --no-settle --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs > $O/bench.json 2> $O/err.log
  python bench.py --distinct-batches $K --steps 40 --no-cpu-baseline --no-alt-mode --no-roofline --no-other-configs > $O/bench_unprofiled.json 2>> $O/err.log
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
done
python - <<PY
import csv, glob, json
out = {"what": "rocprofv3 --pmc FETCH_SIZE over bench.py steps (raw KiB per dispatch as the counter reports them; the gfx950 factor of ~2 for wide reads is NOT applied - the two columns compare like with like), and the un-profiled step time of the same configuration"}
for K in (1, 8):
    vals = {}
    for f in glob.glob(f"gpurun_out/$TAG/distinct_{K}/p/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                vals.setdefault(r["Kernel_Name"][:80], []).append(float(r["Counter_Value"]))
    e = {}
    for k, v in vals.items():
        if any(n in k for n in ("csr_gather_reduce_kernel", "gemm_sp_nt_kernel", "gemm_sp_tn_kernel")):
            e[k] = {"dispatches": len(v), "mean_fetch_kib": sum(v) / len(v)}
    try:
        ms = json.loads(open(f"gpurun_out/$TAG/distinct_{K}/bench_unprofiled.json").read().strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        ms = None
    out[f"distinct_batches_{K}"] = {"ms_per_step": ms, "kernels": e}
json.dump(out, open("gpurun_out/$TAG/${TAG}_distinct_batches.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
