"""rocprofv3 kernel trace of `python bench.py` -> the average duration of the launches bench.py's roofline section
times live (the last 20 launches of the GEMM and of the gather kernel, issued back to back after the timed steps),
to set beside `roofline.ms_per_launch` of the same run.
  python tools/roofline_from_trace.py <kernel_trace.csv> <bench.json> <out.json>"""
import csv
import json
import sys

trace, bench, out = sys.argv[1:4]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
res = {}
for key, pat in (("gemm", "gemm_x3s_kernel"), ("gemm_fp32", "gemm_mfma_kernel"), ("gather", "csr_gather_reduce_kernel")):
    sel = [r for r in rows if pat in r["Kernel_Name"]]
    if len(sel) < 20:
        continue
    last = sel[-20:]
    res[key] = {
        "kernel": last[0]["Kernel_Name"],
        "launches": len(last),
        "avg_us_in_trace": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e3,
    }
b = json.loads(open(bench).read().strip().splitlines()[-1])
for name in ("roofline", "roofline_secondary"):
    r = b.get(name)
    if r:
        res.setdefault("bench_live", {})[r["kernel"][:40]] = {"ms_per_launch": r["ms_per_launch"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
