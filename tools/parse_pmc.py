"""Turn the rocprofv3 --pmc CSVs of tools/pmc_probe.py into profiles/<tag>_pmc_traffic.json.

usage: python tools/parse_pmc.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE counts 128-byte
requests as 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, HBM section): the streaming
calibration kernel in the probe (known 153.6 MB read / written) gives the correction factors that
are applied to the other kernels (and are stored in the JSON)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(d, counter):
    per_kernel = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                per_kernel[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per_kernel.items()}


def pick(d, needle):
    needles = (needle,) if isinstance(needle, str) else needle
    for k, v in d.items():
        if all(n in k for n in needles):
            return v
    return None


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
known = 30000 * 4 * 320 * 4.0  # bytes read and bytes written by the calibration kernel
cal_f = pick(fetch, "act_forward_kernel") * 1024.0
cal_w = pick(write, "act_forward_kernel") * 1024.0
kf, kw = known / cal_f, known / cal_w
res = {"calibration": {"known_bytes_each_way": known, "fetch_raw_bytes": cal_f, "write_raw_bytes": cal_w,
                       "fetch_factor": kf, "write_factor": kw}}
# traffic keys of bench.py's roofline blocks -> substrings of the kernel name
NEEDLES = (
    ("gather", ("csr_gather_reduce_kernel", ", 0, false>")), ("gather_sp", ("csr_gather_reduce_kernel", ", 0, true>")),
    # (round 6: buckets of <= 1.5 edges on average - molecule batches - go two to a lane group: another kernel name)
    ("gather", ("csr_gather_reduce_multi_kernel", ", false, 2, 1>")), ("gather_sp", ("csr_gather_reduce_multi_kernel", ", true, 2, 1>")),
    ("gather_heads", ("csr_gather_reduce_kernel", ", 1, false>")),
    ("gemm", "gemm_mfma_kernel"), ("gemm_bf16x3", "gemm_x3s_kernel"), ("gemm_bf16x3_pipelined", "gemm_x3p_kernel"),
    ("gemm_sp_nt", "gemm_sp_nt_kernel"), ("gemm_sp_tn", "gemm_sp_tn_kernel"), ("gemm_stream", "gemm_x3k_kernel"),
)
for name, needle in NEEDLES:
    if pick(fetch, needle) is None or pick(write, needle) is None or name in res:
        continue
    f, w = pick(fetch, needle) * 1024.0, pick(write, needle) * 1024.0
    res[name] = {"kernel": [k for k in fetch if all(n in k for n in ((needle,) if isinstance(needle, str) else needle))][0][:160],
                 "fetch_raw_bytes": f, "write_raw_bytes": w, "hbm_bytes_per_launch": f * kf + w * kw}
# which kernel sources these counters belong to: bench.py compares it with the sources it runs (a traffic number measured on an
# older kernel is reported as such)
import hashlib  # noqa: E402
import os  # noqa: E402

_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf2_gnn_amd", "csrc")
_h = hashlib.sha256()
for _f in sorted(os.listdir(_csrc)):
    if _f.endswith((".hip", ".hpp")):
        _h.update(open(os.path.join(_csrc, _f), "rb").read())
res["_meta"] = {"kernel_source_sha16": _h.hexdigest()[:16]}
res["all_kernels_raw_kib"] = {k[:120]: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)} for k in sorted(set(fetch) | set(write))}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "all_kernels_raw_kib"}, indent=1))
