"""Probe for cfg-5 (arxiv-rgin): what would the per-relation grouped products cost on split fp16 operands (3 piece products)
instead of bf16x3 (6)?  Times the EXISTING split-operand kernels on the ungrouped equivalents of the grouped shapes - same
rows, same K, one weight matrix for all rows (grouping changes which weight a row tile reads, not the work) - beside the
bf16x3 grouped kernels the workload runs today:
    forward / dX :  [452517, 512] x [512, 512]        (tfgnn_sp_gemm_nt, with and without the split-form output)
    dW           :  [452517, 512]^T x [452517, 512]   (tfgnn_sp_gemm_tn)
-> gpurun_out/grouped_sp_probe.json"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
R, H, L = 452517, 512, 40


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


gen = torch.Generator().manual_seed(0)
X = torch.randn((R, H), generator=gen).to(dev)
G = torch.randn((R, H), generator=gen).to(dev)
W = (torch.randn((L, H, H), generator=gen) * 0.05).to(dev)
# Zipf(1) group sizes like the workload's
w = 1.0 / torch.arange(1, L + 1, dtype=torch.float64)
sizes = (w / w.sum() * R).long()
sizes[0] += R - int(sizes.sum())
off_h = [0] + torch.cumsum(sizes, 0).tolist()
off_dev = torch.tensor(off_h, dtype=torch.int32, device=dev)
res = {}
ops.set_gemm_mode("bf16x3")
res["bf16x3 gemm_grouped_rows (forward, relu)"] = timeit(lambda: ops.gemm_grouped_rows(X, off_dev, off_h, W, act="relu"))
res["bf16x3 gemm_grouped_rows (dX, trans_b)"] = timeit(lambda: ops.gemm_grouped_rows(G, off_dev, off_h, W, trans_b=True))
res["bf16x3 gemm_grouped_k (dW)"] = timeit(lambda: ops.gemm_grouped_k(X, G, off_dev, off_h, L))
ops.set_gemm_mode("f16x2")
x_sp = ops.sp_split_rows(X)
g_sp = ops.sp_split_rows(G)
wt_sp = ops.sp_split_rows(W[0].t().contiguous())
out = torch.empty((R, H), device=dev)
res["f16x2 sp_gemm_nt ungrouped [R,512]x[512,512] relu"] = timeit(lambda: ops.sp_gemm_nt(x_sp, wt_sp, act="relu", out=out))
res["f16x2 split pass over the result (what an SP16-writing epilogue saves)"] = timeit(lambda: ops.sp_split_rows(out))
dw = torch.empty((H, H), device=dev)
res["f16x2 sp_gemm_tn ungrouped [R,512]^T x [R,512]"] = timeit(lambda: ops.sp_gemm_tn(x_sp, g_sp, out=dw))
res["f16x2 sp_split_rows [R,512] (operand conversion where no producer writes it)"] = timeit(lambda: ops.sp_split_rows(X))
res["gather_reduce compact-row expansion [170000,512] -> [R,512]"] = None
for k, v in res.items():
    print(f"{k:80s} {v if v is None else round(v, 3)} ms")
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
json.dump({"rows": R, "H": H, "groups": L, "ms": res}, open(ROOT / "gpurun_out" / "grouped_sp_probe.json", "w"), indent=1)
