"""Timing of the split-operand (f16x2) NT product against the bf16x3 / fp32 kernels on the hot-path shapes.
Run on the GPU box: python tools/sp_probe.py"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


res = {}
for name, M, N, K, sb in [("fwd 30000x320x1280", 30000, 320, 1280, 320), ("fwd per-row scale", 30000, 320, 1280, 0)]:
    g = torch.Generator(device="cpu").manual_seed(0)
    A = torch.randn((M, K), device=dev)
    Bt = torch.randn((N, K), device=dev) * 0.05
    a_op = ops.sp_split_rows(A, scale_block=sb)
    b_op = ops.sp_split_rows(Bt)
    out = torch.empty((M, N), device=dev)
    t_sp = timeit(lambda: ops.sp_gemm_nt(a_op, b_op, act="relu", out=out))
    t_split = timeit(lambda: ops.sp_split_rows(A, scale_block=sb, out=a_op))
    row = {"f16x2_us": t_sp, "split_A_us": t_split}
    for mode in ("bf16x3", "fp32"):
        prev = ops.set_gemm_mode(mode)
        row[mode + "_us"] = timeit(lambda: ops.gemm(A, Bt, trans_b=True, act="relu", out=out))
        ops.set_gemm_mode(prev)
    flops = 2.0 * M * N * K
    row["f16x2_alg_tflops"] = flops / (t_sp[0] * 1e-6) / 1e12
    row["f16x2_mfma_frac_of_2.5PF"] = 3 * flops / (t_sp[0] * 1e-6) / 2.5e15
    res[name] = row
    print(name, json.dumps(row), flush=True)

# weight-gradient product and the SP16-writing gather at the cfg-2 shapes
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
gr = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
Gm = torch.randn((V, L * H), device=dev) * 1e-3
xs = ops.sp_split_rows(X)
gs = ops.sp_split_rows(Gm, scale_block=H)
dW = torch.empty((L, H, H), device=dev)
row = {"f16x2_tn_us": timeit(lambda: ops.sp_gemm_tn(gs, xs, out=dW, scatter=(H, H * H, 1, H)))}
for mode in ("bf16x3", "fp32"):
    prev = ops.set_gemm_mode(mode)
    row[mode + "_tn_us"] = timeit(lambda: ops.gemm(Gm, X, trans_a=True))
    ops.set_gemm_mode(prev)
row["absmax_X_us"] = timeit(lambda: ops.absmax(X))
row["split_X_us"] = timeit(lambda: ops.sp_split_rows(X, out=xs))
res["dW 1280x320 K=30000"] = row
print("dW", json.dumps(row), flush=True)
rs = gr.array(ops.G_INVDEG_BY_DST)
A = torch.empty((V * L, H), device=dev)
row = {"gather_fp32_us": timeit(lambda: ops.graph_gather(gr, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, out=A)),
       "gather_sp_us": timeit(lambda: ops.graph_gather_sp(gr, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, rows_per_operand_row=L)),
       "gather_sp_fixed_us": timeit(lambda: ops.graph_gather_sp(gr, ops.VIEW_BY_SRC_TYPED, X, fixed_inv_scale=ops.tensor_inv_scale(ops.absmax(X)), rows_per_operand_row=L))}
res["gather cfg-2"] = row
print("gather", json.dumps(row), flush=True)
json.dump(res, open("gpurun_out/sp_probe.json", "w"), indent=1)
