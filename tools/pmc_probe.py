"""Workload for the rocprofv3 --pmc passes (HBM traffic of the two dominant kernels at cfg-2 shape).
Run under:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/pmc_probe.py
            rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir> -- python tools/pmc_probe.py
The streaming activation kernel (known bytes: read N*4, write N*4) calibrates the counters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
A = torch.empty((V * L, H), device=dev)
W = torch.randn((L * H, H), device=dev) * 0.05
out = torch.empty((V, H), device=dev)
big = torch.randn(V * L * H, device=dev)  # 153.6 MB
big2 = torch.empty_like(big)
rs = g.array(ops.G_INVDEG_BY_DST)
Wt = W.t().contiguous()
Wt_sp = ops.sp_split_cols(W)
X_sp = ops.sp_split_rows(X)
G_sp = ops.sp_split_rows(torch.randn((V, L * H), device=dev) * 1e-3, scale_block=H)
dW = torch.empty((L, H, H), device=dev)
for _ in range(5):
    ops.activation_forward("relu", big, out=big2)  # calibration: 153.6 MB read + 153.6 MB written
    ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, out=A)
    ops.set_gemm_mode("fp32")
    ops.gemm(A.view(V, L * H), W, act="relu", out=out)
    ops.set_gemm_mode("bf16x3")  # the same product on the split-operand kernel (W^T, as the layers call it)
    ops.gemm(A.view(V, L * H), Wt, trans_b=True, act="relu", out=out)
    # f16x2 mode: the gather writes the split operand, NT product forward, TN product for the weight gradient
    A_sp = ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, rows_per_operand_row=L)
    ops.sp_gemm_nt(A_sp, Wt_sp, act="relu", out=out)
    ops.sp_gemm_tn(G_sp, X_sp, out=dW, scatter=(H, H * H, 1, H))
torch.cuda.synchronize()
print("pmc probe done")
