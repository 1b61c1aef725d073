"""What bounds the (node, type)-bucket gather at cfg-2 shape?  Same kernel, same edges, source rows remapped so that the
distinct source rows fit (a) one XCD's L2 (2048 rows = 2.6 MB), (b) the aggregate L2 (16384 rows = 21 MB), (c) as is
(30000 rows = 38 MB, Infinity Cache).  Also times the weight-gradient product and its factor pass."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
rs = g.array(ops.G_INVDEG_BY_DST)
col = g.array(ops.G_COL_BY_DST)
A = torch.empty((V * L, H), device=dev)
for label, c in (("as is (30000 rows)", None), ("16384 distinct rows", (col % 16384).contiguous()), ("2048 distinct rows", (col % 2048).contiguous()),
                 ("sorted sources (col = position * V / E)", (torch.arange(col.numel(), device=dev, dtype=torch.int64) * V // col.numel()).int())):
    ms_sp = time_kernel(lambda: ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, X, col=c, row_scale=rs, rows_per_operand_row=L))
    ms = time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, col=c, row_scale=rs, out=A))
    print(f"gather {label:42s}: SP16 out {1000 * ms_sp:7.1f} us   fp32 out {1000 * ms:7.1f} us")
Xs = ops.sp_split_rows(X)
Gs = ops.sp_split_rows(torch.randn((V, L * H), device=dev) * 1e-3, scale_block=H)
dW = torch.empty((L, H, H), device=dev)
print(f"sp_gemm_tn (factors + product + reduce): {1000 * time_kernel(lambda: ops.sp_gemm_tn(Gs, Xs, out=dW, scatter=(H, H * H, 1, H))):7.1f} us")
