"""Gather probe at cfg-2 shape: TFGNN_GATHER_SLICED=0/1 (L2-resident feature windows per XCD)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402


def t(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


dev = torch.device("cuda", 0)
V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
A = torch.empty((V * L, H), device=dev)
rs = g.array(ops.G_INVDEG_BY_DST)
ews = g.array(ops.G_INVDEG_EDGE_BY_SRC)
Y = torch.randn((V * L, H), device=dev)
out = torch.empty((V, H), device=dev)
bytes_alg = E * (4 * H + 4) + (V * L + 1) * 4 + V * L * 4 + V * L * H * 4
print("TFGNN_GATHER_SLICED =", os.environ.get("TFGNN_GATHER_SLICED", "auto"))
ms = t(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, out=A))
print(f"fwd  aggregate (rows (v,l), src X 38 MB)   {ms*1000:7.1f} us  {bytes_alg/ms/1e6:8.0f} GB/s algorithmic")
ms = t(lambda: ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, X, edge_weight=ews, out=A))
print(f"bwd  aggregate (rows (u,l), edge weights)  {ms*1000:7.1f} us")
ms = t(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Y, out=out))
print(f"node gather from [V*L,H] (154 MB source)   {ms*1000:7.1f} us")
ref = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs)
print("checksum", float(ref.double().sum()), float(ref.double().abs().sum()))

# what bounds the gather?  same edges, but every source row index folded into the first 1024 rows (4 x 1.3 MB:
# L2-resident on every XCD): the difference to the real run is what the cache misses cost
col = g.array(ops.G_COL_BY_DST)
small = (col % 1024).contiguous()
ms = t(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, col=small, row_scale=rs, out=A))
print(f"fwd  aggregate, sources folded into 1024 rows (L2 hits) {ms*1000:7.1f} us")
onecol = torch.zeros_like(col)
ms = t(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, col=onecol, row_scale=rs, out=A))
print(f"fwd  aggregate, every edge reads row 0                  {ms*1000:7.1f} us")
