"""Can the weight-gradient product (MFMA-bound, off the critical path of the backward pass) hide behind the next layer's
gather (latency-bound)?  Times gather and sp_gemm_tn back to back on one stream and concurrently on two streams."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
rs = g.array(ops.G_INVDEG_BY_DST)
Xs = ops.sp_split_rows(X)
Gs = ops.sp_split_rows(torch.randn((V, L * H), device=dev) * 1e-3, scale_block=H)
Wt = ops.sp_split_cols(torch.randn((L * H, H), device=dev) * 0.05)
dW = torch.empty((L, H, H), device=dev)
out = torch.empty((V, H), device=dev)
side = torch.cuda.Stream()


def gather():
    return ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, rows_per_operand_row=L)


def tn():
    ops.sp_gemm_tn(Gs, Xs, out=dW, scatter=(H, H * H, 1, H))


def nt():
    ops.sp_gemm_nt(Gs, Wt, out=out)


def wall(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def both(main_fn, side_fn):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            side_fn()
        main_fn()
        torch.cuda.current_stream().wait_stream(side)
    return run


with torch.cuda.stream(side):
    tn(); nt()  # the side stream's workspace
torch.cuda.synchronize()
print(f"gather {wall(gather):.1f} us   tn {wall(tn):.1f} us   nt {wall(nt):.1f} us")
print(f"gather then tn, one stream : {wall(lambda: (gather(), tn())):.1f} us")
print(f"gather || tn, two streams  : {wall(both(gather, tn)):.1f} us")
print(f"gather then nt, one stream : {wall(lambda: (gather(), nt())):.1f} us")
print(f"gather || nt, two streams  : {wall(both(gather, nt)):.1f} us")
print(f"nt then tn, one stream     : {wall(lambda: (nt(), tn())):.1f} us")
print(f"nt || tn, two streams      : {wall(both(nt, tn)):.1f} us")
