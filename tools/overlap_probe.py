"""Do an MFMA-bound weight-gradient GEMM and a cache-bound gather overlap when issued on two HIP streams?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
ops.set_gemm_mode("bf16x3")
V, E, L, H = 30000, 900000, 4, 320
_, adjs = make_synthetic_batch(V, E, L, 8, seed=1)
g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
X = torch.randn((V, H), device=dev)
G2 = torch.randn((V, L * H), device=dev)
A = torch.empty((V * L, H), device=dev)
Wh = torch.randn((H, L * H), device=dev) * 0.05
dX = torch.empty((V, H), device=dev)
ews = g.array(ops.G_INVDEG_EDGE_BY_SRC)
side = torch.cuda.Stream()


def gather():
    ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, X, edge_weight=ews, out=A)


def dw():
    return ops.gemm(G2, X, trans_a=True)


def dx():
    ops.gemm(G2, Wh, trans_b=True, out=dX)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def seq():
    dw()
    gather()
    dx()


def par():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        dw()
    gather()
    dx()
    torch.cuda.current_stream().wait_stream(side)


print(f"dW {timeit(dw):.1f} us  gather {timeit(gather):.1f} us  dX {timeit(dx):.1f} us")
print(f"sequential dW + gather + dX: {timeit(seq):.1f} us")
print(f"dW on a second stream beside gather + dX: {timeit(par):.1f} us")
