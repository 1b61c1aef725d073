"""GEMM shapes of the QM9-sized configs (cfg-4: M = 1.15 M rows, H = 128) in both evaluation modes.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402


def t(fn, iters=5):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1000


dev = torch.device("cuda", 0)
M = 1151896
shapes = [("GRU  [M,128]x[128,384] NN", 128, 384, False), ("msg  [M,640]x[128,640]^T NT", 640, 128, True),
          ("dAgg [M,384]x[128,384]^T NT", 384, 128, True), ("dA   [M,128]x[640,128]^T NT", 128, 640, True)]
for name, K, N, tb in shapes:
    A = torch.randn((M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev) * 0.05
    out = torch.empty((M, N), device=dev)
    res = []
    for mode in ("fp32", "bf16x3"):
        ops.set_gemm_mode(mode)
        res.append(t(lambda: ops.gemm(A, B, trans_b=tb, out=out)))
    gb = (M * K + M * N) * 4 / 1e9
    print(f"{name:30s} fp32 {res[0]:8.1f} us  bf16x3 {res[1]:8.1f} us   ({gb:.2f} GB in+out -> {gb / 6e3 * 1e6:.0f} us at 6 TB/s)", flush=True)
    del A, B, out
