"""How much does the per-(row, block) scaling of the A fragments (ABLK variant of gemm_sp_nt_kernel) cost at the headline
shape?  Same operands split with one scale per row (ABLK off) and with one per 320-column block (ABLK on)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N, K = 30000, 320, 1280
A = torch.randn((M, K), device=dev)
W = torch.randn((K, N), device=dev) * 0.05
Wt = ops.sp_split_cols(W)
out = torch.empty((M, N), device=dev)
res = {}
for name, sb in (("one scale per row", 0), ("one scale per 320-column block", 320)):
    a = ops.sp_split_rows(A, scale_block=sb)
    res[name] = [1e3 * time_kernel(lambda: ops.sp_gemm_nt(a, Wt, act="relu", out=out), iters=30) for _ in range(3)]
print(json.dumps(res, indent=1))
