"""The K = 320 products of the headline stack (projection, Dense, Dense input gradient: [30000, 320] x [320, 320]) take 55-64 us
inside the step against ~36 us for the plain product (tools/sp_fixed_cost_probe.py): which epilogue feature costs what?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N = 30000, 320
out = torch.empty((M, N), device=dev)
for K in (320, 1280):
    a = ops.sp_split_rows(torch.randn((M, K), device=dev))
    b = ops.sp_split_rows(torch.randn((N, K), device=dev) * 0.05)
    saved = torch.tanh(torch.randn((M, N), device=dev))
    rows = [
        ("plain", lambda: ops.sp_gemm_nt(a, b, out=out)),
        ("relu", lambda: ops.sp_gemm_nt(a, b, act="relu", out=out)),
        ("tanh", lambda: ops.sp_gemm_nt(a, b, act="tanh", out=out)),
        ("relu + split result", lambda: ops.sp_gemm_nt_split(a, b, act="relu")),
        ("tanh + split result", lambda: ops.sp_gemm_nt_split(a, b, act="tanh")),
        ("tanh + dropout + split result (Dense forward)", lambda: ops.sp_gemm_nt_split(a, b, act="tanh", dropout=(0.1, 7))),
        ("relu + dropout + split result", lambda: ops.sp_gemm_nt_split(a, b, act="relu", dropout=(0.1, 7))),
        ("split result only (no fp32)", lambda: ops.sp_gemm_nt_split(a, b, act="relu", want_fp32=False)),
        ("x tanh'(saved) (Dense input gradient)", lambda: ops.sp_gemm_nt(a, b, act_grad=("tanh", saved), out=out)),
        ("x relu'(saved)", lambda: ops.sp_gemm_nt(a, b, act_grad=("relu", saved), out=out)),
    ]
    for name, fn in rows:
        print(f"K={K:5d} {name:48s} {1000 * time_kernel(fn, iters=30):7.1f} us")
