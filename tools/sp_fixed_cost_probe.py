"""Fixed cost of the split-operand products: time vs K at the benchmark's M, N (the slope is the main loop, the intercept is
launch + prologue + epilogue)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N = 30000, 320
out = torch.empty((M, N), device=dev)
for K in (16, 64, 320, 640, 1280, 2560):
    a = ops.sp_split_rows(torch.randn((M, K), device=dev))
    b = ops.sp_split_rows(torch.randn((N, K), device=dev) * 0.05)
    t0 = time_kernel(lambda: ops.sp_gemm_nt(a, b, out=out), iters=30)
    t1 = time_kernel(lambda: ops.sp_gemm_nt(a, b, act="relu", out=out), iters=30)
    t2 = time_kernel(lambda: ops.sp_gemm_nt_split(a, b, act="relu"), iters=30)
    print(f"NT  M={M} N={N} K={K:5d}: plain {1000 * t0:6.1f} us   relu {1000 * t1:6.1f} us   relu + split result {1000 * t2:6.1f} us")
for Kr in (480, 3000, 12000, 30000):
    x = ops.sp_split_rows(torch.randn((Kr, 1280), device=dev), scale_block=320)
    g = ops.sp_split_rows(torch.randn((Kr, 320), device=dev))
    dW = torch.empty((1280, 320), device=dev)
    t = time_kernel(lambda: ops.sp_gemm_tn(x, g, out=dW), iters=30)
    print(f"TN  M=1280 N=320 K={Kr:5d}: factors + product + reduce {1000 * t:6.1f} us")
