"""Does the split GEMM's time depend on the operand VALUES?  Same shapes, same kernel, operands all zero / a constant /
relu-like half zero / dense N(0,1).  Run on the GPU box (TFGNN_GEMM_MODE is set here)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402


def t(fn, iters=30):
    for _ in range(5):
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
V, K, N = 30000, 1280, 320
out = torch.empty((V, N), device=dev)
gens = {
    "zeros": lambda s: torch.zeros(s, device=dev),
    "constant 1.0": lambda s: torch.ones(s, device=dev),
    "N(0,1), 50 % zeroed (relu-like)": lambda s: torch.relu(torch.randn(s, device=dev)),
    "N(0,1), 45 % of rows zero": lambda s: torch.randn(s, device=dev) * (torch.rand((s[0], 1), device=dev) > 0.45),
    "dense N(0,1)": lambda s: torch.randn(s, device=dev),
}
for mode in ("bf16x3", "fp32"):
    ops.set_gemm_mode(mode)
    for name, gen in gens.items():
        A = gen((V, K))
        B = gen((N, K)) if name in ("zeros", "constant 1.0") else torch.randn((N, K), device=dev) * 0.05
        print(f"{mode:7s} A = {name:34s} {t(lambda: ops.gemm(A, B, trans_b=True, out=out)):7.1f} us", flush=True)
