"""What the epilogue of the split-operand NT product costs at the benchmark's shapes (round 4): tfgnn_sp_gemm_nt_dropout with
K = 320 (projection / Dense) and K = 1280 (message product), activation none / relu / tanh, plain or split (SP16) output,
with and without the fused dropout.  Prints microseconds per launch (HIP events, 30 launches)."""
import itertools
import sys

import torch

sys.path.insert(0, ".")
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N = 30000, 320


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1000.0 * a.elapsed_time(b) / iters


g = torch.Generator().manual_seed(0)
for K in (320, 1280):
    A = ops.sp_split_rows(torch.randn((M, K), generator=g).to(dev), scale_block=320)
    B = ops.sp_split_rows(torch.randn((N, K), generator=g).to(dev) * 0.05)
    saved = torch.relu(torch.randn((M, N), generator=g)).to(dev)
    for act, split, drop in itertools.product((None, "relu", "tanh"), (False, True), (None, (0.1, 7))):
        if split:
            t = timeit(lambda: ops.sp_gemm_nt_split(A, B, act=act, dropout=drop))
        else:
            t = timeit(lambda: ops.sp_gemm_nt(A, B, act=act, dropout=drop))
        print(f"K={K:5d} act={str(act):5s} split_out={int(split)} dropout={int(drop is not None)}  {t:7.1f} us")
    for what, kw in (("grad relu'(saved)", dict(act_grad=("relu", saved))),
                     ("grad relu'(saved) x recomputed mask", dict(act_grad=("relu", saved), dropout=(0.1, 7))),
                     ("grad tanh'(saved*0.9) x recomputed mask", dict(act_grad=("tanh", saved), dropout=(0.1, 7), saved_scale=0.9)),
                     ("grad relu mask-from-saved", dict(act_grad=("relu", saved), dropout=(0.1, 0xFFFFFFFFFFFFFFFF), saved_scale=0.9))):
        t = timeit(lambda: ops.sp_gemm_nt(A, B, **kw))
        print(f"K={K:5d} {what:45s} {t:7.1f} us")
