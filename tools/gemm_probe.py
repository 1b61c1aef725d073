"""GEMM probe: where does the fp32 MFMA kernel lose time?  (run on the GPU box)
NOTE: the TFGNN_GEMM_DEBUG / X3_PROBE arms this script drives were part of the round-1 kernels (commit 058dff9) and have been
removed from the shipped sources; with the current library every setting times the full kernel.
  TFGNN_GEMM_DEBUG=0 full kernel | 1 no staging after the first tile | 2 no LDS fragment reads | 3 both
Results are wrong for debug != 0; only the time matters."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402


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
V, L, H = 30000, 4, 320
A = torch.randn((V, L * H), device=dev)
W = torch.randn((L * H, H), device=dev) * 0.05
X = torch.randn((V, H), device=dev)
G = torch.randn((V, L * H), device=dev)
Wh = torch.randn((H, L * H), device=dev) * 0.05
out = torch.empty((V, H), device=dev)
shapes = {
    "fwd  [V,1280]x[1280,320]": (lambda: ops.gemm(A, W, act="relu", out=out), 2.0 * V * L * H * H),
    "dX   [V,1280]x[320,1280]^T": (lambda: ops.gemm(G, Wh, trans_b=True, out=out), 2.0 * V * L * H * H),
    "dW   [V,320]^T x [V,1280]": (lambda: ops.gemm(X, G, trans_a=True), 2.0 * V * L * H * H),
    "proj [V,320]x[320,320]": (lambda: ops.gemm(X, W[:H], act="tanh", out=out), 2.0 * V * H * H),
}
print("TFGNN_GEMM_DEBUG =", os.environ.get("TFGNN_GEMM_DEBUG", "0"))
for name, (fn, flops) in shapes.items():
    ms = t(fn)
    print(f"{name:32s} {ms*1000:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")

if int(os.environ.get("TFGNN_GEMM_DEBUG", "0")) & 4:
    # shader clock during the fwd GEMM (workgroup 0): clock64 / wall_clock64 (100 MHz)
    from tf2_gnn_amd import _lib
    import ctypes
    lib = _lib.load()
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    Wt = W.t().contiguous()
    for name, (M, N, K, a, b, ta, tb) in {
        "fwd": (V, H, L * H, A, W, 0, 0),
        "fwdT": (V, H, L * H, A, Wt, 0, 1),
    }.items():
        for _ in range(3):
            lib.tfgnn_gemm(ta, tb, M, N, K, ctypes.c_void_p(a.data_ptr()), a.stride(0), ctypes.c_void_p(b.data_ptr()), b.stride(0),
                           ctypes.c_void_p(out.data_ptr()), N, None, 1, 0, ctypes.c_void_p(ws.data_ptr()), 0, None)
        torch.cuda.synchronize()
        nb = (M + 127) // 128
        c = ws[: nb * 32].view(torch.int64).cpu().view(nb, 4).double()
        t0 = c[:, 1].min()
        start, loop_end, end = (c[:, 1] - t0) / 100.0, (c[:, 2] - t0) / 100.0, (c[:, 3] - t0) / 100.0
        mhz = c[:, 0] / (c[:, 2] - c[:, 1]) * 100.0
        print(f"{name}: {nb} workgroups; start skew max {start.max():.1f} us; main loop {float((loop_end-start).min()):.1f}/{float((loop_end-start).mean()):.1f}/{float((loop_end-start).max()):.1f} us (min/mean/max); "
              f"epilogue {float((end-loop_end).min()):.1f}/{float((end-loop_end).mean()):.1f}/{float((end-loop_end).max()):.1f} us; last end {end.max():.1f} us; clock {float(mhz.mean()):.0f} MHz")

