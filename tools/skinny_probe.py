"""Times the QM9-sized products (configs[3]: V = 1.15 M nodes, ~0.68 M edges per type, width 128) through ops.gemm /
ops.gemm_grad in bf16x3 mode:  python tools/skinny_probe.py            (shipped routes)
                               TFGNN_X3_STREAM_MIN_ROWS=0 TFGNN_LONG_K_SPLITS=0 python tools/skinny_probe.py   (tiled kernels, <= 64 splits)
Prints one JSON object: shape -> microseconds and the fraction of the 8 TB/s HBM roof its compulsory bytes reach."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from tf2_gnn_amd import ops  # noqa: E402


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    dev = torch.device("cuda:0")
    ops.set_gemm_mode("bf16x3")
    V, E = 1151896, 680828
    out = {}
    g = torch.Generator(device=dev).manual_seed(1)
    for name, M, K, N, tb, kind in [
        ("dense V x128x128 nn", V, 128, 128, False, "plain"),
        ("dense V x128x128 nt relu", V, 128, 128, True, "relu"),
        ("edge-mlp hidden E x128x128 nn relu", E, 128, 128, False, "relu"),
        ("edge-mlp d(hidden) E x128x128 nt relu'", E, 128, 128, True, "grad"),
        ("gru input V x128x384 nt", V, 128, 384, True, "plain"),
        ("edge-mlp P|Q V x128x640 nn", V, 128, 640, False, "plain"),
        ("edge-mlp P|Q V x128x1280 nn", V, 128, 1280, False, "plain"),
    ]:
        A = torch.randn((M, K), device=dev, generator=g)
        B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g) * 0.1
        C = torch.empty((M, N), device=dev)
        saved = torch.randn((M, N), device=dev, generator=g) if kind == "grad" else None
        if kind == "grad":
            fn = lambda: ops.gemm_grad(A, B, trans_b=tb, out=C, act_grad=("relu", saved))  # noqa: E731
        else:
            fn = lambda: ops.gemm(A, B, trans_b=tb, act="relu" if kind == "relu" else None, out=C)  # noqa: E731
        us = timed(fn)
        nbytes = 4 * (M * K + M * N + (M * N if kind == "grad" else 0))
        out[name] = {"us": round(us, 1), "hbm_frac": round(nbytes / (us * 1e-6) / 8e12, 3)}
        del A, B, C, saved
    for name, K, M, N in [
        ("dW V rows 128x128", V, 128, 128),
        ("dW E rows 128x128", E, 128, 128),
        ("dW V rows 128x640", V, 128, 640),
        ("dW V rows 128x384", V, 128, 384),
    ]:
        X = torch.randn((K, M), device=dev, generator=g)
        G = torch.randn((K, N), device=dev, generator=g)
        C = torch.empty((M, N), device=dev)
        us = timed(lambda: ops.gemm(X, G, trans_a=True, out=C))
        out[name] = {"us": round(us, 1), "hbm_frac": round(4 * K * (M + N) / (us * 1e-6) / 8e12, 3)}
        del X, G, C
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
