"""Where does the time of gemm_sp_nt_kernel go?  Builds probe variants of the library with parts of the main loop left
out and times the forward shape with each.  The shipped source (csrc/gemm_sp.hip) carries NO probe code: the
instrumentation lives in tools/sp_ablate.patch and is applied to a temporary copy of the source here (-DSP_ABLATE=bits:
1 no DMA, 2 no fragment reads, 4 no MFMAs, 8 no barrier, 16 no stores, 32 no DMA of A, 64 no DMA of B, 128 hardware ids,
256 full-line DMA pattern (data lands wrongly), 512 no epilogue).
  python tools/sp_ablate.py build      (here: cross-compiles the variants into tools/_probe/)
  python tools/sp_ablate.py            (on the GPU box)"""
import ctypes
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tools" / "_probe"
VARIANTS = {0: "full", 1: "no DMA", 2: "no fragment reads", 4: "no MFMA", 8: "no barrier", 16: "no stores", 32: "no DMA of A",
            64: "no DMA of B", 1 | 2: "MFMA + barrier only", 4 | 2: "DMA + barrier only", 1 | 4: "fragment reads only", 1 | 2 | 8: "MFMA only, no barrier", 1 | 2 | 128: "hw ids", 256: "full, 128-B-line DMA pattern", 256 | 6: "DMA + barrier only, 128-B lines", 256 | 6 | 32: "DMA of B only, 128-B lines", 6 | 32: "DMA of B only", 6 | 64: "DMA of A only", 256 | 6 | 64: "DMA of A only, 128-B lines"}


import os

if os.environ.get("SP_ABLATE_VARIANTS"):
    VARIANTS = {int(b): VARIANTS.get(int(b), f"bits {b}") for b in os.environ["SP_ABLATE_VARIANTS"].split(",")}


def build():
    OUT.mkdir(exist_ok=True)
    objs = [str(p) for p in (ROOT / "tf2_gnn_amd" / "csrc" / "_obj").glob("*.o") if p.name != "gemm_sp.o"]
    src = OUT / "gemm_sp_instrumented.hip"  # the shipped kernel + the probe arms
    src.write_text((ROOT / "tf2_gnn_amd" / "csrc" / "gemm_sp.hip").read_text())
    subprocess.check_call(["patch", "-s", str(src), str(ROOT / "tools" / "sp_ablate.patch")])
    for bits in VARIANTS:
        obj = OUT / f"gemm_sp_{bits}.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-pass-failed",
                               f"-DSP_ABLATE={bits}", f"-I{ROOT / 'include'}", f"-I{ROOT / 'tf2_gnn_amd' / 'csrc'}", "-c",
                               str(src), "-o", str(obj)])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, str(obj), "-o", str(OUT / f"libtfgnn_abl_{bits}.so")])
        obj.unlink()
        print("built", bits, VARIANTS[bits], flush=True)


def run():
    import torch

    sys.path.insert(0, str(ROOT))
    from tf2_gnn_amd import ops

    dev = torch.device("cuda", 0)
    M, N, K = 30000, 320, int(os.environ.get("SP_K", "1280"))
    A = torch.randn((M, K), device=dev)
    Bt = torch.randn((N, K), device=dev) * 0.05
    a_op, b_op = ops.sp_split_rows(A), ops.sp_split_rows(Bt)
    out = torch.empty((M, N), device=dev)
    res = {}
    vp = ctypes.c_void_p
    for bits, name in VARIANTS.items():
        lib = ctypes.CDLL(str(OUT / f"libtfgnn_abl_{bits}.so"))
        fn = lib.tfgnn_sp_gemm_nt
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int64] * 3 + [vp, ctypes.c_int64, vp, ctypes.c_int, vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp,
                                              ctypes.c_int, ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int, vp, ctypes.c_int64, vp]
        st = torch.cuda.current_stream().cuda_stream

        def call():
            rc = fn(M, N, K, a_op.data.data_ptr(), a_op.data.stride(0), a_op.inv_scale.data_ptr(), K, b_op.data.data_ptr(),
                    b_op.data.stride(0), b_op.inv_scale.data_ptr(), out.data_ptr(), N, None, 1, 0, None, 0, 0, None, 0, st)
            assert rc == 0, rc

        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for a, b in ev:
            a.record()
            call()
            b.record()
        torch.cuda.synchronize()
        if bits & 128:
            ids = out.view(torch.int32).flatten()[: 235 * 4].cpu().view(235, 4)
            simd = (ids >> 4) & 3
            cu = (ids >> 8) & 15
            from collections import Counter
            print("SIMD ids of the 4 waves per workgroup:", Counter(tuple(sorted(r.tolist())) for r in simd).most_common(8))
            print("distinct CUs per workgroup:", Counter(len(set(r.tolist())) for r in cu))
            print("first rows (simd):", simd[:6].tolist(), "raw", [hex(v) for v in ids[0].tolist()])
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        res[name] = {"median_us": ts[len(ts) // 2], "min_us": ts[0]}
        print(f"{bits:3d} {name:24s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us", flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    json.dump(res, open(ROOT / "gpurun_out" / "sp_ablate.json", "w"), indent=1)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else run()
