"""Workload for rocprofv3 --pmc passes over gemm_sp_nt_kernel (forward shape): dispatches, in order,
5 x full kernel on random operands, 5 x full on zeros, 5 x MFMA-only variant on random, 5 x MFMA-only on zeros."""
import ctypes
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tf2_gnn_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M, N, K = 30000, 320, 1280
vp = ctypes.c_void_p
out = torch.empty((M, N), device=dev)


def bind(bits):
    lib = ctypes.CDLL(str(ROOT / "tools" / "_probe" / f"libtfgnn_abl_{bits}.so"))
    fn = lib.tfgnn_sp_gemm_nt
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64] * 3 + [vp, ctypes.c_int64, vp, ctypes.c_int, vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp,
                                          ctypes.c_int, ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int, vp, ctypes.c_int64, vp]
    return fn


for bits in (0, 3):
    fn = bind(bits)
    for zero in (False, True):
        A = torch.zeros((M, K), device=dev) if zero else torch.randn((M, K), device=dev)
        Bt = torch.zeros((N, K), device=dev) if zero else torch.randn((N, K), device=dev) * 0.05
        a_op, b_op = ops.sp_split_rows(A), ops.sp_split_rows(Bt)
        torch.cuda.synchronize()
        for _ in range(5):
            rc = fn(M, N, K, a_op.data.data_ptr(), a_op.data.stride(0), a_op.inv_scale.data_ptr(), K, b_op.data.data_ptr(),
                    b_op.data.stride(0), b_op.inv_scale.data_ptr(), out.data_ptr(), N, None, 1, 0, None, 0, 0, None, 0,
                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        torch.cuda.synchronize()
print("done")
