"""Seeded fuzz of the Dense entry point over shapes, layouts, leading dimensions, epilogues and both evaluation
modes: every case against the fp64 product.  Catches dispatch mistakes between the fp32-MFMA kernel and the three
tile widths / two kernels of the split-operand path (K tails, ragged M, padded views, unaligned operands)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        N = int(rng.choice([128, 256, 320, 384, 512, 640, 96, 200, 1280]))
        M = int(rng.choice([1, 7, 64, 127, 128, 129, 300, 1000, 2050]))
        K = int(rng.choice([4, 60, 64, 68, 100, 128, 320, 516, 1280, 4100]))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        pad_a, pad_b, pad_c = (int(rng.choice([0, 0, 4, 8, 3])) for _ in range(3))
        epi = int(rng.integers(0, 4))  # 0 none, 1 bias+act, 2 accumulate, 3 bias+act+accumulate
        out.append((i, ta, tb, M, N, K, pad_a, pad_b, pad_c, epi))
    return out


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16x3_9"])
@pytest.mark.parametrize("case", _cases(40, 20260925), ids=lambda c: f"c{c[0]}")
def test_gemm_fuzz(dev, mode, case):
    from tf2_gnn_amd import ops

    _, ta, tb, M, N, K, pad_a, pad_b, pad_c, epi = case
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(mode)
    try:
        g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
        a_shape = (K, M) if ta else (M, K)
        b_shape = (N, K) if tb else (K, N)
        A_full = torch.randn((a_shape[0], a_shape[1] + pad_a), generator=g)
        B_full = torch.randn((b_shape[0], b_shape[1] + pad_b), generator=g) * 0.2
        C_full = torch.randn((M, N + pad_c), generator=g)
        A, B = A_full[:, : a_shape[1]], B_full[:, : b_shape[1]]
        bias = torch.randn(N, generator=g) if epi in (1, 3) else None
        act = "tanh" if epi in (1, 3) else None
        acc = epi in (2, 3)
        ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
        if bias is not None:
            ref = torch.tanh(ref + bias.double())
        if acc:
            ref = ref + C_full[:, :N].double()
        Ad, Bd, Cd = A_full.to(dev)[:, : a_shape[1]], B_full.to(dev)[:, : b_shape[1]], C_full.to(dev)
        out_view = Cd[:, :N]
        ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb, bias=None if bias is None else bias.to(dev), act=act, out=out_view,
                 accumulate=acc)
        # fp32 accumulation noise of a length-K product grows like sqrt(K) (x 0.2: the scale of B); tanh' <= 1
        scale = max(1.0, 0.2 * float(K) ** 0.5) if act is None else 1.0
        tol = 1e-5 if act is None else max(1e-5, 6e-7 * float(K) ** 0.5)
        assert_close(out_view.cpu() / scale, (ref / scale).float(), tol=tol, what=f"fuzz {mode} {case}")
        if pad_c:  # the padding columns of the output view are untouched
            assert torch.equal(Cd[:, N:].cpu(), C_full[:, N:])
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("M,N,K", [(256, 128, 1001), (320, 1280, 67), (128, 320, 30001)])
def test_gemm_k_major_operands_with_odd_k(dev, M, N, K):
    """Weight-gradient layout (both operands K-major, K = number of edges / nodes: any value): the split kernels mask
    whole k rows, so K need not be a multiple of 4."""
    from tf2_gnn_amd import ops

    prev = ops.set_gemm_mode("bf16x3")
    try:
        g = torch.Generator().manual_seed(K)
        A = torch.randn((K, M), generator=g)
        B = torch.randn((K, N), generator=g) * 0.2
        out = ops.gemm(A.to(dev), B.to(dev), trans_a=True)
        ref = A.double().t() @ B.double()
        scale = max(1.0, 0.2 * float(K) ** 0.5)
        assert_close(out.cpu() / scale, (ref / scale).float(), tol=1e-5, what=f"TN odd K {M}x{N}x{K}")
    finally:
        ops.set_gemm_mode(prev)
