"""The two QM9-sized product routes of the bf16x3 path (csrc/gemm_x3.hip): the streaming kernel with the weight block
resident in LDS (short K, very many rows: Dense / edge-MLP / GRU-input products of configs[3]) and the long-K split of
skinny weight gradients (1 - 8 output tiles under ~10^6 rows).  Every case against the fp64 product, and the kernel
family counters show that the route under test really ran."""
import pytest
import torch

from tests.helpers import KernelsUsed, assert_close

pytestmark = pytest.mark.gpu


# (f16x2, round 6: the same streaming kernel in the f16x2 arithmetic - operands split on the fly into two fp16 pieces under a
#  power-of-two scale per row of A / column of B, 3 piece products; counted as kernel family stream_f16x2)
@pytest.fixture(params=["bf16x3", "bf16x3_9", "f16x2"])
def x3_mode(request):
    from tf2_gnn_amd import ops

    prev = ops.get_gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(prev)


STREAM_CASES = [
    # M, K, N, trans_b, epilogue, (pad_a, pad_b, pad_c)
    (65536, 128, 128, False, "none", (0, 0, 0)),
    (65536 + 37, 128, 128, True, "bias_relu", (0, 0, 0)),
    (70001, 64, 384, False, "bias_tanh", (4, 0, 8)),
    (70001, 96, 128, True, "accumulate", (0, 4, 4)),
    (100003, 128, 640, True, "none", (8, 0, 0)),
    (65540, 128, 256, False, "grad_relu", (0, 0, 0)),
    (65540, 64, 128, True, "grad_mask_tanh", (0, 0, 4)),
    (131072 + 5, 32 * 3, 1280, False, "bias_relu", (0, 0, 0)),
    # several row blocks per workgroup at every K (256 row streams at N = 128, 80 at N = 384)
    (300001, 128, 128, False, "bias_relu", (0, 0, 0)),
    (300001, 128, 384, True, "grad_relu", (0, 0, 0)),
    (200003, 64, 128, True, "none", (0, 0, 0)),
    (200003, 96, 128, False, "accumulate", (0, 0, 0)),
    (70001, 128, 128, True, "grad_mask_relu_accumulate", (0, 0, 0)),
]


@pytest.mark.parametrize("case", STREAM_CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}-{'nt' if c[3] else 'nn'}-{c[4]}")
def test_streaming_kernel_against_fp64(dev, x3_mode, case):
    from tf2_gnn_amd import ops

    M, K, N, tb, epi, (pad_a, pad_b, pad_c) = case
    g = torch.Generator().manual_seed(M + 3 * K + 7 * N)
    A_full = torch.randn((M, K + pad_a), generator=g)
    b_shape = (N, K) if tb else (K, N)
    B_full = torch.randn((b_shape[0], b_shape[1] + pad_b), generator=g) * 0.2
    C_full = torch.randn((M, N + pad_c), generator=g)
    A, B = A_full[:, :K], B_full[:, : b_shape[1]]
    Ad, Bd, Cd = A_full.to(dev)[:, :K], B_full.to(dev)[:, : b_shape[1]], C_full.to(dev)
    out_view = Cd[:, :N]
    ref = A.double() @ (B.double().t() if tb else B.double())
    with KernelsUsed() as k:
        if epi in ("none", "accumulate"):
            ops.gemm(Ad, Bd, trans_b=tb, out=out_view, accumulate=epi == "accumulate")
            if epi == "accumulate":
                ref = ref + C_full[:, :N].double()
            res = out_view
        elif epi in ("bias_relu", "bias_tanh"):
            bias = torch.randn(N, generator=g)
            act = epi.split("_")[1]
            ops.gemm(Ad, Bd, trans_b=tb, bias=bias.to(dev), act=act, out=out_view)
            ref = ref + bias.double()
            ref = torch.relu(ref) if act == "relu" else torch.tanh(ref)
            res = out_view
        else:
            saved = torch.randn((M, N), generator=g)
            mask = (torch.rand((M, N), generator=g) > 0.2).float() * 1.25 if "mask" in epi else None
            act = "relu" if "relu" in epi else "tanh"
            res = ops.gemm_grad(Ad, Bd, trans_b=tb, out=out_view, out_mul=None if mask is None else mask.to(dev),
                                act_grad=(act, saved.to(dev)), accumulate="accumulate" in epi)
            dact = (saved > 0).double() if act == "relu" else 1.0 - saved.double() ** 2
            ref = ref * dact * (1.0 if mask is None else mask.double())
            if "accumulate" in epi:
                ref = ref + C_full[:, :N].double()
    assert k.delta["gemm_stream"] == 1 and k.delta["stream_f16x2"] == (1 if x3_mode == "f16x2" else 0), k.delta
    assert res.data_ptr() == out_view.data_ptr()
    scale = max(1.0, 0.2 * float(K) ** 0.5)
    assert_close(res.cpu() / scale, (ref / scale).float(), tol=1e-5, what=f"streaming {x3_mode} {case}")
    if pad_c:
        assert torch.equal(Cd[:, N:].cpu(), C_full[:, N:])


def test_streaming_kernel_special_values_stay_in_their_rows(dev, x3_mode):
    """inf / nan rows stay in their rows, zero rows give exact zeros."""
    from tf2_gnn_amd import ops

    M, K, N = 65536 + 64, 128, 128
    g = torch.Generator().manual_seed(5)
    A = torch.randn((M, K), generator=g)
    A[7, 3] = float("inf")
    A[M - 1, 100] = float("nan")
    A[12345] = 0.0
    B = torch.randn((K, N), generator=g).abs() + 0.1  # positive: inf * b stays +inf, no inf - inf
    out = ops.gemm(A.to(dev), B.to(dev)).cpu()
    # inf * b is +inf in fp32; the split evaluation also multiplies inf by b's lower pieces, and where such a piece is exactly
    # zero inf * 0 = nan joins the sum (tests/test_gpu_bf16x3_mode.py): non-finite everywhere, +inf almost everywhere
    assert (~torch.isfinite(out[7])).all()
    if x3_mode != "f16x2":  # (the round-to-nearest split of f16x2 has lower pieces of either sign: inf * l_b may be -inf, the sum nan)
        assert (out[7][torch.isinf(out[7])] > 0).all() and float(torch.isnan(out[7]).float().mean()) < 0.2
    assert torch.isnan(out[M - 1]).all()
    assert torch.equal(out[12345], torch.zeros(N))
    keep = torch.ones(M, dtype=torch.bool)
    keep[7] = keep[M - 1] = False
    assert torch.isfinite(out[keep]).all()


@pytest.mark.parametrize("K,M,N", [(150001, 128, 128), (140000, 128, 640), (200003, 256, 128)])
def test_long_k_weight_gradient_split(dev, x3_mode, K, M, N):
    """dW = X^T G with a 1 - 4 tile output under K >= 2^17 rows runs up to 512 splits and the sliced reduction."""
    from tf2_gnn_amd import ops

    g = torch.Generator().manual_seed(K + M)
    X = torch.randn((K, M), generator=g)
    G = torch.randn((K, N), generator=g) * 0.05
    out = ops.gemm(X.to(dev), G.to(dev), trans_a=True)
    ref = X.double().t() @ G.double()
    scale = 0.05 * float(K) ** 0.5
    assert_close(out.cpu() / scale, (ref / scale).float(), tol=1e-5, what=f"long-K split {x3_mode} {(K, M, N)}")
    again = ops.gemm(X.to(dev), G.to(dev), trans_a=True)
    assert torch.equal(out, again)  # fixed summation order


@pytest.mark.parametrize("M,V,K,N,tb", [(70000, 20000, 128, 128, False), (150001, 9000, 64, 256, True), (5000, 3000, 128, 128, False),
                                        (70000, 20000, 72, 128, False)])
@pytest.mark.parametrize("epi", ["plain", "bias_relu", "after", "before_relu", "before_tanh_bias"])
def test_gathered_rows_product(dev, x3_mode, M, V, K, N, tb, epi):
    """tfgnn_gemm_gathered = embedding_lookup + Dense: rows of A read through an index inside the streaming kernel, or (small /
    odd shapes) a row gather followed by tfgnn_gemm; both accumulate orders."""
    from tf2_gnn_amd import ops

    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn((V, K), generator=g)
    idx = torch.randint(0, V, (M,), generator=g, dtype=torch.int64)
    idx[0], idx[-1] = V - 1, 0
    B = torch.randn((N, K) if tb else (K, N), generator=g) * 0.2
    C0 = torch.randn((M, N), generator=g)
    bias = torch.randn(N, generator=g) if "bias" in epi else None
    ref = X.double()[idx] @ (B.double().t() if tb else B.double())
    if bias is not None:
        ref = ref + bias.double()
    act = "relu" if "relu" in epi else ("tanh" if "tanh" in epi else None)
    f = {"relu": torch.relu, "tanh": torch.tanh, None: lambda t: t}[act]
    if epi.startswith("before"):
        ref = f(ref + C0.double())
        acc = "before"
    elif epi == "after":
        ref = f(ref) + C0.double()
        acc = "after"
    else:
        ref = f(ref)
        acc = None
    out = C0.to(dev) if acc else None
    streaming = M >= 65536 and K in (64, 96, 128)
    with KernelsUsed() as k:
        res = ops.gemm_gathered(X.to(dev), idx.to(torch.int32).to(dev), B.to(dev), trans_b=tb, bias=None if bias is None else bias.to(dev),
                                act=act, out=out, accumulate=acc)
    assert k.delta["gemm_stream"] == (1 if streaming else 0), k.delta
    assert k.delta["stream_f16x2"] == (1 if streaming and x3_mode == "f16x2" else 0), k.delta
    scale = max(1.0, 0.2 * float(K) ** 0.5)
    assert_close(res.cpu() / scale, (ref / scale).float(), tol=1e-5, what=f"gathered product {x3_mode} {(M, V, K, N, tb, epi)}")


def test_gathered_rows_product_reads_an_out_of_range_index_as_a_zero_row(dev, x3_mode):
    """row indices reach the streaming kernel unchecked and its 24-bit multiply would alias another row.  An index
    outside [0, a_rows) now reads as zeros (negative, one past the end, far beyond 2^24) - the neighbours are untouched."""
    from tf2_gnn_amd import ops

    M, V, K, N = 70000, 5000, 128, 128
    g = torch.Generator().manual_seed(1)
    X = torch.randn((V, K), generator=g)
    B = torch.randn((K, N), generator=g) * 0.2
    idx = torch.randint(0, V, (M,), generator=g, dtype=torch.int64)
    bad = {3: -1, 64: V, 40000: (1 << 24) + 5, M - 1: -(1 << 24)}
    good = idx.clone()
    for pos, val in bad.items():
        idx[pos] = val
    with KernelsUsed() as k:
        res = ops.gemm_gathered(X.to(dev), idx.to(torch.int32).to(dev), B.to(dev))
    assert k.delta["gemm_stream"] == 1
    ref = ops.gemm_gathered(X.to(dev), good.to(torch.int32).to(dev), B.to(dev))
    keep = torch.ones(M, dtype=torch.bool)
    for pos in bad:
        keep[pos] = False
        assert torch.equal(res[pos].cpu(), torch.zeros(N)), pos
    assert torch.equal(res.cpu()[keep], ref.cpu()[keep])


@pytest.mark.parametrize("tb", [False, True])
def test_streaming_kernel_f16x2_arithmetic_over_wide_dynamic_range(dev, tb):
    """The f16x2 form scales every ROW of the streamed operand and every COLUMN of the resident one by its own power of two:
    rows 2^+-30 apart, columns 2^+-20 apart and entries far below their row's maximum must all come out to fp32 accuracy,
    relative to sum_k |a||b| of the entry (the yardstick of the other split-operand products, tests/test_gpu_gemm_sp.py)."""
    from tf2_gnn_amd import ops

    M, K, N = 70000, 128, 256
    g = torch.Generator().manual_seed(9)
    A = torch.randn((M, K), generator=g) * torch.exp2(torch.randint(-30, 31, (M, 1), generator=g).float())
    A[:, ::5] *= 2.0 ** -12  # entries far below the row maximum
    B = torch.randn((N, K) if tb else (K, N), generator=g) * 0.2
    colscale = torch.exp2(torch.randint(-20, 21, (N,), generator=g).float())
    B = B * (colscale.unsqueeze(1) if tb else colscale.unsqueeze(0))
    prev = ops.set_gemm_mode("f16x2")
    try:
        with KernelsUsed() as k:
            out = ops.gemm(A.to(dev), B.to(dev), trans_b=tb).cpu()
        assert k.delta["stream_f16x2"] == 1, k.delta
    finally:
        ops.set_gemm_mode(prev)
    Bm = B.double().t() if tb else B.double()
    ref = A.double() @ Bm
    mag = A.double().abs() @ Bm.abs()
    err = float(((out.double() - ref).abs() / mag).max())
    assert err <= 2e-6, err


@pytest.mark.parametrize("V,H", [(70001, 128), (65536 + 40, 64)])
def test_gru_cell_with_both_products_in_the_kernel(dev, V, H):
    """tfgnn_gemm_gru2 (round 6): h' = GRUCell(x, h) with x @ kernel AND h @ recurrent_kernel inside the gate kernel (mode f16x2,
    QM9-sized row counts) against the fp64 cell ([ext] TF2 GRUCell, reset_after: oracle gru_cell); the gates saved for the
    backward pass and the one third of mh it reads (the candidate part h U_h + b_h) against fp64 too."""
    from oracle import tf2gnn_oracle as orc
    from tf2_gnn_amd import ops

    g = torch.Generator().manual_seed(V)
    x = torch.randn((V, H), generator=g) * torch.exp2(torch.randint(-6, 7, (V, 1), generator=g).float())
    h = torch.tanh(torch.randn((V, H), generator=g))
    kernel = (torch.rand((H, 3 * H), generator=g) * 2 - 1) * (6.0 / (4 * H)) ** 0.5
    recurrent = torch.linalg.qr(torch.randn((3 * H, H), generator=g))[0].t().contiguous()
    bias = torch.randn((2, 3 * H), generator=g) * 0.1
    prev = ops.set_gemm_mode("f16x2")
    try:
        with KernelsUsed() as k:
            res = ops.gemm_gru2(x.to(dev), kernel.to(dev), bias[0].to(dev), h.to(dev), recurrent.to(dev), bias[1].to(dev))
        assert res is not None and k.delta["stream_f16x2"] == 1 and k.delta["gemm_stream"] == 1, k.delta
        ops.set_gemm_mode("bf16x3")
        assert ops.gemm_gru2(x.to(dev), kernel.to(dev), bias[0].to(dev), h.to(dev), recurrent.to(dev), bias[1].to(dev)) is None
    finally:
        ops.set_gemm_mode(prev)
    h_new, gates, mh = (t.cpu() for t in res)
    x64, h64, k64, r64, b64 = x.double(), h.double(), kernel.double(), recurrent.double(), bias.double()
    ref = orc.gru_cell(x64, h64, k64, r64, b64)
    mx, mh64 = x64 @ k64 + b64[0], h64 @ r64 + b64[1]
    z = torch.sigmoid(mx[:, :H] + mh64[:, :H])
    r = torch.sigmoid(mx[:, H:2 * H] + mh64[:, H:2 * H])
    c = torch.tanh(mx[:, 2 * H:] + r * mh64[:, 2 * H:])
    # yardstick: the same cell evaluated in fp32 in the reference's order (rows of x reach |x| ~ 200: the pre-activations carry
    # fp32 rounding of sums of that size) - the rule of the full-size tests: err <= max(1e-5, 2 x reference-order fp32)
    ref32 = orc.gru_cell(x, h, kernel, recurrent, bias)
    from tests.helpers import scaled_error

    e32 = scaled_error(ref32, ref)
    assert_close(h_new, ref.float(), tol=max(1e-5, 2 * e32), what=f"gru2 h' V={V} H={H}")
    for name, got, want in (("z", gates[:, :H], z), ("r", gates[:, H:2 * H], r), ("candidate", gates[:, 2 * H:], c),
                            ("candidate part of mh", mh[:, 2 * H:], mh64[:, 2 * H:])):
        assert_close(got, want.float(), tol=max(1e-5, 2 * e32), what=f"gru2 {name} V={V} H={H}")
    # and against the two-kernel route it replaces (mh by a product of its own, tfgnn_gemm_gru): the same numbers to fp32 rounding
    prev = ops.set_gemm_mode("f16x2")
    try:
        mh_old = ops.gemm(h.to(dev), recurrent.to(dev), bias=bias[1].to(dev))
        h_old, _ = ops.gemm_gru(x.to(dev), kernel.to(dev), bias[0].to(dev), mh_old, h.to(dev))
    finally:
        ops.set_gemm_mode(prev)
    assert_close(h_new, h_old.cpu(), tol=max(1e-5, 2 * e32), what=f"gru2 vs the two-kernel route V={V} H={H}")
