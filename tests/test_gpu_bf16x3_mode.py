"""The parity tests of the benchmarked configuration with the GEMMs in the bf16x3 mode bench.py runs in by default
(fp32 operands split exactly into three bf16 pieces, csrc/gemm_x3.hip).  Same oracles, same 1e-5 tolerance as
in the fp32-MFMA mode: the mode changes how the fp32 product is evaluated, not the contract."""
import pytest
import torch

from tests.test_gpu_full_size import cfg2  # noqa: F401  (the full-size tests themselves run per mode in their own module)
from tests.test_gpu_layers import check_layer_backward

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def bf16x3_mode():
    from tf2_gnn_amd import ops

    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("bf16x3")
    yield
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("over", [{}, {"message_activation_function": "tanh", "aggregation_function": "mean"}],
                         ids=["rgcn", "rgcn_tanh_mean"])
def test_rgcn_h320_backward_parity_in_bf16x3_mode(dev, over):
    """H = D = 320, 4 edge types: every GEMM of the layer (forward [V,1280]x[1280,320], dX, dW^T) takes the split-
    operand kernels; forward, dX and dW against fp64 autograd through the oracle."""
    from tf2_gnn_amd import ops

    assert ops.get_gemm_mode() == ops.GEMM_BF16X3
    check_layer_backward(dev, "rgcn_h320_bf16x3", "RGCN", over, V=700, E=6000, L=4, H=320)


def test_modes_agree_on_the_benchmarked_layer(dev, cfg2):
    """fp32-MFMA and bf16x3 evaluations of the same RGCN layer (V=30k, E=900k, H=320) differ by fp32 rounding only."""
    from tests.test_gpu_full_size import _build
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    layer, _ = _build("RGCN", {"hidden_dim": cfg2["H"]}, cfg2["H"], cfg2["L"])
    inp = MessagePassingInput(cfg2["X"], cfg2["graph"])
    out3 = layer(inp, training=False)
    ops.set_gemm_mode("fp32")
    out32 = layer(inp, training=False)
    scale = float(out32.abs().max())
    assert float((out3 - out32).abs().max()) <= 2e-6 * max(1.0, scale)


# the last three run on the streaming kernel (V >= 65536, K in {64, 96, 128}: three column tiles z | r | h per 32 units)
@pytest.mark.parametrize("V,K,H", [(1000, 128, 128), (4097, 640, 64), (300, 72, 192), (70001, 128, 128), (150003, 64, 64),
                                   (300001, 96, 192)])
def test_gemm_gru_matches_unfused(dev, V, K, H):
    """tfgnn_gemm_gru (matmul + GRUCell gate math in one kernel) against tfgnn_gemm + tfgnn_gru_gates_forward in the same
    mode: h' and the saved gates agree to fp32 rounding; and against an fp64 evaluation of the cell."""
    from tf2_gnn_amd import ops

    prev = ops.set_gemm_mode("bf16x3")
    try:
        g = torch.Generator().manual_seed(V)
        x = torch.randn((V, K), generator=g).to(dev)
        h = torch.randn((V, H), generator=g).to(dev)
        kernel = (torch.randn((K, 3 * H), generator=g) * 0.1).to(dev)
        rk = (torch.randn((H, 3 * H), generator=g) * 0.1).to(dev)
        bias = torch.randn((2, 3 * H), generator=g).to(dev)
        mh = ops.gemm(h, rk, bias=bias[1])
        from tests.helpers import KernelsUsed

        with KernelsUsed() as used:
            fused = ops.gemm_gru(x, kernel, bias[0], mh, h)
        assert fused is not None
        assert used.delta["gemm_stream"] == (1 if V >= 65536 and K in (64, 96, 128) else 0), used.delta
        mx = ops.gemm(x, kernel, bias=bias[0])
        h_ref, gates_ref = ops.gru_gates_forward(mx, mh, h)
        # (the unfused product may run on another tile width or, for N = 192, on the fp32-MFMA kernel: fp32 rounding apart)
        assert float((fused[0] - h_ref).abs().max()) <= 1e-5
        assert float((fused[1] - gates_ref).abs().max()) <= 1e-5
        # fp64 check of the whole cell
        mx64 = x.double().cpu() @ kernel.double().cpu() + bias[0].double().cpu()
        mh64 = mh.double().cpu()
        z = torch.sigmoid(mx64[:, :H] + mh64[:, :H])
        r = torch.sigmoid(mx64[:, H:2 * H] + mh64[:, H:2 * H])
        c = torch.tanh(mx64[:, 2 * H:] + r * mh64[:, 2 * H:])
        ref = z * h.double().cpu() + (1 - z) * c
        assert float((fused[0].double().cpu() - ref).abs().max()) <= 1e-5
        no_gates = ops.gemm_gru(x, kernel, None, mh, h, save_gates=False)
        assert no_gates[1] is None
    finally:
        ops.set_gemm_mode(prev)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x3_9", "f16x2"])
def test_split_operand_modes_special_values_match_fp32_result_class(dev, mode):
    """+-inf, NaN, +-FLT_MAX rows (the empty-segment value of a max aggregation flows into the next layer's product),
    subnormals, 1e+-30 dynamic range: every output element has the class (finite / +inf / -inf / nan) the fp32-MFMA
    kernel gives it, and the finite ones agree to fp32 rounding of sum |a||b|."""
    from tf2_gnn_amd import ops

    M, N, K = 200, 128, 128
    g = torch.Generator().manual_seed(17)
    A = torch.randn((M, K), generator=g)
    Bt = torch.randn((N, K), generator=g) * 0.1
    fmax = torch.finfo(torch.float32).max
    A[1, 7] = float("inf")
    A[2, 9] = float("-inf")
    A[3, 11] = float("nan")
    A[4, :] = -fmax
    A[5, 13] = fmax
    A[6, :] = 1e-42                       # subnormal inputs
    A[7, :] *= 1e-37                      # smallest normals: lower pieces are subnormal bf16 / fp16 values
    A[8, :] *= 1e30
    A[9, :] *= 1e-30
    A[10, ::2] *= 1e20                    # dynamic range inside a row
    Bt[5, :] = 0.0                        # inf * 0 -> nan in every mode
    Bt[:, 13] *= 1e-3

    def run(m):
        prev = ops.set_gemm_mode(m)
        try:
            if m == "f16x2":
                return ops.sp_gemm_nt(ops.sp_split_rows(A.to(dev)), ops.sp_split_rows(Bt.to(dev))).cpu()
            return ops.gemm(A.to(dev), Bt.to(dev), trans_b=True).cpu()
        finally:
            ops.set_gemm_mode(prev)

    ref32, got = run("fp32"), run(mode)

    def cls(t):
        return torch.where(torch.isnan(t), 3, torch.where(torch.isinf(t), torch.where(t > 0, 1, 2), 0))

    # rows made of +-FLT_MAX: the fp32 chain overflows to +-inf part-way, an evaluation that sums in another order (or in
    # scaled units) may stay finite - both are "the fp32 result"; everywhere else the classes must coincide
    free = torch.zeros(M, dtype=torch.bool)
    free[[4, 5]] = True
    # rows with an infinite input: inf * b is +-inf in fp32; a split evaluation also multiplies inf by b's LOWER pieces,
    # and where such a piece is exactly zero (b representable in fewer bits) inf * 0 = nan joins the sum.  Non-finite in
    # both, the sign of an infinite result agrees.
    infrows = torch.zeros(M, dtype=torch.bool)
    infrows[[1, 2]] = True
    same = cls(ref32) == cls(got)
    strict = ~free & ~infrows
    assert bool(same[strict].all()), f"{mode}: result class differs from the fp32-MFMA kernel at {int((~same[strict]).sum())} elements"
    assert bool((~torch.isfinite(got[infrows])).all()) and bool((~torch.isfinite(ref32[infrows])).all())
    both_inf = torch.isinf(got[infrows]) & torch.isinf(ref32[infrows])
    assert bool((torch.sign(got[infrows])[both_inf] == torch.sign(ref32[infrows])[both_inf]).all())
    if mode != "f16x2":  # truncation split: all pieces of b share its sign, only the inf * 0 case gives nan
        assert float(torch.isnan(got[infrows]).float().mean()) < 0.2
    # (f16x2 splits by round-to-nearest - one more bit - so b's lower piece has either sign and inf * l_b cancels
    # inf * h_b to nan about half the time: an infinite operand gives a non-finite result, not necessarily +-inf)
    assert bool((torch.isfinite(got[free]) | torch.isinf(got[free])).all())
    fin = torch.isfinite(ref32) & torch.isfinite(got) & ~free.unsqueeze(1)
    mag = A.double().abs().nan_to_num(0, 0, 0) @ Bt.double().abs().t()
    ref64 = (A.double().nan_to_num(0, 0, 0) @ Bt.double().t())
    rows_ok = torch.ones(M, dtype=torch.bool)
    rows_ok[[1, 2, 3, 4, 5]] = False
    err = ((got.double() - ref64).abs() / mag.clamp(min=1e-300))[rows_ok]
    # rows 6 / 7 (subnormal and near-subnormal inputs): the lower pieces underflow in every split format; the result is
    # correct to the leading piece (bf16: 2^-8, fp16 after scaling: full precision)
    tiny = torch.zeros(M, dtype=torch.bool)
    tiny[[6, 7]] = True
    assert float(err[~tiny[rows_ok]].max()) <= 1e-6, (mode, float(err[~tiny[rows_ok]].max()))
    # row 7, inputs ~1e-37: the lower pieces of the bf16 split are subnormal bf16 values, which the matrix cores flush -
    # correct to the leading piece (2^-8); the scaled fp16 split keeps full precision
    e7 = float(((got.double() - ref64).abs() / mag.clamp(min=1e-300))[7].max())
    assert e7 <= (2.0 ** -7 if mode != "f16x2" else 1e-5), (mode, e7)
    # row 6, subnormal fp32 inputs (1e-42): results of magnitude 1e-43 - flushed to zero by the bf16 pipes, kept by fp32
    # and by the scaled fp16 split; only that they stay finite and no larger than sum |a||b|
    assert bool(torch.isfinite(got[6]).all()) and bool((got[6].double().abs() <= 2 * mag[6] + 1e-45).all())
    if mode == "f16x2":
        assert float(((got.double() - ref64).abs() / mag.clamp(min=1e-300))[6].max()) <= 1e-3
    assert bool(fin.any())
