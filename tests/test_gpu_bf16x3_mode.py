"""The parity tests of the benchmarked configuration with the GEMMs in the bf16x3 mode bench.py runs in by default
(fp32 operands split exactly into three bf16 pieces, csrc/gemm_x3.hip).  Same oracles, same 1e-5 tolerance as
in the fp32-MFMA mode: the mode changes how the fp32 product is evaluated, not the contract."""
import pytest
import torch

from tests.test_gpu_full_size import (  # noqa: F401  (collected here again, under the mode fixture below)
    cfg2,
    test_cfg2_rgcn_gnn_step_gradients_finite_and_reproducible,
    test_cfg2_rgcn_layer_matches_oracle_on_sampled_targets,
)
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


@pytest.mark.parametrize("V,K,H", [(1000, 128, 128), (4097, 640, 64), (300, 72, 192)])
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
        fused = ops.gemm_gru(x, kernel, bias[0], mh, h)
        assert fused is not None
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
