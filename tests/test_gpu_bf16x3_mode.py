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
