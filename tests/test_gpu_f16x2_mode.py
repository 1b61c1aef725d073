"""The parity tests of the benchmarked configuration with the hot products in the f16x2 mode (pre-split SP16 operands,
csrc/gemm_sp.hip; everything else as in bf16x3).  Same oracles, same 1e-5 tolerance: the mode changes how the fp32 product
is evaluated, not the contract."""
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
def f16x2_mode():
    from tf2_gnn_amd import ops

    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("f16x2")
    yield
    ops.set_gemm_mode(prev)


@pytest.mark.parametrize("H,L", [(320, 4), (128, 4), (256, 2), (128, 5)])
@pytest.mark.parametrize("over", [{}, {"message_activation_function": "tanh", "aggregation_function": "mean"},
                                  {"message_activation_function": "gelu", "normalize_by_num_incoming": False}],
                         ids=["rgcn", "rgcn_tanh_mean", "rgcn_gelu_nonorm"])
def test_rgcn_backward_parity_in_f16x2_mode(dev, over, H, L):
    """Every product of the layer on split operands (forward [V, L H] x [L H, H] from the SP16-writing gather, dX, dW):
    forward, dX and dW against fp64 autograd through the oracle."""
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers.message_passing import RGCN

    assert ops.get_gemm_mode() == ops.GEMM_F16X2
    probe = RGCN(dict(RGCN.get_default_hyperparameters(), hidden_dim=H))
    assert probe._f16x2_eligible(700, H, L, H), "the test shape must take the split-operand path"
    check_layer_backward(dev, f"rgcn_h{H}_f16x2", "RGCN", over, V=700, E=6000, L=L, H=H)


def test_ineligible_shapes_fall_back(dev):
    """3 edge types x H = 320 (L H = 960 is not a multiple of the 128-row tile) and target-state input run the bf16x3
    kernels, transparently."""
    check_layer_backward(dev, "rgcn_h320_L3", "RGCN", {}, V=300, E=3000, L=3, H=320)
    check_layer_backward(dev, "rgcn_target_h128", "RGCN", {"use_target_state_as_input": True}, V=300, E=3000, L=4, H=128)


def test_modes_agree_on_the_benchmarked_layer(dev, cfg2):
    """fp32-MFMA and f16x2 evaluations of the same RGCN layer (V=30k, E=900k, H=320) differ by fp32 rounding only."""
    from tests.test_gpu_full_size import _build
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    layer, _ = _build("RGCN", {"hidden_dim": cfg2["H"]}, cfg2["H"], cfg2["L"])
    inp = MessagePassingInput(cfg2["X"], cfg2["graph"])
    out2 = layer(inp, training=False)
    assert layer._ctx.get("f16x2")
    ops.set_gemm_mode("fp32")
    out32 = layer(inp, training=False)
    scale = float(out32.abs().max())
    assert float((out2 - out32).abs().max()) <= 2e-6 * max(1.0, scale)
