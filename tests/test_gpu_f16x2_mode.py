"""The parity tests of the benchmarked configuration with the hot products in the f16x2 mode (pre-split SP16 operands,
csrc/gemm_sp.hip; everything else as in bf16x3).  Same oracles, same 1e-5 tolerance: the mode changes how the fp32 product
is evaluated, not the contract."""
import pytest
import torch

from tests.test_gpu_full_size import cfg2  # noqa: F401  (the full-size tests themselves run per mode in their own module)
from tests.helpers import assert_close
from tests.test_gpu_layers import check_gnn_stack, check_layer_backward, check_rgat_backward

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


def test_gnn_stack_with_dense_products_on_split_operands_matches_the_default_path(dev, monkeypatch):
    """TFGNN_DENSE_F16X2=1: projection / Dense products and their gradients through tfgnn_sp_gemm_nt(_sp) / _tn, operands
    split by the epilogues of the products before them.  Same stack, same weights, same dropout masks as the default
    path (Dense products in bf16x3): outputs and gradients agree to the error class of the modes."""
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput

    V, E, L, H = 3000, 40000, 4, 320
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=3)
    params = GNN.get_default_hyperparameters("rgcn")
    params.update({"hidden_dim": H, "num_layers": 2, "dense_every_num_layers": 1, "residual_every_num_layers": 10000,
                   "global_exchange_every_num_layers": 10000, "layer_input_dropout_rate": 0.1})
    X = torch.from_numpy(feats).to(dev)
    adj = tuple(torch.from_numpy(a).to(dev) for a in adjs)
    n2g = torch.zeros(V, dtype=torch.int32, device=dev)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(1)).to(dev)
    results = {}
    prev = ops.get_gemm_mode()
    ops.set_gemm_mode("f16x2")
    try:
        for flag in ("0", "1"):
            monkeypatch.setenv("TFGNN_DENSE_F16X2", flag)
            from tf2_gnn_amd.layers.message_passing import set_seed

            set_seed(7)
            gnn = GNN(params)
            gnn.dropout_seed = 11
            out = gnn(GNNInput(X, adj, n2g, 1), training=True)
            dX = gnn.backward(dOut, need_input_grad=True)
            results[flag] = (out.cpu(), dX.cpu(), [v.grad.cpu() for v in gnn.trainable_variables])
    finally:
        ops.set_gemm_mode(prev)
    o0, x0, g0 = results["0"]
    o1, x1, g1 = results["1"]
    assert_close(o1, o0, tol=1e-5, what="dense f16x2 stack output")
    assert_close(x1 / float(x0.abs().max()), x0 / float(x0.abs().max()), tol=1e-5, what="dense f16x2 stack dX")
    for a, b in zip(g1, g0):
        s = max(1.0, float(b.abs().max()))
        assert_close(a / s, b / s, tol=2e-5, what="dense f16x2 stack weight gradient")


# ---- the spread guard of the split-operand weight-gradient product --------------
def test_spread_guard_flags_wide_row_spreads_and_demotes_the_mode(dev):
    """tfgnn_sp_gemm_tn applies ONE combined per-k factor to the A fragments: a non-zero row 2^j below the largest row of
    its column block keeps 22 bits relative to itself up to j = 13, 35 - j bits after that, and drops out at j = 24
    (include/tfgnn.h).  Measured here: the error of the product against fp64, relative to sum |a||b| per entry, for row
    spreads of 2^8 .. 2^18 (no flag: fp32 class) and 2^40 (flag).  The factor pass reports spreads beyond 2^20
    through tfgnn_sp_spread_flag; tf2_gnn_amd.ops then takes the exact bf16x3 kernels until re-armed."""
    import warnings

    from tests.helpers import record_parity
    from tf2_gnn_amd import _lib, ops

    lib = _lib.load()
    K, M, N = 4096, 128, 128
    g = torch.Generator().manual_seed(0)
    a = torch.randn((K, M), generator=g)
    b = torch.randn((K, N), generator=g)

    def product_error(half_range):
        aw = a * torch.exp2(torch.randint(-half_range, half_range + 1, (K, 1), generator=g).float())
        aw[::7] = 0.0  # all-zero rows are not a spread
        got = ops.sp_gemm_tn(ops.sp_split_rows(aw.to(dev), scale_block=M), ops.sp_split_rows(b.to(dev))).cpu()
        torch.cuda.synchronize()
        ref = aw.double().t() @ b.double()
        mag = aw.double().abs().t() @ b.double().abs()
        return float(((got.double() - ref).abs() / mag).max())

    for half_range in (4, 7, 9):  # total spreads 2^8, 2^14, 2^18 (+ 2^2..3 from the rows' own maxima; the benchmark's rows: ~2^16)
        err = product_error(half_range)
        record_parity(f"sp_gemm_tn error / sum |a||b| at a row spread of 2^{2 * half_range}", max_err_over_sum_abs_products=err, bound=2e-6)
        assert err <= 2e-6, (half_range, err)
        assert lib.tfgnn_sp_spread_flag(0) == 0 and ops.get_gemm_mode() == ops.GEMM_F16X2, half_range
    err = product_error(20)  # 2^40: rows drop out
    record_parity("sp_gemm_tn error / sum |a||b| at a row spread of 2^40", max_err_over_sum_abs_products=err, bound=3e-5)
    assert err <= 3e-5  # the documented error model: the error stays absolute in the units of the largest rows
    assert lib.tfgnn_sp_spread_flag(0) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert ops.get_gemm_mode() == ops.GEMM_BF16X3  # demoted (sticky)
        assert ops.get_gemm_mode() == ops.GEMM_BF16X3
    assert sum("spread" in str(x.message) for x in w) <= 1
    ops.set_gemm_mode("f16x2")  # re-arms the guard
    assert lib.tfgnn_sp_spread_flag(0) == 0 and ops.get_gemm_mode() == ops.GEMM_F16X2


@pytest.mark.parametrize("magnitude", [1.0, 1e-9, 1e-20, 1e12])
@pytest.mark.parametrize("K", [4096, 200000])
def test_all_zero_rows_of_small_operands_do_not_trip_the_spread_guard(dev, magnitude, K):
    """Round 4 (the ppi workload): an all-zero row carries the marker scale 2^-126; the guard used to recognise it by the
    size of its FACTOR (marker / largest scale product < 1e-30), so with operands of 1e-9 - a loss gradient - every empty
    bucket looked like a row 2^-90 below the rest and the first step demoted the mode.  Zero rows are now recognised by their
    own scale: no flag at any magnitude (in-kernel factors at K = 4096, the factor pass at K = 200 000), product exact to the
    usual bound; a real spread still trips."""
    from tf2_gnn_amd import _lib, ops

    lib = _lib.load()
    ops.set_gemm_mode("f16x2")
    M, N = 256, 128
    g = torch.Generator().manual_seed(K)
    a = torch.randn((K, M), generator=g) * magnitude
    b = torch.randn((K, N), generator=g)
    a[::3] = 0.0
    a[1::3, :128] = 0.0  # zero in one scale block only
    b[5::11] = 0.0
    got = ops.sp_gemm_tn(ops.sp_split_rows(a.to(dev), scale_block=128), ops.sp_split_rows(b.to(dev))).cpu()
    torch.cuda.synchronize()
    assert lib.tfgnn_sp_spread_flag(0) == 0 and ops.get_gemm_mode() == ops.GEMM_F16X2
    ref = a.double().t() @ b.double()
    mag = a.double().abs().t() @ b.double().abs()
    assert float(((got.double() - ref).abs() / mag).max()) <= 2e-6
    a[7] *= 1e-9  # one real row 2^-30 below the rest
    ops.sp_gemm_tn(ops.sp_split_rows(a.to(dev), scale_block=128), ops.sp_split_rows(b.to(dev)))
    torch.cuda.synchronize()
    assert lib.tfgnn_sp_spread_flag(0) == 1
    ops.set_gemm_mode("f16x2")  # re-arm for the tests that follow


def test_trained_like_gradient_spreads_through_the_rgcn_layer(dev):
    """Gradients of a trained network are not N(0,1): per-node magnitudes over 1e-6 .. 1e2 on hub-normalised rows (1 / degree
    down to 1 / 200).  The split-operand backward keeps every dW entry within 1e-5 of the largest and dX within the
    suite's scaled bound (the error is absolute in the units of the largest rows, below the fp32 rounding of the sums), the
    guard reports the spread, and the NEXT call of the layer runs the exact kernels."""
    from oracle import tf2gnn_oracle as orc
    from tests.helpers import ForcedKinks, mp_weights_from_layer, random_graph, to_dev
    from tests.test_gpu_layers import _build, _to64
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    V, E, L, H = 700, 9000, 4, 128
    adjs = random_graph(V, E, L, seed=3, hub=(5, 200))
    adj_t = [torch.from_numpy(a) for a in adjs]
    layer, p = _build("RGCN", {"hidden_dim": H}, H, L)
    gen = torch.Generator().manual_seed(2)
    X = torch.randn((V, H), generator=gen)
    node_mag = torch.pow(10.0, torch.rand((V, 1), generator=gen) * 8.0 - 6.0)  # 1e-6 .. 1e2
    dOut = torch.randn((V, H), generator=gen) * node_mag
    inp = MessagePassingInput(X.to(dev), to_dev(adjs, dev))
    out = layer(inp, training=True)
    assert layer._ctx.get("f16x2")
    dX = layer.backward(dOut.to(dev)).cpu()
    torch.cuda.synchronize()
    w64 = _to64(mp_weights_from_layer(layer))
    leaves = []
    for l in range(L):
        w64["edge_mlps"][l] = [k.requires_grad_(True) for k in w64["edge_mlps"][l]]
        leaves += w64["edge_mlps"][l]
    X64 = X.double().requires_grad_(True)
    mask = (out > 0).cpu()
    with ForcedKinks(lambda i, x: mask):
        ref = orc.message_passing_call("rgcn", p, w64, X64, adj_t)
    grads = torch.autograd.grad((ref * dOut.double()).sum(), [X64] + leaves)
    for l in range(L):
        r = grads[1 + l]
        err = float((layer._edge_type_mlps.vars[l][0].grad.cpu().double() - r).abs().max()) / float(r.abs().max())
        assert err <= 1e-5, (l, err)
    # dX rows span eight orders of magnitude as well: each row against its own magnitude
    row = grads[0].abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
    assert float(((dX.double() - grads[0]).abs() / row).max()) <= 2e-5
    assert ops.get_gemm_mode() == ops.GEMM_BF16X3, "the guard must have seen the 2^27+ spread of the gradient rows"
    layer(inp, training=True)
    assert not layer._ctx.get("f16x2")


def test_a_tripped_guard_recomputes_the_first_backward_passes_of_a_stack_on_the_exact_kernels(dev):
    """the spread guard reports asynchronously - the pass that trips it has already produced its weight
    gradients.  GNN.backward therefore checks it SYNCHRONOUSLY for the first passes of a model (TFGNN_GUARD_SYNC_PASSES = 3)
    and runs a tripped pass again on the exact kernels: the gradients a caller reads come from the bf16x3 kernels."""
    import warnings

    from tf2_gnn_amd import _lib, ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    V, E, L, H = 2000, 40000, 3, 128
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=4)
    params = GNN.get_default_hyperparameters("rgcn")
    params.update({"hidden_dim": H, "num_layers": 2, "dense_every_num_layers": 10000, "residual_every_num_layers": 10000,
                   "global_exchange_every_num_layers": 10000, "layer_input_dropout_rate": 0.0})
    inp = GNNInput(torch.from_numpy(feats).to(dev), tuple(torch.from_numpy(a).to(dev) for a in adjs),
                   torch.zeros(V, dtype=torch.int32, device=dev), 1)
    gen = torch.Generator().manual_seed(9)
    # per-node gradient magnitudes over 2^60: the rows of the transposed gather are spread far beyond 2^20
    dOut = (torch.randn((V, H), generator=gen) * torch.exp2(torch.randint(-40, 20, (V, 1), generator=gen).float())).to(dev)

    def grads(mode):
        ops.set_gemm_mode(mode)
        set_seed(3)
        gnn = GNN(params)
        gnn(inp, training=True)
        gnn.backward(dOut)
        torch.cuda.synchronize()
        return gnn, [v.grad.clone() for v in gnn.trainable_variables]

    _, exact = grads("bf16x3")
    ops._spread_warned[0] = False  # (the warning is issued once per process)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gnn, got = grads("f16x2")
    assert _lib.load().tfgnn_sp_spread_flag(0) == 1 and ops.get_gemm_mode() == ops.GEMM_BF16X3
    assert any("spread" in str(x.message) for x in w)
    assert gnn._guard_sync_passes == 2
    # (the forward pass ran in f16x2 - its saved activations differ from the bf16x3 run's in the last bits - so the recomputed
    #  gradients agree with the exact run to fp32 rounding, and every weight gradient is finite and complete)
    for v, a, b in zip(gnn.trainable_variables, got, exact):
        scale = max(float(b.abs().max()), 1e-30)
        assert float((a - b).abs().max()) / scale <= 1e-5, v.name
    # the policy is visible to the caller and re-armable.  The pass reported the trip; the stack's own demotion of
    # its Dense products lasts until the mode is armed again, and the re-armed stack is checked synchronously again.
    assert gnn.guard_tripped_last_backward is True and not gnn._dense_split_ok
    ops.set_gemm_mode("f16x2")
    assert gnn._dense_f16x2(H, H) and gnn._dense_split_ok and gnn._guard_sync_passes >= 3
    # a well-conditioned pass on the re-armed stack: checked, not tripped, mode kept
    gnn(inp, training=True)
    gnn.backward(torch.randn((V, H), generator=gen).to(dev))
    assert gnn.guard_tripped_last_backward is False and ops.get_gemm_mode() == ops.GEMM_F16X2
    # after the warm-up passes a trip is only REPORTED, one pass late.  Round 6: the next query of the mode then walks the
    # stack's staged policy one step (ops._f16x2_on -> GNN._on_late_guard_trip) instead of taking the whole mode off the split
    # operands, re-arms the guard and has the stack's next passes checked synchronously again
    gnn._guard_sync_passes = 0
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        gnn(inp, training=True)
        gnn.backward(dOut)  # unchecked: whatever its products report is seen by a later query of the mode, maybe inside the pass
        torch.cuda.synchronize()
        assert ops.get_gemm_mode() == ops.GEMM_F16X2
    assert any("not checked synchronously" in str(x.message) for x in w2)
    assert _lib.load().tfgnn_sp_spread_flag(0) == 0 and gnn._guard_sync_passes == 3 and gnn.guard_state()["stage"] == "1a"
    gnn(inp, training=True)
    gnn.backward(dOut)  # checked again: recomputed stage by stage until nothing trips
    assert gnn.guard_tripped_last_backward is True
    for v, b in zip(gnn.trainable_variables, exact):
        scale = max(float(b.abs().max()), 1e-30)
        assert float((v.grad - b).abs().max()) / scale <= 1e-5, v.name
    # the periodic check: every second pass is checked synchronously (and recomputed) whatever the warm-up count says
    ops.set_gemm_mode("f16x2")
    gnn(inp, training=True)
    gnn.backward(torch.randn((V, H), generator=gen).to(dev))  # (a quiet, checked pass: the re-armed stack's policy starts over)
    gnn._guard_sync_passes = 0
    gnn.guard_check_every = 2
    gnn._backward_passes = 0
    ops._LATE_TRIP_POLICIES.clear()  # (the periodic check alone: without a staged policy a late trip demotes the mode on sight)
    for i in range(2):
        gnn(inp, training=True)
        gnn.backward(dOut)
        if i == 0:
            assert gnn.guard_tripped_last_backward is False
            torch.cuda.synchronize()
            assert _lib.load().tfgnn_sp_spread_flag(0) == 1
            ops.set_gemm_mode("f16x2")  # (the host may or may not have demoted the mode on sight: arm it for the checked pass)
            assert gnn._guard_sync_passes == 0
    assert gnn.guard_tripped_last_backward is True
    for v, b in zip(gnn.trainable_variables, exact):
        scale = max(float(b.abs().max()), 1e-30)
        assert float((v.grad - b).abs().max()) / scale <= 1e-5, v.name
    ops.set_gemm_mode("f16x2")


# ---- realistic gradient magnitudes ----------------------------------------------------
# Every parity test and the headline bench feed d out ~ N(0,1); the gradient of a real loss (a mean over 10^5 node labels:
# utils/param_helpers.py, models/graph_task_model.py:347-357) is ~1e-9 per element, a summed un-normalised one ~1e4.  The
# round-4 spread-guard bug (empty buckets taken for tiny rows) only showed at such magnitudes.  One training step per layer
# class at both: the f16x2 mode must NOT demote, and every gradient must match fp64 relative to its OWN magnitude.
_SCALED_LAYER_CASES = [
    ("rgcn_h320", "RGCN", {}, 320, 4),
    ("ggnn_h128_nonorm", "GGNN", {"normalize_by_num_incoming": False}, 128, 5),
    ("edge_mlp_linear_gelu_h256", "GNN_Edge_MLP", {"num_edge_MLP_hidden_layers": 0, "use_target_state_as_input": False,
                                                   "message_activation_function": "gelu"}, 256, 2),
    ("edge_mlp_default_h128", "GNN_Edge_MLP", {}, 128, 5),
    ("rgin_h128", "RGIN", {}, 128, 4),
]


def _assert_mode_kept():
    from tf2_gnn_amd import _lib, ops

    assert not ops.f16x2_guard_tripped_sync(), "the spread guard tripped on a uniformly scaled gradient"
    assert _lib.load().tfgnn_sp_spread_flag(0) == 0 and ops.get_gemm_mode() == ops.GEMM_F16X2


@pytest.mark.parametrize("dout_scale", [1e-9, 1e4])
@pytest.mark.parametrize("name,cls_name,over,H,L", _SCALED_LAYER_CASES, ids=[c[0] for c in _SCALED_LAYER_CASES])
def test_layer_training_step_at_loss_gradient_magnitudes(dev, name, cls_name, over, H, L, dout_scale):
    check_layer_backward(dev, f"{name}_dout{dout_scale:g}", cls_name, over, V=384, E=4200, L=L, H=H, dout_scale=dout_scale)
    _assert_mode_kept()


def test_rearming_the_mode_restores_both_demotion_stages_of_a_stack(dev, monkeypatch):
    """ADVICE r4 (guard policy re-armable), extended to stage 2: a stack that handed its Dense AND its per-relation weight
    gradients back to the exact kernels takes the split-operand routes again after ops.set_gemm_mode("f16x2"), under the
    synchronous check of the first passes."""
    from tests.helpers import random_graph, to_dev
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import RGIN, set_seed

    monkeypatch.setattr(RGIN, "GROUPED_SPLIT_MIN_ROWS", 64)
    params = GNN.get_default_hyperparameters("rgin")
    params.update({"hidden_dim": 128, "num_layers": 2, "global_exchange_every_num_layers": 10000, "layer_input_dropout_rate": 0.0,
                   "dense_every_num_layers": 2, "residual_every_num_layers": 2})
    set_seed(3)
    gnn = GNN(params)
    V, L = 600, 8
    gen = torch.Generator().manual_seed(2)
    X = torch.randn((V, 128), generator=gen).to(dev)
    dOut = torch.randn((V, 128), generator=gen).to(dev)
    inp = GNNInput(X, to_dev(random_graph(V, 900, L, seed=6), dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
    gnn(inp, training=True)
    gnn.backward(dOut)
    assert any(getattr(mp, "_grouped_tn_used", False) for mp in gnn._mp_layers)
    stages = []
    while True:
        what = gnn._demote_fragile_weight_gradients()
        if what is None:
            break
        stages.append(what)
    assert stages[-1].startswith("the per-relation MLP weight gradients") and len(stages) <= 3
    assert not gnn._dense_split_ok or not gnn._dense_f16x2(128, 128)
    assert all(mp._grouped_tn_split_ok is False for mp in gnn._mp_layers if getattr(mp, "_grouped_tn_used", False))
    gnn._guard_sync_passes = 0
    ops.set_gemm_mode("f16x2")  # re-arm
    gnn(inp, training=True)
    assert gnn._tn_demoted_epoch is None and gnn._guard_sync_passes == gnn._guard_sync_passes_init and not gnn._dense_tn_wide
    assert all(getattr(mp, "_grouped_tn_split_ok", True) for mp in gnn._mp_layers)
    gnn.backward(dOut)
    _assert_mode_kept()


def test_edge_mlp_first_layer_gradients_run_on_split_operands(dev, monkeypatch):
    """Round 5 (BASELINE configs[3], GNN_Edge_MLP with target states): the gradients of the first per-edge MLP layer - both halves,
    source and target states - as SP16-writing typed gathers + split-operand NT / two-factor TN products where two fp32 gathers
    and four bf16x3 products ran; same fp64 parity bound as the route it replaces (TFGNN_EDGE_FIRST_LAYER_F16X2=0)."""
    from tests.helpers import KernelsUsed
    from tf2_gnn_amd import ops

    with KernelsUsed() as k:
        check_layer_backward(dev, "edge_mlp_first_layer_split", "GNN_Edge_MLP", {}, V=384, E=4200, L=5, H=128)
    assert k.delta["gather_sp"] >= 2 and k.delta["sp_tn"] >= 2 and k.delta["sp_nt"] >= 2, k.delta
    assert ops.get_gemm_mode() == ops.GEMM_F16X2 and not ops.f16x2_guard_tripped_sync()
    monkeypatch.setenv("TFGNN_EDGE_FIRST_LAYER_F16X2", "0")
    with KernelsUsed() as k:
        check_layer_backward(dev, "edge_mlp_first_layer_exact", "GNN_Edge_MLP", {}, V=384, E=4200, L=5, H=128)
    assert k.delta["gather_sp"] == 0 and k.delta["sp_tn"] == 0, k.delta


@pytest.mark.parametrize("dout_scale", [1e-9, 1e4])
def test_rgat_training_step_at_loss_gradient_magnitudes(dev, dout_scale):
    check_rgat_backward(dev, 8, "tanh", V=300, E=3200, L=4, H=256, dout_scale=dout_scale)
    _assert_mode_kept()


@pytest.mark.parametrize("mp_style,over", [
    ("rgcn", {"dense_every_num_layers": 1, "residual_every_num_layers": 2}),       # Dense weight gradients on split operands
    ("ggnn", {"dense_every_num_layers": 2, "residual_every_num_layers": 1}),
    ("gnn_edge_mlp", {"dense_every_num_layers": 2, "residual_every_num_layers": 2}),
    ("rgin", {"dense_every_num_layers": 2, "residual_every_num_layers": 2, "use_inter_layer_layernorm": True}),
    ("rgat", {"dense_every_num_layers": 2, "residual_every_num_layers": 2, "num_heads": 4}),
], ids=["rgcn", "ggnn", "gnn_edge_mlp", "rgin", "rgat"])
def test_gnn_stack_gradients_are_linear_in_the_loss_gradient_magnitude(dev, mp_style, over):
    """The stack is where the guard's policy lives (GNN.backward: synchronous check of the first passes, staged demotion): at
    H = 128 the projection / Dense products and their weight gradients run on split operands.  The backward pass is LINEAR in
    d out, so the gradients of d out * 1e-9 and d out * 1e4, scaled back, must equal those of d out (same forward pass, same
    relu decisions: no kink noise - a first version compared each run with the fp64 oracle and tripped over relu units of the
    640 x 128 states flipping between fp32 and fp64, at every scale alike) to 1e-5 of each gradient's largest entry - the
    scale-1 gradients themselves are pinned to fp64 by tests/test_gpu_layers.py::test_gnn_stack_forward_backward_parity and the
    per-layer tests above.  Nothing may demote: neither the stack's Dense products (`_dense_split_ok`) nor the mode."""
    from tests.helpers import random_graph, to_dev
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    V, E, L, D, H = 640, 7000, 3, 128, 128
    params = GNN.get_default_hyperparameters(mp_style)
    params.update({"hidden_dim": H, "num_layers": 2, "global_exchange_every_num_layers": 10000})
    params.update(over)
    set_seed(17)
    gnn = GNN(params)
    gen = torch.Generator().manual_seed(3)
    X = torch.randn((V, D), generator=gen).to(dev)
    dOut = torch.randn((V, H), generator=gen).to(dev)
    inp = GNNInput(X, to_dev(random_graph(V, E, L, seed=21, hub=(3, 200)), dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)

    def grads(scale):
        gnn(inp, training=False)
        dX = gnn.backward(dOut * scale, need_input_grad=True)
        torch.cuda.synchronize()
        assert gnn.guard_tripped_last_backward in (False, None) and gnn._dense_split_ok
        _assert_mode_kept()
        return [dX / scale] + [v.grad / scale for v in gnn.trainable_variables]

    base = grads(1.0)
    assert all(bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0 for g in base)
    for scale in (1e-9, 1e4):
        for name, g1, gs in zip(["d node_features"] + [v.name for v in gnn.trainable_variables], base, grads(scale)):
            top = float(g1.abs().max())
            assert float((gs - g1).abs().max()) <= 1e-5 * top, (mp_style, scale, name, float((gs - g1).abs().max()) / top)


@pytest.mark.parametrize("H,over", [(128, {}), (256, {"num_edge_MLP_hidden_layers": 2}), (128, {"message_activation_function": "tanh",
                                                                                             "aggregation_function": "mean"}),
                                    # two hidden layers wider than one column tile: both TN operands of the middle layer come
                                    # out of grouped products with one scale per row AND tile (this raised)
                                    (512, {"num_edge_MLP_hidden_layers": 2})],
                         ids=["h128", "h256_two_hidden", "h128_tanh_mean", "h512_two_hidden"])
def test_rgin_compact_rows_on_grouped_split_operand_products(dev, monkeypatch, H, over):
    """Round 5 (BASELINE configs[4]): where most (source, type) pairs have no edge, RGIN's per-relation MLPs run over the non-empty
    rows as grouped products on split operands - tfgnn_sp_gemm_nt_grouped forward and input gradients (first layer through the
    row -> node index, hidden layers writing the next operand), one grouped two-factor TN product per MLP layer for the kernel
    gradients (tfgnn_sp_gemm_tn_grouped) - where the bf16x3 grouped kernels ran before.  Forward, dX and every kernel gradient against fp64 autograd through
    the oracle; ragged relation sizes (a hub among the targets).  Then the same with the kernel gradients demoted to the exact
    grouped kernel, as the stack's guard policy does for un-normalised sums spread over more than 2^20."""
    from tests.helpers import KernelsUsed
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers.message_passing import RGIN

    import numpy as np

    L, V = 8, 400
    # few distinct sources (most (source, type) pairs have no edge: the compact-row path), every node a target of ~10 edges (a node
    # without incoming edges would sit exactly at the kink of the output activation); ragged relation sizes
    rng = np.random.default_rng(5)
    hubs = rng.choice(V, size=70, replace=False)
    adjs = []
    for l in range(L):
        e = 900 if l < 2 else (40 if l == 5 else 350)
        adjs.append(np.stack([rng.choice(hubs[: 70 if l % 2 else 25], size=e), rng.integers(0, V, size=e)], axis=1).astype(np.int32))
    monkeypatch.setattr(RGIN, "GROUPED_SPLIT_MIN_ROWS", 64)
    with KernelsUsed() as k:
        check_layer_backward(dev, f"rgin_grouped_split_h{H}", "RGIN", over, V=V, E=0, L=L, H=H, adjs=adjs)
    layers = 2 + (1 if over.get("num_edge_MLP_hidden_layers") == 2 else 0)
    # (the kernel gradients of all relations of one MLP layer are ONE launch of the grouped two-factor product)
    assert k.delta["sp_nt"] >= 2 * layers and k.delta["sp_tn"] >= layers, k.delta
    assert ops.get_gemm_mode() == ops.GEMM_F16X2 and not ops.f16x2_guard_tripped_sync()
    monkeypatch.setattr(RGIN, "_grouped_tn_split_ok", False, raising=False)
    with KernelsUsed() as k:
        check_layer_backward(dev, f"rgin_grouped_split_exact_tn_h{H}", "RGIN", over, V=V, E=0, L=L, H=H, adjs=adjs)
    assert k.delta["sp_nt"] >= 2 * layers and k.delta["sp_tn"] == 0 and k.delta["gemm_bf16x3"] >= layers, k.delta
