"""Parity of every layer class in the GEMM modes bench.py is timed in, at widths the split-operand kernels really take
(VERDICT r2 "next round" item 1): hidden sizes 128 / 256 / 320 so that

  * mode bf16x3 runs `gemm_x3s / gemm_x3p` (N a multiple of 128, K >= 64),
  * mode f16x2 runs `gemm_sp_nt / gemm_sp_tn` on the operand the SP16-writing gather produced (RGCN / GGNN / linear
    GNN_Edge_MLP without target states) and the bf16x3 kernels everywhere else - exactly what `bench.py --workload ...`
    does,
  * mode fp32 runs the fp32-MFMA kernel only.

Which kernels ran is read from the library's launch counters (`tfgnn_launch_counts`), so a test that silently fell back
to another kernel family fails.  Same oracles and the same bounds as in tests/test_gpu_layers.py: fp32 node states
within 1e-5 * max(1, |ref|) of the fp64 oracle (north_star), gradients against fp64 autograd through the oracle.

The second half is the BENCHMARKED stack itself: `ppi_rgcn_params(320, 4)` (PPI_RGCN.json), 4 edge types, training
mode with the dropout masks the HIP path drew injected into `orc.gnn_internal_call` (tf2_gnn/layers/gnn.py:276-329)."""
import numpy as np
import pytest
import torch

from oracle import tf2gnn_oracle as orc
from tests.helpers import ForcedKinks, KernelsUsed, assert_close, random_graph, record_parity, scaled_error, to_dev
from tests.test_gpu_layers import (
    _gnn_oracle_weights,
    _to64,
    check_film,
    check_layer_backward,
    check_layer_forward,
    check_rgat_backward,
    check_weighted_sum,
)

pytestmark = [pytest.mark.gpu, pytest.mark.gemm_modes, pytest.mark.usefixtures("gemm_mode")]


def _assert_kernel_families(mode, used, split_operand_path, forward_only=False):
    """`used`: launches per kernel family during the test body."""
    if forward_only and mode == "f16x2" and split_operand_path:
        assert used["sp_nt"] >= 1, used  # (+ the SP16-writing gather for the edge-MLP family; RGAT's operand is X itself)
        return
    if mode == "fp32":
        assert used["gemm_fp32"] > 0 and used["gemm_bf16x3"] == 0 and used["sp_nt"] == 0 and used["sp_tn"] == 0, used
    elif mode == "bf16x3":
        assert used["gemm_bf16x3"] > 0 and used["sp_nt"] == 0 and used["sp_tn"] == 0, used
    else:  # f16x2
        if split_operand_path:
            assert used["sp_nt"] >= 2 and used["sp_tn"] >= 1 and used["gather_sp"] >= 2, used
        else:
            assert used["gemm_bf16x3"] > 0, used


# name, class, overrides, H, L, takes the split-operand (SP16) path in f16x2 mode
WIDE_CASES = [
    ("rgcn_h128", "RGCN", {}, 128, 4, True),
    ("rgcn_h256_tanh_mean", "RGCN", {"message_activation_function": "tanh", "aggregation_function": "mean"}, 256, 2, True),
    ("rgcn_h320", "RGCN", {}, 320, 4, True),
    # L * H not a multiple of the 128-column tile of the weight-gradient product (PPI: 3 edge types, H = 320; one edge type)
    ("rgcn_h320_three_types", "RGCN", {}, 320, 3, True),
    ("rgcn_h320_one_type", "RGCN", {}, 320, 1, True),
    ("ggnn_h320_three_types", "GGNN", {}, 320, 3, True),
    ("rgcn_h128_max", "RGCN", {"aggregation_function": "max"}, 128, 4, False),
    ("rgcn_h128_target", "RGCN", {"use_target_state_as_input": True}, 128, 4, False),
    ("ggnn_h128_nonorm", "GGNN", {"normalize_by_num_incoming": False}, 128, 5, True),  # the qm9-ggnn workload's layer
    ("ggnn_h256", "GGNN", {}, 256, 2, True),
    ("rgin_h128", "RGIN", {}, 128, 4, False),
    ("rgin_h256_norm_aggr_mlp", "RGIN", {"normalize_by_num_incoming": True, "num_aggr_MLP_hidden_layers": 1}, 256, 3, False),
    ("edge_mlp_default_h128", "GNN_Edge_MLP", {}, 128, 5, False),  # the qm9-edgemlp workload's layer (per-edge MLP)
    ("edge_mlp_src_only_h128", "GNN_Edge_MLP", {"use_target_state_as_input": False}, 128, 4, False),
    ("edge_mlp_linear_gelu_h256", "GNN_Edge_MLP", {"num_edge_MLP_hidden_layers": 0, "use_target_state_as_input": False,
                                                   "message_activation_function": "gelu"}, 256, 2, True),
]


@pytest.mark.parametrize("name,cls_name,over,H,L,sp", WIDE_CASES, ids=[c[0] for c in WIDE_CASES])
def test_wide_layer_forward_backward_parity_per_mode(dev, gemm_mode, name, cls_name, over, H, L, sp):
    """forward, dX and every weight gradient against fp64 autograd through the oracle, 384 nodes incl. a hub."""
    with KernelsUsed() as k:
        check_layer_backward(dev, name, cls_name, over, V=384, E=4200, L=L, H=H)
    _assert_kernel_families(gemm_mode, k.delta, sp)


@pytest.mark.parametrize("name,cls_name,over,H", [
    ("rgcn_sqrt_n_nonorm", "RGCN", {"aggregation_function": "sqrt_n", "normalize_by_num_incoming": False}, 128),
    ("rgcn_act_before", "RGCN", {"message_activation_before_aggregation": True, "message_activation_function": "elu"}, 128),
    ("rgin", "RGIN", {}, 512),  # the arxiv-rgin workload's width (one empty edge type)
    ("rgat_tanh_8", "RGAT", {"num_heads": 8, "message_activation_function": "tanh"}, 256),
], ids=lambda v: v if isinstance(v, str) else None)
def test_wide_layer_forward_parity_per_mode(dev, gemm_mode, name, cls_name, over, H):
    with KernelsUsed() as k:
        check_layer_forward(dev, name, cls_name, over, H, V=300, E=3600, L=4)
    _assert_kernel_families(gemm_mode, k.delta, gemm_mode == "f16x2" and k.delta["sp_nt"] > 0, forward_only=True)


@pytest.mark.parametrize("H,K,act", [(256, 8, "tanh"), (128, 8, "relu"), (128, 4, "gelu")])
def test_wide_rgat_backward_parity_per_mode(dev, gemm_mode, H, K, act):
    """BASELINE configs[2]'s layer (8 heads, H = 256) at 300 nodes: forward, dX, dW_l, d alpha_l vs fp64 autograd."""
    with KernelsUsed() as k:
        check_rgat_backward(dev, K, act, V=300, E=3200, L=4, H=H)
    if gemm_mode == "f16x2":  # Y = X W and dX on gemm_sp_nt (round 3); dW stays on the exact kernel (attention-weighted rows)
        assert k.delta["sp_nt"] >= 2 and k.delta["sp_tn"] == 0 and k.delta["gemm_bf16x3"] >= 1, k.delta
    else:
        _assert_kernel_families(gemm_mode, k.delta, False)


@pytest.mark.parametrize("name,over", [
    ("film_default", {}),
    ("film_hidden_edge_mlp_gelu", {"num_edge_MLP_hidden_layers": 1, "message_activation_function": "gelu"}),
    ("film_target_input_sum", {"use_target_state_as_input": True}),
], ids=lambda v: v if isinstance(v, str) else None)
def test_wide_film_parity_per_mode(dev, gemm_mode, name, over):
    with KernelsUsed() as k:
        check_film(dev, name + "_h128", over, V=256, E=2600, L=3, H=128)
    _assert_kernel_families(gemm_mode, k.delta, False)


@pytest.mark.parametrize("wf", ["softmax", "sigmoid", "average", "none"])
@pytest.mark.parametrize("VD,GD,hidden", [(128, 128, 128), (128, 256, 256)])
def test_wide_pooling_parity_per_mode(dev, gemm_mode, wf, VD, GD, hidden):
    """WeightedSumGraphRepresentation with MLP widths the split kernels tile (GD / hidden multiples of 128)."""
    rng = np.random.default_rng(GD)
    sizes = [int(n) for n in rng.integers(1, 24, size=40)]
    with KernelsUsed() as k:
        check_weighted_sum(dev, wf, sizes=sizes, VD=VD, GD=GD, heads=4, hidden=hidden)
    _assert_kernel_families(gemm_mode, k.delta, False)


# ---- the benchmarked stack (BENCH_rNN's model) against the oracle, training mode ----------------------------------
def _visit_leaves(obj, leaves):
    if isinstance(obj, torch.Tensor):
        obj.requires_grad_(True)
        leaves.append(obj)
    elif isinstance(obj, dict):
        for key in obj:
            _visit_leaves(obj[key], leaves)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _visit_leaves(v, leaves)


@pytest.mark.parametrize("V,E", [(700, 21000), (3000, 90000)])
def test_benchmarked_rgcn_stack_training_step_matches_fp64_oracle(dev, gemm_mode, V, E):
    """`ppi_rgcn_params(320, 4)` = the model bench.py times (projection, dropout -> RGCN x 4, Dense after layer 0,
    cross-layer gradient epilogues), 4 edge types, R-MAT edges at the benchmark's density (30 edges per node), TRAINING
    mode: the masks the HIP dropout kernel drew are read back and injected into the oracle.  Output, all five
    representations, d node_features and EVERY weight gradient against fp64 autograd through `orc.gnn_internal_call`;
    bound 1e-5 * max(1, |ref|) on states, 1e-5 of the largest entry on weight gradients, 2e-5 scaled on d node_features (the
    bound of every input-gradient check of the suite; measured values: parity_r03.json)."""
    from bench import ppi_rgcn_params
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    L, H = 4, 320
    params = ppi_rgcn_params(H, 4)
    assert params["layer_input_dropout_rate"] == 0.1
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=V)
    set_seed(V)
    gnn = GNN(params)
    X = torch.from_numpy(feats)
    inp = GNNInput(X.to(dev), to_dev(adjs, dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(V + 1))
    with KernelsUsed() as k:
        out, all_reprs = gnn(inp, training=True, return_all_representations=True)
        dX = gnn.backward(dOut.to(dev), need_input_grad=True)
    _assert_kernel_families(gemm_mode, k.delta, True)
    if gemm_mode == "f16x2":  # 4 layers x (forward + dX) on split operands, 4 weight-gradient products
        assert k.delta["sp_nt"] >= 8 and k.delta["sp_tn"] >= 4 and k.delta["gather_sp"] >= 8, k.delta
    masks = [m.cpu() for m in gnn.dropout_masks()]
    assert all(m is not None and 0.85 < float((m > 0).float().mean()) < 0.95 for m in masks)

    w = _gnn_oracle_weights(gnn)
    adj_t = [torch.from_numpy(a) for a in adjs]
    w64 = _to64(w)
    leaves = []
    _visit_leaves(w64, leaves)
    X64 = X.double().requires_grad_(True)
    # 4 x V x 320 relu units: some lie within fp32 rounding of 0, where the gradient is discontinuous (tests/helpers.py,
    # "activation kinks").  The fp64 reference is evaluated on the branch the HIP forward took: relu call i of the oracle is
    # the message activation of layer i (tf2_gnn/layers/message_passing/message_passing.py:176-177).
    relu_masks = [(mp._ctx["out"] > 0).cpu() for mp in gnn._mp_layers]
    with ForcedKinks(lambda i, x: relu_masks[i]) as kinks:
        ref64, ref_all = orc.gnn_internal_call(params, w64, X64, adj_t, dropout_masks=[m.double() for m in masks])
    assert kinks.calls == 4 and kinks.flipped <= 1e-4 * kinks.units, (kinks.calls, kinks.flipped, kinks.units)
    tag = f"benchmarked stack V={V}"
    record_parity(f"{tag} relu decisions differing from fp64", max_flipped_units=kinks.flipped, units=kinks.units, bound=1e-4 * kinks.units)
    assert len(all_reprs) == 5
    for i, (a, b) in enumerate(zip(all_reprs, ref_all)):
        assert_close(a.cpu(), b.detach().float(), tol=1e-5, what=f"{tag} representation {i}")
    assert_close(out.cpu(), ref64.detach().float(), tol=1e-5, what=f"{tag} output")
    grads = torch.autograd.grad((ref64 * dOut.double()).sum(), [X64] + leaves)
    # d node_features is a K = 320 product of the projection kernel with gradients that went through four layers: the
    # yardstick is the reference-order fp32 evaluation of the same backward pass (its error against fp64, same masks)
    X32 = X.clone().requires_grad_(True)
    with ForcedKinks(lambda i, x: relu_masks[i]):
        ref32, _ = orc.gnn_internal_call(params, w, X32, adj_t, dropout_masks=masks)
        (dX32,) = torch.autograd.grad((ref32 * dOut).sum(), [X32])
    assert_close(dX.cpu(), grads[0].float(), tol=max(1e-5, 2 * scaled_error(dX32, grads[0])), what=f"{tag} d node_features")
    ref_by_id = {id(t): gr for t, gr in zip(leaves, grads[1:])}

    def check_grad(var, leaf, what):
        r = ref_by_id[id(leaf)]
        scale = max(1.0, float(r.abs().max()))
        assert_close(var.grad.cpu() / scale, (r / scale).float(), tol=1e-5, what=f"{tag} {what}")

    check_grad(gnn._initial_projection_layer, w64["initial_projection"], "d initial projection")
    check_grad(gnn._dense_layers["0"], w64["dense"][0], "d dense 0")
    for i, mp in enumerate(gnn._mp_layers):
        for l in range(L):
            check_grad(mp._edge_type_mlps.vars[l][0], w64["mp"][i]["edge_mlps"][l][0], f"layer {i} dW_{l}")
    # the reference-order fp32 evaluation of the same step for scale: how far is ITS output from fp64?
    record_parity(f"{tag} reference-order fp32 d node_features vs fp64", max_scaled_error=scaled_error(dX32, grads[0]), bound=1e-5)
    record_parity(f"{tag} reference-order fp32 output vs fp64", max_scaled_error=scaled_error(ref32, ref64.detach()), bound=1e-5)


@pytest.mark.parametrize("dense_f16x2", ["0", "1"])
def test_dropout_applied_by_the_producer_equals_the_stand_alone_dropout(dev, monkeypatch, dense_f16x2):
    """Round 4: in f16x2 mode the op that PRODUCES a layer's input applies that layer's input dropout in its epilogue
    (tfgnn_sp_gemm_nt_dropout: no dropout pass, no stored mask - the backward pass recomputes the mask from the seed and
    takes the activation derivative at the dropped value * (1 - rate)) unless the caller asks for the intermediate results.
    The benchmarked stack (PPI_RGCN.json: H = 320, 4 layers, rate 0.1, tanh projection / Dense, relu messages), same seeds,
    both ways: output, d node_features and every weight gradient must agree to fp32 rounding (bit-equal where only relu
    layers are involved), and the fused run must not launch a dropout kernel for the layers whose producer is a
    split-operand product.  TFGNN_DENSE_F16X2=1 also moves the projection / Dense products (tanh: the derivative goes
    through the rescaled dropped value) onto that path.  Against the fp64 oracle the unfused run is held by
    test_benchmarked_rgcn_stack_training_step_matches_fp64_oracle; here the fused one is held to the same bounds through it."""
    from bench import ppi_rgcn_params
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    if ops.get_gemm_mode() != ops.GEMM_F16X2:
        pytest.skip("the fused dropout is a feature of the split-operand products")
    monkeypatch.setenv("TFGNN_DENSE_F16X2", dense_f16x2)
    V, E, L, H = 3000, 90000, 4, 320
    params = ppi_rgcn_params(H, 4)
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=5)
    set_seed(5)
    gnn = GNN(params)
    inp = GNNInput(torch.from_numpy(feats).to(dev), to_dev(adjs, dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(6)).to(dev)

    def run(all_reprs):
        calls0 = gnn._dropout_calls
        res = gnn(inp, training=True, return_all_representations=all_reprs)
        out = res[0] if all_reprs else res
        dX = gnn.backward(dOut, need_input_grad=True)
        grads = [v.grad.clone() for v in gnn.trainable_variables]
        masks = [m.clone() for m in gnn.dropout_masks()]
        fused = [bool(st.get("drop_by_producer")) for st in gnn._ctx["steps"]]  # (a "drop" spec alone: dropped by its own pass, mask not stored)
        gnn._dropout_calls = calls0  # the next run draws the same masks
        return out.clone(), dX.clone(), grads, masks, fused

    run(True)  # builds the layers
    out_u, dX_u, g_u, m_u, fused_u = run(True)
    out_f, dX_f, g_f, m_f, fused_f = run(False)
    assert fused_u == [False, dense_f16x2 == "1", False, False]  # (a Dense output is not among the returned representations)
    assert fused_f == ([True] * 4 if dense_f16x2 == "1" else [False, False, True, True]), fused_f
    for a, b in zip(m_u, m_f):
        assert torch.equal(a, b)  # the regenerated masks are the stored ones
    tol = 0.0 if dense_f16x2 == "0" else 2e-6

    def close(a, b, what):
        scale = max(1.0, float(b.abs().max()))
        err = float((a - b).abs().max()) / scale
        record_parity(f"fused vs stand-alone dropout (dense f16x2 {dense_f16x2}) {what}", max_err_over_max_entry=err, bound=tol)
        assert err <= tol, (what, err)

    close(out_f, out_u, "output")
    close(dX_f, dX_u, "d node_features")
    for v, a, b in zip(gnn.trainable_variables, g_f, g_u):
        close(a, b, "d " + v.name)


@pytest.mark.parametrize("V,E", [(3000, 9000), (3000, 90000)])
def test_forward_products_over_pattern_ordered_nodes_leave_the_training_step_unchanged(dev, monkeypatch, V, E):
    """Round 4: the forward product of the RGCN layers runs over the nodes in the order of their bucket-emptiness pattern and
    skips the all-zero type blocks of each row tile (Graph part DST_PATTERN, TFGNN_NT_SKIP_EMPTY=0 switches it off).  The
    skipped products are exact zeros and the rows go back to node order in the epilogue: the whole training step (output,
    d node_features, every weight gradient) of the benchmarked stack must be BIT-EQUAL with and without it - on a batch where
    most buckets are empty and on one where few are."""
    from bench import ppi_rgcn_params
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    if ops.get_gemm_mode() != ops.GEMM_F16X2:
        pytest.skip("the split-operand products")
    L, H = 4, 320
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=8)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(9)).to(dev)

    def run(flag):
        monkeypatch.setenv("TFGNN_NT_SKIP_EMPTY", flag)
        set_seed(5)
        gnn = GNN(ppi_rgcn_params(H, 4))
        inp = GNNInput(torch.from_numpy(feats).to(dev), to_dev(adjs, dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
        out = gnn(inp, training=True)
        dX = gnn.backward(dOut, need_input_grad=True)
        parts = gnn.graph_parts(V, [len(a) for a in adjs])
        return out.clone(), dX.clone(), [v.grad.clone() for v in gnn.trainable_variables], parts

    out0, dX0, g0, parts0 = run("0")
    out1, dX1, g1, parts1 = run("1")
    assert not parts0 & ops.G_PART_DST_PATTERN and parts1 & ops.G_PART_DST_PATTERN
    assert torch.equal(out0, out1) and torch.equal(dX0, dX1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("model,H,L,over", [
    ("ggnn", 128, 3, {}),
    ("rgat", 128, 4, {"num_heads": 8}),
    ("rgcn", 320, 4, {"use_inter_layer_layernorm": True}),  # LayerNorm between the layers: no producer epilogue for the dropout
    ("rgcn", 128, 2, {"residual_every_num_layers": 2}),     # residual sums: the mask stays a stored tensor at those layers
], ids=["ggnn", "rgat", "rgcn_layernorm", "rgcn_residual"])
def test_layer_input_dropout_without_a_stored_mask_equals_the_stored_one(dev, monkeypatch, model, H, L, over):
    """Round 4: where the layer that reads a dropped input recomputes the mask in an epilogue of its backward pass
    (MessagePassing.recomputes_input_dropout: GGNN - all three terms of d h -, RGAT, RGCN path A), GNN drops the input WITHOUT
    storing the mask (ops.dropout_forward(want_mask=False)) and hands a DropoutSpec down.  Same seeds with
    TFGNN_RECOMPUTE_DROPOUT=0 / 1: identical output (the same masks), gradients equal to fp32 rounding (the mask multiplies
    the terms of a sum instead of the sum), and no stored mask where the layer said it recomputes."""
    from bench import model_params
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    if ops.get_gemm_mode() != ops.GEMM_F16X2:
        pytest.skip("the recomputing epilogues belong to the split-operand products")
    V, E = 2000, 30000
    feats, adjs = make_synthetic_batch(V, E, L, H, seed=3)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(4)).to(dev)
    NL = 4
    params = model_params(model, H, NL, num_heads=over.get("num_heads"))
    params.update({k: v for k, v in over.items() if k != "num_heads"})

    def run(flag):
        monkeypatch.setenv("TFGNN_RECOMPUTE_DROPOUT", flag)
        set_seed(11)
        gnn = GNN(params)
        inp = GNNInput(torch.from_numpy(feats).to(dev), to_dev(adjs, dev), torch.zeros(V, dtype=torch.int32, device=dev), 1)
        gnn(inp, training=False)  # builds the layers (a layer answers recomputes_input_dropout once it is built)
        out = gnn(inp, training=True, return_all_representations=True)[0]  # (all representations: no producer-side dropout)
        dX = gnn.backward(dOut, need_input_grad=True)
        # per layer: "stored" (a mask tensor), "producer" (dropped in the epilogue of the op before - the Dense after layer 0
        # is not among the returned representations), "recomputed" (dropped by its own pass, mask not stored)
        how = ["stored" if "mask" in st else ("producer" if st.get("drop_by_producer") else "recomputed") for st in gnn._ctx["steps"]]
        masks = [m.clone() for m in gnn.dropout_masks()]
        return out.clone(), dX.clone(), [v.grad.clone() for v in gnn.trainable_variables], how, masks, [v.name for v in gnn.trainable_variables]

    out0, dX0, g0, how0, m0, names = run("0")
    out1, dX1, g1, how1, m1, _ = run("1")
    assert "recomputed" not in how0, how0
    r = int(params["residual_every_num_layers"])
    for i in range(NL):
        residual_here = i % r == 0 and not (i == 0 and r >= NL)
        if how0[i] == "stored":
            assert how1[i] == ("stored" if residual_here else "recomputed"), (i, how0, how1)
        else:
            assert how1[i] == how0[i], (i, how0, how1)
    assert "recomputed" in how1, how1
    for a, b in zip(m0, m1):
        assert torch.equal(a, b)
    assert torch.equal(out0, out1)

    def close(a, b, what):
        err = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
        assert err <= 2e-6, (what, err)

    close(dX1, dX0, "d node_features")
    for n, a, b in zip(names, g1, g0):
        close(a, b, "d " + n)
