"""CPU-side checks: the C-ABI library loads and exports every symbol include/tfgnn.h declares,
argument validation that needs no GPU, and the host-side mirror of the reference interface
(registry, hyper-parameter dictionaries, parameter shapes)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "tfgnn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfgnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tf2_gnn_amd import _lib

    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libtfgnn.so does not export {name}"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "ctypes table and header disagree"
    assert b"gfx950" in lib.tfgnn_version()


def test_argument_validation_without_gpu():
    from tf2_gnn_amd import _lib

    lib = _lib.load()
    # negative sizes / bad enums are rejected before any HIP call
    assert lib.tfgnn_gemm(0, 0, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, None, 0, None) == -1
    assert b"negative" in lib.tfgnn_last_error()
    assert lib.tfgnn_csr_gather_reduce(None, None, None, None, 4, None, 8, 8, None, 8, 7, 0, 0, None) == -1
    assert lib.tfgnn_activation_forward(99, None, None, 4, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-1)
    # empty problems are no-ops
    assert lib.tfgnn_gemm(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, None, 0, None) == 0
    assert lib.tfgnn_rgat_node_scores(None, None, 0, 2, 2, 4, None, None, None) == 0
    assert lib.tfgnn_rgat_node_scores(None, None, 4, 2, 3, 4, None, None, None) == -1  # 4 % 3


def test_ops_refuse_cpu_tensors():
    from tf2_gnn_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.Graph([torch.zeros((1, 2), dtype=torch.int32)], 2)


def test_registry_and_errors(kats):
    """tf2_gnn/layers/message_passing/__init__.py:9-13, message_passing.py:221-227, utils/param_helpers.py"""
    from tf2_gnn_amd.layers import get_known_message_passing_classes, get_message_passing_class
    from tf2_gnn_amd.utils import get_activation_function, get_aggregation_function

    assert get_message_passing_class("RGCN").__name__ == "RGCN"
    assert get_message_passing_class("gnn_edge_mlp").__name__ == "GNN_Edge_MLP"
    assert set(get_known_message_passing_classes()) >= {"RGCN", "RGAT", "RGIN", "GGNN", "GNN_Edge_MLP", "GNN_FiLM"}
    with pytest.raises(ValueError, match="Unknown message passing type"):
        get_message_passing_class("nope")
    with pytest.raises(ValueError, match="Unknown aggregation function"):
        get_aggregation_function("median")
    with pytest.raises(ValueError, match="Unknown activation function"):
        get_activation_function("swish")
    with pytest.raises(ValueError):  # the reference raises for "linear" too (param_helpers.py:28,36-38)
        get_activation_function("linear")
    assert get_activation_function(None) is None
    assert get_activation_function("ReLU").tfgnn_name == "relu"


@pytest.mark.parametrize("cls_name", ["MessagePassing", "GNN_Edge_MLP", "RGCN", "RGIN", "GGNN", "RGAT", "GNN_FiLM"])
def test_default_hyperparameters_match_reference(kats, cls_name):
    import tf2_gnn_amd.layers.message_passing as mp

    cls = getattr(mp, cls_name)
    got = cls.get_default_hyperparameters()
    ref_own = kats["default_hyperparameters"][cls_name]
    for k, v in ref_own.items():
        assert got[k] == v, (cls_name, k)
    # inherited keys of the base class are present everywhere
    for k in kats["default_hyperparameters"]["MessagePassing"]:
        assert k in got


def test_gnn_default_hyperparameters_match_reference(kats):
    from tf2_gnn_amd.layers import GNN

    got = GNN.get_default_hyperparameters()
    for k, v in kats["default_hyperparameters"]["GNN"].items():
        assert got[k] == v, k
    assert got["use_target_state_as_input"] is False  # merged from RGCN (gnn.py:74-78)
    assert GNN.get_default_hyperparameters("ggnn")["message_calculation_class"] == "ggnn"
    bad = dict(got, global_exchange_mode="nope")
    with pytest.raises(ValueError, match="global_exchange_mode"):
        GNN(bad)


@pytest.mark.parametrize("use_target", [False, True])
def test_rgcn_trainable_variable_shapes(kats, use_target):
    """tf2_gnn/test/layers/test_RGCN.py:15-65: exactly L bias-free kernels of shape (D or 2D, H)."""
    from tf2_gnn_amd.layers import MessagePassingInput, RGCN

    for D, L, H in kats["rgcn_shape_cases"]:
        params = RGCN.get_default_hyperparameters()
        params["hidden_dim"] = H
        params["use_target_state_as_input"] = use_target
        layer = RGCN(params)
        layer.build(MessagePassingInput((None, D), tuple((None, 2) for _ in range(L))))
        tv = layer.trainable_variables
        assert len(tv) == L and len(layer.variables) == L
        for v in tv:
            assert tuple(v.shape) == ((2 * D if use_target else D), H)


def test_rgat_trainable_variable_shapes(kats):
    """tf2_gnn/test/layers/test_RGAT.py:35-64"""
    from tf2_gnn_amd.layers import MessagePassingInput, RGAT

    for D, L, H, K in kats["rgat_shape_cases"]:
        params = RGAT.get_default_hyperparameters()
        params["hidden_dim"] = H
        params["num_heads"] = K
        layer = RGAT(params)
        layer.build(MessagePassingInput((None, D), tuple((None, 2) for _ in range(L))))
        tv = layer.trainable_variables
        assert len(tv) == 2 * L and len(layer.variables) == 2 * L
        for v in tv:
            if "kernel" in v.name:
                assert tuple(v.shape) == (D, H)
            elif "attention" in v.name:
                assert tuple(v.shape) == (K, 2 * H // K)
            else:
                raise AssertionError(v.name)


def test_gnn_build_creates_reference_layer_set():
    """gnn.py:117-232 / SURVEY 3.3: with dense_every_num_layers=10000 there is exactly one Dense (layer 0)."""
    from tf2_gnn_amd.layers import GNN, GNNInput

    params = GNN.get_default_hyperparameters()
    params.update({"hidden_dim": 8, "num_layers": 4, "dense_every_num_layers": 10000,
                   "residual_every_num_layers": 10000, "global_exchange_every_num_layers": 10000})
    gnn = GNN(params)
    gnn.build(GNNInput((None, 5), tuple((None, 2) for _ in range(3)), (None,), ()))
    names = [v.name for v in gnn.trainable_variables]
    assert sum("Dense/kernel" in n for n in names) == 1
    assert sum("MessagePassing" in n for n in names) == 4 * 3
    assert tuple(gnn.trainable_variables[0].shape) == (5, 8)
    # default GNN params (gnn.py:53-79) enable a GRU global exchange every 2 layers: layer 2 of 4 gets one
    g2 = GNN(GNN.get_default_hyperparameters())
    g2.build(GNNInput((None, 5), ((None, 2),), (None,), ()))
    assert sorted(g2._global_exchange_layers) == ["2"]
    ex_names = [v.name for v in g2.trainable_variables if "Global_Exchange" in v.name]
    # graph_global_exchange.py:141-145 + nodes_to_graph_representation.py:151: name scopes of the reference's variables
    pre = "RGCN_GNN/Layer_2/Global_Exchange/GraphGlobalGRUExchange/"
    assert {pre + "kernel", pre + "recurrent_kernel", pre + "bias",
            pre + "WeightedSumGraphRepresentation/ScoringMLP_final_layer/kernel"} <= set(ex_names)


def test_data_utils_helpers_match_reference_semantics():
    """tf2_gnn/data/utils.py:61-85 (pure-Python helpers of the batch finalisation)."""
    from tf2_gnn_amd.data import compute_number_of_edge_types, get_tied_edge_types

    assert get_tied_edge_types(True, 3) == {0, 1, 2}
    assert get_tied_edge_types(False, 3) == set()
    assert get_tied_edge_types([0, 2], 3) == {0, 2}
    assert compute_number_of_edge_types({0, 2}, 3, True) == 2 * 3 - 2 + 1
    assert compute_number_of_edge_types(set(), 4, False) == 8


def test_task_model_requires_edge_type_count_and_labels():
    from tf2_gnn_amd.tasks import GraphTaskModel, NodeMulticlassTask, QM9RegressionTask

    with pytest.raises(ValueError):
        GraphTaskModel(GraphTaskModel.get_default_hyperparameters("rgcn"))

    class DS:
        num_edge_types = 3

    with pytest.raises(ValueError, match="num_node_target_labels"):
        NodeMulticlassTask(NodeMulticlassTask.get_default_hyperparameters("rgcn"), dataset=DS())
    p = QM9RegressionTask.get_default_hyperparameters("ggnn")
    assert p["gnn_message_calculation_class"] == "ggnn" and p["out_layer_dropout_keep_prob"] == 1.0
    assert p["use_intermediate_gnn_results"] is False and "optimizer" in p


def test_round4_host_side_queries_and_validation_without_gpu():
    """Entry points of round 4 that decide or validate on the host: the shape query of the fused edge products, the ABI number,
    the argument checks of the new element-wise / grouped calls (rejected before any HIP call)."""
    from tf2_gnn_amd import _lib

    lib = _lib.load()
    assert lib.tfgnn_abi_version() == _lib.ABI_VERSION
    # heads of 4 .. 64 floats (a power of two) that fit the gather's lane groups
    table = {(256, 8): 1, (128, 8): 1, (128, 4): 1, (96, 3): 1, (64, 16): 1, (512, 8): 1, (64, 2): 1, (2048, 32): 1,
             (256, 1): 0, (96, 8): 0, (100, 4): 0, (0, 4): 0, (256, 0): 0, (30, 3): 0, (1024, 8): 0, (24, 2): 0}
    for (width, heads), want in table.items():
        assert lib.tfgnn_graph_gather_dot_supported(width, heads) == want, (width, heads)
    assert lib.tfgnn_activation_backward_mul(0, None, None, None, None, -1, None) == -1
    assert lib.tfgnn_activation_backward_mul(0, None, None, None, None, 0, None) == 0       # empty: a no-op
    assert lib.tfgnn_activation_backward_mul(99, None, None, None, None, 4, None) == -1      # (NULL pointers are caught first)
    assert lib.tfgnn_gemm_grouped_rows_grad(0, 0, None, 0, 8, 8, None, 8, None, 8, 64, None, 8, 1, None, 8, None) == 0
    assert lib.tfgnn_gemm_grouped_rows_grad(0, 2, None, 4, 8, 8, None, 8, None, 8, 64, None, 8, 1, None, 8, None) == -1
    assert lib.tfgnn_graph_gather_reduce_dot(None, 0, None, 8, None, 256, 256, None, 256, None, 256, None, None, None, 0, None) == -1
    assert lib.tfgnn_gru_gates_backward_sp_dropout(None, None, None, None, None, None, None, None, None, 0.1, 7, None, 0, 100, None, 0,
                                                   None) == -4  # H % 64 != 0: no such kernel, the caller takes the fp32 route
    assert lib.tfgnn_gru_gates_backward_sp_dropout(None, None, None, None, None, None, None, None, None, 1.5, 7, None, 4, 128, None, 0,
                                                   None) == -1  # rate outside [0, 1)


def test_layers_answer_the_stack_queries_of_round4_before_and_after_build():
    """MessagePassing.recomputes_input_dropout / graph_parts and the skip predicate are host logic: defaults, and the answers of
    a layer that cannot take the split-operand route (CPU: mode queries need no device)."""
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers.message_passing import MessagePassing
    from tf2_gnn_amd.layers.message_passing.gnn_edge_mlp import _skip_empty_blocks

    assert MessagePassing.recomputes_input_dropout(object(), 100, 32, 3) is False
    assert _skip_empty_blocks(4, 320) and _skip_empty_blocks(8, 1024) and _skip_empty_blocks(2, 16)
    assert not _skip_empty_blocks(1, 320) and not _skip_empty_blocks(9, 320) and not _skip_empty_blocks(4, 1040) and not _skip_empty_blocks(4, 24)
    assert ops.G_PARTS_DEFAULT == 31 and ops.G_PARTS_ALL == 63 and not ops.G_PARTS_DEFAULT & ops.G_PART_DST_PATTERN
    assert ops._VIEW_PARTS[ops.VIEW_BY_DST_TYPED_PATTERN] == ops.G_PART_PLAN_TYPED | ops.G_PART_DST_PATTERN
    assert ops._array_parts(ops.G_PATTERN_TILEMASK_BY_DST) == ops.G_PART_DST_PATTERN


def test_round5_host_side_queries_and_validation_without_gpu():
    """Round 5 entry points that decide or validate on the host: the slab count of the column sums, the workspace of the two-factor
    TN product, the K-split switch, the argument checks of the grouped products (rejected before any HIP call), and the row limit
    above which the host mirror runs the two-factor product as row ranges."""
    from tf2_gnn_amd import _lib, ops

    lib = _lib.load()
    assert _lib.ABI_VERSION == 4 and lib.tfgnn_abi_version() == 4
    # column sums: one slab per 64 rows above 256 rows, at most 2048 slabs; no workspace for a single slab
    assert lib.tfgnn_colsum_workspace_bytes(256, 121) == 0 and lib.tfgnn_colsum_workspace_bytes(100, 0) == 0
    assert lib.tfgnn_colsum_workspace_bytes(7110, 121) == 112 * 121 * 4
    assert lib.tfgnn_colsum_workspace_bytes(10 ** 7, 320) == 2048 * 320 * 4
    # the two-factor TN product needs the split partials plus its factor tables; more than the one-factor form never less
    one = lib.tfgnn_sp_gemm_tn_workspace_bytes(1280, 320, 30000, 1280, 320)
    two = lib.tfgnn_sp_gemm_tn_wide_workspace_bytes(1280, 320, 30000, 1280, 320)
    assert one > 1280 * 320 * 4 and two > 1280 * 320 * 4
    assert ops.TN_WIDE_MAX_ROWS == 512 * ops.TN_GROUPED_MAX_CHUNK == 512 * 2016
    # the in-launch K split is a host-side switch with counters (no device needed to flip and read it)
    import ctypes

    timed_out, launches = ctypes.c_int(-1), ctypes.c_int64(-1)
    assert lib.tfgnn_sp_gemm_nt_splitk_status(-1, ctypes.byref(timed_out), ctypes.byref(launches)) == 0  # no workspace set: no splits
    assert timed_out.value == 0 and launches.value == 0
    assert lib.tfgnn_sp_gemm_nt_splitk_status(0, None, None) == 0 and lib.tfgnn_sp_gemm_nt_splitk_status(1, None, None) == 0
    assert not hasattr(lib, "tfgnn_sp_gemm_nt_balance")  # round 6: the rejected heavy-tile helpers left the ABI (version 4)
    # grouped TN: null operands, then an unsupported width (N must tile: 128 / 256 / 320 columns)
    args = [512, 512, None, 2048, None, 512, 512, None, 2048, None, 2, None, 4, None, None, 512 * 512, 512, 1, None, 0, None]
    assert lib.tfgnn_sp_gemm_tn_grouped(*args) == -1
    assert "null pointer" in lib.tfgnn_last_error().decode()
    # compact views take the split-form gather since round 5; what they need from the handle
    assert ops._VIEW_PARTS[ops.VIEW_BY_SRC_TYPED_COMPACT] == ops.G_PART_PLAN_TYPED | ops.G_PART_COMPACT
    assert ops.sp_tile_width(512) == 256 and ops.sp_tile_width(320) == 320 and ops.sp_tile_width(121) == 0


def test_row_groups_tables_of_the_grouped_products():
    """ops.RowGroups is host logic: row tiles never straddle two groups; the K ranges of the grouped weight-gradient product are
    whole k16 steps, at most TN_GROUPED_MAX_CHUNK rows, equal shares of a group, and cover every row exactly once."""
    from tf2_gnn_amd import ops

    off = [0, 300, 305, 305, 5000, 5129, 12000]
    rg = ops.RowGroups(off, "cpu")
    assert rg.num_groups == 6 and rg.num_rows == 12000 and rg.num_tiles == sum(-(-(b - a) // 128) for a, b in zip(off, off[1:]))
    tab = rg.table.tolist()
    for r0, n, gi, _ in tab:
        assert off[gi] <= r0 and r0 + n <= off[gi + 1] and 0 < n <= 128
    ranges, first, S = rg.tn_tables()
    ranges, first = ranges.tolist(), first.tolist()
    assert S == len(ranges) == first[-1] and len(first) == 7 and first[3] == first[2]  # (the empty group has no range)
    covered = 0
    for gi in range(6):
        rows = 0
        for r0, n in ranges[first[gi]:first[gi + 1]]:
            assert off[gi] <= r0 and r0 + n <= off[gi + 1] and 0 < n <= ops.TN_GROUPED_MAX_CHUNK
            assert (r0 - off[gi]) % 16 == 0
            rows += n
        assert rows == off[gi + 1] - off[gi]
        covered += rows
    assert covered == 12000 and S <= ops.TN_GROUPED_MAX_RANGES
    assert ops.RowGroups([0, 0], "cpu").tn_tables()[2] == 0


def test_device_tanh_formula_is_relatively_accurate_restated_in_numpy():
    """csrc/common.hpp fast_tanh (TFGNN_ACT_TANH, and inside GELU, in every kernel epilogue), restated with the coefficients
    parsed from the source: the odd polynomial branch below 0.625 (fused multiply-adds emulated in float64, rounded once) and
    the 1 - 2 / (e^{2|x|} + 1) branch above it, against float64 tanh over log-spaced 1e-7 .. 10, both signs: <= 4 ulp - the
    algorithm-level pin of tests/test_gpu_ops.py::test_tanh_relative_accuracy (which measures the device, native exp included)."""
    import os
    import re

    import numpy as np

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tf2_gnn_amd", "csrc", "common.hpp")).read()
    body = src[src.index("float fast_tanh(float x)"):]
    body = body[: body.index("}")]
    coef = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    assert len(coef) == 5 and "0.625f" in body
    f32 = np.float32

    def fma(a, b, c):  # one rounding
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    x = np.concatenate([np.logspace(-7, 1, 4001), -np.logspace(-7, 1, 4001)]).astype(np.float32)
    u = (x * x).astype(f32)
    q = fma(u, np.full_like(u, coef[0]), np.full_like(u, coef[1]))
    for c in coef[2:]:
        q = fma(u, q, np.full_like(u, c))
    small = fma((x * u).astype(f32), q, x)
    t = np.exp((f32(2.0) * np.abs(x)).astype(np.float64)).astype(f32)  # (a correctly rounded exp; the device's native one: GPU test)
    with np.errstate(over="ignore"):
        big = np.copysign(f32(1.0) - f32(2.0) / (t + f32(1.0)), x).astype(f32)
    got = np.where(np.abs(x) < f32(0.625), small, big)
    ref = np.tanh(x.astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    err = np.abs(got.astype(np.float64) - ref) / ulp
    assert float(err.max()) <= 4.0, (float(err.max()), float(x[int(err.argmax())]))
    assert float(np.abs(got[np.abs(x) < 1e-4] - x[np.abs(x) < 1e-4]).max()) == 0.0  # tanh(x) == x to fp32 down there


def test_captured_step_fails_loudly_without_a_device():
    """capture.CapturedStep is host plumbing around the HIP path: without a ROCm device it raises (no CPU fallback), a second
    capture of the same object is refused, and nothing runs at construction."""
    import torch

    from tf2_gnn_amd import CapturedStep

    calls = []
    step = CapturedStep(lambda: calls.append(1), warmup=2)
    assert not step.captured and step.replays == 0 and calls == []
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            step.capture()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            step.replay()
        assert calls == []


def test_layer_entry_points_validate_their_arguments_without_a_gpu():
    """tfgnn_mp_forward / tfgnn_mp_backward (round 6, include/tfgnn.h): the argument structs of the binding have the library's
    size (the call checks struct_size first), NULL handles are rejected before any HIP call."""
    import ctypes

    from tf2_gnn_amd import _lib

    lib = _lib.load()
    for fn, struct in ((lib.tfgnn_mp_forward, _lib.MpForwardArgs), (lib.tfgnn_mp_backward, _lib.MpBackwardArgs)):
        a = struct()
        a.struct_size = ctypes.sizeof(struct) - 8  # a binding built against another header
        assert fn(ctypes.byref(a), None) == -1 and b"struct_size" in lib.tfgnn_last_error()
        a.struct_size = ctypes.sizeof(struct)
        a.kind = 7
        assert fn(ctypes.byref(a), None) == -1 and b"unknown layer kind" in lib.tfgnn_last_error()
        a.kind = 0
        assert fn(ctypes.byref(a), None) == -1 and b"NULL pointer" in lib.tfgnn_last_error()
        assert fn(None, None) == -1
