"""Independent cross-checks of the oracle's [ext] restatements (semantics that live in TensorFlow / Keras / dpu_utils,
which cannot be imported here): every one is compared with a SECOND implementation that shares no code with
oracle/tf2gnn_oracle.py - torch.nn modules and functionals with their own (documented) conventions, or a plain
python-loop evaluation of the published formula - so that a typo in the restatement cannot pass silently.
("what's weak" 1: GRUCell, dpu_utils MLP, segment softmax epsilon, LayerNorm epsilon, leaky-relu alpha,
unsorted_segment_max empty value.)  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import tf2gnn_oracle as orc


def test_gru_cell_matches_torch_grucell():
    """Keras GRUCell(reset_after=True): z | r | h gate order, h' = z h + (1 - z) c, c = tanh(x_h + r (h W_h + b_h)).
    torch.nn.GRUCell is the same reset-after cell with gates ordered r | z | n and h' = (1 - z) n + z h."""
    torch.manual_seed(0)
    H, D, V = 7, 5, 11
    cell = torch.nn.GRUCell(D, H).double()
    x, h = torch.randn(V, D).double(), torch.randn(V, H).double()

    def regroup(w):  # torch rows r | z | n  ->  Keras columns z | r | h
        r, z, n = w[:H], w[H:2 * H], w[2 * H:]
        return torch.cat([z, r, n], dim=0)

    kernel = regroup(cell.weight_ih.detach()).t().contiguous()            # [D, 3H]
    rec = regroup(cell.weight_hh.detach()).t().contiguous()               # [H, 3H]
    bias = torch.stack([regroup(cell.bias_ih.detach()), regroup(cell.bias_hh.detach())])  # [2, 3H]
    got = orc.gru_cell(x, h, kernel, rec, bias)
    want = cell(x, h).detach()
    assert float((got - want).abs().max()) < 1e-12


def test_layer_norm_matches_torch_layernorm_eps_1e3():
    torch.manual_seed(1)
    x = torch.randn(13, 9).double() * 3 + 1
    ln = torch.nn.LayerNorm(9, eps=1e-3).double()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(9))
        ln.bias.copy_(torch.randn(9))
    got = orc.layer_norm(x, ln.weight.detach(), ln.bias.detach())
    assert float((got - ln(x).detach()).abs().max()) < 1e-12
    # the epsilon matters at this precision: 1e-5 (torch's default) is a different function
    other = torch.nn.functional.layer_norm(x, (9,), ln.weight.detach(), ln.bias.detach(), eps=1e-5)
    assert float((got - other).abs().max()) > 1e-6


@pytest.mark.parametrize("name,fn", [
    ("tanh", torch.tanh), ("relu", torch.relu),
    ("leaky_relu", lambda x: torch.where(x > 0, x, 0.2 * x)),                    # tf.nn.leaky_relu default alpha
    ("elu", lambda x: torch.where(x > 0, x, torch.expm1(x))),
    ("selu", lambda x: 1.0507009873554805 * torch.where(x > 0, x, 1.6732632423543772 * torch.expm1(x))),
    ("gelu", lambda x: torch.nn.functional.gelu(x, approximate="tanh")),          # utils/activation.py:7-14
])
def test_activations_match_independent_formulas(name, fn):
    x = torch.linspace(-6, 6, 241, dtype=torch.float64)
    got = orc.get_activation_function(name)(x)
    assert float((got - fn(x)).abs().max()) < 1e-12, name
    assert orc.get_activation_function(name.upper()) is not None  # case-insensitive (param_helpers.py:25)


def test_linear_activation_raises_like_the_reference():
    with pytest.raises(ValueError):
        orc.get_activation_function("linear")  # param_helpers.py:28,36-38
    assert orc.get_activation_function_by_name("linear") is None  # dpu_utils: identity


def _segments(ids, n):
    return [np.nonzero(ids == s)[0] for s in range(n)]


def test_segment_reductions_match_python_loops_incl_empty_segments():
    rng = np.random.default_rng(0)
    n, M, H = 6, 40, 3
    ids = rng.integers(0, n - 1, size=M)  # segment n-1 stays empty
    ids[ids == 2] = 3                      # and so does segment 2
    data = rng.standard_normal((M, H))
    d, i = torch.from_numpy(data), torch.from_numpy(ids)
    segs = _segments(ids, n)
    lowest = np.finfo(np.float64).min
    want = {
        "sum": np.stack([data[s].sum(0) if len(s) else np.zeros(H) for s in segs]),
        "mean": np.stack([data[s].sum(0) / max(len(s), 1) for s in segs]),
        "sqrt_n": np.stack([data[s].sum(0) / math.sqrt(max(len(s), 1)) for s in segs]),
        "max": np.stack([data[s].max(0) if len(s) else np.full(H, lowest) for s in segs]),  # tf: dtype's lowest value
    }
    for name, w in want.items():
        got = orc.get_aggregation_function(name)(d, i, n).numpy()
        np.testing.assert_allclose(got, w, rtol=1e-13, atol=1e-13, err_msg=name)
    f32 = orc.unsorted_segment_max(d.float(), i, n)
    assert float(f32[2, 0]) == float(np.finfo(np.float32).min)


def test_segment_softmax_matches_dense_softmax_per_segment():
    """dpu_utils unsorted_segment_softmax = per-segment softmax up to the 1e-7 added to the denominator (>= 1 after
    the max subtraction); the log form has no epsilon: exp(log_softmax) sums to one exactly."""
    rng = np.random.default_rng(1)
    n, M = 5, 37
    ids = rng.integers(0, n, size=M)
    logits = rng.standard_normal(M) * 4
    l, i = torch.from_numpy(logits), torch.from_numpy(ids)
    got = orc.unsorted_segment_softmax(l, i, n).numpy()
    got_log = orc.unsorted_segment_log_softmax(l, i, n).numpy()
    for s in _segments(ids, n):
        if len(s) == 0:
            continue
        dense = torch.softmax(torch.from_numpy(logits[s]), dim=0).numpy()
        assert np.max(np.abs(got[s] - dense)) <= 1.01e-7 * np.max(dense)   # the epsilon's whole effect
        np.testing.assert_allclose(np.exp(got_log[s]), dense, rtol=1e-12)
        assert abs(np.exp(got_log[s]).sum() - 1.0) < 1e-12


def test_mlp_is_a_stack_of_bias_free_dense_layers_with_relu_between():
    """dpu_utils.tf2utils.MLP as the call sites pin it (test/layers/test_RGCN.py: one (Din, H) kernel per edge type for
    0 hidden layers; an int n = n hidden layers of out_size units): torch.nn.Sequential of Linear(bias=False) + ReLU."""
    torch.manual_seed(2)
    sizes = orc.mlp_hidden_sizes(6, 2) + [6]
    assert sizes == [6, 6, 6] and orc.mlp_hidden_sizes(6, [4, 3]) == [4, 3]
    layers, kernels, last = [], [], 5
    for j, sz in enumerate(sizes):
        lin = torch.nn.Linear(last, sz, bias=False).double()
        kernels.append(lin.weight.detach().t().contiguous())  # Keras kernel [in, out]
        layers.append(lin)
        if j < len(sizes) - 1:
            layers.append(torch.nn.ReLU())
        last = sz
    x = torch.randn(9, 5).double()
    want = torch.nn.Sequential(*layers)(x).detach()
    assert float((orc.mlp_forward(x, kernels) - want).abs().max()) < 1e-12


def test_in_degree_counts_match_bincount():
    rng = np.random.default_rng(3)
    V = 17
    adjs = [rng.integers(0, V, size=(e, 2)).astype(np.int32) for e in (30, 0, 5)]
    got = orc.calculate_type_to_num_incoming_edges(torch.zeros(V, 3), [torch.from_numpy(a) for a in adjs]).numpy()
    want = np.stack([np.bincount(a[:, 1], minlength=V) for a in adjs]).astype(np.float32)
    np.testing.assert_array_equal(got, want)


def test_sigmoid_cross_entropy_and_f1_match_independent_forms():
    torch.manual_seed(4)
    x, z = torch.randn(8, 5).double() * 3, (torch.rand(8, 5) > 0.5).double()
    want = torch.nn.functional.binary_cross_entropy_with_logits(x, z, reduction="none")
    assert float((orc.sigmoid_cross_entropy_with_logits(x, z) - want).abs().max()) < 1e-12
    pred = (torch.sigmoid(x) >= 0.5).double()  # round(sigmoid) - no logits at exactly 0 in this sample
    tp = float((pred * z).sum()); fp = float((pred * (1 - z)).sum()); fn = float(((1 - pred) * z).sum())
    f1 = 2 * tp / (2 * tp + fp + fn)
    got, counts = orc.micro_f1(x, z)
    assert abs(got - f1) < 1e-12 and counts == (int(tp), int(fp), int(fn))


def test_bucket_emptiness_patterns_hand_worked():
    """oracle/adjacency_oracle.py: the specification of the device's pattern order (NOTEBOOK.md 4.8) on a case small enough to
    check by hand: 5 nodes, 3 edge types."""
    import numpy as np

    from oracle import adjacency_oracle as ao

    adjs = [np.array([[0, 1], [2, 1], [3, 4]]),        # type 0 enters nodes 1, 4
            np.array([[1, 1], [0, 2]]),                # type 1 enters nodes 1, 2
            np.zeros((0, 2), dtype=np.int64)]          # type 2: no edges
    pat = ao.bucket_emptiness_patterns(adjs, 5)
    assert pat.tolist() == [0, 3, 2, 0, 1]
    key = ao.pattern_order_key(pat)
    assert key.tolist() == [8 * 256, 6 * 256 + 3, 7 * 256 + 2, 8 * 256, 7 * 256 + 1]
    order = np.argsort(key, kind="stable")
    assert order.tolist() == [1, 4, 2, 0, 3]           # two buckets first, then pattern 1 before pattern 2, the empty nodes last
    assert ao.tile_masks(pat[order], tile_rows=2).tolist() == [3, 2, 0]
    assert ao.tile_masks(pat[order]).tolist() == [3]
    # consistent with the bucketing oracle: a set bit <=> a non-empty row of the by-target CSR
    rowptr, _, _ = ao.bucket_edges(adjs, 5, by="dst")
    nonempty = (np.diff(rowptr) > 0).reshape(5, 3)
    assert ((nonempty * (1 << np.arange(3))).sum(axis=1) == pat).all()
    with np.testing.assert_raises(ValueError):
        ao.bucket_emptiness_patterns([np.zeros((0, 2))] * 9, 3)
