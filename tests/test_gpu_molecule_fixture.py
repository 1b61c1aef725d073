"""The reference's real-graph fixture (tf2_gnn/test/test_datasets/train.jsonl.gz, 10 molecules)
as a parity batch: tests/golden/reference_molecule_batch.json holds the batches the REFERENCE's data pipeline produced.
  * device-side batching (tf2_gnn_amd.data.batching, tfgnn_batch_*) and process_adjacency_lists (tfgnn_adjacency_*)
    reproduce them bit for bit from the raw per-graph data;
  * that batch goes through all six message passing classes and both pooling layers, HIP against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import tf2gnn_oracle as orc
from tests.helpers import assert_close, mp_weights_from_layer, to_dev

# every test of this module runs in the three GEMM modes (conftest.py: gemm_modes)
pytestmark = [pytest.mark.gpu, pytest.mark.gemm_modes, pytest.mark.usefixtures("gemm_mode")]


@pytest.fixture(scope="module")
def fixture():
    return json.loads((Path(__file__).resolve().parent / "golden" / "reference_molecule_batch.json").read_text())


@pytest.mark.parametrize("cfg_idx", [0, 1])
def test_device_batching_reproduces_the_reference_batches(dev, fixture, cfg_idx):
    from tf2_gnn_amd import data

    cfg = fixture["configs"][cfg_idx]
    p = cfg["params"]
    tied = data.get_tied_edge_types(p["tie_fwd_bkwd_edges"], p["num_fwd_edge_types"])
    samples = []
    for g, ref in zip(fixture["graphs"], cfg["samples"]):
        raw = [torch.tensor(a, dtype=torch.int32).reshape(-1, 2).to(dev) for a in g["adjacency_lists"]]
        adj, counts = data.process_adjacency_lists(raw, len(g["node_features"]), p["add_self_loop_edges"], tied)
        for got, exp in zip(adj, ref["adjacency_lists"]):
            assert np.array_equal(got.cpu().numpy(), np.array(exp, dtype=np.int32).reshape(-1, 2))
        assert np.array_equal(counts.cpu().numpy(), np.array(ref["type_to_node_to_num_inedges"], dtype=np.float32))
        samples.append(data.GraphSample([a.cpu().numpy() for a in adj], counts.cpu().numpy(), g["node_features"]))
    batches = list(data.graph_batch_iterator_from_graph_iterator(iter(samples), cfg["num_edge_types"], p["max_nodes_per_batch"], dev))
    assert len(batches) == len(cfg["batches"])
    for got, exp in zip(batches, cfg["batches"]):
        data.check_batch(got)
        assert got["num_graphs_in_batch"] == exp["num_graphs_in_batch"]
        assert np.array_equal(got["node_to_graph_map"].cpu().numpy(), np.array(exp["node_to_graph_map"], dtype=np.int32))
        assert np.array_equal(got["node_features"].cpu().numpy(), np.array(exp["node_features"], dtype=np.float32))
        for t in range(cfg["num_edge_types"]):
            a = got[f"adjacency_list_{t}"]
            assert a.dtype == torch.int32 and tuple(a.shape)[1:] == (2,)
            assert np.array_equal(a.cpu().numpy(), np.array(exp["adjacency_lists"][t], dtype=np.int32).reshape(-1, 2))


def test_device_batching_flags_bad_local_indices(dev):
    from tf2_gnn_amd import data

    good = data.GraphSample([np.array([[0, 1], [1, 2]], dtype=np.int32)], None, np.zeros((3, 4), dtype=np.float32))
    bad = data.GraphSample([np.array([[0, 3]], dtype=np.int32)], None, np.zeros((3, 4), dtype=np.float32))  # node 3 of 3
    (b,) = list(data.graph_batch_iterator_from_graph_iterator(iter([good, bad]), 1, 100, dev))
    with pytest.raises(ValueError):
        data.check_batch(b)


LAYER_CASES = [
    ("RGCN", {}), ("RGIN", {}), ("GGNN", {}), ("GNN_Edge_MLP", {}), ("GNN_FiLM", {}), ("RGAT", {"num_heads": 4}),
    ("RGCN", {"aggregation_function": "max"}), ("RGIN", {"aggregation_function": "mean", "num_aggr_MLP_hidden_layers": 1}),
]


@pytest.mark.parametrize("cls_name,over", LAYER_CASES, ids=[c[0] + ("_" + "_".join(map(str, c[1].values())) if c[1] else "") for c in LAYER_CASES])
def test_molecule_batch_through_every_layer_class(dev, fixture, cls_name, over):
    """The first reference batch (10 molecules, 5 edge types incl. self loops, 35 one-hot node features) through an input
    projection + the layer, forward, against the oracle (fp64) - the realistic small batch the reference itself trains
    on in test/test_training_loop.py."""
    from tests.test_gpu_full_size import _build
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import MessagePassingInput

    cfg = fixture["configs"][0]
    b = cfg["batches"][0]
    L = cfg["num_edge_types"]
    feats = torch.tensor(b["node_features"], dtype=torch.float32)
    adjs = [np.array(a, dtype=np.int32).reshape(-1, 2) for a in b["adjacency_lists"]]
    H = 32
    g = torch.Generator().manual_seed(0)
    proj = torch.randn((feats.shape[1], H), generator=g) * 0.3
    X = torch.tanh(feats @ proj)
    layer, p = _build(cls_name, dict(over, hidden_dim=H), H, L)
    out = layer(MessagePassingInput(X.to(dev), to_dev(adjs, dev)), training=False)
    ref = orc.message_passing_call(cls_name, p, _to64(mp_weights_from_layer(layer)), X.double(), [torch.from_numpy(a) for a in adjs])
    assert_close(out.cpu(), ref.float(), tol=1e-5, what=f"molecule batch {cls_name} {over}")


def _to64(w):
    if isinstance(w, torch.Tensor):
        return w.double()
    if isinstance(w, dict):
        return {k: _to64(v) for k, v in w.items()}
    if isinstance(w, (list, tuple)):
        return [_to64(v) for v in w]
    return w


@pytest.mark.parametrize("wf", ["softmax", "sigmoid", "average", "none"])
def test_molecule_batch_pooling(dev, fixture, wf):
    from tf2_gnn_amd.layers import NodesToGraphRepresentationInput, WeightedSumGraphRepresentation

    b = fixture["configs"][0]["batches"][0]
    feats = torch.tensor(b["node_features"], dtype=torch.float32)
    n2g = torch.tensor(b["node_to_graph_map"], dtype=torch.int32)
    G = b["num_graphs_in_batch"]
    from tests.test_gpu_layers import _pool_weights

    layer = WeightedSumGraphRepresentation(16, 4, weighting_fun=wf, scoring_mlp_layers=[8], transformation_mlp_layers=[8])
    out = layer(NodesToGraphRepresentationInput(feats.to(dev), n2g.to(dev), G), training=False)
    cfg = {"graph_representation_size": 16, "num_heads": 4, "weighting_fun": wf,
           "scoring_mlp_activation_fun": "ReLU", "transformation_mlp_activation_fun": "ReLU"}
    w = _pool_weights(layer)
    w64 = {k: ([t.double() for t in ks], [None if b is None else b.double() for b in bs]) for k, (ks, bs) in w.items()}
    ref = orc.weighted_sum_graph_representation(cfg, w64, feats.double(), n2g, G)
    assert_close(out.cpu(), ref.float(), tol=1e-5, what=f"molecule batch pooling {wf}")
