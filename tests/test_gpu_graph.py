"""Bit-exact adjacency indexing: the GPU bucketing (csrc/graph.hip) against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import adjacency_oracle as ao
from tests.helpers import random_graph, to_dev

pytestmark = pytest.mark.gpu


def _check_graph(adjs, V, dev):
    from tf2_gnn_amd import ops

    g = ops.Graph(to_dev(adjs, dev), V)
    L = len(adjs)
    for by, ids in (("dst", (ops.G_ROWPTR_BY_DST, ops.G_COL_BY_DST, ops.G_EID_BY_DST, ops.G_COLL_BY_DST, ops.G_NODEPTR_BY_DST)),
                    ("src", (ops.G_ROWPTR_BY_SRC, ops.G_COL_BY_SRC, ops.G_EID_BY_SRC, ops.G_COLL_BY_SRC, ops.G_NODEPTR_BY_SRC))):
        rowptr, col, typ = ao.bucket_edges(adjs, V, by=by)
        got_rowptr = g.array(ids[0]).cpu().numpy()
        got_col = g.array(ids[1]).cpu().numpy()
        np.testing.assert_array_equal(got_rowptr, rowptr)
        np.testing.assert_array_equal(got_col, col)
        np.testing.assert_array_equal(g.array(ids[3]).cpu().numpy(), col.astype(np.int64) * L + typ)
        np.testing.assert_array_equal(g.array(ids[4]).cpu().numpy(), rowptr[:: max(L, 1)][: V + 1])
        # eid maps every bucketed edge back to its (src, dst, type) in the concatenated lists
        eid = g.array(ids[2]).cpu().numpy()
        allsrc = np.concatenate([a[:, 0] for a in adjs]) if L else np.zeros(0, np.int32)
        alldst = np.concatenate([a[:, 1] for a in adjs]) if L else np.zeros(0, np.int32)
        assert sorted(eid.tolist()) == list(range(len(allsrc)))
        other = allsrc if by == "dst" else alldst
        np.testing.assert_array_equal(other[eid], col)
    # in-degree normaliser: 1/(c + 1e-7) evaluated in fp32 like the reference
    rowptr, _, _ = ao.bucket_edges(adjs, V, by="dst")
    c = np.diff(rowptr).astype(np.float32)
    exp = np.where(c > 0, np.float32(1.0) / (c + np.float32(1e-7)), np.float32(0)).astype(np.float32)
    np.testing.assert_array_equal(g.array(ops.G_INVDEG_BY_DST).cpu().numpy(), exp)
    # src2dst is the permutation between the two edge orders
    s2d = g.array(ops.G_SRC2DST_POS).cpu().numpy()
    np.testing.assert_array_equal(g.array(ops.G_EID_BY_DST).cpu().numpy()[s2d], g.array(ops.G_EID_BY_SRC).cpu().numpy())
    g.close()


@pytest.mark.parametrize(
    "V,E,L,kw",
    [
        (5, 12, 3, {}),
        (50, 400, 4, {"empty_types": (1,)}),
        (300, 5000, 3, {"hub": (7, 700)}),        # rows > 64 -> block sort
        (2000, 30000, 2, {"hub": (11, 6000)}),    # row > 4096 -> in-place global sort
        (10, 0, 2, {"empty_types": (0, 1)}),
        (7, 30, 1, {}),
    ],
)
def test_bucketing_bit_exact(dev, V, E, L, kw):
    adjs = random_graph(V, E, L, seed=V + E, **kw)
    _check_graph(adjs, V, dev)


def test_bucketing_reference_doctest_graph(dev, kats):
    d = kats["num_incoming_doctest"]
    adjs = [np.array(a, dtype=np.int32) for a in d["adjacency_lists"]]
    _check_graph(adjs, d["num_nodes"], dev)
    from tf2_gnn_amd.layers.message_passing import calculate_type_to_num_incoming_edges

    got = calculate_type_to_num_incoming_edges(torch.zeros((5, 3), device=dev), to_dev(adjs, dev))
    np.testing.assert_array_equal(got.cpu().numpy(), np.array(d["expected"], dtype=np.float32))


def test_bucketing_rmat_and_determinism(dev):
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch

    _, adjs = make_synthetic_batch(3000, 90000, 4, 8, seed=0)
    _check_graph(adjs, 3000, dev)
    a = ops.Graph(to_dev(adjs, dev), 3000)
    b = ops.Graph(to_dev(adjs, dev), 3000)
    for i in range(8):
        assert torch.equal(a.array(i), b.array(i))


def test_out_of_range_index_raises(dev):
    from tf2_gnn_amd import ops

    bad = torch.tensor([[0, 1], [2, 5]], dtype=torch.int32, device=dev)
    with pytest.raises(ValueError, match="outside"):
        ops.Graph([bad], 5)
    neg = torch.tensor([[0, -1]], dtype=torch.int32, device=dev)
    with pytest.raises(ValueError):
        ops.Graph([neg], 5)
