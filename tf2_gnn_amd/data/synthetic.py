"""Synthetic graph batches shaped like the reference's inputs.

The reference's data sets (PPI, QM9) are not available offline; bench.py and the parity tests use
R-MAT graphs with the batch layout ``GraphDataset._finalise_batch`` produces
(tf2_gnn/data/graph_dataset.py:224-246): float32 node features [V, D], one int32 [E_l, 2]
(source, target) list per edge type, unsorted, duplicates allowed.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def rmat_edges(num_nodes: int, num_edges: int, rng: np.random.Generator, abcd=(0.57, 0.19, 0.19, 0.05)) -> np.ndarray:
    """R-MAT: recursive quadrant choice over ceil(log2 V) levels; samples outside [0, V) are
    redrawn; duplicates are kept (the reference never de-duplicates).  -> int32 [E, 2] (src, dst)."""
    scale = max(1, int(np.ceil(np.log2(max(num_nodes, 2)))))
    a, b, c, _ = abcd
    src = np.zeros(0, dtype=np.int64)
    dst = np.zeros(0, dtype=np.int64)
    while src.shape[0] < num_edges:
        n = int((num_edges - src.shape[0]) * 1.3) + 16
        s = np.zeros(n, dtype=np.int64)
        t = np.zeros(n, dtype=np.int64)
        for _ in range(scale):
            r = rng.random(n)
            s_bit = (r >= a + b).astype(np.int64)
            t_bit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)
            s = (s << 1) | s_bit
            t = (t << 1) | t_bit
        ok = (s < num_nodes) & (t < num_nodes)
        src = np.concatenate([src, s[ok]])
        dst = np.concatenate([dst, t[ok]])
    return np.stack([src[:num_edges], dst[:num_edges]], axis=1).astype(np.int32)


def make_synthetic_batch(
    num_nodes: int, num_edges: int, num_edge_types: int, feature_dim: int, seed: int = 0
) -> Tuple[np.ndarray, List[np.ndarray]]:
    """-> (node_features float32 [V, D] ~ N(0,1), adjacency_lists: L x int32 [E_l, 2]); the edge
    type of every R-MAT edge is uniform in [0, L)."""
    rng = np.random.default_rng(seed)
    edges = rmat_edges(num_nodes, num_edges, rng)
    types = rng.integers(0, num_edge_types, size=num_edges)
    adjacency_lists = [np.ascontiguousarray(edges[types == l]) for l in range(num_edge_types)]
    feats = rng.standard_normal((num_nodes, feature_dim), dtype=np.float32)
    return feats, adjacency_lists


def make_qm9_shaped_batch(num_graphs: int, seed: int = 0, feature_dim: int = 128):
    """QM9-shaped batch: graphs of 5..13 nodes, a random tree + ~0.8 extra bonds, 4 bond types
    tied fwd/bkwd + self-loop type 0 (tf2_gnn/data/qm9_dataset.py:54-77) -> 5 edge types.
    -> (node_features [V, D], adjacency_lists, node_to_graph_map [V] (sorted), node offsets [G + 1])."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(5, 14, size=num_graphs)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    V = int(offs[-1])
    n2g = np.repeat(np.arange(num_graphs, dtype=np.int32), sizes)
    # tree edges: node i>0 of a graph attaches to a random earlier node of the same graph
    local = np.arange(V) - offs[n2g]
    child = np.where(local > 0)[0]
    parent = offs[n2g[child]] + (rng.random(child.shape[0]) * local[child]).astype(np.int64)
    extra_g = np.where(rng.random(num_graphs) < 0.8)[0]
    ea = offs[extra_g] + (rng.random(extra_g.shape[0]) * sizes[extra_g]).astype(np.int64)
    eb = offs[extra_g] + (rng.random(extra_g.shape[0]) * sizes[extra_g]).astype(np.int64)
    src = np.concatenate([child, ea])
    dst = np.concatenate([parent, eb])
    bond = rng.integers(1, 5, size=src.shape[0])
    adjs = [np.stack([np.arange(V), np.arange(V)], axis=1).astype(np.int32)]  # type 0: self loops
    for b in range(1, 5):
        m = bond == b
        fwd = np.stack([src[m], dst[m]], axis=1)
        adjs.append(np.concatenate([fwd, fwd[:, ::-1]], axis=0).astype(np.int32))  # tied fwd/bkwd
    feats = rng.standard_normal((V, feature_dim), dtype=np.float32)
    return feats, adjs, n2g, offs


def make_zipf_typed_batch(num_nodes: int, num_edges: int, num_edge_types: int, feature_dim: int, seed: int = 0):
    """ogbn-arxiv-scale stand-in: one R-MAT graph whose edge types follow Zipf(1.0) over
    ``num_edge_types`` (ragged relation groups)."""
    rng = np.random.default_rng(seed)
    edges = rmat_edges(num_nodes, num_edges, rng)
    pz = 1.0 / np.arange(1, num_edge_types + 1)
    types = rng.choice(num_edge_types, size=num_edges, p=pz / pz.sum())
    adjs = [np.ascontiguousarray(edges[types == l]) for l in range(num_edge_types)]
    feats = rng.standard_normal((num_nodes, feature_dim), dtype=np.float32)
    return feats, adjs


def make_ppi_shaped_batch(num_graphs: int = 3, nodes_per_graph: int = 2370, avg_in_degree: int = 14, feature_dim: int = 50,
                          num_labels: int = 121, seed: int = 0):
    """PPI stand-in (the real data is not available offline): ``num_graphs`` R-MAT graphs of
    ``nodes_per_graph`` nodes with ``avg_in_degree`` forward edges per node, one forward edge type - the batch finalisation
    (tf2_gnn/data/utils.py:9-58 through ``process_adjacency_lists``: self loops + backward edges -> 3 types,
    ppi_dataset.py:50-58) happens on the device.  -> (features float32 [V, 50], forward edges int32 [E, 2] with node ids of the
    disjoint union, node_to_graph_map int32 [V], labels float32 [V, 121] in {0, 1})."""
    rng = np.random.default_rng(seed)
    edges = []
    for g in range(num_graphs):
        e = rmat_edges(nodes_per_graph, nodes_per_graph * avg_in_degree, rng)
        edges.append(e + g * nodes_per_graph)
    V = num_graphs * nodes_per_graph
    feats = rng.standard_normal((V, feature_dim), dtype=np.float32)
    labels = (rng.random((V, num_labels)) < 0.3).astype(np.float32)
    n2g = np.repeat(np.arange(num_graphs, dtype=np.int32), nodes_per_graph)
    return feats, np.ascontiguousarray(np.concatenate(edges, axis=0), dtype=np.int32), n2g, labels

