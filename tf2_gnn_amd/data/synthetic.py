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
