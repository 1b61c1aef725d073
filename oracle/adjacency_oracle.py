"""Integer-side oracle (numpy): adjacency preprocessing and the (dst, edge_type) bucketing.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Set, Tuple, Union

import numpy as np


# --- tf2_gnn/data/utils.py:9-133 ---------------------------------------------------------------
def get_tied_edge_types(tie_fwd_bkwd_edges: Union[bool, List[int]], num_fwd_edge_types: int) -> Set[int]:
    """data/utils.py:61-78"""
    if isinstance(tie_fwd_bkwd_edges, list):
        return set(tie_fwd_bkwd_edges)
    if tie_fwd_bkwd_edges:
        return set(range(num_fwd_edge_types))
    return set()


def process_adjacency_lists(
    adjacency_lists: Sequence[Sequence[Tuple[int, int]]],
    num_nodes: int,
    add_self_loop_edges: bool,
    tied_fwd_bkwd_edge_types: Iterable[int],
    self_loop_edge_type: int = 0,
):
    """data/utils.py:9-58: add backward edges (tied types share the forward type, others get a new
    type appended in forward-type order, :99-113), then self loops inserted at ``self_loop_edge_type``
    (negative values count from the end, :37-53, :88-96), then in-degree counts (:116-124)."""
    tied = set(tied_fwd_bkwd_edge_types)
    lists = [list(map(tuple, a)) for a in adjacency_lists]
    n_fwd = len(lists)
    for t in range(n_fwd):
        flipped = [(d, s) for (s, d) in adjacency_lists[t]]
        if t in tied:
            lists[t] = lists[t] + flipped
        else:
            lists.append(flipped)
    if add_self_loop_edges:
        n = len(lists)
        assert -(n + 1) <= self_loop_edge_type <= n
        if self_loop_edge_type < 0:
            self_loop_edge_type += n + 1
        lists.insert(self_loop_edge_type, [(i, i) for i in range(num_nodes)])
    counts = np.zeros((len(lists), num_nodes))
    for t, edges in enumerate(lists):
        for _, d in edges:
            counts[t, d] += 1
    arrays = [
        np.array(a, dtype=np.int32) if len(a) > 0 else np.zeros((0, 2), dtype=np.int32) for a in lists
    ]
    return arrays, counts


# --- tf2_gnn/data/graph_dataset.py:161-246 --------------------------------------------------------
def batch_graph_samples(samples, num_edge_types: int, max_nodes_per_batch: int):
    """graph_batch_iterator_from_graph_iterator with _new_batch / _batch_would_be_too_full / _add_graph_to_batch /
    _finalise_batch restated: ``samples`` = [(adjacency_lists per type (local ids, [E, 2]), node_features)].  A graph
    that would push the node count over ``max_nodes_per_batch`` closes the batch (:167-171); graph i's node ids are
    shifted by the nodes of the graphs before it (:210-222); node_to_graph_map holds the graph index per node (:211-217);
    empty types become int32 [0, 2] (:239-243).  Pinned by tests/golden/reference_molecule_batch.json (the reference's
    own output).  -> list of dicts with the reference's batch keys."""
    batches, cur, nodes = [], [], 0

    def finalise(graphs):
        out = {"node_features": [], "node_to_graph_map": [], "num_graphs_in_batch": len(graphs)}
        adj = [[] for _ in range(num_edge_types)]
        offset = 0
        for gi, (lists, feats) in enumerate(graphs):
            n = len(feats)
            out["node_features"].extend(feats)
            out["node_to_graph_map"].append(np.full(n, gi, dtype=np.int32))
            for t in range(num_edge_types):
                adj[t].append(np.asarray(lists[t], dtype=np.int64).reshape(-1, 2) + offset)
            offset += n
        out["node_features"] = np.array(out["node_features"])
        out["node_to_graph_map"] = np.concatenate(out["node_to_graph_map"]) if graphs else np.zeros(0, np.int32)
        for t in range(num_edge_types):
            a = np.concatenate(adj[t]) if adj[t] else np.zeros((0, 2), dtype=np.int64)
            out[f"adjacency_list_{t}"] = a.astype(np.int32) if a.size else np.zeros((0, 2), dtype=np.int32)
        return out

    for lists, feats in samples:
        n = len(feats)
        if nodes + n > max_nodes_per_batch:
            batches.append(finalise(cur))
            cur, nodes = [], 0
        cur.append((lists, feats))
        nodes += n
    batches.append(finalise(cur))
    return batches


# --- bucketing used by the HIP path (ours; the reference has no equivalent, it concatenates edge
# --- lists and scatter-adds, message_passing.py:166-174) -----------------------------------------
def bucket_edges(adjacency_lists: Sequence[np.ndarray], num_nodes: int, by: str = "dst"):
    """Canonical CSR over rows r = node * L + edge_type.

    by="dst": row node = edge target, col = edge source   (forward gather)
    by="src": row node = edge source, col = edge target   (backward / transposed gather)
    Within a row the edges keep the order of the adjacency lists (a stable sort by row: what the device bucketing does since
    round 4 - it sorts over the row bits only; rounds 1-3 also ordered the columns of a row).
    Returns rowptr int32 [V*L+1], col int32 [E], etype int32 [E]."""
    L = len(adjacency_lists)
    srcs, dsts, types = [], [], []
    for l, adj in enumerate(adjacency_lists):
        adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
        srcs.append(adj[:, 0])
        dsts.append(adj[:, 1])
        types.append(np.full(adj.shape[0], l, dtype=np.int64))
    src = np.concatenate(srcs) if srcs else np.zeros(0, np.int64)
    dst = np.concatenate(dsts) if dsts else np.zeros(0, np.int64)
    typ = np.concatenate(types) if types else np.zeros(0, np.int64)
    row_node, col = (dst, src) if by == "dst" else (src, dst)
    key = row_node * L + typ
    order = np.argsort(key, kind="stable")
    rowptr = np.zeros(num_nodes * L + 1, dtype=np.int64)
    np.add.at(rowptr, key + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), col[order].astype(np.int32), typ[order].astype(np.int32)


def bucket_emptiness_patterns(adjacency_lists: Sequence[np.ndarray], num_nodes: int):
    """Which by-target buckets of a node hold edges: pattern[v] = sum_l 2^l [bucket (v, l) is not empty] (at most 8 edge types).
    No reference counterpart - the reference multiplies every edge - this is the specification of the device's
    TFGNN_GRAPH_PART_DST_PATTERN (include/tfgnn.h): nodes are placed in the order of ``pattern_order_key`` (many non-empty
    buckets first, equal patterns adjacent; the order INSIDE a pattern is free) and ``tile_masks`` ORs the patterns of every
    run of 128 positions."""
    L = len(adjacency_lists)
    if L > 8:
        raise ValueError("at most 8 edge types")
    pattern = np.zeros(num_nodes, dtype=np.int64)
    for l, adj in enumerate(adjacency_lists):
        adj = np.asarray(adj, dtype=np.int64).reshape(-1, 2)
        if adj.shape[0]:
            pattern[np.unique(adj[:, 1])] |= 1 << l
    return pattern


def pattern_order_key(pattern: np.ndarray) -> np.ndarray:
    """sort key of a node: (8 - number of non-empty buckets) * 256 + pattern"""
    pop = np.zeros_like(pattern)
    for b in range(8):
        pop += (pattern >> b) & 1
    return (8 - pop) * 256 + pattern


def tile_masks(pattern_by_position: np.ndarray, tile_rows: int = 128) -> np.ndarray:
    """uint8 [ceil(V / tile_rows)]: OR of the patterns of each run of ``tile_rows`` positions"""
    n = pattern_by_position.shape[0]
    pad = (-n) % tile_rows
    p = np.concatenate([pattern_by_position, np.zeros(pad, dtype=pattern_by_position.dtype)]).reshape(-1, tile_rows)
    return np.bitwise_or.reduce(p, axis=1).astype(np.uint8)
