"""Batching of graph samples on the device - mirror of GraphDataset.graph_batch_iterator_from_graph_iterator and its
helpers (tf2_gnn/data/graph_dataset.py:161-246; "next" row f2 of SURVEY.md section 8).

The reference grows Python lists graph by graph (``_add_graph_to_batch`` adds the node offset to every edge array,
``_finalise_batch`` concatenates).  Here the host only lays the graphs' arrays end to end - local node ids untouched -
and computes the per-graph prefix sums; the per-edge offset addition and ``node_to_graph_map`` run as two streaming HIP
kernels (csrc/edge.hip tfgnn_batch_*), and the result feeds ``ops.Graph`` / ``GNNInput`` directly.  Same batch-size
rule (a graph that would push the node count over ``max_nodes_per_batch`` starts a new batch, :167-171) and the same
batch keys as the reference."""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Iterable, Iterator, List, NamedTuple, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, ops


class GraphSample(NamedTuple):
    """data/graph_dataset.py:21-54: per-type adjacency lists int [E_l, 2] (local node ids), in-degree counts, node features."""

    adjacency_lists: Sequence[np.ndarray]
    type_to_node_to_num_inedges: Any
    node_features: Any


def _finalise_on_device(graphs: List[GraphSample], num_edge_types: int, device) -> Dict[str, Any]:
    """_finalise_batch (graph_dataset.py:224-246) for the graphs collected so far."""
    lib = _lib.load()
    G = len(graphs)
    node_counts = np.array([len(g.node_features) for g in graphs], dtype=np.int64)
    node_ptr = np.zeros(G + 1, dtype=np.int64)
    np.cumsum(node_counts, out=node_ptr[1:])
    V = int(node_ptr[-1])
    if V >= 2 ** 31:
        raise ValueError("batch has too many nodes for int32 node ids")
    feats = np.concatenate([np.asarray(g.node_features, dtype=np.float32).reshape(len(g.node_features), -1) for g in graphs]) \
        if G else np.zeros((0, 0), dtype=np.float32)
    node_ptr_d = torch.from_numpy(node_ptr.astype(np.int32)).to(device)
    n2g = torch.empty(V, dtype=torch.int32, device=device)
    if G:
        _lib.check(lib.tfgnn_batch_node_to_graph_map(ops._ptr(node_ptr_d), G, V, ops._ptr(n2g), ops._stream()))
    batch: Dict[str, Any] = {
        "node_features": torch.from_numpy(feats).to(device),
        "node_to_graph_map": n2g,
        "num_graphs_in_batch": G,
    }
    bad = torch.zeros(1, dtype=torch.int32, device=device)
    for t in range(num_edge_types):
        per_graph = [np.asarray(g.adjacency_lists[t], dtype=np.int32).reshape(-1, 2) for g in graphs]
        counts = np.array([a.shape[0] for a in per_graph], dtype=np.int64)
        edge_ptr = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(counts, out=edge_ptr[1:])
        E = int(edge_ptr[-1])
        out = torch.empty((E, 2), dtype=torch.int32, device=device)
        if E:
            local = torch.from_numpy(np.concatenate(per_graph)).to(device)
            edge_ptr_d = torch.from_numpy(edge_ptr.astype(np.int32)).to(device)
            _lib.check(lib.tfgnn_batch_offset_edges(ops._ptr(local), E, ops._ptr(edge_ptr_d), ops._ptr(node_ptr_d), G,
                                                    ops._ptr(out), ctypes.c_void_p(bad.data_ptr()), ops._stream()))
        batch[f"adjacency_list_{t}"] = out
    batch["_bad_local_index"] = bad  # device flag: checked lazily (``check_batch``) so that batching never synchronises
    return batch


def check_batch(batch: Dict[str, Any]) -> None:
    """Raise ValueError if a graph referred to a node outside itself (one small device read)."""
    if int(batch["_bad_local_index"].item()):
        raise ValueError("a graph sample's adjacency list refers to a node index outside the graph")


def graph_batch_iterator_from_graph_iterator(graph_sample_iterator: Iterable[GraphSample], num_edge_types: int,
                                             max_nodes_per_batch: int, device=None) -> Iterator[Dict[str, Any]]:
    """graph_dataset.py:161-181 with the finalisation on the device.  Yields batch_features dictionaries with the
    reference's keys (node_features, node_to_graph_map, num_graphs_in_batch, adjacency_list_<i>) as device tensors."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    cur: List[GraphSample] = []
    nodes = 0
    for sample in graph_sample_iterator:
        n = len(sample.node_features)
        if nodes + n > max_nodes_per_batch:  # _batch_would_be_too_full
            yield _finalise_on_device(cur, num_edge_types, device)
            cur, nodes = [], 0
        cur.append(sample)
        nodes += n
    yield _finalise_on_device(cur, num_edge_types, device)


def batch_adjacency_lists(batch: Dict[str, Any], num_edge_types: int) -> Tuple[torch.Tensor, ...]:
    return tuple(batch[f"adjacency_list_{t}"] for t in range(num_edge_types))
