"""Per-graph scale vectors (degree normalisation, mean / sqrt_n aggregators), computed by
tfgnn_graph_scales once per (graph, normalize, aggregation) and cached on the Graph."""
from __future__ import annotations

import ctypes

import torch

from .. import _lib, ops

_AGG_MODE = {"sum": 0, "max": 0, "mean": 1, "sqrt_n": 2}


def graph_scales(g: "ops.Graph", normalize: bool, aggregation: str):
    """-> (row_scale [V*L] | None, edge_weight_by_dst [E] | None, edge_weight_by_src [E] | None,
           node_scale [V] | None); None means "all ones" (lets the kernels skip the load)."""
    mode = _AGG_MODE[aggregation]
    key = ("scales", bool(normalize), mode)
    cached = g._cache.get(key)
    if cached is not None:
        return cached
    if not normalize and mode == 0:
        res = (None, None, None, None)
    elif normalize and mode == 0:
        res = (
            g.array(ops.G_INVDEG_BY_DST),
            g.array(ops.G_INVDEG_EDGE_BY_DST),
            g.array(ops.G_INVDEG_EDGE_BY_SRC),
            None,
        )
    else:
        dev = g.device
        R = g.num_nodes * g.num_edge_types
        row_scale = torch.empty(R, dtype=torch.float32, device=dev)
        node_scale = torch.empty(g.num_nodes, dtype=torch.float32, device=dev)
        ew_s = torch.empty(g.num_edges, dtype=torch.float32, device=dev)
        ew_d = torch.empty(g.num_edges, dtype=torch.float32, device=dev)
        _lib.check(
            _lib.load().tfgnn_graph_scales(
                g._h, int(normalize), mode, ops._ptr(row_scale), ops._ptr(node_scale), ops._ptr(ew_s),
                ops._ptr(ew_d), ops._stream(),
            )
        )
        res = (row_scale, ew_d if normalize else None, ew_s, node_scale)
    g._cache[key] = res
    return res


def target_multiplier(g: "ops.Graph", row_scale):
    """-> (k [V*L], ident_ptr [V*L+1], node_of_row [V*L]) for the target-state term of a linear
    edge layer; cached per row_scale identity."""
    key = ("tmul", None if row_scale is None else row_scale.data_ptr())
    cached = g._cache.get(key)
    if cached is not None:
        return cached
    dev = g.device
    R = g.num_nodes * g.num_edge_types
    k = torch.empty(R, dtype=torch.float32, device=dev)
    ident = torch.empty(R + 1, dtype=torch.int32, device=dev)
    node_of_row = torch.empty(R, dtype=torch.int32, device=dev)
    _lib.check(
        _lib.load().tfgnn_graph_target_multiplier(
            g._h, ops._ptr(row_scale), ops._ptr(k), ops._ptr(ident), ops._ptr(node_of_row), ops._stream()
        )
    )
    res = (k, ident, node_of_row)
    g._cache[key] = res
    return res
