"""Adjacency-list finalisation on the device - mirror of tf2_gnn/data/utils.py ("next" row f2 of SURVEY.md
section 8): backward edges, self loops and per-type in-degree counts for one batch, as three streaming HIP
kernels (csrc/edge.hip tfgnn_adjacency_*) instead of Python loops over every edge."""
from typing import List, Sequence, Set, Tuple, Union

import torch

from .. import _lib, ops


def get_tied_edge_types(tie_fwd_bkwd_edges: Union[bool, List[int]], num_fwd_edge_types: int) -> Set[int]:
    """data/utils.py:61-78"""
    if isinstance(tie_fwd_bkwd_edges, list):
        return set(tie_fwd_bkwd_edges)
    if tie_fwd_bkwd_edges:
        return set(range(num_fwd_edge_types))
    return set()


def compute_number_of_edge_types(tied_fwd_bkwd_edge_types: Set[int], num_fwd_edge_types: int, add_self_loop_edges: bool) -> int:
    """data/utils.py:81-85"""
    return 2 * num_fwd_edge_types - len(tied_fwd_bkwd_edge_types) + int(add_self_loop_edges)


def process_adjacency_lists(
    adjacency_lists: Sequence[torch.Tensor],
    num_nodes: int,
    add_self_loop_edges: bool,
    tied_fwd_bkwd_edge_types: Set[int],
    self_loop_edge_type: int = 0,
) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """data/utils.py:9-58 on device tensors: ``adjacency_lists[l]`` is int32 [E_l, 2] (rows (src, dst)) on the
    device.  Returns the processed lists (backward edges appended to tied types / added as fresh types in forward-
    type order, self loops inserted at ``self_loop_edge_type``, negative values counting from the end) and the
    float32 [num_total_edge_types, num_nodes] in-degree counts."""
    lib = _lib.load()
    tied = set(tied_fwd_bkwd_edge_types)
    fwd = []
    for a in adjacency_lists:
        if a.dtype != torch.int32 or not a.is_cuda:
            raise ValueError("adjacency lists must be int32 device tensors of shape [E, 2]")
        fwd.append(a.reshape(-1, 2).contiguous())
    dev = fwd[0].device if fwd else torch.device("cuda", torch.cuda.current_device())
    n_fwd = len(fwd)
    out: List[torch.Tensor] = []
    fresh: List[torch.Tensor] = []
    for t, a in enumerate(fwd):
        n = a.shape[0]
        if t in tied:
            both = torch.empty((2 * n, 2), dtype=torch.int32, device=dev)
            _lib.check(lib.tfgnn_adjacency_append(ops._ptr(a), n, 0, ops._ptr(both), ops._stream()))
            _lib.check(lib.tfgnn_adjacency_append(ops._ptr(a), n, 1, ops._ptr(both[n:]), ops._stream()))
            out.append(both)
        else:
            out.append(a)
            flipped = torch.empty((n, 2), dtype=torch.int32, device=dev)
            _lib.check(lib.tfgnn_adjacency_append(ops._ptr(a), n, 1, ops._ptr(flipped), ops._stream()))
            fresh.append(flipped)
    out.extend(fresh)
    if add_self_loop_edges:
        num_edge_types = len(out)
        lb, ub = -(num_edge_types + 1), num_edge_types
        assert lb <= self_loop_edge_type <= ub, "Self loop edge type {} should be in range [{}, {}].".format(
            self_loop_edge_type, lb, ub
        )
        if self_loop_edge_type < 0:
            self_loop_edge_type += num_edge_types + 1
        loops = torch.empty((num_nodes, 2), dtype=torch.int32, device=dev)
        _lib.check(lib.tfgnn_adjacency_self_loops(num_nodes, ops._ptr(loops), ops._stream()))
        out.insert(self_loop_edge_type, loops)
    counts = torch.empty((len(out), num_nodes), dtype=torch.float32, device=dev)
    for t, a in enumerate(out):
        _lib.check(lib.tfgnn_adjacency_in_degrees(ops._ptr(a), a.shape[0], num_nodes, ops._ptr(counts[t]), ops._stream()))
    return out, counts
