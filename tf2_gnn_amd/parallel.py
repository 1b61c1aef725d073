"""Multi-GPU support for the hot path: one process per GPU, graph batches sharded by graph.

A tf2-gnn batch is a disjoint union of graphs with per-graph node-id offsets and no cross-graph
edges (tf2_gnn/data/graph_dataset.py:202-222), so the message-passing path shards by graph with NO
data-path collective: every rank runs the full layer stack on its own graphs.  The only collectives
are control-plane ones (barrier, MAX of the step time, all-gather of per-rank metric terms), issued
through ``torch.distributed`` - backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).

The reference itself has no distributed code; this module is new.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world_size, dist-or-None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # TFGNN_FORCE_PROCESS_GROUP=1: a communicator of ONE rank - every collective of the N > 1 path then runs through the real
    # backend on a single-GPU box (tests/test_gpu_rccl_smoke.py: bench.py's whole flow on RCCL)
    if world <= 1 and os.environ.get("TFGNN_FORCE_PROCESS_GROUP") != "1":
        return rank, 1, None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("WORLD_SIZE", str(max(world, 1)))
        os.environ.setdefault("RANK", str(rank))
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device  # binds the communicator to this rank's GPU at once (no lazy device guess)
        try:
            dist.init_process_group(backend=backend, **kwargs)
        except (TypeError, ValueError, RuntimeError):
            if not kwargs or dist.is_initialized():
                raise
            dist.init_process_group(backend=backend)  # a torch / RCCL build without eager init: lazy binding
    return rank, max(world, 1), dist


def partition_graphs(cost_per_graph: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time assignment of graphs to ranks by cost (edge count): graphs sorted by
    decreasing cost (ties by index), each given to the currently lightest rank (ties by rank).
    Deterministic; every rank computes the same partition.  Graph order inside a rank is ascending,
    so concatenating rank outputs by ``graph ids`` restores the batch order."""
    order = sorted(range(len(cost_per_graph)), key=lambda g: (-int(cost_per_graph[g]), g))
    loads = [0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for g in order:
        r = min(range(world_size), key=lambda i: (loads[i], i))
        parts[r].append(g)
        loads[r] += int(cost_per_graph[g])
    return [sorted(p) for p in parts]


def shard_batch(
    node_features: np.ndarray,
    adjacency_lists: Sequence[np.ndarray],
    node_to_graph_map: np.ndarray,
    num_graphs: int,
    world_size: int,
    rank: int,
) -> Tuple[np.ndarray, List[np.ndarray], np.ndarray, int, np.ndarray, np.ndarray]:
    """Host-side (numpy) sharding of one batch by graph.

    Returns (local node_features, local adjacency_lists with re-based node ids, local
    node_to_graph_map (0..G_local-1), G_local, global ids of the local graphs, global ids of the
    local nodes).  The node_to_graph_map must be sorted (graph_dataset.py:211-217)."""
    n2g = np.asarray(node_to_graph_map)
    V = n2g.shape[0]
    assert np.all(n2g[1:] >= n2g[:-1]), "node_to_graph_map must be sorted"
    starts = np.searchsorted(n2g, np.arange(num_graphs), side="left")
    ends = np.searchsorted(n2g, np.arange(num_graphs), side="right")
    edges_per_graph = np.zeros(num_graphs, dtype=np.int64)
    for adj in adjacency_lists:
        if adj.shape[0]:
            np.add.at(edges_per_graph, n2g[adj[:, 1]], 1)
    # cost = edges (gather traffic) + nodes (dense work): both scale the per-rank step time
    cost = edges_per_graph + (ends - starts)
    mine = partition_graphs(cost.tolist(), world_size)[rank]
    mine_arr = np.asarray(mine, dtype=np.int64)
    node_ids = (
        np.concatenate([np.arange(starts[g], ends[g]) for g in mine]) if mine else np.zeros(0, dtype=np.int64)
    )
    new_id = np.full(V, -1, dtype=np.int64)
    new_id[node_ids] = np.arange(node_ids.shape[0])
    local_adj = []
    for adj in adjacency_lists:
        if adj.shape[0]:
            keep = new_id[adj[:, 1]] >= 0  # edges never cross graphs: the target decides
            a = adj[keep]
            src = new_id[a[:, 0]]
            assert np.all(src >= 0), "edge crosses graph boundaries"
            local_adj.append(np.stack([src, new_id[a[:, 1]]], axis=1).astype(np.int32))
        else:
            local_adj.append(np.zeros((0, 2), dtype=np.int32))
    local_graph_of = np.full(num_graphs, -1, dtype=np.int64)
    local_graph_of[mine_arr] = np.arange(len(mine))
    local_n2g = local_graph_of[n2g[node_ids]].astype(np.int32)
    return (
        np.ascontiguousarray(node_features[node_ids]),
        local_adj,
        local_n2g,
        len(mine),
        mine_arr,
        node_ids,
    )


def _collective_device(dist, device):
    """device tensors for RCCL, host tensors for gloo (the CPU tests and the single-GPU multi-rank smoke test)"""
    if dist is not None and dist.get_backend() == "gloo":
        return "cpu"
    return device or "cpu"


def barrier(dist) -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def reduce_max(value: float, dist, device=None) -> float:
    """MAX over ranks of a host scalar (step time)."""
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_collective_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_scalars(values: Sequence[float], dist, device=None) -> np.ndarray:
    """all-gather of a few per-rank scalars (edge counts, loss sums, ...) -> [world, len(values)].
    This is the 'final metric reduction' of the north star: the only data any rank sends."""
    mine = torch.tensor(list(values), dtype=torch.float64, device=_collective_device(dist, device))
    if dist is None:
        return mine.cpu().numpy()[None, :]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu().numpy()


def allreduce_gradients(variables, dist, average: bool = True, bucket_bytes: int = 64 << 20,
                        local_count: Optional[float] = None):
    """The one exchange step of a data-parallel *training* step: sum (or mean) over ranks of
    d loss / d variable, in place on ``variable.grad``.

    ``local_count`` (nodes for a node-level loss, graphs for a graph-level one): each rank's loss is a MEAN over its own
    shard and the shards are balanced by cost, not by count, so the plain mean of the ranks' gradients is a mean of means.
    With ``local_count`` the gradients are weighted n_local / n_global (one extra scalar rides in the first bucket) and
    the result equals the single-device gradient of the whole batch.  The reference has no distributed code; this is what a
    weight update over graph shards needs and nothing more - the forward / backward passes stay collective-free.

    Gradients are packed into flat fp32 buckets and each bucket is one all-reduce: over xGMI a ring all-reduce is
    per-link bound (7 links x ~153 GB/s per GPU, no switch), so a few large messages beat one per variable - the
    whole RGCN H=320 L=4 stack (6.6 MB) is a single bucket.  ``variables``: objects with ``.grad`` (None = zeros of
    ``.value``'s shape: a variable one rank did not touch still takes part, every rank must issue the same
    collectives).  Returns the number of all-reduce calls issued."""
    if dist is None:
        return 0
    world = dist.get_world_size()
    variables = list(variables)
    for v in variables:
        if v.grad is None:
            v.grad = torch.zeros_like(v.value)
    weight = None
    if local_count is not None:  # weighted mean: sum_r n_r g_r / sum_r n_r
        dev0 = variables[0].grad.device if variables else "cpu"
        weight = torch.tensor([float(local_count)], dtype=torch.float32, device=dev0)
        for v in variables:
            v.grad = v.grad * float(local_count)
    calls, start = 0, 0
    while start < len(variables):
        end, nbytes = start, 0
        while end < len(variables) and (end == start or nbytes + variables[end].grad.numel() * 4 <= bucket_bytes):
            nbytes += variables[end].grad.numel() * 4
            end += 1
        group = variables[start:end]
        pieces = [v.grad.reshape(-1) for v in group]
        if weight is not None and start == 0:
            pieces.append(weight.to(pieces[0].device))
        flat = torch.cat(pieces)
        if dist.get_backend() == "gloo" and flat.is_cuda:  # single-GPU multi-rank smoke test: host round trip
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat = host.to(flat.device)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if weight is not None and start == 0:
            total = flat[-1].clone()
            flat = flat[:-1]
        if weight is not None:
            flat = flat / total
        elif average:
            flat /= world
        off = 0
        for v in group:
            n = v.grad.numel()
            v.grad = flat[off : off + n].view_as(v.grad)
            off += n
        calls += 1
        start = end
    return calls
