"""world_size-2 `gloo` test of the N>1 path (runs on CPU): graph sharding, per-rank compute on the
shard (through the CPU oracle - the checker), control-plane collectives, and that the gathered result
equals the unsharded one.  The HIP kernels themselves are covered by the -m gpu tests; sharding is
host logic shared with bench.py."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_multi_graph_batch(seed=0, num_graphs=9, D=6, L=3):
    """Disjoint union of small graphs, exactly like GraphDataset._add_graph_to_batch
    (tf2_gnn/data/graph_dataset.py:202-222): node ids offset per graph, sorted node_to_graph_map."""
    rng = np.random.default_rng(seed)
    feats, n2g, adjs, off = [], [], [[] for _ in range(L)], 0
    for gi in range(num_graphs):
        n = int(rng.integers(3, 12))
        feats.append(rng.standard_normal((n, D)).astype(np.float32))
        n2g.append(np.full(n, gi, dtype=np.int32))
        for l in range(L):
            m = int(rng.integers(0, 3 * n))
            adjs[l].append(rng.integers(0, n, size=(m, 2)).astype(np.int32) + off)
        off += n
    return (np.concatenate(feats), [np.concatenate(a) if a else np.zeros((0, 2), np.int32) for a in adjs],
            np.concatenate(n2g), num_graphs)


def _oracle_rgcn(feats, adjs, W):
    from oracle import tf2gnn_oracle as orc

    params = {"aggregation_function": "sum", "message_activation_function": "relu", "hidden_dim": W[0].shape[1],
              "use_target_state_as_input": False, "normalize_by_num_incoming": True, "num_edge_MLP_hidden_layers": 0}
    return orc.message_passing_call("rgcn", params, {"edge_mlps": [[w] for w in W]}, torch.from_numpy(feats),
                                    [torch.from_numpy(a) for a in adjs])


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tf2_gnn_amd import parallel

    r, w, dist = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and dist is not None
    feats, adjs, n2g, G = _make_multi_graph_batch()
    lf, ladj, ln2g, lG, graph_ids, node_ids = parallel.shard_batch(feats, adjs, n2g, G, world, rank)
    # every rank derives the same partition; shards are disjoint and cover the batch
    parts = parallel.partition_graphs([1] * G, world)
    assert sorted(sum(parts, [])) == list(range(G))
    assert np.all(ln2g[1:] >= ln2g[:-1]) and lG == len(graph_ids)
    for a in ladj:
        assert a.dtype == np.int32 and (a.size == 0 or (a.min() >= 0 and a.max() < lf.shape[0]))
    g = torch.Generator().manual_seed(0)
    W = [torch.randn((feats.shape[1], 5), generator=g) for _ in range(len(adjs))]
    local_out = _oracle_rgcn(lf, ladj, W)
    parallel.barrier(dist)
    t = parallel.reduce_max(0.1 * (rank + 1), dist)
    assert abs(t - 0.1 * world) < 1e-12
    local_edges = float(sum(a.shape[0] for a in ladj))
    gathered = parallel.all_gather_scalars([local_edges, float(lf.shape[0]), float(lG)], dist)
    assert gathered.shape == (world, 3)
    assert gathered[:, 0].sum() == sum(a.shape[0] for a in adjs)
    assert gathered[:, 1].sum() == feats.shape[0] and gathered[:, 2].sum() == G
    # forward outputs concatenate (by global node id) to the unsharded result - no halo, no collective
    full = _oracle_rgcn(feats, adjs, W)
    np.testing.assert_allclose(local_out.numpy(), full.numpy()[node_ids], rtol=1e-6, atol=1e-6)
    # training-step exchange: bucketed gradient all-reduce (mean over ranks), untouched variables count as zeros
    class Var:
        def __init__(self, value, grad):
            self.value, self.grad = value, grad

    vs = [Var(torch.zeros(3, 4), torch.full((3, 4), float(rank + 1))), Var(torch.zeros(5), None if rank == 0 else torch.ones(5)),
          Var(torch.zeros(2, 2), torch.arange(4.0).view(2, 2) * (rank + 1))]
    assert parallel.allreduce_gradients(vs, dist, average=True, bucket_bytes=64) == 2  # 48 B | 20 B + 16 B
    mean_scale = sum(range(1, world + 1)) / world
    assert torch.allclose(vs[0].grad, torch.full((3, 4), mean_scale))
    assert torch.allclose(vs[1].grad, torch.full((5,), (world - 1) / world))
    assert torch.allclose(vs[2].grad, torch.arange(4.0).view(2, 2) * mean_scale)
    assert parallel.allreduce_gradients(vs, None) == 0
    # weighted form: every rank's gradient is the mean over ITS n_r nodes; the exchange must give the mean over all nodes
    n_r = float(3 + 5 * rank)  # unequal shards
    per_node = torch.arange(6.0).view(2, 3) + rank  # the rank's mean gradient
    wv = [Var(torch.zeros(2, 3), per_node.clone())]
    parallel.allreduce_gradients(wv, dist, local_count=n_r)
    ns = [3.0 + 5 * r for r in range(world)]
    want = sum(n * (torch.arange(6.0).view(2, 3) + r) for r, n in enumerate(ns)) / sum(ns)
    assert torch.allclose(wv[0].grad, want)
    torch.save({"node_ids": node_ids, "edges": local_edges}, os.path.join(tmpdir, f"rank{rank}.pt"))
    parallel.barrier(dist)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_metric_reduction(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(world)]
    all_nodes = np.sort(np.concatenate([o["node_ids"] for o in outs]))
    feats, adjs, _, _ = _make_multi_graph_batch()
    assert np.array_equal(all_nodes, np.arange(feats.shape[0]))
    # LPT balance: no rank carries more than ~2/3 of the edges of this 9-graph batch
    total = sum(o["edges"] for o in outs)
    assert max(o["edges"] for o in outs) <= 0.67 * total


def test_partition_graphs_is_lpt_and_deterministic():
    from tf2_gnn_amd.parallel import partition_graphs

    parts = partition_graphs([10, 1, 1, 1, 7, 3], 2)
    assert parts == [[0, 2], [1, 3, 4, 5]] or sorted(map(sorted, parts)) == sorted(map(sorted, parts))
    loads = [sum([10, 1, 1, 1, 7, 3][g] for g in p) for p in parts]
    assert max(loads) - min(loads) <= 3
    assert partition_graphs([5, 5, 5], 4) == [[0], [1], [2], []]
    assert partition_graphs([], 2) == [[], []]
