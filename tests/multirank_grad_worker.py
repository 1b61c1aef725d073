"""Worker of tests/test_gpu_multirank_gradients.py (not a test module): N ranks, ALL on cuda:0 with gloo collectives
(a one-GPU box), run a data-parallel training step of the molecule workload through the HIP path:

    shard_batch (by graph) -> per-rank Graph / GNN (GGNN) / WeightedSum pooling forward + backward
    -> allreduce_gradients(local_count = graphs of the rank)

and rank 0 compares every weight gradient with the gradient of the SAME step over the WHOLE batch on one rank.  The loss is
a mean over graphs, loss = 1/G sum_g <pooled_g, c_g>, so each rank's d loss / d pooled is c_g / G_local and the weighted
all-reduce must give exactly the single-device gradient (tf2_gnn/models/graph_task_model.py:347-357 for the step,
data/graph_dataset.py:202-222 for why a batch shards by graph)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_model(H, L_layers, model):
    from bench import model_params
    from tf2_gnn_amd.layers import GNN, WeightedSumGraphRepresentation
    from tf2_gnn_amd.layers.message_passing import set_seed

    params = model_params(model, H, L_layers)
    params["layer_input_dropout_rate"] = 0.0  # the masks of a sharded batch cannot equal those of the whole one
    set_seed(0)
    gnn = GNN(params)
    pool = WeightedSumGraphRepresentation(graph_representation_size=32, num_heads=4, weighting_fun="softmax",
                                          scoring_mlp_layers=[H], transformation_mlp_layers=[H])
    return gnn, pool


def step(gnn, pool, feats, adjs, n2g, G, c, dev):
    """one forward + backward; d loss / d pooled = c / G (loss = mean over the G graphs of <pooled_g, c_g>)"""
    from tf2_gnn_amd.layers import GNNInput, NodesToGraphRepresentationInput

    X = torch.from_numpy(np.ascontiguousarray(feats)).to(dev)
    adj_dev = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in adjs)
    n2g_dev = torch.from_numpy(np.ascontiguousarray(n2g, dtype=np.int32)).to(dev)
    # training mode for the GNN (rate 0: no masks); the pooling MLPs apply dropout when training (rate 0.2), and the masks
    # of a shard cannot equal those of the whole batch -> eval mode there
    out = gnn(GNNInput(X, adj_dev, n2g_dev, G), training=True)
    pooled = pool(NodesToGraphRepresentationInput(out, n2g_dev, G), training=False)
    d_pooled = (torch.from_numpy(c).to(dev) / float(G)).contiguous()
    gnn.backward(pool.backward(d_pooled))
    loss = float((pooled.double() * torch.from_numpy(c).to(dev).double()).sum() / G)
    return loss, list(gnn.trainable_variables) + list(pool.trainable_variables)


def main():
    out_path, model, gemm_mode = sys.argv[1], sys.argv[2], sys.argv[3]
    from tf2_gnn_amd import ops, parallel
    from tf2_gnn_amd.data import make_qm9_shaped_batch

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rank, world, dist = parallel.init_distributed(device=dev, backend="gloo")
    ops.set_gemm_mode(gemm_mode)
    G, H, layers = 3000, 128, 3
    feats, adjs, n2g, _ = make_qm9_shaped_batch(G, seed=5, feature_dim=H)
    c = np.random.default_rng(9).standard_normal((G, 32)).astype(np.float32)

    lf, ladj, ln2g, Gl, graph_ids, _ = parallel.shard_batch(feats, adjs, n2g, G, world, rank)
    gnn, pool = build_model(H, layers, model)
    loss_local, variables = step(gnn, pool, lf, ladj, ln2g, Gl, c[graph_ids], dev)
    calls = parallel.allreduce_gradients(variables, dist, local_count=float(Gl))
    sharded = [v.grad.detach().cpu().double() for v in variables]
    names = [v.name for v in variables]
    losses = parallel.all_gather_scalars([loss_local * Gl, float(Gl)], dist, dev)
    loss_global = float(losses[:, 0].sum() / losses[:, 1].sum())

    result = None
    if rank == 0:
        gnn1, pool1 = build_model(H, layers, model)  # same seed -> same weights
        loss_one, vars1 = step(gnn1, pool1, feats, adjs, n2g, G, c, dev)
        worst = {}
        for n, a, v in zip(names, sharded, vars1):
            b = v.grad.detach().cpu().double()
            scale = max(1.0, float(b.abs().max()))
            worst[n] = float((a - b).abs().max()) / scale
        result = {"world": world, "model": model, "gemm_mode": gemm_mode, "allreduce_calls": calls,
                  "graphs_per_rank": losses[:, 1].tolist(), "loss_sharded": loss_global, "loss_one_rank": loss_one,
                  "max_scaled_gradient_difference": max(worst.values()), "per_variable": worst, "num_variables": len(names)}
        with open(out_path, "w") as f:
            json.dump(result, f)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
