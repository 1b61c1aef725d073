"""Where does the HOST time of an eager PPI step go?  cProfile over the enqueue of 40 steps of `bench.py --workload ppi`'s eager
step (finalisation + bucketing + forward + loss + backward), device idle in between: cumulative time per function, top 45."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, model_params  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_ppi_shaped_batch, process_adjacency_lists  # noqa: E402
from tf2_gnn_amd.layers.message_passing import set_seed  # noqa: E402
from tf2_gnn_amd.tasks import NodeMulticlassTask  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["ppi"]
feats, fwd, n2g, labels = make_ppi_shaped_batch(wl["num_graphs"], wl["nodes_per_graph"], wl["avg_in_degree"], wl["feature_dim"], wl["num_labels"], seed=1)
V = feats.shape[0]
X, fwd_dev, n2g_dev, labels_dev = (torch.from_numpy(a).to(dev) for a in (feats, fwd, n2g, labels))
params = NodeMulticlassTask.get_default_hyperparameters("rgcn")
params.update({f"gnn_{k}": v for k, v in model_params("rgcn", wl["hidden_dim"], wl["num_layers"]).items()})
set_seed(0)
model = NodeMulticlassTask(params, num_edge_types=3, num_node_target_labels=wl["num_labels"])
ops.set_gemm_mode("f16x2")


def step():
    ops.clear_weight_operand_cache()
    adjs, _ = process_adjacency_lists([fwd_dev], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
    batch = {"node_features": X, "node_to_graph_map": n2g_dev, "num_graphs_in_batch": wl["num_graphs"],
             **{f"adjacency_list_{i}": a for i, a in enumerate(adjs)}}
    out = model(batch, training=True)
    model.compute_task_metrics(batch, out, {"node_labels": labels_dev})
    model.backward()


for _ in range(6):
    step()
torch.cuda.synchronize()
N = 40
host = []
for _ in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
host.sort()
print(f"host enqueue per step: median {1e3 * host[N // 2]:.3f} ms, min {1e3 * host[0]:.3f} ms")
pr = cProfile.Profile()
for _ in range(N):
    torch.cuda.synchronize()
    pr.enable()
    step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative")
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((ct / N * 1e3, tt / N * 1e3, nc / N, f"{os.path.basename(fn)}:{line}:{name}"))
rows.sort(reverse=True)
print(f"{'cum ms/step':>12} {'own ms/step':>12} {'calls/step':>10}  function   (under cProfile: ~2x slower than the un-profiled step)")
for ct, tt, nc, name in rows[:45]:
    print(f"{ct:12.4f} {tt:12.4f} {nc:10.1f}  {name}")
