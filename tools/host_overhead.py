"""How long does the HOST need to enqueue one benchmark step (no device sync inside)?  If this approaches the
device time per step, the step is launch-bound and any host hiccup shows up in the metric."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, ppi_rgcn_params  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402
from tf2_gnn_amd.layers import GNN, GNNInput  # noqa: E402

dev = torch.device("cuda", 0)
wl = WORKLOADS["rmat30k"]
V, E, L, D, H, NL = (wl[k] for k in ("num_nodes", "num_edges", "num_edge_types", "feature_dim", "hidden_dim", "num_layers"))
feats, adjs = make_synthetic_batch(V, E, L, D, seed=1)
X = torch.from_numpy(feats).to(dev)
adj_dev = tuple(torch.from_numpy(a).to(dev) for a in adjs)
n2g = torch.zeros(V, dtype=torch.int32, device=dev)
gnn = GNN(ppi_rgcn_params(H, NL))
dOut = torch.randn((V, H), device=dev)
ops.set_gemm_mode(os.environ.get("MODE", "bf16x3"))


def step():
    g = ops.Graph(adj_dev, V)
    gnn(GNNInput(X, g, n2g, 1), training=True)
    gnn.backward(dOut)
    g.close()


for _ in range(5):
    step()
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1000 * t_host / N:.3f} ms/step ; with device {1000 * t_all / N:.3f} ms/step (bucketing on the compute stream, sync build)")
g = ops.Graph(adj_dev, V)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    gnn(GNNInput(X, g, n2g, 1), training=True)
    gnn.backward(dOut)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"graph hoisted: host enqueue {1000 * t_host / N:.3f} ms/step ; with device {1000 * t_all / N:.3f} ms/step")
