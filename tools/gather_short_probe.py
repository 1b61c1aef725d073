"""Gathers over short rows (configs[3]: a bucket of a molecule batch holds 0 - 4 edges): kernel time by lane-group shape.
   python tools/gather_short_probe.py            runs itself once per TFGNN_GATHER_MULTI / TFGNN_GATHER_GRID setting
(the knobs are read once per process), prints us per launch and the bytes moved."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch

    from bench import time_kernel
    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_qm9_shaped_batch

    dev = torch.device("cuda", 0)
    G = int(os.environ.get("PROBE_GRAPHS", "128000"))
    H = 128
    feats, adjs, n2g, _ = make_qm9_shaped_batch(G, seed=1, feature_dim=H)
    V = feats.shape[0]
    g = ops.Graph(tuple(torch.from_numpy(a).to(dev) for a in adjs), V)
    E, L = g.num_edges, g.num_edge_types
    X = torch.randn((V, H), device=dev)
    M = torch.randn((E, H), device=dev)
    ident = torch.arange(E + 1, dtype=torch.int32, device=dev)
    out = torch.empty((V, H), device=dev)
    res = {}
    res["node view, per-edge rows [E,128] -> [V,128]"] = (time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, M, col=ident, out=out)), E * H * 4 + V * H * 4)
    Y = torch.randn((V * L, H), device=dev)  # (the node views read rows (node, type) of the per-type products)
    res["node view, typed rows [V*L,128] -> [V,128]"] = (time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Y, out=out)), E * H * 4 + V * H * 4)
    res["typed by-source view, SP16 out [V,L*128]"] = (time_kernel(lambda: ops.graph_gather_sp(g, ops.VIEW_BY_SRC_TYPED, X, rows_per_operand_row=L)), E * H * 4 + V * L * H * 4)
    A = torch.empty((V * L, H), device=dev)
    res["typed by-target view, fp32 out [V*L,128]"] = (time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, out=A)), E * H * 4 + V * L * H * 4)
    print(f"V={V} E={E} L={L}  multi={os.environ.get('TFGNN_GATHER_MULTI', '0')} grid={os.environ.get('TFGNN_GATHER_GRID', '0')}")
    for k, (ms, b) in res.items():
        print(f"  {k:48s} {1000 * ms:8.1f} us  {b / ms / 1e9:7.2f} TB/s")


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
    else:
        for multi, grid in ((1, 0), (21, 0), (0, 0)):
            env = dict(os.environ, PROBE_CHILD="1", TFGNN_GATHER_MULTI=str(multi), TFGNN_GATHER_GRID=str(grid))
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
