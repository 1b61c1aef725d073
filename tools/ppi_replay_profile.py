"""BASELINE configs[0] (the PPI stand-in of bench.py --workload ppi): the training step captured into one hipGraph and
replayed N times - run under `rocprofv3 --kernel-trace` to see where a replay's device time goes:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ppi_prof -- python tools/ppi_replay_profile.py 20
    python tools/ppi_replay_profile.py --timeline gpurun_out/ppi_prof/.../*kernel_trace.csv"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(n):
    import torch

    import bench
    from tf2_gnn_amd import CapturedStep, ops
    from tf2_gnn_amd.data import make_ppi_shaped_batch, process_adjacency_lists
    from tf2_gnn_amd.layers.message_passing import set_seed
    from tf2_gnn_amd.tasks import NodeMulticlassTask

    wl = bench.WORKLOADS["ppi"]
    dev = torch.device("cuda", 0)
    feats, fwd, n2g, labels = make_ppi_shaped_batch(wl["num_graphs"], wl["nodes_per_graph"], wl["avg_in_degree"], wl["feature_dim"],
                                                    wl["num_labels"], seed=1)
    V = feats.shape[0]
    X = torch.from_numpy(feats).to(dev)
    params = NodeMulticlassTask.get_default_hyperparameters("rgcn")
    params.update({f"gnn_{k}": v for k, v in bench.model_params("rgcn", wl["hidden_dim"], wl["num_layers"]).items()})
    set_seed(0)
    model = NodeMulticlassTask(params, num_edge_types=3, num_node_target_labels=wl["num_labels"])
    ops.set_gemm_mode("f16x2")
    adjs, _ = process_adjacency_lists([torch.from_numpy(fwd).to(dev)], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
    batch = {"node_features": X, "node_to_graph_map": torch.from_numpy(n2g).to(dev), "num_graphs_in_batch": wl["num_graphs"],
             **{f"adjacency_list_{i}": a for i, a in enumerate(adjs)}}
    lab = {"node_labels": torch.from_numpy(labels).to(dev)}

    def step():
        out = model(batch, training=True)
        m = model.compute_task_metrics(batch, out, lab)
        return m, [g for _, g in model.backward()]

    cap = CapturedStep(step)
    cap.capture()
    torch.cuda.synchronize()
    for _ in range(n):
        cap.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        cap.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"replay: {a.elapsed_time(b) / n:.4f} ms per step")


def timeline(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    marks = [i for i, r in enumerate(rows) if "dropout_epoch_kernel" in r["Kernel_Name"]]  # the first node of every replay
    a, b = marks[-3], marks[-2]
    t0 = rows[a]["s"]
    print(f"one replay: {1e-3 * (rows[b]['s'] - t0):.1f} us, {b - a} kernels")
    agg = defaultdict(lambda: [0, 0.0])
    prev = None
    busy = 0.0
    for r in rows[a:b]:
        dur = 1e-3 * (r["e"] - r["s"])
        gap = 1e-3 * (r["s"] - prev) if prev is not None else 0.0
        prev = r["e"]
        busy += dur
        name = r["Kernel_Name"].replace("tfgnn::", "").replace("void ", "")
        print(f"+{1e-3 * (r['s'] - t0):8.1f} dur {dur:6.1f} gap {gap:5.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} {name[:100]}")
        k = name.split("(")[0]
        agg[k][0] += 1
        agg[k][1] += dur
    print(f"kernel time {busy:.1f} us")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:8.1f} us  {c:3d} x  {k[:110]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--timeline":
        timeline(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 20)
