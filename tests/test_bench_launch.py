"""bench.py's N > 1 launch path on CPU: ``python bench.py --gpus 2`` must become two ranks (--gpus was parsed and
ignored).  --plumbing-only does the spawn, the gloo rendezvous on 127.0.0.1, the graph sharding of the workload and the
metric collectives without device work (the compute path has no CPU fallback)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-only", *extra], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_shards_the_molecule_batch():
    r = _run(["--gpus", "2", "--workload", "qm9-tiny"])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong"
    assert sum(r["graphs_per_rank"]) == 2000  # one batch, split by graph
    e = r["edges_per_rank"]
    assert abs(e[0] - e[1]) <= 0.02 * max(e)  # LPT balance by edges + nodes


def test_gpus_2_replicas_for_single_graph_workloads():
    r = _run(["--gpus", "2", "--workload", "tiny"])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["nodes_per_rank"] == [2000.0, 2000.0]


def test_single_rank_needs_no_launcher():
    r = _run(["--workload", "tiny"])
    assert r["n_gpus"] == 1


def test_cpu_baseline_runs_for_every_workload_model():
    """bench.py's cpu_baseline leg (the oracle's training step of the workload's own model on a bounded sample) on tiny
    shapes of every model the default run reports: no device involved."""
    import bench

    for model in ("rgcn", "rgat", "ggnn", "gnn_edge_mlp", "rgin"):
        if model in ("ggnn", "gnn_edge_mlp"):
            wl = dict(model=model, num_graphs=60, feature_dim=16, hidden_dim=16, num_layers=2, sharded=True)
        else:
            wl = dict(model=model, num_nodes=200, num_edges=1500, num_edge_types=3, feature_dim=16, hidden_dim=16, num_layers=2,
                      num_heads=4)
        batch = bench.build_batch(wl, 0, 1)
        params = bench.model_params(model, 16, 2, wl.get("num_heads"))
        r = bench.cpu_baseline(wl, batch, params, budget_seconds=1.0)
        assert r["value"] > 0 and r["kind"] == "port" and r["cores"] >= 1 and model.upper() in r["sample"], (model, r)
