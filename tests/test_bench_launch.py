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


def test_final_line_is_short_and_carries_the_contract():
    """the driver parses the LAST stdout line out of an 8 KB tail.  A complete record of the default run
    (the committed round-5 one: five workloads, step breakdowns, prose - 21 KB) must come out of bench.compact_line as one
    JSON object under 4 KB that still holds the contract's keys and the roofline / cpu_baseline objects."""
    import bench

    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 8192  # the record that did not fit the driver's tail
    full["config"]["gemm_mode_name"] = "f16x2"
    full["config"]["products_per_step"] = {"sp_nt": 11.0, "sp_tn": 6.0, "x3": 0.0, "x3_stream": 0.0, "fp32": 0.0, "gather_sp": 8.0, "gather": 0.0}
    full["config"]["guard"] = {"tripped": False, "stage": "none", "checked_passes_left": 0}
    line = bench.compact_line(full, "gpurun_out/bench_detail_rmat30k.json")
    text = json.dumps(line)
    assert len(text) < bench.FINAL_LINE_LIMIT and "\n" not in text
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert back["metric"] == full["metric"] and abs(back["value"] - full["value"]) <= 1e-4 * full["value"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["config"]["workload"].startswith("rmat30k") and back["config"]["products_per_step"]["sp_nt"] == 11.0
    assert set(back["other_configs"]) == set(full["other_configs"])  # one short row per BASELINE workload
    # a record ten times as wordy still fits: optional parts go first
    full["data"] = full["data"] * 40
    full["cpu_baseline"]["sample"] = full["cpu_baseline"]["sample"] * 10
    assert len(json.dumps(bench.compact_line(full))) < bench.FINAL_LINE_LIMIT


def test_plumbing_run_prints_detail_before_the_final_line():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-only", "--gpus", "2", "--workload", "qm9-tiny"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{") and len(lines[-1]) < 4096
    assert any(l.startswith("BENCH_DETAIL {") for l in lines[:-1])
    last = json.loads(lines[-1])
    assert last["n_gpus"] == 2 and last["config"]["rccl"]["backend"] == "gloo" and last["config"]["rccl"]["world_size_reported"] == 2
