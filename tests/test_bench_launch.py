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
