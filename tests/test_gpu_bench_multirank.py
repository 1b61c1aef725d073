"""bench.py's N > 1 path THROUGH THE HIP KERNELS on a box with one GPU: two ranks share cuda:0 (TFGNN_BENCH_SINGLE_DEVICE=1,
gloo collectives).  Covers what the 8-GPU driver run exercises - self-spawn, graph sharding of the molecule batch
(parallel.shard_batch -> per-rank Graph / GNN / pooling step), replicas for the single-graph workloads, the weighted
gradient all-reduce, the metric collectives - short of RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    # (TFGNN_BENCH_WATCHDOG: every process of a run that is still alive after 100 s dumps its Python stacks to stderr - a hang
    #  then fails with the place it hangs at instead of a bare timeout)
    env = dict(os.environ, TFGNN_BENCH_SINGLE_DEVICE="1", TFGNN_BENCH_WATCHDOG="100")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                              "--no-alt-mode", *extra], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode("utf-8", "replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        raise AssertionError("bench.py " + " ".join(extra) + " hung; stacks after 100 s:\n" + err[-6000:])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    # the driver's contract for N > 1 too: the LAST stdout line is the short object, the complete record precedes it
    last = [l for l in out.stdout.splitlines() if l.strip()][-1]
    assert last == lines[0] and len(last) < 4096 and any(l.startswith("BENCH_DETAIL {") for l in out.stdout.splitlines())
    return json.loads(lines[0])


def test_two_ranks_shard_the_molecule_batch_through_the_hip_path():
    one = _bench(["--workload", "qm9-tiny", "--no-roofline"])
    two = _bench(["--gpus", "2", "--workload", "qm9-tiny", "--no-roofline", "--allreduce-grads"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert sum(two["config"]["edges_per_rank"]) == one["config"]["edges_per_rank"][0]  # the same batch, split by graph
    assert two["value"] > 0 and one["value"] > 0
    assert "all-reduce" in two["config"]["collectives_per_step"]
    # what the communicator is made of travels in the line (the first real RCCL run will say "nccl" and a version here)
    assert two["config"]["rccl"]["backend"] == "gloo" and two["config"]["rccl"]["world_size_reported"] == 2
    assert len(two["config"]["ms_per_step_per_rank"]) == 2 and len(two["config"]["allreduce_ms_per_step_per_rank"]) == 2
    assert two["config"]["products_per_step"]["gather"] + two["config"]["products_per_step"]["gather_sp"] > 0


def test_two_replica_ranks_for_the_single_graph_workload():
    r = _bench(["--gpus", "2", "--workload", "tiny"])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["config"]["edges_per_rank"] == [40000.0, 40000.0]
    assert r["roofline"]["bound"] in ("hbm", "mfma") and 0 < r["roofline"]["frac"] <= 1.0
