"""The collective helpers of tf2_gnn_amd.parallel through the REAL backend ("nccl" = RCCL on ROCm), with the one rank a
single-GPU box allows (tools/rccl_smoke.py; own process: a process group per interpreter).  Multi-rank behaviour is covered
by the gloo tests; this one covers what gloo cannot - that RCCL loads on this driver, takes the dtypes / ops / device tensors the
helpers hand it and tears down cleanly."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parallel_helpers_run_on_rccl_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=240)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "backend nccl world 1" in out.stdout and "rccl smoke ok" in out.stdout, out.stdout


def test_bench_flow_on_rccl_with_one_rank():
    """bench.py's N > 1 flow - rank-consistent settle loop, barrier, MAX of the step time, all-gather of the per-rank counts, the
    per-step gradient all-reduce - with a communicator of ONE rank on the real backend (TFGNN_FORCE_PROCESS_GROUP=1): the line then
    says which backend and RCCL version carried it."""
    import json

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TFGNN_FORCE_PROCESS_GROUP="1", TFGNN_BENCH_WATCHDOG="150")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TFGNN_BENCH_SINGLE_DEVICE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-alt-mode", "--no-roofline", "--allreduce-grads"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    rccl = line["config"]["rccl"]
    assert rccl["backend"] == "nccl" and rccl["world_size_reported"] == 1 and rccl["rccl_version"][0].isdigit(), rccl
    assert "all-reduce" in line["config"]["collectives_per_step"] and len(line["config"]["allreduce_ms_per_step_per_rank"]) == 1
