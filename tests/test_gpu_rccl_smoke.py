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
