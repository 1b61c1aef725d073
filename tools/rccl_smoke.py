"""RCCL itself has never executed in this project (no multi-GPU lease in any round): every collective helper of
tf2_gnn_amd.parallel once through the REAL backend ("nccl" = RCCL on ROCm) with the one rank a single-GPU box allows -
communicator creation bound to the device, float64 MAX all-reduce of a host scalar (the step time), all-gather of per-rank
scalars, the bucketed weighted all-reduce of gradients, barrier, teardown.  Catches what the gloo tests cannot: dtypes / ops the
backend refuses, device placement of the collective tensors, the environment RCCL needs on this driver."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import parallel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
import socket  # noqa: E402

with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as _s:  # a free port: the suite may run this beside other rendezvous
    _s.bind(("127.0.0.1", 0))
    os.environ.setdefault("MASTER_PORT", str(_s.getsockname()[1]))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
print("backend", dist.get_backend(), "world", dist.get_world_size(), "rccl", ".".join(map(str, torch.cuda.nccl.version())),
      "device", torch.cuda.get_device_name(dev))
parallel.barrier(dist)
assert parallel.reduce_max(1.25, dist, dev) == 1.25
g = parallel.all_gather_scalars([3.0, 4.0, 5.0], dist, dev)
assert g.shape == (1, 3) and g[0].tolist() == [3.0, 4.0, 5.0], g


class Var:
    def __init__(self, shape):
        self.value = torch.zeros(shape, device=dev)
        self.grad = torch.full(shape, 2.0, device=dev)


vs = [Var((320, 320)), Var((4, 320, 320)), Var((121,))]
vs[2].grad = None  # a variable this rank did not touch still takes part
calls = parallel.allreduce_gradients(vs, dist, local_count=7110.0, bucket_bytes=1 << 20)
torch.cuda.synchronize()
assert calls >= 2, calls
assert torch.allclose(vs[0].grad, torch.full((320, 320), 2.0, device=dev)) and float(vs[2].grad.abs().max()) == 0.0
parallel.barrier(dist)
dist.destroy_process_group()
print("rccl smoke ok:", calls, "bucketed all-reduces, weighted mean exact with one rank")
