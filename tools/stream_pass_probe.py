"""The streaming passes of configs[3] (V = 1.15 M node rows, H = 128) one by one: us per launch and bytes moved per second.
   python tools/stream_pass_probe.py            (the kernels all sit at 5.0 - 5.3 TB/s: the ceiling of mixed read + write streams on this part)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch

    from bench import time_kernel
    from tf2_gnn_amd import ops

    dev = torch.device("cuda", 0)
    V, H = 1151896, 128
    g = torch.Generator(device="cpu").manual_seed(0)
    gates = torch.rand((V, 3 * H), generator=g).to(dev)
    mh = torch.randn((V, 3 * H), generator=g).to(dev)
    h = torch.randn((V, H), generator=g).to(dev)
    dh = torch.randn((V, H), generator=g).to(dev)
    arr = V * H * 4
    rows = []
    rows.append(("gru_gates_backward_sp", time_kernel(lambda: ops.gru_gates_backward_sp(dh, gates, mh, h)), 13 * arr))
    rows.append(("add_scale [V,H]", time_kernel(lambda: ops.add_scale(h, dh, 1.0)), 3 * arr))
    rows.append(("mul [V,H]", time_kernel(lambda: ops.mul(h, dh)), 3 * arr))
    rows.append(("activation_backward tanh [V,H]", time_kernel(lambda: ops.activation_backward("tanh", dh, h)), 3 * arr))
    rows.append(("sp_split_rows [V,H]", time_kernel(lambda: ops.sp_split_rows(h)), 2 * arr))
    print(f"V={V} H={H} TFGNN_GATES_GROUPS={os.environ.get('TFGNN_GATES_GROUPS', '-')}")
    for name, ms, b in rows:
        print(f"  {name:36s} {1000 * ms:8.1f} us  {b / ms / 1e9:6.2f} TB/s")


if __name__ == "__main__":
    if os.environ.get("PROBE_CHILD"):
        child()
    else:
        for groups in ("-",):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PROBE_CHILD="1", TFGNN_GATES_GROUPS=groups), check=False)
