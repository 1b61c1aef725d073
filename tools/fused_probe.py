"""Stage-1 probe for the gather-fused RGCN product: the rate of the PRODUCER side alone
(tools/fused_probe.hip), and variant (ii) transform-then-gather built from existing kernels, against the shipped
gather + product pair.  Run on the GPU box: python tools/fused_probe.py"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel  # noqa: E402
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch  # noqa: E402

dev = torch.device("cuda", 0)
V, E, L, H = 30000, 900000, 4, 320
feats, adjs = make_synthetic_batch(V, E, L, H, seed=1)
X = torch.from_numpy(feats).to(dev)
adj_dev = tuple(torch.from_numpy(a).to(dev) for a in adjs)
g = ops.Graph(adj_dev, V)
ops.set_gemm_mode("f16x2")
res = {}

# ---- shipped pair -----------------------------------------------------------------------------------------------
rs = g.array(ops.G_INVDEG_BY_DST)
W = (torch.randn((L * H, H), device=dev) * 0.05)
Wt_sp = ops.sp_split_cols(W)
out = torch.empty((V, H), device=dev)
res["shipped_gather_sp_us"] = 1e3 * time_kernel(lambda: ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, rows_per_operand_row=L))
A_sp = ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs, rows_per_operand_row=L)
res["shipped_nt_us"] = 1e3 * time_kernel(lambda: ops.sp_gemm_nt(A_sp, Wt_sp, act="relu", out=out))
ref_out = out.clone()

# ---- variant (ii): Y = X [W_0 | ... | W_{L-1}] (one NT product, N = L*H), then gather rows (src, l) per target node ---
X_sp = ops.sp_split_rows(X)
Wr = W.view(L, H, H)  # W_l [D, H]; operand rows n = (l, h): [L*H, D] K-contiguous = W_l^T stacked
Wn = Wr.permute(0, 2, 1).contiguous().view(L * H, H)
Wn_sp = ops.sp_split_rows(Wn)
Y = torch.empty((V, L * H), device=dev)
res["ii_nt_us"] = 1e3 * time_kernel(lambda: ops.sp_gemm_nt(X_sp, Wn_sp, out=Y))
ew = g.array(ops.G_INVDEG_EDGE_BY_DST)
coll = g.array(ops.G_COLL_BY_DST)
out2 = torch.empty((V, H), device=dev)
res["ii_gather_us"] = 1e3 * time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Y.view(V * L, H), col=coll, edge_weight=ew, post_act="relu", out=out2))
res["ii_max_abs_diff_vs_shipped"] = float((out2 - ref_out).abs().max())

# ---- producer-side probe -------------------------------------------------------------------------------------------
_so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_probe", "libfusedprobe.so")
if not os.path.exists(_so):  # tools/_probe/ is not tracked: build the probe kernel from tools/fused_probe.hip
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.dirname(_so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           f"-I{root}/tf2_gnn_amd/csrc", f"-I{root}/include", f"{root}/tools/fused_probe.hip", "-o", _so])
lib = ctypes.CDLL(_so)
rowptr = g.array(ops.G_ROWPTR_BY_DST)
col = g.array(ops.G_COL_BY_DST)
lens = (rowptr[1:] - rowptr[:-1]).view(V, L)
side = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=rs)  # stand-in for the long-row side table
for T in (48, 32, 1 << 30):
    short = torch.where(lens > T, torch.zeros_like(lens), lens).sum(1)
    order = torch.argsort(short, descending=True).cpu().numpy()
    tiles = (V + 127) // 128
    tn = np.full((tiles, 128), -1, dtype=np.int32)
    # snake deal: balanced short-edge totals per tile
    idx = np.arange(V)
    rnd, pos = idx // tiles, idx % tiles
    tile_of = np.where(rnd % 2 == 0, pos, tiles - 1 - pos)
    slot = rnd
    tn[tile_of, slot] = order
    tn_dev = torch.from_numpy(tn).to(dev)
    tot = short.cpu().numpy()[np.where(tn >= 0, tn, 0)] * (tn >= 0)
    res[f"T{T}_tile_short_edges_mean_max"] = [float(tot.sum(1).mean()), float(tot.sum(1).max())]
    chk = torch.zeros(tiles, device=dev)
    for mode in (0, 1, 2):
        for unr in (2, 4, 8):
            def run():
                rc = lib.fused_probe_launch(ctypes.c_void_p(tn_dev.data_ptr()), tiles, ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                            ctypes.c_void_p(rs.data_ptr()), ctypes.c_void_p(X.data_ptr()), ctypes.c_int64(H), ctypes.c_void_p(side.data_ptr()), L, H, int(min(T, 1 << 30)),
                                            mode, unr, ctypes.c_void_p(chk.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            res[f"probe_T{T}_mode{mode}_unr{unr}_us"] = 1e3 * time_kernel(run, iters=10, warmup=2)
    # natural node order for comparison (no balancing)
    tn2 = np.full((tiles, 128), -1, dtype=np.int32)
    tn2.reshape(-1)[:V] = np.arange(V)
    tn2_dev = torch.from_numpy(tn2).to(dev)

    def run_nat():
        lib.fused_probe_launch(ctypes.c_void_p(tn2_dev.data_ptr()), tiles, ctypes.c_void_p(rowptr.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                               ctypes.c_void_p(rs.data_ptr()), ctypes.c_void_p(X.data_ptr()), ctypes.c_int64(H), ctypes.c_void_p(side.data_ptr()), L, H, int(min(T, 1 << 30)), 0, 4,
                               ctypes.c_void_p(chk.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    res[f"probe_T{T}_natural_order_unr4_us"] = 1e3 * time_kernel(run_nat, iters=5, warmup=1)
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out/r03d", exist_ok=True)
json.dump(res, open("gpurun_out/r03d/fused_probe.json", "w"), indent=1)
