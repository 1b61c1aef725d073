"""How are the per-row magnitudes of the weight-gradient products' operands distributed in the ppi workload?
     python tools/tn_spread_probe.py      (on the GPU box)
Wraps ops.sp_gemm_tn, runs one step of bench.run_ppi and prints per call and scale block: rows with an all-zero operand row
(marker scale 2^-126), quantiles of log2(row scale product / largest), rows more than 2^14 / 2^20 below the largest.
This is how the round-4 guard bug was found (NOTEBOOK.md 4.8): every product showed ~25 % all-zero rows (empty buckets) and no
real row below 2^-16, yet the guard tripped - it judged zero rows by their factor, which is not tiny when the operands are."""
import sys, types
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from tf2_gnn_amd import ops

orig = ops.sp_gemm_tn
calls = []
def spy(a, b, **kw):
    ai = a.inv_scale.reshape(a.rows, -1).double()
    bi = b.inv_scale.reshape(b.rows, -1).double()[:, :1]
    ai = torch.where(ai <= 2.4e-38, torch.zeros_like(ai), ai)
    bi = torch.where(bi <= 2.4e-38, torch.zeros_like(bi), bi)
    mass = (ai * bi)
    for blk in range(mass.shape[1]):
        m = mass[:, blk]
        nz = m[m > 0]
        ref = float(nz.max()) if nz.numel() else 1.0
        lg = torch.log2(nz / ref)
        small = lg < -20
        calls.append(dict(K=a.rows, blk=blk, zero_rows=int((m == 0).sum()), q=[float(x) for x in torch.quantile(lg, torch.tensor([0.0, 0.001, 0.01, 0.1, 0.5], dtype=torch.float64, device=lg.device))],
                          n_small=int(small.sum()), small_mass_rel=float((nz[small] / ref).sum()), lt14=int((lg < -14).sum()),
                          amax_ratio=float(ai[:, blk].max() / ai[:, blk][ai[:, blk] > 0].min()), bmax_ratio=float(bi.max() / bi[bi > 0].min())))
    return orig(a, b, **kw)
ops.sp_gemm_tn = spy
import tf2_gnn_amd.layers.message_passing.gnn_edge_mlp as gem
args = types.SimpleNamespace(warmup=1, steps=1, gemm_mode="f16x2")
import tf2_gnn_amd.layers.gnn as gnnmod
try:
    bench.run_ppi(args, dict(bench.WORKLOADS["ppi"]) if hasattr(bench, "WORKLOADS") else None)
except Exception as e:
    print("run_ppi raised", repr(e)[:300])
for i, c in enumerate(calls):
    print(i, {k: (v if not isinstance(v, float) else float('%.3g' % v)) for k, v in c.items() if k != 'q'}, [round(x) for x in c['q']])
print("calls", len(calls), "mode now", ops.get_gemm_mode())
