#!/usr/bin/env python3
"""Benchmark of the message-passing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload NAME]

Headline (default workload ``rmat30k``): BASELINE.json's metric - edges/sec (fwd+bwd), RGCN H=320 L=4 on an R-MAT batch
shaped like configs[1] (30k nodes, 900k edges, 4 edge types) with the PPI_RGCN.json hyper-parameters.  The other
BASELINE configs are workloads of the same script (``rgat``, ``qm9-ggnn``, ``qm9-edgemlp``, ``arxiv-rgin``), each with its
own roofline block.

One "step" = one pass of the hot path over one batch: bucket the batch's edges (ops.Graph), GNN forward in training mode
(dropout) and the full backward (all weight gradients) - plus node->graph pooling forward/backward for the QM9-shaped
workloads.  Inputs are resident in HBM before the timed region.

Multi-GPU (one process per GPU, torch.distributed backend nccl = RCCL over xGMI).  With N > 1 and no torchrun environment
the script re-executes itself under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` (rendezvous on
127.0.0.1), so ``python bench.py --gpus 8`` IS an 8-rank run; the world size of the communicator is asserted against
--gpus and reported as ``n_gpus``.
  * single-graph workloads (rmat30k, rgat, arxiv-rgin): one independently seeded batch of the same shape per rank -
    replicas, weak scaling, no data-path collective;
  * QM9-shaped workloads: ONE batch of 128k graphs (the same on every rank, seeded), sharded by graph
    (parallel.shard_batch: LPT by edges + nodes, node ids re-based); every rank runs its shard through the HIP path -
    strong scaling, no data-path collective;
  * the only collectives: the barrier, the MAX-reduce of the step time, one all-gather of per-rank scalars (edges,
    nodes) for the metric; --allreduce-grads adds the training-step exchange (one bucketed all-reduce of the weight
    gradients, parallel.allreduce_gradients).

Prints ONE JSON line (rank 0).  ``roofline`` describes the dominant kernel of the workload, timed live with HIP events on
the launch stream; ``cpu_baseline`` times the CPU oracle (the reference's op sequence, torch-CPU, the workload's own model)
on rank 0 at N == 1 on a bounded sample; the non-headline workloads also carry ``step_breakdown`` - every library op of the
real step between HIP events, priced against its roof; the default run (rmat30k, N == 1) times the other four BASELINE
workloads briefly in processes of their own and reports them under ``other_configs`` (--no-other-configs skips that).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
MFMA_FP32_PEAK_TFLOPS = 157.3
MFMA_16BIT_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak (MI355X_MICROARCH.md; no sparsity)
GEMM_MODE_NOTES = {
    "fp32": "fp32: v_mfma_f32_32x32x2_f32 on the fp32 operands",
    "bf16x3": "bf16x3: fp32 operands split exactly into 3 bf16 pieces, 6 largest piece products (each exact in fp32), fp32 "
    "accumulate on v_mfma_f32_32x32x16_bf16; fp32 in/out",
    "bf16x3_9": "bf16x3_9: as bf16x3 with all 9 piece products (products exact, only the fp32 accumulation rounds)",
    "f16x2": "f16x2: the layers' products on pre-split operands - every fp32 value scaled by a power of two per row and split "
    "into two fp16 pieces by round-to-nearest (22+ significand bits), 3 piece products (each exact in fp32), fp32 accumulate "
    "on v_mfma_f32_32x32x16_f16, the gather writes the split operand; fp32 in/out; error vs fp64 measured at or below the "
    "fp32-MFMA kernel's (profiles/parity_r06.json - every parity check of the suite in this mode, tools/mfma_acc_probe.hip); "
    "products without a split producer run as bf16x3; the library default since round 3",
}

# BASELINE.json configs -> workloads.  ``sharded``: one global batch split by graph over the ranks (strong scaling);
# otherwise every rank owns a replica batch (weak scaling).
WORKLOADS = {
    # configs[1] shape + the metric's model (RGCN H=320, 4 layers, PPI_RGCN.json hypers)
    "rmat30k": dict(model="rgcn", num_nodes=30000, num_edges=900000, num_edge_types=4, feature_dim=320, hidden_dim=320, num_layers=4),
    # configs[0] stand-in: tf2_gnn_train RGCN PPI - three 2370-node graphs, batch finalisation
    # (self loops + backward edges), RGCN H=320 L=4, NodeMulticlassTask head with its loss; own driver: run_ppi()
    "ppi": dict(model="rgcn", num_graphs=3, nodes_per_graph=2370, avg_in_degree=14, feature_dim=50, hidden_dim=320, num_layers=4, num_labels=121),
    "tiny": dict(model="rgcn", num_nodes=2000, num_edges=40000, num_edge_types=4, feature_dim=64, hidden_dim=64, num_layers=4),
    # configs[2]: RGAT, 8 heads, H=256, 8 layers on the same graph
    "rgat": dict(model="rgat", num_nodes=30000, num_edges=900000, num_edge_types=4, feature_dim=256, hidden_dim=256, num_layers=8, num_heads=8),
    # configs[3]: GGNN / GNN_Edge_MLP on a QM9-shaped batch (128k molecules, H=128) + node->graph pooling, graph-sharded
    "qm9-ggnn": dict(model="ggnn", num_graphs=128000, feature_dim=128, hidden_dim=128, num_layers=8, sharded=True),
    "qm9-edgemlp": dict(model="gnn_edge_mlp", num_graphs=128000, feature_dim=128, hidden_dim=128, num_layers=4, sharded=True),
    "qm9-tiny": dict(model="ggnn", num_graphs=2000, feature_dim=64, hidden_dim=64, num_layers=2, sharded=True),
    # configs[4]: RGIN, 40 edge types (Zipf), H=512 on an ogbn-arxiv-scale graph
    "arxiv-rgin": dict(model="rgin", num_nodes=170000, num_edges=1200000, num_edge_types=40, feature_dim=512, hidden_dim=512, num_layers=4),
}


def model_params(model, hidden_dim, num_layers, num_heads=None):
    """GNN hyper-parameters of a workload: the message passing class' defaults, the stack settings of
    tf2_gnn/cli_utils/default_hypers/PPI_RGCN.json:6-19 (no residual / LayerNorm / global exchange, one Dense after
    layer 0, input dropout 0.1) and the BASELINE config's sizes."""
    from tf2_gnn_amd.layers import GNN

    p = GNN.get_default_hyperparameters(model)
    p.update(
        {
            "num_layers": num_layers,
            "hidden_dim": hidden_dim,
            "layer_input_dropout_rate": 0.1,
            "dense_every_num_layers": 10000,
            "residual_every_num_layers": 10000,
            "global_exchange_every_num_layers": 10000,
            "use_inter_layer_layernorm": False,
            "initial_node_representation_activation": "tanh",
            "dense_intermediate_layer_activation": "tanh",
        }
    )
    if model == "rgcn":
        p.update({"use_target_state_as_input": False, "normalize_by_num_incoming": True, "num_edge_MLP_hidden_layers": 0,
                  "message_activation_function": "ReLU", "aggregation_function": "sum"})
    if model == "rgat":
        p.update({"num_heads": num_heads or 8})
    return p


def ppi_rgcn_params(hidden_dim, num_layers):
    return model_params("rgcn", hidden_dim, num_layers)


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "edges/sec (fwd+bwd) RGCN H=320 L=4, PPI-shaped batch, 1/2/4/8 MI355X"


def time_kernel(fn, iters=20, warmup=3):
    """average launch duration in ms, HIP events on the current (= launch) stream"""
    for _ in range(warmup):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's op sequence (oracle), SURVEY.md 8d protocol (warm-up, median) on a bounded sample
# --------------------------------------------------------------------------------------------------------------------
def _random_oracle_weights(model, params, D, H, L, NL, gen):
    """Glorot-uniform weights in the oracle's layout for every model bench.py times (shapes as the HIP layers build them)."""
    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return ((torch.rand((i, o), generator=gen) * 2 - 1) * lim).requires_grad_(True)

    leaves = []

    def t(x):
        leaves.append(x)
        return x

    mp = []
    for _ in range(NL):
        if model == "rgat":
            K = params["num_heads"]
            mp.append({"kernels": [t(glorot(H, H)) for _ in range(L)], "attn": [t(glorot(K, 2 * H // K)) for _ in range(L)]})
            continue
        din = 2 * H if params.get("use_target_state_as_input") else H
        nh = int(params.get("num_edge_MLP_hidden_layers", 0))
        sizes = [din] + [H] * nh + [H]
        w = {"edge_mlps": [[t(glorot(sizes[k], sizes[k + 1])) for k in range(len(sizes) - 1)] for _ in range(L)], "aggr_mlp": None}
        if model == "ggnn":
            w["gru_kernel"], w["gru_recurrent_kernel"] = t(glorot(H, 3 * H)), t(glorot(H, 3 * H))
            w["gru_bias"] = t(torch.zeros((2, 3 * H), requires_grad=True))
        mp.append(w)
    weights = {"initial_projection": t(glorot(D, H)), "mp": mp, "dense": {0: t(glorot(H, H))}, "layernorm": []}
    return weights, leaves


def cpu_baseline(wl, batch, params, budget_seconds=30.0):
    """The oracle's training step of the workload's model - the reference's literal op sequence: materialised per-edge
    gathers, per-edge matmuls, concat over edge types, scatter-add (torch-CPU fp32), autograd for the backward, plus the
    WeightedSum pooling head for the molecule workloads - timed on the host: fastest thread count of a short sweep, one
    warm-up, the median of up to 5 runs inside ``budget_seconds``, on a BOUNDED SAMPLE of the same batch (the whole batch
    and the whole layer stack when that fits the budget; otherwise whole graphs of a molecule batch / a prefix of every
    edge list of a single-graph batch, and fewer layers - scaled linearly, the per-edge cost dominates)."""
    from oracle import tf2gnn_oracle as orc

    feats, adjs, n2g, G = batch["feats"], batch["adjs"], batch["n2g"], int(batch["num_graphs"])
    model = wl["model"]
    cores_all = os.cpu_count() or 1
    H, NL, L = wl["hidden_dim"], wl["num_layers"], len(adjs)
    gen = torch.Generator().manual_seed(0)
    weights, leaves = _random_oracle_weights(model, params, feats.shape[1], H, L, NL, gen)
    pooled = bool(wl.get("sharded"))
    if pooled:
        def glorot(i, o):
            lim = (6.0 / (i + o)) ** 0.5
            return ((torch.rand((i, o), generator=gen) * 2 - 1) * lim).requires_grad_(True)
        pool_w = {"scoring": ([glorot(H, H), glorot(H, 4)], None), "transformation": ([glorot(H, H), glorot(H, 32)], None)}
        pool_cfg = {"graph_representation_size": 32, "num_heads": 4, "weighting_fun": "softmax",
                    "scoring_mlp_activation_fun": "ReLU", "transformation_mlp_activation_fun": "ReLU"}
        leaves = leaves + pool_w["scoring"][0] + pool_w["transformation"][0]
    E = sum(a.shape[0] for a in adjs)

    def sample(frac):
        """-> (X, adjacency lists, node_to_graph_map, graphs): whole graphs of a molecule batch, else an edge prefix"""
        if pooled:
            g_n = max(1, int(G * frac))
            v_n = int(np.searchsorted(n2g, g_n, side="left"))
            sub = [torch.from_numpy(a[a[:, 1] < v_n]) for a in adjs]  # edges never cross graphs
            return torch.from_numpy(feats[:v_n]), sub, torch.from_numpy(n2g[:v_n]), g_n
        return torch.from_numpy(feats), [torch.from_numpy(a[: max(1, int(a.shape[0] * frac))]) for a in adjs], None, 1

    def run(frac, layers):
        X, adj_t, ids, g_n = sample(frac)
        p = dict(params, num_layers=layers)
        t0 = time.perf_counter()
        out, _ = orc.gnn_internal_call(p, weights, X, adj_t)
        if pooled:
            out = orc.weighted_sum_graph_representation(pool_cfg, pool_w, out, ids, g_n)
        torch.autograd.grad(out.sum(), leaves, allow_unused=True)
        return time.perf_counter() - t0

    # thread count: torch-CPU with every hardware thread of a 256-thread host is several times SLOWER on these per-edge ops
    # than with a few dozen (measured: 11.6 s vs < 1 s for the same 1/16 sample) - take the fastest of a short sweep
    torch.set_num_threads(min(16, cores_all))
    cal = 1.0 / 64
    run(cal, 1)  # first-touch / thread-pool warm-up
    sweep = {}
    candidates = (8, 16, 32, 64, 128, cores_all) if budget_seconds >= 20 else (16, 32)  # (a short budget: a short sweep)
    for nt in sorted({n for n in candidates if n <= cores_all} or {cores_all}):
        torch.set_num_threads(nt)
        sweep[nt] = min(run(cal, 1), run(cal, 1))
        if sweep[nt] > 2.0 * min(sweep.values()):
            break
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    # one layer (+ projection / Dense / pooling) on the whole batch, extrapolated from HALF the batch (the 1/64 runs of the
    # sweep are dominated by fixed costs: they over-estimate a full layer several times)
    # - or from the fraction that takes about a sixth of the budget at that pessimistic estimate: half of an arxiv-sized RGIN
    # layer with 40 edge types is minutes of host time)
    est_full = sweep[threads] / cal
    f_cal = float(min(0.5, max(2 * cal, (budget_seconds / 6.0) / max(est_full, 1e-6))))
    t_layer_full = run(f_cal, 1) / f_cal if est_full > 1.0 else est_full
    # the sample: whole stack on the whole batch if a warm-up and two timed runs fit the budget, else fewer layers, then a
    # fraction of the batch
    per_run = budget_seconds / 4.5  # a warm-up and at least THREE timed runs ("median of 1 runs")
    layers = NL if t_layer_full * NL <= per_run else max(1, min(NL, int(per_run / max(t_layer_full, 1e-6))))
    frac = float(min(1.0, max(cal, per_run / max(t_layer_full * layers, 1e-6))))
    t_used = run(frac, layers)  # warm-up at the sample size
    times = []
    while len(times) < 5 and (len(times) < 3 or t_used + times[-1] < budget_seconds):
        times.append(run(frac, layers))
        t_used += times[-1]
    t = float(np.median(times))
    t_full = t / frac * NL / layers
    what = ("whole graphs" if pooled else "a prefix of every edge list")
    return {
        "value": E / t_full,
        "unit": "edges/s",
        "cores": threads,
        "host_cpu_count": cores_all,  # os.cpu_count() of the box the line was measured on (north_star: "core count stated")
        "kind": "port",
        "thread_sweep_seconds": {str(k): v for k, v in sweep.items()},
        "sample": f"{layers} of {NL} {model.upper()} layers (+ projection / Dense{' / WeightedSum pooling' if pooled else ''}) fwd+bwd on "
        f"{frac:.3f} of the batch ({what}{', all nodes' if not pooled else ''}), {threads} threads (fastest of a sweep up to the host's {cores_all}): "
        f"1 warm-up, median of {len(times)} runs = {t:.2f} s, scaled to the full batch and stack; torch-CPU fp32 restatement of the "
        "reference op sequence (TensorFlow unavailable offline)",
        "seconds_per_step": t_full,
    }


# --------------------------------------------------------------------------------------------------------------------
# output: the full record on an EARLIER line (and in gpurun_out/), the driver's line LAST and short
# --------------------------------------------------------------------------------------------------------------------
FINAL_LINE_LIMIT = 3800  # bytes (< 4 KB with headroom): the driver keeps an 8 KB tail of stdout and parses its last line
DETAIL_PREFIX = "BENCH_DETAIL "
GEMM_MODE_SHORT = {
    "fp32": "fp32 (fp32 MFMA)",
    "bf16x3": "bf16x3 (fp32 in/out, exact 3-way bf16 split, 6 piece products, fp32 accumulate)",
    "bf16x3_9": "bf16x3_9 (fp32 in/out, exact 3-way bf16 split, 9 piece products)",
    "f16x2": "f16x2 (fp32 in/out/accumulate; operands = 2 fp16 pieces per value under a power-of-two block scale, 3 piece products)",
}
_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_boundary", "compulsory_frac", "ms_per_launch",
              "launches_per_step", "share_of_step", "achieved_basis", "mfma_busy_counter")
_CPU_KEYS = ("value", "unit", "cores", "host_cpu_count", "kind", "seconds_per_step", "sample")


def _round(x, digits=5):
    """floats to ``digits`` significant digits (the short line only; the detail record keeps full precision)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, digits) for v in x]
    return x


def _short_roofline(b, kernel_chars=90):
    if not b:
        return None
    out = {k: b[k] for k in _ROOF_KEYS if b.get(k) is not None}
    if "kernel" in out and len(out["kernel"]) > kernel_chars:
        out["kernel"] = out["kernel"][:kernel_chars].rstrip() + "..."
    if "achieved_basis" in out and len(out["achieved_basis"]) > 48:
        out["achieved_basis"] = out["achieved_basis"][:48].rstrip() + "..."
    if isinstance(out.get("mfma_busy_counter"), dict):
        out["mfma_busy_counter"] = {k: out["mfma_busy_counter"].get(k) for k in ("value", "file")}
    return out


def compact_line(result, detail_path=None):
    """The driver's line: the contract's keys, ``roofline`` / ``cpu_baseline`` objects and a one-row summary of every other
    BASELINE workload - everything else (step breakdowns, secondary rooflines, prose) stays in the detail record.
    Always shorter than FINAL_LINE_LIMIT: optional parts are dropped, in a fixed order, until it is."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "graphs_per_s", "host_ms_per_step", "plumbing_only", "edges_per_rank", "nodes_per_rank", "graphs_per_rank",
            "max_time", "batches")
    line = {k: result[k] for k in keep if k in result}
    cfg = result.get("config", {})
    c = {}
    for k in ("workload", "gemm_mode", "per_layer_traversal_rate_edges_per_s", "products_per_step", "guard", "collectives_per_step",
              "edges_per_rank", "ms_per_step_per_rank", "allreduce_ms_per_step_per_rank", "rccl", "loss"):
        if cfg.get(k) is not None:
            c[k] = cfg[k]
    if cfg.get("gemm_mode_name") in GEMM_MODE_SHORT:
        c["gemm_mode"] = GEMM_MODE_SHORT[cfg["gemm_mode_name"]]
    if isinstance(c.get("guard"), dict):
        c["guard"] = {k: c["guard"].get(k) for k in ("tripped", "stage")}
    if isinstance(c.get("workload"), str) and len(c["workload"]) > 230:
        c["workload"] = c["workload"][:230].rstrip() + "..."
    alt = cfg.get("alt_gemm_mode")
    if alt:
        c["alt_gemm_mode"] = {"gemm_mode": alt.get("gemm_mode_name", alt.get("gemm_mode", ""))[:24], "ms_per_step": alt["ms_per_step"],
                              "value": alt["value"]}
    line["config"] = c
    if result.get("roofline"):
        line["roofline"] = _short_roofline(result["roofline"])
    if result.get("roofline_secondary"):
        line["roofline_secondary"] = _short_roofline(result["roofline_secondary"], 60)
    if result.get("cpu_baseline"):
        cb = {k: result["cpu_baseline"][k] for k in _CPU_KEYS if k in result["cpu_baseline"]}
        if len(cb.get("sample", "")) > 160:
            cb["sample"] = cb["sample"][:160].rstrip() + "..."
        line["cpu_baseline"] = cb
    for k in ("eager", "replay_only", "replay_static_batch"):
        if isinstance(result.get(k), dict):
            line[k] = {kk: vv for kk, vv in result[k].items() if kk in ("ms_per_step", "host_ms_per_step", "edges_per_s", "graphs_per_s", "batches")}
    if result.get("other_configs"):
        oc = {}
        for name, e in result["other_configs"].items():
            if "error" in e:
                oc[name] = {"error": str(e["error"])[-80:]}
                continue
            row = {"ms_per_step": e.get("ms_per_step"), "value": e.get("value")}
            if e.get("roofline"):
                r = e["roofline"]
                row["roofline"] = {"bound": r.get("bound"), "frac": r.get("frac")}
            if e.get("cpu_baseline"):
                row["cpu_value"] = e["cpu_baseline"].get("value")
            if e.get("products_per_step"):
                row["products_per_step"] = {k: v for k, v in e["products_per_step"].items() if v and k in ("sp_nt", "sp_tn", "stream_f16x2", "x3", "fp32")}
            if e.get("guard"):
                row["guard_stage"] = e["guard"].get("stage")
            if e.get("host_ms_per_step") is not None:
                row["host_ms_per_step"] = e["host_ms_per_step"]
            if isinstance(e.get("replay_static_batch"), dict):
                row["replay_static_batch_ms"] = e["replay_static_batch"].get("ms_per_step")
            oc[name] = row
        line["other_configs"] = oc
    if detail_path:
        line["detail"] = detail_path
    line = _round(line)
    # shrink, in this order, until the line fits
    for drop in (("roofline_secondary",), ("config", "alt_gemm_mode"), ("cpu_baseline", "sample"), ("other_configs",), ("data",),
                 ("config", "ms_per_step_per_rank"), ("config", "allreduce_ms_per_step_per_rank"), ("config", "edges_per_rank"), ("edges_per_rank",),
                 ("nodes_per_rank",), ("graphs_per_rank",), ("roofline", "achieved_basis"), ("roofline", "kernel")):
        if len(json.dumps(line)) < FINAL_LINE_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    return line


def emit(result, workload, write_file=True):
    """Rank 0's output: the complete record as ``BENCH_DETAIL {...}`` on an earlier line and, where the directory can be
    written, in gpurun_out/bench_detail_<workload>.json; then ONE short JSON line - the last line of stdout."""
    detail_path = None
    try:
        if not write_file:
            raise OSError("no file asked for")
        root = os.path.dirname(os.path.abspath(__file__))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        rel = os.path.join("gpurun_out", f"bench_detail_{workload}.json")
        with open(os.path.join(root, rel), "w") as f:
            json.dump(result, f, indent=1)
        detail_path = rel
    except OSError:
        pass
    print(DETAIL_PREFIX + json.dumps(result), flush=True)
    text = json.dumps(compact_line(result, detail_path))
    assert len(text) < FINAL_LINE_LIMIT and "\n" not in text
    print(text, flush=True)


# --------------------------------------------------------------------------------------------------------------------
# workload construction / launch
# --------------------------------------------------------------------------------------------------------------------
def products_from_counts(c0, c1, steps):
    """tfgnn_launch_counts differences -> launches per step by product family.  ``x3`` = the exact bf16x3 kernels (gemm_x3s /
    x3p / x3k; ``x3_stream`` is the x3k share), ``stream_f16x2`` = the streaming kernel in the f16x2 arithmetic (3 products, operands
    split on the fly), ``fp32`` = gemm_mfma_kernel, ``sp_nt`` / ``sp_tn`` = the split-operand products."""
    d = {k: (c1[k] - c0[k]) / float(steps) for k in c1}
    return {"sp_nt": d["sp_nt"], "sp_tn": d["sp_tn"], "x3": d["gemm_bf16x3"], "stream_f16x2": d["stream_f16x2"],
            "x3_stream": d["gemm_stream"] - d["stream_f16x2"], "fp32": d["gemm_fp32"], "gather_sp": d["gather_sp"], "gather": d["gather"]}


def rccl_info(dist, dev, world):
    """What the communicator of an N > 1 run is made of: backend, the world size IT reports, the RCCL
    version torch was built against, the device name.  None at N == 1."""
    if dist is None:
        return None
    info = {"backend": dist.get_backend(), "world_size_reported": dist.get_world_size(),
            "device": torch.cuda.get_device_name(dev) if dev is not None else "cpu"}
    try:
        info["rccl_version"] = ".".join(map(str, torch.cuda.nccl.version()))
    except Exception as e:  # a torch build without the binding
        info["rccl_version"] = f"unavailable ({type(e).__name__})"
    return info


def build_batch(wl, rank, world, variant=0):
    """-> dict(feats, adjs, n2g, num_graphs).  variant > 0: another draw of the same shape (--distinct-batches)"""
    from tf2_gnn_amd import parallel
    from tf2_gnn_amd.data import make_qm9_shaped_batch, make_synthetic_batch, make_zipf_typed_batch

    if wl.get("sharded"):
        feats, adjs, n2g, _ = make_qm9_shaped_batch(wl["num_graphs"], seed=1 + 1000 * variant, feature_dim=wl["feature_dim"])  # same on every rank
        G = wl["num_graphs"]
        if world > 1:
            feats, adjs, n2g, G, _, _ = parallel.shard_batch(feats, adjs, n2g, G, world, rank)
        return dict(feats=feats, adjs=adjs, n2g=np.ascontiguousarray(n2g, dtype=np.int32), num_graphs=G)
    if wl["model"] == "rgin":
        feats, adjs = make_zipf_typed_batch(wl["num_nodes"], wl["num_edges"], wl["num_edge_types"], wl["feature_dim"], seed=1 + rank + 1000 * variant)
    else:  # timing seeds 1..
        feats, adjs = make_synthetic_batch(wl["num_nodes"], wl["num_edges"], wl["num_edge_types"], wl["feature_dim"], seed=1 + rank + 1000 * variant)
    return dict(feats=feats, adjs=adjs, n2g=np.zeros(feats.shape[0], dtype=np.int32), num_graphs=1)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_torchrun(args):
    """``python bench.py --gpus N`` without a torchrun environment: become N ranks."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def run_ppi(args, wl):
    """configs[0] (the reference's own CPU-runnable case, README.md:47-48: 2.63 / 4.01 graphs/s on real PPI with TensorFlow):
    a full training step of tf2_gnn_train RGCN PPI minus the optimizer update on the synthetic stand-in - batch finalisation
    on the device (process_adjacency_lists: self loops + backward edges -> 3 edge types, data/utils.py:9-58), edge bucketing,
    NodeMulticlassTask forward (RGCN H=320 L=4 + Dense(121)), sigmoid cross-entropy + micro-F1, full backward.  At V = 7110
    the step is ~200 launches of a few microseconds: the line separates the HOST time of a step (Python + ctypes, measured as
    the time to enqueue it) from the device-bound step time."""
    import torch

    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_ppi_shaped_batch, process_adjacency_lists
    from tf2_gnn_amd.layers.message_passing import set_seed
    from tf2_gnn_amd.tasks import NodeMulticlassTask

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # --distinct-batches K (default here: 8): K different draws of the stand-in take turns - every eager step finalises, buckets
    # and trains on a batch it has not seen in the previous K - 1 steps, as a training epoch over PPI does
    K_batches = max(1, int(args.distinct_batches)) if args.distinct_batches != 1 else 8
    batches = []
    for k in range(K_batches):
        feats, fwd, n2g, labels = make_ppi_shaped_batch(wl["num_graphs"], wl["nodes_per_graph"], wl["avg_in_degree"], wl["feature_dim"],
                                                        wl["num_labels"], seed=1 + 1000 * k)
        batches.append(tuple(torch.from_numpy(a).to(dev) for a in (feats, fwd, n2g, labels)))
    X, fwd_dev, n2g_dev, labels_dev = batches[0]
    V = X.shape[0]
    params = NodeMulticlassTask.get_default_hyperparameters("rgcn")
    params.update({f"gnn_{k}": v for k, v in model_params("rgcn", wl["hidden_dim"], wl["num_layers"]).items()})
    set_seed(0)
    model = NodeMulticlassTask(params, num_edge_types=3, num_node_target_labels=wl["num_labels"])
    ops.set_gemm_mode(args.gemm_mode)
    E = [0]
    calls = [0]

    # Input pipeline: like the headline workload (and the reference's background batching thread + tf.data prefetch,
    # data/graph_dataset.py:292-295), the finalisation + bucketing of batch i+1 is enqueued on a second stream at the start of
    # step i; --serial-bucketing keeps it on the compute stream.
    side_eager = None if args.serial_bucketing else torch.cuda.Stream()
    ready = []

    def prepare(k):
        Xb, fwd_b, n2g_b, labels_b = batches[k % K_batches]
        adjs, _ = process_adjacency_lists([fwd_b], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
        E[0] = int(sum(a.shape[0] for a in adjs))
        batch = {"node_features": Xb, "node_to_graph_map": n2g_b, "num_graphs_in_batch": wl["num_graphs"],
                 **{f"adjacency_list_{i}": a for i, a in enumerate(adjs)}}
        if side_eager is not None and model.built:
            batch["bucketed_graph"] = ops.Graph(adjs, V, wait=False, parts=model._gnn.graph_parts(V, [int(a.shape[0]) for a in adjs]))
        return batch, labels_b

    def prepare_async(k):
        if side_eager is None or not model.built:
            return prepare(k)
        side_eager.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_eager):
            return prepare(k)

    def step():
        ops.clear_weight_operand_cache()  # an optimizer update invalidates the split forms of the weights
        if not ready:
            ready.append(prepare_async(calls[0]))
        batch, labels_b = ready.pop(0)
        calls[0] += 1
        g_ = batch.get("bucketed_graph")
        if g_ is not None:
            torch.cuda.current_stream().wait_stream(side_eager)  # compute waits for THIS batch's preparation
            g_.wait()
        ready.append(prepare_async(calls[0]))  # the next batch, under this step
        out = model(batch, training=True)
        metrics = model.compute_task_metrics(batch, out, {"node_labels": labels_b})
        model.backward()
        if g_ is not None:
            g_.close()
        return metrics

    for _ in range(max(args.warmup, 3)):
        step()
    torch.cuda.synchronize()
    # device-bound step time: K steps between synchronisations (the host runs ahead where it can)
    c0 = ops.launch_counts()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    products = products_from_counts(c0, ops.launch_counts(), args.steps)
    # host time: how long the enqueue of a step takes (each step timed on an idle queue, then waited for)
    host = []
    for _ in range(min(args.steps, 10)):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
    host_ms = 1000.0 * float(np.median(host))
    # device time of one step in isolation: events around it
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    m = step()
    b.record()
    torch.cuda.synchronize()
    G = wl["num_graphs"]
    eager = {"ms_per_step": 1000.0 * dt, "host_ms_per_step": host_ms, "device_ms_one_step_alone": a.elapsed_time(b),
             "edges_per_s": E[0] / dt, "graphs_per_s": G / dt,
             "what": "every step driven from Python: finalisation + bucketing + forward + loss + backward (~190 launches through ctypes)"}

    for b_, _ in ready:  # the batch prepared for the step after the last one
        g_ = b_.get("bucketed_graph")
        if g_ is not None:
            g_.wait()
            g_.close()
    ready.clear()
    torch.cuda.synchronize()

    # ---- the same training step replayed from ONE hipGraph (tf2_gnn_amd.capture.CapturedStep; the reference traces its step
    # into one tf.function graph: models/graph_task_model.py:327-357).  The forward + loss + backward of the finalised, bucketed
    # batch is captured once; a replay is one hipGraphLaunch.  The input pipeline's share of a step - finalisation and
    # bucketing of the NEXT batch - still runs per step, eagerly on a second stream (like the headline workload), so the
    # step keeps the work it had in the eager measurement; `replay_only` is the step without it.
    from tf2_gnn_amd import CapturedStep

    adjs_static, _ = process_adjacency_lists([fwd_dev], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
    batch_static = {"node_features": X, "node_to_graph_map": n2g_dev, "num_graphs_in_batch": wl["num_graphs"],
                    **{f"adjacency_list_{i}": t for i, t in enumerate(adjs_static)}}
    edges_per_type = [int(t.shape[0]) for t in adjs_static]

    def train_step():
        out = model(batch_static, training=True)
        metrics = model.compute_task_metrics(batch_static, out, {"node_labels": labels_dev})
        return metrics, [g_ for _, g_ in model.backward()]

    cap = CapturedStep(train_step)
    cap.capture()
    side = torch.cuda.Stream()
    pending = []

    def prepare_next():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            adjs, _ = process_adjacency_lists([fwd_dev], V, add_self_loop_edges=True, tied_fwd_bkwd_edge_types=set())
            return ops.Graph(adjs, V, wait=False, parts=model._gnn.graph_parts(V, edges_per_type))

    def captured_step(with_preparation):
        if with_preparation:
            if not pending:
                pending.append(prepare_next())
            g_ = pending.pop(0)
            torch.cuda.current_stream().wait_stream(side)
            g_.wait()  # the batch this step trains on has been finalised and bucketed
            pending.append(prepare_next())
            res = cap.replay()
            g_.close()
            return res
        return cap.replay()

    def time_captured(with_preparation):
        for _ in range(5):
            captured_step(with_preparation)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            captured_step(with_preparation)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / args.steps
        hs = []
        for _ in range(min(args.steps, 10)):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            captured_step(with_preparation)
            hs.append(time.perf_counter() - h0)
        torch.cuda.synchronize()
        return per, 1000.0 * float(np.median(hs))

    dt_replay, host_replay = time_captured(False)
    dt_cap, host_cap = time_captured(True)
    for g_ in pending:
        g_.wait()
        g_.close()
    pending.clear()
    m, _ = cap.replay()
    torch.cuda.synchronize()
    assert not cap.guard_tripped(), "the spread guard tripped during the replayed steps"
    # The line's `value` is the EAGER step: a stream of distinct batches - what PPI training is and what the reference's
    # tf.function step serves (layers/gnn.py:220-232) - cannot be replayed from a graph frozen to one adjacency.
    # The replayed step of a static batch (full-batch training, revisited batches, inference) is reported beside it.
    result = {
        "metric": "graphs/sec and edges/sec (batch finalisation + fwd + loss + bwd) RGCN H=320 L=4 + NodeMulticlassTask, PPI stand-in (BASELINE configs[0])",
        "value": E[0] / dt,
        "unit": "edges/s",
        "graphs_per_s": G / dt,
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt,
        "host_ms_per_step": host_ms,
        "step": f"every step another batch ({K_batches} distinct batches take turns): finalisation + bucketing of the next batch "
                f"{'on the compute stream' if args.serial_bucketing else '(2nd stream, overlapped)'} + forward + loss + backward, driven from Python (eager)",
        "batches": K_batches,
        "eager": eager,
        "replay_static_batch": {"ms_per_step": 1000.0 * dt_cap, "host_ms_per_step": host_cap, "edges_per_s": E[0] / dt_cap,
                                "graphs_per_s": G / dt_cap,
                                "what": "the SAME batch every step: forward + loss + backward replayed from one hipGraph (CapturedStep) + "
                                        "finalisation and bucketing of a next batch on a second stream (built, not trained on)"},
        "replay_only": {"ms_per_step": 1000.0 * dt_replay, "host_ms_per_step": host_replay, "edges_per_s": E[0] / dt_replay,
                        "graphs_per_s": G / dt_replay,
                        "what": "forward + loss + backward of the finalised, bucketed batch: one hipGraphLaunch per step"},
        "bound": "host" if host_ms > 0.9 * 1000.0 * dt else "device",
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "reference_published": {"graphs_per_s_epoch_1": 2.63, "graphs_per_s_later_epochs": 4.01,
                                "source": "tf2-gnn README.md:47-48 (TensorFlow, real PPI, hardware not stated) - not this stand-in, no ratio taken"},
        "dtype": "f32",
        "data": "synthetic: 3 R-MAT graphs x 2370 nodes, 14 forward edges per node, N(0,1) features [V, 50], Bernoulli(0.3) labels [V, 121], Glorot weights",
        "config": {"workload": f"ppi: V={V} E={E[0]} (3 edge types after finalisation) D_in=50 H=320 L=4 labels=121",
                   "gemm_mode": args.gemm_mode, "gemm_mode_name": args.gemm_mode, "products_per_step": products,
                   "guard": model._gnn.guard_state(), "loss": float(m["loss"])},
    }
    emit(result, "ppi")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rmat30k", choices=sorted(WORKLOADS))
    ap.add_argument("--reuse-graph", action="store_true", help="bucket the edges once, outside the timed steps")
    ap.add_argument("--serial-bucketing", action="store_true", help="bucket each batch on the compute stream (no overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=36.0, help="host time budget of the cpu_baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-mode", default="f16x2", choices=["fp32", "bf16x3", "bf16x3_9", "f16x2"],
                    help="how the fp32 products run on the matrix cores (include/tfgnn.h): fp32 MFMA, exact bf16 operand splitting "
                    "with 6 / 9 piece products, or pre-split fp16 operands with 3 piece products (fp32 in, fp32 accumulate, fp32 out)")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the second timing in another GEMM mode")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="N > 1: add the training-step exchange (one bucketed RCCL all-reduce of the weight gradients) to every "
                         "step; off by default - the fwd+bwd metric itself has no collective")
    ap.add_argument("--distinct-batches", type=int, default=1,
                    help="cycle K different batches of the workload's shape through the steps (default 1: the same batch every step, "
                         "whose source rows are then warm in the Infinity Cache - the optimistic case)")
    ap.add_argument("--no-settle", action="store_true", help="skip the untimed settle loop (launch-path tests)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short timings of the other BASELINE workloads that the default run reports under `other_configs`")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no device work: spawn / rendezvous (gloo) / sharding / collectives only - the CPU test of the N > 1 launch path")
    args = ap.parse_args()

    if os.environ.get("TFGNN_BENCH_WATCHDOG"):  # debugging aid: every process dumps its Python stacks if it still runs after N seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["TFGNN_BENCH_WATCHDOG"]), repeat=False, file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from tf2_gnn_amd import parallel

    wl = WORKLOADS[args.workload]
    sharded = bool(wl.get("sharded"))
    if args.workload == "ppi":
        if args.gpus != 1 or args.plumbing_only:
            print("--workload ppi is a single-device workload", file=sys.stderr)
            sys.exit(2)
        if not torch.cuda.is_available():
            print("bench.py needs a ROCm device (there is no CPU fallback)", file=sys.stderr)
            sys.exit(2)
        return run_ppi(args, wl)
    if args.plumbing_only:
        rank, world, dist = parallel.init_distributed(backend="gloo")
        assert world == args.gpus, f"communicator has {world} ranks, --gpus says {args.gpus}"
        batch = build_batch(wl, rank, world)
        parallel.barrier(dist)
        dt = parallel.reduce_max(0.001 * (rank + 1), dist)
        gathered = parallel.all_gather_scalars([float(sum(a.shape[0] for a in batch["adjs"])), float(batch["feats"].shape[0]),
                                                float(batch["num_graphs"])], dist)
        if rank == 0:
            emit({"metric": baseline_metric(), "value": None, "unit": "edges/s", "n_gpus": world, "plumbing_only": True,
                  "edges_per_rank": gathered[:, 0].tolist(), "nodes_per_rank": gathered[:, 1].tolist(),
                  "graphs_per_rank": gathered[:, 2].tolist(), "max_time": dt, "scaling": "strong" if sharded else "weak",
                  "config": {"workload": args.workload, "rccl": rccl_info(dist, None, world)}}, args.workload, write_file=False)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        print("bench.py needs a ROCm device (there is no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # TFGNN_BENCH_SINGLE_DEVICE=1 (tests/test_gpu_bench_multirank.py): all ranks on cuda:0 with gloo collectives - the N > 1
    # code path (sharding, per-rank batches, metric collectives around the HIP step) on a box with one GPU
    single_device = os.environ.get("TFGNN_BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world, dist = parallel.init_distributed(device=dev, backend="gloo" if single_device else None)  # nccl == RCCL over xGMI
    assert world == args.gpus, f"communicator has {world} ranks, --gpus says {args.gpus}"

    from tf2_gnn_amd import ops
    from tf2_gnn_amd.layers import GNN, GNNInput, NodesToGraphRepresentationInput, WeightedSumGraphRepresentation
    from tf2_gnn_amd.layers.message_passing import set_seed

    batch = build_batch(wl, rank, world)
    feats, adjs = batch["feats"], batch["adjs"]
    V, D = feats.shape
    L = len(adjs)
    E = sum(a.shape[0] for a in adjs)
    H, NL = wl["hidden_dim"], wl["num_layers"]
    X = torch.from_numpy(feats).to(dev)
    adj_dev = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in adjs)
    n2g = torch.from_numpy(batch["n2g"]).to(dev)
    G = int(batch["num_graphs"])
    params = model_params(wl["model"], H, NL, wl.get("num_heads"))
    set_seed(0)
    gnn = GNN(params)
    pool = None
    if sharded:  # QM9 head: node -> graph pooling (GD=32, 4 heads, softmax weights)
        pool = WeightedSumGraphRepresentation(graph_representation_size=32, num_heads=4, weighting_fun="softmax",
                                              scoring_mlp_layers=[H], transformation_mlp_layers=[H])
        dOut = torch.randn((G, 32), generator=torch.Generator().manual_seed(0)).to(dev)
    else:
        dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(0)).to(dev)
    edges_per_type = [int(a.shape[0]) for a in adj_dev]
    graph = ops.Graph(adj_dev, V) if args.reuse_graph else None
    # Input pipeline: every step buckets one batch's edges (ops.Graph).  Like the reference, whose batches are prepared by
    # a background thread + tf.data prefetch while the previous step trains (data/graph_dataset.py:292-295,
    # cli_utils/training_utils.py:114-115), the bucketing of batch i+1 is enqueued on a second HIP stream at the start
    # of step i and overlaps with its compute.  --serial-bucketing keeps it on the compute stream instead.
    side = torch.cuda.Stream() if (graph is None and not args.serial_bucketing) else None
    pending = []

    # f16x2: the first product takes the node features as a split operand; preparing it is part of the batch preparation,
    # like the bucketing.  Two copies of the feature tensor alternate so that the preparation of batch i+1 (second stream)
    # never touches what step i reads.
    f16_features = args.gemm_mode == "f16x2" and D % 16 == 0 and D >= 32
    feats_dev = [X, X.clone()] if (f16_features and graph is None and not args.serial_bucketing) else [X]
    adjs_dev = [adj_dev]
    K_batches = max(1, int(args.distinct_batches))
    if K_batches > 1:
        # --distinct-batches K: K draws of the workload's shape (same V, E, L; other seeds) take turns, so a step's source rows,
        # edge lists and saved tensors are NOT the ones the previous step left in the 256 MB Infinity Cache
        if graph is not None or sharded:
            print("--distinct-batches needs per-step bucketing of an unsharded workload", file=sys.stderr)
            sys.exit(2)
        feats_dev, adjs_dev = [X], [adj_dev]
        for k in range(1, K_batches):
            bk = build_batch(wl, rank, world, variant=k)
            assert bk["feats"].shape == feats.shape and sum(a.shape[0] for a in bk["adjs"]) == E
            feats_dev.append(torch.from_numpy(bk["feats"]).to(dev))
            adjs_dev.append(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in bk["adjs"]))
    prepared = [0]

    def prepare_batch():
        """-> (Graph, features): bucket the edges and (f16x2) split the node features for the first product"""
        x = feats_dev[prepared[0] % len(feats_dev)]
        adj = adjs_dev[prepared[0] % len(adjs_dev)]
        ept = edges_per_type if len(adjs_dev) == 1 else [int(a.shape[0]) for a in adj]
        prepared[0] += 1
        if side is None:
            if f16_features:
                ops.split_rows_remembered(x)
            return ops.Graph(adj, V, parts=gnn.graph_parts(V, ept)), x
        # the preparation may start once the steps enqueued so far are done (the step before last used this copy of the
        # features); it then runs under the step that is enqueued next
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if f16_features:
                ops.split_rows_remembered(x)
            return ops.Graph(adj, V, wait=False, parts=gnn.graph_parts(V, ept)), x  # only the tables this stack reads

    allreduce_events = []

    def step():
        # a training step ends with an in-place weight update, which invalidates the split forms of the weights the f16x2
        # products cache per value: drop them here so every timed step splits its weights like a training step does
        ops.clear_weight_operand_cache()
        if graph is not None:
            g, x = graph, X
            if f16_features:
                ops.split_rows_remembered(x)
        elif side is None:
            g, x = prepare_batch()
        else:
            if not pending:
                pending.append(prepare_batch())
            g, x = pending.pop(0)
            torch.cuda.current_stream().wait_stream(side)  # compute waits for THIS batch's preparation
            g.wait()
            pending.append(prepare_batch())  # next batch, overlapped with this step
        out = gnn(GNNInput(x, g, n2g, G), training=True)
        if pool is not None:
            pool(NodesToGraphRepresentationInput(out, n2g, G), training=True)
            gnn.backward(pool.backward(dOut))
        else:
            gnn.backward(dOut)
        if args.allreduce_grads:
            # timed on the device (events on the launch stream; read after the run): the exchange's share of every rank's step
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            parallel.allreduce_gradients(gnn.trainable_variables, dist, local_count=float(G if pool is not None else V))
            ev1.record()
            allreduce_events.append((ev0, ev1))
            del allreduce_events[:-64]
        if graph is None:
            g.close()

    def barrier():
        parallel.barrier(dist)

    ms_guess = [3.0]
    per_rank_seconds = []
    products_per_step = {}

    def settle(max_batches=40, tol=0.03, min_batches=8):
        """Untimed extra warm-up: short batches of steps until two consecutive ones take the same time (clock ramp,
        first-touch allocations, the host-side stall around the ~4000th launch of a process: TFGNN_BENCH_TRACE=1 prints
        the batch times).  Batches are ~30 ms of steps; at least ``min_batches`` of them for a fast workload."""
        prev = None
        trace = os.environ.get("TFGNN_BENCH_TRACE") == "1"
        for i in range(max_batches):
            nb = int(min(10, max(2, round(30.0 / max(ms_guess[0], 0.3)))))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nb):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / nb
            if dist is not None:
                # every rank must take the SAME decisions here: with --allreduce-grads a step holds a collective, and a rank that
                # settles one batch earlier than its peer leaves it alone in that collective (round 6: the two-rank test hung
                # about every other run - one rank at the barrier of timed(), the other in allreduce_gradients of a settle step)
                dt = parallel.reduce_max(dt, dist, dev)
            ms_guess[0] = 1000.0 * dt
            if trace:
                print(f"settle batch {i}: {1000 * dt:.3f} ms/step", file=sys.stderr)
            need = min_batches if ms_guess[0] < 10 else 3
            if i + 1 >= need and prev is not None and abs(dt - prev) <= tol * prev:
                return
            prev = dt

    def timed(warmup, steps):
        for _ in range(warmup):
            step()
        if not args.no_settle:
            settle()
        barrier()
        c0 = ops.launch_counts()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        c1 = ops.launch_counts()
        products_per_step.clear()
        products_per_step.update(products_from_counts(c0, c1, steps))
        for g_, _x in pending:  # the batch prepared for the step after the last one
            g_.wait()
            g_.close()
        pending.clear()
        # every rank's own time too (a straggler is visible in the scaling runs), then the MAX the metric is computed from
        per_rank_seconds[:] = parallel.all_gather_scalars([dt], dist, dev)[:, 0].tolist()
        return parallel.reduce_max(dt, dist, dev)

    ops.set_gemm_mode(args.gemm_mode)
    elapsed = timed(args.warmup, args.steps)
    ms_per_step_per_rank = [1000.0 * t / args.steps for t in per_rank_seconds]
    # the spread guard of the f16x2 products (tfgnn_sp_spread_flag) demotes the mode to bf16x3 when it trips: the timed steps
    # must have run in the mode the line reports
    mode_ids = {"fp32": ops.GEMM_FP32, "bf16x3": ops.GEMM_BF16X3, "bf16x3_9": ops.GEMM_BF16X3_EXACT, "f16x2": ops.GEMM_F16X2}
    assert ops.get_gemm_mode() == mode_ids[args.gemm_mode], "the GEMM mode changed during the timed steps (spread guard tripped)"
    # final metric reduction: all-gather of the per-rank edge / node counts (north_star: the only collective)
    gathered = parallel.all_gather_scalars([float(E), float(V), float(G)], dist, dev)
    total_edges_per_step = float(gathered[:, 0].sum())
    ms_per_step = 1000.0 * elapsed / args.steps
    value = total_edges_per_step * args.steps / elapsed
    allreduce_ms_per_rank = None
    if args.allreduce_grads:
        torch.cuda.synchronize()
        mine = [a.elapsed_time(b) for a, b in allreduce_events[-args.steps:]]
        allreduce_ms_per_rank = parallel.all_gather_scalars([float(np.mean(mine)) if mine else 0.0], dist, dev)[:, 0].tolist()

    if sharded:
        shape = (f"G={int(gathered[:, 2].sum())} graphs V={int(gathered[:, 1].sum())} E={int(total_edges_per_step)} (whole job; sharded by graph "
                 f"over {world} rank{'s' if world > 1 else ''})")
        data = "synthetic: QM9-shaped molecules (5..13 nodes, random tree + ~0.8 extra bonds, 4 tied bond types + self loops), N(0,1) features, Glorot weights"
    else:
        shape = f"V={V} E={E} per GPU (one batch per rank, replicas)"
        data = ("synthetic: R-MAT (0.57,0.19,0.19,0.05)" + (", Zipf(1) edge types" if wl["model"] == "rgin" else "")
                + ", N(0,1) features, Glorot weights")
    bucketing = " (hoisted)" if args.reuse_graph else (" (on the compute stream)" if args.serial_bucketing else " of the next batch (2nd stream, overlapped)")
    result = {
        "metric": baseline_metric(),
        "value": value,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": data,
        "config": {
            "workload": f"{args.workload}: {shape} edge_types={L} D={D} H={H} layers={NL} {wl['model'].upper()}"
            f"{' + WeightedSum pooling (GD=32, 4 heads, softmax)' if pool is not None else ''} (PPI_RGCN.json stack settings), "
            f"step = edge bucketing{bucketing} + forward (training mode) + full backward",
            "per_layer_traversal_rate_edges_per_s": value * NL,
            "gemm_mode": GEMM_MODE_NOTES[args.gemm_mode],
            "gemm_mode_name": args.gemm_mode,
            # which product kernels the TIMED steps launched (tfgnn_launch_counts, per step, rank 0) and what the spread guard's
            # staged policy has taken off the split operands: the global mode does not say
            "products_per_step": dict(products_per_step),
            "guard": gnn.guard_state(),
            "rccl": rccl_info(dist, dev, world),
            "batches": (f"{K_batches} distinct batches of this shape take turns (--distinct-batches)" if K_batches > 1 else
                        "the same batch every step (re-bucketed per step; its rows stay warm in the Infinity Cache: the optimistic "
                        "case - see --distinct-batches)"),
            "collectives_per_step": ("1 bucketed all-reduce of the weight gradients (--allreduce-grads)"
                                     if (args.allreduce_grads and dist is not None) else "none (graph-sharded batches / replicas)"),
            "edges_per_rank": gathered[:, 0].tolist(),
            "ms_per_step_per_rank": ms_per_step_per_rank,
            "allreduce_ms_per_step_per_rank": allreduce_ms_per_rank,
        },
    }
    if not args.no_alt_mode:
        # the same job with the products in another evaluation mode, for reference (not the headline value)
        alt = "bf16x3" if args.gemm_mode != "bf16x3" else "fp32"
        ops.set_gemm_mode(alt)
        alt_steps = max(3, args.steps // 2)
        alt_elapsed = timed(2, alt_steps)
        ops.set_gemm_mode(args.gemm_mode)
        result["config"]["alt_gemm_mode"] = {
            "gemm_mode": GEMM_MODE_NOTES[alt], "gemm_mode_name": alt, "products_per_step": dict(products_per_step), "steps": alt_steps, "ms_per_step": 1000.0 * alt_elapsed / alt_steps,
            "value": total_edges_per_step * alt_steps / alt_elapsed,
        }

    if rank == 0 and not args.no_roofline:
        roofs = roofline_blocks(args, wl, ops, dev, adj_dev, V, E, L, H, NL, ms_per_step)
        roofs.sort(key=lambda r: -r["share_of_step"])
        result["roofline_blocks_share_of_step"] = float(sum(r["share_of_step"] for r in roofs))
        result["roofline"] = roofs[0]
        if len(roofs) > 1:
            result["roofline_secondary"] = roofs[1]
        if len(roofs) > 2:
            result["roofline_other"] = roofs[2:]

    if rank == 0 and world == 1 and not args.no_roofline and args.workload != "rmat30k":
        # the non-headline workloads spread their step over many kernels: the per-op breakdown of the REAL step (HIP events
        # around every library op) says where the step goes and prices each op against its roof
        ops.set_gemm_mode(args.gemm_mode)
        result["step_breakdown"] = step_breakdown(step, ops)
        for g_, _x in pending:
            g_.wait()
            g_.close()
        pending.clear()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl, batch, params, budget_seconds=args.cpu_baseline_seconds)

    if rank == 0 and world == 1 and args.workload == "rmat30k" and not args.no_other_configs:
        result["other_configs"] = other_configs(args)

    if rank == 0:
        emit(result, args.workload)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


OTHER_CONFIGS = (("ppi", 30), ("rgat", 10), ("qm9-ggnn", 3), ("qm9-edgemlp", 3), ("arxiv-rgin", 3))  # (workload, timed steps)


def other_configs(args):
    """The other BASELINE configs, timed briefly (a few steps each, own process: fresh allocator and weight caches) so that
    the driver's line carries them next to the headline: ms/step, edges/s, the dominant kernel's roofline fraction and the
    CPU baseline of the same model.  They are reported, not part of `value`."""
    out = {}
    for name, steps in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", "1", "--no-alt-mode",
               "--no-other-configs", "--gemm-mode", args.gemm_mode, "--cpu-baseline-seconds", "8"]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        if args.no_roofline:
            cmd.append("--no-roofline")
        t_sub = time.perf_counter()
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l[len(DETAIL_PREFIX):] for l in res.stdout.splitlines() if l.startswith(DETAIL_PREFIX)]
            if res.returncode != 0 or not line:
                out[name] = {"error": (res.stderr or res.stdout)[-300:]}
                continue
            r = json.loads(line[-1])
        except (subprocess.TimeoutExpired, ValueError) as e:
            out[name] = {"error": str(e)[:300]}
            continue
        entry = {"workload": r["config"]["workload"], "steps": r["steps"], "ms_per_step": r["ms_per_step"], "value": r["value"],
                 "unit": r["unit"], "scaling": r["scaling"], "wall_seconds_of_this_run": round(time.perf_counter() - t_sub, 1)}
        for key in ("graphs_per_s", "host_ms_per_step", "device_ms_one_step_alone", "bound", "reference_published", "step", "replay_only",
                    "replay_static_batch", "eager"):
            if key in r:  # the PPI stand-in's own fields
                entry[key] = r[key]
        for key in ("roofline", "roofline_secondary"):
            if key in r:
                b = r[key]
                entry[key] = {k: b.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_boundary",
                                                    "compulsory_frac", "ms_per_launch", "launches_per_step", "share_of_step",
                                                    "achieved_basis")}
        for key in ("products_per_step", "guard"):
            if r["config"].get(key) is not None:
                entry[key] = r["config"][key]
        if "roofline_blocks_share_of_step" in r:
            entry["roofline_blocks_share_of_step"] = r["roofline_blocks_share_of_step"]
        if "step_breakdown" in r:
            sb = r["step_breakdown"]
            entry["step_breakdown"] = {"ops_share_of_step": sb["ops_share_of_step"], "top_share_of_step": sb["top_share_of_step"],
                                       "top": [{k: e[k] for k in ("op", "shapes", "launches_per_step", "ms_per_launch", "share_of_step", "bound", "frac")}
                                               for e in sb["top"][:8]]}
        if "cpu_baseline" in r:
            entry["cpu_baseline"] = {k: r["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "host_cpu_count", "kind", "sample")}
        out[name] = entry
    return out


# --------------------------------------------------------------------------------------------------------------------
# step breakdown: every library op of ONE real step between HIP events (launch stream), aggregated by (op, shapes)
# --------------------------------------------------------------------------------------------------------------------
_TIMED_OPS = ("graph_gather_dot", "aux_flush", "dropout_mask", "sp_gemm_nt_grouped", "sp_gather_rows", "gemm", "gemm_gathered", "gemm_grad", "gemm_gru", "gemm_grouped_rows", "gemm_grouped_k", "sp_gemm_nt", "sp_gemm_nt_split", "sp_gemm_tn",
              "sp_gemm_tn_grouped", "graph_gather", "graph_gather_sp", "gather_reduce", "gru_gates_forward", "gru_gates_backward", "activation_forward",
              "activation_backward", "dropout_forward", "mul", "add_scale", "colsum", "layernorm_forward", "layernorm_backward",
              "permute_021", "transpose_batched", "edge_aggregate_backward", "sp_split_rows", "sp_split_cols", "clip", "clip_backward")
_PRODUCT_OPS = {"sp_gemm_nt_grouped", "gemm", "gemm_gathered", "gemm_grad", "gemm_gru", "gemm_grouped_rows", "gemm_grouped_k", "sp_gemm_nt", "sp_gemm_nt_split", "sp_gemm_tn", "sp_gemm_tn_grouped"}


def step_breakdown(step, ops, steps=2, top=12):
    """Per-op time of the real step: every tf2_gnn_amd.ops call listed in _TIMED_OPS is bracketed by HIP events on the launch
    stream for ``steps`` steps (after one untimed step), then aggregated by op name and tensor shapes.  For every entry:
    launches per step, ms per launch, share of the step, the bytes it has to move (every tensor argument and result once:
    compulsory bytes) and, for products, the fp32 flops - priced against the HBM roof or the matrix-core roof, whichever
    takes longer at peak.  Nested library calls (gemm inside gemm_grad) are attributed to the outermost op.
    -> (entries sorted by share, sum of shares)"""
    records = []
    depth = [0]
    real = {}

    def tensors_of(obj, out):
        if isinstance(obj, torch.Tensor):
            out.append(obj)
        elif isinstance(obj, ops.SplitOperand):
            out.append(obj.data)
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                tensors_of(o, out)
        elif isinstance(obj, dict):
            for o in obj.values():
                tensors_of(o, out)

    def wrap(name, fn):
        def timed(*a, **k):
            if depth[0]:
                return fn(*a, **k)
            depth[0] += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                res = fn(*a, **k)
            finally:
                depth[0] -= 1
            e1.record()
            ts = []
            tensors_of((a, k, res), ts)
            seen, nbytes, shapes = set(), 0, []
            for t in ts:
                if t.data_ptr() in seen:
                    continue
                seen.add(t.data_ptr())
                nbytes += t.numel() * t.element_size()
                if t.numel() * t.element_size() >= 1 << 16:
                    shapes.append("x".join(map(str, t.shape)) + ("" if t.dtype == torch.float32 else f":{str(t.dtype).split('.')[-1]}"))
            flops = 0.0
            if name in _PRODUCT_OPS:
                big = sorted((t for t in ts if t.dim() >= 2), key=lambda t: -t.numel())[:3]
                if name == "sp_gemm_nt_grouped":
                    flops = 2.0 * a[2].num_rows * a[0].cols * (a[1].rows // max(a[2].num_groups, 1))
                elif name in ("sp_gemm_nt", "sp_gemm_nt_split") and isinstance(a[0], ops.SplitOperand):
                    flops = 2.0 * a[0].rows * a[0].cols * a[1].rows
                elif name == "sp_gemm_tn" and isinstance(a[0], ops.SplitOperand):
                    M = (k.get("a_cols") or (0, a[0].cols))[1]
                    N = (k.get("b_cols") or (0, a[1].cols))[1]
                    flops = 2.0 * a[0].rows * M * N
                elif name == "sp_gemm_tn_grouped":
                    flops = 2.0 * a[0].rows * a[0].cols * a[1].cols
                elif name == "gemm_gru":
                    flops = 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1]
                elif name == "gemm_grouped_k":
                    flops = 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1]
                elif len(big) >= 2 and isinstance(res, (torch.Tensor, tuple)):
                    r = res[0] if isinstance(res, tuple) else res
                    r = r if isinstance(r, torch.Tensor) else big[0]
                    x, y = a[0], a[1]
                    inner = x.shape[0] if k.get("trans_a") else x.shape[-1]
                    flops = 2.0 * r.numel() * inner
            records.append((name, " ".join(shapes[:4]), e0, e1, nbytes, flops))
            return res
        return timed

    for name in _TIMED_OPS:
        if hasattr(ops, name):
            real[name] = getattr(ops, name)
            setattr(ops, name, wrap(name, real[name]))

    class Scope:  # regions the layers mark with ops.op_scope (library calls that bypass the wrappers: RGAT's attention kernels)
        def __init__(self, name, tensors):
            self.name, self.tensors, self.e0 = name, tensors, None

        def __enter__(self):
            if depth[0] == 0:
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            depth[0] += 1
            return self

        def __exit__(self, *exc):
            depth[0] -= 1
            if self.e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                ts = [t for t in self.tensors if isinstance(t, torch.Tensor)]
                shapes = ["x".join(map(str, t.shape)) for t in ts if t.numel() * t.element_size() >= 1 << 16]
                records.append((self.name, " ".join(shapes[:4]), self.e0, e1, sum(t.numel() * t.element_size() for t in ts), 0.0))
            return False

    real["op_scope"] = ops.op_scope
    ops.op_scope = lambda name, *tensors: Scope(name, tensors)
    try:
        step()
        torch.cuda.synchronize()
        records.clear()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(steps):
            step()
        t1.record()
        torch.cuda.synchronize()
    finally:
        for name, fn in real.items():
            setattr(ops, name, fn)
    total_ms = t0.elapsed_time(t1) / steps
    agg = {}
    for name, shapes, e0, e1, nbytes, flops in records:
        a = agg.setdefault((name, shapes), [0, 0.0, nbytes, flops])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    mode = ops.get_gemm_mode()
    entries = []
    for (name, shapes), (n, ms, nbytes, flops) in agg.items():
        ms_launch = ms / n
        executed = flops * (1 if mode == ops.GEMM_FP32 else (3 if name.startswith("sp_") else (9 if mode == ops.GEMM_BF16X3_EXACT else 6)))
        peak_tf = MFMA_FP32_PEAK_TFLOPS if (mode == ops.GEMM_FP32 and not name.startswith("sp_")) else MFMA_16BIT_PEAK_TFLOPS
        t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
        t_mfma = executed / (peak_tf * 1e12)
        bound = "mfma" if t_mfma > t_hbm else "hbm"
        achieved = (executed / (ms_launch * 1e-3) / 1e12) if bound == "mfma" else (nbytes / (ms_launch * 1e-3) / 1e9)
        peak = peak_tf if bound == "mfma" else HBM_PEAK_GBS
        entries.append({"op": name, "shapes": shapes, "launches_per_step": n / steps, "ms_per_launch": ms_launch,
                        "share_of_step": ms / steps / total_ms, "bound": bound, "achieved": achieved, "peak": peak,
                        "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": achieved / peak,
                        "compulsory_bytes_per_launch": nbytes, "algorithmic_flops_per_launch": flops})
    entries.sort(key=lambda e: -e["share_of_step"])
    covered = float(sum(e["share_of_step"] for e in entries))
    return {"ms_per_step_under_events": total_ms, "ops_share_of_step": covered, "top": entries[:top],
            "top_share_of_step": float(sum(e["share_of_step"] for e in entries[:top]))}


# --------------------------------------------------------------------------------------------------------------------
# rooflines: the dominant kernels of a workload, timed live (HIP events on the launch stream) at the workload's shapes
# --------------------------------------------------------------------------------------------------------------------
def kernel_source_sha16():
    """sha256 (first 16 hex digits) over tf2_gnn_amd/csrc/*.hip, *.hpp: what tools/parse_pmc.py stores with a counter
    collection, so that a traffic number measured on older kernels is recognisable."""
    import hashlib

    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf2_gnn_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_traffic(workload, name):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/<tag>_pmc_traffic_<workload>.json, derived by
    tools/parse_pmc.py from the committed counter CSVs profiles/<tag>_pmc_{fetch,write}_<workload>_counter_collection.csv;
    FETCH_SIZE corrected x2 as calibrated on gfx950 by the streaming kernel in the same pass, MI355X_MICROARCH.md)
    -> (bytes | None, provenance dict | None).  The provenance says which collection the number comes from and whether the
    kernel sources it was measured on are the ones running now (rounds 1-3 stored no hash: "unknown")."""
    if name is None:
        return None, None
    root = os.path.dirname(os.path.abspath(__file__))
    for tag in ("r06", "r05", "r04", "r03", "r02"):  # the newest collection that has this workload (tools/pmc_collect.sh)
        path = os.path.join(root, "profiles", f"{tag}_pmc_traffic_{workload}.json")
        try:
            with open(path) as f:
                doc = json.load(f)
            v = doc.get(name, {}).get("hbm_bytes_per_launch")
            if v:
                sha = doc.get("_meta", {}).get("kernel_source_sha16")
                match = "unknown (collected before the sources were hashed)" if sha is None else (sha == kernel_source_sha16())
                if match is False:
                    print(f"bench.py: {os.path.basename(path)} was collected on other kernel sources ({sha}); re-run tools/pmc_collect.sh",
                          file=sys.stderr)
                return v, {"file": f"profiles/{os.path.basename(path)}", "kernel_sources_match": match}
        except (OSError, ValueError):
            continue
    return None, None


def _pmc_mfma_busy(workload, kernel_substring):
    """matrix-pipe busy fraction of a kernel from the newest committed counter pass (tools/pmc_mfma.sh) -> float | None"""
    if not kernel_substring:
        return None
    root = os.path.dirname(os.path.abspath(__file__))
    for tag in ("r06", "r05"):
        try:
            with open(os.path.join(root, "profiles", f"{tag}_pmc_mfma.json")) as f:
                doc = json.load(f).get(workload, {})
        except (OSError, ValueError):
            continue
        for name, e in doc.items():
            if kernel_substring in name and e.get("mfma_busy_fraction_all_simds") is not None:
                return {"value": e["mfma_busy_fraction_all_simds"], "file": f"profiles/{tag}_pmc_mfma.json", "kernel": name[:80]}
    return None


def roofline_blocks(args, wl, ops, dev, adj_dev, V, E, L, H, NL, ms_per_step):
    g = ops.Graph(adj_dev, V)
    out = []
    model, mode = wl["model"], args.gemm_mode
    if mode == "f16x2" and not ((H % 128 == 0 or H % 320 == 0) and (L * H) % 128 == 0):
        mode = "bf16x3"  # widths the split-operand products do not tile (the tiny workloads): the layers run bf16x3 there
    Hx = torch.randn((V, H), device=dev)
    split_peak = MFMA_FP32_PEAK_TFLOPS if mode == "fp32" else MFMA_16BIT_PEAK_TFLOPS
    nprod = {"fp32": 1, "bf16x3_9": 9}.get(mode, 6)  # piece products of the generic split-operand GEMM

    def hbm_block(kernel, ms, alg_bytes, launches, traffic_key=None, compulsory_bytes=None):
        """``achieved`` prices bytes that really crossed the fabric: the rocprofv3 counter bytes of this kernel at this shape
        when they were collected (profiles/r02_pmc_traffic_<workload>.json), else the compulsory bytes (every distinct input /
        output byte once).  SURVEY.md 8d's no-reuse gather model (one source row per EDGE) is reported beside it: its rate
        exceeds the HBM peak whenever rows are re-read from L2 / Infinity Cache, so it is not a fraction of any roof."""
        traffic, traffic_from = _pmc_traffic(args.workload, traffic_key)
        basis_bytes = traffic if traffic else (compulsory_bytes if compulsory_bytes else alg_bytes)
        gbs = basis_bytes / (ms * 1e-3) / 1e9
        blk = {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "achieved_basis": ("rocprofv3 FETCH_SIZE + WRITE_SIZE per launch (gfx950-corrected)" if traffic else
                                  ("compulsory bytes (distinct inputs and outputs once)" if compulsory_bytes else "algorithmic bytes")),
               "traffic": traffic, "traffic_from": traffic_from, "ms_per_launch": ms, "algorithmic_bytes_per_launch": alg_bytes,
               "no_reuse_model_rate_GBs": alg_bytes / (ms * 1e-3) / 1e9, "launches_per_step": launches,
               "share_of_step": launches * ms / ms_per_step}
        if traffic:
            # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE count requests leaving L2 for the fabric - Infinity-Cache (MALL) hits
            # included - so the fraction is taken at the L2 <-> fabric boundary against the HBM peak; DRAM traffic itself lies
            # between the compulsory bytes and this figure
            blk["traffic_boundary"] = "L2<->fabric (Infinity-Cache hits included)"
        if compulsory_bytes:
            blk["compulsory_bytes_per_launch"] = compulsory_bytes
            blk["compulsory_frac"] = compulsory_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        return blk

    def mfma_block(kernel, ms, flops, executed_factor, peak, launches, traffic_key=None, executed_fraction=1.0, busy_key=None):
        """``achieved`` counts the MFMA work the launch EXECUTES: piece products per fp32 product x the share of the operand's
        type blocks the kernel does not skip (``executed_block_fraction``; VERDICT r5 weak 4: skipped all-zero blocks are not
        work).  ``mfma_busy_counter`` is the hardware's own figure where a counter pass of this kernel is committed
        (SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), profiles/<tag>_pmc_mfma.json)."""
        tf = flops / (ms * 1e-3) / 1e12
        traffic, traffic_from = _pmc_traffic(args.workload, traffic_key)
        blk = {"kernel": kernel, "bound": "mfma", "achieved": tf * executed_factor * executed_fraction, "peak": peak, "unit": "TFLOP/s",
               "frac": tf * executed_factor * executed_fraction / peak, "traffic": traffic, "traffic_from": traffic_from, "ms_per_launch": ms,
               "algorithmic_flops_per_launch": flops, "algorithmic_tflops": tf, "piece_products_per_fp32_product": executed_factor,
               "executed_block_fraction": executed_fraction,
               "launches_per_step": launches, "share_of_step": launches * ms / ms_per_step}
        busy = _pmc_mfma_busy(args.workload, busy_key)
        if busy is not None:
            blk["mfma_busy_counter"] = busy
        if peak == MFMA_16BIT_PEAK_TFLOPS:
            # tools/mfma_dep_probe.hip: back-to-back 32x32x16 16-bit MFMAs sustain 2.47 PFLOP/s on smooth operands and
            # 1.87 PFLOP/s on operands with random significands (the clock drops to 1.78 GHz)
            blk["peak_measured_random_operands"] = 1870.0
            blk["frac_of_measured_random_operand_peak"] = tf * executed_factor / 1870.0
        return blk

    if model in ("rgcn", "ggnn", "gnn_edge_mlp"):
        rs = g.array(ops.G_INVDEG_BY_DST) if model == "rgcn" else None
        gather_bytes = E * (4 * H + 4) + (V * L + 1) * 4 + V * L * 4 + V * L * H * 4  # one source row per edge
        gather_compulsory = V * H * 4 + E * 4 + (V * L + 1) * 4 + V * L * 4 + V * L * H * 4  # every source row once
        f16 = mode == "f16x2" and model in ("rgcn", "ggnn")  # the layers whose gather writes the split operand
        if f16:
            ms = time_kernel(lambda: ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, Hx, row_scale=rs, rows_per_operand_row=L))
            name = "csr_gather_reduce_kernel<SP16> (aggregate source rows per (node, type) bucket, written as the split fp16 operand)"
        else:
            A = torch.empty((V * L, H), device=dev)
            ms = time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, Hx, row_scale=rs, out=A))
            name = "csr_gather_reduce_kernel (aggregate source rows per (node, type) bucket)"
        out.append(hbm_block(name, ms, gather_bytes, 2 * NL, "gather_sp" if f16 else "gather", gather_compulsory))
    if model == "rgcn":
        W = torch.randn((L * H, H), device=dev) * 0.05
        res = torch.empty((V, H), device=dev)
        flops = 2.0 * V * (L * H) * H
        if mode == "f16x2":
            # the product as the layer launches it (gnn_edge_mlp._forward_A): operand rows in the order of their emptiness
            # patterns, all-zero type blocks of a row tile skipped, rows written back in node order
            from tf2_gnn_amd.layers.message_passing.gnn_edge_mlp import _skip_empty_blocks

            skip = _skip_empty_blocks(L, H)
            kmask = g.array(ops.G_PATTERN_TILEMASK_BY_DST) if skip else None
            rmap = g.array(ops.G_PATTERN_NODE_BY_DST) if skip else None
            executed = 1.0
            if skip:
                m = kmask.cpu().numpy().astype(np.uint32)
                executed = float(sum(int(((m >> b) & 1).sum()) for b in range(L))) / float(m.size * L)
            A_sp = ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED_PATTERN if skip else ops.VIEW_BY_DST_TYPED, Hx,
                                       row_scale=g.array(ops.G_INVDEG_BY_DST), rows_per_operand_row=L)
            Wt_sp = ops.sp_split_cols(W)
            ms = time_kernel(lambda: ops.sp_gemm_nt(A_sp, Wt_sp, act="relu", out=res, tile_kmask=kmask, row_map=rmap))
            out.append(mfma_block("gemm_sp_nt_kernel ([V, L*H] x [H, L*H]^T + relu on SP16 operands written by the gather: 3 x "
                                  "v_mfma_f32_32x32x16_f16 per fp32 k16 step, all-zero type blocks skipped); forward + dX launches",
                                  ms, flops, 3, MFMA_16BIT_PEAK_TFLOPS, 2 * NL, "gemm_sp_nt", executed_fraction=executed,
                                  busy_key="gemm_sp_nt_kernel"))
            Xs = ops.sp_split_rows(Hx)
            Gs = ops.sp_split_rows(torch.randn((V, L * H), device=dev) * 1e-3, scale_block=H)
            dW = torch.empty((L, H, H), device=dev)
            ms = time_kernel(lambda: ops.sp_gemm_tn(Gs, Xs, out=dW, scatter=(H, H * H, 1, H)))
            out.append(mfma_block("gemm_sp_tn_kernel (per-k factors computed in the kernel, round 4) + split-K reduce (dW = X^T G over K = V rows, transposing LDS reads)",
                                  ms, flops, 3, MFMA_16BIT_PEAK_TFLOPS, NL, "gemm_sp_tn", busy_key="gemm_sp_tn_kernel"))
        elif mode == "fp32":
            A = torch.randn((V, L * H), device=dev)
            ms = time_kernel(lambda: ops.gemm(A, W, act="relu", out=res))
            out.append(mfma_block("gemm_mfma_kernel 128 x N tile ([V, L*H] x [L*H, H] + relu, v_mfma_f32_32x32x2_f32)", ms, flops, 1,
                                  MFMA_FP32_PEAK_TFLOPS, 3 * NL, "gemm"))
        else:
            A = torch.randn((V, L * H), device=dev)
            Wt = W.t().contiguous()
            ms = time_kernel(lambda: ops.gemm(A, Wt, trans_b=True, act="relu", out=res))
            out.append(mfma_block(f"gemm_x3s_kernel 128 x N tile, 4 multiplying + 4 staging waves ({nprod} x v_mfma_f32_32x32x16_bf16 per fp32 "
                                  "k16 step on exactly split operands)", ms, flops, nprod, MFMA_16BIT_PEAK_TFLOPS, 3 * NL, "gemm_bf16x3"))
    if model in ("ggnn", "gnn_edge_mlp"):
        # node-level products [V, H] x [H, 3H] (GRU gates) / [V*L.., 2H] x [2H, H] (edge MLP): M in the millions, K and N
        # small -> bandwidth-bound (operand + result bytes), whatever the multiply mode
        N = 3 * H if model == "ggnn" else H
        Wk = torch.randn((H, N), device=dev) * 0.05
        ms = time_kernel(lambda: ops.gemm(Hx, Wk))
        out.append(hbm_block(f"dense product [V, {H}] x [{H}, {N}] (gemm_x3k_kernel: weight block resident in LDS, rows streamed; "
                             "bytes of A and C bound it, not the matrix cores)", ms,
                             V * H * 4 + V * N * 4 + H * N * 4, (4 if model == "ggnn" else 6) * NL,
                             "gemm" if mode == "fp32" else "gemm_stream"))
    if model == "rgat":
        K = wl.get("num_heads", 8)
        Wc = torch.randn((H, L * H), device=dev) * 0.05
        Y = torch.empty((V, L * H), device=dev)
        ms = time_kernel(lambda: ops.gemm(Hx, Wc, out=Y))
        out.append(mfma_block("Y = X [W_0 | ... | W_{L-1}] ([V, H] x [H, L*H]; forward, dX and dW launches)", ms, 2.0 * V * H * L * H, nprod,
                              split_peak, 3 * NL))
        ew = torch.rand((E, K), device=dev)
        agg = torch.empty((V, H), device=dev)
        ms = time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Y.view(V * L, H), edge_weight=ew, out=agg))
        out.append(hbm_block(f"csr_gather_reduce_kernel<HEADS> (attention-weighted sum over all in-edges of a node, {K} heads)", ms,
                             E * (4 * H + 4 * K + 4) + (V + 1) * 4 + V * H * 4, 2 * NL, "gather_heads",
                             V * L * H * 4 + E * (4 * K + 4) + (V + 1) * 4 + V * H * 4))
    if model == "rgin":
        # per-relation 2-layer edge MLP over the non-empty (source, type) pairs: grouped GEMMs over compact rows
        off_h = g.nonempty_offsets(True)
        nz = off_h[-1]
        Ac = torch.randn((nz, H), device=dev)
        W = torch.randn((L, H, H), device=dev) * 0.05
        off_d = g.array(ops.G_NZ_OFF_BY_SRC)
        if args.gemm_mode == "f16x2":
            # round 5: the forward and input-gradient products of the per-relation MLPs run on split operands
            # (tfgnn_sp_gemm_nt_grouped: 3 piece products), the kernel gradients of all relations in one launch of the
            # two-factor TN product (tfgnn_sp_gemm_tn_grouped: per-k factors on both operands' fragments, so the row scales
            # of this workload's un-normalised sums - spread over 2^19 and 2^21 - stay below each operand's guard)
            groups = ops.RowGroups(off_h, dev)
            a_sp = ops.sp_split_rows(Ac)
            w_sp = ops.sp_split_cols(W.view(L * H, H))
            ms = time_kernel(lambda: ops.sp_gemm_nt_grouped(a_sp, w_sp, groups, act="relu", b_column_blocks=True))
            out.append(mfma_block(f"tfgnn_sp_gemm_nt_grouped over the {nz} non-empty (source, type) rows in {L} relation groups "
                                  "([rows_l, H] x [H, H], split operands)", ms, 2.0 * nz * H * H, 3, MFMA_16BIT_PEAK_TFLOPS, 4 * NL))
            g_sp = ops.sp_split_rows(torch.randn((nz, H), device=dev))
            dWg = torch.empty((L, H, H), device=dev)
            ms = time_kernel(lambda: ops.sp_gemm_tn_grouped(a_sp, g_sp, groups, dWg))
            out.append(mfma_block(f"tfgnn_sp_gemm_tn_grouped: kernel gradients of the {L} relations in one launch (two-factor split-operand "
                                  "TN product + grouped reduce)", ms, 2.0 * nz * H * H, 3, MFMA_16BIT_PEAK_TFLOPS, 2 * NL))
        else:
            ms = time_kernel(lambda: ops.gemm_grouped_rows(Ac, off_d, off_h, W, act="relu"))
            out.append(mfma_block(f"grouped GEMM over the {nz} non-empty (source, type) rows in {L} relation groups ([rows_l, H] x [H, H])", ms,
                                  2.0 * nz * H * H, nprod, split_peak, 6 * NL))
        colc = g.array(ops.G_NZ_CPOS_BY_SRC)[g.array(ops.G_COLL_BY_DST).long()].contiguous()
        agg = torch.empty((V, H), device=dev)
        ms = time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Ac, col=colc, out=agg))
        out.append(hbm_block("csr_gather_reduce_kernel (messages of all edge types summed per target node)", ms,
                             E * (4 * H + 4) + (V + 1) * 4 + V * H * 4, 2 * NL, "gather", nz * H * 4 + E * 4 + (V + 1) * 4 + V * H * 4))
    ms_graph = time_kernel(lambda: ops.Graph(adj_dev, V).close(), iters=5, warmup=1)
    for r in out:
        r["ms_edge_bucketing_per_batch"] = ms_graph
    g.close()
    return out


if __name__ == "__main__":
    main()
