#!/usr/bin/env python3
"""Benchmark of the message-passing hot path: edges/sec (fwd+bwd), RGCN H=320 L=4 on an R-MAT batch
shaped like BASELINE.json configs[1] (30k nodes, 900k edges, 4 edge types), 1..8 MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one batch: bucket the batch's edges (ops.Graph), GNN
forward (initial projection, 4 RGCN layers, dropout, the Dense after layer 0 - the op sequence of
tf2_gnn_train RGCN PPI) and the full backward (all weight gradients).  Inputs are
resident in HBM before the timed region.  Multi-GPU: graph batches are independent, each rank
processes its own batch of the same shape (weak scaling, no data-path collective); one RCCL
all-gather collects the per-rank edge counts / times for the metric.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel, measured live with HIP
events on the launch stream; `cpu_baseline` times the CPU oracle (reference op sequence, torch-CPU)
on rank 0 when N == 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_FP32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; no sparsity)
GEMM_MODE_NOTES = {
    "fp32": "fp32: v_mfma_f32_32x32x2_f32 on the fp32 operands",
    "bf16x3": "bf16x3: fp32 operands split exactly into 3 bf16 pieces, 6 largest piece products (each exact in fp32), fp32 "
    "accumulate on v_mfma_f32_32x32x16_bf16; fp32 in/out; max error vs fp64 measured equal to the fp32-MFMA kernel "
    "(tests/test_gpu_ops.py::test_gemm_bf16x3_matches_fp64)",
    "bf16x3_9": "bf16x3_9: as bf16x3 with all 9 piece products (products exact, only the fp32 accumulation rounds)",
    "f16x2": "f16x2: the layers' products on pre-split operands - every fp32 value scaled by a power of two and split into two "
    "fp16 pieces by round-to-nearest (22+ significand bits), 3 piece products (each exact in fp32), fp32 accumulate on "
    "v_mfma_f32_32x32x16_f16, the gather writes the split operand; fp32 in/out; accumulation error vs fp64 measured below the "
    "fp32-MFMA kernel's (tools/mfma_acc_probe.hip, tests/test_gpu_full_size.py); other products as in bf16x3",
}

WORKLOADS = {
    # BASELINE.json configs[1] shape + metric's model (RGCN H=320, 4 layers, PPI_RGCN.json hypers)
    "rmat30k": dict(num_nodes=30000, num_edges=900000, num_edge_types=4, feature_dim=320, hidden_dim=320, num_layers=4),
    # small variant for smoke runs
    "tiny": dict(num_nodes=2000, num_edges=40000, num_edge_types=4, feature_dim=64, hidden_dim=64, num_layers=4),
}


def ppi_rgcn_params(hidden_dim, num_layers):
    """tf2_gnn/cli_utils/default_hypers/PPI_RGCN.json:6-19 on top of GNN/RGCN defaults."""
    from tf2_gnn_amd.layers import GNN

    p = GNN.get_default_hyperparameters("rgcn")
    p.update(
        {
            "num_layers": num_layers,
            "hidden_dim": hidden_dim,
            "use_target_state_as_input": False,
            "normalize_by_num_incoming": True,
            "num_edge_MLP_hidden_layers": 0,
            "layer_input_dropout_rate": 0.1,
            "dense_every_num_layers": 10000,
            "residual_every_num_layers": 10000,
            "global_exchange_every_num_layers": 10000,
            "use_inter_layer_layernorm": False,
            "initial_node_representation_activation": "tanh",
            "dense_intermediate_layer_activation": "tanh",
            "message_activation_function": "ReLU",
            "aggregation_function": "sum",
        }
    )
    return p


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


def cpu_baseline(wl, feats, adjs, params, max_seconds=30.0):
    """The reference's CPU cost structure: oracle restatement (per-edge gathers and matmuls, concat,
    scatter-add; torch-CPU fp32, autograd for the backward), all host cores."""
    from oracle import tf2gnn_oracle as orc

    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    H, NL, L = wl["hidden_dim"], wl["num_layers"], wl["num_edge_types"]
    g = torch.Generator().manual_seed(0)

    def glorot(i, o):
        lim = (6.0 / (i + o)) ** 0.5
        return ((torch.rand((i, o), generator=g) * 2 - 1) * lim).requires_grad_(True)

    weights = {
        "initial_projection": glorot(wl["feature_dim"], H),
        "mp": [{"edge_mlps": [[glorot(H, H)] for _ in range(L)]} for _ in range(NL)],
        "dense": {0: glorot(H, H)},
        "layernorm": [],
    }
    leaves = [weights["initial_projection"], weights["dense"][0]] + [w[0] for m in weights["mp"] for w in m["edge_mlps"]]
    X = torch.from_numpy(feats)
    adj_t = [torch.from_numpy(a) for a in adjs]
    E = sum(a.shape[0] for a in adjs)

    def step(layers):
        p = dict(params, num_layers=layers)
        out, _ = orc.gnn_internal_call(p, weights, X, adj_t)
        torch.autograd.grad(out.sum(), leaves[:2] + leaves[2 : 2 + layers * L])

    # calibrate on one layer, then time as many layers as fit the budget (scaled to the full stack)
    t0 = time.perf_counter()
    step(1)
    t1 = time.perf_counter() - t0
    layers = NL if t1 * NL * 1.2 < max_seconds else max(1, int(max_seconds / (1.2 * t1)))
    if layers == 1:
        t = t1  # the calibration run is the sample (one layer of this batch already takes ~20 s on 256 threads)
    else:
        t0 = time.perf_counter()
        step(layers)
        t = time.perf_counter() - t0
    # the initial projection / dense glue is <2% of a layer: scale linearly in the number of layers
    t_full = t * NL / layers
    return {
        "value": E / t_full,
        "unit": "edges/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{layers} of {NL} RGCN layers fwd+bwd on the full batch ({t:.2f} s measured"
        f"{'' if layers == NL else ', scaled to ' + str(NL) + ' layers'}); torch-CPU fp32 restatement of the "
        "reference op sequence (TensorFlow unavailable offline)",
        "seconds_per_step": t_full,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="rmat30k", choices=sorted(WORKLOADS))
    ap.add_argument("--reuse-graph", action="store_true", help="bucket the edges once, outside the timed steps")
    ap.add_argument("--serial-bucketing", action="store_true", help="bucket each batch on the compute stream (no overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm-mode", default="f16x2", choices=["fp32", "bf16x3", "bf16x3_9", "f16x2"],
                    help="how the fp32 GEMMs run on the matrix cores (include/tfgnn.h, tfgnn_gemm_set_mode): fp32 MFMA, "
                    "or exact bf16 operand splitting with 6 / 9 piece products (fp32 in, fp32 accumulate, fp32 out)")
    ap.add_argument("--no-alt-mode", action="store_true", help="skip the second timing in the other GEMM mode")
    ap.add_argument("--allreduce-grads", action="store_true",
                    help="N > 1: add the training-step exchange (one bucketed RCCL all-reduce of the weight gradients) to every "
                         "step; off by default - the fwd+bwd metric itself has no collective")
    args = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a ROCm device (there is no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from tf2_gnn_amd import parallel

    rank, world, dist = parallel.init_distributed(device=dev)  # nccl == RCCL over xGMI on ROCm

    from tf2_gnn_amd import ops
    from tf2_gnn_amd.data import make_synthetic_batch
    from tf2_gnn_amd.layers import GNN, GNNInput
    from tf2_gnn_amd.layers.message_passing import set_seed

    wl = WORKLOADS[args.workload]
    V, E, L, D, H, NL = (wl[k] for k in ("num_nodes", "num_edges", "num_edge_types", "feature_dim", "hidden_dim", "num_layers"))
    feats, adjs = make_synthetic_batch(V, E, L, D, seed=1 + rank)  # timing seeds 1..
    X = torch.from_numpy(feats).to(dev)
    adj_dev = tuple(torch.from_numpy(a).to(dev) for a in adjs)
    n2g = torch.zeros(V, dtype=torch.int32, device=dev)
    params = ppi_rgcn_params(H, NL)
    set_seed(0)
    gnn = GNN(params)
    dOut = torch.randn((V, H), generator=torch.Generator().manual_seed(0)).to(dev)
    graph = ops.Graph(adj_dev, V) if args.reuse_graph else None
    # Input pipeline: every step buckets one batch's edges (ops.Graph).  Like the reference, whose
    # batches are prepared by a background thread + tf.data prefetch while the previous step trains
    # (data/graph_dataset.py:292-295, cli_utils/training_utils.py:114-115), the bucketing of batch
    # i+1 is enqueued on a second HIP stream at the start of step i and overlaps with its compute.
    # --serial-bucketing keeps it on the compute stream instead.
    side = torch.cuda.Stream() if (graph is None and not args.serial_bucketing) else None
    pending = []

    def enqueue_bucketing():
        if side is None:
            return ops.Graph(adj_dev, V)
        with torch.cuda.stream(side):
            return ops.Graph(adj_dev, V, wait=False)

    def step():
        if graph is not None:
            g = graph
        elif side is None:
            g = enqueue_bucketing()
        else:
            if not pending:
                pending.append(enqueue_bucketing())
            g = pending.pop(0)
            torch.cuda.current_stream().wait_stream(side)  # compute waits for THIS batch's bucketing
            g.wait()
            pending.append(enqueue_bucketing())  # next batch, overlapped with this step
        gnn(GNNInput(X, g, n2g, 1), training=True)
        gnn.backward(dOut)
        if args.allreduce_grads:
            parallel.allreduce_gradients(gnn.trainable_variables, dist)
        if graph is None:
            g.close()

    def barrier():
        parallel.barrier(dist)

    def settle(max_batches=40, batch=10, tol=0.03, min_batches=8):
        """Untimed extra warm-up: at least ``min_batches`` short batches, then until two consecutive ones take the
        same time.  Every process shows one ~35 ms host-side stall around its 4000th kernel launch (step 45-50 here;
        TFGNN_BENCH_TRACE=1 prints the batch times) - inside a 20-step timed region it reads as 4.9 instead of
        3.1 ms per step; this loop runs past it and past any clock ramp."""
        prev = None
        trace = os.environ.get("TFGNN_BENCH_TRACE") == "1"
        for i in range(max_batches if not trace else 150):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(batch):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if trace:
                print(f"settle batch {i}: {1000 * dt / batch:.3f} ms/step", file=sys.stderr)
            elif i + 1 >= min_batches and prev is not None and abs(dt - prev) <= tol * prev:
                return
            prev = dt

    def timed(warmup, steps):
        for _ in range(warmup):
            step()
        settle()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        for g_ in pending:  # the batch prepared for the step after the last one
            g_.wait()
            g_.close()
        pending.clear()
        return parallel.reduce_max(dt, dist, dev)

    ops.set_gemm_mode(args.gemm_mode)
    elapsed = timed(args.warmup, args.steps)
    # final metric reduction: all-gather of the per-rank edge counts (north_star: the only collective)
    total_edges_per_step = float(parallel.all_gather_scalars([float(E)], dist, dev)[:, 0].sum())
    ms_per_step = 1000.0 * elapsed / args.steps
    value = total_edges_per_step * args.steps / elapsed

    result = {
        "metric": baseline_metric(),
        "value": value,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic R-MAT (0.57,0.19,0.19,0.05), N(0,1) features, Glorot weights",
        "config": {
            "workload": f"{args.workload}: V={V} E={E} edge_types={L} D={D} H={H} layers={NL} RGCN (PPI_RGCN.json hypers), "
            f"step = edge bucketing{' (hoisted)' if args.reuse_graph else (' (on the compute stream)' if args.serial_bucketing else ' of the next batch (2nd stream, overlapped)')} + GNN fwd + full bwd, one batch per GPU",
            "per_layer_traversal_rate_edges_per_s": value * NL,
            "gemm_mode": GEMM_MODE_NOTES[args.gemm_mode],
            "collectives_per_step": ("1 bucketed all-reduce of the weight gradients (--allreduce-grads)"
                                     if (args.allreduce_grads and world > 1) else "none (graph-sharded batches)"),
        },
    }
    if not args.no_alt_mode:
        # the same job with the GEMMs in the other evaluation mode, for reference (not the headline value)
        alt = "fp32" if args.gemm_mode != "fp32" else "bf16x3"
        ops.set_gemm_mode(alt)
        alt_steps = max(3, args.steps // 2)
        alt_elapsed = timed(2, alt_steps)
        ops.set_gemm_mode(args.gemm_mode)
        result["config"]["alt_gemm_mode"] = {
            "gemm_mode": GEMM_MODE_NOTES[alt], "steps": alt_steps, "ms_per_step": 1000.0 * alt_elapsed / alt_steps,
            "value": total_edges_per_step * alt_steps / alt_elapsed,
        }

    if rank == 0 and not args.no_roofline:
        # ---- dominant kernels, measured live on the launch stream -------------------------------
        g = graph if graph is not None else ops.Graph(adj_dev, V)
        rs = g.array(ops.G_INVDEG_BY_DST)
        Hx = torch.randn((V, H), device=dev)
        A = torch.empty((V * L, H), device=dev)
        W = torch.randn((L * H, H), device=dev) * 0.05
        out = torch.empty((V, H), device=dev)
        ms_gather = time_kernel(lambda: ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, Hx, row_scale=rs, out=A))
        if args.gemm_mode == "fp32":
            ms_gemm = time_kernel(lambda: ops.gemm(A.view(V, L * H), W, act="relu", out=out))
        elif args.gemm_mode == "f16x2":  # operands as the layer produces them (gnn_edge_mlp.py:_forward_A)
            ms_gather = time_kernel(lambda: ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, Hx, row_scale=rs, rows_per_operand_row=L))
            A_sp = ops.graph_gather_sp(g, ops.VIEW_BY_DST_TYPED, Hx, row_scale=rs, rows_per_operand_row=L)
            Wt_sp = ops.sp_split_cols(W)
            ms_gemm = time_kernel(lambda: ops.sp_gemm_nt(A_sp, Wt_sp, act="relu", out=out))
        else:  # the layers hand the split-operand kernel W^T (gnn_edge_mlp.py:_forward_A)
            Wt = W.t().contiguous()
            ms_gemm = time_kernel(lambda: ops.gemm(A.view(V, L * H), Wt, trans_b=True, act="relu", out=out))
        ms_graph = time_kernel(lambda: ops.Graph(adj_dev, V).close(), iters=5, warmup=1)
        # algorithmic bytes of one gather launch: one fp32 source row + one int32 col per
        # edge, the row pointer and scale once, one output row per (node, type) bucket
        gather_bytes = E * (4 * H + 4) + (V * L + 1) * 4 + V * L * 4 + V * L * H * 4
        gemm_flops = 2.0 * V * (L * H) * H
        gather_gbs = gather_bytes / (ms_gather * 1e-3) / 1e9
        gemm_tflops = gemm_flops / (ms_gemm * 1e-3) / 1e12
        # per step: 2 gathers and 3 GEMM-equivalents per layer
        share_gather = 2 * NL * ms_gather / ms_per_step
        share_gemm = 3 * NL * ms_gemm / ms_per_step
        roof_gather = {
            "kernel": "csr_gather_reduce_kernel (aggregate source rows per (node, type) bucket)",
            "bound": "hbm", "achieved": gather_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gather_gbs / HBM_PEAK_GBS, "traffic": None, "ms_per_launch": ms_gather,
            "algorithmic_bytes_per_launch": gather_bytes, "share_of_step": share_gather,
        }
        if args.gemm_mode == "fp32":
            gemm_kernel = "gemm_mfma_kernel<4,2,1,5> 128x320 tile ([V, L*H] x [L*H, H] + relu, v_mfma_f32_32x32x2_f32)"
            gemm_peak, executed = MFMA_FP32_PEAK_TFLOPS, gemm_tflops
        elif args.gemm_mode == "f16x2":
            # 3 exact fp16 piece products per fp32 product: the matrix cores execute 3x the algorithmic flops, priced against
            # the dense fp16 MFMA peak (= the bf16 one)
            gemm_kernel = ("gemm_sp_nt_kernel<5> 128x320 tile, LDS-DMA ring, pinned MFMA stream ([V, L*H] x [H, L*H]^T + relu, "
                           "3 x v_mfma_f32_32x32x16_f16 per fp32 k16 step on SP16 operands written by the gather)")
            gemm_peak, executed = MFMA_BF16_PEAK_TFLOPS, gemm_tflops * 3
        else:
            # every fp32 product is evaluated as 6 (9) exact bf16 piece products: the matrix cores execute
            # 6x (9x) the algorithmic flops, priced against the dense bf16 MFMA peak
            nprod = 6 if args.gemm_mode == "bf16x3" else 9
            gemm_kernel = (f"gemm_x3s_kernel 128x320 tile, 4 multiplying + 4 staging waves ([V, L*H] x [H, L*H]^T + relu, {nprod} x v_mfma_f32_32x32x16_bf16 "
                           "per fp32 k16 step on exactly split operands)")
            gemm_peak, executed = MFMA_BF16_PEAK_TFLOPS, gemm_tflops * nprod
        roof_gemm = {
            "kernel": gemm_kernel,
            "bound": "mfma", "achieved": executed, "peak": gemm_peak, "unit": "TFLOP/s",
            "frac": executed / gemm_peak, "traffic": None, "ms_per_launch": ms_gemm,
            "algorithmic_flops_per_launch": gemm_flops, "algorithmic_tflops": gemm_tflops, "share_of_step": share_gemm,
        }
        if args.gemm_mode != "fp32":
            # tools/mfma_dep_probe.hip on this part: back-to-back v_mfma_f32_32x32x16_bf16 sustains 2.47 PFLOP/s on smooth
            # operands and 1.87 PFLOP/s on operands with random significands (the clock drops to 1.78 GHz)
            roof_gemm["peak_measured_random_operands"] = 1870.0
            roof_gemm["frac_of_measured_random_operand_peak"] = executed / 1870.0
        # HBM traffic per launch from the rocprofv3 PMC passes (tools/pmc_probe.py, tools/parse_pmc.py;
        # FETCH_SIZE corrected x2 as calibrated on gfx950), committed under profiles/
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r01_pmc_traffic_{args.workload}.json")
        if os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            roof_gather["traffic"] = pmc["gather"]["hbm_bytes_per_launch"]
            key = "gemm" if args.gemm_mode == "fp32" else "gemm_bf16x3"
            if key in pmc:
                roof_gemm["traffic"] = pmc[key]["hbm_bytes_per_launch"]
            roof_gather["traffic_source"] = roof_gemm["traffic_source"] = os.path.relpath(pmc_path, os.path.dirname(os.path.abspath(__file__)))
        if share_gemm >= share_gather:
            result["roofline"], result["roofline_secondary"] = roof_gemm, roof_gather
        else:
            result["roofline"], result["roofline_secondary"] = roof_gather, roof_gemm
        result["config"]["ms_edge_bucketing"] = ms_graph
        if graph is None:
            g.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl, feats, adjs, params)

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
