"""Timings of the other BASELINE.json configs (parity-test cases, not bench lines): one message passing
layer forward and forward+backward at full size, HIP events on the launch stream.
  python tools/bench_configs.py [cfg2 cfg3 cfg4_ggnn cfg4_edge_mlp cfg5]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf2_gnn_amd import ops  # noqa: E402
from tf2_gnn_amd.data import make_synthetic_batch, rmat_edges  # noqa: E402
import tf2_gnn_amd.layers.message_passing as mp  # noqa: E402
from tests.test_gpu_full_size import _qm9_shaped_batch  # noqa: E402


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def build(cls_name, over, D, L):
    cls = getattr(mp, cls_name)
    p = cls.get_default_hyperparameters()
    p.update(over)
    layer = cls(p)
    layer.build(mp.MessagePassingInput((None, D), tuple((None, 2) for _ in range(L))))
    return layer


dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["cfg2", "cfg3", "cfg4_ggnn", "cfg4_edge_mlp", "cfg5"]
for name in which:
    if name in ("cfg2", "cfg3"):
        V, E, L = 30000, 900000, 4
        H = 320 if name == "cfg2" else 256
        feats, adjs = make_synthetic_batch(V, E, L, H, seed=1)
        layer = build("RGCN", {"hidden_dim": H}, H, L) if name == "cfg2" else build(
            "RGAT", {"hidden_dim": H, "num_heads": 8, "message_activation_function": "tanh"}, H, L)
    elif name.startswith("cfg4"):
        H = 128
        feats, adjs, _, _ = _qm9_shaped_batch(128000, seed=1, D=H)
        V, L = feats.shape[0], len(adjs)
        layer = build("GGNN", {"hidden_dim": H, "normalize_by_num_incoming": False}, H, L) if name == "cfg4_ggnn" else build(
            "GNN_Edge_MLP", {"hidden_dim": H}, H, L)
    else:
        V, E, L, H = 170000, 1200000, 40, 512
        rng = np.random.default_rng(1)
        edges = rmat_edges(V, E, rng)
        pz = 1.0 / np.arange(1, L + 1)
        types = rng.choice(L, size=E, p=pz / pz.sum())
        adjs = [np.ascontiguousarray(edges[types == l]) for l in range(L)]
        feats = rng.standard_normal((V, H), dtype=np.float32)
        layer = build("RGIN", {"hidden_dim": H}, H, L)
    E = sum(a.shape[0] for a in adjs)
    adj_dev = tuple(torch.from_numpy(a).to(dev) for a in adjs)
    X = torch.from_numpy(feats).to(dev)
    g = ops.Graph(adj_dev, V)
    ms_graph = timeit(lambda: ops.Graph(adj_dev, V).close(), iters=5, warmup=1)
    inp = mp.MessagePassingInput(X, g)
    ms_fwd = timeit(lambda: layer(inp, training=False))
    out = layer(inp, training=True)
    dOut = torch.ones_like(out)

    def fb():
        layer(inp, training=True)
        layer.backward(dOut)

    try:
        ms_fb = timeit(fb)
    except NotImplementedError as e:
        ms_fb = float("nan")
    print(f"{name:14s} {type(layer).__name__:13s} V={V} E={E} L={L} H={H}: bucketing {ms_graph:7.3f} ms | layer fwd {ms_fwd:8.3f} ms "
          f"({E/ms_fwd/1e6:7.2f} G edges/s) | fwd+bwd {ms_fb:8.3f} ms ({E/ms_fb/1e6:7.2f} G edges/s)", flush=True)
    g.close()
    del layer, X, out, dOut, g
    torch.cuda.empty_cache()
