"""Task heads behind the path against the CPU oracle: metric kernels, and the whole
batch -> loss -> gradients computation of NodeMulticlassTask / QM9RegressionTask / GraphRegressionTask.
Tolerance: 1e-5 scaled (fp32 north star); counts are exact.  PARITY UNPINNED (oracle/tf2gnn_oracle.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import tf2gnn_oracle as orc
from tests.helpers import assert_close, to_dev
from tests.test_gpu_layers import _gnn_oracle_weights, _pool_weights, _to64

# every test of this module runs in the three GEMM modes (conftest.py: gemm_modes)
pytestmark = [pytest.mark.gpu, pytest.mark.gemm_modes, pytest.mark.usefixtures("gemm_mode")]


@pytest.mark.parametrize("V,C", [(1, 1), (257, 121), (5000, 40)])
def test_sigmoid_ce_metrics_kernel(dev, V, C):
    """node_multiclass_task.py:10-23,62-70: loss, micro-F1 (exact counts, round-half-even at sigmoid = 0.5), gradient."""
    from tf2_gnn_amd import ops

    g = torch.Generator().manual_seed(V + C)
    logits = torch.randn((V, C), generator=g) * 3
    logits[::7, ::3] = 0.0  # sigmoid = 0.5 exactly -> rounds to 0
    logits.view(-1)[:: 11] *= 30  # saturated entries
    labels = (torch.rand((V, C), generator=g) < 0.3).float()
    wide = torch.zeros((V, C + 5))
    wide[:, :C] = logits
    metrics, counts, grad = ops.sigmoid_ce_metrics(wide.to(dev)[:, :C], labels.to(dev))  # strided logits
    ref_loss = torch.mean(torch.sum(orc.sigmoid_cross_entropy_with_logits(logits.double(), labels.double()), dim=-1))
    f1, (tp, fp, fn) = orc.micro_f1(logits, labels)
    assert counts.cpu().tolist() == [tp, fp, fn]
    assert abs(float(metrics[0]) - float(ref_loss)) <= 2e-6 * max(1.0, abs(float(ref_loss)))
    if math.isnan(f1):
        assert math.isnan(float(metrics[1]))
    else:
        assert abs(float(metrics[1]) - f1) <= 1e-6
    x64 = logits.double().requires_grad_(True)
    loss64 = torch.mean(torch.sum(orc.sigmoid_cross_entropy_with_logits(x64, labels.double()), dim=-1))
    (gx,) = torch.autograd.grad(loss64, x64)
    assert float((grad.cpu().double() - gx).abs().max()) <= 1e-6 / V + 1e-7 * float(gx.abs().max())
    # reproducible: fixed-order reduction
    again, _, _ = ops.sigmoid_ce_metrics(wide.to(dev)[:, :C], labels.to(dev), need_grad=False)
    assert torch.equal(again[:1], metrics[:1])  # (the F1 may be nan)


def test_sigmoid_ce_micro_f1_undefined_is_nan(dev):
    """No positive prediction and no positive label: 0/0 -> nan, as the reference's float64 division."""
    from tf2_gnn_amd import ops

    logits = torch.full((4, 3), -2.0)
    labels = torch.zeros((4, 3))
    metrics, counts, _ = ops.sigmoid_ce_metrics(logits.to(dev), labels.to(dev))
    assert counts.cpu().tolist() == [0, 0, 0]
    assert math.isnan(float(metrics[1]))
    with pytest.raises(Exception):
        ops.sigmoid_ce_metrics(torch.zeros((0, 3), device=dev), torch.zeros((0, 3), device=dev))


@pytest.mark.parametrize("G", [1, 37, 200000])
def test_regression_metrics_kernel(dev, G):
    from tf2_gnn_amd import ops

    g = torch.Generator().manual_seed(G)
    pred, target = torch.randn(G, generator=g) * 4, torch.randn(G, generator=g)
    metrics, grad = ops.regression_metrics(pred.to(dev), target.to(dev))
    mse, mae = orc.regression_metrics(target.double(), pred.double())
    assert abs(float(metrics[0]) - float(mse)) <= 2e-6 * float(mse)
    assert abs(float(metrics[1]) - float(mae)) <= 2e-6 * float(mae)
    assert_close(grad.cpu() * G, (2 * (pred - target)), tol=1e-6, what="d mse / d pred")


# ---------------------------------------------------------------------------------------------------------------
def _batch(sizes, L, D0, seed):
    rng = np.random.default_rng(seed)
    adjs = []
    for _ in range(L):
        parts, base = [], 0
        for n in sizes:
            parts.append(rng.integers(0, n, size=(3 * n, 2)) + base)
            base += n
        adjs.append(np.concatenate(parts).astype(np.int32))
    n2g = np.repeat(np.arange(len(sizes)), sizes).astype(np.int32)
    X = torch.from_numpy(rng.standard_normal((sum(sizes), D0)).astype(np.float32))
    return X, adjs, n2g


def _features(X, adjs, n2g, G, dev):
    f = {"node_features": X.to(dev), "node_to_graph_map": torch.from_numpy(n2g).to(dev), "num_graphs_in_batch": G}
    for i, a in enumerate(to_dev(adjs, dev)):
        f[f"adjacency_list_{i}"] = a
    return f


def _leaves(w64):
    leaves = []

    def visit(obj):
        if isinstance(obj, torch.Tensor):
            obj.requires_grad_(True)
            leaves.append(obj)
        elif isinstance(obj, dict):
            for k in obj:
                visit(obj[k])
        elif isinstance(obj, (list, tuple)):
            for v in obj:
                if v is not None:
                    visit(v)

    visit(w64)
    return leaves


def _gnn_pairs(gnn, w64, L):
    pairs = [(gnn._initial_projection_layer, w64["initial_projection"])]
    for i, mp in enumerate(gnn._mp_layers):
        for l in range(L):
            for j, v in enumerate(mp._edge_type_mlps.vars[l]):
                pairs.append((v, w64["mp"][i]["edge_mlps"][l][j]))
        if str(i) in gnn._dense_layers:
            pairs.append((gnn._dense_layers[str(i)], w64["dense"][i]))
    return pairs


def _mlp_pairs(mlp, kernels64, biases64):
    pairs = list(zip(mlp.kernels, kernels64))
    pairs += [(b, t) for b, t in zip(mlp.biases, biases64) if b is not None]
    return pairs


def _check_grads(model, pairs, loss64, what):
    leaves = [t for _, t in pairs]
    grads = torch.autograd.grad(loss64, leaves, allow_unused=True)
    assert len(pairs) == len(model.trainable_variables), (len(pairs), len(model.trainable_variables))
    for (v, _), r in zip(pairs, grads):
        assert v.grad is not None, f"{what}: no gradient for {v.name}"
        r = torch.zeros_like(v.grad.cpu().double()) if r is None else r
        scale = max(1e-3, float(r.abs().max()))
        assert_close(v.grad.cpu() / scale, (r / scale).float(), tol=5e-5, what=f"{what} d {v.name}")


def _gnn_params(model_cls, mp_style, H, layers, extra=()):
    params = model_cls.get_default_hyperparameters(mp_style)
    params.update({"gnn_hidden_dim": H, "gnn_num_layers": layers, "gnn_global_exchange_every_num_layers": 10000,
                   "gnn_dense_every_num_layers": 2, "gnn_residual_every_num_layers": 2})
    params.update(dict(extra))
    return params


def _oracle_gnn(params, w, X, adjs):
    gp = {k[4:]: v for k, v in params.items() if k.startswith("gnn_")}
    return orc.gnn_internal_call(gp, w, X, [torch.from_numpy(a) for a in adjs])


def test_node_multiclass_task_loss_and_gradients(dev):
    """models/node_multiclass_task.py + graph_task_model.py:158-183: batch -> logits -> loss / F1 -> every gradient."""
    from tf2_gnn_amd.tasks import NodeMulticlassTask

    sizes, L, D0, H, C = [30, 45, 12], 3, 10, 16, 7
    X, adjs, n2g = _batch(sizes, L, D0, seed=4)
    params = _gnn_params(NodeMulticlassTask, "rgcn", H, 3)
    model = NodeMulticlassTask(params, num_edge_types=L, num_node_target_labels=C)
    feats = _features(X, adjs, n2g, len(sizes), dev)
    model.build({"node_features": (None, D0)})
    model._bias.value.copy_(torch.randn(C, generator=torch.Generator().manual_seed(1)))
    labels = (torch.rand((X.shape[0], C), generator=torch.Generator().manual_seed(2)) < 0.4).float()
    out = model(feats, training=False)
    m = model.compute_task_metrics(feats, out, {"node_labels": labels.to(dev)})
    model.backward()

    w = _gnn_oracle_weights(model._gnn)
    head = {"kernel": model._kernel.value.cpu().clone(), "bias": model._bias.value.cpu().clone()}
    h, _ = _oracle_gnn(params, w, X, adjs)
    logits, loss = orc.node_multiclass_task(h, head["kernel"], head["bias"], labels)
    assert_close(out[0].cpu(), logits, tol=1e-5, what="per-node logits")
    assert abs(float(m["loss"]) - float(loss)) <= 2e-5 * max(1.0, float(loss))
    f1, counts = orc.micro_f1(out[0].cpu(), labels)  # counts from the HIP logits: a logit within 1e-6 of 0 may flip
    assert m["f1_counts"].cpu().tolist() == list(counts)
    assert abs(float(m["f1_score"]) - f1) <= 1e-6
    assert model.compute_epoch_metrics([m])[1].startswith("Avg MicroF1")

    w64, head64 = _to64(w), _to64(head)
    _leaves(w64), _leaves(head64)
    h64, _ = _oracle_gnn(params, w64, X.double(), adjs)
    _, loss64 = orc.node_multiclass_task(h64, head64["kernel"], head64["bias"], labels.double())
    pairs = _gnn_pairs(model._gnn, w64, L) + [(model._kernel, head64["kernel"]), (model._bias, head64["bias"])]
    _check_grads(model, pairs, loss64, "NodeMulticlassTask")


def test_qm9_regression_task_loss_and_gradients(dev):
    """models/qm9_regression.py:83-130: gated per-graph sum, mse / mae, every gradient."""
    from tf2_gnn_amd.tasks import QM9RegressionTask

    sizes, L, D0, H = [9, 5, 13, 7, 1, 11], 4, 12, 16
    X, adjs, n2g = _batch(sizes, L, D0, seed=6)
    G = len(sizes)
    params = _gnn_params(QM9RegressionTask, "ggnn", H, 2)
    model = QM9RegressionTask(params, num_edge_types=L)
    feats = _features(X, adjs, n2g, G, dev)
    model.build({"node_features": (None, D0)})
    for v in model._task_variables():
        if v.name.endswith("bias"):
            v.value.fill_(0.3)
    target = torch.randn(G, generator=torch.Generator().manual_seed(3))
    out = model(feats, training=False)
    m = model.compute_task_metrics(feats, out, {"target_value": target.to(dev)})
    model.backward()

    w = _gnn_oracle_weights(model._gnn)
    gate = (model._regression_gate.kernels[0].value.cpu().clone(), model._regression_gate.biases[0].value.cpu().clone())
    tr = (model._regression_transform.kernels[0].value.cpu().clone(), model._regression_transform.biases[0].value.cpu().clone())
    ids = torch.from_numpy(n2g)
    h, _ = _oracle_gnn(params, w, X, adjs)
    ref = orc.qm9_regression_output(X, h, gate, tr, ids, G)
    assert_close(out.cpu(), ref, tol=1e-5, what="per-graph output")
    mse, mae = orc.regression_metrics(target, ref)
    assert abs(float(m["loss"]) - float(mse)) <= 5e-5 * max(1.0, float(mse))
    assert abs(float(m["batch_absolute_error"]) - float(mae) * G) <= 5e-5 * max(1.0, float(mae) * G)
    assert m["num_graphs"] == float(G)
    assert "MAE" in model.compute_epoch_metrics([m])[1]

    w64, gate64, tr64 = _to64(w), _to64(gate), _to64(tr)
    for obj in (w64, gate64, tr64):
        _leaves(obj)
    h64, _ = _oracle_gnn(params, w64, X.double(), adjs)
    ref64 = orc.qm9_regression_output(X.double(), h64, gate64, tr64, ids, G)
    mse64, _ = orc.regression_metrics(target.double(), ref64)
    # GGNN: edge kernels + GRU weights per layer
    pairs = [(model._gnn._initial_projection_layer, w64["initial_projection"])]
    for i, mp in enumerate(model._gnn._mp_layers):
        for l in range(L):
            for j, v in enumerate(mp._edge_type_mlps.vars[l]):
                pairs.append((v, w64["mp"][i]["edge_mlps"][l][j]))
        ru = mp._recurrent_unit
        pairs += [(ru["kernel"], w64["mp"][i]["gru_kernel"]), (ru["recurrent_kernel"], w64["mp"][i]["gru_recurrent_kernel"]),
                  (ru["bias"], w64["mp"][i]["gru_bias"])]
        if str(i) in model._gnn._dense_layers:
            pairs.append((model._gnn._dense_layers[str(i)], w64["dense"][i]))
    pairs += [(model._regression_gate.kernels[0], gate64[0]), (model._regression_gate.biases[0], gate64[1]),
              (model._regression_transform.kernels[0], tr64[0]), (model._regression_transform.biases[0], tr64[1])]
    by_name = {id(v): k for k, (v, _) in enumerate(pairs)}
    order = [pairs[by_name[id(v)]] for v in model.trainable_variables]
    _check_grads(model, order, mse64, "QM9RegressionTask")


@pytest.mark.parametrize("intermediate", [True, False])
def test_graph_regression_task_loss_and_gradients(dev, intermediate):
    """models/graph_regression_task.py:104-166; with use_intermediate_gnn_results the gradient enters the stack at
    every layer's output (GNN.backward grad_all_representations)."""
    from tf2_gnn_amd.tasks import GraphRegressionTask

    sizes, L, D0, H = [14, 3, 21, 8], 2, 6, 16
    X, adjs, n2g = _batch(sizes, L, D0, seed=8)
    G = len(sizes)
    params = _gnn_params(GraphRegressionTask, "rgcn", H, 3, extra={
        "use_intermediate_gnn_results": intermediate, "graph_aggregation_output_size": 8, "graph_aggregation_num_heads": 2,
        "graph_aggregation_layers": [12], "regression_mlp_layers": [10, 6],
        "gnn_dense_every_num_layers": 10000 if intermediate else 2})  # no Dense: cross-layer fusions must switch off
    model = GraphRegressionTask(params, num_edge_types=L)
    feats = _features(X, adjs, n2g, G, dev)
    model.build({"node_features": (None, D0)})
    for b in model._regression_mlp.biases:
        b.value.copy_(torch.randn(b.shape, generator=torch.Generator().manual_seed(5)) * 0.2)
    target = torch.randn(G, generator=torch.Generator().manual_seed(7))
    out = model(feats, training=False)
    m = model.compute_task_metrics(feats, out, {"target_value": target.to(dev)})
    model.backward()

    def mlp_w(mlp):
        return [k.value.cpu().clone() for k in mlp.kernels], [b.value.cpu().clone() for b in mlp.biases]

    w = _gnn_oracle_weights(model._gnn)
    tw = {"avg": _pool_weights(model._weighted_avg_of_nodes_to_graph_repr),
          "sum": _pool_weights(model._weighted_sum_of_nodes_to_graph_repr), "regression": mlp_w(model._regression_mlp)}
    ids = torch.from_numpy(n2g)

    def oracle(w_, tw_, X_):
        final, all_reprs = _oracle_gnn(params, w_, X_, adjs)
        return orc.graph_regression_output(params, tw_, X_, (final, all_reprs) if intermediate else final, ids, G)

    ref = oracle(w, tw, X)
    assert_close(out.cpu(), ref, tol=1e-5, what="per-graph output")
    mse, _ = orc.regression_metrics(target, ref)
    assert abs(float(m["loss"]) - float(mse)) <= 5e-5 * max(1.0, float(mse))

    w64, tw64 = _to64(w), _to64(tw)
    _leaves(w64), _leaves(tw64)
    mse64, _ = orc.regression_metrics(target.double(), oracle(w64, tw64, X.double()))
    pairs = _gnn_pairs(model._gnn, w64, L)
    for key, pool in (("avg", model._weighted_avg_of_nodes_to_graph_repr), ("sum", model._weighted_sum_of_nodes_to_graph_repr)):
        pairs += list(zip(pool._scoring_mlp.kernels, tw64[key]["scoring"][0]))
        pairs += list(zip(pool._transformation_mlp.kernels, tw64[key]["transformation"][0]))
    ks, bs = tw64["regression"]
    for k, b, vk, vb in zip(ks, bs, model._regression_mlp.kernels, model._regression_mlp.biases):
        pairs += [(vk, k), (vb, b)]
    by_var = {id(v): (v, t) for v, t in pairs}
    order = [by_var[id(v)] for v in model.trainable_variables]
    _check_grads(model, order, mse64, f"GraphRegressionTask intermediate={intermediate}")

