"""CPU restatement (torch-CPU, fp32 or fp64) of the tf2-gnn message-passing hot path.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  Every function cites the reference
file:line it follows (paths relative to /root/reference).  The op sequence is kept *literal*
(materialised per-edge gathers, per-edge matmul, concat over edge types, unsorted segment op) so
that (a) the floating-point summation order is the reference's, and (b) timing it gives the
reference's CPU cost structure (bench.py ``cpu_baseline``, kind "port").

[ext] marks semantics that live in TensorFlow / Keras / dpu_utils (not under /root/reference) and
are restated from their published behaviour.

Weights are passed explicitly as plain dicts / lists of torch tensors so that the HIP layers and the
oracle can be fed bit-identical parameters.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

SMALL_NUMBER = 1e-7  # tf2_gnn/utils/constants.py:2


# --------------------------------------------------------------------------------------------
# tf.math.unsorted_segment_* [ext]  (used through tf2_gnn/utils/param_helpers.py:7-18)
# --------------------------------------------------------------------------------------------
def unsorted_segment_sum(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int):
    out = torch.zeros((num_segments,) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
    if data.shape[0]:
        out.index_add_(0, segment_ids.long(), data)
    return out


def _segment_counts(segment_ids: torch.Tensor, num_segments: int, dtype):
    cnt = torch.zeros(num_segments, dtype=dtype, device=segment_ids.device)
    if segment_ids.shape[0]:
        cnt.index_add_(0, segment_ids.long(), torch.ones(segment_ids.shape[0], dtype=dtype, device=segment_ids.device))
    return cnt


def unsorted_segment_mean(data, segment_ids, num_segments):
    # [ext] tf.math.unsorted_segment_mean: sum / max(N, 1)
    s = unsorted_segment_sum(data, segment_ids, num_segments)
    n = _segment_counts(segment_ids, num_segments, data.dtype).clamp(min=1.0)
    return s / n.reshape((-1,) + (1,) * (data.dim() - 1))


def unsorted_segment_sqrt_n(data, segment_ids, num_segments):
    # [ext] tf.math.unsorted_segment_sqrt_n: sum / sqrt(max(N, 1))
    s = unsorted_segment_sum(data, segment_ids, num_segments)
    n = _segment_counts(segment_ids, num_segments, data.dtype).clamp(min=1.0)
    return s / torch.sqrt(n).reshape((-1,) + (1,) * (data.dim() - 1))


def unsorted_segment_max(data, segment_ids, num_segments):
    # [ext] tf.math.unsorted_segment_max: empty segments hold the lowest finite value of the dtype.
    lowest = torch.finfo(data.dtype).min
    out = torch.full((num_segments,) + tuple(data.shape[1:]), lowest, dtype=data.dtype, device=data.device)
    if data.shape[0]:
        idx = segment_ids.long().reshape((-1,) + (1,) * (data.dim() - 1)).expand_as(data)
        out = out.scatter_reduce(0, idx, data, reduce="amax", include_self=True)
    return out


def get_aggregation_function(name: str):
    """tf2_gnn/utils/param_helpers.py:7-18"""
    fns = {
        "sum": unsorted_segment_sum,
        "max": unsorted_segment_max,
        "mean": unsorted_segment_mean,
        "sqrt_n": unsorted_segment_sqrt_n,
    }
    fn = fns.get(name)
    if fn is None:
        raise ValueError(f"Unknown aggregation function: {name}")
    return fn


# --------------------------------------------------------------------------------------------
# activations: tf2_gnn/utils/param_helpers.py:21-39, tf2_gnn/utils/activation.py:7-14
# --------------------------------------------------------------------------------------------
def gelu(x: torch.Tensor):
    """tf2_gnn/utils/activation.py:7-14 (tanh approximation)"""
    cdf = 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * torch.pow(x, 3))))
    return x * cdf


def get_activation_function(name: Optional[str]):
    if name is None:
        return None
    name = name.lower()
    fns = {
        "linear": None,
        "tanh": torch.tanh,
        "relu": torch.relu,
        "leaky_relu": lambda x: torch.nn.functional.leaky_relu(x, 0.2),  # [ext] tf.nn.leaky_relu alpha=0.2
        "elu": torch.nn.functional.elu,
        "selu": torch.nn.functional.selu,
        "gelu": gelu,
    }
    fn = fns.get(name)
    if fn is None:
        # NB: the reference raises for "linear" too (param_helpers.py:28,36-38).
        raise ValueError(f"Unknown activation function: {name}")
    return fn


def get_activation_function_by_name(name: Optional[str]):
    """[ext] dpu_utils.tf2utils.get_activation_function_by_name (used by
    tf2_gnn/layers/nodes_to_graph_representation.py:122,134,145): like the above but "linear"/None
    mean identity and "sigmoid" exists."""
    if name is None:
        return None
    name = name.lower()
    if name == "linear":
        return None
    if name == "sigmoid":
        return torch.sigmoid
    return get_activation_function(name)


# --------------------------------------------------------------------------------------------
# [ext] dpu_utils.tf2utils.MLP  (call sites gnn_edge_mlp.py:76-80,100; rgin.py:81-85,104;
#       nodes_to_graph_representation.py:130-148)
# --------------------------------------------------------------------------------------------
def mlp_hidden_sizes(out_size: int, hidden_layers) -> List[int]:
    if isinstance(hidden_layers, int):
        return [out_size] * hidden_layers
    return list(hidden_layers)


def mlp_forward(
    x: torch.Tensor,
    kernels: Sequence[torch.Tensor],
    biases: Optional[Sequence[Optional[torch.Tensor]]] = None,
    activation=torch.relu,
    dropout_masks: Optional[Sequence[Optional[torch.Tensor]]] = None,
):
    """Stack of Keras Dense layers: hidden layers with ``activation`` then a final linear Dense.
    Training-mode dropout ([ext] tf.nn.dropout on the inputs of the hidden layers - see the note on which layers in
    tf2_gnn_amd/layers/nodes_to_graph_representation.py MLP) is restated through ``dropout_masks``: entry i (already scaled
    by 1/(1-rate), or None) multiplies the input of Dense layer i, so that a test can hand over the masks the HIP path
    drew (RNG streams cannot match)."""
    h = x
    n = len(kernels)
    for i, k in enumerate(kernels):
        if dropout_masks is not None and dropout_masks[i] is not None:
            h = h * dropout_masks[i]
        h = h @ k
        if biases is not None and biases[i] is not None:
            h = h + biases[i]
        if i < n - 1 and activation is not None:
            h = activation(h)
    return h


# --------------------------------------------------------------------------------------------
# [ext] dpu_utils.tf2utils.unsorted_segment_{log_}softmax (rgat.py:147-151,
#       nodes_to_graph_representation.py:180-184)
# --------------------------------------------------------------------------------------------
def unsorted_segment_log_softmax(logits, segment_ids, num_segments):
    mx = unsorted_segment_max(logits, segment_ids, num_segments)
    rec = logits - mx[segment_ids.long()]
    s = unsorted_segment_sum(torch.exp(rec), segment_ids, num_segments)
    return rec - torch.log(s)[segment_ids.long()]


def unsorted_segment_softmax(logits, segment_ids, num_segments):
    mx = unsorted_segment_max(logits, segment_ids, num_segments)
    rec = logits - mx[segment_ids.long()]
    e = torch.exp(rec)
    s = unsorted_segment_sum(e, segment_ids, num_segments)
    return e / (s[segment_ids.long()] + SMALL_NUMBER)


# --------------------------------------------------------------------------------------------
# message passing: tf2_gnn/layers/message_passing/message_passing.py
# --------------------------------------------------------------------------------------------
def calculate_type_to_num_incoming_edges(node_embeddings, adjacency_lists):
    """message_passing.py:230-263 -> float tensor [L, V]"""
    V = node_embeddings.shape[0]
    rows = []
    for adj in adjacency_lists:
        targets = adj[:, 1].long()
        cnt = torch.zeros(V, dtype=node_embeddings.dtype, device=node_embeddings.device)
        if targets.shape[0]:
            cnt.index_add_(0, targets, torch.ones(targets.shape[0], dtype=node_embeddings.dtype, device=node_embeddings.device))
        rows.append(cnt)
    return torch.stack(rows) if rows else torch.zeros((0, V), dtype=node_embeddings.dtype, device=node_embeddings.device)


def _edge_mlp_message(params, mlp_kernels_l, src_states, tgt_states, num_incoming):
    """GNN_Edge_MLP._message_function, gnn_edge_mlp.py:84-107."""
    if params["use_target_state_as_input"]:
        inp = torch.cat([src_states, tgt_states], dim=1)
    else:
        inp = src_states
    msg = mlp_forward(inp, mlp_kernels_l)
    if params["normalize_by_num_incoming"]:
        msg = (1.0 / (num_incoming + SMALL_NUMBER)).unsqueeze(-1) * msg
    return msg


def _rgat_message(params, kernel_l, attn_l, src_states, tgt_states):
    """RGAT._message_function, rgat.py:91-123."""
    K = params["num_heads"]
    H = params["hidden_dim"]
    ys = (src_states @ kernel_l).reshape(-1, K, H // K)
    yt = (tgt_states @ kernel_l).reshape(-1, K, H // K)
    both = torch.cat([ys, yt], dim=-1)  # [E, K, 2H/K]
    scores = torch.nn.functional.leaky_relu(torch.einsum("vki,ki->vk", both, attn_l), 0.2)
    return ys, scores


def message_passing_call(
    kind: str,
    params: Dict[str, Any],
    weights: Dict[str, Any],
    node_embeddings: torch.Tensor,
    adjacency_lists: Sequence[torch.Tensor],
):
    """MessagePassing.call (message_passing.py:95-133) for kind in
    {rgcn, ggnn, rgin, gnn_edge_mlp, gnn_film, rgat, pass_source_states}; eval mode.

    weights:
      edge-MLP family: weights["edge_mlps"][l] = list of kernels ([in,out]) of MLP_l
      rgin:  + weights["aggr_mlp"] = list of kernels or None
      gnn_film: + weights["film_mlps"][l] = list of kernels of the FiLM parameter MLP_l (out 2H; gnn_film.py:70-82)
      ggnn:  + weights["gru_kernel"] [D,3H], ["gru_recurrent_kernel"] [H,3H], ["gru_bias"] [2,3H]
      rgat:  weights["kernels"][l] [D,H], weights["attn"][l] [K, 2H/K]
    """
    kind = kind.lower()
    X = node_embeddings
    V = X.shape[0]
    agg_fn = get_aggregation_function(params.get("aggregation_function", "sum"))
    act_fn = get_activation_function(params.get("message_activation_function", "relu"))
    act_before = params.get("message_activation_before_aggregation", False)

    # _calculate_messages_per_type, message_passing.py:181-218
    cnt = calculate_type_to_num_incoming_edges(X, adjacency_lists)
    messages_per_type = []
    for l, adj in enumerate(adjacency_lists):
        src = adj[:, 0].long()
        tgt = adj[:, 1].long()
        xs = X[src]
        xt = X[tgt]
        c = cnt[l][tgt]
        if kind == "pass_source_states":  # test/layers/test_message_passing.py:11-27
            m = xs
        elif kind == "rgat":
            m = _rgat_message(params, weights["kernels"][l], weights["attn"][l], xs, xt)
        else:
            m = _edge_mlp_message(params, weights["edge_mlps"][l], xs, xt, c)
            if kind == "gnn_film":  # gnn_film.py:84-108: the target state modulates the message feature-wise
                film = mlp_forward(xt, weights["film_mlps"][l])  # [E, 2H]
                H_ = params["hidden_dim"]
                m = film[:, :H_] * m + film[:, H_:]
        messages_per_type.append(m)
    targets = [adj[:, 1] for adj in adjacency_lists]
    message_targets = torch.cat(targets, dim=0) if targets else torch.zeros(0, dtype=torch.int32, device=X.device)

    if kind == "rgat":  # rgat.py:125-163
        K = params["num_heads"]
        msgs = torch.cat([m[0] for m in messages_per_type], dim=0)  # [M,K,H/K]
        scores = torch.cat([m[1] for m in messages_per_type], dim=0)  # [M,K]
        heads = []
        for k in range(K):
            att = torch.exp(unsorted_segment_log_softmax(scores[:, k], message_targets, V))
            heads.append(unsorted_segment_sum(att.unsqueeze(-1) * msgs[:, k, :], message_targets, V))
        return act_fn(torch.cat(heads, dim=-1))

    H = params["hidden_dim"]
    messages = (
        torch.cat(messages_per_type, dim=0) if messages_per_type else torch.zeros((0, H), dtype=X.dtype, device=X.device)
    )

    if kind == "rgin":  # rgin.py:88-106 (ignores message_activation_before_aggregation)
        agg = agg_fn(messages, message_targets, V)
        if weights.get("aggr_mlp") is not None:
            agg = mlp_forward(agg, weights["aggr_mlp"])
        return act_fn(agg)

    if kind == "ggnn":  # ggnn.py:68-89 (no message activation at all)
        agg = agg_fn(messages, message_targets, V)
        return gru_cell(agg, X, weights["gru_kernel"], weights["gru_recurrent_kernel"], weights["gru_bias"])

    # base class, message_passing.py:135-179
    if act_before:
        messages = act_fn(messages)
    agg = agg_fn(messages, message_targets, V)
    if not act_before:
        agg = act_fn(agg)
    return agg


def gru_cell(x, h, kernel, recurrent_kernel, bias):
    """[ext] tf.keras.layers.GRUCell TF2 defaults (reset_after=True, gates z|r|h, tanh / sigmoid),
    as used by ggnn.py:64,84-87.  bias has shape [2, 3H] (input bias, recurrent bias)."""
    H = h.shape[1]
    mx = x @ kernel + bias[0]
    mh = h @ recurrent_kernel + bias[1]
    xz, xr, xh = mx[:, :H], mx[:, H : 2 * H], mx[:, 2 * H :]
    hz, hr, hh = mh[:, :H], mh[:, H : 2 * H], mh[:, 2 * H :]
    z = torch.sigmoid(xz + hz)
    r = torch.sigmoid(xr + hr)
    cand = torch.tanh(xh + r * hh)
    return z * h + (1.0 - z) * cand


def layer_norm(x, gamma, beta, eps=1e-3):
    """[ext] tf.keras.layers.LayerNormalization defaults (axis=-1, epsilon=1e-3), gnn.py:157-161."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


# --------------------------------------------------------------------------------------------
# GNN layer stack: tf2_gnn/layers/gnn.py:276-329 (eval mode; global exchange not restated - it is
# disabled in every PPI/QM9 default_hypers file and out of scope)
# --------------------------------------------------------------------------------------------
def gnn_internal_call(
    params: Dict[str, Any],
    weights: Dict[str, Any],
    node_features: torch.Tensor,
    adjacency_lists: Sequence[torch.Tensor],
    dropout_masks: Optional[Sequence[Optional[torch.Tensor]]] = None,
    node_to_graph_map: Optional[torch.Tensor] = None,
    num_graphs: Optional[int] = None,
    exchange_dropout_masks: Optional[Dict[int, torch.Tensor]] = None,
):
    """weights: {"initial_projection": [Din,H], "mp": [per-layer weights dict],
                 "dense": {layer_idx: [H,H]}, "layernorm": [(gamma, beta)],
                 "global_exchange": {layer_idx: weights of graph_global_exchange} }.
    ``dropout_masks[i]`` (already scaled by 1/(1-rate), or None) multiplies the input of layer i:
    lets tests inject the exact mask the HIP path drew ([ext] tf.nn.dropout scales kept units by
    1/(1-rate))."""
    kind = params["message_calculation_class"]
    act_init = get_activation_function(params["initial_node_representation_activation"])
    act_dense = get_activation_function(params["dense_intermediate_layer_activation"])
    cur = node_features @ weights["initial_projection"]
    if act_init is not None:
        cur = act_init(cur)
    last = cur
    all_reprs = [cur]
    for i in range(params["num_layers"]):
        if dropout_masks is not None and dropout_masks[i] is not None:
            cur = cur * dropout_masks[i]
        if i % params["residual_every_num_layers"] == 0:
            tmp = cur
            if i > 0:
                cur = cur + last
                cur = cur / 2
            last = tmp
        cur = message_passing_call(kind, params, weights["mp"][i], cur, adjacency_lists)
        all_reprs.append(cur)
        if i and i % params["global_exchange_every_num_layers"] == 0:
            ex = weights["global_exchange"][i]
            cur = graph_global_exchange(
                params["global_exchange_mode"], params, ex, cur, node_to_graph_map, num_graphs,
                None if exchange_dropout_masks is None else exchange_dropout_masks.get(i),
            )
        if params["use_inter_layer_layernorm"]:
            g, b = weights["layernorm"][i]
            cur = layer_norm(cur, g, b)
        if i % params["dense_every_num_layers"] == 0:
            cur = cur @ weights["dense"][i]
            if act_dense is not None:
                cur = act_dense(cur)
    return cur, tuple(all_reprs)


# --------------------------------------------------------------------------------------------
# graph-global exchange: tf2_gnn/layers/graph_global_exchange.py
# --------------------------------------------------------------------------------------------
def graph_global_exchange(mode, params, weights, node_embeddings, node_to_graph_map, num_graphs, dropout_mask=None):
    """GraphGlobal{Mean,GRU,MLP}Exchange.call (graph_global_exchange.py:83-183).
    weights: {"pool": weights of the WeightedSumGraphRepresentation(graph_representation_size=H, weighting_fun,
              num_heads, scoring_mlp_layers=[H]) of :46-58,
              gru:  "gru_kernel" [H,3H], "gru_recurrent_kernel" [H,3H], "gru_bias" [2,3H]   (:140-153)
              mlp:  "mlp" = list of kernels of MLP(out_size=H) on [graph repr | node state]  (:168-182)}
    dropout_mask: the (already 1/(1-rate)-scaled) mask of :104-107, or None (eval)."""
    H = params["hidden_dim"]
    cfg = {"graph_representation_size": H, "num_heads": params["global_exchange_num_heads"],
           "weighting_fun": params["global_exchange_weighting_fun"]}
    graph_reprs = weighted_sum_graph_representation(cfg, weights["pool"], node_embeddings, node_to_graph_map, num_graphs)
    per_node = graph_reprs[node_to_graph_map.long()]  # gather_dense_gradient, :99-101
    if dropout_mask is not None:
        per_node = per_node * dropout_mask
    mode = mode.lower()
    if mode == "mean":
        return (node_embeddings + per_node) / 2
    if mode == "gru":
        return gru_cell(per_node, node_embeddings, weights["gru_kernel"], weights["gru_recurrent_kernel"], weights["gru_bias"])
    if mode == "mlp":
        return mlp_forward(torch.cat([per_node, node_embeddings], dim=-1), weights["mlp"])
    raise ValueError(f"Unknown global_exchange_mode mode {mode}")


# --------------------------------------------------------------------------------------------
# node -> graph pooling: tf2_gnn/layers/nodes_to_graph_representation.py
# --------------------------------------------------------------------------------------------
def weighted_sum_graph_representation(
    cfg: Dict[str, Any],
    weights: Dict[str, Any],
    node_embeddings: torch.Tensor,
    node_to_graph_map: torch.Tensor,
    num_graphs: int,
    dropout_masks: Optional[Dict[str, Any]] = None,
):
    """WeightedSumGraphRepresentation.call, nodes_to_graph_representation.py:170-229; ``dropout_masks`` =
    {"scoring": [...], "transformation": [...]} (see mlp_forward) restates training mode.
    cfg: graph_representation_size, num_heads, weighting_fun, scoring_mlp_activation_fun,
         transformation_mlp_activation_fun, transformation_mlp_result_{lower,upper}_bound
    weights: {"scoring": (kernels, biases|None), "transformation": (kernels, biases|None)}"""
    GD = cfg["graph_representation_size"]
    heads = cfg["num_heads"]
    wf = cfg.get("weighting_fun", "softmax").lower()
    act_s = get_activation_function_by_name(cfg.get("scoring_mlp_activation_fun", "ReLU"))
    act_t = get_activation_function_by_name(cfg.get("transformation_mlp_activation_fun", "ReLU"))
    ids = node_to_graph_map
    if wf not in ("none", "average"):
        ks, bs = weights["scoring"]
        scores = mlp_forward(node_embeddings, ks, bs, act_s, (dropout_masks or {}).get("scoring"))  # [V, heads]
        if wf == "sigmoid":
            w = torch.sigmoid(scores)
        elif wf == "softmax":
            w = torch.stack(
                [unsorted_segment_softmax(scores[:, h], ids, num_graphs) for h in range(heads)], dim=1
            )
        else:
            raise ValueError()
    kt, bt = weights["transformation"]
    reprs = mlp_forward(node_embeddings, kt, bt, act_t, (dropout_masks or {}).get("transformation"))
    if act_t is not None:
        reprs = act_t(reprs)  # :191-193 applies the activation to the MLP *output* as well
    lo = cfg.get("transformation_mlp_result_lower_bound")
    hi = cfg.get("transformation_mlp_result_upper_bound")
    if lo is not None:
        reprs = torch.clamp(reprs, min=lo)
    if hi is not None:
        reprs = torch.clamp(reprs, max=hi)
    reprs = reprs.reshape(-1, heads, GD // heads)
    if wf == "none":
        return unsorted_segment_sum(reprs.reshape(-1, GD), ids, num_graphs)  # tf.math.segment_sum
    if wf == "average":
        return unsorted_segment_mean(reprs.reshape(-1, GD), ids, num_graphs)  # tf.math.segment_mean
    weighted = (w.unsqueeze(-1) * reprs).reshape(-1, GD)
    return unsorted_segment_sum(weighted, ids, num_graphs)


def was_graph_representation(cfg, weights, node_embeddings, node_to_graph_map, num_graphs):
    """WASGraphRepresentation.call, nodes_to_graph_representation.py:311-314."""
    base = dict(cfg)
    a = weighted_sum_graph_representation(
        {**base, "weighting_fun": "softmax"}, weights["avg"], node_embeddings, node_to_graph_map, num_graphs
    )
    s = weighted_sum_graph_representation(
        {**base, "weighting_fun": "sigmoid"}, weights["sum"], node_embeddings, node_to_graph_map, num_graphs
    )
    return torch.cat([a, s], dim=-1) @ weights["out_projection"]


# --------------------------------------------------------------------------------------------
# Task heads behind the path.  PARITY UNPINNED: the reference's tests hold no
# numeric vector for any of them; the restatements follow the source line by line, [ext] TensorFlow semantics
# (sigmoid_cross_entropy_with_logits, losses.mean_*_error, round-half-even) from their published definitions.
# --------------------------------------------------------------------------------------------
def micro_f1(logits: torch.Tensor, labels: torch.Tensor):
    """tf2_gnn/models/node_multiclass_task.py:10-23 -> (f1 as float32-cast python float, (tp, fp, fn))."""
    predicted = torch.round(torch.sigmoid(logits.to(torch.float32))).to(torch.int32)  # [ext] tf.math.round: half to even
    labels = labels.to(torch.int32)
    true_pos = int(torch.count_nonzero(predicted * labels))
    false_pos = int(torch.count_nonzero(predicted * (labels - 1)))
    false_neg = int(torch.count_nonzero((predicted - 1) * labels))
    nan = float("nan")
    precision = true_pos / (true_pos + false_pos) if (true_pos + false_pos) else nan  # int64 / int64 -> float64
    recall = true_pos / (true_pos + false_neg) if (true_pos + false_neg) else nan
    denom = precision + recall
    fmeasure = (2 * precision * recall) / denom if denom == denom and denom != 0 else nan
    return fmeasure, (true_pos, false_pos, false_neg)


def sigmoid_cross_entropy_with_logits(logits: torch.Tensor, labels: torch.Tensor):
    """[ext] tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x z + log(1 + exp(-|x|)), written as TensorFlow
    writes it (two selects on x >= 0) so that autograd at x = 0 gives sigmoid(0) - z like tf.GradientTape."""
    cond = logits >= 0
    zeros = torch.zeros_like(logits)
    relu_logits = torch.where(cond, logits, zeros)
    neg_abs_logits = torch.where(cond, -logits, logits)
    return relu_logits - logits * labels + torch.log1p(torch.exp(neg_abs_logits))


def node_multiclass_task(final_node_representations, kernel, bias, node_labels):
    """NodeMulticlassTask.compute_task_output + _fast_task_metrics (node_multiclass_task.py:46-70):
    -> (per-node logits, loss = mean over nodes of the per-node label sum)."""
    per_node_logits = final_node_representations @ kernel + bias  # Dense(units=num_labels, use_bias=True), :43
    per_node_losses = sigmoid_cross_entropy_with_logits(per_node_logits, node_labels)
    loss = torch.mean(torch.sum(per_node_losses, dim=-1))
    return per_node_logits, loss


def qm9_regression_output(node_features, final_node_representations, gate, transform, node_to_graph_map, num_graphs):
    """QM9RegressionTask.compute_task_output (qm9_regression.py:83-114).  gate / transform: (kernel, bias) of the
    single-Dense MLPs (out_size=1, hidden_layers=[], use_biases=True, :43-57)."""
    per_node_output = final_node_representations @ transform[0] + transform[1]  # [V, 1]
    per_node_weight = torch.cat([node_features, final_node_representations], dim=-1) @ gate[0] + gate[1]  # [V, 1]
    per_node_weighted_output = (torch.sigmoid(per_node_weight) * per_node_output).squeeze(-1)
    return unsorted_segment_sum(per_node_weighted_output.unsqueeze(-1), node_to_graph_map, num_graphs).squeeze(-1)


def graph_regression_output(params, weights, node_features, final_or_all, node_to_graph_map, num_graphs):
    """GraphRegressionTask.compute_task_output (graph_regression_task.py:104-148).  ``final_or_all``: the GNN result
    (final, or (final, all) with use_intermediate_gnn_results).  weights: {"avg", "sum": pooling weights of
    weighted_sum_graph_representation, "regression": (kernels, biases)}."""
    if params["use_intermediate_gnn_results"]:
        _, intermediate = final_or_all
        node_representations = torch.cat((node_features,) + tuple(intermediate[1:]), dim=-1)
    else:
        node_representations = torch.cat([node_features, final_or_all], dim=-1)
    cfg = {
        "graph_representation_size": params["graph_aggregation_output_size"],
        "num_heads": params["graph_aggregation_num_heads"],
        "scoring_mlp_activation_fun": "elu",
        "transformation_mlp_activation_fun": "elu",
    }
    avg = weighted_sum_graph_representation({**cfg, "weighting_fun": "softmax"}, weights["avg"], node_representations,
                                            node_to_graph_map, num_graphs)
    tot = weighted_sum_graph_representation({**cfg, "weighting_fun": "sigmoid"}, weights["sum"], node_representations,
                                            node_to_graph_map, num_graphs)
    graph_representations = torch.cat([avg, tot], dim=-1)
    ks, bs = weights["regression"]
    return mlp_forward(graph_representations, ks, bs, torch.relu).squeeze(-1)


def regression_metrics(target_value, task_output):
    """[ext] tf.losses.mean_squared_error / mean_absolute_error (graph_regression_task.py:157-158) -> (mse, mae)."""
    d = task_output - target_value
    return torch.mean(d * d), torch.mean(torch.abs(d))
