"""Relational Graph Attention Network layer - mirror of tf2_gnn/layers/message_passing/rgat.py."""
from typing import Any, Dict, List

import torch

from ... import _lib, ops
from .message_passing import (
    MessagePassing,
    MessagePassingInput,
    default_device,
    get_graph,
    glorot_uniform,
    register_message_passing_implementation,
)


@register_message_passing_implementation
class RGAT(MessagePassing):
    """Compute new graph states by neural message passing using attention (rgat.py:11-163):
        h^{t+1}_v := sigma( sum_l sum_{(u,v) in A_l} a_l(h_u, h_v) * W_l h_u )
    Weights (rgat.py:68-89, pinned by test/layers/test_RGAT.py): per edge type one bias-free Dense
    kernel [D, H] and one attention parameter [K, 2H/K].  The L kernels live side by side in one
    [D, L*H] buffer so that Y = X W for all types is a single GEMM.

    The layer ignores ``aggregation_function`` and ``message_activation_before_aggregation``
    (rgat.py:125-163), like the reference."""

    @classmethod
    def get_default_hyperparameters(cls):
        these_hypers = {
            "num_heads": 3,
        }
        mp_hypers = super().get_default_hyperparameters()
        mp_hypers.update(these_hypers)
        return mp_hypers

    def __init__(self, params: Dict[str, Any], **kwargs):
        super().__init__(params, **kwargs)
        self._num_heads: int = params["num_heads"]
        self._edge_type_to_message_computation_layer = []
        self._edge_type_to_attention_parameters = []
        self._kernels = None  # [D, L*H]
        self._attn = None  # [L, K, 2H/K]
        self._num_edge_types = None

    def build(self, input_shapes: MessagePassingInput):
        D = int(input_shapes.node_embeddings[-1])
        L = len(input_shapes.adjacency_lists)
        H, K = self._hidden_dim, self._num_heads
        if H % K != 0:
            raise ValueError(f"hidden_dim {H} must be divisible by num_heads {K}")
        per_head_dim = H // K
        dev = default_device()
        self._num_edge_types = L
        self._kernels = torch.empty((D, L * H), dtype=torch.float32, device=dev)
        self._attn = torch.empty((L, K, 2 * per_head_dim), dtype=torch.float32, device=dev)
        for i in range(L):
            self._kernels[:, i * H : (i + 1) * H].copy_(glorot_uniform((D, H), device=dev))
            self._edge_type_to_message_computation_layer.append(
                self.add_weight(f"edge_type_{i}/Edge_weight_{i}/kernel", self._kernels[:, i * H : (i + 1) * H])
            )
            self._attn[i].copy_(glorot_uniform((K, 2 * per_head_dim), device=dev))
            self._edge_type_to_attention_parameters.append(
                self.add_weight(f"edge_type_{i}/Edge_attention_parameters_{i}", self._attn[i])
            )
        super().build(input_shapes)

    def _message_function(self, edge_source_states, edge_target_states, num_incoming_to_node_per_message,
                          edge_type_idx, training):
        raise NotImplementedError(
            "RGAT evaluates attention on the bucketed graph (csrc/rgat.hip); the per-edge form lives in "
            "oracle/tf2gnn_oracle.py:_rgat_message"
        )

    def call(self, inputs: MessagePassingInput, training: bool = False):
        X = inputs.node_embeddings
        V = X.shape[0]
        g = get_graph(inputs.adjacency_lists, V)
        L, H, K = g.num_edge_types, self._hidden_dim, self._num_heads
        if L != self._num_edge_types:
            raise ValueError(f"layer was built for {self._num_edge_types} edge types, got {L}")
        lib = _lib.load()
        dev = X.device
        Y = ops.gemm(X, self._kernels)  # [V, L*H] == rows (v, l) of width H
        s_src = torch.empty((V * L, K), dtype=torch.float32, device=dev)
        s_tgt = torch.empty((V * L, K), dtype=torch.float32, device=dev)
        _lib.check(
            lib.tfgnn_rgat_node_scores(
                ops._ptr(Y), ops._ptr(self._attn), V, L, K, H, ops._ptr(s_src), ops._ptr(s_tgt), ops._stream()
            )
        )
        out = torch.empty((V, H), dtype=torch.float32, device=dev)
        att = torch.empty((g.num_edges, K), dtype=torch.float32, device=dev)
        act = self._activation_name
        fused = None if act == "gelu" else act
        _lib.check(
            lib.tfgnn_rgat_aggregate(
                ops._ptr(g.array(ops.G_NODEPTR_BY_DST)), ops._ptr(g.array(ops.G_COLL_BY_DST)), ops._ptr(Y),
                ops._ptr(s_src), ops._ptr(s_tgt), V, L, K, H, ops.act_id(fused), ops._ptr(out), ops._ptr(att),
                ops._stream(),
            )
        )
        ctx = {"graph": g, "X": X, "Y": Y, "s_src": s_src, "s_tgt": s_tgt, "att": att, "fused_act": act}
        if act == "gelu":
            ctx["pre"] = out
            out = ops.activation_forward("gelu", out)
        ctx["out"] = out
        self._ctx = ctx
        return out

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("RGAT backward is not implemented yet")
