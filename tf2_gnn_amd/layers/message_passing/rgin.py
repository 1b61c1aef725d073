"""Relational Graph Isomorphism Network layer - mirror of tf2_gnn/layers/message_passing/rgin.py."""
from typing import Any, Dict, List, Optional

import torch

from ... import ops
from .gnn_edge_mlp import GNN_Edge_MLP, mlp_hidden_sizes
from .message_passing import (
    MessagePassingInput,
    default_device,
    glorot_uniform,
    register_message_passing_implementation,
)


@register_message_passing_implementation
class RGIN(GNN_Edge_MLP):
    """h^{t+1}_v := sigma( MLP_aggr( sum_l sum_{(u,v) in A_l} MLP_l(h^t_u) ) )   (rgin.py:13-106).
    Aggregation MLP only if ``num_aggr_MLP_hidden_layers`` is not None (rgin.py:79-85); the layer
    ignores ``message_activation_before_aggregation`` (rgin.py:88-106)."""

    def graph_parts(self, num_nodes, edges_per_type, in_dim) -> int:
        return ops.G_PARTS_DEFAULT  # compact buckets / per-edge forms: every derived table of the handle

    @classmethod
    def get_default_hyperparameters(cls):
        these_hypers = {
            "use_target_state_as_input": False,
            "num_edge_MLP_hidden_layers": 1,
            "num_aggr_MLP_hidden_layers": None,
        }
        gnn_edge_mlp_hypers = super().get_default_hyperparameters()
        gnn_edge_mlp_hypers.update(these_hypers)
        return gnn_edge_mlp_hypers

    def __init__(self, params: Dict[str, Any], **kwargs):
        super().__init__(params, **kwargs)
        self._num_aggr_MLP_hidden_layers: Optional[int] = params["num_aggr_MLP_hidden_layers"]
        self._aggregation_mlp: Optional[List[torch.Tensor]] = None
        self._aggregation_mlp_vars = []

    def build(self, input_shapes: MessagePassingInput):
        if self._num_aggr_MLP_hidden_layers is not None:
            H = self._hidden_dim
            sizes = [H] * self._num_aggr_MLP_hidden_layers + [H]
            self._aggregation_mlp = []
            for j, _ in enumerate(sizes):
                w = glorot_uniform((H, H), device=default_device())
                self._aggregation_mlp.append(w)
                tag = f"dense_{j}" if j < len(sizes) - 1 else "final_layer"
                self._aggregation_mlp_vars.append(self.add_weight(f"aggregation_MLP/MLP_{tag}/kernel", w))
        super().build(input_shapes)

    def _uses_base_aggregation(self) -> bool:
        return False

    def _post_activation_name(self):
        # with an aggregation MLP the activation comes after it; otherwise fuse it
        return None if self._aggregation_mlp is not None else self._activation_name

    def _finish(self, agg, X, ctx, training):
        if self._aggregation_mlp is None:
            ctx["out"] = agg
            return agg
        hs = [agg]
        cur = agg
        n = len(self._aggregation_mlp)
        for j, w in enumerate(self._aggregation_mlp):
            last = j == n - 1
            act = None if not last else (None if self._activation_name == "gelu" else self._activation_name)
            cur = ops.gemm(cur, w, act=act if last else "relu")
            hs.append(cur)
        ctx["aggr_hs"] = hs
        if self._activation_name == "gelu":
            ctx["aggr_pre"] = cur
            cur = ops.activation_forward("gelu", cur)
        ctx["out"] = cur
        return cur

    def _backward_finish(self, grad_output, ctx):
        if self._aggregation_mlp is None:
            return super()._backward_finish(grad_output, ctx)
        act = self._activation_name
        saved = ctx["aggr_pre"] if act == "gelu" else ctx["out"]
        d = ops.activation_backward(act, grad_output, saved)
        hs = ctx["aggr_hs"]
        n = len(self._aggregation_mlp)
        for j in range(n - 1, -1, -1):
            w = self._aggregation_mlp[j]
            self._aggregation_mlp_vars[j].grad = ops.gemm(hs[j], d, trans_a=True)
            d = ops.gemm(d, w, trans_b=True)
            if j > 0:
                d = ops.activation_backward("relu", d, hs[j])
        return d
