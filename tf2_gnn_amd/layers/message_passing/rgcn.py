"""Relational Graph Convolutional Network layer - mirror of tf2_gnn/layers/message_passing/rgcn.py."""
from typing import Any, Dict

from .gnn_edge_mlp import GNN_Edge_MLP
from .message_passing import register_message_passing_implementation


@register_message_passing_implementation
class RGCN(GNN_Edge_MLP):
    """Compute new graph states by neural message passing (rgcn.py:12-62):
        h^{t+1}_v := sigma( sum_l sum_{(u,v) in A_l} W_l h^t_u / c_{v,l} )
    i.e. GNN_Edge_MLP with no target input, degree normalisation and a single bias-free kernel per
    edge type (rgcn.py:52-56).  Runs as one gather kernel + one MFMA GEMM (gnn_edge_mlp path A)."""

    @classmethod
    def get_default_hyperparameters(cls):
        these_hypers = {
            "use_target_state_as_input": False,
            "normalize_by_num_incoming": True,
            "num_edge_MLP_hidden_layers": 0,
        }
        mp_hypers = super().get_default_hyperparameters()
        mp_hypers.update(these_hypers)
        return mp_hypers

    def __init__(self, params: Dict[str, Any], **kwargs):
        super().__init__(params, **kwargs)
