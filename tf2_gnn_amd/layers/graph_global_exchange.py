"""Graph-global information exchange between message passing layers - mirror of
tf2_gnn/layers/graph_global_exchange.py ("next" row f3 of SURVEY.md section 8).

Every piece runs on kernels the hot path already has: the per-graph representation is the pooling layer
(WeightedSumGraphRepresentation: MFMA GEMMs + pool.hip segment softmax / weighted sum), the broadcast back to
the nodes and its gradient (gather_dense_gradient, utils/gather_dense_gradient.py:9-14) are row gathers /
segment sums of the generic gather kernel over the sorted node_to_graph_map, the GRU update is the GGNN gate
kernel."""
from typing import List, NamedTuple

import torch

from .. import ops
from .message_passing.message_passing import _INIT_GEN, Variable, default_device, glorot_uniform
from .nodes_to_graph_representation import (
    prefix_names,
    MLP,
    NodesToGraphRepresentationInput,
    WeightedSumGraphRepresentation,
    segment_offsets,
)


class GraphGlobalExchangeInput(NamedTuple):
    """Input named tuple for graph global information exchange in GNNs (graph_global_exchange.py:12-17)."""

    node_embeddings: torch.Tensor
    node_to_graph_map: torch.Tensor
    num_graphs: int


class GraphGlobalExchange:
    """Update node representations based on graph-global information (graph_global_exchange.py:20-109)."""

    def __init__(self, hidden_dim: int, weighting_fun: str = "softmax", num_heads: int = 4, dropout_rate: float = 0.0):
        self._hidden_dim = hidden_dim
        self._weighting_fun = weighting_fun
        self._num_heads = num_heads
        self._dropout_rate = dropout_rate
        self._node_to_graph_representation_layer = None
        self.dropout_seed = 0
        self._ctx = None

    def build(self, tensor_shapes: GraphGlobalExchangeInput):
        self._node_to_graph_representation_layer = WeightedSumGraphRepresentation(
            graph_representation_size=self._hidden_dim,
            weighting_fun=self._weighting_fun,
            num_heads=self._num_heads,
            scoring_mlp_layers=[self._hidden_dim],
        )
        self._node_to_graph_representation_layer.build(
            NodesToGraphRepresentationInput(tensor_shapes.node_embeddings, tensor_shapes.node_to_graph_map, tensor_shapes.num_graphs)
        )
        # graph_global_exchange.py:119,141,170: the subclasses build everything under tf.name_scope(<class name>)
        prefix_names(self._node_to_graph_representation_layer.trainable_variables, self.__class__.__name__)

    @property
    def trainable_variables(self) -> List[Variable]:
        return self._own_variables() + self._node_to_graph_representation_layer.trainable_variables

    def _own_variables(self) -> List[Variable]:
        return []

    def __call__(self, inputs: GraphGlobalExchangeInput, training: bool = False):
        return self.call(inputs, training)

    # ---- shared pieces ----------------------------------------------------------------------------
    def _compute_per_node_graph_representations(self, inputs: GraphGlobalExchangeInput, training: bool = False):
        """graph_global_exchange.py:83-109 -> [V, hidden_dim]"""
        X = inputs.node_embeddings
        V = X.shape[0]
        G = int(inputs.num_graphs)
        ids = inputs.node_to_graph_map.to(torch.int32).contiguous()
        graph_reprs = self._node_to_graph_representation_layer(
            NodesToGraphRepresentationInput(X, ids, G), training=training
        )  # [G, hidden_dim]
        ident = torch.arange(V + 1, dtype=torch.int32, device=X.device)
        per_node = ops.gather_reduce(ident, ids, graph_reprs)  # row v = graph_reprs[node_to_graph_map[v]]
        mask = None
        if training and self._dropout_rate > 0.0:
            per_node, mask = ops.dropout_forward(per_node, float(self._dropout_rate), self.dropout_seed)
        self._bcast = {"ids": ids, "ident": ident, "V": V, "G": G, "mask": mask}
        return per_node

    def _backward_per_node_graph_representations(self, d_per_node: torch.Tensor) -> torch.Tensor:
        """gradient of the above w.r.t. the node embeddings (through the pooling layer)."""
        b = self._bcast
        if b["mask"] is not None:
            d_per_node = ops.mul(d_per_node, b["mask"])
        # scatter_nd of gather_dense_gradient: node_to_graph_map is sorted, a graph's nodes are one segment
        ptr = segment_offsets(b["ids"], b["G"])
        d_graph = ops.gather_reduce(ptr, b["ident"][: b["V"]], d_per_node.contiguous())
        return self._node_to_graph_representation_layer.backward(d_graph)


class GraphGlobalMeanExchange(GraphGlobalExchange):
    """(node state + graph representation) / 2 (graph_global_exchange.py:112-131)."""

    def call(self, inputs: GraphGlobalExchangeInput, training: bool = False):
        per_node = self._compute_per_node_graph_representations(inputs, training)
        return ops.add_scale(inputs.node_embeddings, per_node, 0.5)

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        half = ops.add_scale(grad_output, grad_output, 0.25)  # 0.5 * g
        return ops.add_scale(half, self._backward_per_node_graph_representations(half), 1.0)


class GraphGlobalGRUExchange(GraphGlobalExchange):
    """GRUCell(inputs = graph representation, state = node state) (graph_global_exchange.py:134-159)."""

    def build(self, tensor_shapes: GraphGlobalExchangeInput):
        H = self._hidden_dim
        dev = default_device()
        kernel = glorot_uniform((H, 3 * H), device=dev)
        q, _ = torch.linalg.qr(torch.randn((3 * H, H), generator=_INIT_GEN, dtype=torch.float32))  # [ext] orthogonal
        self._gru = {
            "kernel": Variable("GraphGlobalGRUExchange/kernel", kernel),
            "recurrent_kernel": Variable("GraphGlobalGRUExchange/recurrent_kernel", q.t().contiguous().to(dev)),
            "bias": Variable("GraphGlobalGRUExchange/bias", torch.zeros((2, 3 * H), dtype=torch.float32, device=dev)),
        }
        super().build(tensor_shapes)

    def _own_variables(self):
        return [self._gru["kernel"], self._gru["recurrent_kernel"], self._gru["bias"]]

    def call(self, inputs: GraphGlobalExchangeInput, training: bool = False):
        X = inputs.node_embeddings
        per_node = self._compute_per_node_graph_representations(inputs, training)
        b = self._gru["bias"].value
        mh = ops.gemm(X, self._gru["recurrent_kernel"].value, bias=b[1])
        fused = ops.gemm_gru(per_node, self._gru["kernel"].value, b[0], mh, X)  # as in GGNN: mx stays on chip
        if fused is not None:
            h_new, gates = fused
        else:
            mx = ops.gemm(per_node, self._gru["kernel"].value, bias=b[0])
            h_new, gates = ops.gru_gates_forward(mx, mh, X)
        self._ctx = {"X": X, "per_node": per_node, "mh": mh, "gates": gates}
        return h_new

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        c, ru = self._ctx, self._gru
        X = c["X"]
        dmx, dmh, dh_direct = ops.gru_gates_backward(grad_output.contiguous(), c["gates"], c["mh"], X)
        ru["kernel"].grad = ops.gemm(c["per_node"], dmx, trans_a=True)
        ru["recurrent_kernel"].grad = ops.gemm(X, dmh, trans_a=True)
        ru["bias"].grad = torch.stack([ops.colsum(dmx), ops.colsum(dmh)], dim=0)
        d_per_node = ops.gemm(dmx, ru["kernel"].value, trans_b=True)
        dX = ops.gemm(dmh, ru["recurrent_kernel"].value, trans_b=True, out=dh_direct, accumulate=True)
        return ops.add_scale(dX, self._backward_per_node_graph_representations(d_per_node), 1.0)


class GraphGlobalMLPExchange(GraphGlobalExchange):
    """MLP([graph representation | node state]) (graph_global_exchange.py:162-183; dpu_utils MLP defaults [ext]:
    one hidden layer of out_size units, relu, no biases)."""

    def build(self, tensor_shapes: GraphGlobalExchangeInput):
        self._mlp = MLP(out_size=self._hidden_dim, hidden_layers=1)  # default name "MLP"
        self._mlp.build(2 * self._hidden_dim)
        prefix_names(self._mlp.variables, self.__class__.__name__)
        super().build(tensor_shapes)

    def _own_variables(self):
        return self._mlp.variables

    def call(self, inputs: GraphGlobalExchangeInput, training: bool = False):
        X = inputs.node_embeddings
        H = self._hidden_dim
        per_node = self._compute_per_node_graph_representations(inputs, training)
        cat = torch.empty((X.shape[0], 2 * H), dtype=torch.float32, device=X.device)
        cat[:, :H].copy_(per_node)
        cat[:, H:].copy_(X)
        return self._mlp(cat, training=training, dropout_masks=getattr(self, "mlp_dropout_masks", None))

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        H = self._hidden_dim
        d_cat = self._mlp.backward(grad_output.contiguous())
        d_per_node = d_cat[:, :H].contiguous()
        dX = d_cat[:, H:].contiguous()
        return ops.add_scale(dX, self._backward_per_node_graph_representations(d_per_node), 1.0)
