"""GNN layer with feature-wise linear modulation - mirror of tf2_gnn/layers/message_passing/gnn_film.py."""
from typing import Any, Dict

import torch

from ... import _lib, ops
from .gnn_edge_mlp import GNN_Edge_MLP, StackedEdgeMLPs
from .message_passing import (
    MessagePassing,
    MessagePassingInput,
    default_device,
    get_graph,
    register_message_passing_implementation,
)


@register_message_passing_implementation
class GNN_FiLM(GNN_Edge_MLP):
    """Compute new graph states by neural message passing modulated by the target state (gnn_film.py:12-108):
        h^{t+1}_v := sigma( sum_l sum_{(u,v) in A_l} gamma_{l,v} * (1/c_{v,l} * W_l h^t_u) + beta_{l,v} )
        (gamma_{l,v} | beta_{l,v}) := FiLM-MLP_l(h^t_v)
    Weights (gnn_film.py:67-82): per edge type one FiLM parameter MLP (out 2H, created first) and one edge MLP.

    On the device every edge of a (target, type) bucket shares gamma / beta, so the modulation is applied to the
    bucket sums Z[v,l] (gather kernel + MFMA GEMMs as in GNN_Edge_MLP) by a node-side epilogue
    (csrc/edge.hip tfgnn_film_combine_*):  out[v] = sigma( scale_v * sum_l (gamma_{l,v} * Z[v,l] + cnt[v,l] * beta_{l,v}) ).
    That form covers sum / mean / sqrt_n aggregation with the activation after aggregation and source-only edge MLPs of
    any depth (the class defaults).  Max aggregation, message_activation_before_aggregation and
    use_target_state_as_input take the per-edge form of the reference (``_call_per_edge``): messages per edge from the
    edge-MLP machinery of GNN_Edge_MLP, modulated per edge (csrc/edge.hip tfgnn_film_edge_*), then the general
    aggregation kernel."""

    def graph_parts(self, num_nodes, edges_per_type, in_dim) -> int:
        return ops.G_PARTS_DEFAULT  # compact buckets / per-edge forms: every derived table of the handle

    @classmethod
    def get_default_hyperparameters(cls):
        these_hypers = {
            "use_target_state_as_input": False,
            "normalize_by_num_incoming": False,
            "num_edge_MLP_hidden_layers": 0,
            "film_parameter_MLP_hidden_layers": [],
        }
        mp_hypers = super().get_default_hyperparameters()
        mp_hypers.update(these_hypers)
        return mp_hypers

    def __init__(self, params: Dict[str, Any], **kwargs):
        super().__init__(params, **kwargs)
        self._film_parameter_MLP_hidden_layers = params["film_parameter_MLP_hidden_layers"]
        self._film_mlps = None

    def build(self, input_shapes: MessagePassingInput):
        D = int(input_shapes.node_embeddings[-1])
        L = len(input_shapes.adjacency_lists)
        # the FiLM MLPs are created before the edge MLPs (gnn_film.py:72-82)
        self._film_mlps = StackedEdgeMLPs(self, L, D, 2 * self._hidden_dim, self._film_parameter_MLP_hidden_layers,
                                          default_device(), scope="edge_type_{l}-FiLM")
        super().build(input_shapes)

    def _message_function(self, edge_source_states, edge_target_states, num_incoming_to_node_per_message,
                          edge_type_idx, training):
        """gnn_film.py:83-108 in torch operations, for user subclasses (see GNN_Edge_MLP._message_function)."""
        messages = GNN_Edge_MLP._message_function(self, edge_source_states, edge_target_states,
                                                  num_incoming_to_node_per_message, edge_type_idx, training)
        film = edge_target_states
        kernels = self._film_mlps.vars[edge_type_idx]
        for j, var in enumerate(kernels):
            film = film @ var.value
            if j < len(kernels) - 1:
                film = torch.relu(film)
        H = self._hidden_dim
        return film[:, :H] * messages + film[:, H:]

    _message_function._tfgnn_builtin = True

    def _node_side(self) -> bool:
        return not (self._aggregation_name == "max" or self._pre_activation() or self._use_target_state_as_input)

    def _film_scales(self, g):
        """(per-bucket 1/(c+1e-7) | None, the same per edge in by-src order | None, mean / sqrt_n factor per node | None):
        unlike the other layers the two factors cannot be merged - beta is added between them."""
        from ..graph_scales import graph_scales

        row_scale, _, ew_s, _ = graph_scales(g, bool(self._normalize_by_num_incoming), "sum")
        node_scale = graph_scales(g, bool(self._normalize_by_num_incoming), self._aggregation_name)[3]
        return row_scale, ew_s, node_scale

    # ---- per-edge form: gamma_{l,v} * (w_e * MLP_l([x_u | x_v])) + beta_{l,v} per edge, then any aggregation --------------
    def _call_per_edge(self, X, g):
        V, D = X.shape
        L, H, E = g.num_edge_types, self._hidden_dim, g.num_edges
        lib = _lib.load()
        _, ew_d, _, node_scale = self._scales(g)
        fuse_act = self._post_activation_name()
        ctx = {"graph": g, "X": X, "per_edge": True, "fused_act": fuse_act}
        film = self._mlp_all_types(X, L, ctx, mlps=self._film_mlps, key="film_acts")  # [V, L, 2H]
        if E == 0:
            ctx["out"] = self._aggregate_nothing(V, X, fuse_act, ctx)
            self._ctx = ctx
            return ctx["out"]
        src_l, tgt_l, tgt_node, w_orig, off, _ = self._original_order(g, ew_d)  # edge-list order
        if self._use_target_state_as_input:
            msgs, ctx["edge_acts"] = self._edge_messages_C(X, g, ew_d)  # [E, H]
            mrow = None
        else:
            msgs = self._mlp_all_types(X, L, ctx).view(V * L, H)  # row (u, l) = MLP_l(x_u)
            mrow = src_l
        M = torch.empty((E, H), dtype=torch.float32, device=X.device)
        _lib.check(lib.tfgnn_film_edge_forward(ops._ptr(msgs), ops._ptr(mrow), ops._ptr(film), ops._ptr(tgt_l), ops._ptr(w_orig),
                                               E, H, ops._ptr(M), ops._stream()))
        ctx.update({"film": film, "msgs": msgs, "mrow": mrow, "M": M})
        out = self._gather_messages(g, M, g.array(ops.G_EID_BY_DST), None, node_scale, fuse_act, ctx)
        ctx["out"] = out
        self._ctx = ctx
        return out

    def _backward_per_edge(self, grad_output, ctx):
        g, X = ctx["graph"], ctx["X"]
        V, D = X.shape
        L, H, E = g.num_edge_types, self._hidden_dim, g.num_edges
        lib = _lib.load()
        mlps = self._edge_type_mlps
        dX = torch.empty_like(X)
        if E == 0:
            dfilm = torch.zeros((V, L, 2 * H), dtype=torch.float32, device=X.device)
            self._mlp_all_types_backward(self._film_mlps, X, ctx["film_acts"], dfilm, dX, accumulate=False)
            self._film_mlps.publish_grads()
            mlps.grads = [torch.zeros_like(W) for W in mlps.kernels]
            mlps.publish_grads()
            return dX
        d_agg = self._backward_finish(grad_output, ctx)
        _, ew_d, _, node_scale = self._scales(g)
        src_l, tgt_l, tgt_node, w_orig, off, _ = self._original_order(g, ew_d)
        eid_d = g.array(ops.G_EID_BY_DST)
        dM = self._message_grads(g, d_agg, ctx, ctx["M"], None, tgt_node, None, node_scale, eid_d)  # [E, H], edge-list order
        dmsg = torch.empty((E, H), dtype=torch.float32, device=X.device)
        dfilm_e = torch.empty((E, 2 * H), dtype=torch.float32, device=X.device)
        _lib.check(lib.tfgnn_film_edge_backward(ops._ptr(dM), ops._ptr(ctx["msgs"]), ops._ptr(ctx["mrow"]), ops._ptr(ctx["film"]),
                                                ops._ptr(tgt_l), ops._ptr(w_orig), E, H, ops._ptr(dmsg), ops._ptr(dfilm_e),
                                                ops._stream()))
        # gamma / beta of (v, l) were used by every edge of that bucket
        dfilm = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, dfilm_e, col=eid_d).view(V, L, 2 * H)
        self._mlp_all_types_backward(self._film_mlps, X, ctx["film_acts"], dfilm, dX, accumulate=False)
        self._film_mlps.publish_grads()
        if self._use_target_state_as_input:
            dX = ops.add_scale(dX, self._backward_C(None, ctx, dcur=dmsg), 1.0)
        else:
            # MLP_l(x_u) was used by every edge leaving (u, l)
            G = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, dmsg, col=g.array(ops.G_EID_BY_SRC)).view(V, L, H)
            self._mlp_all_types_backward(mlps, X, ctx["mlp_acts"], G, dX, accumulate=True)
            mlps.publish_grads()
        return dX

    def call(self, inputs: MessagePassingInput, training: bool = False):
        if self._user_message_function():
            return MessagePassing.call(self, inputs, training)
        X = inputs.node_embeddings
        if not self._node_side():
            V = X.shape[0]
            g = get_graph(inputs.adjacency_lists, V)
            if g.num_edge_types != self._num_edge_types:
                raise ValueError(f"layer was built for {self._num_edge_types} edge types, got {g.num_edge_types}")
            return self._call_per_edge(X, g)
        V, D = X.shape
        g = get_graph(inputs.adjacency_lists, V)
        L, H = g.num_edge_types, self._hidden_dim
        if L != self._num_edge_types:
            raise ValueError(f"layer was built for {self._num_edge_types} edge types, got {L}")
        row_scale, ew_s, node_scale = self._film_scales(g)
        mlps = self._edge_type_mlps
        ctx = {"graph": g, "X": X}
        Z = torch.empty((V, L, H), dtype=torch.float32, device=X.device)
        if mlps.num_layers == 1:
            # Z[v,l] = (scale_{v,l} * sum_{(u,v) in A_l} x_u) @ W_l
            A = torch.empty((V, L, D), dtype=torch.float32, device=X.device)
            ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=row_scale, out=A.view(V * L, D))
            for l in range(L):
                ops.gemm(A[:, l, :], mlps.kernels[0][l], out=Z[:, l, :])
            ctx["A"] = A
        else:
            # Z[v,l] = scale_{v,l} * sum_{(u,v) in A_l} MLP_l(x_u)
            Y = self._mlp_all_types(X, L, ctx)
            if L > 0:
                ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, Y.view(V * L, H), col=g.array(ops.G_COLL_BY_DST),
                                 row_scale=row_scale, out=Z.view(V * L, H))
        film = self._mlp_all_types(X, L, ctx, mlps=self._film_mlps, key="film_acts")  # [V, L, 2H]
        act = self._activation_name
        out = torch.empty((V, H), dtype=torch.float32, device=X.device)
        pre = torch.empty_like(out) if act == "gelu" else None
        _lib.check(
            _lib.load().tfgnn_film_combine_forward(
                ops._ptr(Z), ops._ptr(film), ops._ptr(g.array(ops.G_ROWPTR_BY_DST)), ops._ptr(node_scale), V, L, H,
                ops.act_id(act), ops._ptr(pre), ops._ptr(out), ops._stream(),
            )
        )
        ctx.update({"Z": Z, "film": film, "out": out, "pre": pre, "node_scale": node_scale})
        self._ctx = ctx
        return out

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        if ctx.get("generic"):  # user message function on the generic path
            return MessagePassing.backward(self, grad_output)
        if ctx.get("per_edge"):
            return self._backward_per_edge(grad_output, ctx)
        g, X, Z, film = ctx["graph"], ctx["X"], ctx["Z"], ctx["film"]
        V, D = X.shape
        L, H = g.num_edge_types, self._hidden_dim
        act = self._activation_name
        d_pre = grad_output
        if act is not None:
            d_pre = ops.activation_backward(act, grad_output, ctx["pre"] if act == "gelu" else ctx["out"])
        d_pre = d_pre.contiguous()
        dZ = torch.empty_like(Z)
        dfilm = torch.empty_like(film)
        _lib.check(
            _lib.load().tfgnn_film_combine_backward(
                ops._ptr(d_pre), ops._ptr(Z), ops._ptr(film), ops._ptr(g.array(ops.G_ROWPTR_BY_DST)),
                ops._ptr(ctx["node_scale"]), V, L, H, ops._ptr(dZ), ops._ptr(dfilm), ops._stream(),
            )
        )
        dX = torch.empty_like(X)
        # the FiLM parameter MLPs read the node's own state
        self._mlp_all_types_backward(self._film_mlps, X, ctx["film_acts"], dfilm, dX, accumulate=False)
        self._film_mlps.publish_grads()
        mlps = self._edge_type_mlps
        _, ew_s, _ = self._film_scales(g)
        if L == 0:
            mlps.grads = [torch.zeros_like(W) for W in mlps.kernels]
        elif mlps.num_layers == 1:
            A = ctx["A"]
            W = mlps.kernels[0]
            gW = torch.empty_like(W)
            dA = torch.empty_like(A)
            for l in range(L):
                ops.gemm(A[:, l, :], dZ[:, l, :], trans_a=True, out=gW[l])
                ops.gemm(dZ[:, l, :], W[l], trans_b=True, out=dA[:, l, :])
            mlps.grads = [gW]
            # dX[u] += sum over edges (u -> v) of type l of scale_{v,l} * dA[v,l]
            dXe = ops.graph_gather(g, ops.VIEW_BY_SRC_NODE, dA.view(V * L, D), col=g.array(ops.G_COLL_BY_SRC),
                                   edge_weight=ew_s)
            dX = ops.add_scale(dX, dXe, 1.0)
        else:
            # dY[u,l] = sum over edges (u -> v) of type l of scale_{v,l} * dZ[v,l]
            G = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, dZ.view(V * L, H), col=g.array(ops.G_COLL_BY_SRC),
                                 edge_weight=ew_s).view(V, L, H)
            self._mlp_all_types_backward(mlps, X, ctx["mlp_acts"], G, dX, accumulate=True)
        mlps.publish_grads()
        return dX
