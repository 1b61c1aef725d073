"""GNN_Edge_MLP and its hyper-parameter specialisations - mirror of
tf2_gnn/layers/message_passing/gnn_edge_mlp.py (class, hyper-parameters, weights: one bias-free
MLP per edge type, gnn_edge_mlp.py:64-82, pinned by test/layers/test_RGCN.py).

The reference computes, per edge type l and per EDGE (gnn_edge_mlp.py:84-107):
    m_e = MLP_l([x_src | x_tgt] or x_src) ;  m_e *= 1 / (c_{l,tgt} + 1e-7)   (optional)
and then concat -> unsorted_segment_<agg> -> activation (message_passing.py:135-179).
Here the same function is evaluated with the per-edge matmul moved to the node side:

  path A (0 hidden layers, sum/mean/sqrt_n, activation after aggregation - RGCN, GGNN):
      A[v, l, :] = scale_{l,v} * sum_{(u,v) in A_l} x_u          gather kernel over rows (v, l)
      out        = act( [A_0 | ... | A_{L-1}] @ [W_0; ...; W_{L-1}] )     one MFMA GEMM
      (+ target states: sum_e [x_u|x_v] W = (sum x_u) W_s + c_{l,v} x_v W_t)
  path B (>= 1 hidden layer, or max aggregation, or activation before aggregation, source-only):
      Y[u, l, :] = MLP_l(x_u) for every node        MFMA GEMMs, [V, L, H]
      out[v]     = post( agg_{(u,v,l)} pre( w_e * Y[u, l, :] ) )         gather kernel over rows v
  path C (target states AND >= 1 hidden layer): genuine per-edge MLP (first layer separable:
      x_u W_s + x_v W_t), evaluated per edge in edge-list order.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from ... import _lib, ops
from .message_passing import (
    apply_gradient_epilogue,
    MessagePassing,
    MessagePassingInput,
    Variable,
    default_device,
    get_graph,
    glorot_uniform,
    register_message_passing_implementation,
)


def mlp_hidden_sizes(out_size: int, hidden_layers) -> List[int]:
    """[ext] dpu_utils.tf2utils.MLP: an int n means n hidden layers of ``out_size`` units."""
    if isinstance(hidden_layers, int):
        return [out_size] * hidden_layers
    return list(hidden_layers)


PER_EDGE_MIN_ROWS = 65536  # (tests lower it; the library's own threshold is asked through tfgnn_gemm_gathered_supported)


def _skip_empty_blocks(L: int, Din: int) -> bool:
    """Does the forward product of path A run over the nodes in pattern order and skip the all-zero type blocks of a row tile
    (tfgnn_sp_gemm_nt_dropout d_tile_kmask)?  One edge type has nothing to skip; the library takes at most 8 blocks of at
    most 1024 columns.  TFGNN_NT_SKIP_EMPTY=0 keeps the node order (A/B measurements)."""
    import os

    return 2 <= L <= 8 and Din % 16 == 0 and Din <= 1024 and ops.env("TFGNN_NT_SKIP_EMPTY", "1") == "1"


def messages_per_edge(layer, g, D, H) -> bool:
    """one product row per EDGE (rows of X read through an index) rather than per (node, type) bucket?  Pays when edges are
    fewer than buckets and the fused gathered product (tfgnn_gemm_gathered) takes every edge type's product - the library is
    asked (tfgnn_gemm_gathered_supported), the shape conditions are not restated here."""
    if ops.get_gemm_mode() == ops.GEMM_FP32:
        return False
    L, E, V = g.num_edge_types, g.num_edges, g.num_nodes
    if E == 0 or E >= V * L or layer._aggregation_name == "max" or layer._pre_activation():
        return False
    lib = _lib.load()
    forced = PER_EDGE_MIN_ROWS < 65536  # the layer-logic tests run the formulation on small graphs (separate gather + product)
    for c in g.edges_per_type:
        if c == 0:
            continue
        if forced:
            if c < PER_EDGE_MIN_ROWS or D not in (64, 96, 128) or H % 128 != 0:
                return False
        elif not lib.tfgnn_gemm_gathered_supported(int(c), int(H), int(D), int(D), int(V)):
            return False
    return True


def _relu_input_grad(d_out, W, layer_input, out):
    """out = (d_out @ W^T) * relu'(layer_input), the gradient w.r.t. the pre-activation of the layer below, with the
    factor applied in the product's epilogue when the active kernel has one (one pass over [rows, width] less)."""
    res = ops.gemm_grad(d_out, W, trans_b=True, out=out, act_grad=("relu", layer_input))
    if res.data_ptr() != out.data_ptr():  # unfused route: the factor was applied out of place
        out.copy_(res)


class StackedEdgeMLPs:
    """L bias-free MLPs (one per edge type) stored layer-wise as [L, in, out] tensors so that the
    per-type kernels of one layer are contiguous ([L*in, out] is the vertical stack)."""

    def __init__(self, owner: MessagePassing, num_types: int, in_size: int, out_size: int, hidden_layers, device,
                 scope: str = "edge_type_{l}"):
        sizes = mlp_hidden_sizes(out_size, hidden_layers) + [out_size]
        self.L = num_types
        self.kernels: List[torch.Tensor] = []  # per MLP layer j: [L, in_j, out_j]
        self.vars: List[List[Variable]] = [[] for _ in range(num_types)]
        last = in_size
        for j, size in enumerate(sizes):
            w = torch.empty((num_types, last, size), dtype=torch.float32, device=device)
            self.kernels.append(w)
            last = size
        # create in the reference's order (all layers of type 0, then type 1, ...) so that seeded
        # initialisation and variable order match a per-type construction (gnn_edge_mlp.py:73-81)
        for l in range(num_types):
            for j, w in enumerate(self.kernels):
                w[l].copy_(glorot_uniform(w[l].shape, device=device))
                tag = f"dense_{j}" if j < len(sizes) - 1 else "final_layer"
                self.vars[l].append(owner.add_weight(f"{scope.format(l=l)}/MLP_{tag}/kernel", w[l]))
        self.grads: List[Optional[torch.Tensor]] = [None] * len(self.kernels)

    @property
    def num_layers(self):
        return len(self.kernels)

    def publish_grads(self):
        for j, g in enumerate(self.grads):
            for l in range(self.L):
                self.vars[l][j].grad = None if g is None else g[l]


@register_message_passing_implementation
class GNN_Edge_MLP(MessagePassing):
    """Compute new graph states by neural message passing using an edge MLP (gnn_edge_mlp.py:12-107):
        h^{t+1}_v := sigma( sum_l sum_{(u,v) in A_l} MLP_l(h^t_u || h^t_v) )
    """

    @classmethod
    def get_default_hyperparameters(cls):
        these_hypers = {
            "use_target_state_as_input": True,
            "normalize_by_num_incoming": False,
            "num_edge_MLP_hidden_layers": 1,
        }
        mp_hypers = super().get_default_hyperparameters()
        mp_hypers.update(these_hypers)
        return mp_hypers

    def __init__(self, params: Dict[str, Any], **kwargs):
        super().__init__(params, **kwargs)
        self._use_target_state_as_input = params["use_target_state_as_input"]
        self._normalize_by_num_incoming = params["normalize_by_num_incoming"]
        self._num_edge_MLP_hidden_layers = params["num_edge_MLP_hidden_layers"]
        self._compact_opt_in = bool(params.get("use_compact_buckets", False))  # not a reference hyper-parameter
        self._edge_type_mlps: Optional[StackedEdgeMLPs] = None
        self._in_dim = None
        self._num_edge_types = None

    def build(self, input_shapes: MessagePassingInput):
        node_embedding_shapes = input_shapes.node_embeddings
        adjacency_list_shapes = input_shapes.adjacency_lists
        num_edge_types = len(adjacency_list_shapes)
        D = int(node_embedding_shapes[-1])
        edge_layer_input_size = 2 * D if self._use_target_state_as_input else D
        self._in_dim = D
        self._num_edge_types = num_edge_types
        self._edge_type_mlps = StackedEdgeMLPs(
            self, num_edge_types, edge_layer_input_size, self._hidden_dim,
            self._num_edge_MLP_hidden_layers, default_device(),
        )
        super().build(input_shapes)

    # The reference's per-edge definition (gnn_edge_mlp.py:84-107), written with torch operations on the layer's
    # variables.  The built-in classes never call it (they evaluate the same function on the node side, module
    # docstring); it exists for USER subclasses that override ``_message_function`` and call ``super()`` - those run on
    # the generic path of MessagePassing (library kernels around the user function, torch autograd through it).
    def _message_function(self, edge_source_states, edge_target_states, num_incoming_to_node_per_message,
                          edge_type_idx, training):
        cur = (torch.cat([edge_source_states, edge_target_states], dim=-1) if self._use_target_state_as_input
               else edge_source_states)
        kernels = self._edge_type_mlps.vars[edge_type_idx]
        for j, var in enumerate(kernels):
            cur = cur @ var.value
            if j < len(kernels) - 1:
                cur = torch.relu(cur)  # dpu_utils MLP [ext]: relu between the bias-free Dense layers
        if self._normalize_by_num_incoming:
            cur = (1.0 / (num_incoming_to_node_per_message + 1e-7)).unsqueeze(-1) * cur
        return cur

    _message_function._tfgnn_builtin = True

    def _user_message_function(self) -> bool:
        """Did a subclass replace the message function?  Then the node-side formulations below do not apply."""
        return not getattr(type(self)._message_function, "_tfgnn_builtin", False)

    def graph_parts(self, num_nodes: int, edges_per_type, in_dim: int) -> int:
        """path A without target states reads the typed views only (forward: buckets by target, backward: by source); with
        one message per edge (molecule-sized batches) the node view as well.  Everything else: all parts."""
        import os
        from types import SimpleNamespace

        if not self._user_message_function() and self._path() == "C":
            # the first-layer gradients on split operands walk the nodes in the order of their emptiness patterns
            L, H0 = len(edges_per_type), int(self._edge_type_mlps.kernels[0].shape[2])
            if self._first_layer_grads_split_ok(int(num_nodes), int(in_dim), L, H0) and _skip_empty_blocks(L, H0):
                return ops.G_PARTS_DEFAULT | ops.G_PART_DST_PATTERN
        if (self._user_message_function() or self._path() != "A" or self._use_target_state_as_input or self._compact_opt_in
                or ops.env("TFGNN_COMPACT_BUCKETS") == "1"):
            return ops.G_PARTS_DEFAULT
        shape = SimpleNamespace(num_edge_types=len(edges_per_type), num_edges=int(sum(edges_per_type)), num_nodes=int(num_nodes),
                                edges_per_type=tuple(int(c) for c in edges_per_type))
        if messages_per_edge(self, shape, in_dim, self._hidden_dim):
            parts = ops.G_PART_PLAN_TYPED | ops.G_PART_PLAN_NODE | ops.G_PART_EDGE_IDS
            # the backward pass keeps the bucket formulation; on split operands its input-gradient product walks the nodes in the
            # order of their by-source emptiness patterns (built with the batch, not in the middle of the step)
            if (self._f16x2_eligible(shape.num_nodes, in_dim, shape.num_edge_types, self._hidden_dim)
                    and _skip_empty_blocks(shape.num_edge_types, self._hidden_dim)):
                parts |= ops.G_PART_DST_PATTERN
            return parts
        if _skip_empty_blocks(shape.num_edge_types, in_dim):
            return ops.G_PART_PLAN_TYPED | ops.G_PART_DST_PATTERN
        return ops.G_PART_PLAN_TYPED

    # ---- which formulation ------------------------------------------------------------------
    def _path(self) -> str:
        linear = self._edge_type_mlps.num_layers == 1
        sum_like = self._aggregation_name in ("sum", "mean", "sqrt_n")
        if linear and sum_like and not self._pre_activation():
            return "A"
        if not self._use_target_state_as_input:
            return "B"
        return "C"

    def _pre_activation(self) -> bool:
        """activation applied per message before aggregation? (base class only, message_passing.py:169)"""
        return bool(self._message_activation_before_aggregation) and self._uses_base_aggregation()

    def _uses_base_aggregation(self) -> bool:
        return True  # RGIN / GGNN override _compute_new_node_embeddings and ignore the flag

    def _post_activation_name(self) -> Optional[str]:
        """activation fused after the aggregation of messages (message_passing.py:176-177)."""
        return None if self._pre_activation() else self._activation_name

    # ---- scale helpers ----------------------------------------------------------------------
    def _scales(self, g: "ops.Graph"):
        """(row_scale over rows (v,l) | None, edge weights by-dst | None, edge weights by-src | None,
        node scale [V] | None) for the configured normalisation / aggregation."""
        from ..graph_scales import graph_scales

        return graph_scales(g, bool(self._normalize_by_num_incoming), self._aggregation_name)

    # ---- forward ----------------------------------------------------------------------------
    def call(self, inputs: MessagePassingInput, training: bool = False):
        if self._user_message_function():
            if not self._uses_base_aggregation():
                raise NotImplementedError(
                    f"{type(self).__name__} overrides _message_function on a layer whose aggregation is not the base class's "
                    "(RGIN / GGNN): override call() and backward() as well"
                )
            return MessagePassing.call(self, inputs, training)
        X = inputs.node_embeddings
        V = X.shape[0]
        g = get_graph(inputs.adjacency_lists, V)
        if g.num_edge_types != self._num_edge_types:
            raise ValueError(
                f"layer was built for {self._num_edge_types} edge types, got {g.num_edge_types}"
            )
        agg, ctx = self._aggregate_messages(X, g, fuse_act=self._post_activation_name())
        ctx["graph"] = g
        ctx["X"] = X
        self._ctx = ctx
        return self._finish(agg, X, ctx, training)

    def _finish(self, agg, X, ctx, training):
        """base class: the activation was fused into the aggregation step."""
        ctx["out"] = agg
        return agg

    def _aggregate_messages(self, X, g, fuse_act):
        """-> (act?(aggregated messages) [V, H], ctx)"""
        path = self._path()
        if path == "A":
            return self._forward_A(X, g, fuse_act)
        if path == "B":
            return self._forward_B(X, g, fuse_act)
        return self._forward_C(X, g, fuse_act)

    # ---- general aggregation (max / activation before aggregation) ----------------------------------
    def _agg_general(self) -> bool:
        return self._aggregation_name == "max" or self._pre_activation()

    def _gather_messages(self, g, msgs, col, ew_d, node_scale, fuse_act, ctx):
        """act?( node_scale * REDUCE_{e -> v} pre_act?( w_e * msgs[col_e] ) ) over the by-dst node view; keeps what
        the backward pass of a max aggregation needs (the raw maxima)."""
        is_max = self._aggregation_name == "max"
        pre = self._activation_name if self._pre_activation() else None
        gelu_split = fuse_act == "gelu"
        separate_post = is_max and fuse_act is not None and not gelu_split
        out = ops.graph_gather(
            g, ops.VIEW_BY_DST_NODE, msgs, col=col, edge_weight=ew_d, row_scale=node_scale,
            reduce=ops.REDUCE_MAX if is_max else ops.REDUCE_SUM, pre_act=pre,
            post_act=None if (gelu_split or separate_post) else fuse_act,
        )
        if is_max:
            ctx["agg_max"] = out
        if gelu_split:
            ctx["pre"] = out
            return ops.activation_forward("gelu", out)
        if separate_post:
            return ops.activation_forward(fuse_act, out)
        return out

    def _message_grads(self, g, d_agg, ctx, msgs, msg_row, target, ew, node_scale, sel_col):
        """d(loss)/d(message of every edge) [E, H], in the edge order of msg_row / target / ew
        (csrc/edge.hip tfgnn_edge_aggregate_backward); sel_col maps by-dst positions to rows of that order."""
        pre = self._activation_name if self._pre_activation() else None
        d_agg = d_agg.contiguous()
        if self._aggregation_name == "max":
            sel = ops.edge_aggregate_backward(msgs, target, None, msg_row=msg_row, edge_weight=ew, pre_act=pre,
                                              reduce=ops.REDUCE_MAX, agg_max=ctx["agg_max"], phase=0)
            nsel = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, sel, col=sel_col)  # ties share the gradient evenly
            return ops.edge_aggregate_backward(msgs, target, d_agg, msg_row=msg_row, edge_weight=ew, pre_act=pre,
                                               reduce=ops.REDUCE_MAX, agg_max=ctx["agg_max"], num_selected=nsel)
        return ops.edge_aggregate_backward(msgs, target, d_agg, msg_row=msg_row, edge_weight=ew, node_scale=node_scale,
                                           pre_act=pre, reduce=ops.REDUCE_SUM)

    @staticmethod
    def _ident_e(g):
        ident = g._cache.get("ident_e")
        if ident is None or ident.numel() < g.num_edges + 1:
            ident = torch.arange(g.num_edges + 1, dtype=torch.int32, device=g.device)
            g._cache["ident_e"] = ident
        return ident

    # Buckets (node, type) that received no edge contribute nothing: 45 % of them are empty on an R-MAT
    # batch, and the dense multiply can run over the non-empty ones only (grouped GEMMs over compact
    # rows).  Measured on MI355X (profiles/r01e_compact_kernel_stats.csv): 45 % fewer FLOPs buy only
    # ~12 % on the forward GEMM because the per-relation groups have K = D (10 K-tiles) and the
    # workgroup prologue / store-burst epilogue dominates; with the extra combine passes the step time
    # is unchanged.  Kept opt-in (hyper-parameter "use_compact_buckets" or TFGNN_COMPACT_BUCKETS=1).
    SPARSE_BUCKET_THRESHOLD = 0.85

    def _use_compact_buckets(self, g) -> bool:
        import os

        if not (self._compact_opt_in or ops.env("TFGNN_COMPACT_BUCKETS") == "1"):
            return False
        if self._use_target_state_as_input or g.num_edge_types == 0 or g.num_edges == 0:
            return False
        nz = g.nonempty_offsets(False)[-1]
        return nz < self.SPARSE_BUCKET_THRESHOLD * g.num_nodes * g.num_edge_types

    def _forward_A_compact(self, X, g, fuse_act):
        """path A over the non-empty buckets only (type-major compact rows):
             A_c[c]  = scale * sum of source rows of bucket c          gather, compact output
             Y_c     = A_c[rows of type l] @ W_l  for every l          grouped MFMA GEMM
             out[v]  = act( sum over the non-empty buckets of v )       <= L rows per node"""
        row_scale, _, _, _ = self._scales(g)
        W = self._edge_type_mlps.kernels[0]  # [L, D, H]
        off_h = g.nonempty_offsets(False)
        Ac = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED_COMPACT, X, row_scale=row_scale)
        Yc = ops.gemm_grouped_rows(Ac, g.array(ops.G_NZ_OFF_BY_DST), off_h, W)
        gelu_split = fuse_act == "gelu"
        pre = ops.gather_reduce(
            g.array(ops.G_NZ_NODEPTR_BY_DST), g.array(ops.G_NZ_COL_BY_DST), Yc, post_act=None if gelu_split else fuse_act
        )
        ctx = {"path": "Ac", "fused_act": fuse_act}
        if gelu_split:
            ctx["pre"] = pre
            return ops.activation_forward("gelu", pre), ctx
        return pre, ctx

    def _backward_A_compact(self, d_agg, ctx):
        g, X = ctx["graph"], ctx["X"]
        mlps = self._edge_type_mlps
        W = mlps.kernels[0]  # [L, D, H]
        L = g.num_edge_types
        _, _, ew_s, _ = self._scales(g)
        off_h = g.nonempty_offsets(True)
        nz = off_h[-1]
        off_d = g.array(ops.G_NZ_OFF_BY_SRC)
        # G_c[c] = sum over the out-edges of bucket c = (u, l) of w_e * d_agg[target]
        Gc = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED_COMPACT, d_agg, edge_weight=ew_s)  # [nz, H]
        Zc = ops.gemm_grouped_rows(Gc, off_d, off_h, W, trans_b=True)  # [nz, D] = G_c,l @ W_l^T
        dX = ops.gather_reduce(g.array(ops.G_NZ_NODEPTR_BY_SRC), g.array(ops.G_NZ_COL_BY_SRC), Zc)
        ident = g._cache.get(("ident_nz", nz))
        if ident is None:
            ident = torch.arange(nz + 1, dtype=torch.int32, device=X.device)
            g._cache[("ident_nz", nz)] = ident
        Xc = ops.gather_reduce(ident, g.array(ops.G_NZ_NODE_BY_SRC), X)  # source states of the compact rows
        mlps.grads = [ops.gemm_grouped_k(Xc, Gc, off_d, off_h, L)]  # dW_l = X_c,l^T @ G_c,l
        mlps.publish_grads()
        return dX

    def _f16x2_eligible(self, V, D, L, H) -> bool:
        """path A without target states on shapes the split-operand kernels tile (include/tfgnn.h tfgnn_sp_gemm_*)."""
        def tiles(n):  # output widths the kernels tile: 320 / 256 / 128 columns
            return n % 128 == 0 or n % 320 == 0

        return (ops.get_gemm_mode() == ops.GEMM_F16X2 and not self._use_target_state_as_input and L > 0 and V > 0
                and D % 16 == 0 and D <= 512 and 32 <= H <= 512 and tiles(H) and tiles(D))

    def _forward_A(self, X, g, fuse_act):
        if self._use_compact_buckets(g):
            return self._forward_A_compact(X, g, fuse_act)
        V, D = X.shape
        L, H = g.num_edge_types, self._hidden_dim
        T = self._use_target_state_as_input
        row_scale, _, _, _ = self._scales(g)
        W = self._edge_type_mlps.kernels[0]  # [L, Din, H]
        Din = W.shape[1]
        if not T and messages_per_edge(self, g, D, H):
            # molecule-sized graphs have fewer edges than (node, type) buckets (QM9: 3.4 M vs 5.8 M): the bucket matrix
            # [V, L D] would be mostly zeros.  One message per EDGE instead - x_u W_l on rows of X read through the edge's
            # source index - summed per target by the node-view gather (the machinery of path C)
            _, ew_d, _, node_scale = self._scales(g)
            src_l, _, _, _, off, _ = self._original_order(g, ew_d)
            src_node = g._cache.get("orig_src_node")
            if src_node is None:
                src_node = torch.div(src_l, L, rounding_mode="floor").to(torch.int32)
                g._cache["orig_src_node"] = src_node
            msgs = torch.empty((g.num_edges, H), dtype=torch.float32, device=X.device)
            for l in range(L):
                if off[l + 1] > off[l]:
                    ops.gemm_gathered(X, src_node[off[l] : off[l + 1]], W[l], out=msgs[off[l] : off[l + 1]])
            ctx = {"path": "A", "A": None, "fused_act": fuse_act, "f16x2": self._f16x2_eligible(V, D, L, H)}
            return self._gather_messages(g, msgs, g.array(ops.G_EID_BY_DST), ew_d, node_scale, fuse_act, ctx), ctx
        if self._f16x2_eligible(V, D, L, H):
            # f16x2: the gather writes [A_0 | ... | A_{L-1}] directly as the split operand (one scale per (node, type)
            # bucket), the kernels are split once per value, the product only moves data and multiplies
            # most nodes of a typed graph receive edges of only some types: with the nodes in the order of their emptiness
            # pattern (Graph part DST_PATTERN) whole 128-row tiles of A have all-zero type blocks, which the product skips.
            # The product's epilogue writes its rows back in node order, so nothing downstream sees the order.
            skip = _skip_empty_blocks(L, Din)
            kmask = g.array(ops.G_PATTERN_TILEMASK_BY_DST) if skip else None
            rmap = g.array(ops.G_PATTERN_NODE_BY_DST) if skip else None
            # (a gather that leaves the zero rows of skipped blocks unwritten - 38 % of the operand's 154 MB at the benchmark
            #  batch - measured no gain in round 5, 2.370 vs 2.368 ms per step: the gather's time is its reads.  Removed.)
            view = ops.VIEW_BY_DST_TYPED_PATTERN if skip else ops.VIEW_BY_DST_TYPED
            gelu_split = fuse_act == "gelu"
            want_split = getattr(self, "_want_split_output", False) or getattr(self, "_always_split_output", False)
            drop = getattr(self, "_fused_output_dropout", None)  # (rate, seed) of the NEXT layer's input dropout (GNN stack)
            out_scale = 1.0
            # this layer's output is only ever read through that dropout: apply the mask in the product's epilogue and write the
            # dropped result in both forms (fp32 for the next gather, SP16 for its weight-gradient product)
            fuse_drop = drop is not None and not gelu_split and H in (128, 256, 320) and type(self)._finish is GNN_Edge_MLP._finish
            # (a split result: the consumer finds it with sp_rows_of)
            split_out = fuse_drop or (want_split and not gelu_split and H in (128, 256, 320))
            if ops.mp_entry_enabled() and Din == D:
                # round 6: gather + weight split + merged small passes + product in ONE library call (tfgnn_mp_forward) - the
                # kernels and arguments of the op-level route below, without Python between the launches
                pre, _ = ops.mp_forward(g, view, X, W, row_scale=row_scale, act=None if gelu_split else fuse_act,
                                        dropout=drop if fuse_drop else None, tile_kmask=kmask, row_map=rmap, want_split=split_out)
            else:
                A_sp = ops.graph_gather_sp(g, view, X, row_scale=row_scale, rows_per_operand_row=L, defer_combine=True)
                Wt_sp = ops.sp_weight_operand(W, "cols", lambda: ops.sp_split_cols(W.view(L * Din, H), defer=True))
                if fuse_drop:
                    pre, _ = ops.sp_gemm_nt_split(A_sp, Wt_sp, act=fuse_act, dropout=drop, tile_kmask=kmask, row_map=rmap)
                elif split_out:
                    pre, _ = ops.sp_gemm_nt_split(A_sp, Wt_sp, act=fuse_act, tile_kmask=kmask, row_map=rmap)
                else:
                    pre = ops.sp_gemm_nt(A_sp, Wt_sp, act=None if gelu_split else fuse_act, tile_kmask=kmask, row_map=rmap)
            if fuse_drop:
                self._fused_output_dropout_done = True
                out_scale = 1.0 - float(drop[0])
            ctx = {"path": "A", "A": None, "fused_act": fuse_act, "f16x2": True, "out_scale": out_scale}
            if gelu_split:
                ctx["pre"] = pre
                return ops.activation_forward("gelu", pre), ctx
            return pre, ctx
        A = torch.empty((V, L * Din), dtype=torch.float32, device=X.device)
        Arows = A.view(V * L, Din)
        ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, X, row_scale=row_scale, out=Arows[:, :D])
        if T:
            # sum_e s_e [x_u | x_v] W = (sum_e s_e x_u) W_s + (c * s) x_v W_t
            from ..graph_scales import target_multiplier

            k, ident_ptr, node_of_row = target_multiplier(g, row_scale)
            ops.gather_reduce(ident_ptr, node_of_row, X, edge_weight=k, out=Arows[:, D:])
        gelu_split = fuse_act == "gelu"
        act = None if gelu_split else fuse_act
        if ops.get_gemm_mode() != ops.GEMM_FP32:
            # the split-operand kernel stages K-contiguous operands fastest: hand it W^T ([H, L*Din], 1.6 MB copy)
            Wt = ops.sp_weight_operand(W, "transposed", lambda: ops.transpose_batched(W.view(L * Din, H)))  # once per weight value
            pre = ops.gemm(A, Wt, trans_b=True, act=act)
        else:
            pre = ops.gemm(A, W.view(L * Din, H), act=act)
        ctx = {"path": "A", "A": A, "fused_act": fuse_act}
        if gelu_split:
            ctx["pre"] = pre
            return ops.activation_forward("gelu", pre), ctx
        return pre, ctx

    def _backward_A_f16x2(self, d_agg, ctx, g, X, ew_s):
        """path A backward on split operands: the transposed gather writes G = [G_0 | ... | G_{L-1}] as an SP16 operand
        (one scale per (source, type) bucket) that serves both dX = G @ [W_0 | ... | W_{L-1}]^T (NT) and
        dW_l = X^T G_l (TN, where the per-row scales become per-k factors: tfgnn_sp_gemm_tn)."""
        V, D = X.shape
        L, H = g.num_edge_types, self._hidden_dim
        mlps = self._edge_type_mlps
        W = mlps.kernels[0]  # [L, D, H]
        import os

        overlap = ops.env("TFGNN_TN_OVERLAP", "0") == "1"
        if ops.mp_entry_enabled() and not overlap:
            # round 6: the whole pass in ONE library call (tfgnn_mp_backward): gather over the by-source buckets, the rows form of
            # the kernels when stale, the merged small passes, dX = G W^T with its epilogue, dW = X^T G - the op-level sequence below
            skip = {}
            if _skip_empty_blocks(L, H) and V > 0:
                node_at = g.array(ops.G_PATTERN_NODE_BY_SRC)
                skip = dict(tile_kmask=g.array(ops.G_PATTERN_TILEMASK_BY_SRC), a_rows=node_at, row_map=node_at)
            epi = getattr(self, "_out_epilogue", None)
            acc = getattr(self, "_dx_accumulate", None)
            kw = {}
            if acc is not None:
                kw = dict(out=acc[0], accumulate=True, out_mul=acc[1])
                self._dx_accumulate = None  # consumed
            elif epi is not None:
                kw = dict(out_mul=epi[0], act_grad=epi[1], want_split=bool(getattr(self, "_want_split_input_grad", False)) and D in (128, 256, 320))
                self._out_epilogue = None  # consumed
            X_sp = ops.sp_rows_of(X)  # written by the dropout kernel when X came out of one
            dX, _, dW = ops.mp_backward(g, d_agg.contiguous(), W, X_sp, edge_weight=ew_s, skip=skip, **kw)
            mlps.grads = [dW]
            mlps.publish_grads()
            return dX
        G_sp = ops.graph_gather_sp(g, ops.VIEW_BY_SRC_TYPED, d_agg.contiguous(), edge_weight=ew_s, rows_per_operand_row=L,
                                   defer_combine=True)
        # the weight gradient dW = X^T G is off the critical path of the backward pass: its two small passes (per-k factors,
        # split reduction) run on the library's second stream beside the big kernels around them
        tn = None
        if overlap:
            dW = torch.empty_like(W)
            tn = ops.SpGemmTnOverlapped(G_sp, ops.sp_rows_of(X), out=dW, scatter=(H, D * H, 1, H))  # factors: second stream
        Wh_sp = ops.sp_weight_operand(W, "rows", lambda: ops.sp_split_rows(W[0], segments=(H, D * H, L * H), defer=True))
        epi = getattr(self, "_out_epilogue", None)
        acc = getattr(self, "_dx_accumulate", None)
        # Round 5: the by-source buckets are as sparse as the by-target ones (45 % empty on the R-MAT batch).  G stays in node
        # order - the weight-gradient product below pairs its rows with X's -, the input-gradient product reads its rows in the
        # order of the by-source emptiness patterns (a_rows), skips the all-zero type blocks of a row tile and writes node order
        # again (row_map; mask, saved activation and dropout index follow it).  Skipped terms are exact zeros: bit-equal.
        skip = {}
        if _skip_empty_blocks(L, H) and V > 0:
            node_at = g.array(ops.G_PATTERN_NODE_BY_SRC)
            skip = dict(tile_kmask=g.array(ops.G_PATTERN_TILEMASK_BY_SRC), a_rows=node_at, row_map=node_at)
        if acc is not None:
            # a subclass (GGNN) already holds other terms of d(node_embeddings): add this one in the product's epilogue
            dX = ops.sp_gemm_nt(G_sp, Wh_sp, out=acc[0], accumulate=True, out_mul=acc[1], **skip)
            self._dx_accumulate = None  # consumed
        elif epi is not None:
            if getattr(self, "_want_split_input_grad", False) and D in (128, 256, 320):
                dX, _ = ops.sp_gemm_nt_split(G_sp, Wh_sp, out_mul=epi[0], act_grad=epi[1], **skip)
            else:
                dX = ops.sp_gemm_nt(G_sp, Wh_sp, out_mul=epi[0], act_grad=epi[1], **skip)
            self._out_epilogue = None  # consumed
        else:
            dX = ops.sp_gemm_nt(G_sp, Wh_sp, **skip)
        if tn is not None:
            tn.product()
            tn.finish()  # the reduction runs beside the next layer's gather; GNN.backward joins the second stream at its end
            if not getattr(self, "_defer_aux_join", False):
                ops.join_aux_stream()  # stand-alone layer call: the gradients are complete when backward() returns
        else:
            X_sp = ops.sp_rows_of(X)  # written by the dropout kernel when X came out of one
            dW = torch.empty_like(W)
            # element ((l, h), d) -> dW[l, d, h].  (The split reduction stays right behind the product: deferred into a later
            # merged launch - ops.sp_gemm_tn(defer_reduce=True) - it finds its 41 MB of partials evicted and takes 45 us
            # instead of 14.)
            ops.sp_gemm_tn(G_sp, X_sp, out=dW, scatter=(H, D * H, 1, H))
        mlps.grads = [dW]
        mlps.publish_grads()
        return dX

    def _mlp_all_types(self, X, L, ctx, mlps=None, key="mlp_acts"):
        """Y[:, l, :] = MLP_l(X) for all nodes -> [V, L, H]; hidden activations saved in ctx[key]."""
        V = X.shape[0]
        mlps = self._edge_type_mlps if mlps is None else mlps
        acts = []
        cur = None
        for j, W in enumerate(mlps.kernels):
            out_j = W.shape[2]
            Z = torch.empty((V, L, out_j), dtype=torch.float32, device=X.device)
            last = j == mlps.num_layers - 1
            for l in range(L):
                inp = X if j == 0 else cur[:, l, :]
                ops.gemm(inp, W[l], act=None if last else "relu", out=Z[:, l, :])
            acts.append(Z)
            cur = Z
        ctx[key] = acts
        return cur

    @staticmethod
    def _mlp_all_types_backward(mlps, X, acts, dY, dX, accumulate):
        """Gradient of _mlp_all_types: dY [V, L, out] -> kernel gradients (mlps.grads) and dX (+= if accumulate)."""
        L = mlps.L
        grads = [torch.empty_like(W) for W in mlps.kernels]
        dcur = dY
        for j in range(mlps.num_layers - 1, -1, -1):
            W = mlps.kernels[j]
            inp_all = None if j == 0 else acts[j - 1]
            if j > 0:
                dprev = torch.empty_like(inp_all)
            for l in range(L):
                inp = X if j == 0 else inp_all[:, l, :]
                ops.gemm(inp, dcur[:, l, :], trans_a=True, out=grads[j][l])
                if j > 0:
                    # hidden layers use relu ([ext] dpu_utils MLP default activation): its derivative rides in the
                    # epilogue of the product where the active kernel has one
                    _relu_input_grad(dcur[:, l, :], W[l], inp, dprev[:, l, :])
                else:
                    ops.gemm(dcur[:, l, :], W[l], trans_b=True, out=dX, accumulate=accumulate or l > 0)
            if j > 0:
                dcur = dprev
        mlps.grads = grads
        if L == 0 and not accumulate:
            dX.zero_()

    # Path B evaluates MLP_l(x_u) for every (node, type) pair that is the SOURCE of at least one edge.
    # When fewer than this fraction of the V*L pairs are (cfg-5: 40 edge types, ~13 %), the MLPs run as
    # grouped GEMMs over the compact rows only.
    SPARSE_SOURCE_THRESHOLD = 0.6

    # ---- path Bc on split operands (round 5; BASELINE configs[4]) ------------------------------------------------------------
    # The per-relation MLPs over the non-empty (source, type) rows as grouped products on SP16 operands (3 piece products per
    # fp32 product; the bf16x3 grouped kernels spend 6): the first layer reads the node states through the row -> node index
    # (no expanded copy of X for the forward pass), a hidden layer's product writes its relu output ALSO as the operand of the
    # next product, the input-gradient products carry relu' of the saved activations and write the operand of the next
    # gradient product; the weight gradients are one split-operand TN product per relation over that relation's row range.
    GROUPED_SPLIT_MIN_ROWS = 4096  # (tests lower it: below this the grouped bf16x3 kernels are as good)

    def _grouped_split_ok(self, X, g) -> bool:
        import os

        if ops.get_gemm_mode() != ops.GEMM_F16X2 or ops.env("TFGNN_GROUPED_F16X2", "1") == "0":
            return False
        dims = [X.shape[1]] + [int(W.shape[2]) for W in self._edge_type_mlps.kernels]
        # every width a column tile of the split-operand product and a multiple of its scale blocks; rows x bytes below 4 GB
        return (all(d % 128 == 0 and d <= 1024 for d in dims) and X.shape[0] * X.shape[1] * 4 < (1 << 32) - 65536
                and g.nonempty_offsets(True)[-1] >= self.GROUPED_SPLIT_MIN_ROWS)

    @staticmethod
    def _row_groups(g):
        rg = g._cache.get("row_groups_by_src")
        if rg is None:
            rg = ops.RowGroups(g.nonempty_offsets(True), g.device)
            g._cache["row_groups_by_src"] = rg
        return rg

    def _grouped_tn_route(self, g) -> bool:
        """kernel gradients of the compact-row MLPs on the grouped two-factor TN product?  Not after the stack's guard policy
        handed them back (``_grouped_tn_split_ok``), and not for more K ranges than one launch takes (~10^6 compact rows)."""
        return bool(getattr(self, "_grouped_tn_split_ok", True)) and self._row_groups(g).tn_tables()[2] <= ops.TN_GROUPED_MAX_RANGES

    @staticmethod
    def _stacked_transposed_operand(W):
        """[L, in, out] kernels -> SP16 [out, L * in]: column block l is W_l^T, relation l's [N, K] operand of the forward product
        (ONE split launch for the whole stack; a row's scale is shared by the relations)"""
        L, K, N = W.shape
        return ops.sp_split_cols(W.view(L * K, N), defer=True)

    def _forward_B_compact_split(self, X, g, fuse_act):
        mlps = self._edge_type_mlps
        _, ew_d, _, node_scale = self._scales(g)
        groups = self._row_groups(g)
        node_of = g.array(ops.G_NZ_NODE_BY_SRC)
        x_sp = ops.sp_rows_of(X)
        acts, cur_sp, cur32 = [], x_sp, None
        for j, W in enumerate(mlps.kernels):
            last = j == mlps.num_layers - 1
            wt = ops.sp_weight_operand(W, "grouped_cols", lambda W=W: self._stacked_transposed_operand(W))
            cur32, nxt_sp = ops.sp_gemm_nt_grouped(cur_sp, wt, groups, a_rows=node_of if j == 0 else None,
                                                   act=None if last else "relu", want_split=not last, b_column_blocks=True)
            acts.append(cur32)
            cur_sp = nxt_sp
        colc = g._cache.get("compact_src_col_by_dst")
        if colc is None:
            colc = g.array(ops.G_NZ_CPOS_BY_SRC)[g.array(ops.G_COLL_BY_DST).long()].contiguous()
            g._cache["compact_src_col_by_dst"] = colc
        ctx = {"path": "Bc", "fused_act": fuse_act, "Xc": None, "x_sp": x_sp, "mlp_acts": acts, "colc": colc, "grouped_split": True}
        return self._gather_messages(g, cur32, colc, ew_d, node_scale, fuse_act, ctx), ctx

    def _backward_B_compact_split(self, dcur, ctx, g):
        """d(MLP outputs) [nz, H] -> d(compact inputs) [nz, D]; fills the kernel gradients (see _forward_B_compact_split).
        The kernel gradients dW_l = inp_l^T d_l are split-operand TN products while ``_grouped_tn_split_ok``: that product
        applies one combined fp16 factor per operand row pair, and its spread guard trips when a relation's rows are spread over
        more than 2^20 in scale (un-normalised RGIN sums grow by the degree per layer: the arxiv-rgin workload reaches 2^25) -
        the stack then hands these gradients to the exact bf16x3 grouped kernel (GNN.backward, like its Dense products); the
        row-wise products above have per-row scales and no such limit."""
        mlps = self._edge_type_mlps
        groups = self._row_groups(g)
        off = groups.offsets
        acts = ctx["mlp_acts"]
        tn_split = self._grouped_tn_route(g)
        if isinstance(dcur, ops.SplitOperand):  # written by the compact by-source gather
            d_sp, d32 = dcur, None
        else:
            d_sp, d32 = ops.sp_split_rows(dcur), dcur
        assert tn_split or d32 is not None
        grads = [None] * mlps.num_layers
        dcur32 = None
        for j in range(mlps.num_layers - 1, -1, -1):
            W = mlps.kernels[j]  # [L, in, out]
            if not tn_split:
                if j > 0:
                    inp32 = acts[j - 1]
                else:  # the expanded node states, as the bf16x3 path keeps them
                    nz = groups.num_rows
                    ident = g._cache.get(("ident_nz", nz))
                    if ident is None:
                        ident = torch.arange(nz + 1, dtype=torch.int32, device=dcur.device)
                        g._cache[("ident_nz", nz)] = ident
                    inp32 = ops.gather_reduce(ident, g.array(ops.G_NZ_NODE_BY_SRC), ctx["X"])
                grads[j] = ops.gemm_grouped_k(inp32, d32, g.array(ops.G_NZ_OFF_BY_SRC), off, groups.num_groups)
            else:
                self._grouped_tn_used = True  # (what the stack's guard policy demotes if the spread guard trips: GNN.backward)
                inp_sp = ops.sp_rows_of(acts[j - 1]) if j > 0 else ops.sp_gather_rows(ctx["x_sp"], g.array(ops.G_NZ_NODE_BY_SRC))
                # The operand with per-block scales is the product's left one: the layer input for j > 0 (a product wrote it:
                # blocks of a column tile), the gradient for j = 0 (then the transpose is stored)
                dW = torch.empty_like(W)
                if inp_sp.scale_block != inp_sp.cols and d_sp.scale_block != d_sp.cols:
                    # both operands written by grouped products wider than one column tile (two or more hidden layers of
                    # width >= 384: one scale per row AND column tile on either side): the product takes such
                    # scales on its left operand only - the gradient is split again with one scale per row
                    d_sp = ops.sp_split_rows(d32)
                left_is_input = inp_sp.scale_block != inp_sp.cols or d_sp.scale_block == d_sp.cols
                # all relations in ONE launch of the two-factor ("wide range") product: both operands' rows are un-normalised
                # sums, the guard looks at each operand's own spread inside a K range of <= 2016 rows (a launch per relation
                # cost as much as the exact grouped kernel: 40 small products, 40 reductions)
                if left_is_input:
                    ops.sp_gemm_tn_grouped(inp_sp, d_sp, groups, dW)
                else:  # (d_l^T inp_l)^T: element (m, n) of the product is dW_l[n, m]
                    ops.sp_gemm_tn_grouped(d_sp, inp_sp, groups, dW, transposed=True)
                grads[j] = dW
            wr = ops.sp_weight_operand(W, "grouped_rows", lambda W=W: ops.sp_split_rows(W.view(W.shape[0] * W.shape[1], W.shape[2])))
            # (the fp32 copy of a hidden layer's gradient is needed by the exact route and by the re-split above: when the
            #  layer below is itself a hidden layer and this product's split result carries per-tile scales)
            resplit_below = j > 1 and ops.sp_tile_width(int(W.shape[1])) != int(W.shape[1])
            dcur32, d_sp = ops.sp_gemm_nt_grouped(d_sp, wr, groups, act_grad=("relu", acts[j - 1]) if j > 0 else None,
                                                  want_fp32=(j == 0 or not tn_split or resplit_below), want_split=(j > 0))
            d32 = dcur32
        mlps.grads = grads
        mlps.publish_grads()
        return dcur32

    def _forward_B_compact(self, X, g, fuse_act):
        if self._grouped_split_ok(X, g):
            return self._forward_B_compact_split(X, g, fuse_act)
        L, H = g.num_edge_types, self._hidden_dim
        mlps = self._edge_type_mlps
        _, ew_d, _, node_scale = self._scales(g)
        off_h = g.nonempty_offsets(True)
        nz = off_h[-1]
        off_dev = g.array(ops.G_NZ_OFF_BY_SRC)
        ident = g._cache.get(("ident_nz", nz))
        if ident is None:
            ident = torch.arange(nz + 1, dtype=torch.int32, device=X.device)
            g._cache[("ident_nz", nz)] = ident
        Xc = ops.gather_reduce(ident, g.array(ops.G_NZ_NODE_BY_SRC), X)  # states of the non-empty (source, type) pairs
        acts, cur = [], Xc
        for j, W in enumerate(mlps.kernels):
            last = j == mlps.num_layers - 1
            cur = ops.gemm_grouped_rows(cur, off_dev, off_h, W, act=None if last else "relu")
            acts.append(cur)
        # column of every bucketed edge (by-dst order) in the compact table: cpos_src[source * L + type]
        colc = g._cache.get("compact_src_col_by_dst")
        if colc is None:
            colc = g.array(ops.G_NZ_CPOS_BY_SRC)[g.array(ops.G_COLL_BY_DST).long()].contiguous()
            g._cache["compact_src_col_by_dst"] = colc
        ctx = {"path": "Bc", "fused_act": fuse_act, "Xc": Xc, "mlp_acts": acts, "colc": colc}
        return self._gather_messages(g, cur, colc, ew_d, node_scale, fuse_act, ctx), ctx

    def _backward_B_compact(self, d_agg, ctx):
        g = ctx["graph"]
        L = g.num_edge_types
        mlps = self._edge_type_mlps
        _, _, ew_s, _ = self._scales(g)
        off_h = g.nonempty_offsets(True)
        off_dev = g.array(ops.G_NZ_OFF_BY_SRC)
        acts = ctx["mlp_acts"]
        if self._agg_general():
            _, ew_d, _, node_scale = self._scales(g)
            dM = self._message_grads(g, d_agg, ctx, acts[-1], ctx["colc"], g.array(ops.G_TARGET_BY_DST), ew_d, node_scale,
                                     self._ident_e(g)[: g.num_edges])
            dcur = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED_COMPACT, dM, col=g.array(ops.G_SRC2DST_POS))
        elif ctx.get("grouped_split") and self._grouped_tn_route(g) and d_agg.shape[1] % 16 == 0:
            # every consumer of d(MLP outputs) takes the split form: the gather writes it (no fp32 [nz, H], no split pass)
            dcur = ops.graph_gather_sp(g, ops.VIEW_BY_SRC_TYPED_COMPACT, d_agg.contiguous(), edge_weight=ew_s)
        else:
            dcur = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED_COMPACT, d_agg, edge_weight=ew_s)  # d(MLP outputs) [nz, H]
        if ctx.get("grouped_split"):
            dxc = self._backward_B_compact_split(dcur, ctx, g)
            return ops.gather_reduce(g.array(ops.G_NZ_NODEPTR_BY_SRC), g.array(ops.G_NZ_COL_BY_SRC), dxc)
        grads = [None] * mlps.num_layers
        for j in range(mlps.num_layers - 1, -1, -1):
            inp = ctx["Xc"] if j == 0 else acts[j - 1]
            grads[j] = ops.gemm_grouped_k(inp, dcur, off_dev, off_h, L)
            # (hidden layers: relu' of the layer below rides in the product's epilogue)
            dcur = ops.gemm_grouped_rows(dcur, off_dev, off_h, mlps.kernels[j], trans_b=True,
                                         act_grad=("relu", inp) if j > 0 else None)
        mlps.grads = grads
        mlps.publish_grads()
        # dX[u] = sum over the non-empty (u, l) pairs
        return ops.gather_reduce(g.array(ops.G_NZ_NODEPTR_BY_SRC), g.array(ops.G_NZ_COL_BY_SRC), dcur)

    def _forward_B(self, X, g, fuse_act):
        V = X.shape[0]
        L, H = g.num_edge_types, self._hidden_dim
        if L > 0 and g.num_edges > 0 and g.nonempty_offsets(True)[-1] < self.SPARSE_SOURCE_THRESHOLD * V * L:
            return self._forward_B_compact(X, g, fuse_act)
        _, ew_d, _, node_scale = self._scales(g)
        ctx = {"path": "B", "fused_act": fuse_act}
        Y = self._mlp_all_types(X, L, ctx)
        return self._gather_messages(g, Y.view(V * L, H), None, ew_d, node_scale, fuse_act, ctx), ctx

    def _original_order(self, g, ew_d):
        """index arrays in concatenated-adjacency-list order (type-contiguous), cached on the Graph."""
        from ... import _lib

        key = ("orig", None if ew_d is None else ew_d.data_ptr())
        cached = g._cache.get(key)
        if cached is None:
            E, dev = g.num_edges, g.device
            src_l = torch.empty(E, dtype=torch.int32, device=dev)
            tgt_l = torch.empty(E, dtype=torch.int32, device=dev)
            tgt_node = torch.empty(E, dtype=torch.int32, device=dev)
            w = torch.empty(E, dtype=torch.float32, device=dev) if ew_d is not None else None
            g.ensure(ops.G_PART_EDGE_IDS)
            _lib.check(
                _lib.load().tfgnn_graph_original_order(
                    g._h, ops._ptr(ew_d), ops._ptr(src_l), ops._ptr(tgt_l), ops._ptr(tgt_node), ops._ptr(w), ops._stream()
                )
            )
            counts = g.edges_per_type  # edges of type l occupy [off[l], off[l+1]) in this order (host-side sizes: no sync)
            off = [0]
            for c in counts:
                off.append(off[-1] + int(c))
            cached = (src_l, tgt_l, tgt_node, w, off, torch.arange(E + 1, dtype=torch.int32, device=dev))
            g._cache[key] = cached
        return cached

    def _edge_messages_C(self, X, g, ew_d):
        """per-edge MLP outputs [E, H] in edge-list order (target states + hidden layers); see csrc/edge.hip."""
        from ... import _lib

        V, D = X.shape
        L, E = g.num_edge_types, g.num_edges
        mlps = self._edge_type_mlps
        src_l, tgt_l, tgt_node, _, off, _ = self._original_order(g, ew_d)
        H0 = mlps.kernels[0].shape[2]
        first_act = "relu" if mlps.num_layers > 1 else None
        if self._first_layer_per_edge(g, D, H0, off):
            # few edges per (node, type) pair (molecules: E < V L): the first layer runs once per EDGE on rows of X read
            # through the edge's source / target index - x_u W_s, then act(. + x_v W_t) - instead of once per (node, type)
            # pair followed by a pass that adds the two gathered rows
            src_node = g._cache.get("orig_src_node")
            if src_node is None:
                src_node = torch.div(src_l, L, rounding_mode="floor").to(torch.int32)
                g._cache["orig_src_node"] = src_node
            W0 = mlps.kernels[0]  # [L, 2D, H0]
            Z = torch.empty((E, H0), dtype=torch.float32, device=X.device)
            for l in range(L):
                if off[l + 1] > off[l]:
                    sl = slice(off[l], off[l + 1])
                    ops.gemm_gathered(X, src_node[sl], W0[l, :D], out=Z[sl])
                    ops.gemm_gathered(X, tgt_node[sl], W0[l, D:], out=Z[sl], act=first_act, accumulate="before")
            return self._edge_hidden_layers_C(Z, off, L)
        Wh = ops.permute_021(mlps.kernels[0])  # [2D, L, H0]
        P = ops.gemm(X, Wh[:D].view(D, L * H0))  # x_u W_s for every (node, type)
        Q = ops.gemm(X, Wh[D:].view(D, L * H0))  # x_v W_t
        Z = torch.empty((E, H0), dtype=torch.float32, device=X.device)
        # a single Dense layer has no activation (dpu_utils MLP [ext]: the final layer is linear)
        _lib.check(
            _lib.load().tfgnn_edge_pair_combine(
                ops._ptr(src_l), ops._ptr(tgt_l), ops._ptr(P), ops._ptr(Q), E, H0, ops.act_id(first_act), ops._ptr(Z),
                ops._stream(),
            )
        )
        return self._edge_hidden_layers_C(Z, off, L)

    def _edge_hidden_layers_C(self, Z, off, L):
        """layers 1 .. of the per-edge MLPs on the first layer's activations Z [E, H0] (edge-list order) -> (outputs, acts)."""
        mlps = self._edge_type_mlps
        E = Z.shape[0]
        acts = [Z]
        cur = Z
        for j in range(1, mlps.num_layers):
            W = mlps.kernels[j]
            last = j == mlps.num_layers - 1
            nxt = torch.empty((E, W.shape[2]), dtype=torch.float32, device=Z.device)
            for l in range(L):
                if off[l + 1] > off[l]:
                    ops.gemm(cur[off[l] : off[l + 1]], W[l], act=None if last else "relu", out=nxt[off[l] : off[l + 1]])
            acts.append(nxt)
            cur = nxt
        return cur, acts

    def _first_layer_per_edge(self, g, D, H0, off) -> bool:
        """the first layer of path C once per edge (two gathered products)?  The same question as ``messages_per_edge`` with the
        hidden width as the output width (one predicate, asked of the library)."""
        if g.num_edges == 0 or g.num_edges >= g.num_nodes * g.num_edge_types or ops.get_gemm_mode() == ops.GEMM_FP32:
            return False
        lib = _lib.load()
        forced = PER_EDGE_MIN_ROWS < 65536
        for l in range(g.num_edge_types):
            c = off[l + 1] - off[l]
            if c == 0:
                continue
            if forced:
                if c < PER_EDGE_MIN_ROWS or D not in (64, 96, 128) or H0 % 128 != 0:
                    return False
            elif not lib.tfgnn_gemm_gathered_supported(int(c), int(H0), int(D), int(D), int(g.num_nodes)):
                return False
        return True

    def _aggregate_nothing(self, V, X, fuse_act, ctx):
        """aggregation over zero edges: zeros (sum-like) / the float minimum (max), then the activation."""
        out = torch.zeros((V, self._hidden_dim), dtype=torch.float32, device=X.device)
        if self._aggregation_name == "max":
            out.fill_(torch.finfo(torch.float32).min)
        if fuse_act == "gelu":
            ctx["pre"] = out
            return ops.activation_forward("gelu", out)
        if fuse_act is not None:
            out = ops.activation_forward(fuse_act, out)
        return out

    def _forward_C(self, X, g, fuse_act):
        V = X.shape[0]
        _, ew_d, _, node_scale = self._scales(g)
        cur, acts = self._edge_messages_C(X, g, ew_d)
        H0 = self._edge_type_mlps.kernels[0].shape[2]
        ctx = {"path": "C", "fused_act": fuse_act, "edge_acts": acts, "P_shape": (V, g.num_edge_types * H0)}
        if g.num_edges == 0:
            return self._aggregate_nothing(V, X, fuse_act, ctx), ctx
        return self._gather_messages(g, cur, g.array(ops.G_EID_BY_DST), ew_d, node_scale, fuse_act, ctx), ctx

    def _backward_C(self, d_agg, ctx, dcur=None):
        """``dcur``: d(per-edge MLP outputs) [E, H] in edge-list order when the caller already has it (GNN_FiLM)."""
        g, X = ctx["graph"], ctx["X"]
        V, D = X.shape
        L, E = g.num_edge_types, g.num_edges
        mlps = self._edge_type_mlps
        if E == 0:
            mlps.grads = [torch.zeros_like(W) for W in mlps.kernels]
            mlps.publish_grads()
            return torch.zeros_like(X)
        _, ew_d, _, node_scale = self._scales(g)
        general = self._agg_general()
        acts = ctx["edge_acts"]
        if dcur is not None:
            off = self._original_order(g, ew_d)[4]
        elif general:
            # per-edge messages are acts[-1] in edge-list order; the normalisation weight in that order
            src_l, tgt_l, tgt_node, w_orig, off, ident = self._original_order(g, ew_d)
            dcur = self._message_grads(g, d_agg, ctx, acts[-1], None, tgt_node, w_orig, node_scale, g.array(ops.G_EID_BY_DST))
        else:
            # per-edge weight of the aggregation in by-dst order, including the mean / sqrt_n factor
            if node_scale is not None:
                m_e = ops.gather_reduce(self._ident_e(g), g.array(ops.G_TARGET_BY_DST), node_scale.view(-1, 1)).view(-1)
                w_full = m_e if ew_d is None else ops.mul(m_e, ew_d)
            else:
                w_full = ew_d
            src_l, tgt_l, tgt_node, w_orig, off, ident = self._original_order(g, w_full)
            # d messages, in edge-list order: dM[e] = w_e * d_agg[target_e]
            dcur = ops.gather_reduce(ident, tgt_node, d_agg, edge_weight=w_orig)
        grads = [None] * mlps.num_layers
        for j in range(mlps.num_layers - 1, 0, -1):
            W = mlps.kernels[j]
            inp = acts[j - 1]
            gW = torch.zeros_like(W)
            dprev = torch.empty_like(inp)
            for l in range(L):
                if off[l + 1] > off[l]:
                    sl = slice(off[l], off[l + 1])
                    ops.gemm(inp[sl], dcur[sl], trans_a=True, out=gW[l])
                    _relu_input_grad(dcur[sl], W[l], inp[sl], dprev[sl])  # hidden layers are relu (dpu_utils MLP)
            grads[j] = gW
            dcur = dprev
        # first layer: z0[e] = relu(P[(src,l)] + Q[(tgt,l)]); dcur is d(P+Q) per edge
        H0 = mlps.kernels[0].shape[2]
        if self._first_layer_grads_split_ok(V, D, L, H0):
            grads[0], dX = self._backward_C_first_layer_split(dcur, g, X, L, H0)
            mlps.grads = grads
            mlps.publish_grads()
            return dX
        dP = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, dcur, col=g.array(ops.G_EID_BY_SRC)).view(V, L * H0)
        dQ = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, dcur, col=g.array(ops.G_EID_BY_DST)).view(V, L * H0)
        Wh = ops.permute_021(mlps.kernels[0])  # [2D, L, H0]
        dWh = torch.empty_like(Wh)
        ops.gemm(X, dP, trans_a=True, out=dWh[:D].view(D, L * H0))
        ops.gemm(X, dQ, trans_a=True, out=dWh[D:].view(D, L * H0))
        epi = getattr(self, "_out_epilogue", None)
        if epi is not None:
            # the factors of the next backward step (dropout mask, activation derivative of the layer below) distribute over
            # the two terms: both products apply them in their epilogues, the second one adds into the first's result
            dX = ops.gemm_grad(dP, Wh[:D].view(D, L * H0), trans_b=True, out_mul=epi[0], act_grad=epi[1])
            dX = ops.gemm_grad(dQ, Wh[D:].view(D, L * H0), trans_b=True, out=dX, accumulate=True, out_mul=epi[0], act_grad=epi[1])
            self._out_epilogue = None  # consumed
        else:
            dX = ops.gemm(dP, Wh[:D].view(D, L * H0), trans_b=True)
            ops.gemm(dQ, Wh[D:].view(D, L * H0), trans_b=True, out=dX, accumulate=True)
        grads[0] = ops.permute_021(dWh)
        mlps.grads = grads
        mlps.publish_grads()
        return dX

    def _first_layer_grads_split_ok(self, V, D, L, H0) -> bool:
        """path C, gradients of the first MLP layer on split operands?  (widths the split-operand kernels tile; the stack's guard
        policy may have handed these weight gradients back to the exact kernels: GNN._demote_fragile_weight_gradients)"""
        import os

        def tiles(n):
            return n % 128 == 0 or n % 320 == 0

        return (ops.get_gemm_mode() == ops.GEMM_F16X2 and ops.env("TFGNN_EDGE_FIRST_LAYER_F16X2", "1") == "1"
                and getattr(self, "_grouped_tn_split_ok", True) and V > 0 and L > 0 and H0 % 16 == 0 and H0 <= 512
                and 32 <= D <= 512 and tiles(D) and tiles(H0))

    def _backward_C_first_layer_split(self, dcur, g, X, L, H0):
        """d(first-layer pre-activations) per edge [E, H0] -> (d kernels[0] [L, 2D, H0], dX [V, D]) on split operands (f16x2), the
        two halves of the layer - source states, target states - alike (as _backward_A_f16x2 does for linear messages):
          G = [G_0 | .. | G_{L-1}], G_l[u] = sum of dcur over the edges of type l that leave (enter) u: the typed gather writes
              it as an SP16 operand, one scale per (node, type) bucket;
          dX += G @ [W_0 | .. | W_{L-1}]^T   split-operand NT product over the nodes in the order of their emptiness patterns
              (all-zero type blocks of a row tile skipped), the factors of the next backward step in its epilogue;
          dW_l = X^T G_l                      the two-factor TN product (both operands are un-normalised sums).
        Replaces two fp32 typed gathers + four bf16x3 products over [V, L H0] (QM9-sized batches: 2.4 -> 1.9 ms per half)."""
        V, D = X.shape
        W0 = self._edge_type_mlps.kernels[0]  # [L, 2D, H0]
        X_sp = ops.sp_rows_of(X)
        gW0 = torch.empty_like(W0)
        epi = getattr(self, "_out_epilogue", None)
        kw = dict(out_mul=epi[0], act_grad=epi[1]) if epi is not None else {}
        self._grouped_tn_used = True  # (what the stack's guard policy demotes if the spread guard trips: GNN.backward)
        dX = None
        halves = ((ops.VIEW_BY_SRC_TYPED, ops.G_EID_BY_SRC, ops.G_PATTERN_NODE_BY_SRC, ops.G_PATTERN_TILEMASK_BY_SRC, 0),
                  (ops.VIEW_BY_DST_TYPED, ops.G_EID_BY_DST, ops.G_PATTERN_NODE_BY_DST, ops.G_PATTERN_TILEMASK_BY_DST, D))
        for view, eid, pat_node, pat_mask, d0 in halves:
            G_sp = ops.graph_gather_sp(g, view, dcur, col=g.array(eid), rows_per_operand_row=L, defer_combine=True)
            Wh_sp = ops.sp_weight_operand(W0, f"rows_half{d0}", lambda d0=d0: ops.sp_split_rows(
                W0[0, d0:d0 + D], segments=(H0, 2 * D * H0, L * H0), defer=True))  # row d = [W_0[d0 + d, :] | W_1[d0 + d, :] | ..]
            skip = {}
            if _skip_empty_blocks(L, H0):
                node_at = g.array(pat_node)
                skip = dict(tile_kmask=g.array(pat_mask), a_rows=node_at, row_map=node_at)
            if dX is None:
                dX = ops.sp_gemm_nt(G_sp, Wh_sp, **kw, **skip)
            else:  # the factors distribute over the two terms: the second product adds into the first's result
                ops.sp_gemm_nt(G_sp, Wh_sp, out=dX, accumulate=True, **kw, **skip)
            dWh = torch.empty((L, D, H0), dtype=torch.float32, device=X.device)
            ops.sp_gemm_tn(G_sp, X_sp, out=dWh, scatter=(H0, D * H0, 1, H0), wide=True)  # element ((l, h), d) -> dWh[l, d, h]
            gW0[:, d0:d0 + D].copy_(dWh)
        self._out_epilogue = None  # consumed
        return gW0, dX

    # ---- backward ---------------------------------------------------------------------------
    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        if ctx.get("generic"):  # the forward pass ran a user message function on the generic path
            return MessagePassing.backward(self, grad_output)
        d_agg = self._backward_finish(grad_output, ctx)
        return self._backward_messages(d_agg, ctx)

    def _plain_base_backward(self) -> bool:
        cls = type(self)
        return cls.backward is GNN_Edge_MLP.backward and cls._backward_finish is GNN_Edge_MLP._backward_finish

    def activation_backward_spec(self):
        ctx = self._ctx
        if ctx is None or not self._plain_base_backward() or ctx.get("fused_act") is None:
            return None
        act = ctx["fused_act"]
        # (third entry: the saved output is a DROPPED one - the next layer's dropout ran in this layer's product epilogue -
        # and carries 1 / (1 - rate) where kept: the derivative is taken at saved * scale)
        return act, (ctx["pre"] if act == "gelu" else ctx["out"]), (1.0 if act == "gelu" else ctx.get("out_scale", 1.0))

    def recomputes_input_dropout(self, num_nodes: int, in_dim: int, num_edge_types: int) -> bool:
        """path A on split operands: the mask rides in the epilogue of dX = G W^T (tfgnn_sp_gemm_nt_dropout recomputes it)"""
        if (not self._plain_base_backward() or self._user_message_function() or self._path() != "A" or self._use_target_state_as_input
                or not self._f16x2_eligible(num_nodes, in_dim, num_edge_types, self._hidden_dim)):
            return False
        return True

    def backward_with_epilogue(self, grad_output, grad_is_pre_activation=False, out_mul=None, out_act_grad=None):
        if not self._plain_base_backward() or (self._ctx is not None and self._ctx.get("generic")):
            return super().backward_with_epilogue(grad_output, grad_is_pre_activation, out_mul, out_act_grad)
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        d_agg = grad_output if grad_is_pre_activation else self._backward_finish(grad_output, ctx)
        self._out_epilogue = (out_mul, out_act_grad)
        try:
            dX = self._backward_messages(d_agg, ctx)
            if self._out_epilogue is not None:  # the path taken had no GEMM to fold the factors into
                dX = apply_gradient_epilogue(dX, out_mul, out_act_grad)
        finally:
            self._out_epilogue = None
        return dX

    def _backward_finish(self, grad_output, ctx):
        """base class: undo the fused activation."""
        act = ctx["fused_act"]
        if act is None:
            return grad_output
        saved = ctx["pre"] if act == "gelu" else ctx["out"]
        _, spec = ops.plain_epilogue(None, (act, saved, 1.0 if act == "gelu" else ctx.get("out_scale", 1.0)))
        return ops.activation_backward(act, grad_output, spec[1])

    def _backward_messages(self, d_agg, ctx):
        """d(aggregated messages) [V, H] -> dX [V, D]; fills the edge-MLP kernel gradients."""
        if ctx["path"] == "C":
            return self._backward_C(d_agg, ctx)
        if ctx["path"] == "Bc":
            return self._backward_B_compact(d_agg, ctx)
        if ctx["path"] == "Ac":
            return self._backward_A_compact(d_agg, ctx)
        g = ctx["graph"]
        X = ctx["X"]
        V, D = X.shape
        L, H = g.num_edge_types, self._hidden_dim
        mlps = self._edge_type_mlps
        row_scale, ew_d, ew_s, node_scale = self._scales(g)
        if ctx["path"] == "B" and self._agg_general():
            # max / activation before aggregation: per-edge gradients first, then the sum over every (source, type) bucket
            Y = ctx["mlp_acts"][-1].view(V * L, H)
            dM = self._message_grads(g, d_agg, ctx, Y, g.array(ops.G_COLL_BY_DST), g.array(ops.G_TARGET_BY_DST), ew_d,
                                     node_scale, self._ident_e(g)[: g.num_edges])
            G = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, dM, col=g.array(ops.G_SRC2DST_POS)).view(V, L, H)
        elif ctx.get("f16x2") and ops.get_gemm_mode() == ops.GEMM_F16X2:  # (not after the spread guard demoted the mode)
            return self._backward_A_f16x2(d_agg, ctx, g, X, ew_s)
        else:
            # G[u, l, :] = sum over edges (u -> v) of type l of w_e * d_agg[v, :]
            G = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, d_agg, edge_weight=ew_s).view(V, L, H)
        dX = torch.empty_like(X)
        if ctx["path"] == "A":
            W = mlps.kernels[0]  # [L, Din, H]
            Din = W.shape[1]
            # horizontal stack [Din, L*H] of the L kernels: the backward pass becomes two large GEMMs
            #   dX = [G_0|...|G_{L-1}] @ [W_0|...|W_{L-1}]^T      dW_h = X^T @ [G_0|...|G_{L-1}]
            Wh = ops.sp_weight_operand(W, "stacked_rows", lambda: ops.permute_021(W))  # [Din, L, H], once per weight value
            G2 = G.view(V, L * H)
            if L > 0 and not self._use_target_state_as_input:
                # dW^T = G^T X [L*H, D]: ten full 128-row output tiles instead of the 2.5 x 4 ragged ones of X^T G
                epi = getattr(self, "_out_epilogue", None)
                if epi is not None:
                    dX = ops.gemm_grad(G2, Wh.view(D, L * H), trans_b=True, out=dX, out_mul=epi[0], act_grad=epi[1])
                    self._out_epilogue = None  # consumed
                else:
                    ops.gemm(G2, Wh.view(D, L * H), trans_b=True, out=dX)
                mlps.grads = [ops.transpose_batched(ops.gemm(G2, X, trans_a=True).view(L, H, D))]
                mlps.publish_grads()
                return dX
            dWh = torch.empty((Din, L, H), dtype=torch.float32, device=X.device)
            if L == 0:
                dX.zero_()
            else:
                ops.gemm(G2, Wh[:D].view(D, L * H), trans_b=True, out=dX)
                ops.gemm(X, G2, trans_a=True, out=dWh[:D].view(D, L * H))
            if self._use_target_state_as_input and L > 0:
                # target part: pre += (k_{l,v} x_v) W_t  ->  dW_t = (k x)^T d_agg ; dX_v += k d_agg W_t^T
                from ..graph_scales import target_multiplier

                k, ident_ptr, node_of_row = target_multiplier(g, row_scale)
                A = ctx["A"].view(V, L, 2 * D)
                for l in range(L):
                    ops.gemm(A[:, l, D:], d_agg, trans_a=True, out=dWh[D:, l, :])
                # dX_v += sum_l k_{l,v} * (d_agg[v] @ W_t[l]^T)
                kd = ops.gather_reduce(ident_ptr, node_of_row, d_agg, edge_weight=k).view(V, L * H)
                ops.gemm(kd, Wh[D:].view(D, L * H), trans_b=True, out=dX, accumulate=True)
            mlps.grads = [ops.permute_021(dWh)]
        else:
            self._mlp_all_types_backward(mlps, X, ctx["mlp_acts"], G, dX, accumulate=False)
        mlps.publish_grads()
        return dX
