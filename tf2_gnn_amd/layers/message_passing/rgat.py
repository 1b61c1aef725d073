"""Relational Graph Attention Network layer - mirror of tf2_gnn/layers/message_passing/rgat.py."""
import os
from typing import Any, Dict

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

    Forward on the device (csrc/rgat.hip, csrc/spmm.hip):
      Y = X @ [W_0|...|W_{L-1}]                                     MFMA GEMM, rows (v,l) of width H
      s_src, s_tgt [V*L, K]   the two halves of every attention logit, per (node, type, head)
      a [E, K]                per-head softmax over all edges entering a node (all types)
      out = act( sum_e a_e * Y[(src_e, l_e)] )                      gather kernel, per-head edge weights
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

    _message_function._tfgnn_builtin = True

    def call(self, inputs: MessagePassingInput, training: bool = False):
        if not getattr(type(self)._message_function, "_tfgnn_builtin", False):
            raise NotImplementedError(
                f"{type(self).__name__} overrides RGAT._message_function: RGAT's aggregation (per-head softmax over all incoming "
                "edges, rgat.py:125-163) is not the base class's, override call() and backward() as well"
            )
        X = inputs.node_embeddings
        V = X.shape[0]
        g = get_graph(inputs.adjacency_lists, V)
        L, H, K = g.num_edge_types, self._hidden_dim, self._num_heads
        if L != self._num_edge_types:
            raise ValueError(f"layer was built for {self._num_edge_types} edge types, got {L}")
        lib = _lib.load()
        dev = X.device
        f16 = self._f16x2_eligible(V, X.shape[1], L, H)
        if f16:
            # f16x2: Y = X [W_0 | ... | W_{L-1}] on split operands (X's split form comes from the dropout kernel in a GNN stack)
            Wn = ops.sp_weight_operand(self._kernels, "cols", lambda: ops.sp_split_cols(self._kernels, defer=True))
            Y = ops.sp_gemm_nt(ops.sp_rows_of(X), Wn)
        else:
            Y = ops.gemm(X, self._kernels)  # [V, L*H] == rows (v, l) of width H
        s_src = torch.empty((V * L, K), dtype=torch.float32, device=dev)
        s_tgt = torch.empty((V * L, K), dtype=torch.float32, device=dev)
        with ops.op_scope("rgat_node_scores", Y, s_src, s_tgt):
            _lib.check(
                lib.tfgnn_rgat_node_scores(
                    ops._ptr(Y), ops._ptr(self._attn), V, L, K, H, ops._ptr(s_src), ops._ptr(s_tgt), ops._stream()
                )
            )
        att, att_by_src = self._edge_attention(g, s_src, s_tgt, K, training)
        act = self._activation_name
        fused = None if act == "gelu" else act
        if L == 0 or g.num_edges == 0:
            out = torch.zeros((V, H), dtype=torch.float32, device=dev)
            if fused is not None:
                out = ops.activation_forward(fused, out)
        else:
            out = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, Y.view(V * L, H), edge_weight=att, post_act=fused)
        ctx = {"graph": g, "X": X, "Y": Y, "s_src": s_src, "s_tgt": s_tgt, "att": att, "fused_act": act, "f16x2": f16,
               "att_by_src": att_by_src}
        if act == "gelu":
            ctx["pre"] = out
            out = ops.activation_forward("gelu", out)
        ctx["out"] = out
        self._ctx = ctx
        return out

    def _f16x2_eligible(self, V, D, L, H) -> bool:
        """the three products of the layer (Y = X W, dX = dY W^T, dW = X^T dY) on split operands: widths the kernels tile"""
        def tiles(n):
            return n % 128 == 0 or n % 320 == 0

        return (ops.get_gemm_mode() == ops.GEMM_F16X2 and V > 0 and L > 0 and D % 16 == 0 and 32 <= D <= 512 and tiles(D)
                and tiles(L * H) and L * H <= 2048 and (H // self._num_heads) % 4 == 0)

    @staticmethod
    def _ident(g, n):
        ident = g._cache.get("ident_e")
        if ident is None or ident.numel() < n + 1:
            ident = torch.arange(n + 1, dtype=torch.int32, device=g.device)
            g._cache["ident_e"] = ident
        return ident

    def _edge_attention(self, g, s_src, s_tgt, K, training=True):
        """a[e,k]: per head, softmax over all edges entering the target (rgat.py:142-151) -> (a in by-target edge order,
        the same weights in by-source order or None).  One pass structure per CSR row (csrc/rgat.hip,
        tfgnn_rgat_attention_forward) when the head count is a power of two; otherwise edge-parallel kernels + two generic
        segment reductions over the node view (identity columns)."""
        lib = _lib.load()
        E, V, L = g.num_edges, g.num_nodes, g.num_edge_types
        dev = s_src.device
        att = torch.empty((E, K), dtype=torch.float32, device=dev)
        if E == 0:
            return att, None
        att_by_src = torch.empty((E, K), dtype=torch.float32, device=dev) if training else None
        g.ensure(ops.G_PART_PLAN_NODE | ops.G_PART_EDGE_MAPS)
        ws_bytes = lib.tfgnn_rgat_attention_workspace_bytes(g._h, K)
        ws = ops._workspace(dev, ws_bytes) if ws_bytes else None
        with ops.op_scope("rgat_attention_forward", s_src, s_tgt, att, att_by_src):
            rc = lib.tfgnn_rgat_attention_forward(g._h, ops._ptr(s_src), ops._ptr(s_tgt), K, ops._ptr(att), ops._ptr(att_by_src),
                                                  ops._ptr(ws), ws.numel() if ws is not None else 0, ops._stream())
        if rc == 0:
            return att, att_by_src
        if rc != -4:
            _lib.check(rc)
        coll, tgt = g.array(ops.G_COLL_BY_DST), g.array(ops.G_TARGET_BY_DST)
        ident = self._ident(g, E)[:E]
        scores = torch.empty((E, K), dtype=torch.float32, device=dev)
        _lib.check(lib.tfgnn_rgat_edge_scores(ops._ptr(coll), ops._ptr(tgt), ops._ptr(s_src), ops._ptr(s_tgt), E, L, K,
                                              ops._ptr(scores), ops._stream()))
        m = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, scores, col=ident, reduce=ops.REDUCE_MAX)  # [V, K]
        _lib.check(lib.tfgnn_rgat_edge_node_op(ops._ptr(scores), ops._ptr(tgt), ops._ptr(m), E, K, 0, ops._ptr(scores),
                                               ops._stream()))  # in place: p = exp(score - m[tgt])
        den = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, scores, col=ident)  # [V, K]
        _lib.check(lib.tfgnn_rgat_edge_node_op(ops._ptr(scores), ops._ptr(tgt), ops._ptr(den), E, K, 1, ops._ptr(att),
                                               ops._stream()))
        return att, None

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        """d(loss)/d(out) -> d(loss)/d(node_embeddings); fills the kernel / attention gradients."""
        return self._backward(grad_output, False, None, None)

    def activation_backward_spec(self):
        ctx = self._ctx
        if ctx is None or type(self).backward is not RGAT.backward or ctx.get("fused_act") is None:
            return None
        act = ctx["fused_act"]
        return act, (ctx["pre"] if act == "gelu" else ctx["out"])

    def recomputes_input_dropout(self, num_nodes: int, in_dim: int, num_edge_types: int) -> bool:
        return (type(self).backward is RGAT.backward and num_edge_types == self._num_edge_types
                and self._f16x2_eligible(num_nodes, in_dim, num_edge_types, self._hidden_dim))

    def backward_with_epilogue(self, grad_output, grad_is_pre_activation=False, out_mul=None, out_act_grad=None):
        """backward() whose input-gradient product dX = dY W^T applies the caller's next element-wise steps (dropout mask of
        this layer's input, activation derivative of the layer below) in its epilogue, and which takes a gradient the layer
        above already multiplied by this layer's activation derivative."""
        if type(self).backward is not RGAT.backward:
            return super().backward_with_epilogue(grad_output, grad_is_pre_activation, out_mul, out_act_grad)
        return self._backward(grad_output, grad_is_pre_activation, out_mul, out_act_grad)

    def _backward(self, grad_output, grad_is_pre_activation, out_mul, out_act_grad) -> torch.Tensor:
        from .message_passing import apply_gradient_epilogue

        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        lib = _lib.load()
        g, X, Y, att = ctx["graph"], ctx["X"], ctx["Y"], ctx["att"]
        s_src, s_tgt = ctx["s_src"], ctx["s_tgt"]
        V, D = X.shape
        L, H, K = g.num_edge_types, self._hidden_dim, self._num_heads
        E = g.num_edges
        dev = X.device
        act = ctx["fused_act"]
        d_agg = grad_output
        if act is not None and not grad_is_pre_activation:
            d_agg = ops.activation_backward(act, grad_output, ctx["pre"] if act == "gelu" else ctx["out"])
        d_agg = d_agg.contiguous()
        if L == 0 or E == 0:
            for v in self._variables:
                v.grad = torch.zeros_like(v.value)
            return torch.zeros_like(X)  # (zero times the caller's factors)
        s2d = g.array(ops.G_SRC2DST_POS)
        ident_e = self._ident(g, E)
        # (1) dY[(u,l),k,:] = sum over out-edges e of (u,l): a_ek * d_agg[tgt_e, k, :]
        att_s = ctx.get("att_by_src")  # written by the forward row kernels in training mode
        if att_s is None:
            att_s = ops.gather_reduce(ident_e[: E + 1], s2d, att)  # attention re-ordered to the by-src edge order
        # (2) gradient w.r.t. the attention values: da[e,k] = <Y[(src_e, l_e)], d_agg[tgt_e]>_k.  The gather of (1) reads
        #     d_agg[tgt_e] for every out-edge of (u, l) with Y[(u, l)] fixed per row: the products ride on it and land in
        #     by-target order through the edge map (round 4; a pass of its own over the edges before, 112 us at configs[2])
        if K > 1 and ops.graph_gather_dot_supported(H, K) and ops.env("TFGNN_RGAT_FUSED_DOT", "1") == "1":
            dY, da = ops.graph_gather_dot(g, ops.VIEW_BY_SRC_TYPED, d_agg, edge_weight=att_s, dot_rows=Y.view(V * L, H), dot_pos=s2d)
        else:
            dY = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, d_agg, edge_weight=att_s)  # [V*L, H]
            da = torch.empty((E, K), dtype=torch.float32, device=dev)
            _lib.check(
                lib.tfgnn_rgat_edge_dot(
                    ops._ptr(g.array(ops.G_COLL_BY_DST)), ops._ptr(g.array(ops.G_TARGET_BY_DST)), ops._ptr(Y),
                    ops._ptr(d_agg), E, K, H, ops._ptr(da), ops._stream(),
                )
            )
        # softmax + leaky_relu backward
        dz = torch.empty((E, K), dtype=torch.float32, device=dev)
        g.ensure(ops.G_PART_PLAN_NODE | ops.G_PART_EDGE_MAPS)
        ws_bytes = lib.tfgnn_rgat_attention_workspace_bytes(g._h, K)
        ws = ops._workspace(dev, ws_bytes) if ws_bytes else None
        with ops.op_scope("rgat_attention_backward", s_src, s_tgt, att, da, dz):
            rc = lib.tfgnn_rgat_attention_backward(g._h, ops._ptr(s_src), ops._ptr(s_tgt), ops._ptr(att), ops._ptr(da), K, ops._ptr(dz),
                                                   ops._ptr(ws), ws.numel() if ws is not None else 0, ops._stream())
        if rc == -4:  # head count not a power of two: the piecewise form
            t = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, ops.mul(att, da), col=ident_e[:E])  # [V, K] sum of a * da
            _lib.check(
                lib.tfgnn_rgat_edge_softmax_backward(
                    ops._ptr(g.array(ops.G_COLL_BY_DST)), ops._ptr(g.array(ops.G_TARGET_BY_DST)), ops._ptr(s_src),
                    ops._ptr(s_tgt), ops._ptr(att), ops._ptr(da), ops._ptr(t), E, L, K, ops._ptr(dz), ops._stream(),
                )
            )
        else:
            _lib.check(rc)
        # (3) logits are s_src[(src,l)] + s_tgt[(tgt,l)]: segment sums of dz over both bucketings
        ds_tgt = ops.graph_gather(g, ops.VIEW_BY_DST_TYPED, dz, col=ident_e[:E])  # [V*L, K]
        ds_src = ops.graph_gather(g, ops.VIEW_BY_SRC_TYPED, dz, col=s2d)  # [V*L, K]
        # (4) through the inner products with alpha
        #     d alpha[l,k,:Hk] = sum_v ds_src[(v,l),k] * Y[(v,l),k,:]   (block diagonal of a small GEMM)
        #     (tfgnn_rgat_alpha_grad: Y is read once; the two [V, L K]^T x [V, L H] products it replaces computed the whole
        #      [L K, L H] matrix for its block diagonal)
        d_attn = torch.empty_like(self._attn)
        ws_bytes = lib.tfgnn_rgat_alpha_grad_workspace_bytes(V, L, H)
        ws = ops._workspace(dev, ws_bytes) if ws_bytes else None
        with ops.op_scope("rgat_alpha_grad", ds_src, ds_tgt, Y):
            _lib.check(
                lib.tfgnn_rgat_alpha_grad(ops._ptr(ds_src), ops._ptr(ds_tgt), ops._ptr(Y), V, L, K, H, ops._ptr(d_attn), ops._ptr(ws),
                                          ws.numel() if ws is not None else 0, ops._stream())
            )
        d_kernels = dX = None
        if ctx.get("f16x2") and ops.get_gemm_mode() == ops.GEMM_F16X2:
            # the score terms are added while dY is ALSO written as the split operand of dX = dY W^T.  The weight gradient stays
            # on the exact bf16x3 kernel: dY's rows carry attention weights (1e-9 into a hub) - beyond the spread guard of the
            # split-operand TN product (measured: it trips on the first step of the rgat workload)
            dY_sp = ops.SplitOperand(torch.empty((V, L * H * 4), dtype=torch.uint8, device=dev),
                                     torch.empty((V, 1), dtype=torch.float32, device=dev), V, L * H, L * H)
            with ops.op_scope("rgat_scores_backward", ds_src, ds_tgt, dY, dY_sp.data):
                rc = lib.tfgnn_rgat_scores_backward_sp(ops._ptr(ds_src), ops._ptr(ds_tgt), ops._ptr(self._attn), ops._ptr(dY), 1, V, L, K, H,
                                                       ops._ptr(dY_sp.data), ops._ptr(dY_sp.inv_scale), ops._stream())
            if rc == 0:
                d_kernels = ops.gemm(X, dY.view(V, L * H), trans_a=True)  # X^T dY  [D, L*H]
                Wr = ops.sp_weight_operand(self._kernels, "rows", lambda: ops.sp_split_rows(self._kernels, defer=True))
                if getattr(self, "_want_split_input_grad", False) and D in (128, 256, 320):
                    dX, _ = ops.sp_gemm_nt_split(dY_sp, Wr, out_mul=out_mul, act_grad=out_act_grad)
                else:
                    dX = ops.sp_gemm_nt(dY_sp, Wr, out_mul=out_mul, act_grad=out_act_grad)
                out_mul = out_act_grad = None  # applied
            elif rc != -4:
                _lib.check(rc)
        if dX is None:
            with ops.op_scope("rgat_scores_backward", ds_src, ds_tgt, dY):
                _lib.check(
                    lib.tfgnn_rgat_scores_backward(
                        ops._ptr(ds_src), ops._ptr(ds_tgt), ops._ptr(self._attn), V, L, K, H, ops._ptr(dY), ops._stream()
                    )
                )
            # (5) Y = X @ W
            dYv = dY.view(V, L * H)
            d_kernels = ops.gemm(X, dYv, trans_a=True)  # [D, L*H]
            dX = ops.gemm(dYv, self._kernels, trans_b=True)
        for i in range(L):
            self._edge_type_to_message_computation_layer[i].grad = d_kernels[:, i * H : (i + 1) * H]
            self._edge_type_to_attention_parameters[i].grad = d_attn[i]
        if out_mul is not None or out_act_grad is not None:
            dX = apply_gradient_epilogue(dX, out_mul, out_act_grad)
        return dX
