"""Gated Graph Neural Network layer - mirror of tf2_gnn/layers/message_passing/ggnn.py."""
from typing import Any, Dict

import torch

from ... import ops
from .gnn_edge_mlp import GNN_Edge_MLP
from .message_passing import (
    MessagePassingInput,
    _INIT_GEN,
    default_device,
    glorot_uniform,
    register_message_passing_implementation,
)


@register_message_passing_implementation
class GGNN(GNN_Edge_MLP):
    """h^{t+1}_v := GRU( sum_l sum_{(u,v) in A_l} W_l h^t_u , h^t_v )   (ggnn.py:12-89).
    No message activation; requires hidden_dim == input dim (ggnn.py:30).
    GRUCell = [ext] tf.keras.layers.GRUCell(units=H) TF2 defaults (reset_after=True): kernel [D,3H]
    glorot_uniform, recurrent_kernel [H,3H] orthogonal, bias [2,3H] zeros, gate order z|r|h."""

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
        self._recurrent_unit = None

    def build(self, input_shapes: MessagePassingInput):
        D = int(input_shapes.node_embeddings[-1])
        H = self._hidden_dim
        dev = default_device()
        kernel = glorot_uniform((H, 3 * H), device=dev)  # GRUCell input = aggregated messages [V, H]
        a = torch.randn((3 * H, H), generator=_INIT_GEN, dtype=torch.float32)
        q, _ = torch.linalg.qr(a)  # [ext] orthogonal initialiser
        recurrent = q.t().contiguous().to(dev)
        if D != H:
            raise ValueError("GGNN requires hidden_dim == node embedding dimension (GRU state size)")
        bias = torch.zeros((2, 3 * H), dtype=torch.float32, device=dev)
        self._recurrent_unit = {
            "kernel": self.add_weight("kernel", kernel),
            "recurrent_kernel": self.add_weight("recurrent_kernel", recurrent),
            "bias": self.add_weight("bias", bias),
        }
        super().build(input_shapes)

    def _uses_base_aggregation(self) -> bool:
        return False

    # f16x2: the aggregate [V, H] is the K operand of the GRU kernel's gradient product - let the forward product's epilogue
    # write its split form (tfgnn_sp_gemm_nt_sp) instead of splitting it in a pass of its own
    _always_split_output = True

    def _post_activation_name(self):
        return None  # ggnn.py:80-89: aggregated messages go straight into the GRU

    def _finish(self, agg, X, ctx, training):
        ru = self._recurrent_unit
        b = ru["bias"].value
        # round 6 (QM9-sized batches, mode f16x2): both products of the cell inside the gate kernel - mh = h U + b_1 is neither
        # written by a product of its own nor read back; of it the backward pass needs the candidate third only
        both = ops.gemm_gru2(agg, ru["kernel"].value, b[0], X, ru["recurrent_kernel"].value, b[1])
        if both is not None:
            h_new, gates, mh = both
            ctx.update({"agg": agg, "mh": mh, "gates": gates, "out": h_new})
            return h_new
        mh = ops.gemm(X, ru["recurrent_kernel"].value, bias=b[1])
        fused = ops.gemm_gru(agg, ru["kernel"].value, b[0], mh, X)  # mx stays on chip (bf16x3 modes, H % 64 == 0)
        if fused is not None:
            h_new, gates = fused
        else:
            mx = ops.gemm(agg, ru["kernel"].value, bias=b[0])
            h_new, gates = ops.gru_gates_forward(mx, mh, X)
        ctx.update({"agg": agg, "mh": mh, "gates": gates, "out": h_new})
        return h_new

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        ru = self._recurrent_unit
        X = ctx["X"]
        if ctx.get("f16x2") and ops.get_gemm_mode() == ops.GEMM_F16X2:
            dX = self._backward_f16x2(grad_output, ctx, X)
            if dX is not None:
                return dX
        dmx, dmh, dh_direct = ops.gru_gates_backward(grad_output, ctx["gates"], ctx["mh"], X)
        ru["kernel"].grad = ops.gemm(ctx["agg"], dmx, trans_a=True)
        ru["recurrent_kernel"].grad = ops.gemm(X, dmh, trans_a=True)
        ru["bias"].grad = torch.stack([ops.colsum(dmx), ops.colsum(dmh)], dim=0)
        d_agg = ops.gemm(dmx, ru["kernel"].value, trans_b=True)
        dX_state = ops.gemm(dmh, ru["recurrent_kernel"].value, trans_b=True, out=dh_direct, accumulate=True)
        dX_msgs = self._backward_messages(d_agg, ctx)
        return ops.add_scale(dX_msgs, dX_state, 1.0)

    def recomputes_input_dropout(self, num_nodes: int, in_dim: int, num_edge_types: int) -> bool:
        """the three terms of d(node_embeddings) each take the recomputed mask in their epilogue (_backward_f16x2)"""
        H = self._hidden_dim
        return (type(self).backward is GGNN.backward and H % 64 == 0 and H <= 512 and in_dim == H and not self._user_message_function()
                and self._path() == "A" and self._f16x2_eligible(num_nodes, in_dim, num_edge_types, H))

    def backward_with_epilogue(self, grad_output, grad_is_pre_activation=False, out_mul=None, out_act_grad=None):
        """With a DropoutSpec as ``out_mul`` (a layer input dropped without a stored mask) the mask is recomputed in the epilogues
        of the three terms of d(node_embeddings) - the gate-gradient kernel's dh_new * z and the two accumulating products -
        instead of a pass over [V, H] afterwards (a STORED mask costs three more reads there: measured slower, so tensors keep
        the generic route)."""
        from .message_passing import apply_gradient_epilogue

        ctx = self._ctx
        if (grad_is_pre_activation or ctx is None or not isinstance(out_mul, ops.DropoutSpec) or type(self).backward is not GGNN.backward
                or not (ctx.get("f16x2") and ops.get_gemm_mode() == ops.GEMM_F16X2)):
            return super().backward_with_epilogue(grad_output, grad_is_pre_activation, out_mul, out_act_grad)
        dX = self._backward_f16x2(grad_output, ctx, ctx["X"], out_mul=out_mul)
        if dX is None:
            return super().backward_with_epilogue(grad_output, grad_is_pre_activation, out_mul, out_act_grad)
        return apply_gradient_epilogue(dX, None, out_act_grad) if out_act_grad is not None else dX

    def _backward_f16x2(self, grad_output, ctx, X, out_mul=None):
        """The GRU part of the backward pass on split operands (round 3): the gate-gradient kernel writes dmx / dmh only as
        SP16 operands (with the bias gradients folded in), the two kernel gradients are tfgnn_sp_gemm_tn products (K = V
        rows), the two input gradients tfgnn_sp_gemm_nt products - no fp32 [V, 3H] tensor is written or re-read, and the
        products run 3 piece products instead of 6.  The three terms of d(node_embeddings) - dh_new * z, dmh recurrent^T and the
        message path's G W^T - accumulate in ONE buffer through the products' epilogues.  (``out_mul``, the dropout mask of the
        layer's input, can ride along in the three epilogues; measured at the QM9 size that costs three more reads of the mask
        for one saved pass over [V, H] - 129.5 vs 124.2 ms per step - so GNN.backward keeps applying it afterwards.)
        None when the width has no such kernel."""
        res = ops.gru_gates_backward_sp(grad_output, ctx["gates"], ctx["mh"], X, out_mul=out_mul)
        if res is None:
            return None
        dmx_sp, dmh_sp, dh_direct, bias_grad = res
        ru = self._recurrent_unit
        Wk, Wr = ru["kernel"].value, ru["recurrent_kernel"].value  # [H, 3H]: rows are the N = H outputs, K = 3H contiguous
        ru["kernel"].grad = ops.sp_gemm_tn(ops.sp_rows_of(ctx["agg"]), dmx_sp)  # agg^T dmx  [H, 3H]
        ru["recurrent_kernel"].grad = ops.sp_gemm_tn(ops.sp_rows_of(X), dmh_sp)  # h^T dmh
        ru["bias"].grad = bias_grad
        d_agg = ops.sp_gemm_nt(dmx_sp, ops.sp_weight_operand(Wk, "rows", lambda: ops.sp_split_rows(Wk, defer=True)))
        dX = ops.sp_gemm_nt(dmh_sp, ops.sp_weight_operand(Wr, "rows", lambda: ops.sp_split_rows(Wr, defer=True)), out=dh_direct,
                            accumulate=True, out_mul=out_mul)
        # the message path adds its term into the same buffer (GNN_Edge_MLP._backward_A_f16x2 consumes the request)
        self._dx_accumulate = (dX, out_mul)
        try:
            dX_msgs = self._backward_messages(d_agg, ctx)
            if self._dx_accumulate is None:
                return dX_msgs  # == dX, accumulated in place
        finally:
            self._dx_accumulate = None
        if out_mul is not None:  # the message path took a route without an accumulating product
            dX_msgs = ops.mul(dX_msgs, out_mul.mask() if isinstance(out_mul, ops.DropoutSpec) else out_mul)
        return ops.add_scale(dX_msgs, dX, 1.0)
