"""GNN encoder - mirror of tf2_gnn/layers/gnn.py (GNNInput, GNN: hyper-parameters, build, call
signature, op order of ``_internal_call`` gnn.py:276-329).

The layer stack is the caller of the hot path; its glue (bias-free Dense + activation, dropout,
residual averaging, LayerNorm) runs on the same HIP library so that a whole forward+backward step
stays on the device without a second framework in the loop.  The batch's edges are bucketed once
(ops.Graph) and shared by all message passing layers and both passes.

Graph global exchange (layers/graph_global_exchange.py) is built from the same kernels - disabled in every
PPI / QM9 default_hypers file; see DESIGN.md.
"""
from __future__ import annotations

from typing import Any, Dict, List, NamedTuple, Optional, Sequence, Tuple

import os

import torch

from .. import ops
from ..utils.param_helpers import get_activation_function
from .graph_global_exchange import (
    GraphGlobalExchange,
    GraphGlobalExchangeInput,
    GraphGlobalGRUExchange,
    GraphGlobalMeanExchange,
    GraphGlobalMLPExchange,
)
from .message_passing import MessagePassing, MessagePassingInput, get_message_passing_class
from .message_passing.message_passing import Variable, _num_edge_types, default_device, get_graph, glorot_uniform


class GNNInput(NamedTuple):
    """Input named tuple for the GNN (gnn.py:21-27)."""

    node_features: torch.Tensor
    adjacency_lists: Tuple[torch.Tensor, ...]
    node_to_graph_map: torch.Tensor
    num_graphs: Any


class GNN:
    """Encode graph states using a combination of graph message passing layers and dense layers."""

    @classmethod
    def get_default_hyperparameters(cls, mp_style: Optional[str] = None) -> Dict[str, Any]:
        """Get the default hyperparameter dictionary for the class (gnn.py:53-79)."""
        these_hypers = {
            "message_calculation_class": "rgcn",
            "initial_node_representation_activation": "tanh",
            "dense_intermediate_layer_activation": "tanh",
            "num_layers": 4,
            "dense_every_num_layers": 2,
            "residual_every_num_layers": 2,
            "use_inter_layer_layernorm": False,
            "hidden_dim": 16,
            "layer_input_dropout_rate": 0.0,
            "global_exchange_mode": "gru",  # One of "mean", "mlp", "gru"
            "global_exchange_every_num_layers": 2,
            "global_exchange_weighting_fun": "softmax",  # One of "softmax", "sigmoid"
            "global_exchange_num_heads": 4,
            "global_exchange_dropout_rate": 0.2,
        }  # type: Dict[str, Any]
        if mp_style is not None:
            these_hypers["message_calculation_class"] = mp_style
        message_passing_class = get_message_passing_class(these_hypers["message_calculation_class"])
        message_passing_hypers = message_passing_class.get_default_hyperparameters()
        message_passing_hypers.update(these_hypers)
        return message_passing_hypers

    def __init__(self, params: Dict[str, Any]):
        """Initialise the layer (gnn.py:81-115)."""
        self._params = params
        self._hidden_dim = params["hidden_dim"]
        self._num_layers = params["num_layers"]
        self._dense_every_num_layers = params["dense_every_num_layers"]
        self._residual_every_num_layers = params["residual_every_num_layers"]
        self._use_inter_layer_layernorm = params["use_inter_layer_layernorm"]
        self._initial_node_representation_activation_fn = get_activation_function(
            params["initial_node_representation_activation"]
        )
        self._dense_intermediate_layer_activation_fn = get_activation_function(
            params["dense_intermediate_layer_activation"]
        )
        self._init_act = _act_name(self._initial_node_representation_activation_fn)
        self._dense_act = _act_name(self._dense_intermediate_layer_activation_fn)
        self._message_passing_class = get_message_passing_class(params["message_calculation_class"])

        if not params["global_exchange_mode"].lower() in {"mean", "mlp", "gru"}:
            raise ValueError(
                f"Unknown global_exchange_mode mode {params['global_exchange_mode']} - has to be one of 'mean', 'mlp', 'gru'!"
            )
        self._global_exchange_every_num_layers = params["global_exchange_every_num_layers"]

        self._initial_projection_layer: Optional[Variable] = None
        self._mp_layers: List[MessagePassing] = []
        self._inter_layer_layernorms: List[Tuple[Variable, Variable]] = []
        self._dense_layers: Dict[str, Variable] = {}
        self._global_exchange_layers: Dict[str, GraphGlobalExchange] = {}
        self.built = False
        self._ctx = None
        self._dropout_calls = 0
        self.dropout_seed = 0
        # Spread-guard policy of the f16x2 mode (backward()): the first `_guard_sync_passes` backward passes are checked
        # synchronously and recomputed on exact kernels when they trip; after that every `guard_check_every`-th pass is
        # (TFGNN_GUARD_CHECK_EVERY, 0 = never: a later trip only demotes the NEXT pass).  `guard_tripped_last_backward`
        # tells the caller what the last pass saw: False / True after a checked pass (True: recomputed on exact kernels),
        # for an unchecked pass the asynchronous flag as it stood when the pass was enqueued (a trip of that very pass shows
        # one pass later, together with the demotion warning) - a training loop that must not consume such gradients tests it.
        self._guard_sync_passes_init = int(ops.env("TFGNN_GUARD_SYNC_PASSES", "3"))
        self._guard_sync_passes = self._guard_sync_passes_init
        self.guard_check_every = int(ops.env("TFGNN_GUARD_CHECK_EVERY", "0"))
        self.guard_tripped_last_backward: Optional[bool] = None
        self._backward_passes = 0
        self._unchecked_split_passes = 0  # backward passes in mode f16x2 since the last synchronous check (late trips: below)
        self._unchecked_epoch = -1        # ops.REARM_EPOCH at the last of them
        self._late_trip_pass = -1         # the backward pass whose late trip moved the policy (one step per pass)
        self._late_trip_hook = ops.register_late_trip_policy(self)
        self._dense_split_ok = True  # cleared when the spread guard trips on this stack's Dense products (backward())
        self._dense_demoted_epoch = -1  # ops.REARM_EPOCH at that moment: set_gemm_mode("f16x2") re-arms this stack as well
        self._tn_demoted_epoch = None   # the same for stage 2 (per-relation weight gradients of the message-passing layers)
        # stage 1a: the Dense / projection weight gradients on the TWO-FACTOR product (each operand's rows guarded on their own,
        # 2^22) before they leave the split operands altogether (stage 1b)
        self._dense_tn_wide = False

    # ---- Keras-like plumbing ----------------------------------------------------------------
    @property
    def trainable_variables(self) -> List[Variable]:
        out = [self._initial_projection_layer]
        for i, mp in enumerate(self._mp_layers):
            out.extend(mp.trainable_variables)
            if str(i) in self._global_exchange_layers:
                out.extend(self._global_exchange_layers[str(i)].trainable_variables)
            if self._use_inter_layer_layernorm:
                out.extend(self._inter_layer_layernorms[i])
            if str(i) in self._dense_layers:
                out.append(self._dense_layers[str(i)])
        return out

    variables = trainable_variables

    def zero_grad(self):
        for v in self.trainable_variables:
            v.grad = None

    def build(self, tensor_shapes: GNNInput):
        """gnn.py:117-232."""
        dev = default_device()
        D_in = int(tensor_shapes.node_features[-1])
        adjacency_list_shapes = tuple(tensor_shapes.adjacency_lists)
        H = self._hidden_dim
        embedded_shape = (None, H)
        name = f"{self._message_passing_class.__name__}_GNN"
        self._initial_projection_layer = Variable(
            f"{name}/gnn_initial_node_projection/kernel", glorot_uniform((D_in, H), device=dev)
        )
        for layer_idx in range(self._num_layers):
            mp = self._message_passing_class(self._params)
            mp.build(MessagePassingInput(embedded_shape, adjacency_list_shapes))
            for v in mp.variables:
                v.name = f"{name}/Layer_{layer_idx}/MessagePassing/{v.name}"
            self._mp_layers.append(mp)
            if self._use_inter_layer_layernorm:
                self._inter_layer_layernorms.append(
                    (
                        Variable(f"{name}/Layer_{layer_idx}/LayerNorm/gamma", torch.ones(H, dtype=torch.float32, device=dev)),
                        Variable(f"{name}/Layer_{layer_idx}/LayerNorm/beta", torch.zeros(H, dtype=torch.float32, device=dev)),
                    )
                )
            if layer_idx % self._dense_every_num_layers == 0:
                self._dense_layers[str(layer_idx)] = Variable(
                    f"{name}/Layer_{layer_idx}/Dense/kernel", glorot_uniform((H, H), device=dev)
                )
            if layer_idx and layer_idx % self._global_exchange_every_num_layers == 0:
                mode = self._params["global_exchange_mode"].lower()
                cls = {"mean": GraphGlobalMeanExchange, "gru": GraphGlobalGRUExchange, "mlp": GraphGlobalMLPExchange}[mode]
                ex = cls(
                    hidden_dim=H,
                    weighting_fun=self._params["global_exchange_weighting_fun"],
                    num_heads=self._params["global_exchange_num_heads"],
                    dropout_rate=self._params["global_exchange_dropout_rate"],
                )
                ex.build(GraphGlobalExchangeInput(embedded_shape, (None,), ()))
                for v in ex.trainable_variables:
                    v.name = f"{name}/Layer_{layer_idx}/Global_Exchange/{v.name}"
                self._global_exchange_layers[str(layer_idx)] = ex
        self.built = True

    def __call__(self, inputs: GNNInput, training: bool = False, return_all_representations: bool = False):
        if not self.built:
            self.build(
                GNNInput(
                    tuple(inputs.node_features.shape),
                    tuple((None, 2) for _ in range(_num_edge_types(inputs.adjacency_lists))),
                    (None,),
                    (),
                )
            )
        return self.call(inputs, training, return_all_representations)

    def call(self, inputs: GNNInput, training: bool = False, return_all_representations: bool = False):
        """gnn.py:234-274: [V, hidden_dim], or (that, tuple of num_layers+1 x [V, hidden_dim])."""
        cur, all_reprs = self._internal_call(inputs, training, need_all_representations=return_all_representations)
        if return_all_representations:
            return cur, all_reprs
        return cur

    def graph_parts(self, num_nodes: int, edges_per_type) -> int:
        """Union of what the layers of this stack read from a batch's graph handle (``MessagePassing.graph_parts``): hand it
        to ``ops.Graph(..., parts=)`` when the handle is built ahead of the step (input pipeline)."""
        if not self._mp_layers:  # not built yet: the first call builds everything
            return ops.G_PARTS_DEFAULT
        parts = 0
        for mp in self._mp_layers:
            parts |= mp.graph_parts(num_nodes, edges_per_type, self._hidden_dim)
        return parts

    @staticmethod
    def _tiles(n: int) -> bool:
        return n % 128 == 0 or n % 320 == 0  # output widths the split-operand products tile

    def _dense_f16x2(self, in_dim: int, out_dim: int) -> bool:
        """Dense products on split operands (mode f16x2): the operands come split from their producers - the epilogue of
        the product before (tfgnn_sp_gemm_nt_sp), the dropout kernel, the input pipeline (ops.split_rows_remembered) -
        or from one split pass where no producer wrote them."""
        import os

        # On by default since round 4 (TFGNN_DENSE_F16X2=0 turns it off).  Round 3 measured a break-even (2.66 vs 2.65 ms per
        # step): the three K = 320 products got 26 us faster each, the per-step splits of two more weight matrices and the
        # factor and reduce passes of two more weight-gradient products took it back.  With the weight splits riding in the
        # merged small-pass launches, the factors computed inside the weight-gradient kernel and the layer-input dropout in
        # these products' epilogues (all four dropout passes of the benchmark stack gone) it is worth 2.48 vs 2.51 ms.
        if (not self._dense_split_ok or self._dense_tn_wide) and ops.REARM_EPOCH[0] != self._dense_demoted_epoch:
            # the mode was re-armed (ops.set_gemm_mode("f16x2")) after this stack moved its Dense weight gradients to the
            # two-factor product or its Dense products off the split operands: try again, with the synchronous check of the
            # first passes
            self._dense_split_ok = True
            self._dense_tn_wide = False
            self._guard_sync_passes = max(self._guard_sync_passes, self._guard_sync_passes_init)
        if ops.env("TFGNN_DENSE_F16X2", "1") == "0" or not self._dense_split_ok:
            return False
        return ops.get_gemm_mode() == ops.GEMM_F16X2 and in_dim % 16 == 0 and in_dim >= 32 and self._tiles(out_dim)

    def _dense(self, x, w: Variable, act_name, drop=None):
        """bias-free Dense + activation (gnn.py:136-141,163-170); gelu keeps its pre-activation.  ``drop`` = (rate, seed) of
        the layer-input dropout that follows: applied in the product's epilogue where the split-operand kernel runs it.
        -> (output, pre-activation | None, dropout applied?)"""
        if x.shape[0] > 0 and self._dense_f16x2(w.value.shape[0], w.value.shape[1]):
            wt = ops.sp_weight_operand(w.value, "cols", lambda: ops.sp_split_cols(w.value, defer=True))
            if act_name == "gelu":
                pre = ops.sp_gemm_nt(ops.sp_rows_of(x), wt)
                return ops.activation_forward("gelu", pre), pre, False
            if drop is not None and w.value.shape[1] in (128, 256, 320):
                out, _ = ops.sp_gemm_nt_split(ops.sp_rows_of(x), wt, act=act_name, dropout=drop)  # dropped, fp32 + split form
                return out, None, True
            return ops.sp_gemm_nt(ops.sp_rows_of(x), wt, act=act_name), None, False
        if act_name == "gelu":
            pre = ops.gemm(x, w.value)
            return ops.activation_forward("gelu", pre), pre, False
        return ops.gemm(x, w.value, act=act_name), None, False

    def _dense_backward(self, x, w: Variable, gpre, act_grad=None, need_input_grad=True):
        """w.grad = x^T gpre; returns gpre w^T (* act'(saved) of the op below, ``act_grad``) or None."""
        d_in, d_out = w.value.shape
        if x.shape[0] > 0 and self._dense_f16x2(d_in, d_out):
            g_sp = ops.sp_rows_of(gpre)  # written by the epilogue of the product above when that is a split-operand product
            if self._dense_tn_wide:  # (guard policy stage 1a: both operands' rows are un-normalised sums)
                w.grad = ops.sp_gemm_tn(ops.sp_rows_of(x), g_sp, out=torch.empty_like(w.value), wide=True)
            else:
                w.grad = ops.sp_gemm_tn(ops.sp_rows_of(x), g_sp)  # [in, out] = x^T gpre
            if not need_input_grad:
                return None
            if self._tiles(d_in):
                wr = ops.sp_weight_operand(w.value, "rows", lambda: ops.sp_split_rows(w.value, defer=True))
                return ops.sp_gemm_nt(g_sp, wr, act_grad=act_grad)
            return ops.gemm_grad(gpre, w.value, trans_b=True, act_grad=act_grad)
        w.grad = ops.gemm(x, gpre, trans_a=True)
        return ops.gemm_grad(gpre, w.value, trans_b=True, act_grad=act_grad) if need_input_grad else None

    def _fuse_dropout_statically(self, layer_idx: int) -> bool:
        """May the dropout at the input of layer ``layer_idx`` be applied by the op that produces that input?  Only where the
        backward pass hands the mask to an input-gradient product as well (no residual sum at this layer) and nothing but a
        Dense / the projection / a message passing layer precedes it."""
        if ops.env("TFGNN_FUSED_DROPOUT", "1") == "0" or ops.get_gemm_mode() != ops.GEMM_F16X2:
            return False
        if layer_idx % self._residual_every_num_layers == 0 and not (layer_idx == 0 and self._residual_every_num_layers >= self._num_layers):
            return False
        if layer_idx == 0:
            return True
        prev = layer_idx - 1
        return not self._use_inter_layer_layernorm and str(prev) not in self._global_exchange_layers

    def _recompute_dropout(self, layer_idx: int, mp_layer, V: int, D: int, L: int) -> bool:
        """Drop the input of layer ``layer_idx`` WITHOUT storing the mask?  Where the backward pass hands the mask to the layer
        (no residual sum here) and the layer recomputes it in a product epilogue (MessagePassing.recomputes_input_dropout).
        TFGNN_RECOMPUTE_DROPOUT=0: always store it."""
        if ops.env("TFGNN_RECOMPUTE_DROPOUT", "1") == "0" or ops.get_gemm_mode() != ops.GEMM_F16X2:
            return False
        if layer_idx % self._residual_every_num_layers == 0 and not (layer_idx == 0 and self._residual_every_num_layers >= self._num_layers):
            return False
        # (a layer that has not been built yet - the first call of the stack - keeps the stored mask)
        return bool(getattr(mp_layer, "built", False) and mp_layer.recomputes_input_dropout(V, D, L))

    def _internal_call(self, inputs: GNNInput, training: bool = False, need_all_representations: bool = True):
        """gnn.py:276-329, same op order.  With ``need_all_representations=False`` (what ``call`` passes unless asked for the
        tuple) the outputs nobody reads - a layer's result BEFORE the next layer's input dropout - are not materialised: their
        producer drops them in its epilogue, and the second result is None."""
        X = inputs.node_features
        V = X.shape[0]
        adj = inputs.adjacency_lists
        if self._tn_demoted_epoch is not None and ops.REARM_EPOCH[0] != self._tn_demoted_epoch:
            # the mode was re-armed (ops.set_gemm_mode("f16x2")) after stage 2 of the guard policy took the per-relation weight
            # gradients of this stack's layers off the split operands: they try again, under the synchronous check
            for mp in self._mp_layers:
                if getattr(mp, "_grouped_tn_split_ok", True) is False:
                    mp._grouped_tn_split_ok = True
            self._tn_demoted_epoch = None
            self._guard_sync_passes = max(self._guard_sync_passes, self._guard_sync_passes_init)
        graph = adj if isinstance(adj, ops.Graph) else get_graph(
            adj, V, parts=self.graph_parts(V, [int(a.shape[0]) if a.numel() else 0 for a in adj]))
        graph = get_graph(graph, V)
        rate = float(self._params["layer_input_dropout_rate"])
        steps = []
        NL = self._num_layers

        if training and ops.get_gemm_mode() == ops.GEMM_F16X2 and ops.env("TFGNN_BATCHED_WEIGHT_SPLIT", "0") == "1":
            # Opt-in: both split forms of every layer's kernel stack in ONE launch at the start of the step instead of two
            # small launches per layer.  Measured a LOSS (2.67 vs 2.62 ms per step): the batched launch is 59 us less
            # split-kernel time, but every layer product then starts on a weight operand that is no longer in L2 and takes
            # 15-17 us longer (80 -> 97 us forward, 95 -> 111 us dX) - splitting a kernel right before the product that
            # re-reads it 235 times doubles as its prefetch.
            stacks = [mp._edge_type_mlps.kernels[0] for mp in self._mp_layers
                      if getattr(mp, "_edge_type_mlps", None) is not None and hasattr(mp, "_f16x2_eligible")
                      and mp._path() == "A" and mp._f16x2_eligible(V, self._hidden_dim, graph.num_edge_types, self._hidden_dim)]
            if len(stacks) > 1 and len({tuple(w.shape) for w in stacks}) == 1:
                ops.sp_split_weights(stacks)
        # the dropout seeds, in the order the stack draws them (one per layer input, one per global exchange)
        drop_seed, ex_seed = [None] * NL, {}
        for i in range(NL):
            if training and rate > 0.0:
                self._dropout_calls += 1
                drop_seed[i] = self.dropout_seed * 1000003 + self._dropout_calls
            if str(i) in self._global_exchange_layers:
                self._dropout_calls += 1
                ex_seed[i] = self.dropout_seed * 1000003 + self._dropout_calls

        def drop_for(i, producer_output_is_read):
            """(rate, seed) if the producer of layer i's input may apply that layer's dropout itself"""
            if i >= NL or drop_seed[i] is None or producer_output_is_read or not self._fuse_dropout_statically(i):
                return None
            return (rate, drop_seed[i])

        cur, pre0, dropped = self._dense(X, self._initial_projection_layer, self._init_act,
                                         drop=drop_for(0, need_all_representations))
        ctx = {"X": X, "h0": cur, "pre0": pre0, "steps": steps, "h0_scale": (1.0 - rate) if dropped else 1.0,
               "all_representations": need_all_representations}
        last = cur
        all_reprs = [cur]
        for layer_idx, mp_layer in enumerate(self._mp_layers):
            st = {}
            if drop_seed[layer_idx] is not None:
                if dropped:  # the producer of `cur` applied this mask in its epilogue; nothing is stored
                    st["drop"] = ops.DropoutSpec(rate, drop_seed[layer_idx], tuple(cur.shape), cur.device)
                    st["drop_by_producer"] = True
                elif self._recompute_dropout(layer_idx, mp_layer, V, int(cur.shape[1]), graph.num_edge_types):
                    # the layer's backward pass recomputes the mask in a product epilogue: drop without storing it
                    st["drop"] = ops.DropoutSpec(rate, drop_seed[layer_idx], tuple(cur.shape), cur.device)
                    cur, _ = ops.dropout_forward(cur, rate, drop_seed[layer_idx], want_mask=False)
                else:
                    cur, st["mask"] = ops.dropout_forward(cur, rate, drop_seed[layer_idx])
            dropped = False
            if layer_idx % self._residual_every_num_layers == 0:
                tmp = cur
                if layer_idx > 0:
                    cur = ops.add_scale(cur, last, 0.5)
                last = tmp
            dense_here = layer_idx % self._dense_every_num_layers == 0
            has_ex = str(layer_idx) in self._global_exchange_layers
            # a Dense right behind this layer takes its input as a split operand: let the layer's product write it
            mp_layer._want_split_output = bool(getattr(mp_layer, "_always_split_output", False)) or (
                dense_here and not has_ex
                and not self._use_inter_layer_layernorm and self._dense_f16x2(self._hidden_dim, self._hidden_dim)
            )
            # ... and the next layer's input dropout, when this layer's output goes straight into it
            direct = not (dense_here or has_ex or self._use_inter_layer_layernorm)
            mp_layer._fused_output_dropout = drop_for(layer_idx + 1, need_all_representations) if direct else None
            mp_layer._fused_output_dropout_done = False
            cur = mp_layer(MessagePassingInput(node_embeddings=cur, adjacency_lists=graph), training=training)
            dropped = bool(mp_layer._fused_output_dropout_done)
            mp_layer._fused_output_dropout = None
            all_reprs.append(cur)
            if has_ex:  # gnn.py:307-315
                ex = self._global_exchange_layers[str(layer_idx)]
                ex.dropout_seed = ex_seed[layer_idx]
                cur = ex(GraphGlobalExchangeInput(cur, inputs.node_to_graph_map, inputs.num_graphs), training=training)
            if self._use_inter_layer_layernorm:
                g_, b_ = self._inter_layer_layernorms[layer_idx]
                st["ln_in"] = cur
                cur, st["ln_mean"], st["ln_rstd"] = ops.layernorm_forward(cur, g_.value, b_.value, 1e-3)
            if dense_here:
                st["dense_in"] = cur
                cur, st["dense_pre"], dropped = self._dense(cur, self._dense_layers[str(layer_idx)], self._dense_act,
                                                            drop=drop_for(layer_idx + 1, False))
                st["dense_out"] = cur
                st["dense_out_scale"] = (1.0 - rate) if dropped else 1.0
            steps.append(st)
        self._ctx = ctx
        return cur, (tuple(all_reprs) if need_all_representations else None)

    def dropout_masks(self):
        """The layer-input dropout masks of the last training-mode forward pass, one [V, H] tensor (0 or 1/(1-rate)) or None
        per layer - stored ones as they are, fused ones (applied in a producer's epilogue, never stored) regenerated from
        their seed."""
        if self._ctx is None:
            raise RuntimeError("no forward pass recorded")
        out = []
        for st in self._ctx["steps"]:
            out.append(st["mask"] if "mask" in st else (st["drop"].mask() if "drop" in st else None))
        return out

    # ---- backward (stands in for tf.GradientTape, models/graph_task_model.py:347-357) ---------
    def _tail_first_backward_op(self, layer_idx: int, ctx):
        """What the backward pass meets first below layer ``layer_idx``'s input gradient (after its dropout mask): the
        activation of the last forward op of layer ``layer_idx - 1`` (Dense, else the message passing layer itself), or
        of the initial projection for layer 0.  -> (activation name, saved tensor[, saved_scale]) or None if that op is not a
        plain activation (LayerNorm / global exchange in between, no activation); saved_scale = 1 - rate when the saved
        tensor is the DROPPED output of that op (its producer applied the next dropout): the derivative is taken at
        saved * saved_scale."""
        if layer_idx == 0:
            if self._init_act is None:
                return None
            return self._init_act, (ctx["pre0"] if self._init_act == "gelu" else ctx["h0"]), ctx.get("h0_scale", 1.0)
        prev = layer_idx - 1
        st = ctx["steps"][prev]
        if prev % self._dense_every_num_layers == 0:
            if self._dense_act is None:
                return None
            return (self._dense_act, (st["dense_pre"] if self._dense_act == "gelu" else st["dense_out"]),
                    st.get("dense_out_scale", 1.0))
        if self._use_inter_layer_layernorm or str(prev) in self._global_exchange_layers:
            return None
        return self._mp_layers[prev].activation_backward_spec()

    def backward(self, grad_output: Optional[torch.Tensor], need_input_grad: bool = False,
                 grad_all_representations: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """Back-propagate d(loss)/d(final node representations) through the stack; fills ``.grad``
        of every trainable variable; returns d(loss)/d(node_features) if requested.
        ``grad_all_representations`` (num_layers + 1 entries, None = no gradient): d(loss)/d(the tuple returned with
        ``return_all_representations=True``), for task heads that read the intermediate results
        (tf2_gnn/models/graph_regression_task.py:112-121); entry i + 1 is the output of message passing layer i.

        Between two layers the gradient only meets element-wise factors - the dropout mask of the upper layer's
        input and the activation derivative of the lower layer's last op; where no residual / LayerNorm / exchange
        intervenes they are handed to the upper layer's input-gradient GEMM (backward_with_epilogue) instead of
        running as two more passes over [V, H]."""
        ctx = self._ctx
        if ctx is None:
            raise RuntimeError("backward called before a forward pass")
        extras = list(grad_all_representations) if grad_all_representations is not None else [None] * (self._num_layers + 1)
        if len(extras) != self._num_layers + 1:
            raise ValueError(f"grad_all_representations needs {self._num_layers + 1} entries, got {len(extras)}")
        if grad_all_representations is not None and not ctx.get("all_representations", True) and any(e is not None for e in extras):
            raise ValueError("grad_all_representations given, but the forward pass ran without return_all_representations=True "
                             "(the intermediate results were not materialised)")
        g = grad_output
        if g is None:  # only intermediate results were read: the last op's output gets no gradient of its own
            g = torch.zeros_like(ctx["steps"][-1].get("dense_out", ctx["h0"])) if self._num_layers else None
        g_is_pre = False  # g already carries the activation derivative of the op differentiated next
        g_last = None
        try:
            was_f16x2 = ops.get_gemm_mode() == ops.GEMM_F16X2
            self._backward_passes += 1
            periodic = self.guard_check_every > 0 and self._backward_passes % self.guard_check_every == 0
            if ops.capturing():  # a step being captured into a hipGraph cannot wait for the device: never a checked pass
                if was_f16x2 and self._guard_sync_passes > 0:
                    raise RuntimeError("tf2_gnn_amd: this GNN's first backward passes are checked synchronously "
                                       f"(TFGNN_GUARD_SYNC_PASSES, {self._guard_sync_passes} left); run them eagerly before "
                                       "capturing the step (capture.CapturedStep warmup)")
                periodic = False
            if not (was_f16x2 and (self._guard_sync_passes > 0 or periodic)):
                self.guard_tripped_last_backward = bool(was_f16x2 and ops.f16x2_guard_flag_async())
                if was_f16x2:
                    self._unchecked_split_passes += 1
                    self._unchecked_epoch = ops.REARM_EPOCH[0]
                return self._backward_walk(ctx, g, g_is_pre, g_last, extras, need_input_grad)
            # The spread guard of the split weight-gradient products reports through a host-visible flag WITHOUT a stream
            # synchronisation: a pass that trips it has produced its gradients by the time the host notices.  For the
            # first passes of a model (TFGNN_GUARD_SYNC_PASSES, default 3: whether a model's gradient rows are spread
            # that far shows at once - RGAT's attention-weighted rows trip it on the first step) wait for the pass and,
            # if it tripped, run it again on sturdier kernels, one step per attempt (_demote_fragile_weight_gradients): this
            # stack's Dense / projection weight gradients on the two-factor product, then its Dense products off the split
            # operands (their operand rows - un-normalised sums, attention-weighted gradients - are the usual culprit; the
            # message products keep their split operands), then the per-relation MLP weight gradients, then the whole mode.  Later
            # trips demote the mode from the NEXT pass on and say so (ops.get_gemm_mode warns), but the tripping pass
            # itself is not recomputed (README.md).  While this section runs, a set flag does not demote the mode on sight.
            if self._guard_sync_passes > 0:
                self._guard_sync_passes -= 1
            self.guard_tripped_last_backward = False
            self._unchecked_split_passes = 0
            with ops.hold_spread_guard():
                result = self._backward_walk(ctx, g, g_is_pre, g_last, extras, need_input_grad)
                for attempt in range(4):
                    if not ops.f16x2_guard_tripped_sync():
                        break
                    self.guard_tripped_last_backward = True
                    what = self._demote_fragile_weight_gradients() if attempt < 3 else None
                    if what:
                        ops.rearm_spread_guard()
                        import warnings

                        warnings.warn(f"tf2_gnn_amd: operand rows of a weight-gradient product of this GNN are spread beyond the range "
                                      f"of the split-operand product that ran: {what}; the pass was recomputed")
                    else:
                        ops.demote_gemm_mode()  # the whole mode (sticky), with a warning
                    for v in self.trainable_variables:
                        v.grad = None
                    result = self._backward_walk(ctx, g, False, None, extras, need_input_grad)
            return result
        finally:
            # stand-alone layer.backward() calls after this pass join the second stream themselves again
            for mp in self._mp_layers:
                mp._defer_aux_join = False
            ops.aux_flush()        # the deferred split reductions of the weight gradients: one launch for all layers
            ops.join_aux_stream()  # weight gradients whose last pass ran on the second stream

    def guard_state(self) -> Dict[str, Any]:
        """What the spread-guard policy has done to this stack so far: ``tripped`` (the last backward pass, see
        ``guard_tripped_last_backward``) and ``stage`` - "none", "1a" (Dense / projection weight gradients on the two-factor
        product), "1b" (Dense products off the split operands), "2" (per-relation weight gradients on the exact kernels),
        "3" (the whole mode demoted to bf16x3).  bench.py prints it with every line."""
        stage = "none"
        if self._dense_tn_wide:
            stage = "1a"
        if not self._dense_split_ok:
            stage = "1b"
        if self._tn_demoted_epoch is not None:
            stage = "2"
        if self._backward_passes and ops.get_gemm_mode() != ops.GEMM_F16X2:
            stage = "3" if stage != "none" or ops.f16x2_guard_flag_async() else stage
        return {"tripped": self.guard_tripped_last_backward, "stage": stage, "checked_passes_left": self._guard_sync_passes}

    def _on_late_guard_trip(self) -> Optional[str]:
        """A backward pass that was NOT checked synchronously tripped the spread guard (ops._f16x2_on sees the flag some time
        later - possibly while that pass is still being enqueued).  Instead of losing the whole mode, a stack that ran such
        passes walks its staged policy one step - one step per pass, however many of its products report - and has its next
        passes checked (and recomputed when they trip) again.  -> what changed kernels, None: nothing left / not involved."""
        if self._unchecked_split_passes == 0 or self._unchecked_epoch != ops.REARM_EPOCH[0]:
            return None  # (no unchecked pass of this stack since the mode was last armed: somebody else's products)
        if self._late_trip_pass == self._backward_passes:
            return "(this stack's policy has moved a step for that pass already)"
        what = self._demote_fragile_weight_gradients()
        if what:
            self._late_trip_pass = self._backward_passes
            self._guard_sync_passes = max(self._guard_sync_passes, self._guard_sync_passes_init)
            self.guard_tripped_last_backward = True
        return what

    def _demote_fragile_weight_gradients(self) -> Optional[str]:
        """The stages of the spread guard's policy before the whole mode is demoted: the weight-gradient products whose operand
        ROWS are un-normalised sums change kernels, one step per call - first this stack's Dense / projection weight gradients
        go from the combined-factor TN product (range 2^20) to the two-factor one (2^22 per operand), then these products leave
        the split operands, then the per-relation TN products of the compact-row MLP path (RGIN, GNN_Edge_MLP: the two-factor
        product) go to the exact kernels; the message products keep their split operands.
        -> what was demoted, or None when nothing is left to demote."""
        if self._dense_split_ok and self._dense_f16x2(self._hidden_dim, self._hidden_dim):
            self._dense_demoted_epoch = ops.REARM_EPOCH[0]
            if not self._dense_tn_wide and ops.env("TFGNN_DENSE_TN_WIDE", "1") == "1":
                self._dense_tn_wide = True
                return "the Dense / projection weight gradients (now on the two-factor split-operand product)"
            self._dense_split_ok = False
            return "the Dense / projection products (now on the exact bf16x3 kernels)"
        did = False
        for mp in self._mp_layers:
            if getattr(mp, "_grouped_tn_used", False) and getattr(mp, "_grouped_tn_split_ok", True):
                mp._grouped_tn_split_ok = False
                did = True
        if did:
            self._tn_demoted_epoch = ops.REARM_EPOCH[0]
        return "the per-relation MLP weight gradients (now on the exact bf16x3 kernels)" if did else None

    @staticmethod
    def _activation_backward_scaled(act, g, saved, saved_scale):
        _, spec = ops.plain_epilogue(None, (act, saved, saved_scale))
        return ops.activation_backward(act, g, spec[1])

    def _backward_walk(self, ctx, g, g_is_pre, g_last, extras, need_input_grad):
        for layer_idx in range(self._num_layers - 1, -1, -1):
            st = ctx["steps"][layer_idx]
            mp = self._mp_layers[layer_idx]
            mp._defer_aux_join = True  # joined once, at the end of this backward pass
            has_ln = self._use_inter_layer_layernorm
            has_ex = str(layer_idx) in self._global_exchange_layers
            if layer_idx % self._dense_every_num_layers == 0:
                w = self._dense_layers[str(layer_idx)]
                gpre = g
                if self._dense_act is not None and not g_is_pre:
                    gpre = self._activation_backward_scaled(
                        self._dense_act, g, st["dense_pre"] if self._dense_act == "gelu" else st["dense_out"],
                        st.get("dense_out_scale", 1.0))
                nxt = None if (has_ln or has_ex or extras[layer_idx + 1] is not None) else mp.activation_backward_spec()
                g = self._dense_backward(st["dense_in"], w, gpre, act_grad=nxt)
                g_is_pre = nxt is not None
            if has_ln:
                gam, bet = self._inter_layer_layernorms[layer_idx]
                g, gam.grad, bet.grad = ops.layernorm_backward(g, st["ln_in"], gam.value, st["ln_mean"], st["ln_rstd"])
            if has_ex:
                g = self._global_exchange_layers[str(layer_idx)].backward(g)
            if extras[layer_idx + 1] is not None:  # g is the plain gradient here (the fusions above were switched off)
                g = ops.add_scale(g, extras[layer_idx + 1], 1.0)
            residual_here = layer_idx % self._residual_every_num_layers == 0 and not (layer_idx == 0 and g_last is None)
            mask = st.get("mask", st.get("drop"))  # a stored mask tensor, or the spec of one applied by the producer
            if not residual_here:
                nxt = None if extras[layer_idx] is not None else self._tail_first_backward_op(layer_idx, ctx)
                # the gradient this layer hands down goes straight into a Dense / projection weight-gradient product when
                # the op below is one and its activation derivative is folded in here: ask for it as a split operand too
                below_is_dense = layer_idx == 0 or (layer_idx - 1) % self._dense_every_num_layers == 0
                mp._want_split_input_grad = nxt is not None and below_is_dense and self._dense_f16x2(self._hidden_dim, self._hidden_dim)
                g = mp.backward_with_epilogue(g, grad_is_pre_activation=g_is_pre, out_mul=mask, out_act_grad=nxt)
                g_is_pre = nxt is not None
            else:
                g = mp.backward_with_epilogue(g, grad_is_pre_activation=g_is_pre)
                g_is_pre = False
                if layer_idx > 0:
                    half = ops.add_scale(g, g, 0.25)  # 0.5 * g
                    g = half if g_last is None else ops.add_scale(half, g_last, 1.0)
                    g_last = half
                else:
                    g = ops.add_scale(g, g_last, 1.0)
                    g_last = None
                if mask is not None:
                    g = ops.mul(g, mask.mask() if isinstance(mask, ops.DropoutSpec) else mask)
        if extras[0] is not None:
            g = ops.add_scale(g, extras[0], 1.0)
        gpre = g
        if self._init_act is not None and not g_is_pre:
            gpre = self._activation_backward_scaled(self._init_act, g, ctx["pre0"] if self._init_act == "gelu" else ctx["h0"],
                                                    ctx.get("h0_scale", 1.0))
        return self._dense_backward(ctx["X"], self._initial_projection_layer, gpre, need_input_grad=need_input_grad)


def _act_name(fn):
    return None if fn is None else fn.tfgnn_name
