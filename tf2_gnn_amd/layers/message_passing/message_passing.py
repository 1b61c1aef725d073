"""Message passing layer - MI355X-native mirror of
tf2_gnn/layers/message_passing/message_passing.py (same class / method names, hyper-parameter
keys, call signature and error behaviour; the compute runs in libtfgnn.so).

Differences a caller can observe, by design:
  * tensors are torch tensors on a ROCm device (fp32 node states, int32 adjacency lists);
  * the edges of a batch are bucketed once into a ``Graph`` (ops.Graph) that all layers and both
    passes of a step share; ``MessagePassingInput.adjacency_lists`` may be that Graph directly;
  * gradients come from explicit ``backward`` methods (the reference relies on tf.GradientTape,
    models/graph_task_model.py:347-357).
"""
from __future__ import annotations

from abc import abstractmethod
from collections import OrderedDict
from typing import Any, Dict, List, NamedTuple, Optional, Sequence, Tuple, Union

import torch

from ... import ops
from ...utils.param_helpers import get_activation_function, get_aggregation_function


class MessagePassingInput(NamedTuple):
    """A named tuple to hold input to the message passing layer (message_passing.py:13-17)."""

    node_embeddings: torch.Tensor
    adjacency_lists: Union[Tuple[torch.Tensor, ...], "ops.Graph"]


class Variable:
    """A named trainable tensor (stand-in for tf.Variable: ``.name``, ``.shape``, ``.value``).

    Updating the value: ``assign(new)`` or in-place torch arithmetic on ``.value`` (``var.value.add_(...)``,
    ``var.value.copy_(...)``) - both are seen by the library, which keeps derived forms of the weights (split fp16
    operands, transposed copies) per weight VALUE.  Updates torch's version counter does not see are NOT supported without
    telling the library: after ``var.value.data.add_(...)``, an optimizer kernel writing through ``data_ptr()``, or any
    other out-of-band write call ``var.mark_updated()`` (ops.notify_weights_changed), otherwise the next forward / backward
    pass may use the stale derived forms.  TFGNN_CHECK_WEIGHT_CACHE=1 makes the library verify this with checksums."""

    def __init__(self, name: str, value: torch.Tensor, trainable: bool = True):
        self.name = name
        self.value = value
        self.trainable = trainable
        self.grad: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return tuple(self.value.shape)

    def assign(self, value) -> "Variable":
        """In-place update (tf.Variable.assign).  ``value`` tensors of several layers are VIEWS into one fused buffer (the
        per-type kernels of an edge-MLP stack, the RGAT kernels): an optimizer or a checkpoint loader must write through
        ``assign`` / ``value.copy_`` / in-place arithmetic - rebinding ``var.value`` to a new tensor detaches the variable
        from the buffer the kernels read.  Cached derived forms of the buffer are dropped explicitly."""
        new = torch.as_tensor(value, dtype=self.value.dtype)
        if tuple(new.shape) != tuple(self.value.shape):
            raise ValueError(f"Shape mismatch for {self.name}: {tuple(self.value.shape)} vs {tuple(new.shape)}")
        with torch.no_grad():
            self.value.copy_(new.to(self.value.device))
        self.mark_updated()
        return self

    def mark_updated(self) -> "Variable":
        """The value was changed in place by something torch cannot see (``.data`` arithmetic, a raw-pointer kernel):
        drop every derived form of this variable's buffer."""
        if self.value.is_cuda:
            ops.notify_weights_changed(self.value)
        return self

    def __repr__(self):
        return f"Variable({self.name!r}, shape={self.shape})"


_PARAM_DEVICE = [None]


def default_device() -> torch.device:
    if _PARAM_DEVICE[0] is not None:
        return _PARAM_DEVICE[0]
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")  # shape/plumbing tests only; every op raises on CPU tensors


def set_default_device(device):
    _PARAM_DEVICE[0] = None if device is None else torch.device(device)


_INIT_GEN = torch.Generator(device="cpu")
_INIT_GEN.manual_seed(0)


def set_seed(seed: int):
    _INIT_GEN.manual_seed(seed)


def glorot_uniform(shape, fan_in=None, fan_out=None, device=None) -> torch.Tensor:
    """[ext] Keras default kernel initializer (glorot_uniform): U(-l, l), l = sqrt(6/(fan_in+fan_out))."""
    if fan_in is None:
        fan_in, fan_out = shape[-2], shape[-1]
    limit = (6.0 / max(fan_in + fan_out, 1)) ** 0.5
    t = (torch.rand(tuple(shape), generator=_INIT_GEN, dtype=torch.float32) * 2.0 - 1.0) * limit
    return t.to(device or default_device())


# --------------------------------------------------------------------------------------------
# graph cache: GNN builds the Graph once per batch; stand-alone layer calls share it too
# --------------------------------------------------------------------------------------------
_GRAPH_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()  # key -> (Graph, the adjacency tensors it was built from)
_GRAPH_CACHE_SIZE = 4


def _graph_key(adjacency_lists, num_nodes: int) -> tuple:
    return (int(num_nodes),) + tuple(
        (a.data_ptr(), tuple(a.shape), tuple(a.stride()), str(a.device), a._version) for a in adjacency_lists
    )


def get_graph(adjacency_lists, num_nodes: int, parts: int = ops.G_PARTS_DEFAULT) -> "ops.Graph":
    """The bucketed Graph of a batch, built once and shared by every layer / pass that is handed the same adjacency
    tensors.  An entry holds references to those tensors, so their addresses cannot be given to another batch while
    the entry is alive (a key of addresses alone would return a stale Graph when the allocator reuses them); an
    in-place edit bumps ``_version`` and misses.  Evicted Graphs are released by reference count, not closed: a layer
    context of an earlier forward pass may still hold them."""
    if isinstance(adjacency_lists, ops.Graph):
        if adjacency_lists.num_nodes != num_nodes:
            raise ValueError("Graph was built for a different number of nodes")
        return adjacency_lists
    adjacency_lists = tuple(adjacency_lists)
    key = _graph_key(adjacency_lists, num_nodes)
    entry = _GRAPH_CACHE.get(key)
    if entry is None:
        g = ops.Graph(adjacency_lists, num_nodes, parts=parts)
        _GRAPH_CACHE[key] = (g, adjacency_lists)
        while len(_GRAPH_CACHE) > _GRAPH_CACHE_SIZE:
            _GRAPH_CACHE.popitem(last=False)
    else:
        _GRAPH_CACHE.move_to_end(key)
        g = entry[0]
    return g


def clear_graph_cache():
    _GRAPH_CACHE.clear()


class MessagePassing:
    """Abstract class to compute new graph states by neural message passing
    (message_passing.py:20-218).

    Users create a specific type of message passing layer by implementing ``_message_function``
    (and optionally overriding ``_compute_new_node_embeddings``), exactly as in the reference.
    A subclass that only implements ``_message_function`` runs on the generic path: per-edge source
    / target states are gathered by the HIP gather kernel and handed to the user function, whose
    result is aggregated by the HIP segment kernels.  The built-in subclasses override ``call``
    with fused node-side formulations.

    Shape abbreviations as in the reference: V nodes, D input dim, L edge types, E edges of a
    type, H = hidden_dim.
    """

    @classmethod
    def get_default_hyperparameters(cls):
        return {
            "aggregation_function": "sum",  # One of sum, mean, max, sqrt_n
            "message_activation_function": "relu",  # One of relu, leaky_relu, elu, gelu, tanh
            "message_activation_before_aggregation": False,  # Change to True to apply activation _before_ aggregation.
            "hidden_dim": 7,
        }

    def graph_parts(self, num_nodes: int, edges_per_type, in_dim: int) -> int:
        """Which derived tables of the batch's graph handle this layer reads (ops.G_PART_*), for a batch of that shape: a
        stack whose layers need only some of them skips the preparation kernels of the rest (include/tfgnn.h
        tfgnn_graph_create_parts_async).  A hint, not a contract: a missing part is built on first use.  Default: all."""
        return ops.G_PARTS_DEFAULT

    def __init__(self, params: Dict[str, Any], **kwargs):
        self.name = kwargs.get("name", type(self).__name__)
        self._hidden_dim = int(params["hidden_dim"])

        aggregation_fn_name = params["aggregation_function"]
        self._aggregation_fn = get_aggregation_function(aggregation_fn_name)
        self._aggregation_name = aggregation_fn_name

        self._message_activation_before_aggregation = params.get(
            "message_activation_before_aggregation", False
        )

        activation_fn_name = params["message_activation_function"]
        self._activation_fn = get_activation_function(activation_fn_name)
        self._activation_name = None if activation_fn_name is None else activation_fn_name.lower()

        self.built = False
        self._variables: List[Variable] = []
        self._ctx = None  # saved tensors of the last recorded forward

    # ---- Keras-like plumbing ----------------------------------------------------------------
    @property
    def trainable_variables(self) -> List[Variable]:
        return [v for v in self._variables if v.trainable]

    @property
    def variables(self) -> List[Variable]:
        return list(self._variables)

    def add_weight(self, name: str, value: torch.Tensor, trainable: bool = True) -> Variable:
        v = Variable(name, value, trainable)
        self._variables.append(v)
        return v

    def zero_grad(self):
        for v in self._variables:
            v.grad = None

    def build(self, input_shapes: MessagePassingInput):
        self.built = True

    def __call__(self, inputs: MessagePassingInput, training: bool = False):
        if not self.built:
            self.build(
                MessagePassingInput(
                    tuple(inputs.node_embeddings.shape),
                    tuple((None, 2) for _ in range(_num_edge_types(inputs.adjacency_lists))),
                )
            )
        return self.call(inputs, training=training)

    # ---- the reference's extension points ---------------------------------------------------
    @abstractmethod
    def _message_function(
        self,
        edge_source_states: torch.Tensor,
        edge_target_states: torch.Tensor,
        num_incoming_to_node_per_message: torch.Tensor,
        edge_type_idx: int,
        training: bool,
    ) -> torch.Tensor:
        """Calculate the messages passed from source nodes to target nodes for ONE edge type
        (message_passing.py:64-93): [E, D], [E, D], [E] -> [E, H]."""

    def call(self, inputs: MessagePassingInput, training: bool = False):
        """message_passing.py:95-133 (generic path).  Returns float32 [V, hidden_dim].

        The user's ``_message_function`` runs under torch autograd on the gathered per-edge states (leaf tensors) and the
        layer's variables; gather, aggregation and activation are this library's kernels.  ``backward`` then needs no
        code from the subclass (the reference gets the same from tf.GradientTape)."""
        node_embeddings, adjacency_lists = inputs.node_embeddings, inputs.adjacency_lists
        if isinstance(adjacency_lists, ops.Graph):
            raise ValueError("the generic MessagePassing path needs the adjacency list tensors")
        num_nodes = node_embeddings.shape[0]
        self._release_tape()  # a previous training-mode call whose backward never ran
        # the autograd tape (per-edge states as leaves, variables with requires_grad) is only recorded when a backward
        # pass can follow: training mode, or ``record_tape_in_eval = True`` on the layer (gradient checks in eval mode)
        record_tape = bool(training) or bool(getattr(self, "record_tape_in_eval", False))
        record = {"generic": True, "per_type": []} if record_tape else {"generic": True}
        self._generic_record = record if record_tape else None
        try:
            messages_per_type = self._calculate_messages_per_type(adjacency_lists, node_embeddings, training)
        finally:
            self._generic_record = None
        edge_type_to_message_targets = [adj[:, 1] for adj in adjacency_lists]
        out = self._compute_new_node_embeddings(
            node_embeddings, [m.detach() for m in messages_per_type], edge_type_to_message_targets, num_nodes, training
        )
        record.update({"X": node_embeddings, "adjacency_lists": adjacency_lists, "messages": messages_per_type, "out": out})
        self._ctx = record
        return out

    def _compute_new_node_embeddings(
        self,
        cur_node_embeddings: torch.Tensor,
        messages_per_type: List[torch.Tensor],
        edge_type_to_message_targets: List[torch.Tensor],
        num_nodes: int,
        training: bool,
    ):
        """message_passing.py:135-179: concat, (activation), aggregation, (activation)."""
        message_targets = torch.cat(list(edge_type_to_message_targets), dim=0)  # [M]
        messages = torch.cat(list(messages_per_type), dim=0)  # [M, H]
        if self._message_activation_before_aggregation:
            messages = self._activation_fn(messages)
        aggregated_messages = self._aggregation_fn(
            data=messages, segment_ids=message_targets, num_segments=num_nodes
        )
        if not self._message_activation_before_aggregation:
            aggregated_messages = self._activation_fn(aggregated_messages)
        return aggregated_messages

    def _calculate_messages_per_type(self, adjacency_lists, node_embeddings, training: bool = False):
        """message_passing.py:181-218; the three embedding_lookups are HIP row gathers."""
        messages_per_type = []
        type_to_num_incoming_edges = calculate_type_to_num_incoming_edges(node_embeddings, adjacency_lists)
        for edge_type_idx, adj in enumerate(adjacency_lists):
            edge_sources = adj[:, 0].contiguous()
            edge_targets = adj[:, 1].contiguous()
            edge_source_states = gather_rows(node_embeddings, edge_sources)
            edge_target_states = gather_rows(node_embeddings, edge_targets)
            num_incoming = gather_rows(
                type_to_num_incoming_edges[edge_type_idx].reshape(-1, 1), edge_targets
            ).reshape(-1)
            record = getattr(self, "_generic_record", None)
            if record is None:
                messages_per_type.append(
                    self._message_function(edge_source_states, edge_target_states, num_incoming, edge_type_idx, training)
                )
                continue
            # recorded forward: the states are autograd leaves, the variables' tensors take part in the tape
            edge_source_states.requires_grad_(True)
            edge_target_states.requires_grad_(True)
            for v in self._variables:
                if v.trainable and not v.value.requires_grad:
                    v.value.requires_grad_(True)
            with torch.enable_grad():
                m = self._message_function(edge_source_states, edge_target_states, num_incoming, edge_type_idx, training)
            record["per_type"].append((edge_source_states, edge_target_states))
            messages_per_type.append(m)
        return messages_per_type

    def _release_tape(self):
        """Drop the autograd tape of the generic path and make the variables plain tensors again (in-place optimizer
        updates of a leaf that requires grad raise outside no_grad)."""
        ctx = getattr(self, "_ctx", None)
        if isinstance(ctx, dict) and "per_type" in ctx:
            ctx.pop("per_type", None)
            ctx.pop("messages", None)
        for v in getattr(self, "_variables", []):
            if v.trainable and v.value.requires_grad:
                v.value.grad = None
                v.value.requires_grad_(False)

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        """Generic backward: d out / d messages through this library's activation / aggregation kernels, the user's
        ``_message_function`` through torch autograd, the two per-edge state gradients scattered back to the nodes by the
        gather kernel.  Fills ``Variable.grad`` and returns d(node_embeddings).  Needs a forward pass that recorded the
        tape: ``training=True`` (or ``layer.record_tape_in_eval = True``)."""
        ctx = self._ctx
        if ctx is None or "per_type" not in ctx:
            raise RuntimeError("backward called before a forward pass in training mode (the generic path records its "
                               "autograd tape only when training=True or layer.record_tape_in_eval is set)")
        try:
            return self._generic_backward(ctx, grad_output)
        finally:
            self._release_tape()

    def _generic_backward(self, ctx, grad_output: torch.Tensor) -> torch.Tensor:
        if type(self)._compute_new_node_embeddings is not MessagePassing._compute_new_node_embeddings:
            raise NotImplementedError(
                f"{type(self).__name__} overrides _compute_new_node_embeddings; the generic backward covers the base class "
                "aggregation (message_passing.py:135-179) - implement backward() next to the override"
            )
        X, adjacency_lists, messages_per_type = ctx["X"], ctx["adjacency_lists"], ctx["messages"]
        V, H = X.shape[0], self._hidden_dim
        g = get_graph(adjacency_lists, V)
        E = g.num_edges
        dX = torch.zeros_like(X)
        for v in self._variables:
            v.grad = None
        if E == 0:
            for v in self._variables:
                if v.trainable:
                    v.grad = torch.zeros_like(v.value)
            return dX
        from ..graph_scales import graph_scales

        act = self._activation_name
        pre = act if self._message_activation_before_aggregation else None
        is_max = self._aggregation_name == "max"
        node_scale = graph_scales(g, False, self._aggregation_name)[3]
        eid_d = g.array(ops.G_EID_BY_DST)
        messages = torch.cat([m.detach() for m in messages_per_type], dim=0).contiguous()  # edge-list order
        target = torch.cat([a[:, 1] for a in adjacency_lists], dim=0).to(torch.int32).contiguous()
        agg_raw = None
        if is_max or (pre is None and act == "gelu"):  # the raw aggregate: maxima for the tie split, gelu's argument
            agg_raw = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, messages, col=eid_d, row_scale=node_scale,
                                       reduce=ops.REDUCE_MAX if is_max else ops.REDUCE_SUM, pre_act=pre)
        d_agg = grad_output.contiguous()
        if pre is None and act is not None:
            d_agg = ops.activation_backward(act, d_agg, agg_raw if act == "gelu" else ctx["out"])
        if is_max:
            sel = ops.edge_aggregate_backward(messages, target, None, pre_act=pre, reduce=ops.REDUCE_MAX, agg_max=agg_raw, phase=0)
            nsel = ops.graph_gather(g, ops.VIEW_BY_DST_NODE, sel, col=eid_d)
            dM = ops.edge_aggregate_backward(messages, target, d_agg, pre_act=pre, reduce=ops.REDUCE_MAX, agg_max=agg_raw,
                                             num_selected=nsel)
        else:
            dM = ops.edge_aggregate_backward(messages, target, d_agg, node_scale=node_scale, pre_act=pre, reduce=ops.REDUCE_SUM)
        d_src, d_tgt, off = [], [], 0
        leaves = [v.value for v in self._variables if v.trainable]
        for (xs, xt), m in zip(ctx["per_type"], messages_per_type):
            n = m.shape[0]
            if n and m.grad_fn is None:
                raise NotImplementedError(
                    f"{type(self).__name__}._message_function returned a tensor without an autograd history (built from "
                    "kernels outside torch autograd?): write it with torch operations, or override backward()"
                )
            if n:
                torch.autograd.backward([m], [dM[off : off + n]], inputs=[xs, xt] + leaves)
            d_src.append(xs.grad if xs.grad is not None else torch.zeros_like(xs))
            d_tgt.append(xt.grad if xt.grad is not None else torch.zeros_like(xt))
            off += n
        # d X[u] += sum over out-edges of d(source state); d X[v] += sum over in-edges of d(target state)
        dX = ops.graph_gather(g, ops.VIEW_BY_SRC_NODE, torch.cat(d_src, dim=0).contiguous(), col=g.array(ops.G_EID_BY_SRC))
        dX = ops.add_scale(dX, ops.graph_gather(g, ops.VIEW_BY_DST_NODE, torch.cat(d_tgt, dim=0).contiguous(), col=eid_d), 1.0)
        for v in self._variables:
            if v.trainable:
                v.grad = v.value.grad if v.value.grad is not None else torch.zeros_like(v.value)
        return dX  # backward() releases the tape and resets requires_grad

    # ---- hooks that let the layer stack fold element-wise backward steps into this layer's last GEMM ----------
    def activation_backward_spec(self):
        """(activation name, saved tensor) if backward() starts with activation_backward(name, grad, saved) and can
        skip it when handed the already multiplied gradient (grad_is_pre_activation); None otherwise."""
        return None

    def recomputes_input_dropout(self, num_nodes: int, in_dim: int, num_edge_types: int) -> bool:
        """Will backward_with_epilogue apply the dropout mask of this layer's input by RECOMPUTING it from (rate, seed) in a
        product epilogue?  Then the layer stack does not store the mask (ops.dropout_forward(want_mask=False)) and hands a
        DropoutSpec down.  A layer that answers True and then takes another route materialises the mask once
        (ops.plain_epilogue): correct, one pass slower."""
        return False

    def backward_with_epilogue(self, grad_output, grad_is_pre_activation=False, out_mul=None, out_act_grad=None):
        """backward(), then d(node_embeddings) * out_mul * act'(saved) - the element-wise factors the caller would
        apply next (dropout mask of this layer's input, activation derivative of the layer below).  Generic form:
        separate kernels; GNN_Edge_MLP folds them into its input-gradient GEMM."""
        if grad_is_pre_activation:
            raise ValueError(f"{type(self).__name__} does not accept an already activation-multiplied gradient")
        return apply_gradient_epilogue(self.backward(grad_output), out_mul, out_act_grad)


def apply_gradient_epilogue(g, out_mul, out_act_grad):
    out_mul, out_act_grad = ops.plain_epilogue(out_mul, out_act_grad)
    if out_mul is not None and out_act_grad is not None:
        return ops.activation_backward(out_act_grad[0], g, out_act_grad[1], mul=out_mul)  # one pass, same bits
    if out_mul is not None:
        g = ops.mul(g, out_mul)
    if out_act_grad is not None:
        g = ops.activation_backward(out_act_grad[0], g, out_act_grad[1])
    return g


def _num_edge_types(adjacency_lists) -> int:
    if isinstance(adjacency_lists, ops.Graph):
        return adjacency_lists.num_edge_types
    return len(adjacency_lists)


def gather_rows(params: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """tf.nn.embedding_lookup(params, ids) (message_passing.py:197-206) through the gather kernel
    (identity row pointer: one edge per output row)."""
    n = ids.shape[0]
    if params.shape[0] == 0 and n:
        raise ValueError("gather from an empty tensor")
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=params.device)
    if n and (int(ids.min()) < 0 or int(ids.max()) >= params.shape[0]):
        raise ValueError("index out of range in gather")  # TF: InvalidArgumentError
    return ops.gather_reduce(rowptr, ids.to(torch.int32).contiguous(), params.reshape(params.shape[0], -1))


MESSAGE_PASSING_IMPLEMENTATIONS: Dict[str, type] = {}


def register_message_passing_implementation(cls):
    """Decorator used to register a message passing class implementation (message_passing.py:224-227)."""
    MESSAGE_PASSING_IMPLEMENTATIONS[cls.__name__.lower()] = cls
    return cls


def calculate_type_to_num_incoming_edges(node_embeddings, adjacency_lists):
    """float32 tensor [L, V]: number of incoming edges of each type per node
    (message_passing.py:230-263).  Here it is read off the bucketed graph: the count for (l, v) is
    the length of row v*L+l."""
    num_nodes = node_embeddings.shape[0]
    g = get_graph(adjacency_lists, num_nodes)
    L = g.num_edge_types
    rowptr = g.array(ops.G_ROWPTR_BY_DST)
    counts = (rowptr[1:] - rowptr[:-1]).reshape(num_nodes, L)
    return counts.t().to(torch.float32).contiguous()
