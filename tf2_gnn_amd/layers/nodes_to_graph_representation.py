"""Graph representation aggregation layer - mirror of
tf2_gnn/layers/nodes_to_graph_representation.py (NodesToGraphRepresentationInput,
WeightedSumGraphRepresentation, WASGraphRepresentation; same constructor arguments)."""
from __future__ import annotations

from abc import abstractmethod
from typing import List, NamedTuple, Optional

import torch

from .. import _lib, ops
from .message_passing.message_passing import Variable, default_device, glorot_uniform


class NodesToGraphRepresentationInput(NamedTuple):
    """Input to layers computing graph representations from node representations
    (nodes_to_graph_representation.py:9-15)."""

    node_embeddings: torch.Tensor
    node_to_graph_map: torch.Tensor
    num_graphs: int


def get_activation_function_by_name(name: Optional[str]) -> Optional[str]:
    """[ext] dpu_utils.tf2utils.get_activation_function_by_name: canonical lower-case name or None
    ("linear"/None = identity)."""
    if name is None:
        return None
    name = name.lower()
    if name == "linear":
        return None
    if name not in ("relu", "tanh", "leaky_relu", "elu", "selu", "gelu", "sigmoid"):
        raise ValueError(f"Unknown activation function: {name}")
    return name


def prefix_names(variables, prefix: str) -> None:
    """What building under ``with tf.name_scope(prefix)`` does to the variables' names."""
    for v in variables:
        v.name = f"{prefix}/{v.name}"


class MLP:
    """[ext] dpu_utils.tf2utils.MLP on the HIP GEMMs: hidden Dense layers (activation), final linear Dense; optional
    biases; in training mode tf.nn.dropout(rate) on the input of every HIDDEN Dense layer (call sites:
    nodes_to_graph_representation.py:130-148 with rate 0.2 by default, graph_global_exchange.py:171-183,
    models/graph_regression_task.py:65-69).  [ext] dpu_utils is not installed here, so which layer inputs are dropped
    is unpinned; the choice "all but the final layer" is the one under which the reference's own QM9 task works:
    models/qm9_regression.py:49-62 passes out_layer_dropout_keep_prob = 1.0 as the dropout RATE of two MLPs without
    hidden layers - tf.nn.dropout(rate=1.0) on their inputs would raise (rate must be in [0, 1)) or zero them.
    ``dropout_masks`` (one tensor per Dense layer, already scaled by 1/(1-rate), or None entries) replaces the drawn
    masks - the parity tests hand the oracle the same masks."""

    _seed_counter = [0]

    def __init__(self, out_size, hidden_layers, use_biases=False, activation_fun="relu", dropout_rate=0.0, name="MLP"):
        self._out_size = out_size
        self._sizes = ([out_size] * hidden_layers if isinstance(hidden_layers, int) else list(hidden_layers)) + [out_size]
        self._use_biases = use_biases
        self._act = activation_fun
        self._dropout_rate = float(dropout_rate)
        self._name = name
        self.kernels: List[Variable] = []
        self.biases: List[Optional[Variable]] = []
        self.dropout_seed = 0
        self.last_dropout_masks: List[Optional[torch.Tensor]] = []
        self._ctx = None

    def build(self, in_size: int):
        """Variable names: the dpu_utils MLP [ext] builds every Dense under tf.name_scope(f"{name}_dense_{i}") /
        f"{name}_final_layer"; callers prefix the scopes they build it under (``prefix_names``)."""
        dev = default_device()
        last = in_size
        for j, size in enumerate(self._sizes):
            tag = f"dense_{j}" if j < len(self._sizes) - 1 else "final_layer"
            self.kernels.append(Variable(f"{self._name}_{tag}/kernel", glorot_uniform((last, size), device=dev)))
            self.biases.append(
                Variable(f"{self._name}_{tag}/bias", torch.zeros(size, dtype=torch.float32, device=dev))
                if self._use_biases
                else None
            )
            last = size

    @property
    def variables(self):
        out = []
        for k, b in zip(self.kernels, self.biases):
            out.append(k)
            if b is not None:
                out.append(b)
        return out

    def __call__(self, x, final_act=None, training: bool = False, dropout_masks=None):
        """-> MLP(x), optionally with ``final_act`` applied to the output."""
        ins, outs, pres, masks = [], [], [], []
        cur = x
        n = len(self.kernels)
        for j, (k, b) in enumerate(zip(self.kernels, self.biases)):
            mask = None
            if dropout_masks is not None:
                mask = dropout_masks[j]
                if mask is not None:
                    cur = ops.mul(cur, mask)
            elif training and self._dropout_rate > 0.0 and j < n - 1:
                if self._dropout_rate >= 1.0:
                    raise ValueError("rate must be a scalar tensor or a float in the range [0, 1), got %g" % self._dropout_rate)
                MLP._seed_counter[0] += 1
                cur, mask = ops.dropout_forward(cur, self._dropout_rate, self.dropout_seed * 1000003 + MLP._seed_counter[0])
            masks.append(mask)
            ins.append(cur)  # the (dropped) input of Dense layer j
            act = self._act if j < n - 1 else final_act
            bias = None if b is None else b.value
            if act == "gelu":
                pre = ops.gemm(cur, k.value, bias=bias)
                cur = ops.activation_forward("gelu", pre)
                pres.append(pre)
            else:
                cur = ops.gemm(cur, k.value, bias=bias, act=act)
                pres.append(None)
            outs.append(cur)  # the output of layer j (before the next layer's dropout)
        self.last_dropout_masks = masks
        self._ctx = (ins, outs, pres, final_act, masks)
        return cur

    def backward(self, grad):
        ins, outs, pres, final_act, masks = self._ctx
        n = len(self.kernels)
        d = grad
        for j in range(n - 1, -1, -1):
            act = self._act if j < n - 1 else final_act
            if act is not None:
                d = ops.activation_backward(act, d, pres[j] if act == "gelu" else outs[j])
            self.kernels[j].grad = ops.gemm(ins[j], d, trans_a=True)
            if self.biases[j] is not None:
                self.biases[j].grad = ops.colsum(d)
            d = ops.gemm(d, self.kernels[j].value, trans_b=True)
            if masks[j] is not None:
                d = ops.mul(d, masks[j])
        return d


class NodesToGraphRepresentation:
    """Abstract class to compute graph representations from node representations
    (nodes_to_graph_representation.py:18-48).  V nodes, VD node dim, G graphs, GD graph dim."""

    def __init__(self, graph_representation_size: int, **kwargs):
        self._graph_representation_size = graph_representation_size
        self.built = False

    def build(self, input_shapes: NodesToGraphRepresentationInput):
        self.built = True

    def __call__(self, inputs: NodesToGraphRepresentationInput, training: bool = False):
        if not self.built:
            self.build(NodesToGraphRepresentationInput(tuple(inputs.node_embeddings.shape), (None,), ()))
        return self.call(inputs, training)

    @abstractmethod
    def call(self, inputs: NodesToGraphRepresentationInput, training: bool = False):
        """-> float32 [G, GD]"""


_OFFSETS_CACHE: "dict" = {}


def segment_offsets(node_to_graph_map: torch.Tensor, num_graphs: int) -> torch.Tensor:
    """ptr [G + 1] of the sorted node_to_graph_map (tfgnn_segment_offsets_async), computed and VALIDATED once per batch:
    the pooling layers, the global exchanges and the task head of a step all ask for the same map - the entry is keyed
    on the tensor (address, length, version, G) and keeps it alive; only the first request reads the error word back
    (unsorted / out-of-range ids raise like tf.math.segment_sum's InvalidArgument).  No allocation or synchronisation
    inside the library."""
    ids = node_to_graph_map
    if ids.dtype != torch.int32 or not ids.is_contiguous():
        ids = ids.to(torch.int32).contiguous()
    key = (ids.data_ptr(), ids.numel(), ids._version, int(num_graphs), str(ids.device))
    hit = _OFFSETS_CACHE.get(key)
    if hit is not None:
        return hit[0]
    ptr = torch.empty(num_graphs + 1, dtype=torch.int32, device=ids.device)
    flag = torch.zeros(1, dtype=torch.int32, device=ids.device)
    _lib.check(_lib.load().tfgnn_segment_offsets_async(ops._ptr(ids), ids.numel(), num_graphs, ops._ptr(ptr), ops._ptr(flag),
                                                       ops._stream()))
    err = int(flag.item())  # the one read-back per batch
    if err & 1:
        raise ValueError(f"tfgnn: node_to_graph_map contains an id outside [0, {num_graphs})")
    if err & 2:
        raise ValueError("tfgnn: node_to_graph_map is not sorted (tf.math.segment_sum requires sorted segment ids)")
    if len(_OFFSETS_CACHE) >= 8:
        _OFFSETS_CACHE.pop(next(iter(_OFFSETS_CACHE)))
    _OFFSETS_CACHE[key] = (ptr, ids)
    return ptr


class WeightedSumGraphRepresentation(NodesToGraphRepresentation):
    """Graph representations as weighted sum of node representations
    (nodes_to_graph_representation.py:51-229): weights from a scoring MLP through a sigmoid or a
    per-graph softmax, per head; "average" / "none" use fixed weights."""

    def __init__(
        self,
        graph_representation_size: int,
        num_heads: int,
        weighting_fun: str = "softmax",  # One of {"softmax", "sigmoid"}
        scoring_mlp_layers: List[int] = [128],
        scoring_mlp_activation_fun: str = "ReLU",
        scoring_mlp_use_biases: bool = False,
        scoring_mlp_dropout_rate: float = 0.2,
        transformation_mlp_layers: List[int] = [128],
        transformation_mlp_activation_fun: str = "ReLU",
        transformation_mlp_use_biases: bool = False,
        transformation_mlp_dropout_rate: float = 0.2,
        transformation_mlp_result_lower_bound: Optional[float] = None,
        transformation_mlp_result_upper_bound: Optional[float] = None,
        **kwargs,
    ):
        super().__init__(graph_representation_size, **kwargs)
        assert (
            graph_representation_size % num_heads == 0
        ), f"Number of heads {num_heads} needs to divide final representation size {graph_representation_size}!"
        assert weighting_fun.lower() in {
            "none",
            "average",
            "softmax",
            "sigmoid",
        }, f"Weighting function {weighting_fun} unknown, {{'softmax', 'sigmoid', 'none', 'average'}} supported."
        self._num_heads = num_heads
        self._weighting_fun = weighting_fun.lower()
        self._transformation_mlp_activation_fun = get_activation_function_by_name(transformation_mlp_activation_fun)
        self._transformation_mlp_result_lower_bound = transformation_mlp_result_lower_bound
        self._transformation_mlp_result_upper_bound = transformation_mlp_result_upper_bound
        if self._weighting_fun not in ("none", "average"):
            self._scoring_mlp = MLP(
                out_size=self._num_heads,
                hidden_layers=scoring_mlp_layers,
                use_biases=scoring_mlp_use_biases,
                activation_fun=get_activation_function_by_name(scoring_mlp_activation_fun),
                dropout_rate=scoring_mlp_dropout_rate,
                name="ScoringMLP",
            )
        self._transformation_mlp = MLP(
            out_size=self._graph_representation_size,
            hidden_layers=transformation_mlp_layers,
            use_biases=transformation_mlp_use_biases,
            activation_fun=self._transformation_mlp_activation_fun,
            dropout_rate=transformation_mlp_dropout_rate,
            name="TransformationMLP",
        )
        self._ctx = None
        # tests: {"scoring": [mask per Dense layer], "transformation": [...]} replaces the drawn dropout masks
        self.dropout_masks = None

    def build(self, input_shapes: NodesToGraphRepresentationInput):
        vd = int(input_shapes.node_embeddings[-1])
        if self._weighting_fun not in ("none", "average"):
            self._scoring_mlp.build(vd)
        self._transformation_mlp.build(vd)
        prefix_names(self.trainable_variables, "WeightedSumGraphRepresentation")  # nodes_to_graph_representation.py:151
        super().build(input_shapes)

    @property
    def trainable_variables(self):
        out = []
        if self._weighting_fun not in ("none", "average"):
            out.extend(self._scoring_mlp.variables)
        out.extend(self._transformation_mlp.variables)
        return out

    def call(self, inputs: NodesToGraphRepresentationInput, training: bool = False):
        X = inputs.node_embeddings
        V = X.shape[0]
        G = int(inputs.num_graphs)
        GD, heads = self._graph_representation_size, self._num_heads
        lib = _lib.load()
        ids = inputs.node_to_graph_map.to(torch.int32).contiguous()
        ptr = segment_offsets(ids, G)
        masks = self.dropout_masks or {}
        w = None
        if self._weighting_fun == "sigmoid":
            w = self._scoring_mlp(X, final_act="sigmoid", training=training, dropout_masks=masks.get("scoring"))  # [V, heads]
        elif self._weighting_fun == "softmax":
            scores = self._scoring_mlp(X, training=training, dropout_masks=masks.get("scoring"))
            w = torch.empty_like(scores)
            _lib.check(
                lib.tfgnn_segment_softmax(ops._ptr(scores), heads, heads, ops._ptr(ptr), G, ops._ptr(w), heads, ops._stream())
            )
        # nodes_to_graph_representation.py:191-193: the activation is applied to the MLP *output* too
        R = self._transformation_mlp(X, final_act=self._transformation_mlp_activation_fun, training=training,
                                     dropout_masks=masks.get("transformation"))  # [V, GD]
        lo, hi = self._transformation_mlp_result_lower_bound, self._transformation_mlp_result_upper_bound
        R_unclipped = None
        if lo is not None or hi is not None:  # :194-197
            R_unclipped = R
            R = ops.clip(R, lo, hi)
        out = torch.empty((G, GD), dtype=torch.float32, device=X.device)
        _lib.check(
            lib.tfgnn_segment_weighted_sum(
                ops._ptr(R), ops._ptr(w), ops._ptr(ptr), G, GD, heads, int(self._weighting_fun == "average"),
                ops._ptr(out), ops._stream(),
            )
        )
        self._ctx = {"ids": ids, "ptr": ptr, "w": w, "R": R, "V": V, "G": G, "R_unclipped": R_unclipped}
        return out

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        """d(loss)/d(graph representations) [G, GD] -> d(loss)/d(node_embeddings) [V, VD]."""
        c = self._ctx
        lib = _lib.load()
        GD, heads = self._graph_representation_size, self._num_heads
        V, G = c["V"], c["G"]
        g = grad_output.contiguous()
        dR = torch.empty((V, GD), dtype=torch.float32, device=g.device)
        w = c["w"]
        dW = torch.empty((V, heads), dtype=torch.float32, device=g.device) if w is not None else None
        _lib.check(
            lib.tfgnn_segment_weighted_sum_backward(
                ops._ptr(g), ops._ptr(c["R"]) if w is not None else None, ops._ptr(w), ops._ptr(c["ids"]),
                ops._ptr(c["ptr"]), V, GD, heads, int(self._weighting_fun == "average"), ops._ptr(dR), ops._ptr(dW),
                ops._stream(),
            )
        )
        if c["R_unclipped"] is not None:
            dR = ops.clip_backward(dR, c["R_unclipped"], self._transformation_mlp_result_lower_bound,
                                   self._transformation_mlp_result_upper_bound)
        dX = self._transformation_mlp.backward(dR)
        if self._weighting_fun == "sigmoid":
            dXs = self._scoring_mlp.backward(dW)  # sigmoid handled as the MLP's final activation
            dX = ops.add_scale(dX, dXs, 1.0)
        elif self._weighting_fun == "softmax":
            ds = torch.empty_like(dW)
            _lib.check(
                lib.tfgnn_segment_softmax_backward(ops._ptr(w), ops._ptr(dW), heads, ops._ptr(c["ptr"]), G, ops._ptr(ds), ops._stream())
            )
            dXs = self._scoring_mlp.backward(ds)
            dX = ops.add_scale(dX, dXs, 1.0)
        return dX


class WASGraphRepresentation(NodesToGraphRepresentation):
    """_W_eighted _A_verage and _S_um graph representation (nodes_to_graph_representation.py:232-314)."""

    def __init__(
        self,
        graph_representation_size: int = 128,
        num_heads: int = 8,
        pooling_mlp_layers: List[int] = [128, 128],
        pooling_mlp_activation_fun: str = "elu",
        pooling_mlp_use_biases: bool = True,
        pooling_mlp_dropout_rate: float = 0.0,
        **kwargs,
    ):
        super().__init__(graph_representation_size, **kwargs)
        common = dict(
            graph_representation_size=graph_representation_size,
            num_heads=num_heads,
            scoring_mlp_layers=pooling_mlp_layers,
            scoring_mlp_dropout_rate=pooling_mlp_dropout_rate,
            scoring_mlp_use_biases=pooling_mlp_use_biases,
            scoring_mlp_activation_fun=pooling_mlp_activation_fun,
            transformation_mlp_layers=pooling_mlp_layers,
            transformation_mlp_dropout_rate=pooling_mlp_dropout_rate,
            transformation_mlp_use_biases=pooling_mlp_use_biases,
            transformation_mlp_activation_fun=pooling_mlp_activation_fun,
        )
        self._weighted_avg_graph_repr_layer = WeightedSumGraphRepresentation(weighting_fun="softmax", **common)
        self._weighted_sum_graph_repr_layer = WeightedSumGraphRepresentation(weighting_fun="sigmoid", **common)
        self._out_projection: Optional[Variable] = None
        self._ctx = None

    def build(self, input_shapes: NodesToGraphRepresentationInput):
        # nodes_to_graph_representation.py:281-291: scopes <class name>/Weighted{Avg,Sum}GraphRepresentation/...
        cls = self.__class__.__name__
        self._weighted_avg_graph_repr_layer.build(input_shapes)
        prefix_names(self._weighted_avg_graph_repr_layer.trainable_variables, f"{cls}/WeightedAvgGraphRepresentation")
        self._weighted_sum_graph_repr_layer.build(input_shapes)
        prefix_names(self._weighted_sum_graph_repr_layer.trainable_variables, f"{cls}/WeightedSumGraphRepresentation")
        GD = self._graph_representation_size
        self._out_projection = Variable(f"{cls}/kernel", glorot_uniform((2 * GD, GD), device=default_device()))
        super().build(input_shapes)

    @property
    def trainable_variables(self):
        return (
            self._weighted_avg_graph_repr_layer.trainable_variables
            + self._weighted_sum_graph_repr_layer.trainable_variables
            + [self._out_projection]
        )

    def call(self, inputs: NodesToGraphRepresentationInput, training: bool = False):
        GD = self._graph_representation_size
        G = int(inputs.num_graphs)
        cat = torch.empty((G, 2 * GD), dtype=torch.float32, device=inputs.node_embeddings.device)
        avg = self._weighted_avg_graph_repr_layer(inputs, training)
        summed = self._weighted_sum_graph_repr_layer(inputs, training)
        # tf.concat([avg, sum], axis=-1) (:311-313): plain device copies
        cat[:, :GD].copy_(avg)
        cat[:, GD:].copy_(summed)
        self._ctx = cat
        return ops.gemm(cat, self._out_projection.value)

    def backward(self, grad_output: torch.Tensor) -> torch.Tensor:
        GD = self._graph_representation_size
        cat = self._ctx
        self._out_projection.grad = ops.gemm(cat, grad_output, trans_a=True)
        dcat = ops.gemm(grad_output, self._out_projection.value, trans_b=True)
        da = self._weighted_avg_graph_repr_layer.backward(dcat[:, :GD].contiguous())
        ds = self._weighted_sum_graph_repr_layer.backward(dcat[:, GD:].contiguous())
        return ops.add_scale(da, ds, 1.0)
