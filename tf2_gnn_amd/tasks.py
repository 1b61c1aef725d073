"""Task models around the path: what the reference wraps the GNN stack in, on the same
kernels - the computation between a finalised batch and the loss, and back.

Mirrors ``tf2_gnn.models`` for that computation only: ``compute_final_node_representations``,
``compute_task_output``, ``compute_task_metrics`` keep the reference's names, arguments (dicts keyed like the batches
of ``GraphDataset._finalise_batch``: "node_features", "adjacency_list_<i>", "node_to_graph_map",
"num_graphs_in_batch"; labels "node_labels" / "target_value") and result keys.  ``backward()`` stands in for the
``tf.GradientTape`` of ``GraphTaskModel._run_step`` (tf2_gnn/models/graph_task_model.py:327-357): it fills ``.grad``
of every trainable variable with d loss / d variable.  The optimizer, the epoch loop, datasets and checkpoint I/O are
the reference's control plane and stay out (out of scope).

MLP-input dropout of the heads (``regression_mlp_dropout``, ``graph_aggregation_dropout_rate``; QM9 hands
``out_layer_dropout_keep_prob`` over as a rate) is applied in training mode as in the pooling layers (layers/nodes_to_graph_representation.py MLP).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from .layers.gnn import GNN, GNNInput
from .layers.message_passing.message_passing import Variable, default_device, glorot_uniform
from .layers.nodes_to_graph_representation import (
    prefix_names,
    MLP,
    NodesToGraphRepresentationInput,
    WeightedSumGraphRepresentation,
    segment_offsets,
)


class GraphTaskModel:
    """tf2_gnn/models/graph_task_model.py:14-222 without the Keras / optimizer / epoch-loop plumbing."""

    @classmethod
    def get_default_hyperparameters(cls, mp_style: Optional[str] = None) -> Dict[str, Any]:
        """graph_task_model.py:16-33 (the optimizer keys are kept: they are part of the dict callers pass around)."""
        params = {f"gnn_{name}": value for name, value in GNN.get_default_hyperparameters(mp_style).items()}
        params.update(
            {
                "optimizer": "Adam",
                "learning_rate": 0.001,
                "learning_rate_warmup_steps": None,
                "learning_rate_decay_steps": None,
                "momentum": 0.85,
                "rmsprop_rho": 0.98,
                "gradient_clip_value": None,
                "gradient_clip_norm": None,
                "gradient_clip_global_norm": None,
                "use_intermediate_gnn_results": False,
            }
        )
        return params

    def __init__(self, params: Dict[str, Any], dataset: Any = None, name: Optional[str] = None, *,
                 num_edge_types: Optional[int] = None):
        """``dataset``: anything with ``num_edge_types`` (graph_task_model.py:41-43), or pass ``num_edge_types``."""
        self._params = params
        if num_edge_types is None:
            if dataset is None or not hasattr(dataset, "num_edge_types"):
                raise ValueError("GraphTaskModel needs a dataset with num_edge_types, or num_edge_types=...")
            num_edge_types = dataset.num_edge_types
        self._num_edge_types = int(num_edge_types)
        self._use_intermediate_gnn_results = params.get("use_intermediate_gnn_results", False)
        self.name = name or self.__class__.__name__
        self._gnn: Optional[GNN] = None
        self.built = False
        self._step = None  # what backward() needs from the last forward / metrics call

    # ---- structure (graph_task_model.py:93-131) --------------------------------------------------------------
    def build(self, input_shapes: Dict[str, Any]):
        graph_params = {name[4:]: value for name, value in self._params.items() if name.startswith("gnn_")}
        self._gnn = GNN(graph_params)
        self._gnn.build(
            GNNInput(
                node_features=self.get_initial_node_feature_shape(input_shapes),
                adjacency_lists=tuple((None, 2) for _ in range(self._num_edge_types)),
                node_to_graph_map=(None,),
                num_graphs=(),
            )
        )
        self.built = True

    def get_initial_node_feature_shape(self, input_shapes):
        return tuple(input_shapes["node_features"])

    def compute_initial_node_features(self, inputs, training: bool):
        return inputs["node_features"]

    @property
    def trainable_variables(self) -> List[Variable]:
        return list(self._gnn.trainable_variables) + self._task_variables()

    def _task_variables(self) -> List[Variable]:
        return []

    @property
    def variables(self) -> List[Variable]:
        """what save_model / load_weights_verbosely walk (utils/model_utils.py); the reference's non-trainable
        training_step counter is not part of the hot path"""
        return self.trainable_variables

    def zero_grad(self):
        for v in self.trainable_variables:
            v.grad = None

    # ---- forward (graph_task_model.py:158-183) -----------------------------------------------------------------
    def compute_final_node_representations(self, inputs, training: bool):
        # An input pipeline may hand over the batch already bucketed - ``bucketed_graph``: the ops.Graph of these adjacency
        # lists, built on a prefetch stream while the previous step trained (the reference prepares batches in a background
        # thread + tf.data prefetch: data/graph_dataset.py:292-295); the caller has ordered the compute stream after it.
        adjacency_lists = inputs.get("bucketed_graph") if hasattr(inputs, "get") else None
        if adjacency_lists is None:
            adjacency_lists = tuple(inputs[f"adjacency_list_{i}"] for i in range(self._num_edge_types))
        gnn_input = GNNInput(
            node_features=self.compute_initial_node_features(inputs, training),
            adjacency_lists=adjacency_lists,
            node_to_graph_map=inputs["node_to_graph_map"],
            num_graphs=inputs["num_graphs_in_batch"],
        )
        return self._gnn(gnn_input, training=training, return_all_representations=self._use_intermediate_gnn_results)

    def __call__(self, inputs, training: bool = False):
        if not self.built:
            self.build({"node_features": tuple(inputs["node_features"].shape)})
        final_node_representations = self.compute_final_node_representations(inputs, training)
        return self.compute_task_output(inputs, final_node_representations, training)

    call = __call__

    def compute_task_output(self, batch_features, final_node_representations, training: bool):
        raise NotImplementedError

    def compute_task_metrics(self, batch_features, task_output, batch_labels) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    # ---- backward (graph_task_model.py:347-357: tape.gradient(loss, trainable_variables)) -----------------------
    def backward(self) -> List[Tuple[Variable, Optional[torch.Tensor]]]:
        """d loss / d variable for the loss of the last ``compute_task_metrics`` call -> [(variable, grad)]."""
        if self._step is None or "dloss" not in self._step:
            raise RuntimeError("backward() needs a forward pass followed by compute_task_metrics()")
        grad_final, grad_all = self._task_backward()
        self._gnn.backward(grad_final, grad_all_representations=grad_all)
        return [(v, v.grad) for v in self.trainable_variables]

    def _task_backward(self):
        raise NotImplementedError


class NodeMulticlassTask(GraphTaskModel):
    """tf2_gnn/models/node_multiclass_task.py:26-76: Dense(num_labels) on the final node representations, sigmoid
    cross entropy summed over labels and averaged over nodes, micro-F1."""

    def __init__(self, params: Dict[str, Any], dataset: Any = None, name: Optional[str] = None, *,
                 num_edge_types: Optional[int] = None, num_node_target_labels: Optional[int] = None):
        super().__init__(params, dataset=dataset, name=name, num_edge_types=num_edge_types)
        if num_node_target_labels is None:
            if not hasattr(dataset, "num_node_target_labels"):
                raise ValueError(
                    f"Provided dataset of type {type(dataset)} does not provide num_node_target_labels information."
                )
            num_node_target_labels = dataset.num_node_target_labels
        self._num_labels = int(num_node_target_labels)

    def build(self, input_shapes):
        H = int(self._params["gnn_hidden_dim"])
        dev = default_device()
        # node_multiclass_task.py:41-43: a Dense built directly under tf.name_scope(<class name>) - Keras adds no layer name
        cls = self.__class__.__name__
        self._kernel = Variable(f"{cls}/kernel", glorot_uniform((H, self._num_labels), device=dev))
        self._bias = Variable(f"{cls}/bias", torch.zeros(self._num_labels, dtype=torch.float32, device=dev))
        super().build(input_shapes)

    def _task_variables(self):
        return [self._kernel, self._bias]

    def compute_task_output(self, batch_features, final_node_representations, training: bool):
        h = final_node_representations[0] if isinstance(final_node_representations, tuple) else final_node_representations
        per_node_logits = ops.gemm(h, self._kernel.value, bias=self._bias.value)
        self._step = {"h": h}
        return (per_node_logits,)

    def compute_task_metrics(self, batch_features, task_output, batch_labels) -> Dict[str, torch.Tensor]:
        (per_node_logits,) = task_output
        metrics, counts, dlogits = ops.sigmoid_ce_metrics(per_node_logits, batch_labels["node_labels"])
        if self._step is not None:
            self._step["dloss"] = dlogits
        return {"loss": metrics[0], "f1_score": metrics[1], "f1_counts": counts}

    def compute_epoch_metrics(self, task_results: List[Any]) -> Tuple[float, str]:
        """node_multiclass_task.py:72-74."""
        avg_microf1 = float(sum(float(r["f1_score"]) for r in task_results) / len(task_results))
        return -avg_microf1, f"Avg MicroF1: {avg_microf1:.3f}"

    def _task_backward(self):
        h, d = self._step["h"], self._step["dloss"]
        self._kernel.grad = ops.gemm(h, d, trans_a=True)
        self._bias.grad = ops.colsum(d)
        return ops.gemm(d, self._kernel.value, trans_b=True), None


def _regression_task_metrics(task, task_output, batch_features, batch_labels):
    """graph_regression_task.py:150-166 / qm9_regression.py:116-130."""
    metrics, dpred = ops.regression_metrics(task_output, batch_labels["target_value"])
    if task._step is not None:
        task._step["dloss"] = dpred
    num_graphs = float(int(batch_features["num_graphs_in_batch"]))
    return {
        "loss": metrics[0],
        "batch_squared_error": metrics[0] * num_graphs,
        "batch_absolute_error": metrics[1] * num_graphs,
        "num_graphs": num_graphs,
    }


def _regression_epoch_metrics(task_results):
    total_num_graphs = sum(r["num_graphs"] for r in task_results)
    epoch_mse = float(sum(float(r["batch_squared_error"]) for r in task_results) / total_num_graphs)
    epoch_mae = float(sum(float(r["batch_absolute_error"]) for r in task_results) / total_num_graphs)
    return epoch_mse, epoch_mae


class QM9RegressionTask(GraphTaskModel):
    """tf2_gnn/models/qm9_regression.py:30-145: per node sigmoid(gate([x0 | h])) * transform(h), summed per graph."""

    @classmethod
    def get_default_hyperparameters(cls, mp_style: Optional[str] = None) -> Dict[str, Any]:
        params = super().get_default_hyperparameters(mp_style)
        params.update({"use_intermediate_gnn_results": False, "out_layer_dropout_keep_prob": 1.0})
        return params

    def __init__(self, params: Dict[str, Any], dataset: Any = None, name: Optional[str] = None, *,
                 num_edge_types: Optional[int] = None, task_id: int = 0):
        super().__init__(params, dataset=dataset, name=name, num_edge_types=num_edge_types)
        self._task_id = int(getattr(dataset, "_params", {}).get("task_id", task_id)) if dataset is not None else int(task_id)
        # qm9_regression.py:49-62: the "keep prob" hyper-parameter is handed over as the dropout RATE; without hidden layers
        # the MLPs never draw a mask (see layers/nodes_to_graph_representation.py MLP)
        rate = float(self._params["out_layer_dropout_keep_prob"])
        self._regression_gate = MLP(out_size=1, hidden_layers=[], use_biases=True, dropout_rate=rate, name="gate")
        self._regression_transform = MLP(out_size=1, hidden_layers=[], use_biases=True, dropout_rate=rate, name="transform")

    def build(self, input_shapes):
        H = int(self._params["gnn_hidden_dim"])
        self._regression_gate.build(int(input_shapes["node_features"][-1]) + H)
        self._regression_transform.build(H)
        cls = self.__class__.__name__  # qm9_regression.py:65-80
        prefix_names(self._regression_gate.variables, f"{cls}/node_gate")
        prefix_names(self._regression_transform.variables, f"{cls}/node_transform")
        super().build(input_shapes)

    def _task_variables(self):
        return self._regression_gate.variables + self._regression_transform.variables

    def compute_task_output(self, batch_features, final_node_representations, training: bool):
        if self._params["use_intermediate_gnn_results"]:
            final_node_representations, _ = final_node_representations
        h = final_node_representations
        x0 = batch_features["node_features"]
        G = int(batch_features["num_graphs_in_batch"])
        per_node_output = self._regression_transform(h, training=training)  # [V, 1]
        per_node_weight = self._regression_gate(torch.cat([x0, h], dim=1), final_act="sigmoid", training=training)  # [V, 1]
        ids = batch_features["node_to_graph_map"].to(torch.int32).contiguous()
        ptr = segment_offsets(ids, G)
        out = torch.empty((G, 1), dtype=torch.float32, device=h.device)
        lib = _lib.load()
        _lib.check(
            lib.tfgnn_segment_weighted_sum(ops._ptr(per_node_output), ops._ptr(per_node_weight), ops._ptr(ptr), G, 1, 1, 0,
                                           ops._ptr(out), ops._stream())
        )
        self._step = {"ids": ids, "ptr": ptr, "R": per_node_output, "w": per_node_weight, "D0": x0.shape[1], "V": h.shape[0]}
        return out.view(G)

    def compute_task_metrics(self, batch_features, task_output, batch_labels):
        return _regression_task_metrics(self, task_output, batch_features, batch_labels)

    def compute_epoch_metrics(self, task_results: List[Any]) -> Tuple[float, str]:
        """qm9_regression.py:132-158."""
        epoch_mse, epoch_mae = _regression_epoch_metrics(task_results)
        return epoch_mae, f"Task {self._task_id} | MSE = {epoch_mse:.3f} | MAE = {epoch_mae:.3f}"

    def _task_backward(self):
        s = self._step
        V = s["V"]
        d_out = s["dloss"].contiguous().view(-1, 1)
        dR = torch.empty((V, 1), dtype=torch.float32, device=d_out.device)
        dw = torch.empty((V, 1), dtype=torch.float32, device=d_out.device)
        lib = _lib.load()
        _lib.check(
            lib.tfgnn_segment_weighted_sum_backward(ops._ptr(d_out), ops._ptr(s["R"]), ops._ptr(s["w"]), ops._ptr(s["ids"]),
                                                    ops._ptr(s["ptr"]), V, 1, 1, 0, ops._ptr(dR), ops._ptr(dw), ops._stream())
        )
        d_cat = self._regression_gate.backward(dw)  # [V, D0 + H]
        d_h = self._regression_transform.backward(dR)
        d_h = ops.add_scale(d_h, d_cat[:, s["D0"]:], 1.0)
        if self._params["use_intermediate_gnn_results"]:
            L = int(self._params["gnn_num_layers"])
            return d_h, [None] * (L + 1)
        return d_h, None


class GraphRegressionTask(GraphTaskModel):
    """tf2_gnn/models/graph_regression_task.py:15-185: softmax- and sigmoid-weighted sums of the node representations
    (the input features next to every layer's output, or the last one), concatenated, through a regression MLP."""

    @classmethod
    def get_default_hyperparameters(cls, mp_style: Optional[str] = None) -> Dict[str, Any]:
        params = super().get_default_hyperparameters(mp_style)
        params.update(
            {
                "use_intermediate_gnn_results": True,
                "graph_aggregation_output_size": 32,
                "graph_aggregation_num_heads": 4,
                "graph_aggregation_layers": [32, 32],
                "graph_aggregation_dropout_rate": 0.1,
                "regression_mlp_layers": [64, 32],
                "regression_mlp_dropout": 0.1,
            }
        )
        return params

    def __init__(self, params: Dict[str, Any], dataset: Any = None, name: Optional[str] = None, *,
                 num_edge_types: Optional[int] = None):
        super().__init__(params, dataset=dataset, name=name, num_edge_types=num_edge_types)

        def pooling(weighting_fun):
            return WeightedSumGraphRepresentation(
                graph_representation_size=params["graph_aggregation_output_size"],
                num_heads=params["graph_aggregation_num_heads"],
                weighting_fun=weighting_fun,
                scoring_mlp_layers=params["graph_aggregation_layers"],
                scoring_mlp_dropout_rate=params["graph_aggregation_dropout_rate"],
                scoring_mlp_activation_fun="elu",
                transformation_mlp_layers=params["graph_aggregation_layers"],
                transformation_mlp_dropout_rate=params["graph_aggregation_dropout_rate"],
                transformation_mlp_activation_fun="elu",
            )

        self._weighted_avg_of_nodes_to_graph_repr = pooling("softmax")
        self._weighted_sum_of_nodes_to_graph_repr = pooling("sigmoid")
        self._regression_mlp = MLP(out_size=1, hidden_layers=params["regression_mlp_layers"], use_biases=True,
                                   activation_fun="relu", dropout_rate=params["regression_mlp_dropout"])  # default name "MLP"
        self.regression_mlp_dropout_masks = None  # tests: masks per Dense layer instead of drawn ones

    def build(self, input_shapes):
        D0 = int(input_shapes["node_features"][-1])
        H = int(self._params["gnn_hidden_dim"])
        if self._params["use_intermediate_gnn_results"]:
            node_repr_size = D0 + H * int(self._params["gnn_num_layers"])
        else:
            node_repr_size = D0 + H
        shapes = NodesToGraphRepresentationInput((None, node_repr_size), (None,), ())
        self._weighted_avg_of_nodes_to_graph_repr.build(shapes)
        self._weighted_sum_of_nodes_to_graph_repr.build(shapes)
        self._regression_mlp.build(2 * int(self._params["graph_aggregation_output_size"]))
        # graph_regression_task.py:91-106: name scopes of the reference's variables
        cls = self.__class__.__name__
        prefix_names(self._weighted_avg_of_nodes_to_graph_repr.trainable_variables, f"{cls}/graph_representation_computation/weighted_avg")
        prefix_names(self._weighted_sum_of_nodes_to_graph_repr.trainable_variables, f"{cls}/graph_representation_computation/weighted_sum")
        prefix_names(self._regression_mlp.variables, cls)
        super().build(input_shapes)

    def _task_variables(self):
        return (
            list(self._weighted_avg_of_nodes_to_graph_repr.trainable_variables)
            + list(self._weighted_sum_of_nodes_to_graph_repr.trainable_variables)
            + self._regression_mlp.variables
        )

    def compute_task_output(self, batch_features, final_node_representations, training: bool):
        x0 = batch_features["node_features"]
        if self._params["use_intermediate_gnn_results"]:
            _, intermediate = final_node_representations
            pieces = (x0,) + tuple(intermediate[1:])  # skip the output of the initial projection (:112-121)
        else:
            pieces = (x0, final_node_representations)
        node_representations = torch.cat(pieces, dim=1)
        pool_in = NodesToGraphRepresentationInput(
            node_representations, batch_features["node_to_graph_map"], batch_features["num_graphs_in_batch"]
        )
        avg = self._weighted_avg_of_nodes_to_graph_repr(pool_in, training=training)
        tot = self._weighted_sum_of_nodes_to_graph_repr(pool_in, training=training)
        graph_representations = torch.cat([avg, tot], dim=1)  # [G, 2 GD]
        per_graph_results = self._regression_mlp(graph_representations, training=training,
                                                 dropout_masks=self.regression_mlp_dropout_masks)  # [G, 1]
        self._step = {"widths": [p.shape[1] for p in pieces], "GD": avg.shape[1]}
        return per_graph_results.view(-1)

    def compute_task_metrics(self, batch_features, task_output, batch_labels):
        return _regression_task_metrics(self, task_output, batch_features, batch_labels)

    def compute_epoch_metrics(self, task_results: List[Any]) -> Tuple[float, str]:
        """graph_regression_task.py:168-185."""
        epoch_mse, epoch_mae = _regression_epoch_metrics(task_results)
        return epoch_mae, f" MSE = {epoch_mse:.3f} | MAE = {epoch_mae:.3f}"

    def _task_backward(self):
        s = self._step
        d_graph = self._regression_mlp.backward(s["dloss"].contiguous().view(-1, 1))  # [G, 2 GD]
        GD = s["GD"]
        d_nodes = ops.add_scale(
            self._weighted_avg_of_nodes_to_graph_repr.backward(d_graph[:, :GD].contiguous()),
            self._weighted_sum_of_nodes_to_graph_repr.backward(d_graph[:, GD:].contiguous()),
            1.0,
        )  # [V, D0 + ...]
        widths = s["widths"]
        if self._params["use_intermediate_gnn_results"]:
            grads: List[Optional[torch.Tensor]] = [None]
            col = widths[0]
            for w in widths[1:]:
                grads.append(d_nodes[:, col : col + w])
                col += w
            return None, grads
        return d_nodes[:, widths[0] :], None
