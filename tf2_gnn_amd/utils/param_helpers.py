"""Name -> function registries, mirroring tf2_gnn/utils/param_helpers.py:7-39.

The returned callables have the reference's signatures
    aggregation_fn(data=[M, H], segment_ids=[M], num_segments=V) -> [V, H]
    activation_fn(tensor) -> tensor
and run on the HIP kernels (device tensors only).  ``.tfgnn_name`` carries the canonical name so
that layers can fuse the op into a producing kernel instead of calling it.
"""
from __future__ import annotations

import torch

from .. import ops

_AGGREGATIONS = ("sum", "max", "mean", "sqrt_n")
_ACTIVATIONS = ("tanh", "relu", "leaky_relu", "elu", "selu", "gelu")


def _make_aggregation(name: str):
    def aggregation_fn(data: torch.Tensor, segment_ids: torch.Tensor, num_segments: int) -> torch.Tensor:
        """tf.math.unsorted_segment_<name>(data, segment_ids, num_segments) on the HIP path."""
        from ..segment import unsorted_segment_reduce

        return unsorted_segment_reduce(name, data, segment_ids, int(num_segments))

    aggregation_fn.tfgnn_name = name
    aggregation_fn.__name__ = f"unsorted_segment_{name}"
    return aggregation_fn


def get_aggregation_function(aggregation_fn_name: str):
    """Convert from an aggregation function name to the function itself (param_helpers.py:7-18)."""
    if aggregation_fn_name not in _AGGREGATIONS:
        raise ValueError(f"Unknown aggregation function: {aggregation_fn_name}")
    return _make_aggregation(aggregation_fn_name)


def _make_activation(name: str):
    def activation_fn(x: torch.Tensor) -> torch.Tensor:
        return ops.activation_forward(name, x)

    activation_fn.tfgnn_name = name
    activation_fn.__name__ = name
    return activation_fn


def get_activation_function(activation_fn_name):
    """Convert from an activation function name to the function itself (param_helpers.py:21-39).
    Like the reference: ``None`` -> ``None``; names are lower-cased; "linear" maps to ``None`` in
    the table and therefore raises ValueError (param_helpers.py:28,36-38)."""
    if activation_fn_name is None:
        return None
    name = activation_fn_name.lower()
    if name not in _ACTIVATIONS:
        raise ValueError(f"Unknown activation function: {activation_fn_name}")
    return _make_activation(name)
