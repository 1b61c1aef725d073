from .message_passing import (
    MESSAGE_PASSING_IMPLEMENTATIONS,
    MessagePassing,
    MessagePassingInput,
    Variable,
    calculate_type_to_num_incoming_edges,
    clear_graph_cache,
    get_graph,
    register_message_passing_implementation,
    set_default_device,
    set_seed,
)
from .rgat import RGAT
from .rgcn import RGCN
from .rgin import RGIN
from .ggnn import GGNN
from .gnn_edge_mlp import GNN_Edge_MLP
from .gnn_film import GNN_FiLM


def get_message_passing_class(message_calculation_class_name: str):
    """tf2_gnn/layers/message_passing/__init__.py:9-13"""
    calculation_class = MESSAGE_PASSING_IMPLEMENTATIONS.get(message_calculation_class_name.lower())
    if calculation_class is None:
        raise ValueError(f"Unknown message passing type: {message_calculation_class_name}")
    return calculation_class


def get_known_message_passing_classes():
    for message_passing_implementation in MESSAGE_PASSING_IMPLEMENTATIONS.values():
        yield message_passing_implementation.__name__
