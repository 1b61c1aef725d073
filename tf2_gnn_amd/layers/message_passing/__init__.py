"""Message passing layers on the HIP path.  The registry keys are the lower-cased class names, as in the
reference (tf2_gnn/layers/message_passing/message_passing.py:221-227), so ``params["message_calculation_class"]``
resolves to the same layer types."""
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

# importing a module registers its class
from .gnn_edge_mlp import GNN_Edge_MLP  # isort: skip
from .gnn_film import GNN_FiLM  # isort: skip
from .ggnn import GGNN  # isort: skip
from .rgat import RGAT  # isort: skip
from .rgcn import RGCN  # isort: skip
from .rgin import RGIN  # isort: skip


def get_message_passing_class(message_calculation_class_name: str):
    """Registry lookup by (case-insensitive) class name; unknown names raise ValueError with the reference's
    message (tf2_gnn/layers/message_passing/__init__.py:9-13)."""
    try:
        return MESSAGE_PASSING_IMPLEMENTATIONS[message_calculation_class_name.lower()]
    except KeyError:
        raise ValueError(f"Unknown message passing type: {message_calculation_class_name}") from None


def get_known_message_passing_classes():
    """Names of the registered layer classes."""
    return (cls.__name__ for cls in MESSAGE_PASSING_IMPLEMENTATIONS.values())
