from .gnn import GNN, GNNInput
from .graph_global_exchange import (
    GraphGlobalExchange,
    GraphGlobalExchangeInput,
    GraphGlobalGRUExchange,
    GraphGlobalMeanExchange,
    GraphGlobalMLPExchange,
)
from .message_passing import (
    GGNN,
    GNN_Edge_MLP,
    GNN_FiLM,
    MessagePassing,
    MessagePassingInput,
    RGAT,
    RGCN,
    RGIN,
    get_known_message_passing_classes,
    get_message_passing_class,
)
from .nodes_to_graph_representation import (
    NodesToGraphRepresentation,
    NodesToGraphRepresentationInput,
    WASGraphRepresentation,
    WeightedSumGraphRepresentation,
)
