from .synthetic import make_qm9_shaped_batch, make_synthetic_batch, make_zipf_typed_batch, rmat_edges
from .utils import compute_number_of_edge_types, get_tied_edge_types, process_adjacency_lists
from .batching import GraphSample, batch_adjacency_lists, check_batch, graph_batch_iterator_from_graph_iterator
