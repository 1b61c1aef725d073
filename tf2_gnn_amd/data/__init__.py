from .synthetic import make_synthetic_batch, rmat_edges
