"""Project wide constants (tf2_gnn/utils/constants.py:2)."""
SMALL_NUMBER = 1e-7
