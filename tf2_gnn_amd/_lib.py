"""ctypes binding of libtfgnn.so (the C ABI declared in include/tfgnn.h).

There is deliberately NO fallback: if the HIP library has not been built (``python -m
tf2_gnn_amd.build``) or cannot be loaded, every op raises ``RuntimeError``.  PyTorch is used for
device allocation and the current HIP stream only.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libtfgnn.so"

# (name, restype, argtypes) - one row per symbol declared in include/tfgnn.h
_SIGNATURES = [
    ("tfgnn_last_error", c_char_p, []),
    ("tfgnn_version", c_char_p, []),
    ("tfgnn_abi_version", c_int, []),
    ("tfgnn_gemm_gathered_supported", c_int, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    ("tfgnn_launch_counts", c_int, [POINTER(c_int64), c_int]),
    ("tfgnn_sp_spread_flag", c_int, [c_int]),
    (
        "tfgnn_graph_create",
        c_int,
        [c_int, c_int64, POINTER(c_void_p), POINTER(c_int64), c_void_p, POINTER(c_void_p)],
    ),
    ("tfgnn_graph_destroy", c_int, [c_void_p]),
    (
        "tfgnn_graph_create_async",
        c_int,
        [c_int, c_int64, POINTER(c_void_p), POINTER(c_int64), c_void_p, POINTER(c_void_p)],
    ),
    ("tfgnn_graph_wait", c_int, [c_void_p]),
    (
        "tfgnn_graph_create_parts_async",
        c_int,
        [c_int, c_int64, POINTER(c_void_p), POINTER(c_int64), ctypes.c_uint, c_void_p, POINTER(c_void_p)],
    ),
    ("tfgnn_graph_ensure", c_int, [c_void_p, ctypes.c_uint, c_void_p]),
    ("tfgnn_graph_parts", ctypes.c_uint, [c_void_p]),
    ("tfgnn_graph_destroy_async", c_int, [c_void_p, c_void_p]),
    ("tfgnn_graph_array", c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_int64)]),
    ("tfgnn_graph_dims", c_int, [c_void_p, POINTER(c_int64), POINTER(c_int), POINTER(c_int64)]),
    ("tfgnn_graph_scales", c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_graph_target_multiplier", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    (
        "tfgnn_csr_gather_reduce",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64,
         c_int, c_int, c_int, c_void_p],
    ),
    ("tfgnn_graph_gather_workspace_bytes", c_size_t, [c_void_p, c_int, c_int]),
    (
        "tfgnn_graph_gather_reduce",
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
         c_int, c_int, c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_graph_gather_dot_supported", c_int, [c_int, c_int]),
    (
        "tfgnn_graph_gather_reduce_dot",
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_edge_pair_combine", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    ("tfgnn_graph_original_order", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_gemm_workspace_bytes", c_size_t, [c_int64, c_int64, c_int64]),
    (
        "tfgnn_gemm_grad_epilogue",
        c_int,
        [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
         c_int, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p],
    ),
    (
        "tfgnn_gemm_gathered",
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
         c_int, c_int, c_void_p],
    ),
    ("tfgnn_gemm_set_mode", c_int, [c_int]),
    ("tfgnn_gemm_get_mode", c_int, []),
    (
        "tfgnn_gemm",
        c_int,
        [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
         c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p],
    ),
    (
        "tfgnn_gemm_grouped_rows",
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
         c_int64, c_int, c_void_p],
    ),
    (
        "tfgnn_gemm_grouped_rows_grad",
        c_int,
        [c_int, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
         c_int64, c_int, c_void_p, c_int64, c_void_p],
    ),
    ("tfgnn_gemm_grouped_k_workspace_bytes", c_size_t, [c_int, c_int64, c_int64, c_int64]),
    (
        "tfgnn_gemm_grouped_k",
        c_int,
        [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
         c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_graph_nonempty_offsets", c_int, [c_void_p, c_int, POINTER(c_int32)]),
    ("tfgnn_activation_forward", c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    ("tfgnn_activation_backward", c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    ("tfgnn_activation_backward_mul", c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    (
        "tfgnn_gemm_gru",
        c_int,
        [c_int64, c_int, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_gemm_gru2", c_int, [c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    ("tfgnn_gru_gates_forward", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    (
        "tfgnn_gru_gates_backward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p],
    ),
    ("tfgnn_gru_gates_backward_sp_workspace_bytes", c_size_t, [c_int64, c_int]),
    (
        "tfgnn_gru_gates_backward_sp",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
         c_int, c_void_p, c_size_t, c_void_p],
    ),
    (
        "tfgnn_gru_gates_backward_sp_dropout",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, ctypes.c_uint64, c_void_p,
         c_int64, c_int, c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_colsum_workspace_bytes", c_size_t, [c_int64, c_int]),
    ("tfgnn_colsum", c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("tfgnn_add_scale", c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    ("tfgnn_rgat_node_scores", c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_rgat_edge_scores", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    ("tfgnn_rgat_edge_node_op", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    ("tfgnn_rgat_edge_dot", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    (
        "tfgnn_rgat_edge_softmax_backward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p],
    ),
    ("tfgnn_rgat_scores_backward", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("tfgnn_rgat_scores_backward_sp", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_rgat_attention_workspace_bytes", c_size_t, [c_void_p, c_int]),
    ("tfgnn_rgat_attention_forward", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("tfgnn_rgat_attention_backward", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    ("tfgnn_rgat_alpha_grad_workspace_bytes", c_size_t, [c_int64, c_int, c_int]),
    ("tfgnn_rgat_alpha_grad", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    (
        "tfgnn_layernorm_forward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    (
        "tfgnn_layernorm_backward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_segment_offsets", c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    ("tfgnn_segment_offsets_async", c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_clip", c_int, [c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    ("tfgnn_clip_backward", c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    ("tfgnn_segment_softmax", c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    ("tfgnn_segment_weighted_sum", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    (
        "tfgnn_segment_weighted_sum_backward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_segment_softmax_backward", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    ("tfgnn_dropout_forward", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, ctypes.c_uint64, c_void_p]),
    ("tfgnn_film_edge_forward", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    ("tfgnn_film_edge_backward", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                         c_void_p]),
    ("tfgnn_sp_gemm_nt_set_splitk_workspace", c_int, [c_void_p, ctypes.c_size_t]),
    ("tfgnn_sp_gemm_nt_splitk_status", c_int, [c_int, c_void_p, c_void_p]),
    ("tfgnn_dropout_epoch_advance", c_int, [c_void_p]),
    ("tfgnn_dropout_epoch_set", c_int, [ctypes.c_uint32, c_void_p]),
    ("tfgnn_dropout_epoch_get", c_int, [c_void_p, c_void_p]),
    ("tfgnn_dropout_forward_sp", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, ctypes.c_uint64, c_void_p, c_int64,
                                         c_void_p, c_void_p]),
    ("tfgnn_transpose_batched", c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    (
        "tfgnn_edge_aggregate_backward",
        c_int,
        [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
         c_void_p, c_int, c_void_p, c_void_p],
    ),
    (
        "tfgnn_film_combine_forward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    ),
    (
        "tfgnn_film_combine_backward",
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_batch_offset_edges", c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    ("tfgnn_batch_node_to_graph_map", c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    ("tfgnn_adjacency_append", c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    ("tfgnn_adjacency_self_loops", c_int, [c_int64, c_void_p, c_void_p]),
    ("tfgnn_adjacency_in_degrees", c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    ("tfgnn_permute_021", c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    ("tfgnn_mul", c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    ("tfgnn_task_metrics_workspace_bytes", c_size_t, []),
    (
        "tfgnn_sigmoid_ce_metrics",
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_regression_metrics", c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    (
        "tfgnn_graph_gather_reduce_sp",
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_sp_gemm_tn_workspace_bytes", c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int]),
    (
        "tfgnn_sp_gemm_tn_grouped",
        c_int,
        [c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_sp_gemm_tn_wide_workspace_bytes", c_size_t, [c_int64, c_int64, c_int64, c_int64, c_int]),
    (
        "tfgnn_sp_gemm_tn_wide",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_tn",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_tn_phase",
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p],
    ),
    ("tfgnn_absmax", c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    ("tfgnn_sp_inv_scale_from_bound", c_int, [c_void_p, c_void_p, c_void_p]),
    ("tfgnn_sp_bytes", c_size_t, [c_int64, c_int64]),
    (
        "tfgnn_sp_split_rows",
        c_int,
        [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_sp_split_cols", c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    (
        "tfgnn_sp_gemm_nt",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p],
    ),
    ("tfgnn_sp_split_weights", c_int, [c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    (
        "tfgnn_sp_gemm_nt_sp",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    ("tfgnn_aux_launch", c_int, [c_void_p, c_int, c_void_p]),
    (
        "tfgnn_sp_split_rows_job",
        c_int,
        [c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    ("tfgnn_sp_split_cols_job", c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    ("tfgnn_sp_split_cols_two_pass_bytes", c_size_t, [c_int64, c_int64]),
    ("tfgnn_sp_split_cols_jobs", c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p,
                                         c_void_p]),
    (
        "tfgnn_graph_gather_reduce_sp_deferred",
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
         c_size_t, c_void_p, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_tn_jobs",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_tn_deferred",
        c_int,
        [c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
         c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_nt_dropout",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p, c_float,
         ctypes.c_uint64, c_void_p, c_void_p, c_void_p],
    ),
    (
        "tfgnn_sp_gemm_nt_grouped",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64,
         c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64,
         c_void_p, c_void_p],
    ),
    ("tfgnn_sp_gather_rows", c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                     c_void_p, c_void_p]),
    (
        "tfgnn_sp_gemm_nt_rows",
        c_int,
        [c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_int, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_float, c_void_p, c_int64, c_void_p, c_float,
         ctypes.c_uint64, c_void_p, c_void_p, c_void_p],
    ),
]

# layer-level entry points (include/tfgnn.h, round 6): the argument structs travel by pointer
_SIGNATURES += [
    ("tfgnn_mp_forward", c_int, [c_void_p, c_void_p]),
    ("tfgnn_mp_backward", c_int, [c_void_p, c_void_p]),
]

EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]
ABI_VERSION = 4  # include/tfgnn.h TFGNN_ABI_VERSION


class AuxJob(ctypes.Structure):
    """tfgnn_aux_job (include/tfgnn.h): one small pass of a merged launch"""

    _fields_ = [("kind", c_int), ("num_blocks", ctypes.c_uint), ("payload", ctypes.c_ubyte * 248)]


class MpForwardArgs(ctypes.Structure):
    """tfgnn_mp_forward_args (include/tfgnn.h), field for field"""

    _fields_ = [
        ("struct_size", ctypes.c_size_t), ("kind", c_int), ("graph", c_void_p), ("view", c_int), ("x", c_void_p), ("ldx", c_int64),
        ("in_dim", c_int), ("hidden_dim", c_int), ("row_scale", c_void_p), ("w", c_void_p), ("wt_sp", c_void_p),
        ("ld_wt_sp_bytes", c_int64), ("wt_inv_scale", c_void_p), ("agg_sp", c_void_p), ("agg_inv_scale", c_void_p), ("bias", c_void_p),
        ("act", c_int), ("dropout_rate", ctypes.c_float), ("dropout_seed", ctypes.c_uint64), ("tile_kmask", c_void_p),
        ("row_map", c_void_p), ("out", c_void_p), ("ld_out", c_int64), ("out_sp", c_void_p), ("ld_out_sp_bytes", c_int64),
        ("out_inv_scale", c_void_p), ("extra_jobs", c_void_p), ("num_extra_jobs", c_int), ("workspace", c_void_p),
        ("workspace_bytes", ctypes.c_size_t),
    ]


class MpBackwardArgs(ctypes.Structure):
    """tfgnn_mp_backward_args (include/tfgnn.h), field for field"""

    _fields_ = [
        ("struct_size", ctypes.c_size_t), ("kind", c_int), ("graph", c_void_p), ("d_pre", c_void_p), ("ld_d_pre", c_int64),
        ("in_dim", c_int), ("hidden_dim", c_int), ("edge_weight", c_void_p), ("w", c_void_p), ("wh_sp", c_void_p),
        ("ld_wh_sp_bytes", c_int64), ("wh_inv_scale", c_void_p), ("g_sp", c_void_p), ("g_inv_scale", c_void_p), ("dx", c_void_p),
        ("ld_dx", c_int64), ("accumulate", c_int), ("mul", c_void_p), ("ld_mul", c_int64), ("act_of_saved", c_int), ("saved", c_void_p),
        ("ld_saved", c_int64), ("saved_scale", ctypes.c_float), ("dropout_rate", ctypes.c_float), ("dropout_seed", ctypes.c_uint64),
        ("dx_sp", c_void_p), ("ld_dx_sp_bytes", c_int64), ("dx_inv_scale", c_void_p), ("tile_kmask", c_void_p), ("a_rows", c_void_p),
        ("row_map", c_void_p), ("dw", c_void_p), ("x_sp", c_void_p), ("ld_x_sp_bytes", c_int64), ("x_inv_scale", c_void_p),
        ("tn_workspace", c_void_p), ("tn_workspace_bytes", ctypes.c_size_t), ("extra_jobs", c_void_p), ("num_extra_jobs", c_int),
        ("workspace", c_void_p), ("workspace_bytes", ctypes.c_size_t),
    ]


_lib = None


class TfgnnError(RuntimeError):
    pass


def load():
    """Load libtfgnn.so (once).  Raises RuntimeError when it is missing - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run `python -m tf2_gnn_amd.build` "
            "(hipcc --offload-arch=gfx950). tf2_gnn_amd has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.tfgnn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} speaks ABI version {lib.tfgnn_abi_version()}, this binding was written against {ABI_VERSION} "
                           "(include/tfgnn.h TFGNN_ABI_VERSION): rebuild with `python -m tf2_gnn_amd.build`")
    _lib = lib
    return lib


def check(status: int):
    if status == 0:
        return
    msg = load().tfgnn_last_error().decode("utf-8", "replace")
    if status in (-1, -2):
        # same convention as the reference's Python layer: bad arguments / indices -> ValueError
        raise ValueError(f"tfgnn: {msg}")
    raise TfgnnError(f"tfgnn (status {status}): {msg}")
