/*
 * tfgnn.h - C ABI of the MI355X-native (gfx950) message-passing hot path of microsoft/tf2-gnn.
 *
 * The reference (/root/reference, pure Python on TensorFlow 2) has no FFI of its own: every
 * "kernel" on its hot path is a stock TensorFlow op dispatched from
 * tf2_gnn/layers/message_passing/{message_passing,gnn_edge_mlp,rgcn,rgin,ggnn,rgat}.py and tf2_gnn/layers/nodes_to_graph_representation.py.
 * This header therefore declares one entry point per TensorFlow op (or fused group of ops) that
 * path dispatches; each declaration cites the reference call site it replaces (file:line relative
 * to /root/reference).  A maintainer binds them with ctypes.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HIP), row-major, contiguous unless a leading
 *     dimension (ld*, counted in elements) is given; floats are IEEE fp32, indices int32.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it,
 *     no entry point synchronises except tfgnn_graph_create (one small D2H validity read).
 *   - return value: 0 on success, negative tfgnn_status otherwise; tfgnn_last_error() returns a
 *     thread-local description of the last failure.
 *   - inputs are never written; outputs must not alias inputs unless stated.
 */
#ifndef TFGNN_H
#define TFGNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  TFGNN_OK = 0,
  TFGNN_ERR_INVALID_ARGUMENT = -1, /* bad shape / null pointer / unknown enum           */
  TFGNN_ERR_OUT_OF_RANGE = -2,     /* node index outside [0, V) (TF Gather: InvalidArgument) */
  TFGNN_ERR_HIP = -3,              /* a HIP runtime call failed                          */
  TFGNN_ERR_UNSUPPORTED = -4
} tfgnn_status;

/* tf2_gnn/utils/param_helpers.py:25-33 (+ "none" = identity, "sigmoid" for pooling weights,
 * nodes_to_graph_representation.py:176) */
typedef enum {
  TFGNN_ACT_NONE = 0,
  TFGNN_ACT_RELU = 1,
  TFGNN_ACT_TANH = 2,
  TFGNN_ACT_LEAKY_RELU = 3, /* alpha = 0.2 (tf.nn.leaky_relu default) */
  TFGNN_ACT_ELU = 4,
  TFGNN_ACT_SELU = 5,
  TFGNN_ACT_GELU = 6, /* tanh approximation, tf2_gnn/utils/activation.py:7-14 */
  TFGNN_ACT_SIGMOID = 7
} tfgnn_activation;

/* tf2_gnn/utils/param_helpers.py:9-14 */
typedef enum {
  TFGNN_REDUCE_SUM = 0,
  TFGNN_REDUCE_MAX = 1 /* empty segment -> lowest finite float, like tf.math.unsorted_segment_max */
} tfgnn_reduce;

const char* tfgnn_last_error(void);
/* "tfgnn <version> gfx950" */
const char* tfgnn_version(void);
/* Bumped whenever an existing entry point changes its signature or meaning (round 4: 2 - tfgnn_gemm_grad_epilogue gained
 * `accumulate` in round 3, tfgnn_gemm_get_mode can return TFGNN_GEMM_F16X2, the bucketing keeps list order inside a bucket).
 * A binding compares tfgnn_abi_version() with the TFGNN_ABI_VERSION it was written against (tf2_gnn_amd/_lib.py does). */
#define TFGNN_ABI_VERSION 4
int tfgnn_abi_version(void);

/* Diagnostics (no reference counterpart): number of launches of each product-kernel family this process has enqueued
 * since the library was loaded - host-side counters, no device work.  The parity tests read them around a layer call
 * to assert that a GEMM mode (tfgnn_gemm_set_mode / the f16x2 layer paths) really ran the kernels it names, so that
 * parity evidence and timing evidence are about the same code. */
typedef enum {
  TFGNN_KFAM_GEMM_FP32 = 0,   /* gemm_mfma_kernel (v_mfma_f32_32x32x2_f32)                       */
  TFGNN_KFAM_GEMM_BF16X3 = 1, /* gemm_x3s / gemm_x3p kernels (3-way bf16 split inside the GEMM)    */
  TFGNN_KFAM_SP_NT = 2,       /* gemm_sp_nt_kernel (pre-split fp16 pairs, forward / input gradient) */
  TFGNN_KFAM_SP_TN = 3,       /* gemm_sp_tn_kernel (pre-split fp16 pairs, weight gradient)          */
  TFGNN_KFAM_GATHER_SP = 4,   /* csr_gather_reduce_kernel writing the SP16 operand                  */
  TFGNN_KFAM_GATHER = 5,      /* csr_gather_reduce_kernel, fp32 output                              */
  TFGNN_KFAM_FUSED_NT = 6,    /* reserved (the gather-producing product was measured and not shipped) */
  TFGNN_KFAM_GEMM_STREAM = 7, /* gemm_x3k_kernel: bf16x3 products with the weight block resident in LDS (short K,
                                 very many rows); every such launch also counts as TFGNN_KFAM_GEMM_BF16X3 */
  TFGNN_KFAM_STREAM_F16X2 = 8, /* gemm_x3k_kernel in the f16x2 arithmetic (round 6: mode f16x2; 3 products, operands split on the fly);
                                  such a launch counts here and as TFGNN_KFAM_GEMM_STREAM, NOT as TFGNN_KFAM_GEMM_BF16X3 */
  TFGNN_KFAM_COUNT = 9
} tfgnn_kernel_family;
int tfgnn_launch_counts(int64_t* out_counts, int n);

/* ------------------------------------------------------------------------------------------
 * Graph handle: the (dst, edge_type)- and (src, edge_type)-bucketed adjacency of one batch.
 * Replaces, for all L layers and both passes of a step, the per-layer work of
 *   message_passing.py:195-206 (slicing src/dst columns, embedding_lookup of the degree),
 *   message_passing.py:230-263 calculate_type_to_num_incoming_edges (scatter_nd of ones),
 *   message_passing.py:166-167 (concat of targets / messages over edge types).
 * Row r = node * L + edge_type.  Within a row, columns ascend (canonical, deterministic).
 * ------------------------------------------------------------------------------------------ */
typedef struct tfgnn_graph tfgnn_graph;

/* d_adjacency[l] : device int32 [num_edges[l], 2], row k = (source, target)  (gnn.py:241-244).
 * The array of pointers and num_edges live on the HOST. */
int tfgnn_graph_create(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                       const int64_t* num_edges, void* stream, tfgnn_graph** out_graph);
int tfgnn_graph_destroy(tfgnn_graph* graph);
/* Pipelined form (the reference prepares batches in a background thread + tf.data prefetch,
 * data/graph_dataset.py:292-295, cli_utils/training_utils.py:114-115): enqueue the whole build on
 * `stream` and return at once; tfgnn_graph_wait blocks the HOST until the build has finished, then
 * reports bad indices.  Consumers on other streams must order themselves after `stream` (event /
 * stream wait) and call tfgnn_graph_wait before the first gather.  tfgnn_graph_destroy_async returns
 * the handle's memory to the library's pool, to be reused only after the work already enqueued on
 * `last_use_stream` (tfgnn_graph_destroy waits for the device instead). */
int tfgnn_graph_create_async(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                             const int64_t* num_edges, void* stream, tfgnn_graph** out_graph);
int tfgnn_graph_wait(tfgnn_graph* graph);
int tfgnn_graph_destroy_async(tfgnn_graph* graph, void* last_use_stream);

/* What a batch's handle holds beyond the two sorted edge orders (row pointers, columns, edge ids, degrees: always built).
 * A layer stack that only runs the aggregate-first RGCN / GGNN path needs the typed plans alone; everything else costs
 * preparation kernels per batch for arrays nobody reads.  tfgnn_graph_create / _async build everything (as before);
 * tfgnn_graph_create_parts_async builds the requested parts; tfgnn_graph_ensure adds missing ones later on `stream`
 * (it blocks the host: the sizes of a part come back from the device).  Entry points that need a part the handle does
 * not have fail with TFGNN_ERR_INVALID_ARGUMENT and name the part - they never read unbuilt arrays.
 * LIFETIME / ORDERING of tfgnn_graph_ensure(EDGE_IDS | EDGE_MAPS) on a handle created without EDGE_IDS: it sorts again - it
 * reads the adjacency lists of the creation call a second time (the caller must still own them) and REWRITES the handle's
 * row pointers / columns / degrees in place with identical values, on `stream`.  Kernels reading the handle on ANOTHER
 * stream at that moment race with those writes: call it only from the stream that uses the handle, or - better - name the
 * parts at creation (a stack with RGAT or per-edge messages: EDGE_MAPS / EDGE_IDS; tf2_gnn_amd: GNN.graph_parts does). */
typedef enum {
  TFGNN_GRAPH_PART_PLAN_TYPED = 1, /* long-row plans + length-ordered short rows of the views BY_DST_TYPED / BY_SRC_TYPED */
  TFGNN_GRAPH_PART_PLAN_NODE = 2,  /* ... of the views BY_DST_NODE / BY_SRC_NODE (RGAT, per-edge messages)                */
  TFGNN_GRAPH_PART_COMPACT = 4,    /* non-empty (node, type) buckets, type-major: TFGNN_G_NZ_* and the COMPACT views        */
  TFGNN_GRAPH_PART_EDGE_MAPS = 8,  /* TFGNN_G_SRC2DST_POS and its inverse (RGAT: attention weights in both edge orders);
                                      implies EDGE_IDS                                                                      */
  TFGNN_GRAPH_PART_EDGE_IDS = 16,  /* TFGNN_G_EID_BY_DST / _SRC: the position of every bucketed edge in the caller's lists
                                      (per-edge message forms).  Without it the sort moves 4 bytes per edge and digit instead
                                      of 8.  Added later by tfgnn_graph_ensure it costs a second sort, which READS THE
                                      ADJACENCY LISTS OF THE CREATION CALL AGAIN: keep them alive.                          */
  TFGNN_GRAPH_PART_DST_PATTERN = 32, /* TFGNN_G_PATTERN_*: the nodes ordered by which of their by-target buckets are empty, and per
                                      128 positions the union of those patterns - lets the forward product of the aggregate-first
                                      layers skip the all-zero K blocks of a row tile (TFGNN_VIEW_BY_DST_TYPED_PATTERN, the tile
                                      mask / row map arguments of tfgnn_sp_gemm_nt_dropout); at most 8 edge types, empty beyond  */
  TFGNN_GRAPH_PARTS_DEFAULT = 31, /* what tfgnn_graph_create_async builds: everything but the pattern order (on request) */
  TFGNN_GRAPH_PARTS_ALL = 63
} tfgnn_graph_part;
int tfgnn_graph_create_parts_async(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                                   const int64_t* num_edges, unsigned parts, void* stream, tfgnn_graph** out_graph);
int tfgnn_graph_ensure(tfgnn_graph* graph, unsigned parts, void* stream);
unsigned tfgnn_graph_parts(const tfgnn_graph* graph);

typedef enum {
  TFGNN_G_ROWPTR_BY_DST = 0, /* int32 [V*L+1]                                                  */
  TFGNN_G_COL_BY_DST = 1,    /* int32 [E]  source node of each bucketed edge                  */
  TFGNN_G_EID_BY_DST = 2,    /* int32 [E]  position of the edge in concat(adjacency_lists)    */
  TFGNN_G_COLL_BY_DST = 3,   /* int32 [E]  source * L + edge_type (row of a [V*L, H] tensor)  */
  TFGNN_G_ROWPTR_BY_SRC = 4, /* int32 [V*L+1]                                                  */
  TFGNN_G_COL_BY_SRC = 5,    /* int32 [E]  target node                                        */
  TFGNN_G_EID_BY_SRC = 6,    /* int32 [E]                                                     */
  TFGNN_G_COLL_BY_SRC = 7,   /* int32 [E]  target * L + edge_type                             */
  TFGNN_G_INVDEG_BY_DST = 8, /* float [V*L] 1/(in_degree[l,v] + 1e-7) (gnn_edge_mlp.py:102-106), 0 for empty rows */
  TFGNN_G_INVDEG_EDGE_BY_SRC = 9, /* float [E] same quantity per edge, in by-src order        */
  TFGNN_G_NODEPTR_BY_DST = 10,    /* int32 [V+1] = ROWPTR_BY_DST[::L] (all edge types of a target) */
  TFGNN_G_NODEPTR_BY_SRC = 11,    /* int32 [V+1]                                               */
  TFGNN_G_INVDEG_EDGE_BY_DST = 12, /* float [E] per edge, by-dst order                          */
  TFGNN_G_SRC2DST_POS = 13,       /* int32 [E] position in by-dst order of the edge at each by-src position */
  TFGNN_G_TARGET_BY_DST = 14,     /* int32 [E] target node of each bucketed edge, by-dst order */
  /* non-empty buckets in type-major order (compact index c): the per-relation multiply only has to
   * visit buckets that received at least one edge (45 % of the (node,type) buckets of an R-MAT batch
   * are empty) */
  TFGNN_G_NZ_CPOS_BY_DST = 15,    /* int32 [V*L] bucket row -> compact index or -1            */
  TFGNN_G_NZ_ROW_BY_DST = 16,     /* int32 [nz]  compact index -> bucket row                  */
  TFGNN_G_NZ_NODE_BY_DST = 17,    /* int32 [nz]  compact index -> node                        */
  TFGNN_G_NZ_OFF_BY_DST = 18,     /* int32 [L+1] first compact index of each edge type        */
  TFGNN_G_NZ_NODEPTR_BY_DST = 19, /* int32 [V+1] CSR over nodes of their non-empty buckets    */
  TFGNN_G_NZ_COL_BY_DST = 20,     /* int32 [nz]  compact indices grouped by node              */
  TFGNN_G_NZ_CPOS_BY_SRC = 21,
  TFGNN_G_NZ_ROW_BY_SRC = 22,
  TFGNN_G_NZ_NODE_BY_SRC = 23,
  TFGNN_G_NZ_OFF_BY_SRC = 24,
  TFGNN_G_NZ_NODEPTR_BY_SRC = 25,
  TFGNN_G_NZ_COL_BY_SRC = 26,
  /* part DST_PATTERN (at most 8 edge types; count 0 beyond): nodes ordered by the emptiness pattern of their by-target buckets */
  TFGNN_G_PATTERN_POS_BY_DST = 27,      /* int32 [V]: node -> position                                  */
  TFGNN_G_PATTERN_NODE_BY_DST = 28,     /* int32 [V]: position -> node (the row map of a product's output) */
  TFGNN_G_PATTERN_TILEMASK_BY_DST = 29, /* uint8 [ceil(V/128)]: union of the patterns of 128 positions (bit l: type l non-empty) */
  /* the same order for the by-SOURCE buckets (also part DST_PATTERN; round 5): the input-gradient product over the by-source
   * sums reads its operand rows through NODE_BY_SRC (tfgnn_sp_gemm_nt_rows d_a_rows) and skips a tile's all-zero blocks */
  TFGNN_G_PATTERN_NODE_BY_SRC = 30,     /* int32 [V]: position -> node */
  TFGNN_G_PATTERN_TILEMASK_BY_SRC = 31  /* uint8 [ceil(V/128)] */
} tfgnn_graph_array_id;

/* Borrow a device array owned by the handle (valid until tfgnn_graph_destroy). */
int tfgnn_graph_array(const tfgnn_graph* graph, int array_id, const void** d_ptr, int64_t* count);
/* host copy of TFGNN_G_NZ_OFF_* ([L+1] int32; offsets[L] = number of non-empty buckets) */
int tfgnn_graph_nonempty_offsets(const tfgnn_graph* graph, int by_src, int32_t* h_offsets);
int tfgnn_graph_dims(const tfgnn_graph* graph, int64_t* num_nodes, int* num_edge_types,
                     int64_t* num_edges);

/* Scales that fold the degree normalisation (gnn_edge_mlp.py:102-106) and the mean / sqrt_n
 * aggregators (utils/param_helpers.py:12-13) into the gather kernel.
 *   aggregation_mode 0 = sum, 1 = mean (1/max(N_v,1)), 2 = sqrt_n (1/sqrt(max(N_v,1))), N_v = number
 *   of messages entering v over all edge types.
 *   d_row_scale [V*L]          per bucket (v,l): (normalize ? 1/(c_lv+1e-7) : 1) * m_v
 *   d_node_scale [V]           m_v
 *   d_edge_weight_by_src [E]   per edge in by-src order: (normalize ? 1/(c_{l,tgt}+1e-7) : 1) * m_tgt
 *   d_edge_weight_by_dst [E]   per edge in by-dst order: (normalize ? 1/(c_{l,tgt}+1e-7) : 1)          */
int tfgnn_graph_scales(const tfgnn_graph* graph, int normalize_by_num_incoming, int aggregation_mode,
                       float* d_row_scale, float* d_node_scale, float* d_edge_weight_by_src,
                       float* d_edge_weight_by_dst, void* stream);

/* Helper for use_target_state_as_input with a linear edge layer (gnn_edge_mlp.py:92-97):
 * sum_e s_e [x_u | x_v] W = (sum_e s_e x_u) W_s + k_{l,v} x_v W_t with k = c_{l,v} * row_scale.
 *   d_k [V*L], d_ident_ptr [V*L+1] = 0..V*L, d_node_of_row [V*L] = r / L                           */
int tfgnn_graph_target_multiplier(const tfgnn_graph* graph, const float* d_row_scale, float* d_k,
                                  int32_t* d_ident_ptr, int32_t* d_node_of_row, void* stream);

/* ------------------------------------------------------------------------------------------
 * Gather + segment reduce over a CSR (the hot kernel):
 *   out[r, :] = post_act( row_scale[r] * REDUCE_{e in [rowptr[r], rowptr[r+1])}
 *                                 pre_act( edge_weight[e] * in[col[e], :] ) )
 * Replaces tf.nn.embedding_lookup (message_passing.py:197-206) fused with
 * tf.math.unsorted_segment_{sum,max,mean,sqrt_n} (utils/param_helpers.py:9-14 via
 * message_passing.py:172-174) and the per-message 1/(c+1e-7) scaling (gnn_edge_mlp.py:102-106);
 * mean / sqrt_n are SUM with a row_scale.  d_edge_weight and d_row_scale may be NULL (= 1).
 * width = number of floats per row; ld_in / ld_out = row strides in floats.
 * ------------------------------------------------------------------------------------------ */
int tfgnn_csr_gather_reduce(const int32_t* d_rowptr, const int32_t* d_col,
                            const float* d_edge_weight, const float* d_row_scale,
                            int64_t num_rows, const float* d_in, int64_t ld_in, int width,
                            float* d_out, int64_t ld_out, int reduce_op, int pre_act, int post_act,
                            void* stream);

/* The same reduction over one of the graph handle's four bucketed views, with the handle's plan for
 * long rows (rows with more than 32 edges are cut into 256-edge items, one workgroup each, combined
 * deterministically) - the form the layers use:
 *   TFGNN_VIEW_BY_DST_TYPED  rows (v,l) -> col = source             (aggregate-first forward)
 *   TFGNN_VIEW_BY_DST_NODE   rows v     -> col = source*L + type    (transform-first forward, RGAT)
 *   TFGNN_VIEW_BY_SRC_TYPED  rows (u,l) -> col = target             (backward: scatter by source)
 *   TFGNN_VIEW_BY_SRC_NODE   rows u     -> col = target*L + type
 * d_col_override (nullable) replaces the view's column array (same edge order).  d_edge_weight is
 * [E] (ew_heads == 1) or [E, ew_heads] applied per head of width/ew_heads floats (RGAT attention,
 * rgat.py:154-160).  d_workspace must hold tfgnn_graph_gather_workspace_bytes(graph, view, width). */
typedef enum {
  TFGNN_VIEW_BY_DST_TYPED = 0,
  TFGNN_VIEW_BY_DST_NODE = 1,
  TFGNN_VIEW_BY_SRC_TYPED = 2,
  TFGNN_VIEW_BY_SRC_NODE = 3,
  /* the two typed views with COMPACT output: row c of the output belongs to the c-th non-empty
   * bucket in type-major order (TFGNN_G_NZ_* arrays); empty buckets produce no row (and, since round 5, no work: the
   * launch covers the non-empty buckets only).  tfgnn_graph_gather_reduce_sp takes them too: one operand row and one
   * scale per non-empty bucket */
  TFGNN_VIEW_BY_DST_TYPED_COMPACT = 4,
  TFGNN_VIEW_BY_SRC_TYPED_COMPACT = 5,
  /* the by-target typed view with its output rows in PATTERN order: bucket (v, l) is written at row pos[v] * L + l
   * (TFGNN_G_PATTERN_POS_BY_DST), every bucket has a row (part DST_PATTERN) */
  TFGNN_VIEW_BY_DST_TYPED_PATTERN = 6
  /* (7, round 5 only: the same rows without the zero rows of blocks a masked consumer skips - measured no gain, the gather's
   *  time is its reads; removed in round 6 with ABI version 4) */
} tfgnn_graph_view;
size_t tfgnn_graph_gather_workspace_bytes(const tfgnn_graph* graph, int view, int width);
int tfgnn_graph_gather_reduce(const tfgnn_graph* graph, int view, const int32_t* d_col_override,
                              const float* d_edge_weight, int ew_heads, const float* d_row_scale,
                              const float* d_in, int64_t ld_in, int width, float* d_out, int64_t ld_out,
                              int reduce_op, int pre_act, int post_act, void* d_workspace,
                              size_t workspace_bytes, void* stream);

/* Per-edge pieces of GNN_Edge_MLP with target states AND hidden layers (gnn_edge_mlp.py:92-100): the
 * first Dense is separable, [x_u|x_v] W = x_u W_s + x_v W_t, computed per node; per edge only
 *   out[e,:] = act( P[index_a[e],:] + Q[index_b[e],:] )
 * remains.  tfgnn_graph_original_order gives, in the order of the concatenated adjacency lists
 * (type-contiguous, so later Dense layers are GEMMs over [E_l, H] blocks): source*L+type, target*L+type,
 * target node, and (optional) a per-edge weight array re-ordered from by-dst order. */
int tfgnn_edge_pair_combine(const int32_t* d_index_a, const int32_t* d_index_b, const float* d_P,
                            const float* d_Q, int64_t num_edges, int width, int act, float* d_out,
                            void* stream);

/* GNN_FiLM in its per-edge form (gnn_film.py:83-108), for the configurations whose modulated messages go through a max
 * aggregation or a per-message activation (sums are modulated on the node side: tfgnn_film_combine_*):
 *   out[e,:] = gamma[film_row[e],:] * (weight[e] * msg[msg_row[e],:]) + beta[film_row[e],:]
 * d_film rows are [gamma | beta] of width 2 * width; msg_row NULL = identity, edge_weight NULL = 1.  The backward entry
 * writes per-edge terms: grad_msg [E, width] and grad_film [E, 2 * width]; the caller reduces them over message rows /
 * (target, type) buckets with the gather kernel. */
int tfgnn_film_edge_forward(const float* d_msg, const int32_t* d_msg_row, const float* d_film, const int32_t* d_film_row,
                            const float* d_edge_weight, int64_t num_edges, int width, float* d_out, void* stream);
int tfgnn_film_edge_backward(const float* d_grad, const float* d_msg, const int32_t* d_msg_row, const float* d_film,
                             const int32_t* d_film_row, const float* d_edge_weight, int64_t num_edges, int width,
                             float* d_grad_msg, float* d_grad_film, void* stream);

/* Backward through a general aggregation, i.e. the cases the sum-only algebra does not cover:
 * aggregation_function "max" (utils/param_helpers.py:11) and message_activation_before_aggregation
 * (message_passing.py:169-172).  Forward: agg[t,:] = node_scale[t] * REDUCE_{e->t} pre_act(w_e * msg[row_e,:])
 * (tfgnn_graph_gather_reduce).  Per edge, in the order of d_target / d_msg_row / d_edge_weight:
 *   phase 0 (max):  d_out[e,:] = 1 where the edge attains the maximum of its target, else 0
 *                   (segment-sum it to get d_num_selected: TF splits the gradient evenly among ties [ext])
 *   phase 1:        d_out[e,:] = d(loss)/d(msg[row_e,:]) through edge e
 * d_msg_row NULL = identity (messages already per edge); d_agg_max / d_num_selected only for max. */
int tfgnn_edge_aggregate_backward(int64_t num_edges, int64_t width, const float* d_msg, int64_t ld_msg,
                                  const int32_t* d_msg_row, const int32_t* d_target, const float* d_edge_weight,
                                  const float* d_node_scale, int pre_act, int reduce_op, const float* d_grad_agg,
                                  const float* d_agg_max, const float* d_num_selected, int phase, float* d_out,
                                  void* stream);

/* GNN_FiLM (gnn_film.py:84-108): every message is modulated by its target, m' = gamma[l,tgt] * m + beta[l,tgt],
 * with (gamma | beta) = FiLM-MLP_l(h_tgt) [2H].  For sum / mean / sqrt_n aggregation this is a node-side epilogue
 * over the per-bucket message sums Z [V, L, H] (bucket row r = node * L + type):
 *   pre[v,:] = node_scale[v] * sum_l ( film[v,l,:H] * Z[v,l,:] + cnt[v,l] * film[v,l,H:] ),  out = act(pre)
 * cnt[v,l] = d_rowptr_typed[r+1] - d_rowptr_typed[r] (TFGNN_G_ROWPTR_BY_DST).  d_pre may be NULL.
 * Backward: d_grad_pre = d(loss)/d(pre)  ->  d_dZ [V,L,H], d_dfilm [V,L,2H]. */
int tfgnn_film_combine_forward(const float* d_Z, const float* d_film, const int32_t* d_rowptr_typed,
                               const float* d_node_scale, int64_t num_nodes, int num_edge_types, int64_t hidden,
                               int act, float* d_pre, float* d_out, void* stream);
int tfgnn_film_combine_backward(const float* d_grad_pre, const float* d_Z, const float* d_film,
                                const int32_t* d_rowptr_typed, const float* d_node_scale, int64_t num_nodes,
                                int num_edge_types, int64_t hidden, float* d_dZ, float* d_dfilm, void* stream);

/* Batch finalisation of the adjacency lists on the device (tf2_gnn/data/utils.py:9-124, called per batch from
 * data/graph_dataset.py:161-246): the building blocks of process_adjacency_lists.
 *   tfgnn_adjacency_append      d_out[i] = flip ? (dst, src) : (src, dst) of edge i   (_add_backward_edges, :99-113;
 *                               the caller points d_out at the tail of the tied type's list or at a new type's list)
 *   tfgnn_adjacency_self_loops  d_out[i] = (i, i)                                       (_add_self_loop_edges, :88-96)
 *   tfgnn_adjacency_in_degrees  d_counts[v] = number of edges of the list entering v, as float like the reference
 *                               (_compute_type_to_num_inedges, :116-124); targets outside [0, V) are skipped here and
 *                               reported by tfgnn_graph_create
 * Edge lists are int32 [n, 2], 8-byte aligned. */
/* Batching of graph samples into one disjoint-union graph (data/graph_dataset.py:161-246: _add_graph_to_batch shifts
 * graph i's node ids by the nodes before it, :210-222; node_to_graph_map, :211-217; _finalise_batch concatenates).  Per
 * edge type the host lays the graphs' edge lists end to end with their LOCAL node ids (d_local_edges int32 [E, 2]) and
 * passes the prefix sums d_edge_ptr [G+1] (edges of graphs 0..i-1) and d_node_ptr [G+1] (nodes):
 *   tfgnn_batch_offset_edges      d_out[e] = d_local_edges[e] + d_node_ptr[graph of e]; *d_bad_flag (nullable, zeroed by
 *                                 the caller) is set when a local id is outside its graph
 *   tfgnn_batch_node_to_graph_map d_out[v] = graph of node v */
int tfgnn_batch_offset_edges(const int32_t* d_local_edges, int64_t num_edges, const int32_t* d_edge_ptr,
                             const int32_t* d_node_ptr, int num_graphs, int32_t* d_out, int* d_bad_flag, void* stream);
int tfgnn_batch_node_to_graph_map(const int32_t* d_node_ptr, int num_graphs, int64_t num_nodes, int32_t* d_out,
                                  void* stream);
int tfgnn_adjacency_append(const int32_t* d_edges, int64_t num_edges, int flip, int32_t* d_out, void* stream);
int tfgnn_adjacency_self_loops(int64_t num_nodes, int32_t* d_out, void* stream);
int tfgnn_adjacency_in_degrees(const int32_t* d_edges, int64_t num_edges, int64_t num_nodes, float* d_counts,
                               void* stream);
int tfgnn_graph_original_order(const tfgnn_graph* graph, const float* d_weight_by_dst, int32_t* d_src_l,
                               int32_t* d_tgt_l, int32_t* d_tgt_node, float* d_weight, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense layer: C[M,N] = act( op(A)[M,K] @ op(B)[K,N] + bias[N] ) (+ C if accumulate)
 * fp32 in / fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32).
 * Replaces Keras Dense inside dpu_utils MLP (gnn_edge_mlp.py:100, rgin.py:104), rgat.py:102-109,
 * gnn.py:279,324-327, the GRUCell matmuls (ggnn.py:84-87) and their tf.GradientTape gradients
 * (models/graph_task_model.py:347-357).
 *   trans_a = 0: A stored [M,K] (lda >= K);  1: A stored [K,M] (lda >= M)
 *   trans_b = 0: B stored [K,N] (ldb >= N);  1: B stored [N,K] (ldb >= K)
 * d_workspace: scratch for split-K partial sums (may be NULL / 0 -> no split-K).
 * ------------------------------------------------------------------------------------------ */
size_t tfgnn_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int tfgnn_gemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A,
               int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
               const float* d_bias, int act, int accumulate, void* d_workspace,
               size_t workspace_bytes, void* stream);

/* How the fp32 products are evaluated (process-wide; initial value from the environment variable
 * TFGNN_GEMM_MODE = fp32 | bf16x3 | bf16x3_9 | f16x2; unset = f16x2, the mode bench.py is timed in):
 *   TFGNN_GEMM_FP32          v_mfma_f32_32x32x2_f32 on the fp32 operands
 *   TFGNN_GEMM_BF16X3        every fp32 operand is split EXACTLY into three bf16 pieces (x = h + m + l) and
 *                            the six largest piece products - each exact in fp32 - are accumulated in fp32 on
 *                            v_mfma_f32_32x32x16_bf16; dropped terms are < 2^-23 |a b| (csrc/gemm_x3.hip)
 *   TFGNN_GEMM_BF16X3_EXACT  all nine piece products: products exact, only the fp32 accumulation rounds
 *   TFGNN_GEMM_F16X2         (default) the products whose operands come out of a kernel that can write them SPLIT (the gather,
 *                            a product's epilogue, the dropout pass) go to the tfgnn_sp_* entry points - operands as two
 *                            rounded fp16 pieces with a power-of-two block scale, 3 piece products ("Split fp16 operands"
 *                            below); the tfgnn_gemm* entry points themselves run as TFGNN_GEMM_BF16X3.  The choice of entry
 *                            point is the binder's: take the tfgnn_sp_* route while tfgnn_gemm_get_mode() returns
 *                            TFGNN_GEMM_F16X2.  The spread guard of the split weight-gradient product
 *                            (tfgnn_sp_spread_flag) demotes the mode inside the library: once it has tripped,
 *                            tfgnn_gemm_get_mode() returns TFGNN_GEMM_BF16X3 until tfgnn_gemm_set_mode(TFGNN_GEMM_F16X2)
 *                            re-arms it (which waits for the device, so that no factor pass in flight can trip it again).
 * Inputs, outputs and accumulators are fp32 in every mode.  The split modes cover N % 128 == 0 (tiles of 320, 256
 * or 128 columns; NN, NT, TN layouts, K >= 64, 16-byte aligned operands); other shapes run the fp32 kernel
 * whatever the mode. */
enum { TFGNN_GEMM_FP32 = 0, TFGNN_GEMM_F16X2 = 3, TFGNN_GEMM_BF16X3 = 6, TFGNN_GEMM_BF16X3_EXACT = 9 };
int tfgnn_gemm_set_mode(int mode);
int tfgnn_gemm_get_mode(void);

/* The input-gradient product with the element-wise factors of the NEXT backward step applied on the way out:
 *   C[M,N] = (op(A) @ op(B)) * mul * act'(saved)      (+ C[M,N] if accumulate: a second term of the same gradient,
 *                                                       e.g. the target half of an edge MLP's concatenated input)
 * mul    = the dropout mask of tf.nn.dropout on the layer input (gnn.py:285-288), NULL = absent;
 * saved  = what tfgnn_activation_backward takes for `act_of_saved` (the activation's output, or its input for
 *          gelu) of the layer below (message_passing.py:176-177 / Dense activation gnn.py:324-327), NULL = absent.
 * Saves two passes over [V, H] per layer boundary.  Returns TFGNN_ERR_UNSUPPORTED when the active GEMM mode / shape
 * has no fused epilogue (only the split-operand NN / NT kernel has one): the caller then runs tfgnn_gemm, tfgnn_mul
 * and tfgnn_activation_backward. */
int tfgnn_gemm_grad_epilogue(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A,
                             int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                             const float* d_mul, int64_t ld_mul, int act_of_saved, const float* d_saved,
                             int64_t ld_saved, int accumulate, void* d_workspace, size_t workspace_bytes, void* stream);

/* The Dense layer over GATHERED rows: C[m, :] = act( (C[m, :] if accumulate == 2) + A[row_index[m], :] @ op(B) + bias )
 * (+ C[m, :] if accumulate == 1), A [a_rows, lda] fp32, row_index [M] int32 in [0, a_rows); an index outside that range
 * reads as a row of zeros (it never aliases another row).
 * Replaces tf.nn.embedding_lookup(node_embeddings, sources / targets) followed by the first Dense layer of an edge MLP
 * (gnn_edge_mlp.py:84-100 with message_passing.py:195-206): the per-edge input [x_src || x_tgt] W = x_src W_s + x_tgt W_t is
 * two calls (the second with accumulate = 2 and the MLP's activation) and the [E, 2D] concatenation is never written.
 * Only the streaming bf16x3 kernel reads rows through an index (K in {64, 96, 128}, N % 128 == 0, M >= 65536, 16-byte aligned
 * operands, a_rows * lda < 2^30); otherwise TFGNN_ERR_UNSUPPORTED and the caller gathers (tfgnn_gather_reduce with an
 * identity row pointer) and multiplies (tfgnn_gemm) - what the Python mirror does. */
int tfgnn_gemm_gathered(int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A, int64_t lda, int64_t a_rows,
                        const int32_t* d_row_index, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                        const float* d_bias, int act, int accumulate, void* stream);
/* 1 if tfgnn_gemm_gathered takes a product of this shape in the current GEMM mode (16-byte aligned operands assumed), else 0:
 * ask before choosing a formulation that only pays with the fused kernel (one message per edge, gnn_edge_mlp.py:84-100). */
int tfgnn_gemm_gathered_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t a_rows);

/* Grouped forms of the Dense layer for the per-relation multiply over NON-EMPTY buckets only (rows of
 * the stacked operand are grouped by edge type: group g owns rows [d_group_offsets[g],
 * d_group_offsets[g+1]), TFGNN_G_NZ_OFF_*; max_group_rows bounds the launch grid):
 *   tfgnn_gemm_grouped_rows: C[rows g] = act( A[rows g] @ op(B + g*stride_b) )   (A [rows,K], B_g [K,N] or [N,K])
 *   tfgnn_gemm_grouped_k:    C + g*stride_c = A[rows g]^T @ B[rows g]             (A [rows,M], B [rows,N]; the
 *                            weight gradient of relation g; split-K with a deterministic second pass) */
int tfgnn_gemm_grouped_rows(int trans_b, int num_groups, const int32_t* d_group_offsets, int64_t max_group_rows,
                            int64_t N, int64_t K, const float* d_A, int64_t lda, const float* d_B, int64_t ldb,
                            int64_t stride_b, float* d_C, int64_t ldc, int act, void* stream);
/* tfgnn_gemm_grouped_rows (act none) times act'(d_saved) - d_saved [rows, N] indexed like C - in the product's epilogue: the
 * input-gradient product of a per-relation MLP layer (gnn_edge_mlp.py:84-100 under tf.GradientTape) with the derivative of the
 * hidden relu below it.  TFGNN_ERR_UNSUPPORTED when the active kernel has no such epilogue (fp32 mode, odd shapes): the
 * caller runs tfgnn_gemm_grouped_rows and tfgnn_activation_backward. */
int tfgnn_gemm_grouped_rows_grad(int trans_b, int num_groups, const int32_t* d_group_offsets, int64_t max_group_rows, int64_t N,
                                 int64_t K, const float* d_A, int64_t lda, const float* d_B, int64_t ldb, int64_t stride_b,
                                 float* d_C, int64_t ldc, int act_of_saved, const float* d_saved, int64_t ld_saved, void* stream);
size_t tfgnn_gemm_grouped_k_workspace_bytes(int num_groups, int64_t max_group_rows, int64_t M, int64_t N);
int tfgnn_gemm_grouped_k(int num_groups, const int32_t* d_group_offsets, int64_t max_group_rows, int64_t M,
                         int64_t N, const float* d_A, int64_t lda, const float* d_B, int64_t ldb, float* d_C,
                         int64_t ldc, int64_t stride_c, void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise: activations (utils/param_helpers.py:21-39) and their gradients.
 * tfgnn_activation_backward: dx = dy * act'(.) where the derivative is evaluated from the saved
 * OUTPUT y for relu/tanh/leaky_relu/elu/selu/sigmoid and from the saved INPUT x for gelu
 * (pass the matching tensor as d_saved).
 * ------------------------------------------------------------------------------------------ */
int tfgnn_activation_forward(int act, const float* d_x, float* d_y, int64_t n, void* stream);
int tfgnn_activation_backward(int act, const float* d_dy, const float* d_saved, float* d_dx,
                              int64_t n, void* stream);
/* dx = (dy * mul) * act'(saved): tfgnn_mul (the dropout mask of a layer input, gnn.py:285-288) followed by
 * tfgnn_activation_backward (the activation of the layer below) in one pass, bit-equal to the two calls.  16-byte aligned. */
int tfgnn_activation_backward_mul(int act, const float* d_dy, const float* d_saved, const float* d_mul, float* d_dx, int64_t n,
                                  void* stream);

/* ------------------------------------------------------------------------------------------
 * GRUCell gate math ([ext] tf.keras.layers.GRUCell TF2 defaults: reset_after=True, gates z|r|h,
 * sigmoid / tanh), used by GGNN (ggnn.py:64,84-87).  The two matmuls are tfgnn_gemm calls:
 *   mx = x @ kernel + bias[0]   mh = h @ recurrent_kernel + bias[1]      (both [V, 3H])
 * forward:  z = s(mx_z+mh_z), r = s(mx_r+mh_r), c = tanh(mx_h + r*mh_h), h' = z*h + (1-z)*c;
 *           d_gates (nullable) [V,3H] receives z|r|c for the backward pass.
 * backward: d_dmx, d_dmh [V,3H] and d_dh_direct [V,H] = dh' * z.
 * ------------------------------------------------------------------------------------------ */
/* tfgnn_gemm_gru (bf16x3 GEMM modes, H a multiple of 64): the first matmul and the gate math in one kernel -
 *   h' = GRU(mx = x @ kernel + bias[0], mh, h) with mx kept in registers / LDS (never written to HBM).
 * d_kernel_t: the kernel transposed to [3H, K] (K contiguous) with its rows regrouped per block of 192 rows as
 * [z | r | h] of the same 64 units (row t*192 + g*64 + c = column g*H + t*64 + c of the Keras kernel); d_bias
 * (nullable) regrouped likewise.  d_mh as above; d_gates nullable.  TFGNN_ERR_UNSUPPORTED otherwise: callers then
 * use tfgnn_gemm + tfgnn_gru_gates_forward. */
int tfgnn_gemm_gru(int64_t V, int H, int64_t K, const float* d_x, int64_t ld_x, const float* d_kernel_t,
                   const float* d_bias, const float* d_mh, const float* d_h, float* d_h_new, float* d_gates,
                   void* stream);
/* tfgnn_gemm_gru2 (round 6; mode f16x2, H = K in {64, 128}, >= 65536 rows: QM9-sized batches): BOTH products of the cell in the
 * kernel - h' = GRU(x @ kernel + bias[0], h @ recurrent_kernel + bias[1], h) - so that mh [V, 3H] is neither written by a
 * product of its own nor read back.  Both kernels transposed and regrouped as for tfgnn_gemm_gru (d_recurrent_bias likewise,
 * nullable).  d_mh_out (nullable, [V, 3H]): only the candidate third (columns 2H .. 3H) is written - all the backward pass
 * reads of mh (tfgnn_gru_gates_backward*).  TFGNN_ERR_UNSUPPORTED otherwise: callers use tfgnn_gemm + tfgnn_gemm_gru. */
int tfgnn_gemm_gru2(int64_t V, int H, const float* d_x, int64_t ld_x, const float* d_kernel_t, const float* d_bias,
                    const float* d_h, const float* d_recurrent_kernel_t, const float* d_recurrent_bias, float* d_h_new,
                    float* d_gates, float* d_mh_out, void* stream);
int tfgnn_gru_gates_forward(const float* d_mx, const float* d_mh, const float* d_h, float* d_h_new,
                            float* d_gates, int64_t V, int H, void* stream);
int tfgnn_gru_gates_backward(const float* d_dh_new, const float* d_gates, const float* d_mh,
                             const float* d_h, float* d_dmx, float* d_dmh, float* d_dh_direct,
                             int64_t V, int H, void* stream);
/* The same gate gradients for the f16x2 products that consume them (round 3): dmx and dmh [V, 3H] are written ONLY as SP16
 * split operands (one scale per row, d_*_inv_scale [V]) - the operands of tfgnn_sp_gemm_tn (kernel gradients agg^T dmx,
 * h^T dmh) and tfgnn_sp_gemm_nt (d agg = dmx kernel^T, d h += dmh recurrent^T) - and the bias gradients [2, 3H] (column sums
 * of dmx, dmh: fixed-order two-stage reduction) come out of the same pass.  d_out_mul (nullable, [V, H]): an element-wise
 * factor of the state gradient (the dropout mask of the layer's input) applied to d_dh_direct = dh_new * z on the way out.
 * H % 64 == 0, H <= 512 (TFGNN_ERR_UNSUPPORTED otherwise); workspace: tfgnn_gru_gates_backward_sp_workspace_bytes. */
size_t tfgnn_gru_gates_backward_sp_workspace_bytes(int64_t V, int H);
int tfgnn_gru_gates_backward_sp(const float* d_dh_new, const float* d_gates, const float* d_mh, const float* d_h,
                                void* d_dmx_sp, float* d_dmx_inv_scale, void* d_dmh_sp, float* d_dmh_inv_scale,
                                float* d_dh_direct, const float* d_out_mul, float* d_bias_grad, int64_t V, int H,
                                void* d_workspace, size_t workspace_bytes, void* stream);
/* ... with that factor = the mask tfgnn_dropout_forward draws for (dropout_seed, dropout_rate) over the [V, H] layer input,
 * recomputed per element: no mask tensor is read (none exists when the layer input was dropped with d_mask = NULL). */
int tfgnn_gru_gates_backward_sp_dropout(const float* d_dh_new, const float* d_gates, const float* d_mh, const float* d_h,
                                        void* d_dmx_sp, float* d_dmx_inv_scale, void* d_dmh_sp, float* d_dmh_inv_scale,
                                        float* d_dh_direct, float dropout_rate, uint64_t dropout_seed, float* d_bias_grad, int64_t V,
                                        int H, void* d_workspace, size_t workspace_bytes, void* stream);

/* tf.maximum(x, lower) then tf.minimum(., upper) (nodes_to_graph_representation.py:194-197; pass -inf / +inf for an absent
 * bound) and its gradient: dx = dy where lower <= x <= upper, else 0 (TensorFlow's MaximumMinimumGrad). */
int tfgnn_clip(const float* d_x, int64_t n, float lower, float upper, float* d_y, void* stream);
int tfgnn_clip_backward(const float* d_dy, const float* d_x, int64_t n, float lower, float upper, float* d_dx, void* stream);
/* out[n] = sum_m in[m, n] (bias gradients of Dense / GRUCell). */
size_t tfgnn_colsum_workspace_bytes(int64_t M, int N);
int tfgnn_colsum(const float* d_in, int64_t M, int N, int64_t ld, float* d_out, void* d_workspace,
                 size_t workspace_bytes, void* stream);
/* out = alpha * (x + y): residual averaging of the layer stack, gnn.py:291-296. */
int tfgnn_add_scale(const float* d_x, const float* d_y, float alpha, float* d_out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * RGAT (rgat.py:91-163).  Y = X @ [W_0 | ... | W_{L-1}] is a tfgnn_gemm call ([V*L, H] rows (v,l)).
 * tfgnn_rgat_node_scores: the two halves of the attention logit (rgat.py:111-121) per (node,type,head):
 *     s_src[(v,l),k] = <Y[(v,l),k,:], alpha[l,k,:H/K]>   s_tgt[(v,l),k] = <Y[(v,l),k,:], alpha[l,k,H/K:]>
 *     d_alpha: [L, K, 2H/K] (the L "Edge_attention_parameters" stacked, rgat.py:82-86)
 * tfgnn_rgat_edge_scores: score_ek = leaky_relu(s_src[src,l,k] + s_tgt[tgt,l,k]) per bucketed edge (by-dst
 *     order).  Per head, the softmax over all edges entering a node (dpu_utils unsorted_segment_log_softmax
 *     + exp, rgat.py:147-151) is: m = segment max, p = exp(score - m[tgt]), den = segment sum of p,
 *     a = p / den[tgt] - the two segment reductions are tfgnn_graph_gather_reduce over view BY_DST_NODE with
 *     the identity as column array (so hub nodes use the long-row plan), the edge-parallel steps are
 *     tfgnn_rgat_edge_node_op.  The weighted sum out[v,k,:] = sum_e a_ek Y[(src,l),k,:] (rgat.py:154-163) is
 *     tfgnn_graph_gather_reduce(view BY_DST_NODE, ew_heads = K).
 * ------------------------------------------------------------------------------------------ */
int tfgnn_rgat_node_scores(const float* d_Y, const float* d_alpha, int64_t num_nodes, int num_edge_types,
                           int num_heads, int hidden_dim, float* d_s_src, float* d_s_tgt, void* stream);
int tfgnn_rgat_edge_scores(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst, const float* d_s_src,
                           const float* d_s_tgt, int64_t num_edges, int num_edge_types, int num_heads,
                           float* d_scores, void* stream);
/* mode 0: out[e,k] = exp(x[e,k] - node[tgt_e,k]);  mode 1: out[e,k] = x[e,k] / node[tgt_e,k] */
int tfgnn_rgat_edge_node_op(const float* d_x, const int32_t* d_target_by_dst, const float* d_node,
                            int64_t num_edges, int num_heads, int mode, float* d_out, void* stream);
/* backward pieces (tf.GradientTape in the reference):
 *   tfgnn_rgat_edge_dot:              da[e,k] = < d_agg[tgt_e,k,:], Y[(src_e,l_e),k,:] >
 *   tfgnn_rgat_edge_softmax_backward: dz[e,k] = a_ek (da_ek - t[tgt_e,k]) * leaky_relu'(z_ek),
 *                                     t[v,k] = sum over in-edges of a*da (a generic gather)
 *   tfgnn_rgat_scores_backward:       dY[(v,l),k,:] += ds_src[(v,l),k] alpha[l,k,:H/K] + ds_tgt[(v,l),k] alpha[l,k,H/K:] */
int tfgnn_rgat_edge_dot(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst, const float* d_Y,
                        const float* d_dagg, int64_t num_edges, int num_heads, int hidden_dim, float* d_da,
                        void* stream);
int tfgnn_rgat_edge_softmax_backward(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst,
                                     const float* d_s_src, const float* d_s_tgt, const float* d_att,
                                     const float* d_da, const float* d_t, int64_t num_edges, int num_edge_types,
                                     int num_heads, float* d_dz, void* stream);
int tfgnn_rgat_scores_backward(const float* d_ds_src, const float* d_ds_tgt, const float* d_alpha,
                               int64_t num_nodes, int num_edge_types, int num_heads, int hidden_dim,
                               float* d_dY, void* stream);
/* As above, the result dY' = dY + (score terms) written as an SP16 split operand (rows = nodes, L * H columns, one scale per
 * row) - the operand of dX = dY' W^T (tfgnn_sp_gemm_nt) in mode f16x2 - and, when update_fp32 != 0, back into d_dY as well
 * (the weight gradient X^T dY' stays on tfgnn_gemm: the rows of dY' carry attention weights and spread over more than the
 * 2^20 tfgnn_sp_gemm_tn's guard allows).  (H / K) % 4 == 0, L * H <= 2048 and a multiple of 16 (TFGNN_ERR_UNSUPPORTED otherwise). */
int tfgnn_rgat_scores_backward_sp(const float* d_ds_src, const float* d_ds_tgt, const float* d_alpha, float* d_dY,
                                  int update_fp32, int64_t num_nodes, int num_edge_types, int num_heads, int hidden_dim,
                                  void* d_dY_sp, float* d_inv_scale, void* stream);

/* Round 3: the per-target softmax of rgat.py:142-151 row by row over the node view (all incoming edges of a node, every
 * type) instead of edge scores + segment max + exp + segment sum + divide, and its gradient likewise:
 *   forward:  att[e, k] = softmax over the in-edges e of v of leaky_relu(s_src[(src_e, l_e), k] + s_tgt[(v, l_e), k])   [E, K], by-dst order;
 *             d_att_by_src (nullable): the same weights in the by-source edge order (for the backward pass's weighted gather)
 *   backward: dz[e, k]  = att (da - sum_{e' into v} att da) leaky_relu'(z)                  (the gradient w.r.t. the logits)
 * num_heads must be a power of two <= 64 (TFGNN_ERR_UNSUPPORTED otherwise: callers keep the piecewise kernels above).
 * Short rows take one wave, single-item rows of the long-row plan one workgroup, hub rows one workgroup PER ITEM with an
 * in-order combine of the items' (max, sum) pairs in between (d_workspace: tfgnn_rgat_attention_workspace_bytes); online
 * softmax, fixed reduction trees and orders: deterministic. */
size_t tfgnn_rgat_attention_workspace_bytes(const tfgnn_graph* graph, int num_heads);
int tfgnn_rgat_attention_forward(const tfgnn_graph* graph, const float* d_s_src, const float* d_s_tgt, int num_heads,
                                 float* d_att, float* d_att_by_src, void* d_workspace, size_t workspace_bytes, void* stream);
int tfgnn_rgat_attention_backward(const tfgnn_graph* graph, const float* d_s_src, const float* d_s_tgt, const float* d_att,
                                  const float* d_da, int num_heads, float* d_dz, void* d_workspace, size_t workspace_bytes,
                                  void* stream);
/* d alpha_l[k, :H/K] = sum_v ds_src[(v,l), k] Y[(v,l), k, :],  d alpha_l[k, H/K:] = sum_v ds_tgt[(v,l), k] Y[(v,l), k, :]
 * (gradient of the einsum of rgat.py:115-121 w.r.t. the attention parameters): d_alpha_grad [L, K, 2 H/K]; two-stage
 * reduction in a fixed order (workspace: tfgnn_rgat_alpha_grad_workspace_bytes). */
size_t tfgnn_rgat_alpha_grad_workspace_bytes(int64_t num_nodes, int num_edge_types, int hidden_dim);
int tfgnn_rgat_alpha_grad(const float* d_ds_src, const float* d_ds_tgt, const float* d_Y, int64_t num_nodes, int num_edge_types,
                          int num_heads, int hidden_dim, float* d_alpha_grad, void* d_workspace, size_t workspace_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Node -> graph pooling (layers/nodes_to_graph_representation.py:170-229).  node_to_graph_map is
 * sorted (data/graph_dataset.py:211-217), so segments are ranges ptr[g]..ptr[g+1].
 *   tfgnn_segment_offsets: ids [V] -> ptr [G+1]; synchronises once to report unsorted /
 *       out-of-range ids (tf.math.segment_sum raises InvalidArgument for both).
 *   tfgnn_segment_softmax: per (graph, head) dpu_utils unsorted_segment_softmax of the node scores
 *       (:178-186): exp(s - max) / (sum + 1e-7).  scores/out: [V, heads] with row strides ld.
 *   tfgnn_segment_weighted_sum: out[g,h,:] = sum_{v in g} w[v,h] * R[v,h,:] (:219-227); w NULL =
 *       tf.math.segment_sum (:204-210); mean=1 = tf.math.segment_mean (:211-217).
 *   *_backward: gradients of the two ops above (tf.GradientTape in the reference).
 * ------------------------------------------------------------------------------------------ */
int tfgnn_segment_offsets(const int32_t* d_ids, int64_t V, int64_t G, int32_t* d_ptr, void* stream);
/* The same without the validity read-back: no allocation, no synchronisation.  *d_error_flag (zeroed by the caller) gets
 * bit 0 = an id outside [0, G), bit 1 = ids not sorted; the caller reads it when it wants to know (the host mirror does so
 * once per batch and shares the offsets between the pooling layers, the global exchanges and the task head). */
int tfgnn_segment_offsets_async(const int32_t* d_ids, int64_t V, int64_t G, int32_t* d_ptr, int32_t* d_error_flag,
                                void* stream);
int tfgnn_segment_softmax(const float* d_scores, int64_t ld, int heads, const int32_t* d_ptr, int64_t G,
                          float* d_out, int64_t ld_out, void* stream);
int tfgnn_segment_weighted_sum(const float* d_R, const float* d_w, const int32_t* d_ptr, int64_t G, int GD,
                               int heads, int mean, float* d_out, void* stream);
int tfgnn_segment_weighted_sum_backward(const float* d_dOut, const float* d_R, const float* d_w,
                                        const int32_t* d_ids, const int32_t* d_ptr, int64_t V, int GD,
                                        int heads, int mean, float* d_dR, float* d_dW, void* stream);
int tfgnn_segment_softmax_backward(const float* d_w, const float* d_dw, int heads, const int32_t* d_ptr,
                                   int64_t G, float* d_ds, void* stream);

/* Layer-input dropout of the GNN stack (gnn.py:285-288; [ext] tf.nn.dropout scales kept units by
 * 1/(1-rate)).  d_mask receives 0 or 1/(1-rate) per element; the gradient is tfgnn_mul(dy, mask).
 * The stream of random numbers is this library's own (counter-based: element i of a call is a pure function of (seed, i)),
 * not TensorFlow's.  d_mask may be NULL (not stored); d_y == NULL (then d_x is ignored) only (re)generates the mask of
 * (seed, rate) - what tfgnn_sp_gemm_nt_dropout applies in its epilogue. */
int tfgnn_dropout_forward(const float* d_x, float* d_y, float* d_mask, int64_t n, float rate,
                          uint64_t seed, void* stream);
/* The same dropout (same random numbers: element (r, c) draws that of flat index r * cols + c) that also writes its result
 * in the SP16 operand format with one power-of-two scale per row (see "Split fp16 operands"): the layer input is an
 * operand of the weight-gradient product downstream, and splitting it where it is produced saves one pass over it.
 * cols % 16 == 0, cols <= 512; d_out_sp 64-byte aligned rows of ld_out_sp_bytes >= 4 * cols; d_inv_scale [rows]. */
int tfgnn_dropout_forward_sp(const float* d_x, float* d_y, float* d_mask, int64_t rows, int64_t cols, float rate,
                             uint64_t seed, void* d_out_sp, int64_t ld_out_sp_bytes, float* d_inv_scale, void* stream);
/* The dropout EPOCH (round 5).  Every mask this library draws - tfgnn_dropout_forward / _sp, the epilogues of
 * tfgnn_sp_gemm_nt_dropout, the recomputed mask of tfgnn_gru_gates_backward_sp_dropout - is a function of (seed, element,
 * epoch), the epoch being one word of DEVICE memory the kernels read when they start.  It exists for steps replayed from a
 * captured hipGraph (tf2_gnn_amd.capture.CapturedStep): a replay re-launches the same kernels with the same seeds, so the
 * first node of the captured step is tfgnn_dropout_epoch_advance (a one-thread kernel: epoch += 1) and every replay draws
 * fresh masks, forward and backward kernels of one replay the same ones.  Epoch 0 - the state of a process that never
 * advances it - gives exactly the masks of (seed, element) alone.  _set stores a value (a kernel on `stream`), _get reads it
 * back (synchronises `stream`).  [ext] tf.nn.dropout draws new random numbers per call: gnn.py:285-288. */
int tfgnn_dropout_epoch_advance(void* stream);
int tfgnn_dropout_epoch_set(uint32_t value, void* stream);
int tfgnn_dropout_epoch_get(uint32_t* h_value, void* stream);
int tfgnn_mul(const float* d_a, const float* d_b, float* d_out, int64_t n, void* stream);

/* Inter-layer LayerNormalization of the GNN stack (gnn.py:157-161,318-321; [ext] Keras defaults
 * axis=-1, epsilon=1e-3).  backward: d_dx, and d_dy_xhat = dy * xhat whose column sum is d gamma
 * (d beta = column sum of dy; both via tfgnn_colsum). */
int tfgnn_layernorm_forward(const float* d_x, const float* d_gamma, const float* d_beta, float eps,
                            int64_t rows, int H, float* d_y, float* d_mean, float* d_rstd, void* stream);
int tfgnn_layernorm_backward(const float* d_dy, const float* d_x, const float* d_gamma, const float* d_mean,
                             const float* d_rstd, int64_t rows, int H, float* d_dx, float* d_dy_xhat,
                             void* stream);

/* dst[b, a, :] = src[a, b, :]: re-pack the stacked per-edge-type kernels [L, D, H] <-> [D, L, H]
 * (one bias-free kernel per edge type, gnn_edge_mlp.py:73-81) so that a layer's backward pass is two
 * large GEMMs. */
int tfgnn_permute_021(const float* d_src, int64_t A, int64_t B, int64_t C, float* d_dst, void* stream);
/* ------------------------------------------------------------------------------------------
 * Split-operand Dense products on the fp16 matrix cores ("f16x2", csrc/gemm_sp.hip): the same Keras Dense /
 * MatMul-gradient call sites as tfgnn_gemm (gnn_edge_mlp.py:100, rgcn.py:52-56, gnn.py:279,324-327,
 * models/graph_task_model.py:347-357), fp32 in / fp32 accumulate / fp32 out, with the operands handed over
 * already split so that the product kernel only moves data (LDS-DMA) and multiplies.
 *
 * SP16 operand format of an fp32 matrix [rows, cols] (cols % 16 == 0; the same number of bytes as the matrix):
 *   element (r, c) = h + l,  h, l fp16:  byte r * ld_bytes + (c / 16) * 64 + plane * 32 + (c % 16) * 2
 *   (plane 0 = h = fp16(x * 2^e), plane 1 = l = fp16(x * 2^e - h)), one power-of-two scale 2^e per row and per
 *   block of `scale_block` columns chosen so that the block maximum lies in [2^14, 2^15);
 *   d_inv_scale[r * (cols / scale_block) + b] = 2^-e.  |x - (h + l) 2^-e| <= 2^-22 |x| (elements within 2^-3 of the
 *   block maximum) or 2^-39 of the block maximum.
 * tfgnn_sp_split_rows: source element (r, c) = d_src[r * ld + (c / seg_len) * seg_stride + c % seg_len]
 *   (seg_len <= 0: plain row-major rows).  d_fixed_inv_scale (nullable, 1 float on the device): use this 2^-e for the
 *   whole tensor instead of per-block maxima (a caller-side bound; what the weight-gradient product needs).
 * tfgnn_sp_split_cols: SP16 row n, column k = d_src[k * ld + n] (a Keras kernel [K, N] -> the [N, K] operand).
 * tfgnn_sp_gemm_nt:   C[M,N] = epilogue( A[M,K] . B[N,K]^T ), A and B in SP16 (K contiguous);
 *   epilogue = act(. + bias) (+ C if accumulate), then * d_mul * act'(d_saved) as in tfgnn_gemm_grad_epilogue.
 *   a_scale_block: columns per scale block of A (0: K, i.e. one scale per row; < 0: d_a_inv_scale points to ONE scale
 *   for the whole tensor); B has one scale per row (d_b_inv_scale [N]); NULL scales = 1.  N % 128 == 0 (tiles of 320, 256 or 128 columns), K % 16 == 0; TFGNN_ERR_UNSUPPORTED otherwise.
 * ------------------------------------------------------------------------------------------ */
size_t tfgnn_sp_bytes(int64_t rows, int64_t cols);
int tfgnn_sp_split_rows(const float* d_src, int64_t ld, int64_t seg_len, int64_t seg_stride, int64_t rows, int64_t cols,
                        int scale_block, void* d_sp, int64_t ld_sp_bytes, float* d_inv_scale,
                        const float* d_fixed_inv_scale, void* stream);
int tfgnn_sp_split_cols(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                        float* d_inv_scale, void* stream);
/* Both operand forms of `count` (<= 16) stacked kernels [L, D, H] of one shape in a single launch - the per-step weight
 * preparation of a layer stack (after an optimizer update every kernel has to be split again; 2 x layers latency-bound
 * launches otherwise): h_cols_sp[i] receives tfgnn_sp_split_cols of the [L D, H] view (rows = H, cols = L D),
 * h_rows_sp[i] tfgnn_sp_split_rows of [W_0 | ... | W_{L-1}] (rows = D, cols = L H), each with one scale per row.  The
 * pointer arrays are HOST arrays of device pointers. */
int tfgnn_sp_split_weights(int count, const float* const* h_src, int64_t L, int64_t D, int64_t H, void* const* h_cols_sp,
                           float* const* h_cols_inv_scale, void* const* h_rows_sp, float* const* h_rows_inv_scale, void* stream);
int tfgnn_sp_gemm_nt(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                     int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                     int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                     int act_of_saved, const float* d_saved, int64_t ld_saved, void* stream);
/* The same product whose epilogue ALSO writes the result as an SP16 operand with one power-of-two scale per row
 * (d_out_sp rows of ld_out_sp_bytes >= 4 N, 64-byte aligned; d_out_inv_scale [M]) - the operand of the next product
 * (a Dense layer behind a message-passing layer, the weight-gradient products of the backward pass) without a split
 * pass.  The row maximum is taken over the FINAL values (after bias / activation / the gradient factors), per COLUMN TILE
 * of the product (N = 128, 256 or 320: one scale per row; since round 5 also several tiles - N = 512 = 2 x 256, 640 = 2 x 320:
 * d_out_inv_scale [M][N / tile], an operand with scale blocks of one tile, a_scale_block of the next product); no
 * accumulation; d_C may be NULL when only the split form is needed. */
int tfgnn_sp_gemm_nt_sp(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                        int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                        int64_t ldc, const float* d_bias, int act, const float* d_mul, int64_t ld_mul, int act_of_saved,
                        const float* d_saved, int64_t ld_saved, void* d_out_sp, int64_t ld_out_sp_bytes,
                        float* d_out_inv_scale, void* stream);
/* tfgnn_sp_gemm_nt_dropout with the rows of A read through an index (round 5): product row r multiplies operand row
 * d_a_rows[r] (int32 [M], every entry in [0, M); NULL: row r) and takes that row's scales; the operand must be below 4 GB.
 * The input-gradient product dX = [G_0|..|G_{L-1}] W^T of the aggregate-first layers walks the by-source sums in the order of
 * their emptiness patterns this way (d_a_rows = d_row_map = TFGNN_G_PATTERN_NODE_BY_SRC, d_tile_kmask =
 * TFGNN_G_PATTERN_TILEMASK_BY_SRC) and skips the all-zero type blocks of a row tile, while the sums themselves stay in node
 * order, as the weight-gradient product needs them.  Skipped products are exact zeros: bit-equal to the unindexed call. */
int tfgnn_sp_gemm_nt_rows(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                          int a_scale_block, const int32_t* d_a_rows, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale,
                          float* d_C, int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                          int act_of_saved, const float* d_saved, int64_t ld_saved, float saved_scale, void* d_out_sp,
                          int64_t ld_out_sp_bytes, float* d_out_inv_scale, float dropout_rate, uint64_t dropout_seed,
                          const uint8_t* d_tile_kmask, const int32_t* d_row_map, void* stream);

/* Grouped rows (round 5; BASELINE configs[4]: the per-relation MLPs of RGIN / GNN_Edge_MLP over the non-empty (source, type)
 * rows, rgin.py:77-106, gnn_edge_mlp.py:84-107): the M product rows fall into num_groups consecutive groups, group g multiplies
 * ITS OWN [N, K] weight operand: group g's starts b_group_stride_bytes behind group g-1's, its column scales / bias
 * b_scale_group_stride elements behind (stacked operands: N * ldb_bytes and N; column blocks of ONE operand [N, G * K] - what
 * tfgnn_sp_split_cols makes of G stacked Keras kernels [G * K, N], one scale per row shared by the groups: 4 * K and 0).
 * d_tile_table: int32 [num_tiles][4] = {first row, rows (1..128), group, 0} - every row tile lies inside one group (the host
 * cuts each group into tiles of 128 rows; a binder builds it from the group offsets).  d_a_rows / a_src_rows as in
 * tfgnn_sp_gemm_nt_rows, with the source operand having a_src_rows rows (the first MLP layer reads the node states through the
 * row -> node index: no expanded copy of them is made).  The split-form result (d_out_sp) carries one scale per row and column
 * tile: N = 512 gives scale blocks of 256 columns - what the next grouped product takes as a_scale_block = 256.
 * 3 piece products per fp32 product where the bf16x3 grouped kernels (tfgnn_gemm_grouped_rows) spend 6. */
int tfgnn_sp_gemm_nt_grouped(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int a_scale_block, const int32_t* d_a_rows, int64_t a_src_rows, const int32_t* d_tile_table,
                             int64_t num_tiles, int num_groups, const void* d_B_sp, int64_t ldb_bytes, int64_t b_group_stride_bytes,
                             const float* d_b_inv_scale, int64_t b_scale_group_stride, float* d_C, int64_t ldc, const float* d_bias,
                             int act, const float* d_mul, int64_t ld_mul, int act_of_saved, const float* d_saved, int64_t ld_saved,
                             void* d_out_sp, int64_t ld_out_sp_bytes, float* d_out_inv_scale, void* stream);
/* dst row r = src row d_index[r] of an SP16 operand, with its scales_per_row scales (an index outside [0, src_rows) gives a
 * zero row).  The expanded operand of the grouped weight-gradient products: the K dimension of tfgnn_sp_gemm_tn cannot be
 * read through an index. */
int tfgnn_sp_gather_rows(const void* d_src_sp, int64_t ld_src_bytes, const float* d_src_inv_scale, int scales_per_row,
                         const int32_t* d_index, int64_t rows, int64_t src_rows, int64_t cols, void* d_dst_sp, int64_t ld_dst_bytes,
                         float* d_dst_inv_scale, void* stream);

/* tfgnn_sp_gemm_tn with a WIDER operand range (round 5): both operands' fragments carry per-k factors instead of one combined
 * factor on A.  Round 6: the two factors of row k share the deficit of the PAIR - (inv_a[k] / its maximum over the K range) x
 * (inv_b[k] / its maximum) = 2^-e is applied as 2^-ceil(e / 2) on A and 2^-floor(e / 2) on B (per scale block of A: the block
 * with the smallest deficit sets B's share) - so a row pair keeps >= 16 bits while its scale PRODUCT lies within 2^44 of the
 * largest of its K range (ranges of <= 2016 rows, factors computed in the kernel), whichever operand the spread comes from;
 * the spread flag is set beyond that (for pairs of non-zero rows).  For products whose rows are un-normalised sums on both sides (the
 * per-relation weight gradients of RGIN at arxiv scale: row scales over 2^19 and 2^21, their products over 2^25 - the combined
 * factor of tfgnn_sp_gemm_tn trips its 2^20 guard there).  Costs 2 (N / 32) packed multiplies more per k16 step (~+25 %) and
 * more K ranges (workspace: tfgnn_sp_gemm_tn_wide_workspace_bytes).  reduce_job non-NULL: the split reduction as a job of
 * tfgnn_aux_launch, as tfgnn_sp_gemm_tn_deferred.  d_b_inv_scale is required.  Same result layout arguments. */
size_t tfgnn_sp_gemm_tn_wide_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t a_total_cols, int a_scale_block);
int tfgnn_sp_gemm_tn_wide(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                          const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                          int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                          int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                          size_t workspace_bytes, struct tfgnn_aux_job* reduce_job, void* stream);

/* The wide-range product over ROW GROUPS of the same two operands in one launch (round 5; the kernel gradients dW_g = A_g^T B_g
 * of the per-relation MLPs): d_split_table int32 [num_splits][2] = (first row, rows <= 2016) of every K range, ranges of a group
 * consecutive; d_group_split_offsets int32 [num_groups + 1] = first range of every group.  Output g at d_C + g *
 * c_group_stride, element (m, n) at m * stride_row + n * stride_col.  Workspace: 4 * 512 * (a_total_cols / a_scale_block)
 * rounded up to 256, + num_splits * roundup(M, 128) * N * 4 bytes.  num_splits <= 512. */
int tfgnn_sp_gemm_tn_grouped(int64_t M, int64_t N, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int64_t a_total_cols, int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale,
                             int num_groups, const int32_t* d_group_split_offsets, int num_splits, const int32_t* d_split_table,
                             float* d_C, int64_t c_group_stride, int64_t stride_row, int64_t stride_col, void* d_workspace,
                             size_t workspace_bytes, void* stream);

/* K split INSIDE the launch of the NT product, for few row tiles (round 5; BASELINE configs[0]: a PPI batch of 7 110 nodes is 56
 * row tiles - 56 workgroups each streaming the whole weight operand while 200 CUs idle, and the product takes as long as at
 * 30 000 nodes).  With a workspace registered here, products of at most 112 output tiles and K >= 480 launch S = 2..4
 * workgroups per tile (S * tiles <= 232: all resident at once); splits 1 .. S-1 hand their raw accumulators to split 0
 * through the workspace (write-through stores + one flag word each), split 0 adds them in split order - the result is
 * bit-reproducible, though not bit-equal to the unsplit product's (another summation order; same error class) - and runs
 * the epilogue.  No extra launch, no extra pass.  Layout: [64 KB of flags][slabs of 128 x tile-width floats]; 32 MB cover
 * every shape eligible for the K split.  The
 * library keeps the flags zero between launches (each reducer clears what it consumed), which is what makes a product
 * captured in a hipGraph replayable.  ONE workspace per process, owned by the device that was current at registration
 * (round 6): a product on another device is never split; a split product on another STREAM than the previous one
 * first waits (host side) for that stream, so two split products never share the flags in flight (a stream under capture
 * cannot wait: synchronise the device before capturing, as capture.CapturedStep does; while another stream is capturing split
 * products, a product stays unsplit).  d_workspace NULL / too small: no split (the default).  The call waits for the device.
 * A reducer that gives up waiting for a producer (~1 s; never expected) leaves ITS product incomplete: the next product call
 * then fails with TFGNN_ERR_HIP and a message, after clearing the flags (a late producer may have left one set).
 * _status: enable >= 0 switches the split on / off (the workspace stays); *timed_out (may be NULL) = number of such timeouts
 * so far; *split_launches (may be NULL) = products launched with a split so far (tests assert that the path under test
 * really ran); returns 1 if splits can happen. */
int tfgnn_sp_gemm_nt_set_splitk_workspace(void* d_workspace, size_t bytes);
int tfgnn_sp_gemm_nt_splitk_status(int enable, int* timed_out, int64_t* split_launches);

/* The superset: tfgnn_sp_gemm_nt / _sp (d_out_sp may be NULL) with the layer-input dropout of the NEXT op in the epilogue
 * (gnn.py:285-288 - the producer of a layer's input drops it, so the stand-alone pass over [V, H], its mask tensor and the
 * split pass disappear): the result, after bias / activation / the gradient factors, is multiplied by the mask
 * tfgnn_dropout_forward draws for (dropout_seed, dropout_rate) at element index row * N + column (rate 0: no dropout).
 * In a gradient product the same argument RECOMPUTES the forward mask (no mask tensor is read), and `saved_scale` lets
 * d_saved be the DROPPED activation: the derivative is taken at d_saved * saved_scale (= 1 - rate: the kept values carry
 * 1 / (1 - rate); dropped positions have mask 0 anyway).  dropout_seed = UINT64_MAX in a gradient product whose saved tensor
 * is a dropped RELU output: no mask is computed at all - that tensor is positive exactly where the unit was kept and active,
 * relu'(d_saved) carries the mask's zeros and the epilogue only applies 1 / (1 - rate).
 * d_tile_kmask (uint8 per 128-row tile of A, NULL = all) with a_scale_block > 0: bit b of a tile's byte says whether scale block b
 * of K holds anything but zeros in that tile - the product skips the other blocks (at most 8 blocks; TFGNN_G_PATTERN_TILEMASK_BY_DST
 * for an operand written through TFGNN_VIEW_BY_DST_TYPED_PATTERN).  d_row_map (int32 [M], NULL = identity): row r of the product is
 * written (and its d_mul / d_saved / dropout index taken) at row d_row_map[r] (TFGNN_G_PATTERN_NODE_BY_DST: back to node order). */
int tfgnn_sp_gemm_nt_dropout(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                             int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                             int act_of_saved, const float* d_saved, int64_t ld_saved, float saved_scale, void* d_out_sp,
                             int64_t ld_out_sp_bytes, float* d_out_inv_scale, float dropout_rate, uint64_t dropout_seed,
                             const uint8_t* d_tile_kmask, const int32_t* d_row_map, void* stream);

/* tfgnn_sp_gemm_tn: the weight-gradient product C[m, n] = sum_k A[k, a_first_col + m] B[k, b_first_col + n] of two SP16
 * operands stored with K as the row index (dW = X^T G of the Dense / edge-MLP kernels, tf.GradientTape in
 * models/graph_task_model.py:347-357).  Scales: A one 2^-e per (row, block of a_scale_block columns) at
 * d_a_inv_scale[k * (a_total_cols / a_scale_block) + b] (a_scale_block >= 32), B one per row (d_b_inv_scale[k]; NULL =
 * 1) - exactly what the producers write.  A per-row scale is a per-k factor of this product: a first pass turns the two
 * scale arrays into one fp16 factor <= 1 per (k, block) - the scale product over its maximum over k, a power of two - which
 * the kernel multiplies into the A fragments; a row whose scale product is 2^-j of the largest keeps 22 bits while
 * j <= 13, 35 - j bits after that, and drops out beyond 2^-24.
 * M % 16 == 0 (row tiles are 128 columns of A; the last one may run past M: it reads neighbouring bytes of the operand rows and
 * its surplus result rows stay in the workspace), N % 128 == 0, first columns multiples of 16.  Split-K with a deterministic second pass that also
 * scatters the result:
 *   C[(m / group_rows) * stride_group + (m % group_rows) * stride_row + n * stride_col] (+)= value
 * (row-major [M, N]: group_rows = M, stride_row = N, stride_col = 1; dW of stacked kernels [L, D, H] from m = (l, h),
 * n = d: group_rows = H, stride_group = D * H, stride_row = 1, stride_col = H).  d_workspace: 256-byte aligned,
 * tfgnn_sp_gemm_tn_workspace_bytes bytes. */
/* ---- small passes that share a launch ------------------------------------------------------------------------------
 * Around the big kernels of a layer sit passes of 5-15 us each that are bound by launch and dependent-load latency, not by
 * work: weight matrices into SP16 form, the combine pass of the gather's long buckets, the split-K reduction of a weight
 * gradient.  A `tfgnn_aux_job` describes one of them; tfgnn_aux_launch runs up to 8 per launch (more: several launches),
 * every job on its own workgroups - the launch takes as long as its longest job.  Jobs of one call must be independent
 * of each other.  The *_job / *_deferred functions FILL a job (host memory, nothing is launched for it) with exactly the
 * work the function of the same name without the suffix would have launched; a job holds device pointers - keep the
 * buffers alive until the launch has run.  (No reference counterpart: TensorFlow schedules these ops one kernel each.) */
typedef struct tfgnn_aux_job {
  int kind;            /* 0: nothing to do */
  unsigned num_blocks; /* workgroups of 256 threads */
  unsigned char payload[248];
} tfgnn_aux_job;
int tfgnn_aux_launch(const tfgnn_aux_job* jobs, int num_jobs, void* stream);
int tfgnn_sp_split_rows_job(const float* d_src, int64_t ld, int64_t seg_len, int64_t seg_stride, int64_t rows, int64_t cols,
                            int scale_block, void* d_sp, int64_t ld_sp_bytes, float* d_inv_scale,
                            const float* d_fixed_inv_scale, tfgnn_aux_job* job);
int tfgnn_sp_split_cols_job(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                            float* d_inv_scale, tfgnn_aux_job* job);
/* tfgnn_sp_split_cols for LONG K (round 6; the stacked kernels of many relations, BASELINE configs[4]: [40 * 512, 512]): the
 * one-pass job lets each of its K slices take the column maxima over all of K itself - 64-byte row pieces, the matrix read 8
 * times.  Two jobs instead: *maxima_job (whole rows, each byte once, per-slab maxima into d_colmax_workspace,
 * tfgnn_sp_split_cols_two_pass_bytes(K, N) bytes; 0 = K is short, the one-pass job is used and *maxima_job is empty) must run
 * in an EARLIER launch than *split_job.  Same operand, bit for bit (a maximum does not depend on the order). */
size_t tfgnn_sp_split_cols_two_pass_bytes(int64_t K, int64_t N);
int tfgnn_sp_split_cols_jobs(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                             float* d_inv_scale, float* d_colmax_workspace, size_t workspace_bytes, tfgnn_aux_job* maxima_job,
                             tfgnn_aux_job* split_job);
/* tfgnn_graph_gather_reduce_sp; the combine pass of the long buckets comes back in *combine_job (kind 0: none) */
int tfgnn_graph_gather_reduce_sp_deferred(const tfgnn_graph* graph, int view, const int32_t* d_col_override,
                                          const float* d_edge_weight, const float* d_row_scale, const float* d_in,
                                          int64_t ld_in, int width, void* d_out_sp, int64_t ld_out_sp_bytes,
                                          float* d_inv_scale, const float* d_fixed_inv_scale, void* d_workspace,
                                          size_t workspace_bytes, tfgnn_aux_job* combine_job, void* stream);
/* tfgnn_sp_gemm_tn with BOTH of its small passes as jobs (K <= 131072 rows; TFGNN_ERR_UNSUPPORTED above): nothing is
 * launched; run *factors_job in a merged launch, then tfgnn_sp_gemm_tn_phase(2, same arguments), then *reduce_job.  A weight
 * gradient is off the critical path of the backward pass, so the whole product can wait for the next merged launch. */
int tfgnn_sp_gemm_tn_jobs(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                          const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                          int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                          int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                          size_t workspace_bytes, tfgnn_aux_job* factors_job, tfgnn_aux_job* reduce_job);
/* tfgnn_sp_gemm_tn: factor pass (with_factors != 0) and product are launched, the split reduction (+ scatter into d_C) comes
 * back in *reduce_job - a weight gradient is not read before the end of the backward pass, so the reductions of all layers
 * can share one launch.  The workspace must stay untouched until that launch. */
int tfgnn_sp_gemm_tn_deferred(int with_factors, int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes,
                              int64_t a_first_col, const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block,
                              const void* d_B_sp, int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C,
                              int64_t group_rows, int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate,
                              void* d_workspace, size_t workspace_bytes, tfgnn_aux_job* reduce_job, void* stream);

size_t tfgnn_sp_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t a_total_cols, int a_scale_block);
/* Guard of the limit above: the factor pass of every tfgnn_sp_gemm_tn sets a library-wide flag (host-visible without a
 * stream synchronisation; it trails the device by however far the stream is behind) when a NON-ZERO operand row lies more
 * than 2^20 below the largest row scale of its column block.  Between 2^-13 and 2^-20 a row keeps 35 - j bits relative to
 * itself and its ABSOLUTE error stays <= 2^-39 of the largest rows' elements - far below the fp32 rounding of the sum (the
 * measured error of the product against fp64, relative to sum |a||b|, as a function of the spread: tests/test_gpu_f16x2_mode.py,
 * profiles/parity_r03.json); beyond 2^-20 fewer than 15 bits are left and at 2^-24 the row drops out.  Returns the flag
 * (0 / 1), clears it when reset != 0.  The host mirror (tf2_gnn_amd.ops) routes the split-operand layer paths to the
 * exact bf16x3 kernels from the next call on once the flag is seen. */
int tfgnn_sp_spread_flag(int reset);
int tfgnn_sp_gemm_tn(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                     const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                     int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                     int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                     size_t workspace_bytes, void* stream);
/* The three passes of tfgnn_sp_gemm_tn separately - phases is a mask of 1 (per-k factors from the two scale arrays), 2 (the
 * product into the workspace), 4 (split reduction + scaling + scatter into d_C) - with the SAME arguments and workspace in
 * every call: the factor pass only needs the scales and can run on another stream beside the product before it, the
 * reduction beside whatever follows (the caller orders the streams; the workspace must live until the last phase). */
int tfgnn_sp_gemm_tn_phase(int phases, int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                           const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                           int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                           int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                           size_t workspace_bytes, void* stream);

/* Tensor-wide scales (d_fixed_inv_scale of the SP16 producers; tfgnn_sp_gemm_nt takes such an operand with a_scale_block < 0):
 * tfgnn_absmax gives *d_out = max(*d_out, scale * max |x|)
 * (start from 0; order-independent, so reproducible), tfgnn_sp_inv_scale_from_bound turns a bound into the 2^-e that
 * puts it in [2^14, 2^15).  A bound may exceed the true maximum (e.g. max |d_pre| times the largest weighted out-degree
 * for the gathered gradient): every factor of two costs one bit of the 18 the format has beyond fp32's significand. */
int tfgnn_absmax(const float* d_x, int64_t n, float scale, float* d_out, void* stream);
int tfgnn_sp_inv_scale_from_bound(const float* d_bound, float* d_inv_scale, void* stream);

/* tfgnn_graph_gather_reduce with per-head edge weights that ALSO writes, per edge e of the view and head k,
 *   d_dot_out[pos(e) * ew_heads + k] = sum_{f in head k} d_in[col(e), f] * d_dot_rows[r(e) * ld_dot + f]
 * r(e) = the row of the view that holds the edge, pos(e) = d_dot_pos[e] (NULL: the edge's position in the view's order).
 * Reference: the gradient of the attention-weighted message sum w.r.t. the attention values (what tf.GradientTape derives for
 * tf2_gnn/layers/message_passing/rgat.py:153-163) - on the by-source gather of the backward pass that reads d_agg[target(e)]
 * anyway, with d_dot_rows = Y and d_dot_pos = TFGNN_G_SRC2DST_POS.  tfgnn_graph_gather_dot_supported: 1 when the head width
 * (4 .. 64 floats, a power of two) fits the kernel's lane groups, else 0 (callers use tfgnn_rgat_edge_dot). */
int tfgnn_graph_gather_dot_supported(int width, int ew_heads);
int tfgnn_graph_gather_reduce_dot(const tfgnn_graph* g, int view, const float* d_edge_weight, int ew_heads, const float* d_in,
                                  int64_t ld_in, int width, float* d_out, int64_t ld_out, const float* d_dot_rows, int64_t ld_dot,
                                  const int32_t* d_dot_pos, float* d_dot_out, void* d_workspace, size_t workspace_bytes,
                                  void* stream);

/* tfgnn_graph_gather_reduce (plain sums: embedding_lookup + 1/(c+1e-7) scaling + unsorted_segment_sum,
 * message_passing.py:197-206,172-174, gnn_edge_mlp.py:102-106) writing its rows directly as the SP16 operand of
 * tfgnn_sp_gemm_* - the [V, L*D] matrix of per-(node, type) sums is never stored in fp32.  Row r of the view gets
 * its own scale d_inv_scale[r] (for a typed view the rows (v, l) are the (row v, scale block l) of the [V, L*width]
 * operand), or every row uses the caller's *d_fixed_inv_scale.  width % 16 == 0, width <= 512. */
int tfgnn_graph_gather_reduce_sp(const tfgnn_graph* graph, int view, const int32_t* d_col_override,
                                 const float* d_edge_weight, const float* d_row_scale, const float* d_in, int64_t ld_in,
                                 int width, void* d_out_sp, int64_t ld_out_sp_bytes, float* d_inv_scale,
                                 const float* d_fixed_inv_scale, void* d_workspace, size_t workspace_bytes, void* stream);

/* [batch, rows, cols] -> [batch, cols, rows] (LDS-tiled).  Used to hand the GEMMs K-contiguous weights
 * (W^T of gnn_edge_mlp.py:100 / rgcn.py:52-56 kernels) and to bring dW^T = G^T X back to the [L, D, H] layout. */
int tfgnn_transpose_batched(const float* d_src, int64_t batch, int64_t rows, int64_t cols, float* d_dst,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Task metrics behind the path: the losses the reference differentiates
 * and the per-batch numbers its epoch loop aggregates.  One streaming pass + a fixed-order two-stage
 * reduction (reproducible run to run); d_workspace: tfgnn_task_metrics_workspace_bytes() bytes.
 *   tfgnn_sigmoid_ce_metrics: NodeMulticlassTask._fast_task_metrics and micro_f1
 *       (models/node_multiclass_task.py:10-23,62-70).  logits / labels [V, C] fp32 (labels 0 / 1);
 *       d_metrics[0] = mean over nodes of the per-node sum of [ext] tf.nn.sigmoid_cross_entropy_with_logits,
 *       d_metrics[1] = micro-F1 of round(sigmoid(logits)) against int32(labels) (nan when undefined, as
 *       there); d_counts (optional) = {true pos, false pos, false neg}; d_dlogits (optional, [V, C]
 *       contiguous) = d loss / d logits = (sigmoid(x) - z) / V.
 *   tfgnn_regression_metrics: tf.losses.mean_squared_error / mean_absolute_error of per-graph outputs
 *       (models/graph_regression_task.py:157-158, models/qm9_regression.py:122-123): d_metrics = {mse, mae};
 *       d_dpred (optional) = d mse / d pred = 2 (pred - target) / G.
 * ------------------------------------------------------------------------------------------ */
size_t tfgnn_task_metrics_workspace_bytes(void);
int tfgnn_sigmoid_ce_metrics(const float* d_logits, int64_t ld_logits, const float* d_labels, int64_t ld_labels,
                             int64_t V, int64_t C, float* d_metrics, int64_t* d_counts, float* d_dlogits,
                             void* d_workspace, size_t workspace_bytes, void* stream);
int tfgnn_regression_metrics(const float* d_pred, const float* d_target, int64_t G, float* d_metrics,
                             float* d_dpred, void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layer-level entry points (round 6): one call per message-passing layer and pass.
 * Replaces, for one layer, MessagePassing.call (message_passing.py:95-133) with
 * _calculate_messages_per_type (:181-218), GNN_Edge_MLP._message_function (gnn_edge_mlp.py:84-107) and
 * _compute_new_node_embeddings (message_passing.py:135-179) - and their share of tape.gradient
 * (models/graph_task_model.py:347-357) - in the aggregate-first formulation on split operands, the route of RGCN, of
 * GGNN's message part and of GNN_Edge_MLP without hidden layers / target states (sum, mean or sqrt_n aggregation, activation
 * after aggregation; hidden_dim and in_dim multiples of 128 or of 320).  Each function enqueues, on `stream`, exactly the
 * kernels the op-level calls named below would: results are bit-identical to that route.
 *
 * All pointers are device pointers unless named h_*; buffers are the caller's (sizes in comments, V = nodes, L = edge types
 * of the graph handle, D = in_dim, H = hidden_dim).  `struct_size` = sizeof of the struct (a binding built against another
 * header is rejected).  `extra_jobs`: small passes the caller has pending (tfgnn_*_job / *_deferred), launched with this
 * call's own in ONE tfgnn_aux_launch.
 * ------------------------------------------------------------------------------------------ */
typedef enum {
  TFGNN_MP_AGGREGATE_FIRST = 0 /* linear message per edge type, aggregated before the per-relation product */
} tfgnn_mp_kind;

typedef struct tfgnn_mp_forward_args {
  size_t struct_size;
  int kind; /* tfgnn_mp_kind */
  const tfgnn_graph* graph;
  int view;                 /* TFGNN_VIEW_BY_DST_TYPED, or ..._PATTERN with tile_kmask / row_map of the handle */
  const float* x;           /* node states [V, D], row pitch ldx floats */
  int64_t ldx;
  int in_dim, hidden_dim;
  const float* row_scale;   /* [V * L] factor per (node, type) bucket - 1 / (c + 1e-7), tfgnn_graph_scales - or NULL */
  const float* w;           /* stacked kernels [L * D, H] row-major: non-NULL = (re)build the operand below from them */
  void* wt_sp;              /* W^T as SP16 [H, L * D] (row pitch ld_wt_sp_bytes) + wt_inv_scale [H]: kept by the caller */
  int64_t ld_wt_sp_bytes;   /*   across calls while the kernels do not change                                          */
  float* wt_inv_scale;
  void* agg_sp;             /* scratch: the aggregate [V, L * D] SP16 (V * L * D * 4 bytes) + agg_inv_scale [V * L] */
  float* agg_inv_scale;
  const float* bias;        /* [H] or NULL */
  int act;                  /* tfgnn_activation applied to the sum (message_passing.py:176-177) */
  float dropout_rate;       /* > 0: the NEXT op's input dropout applied to the result (tfgnn_sp_gemm_nt_dropout) */
  uint64_t dropout_seed;
  const uint8_t* tile_kmask; /* TFGNN_G_PATTERN_TILEMASK_BY_DST with the PATTERN view, else NULL */
  const int32_t* row_map;    /* TFGNN_G_PATTERN_NODE_BY_DST with the PATTERN view, else NULL */
  float* out;               /* [V, H] fp32, row pitch ld_out floats (may be NULL when out_sp is given) */
  int64_t ld_out;
  void* out_sp;             /* optional: the result as a split operand [V, H] + out_inv_scale [V * (H / tile width)] */
  int64_t ld_out_sp_bytes;
  float* out_inv_scale;
  const struct tfgnn_aux_job* extra_jobs;
  int num_extra_jobs;
  void* workspace;          /* tfgnn_graph_gather_workspace_bytes(graph, view, D) bytes */
  size_t workspace_bytes;
} tfgnn_mp_forward_args;

typedef struct tfgnn_mp_backward_args {
  size_t struct_size;
  int kind;
  const tfgnn_graph* graph;
  const float* d_pre;       /* gradient w.r.t. the layer's pre-activation sums [V, H], row pitch ld_d_pre floats */
  int64_t ld_d_pre;
  int in_dim, hidden_dim;
  const float* edge_weight; /* [E] factor per edge in by-source order (tfgnn_graph_scales d_edge_weight_by_src) or NULL */
  const float* w;           /* stacked kernels [L, D, H]: non-NULL = (re)build the operand below */
  void* wh_sp;              /* rows [W_0[d, :] | W_1[d, :] | ..] as SP16 [D, L * H] + wh_inv_scale [D] */
  int64_t ld_wh_sp_bytes;
  float* wh_inv_scale;
  void* g_sp;               /* scratch: G [V, L * H] SP16 + g_inv_scale [V * L] */
  float* g_inv_scale;
  /* d(node states) = epilogue(G W^T): fp32 result dx [V, D] (added into when accumulate), times mul [V, D] (NULL: 1), times
   * act'(saved * saved_scale) of activation act_of_saved (saved NULL: 1), times the dropout mask of (rate, seed) (rate 0: none);
   * optionally also as a split operand */
  float* dx;
  int64_t ld_dx;
  int accumulate;
  const float* mul;
  int64_t ld_mul;
  int act_of_saved;
  const float* saved;
  int64_t ld_saved;
  float saved_scale;
  float dropout_rate;
  uint64_t dropout_seed;
  void* dx_sp;
  int64_t ld_dx_sp_bytes;
  float* dx_inv_scale;
  const uint8_t* tile_kmask; /* TFGNN_G_PATTERN_TILEMASK_BY_SRC + a_rows = row_map = TFGNN_G_PATTERN_NODE_BY_SRC, or all NULL */
  const int32_t* a_rows;
  const int32_t* row_map;
  /* kernel gradients dW [L, D, H] = X^T G_l (dw NULL: skipped): the layer input as a split operand [V, D], one scale per row */
  float* dw;
  const void* x_sp;
  int64_t ld_x_sp_bytes;
  const float* x_inv_scale;
  void* tn_workspace;       /* tfgnn_sp_gemm_tn_workspace_bytes(L * H, D, V, L * H, H) bytes, 256-byte aligned */
  size_t tn_workspace_bytes;
  const struct tfgnn_aux_job* extra_jobs;
  int num_extra_jobs;
  void* workspace;          /* tfgnn_graph_gather_workspace_bytes(graph, TFGNN_VIEW_BY_SRC_TYPED, H) bytes */
  size_t workspace_bytes;
} tfgnn_mp_backward_args;

/* = tfgnn_graph_gather_reduce_sp_deferred + tfgnn_sp_split_cols_job + tfgnn_aux_launch + tfgnn_sp_gemm_nt_rows */
int tfgnn_mp_forward(const tfgnn_mp_forward_args* args, void* stream);
/* = tfgnn_graph_gather_reduce_sp_deferred + tfgnn_sp_split_rows_job + tfgnn_aux_launch + tfgnn_sp_gemm_nt_rows + tfgnn_sp_gemm_tn */
int tfgnn_mp_backward(const tfgnn_mp_backward_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFGNN_H */
