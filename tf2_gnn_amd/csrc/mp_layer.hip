// Layer-level entry points (round 6; SURVEY.md section 8(b) planned them, VERDICT r5 "missing 1"): ONE call per message-passing
// layer and pass, for the formulation the benchmarked stacks run - the aggregate-first path of RGCN / GGNN / GNN_Edge_MLP
// without hidden layers and target states ("path A") on split operands:
//
//   forward   A[v, l, :] = s_{l,v} * sum_{(u,v) in A_l} x_u          csr_gather_reduce_kernel, written as the SP16 operand
//             out        = epilogue([A_0 | .. | A_{L-1}] @ [W_0; ..; W_{L-1}])   gemm_sp_nt_kernel (+ bias, activation, the next
//                                                                                 op's input dropout, the split form of the result)
//   backward  G[u, l, :] = sum_{(u,v) in A_l} w_e * d_pre[v, :]       the same gather over the by-source buckets
//             dX         = epilogue(G @ [W_0 | .. | W_{L-1}]^T)        gemm_sp_nt_kernel (+ gradient factors of the op below)
//             dW_l       = X^T G_l                                      gemm_sp_tn_kernel + split reduction
//
// i.e. the reference's message_passing.py:95-218 + gnn_edge_mlp.py:84-107 for one layer, and its share of
// tf.GradientTape.gradient (models/graph_task_model.py:347-357).  Host code only: each function walks the launch sequence the
// Python layer (tf2_gnn_amd/layers/message_passing/gnn_edge_mlp.py: _forward_A / _backward_A_f16x2) used to drive through
// six to eight op-level calls - the same kernels with the same arguments, so results are bit-identical to the op-level route
// (tests/test_gpu_mp_entry.py) - without Python in between.  A binding from another framework needs
// these two calls per layer, the graph handle, and the loss.
#include <cstring>
#include <vector>

#include "common.hpp"
#include "tfgnn.h"

using namespace tfgnn;

namespace {

int launch_small_passes(const tfgnn_aux_job* own, int num_own, const tfgnn_aux_job* extra, int num_extra, void* stream) {
  std::vector<tfgnn_aux_job> jobs;
  jobs.reserve((size_t)num_own + (size_t)(num_extra > 0 ? num_extra : 0));
  for (int i = 0; i < num_extra; ++i)
    if (extra[i].kind != 0 && extra[i].num_blocks != 0) jobs.push_back(extra[i]);
  for (int i = 0; i < num_own; ++i)
    if (own[i].kind != 0 && own[i].num_blocks != 0) jobs.push_back(own[i]);
  if (jobs.empty()) return TFGNN_OK;
  return tfgnn_aux_launch(jobs.data(), (int)jobs.size(), stream);
}

}  // namespace

extern "C" int tfgnn_mp_forward(const tfgnn_mp_forward_args* a, void* stream) {
  TFGNN_REQUIRE(a != nullptr && a->struct_size == sizeof(tfgnn_mp_forward_args),
                "tfgnn_mp_forward: args is NULL or was built against another header (struct_size)");
  TFGNN_REQUIRE(a->kind == TFGNN_MP_AGGREGATE_FIRST, "tfgnn_mp_forward: unknown layer kind %d", a->kind);
  TFGNN_REQUIRE(a->graph && a->x && a->wt_sp && a->wt_inv_scale && a->agg_sp && a->agg_inv_scale && (a->out || a->out_sp),
                "tfgnn_mp_forward: NULL pointer");
  TFGNN_REQUIRE(a->view == TFGNN_VIEW_BY_DST_TYPED || a->view == TFGNN_VIEW_BY_DST_TYPED_PATTERN,
                "tfgnn_mp_forward: the forward pass gathers over the by-target typed buckets (view %d)", a->view);
  TFGNN_REQUIRE(a->num_extra_jobs >= 0 && (a->num_extra_jobs == 0 || a->extra_jobs), "tfgnn_mp_forward: extra jobs");
  int64_t V = 0, E = 0;
  int L = 0;
  int rc = tfgnn_graph_dims(a->graph, &V, &L, &E);
  if (rc) return rc;
  const int64_t D = a->in_dim, H = a->hidden_dim, K = (int64_t)L * D;
  TFGNN_REQUIRE(V > 0 && L > 0 && D > 0 && H > 0, "tfgnn_mp_forward: empty layer");
  tfgnn_aux_job own[2];
  std::memset(own, 0, sizeof(own));
  // 1. the aggregate, one SP16 row of L blocks per node (one scale per (node, type) bucket)
  rc = tfgnn_graph_gather_reduce_sp_deferred(a->graph, a->view, nullptr, nullptr, a->row_scale, a->x, a->ldx, (int)D, a->agg_sp,
                                             D * 4, a->agg_inv_scale, nullptr, a->workspace, a->workspace_bytes, &own[0], stream);
  if (rc) return rc;
  // 2. W^T as the [H, L * D] operand, when the caller's copy is stale (once per weight value)
  if (a->w) {
    rc = tfgnn_sp_split_cols_job(a->w, H, K, H, a->wt_sp, a->ld_wt_sp_bytes, a->wt_inv_scale, &own[1]);
    if (rc) return rc;
  }
  // 3. the small passes in one launch: the combine pass of the long buckets, the weight split, whatever the caller queued
  rc = launch_small_passes(own, 2, a->extra_jobs, a->num_extra_jobs, stream);
  if (rc) return rc;
  // 4. the product with its epilogue
  return tfgnn_sp_gemm_nt_rows(V, H, K, a->agg_sp, K * 4, a->agg_inv_scale, (int)D, nullptr, a->wt_sp, a->ld_wt_sp_bytes, a->wt_inv_scale,
                               a->out, a->ld_out, a->bias, a->act, 0, nullptr, 0, TFGNN_ACT_NONE, nullptr, 0, 1.f, a->out_sp,
                               a->ld_out_sp_bytes, a->out_inv_scale, a->dropout_rate, a->dropout_seed, a->tile_kmask, a->row_map, stream);
}

extern "C" int tfgnn_mp_backward(const tfgnn_mp_backward_args* a, void* stream) {
  TFGNN_REQUIRE(a != nullptr && a->struct_size == sizeof(tfgnn_mp_backward_args),
                "tfgnn_mp_backward: args is NULL or was built against another header (struct_size)");
  TFGNN_REQUIRE(a->kind == TFGNN_MP_AGGREGATE_FIRST, "tfgnn_mp_backward: unknown layer kind %d", a->kind);
  TFGNN_REQUIRE(a->graph && a->d_pre && a->wh_sp && a->wh_inv_scale && a->g_sp && a->g_inv_scale && (a->dx || a->dx_sp),
                "tfgnn_mp_backward: NULL pointer");
  TFGNN_REQUIRE(!a->dw || (a->x_sp && a->x_inv_scale), "tfgnn_mp_backward: the kernel gradients need the layer input as a split operand");
  TFGNN_REQUIRE(a->num_extra_jobs >= 0 && (a->num_extra_jobs == 0 || a->extra_jobs), "tfgnn_mp_backward: extra jobs");
  int64_t V = 0, E = 0;
  int L = 0;
  int rc = tfgnn_graph_dims(a->graph, &V, &L, &E);
  if (rc) return rc;
  const int64_t D = a->in_dim, H = a->hidden_dim, K = (int64_t)L * H;
  TFGNN_REQUIRE(V > 0 && L > 0 && D > 0 && H > 0, "tfgnn_mp_backward: empty layer");
  tfgnn_aux_job own[2];
  std::memset(own, 0, sizeof(own));
  // 1. G = [G_0 | .. | G_{L-1}]: d_pre summed over the out-edges of every (source, type) bucket
  rc = tfgnn_graph_gather_reduce_sp_deferred(a->graph, TFGNN_VIEW_BY_SRC_TYPED, nullptr, a->edge_weight, nullptr, a->d_pre, a->ld_d_pre,
                                                 (int)H, a->g_sp, H * 4, a->g_inv_scale, nullptr, a->workspace, a->workspace_bytes, &own[0],
                                                 stream);
  if (rc) return rc;
  // 2. the kernels as rows [W_0[d, :] | W_1[d, :] | ..] of the [D, L * H] operand, when the caller's copy is stale
  if (a->w) {
    rc = tfgnn_sp_split_rows_job(a->w, H, H, D * H, D, K, (int)K, a->wh_sp, a->ld_wh_sp_bytes, a->wh_inv_scale, nullptr, &own[1]);
    if (rc) return rc;
  }
  rc = launch_small_passes(own, 2, a->extra_jobs, a->num_extra_jobs, stream);
  if (rc) return rc;
  // 3. dX = G W^T with the gradient factors of the op below in the epilogue; rows in by-source pattern order when asked
  rc = tfgnn_sp_gemm_nt_rows(V, D, K, a->g_sp, K * 4, a->g_inv_scale, (int)H, a->a_rows, a->wh_sp, a->ld_wh_sp_bytes, a->wh_inv_scale, a->dx,
                             a->ld_dx, nullptr, TFGNN_ACT_NONE, a->accumulate, a->mul, a->ld_mul, a->act_of_saved, a->saved, a->ld_saved,
                             a->saved_scale, a->dx_sp, a->ld_dx_sp_bytes, a->dx_inv_scale, a->dropout_rate, a->dropout_seed, a->tile_kmask,
                             a->row_map, stream);
  if (rc) return rc;
  // 4. dW[l, d, h] = sum_v X[v, d] G[v, l, h]: element ((l, h), d) of G^T X scattered into the kernels' [L, D, H] layout
  if (a->dw)
    rc = tfgnn_sp_gemm_tn(K, D, V, a->g_sp, K * 4, 0, a->g_inv_scale, K, (int)H, a->x_sp, a->ld_x_sp_bytes, 0, a->x_inv_scale, a->dw, H, D * H, 1,
                          H, 0, a->tn_workspace, a->tn_workspace_bytes, stream);
  return rc;
}
