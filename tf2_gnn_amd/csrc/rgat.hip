// RGAT attention over the bucketed graph (tf2_gnn/layers/message_passing/rgat.py:91-163).
//
// Reference, per edge e = (u -> v) of type l and head k (rgat.py:102-121,142-160):
//   Ys = reshape(x_u W_l, [K, H/K]) ; Yt = reshape(x_v W_l, [K, H/K])
//   score_ek = leaky_relu( <Ys[k], alpha_l[k, :H/K]> + <Yt[k], alpha_l[k, H/K:]> )
//   a_ek     = exp(log_softmax of score_.k over ALL edges entering v, all types)
//   out[v, k, :] = sum_e a_ek * Ys[k, :]
// The two inner products only depend on (node, type, head): they are computed once per node
// (rgat_node_scores) from Y = X W (one MFMA GEMM), so an edge costs two scalar loads and one row
// gather instead of two [E,D]x[D,H] matmuls.
//
//   tfgnn_rgat_node_scores     s_src / s_tgt [V*L, K]
//   tfgnn_rgat_edge_attention  a[e, k] for every bucketed edge (by-dst order); one wave per target
//                              node, lanes over its incoming edges (hub nodes stay parallel)
//   the weighted sum of source rows is the generic gather kernel with per-head edge weights
//   (tfgnn_graph_gather_reduce, ew_heads = K) - see spmm.hip.
// Backward (stand-in for tf.GradientTape): tfgnn_rgat_edge_dot, tfgnn_rgat_attention_backward,
// tfgnn_rgat_scores_backward + generic gathers / GEMMs (layers/message_passing/rgat.py).
#include <algorithm>

#include "common.hpp"

namespace tfgnn {

constexpr int MAX_HEADS = 64;

// s_src[(v,l), k] = <Y[(v,l), k, :], alpha[l, k, :Hk]> ; s_tgt with alpha[l, k, Hk:]
__global__ void __launch_bounds__(256)
rgat_node_scores_kernel(const float* __restrict__ Y, const float* __restrict__ alpha, int64_t rows, int L,
                        int K, int Hk, float* __restrict__ s_src, float* __restrict__ s_tgt) {
  const int64_t total = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / K;
    const int k = (int)(i - row * K);
    const int l = (int)(row % L);
    const float* y = Y + (row * K + k) * Hk;
    const float* a = alpha + ((int64_t)l * K + k) * 2 * Hk;
    float ss = 0.f, st = 0.f;
    for (int j = 0; j < Hk; ++j) {
      const float v = y[j];
      ss += v * a[j];
      st += v * a[Hk + j];
    }
    s_src[i] = ss;
    s_tgt[i] = st;
  }
}

__device__ __forceinline__ float leaky(float z) { return z > 0.f ? z : 0.2f * z; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// one wave per target node: per head, max and sum-exp over the incoming edges, then the
// normalised attention of every edge.  log_softmax then exp (rgat.py:147-151):
// exp(s - m - log(sum)) == exp(s - m) / sum.
__global__ void __launch_bounds__(256)
rgat_edge_attention_kernel(const int32_t* __restrict__ nodeptr, const int32_t* __restrict__ coll,
                           const float* __restrict__ s_src, const float* __restrict__ s_tgt, int64_t V, int L,
                           int K, float* __restrict__ att) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int32_t beg = nodeptr[v], end = nodeptr[v + 1];
  for (int k = 0; k < K; ++k) {
    float m = kFloatLowest;
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = coll[e];
      m = fmaxf(m, leaky(s_src[(int64_t)cl * K + k] + s_tgt[(v * L + cl % L) * K + k]));
    }
    m = wave_max(m);
    float s = 0.f;
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = coll[e];
      s += expf(leaky(s_src[(int64_t)cl * K + k] + s_tgt[(v * L + cl % L) * K + k]) - m);
    }
    s = wave_add(s);
    const float inv = 1.f / s;
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = coll[e];
      att[(int64_t)e * K + k] = expf(leaky(s_src[(int64_t)cl * K + k] + s_tgt[(v * L + cl % L) * K + k]) - m) * inv;
    }
  }
}

// da[e, k] = < d_agg[tgt_e, k, :], Y[(src_e, l_e), k, :] >    (thread per (edge, head))
__global__ void __launch_bounds__(256)
rgat_edge_dot_kernel(const int32_t* __restrict__ coll, const int32_t* __restrict__ tgt, const float* __restrict__ Y,
                     const float* __restrict__ d_agg, int64_t E, int K, int Hk, float* __restrict__ da) {
  const int64_t total = E * K;
  const int H = K * Hk;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / K;
    const int k = (int)(i - e * K);
    const float* y = Y + (int64_t)coll[e] * H + k * Hk;
    const float* g = d_agg + (int64_t)tgt[e] * H + k * Hk;
    float s = 0.f;
    for (int j = 0; j < Hk; ++j) s += y[j] * g[j];
    da[i] = s;
  }
}

// softmax + leaky_relu backward per target node: dz[e,k] = a (da - sum_e' a da) * lrelu'(z)
__global__ void __launch_bounds__(256)
rgat_attention_backward_kernel(const int32_t* __restrict__ nodeptr, const int32_t* __restrict__ coll,
                               const float* __restrict__ s_src, const float* __restrict__ s_tgt,
                               const float* __restrict__ att, const float* __restrict__ da, int64_t V, int L, int K,
                               float* __restrict__ dz) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= V) return;
  const int32_t beg = nodeptr[v], end = nodeptr[v + 1];
  for (int k = 0; k < K; ++k) {
    float t = 0.f;
    for (int32_t e = beg + lane; e < end; e += 64) t += att[(int64_t)e * K + k] * da[(int64_t)e * K + k];
    t = wave_add(t);
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = coll[e];
      const float z = s_src[(int64_t)cl * K + k] + s_tgt[(v * L + cl % L) * K + k];
      const float ds = att[(int64_t)e * K + k] * (da[(int64_t)e * K + k] - t);
      dz[(int64_t)e * K + k] = ds * (z > 0.f ? 1.f : 0.2f);
    }
  }
}

// dY[(v,l), k, i] += ds_src[(v,l),k] * alpha[l,k,i] + ds_tgt[(v,l),k] * alpha[l,k,Hk+i]
__global__ void __launch_bounds__(256)
rgat_scores_backward_kernel(const float* __restrict__ ds_src, const float* __restrict__ ds_tgt,
                            const float* __restrict__ alpha, int64_t rows, int L, int K, int Hk,
                            float* __restrict__ dY) {
  const int H = K * Hk;
  const int64_t total = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / H;
    const int f = (int)(i - row * H);
    const int k = f / Hk, j = f - k * Hk;
    const int l = (int)(row % L);
    const float* a = alpha + ((int64_t)l * K + k) * 2 * Hk;
    dY[i] += ds_src[row * K + k] * a[j] + ds_tgt[row * K + k] * a[Hk + j];
  }
}

static unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256), 16384)); }

static int check_heads(int num_heads, int hidden_dim) {
  TFGNN_REQUIRE(num_heads > 0 && num_heads <= MAX_HEADS && hidden_dim > 0, "bad sizes");
  TFGNN_REQUIRE(hidden_dim % num_heads == 0, "hidden_dim %d is not divisible by num_heads %d", hidden_dim, num_heads);
  return TFGNN_OK;
}

}  // namespace tfgnn

extern "C" int tfgnn_rgat_node_scores(const float* d_Y, const float* d_alpha, int64_t num_nodes,
                                      int num_edge_types, int num_heads, int hidden_dim, float* d_s_src,
                                      float* d_s_tgt, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0, "bad sizes");
  int rc = check_heads(num_heads, hidden_dim);
  if (rc) return rc;
  const int64_t rows = num_nodes * num_edge_types;
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_Y && d_alpha && d_s_src && d_s_tgt, "NULL pointer");
  hipLaunchKernelGGL(rgat_node_scores_kernel, dim3(grid_for(rows * num_heads)), dim3(256), 0, (hipStream_t)stream,
                     d_Y, d_alpha, rows, num_edge_types, num_heads, hidden_dim / num_heads, d_s_src, d_s_tgt);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_edge_attention(const int32_t* d_nodeptr_by_dst, const int32_t* d_coll_by_dst,
                                         const float* d_s_src, const float* d_s_tgt, int64_t num_nodes,
                                         int num_edge_types, int num_heads, float* d_att, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_heads > 0 && num_heads <= MAX_HEADS, "bad sizes");
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_nodeptr_by_dst, "NULL pointer");
  hipLaunchKernelGGL(rgat_edge_attention_kernel, dim3((unsigned)ceil_div(num_nodes, 4)), dim3(256), 0,
                     (hipStream_t)stream, d_nodeptr_by_dst, d_coll_by_dst, d_s_src, d_s_tgt, num_nodes,
                     num_edge_types > 0 ? num_edge_types : 1, num_heads, d_att);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_edge_dot(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst, const float* d_Y,
                                   const float* d_dagg, int64_t num_edges, int num_heads, int hidden_dim,
                                   float* d_da, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0, "bad sizes");
  int rc = check_heads(num_heads, hidden_dim);
  if (rc) return rc;
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_coll_by_dst && d_target_by_dst && d_Y && d_dagg && d_da, "NULL pointer");
  hipLaunchKernelGGL(rgat_edge_dot_kernel, dim3(grid_for(num_edges * num_heads)), dim3(256), 0, (hipStream_t)stream,
                     d_coll_by_dst, d_target_by_dst, d_Y, d_dagg, num_edges, num_heads, hidden_dim / num_heads, d_da);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_attention_backward(const int32_t* d_nodeptr_by_dst, const int32_t* d_coll_by_dst,
                                             const float* d_s_src, const float* d_s_tgt, const float* d_att,
                                             const float* d_da, int64_t num_nodes, int num_edge_types,
                                             int num_heads, float* d_dz, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_heads > 0 && num_heads <= MAX_HEADS, "bad sizes");
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_nodeptr_by_dst, "NULL pointer");
  hipLaunchKernelGGL(rgat_attention_backward_kernel, dim3((unsigned)ceil_div(num_nodes, 4)), dim3(256), 0,
                     (hipStream_t)stream, d_nodeptr_by_dst, d_coll_by_dst, d_s_src, d_s_tgt, d_att, d_da, num_nodes,
                     num_edge_types > 0 ? num_edge_types : 1, num_heads, d_dz);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_scores_backward(const float* d_ds_src, const float* d_ds_tgt, const float* d_alpha,
                                          int64_t num_nodes, int num_edge_types, int num_heads, int hidden_dim,
                                          float* d_dY, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0, "bad sizes");
  int rc = check_heads(num_heads, hidden_dim);
  if (rc) return rc;
  const int64_t rows = num_nodes * num_edge_types;
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_ds_src && d_ds_tgt && d_alpha && d_dY, "NULL pointer");
  hipLaunchKernelGGL(rgat_scores_backward_kernel, dim3(grid_for(rows * hidden_dim)), dim3(256), 0,
                     (hipStream_t)stream, d_ds_src, d_ds_tgt, d_alpha, rows, num_edge_types, num_heads,
                     hidden_dim / num_heads, d_dY);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
