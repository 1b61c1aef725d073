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
// rgat_aggregate: one wave per target node.  Pass 1 (lanes over edges): per-head max and
// sum-exp of the scores.  Pass 2 (lanes over features, edges in CSR order): a_ek recomputed per
// lane for the head its features belong to, accumulated into registers; a_ek is also written out
// ([E, K], by-dst order) for the backward pass.
#include <algorithm>

#include "common.hpp"

namespace tfgnn {

constexpr int MAX_HEADS = 32;

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

struct RgatArgs {
  const int32_t* nodeptr;  // [V+1] by-dst
  const int32_t* coll;     // [E]  src*L + type, by-dst order
  const float* Y;          // [V*L, H]
  const float* s_src;      // [V*L, K]
  const float* s_tgt;      // [V*L, K]
  int64_t V;
  int L, K, H, Hk;
  int post_act;
  float* out;  // [V, H]
  float* att;  // [E, K] (nullable)
};

// VEC = 4: each lane owns float4 chunks lane, lane+64, ... (requires Hk % 4 == 0); VEC = 1: floats
template <int VEC, int VPL>
__global__ void __launch_bounds__(256) rgat_aggregate_kernel(RgatArgs a) {
  __shared__ float sm_max[4][MAX_HEADS];
  __shared__ float sm_den[4][MAX_HEADS];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t v = (int64_t)blockIdx.x * 4 + w;
  const bool active = v < a.V;
  const int32_t beg = active ? a.nodeptr[v] : 0;
  const int32_t end = active ? a.nodeptr[v + 1] : 0;
  const int K = a.K, L = a.L;

  // ---- pass 1: per-head max / sum-exp over the incoming edges (lanes over edges) ------------
  for (int k = 0; k < K; ++k) {
    float m = kFloatLowest;
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = a.coll[e];
      const int l = cl % L;
      m = fmaxf(m, leaky(a.s_src[(int64_t)cl * K + k] + a.s_tgt[(v * L + l) * K + k]));
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    float s = 0.f;
    for (int32_t e = beg + lane; e < end; e += 64) {
      const int32_t cl = a.coll[e];
      const int l = cl % L;
      s += expf(leaky(a.s_src[(int64_t)cl * K + k] + a.s_tgt[(v * L + l) * K + k]) - m);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) {
      sm_max[w][k] = m;
      sm_den[w][k] = s;
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- pass 2: weighted sum of the source rows (lanes over features) -------------------------
  int head[VPL];
  bool live[VPL], writer[VPL];
  float mx[VPL], inv_den[VPL];
  float acc[VPL][VEC];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int f = (lane + 64 * i) * VEC;
    live[i] = f < a.H;
    head[i] = live[i] ? f / a.Hk : 0;
    writer[i] = live[i] && (f % a.Hk == 0);
    mx[i] = sm_max[w][head[i]];
    // log_softmax then exp (rgat.py:147-151): exp(s - m - log(sum)) == exp(s - m) / sum
    inv_den[i] = 1.f / sm_den[w][head[i]];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[i][c] = 0.f;
  }
  for (int32_t e = beg; e < end; ++e) {
    const int32_t cl = a.coll[e];
    const int l = cl % L;
    const float* yrow = a.Y + (int64_t)cl * a.H;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      if (live[i]) {
        const int f = (lane + 64 * i) * VEC;
        const float sc = leaky(a.s_src[(int64_t)cl * K + head[i]] + a.s_tgt[(v * L + l) * K + head[i]]);
        const float p = expf(sc - mx[i]) * inv_den[i];
        if (writer[i] && a.att) a.att[(int64_t)e * K + head[i]] = p;
        if (VEC == 4) {
          const float4 y = *reinterpret_cast<const float4*>(yrow + f);
          acc[i][0] += p * y.x;
          acc[i][1 % VEC] += p * y.y;
          acc[i][2 % VEC] += p * y.z;
          acc[i][3 % VEC] += p * y.w;
        } else {
          acc[i][0] += p * yrow[f];
        }
      }
    }
  }
  float* orow = a.out + v * a.H;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (live[i]) {
      const int f = (lane + 64 * i) * VEC;
#pragma unroll
      for (int c = 0; c < VEC; ++c) orow[f + c] = act_apply(a.post_act, acc[i][c]);
    }
  }
}

}  // namespace tfgnn

extern "C" int tfgnn_rgat_node_scores(const float* d_Y, const float* d_alpha, int64_t num_nodes,
                                      int num_edge_types, int num_heads, int hidden_dim, float* d_s_src,
                                      float* d_s_tgt, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0 && num_heads > 0 && hidden_dim > 0, "bad sizes");
  TFGNN_REQUIRE(hidden_dim % num_heads == 0, "hidden_dim %d is not divisible by num_heads %d", hidden_dim, num_heads);
  const int64_t rows = num_nodes * num_edge_types;
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_Y && d_alpha && d_s_src && d_s_tgt, "NULL pointer");
  unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(rows * num_heads, 256), 16384);
  hipLaunchKernelGGL(rgat_node_scores_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_Y, d_alpha, rows,
                     num_edge_types, num_heads, hidden_dim / num_heads, d_s_src, d_s_tgt);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_aggregate(const int32_t* d_nodeptr_by_dst, const int32_t* d_coll_by_dst,
                                    const float* d_Y, const float* d_s_src, const float* d_s_tgt,
                                    int64_t num_nodes, int num_edge_types, int num_heads, int hidden_dim,
                                    int post_act, float* d_out, float* d_att, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_heads > 0 && num_heads <= MAX_HEADS && hidden_dim > 0, "bad sizes");
  TFGNN_REQUIRE(hidden_dim % num_heads == 0, "hidden_dim %d is not divisible by num_heads %d", hidden_dim, num_heads);
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_nodeptr_by_dst && d_out, "NULL pointer");
  RgatArgs a{d_nodeptr_by_dst, d_coll_by_dst, d_Y, d_s_src, d_s_tgt, num_nodes, num_edge_types > 0 ? num_edge_types : 1,
             num_heads, hidden_dim, hidden_dim / num_heads, post_act, d_out, d_att};
  dim3 grid((unsigned)ceil_div(num_nodes, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const int Hk = hidden_dim / num_heads;
  const bool vec4 = (Hk % 4 == 0) && ((uintptr_t)d_Y % 16 == 0);
  if (vec4) {
    const int chunks = hidden_dim / 4;
    TFGNN_REQUIRE(chunks <= 64 * 4, "hidden_dim %d too large for the RGAT kernel (max 1024)", hidden_dim);
    if (chunks <= 64) hipLaunchKernelGGL((rgat_aggregate_kernel<4, 1>), grid, block, 0, s, a);
    else if (chunks <= 128) hipLaunchKernelGGL((rgat_aggregate_kernel<4, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rgat_aggregate_kernel<4, 4>), grid, block, 0, s, a);
  } else {
    TFGNN_REQUIRE(hidden_dim <= 64 * 8, "hidden_dim %d with head size %d not a multiple of 4 is limited to 512", hidden_dim, Hk);
    if (hidden_dim <= 64) hipLaunchKernelGGL((rgat_aggregate_kernel<1, 1>), grid, block, 0, s, a);
    else if (hidden_dim <= 128) hipLaunchKernelGGL((rgat_aggregate_kernel<1, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rgat_aggregate_kernel<1, 8>), grid, block, 0, s, a);
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
