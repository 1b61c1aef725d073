// Per-edge kernels for the one configuration whose edge MLP cannot be moved to the node side:
// GNN_Edge_MLP with use_target_state_as_input=True AND hidden layers (gnn_edge_mlp.py:92-100, the
// class defaults): relu sits between the first and second Dense, per edge.  The first Dense is still
// separable, [x_u | x_v] W = x_u W_s + x_v W_t, so both products are computed per NODE (two MFMA
// GEMMs) and an edge only adds two rows:
//     z0[e] = act( P[(src_e, l_e)] + Q[(tgt_e, l_e)] )        tfgnn_edge_pair_combine
// Edges are processed in the order of the concatenated adjacency lists (type-contiguous), so the
// remaining Dense layers are plain GEMMs over contiguous [E_l, H] blocks.
#include <algorithm>

#include "common.hpp"
#include "graph.hpp"

namespace tfgnn {

// out[e, :] = act(P[ia[e], :] + Q[ib[e], :]);  16 lanes x float4 per edge row when width % 4 == 0
__global__ void __launch_bounds__(256)
edge_pair_combine_kernel(const int32_t* __restrict__ ia, const int32_t* __restrict__ ib, const float* __restrict__ P,
                         const float* __restrict__ Q, int64_t E, int width, int act, float* __restrict__ out) {
  const int vec = (width & 3) == 0;
  if (vec) {
    const int chunks = width >> 2;
    const int64_t total = E * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e = i / chunks;
      const int c = (int)(i - e * chunks) * 4;
      const float4 a = *reinterpret_cast<const float4*>(P + (int64_t)ia[e] * width + c);
      const float4 b = *reinterpret_cast<const float4*>(Q + (int64_t)ib[e] * width + c);
      float4 o;
      o.x = act_apply(act, a.x + b.x);
      o.y = act_apply(act, a.y + b.y);
      o.z = act_apply(act, a.z + b.z);
      o.w = act_apply(act, a.w + b.w);
      *reinterpret_cast<float4*>(out + e * width + c) = o;
    }
  } else {
    const int64_t total = E * width;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e = i / width;
      const int c = (int)(i - e * width);
      out[i] = act_apply(act, P[(int64_t)ia[e] * width + c] + Q[(int64_t)ib[e] * width + c]);
    }
  }
}

// index arrays in the order of the concatenated adjacency lists, derived from the by-dst bucketing
__global__ void __launch_bounds__(256)
original_order_kernel(const int32_t* __restrict__ eid_d, const int32_t* __restrict__ coll_d,
                      const int32_t* __restrict__ tgt_d, const float* __restrict__ w_by_dst, int64_t E, int L,
                      int32_t* __restrict__ src_l, int32_t* __restrict__ tgt_l, int32_t* __restrict__ tgt_node,
                      float* __restrict__ w) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t g = eid_d[p];
    const int32_t cl = coll_d[p];
    const int32_t t = tgt_d[p];
    src_l[g] = cl;
    tgt_l[g] = t * L + cl % L;
    tgt_node[g] = t;
    if (w) w[g] = w_by_dst ? w_by_dst[p] : 1.f;
  }
}

}  // namespace tfgnn

extern "C" int tfgnn_edge_pair_combine(const int32_t* d_index_a, const int32_t* d_index_b, const float* d_P,
                                       const float* d_Q, int64_t num_edges, int width, int act, float* d_out,
                                       void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && width >= 0, "negative size");
  if (num_edges == 0 || width == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_index_a && d_index_b && d_P && d_Q && d_out, "NULL pointer");
  TFGNN_REQUIRE((width & 3) != 0 || (((uintptr_t)d_P | (uintptr_t)d_Q | (uintptr_t)d_out) & 15) == 0,
                "operands must be 16-byte aligned when width is a multiple of 4");
  const int64_t work = num_edges * ((width & 3) == 0 ? width / 4 : width);
  unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(work, 256), 65536));
  hipLaunchKernelGGL(edge_pair_combine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_index_a, d_index_b,
                     d_P, d_Q, num_edges, width, act, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_original_order(const tfgnn_graph* g, const float* d_weight_by_dst, int32_t* d_src_l,
                                          int32_t* d_tgt_l, int32_t* d_tgt_node, float* d_weight, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  if (g->E == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_src_l && d_tgt_l && d_tgt_node, "NULL output");
  {
    const int prc = graph_require_parts(g, TFGNN_GRAPH_PART_EDGE_IDS, "tfgnn_graph_original_order");
    if (prc) return prc;
  }
  unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(g->E, 256), 8192);
  hipLaunchKernelGGL(original_order_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g->eid_d, g->coll_d,
                     g->tgt_d, d_weight_by_dst, g->E, g->L, d_src_l, d_tgt_l, d_tgt_node, d_weight);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- FiLM, per-edge form (gnn_film.py:83-108) --------------------------------------------------------
// modulated[e, :] = gamma[frow_e, :] * (w_e * msg[mrow_e, :]) + beta[frow_e, :]     film = [gamma | beta] rows of width 2 * width
// needed when the modulated messages go through a max aggregation or an activation before the aggregation (the node-side
// form of tfgnn_film_combine_* covers sums).  mrow == NULL: identity; w == NULL: 1.
namespace tfgnn {
__global__ void __launch_bounds__(256)
film_edge_forward_kernel(const float* __restrict__ msg, const int32_t* __restrict__ mrow, const float* __restrict__ film,
                         const int32_t* __restrict__ frow, const float* __restrict__ w, int64_t E, int width,
                         float* __restrict__ out) {
  const int64_t total = E * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / width;
    const int c = (int)(i - e * width);
    const float m = msg[(mrow ? (int64_t)mrow[e] : e) * width + c] * (w ? w[e] : 1.f);
    const float* f = film + (int64_t)frow[e] * 2 * width;
    out[i] = f[c] * m + f[width + c];
  }
}

// d_msg[e, :] = d[e, :] * gamma * w_e      d_film[e, :] = [ d[e, :] * (w_e * msg) | d[e, :] ]   (per edge; the caller sums
// d_film over the edges of a (target, type) bucket and d_msg over the edges sharing a message row)
__global__ void __launch_bounds__(256)
film_edge_backward_kernel(const float* __restrict__ d, const float* __restrict__ msg, const int32_t* __restrict__ mrow,
                          const float* __restrict__ film, const int32_t* __restrict__ frow, const float* __restrict__ w,
                          int64_t E, int width, float* __restrict__ d_msg, float* __restrict__ d_film) {
  const int64_t total = E * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / width;
    const int c = (int)(i - e * width);
    const float we = w ? w[e] : 1.f;
    const float m = msg[(mrow ? (int64_t)mrow[e] : e) * width + c] * we;
    const float g = film[(int64_t)frow[e] * 2 * width + c];
    const float dv = d[i];
    d_msg[i] = dv * g * we;
    d_film[e * 2 * width + c] = dv * m;
    d_film[e * 2 * width + width + c] = dv;
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_film_edge_forward(const float* d_msg, const int32_t* d_msg_row, const float* d_film, const int32_t* d_film_row,
                                       const float* d_edge_weight, int64_t num_edges, int width, float* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && width >= 0, "negative size");
  if (num_edges == 0 || width == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_msg && d_film && d_film_row && d_out, "NULL pointer");
  unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(num_edges * width, 256), 65536));
  hipLaunchKernelGGL(film_edge_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_msg, d_msg_row, d_film, d_film_row,
                     d_edge_weight, num_edges, width, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_film_edge_backward(const float* d_grad, const float* d_msg, const int32_t* d_msg_row, const float* d_film,
                                        const int32_t* d_film_row, const float* d_edge_weight, int64_t num_edges, int width,
                                        float* d_grad_msg, float* d_grad_film, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && width >= 0, "negative size");
  if (num_edges == 0 || width == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_grad && d_msg && d_film && d_film_row && d_grad_msg && d_grad_film, "NULL pointer");
  unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(num_edges * width, 256), 65536));
  hipLaunchKernelGGL(film_edge_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_grad, d_msg, d_msg_row, d_film,
                     d_film_row, d_edge_weight, num_edges, width, d_grad_msg, d_grad_film);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- backward through a general aggregation ---------------------------------------------------------
// forward (spmm.hip, MODE_GENERAL):  agg[t, :] = node_scale[t] * REDUCE_{e -> t} pre_act( w_e * msg[row_e, :] )
// phase 0 (max only): out[e, :] = 1 where the edge attains the maximum of its target, else 0
// phase 1: out[e, :] = d(loss)/d(msg[row_e, :]) contributed by edge e
//   sum: g = grad_agg[t] * node_scale[t];   max: g = selected ? grad_agg[t] / num_selected[t] : 0
//        (the gradient of tf.math.unsorted_segment_max is split evenly among ties [ext])
//   out = g * pre_act'(w_e * msg) * w_e
namespace tfgnn {
struct EdgeAggBwdArgs {
  int64_t E;
  int width;
  const float* msg;
  int64_t ld_msg;
  const int32_t* msg_row;
  const int32_t* target;
  const float* edge_weight;
  const float* node_scale;
  int pre_act;
  int is_max;
  const float* grad_agg;
  const float* agg_max;
  const float* num_selected;
  float* out;
  int phase;
};

__global__ void __launch_bounds__(256) edge_aggregate_backward_kernel(EdgeAggBwdArgs a) {
  const int64_t total = a.E * a.width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / a.width;
    const int c = (int)(i - e * a.width);
    const int64_t row = a.msg_row ? a.msg_row[e] : e;
    const int64_t t = a.target[e];
    const float w = a.edge_weight ? a.edge_weight[e] : 1.f;
    const float u = w * a.msg[row * a.ld_msg + c];
    const float z = act_apply(a.pre_act, u);
    const bool selected = !a.is_max || z == a.agg_max[t * a.width + c];
    float r;
    if (a.phase == 0) {
      r = selected ? 1.f : 0.f;
    } else {
      float gsel;
      if (a.is_max) gsel = selected ? a.grad_agg[t * a.width + c] / a.num_selected[t * a.width + c] : 0.f;
      else gsel = a.grad_agg[t * a.width + c] * (a.node_scale ? a.node_scale[t] : 1.f);
      const float da = a.pre_act == TFGNN_ACT_NONE ? 1.f : act_grad(a.pre_act, a.pre_act == TFGNN_ACT_GELU ? u : z);
      r = gsel * da * w;
    }
    a.out[i] = r;
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_edge_aggregate_backward(int64_t num_edges, int64_t width, const float* d_msg, int64_t ld_msg,
                                             const int32_t* d_msg_row, const int32_t* d_target,
                                             const float* d_edge_weight, const float* d_node_scale, int pre_act,
                                             int reduce_op, const float* d_grad_agg, const float* d_agg_max,
                                             const float* d_num_selected, int phase, float* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && width >= 0 && width < (1 << 30), "bad size");
  TFGNN_REQUIRE(phase == 0 || phase == 1, "phase must be 0 or 1");
  TFGNN_REQUIRE(reduce_op == TFGNN_REDUCE_SUM || reduce_op == TFGNN_REDUCE_MAX, "unknown reduce op");
  if (num_edges == 0 || width == 0) return TFGNN_OK;
  const bool is_max = reduce_op == TFGNN_REDUCE_MAX;
  TFGNN_REQUIRE(d_msg && d_target && d_out && ld_msg >= width, "NULL pointer / bad leading dimension");
  TFGNN_REQUIRE(phase == 0 ? is_max : d_grad_agg != nullptr, "phase 0 is for max aggregation; phase 1 needs grad_agg");
  TFGNN_REQUIRE(!is_max || (d_agg_max && (phase == 0 || d_num_selected)), "max aggregation needs agg_max (+ num_selected)");
  EdgeAggBwdArgs a;
  a.E = num_edges; a.width = (int)width; a.msg = d_msg; a.ld_msg = ld_msg; a.msg_row = d_msg_row; a.target = d_target;
  a.edge_weight = d_edge_weight; a.node_scale = d_node_scale; a.pre_act = pre_act; a.is_max = is_max ? 1 : 0;
  a.grad_agg = d_grad_agg; a.agg_max = d_agg_max; a.num_selected = d_num_selected; a.out = d_out; a.phase = phase;
  const int64_t total = num_edges * width;
  hipLaunchKernelGGL(edge_aggregate_backward_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 65535 * 8)),
                     dim3(256), 0, (hipStream_t)stream, a);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- GNN_FiLM: feature-wise linear modulation of the aggregated messages ----------------------------
// gnn_film.py:84-108 modulates every message by its TARGET: m'_e = gamma[l, tgt_e] * m_e + beta[l, tgt_e].
// All edges of a (target, type) bucket share gamma / beta, so for the sum-like aggregations
//   sum_e m'_e = gamma[v,l] * Z[v,l] + cnt[v,l] * beta[v,l],   Z[v,l] = sum of the bucket's messages,
// and the modulation is a node-side epilogue:  out[v] = act( node_scale[v] * sum_l (...) ).
namespace tfgnn {
__global__ void __launch_bounds__(256)
film_combine_forward_kernel(const float* __restrict__ Z, const float* __restrict__ film, const int32_t* __restrict__ rowptr,
                            const float* __restrict__ node_scale, int64_t V, int L, int H, int act,
                            float* __restrict__ pre, float* __restrict__ out) {
  const int64_t total = V * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / H;
    const int h = (int)(i - v * H);
    float s = 0.f;
    for (int l = 0; l < L; ++l) {
      const int64_t r = v * L + l;
      const float cnt = (float)(rowptr[r + 1] - rowptr[r]);
      const float* f = film + r * 2 * H;
      s += f[h] * Z[r * H + h] + cnt * f[H + h];
    }
    if (node_scale) s *= node_scale[v];
    if (pre) pre[i] = s;
    out[i] = act_apply(act, s);
  }
}

__global__ void __launch_bounds__(256)
film_combine_backward_kernel(const float* __restrict__ d_pre, const float* __restrict__ Z, const float* __restrict__ film,
                             const int32_t* __restrict__ rowptr, const float* __restrict__ node_scale, int64_t V, int L,
                             int H, float* __restrict__ dZ, float* __restrict__ dfilm) {
  const int64_t total = V * L * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H;
    const int h = (int)(i - r * H);
    const int64_t v = r / L;
    float g = d_pre[v * H + h];
    if (node_scale) g *= node_scale[v];
    const float cnt = (float)(rowptr[r + 1] - rowptr[r]);
    dZ[i] = film[r * 2 * H + h] * g;
    dfilm[r * 2 * H + h] = Z[i] * g;
    dfilm[r * 2 * H + H + h] = cnt * g;
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_film_combine_forward(const float* d_Z, const float* d_film, const int32_t* d_rowptr_typed,
                                          const float* d_node_scale, int64_t num_nodes, int num_edge_types,
                                          int64_t hidden, int act, float* d_pre, float* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0 && hidden >= 0 && hidden < (1 << 30), "bad size");
  if (num_nodes == 0 || hidden == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_out && (num_edge_types == 0 || (d_Z && d_film && d_rowptr_typed)), "NULL pointer");
  const int64_t total = num_nodes * hidden;
  hipLaunchKernelGGL(film_combine_forward_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 65535 * 8)),
                     dim3(256), 0, (hipStream_t)stream, d_Z, d_film, d_rowptr_typed, d_node_scale, num_nodes,
                     num_edge_types, (int)hidden, act, d_pre, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_film_combine_backward(const float* d_grad_pre, const float* d_Z, const float* d_film,
                                           const int32_t* d_rowptr_typed, const float* d_node_scale, int64_t num_nodes,
                                           int num_edge_types, int64_t hidden, float* d_dZ, float* d_dfilm,
                                           void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0 && hidden >= 0 && hidden < (1 << 30), "bad size");
  const int64_t total = num_nodes * num_edge_types * hidden;
  if (total == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_grad_pre && d_Z && d_film && d_rowptr_typed && d_dZ && d_dfilm, "NULL pointer");
  hipLaunchKernelGGL(film_combine_backward_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 65535 * 8)),
                     dim3(256), 0, (hipStream_t)stream, d_grad_pre, d_Z, d_film, d_rowptr_typed, d_node_scale, num_nodes,
                     num_edge_types, (int)hidden, d_dZ, d_dfilm);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- batch finalisation of the adjacency lists (tf2_gnn/data/utils.py:9-124) ------------------------
// The reference adds backward edges, self loops and per-type in-degree counts with Python loops over every
// edge while batching (data/graph_dataset.py:161-246); here they are three streaming kernels on the device.
namespace tfgnn {
__global__ void __launch_bounds__(256)
adjacency_append_kernel(const int32_t* __restrict__ in, int64_t n, int flip, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int2 e = reinterpret_cast<const int2*>(in)[i];
    reinterpret_cast<int2*>(out)[i] = flip ? make_int2(e.y, e.x) : e;
  }
}
__global__ void __launch_bounds__(256) adjacency_self_loops_kernel(int64_t V, int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<int2*>(out)[i] = make_int2((int)i, (int)i);
}
__global__ void __launch_bounds__(256)
adjacency_in_degrees_kernel(const int32_t* __restrict__ edges, int64_t n, int64_t V, float* __restrict__ counts,
                            int* __restrict__ bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = edges[2 * i + 1];
    if (d < 0 || d >= V) {
      if (bad) *bad = 1;
      continue;
    }
    atomicAdd(&counts[d], 1.0f);  // integer-valued: exact and order-independent below 2^24 edges per node
  }
}

// ---- batching of graph samples (tf2_gnn/data/graph_dataset.py:202-246) ------------------------------------------
// A batch is the disjoint union of its graphs: graph i's node ids are shifted by the number of nodes of the graphs
// before it (_add_graph_to_batch, :210-222) and node_to_graph_map[v] = i (:211-217).  The host hands over, per edge
// type, the graphs' edge lists laid end to end WITH THEIR LOCAL node ids plus the prefix sums; the offsets are added
// here, one thread per edge / node, by a binary search of the element's position in the prefix array.
__device__ __forceinline__ int upper_bound_minus_one(const int32_t* __restrict__ ptr, int n, int64_t pos) {
  int lo = 0, hi = n;  // ptr[0] = 0 <= pos < ptr[n]; returns i with ptr[i] <= pos < ptr[i + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)ptr[mid] <= pos) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) batch_offset_edges_kernel(const int32_t* __restrict__ local_edges, int64_t num_edges,
                                                                 const int32_t* __restrict__ edge_ptr, const int32_t* __restrict__ node_ptr,
                                                                 int num_graphs, int32_t* __restrict__ out, int* __restrict__ bad) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < num_edges; e += (int64_t)gridDim.x * blockDim.x) {
    const int g = upper_bound_minus_one(edge_ptr, num_graphs, e);
    const int2 le = reinterpret_cast<const int2*>(local_edges)[e];
    const int base = node_ptr[g], n = node_ptr[g + 1] - base;
    if (bad && ((unsigned)le.x >= (unsigned)n || (unsigned)le.y >= (unsigned)n)) *bad = 1;
    reinterpret_cast<int2*>(out)[e] = make_int2(le.x + base, le.y + base);
  }
}

__global__ void __launch_bounds__(256) batch_node_to_graph_kernel(const int32_t* __restrict__ node_ptr, int num_graphs, int64_t num_nodes,
                                                                  int32_t* __restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < num_nodes; v += (int64_t)gridDim.x * blockDim.x)
    out[v] = upper_bound_minus_one(node_ptr, num_graphs, v);
}
}  // namespace tfgnn

extern "C" int tfgnn_adjacency_append(const int32_t* d_edges, int64_t num_edges, int flip, int32_t* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0, "negative size");
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_edges && d_out, "NULL pointer");
  TFGNN_REQUIRE(((uintptr_t)d_edges | (uintptr_t)d_out) % 8 == 0, "edge lists must be 8-byte aligned");
  hipLaunchKernelGGL(adjacency_append_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_edges, 256), 65535)), dim3(256),
                     0, (hipStream_t)stream, d_edges, num_edges, flip, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_adjacency_self_loops(int64_t num_nodes, int32_t* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_nodes < ((int64_t)1 << 31), "bad node count");
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_out && (uintptr_t)d_out % 8 == 0, "NULL / unaligned pointer");
  hipLaunchKernelGGL(adjacency_self_loops_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_nodes, 256), 65535)),
                     dim3(256), 0, (hipStream_t)stream, num_nodes, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_adjacency_in_degrees(const int32_t* d_edges, int64_t num_edges, int64_t num_nodes, float* d_counts,
                                          void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && num_nodes >= 0, "negative size");
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_counts != nullptr, "NULL pointer");
  TFGNN_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)num_nodes * 4, (hipStream_t)stream));
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_edges != nullptr, "NULL pointer");
  hipLaunchKernelGGL(adjacency_in_degrees_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_edges, 256), 65535)),
                     dim3(256), 0, (hipStream_t)stream, d_edges, num_edges, num_nodes, d_counts, (int*)nullptr);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_batch_offset_edges(const int32_t* d_local_edges, int64_t num_edges, const int32_t* d_edge_ptr,
                                        const int32_t* d_node_ptr, int num_graphs, int32_t* d_out, int* d_bad_flag, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && num_graphs >= 0, "negative size");
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_local_edges && d_edge_ptr && d_node_ptr && d_out && num_graphs > 0, "NULL pointer / no graphs");
  TFGNN_REQUIRE(((uintptr_t)d_local_edges | (uintptr_t)d_out) % 8 == 0, "edge lists must be 8-byte aligned");
  hipLaunchKernelGGL(batch_offset_edges_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_edges, 256), 65535)), dim3(256), 0,
                     (hipStream_t)stream, d_local_edges, num_edges, d_edge_ptr, d_node_ptr, num_graphs, d_out, d_bad_flag);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_batch_node_to_graph_map(const int32_t* d_node_ptr, int num_graphs, int64_t num_nodes, int32_t* d_out,
                                             void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_graphs >= 0, "negative size");
  if (num_nodes == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_node_ptr && d_out && num_graphs > 0, "NULL pointer / no graphs");
  hipLaunchKernelGGL(batch_node_to_graph_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_nodes, 256), 65535)), dim3(256), 0,
                     (hipStream_t)stream, d_node_ptr, num_graphs, num_nodes, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
