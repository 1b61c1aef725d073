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
//   tfgnn_rgat_edge_scores / tfgnn_rgat_edge_node_op: edge-parallel pieces of the per-target softmax;
//                              the segment max / sum in between are generic gathers (hub nodes stay parallel)
//   the weighted sum of source rows is the generic gather kernel with per-head edge weights
//   (tfgnn_graph_gather_reduce, ew_heads = K) - see spmm.hip.
// Backward (stand-in for tf.GradientTape): tfgnn_rgat_edge_dot, tfgnn_rgat_edge_softmax_backward,
// tfgnn_rgat_scores_backward + generic gathers / GEMMs (layers/message_passing/rgat.py).
#include <algorithm>

#include "common.hpp"
#include "graph.hpp"
#include "sp16.hpp"

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

// The same with H/4 lanes per row (one float4 each, coalesced) and a shuffle reduction inside each head's lane group:
// the thread-per-(row, head) form above walks Hk floats at a 4 Hk-byte lane stride (32x over-fetch at Hk = 32).
// Needs H % 4 == 0, Hk % 4 == 0, Hk / 4 a power of two and H / 4 a divisor of 64.
__global__ void __launch_bounds__(256)
rgat_node_scores_vec_kernel(const float* __restrict__ Y, const float* __restrict__ alpha, int64_t rows, int L, int K,
                            int Hk, float* __restrict__ s_src, float* __restrict__ s_tgt) {
  const int H = K * Hk, lpe = H >> 2, lph = Hk >> 2;
  const int per_block = 256 / lpe;
  const int l = threadIdx.x % lpe;
  for (int64_t row = (int64_t)blockIdx.x * per_block + threadIdx.x / lpe; row < rows; row += (int64_t)gridDim.x * per_block) {
    const int k = l / lph;
    const float4 y = *reinterpret_cast<const float4*>(Y + row * H + 4 * l);
    const float* a = alpha + ((int64_t)(row % L) * K + k) * 2 * Hk + 4 * (l - k * lph);
    const float4 as = *reinterpret_cast<const float4*>(a);
    const float4 at = *reinterpret_cast<const float4*>(a + Hk);
    float ss = y.x * as.x + y.y * as.y + y.z * as.z + y.w * as.w;
    float st = y.x * at.x + y.y * at.y + y.z * at.z + y.w * at.w;
    for (int d = lph >> 1; d > 0; d >>= 1) {
      ss += __shfl_xor(ss, d, 64);
      st += __shfl_xor(st, d, 64);
    }
    if (l % lph == 0) {
      s_src[row * K + k] = ss;
      s_tgt[row * K + k] = st;
    }
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

// Edge-parallel pieces of the per-target softmax (thread per (edge, head)).  The two segment
// reductions in between (max and sum over all edges entering a node) are the generic gather kernel
// over the node view with its long-row plan, so a hub with 15k incoming edges stays parallel:
//   scores[e,k] = leaky_relu(s_src[(src,l),k] + s_tgt[(tgt,l),k])
//   m[v,k]      = max over in-edges           (tfgnn_graph_gather_reduce, REDUCE_MAX, col = identity)
//   p[e,k]      = exp(scores[e,k] - m[tgt,k])
//   den[v,k]    = sum over in-edges of p      (tfgnn_graph_gather_reduce, REDUCE_SUM)
//   a[e,k]      = p[e,k] / den[tgt,k]         == exp(log_softmax) of rgat.py:147-151
__global__ void __launch_bounds__(256)
rgat_edge_scores_kernel(const int32_t* __restrict__ coll, const int32_t* __restrict__ tgt,
                        const float* __restrict__ s_src, const float* __restrict__ s_tgt, int64_t E, int L, int K,
                        float* __restrict__ scores) {
  const int64_t total = E * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / K;
    const int k = (int)(i - e * K);
    const int32_t cl = coll[e];
    scores[i] = leaky(s_src[(int64_t)cl * K + k] + s_tgt[((int64_t)tgt[e] * L + cl % L) * K + k]);
  }
}

// mode 0: out = exp(x - node[tgt]) ; mode 1: out = x / node[tgt]
__global__ void __launch_bounds__(256)
rgat_edge_node_op_kernel(const float* __restrict__ x, const int32_t* __restrict__ tgt, const float* __restrict__ node,
                         int64_t E, int K, int mode, float* __restrict__ out) {
  const int64_t total = E * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / K;
    const int k = (int)(i - e * K);
    const float nv = node[(int64_t)tgt[e] * K + k];
    out[i] = mode == 0 ? expf(x[i] - nv) : x[i] / nv;
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

// H/4 lanes per edge, coalesced float4 reads of both rows, shuffle reduction per head (see rgat_node_scores_vec_kernel)
__global__ void __launch_bounds__(256)
rgat_edge_dot_vec_kernel(const int32_t* __restrict__ coll, const int32_t* __restrict__ tgt, const float* __restrict__ Y,
                         const float* __restrict__ d_agg, int64_t E, int K, int Hk, float* __restrict__ da) {
  const int H = K * Hk, lpe = H >> 2, lph = Hk >> 2;
  const int per_block = 256 / lpe;
  const int l = threadIdx.x % lpe;
  constexpr int UNR = 4;  // consecutive edges per lane group and iteration: 8 row loads in flight (one edge at a time ran at
                          // half the rate: 138 -> us at cfg-3); consecutive by-dst edges mostly share the d_agg row
  for (int64_t e0 = ((int64_t)blockIdx.x * per_block + threadIdx.x / lpe) * UNR; e0 < E; e0 += (int64_t)gridDim.x * per_block * UNR) {
    int32_t c[UNR], t[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t e = e0 + u < E ? e0 + u : E - 1;
      c[u] = coll[e];
      t[u] = tgt[e];
    }
    float4 y[UNR], g[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      y[u] = *reinterpret_cast<const float4*>(Y + (int64_t)c[u] * H + 4 * l);
      g[u] = *reinterpret_cast<const float4*>(d_agg + (int64_t)t[u] * H + 4 * l);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float sum = y[u].x * g[u].x + y[u].y * g[u].y + y[u].z * g[u].z + y[u].w * g[u].w;
      for (int d = lph >> 1; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
      if (l % lph == 0 && e0 + u < E) da[(e0 + u) * K + l / lph] = sum;
    }
  }
}

// softmax + leaky_relu backward, edge-parallel: dz[e,k] = a (da - t[tgt,k]) * lrelu'(z), where
// t[v,k] = sum over in-edges of a * da comes from the generic gather (node view, col = identity)
__global__ void __launch_bounds__(256)
rgat_edge_softmax_backward_kernel(const int32_t* __restrict__ coll, const int32_t* __restrict__ tgt,
                                  const float* __restrict__ s_src, const float* __restrict__ s_tgt,
                                  const float* __restrict__ att, const float* __restrict__ da,
                                  const float* __restrict__ t, int64_t E, int L, int K, float* __restrict__ dz) {
  const int64_t total = E * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i / K;
    const int k = (int)(i - e * K);
    const int32_t cl = coll[e];
    const int64_t v = tgt[e];
    const float z = s_src[(int64_t)cl * K + k] + s_tgt[(v * L + cl % L) * K + k];
    dz[i] = att[i] * (da[i] - t[v * K + k]) * (z > 0.f ? 1.f : 0.2f);
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


// ------------------------------------------------------------------------------------------------------
// Row-centric softmax over all edges entering a node (round 3): the five launches above (edge scores, segment max, exp,
// segment sum, divide) in one pass structure over the CSR rows of the node view, and its backward (t = sum a da, dz) likewise.
// A row's (edge, head) pairs are laid out lane = slot * K + head (K a power of two): a step covers T / K consecutive edges,
// i.e. T consecutive floats of the [E, K] arrays, U steps in flight per lane.  Work units follow the view's long-row plan:
//   * rows of at most long_threshold edges (short-row list, longest first): one wave each, both passes;
//   * rows that are ONE item of the plan (up to item_chunk edges): one 256-thread workgroup, both passes;
//   * rows cut into several items (a 15 000-edge R-MAT hub: 30 items): every item is a workgroup - pass 1 leaves the item's
//     online-softmax pair (max, sum of exp) [backward: its partial sum] in a scratch slot, a tiny kernel combines the slots
//     of a row IN ITEM ORDER and writes the row's (max, sum) back into each of them, pass 2 normalises item by item.  A hub
//     is thus as parallel as the rest of the batch (one 1024-thread workgroup per hub was the tail: 85 us of 130).
// Every lane re-reads only what it wrote itself, the reductions are fixed trees / fixed orders: deterministic.  The order of
// the max / sum differs from the generic gather's; results agree to fp32 rounding.  Forward also writes the weights in the
// by-source edge order (att_s[dst2src[e]]) for the backward pass's weighted gather.
// ------------------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ float row_reduce(float v, bool is_max, int K, float* red /* [T / 64][64] */) {
  for (int d = K; d < 64; d <<= 1) {
    const float o = __shfl_xor(v, d, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  if (T > 64) {  // lanes with the same head sit at the same lane index of every wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    red[w * 64 + lane] = v;
    __syncthreads();
    float r = red[lane];
#pragma unroll
    for (int i = 1; i < T / 64; ++i) r = is_max ? fmaxf(r, red[i * 64 + lane]) : r + red[i * 64 + lane];
    v = r;
  }
  return v;
}

struct RowSoftmaxArgs {
  const int32_t* nodeptr;
  const int32_t* coll;
  const int32_t* dst2src;    // nullable: by-dst edge position -> by-src position
  const int32_t* rows;       // short rows (wave kernel) / item_row (item kernels)
  const int32_t* item_chunk;
  const int32_t* item_slot;  // < 0: the item is a whole row
  int64_t num_rows;
  int chunk_edges;
  const float* s_src;
  const float* s_tgt;
  int L, K;
  float* att;       // forward: out [E, K]; backward: in
  float* att_s;     // forward, nullable: the same weights in by-src edge order
  const float* da;  // backward
  float* dz;        // backward
  float* part;      // [num_partials][K][2] scratch of the multi-item rows
};

constexpr int RS_U = 8;  // edges per lane and step, loads issued together

// pass 1 over edges [beg, end) of node v: forward -> scores parked in att, running (max, sum of exp) in (m, d);
// backward -> t += att * da
template <bool BWD>
__device__ __forceinline__ void row_pass1(const RowSoftmaxArgs& a, int64_t v, int32_t beg, int32_t end, int slot, int k, int epi,
                                          float& m, float& d) {
  const int K = a.K, L = a.L;
  const int32_t* __restrict__ coll = a.coll;
  const float* __restrict__ s_src = a.s_src;
  const float* __restrict__ st = a.s_tgt + (int64_t)v * L * K + k;
  if (!BWD) {
    float* __restrict__ att = a.att;
    for (int32_t e0 = beg + slot; e0 < end; e0 += RS_U * epi) {
      int32_t cl[RS_U];
      float sc[RS_U];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        cl[u] = coll[e < end ? e : end - 1];
      }
#pragma unroll
      for (int u = 0; u < RS_U; ++u) sc[u] = s_src[(int64_t)cl[u] * K + k] + st[(cl[u] % L) * K];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        if (e < end) {
          const float x = leaky(sc[u]);
          att[(int64_t)e * K + k] = x;
          const float mn = fmaxf(m, x);
          d = d * expf(m - mn) + expf(x - mn);
          m = mn;
        }
      }
    }
  } else {
    const float* __restrict__ att = a.att;
    const float* __restrict__ da = a.da;
    for (int32_t e0 = beg + slot; e0 < end; e0 += RS_U * epi) {
      float x[RS_U], y[RS_U];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        const int64_t i = (int64_t)(e < end ? e : end - 1) * K + k;
        x[u] = att[i];
        y[u] = da[i];
      }
#pragma unroll
      for (int u = 0; u < RS_U; ++u)
        if (e0 + u * epi < end) d += x[u] * y[u];
    }
  }
}

// pass 2: forward -> att = exp(score - M) / D (+ by-src copy); backward -> dz = att (da - D) leaky'(z)   (D = t)
template <bool BWD>
__device__ __forceinline__ void row_pass2(const RowSoftmaxArgs& a, int64_t v, int32_t beg, int32_t end, int slot, int k, int epi,
                                          float M, float D) {
  const int K = a.K, L = a.L;
  if (!BWD) {
    float* __restrict__ att = a.att;
    float* __restrict__ att_s = a.att_s;
    const int32_t* __restrict__ d2s = a.dst2src;
    for (int32_t e0 = beg + slot; e0 < end; e0 += RS_U * epi) {
      float x[RS_U];
      int32_t q[RS_U];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        const int32_t ee = e < end ? e : end - 1;
        x[u] = att[(int64_t)ee * K + k];
        q[u] = att_s ? d2s[ee] : 0;
      }
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        if (e < end) {
          const float w = expf(x[u] - M) / D;
          att[(int64_t)e * K + k] = w;
          if (att_s) att_s[(int64_t)q[u] * K + k] = w;
        }
      }
    }
  } else {
    const int32_t* __restrict__ coll = a.coll;
    const float* __restrict__ s_src = a.s_src;
    const float* __restrict__ st = a.s_tgt + (int64_t)v * L * K + k;
    const float* __restrict__ att = a.att;
    const float* __restrict__ da = a.da;
    float* __restrict__ dz = a.dz;
    for (int32_t e0 = beg + slot; e0 < end; e0 += RS_U * epi) {
      int32_t cl[RS_U];
      float x[RS_U], y[RS_U], z[RS_U];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        const int32_t ee = e < end ? e : end - 1;
        cl[u] = coll[ee];
        x[u] = att[(int64_t)ee * K + k];
        y[u] = da[(int64_t)ee * K + k];
      }
#pragma unroll
      for (int u = 0; u < RS_U; ++u) z[u] = s_src[(int64_t)cl[u] * K + k] + st[(cl[u] % L) * K];
#pragma unroll
      for (int u = 0; u < RS_U; ++u) {
        const int32_t e = e0 + u * epi;
        if (e < end) dz[(int64_t)e * K + k] = x[u] * (y[u] - D) * (z[u] > 0.f ? 1.f : 0.2f);
      }
    }
  }
}

// a whole row with T threads: both passes
template <int T, bool BWD>
__device__ __forceinline__ void row_softmax_whole(const RowSoftmaxArgs& a, int64_t v, int tid, float* red) {
  const int K = a.K, epi = T / K, slot = tid / K, k = tid & (K - 1);
  const int32_t beg = a.nodeptr[v], end = a.nodeptr[v + 1];
  float m = -3.402823466e+38f, d = 0.f;
  row_pass1<BWD>(a, v, beg, end, slot, k, epi, m, d);
  if (!BWD) {
    const float M = row_reduce<T>(m, true, K, red);
    const float D = row_reduce<T>(d * expf(m - M), false, K, red);  // lanes without an edge: d = 0
    row_pass2<false>(a, v, beg, end, slot, k, epi, M, D);
  } else {
    const float t = row_reduce<T>(d, false, K, red);
    row_pass2<true>(a, v, beg, end, slot, k, epi, 0.f, t);
  }
}

template <bool BWD>
__global__ void __launch_bounds__(256) rgat_row_softmax_wave_kernel(RowSoftmaxArgs a) {
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= a.num_rows) return;
  row_softmax_whole<64, BWD>(a, a.rows[w], threadIdx.x & 63, nullptr);
}

// one workgroup per item of the long-row plan.  PHASE 1: whole-row items do both passes; items of multi-item rows pass 1 and
// leave their pair in part[slot].  PHASE 2 (after the combine kernel): items of multi-item rows normalise.
template <bool BWD, int PHASE>
__global__ void __launch_bounds__(256) rgat_row_softmax_item_kernel(RowSoftmaxArgs a) {
  __shared__ float red[4 * 64];
  const int item = blockIdx.x;
  const int32_t sl = a.item_slot[item];
  const int64_t v = a.rows[item];
  if (sl < 0) {
    if (PHASE == 1) row_softmax_whole<256, BWD>(a, v, threadIdx.x, red);
    return;
  }
  const int K = a.K, epi = 256 / K, slot = threadIdx.x / K, k = threadIdx.x & (K - 1);
  const int32_t rbeg = a.nodeptr[v], rend = a.nodeptr[v + 1];
  const int32_t beg = rbeg + a.item_chunk[item] * a.chunk_edges;
  const int32_t end = beg + a.chunk_edges < rend ? beg + a.chunk_edges : rend;
  float* pr = a.part + ((int64_t)sl * K + k) * 2;
  if (PHASE == 1) {
    float m = -3.402823466e+38f, d = 0.f;
    row_pass1<BWD>(a, v, beg, end, slot, k, epi, m, d);
    if (!BWD) {
      const float M = row_reduce<256>(m, true, K, red);
      const float D = row_reduce<256>(d * expf(m - M), false, K, red);
      if (slot == 0) { pr[0] = M; pr[1] = D; }
    } else {
      const float t = row_reduce<256>(d, false, K, red);
      if (slot == 0) pr[0] = t;
    }
  } else {
    row_pass2<BWD>(a, v, beg, end, slot, k, epi, pr[0], BWD ? pr[0] : pr[1]);
  }
}

// one wave per multi-item row: lane k combines the row's item pairs in item order and writes the result back to every slot
template <bool BWD>
__global__ void __launch_bounds__(64) rgat_row_softmax_combine_kernel(const int32_t* __restrict__ multi_base,
                                                                     const int32_t* __restrict__ multi_n, int K, float* __restrict__ part) {
  const int k = threadIdx.x;
  if (k >= K) return;
  const int32_t base = multi_base[blockIdx.x], n = multi_n[blockIdx.x];
  if (!BWD) {
    float M = -3.402823466e+38f;
    for (int i = 0; i < n; ++i) M = fmaxf(M, part[((int64_t)(base + i) * K + k) * 2]);
    float D = 0.f;
    for (int i = 0; i < n; ++i) {
      const float* p = part + ((int64_t)(base + i) * K + k) * 2;
      D += p[1] * expf(p[0] - M);
    }
    for (int i = 0; i < n; ++i) {
      float* p = part + ((int64_t)(base + i) * K + k) * 2;
      p[0] = M;
      p[1] = D;
    }
  } else {
    float t = 0.f;
    for (int i = 0; i < n; ++i) t += part[((int64_t)(base + i) * K + k) * 2];
    for (int i = 0; i < n; ++i) part[((int64_t)(base + i) * K + k) * 2] = t;
  }
}

// d alpha[l, k, :Hk] = sum_v ds_src[(v,l), k] Y[(v,l), k, :] ;  d alpha[l, k, Hk:] likewise with ds_tgt: the block diagonal of
// ds^T Y - two [V, L K]^T x [V, L H] products computed 32x too much of it.  Stage 1: blockIdx.x owns a slice of the nodes,
// a thread owns columns (l, f) and walks the slice (Y is read exactly once, coalesced); stage 2 adds the slices in order.
__global__ void __launch_bounds__(256)
rgat_alpha_grad_partial_kernel(const float* __restrict__ ds_src, const float* __restrict__ ds_tgt, const float* __restrict__ Y,
                               int64_t V, int L, int K, int Hk, int64_t nodes_per_block, float* __restrict__ partial) {
  const int H = K * Hk;
  const int64_t LH = (int64_t)L * H;
  const int64_t v0 = (int64_t)blockIdx.x * nodes_per_block;
  const int64_t v1 = v0 + nodes_per_block < V ? v0 + nodes_per_block : V;
  float* out = partial + (int64_t)blockIdx.x * 2 * LH;
  for (int64_t p = threadIdx.x; p < LH; p += 256) {
    const int l = (int)(p / H), f = (int)(p - (int64_t)l * H), k = f / Hk;
    float as = 0.f, at = 0.f;
#pragma unroll 4
    for (int64_t v = v0; v < v1; ++v) {
      const int64_t row = v * L + l;
      const float y = Y[row * H + f];
      as += ds_src[row * K + k] * y;
      at += ds_tgt[row * K + k] * y;
    }
    out[p] = as;
    out[LH + p] = at;
  }
}

// float4 form (Hk % 4 == 0): a thread owns four consecutive columns (one head) and keeps eight node rows in flight - the scalar
// form above has one 4-byte load per thread and node outstanding (47 us for the 123 MB of Y at configs[2], 2.6 TB/s)
__global__ void __launch_bounds__(256)
rgat_alpha_grad_partial_vec_kernel(const float* __restrict__ ds_src, const float* __restrict__ ds_tgt, const float* __restrict__ Y,
                                   int64_t V, int L, int K, int Hk, int64_t nodes_per_block, float* __restrict__ partial) {
  const int H = K * Hk;
  const int64_t LH = (int64_t)L * H;
  const int64_t v0 = (int64_t)blockIdx.x * nodes_per_block;
  const int64_t v1 = v0 + nodes_per_block < V ? v0 + nodes_per_block : V;
  float* out = partial + (int64_t)blockIdx.x * 2 * LH;
  constexpr int U = 8;
  for (int64_t p = (int64_t)threadIdx.x * 4; p < LH; p += 1024) {
    const int l = (int)(p / H), f = (int)(p - (int64_t)l * H), k = f / Hk;
    float4 as = make_float4(0.f, 0.f, 0.f, 0.f), at = as;
    for (int64_t v = v0; v < v1; v += U) {
      float4 y[U];
      float s[U], t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t vv = v + u < v1 ? v + u : v1 - 1;
        const int64_t row = vv * L + l;
        y[u] = *reinterpret_cast<const float4*>(Y + row * H + f);
        s[u] = ds_src[row * K + k];
        t[u] = ds_tgt[row * K + k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (v + u < v1) {  // (fixed order v0, v0 + 1, ...: the same sums as the scalar form)
          as.x += s[u] * y[u].x; as.y += s[u] * y[u].y; as.z += s[u] * y[u].z; as.w += s[u] * y[u].w;
          at.x += t[u] * y[u].x; at.y += t[u] * y[u].y; at.z += t[u] * y[u].z; at.w += t[u] * y[u].w;
        }
      }
    }
    *reinterpret_cast<float4*>(out + p) = as;
    *reinterpret_cast<float4*>(out + LH + p) = at;
  }
}

// 32 outputs per workgroup, 8 lanes per output: each lane adds a CONTIGUOUS run of the node slices (fixed order), the 8 runs
// are then added in order through LDS - deterministic, and 64 workgroups instead of the 8 of a thread-per-output form
__global__ void __launch_bounds__(256)
rgat_alpha_grad_final_kernel(const float* __restrict__ partial, int nblocks, int L, int K, int Hk, float* __restrict__ d_alpha) {
  __shared__ float red[8][32];
  const int H = K * Hk;
  const int64_t LH = (int64_t)L * H;
  const int o = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + o;
  const int per = (nblocks + 7) / 8;
  const int b0 = part * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  float s = 0.f;
  if (i < 2 * LH) {
#pragma unroll 4
    for (int b = b0; b < b1; ++b) s += partial[(int64_t)b * 2 * LH + i];
  }
  red[part][o] = s;
  __syncthreads();
  if (part == 0 && i < 2 * LH) {
    float t = red[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q][o];
    const int side = (int)(i / LH);
    const int64_t p = i - side * LH;
    const int l = (int)(p / H), f = (int)(p - (int64_t)l * H), k = f / Hk, j = f - k * Hk;
    d_alpha[(((int64_t)l * K + k) * 2 + side) * Hk + j] = t;
  }
}

// float4 form (Hk % 4 == 0): a thread owns four consecutive features of one head
__global__ void __launch_bounds__(256)
rgat_scores_backward_vec_kernel(const float* __restrict__ ds_src, const float* __restrict__ ds_tgt, const float* __restrict__ alpha,
                                int64_t rows, int L, int K, int Hk, float* __restrict__ dY) {
  const int H4 = (K * Hk) >> 2, Hk4 = Hk >> 2;
  const int64_t total = rows * H4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / H4;
    const int f4 = (int)(i - row * H4);
    const int k = f4 / Hk4, j = (f4 - k * Hk4) * 4;
    const float* a = alpha + ((int64_t)(row % L) * K + k) * 2 * Hk;
    const float4 as = *reinterpret_cast<const float4*>(a + j);
    const float4 at = *reinterpret_cast<const float4*>(a + Hk + j);
    const float s = ds_src[row * K + k], t = ds_tgt[row * K + k];
    float4 d = reinterpret_cast<float4*>(dY)[i];
    d.x += s * as.x + t * at.x; d.y += s * as.y + t * at.y; d.z += s * as.z + t * at.z; d.w += s * as.w + t * at.w;
    reinterpret_cast<float4*>(dY)[i] = d;
  }
}

// The same update with the result ALSO written as the split operand of the f16x2 product that consumes it (dX = dY W^T on
// tfgnn_sp_gemm_nt): one wave per node row of C = L H columns (C <= 2048, Hk % 4 == 0), one power-of-two scale per row.  The
// weight gradient X^T dY stays on the exact bf16x3 kernel: dY's rows carry the attention weights of their edges (1e-9 for an
// edge into a 15 000-edge hub) and spread over far more than the 2^20 the split-operand TN product's guard allows.
__global__ void __launch_bounds__(256)
rgat_scores_backward_sp_kernel(const float* __restrict__ ds_src, const float* __restrict__ ds_tgt, const float* __restrict__ alpha,
                               float* __restrict__ dY, int write_fp32, int64_t V, int L, int K, int Hk,
                               uint8_t* __restrict__ out_sp, float* __restrict__ inv) {
  const int H = K * Hk, C = L * H, lane = threadIdx.x & 63;
  constexpr int MAXI = 8;
  for (int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); v < V; v += (int64_t)gridDim.x * 4) {
    float4 d[MAXI];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < C) {
        const int l = c / H, f = c - l * H, k = f / Hk, j = f - k * Hk;
        const int64_t row = v * L + l;
        const float* a = alpha + ((int64_t)l * K + k) * 2 * Hk;
        const float4 as = *reinterpret_cast<const float4*>(a + j);
        const float4 at = *reinterpret_cast<const float4*>(a + Hk + j);
        const float s = ds_src[row * K + k], t = ds_tgt[row * K + k];
        float4 x = *reinterpret_cast<const float4*>(dY + v * C + c);
        x.x += s * as.x + t * at.x; x.y += s * as.y + t * at.y; x.z += s * as.z + t * at.z; x.w += s * as.w + t * at.w;
        d[i] = x;
        if (write_fp32) *reinterpret_cast<float4*>(dY + v * C + c) = x;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
      }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float iv;
    const float sc = sp_scale_for_max(mx, &iv);
    if (lane == 0) inv[v] = iv;
    uint8_t* drow = out_sp + v * (int64_t)C * 4;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int c = 4 * (lane + 64 * i);
      if (c < C) sp_store4(drow, c, d[i], sc);
    }
  }
}

static unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256), 16384)); }

// shapes the lane-group kernels take: H / 4 lanes per row inside one wave, Hk / 4 lanes per head, both powers of two
static bool rgat_vec_shape(int num_heads, int hidden_dim) {
  if (num_heads <= 0 || hidden_dim % num_heads) return false;
  const int hk = hidden_dim / num_heads;
  if (hk % 4 || hidden_dim % 4) return false;
  const int lph = hk / 4, lpe = hidden_dim / 4;
  return (lph & (lph - 1)) == 0 && (lpe & (lpe - 1)) == 0 && lpe <= 64;
}

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
  if (rgat_vec_shape(num_heads, hidden_dim) && ((uintptr_t)d_Y | (uintptr_t)d_alpha) % 16 == 0) {
    const int per_block = 256 / (hidden_dim / 4);
    hipLaunchKernelGGL(rgat_node_scores_vec_kernel, dim3(grid_for(ceil_div(rows, per_block) * 256)), dim3(256), 0,
                       (hipStream_t)stream, d_Y, d_alpha, rows, num_edge_types, num_heads, hidden_dim / num_heads, d_s_src,
                       d_s_tgt);
  } else {
    hipLaunchKernelGGL(rgat_node_scores_kernel, dim3(grid_for(rows * num_heads)), dim3(256), 0, (hipStream_t)stream,
                       d_Y, d_alpha, rows, num_edge_types, num_heads, hidden_dim / num_heads, d_s_src, d_s_tgt);
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_edge_scores(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst,
                                      const float* d_s_src, const float* d_s_tgt, int64_t num_edges,
                                      int num_edge_types, int num_heads, float* d_scores, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && num_heads > 0, "bad sizes");
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_coll_by_dst && d_target_by_dst && d_s_src && d_s_tgt && d_scores, "NULL pointer");
  hipLaunchKernelGGL(rgat_edge_scores_kernel, dim3(grid_for(num_edges * num_heads)), dim3(256), 0, (hipStream_t)stream,
                     d_coll_by_dst, d_target_by_dst, d_s_src, d_s_tgt, num_edges, num_edge_types > 0 ? num_edge_types : 1,
                     num_heads, d_scores);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_edge_node_op(const float* d_x, const int32_t* d_target_by_dst, const float* d_node,
                                       int64_t num_edges, int num_heads, int mode, float* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && num_heads > 0 && (mode == 0 || mode == 1), "bad arguments");
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_target_by_dst && d_node && d_out, "NULL pointer");
  hipLaunchKernelGGL(rgat_edge_node_op_kernel, dim3(grid_for(num_edges * num_heads)), dim3(256), 0, (hipStream_t)stream,
                     d_x, d_target_by_dst, d_node, num_edges, num_heads, mode, d_out);
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
  if (rgat_vec_shape(num_heads, hidden_dim) && ((uintptr_t)d_Y | (uintptr_t)d_dagg) % 16 == 0) {
    const int per_block = 256 / (hidden_dim / 4);
    hipLaunchKernelGGL(rgat_edge_dot_vec_kernel, dim3(grid_for(ceil_div(num_edges, per_block * 4) * 256)), dim3(256), 0,
                       (hipStream_t)stream, d_coll_by_dst, d_target_by_dst, d_Y, d_dagg, num_edges, num_heads,
                       hidden_dim / num_heads, d_da);
  } else {
    hipLaunchKernelGGL(rgat_edge_dot_kernel, dim3(grid_for(num_edges * num_heads)), dim3(256), 0, (hipStream_t)stream,
                       d_coll_by_dst, d_target_by_dst, d_Y, d_dagg, num_edges, num_heads, hidden_dim / num_heads, d_da);
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_edge_softmax_backward(const int32_t* d_coll_by_dst, const int32_t* d_target_by_dst,
                                                const float* d_s_src, const float* d_s_tgt, const float* d_att,
                                                const float* d_da, const float* d_t, int64_t num_edges,
                                                int num_edge_types, int num_heads, float* d_dz, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_edges >= 0 && num_heads > 0, "bad sizes");
  if (num_edges == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_coll_by_dst && d_target_by_dst && d_s_src && d_s_tgt && d_att && d_da && d_t && d_dz, "NULL pointer");
  hipLaunchKernelGGL(rgat_edge_softmax_backward_kernel, dim3(grid_for(num_edges * num_heads)), dim3(256), 0,
                     (hipStream_t)stream, d_coll_by_dst, d_target_by_dst, d_s_src, d_s_tgt, d_att, d_da, d_t, num_edges,
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
  if ((hidden_dim / num_heads) % 4 == 0 && ((uintptr_t)d_dY | (uintptr_t)d_alpha) % 16 == 0) {
    hipLaunchKernelGGL(rgat_scores_backward_vec_kernel, dim3(grid_for(rows * hidden_dim / 4)), dim3(256), 0, (hipStream_t)stream,
                       d_ds_src, d_ds_tgt, d_alpha, rows, num_edge_types, num_heads, hidden_dim / num_heads, d_dY);
  } else {
    hipLaunchKernelGGL(rgat_scores_backward_kernel, dim3(grid_for(rows * hidden_dim)), dim3(256), 0,
                       (hipStream_t)stream, d_ds_src, d_ds_tgt, d_alpha, rows, num_edge_types, num_heads,
                       hidden_dim / num_heads, d_dY);
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

static int rgat_row_softmax(const tfgnn_graph* g, const float* s_src, const float* s_tgt, int K, float* att, float* att_s,
                            const float* da, float* dz, bool bwd, void* workspace, size_t workspace_bytes, hipStream_t s) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  if (K <= 0 || K > MAX_HEADS || (K & (K - 1))) return TFGNN_ERR_UNSUPPORTED;  // lane = slot * K + head needs a power of two
  if (g->E == 0) return TFGNN_OK;
  TFGNN_REQUIRE(s_src && s_tgt && att && (!bwd || (da && dz)), "NULL pointer");
  {
    const int prc = graph_require_parts(g, TFGNN_GRAPH_PART_PLAN_NODE | TFGNN_GRAPH_PART_EDGE_MAPS, "tfgnn_rgat_attention");
    if (prc) return prc;
  }
  const GraphView& gv = g->views[1];  // by target, all edge types of a node in one row
  const CsrPlan& pl = gv.plan;
  const size_t need = (size_t)pl.num_partials * K * 2 * 4;
  TFGNN_REQUIRE(pl.num_partials == 0 || (workspace && workspace_bytes >= need), "workspace too small: need %zu bytes", need);
  RowSoftmaxArgs a{};
  a.nodeptr = gv.rowptr; a.coll = gv.col; a.dst2src = g->dst2src; a.s_src = s_src; a.s_tgt = s_tgt; a.L = g->L > 0 ? g->L : 1; a.K = K;
  a.att = att; a.att_s = att_s; a.da = da; a.dz = dz; a.part = (float*)workspace; a.chunk_edges = pl.item_chunk_edges;
  if (pl.num_short > 0) {
    a.rows = pl.short_rows; a.num_rows = pl.num_short;
    const dim3 grid((unsigned)ceil_div(pl.num_short, 4));
    if (bwd) hipLaunchKernelGGL(rgat_row_softmax_wave_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(rgat_row_softmax_wave_kernel<false>, grid, dim3(256), 0, s, a);
  }
  if (pl.num_items > 0) {
    a.rows = pl.item_row; a.num_rows = pl.num_items; a.item_chunk = pl.item_chunk; a.item_slot = pl.item_slot;
    const dim3 grid((unsigned)pl.num_items);
    if (bwd) hipLaunchKernelGGL((rgat_row_softmax_item_kernel<true, 1>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rgat_row_softmax_item_kernel<false, 1>), grid, dim3(256), 0, s, a);
    if (pl.num_multi > 0) {
      if (bwd) {
        hipLaunchKernelGGL(rgat_row_softmax_combine_kernel<true>, dim3((unsigned)pl.num_multi), dim3(64), 0, s, pl.multi_base, pl.multi_n, K, a.part);
        hipLaunchKernelGGL((rgat_row_softmax_item_kernel<true, 2>), grid, dim3(256), 0, s, a);
      } else {
        hipLaunchKernelGGL(rgat_row_softmax_combine_kernel<false>, dim3((unsigned)pl.num_multi), dim3(64), 0, s, pl.multi_base, pl.multi_n, K, a.part);
        hipLaunchKernelGGL((rgat_row_softmax_item_kernel<false, 2>), grid, dim3(256), 0, s, a);
      }
    }
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" size_t tfgnn_rgat_attention_workspace_bytes(const tfgnn_graph* graph, int num_heads) {
  if (!graph || num_heads <= 0) return 0;
  return (size_t)graph->views[1].plan.num_partials * (size_t)num_heads * 2 * 4;
}

extern "C" int tfgnn_rgat_attention_forward(const tfgnn_graph* graph, const float* d_s_src, const float* d_s_tgt, int num_heads,
                                            float* d_att, float* d_att_by_src, void* d_workspace, size_t workspace_bytes, void* stream) {
  return rgat_row_softmax(graph, d_s_src, d_s_tgt, num_heads, d_att, d_att_by_src, nullptr, nullptr, false, d_workspace, workspace_bytes,
                          (hipStream_t)stream);
}

extern "C" int tfgnn_rgat_attention_backward(const tfgnn_graph* graph, const float* d_s_src, const float* d_s_tgt, const float* d_att,
                                             const float* d_da, int num_heads, float* d_dz, void* d_workspace, size_t workspace_bytes,
                                             void* stream) {
  return rgat_row_softmax(graph, d_s_src, d_s_tgt, num_heads, const_cast<float*>(d_att), nullptr, d_da, d_dz, true, d_workspace,
                          workspace_bytes, (hipStream_t)stream);
}

static int alpha_grad_blocks(int64_t V) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, tfgnn::ceil_div(V, 16))); }

extern "C" size_t tfgnn_rgat_alpha_grad_workspace_bytes(int64_t num_nodes, int num_edge_types, int hidden_dim) {
  if (num_nodes <= 0 || num_edge_types <= 0 || hidden_dim <= 0) return 0;
  return (size_t)alpha_grad_blocks(num_nodes) * 2 * (size_t)num_edge_types * (size_t)hidden_dim * 4;
}

extern "C" int tfgnn_rgat_alpha_grad(const float* d_ds_src, const float* d_ds_tgt, const float* d_Y, int64_t num_nodes,
                                     int num_edge_types, int num_heads, int hidden_dim, float* d_alpha_grad, void* d_workspace,
                                     size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0, "bad sizes");
  int rc = check_heads(num_heads, hidden_dim);
  if (rc) return rc;
  if (num_edge_types == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_alpha_grad, "NULL pointer");
  const int64_t LH = (int64_t)num_edge_types * hidden_dim;
  if (num_nodes == 0) {
    TFGNN_HIP_CHECK(hipMemsetAsync(d_alpha_grad, 0, (size_t)2 * LH * 4, (hipStream_t)stream));
    return TFGNN_OK;
  }
  TFGNN_REQUIRE(d_ds_src && d_ds_tgt && d_Y, "NULL pointer");
  const int nb = alpha_grad_blocks(num_nodes);
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= (size_t)nb * 2 * LH * 4, "workspace too small: need %zu bytes",
                (size_t)nb * 2 * LH * 4);
  const int64_t per = ceil_div(num_nodes, nb);
  if ((hidden_dim / num_heads) % 4 == 0 && ((uintptr_t)d_Y | (uintptr_t)d_workspace) % 16 == 0)
    hipLaunchKernelGGL(rgat_alpha_grad_partial_vec_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_ds_src, d_ds_tgt, d_Y,
                       num_nodes, num_edge_types, num_heads, hidden_dim / num_heads, per, (float*)d_workspace);
  else
    hipLaunchKernelGGL(rgat_alpha_grad_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_ds_src, d_ds_tgt, d_Y, num_nodes,
                       num_edge_types, num_heads, hidden_dim / num_heads, per, (float*)d_workspace);
  hipLaunchKernelGGL(rgat_alpha_grad_final_kernel, dim3((unsigned)ceil_div(2 * LH, 32)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)d_workspace, nb, num_edge_types, num_heads, hidden_dim / num_heads, d_alpha_grad);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_rgat_scores_backward_sp(const float* d_ds_src, const float* d_ds_tgt, const float* d_alpha, float* d_dY,
                                             int update_fp32, int64_t num_nodes, int num_edge_types, int num_heads, int hidden_dim,
                                             void* d_dY_sp, float* d_inv_scale, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_nodes >= 0 && num_edge_types >= 0, "bad sizes");
  int rc = check_heads(num_heads, hidden_dim);
  if (rc) return rc;
  const int64_t C = (int64_t)num_edge_types * hidden_dim;
  if ((hidden_dim / num_heads) % 4 != 0 || C > 2048 || C % 16 != 0) return TFGNN_ERR_UNSUPPORTED;
  if (num_nodes == 0 || C == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_ds_src && d_ds_tgt && d_alpha && d_dY && d_dY_sp && d_inv_scale, "NULL pointer");
  TFGNN_REQUIRE(((uintptr_t)d_dY | (uintptr_t)d_alpha) % 16 == 0 && (uintptr_t)d_dY_sp % 64 == 0, "unaligned operand");
  hipLaunchKernelGGL(rgat_scores_backward_sp_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(num_nodes, 4), 8192)), dim3(256), 0,
                     (hipStream_t)stream, d_ds_src, d_ds_tgt, d_alpha, d_dY, update_fp32, num_nodes, num_edge_types, num_heads,
                     hidden_dim / num_heads, (uint8_t*)d_dY_sp, d_inv_scale);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
