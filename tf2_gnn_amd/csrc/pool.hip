// Node -> graph pooling kernels (tf2_gnn/layers/nodes_to_graph_representation.py:170-229).
//
// node_to_graph_map is sorted and contiguous per graph by construction of the batch
// (data/graph_dataset.py:211-217), which is also what tf.math.segment_sum / segment_mean
// (nodes_to_graph_representation.py:208,215,225) require.  The segments are therefore ranges:
//   tfgnn_segment_offsets      : ids [V] -> ptr [G+1]   (lower_bound per graph; flags unsorted ids)
//   tfgnn_segment_softmax      : per (graph, head) softmax of node scores, dpu_utils
//                                unsorted_segment_softmax semantics: exp(s - max) / (sum + 1e-7)
//   tfgnn_segment_weighted_sum : out[g, h, :] = sum_{v in g} w[v, h] * R[v, h, :]  (w NULL -> 1;
//                                mean divides by the node count)
#include <algorithm>

#include "common.hpp"

namespace tfgnn {

__global__ void segment_offsets_kernel(const int32_t* __restrict__ ids, int64_t V, int64_t G,
                                       int32_t* __restrict__ ptr, int32_t* __restrict__ err) {
  const int64_t n = V > G + 1 ? V : G + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i <= G) {  // ptr[g] = first position whose id >= g
      int64_t lo = 0, hi = V;
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < (int32_t)i) lo = mid + 1; else hi = mid;
      }
      ptr[i] = (int32_t)lo;
    }
    if (i < V) {
      const int32_t id = ids[i];
      if (id < 0 || id >= G) atomicOr(err, 1);
      if (i > 0 && ids[i - 1] > id) atomicOr(err, 2);
    }
  }
}

// one wave per graph; lanes over the graph's nodes; loop over heads
__global__ void __launch_bounds__(256)
segment_softmax_kernel(const float* __restrict__ scores, int64_t ld, int heads, const int32_t* __restrict__ ptr,
                       int64_t G, float* __restrict__ out, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const int32_t beg = ptr[g], end = ptr[g + 1];
  for (int h = 0; h < heads; ++h) {
    float m = kFloatLowest;
    for (int32_t v = beg + lane; v < end; v += 64) m = fmaxf(m, scores[(int64_t)v * ld + h]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    float s = 0.f;
    for (int32_t v = beg + lane; v < end; v += 64) s += expf(scores[(int64_t)v * ld + h] - m);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    const float inv = 1.f / (s + kSmallNumber);
    for (int32_t v = beg + lane; v < end; v += 64)
      out[(int64_t)v * ld_out + h] = expf(scores[(int64_t)v * ld + h] - m) * inv;
  }
}

// thread per (graph, feature); nodes of a graph are walked in order (deterministic)
__global__ void __launch_bounds__(256)
segment_weighted_sum_kernel(const float* __restrict__ R, const float* __restrict__ w, const int32_t* __restrict__ ptr,
                            int64_t G, int GD, int heads, int mean, float* __restrict__ out) {
  const int64_t total = G * GD;
  const int per_head = GD / heads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = i / GD;
    const int f = (int)(i - g * GD);
    const int h = f / per_head;
    const int32_t beg = ptr[g], end = ptr[g + 1];
    float s = 0.f;
    for (int32_t v = beg; v < end; ++v) {
      const float x = R[(int64_t)v * GD + f];
      s += w ? w[(int64_t)v * heads + h] * x : x;
    }
    if (mean) s /= (float)(end - beg > 0 ? end - beg : 1);
    out[i] = s;
  }
}

// backward of the weighted sum: dR[v,f] = w[v,h] * dOut[g,f] (w NULL -> 1, mean -> / count);
// dW[v,h] = sum_{f in head h} R[v,f] * dOut[g,f]   (one thread per (node, head))
__global__ void __launch_bounds__(256)
segment_weighted_sum_backward_kernel(const float* __restrict__ dOut, const float* __restrict__ R,
                                     const float* __restrict__ w, const int32_t* __restrict__ ids,
                                     const int32_t* __restrict__ ptr, int64_t V, int GD, int heads, int mean,
                                     float* __restrict__ dR, float* __restrict__ dW) {
  const int per_head = GD / heads;
  const int64_t total = V * heads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / heads;
    const int h = (int)(i - v * heads);
    const int32_t g = ids[v];
    float scale = w ? w[i] : 1.f;
    if (mean) {
      const int32_t cnt = ptr[g + 1] - ptr[g];
      scale /= (float)(cnt > 0 ? cnt : 1);
    }
    const float* dg = dOut + (int64_t)g * GD + h * per_head;
    const float* rr = R ? R + v * GD + h * per_head : nullptr;
    float* dr = dR + v * GD + h * per_head;
    float acc = 0.f;
    for (int j = 0; j < per_head; ++j) {
      const float d = dg[j];
      dr[j] = scale * d;
      if (rr) acc += rr[j] * d;
    }
    if (dW) dW[i] = acc;
  }
}

// softmax backward per (graph, head): ds = w * (dw - sum_{v in g} w * dw)
__global__ void __launch_bounds__(256)
segment_softmax_backward_kernel(const float* __restrict__ w, const float* __restrict__ dw, int heads,
                                const int32_t* __restrict__ ptr, int64_t G, float* __restrict__ ds) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= G) return;
  const int32_t beg = ptr[g], end = ptr[g + 1];
  for (int h = 0; h < heads; ++h) {
    float t = 0.f;
    for (int32_t v = beg + lane; v < end; v += 64) t += w[(int64_t)v * heads + h] * dw[(int64_t)v * heads + h];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d, 64);
    for (int32_t v = beg + lane; v < end; v += 64) {
      const int64_t i = (int64_t)v * heads + h;
      ds[i] = w[i] * (dw[i] - t);
    }
  }
}

static unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256), 16384)); }

}  // namespace tfgnn

// no allocation, no synchronisation: the caller owns the error word (zeroed by it), reads it when it wants to know
extern "C" int tfgnn_segment_offsets_async(const int32_t* d_ids, int64_t V, int64_t G, int32_t* d_ptr, int32_t* d_error_flag,
                                           void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && G >= 0, "negative size");
  TFGNN_REQUIRE(d_ptr && d_error_flag && (V == 0 || d_ids), "NULL pointer");
  TFGNN_REQUIRE(G < ((int64_t)1 << 31) - 1 && V < ((int64_t)1 << 31) - 1, "too large");
  hipLaunchKernelGGL(segment_offsets_kernel, dim3(blocks_for(std::max(V, G + 1))), dim3(256), 0, (hipStream_t)stream, d_ids, V, G,
                     d_ptr, d_error_flag);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_segment_offsets(const int32_t* d_ids, int64_t V, int64_t G, int32_t* d_ptr, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && G >= 0, "negative size");
  TFGNN_REQUIRE(d_ptr && (V == 0 || d_ids), "NULL pointer");
  TFGNN_REQUIRE(G < ((int64_t)1 << 31) - 1 && V < ((int64_t)1 << 31) - 1, "too large");
  hipStream_t s = (hipStream_t)stream;
  int32_t* d_err = nullptr;
  TFGNN_HIP_CHECK(hipMalloc((void**)&d_err, 4));
  TFGNN_HIP_CHECK(hipMemsetAsync(d_err, 0, 4, s));
  hipLaunchKernelGGL(segment_offsets_kernel, dim3(blocks_for(std::max(V, G + 1))), dim3(256), 0, s, d_ids, V, G, d_ptr, d_err);
  int32_t h_err = 0;
  TFGNN_HIP_CHECK(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, s));
  TFGNN_HIP_CHECK(hipStreamSynchronize(s));
  TFGNN_HIP_CHECK(hipFree(d_err));
  if (h_err & 1) {
    set_error("node_to_graph_map contains an id outside [0, %lld)", (long long)G);
    return TFGNN_ERR_OUT_OF_RANGE;
  }
  if (h_err & 2) {
    set_error("node_to_graph_map is not sorted (tf.math.segment_sum requires sorted segment ids)");
    return TFGNN_ERR_INVALID_ARGUMENT;
  }
  return TFGNN_OK;
}

extern "C" int tfgnn_segment_softmax(const float* d_scores, int64_t ld, int heads, const int32_t* d_ptr, int64_t G,
                                     float* d_out, int64_t ld_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(heads >= 0 && G >= 0, "negative size");
  if (G == 0 || heads == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_scores && d_ptr && d_out && ld >= heads && ld_out >= heads, "bad argument");
  hipLaunchKernelGGL(segment_softmax_kernel, dim3((unsigned)ceil_div(G, 4)), dim3(256), 0, (hipStream_t)stream,
                     d_scores, ld, heads, d_ptr, G, d_out, ld_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_segment_weighted_sum(const float* d_R, const float* d_w, const int32_t* d_ptr, int64_t G,
                                          int GD, int heads, int mean, float* d_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(G >= 0 && GD >= 0 && heads > 0 && GD % heads == 0, "bad sizes");
  if (G == 0 || GD == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_R && d_ptr && d_out, "NULL pointer");
  hipLaunchKernelGGL(segment_weighted_sum_kernel, dim3(blocks_for(G * GD)), dim3(256), 0, (hipStream_t)stream, d_R,
                     d_w, d_ptr, G, GD, heads, mean, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_segment_weighted_sum_backward(const float* d_dOut, const float* d_R, const float* d_w,
                                                   const int32_t* d_ids, const int32_t* d_ptr, int64_t V, int GD,
                                                   int heads, int mean, float* d_dR, float* d_dW, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && GD >= 0 && heads > 0 && GD % heads == 0, "bad sizes");
  if (V == 0 || GD == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dOut && d_ids && d_ptr && d_dR, "NULL pointer");
  hipLaunchKernelGGL(segment_weighted_sum_backward_kernel, dim3(blocks_for(V * heads)), dim3(256), 0,
                     (hipStream_t)stream, d_dOut, d_R, d_w, d_ids, d_ptr, V, GD, heads, mean, d_dR, d_dW);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_segment_softmax_backward(const float* d_w, const float* d_dw, int heads, const int32_t* d_ptr,
                                              int64_t G, float* d_ds, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(heads >= 0 && G >= 0, "negative size");
  if (G == 0 || heads == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_w && d_dw && d_ptr && d_ds, "NULL pointer");
  hipLaunchKernelGGL(segment_softmax_backward_kernel, dim3((unsigned)ceil_div(G, 4)), dim3(256), 0,
                     (hipStream_t)stream, d_w, d_dw, heads, d_ptr, G, d_ds);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
