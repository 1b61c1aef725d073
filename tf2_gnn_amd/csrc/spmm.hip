// Gather + segment reduce over a CSR: the bandwidth-bound half of the message-passing hot path.
//
//   out[r, :] = post_act( row_scale[r] * REDUCE_{e in row r} pre_act( edge_weight[e] * in[col[e], :] ) )
//
// One call replaces, for one edge-bucketed view of the graph,
//   tf.nn.embedding_lookup(node_embeddings, edge_sources)          message_passing.py:197-199
//   the per-message 1/(c + 1e-7) scaling                             gnn_edge_mlp.py:102-106
//   tf.concat over edge types                                        message_passing.py:166-167
//   tf.math.unsorted_segment_{sum,max,mean,sqrt_n}                   utils/param_helpers.py:9-14
// without ever materialising the [E, D] gathered tensor (1.15 GB per layer at cfg-2).
//
// Work decomposition (wave64): a wave is split into 64/LPR groups of LPR lanes; a group owns one
// CSR row and a window of LPR*VPL float4 chunks of the feature dimension.  Lane j of the group
// owns chunks j, j+LPR, ... so that each load instruction of a group reads LPR*16 contiguous bytes
// of one source row (>= one 128 B line for LPR >= 8).  Edges are walked UNROLL at a time to keep
// UNROLL*VPL 16-byte loads in flight per lane.  Accumulation order inside a row is the CSR order
// (cols ascending) -> bit-reproducible, no atomics.
//
// blockIdx.y walks feature windows ("slices"): windows narrower than the row let the working set
// V * window_bytes of one pass sit in the 4 MiB L2 of an XCD.
#include "common.hpp"

namespace tfgnn {

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
};
template <>
struct VecT<1> {
  using type = float;
};

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vload(const float* p);
template <>
__device__ __forceinline__ float4 vload<4>(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
template <>
__device__ __forceinline__ float vload<1>(const float* p) {
  return *p;
}

template <int VEC>
__device__ __forceinline__ void vstore(float* p, const float* v);
template <>
__device__ __forceinline__ void vstore<4>(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void vstore<1>(float* p, const float* v) {
  *p = v[0];
}

template <int VEC>
__device__ __forceinline__ void vunpack(const typename VecT<VEC>::type& v, float* o);
template <>
__device__ __forceinline__ void vunpack<4>(const float4& v, float* o) {
  o[0] = v.x;
  o[1] = v.y;
  o[2] = v.z;
  o[3] = v.w;
}
template <>
__device__ __forceinline__ void vunpack<1>(const float& v, float* o) {
  o[0] = v;
}

struct GatherArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const float* ew;         // nullable
  const float* row_scale;  // nullable
  int64_t num_rows;
  const float* in;
  int64_t ld_in;
  int width;  // floats
  float* out;
  int64_t ld_out;
  int pre_act;
  int post_act;
};

template <int LPR, int VPL, int VEC, int UNROLL, bool IS_MAX, bool HAS_PRE>
__global__ void __launch_bounds__(256) csr_gather_reduce_kernel(GatherArgs a) {
  constexpr int GROUPS_PER_BLOCK = 256 / LPR;
  constexpr int WINDOW = LPR * VPL * VEC;  // floats covered per pass
  using V = typename VecT<VEC>::type;

  const int tid = threadIdx.x;
  const int group = tid / LPR;
  const int gl = tid % LPR;
  const int64_t row = (int64_t)blockIdx.x * GROUPS_PER_BLOCK + group;
  if (row >= a.num_rows) return;
  const int f0 = blockIdx.y * WINDOW + gl * VEC;  // first float of this lane's chunk 0

  const int32_t beg = a.rowptr[row];
  const int32_t end = a.rowptr[row + 1];

  float acc[VPL][VEC];
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[i][c] = IS_MAX ? kFloatLowest : 0.f;

  bool live[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) live[i] = (f0 + i * LPR * VEC) < a.width;

  for (int32_t e = beg; e < end; e += UNROLL) {
    int32_t idx[UNROLL];
    float w[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int32_t ee = e + u;
      ok[u] = ee < end;
      ee = ok[u] ? ee : end - 1;
      idx[u] = a.col[ee];
      w[u] = a.ew ? a.ew[ee] : 1.f;
    }
    V v[UNROLL][VPL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float* src = a.in + (int64_t)idx[u] * a.ld_in + f0;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (live[i]) v[u][i] = vload<VEC>(src + i * LPR * VEC);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (ok[u]) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          if (live[i]) {
            float x[VEC];
            vunpack<VEC>(v[u][i], x);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
              float m = w[u] * x[c];
              if (HAS_PRE) m = act_apply(a.pre_act, m);
              acc[i][c] = IS_MAX ? fmaxf(acc[i][c], m) : acc[i][c] + m;
            }
          }
        }
      }
    }
  }

  const float rs = a.row_scale ? a.row_scale[row] : 1.f;
  float* dst = a.out + row * a.ld_out + f0;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (live[i]) {
      float o[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        float y = acc[i][c];
        // an empty max-segment keeps the lowest float (tf.math.unsorted_segment_max); scaling it
        // would overflow to -inf, so the scale only touches real sums
        if (!IS_MAX || end > beg) y *= rs;
        o[c] = act_apply(a.post_act, y);
      }
      vstore<VEC>(dst + i * LPR * VEC, o);
    }
  }
}

template <int LPR, int VPL, int VEC, int UNROLL>
static int launch_variant(const GatherArgs& a, bool is_max, bool has_pre, hipStream_t s) {
  constexpr int GROUPS_PER_BLOCK = 256 / LPR;
  constexpr int WINDOW = LPR * VPL * VEC;
  dim3 grid((unsigned)ceil_div(a.num_rows, GROUPS_PER_BLOCK), (unsigned)ceil_div(a.width, WINDOW));
  dim3 block(256);
  if (!is_max && !has_pre)
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, false, false>), grid, block, 0, s, a);
  else if (!is_max && has_pre)
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, false, true>), grid, block, 0, s, a);
  else if (is_max && !has_pre)
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, true, false>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, true, true>), grid, block, 0, s, a);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

}  // namespace tfgnn

extern "C" int tfgnn_csr_gather_reduce(const int32_t* d_rowptr, const int32_t* d_col,
                                       const float* d_edge_weight, const float* d_row_scale,
                                       int64_t num_rows, const float* d_in, int64_t ld_in, int width,
                                       float* d_out, int64_t ld_out, int reduce_op, int pre_act,
                                       int post_act, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_rows >= 0 && width >= 0, "negative size");
  if (num_rows == 0 || width == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_rowptr && d_in && d_out, "NULL pointer");
  TFGNN_REQUIRE(ld_in >= width && ld_out >= width, "leading dimension smaller than width");
  TFGNN_REQUIRE(reduce_op == TFGNN_REDUCE_SUM || reduce_op == TFGNN_REDUCE_MAX, "unknown reduce op %d", reduce_op);
  TFGNN_REQUIRE(num_rows < ((int64_t)1 << 31), "too many rows");
  hipStream_t s = (hipStream_t)stream;
  GatherArgs a{d_rowptr, d_col,  d_edge_weight, d_row_scale, num_rows, d_in,
               ld_in,    width,  d_out,         ld_out,      pre_act,  post_act};
  const bool is_max = reduce_op == TFGNN_REDUCE_MAX;
  const bool has_pre = pre_act != TFGNN_ACT_NONE;
  const bool vec4 = (width % 4 == 0) && (ld_in % 4 == 0) && (ld_out % 4 == 0) &&
                    (((uintptr_t)d_in | (uintptr_t)d_out) % 16 == 0);
  if (!vec4) {
    // scalar path (odd widths: unit tests, tiny models): 16 lanes x 4 floats per pass
    return launch_variant<16, 4, 1, 2>(a, is_max, has_pre, s);
  }
  const int chunks = width / 4;
  if (chunks <= 8) return launch_variant<8, 1, 4, 8>(a, is_max, has_pre, s);
  if (chunks <= 16) return launch_variant<16, 1, 4, 8>(a, is_max, has_pre, s);
  if (chunks <= 32) return launch_variant<16, 2, 4, 4>(a, is_max, has_pre, s);
  if (chunks <= 64) return launch_variant<16, 4, 4, 2>(a, is_max, has_pre, s);
  if (chunks % 80 == 0 || chunks <= 80) return launch_variant<16, 5, 4, 2>(a, is_max, has_pre, s);
  return launch_variant<32, 4, 4, 2>(a, is_max, has_pre, s);
}
