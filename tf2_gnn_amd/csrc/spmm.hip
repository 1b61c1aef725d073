// Gather + segment reduce over a CSR: the bandwidth-bound half of the message-passing hot path.
//
//   out[r, :] = post_act( row_scale[r] * REDUCE_{e in row r} pre_act( w_e * in[col[e], :] ) )
//
// One call replaces, for one edge-bucketed view of the graph,
//   tf.nn.embedding_lookup(node_embeddings, edge_sources)          message_passing.py:197-199
//   the per-message 1/(c + 1e-7) scaling                             gnn_edge_mlp.py:102-106
//   tf.concat over edge types                                        message_passing.py:166-167
//   tf.math.unsorted_segment_{sum,max,mean,sqrt_n}                   utils/param_helpers.py:9-14
//   (RGAT) attention-weighted per-head unsorted_segment_sum          rgat.py:154-160
// without ever materialising the [E, D] gathered tensor (1.15 GB per layer at cfg-2).
//
// Work decomposition (wave64): a wave is split into 64/LPR groups of LPR lanes; a group owns one
// CSR row and a window of LPR*VPL float4 chunks of the feature dimension.  Lane j of the group
// owns chunks j, j+LPR, ... so that each load instruction of a group reads LPR*16 contiguous bytes
// of one source row (>= one 128 B line for LPR >= 8).  Edges are walked UNROLL at a time to keep
// UNROLL*VPL 16-byte loads in flight per lane.  Accumulation order inside a row is the CSR order
// (cols ascending) -> bit-reproducible, no atomics.
//
// Skew (R-MAT hubs): with a plan (graph views), rows longer than LONG_ROW_THRESHOLD are skipped by
// the row kernel and cut into items of ITEM_CHUNK edges; one workgroup per item, its groups each
// take a contiguous slice of the item, partial sums are combined through LDS in group order.  Rows
// made of several items go through a scratch buffer and a third, tiny combine kernel (item order).
// Everything stays deterministic.
//
// blockIdx.y walks feature windows.
#include <algorithm>
#include <cstdlib>

#include "aux_jobs.hpp"
#include "common.hpp"
#include "graph.hpp"
#include "sp16.hpp"

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

// MODE 0: sum, optional scalar edge weight          (RGCN / GGNN / RGIN forward and backward)
// MODE 1: sum, per-head edge weights ew[e, K]       (RGAT)
// MODE 2: general: runtime max / pre-activation, optional scalar edge weight
// MODE 3: MODE 1 + per edge and head the inner product of the gathered row with a row that belongs to the CSR row
//         (RGAT backward: d attention[e, k] = <Y[(src_e, l_e)], d_agg[tgt_e]>_k rides on the by-source gather that reads
//         d_agg[tgt_e] anyway - round 4; before, rgat_edge_dot read Y[(src_e, l_e)] per edge in a pass of its own)
enum { MODE_SUM = 0, MODE_HEADS = 1, MODE_GENERAL = 2, MODE_HEADS_DOT = 3 };
__host__ __device__ constexpr bool mode_heads(int mode) { return mode == MODE_HEADS || mode == MODE_HEADS_DOT; }

// value of the lane at distance 1 << S inside its aligned group of 2 << S lanes (S = 3, 2: the mirrored lane - same thing for a sum)
template <int S>
__device__ __forceinline__ float dpp_partner(float v) {
  constexpr int ctrl = S == 0 ? 0xB1 : S == 1 ? 0x4E : S == 2 ? 0x141 : 0x140;  // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false));
}
// N values per lane, each to be summed over the 1 << L2 consecutive lanes of the lane's group (N <= 1 << L2): instead of N
// butterflies the lanes split the values among themselves - at the top log2(N) distances a lane keeps the half of its values
// its lane bit selects and receives the partner's partial sums of that half - then one butterfly finishes the value left.
// -> the complete sum of value number `which` (lanes that differ only in the bits below the splitting ones hold copies)
template <int N, int L2>
__device__ __forceinline__ float dpp_split_sum(const float (&p)[N], int lane, int& which) {
  static_assert(N == 1 || N == 2 || N == 4, "values per lane");
  static_assert((1 << L2) >= N && L2 <= 4, "group");
  if constexpr (N == 4) {
    const bool hi = (lane >> (L2 - 1)) & 1;
    float q[2];
    q[0] = (hi ? p[2] : p[0]) + dpp_partner<L2 - 1>(hi ? p[0] : p[2]);
    q[1] = (hi ? p[3] : p[1]) + dpp_partner<L2 - 1>(hi ? p[1] : p[3]);
    int w;
    const float r = dpp_split_sum<2, L2 - 1>(q, lane, w);
    which = (hi ? 2 : 0) + w;
    return r;
  } else if constexpr (N == 2) {
    const bool hi = (lane >> (L2 - 1)) & 1;
    float q[1];
    q[0] = (hi ? p[1] : p[0]) + dpp_partner<L2 - 1>(hi ? p[0] : p[1]);
    int w;
    const float r = dpp_split_sum<1, L2 - 1>(q, lane, w);
    which = hi ? 1 : 0;
    return r;
  } else {
    float v = p[0];
    if constexpr (L2 >= 4) v += dpp_partner<3>(v);
    if constexpr (L2 >= 3) v += dpp_partner<2>(v);
    if constexpr (L2 >= 2) v += dpp_partner<1>(v);
    if constexpr (L2 >= 1) v += dpp_partner<0>(v);
    which = 0;
    return v;
  }
}

// sum over the 1 << log2n (<= 16) consecutive lanes a lane belongs to (DPP: no LDS traffic); every lane gets the sum
__device__ __forceinline__ float dpp_group_sum(float v, int log2n) {
  int x;
  if (log2n >= 1) { x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false); v += __int_as_float(x); }   // quad_perm [1,0,3,2]
  if (log2n >= 2) { x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false); v += __int_as_float(x); }   // quad_perm [2,3,0,1]
  if (log2n >= 3) { x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false); v += __int_as_float(x); }  // row_half_mirror
  if (log2n >= 4) { x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false); v += __int_as_float(x); }  // row_mirror
  return v;
}

struct GatherArgs {
  const int32_t* rowptr;
  const int32_t* col;
  const int32_t* short_rows;  // nullable: the rows the row workgroups handle, longest first (CsrPlan)
  int64_t num_short;
  const float* ew;         // nullable; [E] or [E, ew_heads]
  const float* row_scale;  // nullable
  int64_t num_rows;
  const float* in;
  int64_t ld_in;
  int width;  // floats
  float* out;
  int64_t ld_out;
  int pre_act;
  int post_act;
  int64_t num_src_rows;  // rows of `in` (0 = unknown): drives the L2-slicing heuristic
  int is_max;
  int ew_heads;    // K (MODE_HEADS)
  int head_width;  // floats per head (MODE_HEADS)
  // MODE_HEADS_DOT: dot_out[pos(e) * ew_heads + k] = sum over head k of in[col[e]] * dot_rows[CSR row of e]; pos = dot_pos[e]
  // (nullable: e).  head_width / 4 = 1 << dot_lph_log2 lanes hold a head (<= lanes per row, <= 16)
  const float* dot_rows;
  int64_t ld_dot;
  float* dot_out;
  const int32_t* dot_pos;
  int dot_lph_log2;
  int long_threshold;  // rows longer than this are left to the item kernels (0 = never)
  const int32_t* out_row_map;  // nullable: output row of CSR row r is out_row_map[r]; < 0 = no output
  unsigned xcd_units_pad;      // != 0: 1-D XCD-aware grid over (window, unit); set by the launcher
  unsigned total_units;
  // item pass
  const int32_t* item_row;
  const int32_t* item_chunk;
  const int32_t* item_slot;
  float* partial;  // [num_partials, width]
  int item_chunk_edges;
  // combine pass
  const int32_t* multi_row;
  const int32_t* multi_base;
  const int32_t* multi_n;
  int num_multi;
  // SP16 output (kernels instantiated with SP = true; MODE_SUM, float4 path, one feature window): the row sums are
  // written as the split fp16 operand of the f16x2 products (csrc/gemm_sp.hip) instead of fp32
  uint8_t* out_sp;         // [rows] x ld_out_sp bytes
  int64_t ld_out_sp;
  float* inv_out;          // [rows] 2^-e of every output row (not written when fixed_inv is given)
  const float* fixed_inv;  // nullable: one caller-chosen 2^-e for the whole tensor
  tfgnn_aux_job* combine_job;  // host pointer, nullable: receive the combine pass as a job instead of launching it
  int multi_code;  // host only: 10 R + U = the short rows go R to a lane group, U edges of each per round (gather_rows_block_multi: 21); 0: one row per group
};

// scale of one output row from the maximum over its LANES lanes (lanes of one group are contiguous)
template <int LANES>
__device__ __forceinline__ float sp_row_scale(const GatherArgs& a, float mx, int64_t orow, bool writer) {
  float iv, sc;
  if (a.fixed_inv) {
    iv = a.fixed_inv[0];
    sc = 1.f / iv;
  } else {
#pragma unroll
    for (int o = LANES / 2; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, LANES));
    sc = sp_scale_for_max(mx, &iv);
    if (writer) a.inv_out[orow] = iv;
  }
  return sc;
}

// accumulate edges [beg, end) into acc (lane-private chunks)
template <int LPR, int VPL, int VEC, int UNROLL, int MODE>
__device__ __forceinline__ void accumulate_edges(const GatherArgs& a, int32_t beg, int32_t end, int f0,
                                                 const bool (&live)[VPL], float (&acc)[VPL][VEC], int64_t row) {
  using V = typename VecT<VEC>::type;
  const bool is_max = MODE == MODE_GENERAL && a.is_max;
  int hidx[VPL];
  if (mode_heads(MODE)) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) hidx[i] = live[i] ? (f0 + i * LPR * VEC) / a.head_width : 0;
  }
  float yv[VPL][VEC];
  bool dot_writer = false;
  if constexpr (MODE == MODE_HEADS_DOT) {
    const float* yrow = a.dot_rows + row * a.ld_dot + f0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) yv[i][c] = 0.f;
      if (live[i]) vunpack<VEC>(vload<VEC>(yrow + i * LPR * VEC), yv[i]);
    }
    dot_writer = ((threadIdx.x % LPR) & ((1 << a.dot_lph_log2) - 1)) == 0;
  }
  // the split reduction (one store per edge instead of one per chunk) when the chunk count is 2 or 4 and a head has at
  // least that many lanes
  constexpr bool CAN_SPLIT = MODE == MODE_HEADS_DOT && (VPL == 2 || VPL == 4);
  const bool split = CAN_SPLIT && (1 << a.dot_lph_log2) >= VPL;
  bool split_writer = false;
  int split_head = 0;
  if (split) {  // which value the lane ends up with (dpp_split_sum), fixed per lane
    constexpr int SPLIT_BITS = VPL == 4 ? 2 : 1;
    const int lane = threadIdx.x % LPR;
    const int which = (lane >> (a.dot_lph_log2 - SPLIT_BITS)) & (VPL - 1);
    const int f = f0 + which * LPR * VEC;
    // copies sit in the lanes that differ in the bits below the splitting ones: the lane with those bits clear writes
    split_writer = (lane & ((1 << (a.dot_lph_log2 - SPLIT_BITS)) - 1)) == 0 && f < a.width;
    split_head = f / a.head_width;
  }
  for (int32_t e = beg; e < end; e += UNROLL) {
    int32_t idx[UNROLL];
    int32_t eid[UNROLL];
    int32_t dpos[UNROLL];
    float w[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      int32_t ee = e + u;
      ok[u] = ee < end;
      ee = ok[u] ? ee : end - 1;
      eid[u] = ee;
      idx[u] = a.col[ee];
      if constexpr (MODE == MODE_HEADS_DOT) dpos[u] = a.dot_pos ? a.dot_pos[ee] : ee;
      w[u] = (!mode_heads(MODE) && a.ew) ? a.ew[ee] : 1.f;
    }
    V v[UNROLL][VPL];
    float wh[UNROLL][VPL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float* src = a.in + (int64_t)idx[u] * a.ld_in + f0;
#pragma unroll
      for (int i = 0; i < VPL; ++i) {
        if (live[i]) {
          v[u][i] = vload<VEC>(src + i * LPR * VEC);
          if (mode_heads(MODE)) wh[u][i] = a.ew[(int64_t)eid[u] * a.ew_heads + hidx[i]];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (ok[u]) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          if (live[i]) {
            float x[VEC];
            vunpack<VEC>(v[u][i], x);
            const float wt = mode_heads(MODE) ? wh[u][i] : w[u];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
              float m = wt * x[c];
              if (MODE == MODE_GENERAL) {
                m = act_apply(a.pre_act, m);
                acc[i][c] = is_max ? fmaxf(acc[i][c], m) : acc[i][c] + m;
              } else {
                acc[i][c] += m;
              }
            }
          }
        }
      }
    }
    if constexpr (MODE == MODE_HEADS_DOT) {
      // all lanes of a row group walk the same edges, so the exchange is convergent inside the group (DPP moves data inside
      // a 16-lane row only; other groups of the wave may sit in another iteration, their lanes are not read)
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t pos = dpos[u];
        float p[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          p[i] = 0.f;
          if (live[i]) {
            float x[VEC];
            vunpack<VEC>(v[u][i], x);
#pragma unroll
            for (int c = 0; c < VEC; ++c) p[i] += x[c] * yv[i][c];
          }
        }
        if constexpr (CAN_SPLIT) {
          if (split) {
            const int lane = threadIdx.x % LPR;
            int which = 0;
            float r = 0.f;
            switch (a.dot_lph_log2) {  // (wave-uniform)
              case 1: if constexpr (VPL <= 2) r = dpp_split_sum<VPL, 1>(p, lane, which); break;
              case 2: r = dpp_split_sum<VPL, 2>(p, lane, which); break;
              case 3: r = dpp_split_sum<VPL, 3>(p, lane, which); break;
              default: r = dpp_split_sum<VPL, 4>(p, lane, which); break;
            }
            if (ok[u] && split_writer) a.dot_out[pos * a.ew_heads + split_head] = r;
            continue;
          }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
          const float r = dpp_group_sum(p[i], a.dot_lph_log2);
          if (ok[u] && live[i] && dot_writer) a.dot_out[pos * a.ew_heads + hidx[i]] = r;
        }
      }
    }
  }
}

template <int LPR, int VPL, int VEC, int UNROLL, int MODE, bool SP = false>
__device__ __forceinline__ void gather_rows_block(const GatherArgs& a, unsigned row_block, unsigned window) {
  constexpr int GROUPS_PER_BLOCK = 256 / LPR;
  constexpr int WINDOW = LPR * VPL * VEC;  // floats covered per pass
  const int tid = threadIdx.x;
  const int group = tid / LPR;
  const int gl = tid % LPR;
  const int64_t slot = (int64_t)row_block * GROUPS_PER_BLOCK + group;
  if (slot >= (a.short_rows ? a.num_short : a.num_rows)) return;
  const int64_t row = a.short_rows ? a.short_rows[slot] : slot;
  const int f0 = window * WINDOW + gl * VEC;  // first float of this lane's chunk 0
  const int32_t beg = a.rowptr[row];
  const int32_t end = a.rowptr[row + 1];
  if (a.long_threshold > 0 && end - beg > a.long_threshold) return;  // handled by the item kernels
  const int64_t orow = a.out_row_map ? a.out_row_map[row] : row;
  if (orow < 0) return;  // compact output: empty buckets have no row
  const bool is_max = MODE == MODE_GENERAL && a.is_max;

  float acc[VPL][VEC];
  bool live[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    live[i] = (f0 + i * LPR * VEC) < a.width;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[i][c] = is_max ? kFloatLowest : 0.f;
  }
  accumulate_edges<LPR, VPL, VEC, UNROLL, MODE>(a, beg, end, f0, live, acc, row);

  const float rs = a.row_scale ? a.row_scale[row] : 1.f;
  if constexpr (SP) {  // VEC == 4, one window: the lane group holds the whole row
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        acc[i][c] *= rs;
        if (live[i]) mx = fmaxf(mx, fabsf(acc[i][c]));
      }
    const float sc = sp_row_scale<LPR>(a, mx, orow, gl == 0);
    uint8_t* drow = a.out_sp + orow * a.ld_out_sp;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
      if (live[i]) sp_store4(drow, f0 + i * LPR * VEC, make_float4(acc[i][0], acc[i][VEC > 1 ? 1 : 0], acc[i][VEC > 2 ? 2 : 0], acc[i][VEC > 3 ? 3 : 0]), sc);
    return;
  }
  float* dst = a.out + orow * a.ld_out + f0;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if (live[i]) {
      float o[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        float y = acc[i][c];
        // an empty max-segment keeps the lowest float (tf.math.unsorted_segment_max); scaling it
        // would overflow to -inf, so the scale only touches real results
        if (!is_max || end > beg) y *= rs;
        o[c] = act_apply(a.post_act, y);
      }
      vstore<VEC>(dst + i * LPR * VEC, o);
    }
  }
}

// Short rows (round 6): a lane group takes R rows AT ONCE and walks them side by side, U edges of each per round.  A bucket of
// a molecule batch holds 0 - 4 edges: with one row per group the chain of dependent loads (row list -> row pointers -> columns
// -> source rows) is walked once per row with at most its own few source rows in flight at the end - the node-view gathers of
// the per-edge messages of configs[3] moved 2.3 GB in 0.83 ms (0.35 of the HBM peak) with every CU full of waiting waves.
// Here the R chains of a group advance together (R x U x VPL 16-byte loads in flight per lane) and loads past the end of a row
// are not issued (the one-row walk re-reads the row's last edge up to UNROLL - 1 times).  Every row still adds its edges in CSR
// order with the same arithmetic: results are bit-identical to gather_rows_block.  MODE_SUM, one feature window.
template <int LPR, int VPL, int VEC, bool SP, int R, int U>
__device__ __forceinline__ void gather_rows_block_multi(const GatherArgs& a, unsigned row_block) {
  using V = typename VecT<VEC>::type;
  constexpr int GROUPS_PER_BLOCK = 256 / LPR;
  const int tid = threadIdx.x;
  const int group = tid / LPR;
  const int gl = tid % LPR;
  const int64_t nslots = a.short_rows ? a.num_short : a.num_rows;
  const int64_t slot0 = (int64_t)row_block * (GROUPS_PER_BLOCK * R) + group;  // row r of the group: slot0 + r * GROUPS_PER_BLOCK
  const int f0 = gl * VEC;
  bool live[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) live[i] = (f0 + i * LPR * VEC) < a.width;
  int64_t row[R];
  bool on[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t slot = slot0 + (int64_t)r * GROUPS_PER_BLOCK;
    on[r] = slot < nslots;
    row[r] = on[r] ? (a.short_rows ? (int64_t)a.short_rows[slot] : slot) : 0;
  }
  int32_t beg[R], len[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    beg[r] = a.rowptr[row[r]];
    len[r] = a.rowptr[row[r] + 1] - beg[r];
  }
  int64_t orow[R];
  float rs[R];
  int32_t maxlen = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    orow[r] = a.out_row_map ? (int64_t)a.out_row_map[row[r]] : row[r];
    rs[r] = a.row_scale ? a.row_scale[row[r]] : 1.f;
    // (rows of the item kernels; compact output: empty buckets have no row)
    if ((a.long_threshold > 0 && len[r] > a.long_threshold) || orow[r] < 0) on[r] = false;
    if (!on[r]) len[r] = 0;
    maxlen = len[r] > maxlen ? len[r] : maxlen;
  }
  float acc[R][VPL][VEC];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc[r][i][c] = 0.f;

  for (int32_t j = 0; j < maxlen; j += U) {
    int32_t idx[R][U];
    float w[R][U];
    bool ok[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ok[r][u] = j + u < len[r];
        idx[r][u] = 0;
        w[r][u] = 1.f;
        if (ok[r][u]) {
          const int32_t e = beg[r] + j + u;
          idx[r][u] = a.col[e];
          if (a.ew) w[r][u] = a.ew[e];
        }
      }
    V v[R][U][VPL];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[r][u]) {
          const float* src = a.in + (int64_t)idx[r][u] * a.ld_in + f0;
#pragma unroll
          for (int i = 0; i < VPL; ++i)
            if (live[i]) v[r][u][i] = vload<VEC>(src + i * LPR * VEC);
        }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[r][u]) {
#pragma unroll
          for (int i = 0; i < VPL; ++i)
            if (live[i]) {
              float x[VEC];
              vunpack<VEC>(v[r][u][i], x);
#pragma unroll
              for (int c = 0; c < VEC; ++c) {
                float m = w[r][u] * x[c];
                acc[r][i][c] += m;
              }
            }
        }
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!on[r]) continue;
    if constexpr (SP) {
      float mx = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          acc[r][i][c] *= rs[r];
          if (live[i]) mx = fmaxf(mx, fabsf(acc[r][i][c]));
        }
      const float sc = sp_row_scale<LPR>(a, mx, orow[r], gl == 0);
      uint8_t* drow = a.out_sp + orow[r] * a.ld_out_sp;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (live[i])
          sp_store4(drow, f0 + i * LPR * VEC,
                    make_float4(acc[r][i][0], acc[r][i][VEC > 1 ? 1 : 0], acc[r][i][VEC > 2 ? 2 : 0], acc[r][i][VEC > 3 ? 3 : 0]), sc);
    } else {
      float* dst = a.out + orow[r] * a.ld_out + f0;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (live[i]) {
          float o[VEC];
#pragma unroll
          for (int c = 0; c < VEC; ++c) o[c] = act_apply(a.post_act, acc[r][i][c] * rs[r]);
          vstore<VEC>(dst + i * LPR * VEC, o);
        }
    }
  }
}

// one workgroup per item (a run of <= item_chunk_edges edges of a long row)
template <int LPR, int VPL, int VEC, int UNROLL, int MODE, bool SP = false>
__device__ __forceinline__ void gather_item_block(const GatherArgs& a, int item, unsigned window,
                                                  float (&red)[256 / LPR][LPR * VPL * VEC]) {
  constexpr int GROUPS = 256 / LPR;
  constexpr int WINDOW = LPR * VPL * VEC;
  const int tid = threadIdx.x;
  const int group = tid / LPR;
  const int gl = tid % LPR;
  const int64_t row = a.item_row[item];
  const int32_t rbeg = a.rowptr[row], rend = a.rowptr[row + 1];
  const int32_t ibeg = rbeg + a.item_chunk[item] * a.item_chunk_edges;
  const int32_t iend = min(rend, ibeg + a.item_chunk_edges);
  // the item's edges are dealt evenly to the lane groups (in pairs: the walk is unrolled by UNROLL), so a
  // 60-edge row keeps all 16 groups busy instead of the first five
  int per = (iend - ibeg + GROUPS - 1) / GROUPS;
  per = (per + UNROLL - 1) / UNROLL * UNROLL;
  const int32_t beg = min(iend, ibeg + group * per);
  const int32_t end = min(iend, beg + per);
  const int w0 = window * WINDOW;
  const int f0 = w0 + gl * VEC;
  const bool is_max = MODE == MODE_GENERAL && a.is_max;

  float acc[VPL][VEC];
  bool live[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    live[i] = (f0 + i * LPR * VEC) < a.width;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[i][c] = is_max ? kFloatLowest : 0.f;
  }
  accumulate_edges<LPR, VPL, VEC, UNROLL, MODE>(a, beg, end, f0, live, acc, row);
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int c = 0; c < VEC; ++c) red[group][(gl + i * LPR) * VEC + c] = acc[i][c];
  __syncthreads();
  const int32_t slot = a.item_slot[item];
  const float rs = a.row_scale ? a.row_scale[row] : 1.f;
  if constexpr (SP) {
    if (slot < 0) {  // the item is the whole row: sums -> red[0], row maximum, split store
      float mx = 0.f;
      for (int f = tid; f < WINDOW; f += 256) {
        if (w0 + f >= a.width) break;
        float v = red[0][f];
#pragma unroll
        for (int g = 1; g < GROUPS; ++g) v += red[g][f];
        v *= rs;
        red[0][f] = v;
        mx = fmaxf(mx, fabsf(v));
      }
#pragma unroll
      for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      __syncthreads();  // every sum over red[1..] has been taken
      if ((tid & 63) == 0) red[1][tid >> 6] = mx;
      __syncthreads();
      mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
      const int64_t orow = a.out_row_map ? a.out_row_map[row] : row;
      float iv, sc;
      if (a.fixed_inv) {
        iv = a.fixed_inv[0];
        sc = 1.f / iv;
      } else {
        sc = sp_scale_for_max(mx, &iv);
        if (tid == 0) a.inv_out[orow] = iv;
      }
      uint8_t* drow = a.out_sp + orow * a.ld_out_sp;
      for (int c4 = tid; c4 * 4 < WINDOW; c4 += 256)
        if (w0 + c4 * 4 < a.width)
          sp_store4(drow, w0 + c4 * 4, make_float4(red[0][c4 * 4], red[0][c4 * 4 + 1], red[0][c4 * 4 + 2], red[0][c4 * 4 + 3]), sc);
      return;
    }
  }
  for (int f = tid; f < WINDOW; f += 256) {
    if (w0 + f >= a.width) break;
    float s = red[0][f];
#pragma unroll
    for (int g = 1; g < GROUPS; ++g) s = is_max ? fmaxf(s, red[g][f]) : s + red[g][f];
    if (slot < 0) {
      const int64_t orow = a.out_row_map ? a.out_row_map[row] : row;
      a.out[orow * a.ld_out + w0 + f] = act_apply(a.post_act, s * rs);
    } else {
      a.partial[(int64_t)slot * a.width + w0 + f] = s;
    }
  }
}

// rows made of several items: combine their partial results in item order
__global__ void __launch_bounds__(256) csr_gather_combine_kernel(GatherArgs a) {
  const int64_t total = (int64_t)a.num_multi * a.width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / a.width);
    const int f = (int)(i - (int64_t)m * a.width);
    const int64_t row = a.multi_row[m];
    const int32_t base = a.multi_base[m], n = a.multi_n[m];
    float s = a.partial[(int64_t)base * a.width + f];
    for (int k = 1; k < n; ++k) {
      const float p = a.partial[(int64_t)(base + k) * a.width + f];
      s = a.is_max ? fmaxf(s, p) : s + p;
    }
    const float rs = a.row_scale ? a.row_scale[row] : 1.f;
    const int64_t orow = a.out_row_map ? a.out_row_map[row] : row;
    a.out[orow * a.ld_out + f] = act_apply(a.post_act, s * rs);
  }
}

// SP16 output: one wave per multi-item row (width <= 2048 floats); the body is shared with the merged small-pass launch
static AuxCombineSp combine_sp_args(const GatherArgs& a) {
  AuxCombineSp c{};
  c.multi_row = a.multi_row; c.multi_base = a.multi_base; c.multi_n = a.multi_n; c.num_multi = a.num_multi;
  c.row_scale = a.row_scale; c.partial = a.partial; c.width = a.width; c.out_row_map = a.out_row_map;
  c.out_sp = a.out_sp; c.ld_out_sp = a.ld_out_sp; c.inv_out = a.inv_out; c.fixed_inv = a.fixed_inv;
  return c;
}
__global__ void __launch_bounds__(256) csr_gather_combine_sp_kernel(AuxCombineSp a) { combine_sp_body(a, blockIdx.x); }

// One launch covers both kinds of work: workgroups [0, num_items) each take one item of a long row
// (longest work first), the remaining workgroups take 256/LPR short rows each.
template <int LPR, int VPL, int VEC, int UNROLL, int MODE, bool SP = false>
__global__ void __launch_bounds__(256) csr_gather_reduce_kernel(GatherArgs a, int num_items) {
  __shared__ float red[256 / LPR][LPR * VPL * VEC];
  unsigned unit, window;
  if (a.xcd_units_pad) {
    // XCD-aware order (workgroup b runs on XCD b % 8): every XCD walks the feature windows one after the
    // other over its share of the work units, so the source slice in[:, window] (V * window bytes)
    // it is gathering from stays resident in that XCD's 4 MiB L2
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned per = a.xcd_units_pad >> 3;
    window = j / per;
    unit = (j % per) * 8u + xcd;
    if (unit >= a.total_units) return;
  } else {
    unit = blockIdx.x;
    window = blockIdx.y;
  }
  if ((int)unit < num_items)
    gather_item_block<LPR, VPL, VEC, UNROLL, MODE, SP>(a, (int)unit, window, red);
  else
    gather_rows_block<LPR, VPL, VEC, UNROLL, MODE, SP>(a, unit - (unsigned)num_items, window);
}

// the same launch with the short rows R to a lane group (one feature window, plain sums)
template <int LPR, int VPL, int VEC, int UNROLL, bool SP, int R, int U>
__global__ void __launch_bounds__(256) csr_gather_reduce_multi_kernel(GatherArgs a, int num_items) {
  __shared__ float red[256 / LPR][LPR * VPL * VEC];
  const unsigned unit = blockIdx.x;
  if ((int)unit < num_items)
    gather_item_block<LPR, VPL, VEC, UNROLL, MODE_SUM, SP>(a, (int)unit, 0, red);
  else
    gather_rows_block_multi<LPR, VPL, VEC, SP, R, U>(a, unit - (unsigned)num_items);
}

template <int LPR, int VPL, int VEC, int UNROLL, bool SP, int R, int U>
static void launch_multi(GatherArgs a, int num_items, hipStream_t s) {
  const unsigned units = (unsigned)(num_items + ceil_div(a.short_rows ? a.num_short : a.num_rows, (256 / LPR) * R));
  a.total_units = units;
  hipLaunchKernelGGL((csr_gather_reduce_multi_kernel<LPR, VPL, VEC, UNROLL, SP, R, U>), dim3(units, 1), dim3(256), 0, s, a, num_items);
}

template <int LPR, int VPL, int VEC, int UNROLL, int MODE, bool SP = false>
static int launch_mode(GatherArgs a, int num_items, hipStream_t s) {
  constexpr int GROUPS_PER_BLOCK = 256 / LPR;
  constexpr int WINDOW = LPR * VPL * VEC;
  const unsigned windows = (unsigned)ceil_div(a.width, WINDOW);
  const unsigned units = (unsigned)(num_items + ceil_div(a.short_rows ? a.num_short : a.num_rows, GROUPS_PER_BLOCK));
  dim3 block(256);
  a.total_units = units;
  bool multi_done = false;
  if constexpr (MODE == MODE_SUM && VEC == 4 && LPR * VPL <= 128) {
    if (a.multi_code > 10 && windows == 1 && (!SP || a.width <= 2048)) {
      a.xcd_units_pad = 0;
      multi_done = true;
      switch (a.multi_code) {
        case 21: launch_multi<LPR, VPL, VEC, UNROLL, SP, 2, 1>(a, num_items, s); break;
        default: multi_done = false; break;
      }
    }
  }
  if constexpr (SP) {
    if (windows != 1 || a.width > 2048) {
      set_error("SP16 gather output needs the whole row in one feature window (width %d)", a.width);
      return TFGNN_ERR_UNSUPPORTED;
    }
    a.xcd_units_pad = 0;
    if (!multi_done)
      hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, MODE, true>), dim3(units, 1), block, 0, s, a, num_items);
    TFGNN_LAUNCH_CHECK();
    if (a.num_multi > 0) {
      if (a.combine_job) {  // the caller launches the combine pass itself, together with other small passes
        aux_job_set(a.combine_job, AUX_COMBINE_SP, (unsigned)ceil_div(a.num_multi, 4), combine_sp_args(a));
      } else {
        hipLaunchKernelGGL(csr_gather_combine_sp_kernel, dim3((unsigned)ceil_div(a.num_multi, 4)), block, 0, s, combine_sp_args(a));
        TFGNN_LAUNCH_CHECK();
      }
    }
    return TFGNN_OK;
  }
  if (multi_done) {
  } else if (a.xcd_units_pad && windows > 1) {
    a.xcd_units_pad = (units + 7u) & ~7u;
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, MODE>), dim3(a.xcd_units_pad * windows), block, 0, s,
                       a, num_items);
  } else {
    a.xcd_units_pad = 0;
    hipLaunchKernelGGL((csr_gather_reduce_kernel<LPR, VPL, VEC, UNROLL, MODE>), dim3(units, windows), block, 0, s, a,
                       num_items);
  }
  TFGNN_LAUNCH_CHECK();
  if (a.num_multi > 0) {
    const int64_t total = (int64_t)a.num_multi * a.width;
    hipLaunchKernelGGL(csr_gather_combine_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096)), block, 0, s, a);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}

template <int LPR, int VPL, int VEC, int UNROLL>
static int launch_variant(const GatherArgs& a, int mode, int num_items, hipStream_t s) {
  if (mode == MODE_SUM) return launch_mode<LPR, VPL, VEC, UNROLL, MODE_SUM>(a, num_items, s);
  if (mode == MODE_HEADS) return launch_mode<LPR, VPL, VEC, UNROLL, MODE_HEADS>(a, num_items, s);
  if (mode == MODE_HEADS_DOT) {
    if constexpr (VEC == 4) {
      TFGNN_REQUIRE((1 << a.dot_lph_log2) <= LPR, "fused edge products: a head is wider than a lane group (width %d)", a.width);
      return launch_mode<LPR, VPL, VEC, UNROLL, MODE_HEADS_DOT>(a, num_items, s);
    } else {
      set_error("fused edge products need 16-byte aligned rows");
      return TFGNN_ERR_UNSUPPORTED;
    }
  }
  return launch_mode<LPR, VPL, VEC, UNROLL, MODE_GENERAL>(a, num_items, s);
}

// average edges per walked row up to which the short rows go two to a lane group
constexpr double kMultiAvgLen = 1.5;

static int gather_dispatch(GatherArgs a, int num_items, hipStream_t s) {
  int mode = MODE_SUM;
  if (a.is_max || a.pre_act != TFGNN_ACT_NONE) mode = MODE_GENERAL;
  if (a.ew && a.ew_heads > 1) {
    TFGNN_REQUIRE(mode == MODE_SUM, "per-head edge weights only support plain sums");
    TFGNN_REQUIRE(a.head_width > 0 && a.width % a.head_width == 0 && a.width / a.head_width == a.ew_heads,
                  "width %d is not ew_heads %d x head_width %d", a.width, a.ew_heads, a.head_width);
    mode = MODE_HEADS;
    if (a.dot_out) {
      TFGNN_REQUIRE(a.dot_rows && a.ld_dot >= a.width && a.ld_dot % 4 == 0 && (uintptr_t)a.dot_rows % 16 == 0,
                    "fused edge products: bad row operand");
      mode = MODE_HEADS_DOT;
    }
  }
  TFGNN_REQUIRE(!a.dot_out || mode == MODE_HEADS_DOT, "fused edge products ride on the per-head weighted sums only");
  const bool vec4 = (a.width % 4 == 0) && (a.ld_in % 4 == 0) && (a.ld_out % 4 == 0) &&
                    (((uintptr_t)a.in | (uintptr_t)a.out) % 16 == 0) && (!a.out_sp || a.ld_out_sp % 64 == 0) &&
                    (!mode_heads(mode) || a.head_width % 4 == 0);
  TFGNN_REQUIRE(vec4 || !a.out_sp, "SP16 gather output needs 16-byte aligned rows");
  if (!vec4) {
    // scalar path (odd widths: unit tests, tiny models): 16 lanes x 4 floats per pass
    return launch_variant<16, 4, 1, 2>(a, mode, num_items, s);
  }
  const int chunks = a.width / 4;
  if (a.out_sp) {  // SP16 output: plain sums on the float4 path, whole rows per lane group
    TFGNN_REQUIRE(mode == MODE_SUM && a.width % 16 == 0, "SP16 gather output: plain sums of rows with width %% 16 == 0 only");
    if (chunks <= 8) return launch_mode<8, 1, 4, 8, MODE_SUM, true>(a, num_items, s);
    if (chunks <= 16) return launch_mode<16, 1, 4, 8, MODE_SUM, true>(a, num_items, s);
    if (chunks <= 32) return launch_mode<16, 2, 4, 4, MODE_SUM, true>(a, num_items, s);
    if (chunks <= 64) return launch_mode<16, 4, 4, 2, MODE_SUM, true>(a, num_items, s);
    if (chunks <= 80) return launch_mode<16, 5, 4, 2, MODE_SUM, true>(a, num_items, s);  // UNROLL 1 / 4 and 32 lanes per row measured slower (tools/gather_l2_probe.py)
    return launch_mode<32, 4, 4, 2, MODE_SUM, true>(a, num_items, s);
  }
  // L2-resident slicing: gather 32-float (128 B) windows of the rows, XCD by XCD, when one window of
  // ALL source rows fits an XCD's 4 MiB L2 but the full rows do not (cfg-2: 3.84 MB vs 38 MB).
  static const int sliced_knob = [] { const char* e = getenv("TFGNN_GATHER_SLICED"); return e ? atoi(e) : -1; }();
  const bool can_slice = a.num_src_rows > 0 && chunks > 16 && !mode_heads(mode);
  // Measured at cfg-2 (tools/gather_probe.py): 322 us sliced vs 160 us whole-row - ten passes over the
  // index arrays with 128-byte requests lose more than the L2 hits win.  Off unless TFGNN_GATHER_SLICED=1.
  const bool want_slice = sliced_knob > 0;
  if (can_slice && want_slice) {
    a.xcd_units_pad = 1;
    return launch_variant<8, 1, 4, 8>(a, mode, num_items, s);
  }
  // (rows of 8 floats - RGAT's per-head logit gradients summed per bucket - leave six of the eight lanes of the next variant
  // idle; a two-lane variant measured SLOWER, 28 vs 23 us at configs[2]: the launch is bound by the walk along the rows)
  if (chunks <= 8) return launch_variant<8, 1, 4, 8>(a, mode, num_items, s);
  if (chunks <= 16) return launch_variant<16, 1, 4, 8>(a, mode, num_items, s);
  if (chunks <= 32) return launch_variant<16, 2, 4, 4>(a, mode, num_items, s);
  if (chunks <= 64) return launch_variant<16, 4, 4, 2>(a, mode, num_items, s);
  static const int unroll_knob = [] { const char* e = getenv("TFGNN_GATHER_UNROLL"); return e ? atoi(e) : 0; }();
  if (chunks % 80 == 0 || chunks <= 80) {
    if (unroll_knob == 4) return launch_variant<16, 5, 4, 4>(a, mode, num_items, s);
    if (unroll_knob == 1) return launch_variant<16, 5, 4, 1>(a, mode, num_items, s);
    return launch_variant<16, 5, 4, 2>(a, mode, num_items, s);
  }
  return launch_variant<32, 4, 4, 2>(a, mode, num_items, s);
}

static int check_common(int64_t num_rows, const void* rowptr, const void* in, const void* out, int64_t ld_in,
                        int64_t ld_out, int width, int reduce_op) {
  TFGNN_REQUIRE(rowptr && in && out, "NULL pointer");
  TFGNN_REQUIRE(ld_in >= width && ld_out >= width, "leading dimension smaller than width");
  TFGNN_REQUIRE(reduce_op == TFGNN_REDUCE_SUM || reduce_op == TFGNN_REDUCE_MAX, "unknown reduce op %d", reduce_op);
  TFGNN_REQUIRE(num_rows < ((int64_t)1 << 31), "too many rows");
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
  int rc = check_common(num_rows, d_rowptr, d_in, d_out, ld_in, ld_out, width, reduce_op);
  if (rc) return rc;
  GatherArgs a{};
  a.rowptr = d_rowptr; a.col = d_col; a.ew = d_edge_weight; a.row_scale = d_row_scale;
  a.num_rows = num_rows; a.in = d_in; a.ld_in = ld_in; a.width = width; a.out = d_out; a.ld_out = ld_out;
  a.pre_act = pre_act; a.post_act = post_act; a.is_max = reduce_op == TFGNN_REDUCE_MAX;
  a.ew_heads = 1; a.head_width = width;
  return gather_dispatch(a, 0, (hipStream_t)stream);
}

extern "C" size_t tfgnn_graph_gather_workspace_bytes(const tfgnn_graph* g, int view, int width) {
  if (!g || view < 0 || view > 7 || width <= 0) return 0;
  if (view >= 4) view = (view == 4 || view == 6) ? 0 : 2;
  return (size_t)g->views[view].plan.num_partials * (size_t)width * 4;
}

extern "C" int tfgnn_graph_gather_dot_supported(int width, int ew_heads);
static int graph_gather_impl(const tfgnn_graph* g, int view, const int32_t* d_col_override,
                             const float* d_edge_weight, int ew_heads, const float* d_row_scale,
                             const float* d_in, int64_t ld_in, int width, float* d_out,
                             int64_t ld_out, int reduce_op, int pre_act, int post_act,
                             void* d_workspace, size_t workspace_bytes, void* stream, void* d_out_sp, int64_t ld_out_sp,
                             float* d_inv_scale, const float* d_fixed_inv, tfgnn_aux_job* combine_job = nullptr,
                             const float* d_dot_rows = nullptr, int64_t ld_dot = 0, const int32_t* d_dot_pos = nullptr,
                             float* d_dot_out = nullptr) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  TFGNN_REQUIRE(view >= 0 && view <= 6, "unknown graph view %d", view);
  TFGNN_REQUIRE(width >= 0 && ew_heads >= 1, "bad sizes");
  // views 4 / 5: the typed views 0 / 2 with compact output (one row per NON-EMPTY bucket, type-major)
  {
    unsigned need = (view == 1 || view == 3) ? TFGNN_GRAPH_PART_PLAN_NODE : TFGNN_GRAPH_PART_PLAN_TYPED;
    if (view == 4 || view == 5) need |= TFGNN_GRAPH_PART_COMPACT;
    if (view == 6) need |= TFGNN_GRAPH_PART_DST_PATTERN;
    const int prc = graph_require_parts(g, need, "tfgnn_graph_gather_reduce");
    if (prc) return prc;
  }
  const int32_t* out_map = nullptr;
  int64_t compact_nz = -1;
  if (view == 6) {  // by-target buckets, rows in pattern order
    TFGNN_REQUIRE(g->L <= 8, "TFGNN_VIEW_BY_DST_TYPED_PATTERN: at most 8 edge types");
    out_map = g->pat_rowmap_d;
    view = 0;
  } else if (view >= 4) {
    out_map = g->compact[view - 4].cpos;
    compact_nz = g->compact[view - 4].num_nz;
    view = view == 4 ? 0 : 2;
  }
  const GraphView& gv = g->views[view];
  if (gv.num_rows == 0 || width == 0) return TFGNN_OK;
  int rc = check_common(gv.num_rows, gv.rowptr, d_in, d_out_sp ? (const void*)d_out_sp : (const void*)d_out, ld_in,
                        d_out_sp ? (int64_t)width : ld_out, width, reduce_op);
  if (rc) return rc;
  const CsrPlan& p = gv.plan;
  TFGNN_REQUIRE(p.num_partials == 0 || (d_workspace && workspace_bytes >= (size_t)p.num_partials * width * 4),
                "workspace too small: need %zu bytes", (size_t)p.num_partials * width * 4);
  GatherArgs a{};
  a.rowptr = gv.rowptr; a.col = d_col_override ? d_col_override : gv.col; a.ew = d_edge_weight;
  a.row_scale = d_row_scale; a.num_rows = gv.num_rows; a.in = d_in; a.ld_in = ld_in; a.width = width;
  a.out = d_out; a.ld_out = d_out_sp ? (int64_t)width : ld_out; a.pre_act = pre_act; a.post_act = post_act;
  a.out_sp = (uint8_t*)d_out_sp; a.ld_out_sp = ld_out_sp; a.inv_out = d_inv_scale; a.fixed_inv = d_fixed_inv;
  a.combine_job = combine_job;
  if (combine_job) combine_job->kind = 0, combine_job->num_blocks = 0;
  a.is_max = reduce_op == TFGNN_REDUCE_MAX; a.ew_heads = ew_heads; a.head_width = width / ew_heads;
  a.long_threshold = p.long_threshold;
  a.out_row_map = out_map;
  if (d_dot_out) {
    TFGNN_REQUIRE(ew_heads > 1 && d_edge_weight && width % ew_heads == 0, "tfgnn_graph_gather_reduce_dot: per-head edge weights only");
    const int lph = width / ew_heads / 4;
    TFGNN_REQUIRE(tfgnn_graph_gather_dot_supported(width, ew_heads), "tfgnn_graph_gather_reduce_dot: unsupported head width %d",
                  width / ew_heads);
    a.dot_rows = d_dot_rows; a.ld_dot = ld_dot; a.dot_out = d_dot_out; a.dot_pos = d_dot_pos;
    a.dot_lph_log2 = 31 - __builtin_clz((unsigned)lph);
  }
  a.num_src_rows = d_col_override ? 0 : ((view == 1 || view == 3) ? g->R : g->V);  // rows of `in`
  a.item_row = p.item_row; a.item_chunk = p.item_chunk; a.item_slot = p.item_slot;
  a.partial = (float*)d_workspace; a.item_chunk_edges = p.item_chunk_edges;
  a.multi_row = p.multi_row; a.multi_base = p.multi_base; a.multi_n = p.multi_n; a.num_multi = p.num_multi;
  // bit 0: typed views, bit 1: node views keep the natural row order (probing)
  static const int natural_order = [] { const char* e = getenv("TFGNN_GATHER_NATURAL_ORDER"); return e ? atoi(e) : 0; }();
  if (!(natural_order & ((view & 1) ? 2 : 1))) {
    a.short_rows = p.short_rows;
    a.num_short = p.num_short;
    // compact output: the empty buckets have no row, and they close the length-ordered list - no lane groups for them
    // (configs[4]: 452 517 of 6.8 M by-source buckets are non-empty; walking all of them was 0.85 M workgroups with nothing to do)
    if (compact_nz >= 0) a.num_short = std::max<int64_t>(0, (int64_t)p.num_short - (gv.num_rows - compact_nz));
  }
  {
    // short rows side by side (gather_rows_block_multi) when the rows the row workgroups walk hold <= 1.5 edges on average -
    // the (node, type) buckets of a molecule batch (0.6): tools/gather_short_probe.py, 128k molecules, H = 128: typed by-target
    // view 1109 -> 948 us, typed by-source view with SP16 output 1257 -> 1132 us with two rows per group and one edge of each
    // per round; four rows, or two edges per round: 1070 - 1277 / 1156 - 1306 us; a persistent grid (4 - 16 workgroups per CU
    // walking the units): no better.  The node views (3 edges per row) run at 5.0 - 5.2 TB/s either way and keep one row per
    // group.  TFGNN_GATHER_MULTI = 1: off; 21: every plain-sum launch (tests)
    const char* multi_env = getenv("TFGNN_GATHER_MULTI");  // (read per call: the tests compare the two shapes in one process)
    const int multi_knob = multi_env ? atoi(multi_env) : 0;
    const int64_t nslots = a.short_rows ? a.num_short : a.num_rows;
    a.multi_code = 0;
    if (multi_knob > 10) a.multi_code = multi_knob;
    else if (multi_knob == 0 && nslots > 0 && !a.is_max && ew_heads == 1 && (double)g->E <= kMultiAvgLen * (double)nslots) a.multi_code = 21;
  }
  count_launch(d_out_sp ? TFGNN_KFAM_GATHER_SP : TFGNN_KFAM_GATHER);
  return gather_dispatch(a, p.num_items, (hipStream_t)stream);
}

extern "C" int tfgnn_graph_gather_reduce(const tfgnn_graph* g, int view, const int32_t* d_col_override,
                                         const float* d_edge_weight, int ew_heads, const float* d_row_scale,
                                         const float* d_in, int64_t ld_in, int width, float* d_out,
                                         int64_t ld_out, int reduce_op, int pre_act, int post_act,
                                         void* d_workspace, size_t workspace_bytes, void* stream) {
  return graph_gather_impl(g, view, d_col_override, d_edge_weight, ew_heads, d_row_scale, d_in, ld_in, width, d_out, ld_out,
                           reduce_op, pre_act, post_act, d_workspace, workspace_bytes, stream, nullptr, 0, nullptr, nullptr);
}

/* can tfgnn_graph_gather_reduce_dot take rows of `width` floats in `ew_heads` heads?  (a head = 4, 8, .. 64 floats that a
 * lane group of the gather holds in 1..16 consecutive lanes) */
extern "C" int tfgnn_graph_gather_dot_supported(int width, int ew_heads) {
  if (width <= 0 || ew_heads <= 1 || width % ew_heads) return 0;
  const int hw = width / ew_heads;
  if (hw % 4 || width % 4) return 0;
  const int lph = hw / 4, chunks = width / 4;
  if (lph & (lph - 1)) return 0;
  const int lanes_per_row = chunks <= 8 ? 8 : (chunks <= 80 ? 16 : 32);  // gather_dispatch's choice
  return lph <= 16 && lph <= lanes_per_row;
}

/* tfgnn_graph_gather_reduce with per-head edge weights that ALSO writes, per edge e of the view and head k,
 *   d_dot_out[pos(e) * ew_heads + k] = sum_{f in head k} d_in[col(e), f] * d_dot_rows[r(e) * ld_dot + f]
 * where r(e) is the row of the view the edge belongs to and pos(e) = d_dot_pos[e] (NULL: e, the edge's position in the view's
 * order).  RGAT backward: the by-source gather of d_agg[tgt] with d_dot_rows = Y yields d attention in by-target order through
 * d_dot_pos = TFGNN_G_SRC2DST_POS (replaces tfgnn_rgat_edge_dot, one pass over the edges less). */
extern "C" int tfgnn_graph_gather_reduce_dot(const tfgnn_graph* g, int view, const float* d_edge_weight, int ew_heads,
                                             const float* d_in, int64_t ld_in, int width, float* d_out, int64_t ld_out,
                                             const float* d_dot_rows, int64_t ld_dot, const int32_t* d_dot_pos, float* d_dot_out,
                                             void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(d_dot_rows && d_dot_out, "tfgnn_graph_gather_reduce_dot: NULL pointer");
  return graph_gather_impl(g, view, nullptr, d_edge_weight, ew_heads, nullptr, d_in, ld_in, width, d_out, ld_out, TFGNN_REDUCE_SUM,
                           TFGNN_ACT_NONE, TFGNN_ACT_NONE, d_workspace, workspace_bytes, stream, nullptr, 0, nullptr, nullptr, nullptr,
                           d_dot_rows, ld_dot, d_dot_pos, d_dot_out);
}

extern "C" int tfgnn_graph_gather_reduce_sp(const tfgnn_graph* g, int view, const int32_t* d_col_override,
                                            const float* d_edge_weight, const float* d_row_scale, const float* d_in,
                                            int64_t ld_in, int width, void* d_out_sp, int64_t ld_out_sp_bytes,
                                            float* d_inv_scale, const float* d_fixed_inv_scale, void* d_workspace,
                                            size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(d_out_sp && (d_inv_scale || d_fixed_inv_scale), "tfgnn_graph_gather_reduce_sp: NULL output");
  TFGNN_REQUIRE(ld_out_sp_bytes >= (int64_t)width * 4 && ld_out_sp_bytes % 64 == 0 && (uintptr_t)d_out_sp % 64 == 0,
                "tfgnn_graph_gather_reduce_sp: bad SP16 leading dimension / alignment");
  return graph_gather_impl(g, view, d_col_override, d_edge_weight, 1, d_row_scale, d_in, ld_in, width, nullptr, 0,
                           TFGNN_REDUCE_SUM, TFGNN_ACT_NONE, TFGNN_ACT_NONE, d_workspace, workspace_bytes, stream, d_out_sp,
                           ld_out_sp_bytes, d_inv_scale, d_fixed_inv_scale);
}

/* tfgnn_graph_gather_reduce_sp whose combine pass (the partial sums of buckets cut into several items) comes back as a job
 * for tfgnn_aux_launch instead of being launched: kind 0 = there is nothing to combine. */
extern "C" int tfgnn_graph_gather_reduce_sp_deferred(const tfgnn_graph* g, int view, const int32_t* d_col_override,
                                                     const float* d_edge_weight, const float* d_row_scale, const float* d_in,
                                                     int64_t ld_in, int width, void* d_out_sp, int64_t ld_out_sp_bytes,
                                                     float* d_inv_scale, const float* d_fixed_inv_scale, void* d_workspace,
                                                     size_t workspace_bytes, tfgnn_aux_job* combine_job, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(d_out_sp && (d_inv_scale || d_fixed_inv_scale) && combine_job, "tfgnn_graph_gather_reduce_sp_deferred: NULL output");
  TFGNN_REQUIRE(ld_out_sp_bytes >= (int64_t)width * 4 && ld_out_sp_bytes % 64 == 0 && (uintptr_t)d_out_sp % 64 == 0,
                "tfgnn_graph_gather_reduce_sp_deferred: bad SP16 leading dimension / alignment");
  return graph_gather_impl(g, view, d_col_override, d_edge_weight, 1, d_row_scale, d_in, ld_in, width, nullptr, 0,
                           TFGNN_REDUCE_SUM, TFGNN_ACT_NONE, TFGNN_ACT_NONE, d_workspace, workspace_bytes, stream, d_out_sp,
                           ld_out_sp_bytes, d_inv_scale, d_fixed_inv_scale, combine_job);
}
