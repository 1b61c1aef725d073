// Small passes that ride together in ONE launch (tfgnn_aux_launch): the per-step preparation and finishing work around the
// big kernels of a layer - splitting a weight matrix into SP16 operand form, combining the partial sums of the gather's
// multi-item buckets, summing the split-K partials of a weight gradient - is a handful of kernels of 5-15 us each, every one
// of them bound by launch + dependent-load latency, not by work.  Merged, the launch costs what its longest job costs.
// A job = (kind, number of 256-thread workgroups, payload); the kernel finds a workgroup's job from blockIdx.x by walking the
// table in the kernel-argument segment (scalar loads; no table in device memory, no copy command).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"
#include "sp16.hpp"
#include "tfgnn.h"

namespace tfgnn {

enum AuxKind { AUX_NONE = 0, AUX_SPLIT_ROWS = 1, AUX_SPLIT_COLS = 2, AUX_TN_REDUCE = 3, AUX_COMBINE_SP = 4, AUX_TN_FACTORS = 5, AUX_COL_ABSMAX = 6, AUX_KIND_END = 7 };

struct AuxSplitRows {
  const float* src;
  int64_t ld, seg_len, seg_stride, R, C;
  int sb;
  uint8_t* dst;
  int64_t ld_dst;
  float* inv;
  const float* fixed_inv;
};
struct AuxSplitCols {
  const float* src;
  int64_t ld, K, N;
  uint8_t* dst;
  int64_t ld_dst;
  float* inv;
  unsigned ncx, ncy;
  // round 6 (long K: stacked kernels of many relations): the column maxima come from a pass of their own (AUX_COL_ABSMAX in an
  // EARLIER launch) as nparts slabs [nparts][N]; NULL: every workgroup takes them over all of K itself
  const float* colmax_parts;
  int nparts;
};
// |x| maxima of the columns of a row-major [K, N] matrix over nparts row slabs: part[s][n] = max over the rows of slab s.
// One workgroup per (slab, chunk of 1024 columns); thread t owns the float4 column group t of its chunk, rows in sequence
// (whole rows are read: coalesced, each byte once - the column-strip workgroups of the conversion read 64-byte row pieces).
struct AuxColAbsmax {
  const float* src;
  int64_t ld, K, N;
  float* part;
  int nparts;
  unsigned nchunks;
};
static_assert(sizeof(AuxColAbsmax) <= sizeof(((tfgnn_aux_job*)0)->payload), "tfgnn_aux_job payload too small");
struct AuxTnReduce {
  const float* partial;
  int splits;
  int64_t M, N;
  const float* ref;
  int64_t a_col0;
  int a_sb;
  float* C;
  int64_t group_rows, stride_group, stride_row, stride_col;
  int accumulate;
  int64_t slab;
  int ref_ld;  // 0: one reference scale per block (ref[blk]); > 0: one per (split, block) at ref[z * ref_ld + blk]
};
struct AuxCombineSp {
  const int32_t *multi_row, *multi_base, *multi_n;
  int num_multi;
  const float* row_scale;
  const float* partial;
  int width;
  const int32_t* out_row_map;
  uint8_t* out_sp;
  int64_t ld_out_sp;
  float* inv_out;
  const float* fixed_inv;
};
struct AuxTnFactors {
  const float* inv_a;
  int64_t ld_a;
  const float* inv_b;
  int64_t ld_b;
  int64_t K;
  _Float16* F;
  int64_t f_ld;
  float* ref;
  int* spread_flag;
  int nchunks;
};
static_assert(sizeof(AuxTnFactors) <= sizeof(((tfgnn_aux_job*)0)->payload), "tfgnn_aux_job payload too small");
static_assert(sizeof(AuxSplitRows) <= sizeof(((tfgnn_aux_job*)0)->payload) && sizeof(AuxSplitCols) <= sizeof(((tfgnn_aux_job*)0)->payload) &&
                  sizeof(AuxTnReduce) <= sizeof(((tfgnn_aux_job*)0)->payload) && sizeof(AuxCombineSp) <= sizeof(((tfgnn_aux_job*)0)->payload),
              "tfgnn_aux_job payload too small");

template <class P>
static inline void aux_job_set(tfgnn_aux_job* j, int kind, unsigned blocks, const P& p) {
  j->kind = kind;
  j->num_blocks = blocks;
  memset(j->payload, 0, sizeof(j->payload));
  memcpy(j->payload, &p, sizeof(P));
}

// ---- fp32 -> SP16, rows ------------------------------------------------------------------------------------------
// One wave per (row, scale block).  Source element (r, c): src[r * ld + (c / seg_len) * seg_stride + c % seg_len]
// (seg_len = C, seg_stride = 0: a plain row-major matrix; otherwise a row assembled from C / seg_len segments, e.g.
// row d of [W_0[d,:] | W_1[d,:] | ...] from the stacked kernels [L, D, H]).  Two passes over the block (the second
// one hits L1 / L2).
__device__ __forceinline__ void sp_split_rows_body(const float* __restrict__ src, int64_t ld, int64_t seg_len,
                                                   int64_t seg_stride, int64_t R, int64_t C, int sb,
                                                   uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ inv,
                                                   const float* __restrict__ fixed_inv, unsigned block) {
  const int lane = threadIdx.x & 63;
  const int nblk = (int)(C / sb);
  const int64_t item = (int64_t)block * 4 + (threadIdx.x >> 6);
  if (item >= R * nblk) return;
  const int64_t r = item / nblk;
  const int blk = (int)(item - r * nblk);
  const float* srow = src + r * ld;
  const int64_t c0 = (int64_t)blk * sb;
  float s, iv;
  if (fixed_inv) {  // caller-chosen scale (a tensor-wide bound): inv given, s = 1 / inv (a power of two)
    iv = fixed_inv[0];
    s = 1.f / iv;
  } else {
    float mx = 0.f;
    for (int c = lane * 4; c < sb; c += 256) {
      const int64_t cc = c0 + c;
      const int64_t sg = cc / seg_len;
      const float4 v = *reinterpret_cast<const float4*>(srow + sg * seg_stride + (cc - sg * seg_len));
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    s = sp_scale_for_max(mx, &iv);
  }
  if (inv && lane == 0) inv[item] = iv;
  uint8_t* drow = dst + r * ld_dst;
  for (int c = lane * 4; c < sb; c += 256) {
    const int64_t cc = c0 + c;
    const int64_t sg = cc / seg_len;
    const float4 v = *reinterpret_cast<const float4*>(srow + sg * seg_stride + (cc - sg * seg_len));
    sp_store4(drow, cc, v, s);
  }
}

// ---- fp32 -> SP16, columns -----------------------------------------------------------------------------------------
// SP16 rows from the COLUMNS of a row-major fp32 matrix: dst row n, column k = src[k * ld + n]  (a Keras kernel
// [K, N] -> the [N, K] K-contiguous operand of the NT product), one scale per dst row.  Workgroup (x, y): 16 dst rows, the
// y-th slice of K; every workgroup takes the column maxima over ALL k itself (a weight matrix is L2 resident), so the
// slices need no second launch.
__device__ __forceinline__ void col_absmax_body(const AuxColAbsmax& a, unsigned block) {
  const unsigned s = block / a.nchunks, chunk = block - s * a.nchunks;
  const int64_t per = (a.K + a.nparts - 1) / a.nparts;
  const int64_t k0 = (int64_t)s * per, k1 = k0 + per < a.K ? k0 + per : a.K;
  const int64_t c = ((int64_t)chunk * 256 + threadIdx.x) * 4;  // N % 4 == 0
  if (c >= a.N) return;
  const float* p = a.src + c;
  float4 mx = {0.f, 0.f, 0.f, 0.f};
  int64_t k = k0;
  for (; k + 8 <= k1; k += 8) {  // eight rows in flight
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (k + u) * a.ld);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      mx.x = fmaxf(mx.x, fabsf(v[u].x)); mx.y = fmaxf(mx.y, fabsf(v[u].y));
      mx.z = fmaxf(mx.z, fabsf(v[u].z)); mx.w = fmaxf(mx.w, fabsf(v[u].w));
    }
  }
  for (; k < k1; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(p + k * a.ld);
    mx.x = fmaxf(mx.x, fabsf(v.x)); mx.y = fmaxf(mx.y, fabsf(v.y));
    mx.z = fmaxf(mx.z, fabsf(v.z)); mx.w = fmaxf(mx.w, fabsf(v.w));
  }
  *reinterpret_cast<float4*>(a.part + (int64_t)s * a.N + c) = mx;
}

__device__ __forceinline__ void sp_split_cols_body(const float* __restrict__ src, int64_t ld, int64_t K, int64_t N,
                                                   uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ inv,
                                                   unsigned bx, unsigned by, unsigned ny, const float* __restrict__ colmax_parts = nullptr,
                                                   int nparts = 0) {
  __shared__ float red[64][4];
  __shared__ float tile[2][16][65];
  __shared__ float sc[16];
  const int tid = threadIdx.x;
  const int64_t n0 = (int64_t)bx * 16;
  const int kq = tid >> 2, nq = (tid & 3) * 4;  // this thread: k = kq + 64 i, columns n0 + nq .. +3
  const bool ok = n0 + nq < N;                  // N % 4 == 0
  constexpr int NV = 20;                        // 64-row chunks held in registers: K <= 1280 (every layer kernel stack here)
  const int nv = (int)((K + 63) >> 6);
  const bool in_regs = nv <= NV;
  float4 v[NV];
  float4 mx = {0.f, 0.f, 0.f, 0.f};
  if (!in_regs && colmax_parts) {
    // the maxima were taken by a pass of their own (col_absmax_body): this thread reduces the slabs kq, kq + 64, .. of its four
    // columns; the reduction below then runs over the 64 threads with the same nq as for the direct pass.  (max is exact and
    // order-free: the same scales, bit for bit, as the pass over all of K.)
    if (ok)
      for (int sl = kq; sl < nparts; sl += 64) {
        const float4 u = *reinterpret_cast<const float4*>(colmax_parts + (int64_t)sl * N + n0 + nq);
        mx.x = fmaxf(mx.x, u.x); mx.y = fmaxf(mx.y, u.y); mx.z = fmaxf(mx.z, u.z); mx.w = fmaxf(mx.w, u.w);
      }
  } else if (in_regs) {
    // all of the strip's rows in flight at once (round 4: the loop below waited for four loads at a time - 15 us of latency
    // for 80 KB), kept for the conversion: nothing is read twice
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int64_t k = kq + 64 * i;
      v[i] = (i < nv && ok && k < K) ? *reinterpret_cast<const float4*>(src + k * ld + n0 + nq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      mx.x = fmaxf(mx.x, fabsf(v[i].x)); mx.y = fmaxf(mx.y, fabsf(v[i].y));
      mx.z = fmaxf(mx.z, fabsf(v[i].z)); mx.w = fmaxf(mx.w, fabsf(v[i].w));
    }
  } else {
    for (int64_t k0 = kq; k0 < K; k0 += 4 * 64) {
      float4 u4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + u * 64;
        u4[u] = (ok && k < K) ? *reinterpret_cast<const float4*>(src + k * ld + n0 + nq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mx.x = fmaxf(mx.x, fabsf(u4[u].x)); mx.y = fmaxf(mx.y, fabsf(u4[u].y));
        mx.z = fmaxf(mx.z, fabsf(u4[u].z)); mx.w = fmaxf(mx.w, fabsf(u4[u].w));
      }
    }
  }
  // reduce over the 64 k-rows of threads with the same nq: lanes 4 apart inside a wave (xor 4 .. 32), then the 4 waves
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    mx.x = fmaxf(mx.x, __shfl_xor(mx.x, o, 64)); mx.y = fmaxf(mx.y, __shfl_xor(mx.y, o, 64));
    mx.z = fmaxf(mx.z, __shfl_xor(mx.z, o, 64)); mx.w = fmaxf(mx.w, __shfl_xor(mx.w, o, 64));
  }
  if ((tid & 63) < 4) {
    float* r = &red[(tid >> 6) * 4 + (tid & 3)][0];
    r[0] = mx.x; r[1] = mx.y; r[2] = mx.z; r[3] = mx.w;
  }
  __syncthreads();
  if (tid < 16) {  // column n0 + tid = chunk tid / 4, component tid % 4
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = fmaxf(m, red[w * 4 + (tid >> 2)][tid & 3]);
    float iv;
    sc[tid] = sp_scale_for_max(m, &iv);
    if (inv && by == 0 && n0 + tid < N) inv[n0 + tid] = iv;
  }
  __syncthreads();
  const int64_t kper = (((K + ny - 1) / ny) + 63) & ~63ll;
  const int64_t kend = (by + 1) * kper < K ? (by + 1) * kper : K;
  // 16 rows x 64 k = 256 float4 items per chunk: thread -> row tid / 16, k4 = (tid % 16) * 4
  const int rr = tid >> 4, k4 = (tid & 15) * 4;
  if (in_regs) {
    const int i0 = (int)((by * kper) >> 6), i1 = (int)((kend + 63) >> 6);  // this workgroup's chunks (block-uniform)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i >= i0 && i < i1) {
        float (*t)[65] = tile[i & 1];  // two tiles: one barrier per chunk
        t[nq + 0][kq] = v[i].x; t[nq + 1][kq] = v[i].y; t[nq + 2][kq] = v[i].z; t[nq + 3][kq] = v[i].w;
        __syncthreads();
        const int64_t kb = (int64_t)i * 64;
        if (n0 + rr < N && kb + k4 < K) {
          const float4 o = {t[rr][k4], t[rr][k4 + 1], t[rr][k4 + 2], t[rr][k4 + 3]};
          sp_store4(dst + (n0 + rr) * ld_dst, kb + k4, o, sc[rr]);
        }
      }
    }
    return;
  }
  for (int64_t kb = by * kper; kb < kend; kb += 64) {
    const int64_t k = kb + kq;
    float4 w4 = {0.f, 0.f, 0.f, 0.f};
    if (ok && k < K) w4 = *reinterpret_cast<const float4*>(src + k * ld + n0 + nq);
    tile[0][nq + 0][kq] = w4.x; tile[0][nq + 1][kq] = w4.y; tile[0][nq + 2][kq] = w4.z; tile[0][nq + 3][kq] = w4.w;
    __syncthreads();
    if (n0 + rr < N && kb + k4 < K) {
      const float4 o = {tile[0][rr][k4], tile[0][rr][k4 + 1], tile[0][rr][k4 + 2], tile[0][rr][k4 + 3]};
      sp_store4(dst + (n0 + rr) * ld_dst, kb + k4, o, sc[rr]);
    }
    __syncthreads();
  }
}

// ---- split-K partials of a weight gradient -> dW -----------------------------------------------------------------------
// C[(m / group_rows) * stride_group + (m % group_rows) * stride_row + n * stride_col] (+)= ref[blk(m)] * sum_z partial[z][m][n]
// (`block` of `nblocks` workgroups of 256 threads walk the M * N elements)
__device__ __forceinline__ void sp_tn_reduce_body(const AuxTnReduce& a, unsigned block, unsigned nblocks) {
  const int64_t total = a.M * a.N;  // slab >= total: floats per split (the product pads M to a multiple of 128)
  for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < total; i += (int64_t)nblocks * 256) {
    const int64_t m = i / a.N, n = i - m * a.N;
    const int64_t blk = (a.a_col0 + m) / a.a_sb;
    float s = 0.f;
    int z = 0;
    if (a.ref_ld > 0) {  // every split carries its own reference scale (factors computed inside the product kernel)
      for (; z + 8 <= a.splits; z += 8) {  // eight loads in flight; the sum stays in split order
        float v[8], r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = a.partial[(int64_t)(z + u) * a.slab + i];
          r[u] = a.ref[(int64_t)(z + u) * a.ref_ld + blk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u] * r[u];
      }
      for (; z < a.splits; ++z) s += a.partial[(int64_t)z * a.slab + i] * a.ref[(int64_t)z * a.ref_ld + blk];
    } else {
      for (; z + 8 <= a.splits; z += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a.partial[(int64_t)(z + u) * a.slab + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; z < a.splits; ++z) s += a.partial[(int64_t)z * a.slab + i];
      s *= a.ref[blk];
    }
    const int64_t gi = m / a.group_rows, mi = m - gi * a.group_rows;
    float* dst = a.C + gi * a.stride_group + mi * a.stride_row + n * a.stride_col;
    *dst = a.accumulate ? *dst + s : s;
  }
}

// ---- per-k factors of a weight-gradient product (gemm_sp.hip sp_tn_factors_kernel, as a 256-thread job): workgroup
// (block b, chunk c) takes the maximum of inv_a[k, b] * inv_b[k] over ALL k itself, then writes its slice of F[b][.] ----------
__device__ __forceinline__ void sp_tn_factors_body(const AuxTnFactors& a, unsigned block) {
  __shared__ float red[4];
  const int b = (int)block / a.nchunks, chunk = (int)block % a.nchunks;
  float mx = 0.f;
  for (int64_t k0 = threadIdx.x; k0 < a.K; k0 += 32 * 256) {
    float va[32], vb[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int64_t k = k0 + u * 256;
      va[u] = k < a.K ? a.inv_a[k * a.ld_a + b] : 0.f;
      vb[u] = (k < a.K && a.inv_b) ? a.inv_b[k * a.ld_b] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) mx = fmaxf(mx, va[u] * vb[u]);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (mx == 0.f) mx = 1.f;
  if (threadIdx.x == 0 && chunk == 0) a.ref[b] = mx;
  const float r = 1.f / mx;  // powers of two: exact
  const int64_t per = ((a.f_ld + a.nchunks - 1) / a.nchunks + 7) & ~7ll;
  const int64_t kend = (chunk + 1) * per < a.f_ld ? (chunk + 1) * per : a.f_ld;
  bool wide = false;
  for (int64_t k = chunk * per + threadIdx.x; k < kend; k += 256) {
    const float ia = k < a.K ? a.inv_a[k * a.ld_a + b] : 0.f, ib = (k < a.K && a.inv_b) ? a.inv_b[k * a.ld_b] : 1.f;
    const float f = ia * ib * r;
    a.F[(int64_t)b * a.f_ld + k] = (_Float16)f;
    wide |= sp_row_too_small(f, ia, ib);  // the spread guard of sp_tn_factors_kernel (gemm_sp.hip)
  }
  if (a.spread_flag && __any(wide) && (threadIdx.x & 63) == 0) *a.spread_flag = 1;
}

// ---- partial sums of the gather's multi-item buckets -> SP16 rows: one wave per bucket (width <= 2048 floats) ------------
__device__ __forceinline__ void combine_sp_body(const AuxCombineSp& a, unsigned block) {
  const int m = (int)block * 4 + (threadIdx.x >> 6);
  if (m >= a.num_multi) return;
  const int lane = threadIdx.x & 63;
  const int64_t row = a.multi_row[m];
  const int32_t base = a.multi_base[m], n = a.multi_n[m];
  const float rs = a.row_scale ? a.row_scale[row] : 1.f;
  float4 sum[8];
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (lane + i * 64) * 4;
    sum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < a.width) {
      int k = 0;
      for (; k + 4 <= n; k += 4) {  // four partial rows in flight; the sum stays in item order
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = *reinterpret_cast<const float4*>(a.partial + (int64_t)(base + k + u) * a.width + c);
#pragma unroll
        for (int u = 0; u < 4; ++u) { sum[i].x += p[u].x; sum[i].y += p[u].y; sum[i].z += p[u].z; sum[i].w += p[u].w; }
      }
      for (; k < n; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(a.partial + (int64_t)(base + k) * a.width + c);
        sum[i].x += p.x; sum[i].y += p.y; sum[i].z += p.z; sum[i].w += p.w;
      }
      sum[i].x *= rs; sum[i].y *= rs; sum[i].z *= rs; sum[i].w *= rs;
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(sum[i].x), fabsf(sum[i].y)), fmaxf(fabsf(sum[i].z), fabsf(sum[i].w))));
    }
  }
  const int64_t orow = a.out_row_map ? a.out_row_map[row] : row;
  float iv, sc;
  if (a.fixed_inv) {
    iv = a.fixed_inv[0];
    sc = 1.f / iv;
  } else {
#pragma unroll
    for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    sc = sp_scale_for_max(mx, &iv);
    if (lane == 0) a.inv_out[orow] = iv;
  }
  uint8_t* drow = a.out_sp + orow * a.ld_out_sp;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = (lane + i * 64) * 4;
    if (c < a.width) sp_store4(drow, c, sum[i], sc);
  }
}

}  // namespace tfgnn
