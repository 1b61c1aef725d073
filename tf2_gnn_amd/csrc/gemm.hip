// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   C[M,N] = act( op(A)[M,K] @ op(B)[K,N] + bias[N] )  (+ C)
//
// This is the compute-bound half of the hot path: the per-relation weight multiply of
// RGCN / GGNN / RGIN / GNN_Edge_MLP (Keras Dense inside dpu_utils MLP, gnn_edge_mlp.py:100),
// RGAT's Dense (rgat.py:102-109), the GNN glue Dense layers (gnn.py:279,324-327), GRUCell's two
// matmuls (ggnn.py:84-87) and every MatMul gradient TensorFlow would run for them.
//
// Structure: WM x WN waves per workgroup, each wave owns TM x TN tiles of 32 x 32 (block tile
// 32*WM*TM x 32*WN*TN), BK = 32.  N = 320-family shapes use 8 waves (4 x 2) on a 128 x 320 tile: the
// 40 KB weight tile is staged once per CU and two waves share each SIMD, so one wave's LDS fragment
// reads hide under the other's MFMAs.  Global -> registers -> LDS staging with the next tile's
// global loads in flight while the current tile is multiplied (the fp32 MFMA takes 64 cycles per
// instruction, so one tile of prefetch covers HBM latency).  LDS layouts are chosen so that every
// fragment read is bank-conflict free: operands whose K index is contiguous in memory are stored
// [mn][BK+1] (padded, ds_write_b32), operands whose M/N index is contiguous are stored [k][mn]
// (ds_write_b128).  The staging code is specialised at compile time for 16-byte aligned operands
// (VEC: one global_load_dwordx4 per staged vector, pointers advanced by a constant per K tile) so
// that the steady-state loop is a few hundred instructions; a scalar variant covers odd shapes.
// Split-K (blockIdx.z) with a deterministic second-pass reduction covers the weight-gradient
// shapes (small M x N, K = number of nodes).
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace tfgnn {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

struct GemmArgs {
  int64_t M, N, K;
  const float* A;
  int64_t lda;
  const float* B;
  int64_t ldb;
  float* C;
  int64_t ldc;
  const float* bias;
  int act;
  int accumulate;
  int64_t k_chunk;  // K range per blockIdx.z (multiple of BK)
  int splits;
  float* partial;  // [splits][M][N] when splits > 1
  unsigned n_tiles;
  // grouped variants (blockIdx.y = group g, rows [group_off[g], group_off[g+1]) of the stacked operand):
  //   1: M-grouped  C[rows g] = act(A[rows g] @ op(B + g*strideB))          (per-relation multiply over the
  //                                                                          non-empty buckets of one edge type)
  //   2: K-grouped  C + g*strideC = A[rows g]^T @ B[rows g]  (trans_a = 1)   (its weight gradient)
  int group_mode;
  const int32_t* group_off;
  int64_t strideB, strideC;
  int wide_store;  // C rows are 16-byte aligned and N % 4 == 0: float4 epilogue
};

// ---- staging of one operand tile -------------------------------------------------------------
// KCONTIG: source is [MN_total, K] row-major (ld), tile = MN x BK;  LDS [MN][BK+1]
// else   : source is [K, MN_total] row-major (ld), tile = BK x MN;  LDS [BK][MN]
// Thread t stages NV vectors of 4 floats per tile; vector p covers
//   KCONTIG : row mn = (t >> 3) + (NT/8) p, k = (t & 7) * 4 .. +3
//   else    : q = t + NT p, k = q / (MN/4), mn = (q % (MN/4)) * 4 .. +3
template <int MN, bool KCONTIG, bool VEC, int NT>
struct Stage {
  static constexpr int NV = MN * BK / 4 / NT;  // vectors per thread per tile
  static_assert(MN * BK / 4 % NT == 0, "tile must divide evenly over the workgroup");
  float4 r[NV];
  const float* ptr[NV];  // current global address of vector p (advances by one K tile per load)
  bool row_ok[NV];       // mn inside the matrix (VEC: whole vector in or out)
  int koff[NV];          // k offset of the vector inside the tile
  int mn_rem[NV];        // scalar path: number of valid mn (non-KCONTIG) elements in the vector

  __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int64_t mn0, int64_t mn_total,
                                       int64_t k_begin, int tid) {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      int64_t mn;
      int k;
      if (KCONTIG) {
        mn = mn0 + (tid >> 3) + (NT / 8) * p;
        k = (tid & 7) * 4;
        ptr[p] = src + mn * ld + k_begin + k;
        mn_rem[p] = 4;
      } else {
        const int q = tid + NT * p;
        k = q / (MN / 4);
        mn = mn0 + (q % (MN / 4)) * 4;
        ptr[p] = src + (k_begin + k) * ld + mn;
        const int64_t rem = mn_total - mn;
        mn_rem[p] = rem > 4 ? 4 : (rem < 0 ? 0 : (int)rem);
      }
      koff[p] = k;
      row_ok[p] = mn < mn_total;
    }
  }

  // k_left = number of valid k from the start of this tile
  __device__ __forceinline__ void load(int64_t k_left, int64_t ld) {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok[p]) {
        if (VEC) {
          if (koff[p] < k_left) v = *reinterpret_cast<const float4*>(ptr[p]);
        } else if (KCONTIG) {
          const int64_t n = k_left - koff[p];
          if (n > 0) v.x = ptr[p][0];
          if (n > 1) v.y = ptr[p][1];
          if (n > 2) v.z = ptr[p][2];
          if (n > 3) v.w = ptr[p][3];
        } else if (koff[p] < k_left) {
          const int n = mn_rem[p];
          if (n > 0) v.x = ptr[p][0];
          if (n > 1) v.y = ptr[p][1];
          if (n > 2) v.z = ptr[p][2];
          if (n > 3) v.w = ptr[p][3];
        }
      }
      r[p] = v;
      ptr[p] += KCONTIG ? (int64_t)BK : (int64_t)BK * ld;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      if (KCONTIG) {
        float* d = lds + ((tid >> 3) + (NT / 8) * p) * (BK + 1) + (tid & 7) * 4;
        d[0] = r[p].x;
        d[1] = r[p].y;
        d[2] = r[p].z;
        d[3] = r[p].w;
      } else {
        const int q = tid + NT * p;
        float* d = lds + (q / (MN / 4)) * MN + (q % (MN / 4)) * 4;
        *reinterpret_cast<float4*>(d) = r[p];
      }
    }
  }

  // fragment element for MFMA 32x32x2: lane (i = lane & 31, kk = lane >> 5)
  static __device__ __forceinline__ float frag(const float* __restrict__ lds, int mn, int k) {
    return KCONTIG ? lds[mn * (BK + 1) + k] : lds[k * MN + mn];
  }
  static constexpr int LDS_FLOATS = KCONTIG ? MN * (BK + 1) : BK * MN;
};

// the activation switch of the epilogue as out-of-line functions: inlined into every unrolled store of every
// instantiation it made this translation unit take 160 s to compile (and the kernels 25 000 lines of ISA)
__device__ __noinline__ float4 act_apply4(int act, float4 v) {
  v.x = act_apply(act, v.x); v.y = act_apply(act, v.y);
  v.z = act_apply(act, v.z); v.w = act_apply(act, v.w);
  return v;
}
__device__ __noinline__ float act_apply1(int act, float v) { return act_apply(act, v); }

template <int WM_, int WN_, int TM, int TN, bool TA, bool TB, bool VEC>
__global__ void __launch_bounds__(64 * WM_ * WN_) gemm_mfma_kernel(GemmArgs g) {
  constexpr int NT = 64 * WM_ * WN_;
  constexpr int BM = 32 * WM_ * TM, BN = 32 * WN_ * TN;
  using SA = Stage<BM, !TA, VEC, NT>;  // A stored [M,K] -> K contiguous unless transposed
  using SB = Stage<BN, TB, VEC, NT>;   // B stored [K,N] -> N contiguous unless transposed
  // two LDS stages: tile t+1 is written while tile t is being multiplied -> one barrier per K tile
  constexpr int PATCH_FLOATS = (NT / 64) * 32 * 36;  // epilogue staging, one 32 x 36 patch per wave
  constexpr int STAGE_FLOATS = 2 * (SA::LDS_FLOATS + SB::LDS_FLOATS);
  __shared__ __attribute__((aligned(16))) float lds_raw[STAGE_FLOATS > PATCH_FLOATS ? STAGE_FLOATS : PATCH_FLOATS];
  float(*lds_a)[SA::LDS_FLOATS] = reinterpret_cast<float(*)[SA::LDS_FLOATS]>(lds_raw);
  float(*lds_b)[SB::LDS_FLOATS] = reinterpret_cast<float(*)[SB::LDS_FLOATS]>(lds_raw + 2 * SA::LDS_FLOATS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int64_t m0 = (int64_t)(blockIdx.x / g.n_tiles) * BM;
  const int64_t n0 = (int64_t)(blockIdx.x % g.n_tiles) * BN;
  int64_t k_begin = (int64_t)blockIdx.z * g.k_chunk;
  int64_t k_end = k_begin + g.k_chunk < g.K ? k_begin + g.k_chunk : g.K;
  int64_t Mloc = g.M;
  const float* Ap = g.A;
  const float* Bp = g.B;
  float* Cp = g.C;
  int64_t partial_slab = blockIdx.z;
  if (g.group_mode == 1) {
    const int64_t gb = g.group_off[blockIdx.y], ge = g.group_off[blockIdx.y + 1];
    Mloc = ge - gb;
    if (m0 >= Mloc) return;  // uniform for the whole workgroup
    Ap += gb * g.lda;
    Cp += gb * g.ldc;
    Bp += (int64_t)blockIdx.y * g.strideB;
  } else if (g.group_mode == 2) {
    const int64_t gb = g.group_off[blockIdx.y], ge = g.group_off[blockIdx.y + 1];
    const int64_t per = (ge - gb + g.splits - 1) / g.splits;
    const int64_t chunk = (per + BK - 1) / BK * BK;
    k_begin = gb + (int64_t)blockIdx.z * chunk;
    k_end = k_begin + chunk < ge ? k_begin + chunk : ge;
    Cp += (int64_t)blockIdx.y * g.strideC;
    partial_slab = (int64_t)blockIdx.y * g.splits + blockIdx.z;
  }

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;

  if (k_begin < k_end) {
    SA sa;
    SB sb;
    sa.init(Ap, g.lda, m0, Mloc, k_begin, tid);
    sb.init(Bp, g.ldb, n0, g.N, k_begin, tid);
    sa.load(k_end - k_begin, g.lda);
    sb.load(k_end - k_begin, g.ldb);
    sa.store(lds_a[0], tid);
    sb.store(lds_b[0], tid);
    __syncthreads();
    int stage = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK, stage ^= 1) {
      const bool more = k0 + BK < k_end;
      if (more) {
        sa.load(k_end - k0 - BK, g.lda);
        sb.load(k_end - k0 - BK, g.ldb);
      }
      const float* la = lds_a[stage];
      const float* lb = lds_b[stage];
      // fragments of k-step kk+1 are fetched from LDS while the MFMAs of k-step kk run
      float fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = SA::frag(la, (wm * TM + i) * 32 + li, lk);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = SB::frag(lb, (wn * TN + j) * 32 + li, lk);
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[nxt][i] = SA::frag(la, (wm * TM + i) * 32 + li, 2 * (kk + 1) + lk);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nxt][j] = SB::frag(lb, (wn * TN + j) * 32 + li, 2 * (kk + 1) + lk);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      }
      if (more) {
        // the other stage was last read before the previous barrier: safe to overwrite now
        sa.store(lds_a[stage ^ 1], tid);
        sb.store(lds_b[stage ^ 1], tid);
      }
      __syncthreads();
    }
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const bool split = g.splits > 1;
  float* outp = split ? g.partial + partial_slab * g.M * g.N : Cp;
  const int64_t ldo = split ? g.N : g.ldc;
  if (VEC && g.wide_store) {
    // Wide epilogue: every 32x32 accumulator tile goes through a wave-private LDS patch so that each
    // lane stores 16 contiguous bytes (4x fewer store instructions than the per-register layout).
    constexpr int PS = 36;  // patch row stride (floats): 16-byte aligned rows, conflict-free writes
    __syncthreads();        // the staging buffers are free now
    float* patch = lds_raw + wave * 32 * PS;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PS + li] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");  // LDS only: never wait for the previous tile's global stores
        __builtin_amdgcn_wave_barrier();
        const int64_t col = n0 + (wn * TN + j) * 32 + (lane & 7) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pr = (lane >> 3) + 8 * q;
          const int64_t row = m0 + (wm * TM + i) * 32 + pr;
          float4 v = *reinterpret_cast<const float4*>(patch + pr * PS + (lane & 7) * 4);
          if (row < Mloc && col < g.N) {  // N % 4 == 0 in this mode: the vector is fully in or out
            float* dst = outp + row * ldo + col;
            if (!split) {
              if (g.bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(g.bias + col);
                v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
              }
              v = act_apply4(g.act, v);
              if (g.accumulate) {
                const float4 c4 = *reinterpret_cast<const float4*>(dst);
                v.x += c4.x; v.y += c4.y; v.z += c4.z; v.w += c4.w;
              }
            }
            *reinterpret_cast<float4*>(dst) = v;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t col = n0 + (wn * TN + j) * 32 + li;
        if (col < g.N) {
          const float bv = (!split && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row < Mloc) {
              float v = acc[i][j][r];
              if (!split) {
                v = act_apply1(g.act, v + bv);
                if (g.accumulate) v += outp[row * ldo + col];
              }
              outp[row * ldo + col] = v;
            }
          }
        }
      }
    }
  }
}

__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs g) {
  const int64_t total = g.M * g.N;
  const float* part = g.partial + (int64_t)blockIdx.y * g.splits * total;  // blockIdx.y = group (0 if ungrouped)
  float* C = g.C + (int64_t)blockIdx.y * g.strideC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += part[(int64_t)z * total + i];
    const int64_t row = i / g.N, col = i - row * g.N;
    if (g.bias) s += g.bias[col];
    s = act_apply(g.act, s);
    float* c = C + row * g.ldc + col;
    if (g.accumulate) s += *c;
    *c = s;
  }
}

// tile configurations: {waves_m, waves_n, tm, tn} -> block tile (32*wm*tm) x (32*wn*tn)
enum { CFG_128x320 = 0, CFG_128x128 = 1, CFG_64x64 = 2 };
struct GemmPlan {
  int cfg;
  int bm, bn;
  int splits;
  int64_t k_chunk;
};

static GemmPlan plan_gemm(int64_t M, int64_t N, int64_t K, size_t workspace_bytes, bool vec) {
  GemmPlan p;
  if (!vec) {
    p.cfg = CFG_64x64;
  } else if (N % 320 == 0 || (N > 256 && N <= 320)) {
    p.cfg = CFG_128x320;
  } else if (N > 64 && M > 64) {
    p.cfg = CFG_128x128;
  } else {
    p.cfg = CFG_64x64;
  }
  p.bm = p.cfg == CFG_64x64 ? 64 : 128;
  p.bn = p.cfg == CFG_128x320 ? 320 : (p.cfg == CFG_128x128 ? 128 : 64);
  const int64_t tiles = ceil_div(M, p.bm) * ceil_div(N, p.bn);
  p.splits = 1;
  p.k_chunk = ceil_div(K > 0 ? K : 1, BK) * BK;
  // split-K only when the output grid cannot fill the chip and K is long
  if (tiles < 192 && K >= 1024) {
    // one wave of workgroups on the 256 CUs: more splits only add rounds (12 output tiles x 43 splits ran
    // as 3 rounds of 22 K-tiles; x 21 splits is one round of 45)
    const int64_t want = splitk_want(tiles, K);
    int64_t max_by_k = K / (4 * BK);
    int64_t s = want < max_by_k ? want : max_by_k;
    // (exact division: with "+ 1" a workspace of exactly tfgnn_gemm_workspace_bytes() bytes allowed one split fewer than a larger
    //  one - the same product then summed in another order depending on which buffer the caller happened to hold: round 6)
    int64_t max_by_ws = (int64_t)(workspace_bytes / ((size_t)(M * N) * 4));
    if (s > max_by_ws) s = max_by_ws;
    if (s > 1) {
      p.k_chunk = ceil_div(ceil_div(K, s), BK) * BK;
      p.splits = (int)ceil_div(K, p.k_chunk);
    }
  }
  return p;
}

template <int WM_, int WN_, int TM, int TN, bool VEC>
static void launch_cfg(const GemmArgs& g, int ta, int tb, dim3 grid, hipStream_t s) {
  count_launch(TFGNN_KFAM_GEMM_FP32);
  dim3 block(64 * WM_ * WN_);
  if (!ta && !tb) hipLaunchKernelGGL((gemm_mfma_kernel<WM_, WN_, TM, TN, false, false, VEC>), grid, block, 0, s, g);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_mfma_kernel<WM_, WN_, TM, TN, false, true, VEC>), grid, block, 0, s, g);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_mfma_kernel<WM_, WN_, TM, TN, true, false, VEC>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<WM_, WN_, TM, TN, true, true, VEC>), grid, block, 0, s, g);
}

// float4 staging needs 16-byte aligned rows and the vectorised index to be a multiple of 4
static bool operands_vectorisable(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A,
                                  int64_t lda, const float* B, int64_t ldb) {
  const bool a16 = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
  const bool b16 = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
  return a16 && b16 && ((trans_a ? M : K) % 4 == 0) && ((trans_b ? K : N) % 4 == 0);
}

// gemm_x3.hip
int gemm_x3_mode();
int gemm_x3_set_mode(int mode);
int gemm_x3_try(int nprod, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int accumulate,
                void* workspace, size_t workspace_bytes, hipStream_t s, int* status, const float* mul = nullptr,
                int64_t ld_mul = 0, const float* saved = nullptr, int64_t ld_saved = 0, int dact = 0);

int gemm_x3_gathered_try(int nprod, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t a_rows,
                         const int32_t* index, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                         int accumulate, hipStream_t s, int* status);
int gemm_x3_gru(int nprod, int64_t M, int H, int64_t K, const float* A, int64_t lda, const float* Bt, const float* bias,
                const float* mh, const float* h, float* h_new, float* gates, hipStream_t s);

int gemm_x3_try_grouped_rows(int nprod, int trans_b, int num_groups, const int32_t* group_off, int64_t max_rows, int64_t N,
                             int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t stride_b,
                             float* C, int64_t ldc, int act, hipStream_t s, int* status, int dact = 0, const float* saved = nullptr,
                             int64_t ld_saved = 0);
int gemm_x3_try_grouped_k(int nprod, int num_groups, const int32_t* group_off, int64_t max_rows, int64_t M, int64_t N,
                          const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t stride_c, void* workspace, size_t workspace_bytes, hipStream_t s, int* status);

int gemm_f16x2_mode();
void gemm_f16x2_set(int on);
int gemm_x3_gru2(int nprod, int64_t M, int H, int64_t K, const float* A, int64_t lda, const float* Bt, const float* bias, const float* h,
                 const float* B2t, const float* bias2, float* h_new, float* gates, float* mh_out, hipStream_t s);
int gemm_x3_gathered_supported(int nprod, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t a_rows);
}  // namespace tfgnn

extern "C" int tfgnn_gemm_set_mode(int mode) {
  using namespace tfgnn;
  TFGNN_REQUIRE(mode == TFGNN_GEMM_FP32 || mode == TFGNN_GEMM_BF16X3 || mode == TFGNN_GEMM_BF16X3_EXACT || mode == TFGNN_GEMM_F16X2,
                "unknown GEMM mode %d", mode);
  if (mode == TFGNN_GEMM_F16X2) {
    // (re-)arm the spread guard: a factor pass still in flight could set the flag after it has been cleared - wait for the
    // device first (a mode switch is rare)
    (void)hipDeviceSynchronize();
    (void)tfgnn_sp_spread_flag(1);
    gemm_x3_set_mode(TFGNN_GEMM_BF16X3);
    gemm_f16x2_set(1);
  } else {
    gemm_x3_set_mode(mode);
    gemm_f16x2_set(0);
  }
  return TFGNN_OK;
}

extern "C" int tfgnn_gemm_get_mode(void) {
  using namespace tfgnn;
  if (gemm_f16x2_mode()) {
    if (!tfgnn_sp_spread_flag(0)) return TFGNN_GEMM_F16X2;
    gemm_f16x2_set(0);  // the guard tripped: exact kernels from here on (sticky)
  }
  return gemm_x3_mode();
}

extern "C" int tfgnn_gemm_grad_epilogue(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A,
                                        int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                                        const float* d_mul, int64_t ld_mul, int act_of_saved, const float* d_saved,
                                        int64_t ld_saved, int accumulate, void* d_workspace, size_t workspace_bytes,
                                        void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative GEMM size");
  if (M == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_C && d_A && d_B, "NULL operand");
  TFGNN_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "bad leading dimension");
  TFGNN_REQUIRE((!d_mul || ld_mul >= N) && (!d_saved || ld_saved >= N), "bad leading dimension of an epilogue operand");
  const int nprod = gemm_x3_mode();
  if (!nprod) return TFGNN_ERR_UNSUPPORTED;
  int status = TFGNN_OK;
  if (gemm_x3_try(nprod, trans_a, trans_b, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, nullptr, TFGNN_ACT_NONE, accumulate ? 1 : 0,
                  d_workspace, d_workspace ? workspace_bytes : 0, (hipStream_t)stream, &status, d_mul, ld_mul, d_saved, ld_saved,
                  act_of_saved))
    return status;
  return TFGNN_ERR_UNSUPPORTED;
}

extern "C" int tfgnn_gemm_gathered(int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A, int64_t lda, int64_t a_rows,
                                   const int32_t* d_row_index, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                                   const float* d_bias, int act, int accumulate, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(M >= 0 && N >= 0 && K >= 0 && a_rows >= 0, "negative GEMM size");
  TFGNN_REQUIRE(accumulate >= 0 && accumulate <= 2, "accumulate is 0 (no), 1 (after the activation) or 2 (before it)");
  if (M == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_C && d_A && d_B && d_row_index, "NULL operand");
  TFGNN_REQUIRE(lda >= K && ldb >= (trans_b ? K : N) && ldc >= N, "bad leading dimension");
  const int nprod = gemm_x3_mode();
  if (!nprod) return TFGNN_ERR_UNSUPPORTED;
  int status = TFGNN_OK;
  if (gemm_x3_gathered_try(nprod, trans_b, M, N, K, d_A, lda, a_rows, d_row_index, d_B, ldb, d_C, ldc, d_bias, act, accumulate,
                           (hipStream_t)stream, &status))
    return status;
  return TFGNN_ERR_UNSUPPORTED;
}

extern "C" int tfgnn_gemm_gathered_supported(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t a_rows) {
  return tfgnn::gemm_x3_gathered_supported(tfgnn::gemm_x3_mode(), M, N, K, lda, a_rows);
}

extern "C" int tfgnn_gemm_gru(int64_t V, int H, int64_t K, const float* d_x, int64_t ld_x, const float* d_kernel_t,
                              const float* d_bias, const float* d_mh, const float* d_h, float* d_h_new, float* d_gates,
                              void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && H >= 0 && K >= 0, "negative size");
  if (V == 0 || H == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_kernel_t && d_mh && d_h && d_h_new, "NULL pointer");
  TFGNN_REQUIRE(ld_x >= K, "bad leading dimension");
  const int nprod = gemm_x3_mode();
  if (nprod && gemm_x3_gru(nprod, V, H, K, d_x, ld_x, d_kernel_t, d_bias, d_mh, d_h, d_h_new, d_gates, (hipStream_t)stream)) {
    TFGNN_LAUNCH_CHECK();
    return TFGNN_OK;
  }
  return TFGNN_ERR_UNSUPPORTED;
}

extern "C" int tfgnn_gemm_gru2(int64_t V, int H, const float* d_x, int64_t ld_x, const float* d_kernel_t, const float* d_bias,
                               const float* d_h, const float* d_recurrent_kernel_t, const float* d_recurrent_bias, float* d_h_new,
                               float* d_gates, float* d_mh_out, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && H >= 0, "negative size");
  if (V == 0 || H == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_kernel_t && d_h && d_recurrent_kernel_t && d_h_new, "NULL pointer");
  TFGNN_REQUIRE(ld_x >= H, "bad leading dimension");
  const int nprod = gemm_x3_mode();
  if (nprod && gemm_x3_gru2(nprod, V, H, H, d_x, ld_x, d_kernel_t, d_bias, d_h, d_recurrent_kernel_t, d_recurrent_bias, d_h_new, d_gates,
                            d_mh_out, (hipStream_t)stream)) {
    TFGNN_LAUNCH_CHECK();
    return TFGNN_OK;
  }
  return TFGNN_ERR_UNSUPPORTED;
}

extern "C" size_t tfgnn_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  using namespace tfgnn;
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  // upper bound over both staging variants
  GemmPlan p1 = plan_gemm(M, N, K, (size_t)1 << 40, true);
  GemmPlan p2 = plan_gemm(M, N, K, (size_t)1 << 40, false);
  int64_t s = std::max(p1.splits, p2.splits);
  // ... and over the split-operand kernels of the other GEMM modes (gemm_x3_try: 128-row tiles of up to 320 columns): with
  // exactly this many bytes every mode takes the split count it would take with any larger buffer - the summation order of a
  // product does not depend on which workspace the caller holds (round 6: a captured step and its eager twin differed in the
  // last bit of a weight gradient because their streams' workspaces had different sizes)
  if (K >= 1024) {
    const int64_t tiles_min = ceil_div(M, 128) * ceil_div(N, 320);
    if (tiles_min < 192) s = std::max<int64_t>(s, std::min<int64_t>(splitk_want(tiles_min, K), K / 128));
  }
  return s > 1 ? (size_t)s * (size_t)M * (size_t)N * 4 : 0;
}

extern "C" int tfgnn_gemm(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* d_A,
                          int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                          const float* d_bias, int act, int accumulate, void* d_workspace,
                          size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "negative GEMM size");
  if (M == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_C != nullptr, "C is NULL");
  TFGNN_REQUIRE(K == 0 || (d_A && d_B), "A or B is NULL");
  TFGNN_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "bad leading dimension");
  hipStream_t s = (hipStream_t)stream;
  if (const int nprod = gemm_x3_mode()) {
    int status = TFGNN_OK;
    if (gemm_x3_try(nprod, trans_a, trans_b, M, N, K, d_A, lda, d_B, ldb, d_C, ldc, d_bias, act, accumulate,
                    d_workspace, d_workspace ? workspace_bytes : 0, s, &status))
      return status;
  }
  const bool vec = K > 0 && operands_vectorisable(trans_a, trans_b, M, N, K, d_A, lda, d_B, ldb);
  GemmPlan p = plan_gemm(M, N, K, d_workspace ? workspace_bytes : 0, vec);
  GemmArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc;
  g.bias = d_bias; g.act = act; g.accumulate = accumulate;
  g.k_chunk = p.k_chunk; g.splits = p.splits; g.partial = (float*)d_workspace;
  g.group_mode = 0; g.group_off = nullptr; g.strideB = 0; g.strideC = 0;
  {
    const bool split = p.splits > 1;
    const float* cbase = split ? (const float*)d_workspace : d_C;
    const int64_t ldo = split ? N : ldc;
    g.wide_store = vec && (N % 4 == 0) && (ldo % 4 == 0) && ((uintptr_t)cbase % 16 == 0) &&
                   (d_bias == nullptr || (uintptr_t)d_bias % 16 == 0);
  }
  g.n_tiles = (unsigned)ceil_div(N, p.bn);
  const int64_t tiles = ceil_div(M, p.bm) * (int64_t)g.n_tiles;
  TFGNN_REQUIRE(tiles < ((int64_t)1 << 31), "GEMM grid too large");
  dim3 grid((unsigned)tiles, 1, (unsigned)p.splits);
  if (!vec) launch_cfg<2, 2, 1, 1, false>(g, trans_a, trans_b, grid, s);
  else if (p.cfg == CFG_128x320) launch_cfg<4, 2, 1, 5, true>(g, trans_a, trans_b, grid, s);
  else if (p.cfg == CFG_128x128) launch_cfg<2, 2, 2, 2, true>(g, trans_a, trans_b, grid, s);
  else launch_cfg<2, 2, 1, 1, true>(g, trans_a, trans_b, grid, s);
  TFGNN_LAUNCH_CHECK();
  if (p.splits > 1) {
    int64_t total = M * N;
    unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}

// ---- grouped variants ---------------------------------------------------------------------------
namespace tfgnn {
constexpr int64_t kMaxAccumulationChainRows = 16384;
static int grouped_splits(int num_groups, int64_t max_rows, int64_t M, int64_t N, int bm, int bn) {
  const int64_t tiles = ceil_div(M, bm) * ceil_div(N, bn) * num_groups;
  int64_t s = ceil_div(512, tiles > 0 ? tiles : 1);
  const int64_t max_by_k = max_rows / (4 * BK);
  if (s > max_by_k) s = max_by_k;
  // accuracy: one fp32 accumulation chain per split - keep it under 16384 rows of K (cfg-5's largest relation has 10^5
  // rows: a single chain leaves sqrt(K) 2^-24 ~ 2e-5 of the largest entry; partial sums of <= 16k rows, added in split
  // order, 6e-6)
  const int64_t by_chain = ceil_div(max_rows, kMaxAccumulationChainRows);
  if (s < by_chain) s = by_chain;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}
}  // namespace tfgnn

extern "C" int tfgnn_gemm_grouped_rows(int trans_b, int num_groups, const int32_t* d_group_offsets,
                                       int64_t max_group_rows, int64_t N, int64_t K, const float* d_A, int64_t lda,
                                       const float* d_B, int64_t ldb, int64_t stride_b, float* d_C, int64_t ldc,
                                       int act, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_groups >= 0 && max_group_rows >= 0 && N >= 0 && K >= 0, "negative size");
  if (num_groups == 0 || max_group_rows == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_group_offsets && d_A && d_B && d_C, "NULL pointer");
  TFGNN_REQUIRE(lda >= K && ldb >= (trans_b ? K : N) && ldc >= N, "bad leading dimension");
  TFGNN_REQUIRE(num_groups <= 65535, "too many groups");
  if (const int nprod = gemm_x3_mode()) {
    int status = TFGNN_OK;
    if (gemm_x3_try_grouped_rows(nprod, trans_b, num_groups, d_group_offsets, max_group_rows, N, K, d_A, lda, d_B, ldb,
                                 stride_b, d_C, ldc, act, (hipStream_t)stream, &status))
      return status;
  }
  const bool vec = K > 0 && operands_vectorisable(0, trans_b, 4, N, K, d_A, lda, d_B, ldb) && (stride_b % 4 == 0);
  GemmPlan p = plan_gemm(max_group_rows, N, K, 0, vec);
  GemmArgs g;
  g.M = max_group_rows; g.N = N; g.K = K;
  g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc;
  g.bias = nullptr; g.act = act; g.accumulate = 0;
  g.k_chunk = ceil_div(K > 0 ? K : 1, BK) * BK; g.splits = 1; g.partial = nullptr;
  g.group_mode = 1; g.group_off = d_group_offsets; g.strideB = stride_b; g.strideC = 0;
  g.wide_store = vec && (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)d_C % 16 == 0);
  g.n_tiles = (unsigned)ceil_div(N, p.bn);
  dim3 grid((unsigned)(ceil_div(max_group_rows, p.bm) * g.n_tiles), (unsigned)num_groups, 1);
  hipStream_t s = (hipStream_t)stream;
  if (!vec) launch_cfg<2, 2, 1, 1, false>(g, 0, trans_b, grid, s);
  else if (p.cfg == CFG_128x320) launch_cfg<4, 2, 1, 5, true>(g, 0, trans_b, grid, s);
  else if (p.cfg == CFG_128x128) launch_cfg<2, 2, 2, 2, true>(g, 0, trans_b, grid, s);
  else launch_cfg<2, 2, 1, 1, true>(g, 0, trans_b, grid, s);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

/* tfgnn_gemm_grouped_rows times act'(saved) (saved [rows, N], indexed like C) in the product's epilogue: the input-gradient
 * product of a per-relation MLP layer with the derivative of the hidden activation below it.  Only the bf16x3 kernels have
 * the epilogue: TFGNN_ERR_UNSUPPORTED otherwise (the caller multiplies with tfgnn_activation_backward). */
extern "C" int tfgnn_gemm_grouped_rows_grad(int trans_b, int num_groups, const int32_t* d_group_offsets, int64_t max_group_rows,
                                            int64_t N, int64_t K, const float* d_A, int64_t lda, const float* d_B, int64_t ldb,
                                            int64_t stride_b, float* d_C, int64_t ldc, int act_of_saved, const float* d_saved,
                                            int64_t ld_saved, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_groups >= 0 && max_group_rows >= 0 && N >= 0 && K >= 0, "negative size");
  if (num_groups == 0 || max_group_rows == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_group_offsets && d_A && d_B && d_C && d_saved, "NULL pointer");
  TFGNN_REQUIRE(lda >= K && ldb >= (trans_b ? K : N) && ldc >= N && ld_saved >= N, "bad leading dimension");
  TFGNN_REQUIRE(num_groups <= 65535, "too many groups");
  TFGNN_REQUIRE(act_of_saved >= TFGNN_ACT_NONE && act_of_saved <= TFGNN_ACT_SIGMOID, "unknown activation %d", act_of_saved);
  if (const int nprod = gemm_x3_mode()) {
    int status = TFGNN_OK;
    if (gemm_x3_try_grouped_rows(nprod, trans_b, num_groups, d_group_offsets, max_group_rows, N, K, d_A, lda, d_B, ldb, stride_b, d_C,
                                 ldc, TFGNN_ACT_NONE, (hipStream_t)stream, &status, act_of_saved, d_saved, ld_saved))
      return status;
  }
  set_error("tfgnn_gemm_grouped_rows_grad: no fused epilogue for this mode / shape");
  return TFGNN_ERR_UNSUPPORTED;
}

extern "C" size_t tfgnn_gemm_grouped_k_workspace_bytes(int num_groups, int64_t max_group_rows, int64_t M, int64_t N) {
  using namespace tfgnn;
  if (num_groups <= 0 || M <= 0 || N <= 0) return 0;
  GemmPlan p = plan_gemm(M, N, 0, 0, true);
  GemmPlan p2 = plan_gemm(M, N, 0, 0, false);
  const int s = std::max(grouped_splits(num_groups, max_group_rows, M, N, p.bm, p.bn),
                         grouped_splits(num_groups, max_group_rows, M, N, p2.bm, p2.bn));
  return s > 1 ? (size_t)s * num_groups * (size_t)M * (size_t)N * 4 : 0;
}

extern "C" int tfgnn_gemm_grouped_k(int num_groups, const int32_t* d_group_offsets, int64_t max_group_rows,
                                    int64_t M, int64_t N, const float* d_A, int64_t lda, const float* d_B,
                                    int64_t ldb, float* d_C, int64_t ldc, int64_t stride_c, void* d_workspace,
                                    size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_groups >= 0 && max_group_rows >= 0 && M >= 0 && N >= 0, "negative size");
  if (num_groups == 0 || M == 0 || N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_group_offsets && d_C && (max_group_rows == 0 || (d_A && d_B)), "NULL pointer");
  TFGNN_REQUIRE(lda >= M && ldb >= N && ldc >= N, "bad leading dimension");
  TFGNN_REQUIRE(num_groups <= 65535, "too many groups");
  if (const int nprod = gemm_x3_mode()) {
    int status = TFGNN_OK;
    if (gemm_x3_try_grouped_k(nprod, num_groups, d_group_offsets, max_group_rows, M, N, d_A, lda, d_B, ldb, d_C, ldc,
                              stride_c, d_workspace, d_workspace ? workspace_bytes : 0, (hipStream_t)stream, &status))
      return status;
  }
  const bool vec = operands_vectorisable(1, 0, M, N, 4, d_A, lda, d_B, ldb);
  GemmPlan p = plan_gemm(M, N, 0, 0, vec);
  int splits = grouped_splits(num_groups, max_group_rows, M, N, p.bm, p.bn);
  if (splits > 1 && (!d_workspace || workspace_bytes < (size_t)splits * num_groups * M * N * 4)) splits = 1;
  GemmArgs g;
  g.M = M; g.N = N; g.K = max_group_rows;
  g.A = d_A; g.lda = lda; g.B = d_B; g.ldb = ldb; g.C = d_C; g.ldc = ldc;
  g.bias = nullptr; g.act = TFGNN_ACT_NONE; g.accumulate = 0;
  g.k_chunk = 0; g.splits = splits; g.partial = (float*)d_workspace;
  g.group_mode = 2; g.group_off = d_group_offsets; g.strideB = 0; g.strideC = stride_c;
  g.wide_store = vec && (N % 4 == 0) && ((splits > 1) || ((ldc % 4 == 0) && (stride_c % 4 == 0) && ((uintptr_t)d_C % 16 == 0))) &&
                 (splits == 1 || (uintptr_t)d_workspace % 16 == 0);
  g.n_tiles = (unsigned)ceil_div(N, p.bn);
  dim3 grid((unsigned)(ceil_div(M, p.bm) * g.n_tiles), (unsigned)num_groups, (unsigned)splits);
  hipStream_t s = (hipStream_t)stream;
  if (!vec) launch_cfg<2, 2, 1, 1, false>(g, 1, 0, grid, s);
  else if (p.cfg == CFG_128x320) launch_cfg<4, 2, 1, 5, true>(g, 1, 0, grid, s);
  else if (p.cfg == CFG_128x128) launch_cfg<2, 2, 2, 2, true>(g, 1, 0, grid, s);
  else launch_cfg<2, 2, 1, 1, true>(g, 1, 0, grid, s);
  TFGNN_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = M * N;
    dim3 rgrid((unsigned)std::min<int64_t>(ceil_div(total, 256), 1024), (unsigned)num_groups);
    hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, s, g);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}
