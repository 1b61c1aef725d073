// fp32 GEMM evaluated on the bf16 matrix cores by exact operand splitting ("bf16x3").
//
// Every fp32 operand value is split, exactly, into three bf16 pieces by truncation
//     x = h + m + l ,  h = top 8 significand bits, m = next 8, l = last 8   (h, m, l are bf16 values)
// and a product a*b is evaluated as the sum of the NPROD largest of the nine piece products
//     h*h, h*m, m*h, h*l, l*h, m*m   (+ m*l, l*m, l*l)
// each of which is EXACT in fp32 (8 x 8 significand bits), accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (16x the rate of v_mfma_f32_32x32x2_f32).  With 6 products the dropped
// terms are below 2^-23 of |a*b| (fp32 unit round-off is 2^-24); with 9 the products are exact and
// only the fp32 accumulation rounds, as in the fp32-MFMA kernel of gemm.hip.  This is an fp32
// computation carried out on bf16 hardware, not a bf16 GEMM: inputs and outputs stay fp32 and the
// parity bound (1e-5 vs the fp64 oracle) is tested for it like for the fp32-MFMA path.
//
// Structure: 8 waves (4 x 2) on a 128 x 320 tile like gemm.hip; BK = 16 (one MFMA k-step), two LDS
// stages.  Operands are split while being staged into LDS: three bf16 planes per operand, rows of
// 16 k + 8 pad = 48 bytes, so that a lane's 8 consecutive k - one MFMA operand - is one aligned,
// conflict-free ds_read_b128.  Operands whose K index is contiguous in memory are staged row-wise;
// operands stored K-major ([K, M] / [K, N]) are transposed in registers (a thread loads 4 k rows of a
// 4-wide column strip and writes 4 LDS rows of 4 k).
// The two waves that share a SIMD run in opposite phases ("ping-pong"): waves 0-3 multiply tile t while
// waves 4-7 split/store tile t+1 and fetch tile t+2, then the roles swap - the matrix pipe always has one
// wave feeding it and the splitting arithmetic runs in its shadow (two barriers per K tile).
#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace tfgnn {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

constexpr int X3_BK = 16;
constexpr int X3_ROW = 24;  // bf16 elements per LDS row (16 + 8 pad): 48 bytes
constexpr int X3_BM = 128, X3_BN = 320, X3_NT = 512;

struct X3Args {
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
  int64_t k_chunk;
  int splits;
  float* partial;
  unsigned n_tiles;
  int debug;  // TFGNN_GEMM_DEBUG probe bits: 1 = no split/store/fetch in the loop, 2 = no multiply (results wrong)
};

// exact 3-way split of an fp32 value into bf16 pieces (upper halves of h, m, l)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = hb;
  m = mb;
  l = __float_as_uint(r2);  // at most 8 significant bits left: its upper half is exact
}
// two bf16 (the upper halves of lo_piece, hi_piece) -> one dword
__device__ __forceinline__ unsigned pack2(unsigned lo_piece, unsigned hi_piece) {
  return __builtin_amdgcn_perm(hi_piece, lo_piece, 0x07060302u);
}
__device__ __forceinline__ void split_store4(float x0, float x1, float x2, float x3, unsigned short* d, int plane_stride) {
  unsigned h[4], m[4], l[4];
  split3(x0, h[0], m[0], l[0]);
  split3(x1, h[1], m[1], l[1]);
  split3(x2, h[2], m[2], l[2]);
  split3(x3, h[3], m[3], l[3]);
  const uint2v vh = {pack2(h[0], h[1]), pack2(h[2], h[3])};
  const uint2v vm = {pack2(m[0], m[1]), pack2(m[2], m[3])};
  const uint2v vl = {pack2(l[0], l[1]), pack2(l[2], l[3])};
  *reinterpret_cast<uint2v*>(d) = vh;
  *reinterpret_cast<uint2v*>(d + plane_stride) = vm;
  *reinterpret_cast<uint2v*>(d + 2 * plane_stride) = vl;
}

// ---- staging ------------------------------------------------------------------------------------
// KC: operand stored [MN_total, K] (k contiguous).  item id = tid + NT p: row id >> 2, k quad id & 3.
template <int MN>
struct StageKC {
  static constexpr int ITEMS = MN * 4;
  static constexpr int NP = (ITEMS + X3_NT - 1) / X3_NT;
  float4 r[NP];
  const float* ptr[NP];
  bool ok[NP];
  __device__ __forceinline__ void init(const float* src, int64_t ld, int64_t mn0, int64_t mn_total, int64_t k_begin, int tid, int) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int id = tid + X3_NT * p;
      const int64_t mn = mn0 + (id >> 2);
      ok[p] = id < ITEMS && mn < mn_total;
      ptr[p] = src + (ok[p] ? mn : 0) * ld + k_begin + (id & 3) * 4;
    }
  }
  __device__ __forceinline__ void load(int64_t k_left, int64_t, int tid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok[p] && ((tid + X3_NT * p) & 3) * 4 < k_left) v = *reinterpret_cast<const float4*>(ptr[p]);
      r[p] = v;
      ptr[p] += X3_BK;
    }
  }
  __device__ __forceinline__ void store(unsigned short* planes, int tid) const {  // planes: [3][MN][X3_ROW]
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int id = tid + X3_NT * p;
      if (id < ITEMS)
        split_store4(r[p].x, r[p].y, r[p].z, r[p].w, planes + (id >> 2) * X3_ROW + (id & 3) * 4, MN * X3_ROW);
    }
  }
};

// KM: operand stored [K, MN_total] (mn contiguous).  item id = tid - first: k group id & 3 (4 k rows), column
// strip id >> 2 (4 mn); a thread loads the 4 x 4 block and writes 4 LDS rows of 4 k each.
template <int MN>
struct StageKM {
  static constexpr int ITEMS = MN;
  float4 r[4];
  const float* ptr;
  bool ok, mine;
  int id;
  __device__ __forceinline__ void init(const float* src, int64_t ld, int64_t mn0, int64_t mn_total, int64_t k_begin, int tid, int first) {
    id = tid - first;
    mine = id >= 0 && id < ITEMS;
    const int64_t mn = mn0 + (id >> 2) * 4;
    ok = mine && mn < mn_total;  // mn_total % 4 == 0: a strip is fully in or out
    ptr = src + (k_begin + (id & 3) * 4) * ld + (ok ? mn : 0);
  }
  __device__ __forceinline__ void load(int64_t k_left, int64_t ld, int) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok && (id & 3) * 4 + i < k_left) v = *reinterpret_cast<const float4*>(ptr + i * ld);
      r[i] = v;
    }
    ptr += (int64_t)X3_BK * ld;
  }
  __device__ __forceinline__ void store(unsigned short* planes, int) const {
    if (!mine) return;
    unsigned short* d = planes + ((id >> 2) * 4) * X3_ROW + (id & 3) * 4;
    split_store4(r[0].x, r[1].x, r[2].x, r[3].x, d, MN * X3_ROW);
    split_store4(r[0].y, r[1].y, r[2].y, r[3].y, d + X3_ROW, MN * X3_ROW);
    split_store4(r[0].z, r[1].z, r[2].z, r[3].z, d + 2 * X3_ROW, MN * X3_ROW);
    split_store4(r[0].w, r[1].w, r[2].w, r[3].w, d + 3 * X3_ROW, MN * X3_ROW);
  }
};

template <int MN, bool KM>
struct StageSel {
  using type = StageKC<MN>;
};
template <int MN>
struct StageSel<MN, true> {
  using type = StageKM<MN>;
};

__device__ __forceinline__ bf16x8 frag8(const unsigned short* p) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4v*>(p));
}

template <int NPROD>
__device__ __forceinline__ floatx16 mfma_group(floatx16 c, bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl) {
  // smallest terms first
  if (NPROD >= 9) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, c, 0, 0, 0);
  if (NPROD >= 8) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
  return c;
}

// A_KM: A stored [K, M] (trans_a) ; B_KM: B stored [K, N] (no trans_b)
template <bool A_KM, bool B_KM, int NPROD>
__global__ void __launch_bounds__(X3_NT) gemm_x3_kernel(X3Args g) {
  constexpr int WN_ = 2, TN = 5;
  using SA = typename StageSel<X3_BM, A_KM>::type;
  using SB = typename StageSel<X3_BN, B_KM>::type;
  constexpr int PLANE_A = X3_BM * X3_ROW, PLANE_B = X3_BN * X3_ROW;
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);  // ushorts
  constexpr int PATCH_FLOATS = (X3_NT / 64) * 32 * 36;
  static_assert(2 * STAGE * 2 >= PATCH_FLOATS * 4, "epilogue patch must fit in the stage buffers");
  __shared__ __attribute__((aligned(16))) unsigned short lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;  // waves 0-3 (one per SIMD) and 4-7 form the two phase groups
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  const int64_t m0 = (int64_t)(blockIdx.x / g.n_tiles) * X3_BM;
  const int64_t n0 = (int64_t)(blockIdx.x % g.n_tiles) * X3_BN;
  const int64_t k_begin = (int64_t)blockIdx.z * g.k_chunk;
  const int64_t k_end = k_begin + g.k_chunk < g.K ? k_begin + g.k_chunk : g.K;

  floatx16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  if (k_begin < k_end) {
    SA sa;
    SB sb;
    // K-major operands: one item per thread; A items on threads [0,128), B items on [128,448) when both are K-major
    sa.init(g.A, g.lda, m0, g.M, k_begin, tid, 0);
    sb.init(g.B, g.ldb, n0, g.N, k_begin, tid, A_KM ? X3_BM : 0);
    const int a_frag = (wm * 32 + li) * X3_ROW + lk * 8;
    const int b_frag = 3 * PLANE_A + (wn * TN * 32 + li) * X3_ROW + lk * 8;

    auto multiply = [&](int stage) {
      const unsigned short* base = lds + stage * STAGE;
      const bf16x8 ah = frag8(base + a_frag);
      const bf16x8 am = frag8(base + a_frag + PLANE_A);
      const bf16x8 al = frag8(base + a_frag + 2 * PLANE_A);
      bf16x8 bh = frag8(base + b_frag);
      bf16x8 bm = frag8(base + b_frag + PLANE_B);
      bf16x8 bl = frag8(base + b_frag + 2 * PLANE_B);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8 nh = bh, nm = bm, nl = bl;
        if (j + 1 < TN) {  // next column tile's operands fly under this tile's MFMAs
          const unsigned short* nb = base + b_frag + (j + 1) * 32 * X3_ROW;
          nh = frag8(nb);
          nm = frag8(nb + PLANE_B);
          nl = frag8(nb + 2 * PLANE_B);
        }
        acc[j] = mfma_group<NPROD>(acc[j], ah, am, al, bh, bm, bl);
        bh = nh; bm = nm; bl = nl;
      }
    };
    auto stage_next = [&](int stage, int64_t k_next_left) {  // registers (tile t+1) -> LDS, then fetch tile t+2
      unsigned short* base = lds + stage * STAGE;
      sa.store(base, tid);
      sb.store(base + 3 * PLANE_A, tid);
      sa.load(k_next_left, g.lda, tid);
      sb.load(k_next_left, g.ldb, tid);
    };

    // Phase schedule (one barrier between phases; M = multiply tile t, S = split/store a later tile + fetch):
    //   waves 0-3:  M0 S  M1 S  M2 ...     S stores tile t+1 into stage (t+1)&1, fetches tile t+2
    //   waves 4-7:  S' M0 S  M1 S  ...     S stores tile t+2 into stage t&1,     fetches tile t+3
    // i.e. the second group runs the same loop one phase late and stages one tile further ahead; every
    // stage is complete a full phase before anyone multiplies from it and is overwritten only after both
    // groups have multiplied from it.
    sa.load(k_end - k_begin, g.lda, tid);
    sb.load(k_end - k_begin, g.ldb, tid);
    stage_next(0, k_end - k_begin - X3_BK);  // tile 0 -> stage 0; registers <- tile 1
    __syncthreads();
    if (grp == 1) {
      stage_next(1, k_end - k_begin - 2 * X3_BK);  // S': tile 1 -> stage 1; registers <- tile 2
      __syncthreads();
    }
    int cur = 0;
    for (int64_t k0 = k_begin; k0 < k_end; k0 += X3_BK, cur ^= 1) {
      // sched_barrier: keep the splitting arithmetic (and the wait for its global loads) out of the multiply phase
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (!(g.debug & 2)) multiply(cur);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (!(g.debug & 1)) stage_next(cur ^ 1 ^ grp, k_end - k0 - (2 + grp) * X3_BK);
      __syncthreads();
    }
    if (grp == 0) __syncthreads();  // the second group ran one more phase
  }

  // epilogue (same wide-store scheme as gemm.hip): C/D layout col = lane & 31, row = (r&3) + 8 (r>>2) + 4 (lane>>5)
  const bool split = g.splits > 1;
  float* outp = split ? g.partial + (int64_t)blockIdx.z * g.M * g.N : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
  constexpr int PS = 36;
  float* patch = reinterpret_cast<float*>(lds) + wave * 32 * PS;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PS + li] = acc[j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t col = n0 + (wn * TN + j) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pr = (lane >> 3) + 8 * q;
      const int64_t row = m0 + wm * 32 + pr;
      float4 v = *reinterpret_cast<const float4*>(patch + pr * PS + (lane & 7) * 4);
      if (row < g.M && col < g.N) {
        float* dst = outp + row * ldo + col;
        if (!split) {
          if (g.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(g.bias + col);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          v.x = act_apply(g.act, v.x); v.y = act_apply(g.act, v.y);
          v.z = act_apply(g.act, v.z); v.w = act_apply(g.act, v.w);
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

__global__ void __launch_bounds__(256) x3_splitk_reduce_kernel(X3Args g) {
  const int64_t total = g.M * g.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += g.partial[(int64_t)z * total + i];
    const int64_t row = i / g.N, col = i - row * g.N;
    if (g.bias) s += g.bias[col];
    s = act_apply(g.act, s);
    float* c = g.C + row * g.ldc + col;
    if (g.accumulate) s += *c;
    *c = s;
  }
}

template <bool A_KM, bool B_KM>
static void launch_x3(const X3Args& g, dim3 grid, int nprod, hipStream_t s) {
  if (nprod >= 9) hipLaunchKernelGGL((gemm_x3_kernel<A_KM, B_KM, 9>), grid, dim3(X3_NT), 0, s, g);
  else hipLaunchKernelGGL((gemm_x3_kernel<A_KM, B_KM, 6>), grid, dim3(X3_NT), 0, s, g);
}

// 0 = off (fp32 MFMA), 6 / 9 = number of piece products.  Initialised from TFGNN_GEMM_MODE
// (fp32 | bf16x3 | bf16x3_9), changed at run time by tfgnn_gemm_set_mode().
static int mode_from_env() {
  const char* e = getenv("TFGNN_GEMM_MODE");
  if (!e) return 0;
  if (!strcmp(e, "bf16x3") || !strcmp(e, "bf16x3_6")) return 6;
  if (!strcmp(e, "bf16x3_9")) return 9;
  return 0;
}
static int g_x3_mode = -1;
int gemm_x3_mode() {
  if (g_x3_mode < 0) g_x3_mode = mode_from_env();
  return g_x3_mode;
}
int gemm_x3_set_mode(int mode) {
  const int prev = gemm_x3_mode();
  g_x3_mode = mode;
  return prev;
}

// returns 1 if it took the call, 0 if the shape / layout is not covered (caller falls back to gemm.hip)
int gemm_x3_try(int nprod, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int accumulate,
                void* workspace, size_t workspace_bytes, hipStream_t s, int* status) {
  *status = TFGNN_OK;
  // the 128 x 320 tile only (N = 320 family), 16-byte aligned operands, supported layout pairs:
  //   NN (A [M,K], B [K,N]), NT (A [M,K], B [N,K]), TN (A [K,M], B [K,N])
  if (trans_a && trans_b) return 0;
  if (!(N % 320 == 0) || K < 64 || M < 1) return 0;
  const bool a16 = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
  const bool b16 = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
  if (!a16 || !b16 || K % 4 != 0 || (trans_a && M % 4 != 0) || (ldc % 4 != 0) || ((uintptr_t)C % 16 != 0)) return 0;
  if (bias && (uintptr_t)bias % 16 != 0) return 0;
  X3Args g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.act = act; g.accumulate = accumulate;
  g.n_tiles = (unsigned)ceil_div(N, 320);
  const int64_t tiles = ceil_div(M, 128) * (int64_t)g.n_tiles;
  g.splits = 1;
  g.k_chunk = ceil_div(K, X3_BK) * X3_BK;
  if (tiles < 192 && K >= 1024 && workspace) {
    int64_t want = 256 / tiles;
    const int64_t max_by_k = K / 128;
    if (want > max_by_k) want = max_by_k;
    if (want > 64) want = 64;
    const int64_t max_by_ws = (int64_t)(workspace_bytes / ((size_t)(M * N) * 4 + 1));
    if (want > max_by_ws) want = max_by_ws;
    if (want > 1 && (uintptr_t)workspace % 16 == 0) {
      g.k_chunk = ceil_div(ceil_div(K, want), X3_BK) * X3_BK;
      g.splits = (int)ceil_div(K, g.k_chunk);
    }
  }
  g.partial = (float*)workspace;
  {
    static const int dbg = [] { const char* e = getenv("TFGNN_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
    g.debug = dbg;
  }
  dim3 grid((unsigned)tiles, 1, (unsigned)g.splits);
  if (!trans_a && !trans_b) launch_x3<false, true>(g, grid, nprod, s);       // B stored [K, N]: K-major
  else if (!trans_a && trans_b) launch_x3<false, false>(g, grid, nprod, s);  // B stored [N, K]
  else launch_x3<true, true>(g, grid, nprod, s);                             // A stored [K, M], B [K, N]
  if (hipGetLastError() != hipSuccess) {
    set_error("bf16x3 GEMM launch failed");
    *status = TFGNN_ERR_HIP;
    return 1;
  }
  if (g.splits > 1) {
    const int64_t total = M * N;
    hipLaunchKernelGGL(x3_splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096)), dim3(256), 0, s, g);
  }
  return 1;
}

}  // namespace tfgnn
