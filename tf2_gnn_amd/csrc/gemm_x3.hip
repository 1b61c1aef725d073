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
// Structure: a 128 x 320 output tile per 512-thread workgroup, BK = 16 (one MFMA k-step), a ring of three
// LDS stages.  Operands are split while being staged into LDS: three bf16 planes per operand, unpadded
// 32-byte rows (16 k) with the two 16-byte halves XOR-swizzled by bit 3 of the row, so that a lane's 8
// consecutive k - one MFMA operand - is one aligned, conflict-free ds_read_b128.  Operands whose K index is
// contiguous in memory are staged row-wise; operands stored K-major ([K, M] / [K, N]) are transposed in
// registers (a thread loads 4 k rows of a 4-wide column strip and writes 4 LDS rows of 4 k).
// Two kernels share this layout (launch_x3 picks): gemm_x3s_kernel (waves 0-3 multiply, waves 4-7 stage)
// and gemm_x3p_kernel (all eight waves do both, software-pipelined).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "sp16.hpp"

namespace tfgnn {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

constexpr int X3_BK = 16;
constexpr int X3_BM = 128, X3_NT = 512;

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
  // plain form, XCD-aware order: a 1-D grid of 8 * per_xcd workgroups; 0 = (x, z) grid as launched
  unsigned per_xcd, tiles_x;
  // gradient epilogue (tfgnn_gemm_grad_epilogue): C = (A B) * mul * act'(saved); NULL = factor absent
  const float* mul;
  int64_t ld_mul;
  const float* saved;
  int64_t ld_saved;
  int dact;
  // grouped forms (tfgnn_gemm_grouped_rows / _k): blockIdx.y = group
  int group_mode;  // 0 plain, 1 rows of A / C grouped (own B per group), 2 K range grouped (own C per group)
  const int32_t* group_off;
  int64_t strideB, strideC;
  // GRU epilogue (tfgnn_gemm_gru, kernel variant EPI = 1): the product is mx of a GRUCell with the columns of B
  // regrouped per 192-wide tile as [z | r | h] of the same 64 units; C receives h' [M, H]
  const float* gru_mh;   // [M, 3H]: h @ recurrent_kernel + recurrent bias
  const float* gru_h;    // [M, H]: previous state
  float* gru_gates;      // [M, 3H] z | r | c for the backward pass, or NULL
  int gru_H;
  // streaming kernel only (tfgnn_gemm_gathered): row m of the product reads row a_index[m] of A ([a_rows, lda]); NULL = row m
  const int32_t* a_index;
  int64_t a_rows;
};

__device__ __forceinline__ float4 grad_epilogue(const X3Args& g, float4 v, int64_t row, int64_t col) {
  if (g.mul) {
    const float4 m = *reinterpret_cast<const float4*>(g.mul + row * g.ld_mul + col);
    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
  }
  if (g.saved) {
    const float4 sv = *reinterpret_cast<const float4*>(g.saved + row * g.ld_saved + col);
    v.x *= act_grad(g.dact, sv.x); v.y *= act_grad(g.dact, sv.y);
    v.z *= act_grad(g.dact, sv.z); v.w *= act_grad(g.dact, sv.w);
  }
  return v;
}

// exact 3-way split of an fp32 value into bf16 pieces (upper halves of h, m, l)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xffff0000u;
  // +-inf / nan: the class stays in h, the lower pieces are 0 (inf - inf would make them NaN and turn a product that
  // is +-inf in fp32 into NaN); one v_cmp_class + one v_cndmask per element
  const float r1 = __builtin_isfinite(x) ? x - __uint_as_float(hb) : 0.f;
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

// =====================================================================================================
// Shared geometry.  The output tile is 128 x BN, BN = 64 * TN (TN = 5, 4, 2 -> 320, 256, 128 columns: the hidden
// sizes the layers come with), BK = 16, three LDS stages of unpadded, XOR-swizzled 32-byte rows.
// Plane layout (ushort units): [A rows 0..127 | B rows 128..128+BN-1 | trash], row = 16 k; 16-byte half h of row
// r is stored at half h ^ ((r >> 3) & 1).  The staging code is branch-free (clamped addresses, a per-plane trash
// area for threads without an item).
constexpr int P_ROW = 16;
constexpr int P_TRASH = 320;
template <int TN>
struct Geo {
  static constexpr int BN = 64 * TN;
  static constexpr int TRASH_OFF = (X3_BM + BN) * P_ROW;
  static constexpr int PLANE = TRASH_OFF + P_TRASH;  // TN = 5: 7488 ushorts
  static constexpr int STAGE = 3 * PLANE;            // TN = 5: 44928 bytes
};

__device__ __forceinline__ int swz_off(int row, int kq) {  // ushort offset of k quad kq (4 k) of plane row `row`
  return row * P_ROW + ((((kq >> 1) ^ (row >> 3)) & 1) << 3) + ((kq & 1) << 2);
}

// K-contiguous operand: item = (row, k quad); one float4
template <int TN>
struct SlotKC {
  const float* ptr;
  const float* ptr0;
  int lds_off;
  int kofs;
  float4 r;
  __device__ __forceinline__ void init(const float* src, int64_t ld, int64_t mn0, int64_t mn_total, int64_t k_begin,
                                       int id, int items, int row_base, int lane) {
    const bool valid = id < items;
    const int row = valid ? id >> 2 : 0, kq = valid ? id & 3 : 0;
    int64_t mn = mn0 + row;
    if (mn > mn_total - 1) mn = mn_total - 1;  // rows past the edge: any valid row (their outputs are never stored)
    kofs = kq * 4;
    ptr0 = ptr = src + mn * ld + k_begin + kofs;
    lds_off = valid ? swz_off(row_base + row, kq) : Geo<TN>::TRASH_OFF + lane * 4;
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) {
    if (MASKED) {
      const bool inb = kofs < k_left;
      const float4 v = *reinterpret_cast<const float4*>(inb ? ptr : ptr0);
      r.x = inb ? v.x : 0.f; r.y = inb ? v.y : 0.f; r.z = inb ? v.z : 0.f; r.w = inb ? v.w : 0.f;
    } else {
      r = *reinterpret_cast<const float4*>(ptr);
    }
    ptr += adv * X3_BK;
  }
  __device__ __forceinline__ void skip() { ptr += X3_BK; }
  __device__ __forceinline__ void store(unsigned short* stage) const {
    split_store4(r.x, r.y, r.z, r.w, stage + lds_off, Geo<TN>::PLANE);
  }
};

// K-major operand: item = (4-column strip, group of 4 k rows); four float4, transposed in registers
template <int TN>
struct SlotKM {
  const float* ptr;
  const float* ptr0;
  int64_t ld;
  int lds_off;
  int kofs;
  float4 r[4];
  __device__ __forceinline__ void init(const float* src, int64_t ld_, int64_t mn0, int64_t mn_total, int64_t k_begin,
                                       int id, int items, int row_base, int lane) {
    const bool valid = id >= 0 && id < items;
    const int kg = valid ? id & 3 : 0, strip = valid ? id >> 2 : 0;
    int64_t mn = mn0 + strip * 4;
    if (mn > mn_total - 4) mn = mn_total - 4;  // mn_total % 4 == 0
    ld = ld_;
    kofs = kg * 4;
    ptr0 = ptr = src + (k_begin + kofs) * ld + mn;
    lds_off = valid ? swz_off(row_base + strip * 4, kg) : Geo<TN>::TRASH_OFF + lane * 4;
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MASKED) {
        const bool inb = kofs + i < k_left;
        const float4 v = *reinterpret_cast<const float4*>(inb ? ptr + i * ld : ptr0);
        r[i].x = inb ? v.x : 0.f; r[i].y = inb ? v.y : 0.f; r[i].z = inb ? v.z : 0.f; r[i].w = inb ? v.w : 0.f;
      } else {
        r[i] = *reinterpret_cast<const float4*>(ptr + i * ld);
      }
    }
    ptr += (int64_t)(adv * X3_BK) * ld;
  }
  __device__ __forceinline__ void skip() { ptr += (int64_t)X3_BK * ld; }
  template <int J>
  __device__ __forceinline__ void store_row(unsigned short* stage) const {
    // rows strip*4 + j share (row >> 3) & 1: the swizzle term is the same for the four rows
    unsigned short* d = stage + lds_off + J * P_ROW;
    if (J == 0) split_store4(r[0].x, r[1].x, r[2].x, r[3].x, d, Geo<TN>::PLANE);
    if (J == 1) split_store4(r[0].y, r[1].y, r[2].y, r[3].y, d, Geo<TN>::PLANE);
    if (J == 2) split_store4(r[0].z, r[1].z, r[2].z, r[3].z, d, Geo<TN>::PLANE);
    if (J == 3) split_store4(r[0].w, r[1].w, r[2].w, r[3].w, d, Geo<TN>::PLANE);
  }
  __device__ __forceinline__ void store(unsigned short* stage) const {
    store_row<0>(stage); store_row<1>(stage); store_row<2>(stage); store_row<3>(stage);
  }
};

// =====================================================================================================
// Pipelined kernel (both operands K-major: the weight-gradient products): all 8 waves (4 x 2, wave tile
// 32 x 32 TN) split / store and multiply, software-pipelined, one barrier per K tile.  Iteration t multiplies
// tile t (stage t%3) while it splits/stores tile t+2 into stage (t+2)%3 and fetches tile t+3 into registers; the
// first operands of tile t+1 (complete since the previous barrier) are read before the barrier, so the matrix
// pipe restarts immediately after it.  One K-major item per thread: A strips on threads [0,128), B strips on
// [128, 128+BN).
template <int TN>
struct StagerKM {
  SlotKM<TN> s;
  __device__ __forceinline__ void init(const X3Args& g, int64_t m0, int64_t n0, int64_t k_begin, int tid) {
    const int lane = tid & 63;
    if (tid < X3_BM) s.init(g.A, g.lda, m0, g.M, k_begin, tid, X3_BM, 0, lane);
    else s.init(g.B, g.ldb, n0, g.N, k_begin, tid - X3_BM, Geo<TN>::BN, X3_BM, lane);
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) { s.template load<MASKED>(k_left, adv); }
  __device__ __forceinline__ void skip() { s.skip(); }
  __device__ __forceinline__ void store(unsigned short* stage) const { s.store(stage); }
  // piece J of the split/store work; the last MFMA chunk (LAST) takes whatever pieces are left
  template <int J, bool LAST>
  __device__ __forceinline__ void store_piece(unsigned short* stage) const {
    if (J < 4) s.template store_row<(J < 4 ? J : 0)>(stage);
    if (LAST) {
      if (J + 1 < 4) s.template store_row<(J + 1 < 4 ? J + 1 : 0)>(stage);
      if (J + 2 < 4) s.template store_row<(J + 2 < 4 ? J + 2 : 0)>(stage);
      if (J + 3 < 4) s.template store_row<(J + 3 < 4 ? J + 3 : 0)>(stage);
    }
  }
};

// K-contiguous operand pair (NT) for the pipelined kernel: 512 A items + 4 BN B items of one float4 over the 512
// threads - one A slot and NB B slots per thread, each slot one piece of the split / store work.
template <int TN>
struct StagerKC {
  static constexpr int NB = (Geo<TN>::BN * 4 + X3_NT - 1) / X3_NT;
  static constexpr int P = 1 + NB;
  SlotKC<TN> a, b[NB];
  __device__ __forceinline__ void init(const X3Args& g, int64_t m0, int64_t n0, int64_t k_begin, int tid) {
    const int lane = tid & 63;
    a.init(g.A, g.lda, m0, g.M, k_begin, tid, X3_BM * 4, 0, lane);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].init(g.B, g.ldb, n0, g.N, k_begin, tid + X3_NT * i, Geo<TN>::BN * 4, X3_BM, lane);
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) {
    a.template load<MASKED>(k_left, adv);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].template load<MASKED>(k_left, adv);
  }
  __device__ __forceinline__ void skip() {
    a.skip();
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].skip();
  }
  __device__ __forceinline__ void store(unsigned short* stage) const {
    a.store(stage);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].store(stage);
  }
  template <int Q>
  __device__ __forceinline__ void piece(unsigned short* stage) const {
    if constexpr (Q == 0) a.store(stage);
    else if constexpr (Q < P) b[Q - 1].store(stage);
  }
  template <int J, bool LAST>
  __device__ __forceinline__ void store_piece(unsigned short* stage) const {
    piece<J>(stage);
    if (LAST) {
      piece<J + 1>(stage);
      piece<J + 2>(stage);
      piece<J + 3>(stage);
    }
  }
};

// Which (tile, K split) does this workgroup own?  Workgroups are dealt to the 8 XCDs round-robin by linear id and each
// XCD has its own L2; with the XCD-aware order one XCD walks consecutive logical ids - the column tiles of one row
// tile (they share the A rows), then the next row tile, then the next K split - so operand rows shared by neighbouring
// tiles are fetched over the fabric once, not once per XCD.
struct X3Tile {
  unsigned x, z;
  bool valid;
};
__device__ __forceinline__ X3Tile x3_tile(const X3Args& g) {
  if (g.per_xcd == 0) return {blockIdx.x, blockIdx.z, true};
  const unsigned logical = (blockIdx.x & 7u) * g.per_xcd + (blockIdx.x >> 3);
  if (logical >= g.tiles_x * (unsigned)g.splits) return {0u, 0u, false};
  return {logical % g.tiles_x, logical / g.tiles_x, true};
}

template <int NPROD, int TN, bool NT = false>
__global__ void __launch_bounds__(X3_NT) gemm_x3p_kernel(X3Args g) {
  constexpr int PLANE = Geo<TN>::PLANE, STAGE = Geo<TN>::STAGE, BN = Geo<TN>::BN;
  constexpr int PATCH_FLOATS = (X3_NT / 64) * 32 * 36;
  static_assert(3 * STAGE * 2 >= PATCH_FLOATS * 4, "epilogue patch must fit in the stage buffers");
  __shared__ __attribute__((aligned(16))) unsigned short lds[3 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const X3Tile tile = x3_tile(g);
  if (!tile.valid) return;  // uniform for the whole workgroup
  const unsigned bx = tile.x, bz = tile.z;
  const int64_t m0 = (int64_t)(bx / g.n_tiles) * X3_BM;
  const int64_t n0 = (int64_t)(bx % g.n_tiles) * BN;
  int64_t k_begin = (int64_t)bz * g.k_chunk;
  int64_t k_end = k_begin + g.k_chunk < g.K ? k_begin + g.k_chunk : g.K;
  float* Cp = g.C;
  int64_t partial_slab = bz;
  if (g.group_mode == 2) {  // the K range is the group's rows; every group has its own output
    const int64_t gb = g.group_off[blockIdx.y], ge = g.group_off[blockIdx.y + 1];
    const int64_t per = (ge - gb + g.splits - 1) / g.splits;
    const int64_t chunk = (per + X3_BK - 1) / X3_BK * X3_BK;
    k_begin = gb + (int64_t)bz * chunk;
    k_end = k_begin + chunk < ge ? k_begin + chunk : ge;
    Cp += (int64_t)blockIdx.y * g.strideC;
    partial_slab = (int64_t)blockIdx.y * g.splits + bz;
  }

  floatx16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  if (k_begin < k_end) {
    using Stager = typename std::conditional<NT, StagerKC<TN>, StagerKM<TN>>::type;
    Stager st;
    st.init(g, m0, n0, k_begin, tid);
    const int half = (lk ^ (li >> 3)) & 1;
    const int a_frag = (wm * 32 + li) * P_ROW + half * 8;
    const int b_frag = (X3_BM + wn * TN * 32 + li) * P_ROW + half * 8;
    const int64_t k_len = k_end - k_begin;
    const int T = (int)((k_len + X3_BK - 1) / X3_BK);

    // prologue: tiles 0, 1 -> stages 0, 1; registers <- tile 2
    st.template load<true>(k_len, 1);
    st.store(lds);
    st.template load<true>(k_len - X3_BK, 1);
    st.store(lds + STAGE);
    st.template load<true>(k_len - 2 * X3_BK, 1);
    __syncthreads();
    bf16x8 ah = frag8(lds + a_frag), am = frag8(lds + a_frag + PLANE), al = frag8(lds + a_frag + 2 * PLANE);
    bf16x8 bh = frag8(lds + b_frag), bm = frag8(lds + b_frag + PLANE), bl = frag8(lds + b_frag + 2 * PLANE);

    int s_cur = 0, s_nxt = STAGE, s_st = 2 * STAGE;  // stage of tile t, t+1, t+2
    // One iteration = TN chunks (one per 32-column tile of the wave): the split/store of one piece of tile
    // t+2, the operand reads of the next column tile, 6 (9) MFMAs.  sched_barrier keeps the chunks apart so
    // that the splitting arithmetic stays spread over the iteration; `from` holds tile t+2, the fetch of
    // tile t+3 goes to the other register set `to` first and has the whole iteration to land.
    auto iteration = [&](auto masked, int64_t left3, const Stager& from, Stager& to) {
      constexpr bool MASKED = decltype(masked)::value;
      to.template load<MASKED>(left3, 2);
      __builtin_amdgcn_sched_barrier(0);
      unsigned short* dst = lds + s_st;
      const unsigned short* cur = lds + s_cur;
      const unsigned short* nxs = lds + s_nxt;
      auto chunk = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        if constexpr (J < TN) {
          from.template store_piece<J, J == TN - 1>(dst);
          bf16x8 nh, nm, nl;
          if (J + 1 < TN) {
            const unsigned short* nb = cur + b_frag + (J + 1) * 32 * P_ROW;
            nh = frag8(nb); nm = frag8(nb + PLANE); nl = frag8(nb + 2 * PLANE);
          } else {  // first operands of tile t+1 (complete since the previous barrier)
            nh = frag8(nxs + b_frag); nm = frag8(nxs + b_frag + PLANE); nl = frag8(nxs + b_frag + 2 * PLANE);
          }
          acc[J] = mfma_group<NPROD>(acc[J], ah, am, al, bh, bm, bl);
          bh = nh; bm = nm; bl = nl;
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      chunk(std::integral_constant<int, 0>{});
      chunk(std::integral_constant<int, 1>{});
      chunk(std::integral_constant<int, 2>{});
      chunk(std::integral_constant<int, 3>{});
      chunk(std::integral_constant<int, 4>{});
      ah = frag8(nxs + a_frag); am = frag8(nxs + a_frag + PLANE); al = frag8(nxs + a_frag + 2 * PLANE);
      __syncthreads();
      const int s = s_cur; s_cur = s_nxt; s_nxt = s_st; s_st = s;
    };
    Stager st2 = st;  // two register sets, each fetching every second tile: st2 tiles 3, 5, ...;
    st.skip();              // st (holding tile 2) continues with 4, 6, ...
    int t = 0;
    const int t_fast = (int)(k_len / X3_BK) - 3;  // iterations whose fetched tile t+3 is a full tile
    for (; t + 1 < t_fast; t += 2) {
      iteration(std::false_type{}, 0, st, st2);
      iteration(std::false_type{}, 0, st2, st);
    }
    for (; t < T; t += 2) {
      iteration(std::true_type{}, k_len - (int64_t)(t + 3) * X3_BK, st, st2);
      if (t + 1 < T) iteration(std::true_type{}, k_len - (int64_t)(t + 4) * X3_BK, st2, st);
    }
  }

  const bool split = g.splits > 1;
  float* outp = split ? g.partial + partial_slab * g.M * g.N : Cp;
  const int64_t ldo = split ? g.N : g.ldc;
  if constexpr (NT) {
    // compact epilogue (as in the specialised kernel): the four waves of one column half put their 32 x 32 TN blocks
    // into LDS, all eight waves stream them out with bias / activation / gradient factors / accumulate; two rounds
    constexpr int EP_COLS = 32 * TN, EP_LD = EP_COLS + 4, ROW4 = EP_COLS / 4;
    static_assert(3 * STAGE * 2 >= 4 * 32 * EP_LD * 4, "epilogue blocks must fit in the stage buffers");
    float* ep = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (wn == r) {
        float* blk = ep + wm * 32 * EP_LD;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) blk[((q & 3) + 8 * (q >> 2) + 4 * lk) * EP_LD + j * 32 + li] = acc[j][q];
      }
      __syncthreads();
#pragma unroll 1
      for (int f = tid; f < 4 * 32 * ROW4; f += X3_NT) {
        const int b = f / (32 * ROW4), rem = f - b * (32 * ROW4);
        const int row = rem / ROW4, c4 = rem - row * ROW4;
        const int64_t grow = m0 + b * 32 + row;
        const int64_t gcol = n0 + r * EP_COLS + c4 * 4;
        if (grow < g.M) {
          float4 v = *reinterpret_cast<const float4*>(ep + (b * 32 + row) * EP_LD + c4 * 4);
          float* dst = outp + grow * ldo + gcol;
          if (!split) {
            if (g.bias) {
              const float4 b4 = *reinterpret_cast<const float4*>(g.bias + gcol);
              v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
            }
            v.x = act_apply(g.act, v.x); v.y = act_apply(g.act, v.y);
            v.z = act_apply(g.act, v.z); v.w = act_apply(g.act, v.w);
            v = grad_epilogue(g, v, grow, gcol);
            if (g.accumulate) {
              const float4 c4v = *reinterpret_cast<const float4*>(dst);
              v.x += c4v.x; v.y += c4v.y; v.z += c4v.z; v.w += c4v.w;
            }
          }
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
      __syncthreads();
    }
    return;
  }
  // epilogue: wave-private LDS patch per 32 x 32 tile, 16-byte stores
  constexpr int PS = 36;
  float* patch = reinterpret_cast<float*>(lds) + wave * 32 * PS;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lk) * PS + li] = acc[j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
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

// =====================================================================================================
// Specialised kernel (A K-contiguous): waves 0-3 (one per SIMD) only multiply - each owns a 64 x 32 TN block of
// the 128 x 64 TN tile (2 x TN MFMA tiles, 160 accumulator registers at TN = 5) - and waves 4-7 (their SIMD
// partners) only fetch, split and store.  Per fp32 MAC the splitting costs ~5.5 VALU issue slots per element
// staged; an MFMA leaves room for about five other instructions of the same SIMD, so mixing the two in one wave
// (the pipelined kernel above) is issue-bound.  Here the multiplying wave issues one MFMA per 32 cycles plus a
// ds_read every third MFMA, and its partner's VALU work runs beside it.  Same three-stage LDS ring and one
// barrier per K tile.
constexpr int S_PT = 256;  // producer threads

template <bool B_KM, int TN>
struct Producer;
template <int TN>
struct Producer<false, TN> {  // NT: A 512 items, B 256 TN items of one float4 -> 2 + TN per producer thread
  SlotKC<TN> a[2], b[TN];
  __device__ __forceinline__ void init(const X3Args& g, int64_t m0, int64_t n0, int64_t k_begin, int ptid) {
    const int lane = ptid & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].init(g.A, g.lda, m0, g.M, k_begin, ptid + S_PT * i, X3_BM * 4, 0, lane);
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i].init(g.B, g.ldb, n0, g.N, k_begin, ptid + S_PT * i, Geo<TN>::BN * 4, X3_BM, lane);
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].template load<MASKED>(k_left, adv);
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i].template load<MASKED>(k_left, adv);
  }
  __device__ __forceinline__ void skip() {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].skip();
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i].skip();
  }
  __device__ __forceinline__ void store(unsigned short* stage) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].store(stage);
#pragma unroll
    for (int i = 0; i < TN; ++i) b[i].store(stage);
  }
};
template <int TN>
struct Producer<true, TN> {  // NN: A 512 K-contiguous items (2 per thread), B 64 TN K-major items (NB slots per thread)
  static constexpr int NB = (Geo<TN>::BN + S_PT - 1) / S_PT;
  SlotKC<TN> a[2];
  SlotKM<TN> b[NB];
  __device__ __forceinline__ void init(const X3Args& g, int64_t m0, int64_t n0, int64_t k_begin, int ptid) {
    const int lane = ptid & 63;
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].init(g.A, g.lda, m0, g.M, k_begin, ptid + S_PT * i, X3_BM * 4, 0, lane);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].init(g.B, g.ldb, n0, g.N, k_begin, ptid + S_PT * i, Geo<TN>::BN, X3_BM, lane);
  }
  template <bool MASKED>
  __device__ __forceinline__ void load(int64_t k_left, int adv) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].template load<MASKED>(k_left, adv);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].template load<MASKED>(k_left, adv);
  }
  __device__ __forceinline__ void skip() {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].skip();
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].skip();
  }
  __device__ __forceinline__ void store(unsigned short* stage) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i].store(stage);
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i].store(stage);
  }
};

// Narrow tile (TN = 2, 80 KB of LDS): capped at 128 registers so that two workgroups share a CU - one's prologue /
// epilogue overlaps the other's main loop (GRU products of the QM9-sized configs, K = 128: 2.05 -> 1.53 ms).
template <bool B_KM, int NPROD, int TN, int EPI = 0>
__global__ void __launch_bounds__(X3_NT, TN == 2 ? 4 : 2) gemm_x3s_kernel(X3Args g_in) {
  constexpr int TM = 2;
  constexpr int PLANE = Geo<TN>::PLANE, STAGE = Geo<TN>::STAGE, BN = Geo<TN>::BN;
  constexpr int EP_COLS = BN / 2;      // columns of a multiplying wave's block
  constexpr int EP_LD = EP_COLS + 4;   // floats per row of an epilogue block
  static_assert(3 * STAGE * 2 >= 4 * 32 * EP_LD * 4, "epilogue blocks must fit in the stage buffers");
  __shared__ __attribute__((aligned(16))) unsigned short lds[3 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool multiplier = wave < 4;
  const X3Tile tile = x3_tile(g_in);
  if (!tile.valid) return;  // uniform for the whole workgroup
  const unsigned bx = tile.x, bz = tile.z;
  const int64_t m0 = (int64_t)(bx / g_in.n_tiles) * X3_BM;
  const int64_t n0 = (int64_t)(bx % g_in.n_tiles) * BN;
  if (g_in.group_mode == 1) {  // rows [off[y], off[y+1]) of A and C with the group's own B
    const int64_t gb = g_in.group_off[blockIdx.y], ge = g_in.group_off[blockIdx.y + 1];
    if (m0 >= ge - gb) return;  // uniform for the whole workgroup
  }
  X3Args g = g_in;
  if (g.group_mode == 1) {
    const int64_t gb = g.group_off[blockIdx.y];
    g.M = g.group_off[blockIdx.y + 1] - gb;
    g.A += gb * g.lda;
    g.C += gb * g.ldc;
    g.B += (int64_t)blockIdx.y * g.strideB;
    if (g.mul) g.mul += gb * g.ld_mul;        // the gradient factors are indexed like C
    if (g.saved) g.saved += gb * g.ld_saved;
  }
  const int64_t k_begin = (int64_t)bz * g.k_chunk;
  const int64_t k_end = k_begin + g.k_chunk < g.K ? k_begin + g.k_chunk : g.K;
  const int64_t k_len = k_end - k_begin;
  const int T = k_len > 0 ? (int)((k_len + X3_BK - 1) / X3_BK) : 0;
  const int wm = wave & 1, wn = (wave >> 1) & 1;
  const int li = lane & 31, lk = lane >> 5;
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (!multiplier && T > 0) {
    // ------------------------------- producer waves -------------------------------
    Producer<B_KM, TN> p0;
    p0.init(g, m0, n0, k_begin, tid - S_PT);
    p0.template load<true>(k_len, 1);
    p0.store(lds);
    p0.template load<true>(k_len - X3_BK, 1);
    p0.store(lds + STAGE);
    // three register sets in rotation: the fetch of tile t+4 is issued at the top of iteration t and is split /
    // stored two iterations later (tile u by set u % 3; each set's pointers advance three tiles per fetch)
    Producer<B_KM, TN> p1 = p0;
    p1.skip();
    Producer<B_KM, TN> p2 = p1;
    p2.skip();
    p0.template load<true>(k_len - 2 * X3_BK, 3);
    p1.template load<true>(k_len - 3 * X3_BK, 3);
    __syncthreads();
    int s_st = 2 * STAGE;  // stage of tile t+2
    auto iteration = [&](auto masked, int64_t left4, const Producer<B_KM, TN>& from, Producer<B_KM, TN>& to) {
      constexpr bool MASKED = decltype(masked)::value;
      to.template load<MASKED>(left4, 3);                      // tile t+4 (no branch around the fetch / the stores:
                                                               // with one the compiler waits for vmcnt(6..0) here)
      __builtin_amdgcn_sched_barrier(0);   // fetches first: two iterations to land
      from.store(lds + s_st);                                  // tile t+2
      __syncthreads();
      s_st = s_st == 2 * STAGE ? 0 : s_st + STAGE;
    };
    int t = 0;
    const int t_fast = (int)(k_len / X3_BK) - 4;  // iterations whose fetched tile t+4 is a full tile
    for (; t + 2 < t_fast; t += 3) {
      iteration(std::false_type{}, 0, p0, p2);
      iteration(std::false_type{}, 0, p1, p0);
      iteration(std::false_type{}, 0, p2, p1);
    }
    for (; t < T; t += 3) {
      iteration(std::true_type{}, k_len - (int64_t)(t + 4) * X3_BK, p0, p2);
      if (t + 1 < T) iteration(std::true_type{}, k_len - (int64_t)(t + 5) * X3_BK, p1, p0);
      if (t + 2 < T) iteration(std::true_type{}, k_len - (int64_t)(t + 6) * X3_BK, p2, p1);
    }
  } else if (multiplier && T > 0) {
    // --------------------------------- multiplying waves ---------------------------------
    const int half = (lk ^ (li >> 3)) & 1;
    const int a_frag = (wm * 64 + li) * P_ROW + half * 8;
    const int b_frag = (X3_BM + wn * TN * 32 + li) * P_ROW + half * 8;
    __builtin_amdgcn_s_setprio(2);
    __syncthreads();  // tiles 0 and 1 are in stages 0 and 1
    bf16x8 ah[TM], am[TM], al[TM], bh, bm, bl;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ah[i] = frag8(lds + a_frag + i * 32 * P_ROW);
      am[i] = frag8(lds + a_frag + i * 32 * P_ROW + PLANE);
      al[i] = frag8(lds + a_frag + i * 32 * P_ROW + 2 * PLANE);
    }
    bh = frag8(lds + b_frag); bm = frag8(lds + b_frag + PLANE); bl = frag8(lds + b_frag + 2 * PLANE);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): enter the loop with no LDS read pending, as every back edge does
    int s_cur = 0, s_nxt = STAGE;
    for (int t = 0; t < T; ++t) {
      const unsigned short* cur = lds + s_cur;
      const unsigned short* nxs = lds + s_nxt;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bf16x8 nh, nm, nl;
        if (j + 1 < TN) {
          const unsigned short* nb = cur + b_frag + (j + 1) * 32 * P_ROW;
          nh = frag8(nb); nm = frag8(nb + PLANE); nl = frag8(nb + 2 * PLANE);
        } else {  // first operands of tile t+1 (complete since the previous barrier)
          nh = frag8(nxs + b_frag); nm = frag8(nxs + b_frag + PLANE); nl = frag8(nxs + b_frag + 2 * PLANE);
        }
        __builtin_amdgcn_sched_barrier(0);  // the reads above stay a full MFMA group (12 x 32 cycles) ahead of their use
        // the two row tiles alternate: consecutive MFMAs never hit the same accumulator; smallest terms first
#define X3S_PAIR(PA, PB)                                                                   \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[0], PB, acc[0][j], 0, 0, 0); \
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[1], PB, acc[1][j], 0, 0, 0);
        if (NPROD >= 9) { X3S_PAIR(al, bl) }
        if (NPROD >= 8) { X3S_PAIR(am, bl) X3S_PAIR(al, bm) }
        X3S_PAIR(al, bh) X3S_PAIR(ah, bl) X3S_PAIR(am, bm) X3S_PAIR(am, bh) X3S_PAIR(ah, bm) X3S_PAIR(ah, bh)
#undef X3S_PAIR
        bh = nh; bm = nm; bl = nl;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag8(nxs + a_frag + i * 32 * P_ROW);
        am[i] = frag8(nxs + a_frag + i * 32 * P_ROW + PLANE);
        al[i] = frag8(nxs + a_frag + i * 32 * P_ROW + 2 * PLANE);
      }
      __syncthreads();
      s_cur = s_nxt;
      s_nxt = s_nxt == 2 * STAGE ? 0 : s_nxt + STAGE;
    }
    __builtin_amdgcn_s_setprio(0);
  }

  // Epilogue in two rounds of 32 rows per multiplying wave: the accumulators go to LDS as four 32 x 32 TN
  // blocks, then all eight waves stream them out with bias / activation / accumulate (and the gradient factors)
  // applied in a compact loop (row-contiguous 16-byte stores; no wait on earlier stores anywhere).
  const bool split = g.splits > 1;
  float* outp = split ? g.partial + (int64_t)bz * g.M * g.N : g.C;
  const int64_t ldo = split ? g.N : g.ldc;
  float* ep = reinterpret_cast<float*>(lds);
  constexpr int ROW4 = EP_COLS / 4;  // float4 per block row
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (multiplier) {
      float* blk = ep + wave * 32 * EP_LD;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) blk[((r & 3) + 8 * (r >> 2) + 4 * lk) * EP_LD + j * 32 + li] = acc[i][j][r];
    }
    __syncthreads();
    if constexpr (EPI == 1) {
      // GRU epilogue ([ext] GRUCell, reset_after: ggnn.py:84-87): tile column j = gate * 64 + c holds mx_gate of unit
      // c of this tile's 64 units; z and the first half of r are in the wn = 0 blocks, the rest in the wn = 1 blocks.
      static_assert(EPI != 1 || TN == 3, "the GRU epilogue is written for the 192-wide tile");
      const int H = g.gru_H;
      const int64_t unit0 = (n0 / BN) * 64;
#pragma unroll 1
      for (int f = tid; f < 2 * 32 * 16; f += X3_NT) {
        const int wmb = f >> 9, rem = f & 511;
        const int row = rem >> 4, cl = (rem & 15) * 4;
        const int64_t grow = m0 + wmb * 64 + i * 32 + row;
        if (grow < g.M) {
          const float* blk0 = ep + (wmb * 32 + row) * EP_LD;
          const float* blk1 = ep + ((wmb + 2) * 32 + row) * EP_LD;
          float4 xz = *reinterpret_cast<const float4*>(blk0 + cl);
          float4 xr = cl < 32 ? *reinterpret_cast<const float4*>(blk0 + 64 + cl) : *reinterpret_cast<const float4*>(blk1 + cl - 32);
          float4 xh = *reinterpret_cast<const float4*>(blk1 + 32 + cl);
          if (g.bias) {
            const float4 bz = *reinterpret_cast<const float4*>(g.bias + n0 + cl);
            const float4 br = *reinterpret_cast<const float4*>(g.bias + n0 + 64 + cl);
            const float4 bh = *reinterpret_cast<const float4*>(g.bias + n0 + 128 + cl);
            xz.x += bz.x; xz.y += bz.y; xz.z += bz.z; xz.w += bz.w;
            xr.x += br.x; xr.y += br.y; xr.z += br.z; xr.w += br.w;
            xh.x += bh.x; xh.y += bh.y; xh.z += bh.z; xh.w += bh.w;
          }
          const int64_t unit = unit0 + cl;
          const float* pm = g.gru_mh + grow * 3 * H + unit;
          const float4 hz = *reinterpret_cast<const float4*>(pm);
          const float4 hr = *reinterpret_cast<const float4*>(pm + H);
          const float4 hc = *reinterpret_cast<const float4*>(pm + 2 * H);
          const float4 hv = *reinterpret_cast<const float4*>(g.gru_h + grow * H + unit);
          float4 z, r, c, o;
#define X3_GRU1(q)                                          \
          z.q = 1.f / (1.f + expf(-(xz.q + hz.q)));         \
          r.q = 1.f / (1.f + expf(-(xr.q + hr.q)));         \
          c.q = tanhf(xh.q + r.q * hc.q);                   \
          o.q = z.q * hv.q + (1.f - z.q) * c.q;
          X3_GRU1(x) X3_GRU1(y) X3_GRU1(z) X3_GRU1(w)
#undef X3_GRU1
          *reinterpret_cast<float4*>(g.C + grow * g.ldc + unit) = o;
          if (g.gru_gates) {
            float* pg = g.gru_gates + grow * 3 * H + unit;
            *reinterpret_cast<float4*>(pg) = z;
            *reinterpret_cast<float4*>(pg + H) = r;
            *reinterpret_cast<float4*>(pg + 2 * H) = c;
          }
        }
      }
    } else
#pragma unroll 1
    for (int f = tid; f < 4 * 32 * ROW4; f += X3_NT) {
      const int b = f / (32 * ROW4), rem = f - b * (32 * ROW4);
      const int row = rem / ROW4, c4 = rem - row * ROW4;
      const int64_t grow = m0 + (b & 1) * 64 + i * 32 + row;
      const int64_t gcol = n0 + (b >> 1) * EP_COLS + c4 * 4;
      if (grow < g.M) {
        float4 v = *reinterpret_cast<const float4*>(ep + (b * 32 + row) * EP_LD + c4 * 4);
        float* dst = outp + grow * ldo + gcol;
        if (!split) {
          if (g.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(g.bias + gcol);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          v.x = act_apply(g.act, v.x); v.y = act_apply(g.act, v.y);
          v.z = act_apply(g.act, v.z); v.w = act_apply(g.act, v.w);
          v = grad_epilogue(g, v, grow, gcol);
          if (g.accumulate) {
            const float4 c4v = *reinterpret_cast<const float4*>(dst);
            v.x += c4v.x; v.y += c4v.y; v.z += c4v.z; v.w += c4v.w;
          }
        }
        *reinterpret_cast<float4*>(dst) = v;
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) x3_splitk_reduce_kernel(X3Args g) {
  const int64_t total = g.M * g.N;
  const float* part = g.partial + (int64_t)blockIdx.y * g.splits * total;  // blockIdx.y = group (0 if ungrouped)
  g.C += (int64_t)blockIdx.y * g.strideC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < g.splits; ++z) s += part[(int64_t)z * total + i];
    const int64_t row = i / g.N, col = i - row * g.N;
    if (g.bias) s += g.bias[col];
    s = act_apply(g.act, s);
    if (g.mul) s *= g.mul[row * g.ld_mul + col];
    if (g.saved) s *= act_grad(g.dact, g.saved[row * g.ld_saved + col]);
    float* c = g.C + row * g.ldc + col;
    if (g.accumulate) s += *c;
    *c = s;
  }
}

// The same sum for many splits (skinny outputs under a long K run up to 512 of them, splitk_want): a block owns 64
// consecutive elements, 16 lanes x float4 wide, and its 16 lane rows walk the splits 16 apart; the 16 slice sums are
// added in slice order through LDS.  Fixed order, so results repeat bit for bit.
__global__ void __launch_bounds__(256) x3_splitk_reduce_sliced_kernel(X3Args g) {
  __shared__ float4 acc[16][16];
  const int64_t total = g.M * g.N;  // a multiple of 64 (launch site)
  const float* part = g.partial + (int64_t)blockIdx.y * g.splits * total;
  g.C += (int64_t)blockIdx.y * g.strideC;
  const int q = threadIdx.x & 15, slice = threadIdx.x >> 4;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t i = base + q * 4;
    float4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int z = slice; z < g.splits; z += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)z * total + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    acc[slice][q] = s;
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int t = 1; t < 16; ++t) {
        const float4 v = acc[t][q];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int64_t row = i / g.N, col = i - row * g.N;  // N % 4 == 0: the four elements share a row
      float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r = o[e];
        if (g.bias) r += g.bias[col + e];
        r = act_apply(g.act, r);
        if (g.mul) r *= g.mul[row * g.ld_mul + col + e];
        if (g.saved) r *= act_grad(g.dact, g.saved[row * g.ld_saved + col + e]);
        float* c = g.C + row * g.ldc + col + e;
        if (g.accumulate) r += *c;
        *c = r;
      }
    }
    __syncthreads();
  }
}

static void launch_x3_splitk_reduce(const X3Args& g, unsigned groups, hipStream_t s) {
  const int64_t total = g.M * g.N;
  if (g.splits > 64 && total % 64 == 0 && g.N % 4 == 0)
    hipLaunchKernelGGL(x3_splitk_reduce_sliced_kernel, dim3((unsigned)std::min<int64_t>(total / 64, 8192), groups), dim3(256), 0, s, g);
  else
    hipLaunchKernelGGL(x3_splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 4096), groups), dim3(256), 0, s, g);
}

// Layouts: NN / NT (A K-contiguous) -> specialised kernel (NT 140 vs 160 us, NN 68 vs 72 us for the pipelined one
// on the hot-path shapes); TN (both K-major: weight gradients) -> pipelined kernel (with two K-major operands four
// producer waves cannot keep up with the register transposes: 185 vs 207 us).
// NT products on the pipelined kernel: the narrow tile only (106 registers there: two workgroups per CU and all eight
// waves of each multiplying - the QM9-sized shapes run ~10 % faster than on the specialised kernel; on the wide tiles the
// two kernels take the same time).  TFGNN_X3_NT_PIPELINED = 0 / 1 forces never / always.
template <int TN>
static bool pipelined_nt() {
  static const int knob = [] { const char* e = getenv("TFGNN_X3_NT_PIPELINED"); return e ? atoi(e) : -1; }();
  return knob < 0 ? TN == 2 : knob != 0;
}

template <int TN>
static void launch_x3(const X3Args& g, dim3 grid, int nprod, int trans_a, int trans_b, hipStream_t s) {
  const bool nine = nprod >= 9;
  count_launch(TFGNN_KFAM_GEMM_BF16X3);
  if (trans_a) {
    if (nine) hipLaunchKernelGGL((gemm_x3p_kernel<9, TN>), grid, dim3(X3_NT), 0, s, g);
    else hipLaunchKernelGGL((gemm_x3p_kernel<6, TN>), grid, dim3(X3_NT), 0, s, g);
  } else if (trans_b && g.group_mode == 0 && pipelined_nt<TN>()) {
    if (nine) hipLaunchKernelGGL((gemm_x3p_kernel<9, TN, true>), grid, dim3(X3_NT), 0, s, g);
    else hipLaunchKernelGGL((gemm_x3p_kernel<6, TN, true>), grid, dim3(X3_NT), 0, s, g);
  } else if (trans_b) {
    if (nine) hipLaunchKernelGGL((gemm_x3s_kernel<false, 9, TN>), grid, dim3(X3_NT), 0, s, g);
    else hipLaunchKernelGGL((gemm_x3s_kernel<false, 6, TN>), grid, dim3(X3_NT), 0, s, g);
  } else {
    if (nine) hipLaunchKernelGGL((gemm_x3s_kernel<true, 9, TN>), grid, dim3(X3_NT), 0, s, g);
    else hipLaunchKernelGGL((gemm_x3s_kernel<true, 6, TN>), grid, dim3(X3_NT), 0, s, g);
  }
}

// =====================================================================================================
// Streaming kernel for short-K products over very many rows (QM9-sized batches: [10^6, 128] x [128, 128 n], the Dense
// layers, the per-edge MLP layers and the GRU inputs of configs[3]).  On these shapes a tiled kernel spends its time
// re-staging and re-splitting the SAME weight tile for every 128 rows (K = 128 is 8 k-steps: prologue and epilogue
// never overlap anything).  Here the weight block is the resident operand: a workgroup of eight waves splits its 128
// columns of B once into three bf16 planes in LDS (K x 128 x 3 x 2 B <= 102 KB) and then only streams rows.  Every
// wave owns whole 32-row tiles - no barrier after the fill: a lane loads its row's fp32 values straight from global
// memory in MFMA operand order (lane (row i, half kg) takes k in [kg K/2, (kg + 1) K/2), 8 per k-step; a sum over k
// does not care about the order as long as B is read with the same map), splits them in registers, reads the B
// pieces as ds_read_b128 (rows padded by 16 B: conflict-free) and keeps the next tile's loads in flight under the 24 K/16
// MFMAs of the current one (16 KB per wave, 128 KB per CU).  The accumulators leave through a wave-private LDS patch as
// 16-byte stores.  Column blocks of one row range run as neighbouring workgroups of the
// SAME XCD, so the rows cross the fabric once and are re-read from that XCD's L2.
constexpr int XK_COLS = 128, XK_NT = 512, XK_WAVES = 8;
typedef __attribute__((address_space(3))) void x3k_lds_void;
template <int OFF>
__device__ __forceinline__ void x3k_request(uint4v (&dst)[3], const unsigned (&addr)[3]) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(addr[0]), "n"(OFF) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(addr[1]), "n"(OFF) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(addr[2]), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void x3k_request(uint4v (&dst)[2], const unsigned (&addr)[2]) {  // two planes: the f16x2 form
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(addr[0]), "n"(OFF) : "memory");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(addr[1]), "n"(OFF) : "memory");
}
__device__ __forceinline__ void x3k_wait_all_but_3(uint4v (&b)[3]) {  // the three reads of the group BEFORE the last request
  asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2])::"memory");
}
__device__ __forceinline__ void x3k_wait_all_but_3(uint4v (&b)[2]) {  // (two planes: all but the last request's two reads)
  asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(b[0]), "+v"(b[1])::"memory");
}
__device__ __forceinline__ void x3k_wait_all(uint4v (&b)[3]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2])::"memory");
}
__device__ __forceinline__ void x3k_wait_all(uint4v (&b)[2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1])::"memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void x3k_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    x3k_static_for<I + 1, N>(f);
  }
}
constexpr int XK_PS = 36;  // row stride (floats) of a wave's 32 x 32 epilogue patch

// MODE: 0 plain epilogue (bias, activation), 1 + gradient factors / accumulate, 2 GRU gate math (three column tiles: z | r | h
// of 32 units per workgroup; B in the regrouped layout of tfgnn_gemm_gru)
constexpr int XK_PLAIN = 0, XK_EXTRAS = 1, XK_GRU = 2;
// NPROD = 3 (round 6): the f16x2 arithmetic of gemm_sp.hip inside this kernel - what mode f16x2 runs for the products that have
// no split PRODUCER (the per-edge and Dense products of QM9-sized batches: 3 MFMAs per k-step and column tile instead of 6).
// B is split when the workgroup fills its block: two fp16 planes under one power-of-two scale per column; a lane scales the
// K / 2 values of its row half by the power of two of the ROW maximum (one exchange with lane ^ 32), splits them into
// h = fp16(xs), l = fp16(xs - h) and multiplies l_a h_b + h_a l_b + h_a h_b (each product exact in fp32); the two scales leave
// in the epilogue.  Same error class as the other f16x2 products (gemm_sp.hip: >= 22 significand bits per value); no spread
// guard is needed - the scales belong to the rows / columns of the RESULT, not to k.
template <int NPROD, int KS, int MODE>
__global__ void __launch_bounds__(XK_NT) gemm_x3k_kernel(X3Args g, int b_kmajor, int ncb, int spx) {
  constexpr bool EXTRAS = MODE == XK_EXTRAS;
  constexpr bool F16 = NPROD == 3;
  constexpr int NPL = F16 ? 2 : 3;  // operand planes
  constexpr int CT = MODE == XK_GRU ? 3 : 4, COLS = CT * 32;  // column tiles / columns per workgroup
  constexpr int K = KS * 16, ROWB = 2 * K + 16, PLANE_B = COLS * ROWB;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kg = lane >> 5;
  // workgroup id -> (XCD, slot); the ncb column blocks of a row stream sit in neighbouring slots of one XCD
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned cb = slot % (unsigned)ncb, stream_local = slot / (unsigned)ncb;
  if ((int)stream_local >= spx) return;
  const int stream = (int)(stream_local * 8 + xcd), nstreams = spx * 8;
  const int64_t n0 = (int64_t)cb * COLS;
  // row of B (K-contiguous form) behind column n of this workgroup; GRU: unit block cb = half (cb & 1) of the 64 units of
  // the 192-row group cb / 2, rows [z 64 | r 64 | h 64]
  auto b_row = [&](int n) -> int64_t {
    if constexpr (MODE == XK_GRU) return (int64_t)(cb >> 1) * 192 + (n >> 5) * 64 + (cb & 1) * 32 + (n & 31);
    else return n0 + n;
  };

  // ---- fill: B[:, n0 .. n0 + 127] -> planes[p][n][k] (bf16, k natural order) ---------------------------------
  float* const lds_f32 = reinterpret_cast<float*>(lds + NPL * PLANE_B);  // [patches 8 x 32 x XK_PS | bias COLS | (F16) col inv COLS | col max COLS]
  if constexpr (F16) {
    float* colinv = lds_f32 + XK_WAVES * 32 * XK_PS + COLS;
    unsigned* colmax = reinterpret_cast<unsigned*>(colinv + COLS);
    if (tid < COLS) colmax[tid] = 0u;
    __syncthreads();
    // column maxima (non-negative floats order like their bit patterns: an LDS atomic maximum, order-free)
    if (!b_kmajor) {
      for (int idx = tid; idx < COLS * (K / 4); idx += XK_NT) {
        const int n = idx / (K / 4), kq = idx - n * (K / 4);
        const float4 v = *reinterpret_cast<const float4*>(g.B + b_row(n) * g.ldb + kq * 4);
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        if (!(m == m)) m = __builtin_inff();  // a NaN in the column: no scaling (sp_scale_for_max), the result is NaN anyway
        atomicMax(colmax + n, __float_as_uint(m));
      }
    } else {
      for (int idx = tid; idx < K * (COLS / 4); idx += XK_NT) {
        const int k = idx / (COLS / 4), nq = idx - k * (COLS / 4);
        const float4 v = *reinterpret_cast<const float4*>(g.B + (int64_t)k * g.ldb + n0 + nq * 4);
        const float e[4] = {fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicMax(colmax + nq * 4 + q, __float_as_uint(e[q] == e[q] ? e[q] : __builtin_inff()));
      }
    }
    __syncthreads();
    if (tid < COLS) {
      float iv;
      const float sc = sp_scale_for_max(__uint_as_float(colmax[tid]), &iv);
      colinv[tid] = iv;
      colmax[tid] = __float_as_uint(sc);  // from here on: the column's scale
    }
    __syncthreads();
    const float* colscale = reinterpret_cast<const float*>(colmax);
    if (!b_kmajor) {
      for (int idx = tid; idx < COLS * (K / 4); idx += XK_NT) {
        const int n = idx / (K / 4), kq = idx - n * (K / 4);
        const float4 v = *reinterpret_cast<const float4*>(g.B + b_row(n) * g.ldb + kq * 4);
        const float sc = colscale[n];
        _Float16 h[4], l[4];
        sp_split(v.x * sc, h[0], l[0]); sp_split(v.y * sc, h[1], l[1]); sp_split(v.z * sc, h[2], l[2]); sp_split(v.w * sc, h[3], l[3]);
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<half4v*>(lds + n * ROWB + kq * 8) = half4v{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<half4v*>(lds + PLANE_B + n * ROWB + kq * 8) = half4v{l[0], l[1], l[2], l[3]};
      }
    } else {
      for (int idx = tid; idx < K * (COLS / 4); idx += XK_NT) {
        const int k = idx / (COLS / 4), nq = idx - k * (COLS / 4);
        const float4 v = *reinterpret_cast<const float4*>(g.B + (int64_t)k * g.ldb + n0 + nq * 4);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          _Float16 h, l;
          sp_split(e[q] * colscale[nq * 4 + q], h, l);
          _Float16* d = reinterpret_cast<_Float16*>(lds + (nq * 4 + q) * ROWB + k * 2);
          d[0] = h;
          d[PLANE_B / 2] = l;
        }
      }
    }
  } else if (!b_kmajor) {  // B given as [N, K] (K-contiguous)
    for (int idx = tid; idx < COLS * (K / 4); idx += XK_NT) {
      const int n = idx / (K / 4), kq = idx - n * (K / 4);
      const float4 v = *reinterpret_cast<const float4*>(g.B + b_row(n) * g.ldb + kq * 4);
      split_store4(v.x, v.y, v.z, v.w, reinterpret_cast<unsigned short*>(lds + n * ROWB + kq * 8), PLANE_B / 2);
    }
  } else {  // B given as [K, N]
    for (int idx = tid; idx < K * (COLS / 4); idx += XK_NT) {  // (not reached in GRU mode: its B is K-contiguous)
      const int k = idx / (COLS / 4), nq = idx - k * (COLS / 4);
      const float4 v = *reinterpret_cast<const float4*>(g.B + (int64_t)k * g.ldb + n0 + nq * 4);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned h, m, l;
        split3(e[q], h, m, l);
        unsigned short* d = reinterpret_cast<unsigned short*>(lds + (nq * 4 + q) * ROWB + k * 2);
        d[0] = (unsigned short)(h >> 16);
        d[PLANE_B / 2] = (unsigned short)(m >> 16);
        d[PLANE_B] = (unsigned short)(l >> 16);
      }
    }
  }
  if (g.bias && tid < COLS) lds_f32[XK_WAVES * 32 * XK_PS + tid] = g.bias[b_row(tid)];
  __syncthreads();

  const int ntiles = (int)((g.M + 31) / 32);  // M * lda < 2^30 (launch site): rows, tiles and blocks fit 32 bits
  const int nblocks = (ntiles + XK_WAVES - 1) / XK_WAVES;
  // byte offset of this lane's half row (32 bits: the launch site requires M * lda * 4 < 2^32).  The streamed operand is read
  // through a raw buffer descriptor: one address register per lane and tile, the 16 loads differ in the immediate offset
  const int Mi = (int)g.M;
  const unsigned a_row_bytes = (unsigned)g.lda * 4u, a_lane_bytes = (unsigned)kg * (K / 2) * 4u;
  const int32_t* __restrict__ a_index = g.a_index;
  const unsigned a_rows_u = (unsigned)g.a_rows;
  auto row_off = [&](int tile) {
    int row = tile * 32 + li;
    if (row >= Mi) row = Mi - 1;
    if (a_index) {  // gathered rows (per-edge products over node states): a lane's row is its own anyway
      row = a_index[row];
      // an index outside [0, a_rows) must not alias another row through the 24-bit multiply: an offset past the buffer
      // descriptor (its size stays 8 KiB below 2^32: x3k_shape_ok) reads as a row of zeros
      if ((unsigned)row >= a_rows_u) return 0xFFFFE000u;
    }
    return __umul24((unsigned)row, a_row_bytes) + a_lane_bytes;  // v_mad_u32_u24 (rows, 4 lda < 2^24: launch site)
  };
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)(unsigned)((g.a_index ? g.a_rows : g.M) * g.lda * 4), 0x00020000);
  auto a_load = [&](unsigned off, int i) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off + i * 16, 0, 0));
  };
  // B piece addresses: one LDS byte address per plane (+ col tile * 32 * ROWB + step * 16 as the immediate offset).  The reads
  // are written as asm so that they are issued ONE GROUP (6 - 9 MFMAs) AHEAD of their use; left to the compiler every read
  // sits right in front of its MFMAs and the LDS latency is paid 4 K/16 times per tile (the multiply side alone took 198 us
  // of a 320 us kernel, twice the time of its MFMAs).  A request's registers reach the MFMAs through the "+v" operands of the
  // wait, so neither can move above it.
  unsigned b_addr[NPL];
#pragma unroll
  for (int p = 0; p < NPL; ++p) b_addr[p] = (unsigned)(uintptr_t)(x3k_lds_void*)lds + p * PLANE_B + li * ROWB + kg * K;
  uint4v bq[2][NPL];
  auto b_request = [&](auto group_c) {  // group = step * CT + column tile
    constexpr int grp = decltype(group_c)::value;
    x3k_request<(grp % CT) * 32 * ROWB + (grp / CT) * 16>(bq[grp & 1], b_addr);
  };
  float* patch = lds_f32 + wave * 32 * XK_PS;
  const float* lds_bias = lds_f32 + XK_WAVES * 32 * XK_PS;  // [128], filled above
  const float* lds_colinv = lds_bias + COLS;                // (F16) 2^-e of the columns
  float4 araw[2 * KS];
  int blk = stream;
  {
    const int t0 = blk * XK_WAVES + wave;
    const unsigned ao = row_off(t0 < ntiles ? t0 : ntiles - 1);
#pragma unroll
    for (int j = 0; j < 2 * KS; ++j) araw[j] = a_load(ao, j);
  }
  for (; blk < nblocks; blk += nstreams) {
    const int tile = blk * XK_WAVES + wave;
    int next = (blk + nstreams) * XK_WAVES + wave;
    if (next >= ntiles) next = ntiles - 1;  // nothing left: a harmless re-load
    const unsigned no = row_off(next);
    floatx16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    // requests are pending only inside the step code below (VALU + MFMA, no spill code: the compiler does not know that the
    // registers of an asm read are still in flight - around the epilogue it spilled them and saved the OLD contents)
    // F16: the power-of-two scale of this lane's row - the maximum over its K / 2 values and, through lane ^ 32, the other half
    float a_scale = 1.f, a_inv = 1.f;
    if constexpr (F16) {
      float mx = 0.f;
      bool bad = false;
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        const float4 v = araw[i];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        bad = bad || !(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w);
      }
      if (bad) mx = __builtin_inff();  // a NaN in the row: no scaling, the row is NaN anyway (fmaxf drops NaNs)
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      a_scale = sp_scale_for_max(mx, &a_inv);
    }
    b_request(std::integral_constant<int, 0>{});
    x3k_static_for<0, KS>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const float4 x0 = araw[2 * j], x1 = araw[2 * j + 1];
      const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if constexpr (F16) {
        half8v ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 hh, ll;
          sp_split(xs[e] * a_scale, hh, ll);
          ah[e] = hh;
          al[e] = ll;
        }
        x3k_static_for<0, CT>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value, cur = (j * CT + c) & 1;
          if constexpr (j * CT + c + 1 < CT * KS) {
            b_request(std::integral_constant<int, j * CT + c + 1>{});
            x3k_wait_all_but_3(bq[cur]);
          } else {
            x3k_wait_all(bq[cur]);  // last group of the tile: nothing stays pending across the epilogue
          }
          const half8v bh = __builtin_bit_cast(half8v, bq[cur][0]), bl = __builtin_bit_cast(half8v, bq[cur][1]);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[c], 0, 0, 0);  // smallest terms first
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[c], 0, 0, 0);
        });
      } else {
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(xs[e], h[e], m[e], l[e]);
      const bf16x8 ah = __builtin_bit_cast(bf16x8, uint4v{pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7])});
      const bf16x8 am = __builtin_bit_cast(bf16x8, uint4v{pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7])});
      const bf16x8 al = __builtin_bit_cast(bf16x8, uint4v{pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7])});
      x3k_static_for<0, CT>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value, cur = (j * CT + c) & 1;
        if constexpr (j * CT + c + 1 < CT * KS) {
          b_request(std::integral_constant<int, j * CT + c + 1>{});
          x3k_wait_all_but_3(bq[cur]);
        } else {
          x3k_wait_all(bq[cur]);  // last group of the tile: nothing stays pending across the epilogue
        }
        if constexpr (!F16)
          acc[c] = mfma_group<NPROD>(acc[c], ah, am, al, __builtin_bit_cast(bf16x8, bq[cur][0]), __builtin_bit_cast(bf16x8, bq[cur][1]),
                                     __builtin_bit_cast(bf16x8, bq[cur][NPL - 1]));
      });
      }
      // the next tile's values, one burst of whole 128-byte lines per lane every four steps (with two loads per step a line
      // stays half-read for four steps while the other seven waves push it out of the 32 KB L1)
      if ((j & 3) == 3 || j == KS - 1) {
#pragma unroll
        for (int i = 2 * (j & ~3); i <= 2 * j + 1; ++i) araw[i] = a_load(no, i);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: hoisting every split to the top spills 370 registers
    });
    if (tile >= ntiles) continue;
    // ---- epilogue, straight from the accumulator layout: register r of lane (li, kg) is element
    // (row (r & 3) + 8 (r >> 2) + 4 kg, column li) of its 32 x 32 block ------------------------------------------
    if constexpr (F16) {  // the two power-of-two scales leave: row r's from the lane that holds row r, the column's from LDS
      float rf[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rf[r] = __shfl(a_inv, (r & 3) + 8 * (r >> 2) + 4 * kg, 64);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float ci = lds_colinv[c * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] *= rf[r] * ci;
      }
    }
    if (g.bias) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float bias = lds_bias[c * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] += bias;
      }
    }
    if constexpr (MODE == XK_GRU) {
      // GRU gate math ([ext] GRUCell, reset_after: ggnn.py:84-87): the three tiles are mx_z, mx_r, mx_h of the SAME 32 units;
      // each goes through the patch so that a lane ends up with four units of one row of all three
      float4 xg[3][4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * XK_PS + li] = acc[c][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xg[c][q] = *reinterpret_cast<const float4*>(patch + ((lane >> 3) + 8 * q) * XK_PS + (lane & 7) * 4);
        __builtin_amdgcn_wave_barrier();
      }
      const int H = g.gru_H;
      const int64_t unit = (int64_t)cb * 32 + (lane & 7) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t row = (int64_t)tile * 32 + (lane >> 3) + 8 * q;
        if (row < g.M) {
          const float* pm = g.gru_mh + row * 3 * H + unit;
          const float4 hz = *reinterpret_cast<const float4*>(pm);
          const float4 hr = *reinterpret_cast<const float4*>(pm + H);
          const float4 hc = *reinterpret_cast<const float4*>(pm + 2 * H);
          const float4 hv = *reinterpret_cast<const float4*>(g.gru_h + row * H + unit);
          const float4 xz = xg[0][q], xr = xg[1][q], xh = xg[2][q];
          float4 z, r, c, o;
#define X3K_GRU1(e)                                  \
          z.e = 1.f / (1.f + expf(-(xz.e + hz.e)));  \
          r.e = 1.f / (1.f + expf(-(xr.e + hr.e)));  \
          c.e = tanhf(xh.e + r.e * hc.e);            \
          o.e = z.e * hv.e + (1.f - z.e) * c.e;
          X3K_GRU1(x) X3K_GRU1(y) X3K_GRU1(z) X3K_GRU1(w)
#undef X3K_GRU1
          *reinterpret_cast<float4*>(g.C + row * g.ldc + unit) = o;
          if (g.gru_gates) {
            float* pg = g.gru_gates + row * 3 * H + unit;
            *reinterpret_cast<float4*>(pg) = z;
            *reinterpret_cast<float4*>(pg + H) = r;
            *reinterpret_cast<float4*>(pg + 2 * H) = c;
          }
        }
      }
      continue;
    }
    // one wave-uniform dispatch per tile on the activation (a switch per element is a taken branch per element)
    auto act_all = [&](auto act_c) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = act_apply(decltype(act_c)::value, acc[c][r]);
    };
    // accumulate == 2: C = act(C + A B + bias) - the activation then waits for the old values (after the patch)
    const bool act_late = EXTRAS && g.accumulate == 2;
    switch (act_late ? TFGNN_ACT_NONE : g.act) {
      case TFGNN_ACT_RELU: act_all(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
      case TFGNN_ACT_TANH: act_all(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
      case TFGNN_ACT_LEAKY_RELU: act_all(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
      case TFGNN_ACT_ELU: act_all(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
      case TFGNN_ACT_SELU: act_all(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
      case TFGNN_ACT_GELU: act_all(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
      case TFGNN_ACT_SIGMOID: act_all(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
      default: break;
    }
    // a 32 x 32 block at a time through the wave's LDS patch, so that a lane stores 16 contiguous bytes: dword stores
    // straight from the accumulator layout are store-issue bound (355 us for [1.15 M, 128] x [128, 128], 0.4 of HBM)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * XK_PS + li] = acc[c][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      const int64_t col = n0 + c * 32 + (lane & 7) * 4;
      float4 v[4];
      bool ok[4];
      int64_t rowq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pr = (lane >> 3) + 8 * q;
        rowq[q] = tile * 32 + pr;
        ok[q] = rowq[q] < g.M;
        if (!ok[q]) rowq[q] = g.M - 1;
        v[q] = *reinterpret_cast<const float4*>(patch + pr * XK_PS + (lane & 7) * 4);
      }
      if constexpr (EXTRAS) {
        if (g.mul) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 m = *reinterpret_cast<const float4*>(g.mul + rowq[q] * g.ld_mul + col);
            v[q].x *= m.x; v[q].y *= m.y; v[q].z *= m.z; v[q].w *= m.w;
          }
        }
        if (g.saved) {
          float4 sv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) sv[q] = *reinterpret_cast<const float4*>(g.saved + rowq[q] * g.ld_saved + col);
          auto dact_all = [&](auto act_c) {
            constexpr int A = decltype(act_c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q].x *= act_grad(A, sv[q].x); v[q].y *= act_grad(A, sv[q].y);
              v[q].z *= act_grad(A, sv[q].z); v[q].w *= act_grad(A, sv[q].w);
            }
          };
          switch (g.dact) {
            case TFGNN_ACT_RELU: dact_all(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
            case TFGNN_ACT_TANH: dact_all(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
            case TFGNN_ACT_LEAKY_RELU: dact_all(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
            case TFGNN_ACT_ELU: dact_all(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
            case TFGNN_ACT_SELU: dact_all(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
            case TFGNN_ACT_GELU: dact_all(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
            case TFGNN_ACT_SIGMOID: dact_all(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
            default: dact_all(std::integral_constant<int, TFGNN_ACT_NONE>{}); break;
          }
        }
        if (g.accumulate) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 c4 = *reinterpret_cast<const float4*>(g.C + rowq[q] * g.ldc + col);
            v[q].x += c4.x; v[q].y += c4.y; v[q].z += c4.z; v[q].w += c4.w;
          }
        }
        if (act_late) {
          auto late = [&](auto act_c) {
            constexpr int A = decltype(act_c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              v[q].x = act_apply(A, v[q].x); v[q].y = act_apply(A, v[q].y);
              v[q].z = act_apply(A, v[q].z); v[q].w = act_apply(A, v[q].w);
            }
          };
          switch (g.act) {
            case TFGNN_ACT_RELU: late(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
            case TFGNN_ACT_TANH: late(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
            case TFGNN_ACT_LEAKY_RELU: late(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
            case TFGNN_ACT_ELU: late(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
            case TFGNN_ACT_SELU: late(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
            case TFGNN_ACT_GELU: late(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
            case TFGNN_ACT_SIGMOID: late(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
            default: break;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q]) *reinterpret_cast<float4*>(g.C + rowq[q] * g.ldc + col) = v[q];
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// GRUCell with BOTH of its products in the kernel (round 6; ggnn.py:84-87): h' = GRU(x W + b_0, h U + b_1, h).
// The GRU form above still read mh = h U + b_1 - a [V, 3H] tensor another product had written: 3.5 GB per layer at the QM9 size
// for something that is consumed once.  Here a workgroup keeps the [z | r | h] column blocks of its 32 units of BOTH kernels in
// LDS (f16x2 planes: 2 x 192 columns x (2 K + 16) bytes = 104 KB at K = 128; the three bf16 planes of the exact form would not
// fit - this kernel exists in the f16x2 arithmetic only) and a tile of 32 rows is multiplied twice: the rows of x against W
// (while the rows of h stream into the registers the consumed values leave), then the rows of h against U (while the next
// tile's x streams in).  The gate math needs of mh only its candidate third: written to the last third of d_mh_out's rows for
// the backward pass (tfgnn_gru_gates_backward* read nothing else of it).  K = H (GGNN: message width = state width).
template <int KS>
__global__ void __launch_bounds__(XK_NT) gemm_x3k_gru2_kernel(X3Args g, const float* __restrict__ B2, const float* __restrict__ bias2,
                                                             float* __restrict__ mh_out, int ncb, int spx) {
  constexpr int CT = 3, COLS = 2 * CT * 32;  // 96 columns of W + 96 of U
  constexpr int K = KS * 16, ROWB = 2 * K + 16, PLANE_B = COLS * ROWB;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kg = lane >> 5;
  const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const unsigned cb = slot % (unsigned)ncb, stream_local = slot / (unsigned)ncb;
  if ((int)stream_local >= spx) return;
  const int stream = (int)(stream_local * 8 + xcd), nstreams = spx * 8;
  // row of the regrouped kernels ([3H, K], blocks of 192 rows = [z 64 | r 64 | h 64] of 64 units) behind column n < 96
  auto b_row = [&](int n) -> int64_t { return (int64_t)(cb >> 1) * 192 + (n >> 5) * 64 + (cb & 1) * 32 + (n & 31); };
  float* const lds_f32 = reinterpret_cast<float*>(lds + 2 * PLANE_B);  // [patches | bias 192 | col inv 192 | col max 192]
  float* lds_bias = lds_f32 + XK_WAVES * 32 * XK_PS;
  float* colinv = lds_bias + COLS;
  unsigned* colmax = reinterpret_cast<unsigned*>(colinv + COLS);
  if (tid < COLS) colmax[tid] = 0u;
  __syncthreads();
  for (int idx = tid; idx < COLS * (K / 4); idx += XK_NT) {
    const int n = idx / (K / 4), kq = idx - n * (K / 4);
    const float* src = n < 96 ? g.B + b_row(n) * g.ldb : B2 + b_row(n - 96) * (int64_t)K;
    const float4 v = *reinterpret_cast<const float4*>(src + kq * 4);
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (!(m == m)) m = __builtin_inff();
    atomicMax(colmax + n, __float_as_uint(m));
  }
  __syncthreads();
  if (tid < COLS) {
    float iv;
    const float sc = sp_scale_for_max(__uint_as_float(colmax[tid]), &iv);
    colinv[tid] = iv;
    colmax[tid] = __float_as_uint(sc);
    const float* bsrc = tid < 96 ? g.bias : bias2;
    lds_bias[tid] = bsrc ? bsrc[b_row(tid < 96 ? tid : tid - 96)] : 0.f;
  }
  __syncthreads();
  {
    const float* colscale = reinterpret_cast<const float*>(colmax);
    for (int idx = tid; idx < COLS * (K / 4); idx += XK_NT) {
      const int n = idx / (K / 4), kq = idx - n * (K / 4);
      const float* src = n < 96 ? g.B + b_row(n) * g.ldb : B2 + b_row(n - 96) * (int64_t)K;
      const float4 v = *reinterpret_cast<const float4*>(src + kq * 4);
      const float sc = colscale[n];
      _Float16 h[4], l[4];
      sp_split(v.x * sc, h[0], l[0]); sp_split(v.y * sc, h[1], l[1]); sp_split(v.z * sc, h[2], l[2]); sp_split(v.w * sc, h[3], l[3]);
      typedef _Float16 half4v __attribute__((ext_vector_type(4)));
      *reinterpret_cast<half4v*>(lds + n * ROWB + kq * 8) = half4v{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<half4v*>(lds + PLANE_B + n * ROWB + kq * 8) = half4v{l[0], l[1], l[2], l[3]};
    }
  }
  __syncthreads();

  const int ntiles = (int)((g.M + 31) / 32);
  const int nblocks = (ntiles + XK_WAVES - 1) / XK_WAVES;
  const int Mi = (int)g.M;
  const int H = g.gru_H;
  const unsigned a_row_bytes = (unsigned)g.lda * 4u, h_row_bytes = (unsigned)H * 4u, lane_bytes = (unsigned)kg * (K / 2) * 4u;
  auto row_of = [&](int tile) {
    const int row = tile * 32 + li;
    return row < Mi ? row : Mi - 1;
  };
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)(unsigned)(g.M * g.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.gru_h, 0, (int)(unsigned)(g.M * (int64_t)H * 4), 0x00020000);
  auto a_load = [&](unsigned off, int i) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, off + i * 16, 0, 0)); };
  auto h_load = [&](unsigned off, int i) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, off + i * 16, 0, 0)); };
  unsigned b_addr[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) b_addr[p] = (unsigned)(uintptr_t)(x3k_lds_void*)lds + p * PLANE_B + li * ROWB + kg * K;
  uint4v bq[2][2];
  float* patch = lds_f32 + wave * 32 * XK_PS;
  float4 araw[2 * KS];
  int blk = stream;
  {
    const int t0 = blk * XK_WAVES + wave;
    const unsigned ao = __umul24((unsigned)row_of(t0 < ntiles ? t0 : ntiles - 1), a_row_bytes) + lane_bytes;
#pragma unroll
    for (int j = 0; j < 2 * KS; ++j) araw[j] = a_load(ao, j);
  }
  // one product of the tile: the values in araw against column tiles PH * 3 .. PH * 3 + 2 into acc; the registers a step has
  // consumed are refilled from the other operand (PH = 0: this tile's rows of h; PH = 1: the next tile's rows of x)
  auto row_scale = [&](float& inv) {
    float mx = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) {
      const float4 v = araw[i];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      bad = bad || !(v.x == v.x) || !(v.y == v.y) || !(v.z == v.z) || !(v.w == v.w);
    }
    if (bad) mx = __builtin_inff();
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    return sp_scale_for_max(mx, &inv);
  };
  for (; blk < nblocks; blk += nstreams) {
    const int tile = blk * XK_WAVES + wave;
    int next = (blk + nstreams) * XK_WAVES + wave;
    if (next >= ntiles) next = ntiles - 1;
    const unsigned ho = __umul24((unsigned)row_of(tile < ntiles ? tile : ntiles - 1), h_row_bytes) + lane_bytes;
    const unsigned no = __umul24((unsigned)row_of(next), a_row_bytes) + lane_bytes;
    floatx16 acc[2][CT];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][c][r] = 0.f;
    float a_inv[2];
    x3k_static_for<0, 2>([&](auto ph_c) {
      constexpr int PH = decltype(ph_c)::value;
      const float a_scale = row_scale(a_inv[PH]);
      x3k_request<(PH * 96) * ROWB>(bq[0], b_addr);
      x3k_static_for<0, KS>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        const float4 x0 = araw[2 * j], x1 = araw[2 * j + 1];
        const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        half8v ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          _Float16 hh, ll;
          sp_split(xs[e] * a_scale, hh, ll);
          ah[e] = hh;
          al[e] = ll;
        }
        x3k_static_for<0, CT>([&](auto c_c) {
          constexpr int c = decltype(c_c)::value, grp = j * CT + c, cur = grp & 1;
          if constexpr (grp + 1 < CT * KS) {
            constexpr int nxt = grp + 1;
            x3k_request<(PH * 96 + (nxt % CT) * 32) * ROWB + (nxt / CT) * 16>(bq[nxt & 1], b_addr);
            x3k_wait_all_but_3(bq[cur]);
          } else {
            x3k_wait_all(bq[cur]);
          }
          const half8v bh = __builtin_bit_cast(half8v, bq[cur][0]), bl = __builtin_bit_cast(half8v, bq[cur][1]);
          acc[PH][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[PH][c], 0, 0, 0);
          acc[PH][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[PH][c], 0, 0, 0);
          acc[PH][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[PH][c], 0, 0, 0);
        });
        if ((j & 3) == 3 || j == KS - 1) {
#pragma unroll
          for (int i = 2 * (j & ~3); i <= 2 * j + 1; ++i) araw[i] = PH == 0 ? h_load(ho, i) : a_load(no, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (tile >= ntiles) continue;
    // ---- the two scales and the biases, in the accumulator layout ----
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float rf[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) rf[r] = __shfl(a_inv[p], (r & 3) + 8 * (r >> 2) + 4 * kg, 64);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float ci = colinv[p * 96 + c * 32 + li], bias = lds_bias[p * 96 + c * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][c][r] = acc[p][c][r] * (rf[r] * ci) + bias;
      }
    }
    // ---- gate math: a 32 x 32 block through the patch puts four units of one row into a lane ----
    auto to_rows = [&](const floatx16& a, float4 (&out)[4]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * kg) * XK_PS + li] = a[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) out[q] = *reinterpret_cast<const float4*>(patch + ((lane >> 3) + 8 * q) * XK_PS + (lane & 7) * 4);
      __builtin_amdgcn_wave_barrier();
    };
    float4 z[4], r[4], t0[4], t1[4];
    to_rows(acc[0][0], t0);
    to_rows(acc[1][0], t1);
#define X3K_SIG(d, a, b) d.x = 1.f / (1.f + expf(-(a.x + b.x))); d.y = 1.f / (1.f + expf(-(a.y + b.y))); \
                         d.z = 1.f / (1.f + expf(-(a.z + b.z))); d.w = 1.f / (1.f + expf(-(a.w + b.w)));
#pragma unroll
    for (int q = 0; q < 4; ++q) { X3K_SIG(z[q], t0[q], t1[q]) }
    to_rows(acc[0][1], t0);
    to_rows(acc[1][1], t1);
#pragma unroll
    for (int q = 0; q < 4; ++q) { X3K_SIG(r[q], t0[q], t1[q]) }
#undef X3K_SIG
    to_rows(acc[0][2], t0);  // x_h
    to_rows(acc[1][2], t1);  // h_h
    const int64_t unit = (int64_t)cb * 32 + (lane & 7) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t row = (int64_t)tile * 32 + (lane >> 3) + 8 * q;
      if (row < g.M) {
        const float4 hv = *reinterpret_cast<const float4*>(g.gru_h + row * H + unit);
        const float4 xh = t0[q], hc = t1[q];
        float4 c, o;
#define X3K_GRU1(e)                         \
        c.e = tanhf(xh.e + r[q].e * hc.e);  \
        o.e = z[q].e * hv.e + (1.f - z[q].e) * c.e;
        X3K_GRU1(x) X3K_GRU1(y) X3K_GRU1(z) X3K_GRU1(w)
#undef X3K_GRU1
        *reinterpret_cast<float4*>(g.C + row * g.ldc + unit) = o;
        if (g.gru_gates) {
          float* pg = g.gru_gates + row * 3 * H + unit;
          *reinterpret_cast<float4*>(pg) = z[q];
          *reinterpret_cast<float4*>(pg + H) = r[q];
          *reinterpret_cast<float4*>(pg + 2 * H) = c;
        }
        if (mh_out) *reinterpret_cast<float4*>(mh_out + row * 3 * H + 2 * H + unit) = hc;
      }
    }
  }
}

template <int NPROD, int MODE>
static void launch_x3k_ks(const X3Args& g, int ks, int b_kmajor, int ncb, int spx, dim3 grid, hipStream_t s) {
  constexpr int COLS = MODE == XK_GRU ? 96 : 128;
  constexpr int NPL = NPROD == 3 ? 2 : 3, EXTRA = NPROD == 3 ? 2 * COLS * 4 : 0;  // (f16x2 form: column scales + maxima)
  const size_t lds_bytes = (size_t)NPL * COLS * (2 * ks * 16 + 16) + XK_WAVES * 32 * XK_PS * 4 + COLS * 4 + EXTRA;
#define TFGNN_X3K_CASE(KSV)                                                                                                   \
  case KSV: {                                                                                                                 \
    static const bool raised = [] {                                                                                           \
      (void)hipFuncSetAttribute((const void*)gemm_x3k_kernel<NPROD, KSV, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                NPL * COLS * (2 * KSV * 16 + 16) + XK_WAVES * 32 * XK_PS * 4 + COLS * 4 + EXTRA);             \
      return true;                                                                                                            \
    }();                                                                                                                      \
    (void)raised;                                                                                                             \
    hipLaunchKernelGGL((gemm_x3k_kernel<NPROD, KSV, MODE>), grid, dim3(XK_NT), lds_bytes, s, g, b_kmajor, ncb, spx);          \
  } break;
  switch (ks) {
    TFGNN_X3K_CASE(2)
    TFGNN_X3K_CASE(4)
    TFGNN_X3K_CASE(6)
    TFGNN_X3K_CASE(8)
    default: break;
  }
#undef TFGNN_X3K_CASE
}

int gemm_f16x2_mode();
// mode f16x2 (while its guard has not demoted it): the streaming kernel multiplies in the f16x2 arithmetic, 3 products instead of
// 6 (TFGNN_X3K_F16=0: the exact bf16x3 form, for A/B runs)
static bool x3k_f16_arithmetic(int nprod) {
  static const bool on = [] { const char* e = getenv("TFGNN_X3K_F16"); return !e || atoi(e) != 0; }();
  return on && nprod == 6 && gemm_f16x2_mode() == 1;
}

static int64_t x3k_min_rows() {
  static const int64_t min_rows = [] { const char* e = getenv("TFGNN_X3_STREAM_MIN_ROWS"); return e ? atoll(e) : 65536ll; }();
  return min_rows;
}
// shape limits shared by the forms of the streaming kernel: K in {32, 64, 96, 128}, many rows, 32-bit byte offsets and 24-bit
// factors for the streamed operand
static bool x3k_shape_ok(const X3Args& g) {
  if (x3k_min_rows() <= 0 || g.M < x3k_min_rows() || g.K % 32 || g.K < 32 || g.K > 128) return false;
  const int64_t src_rows = g.a_index ? g.a_rows : g.M;
  return src_rows * g.lda < (1ll << 30) - 2048 && src_rows < (1 << 24) && g.M < (1ll << 31) - 64 && g.lda < (1 << 22);
}

// 1 = the streaming kernel took the product (NN / NT, K in {32, 64, 96, 128}, N a multiple of 128, many rows)
static int gemm_x3k_try(int nprod, int trans_b, const X3Args& g, hipStream_t s) {
  if (!x3k_shape_ok(g) || g.N % XK_COLS) return 0;
  const int ncb = (int)(g.N / XK_COLS);
  if (ncb > 32) return 0;
  const int spx = 32 / ncb;  // row streams per XCD (32 CUs each)
  const bool f16 = x3k_f16_arithmetic(nprod);
  count_launch(f16 ? TFGNN_KFAM_STREAM_F16X2 : TFGNN_KFAM_GEMM_BF16X3);
  count_launch(TFGNN_KFAM_GEMM_STREAM);
  dim3 grid((unsigned)(8 * spx * ncb));
  const bool extras = g.mul || g.saved || g.accumulate;  // the plain forward kernels carry none of that code
  const int ks = (int)(g.K / 16), bkm = trans_b ? 0 : 1;
  if (f16) {
    if (extras) launch_x3k_ks<3, XK_EXTRAS>(g, ks, bkm, ncb, spx, grid, s);
    else launch_x3k_ks<3, XK_PLAIN>(g, ks, bkm, ncb, spx, grid, s);
  } else if (nprod >= 9) {
    if (extras) launch_x3k_ks<9, XK_EXTRAS>(g, ks, bkm, ncb, spx, grid, s);
    else launch_x3k_ks<9, XK_PLAIN>(g, ks, bkm, ncb, spx, grid, s);
  } else {
    if (extras) launch_x3k_ks<6, XK_EXTRAS>(g, ks, bkm, ncb, spx, grid, s);
    else launch_x3k_ks<6, XK_PLAIN>(g, ks, bkm, ncb, spx, grid, s);
  }
  return 1;
}

// the GRU form (g filled by gemm_x3_gru: B = the regrouped kernel [3H, K], C = h' [M, H]): one workgroup per 32 units
static int gemm_x3k_gru_try(int nprod, const X3Args& g, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("TFGNN_X3_STREAM_GRU"); return !e || atoi(e) != 0; }();  // 0: A/B probe
  const int H = g.gru_H;
  if (!on || !x3k_shape_ok(g) || H % 64 || H / 32 > 32 || g.ldc % 4) return 0;
  const int ncb = H / 32, spx = 32 / ncb;
  const bool f16 = x3k_f16_arithmetic(nprod);
  count_launch(f16 ? TFGNN_KFAM_STREAM_F16X2 : TFGNN_KFAM_GEMM_BF16X3);
  count_launch(TFGNN_KFAM_GEMM_STREAM);
  dim3 grid((unsigned)(8 * spx * ncb));
  if (f16) launch_x3k_ks<3, XK_GRU>(g, (int)(g.K / 16), 0, ncb, spx, grid, s);
  else if (nprod >= 9) launch_x3k_ks<9, XK_GRU>(g, (int)(g.K / 16), 0, ncb, spx, grid, s);
  else launch_x3k_ks<6, XK_GRU>(g, (int)(g.K / 16), 0, ncb, spx, grid, s);
  return 1;
}

// 0 = off (fp32 MFMA), 6 / 9 = number of piece products of the tfgnn_gemm* entry points.  Initialised from TFGNN_GEMM_MODE
// (fp32 | bf16x3 | bf16x3_9 | f16x2; unset = f16x2, the mode the benchmark is timed in and - since round 3 - the whole parity
// suite runs in), changed at run time by tfgnn_gemm_set_mode().  "f16x2" is a mode of the LIBRARY since round 4: the
// tfgnn_gemm* entry points then run as bf16x3 and tfgnn_gemm_get_mode() tells a binder to hand the products that have a
// split-operand producer to the tfgnn_sp_* entry points - until the spread guard of the weight-gradient product
// (tfgnn_sp_spread_flag) trips, from where on it reports TFGNN_GEMM_BF16X3 (sticky; tfgnn_gemm_set_mode re-arms).
static int g_x3_mode = -1;
static int g_f16x2 = -1;
static void mode_from_env() {
  const char* e = getenv("TFGNN_GEMM_MODE");
  g_f16x2 = (!e || !*e || !strcmp(e, "f16x2")) ? 1 : 0;
  if (e && !strcmp(e, "fp32")) g_x3_mode = 0;
  else if (e && !strcmp(e, "bf16x3_9")) g_x3_mode = 9;
  else g_x3_mode = 6;  // bf16x3, bf16x3_6, f16x2
}
int gemm_x3_mode() {
  if (g_x3_mode < 0) mode_from_env();
  return g_x3_mode;
}
int gemm_x3_set_mode(int mode) {
  const int prev = gemm_x3_mode();
  g_x3_mode = mode;
  return prev;
}
int gemm_f16x2_mode() {
  if (g_f16x2 < 0) mode_from_env();
  return g_f16x2;
}
void gemm_f16x2_set(int on) {
  (void)gemm_x3_mode();
  g_f16x2 = on ? 1 : 0;
}

// returns 1 if it took the call, 0 if the shape / layout is not covered (caller falls back to gemm.hip)
int gemm_x3_try(int nprod, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int accumulate,
                void* workspace, size_t workspace_bytes, hipStream_t s, int* status, const float* mul, int64_t ld_mul,
                const float* saved, int64_t ld_saved, int dact) {
  *status = TFGNN_OK;
  if (mul || saved) {  // the gradient epilogue lives in the specialised kernel (NN / NT) only
    if (trans_a) return 0;
    if ((mul && (((uintptr_t)mul % 16) || ld_mul % 4)) || (saved && (((uintptr_t)saved % 16) || ld_saved % 4))) return 0;
  }
  // 128 x {320, 256, 128} tiles (N a multiple of one of them), 16-byte aligned operands, supported layout pairs:
  //   NN (A [M,K], B [K,N]), NT (A [M,K], B [N,K]), TN (A [K,M], B [K,N])
  if (trans_a && trans_b) return 0;
  int bn = N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : (N % 128 == 0 ? 128 : 0));
  // short K, many row blocks (QM9-sized batches): the 128-wide tile runs two workgroups per CU, which overlaps one's
  // prologue / output burst with the other's few K tiles
  static const int narrow_k = [] { const char* e = getenv("TFGNN_X3_NARROW_K"); return e ? atoi(e) : 256; }();
  if (!trans_a && K <= narrow_k && N % 128 == 0 && M >= 128 * 1024) bn = 128;
  if (!bn || K < 64 || M < 1) return 0;
  const bool a16 = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
  const bool b16 = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
  // K % 4: only where an operand is K-contiguous (16-byte fetches along K); K-major operands mask whole k rows
  const bool k_aligned = K % 4 == 0 || (trans_a && !trans_b);
  if (!a16 || !b16 || !k_aligned || (trans_a && M % 4 != 0) || (ldc % 4 != 0) || ((uintptr_t)C % 16 != 0)) return 0;
  if (bias && (uintptr_t)bias % 16 != 0) return 0;
  X3Args g;
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.act = act; g.accumulate = accumulate;
  g.mul = mul; g.ld_mul = ld_mul; g.saved = saved; g.ld_saved = ld_saved; g.dact = dact;
  g.group_mode = 0; g.group_off = nullptr; g.strideB = 0; g.strideC = 0;
  g.splits = 1; g.partial = nullptr;
  g.a_index = nullptr; g.a_rows = 0;
  if (!trans_a && gemm_x3k_try(nprod, trans_b, g, s)) {
    if (hipGetLastError() != hipSuccess) {
      set_error("bf16x3 streaming GEMM launch failed");
      *status = TFGNN_ERR_HIP;
    }
    return 1;
  }
  g.n_tiles = (unsigned)(N / bn);
  const int64_t tiles = ceil_div(M, 128) * (int64_t)g.n_tiles;
  g.splits = 1;
  g.k_chunk = ceil_div(K, X3_BK) * X3_BK;
  if (tiles < 192 && K >= 1024 && workspace) {
    int64_t want = splitk_want(tiles, K);
    const int64_t max_by_k = K / 128;
    if (want > max_by_k) want = max_by_k;
    // (exact division: with "+ 1" a workspace of exactly tfgnn_gemm_workspace_bytes() bytes allowed one split fewer than a larger
    //  one - the same product then summed in another order depending on which buffer the caller happened to hold: round 6)
    const int64_t max_by_ws = (int64_t)(workspace_bytes / ((size_t)(M * N) * 4));
    if (want > max_by_ws) want = max_by_ws;
    if (want > 1 && (uintptr_t)workspace % 16 == 0) {
      g.k_chunk = ceil_div(ceil_div(K, want), X3_BK) * X3_BK;
      g.splits = (int)ceil_div(K, g.k_chunk);
    }
  }
  g.partial = (float*)workspace;
  dim3 grid((unsigned)tiles, 1, (unsigned)g.splits);
  g.per_xcd = 0; g.tiles_x = (unsigned)tiles;
  if ((g.n_tiles > 1 || g.splits > 1) && tiles * g.splits < (1ll << 30)) {  // tiles share operand rows: XCD-aware order (x3_tile)
    g.per_xcd = (unsigned)ceil_div(tiles * g.splits, 8);
    grid = dim3(8 * g.per_xcd, 1, 1);
  }
  if (bn == 320) launch_x3<5>(g, grid, nprod, trans_a, trans_b, s);
  else if (bn == 256) launch_x3<4>(g, grid, nprod, trans_a, trans_b, s);
  else launch_x3<2>(g, grid, nprod, trans_a, trans_b, s);
  if (hipGetLastError() != hipSuccess) {
    set_error("bf16x3 GEMM launch failed");
    *status = TFGNN_ERR_HIP;
    return 1;
  }
  if (g.splits > 1) {
    launch_x3_splitk_reduce(g, 1, s);
  }
  return 1;
}

// C[m] = act?( (C[m] if accumulate) + A[index[m]] @ op(B) + bias ) on the streaming kernel; accumulate: 0 no, 1 after the
// activation, 2 before it.  1 = taken, 0 = shape not covered (caller gathers and multiplies separately).
// would gemm_x3_gathered_try take a product of this shape (aligned operands assumed)?  The layers ask before they choose the
// one-message-per-edge formulation (the shape conditions used to be copied into the host mirror)
int gemm_x3_gathered_supported(int nprod, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t a_rows) {
  if (!nprod) return 0;
  X3Args g{};
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.a_rows = a_rows;
  g.a_index = reinterpret_cast<const int32_t*>(1);  // "indexed": the row limits apply to a_rows
  return x3k_shape_ok(g) && N % XK_COLS == 0 && N / XK_COLS <= 32 && lda % 4 == 0 ? 1 : 0;
}

int gemm_x3_gathered_try(int nprod, int trans_b, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, int64_t a_rows,
                         const int32_t* index, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int act,
                         int accumulate, hipStream_t s, int* status) {
  *status = TFGNN_OK;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 || lda % 4 || ldb % 4 || ldc % 4) return 0;
  X3Args g{};
  g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.act = act; g.accumulate = accumulate; g.splits = 1;
  g.a_index = index; g.a_rows = a_rows;
  if (!gemm_x3k_try(nprod, trans_b, g, s)) return 0;
  if (hipGetLastError() != hipSuccess) {
    set_error("bf16x3 streaming GEMM launch failed");
    *status = TFGNN_ERR_HIP;
  }
  return 1;
}

// h' = GRU(mx = A Bt^T + bias, mh, h) with the gate math in the epilogue (mx is never written).  Bt: [3H, K]
// K-contiguous with its rows regrouped per 192-row tile as [z | r | h] of 64 units (bias likewise).  1 = taken.
int gemm_x3_gru(int nprod, int64_t M, int H, int64_t K, const float* A, int64_t lda, const float* Bt, const float* bias,
                const float* mh, const float* h, float* h_new, float* gates, hipStream_t s) {
  if (H % 64 != 0 || K < 64 || K % 4 != 0 || M < 1 || lda % 4 != 0) return 0;
  for (const void* ptr : {(const void*)A, (const void*)Bt, (const void*)mh, (const void*)h, (const void*)h_new})
    if ((uintptr_t)ptr % 16) return 0;
  if ((bias && (uintptr_t)bias % 16) || (gates && (uintptr_t)gates % 16)) return 0;
  X3Args g{};
  g.M = M; g.N = 3 * (int64_t)H; g.K = K; g.A = A; g.lda = lda; g.B = Bt; g.ldb = K; g.C = h_new; g.ldc = H;
  g.bias = bias; g.act = TFGNN_ACT_NONE; g.splits = 1; g.k_chunk = ceil_div(K, X3_BK) * X3_BK;
  g.n_tiles = (unsigned)(3 * H / 192);
  g.gru_mh = mh; g.gru_h = h; g.gru_gates = gates; g.gru_H = H;
  if (gemm_x3k_gru_try(nprod, g, s)) return 1;  // QM9-sized batches: the weight block stays in LDS, rows stream
  const int64_t tiles = ceil_div(M, X3_BM) * g.n_tiles;
  if (tiles > 0x3fffffff) return 0;
  dim3 grid((unsigned)tiles, 1, 1);
  if (g.n_tiles > 1) {  // the column tiles of a row tile share the A rows: XCD-aware order (x3_tile)
    g.tiles_x = (unsigned)tiles;
    g.per_xcd = (unsigned)ceil_div(tiles, 8);
    grid = dim3(8 * g.per_xcd, 1, 1);
  }
  count_launch(TFGNN_KFAM_GEMM_BF16X3);
  if (nprod >= 9) hipLaunchKernelGGL((gemm_x3s_kernel<false, 9, 3, 1>), grid, dim3(X3_NT), 0, s, g);
  else hipLaunchKernelGGL((gemm_x3s_kernel<false, 6, 3, 1>), grid, dim3(X3_NT), 0, s, g);
  return 1;
}

// GRUCell with both products in the streaming kernel (gemm_x3k_gru2_kernel): 1 = taken.  Mode f16x2 only, K == H in {64, 128},
// QM9-sized row counts (x3k_shape_ok).
int gemm_x3_gru2(int nprod, int64_t M, int H, int64_t K, const float* A, int64_t lda, const float* Bt, const float* bias, const float* h,
                 const float* B2t, const float* bias2, float* h_new, float* gates, float* mh_out, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("TFGNN_X3_STREAM_GRU2"); return !e || atoi(e) != 0; }();  // 0: A/B probe
  if (!on || !x3k_f16_arithmetic(nprod) || K != H || (H != 64 && H != 128) || M < 1 || lda % 4 != 0) return 0;
  for (const void* ptr : {(const void*)A, (const void*)Bt, (const void*)B2t, (const void*)h, (const void*)h_new})
    if ((uintptr_t)ptr % 16) return 0;
  if ((bias && (uintptr_t)bias % 16) || (bias2 && (uintptr_t)bias2 % 16) || (gates && (uintptr_t)gates % 16) || (mh_out && (uintptr_t)mh_out % 16))
    return 0;
  X3Args g{};
  g.M = M; g.N = 3 * (int64_t)H; g.K = K; g.A = A; g.lda = lda; g.B = Bt; g.ldb = K; g.C = h_new; g.ldc = H;
  g.bias = bias; g.act = TFGNN_ACT_NONE; g.splits = 1; g.k_chunk = ceil_div(K, X3_BK) * X3_BK;
  g.gru_mh = nullptr; g.gru_h = h; g.gru_gates = gates; g.gru_H = H;
  if (!x3k_shape_ok(g) || M * (int64_t)H >= (1ll << 30) - 2048) return 0;
  const int ncb = H / 32, spx = 32 / ncb;
  count_launch(TFGNN_KFAM_STREAM_F16X2);
  count_launch(TFGNN_KFAM_GEMM_STREAM);
  dim3 grid((unsigned)(8 * spx * ncb));
  const int ks = (int)(K / 16);
  const size_t lds_bytes = (size_t)2 * 192 * (2 * ks * 16 + 16) + XK_WAVES * 32 * XK_PS * 4 + 3 * 192 * 4;
  if (ks == 8) {
    static const bool raised = [] {
      (void)hipFuncSetAttribute((const void*)gemm_x3k_gru2_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * 192 * (2 * 128 + 16) + XK_WAVES * 32 * XK_PS * 4 + 3 * 192 * 4);
      return true;
    }();
    (void)raised;
    hipLaunchKernelGGL((gemm_x3k_gru2_kernel<8>), grid, dim3(XK_NT), lds_bytes, s, g, B2t, bias2, mh_out, ncb, spx);
  } else {
    static const bool raised = [] {
      (void)hipFuncSetAttribute((const void*)gemm_x3k_gru2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                2 * 192 * (2 * 64 + 16) + XK_WAVES * 32 * XK_PS * 4 + 3 * 192 * 4);
      return true;
    }();
    (void)raised;
    hipLaunchKernelGGL((gemm_x3k_gru2_kernel<4>), grid, dim3(XK_NT), lds_bytes, s, g, B2t, bias2, mh_out, ncb, spx);
  }
  return 1;
}

static int x3_tile_width(int64_t N) { return N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : (N % 128 == 0 ? 128 : 0)); }

static void launch_bn(const X3Args& g, dim3 grid, int bn, int nprod, int trans_a, int trans_b, hipStream_t s) {
  if (bn == 320) launch_x3<5>(g, grid, nprod, trans_a, trans_b, s);
  else if (bn == 256) launch_x3<4>(g, grid, nprod, trans_a, trans_b, s);
  else launch_x3<2>(g, grid, nprod, trans_a, trans_b, s);
}

// C[rows g] = act(A[rows g] @ op(B + g stride_b)) on the specialised kernel; 1 = taken
int gemm_x3_try_grouped_rows(int nprod, int trans_b, int num_groups, const int32_t* group_off, int64_t max_rows, int64_t N,
                             int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, int64_t stride_b,
                             float* C, int64_t ldc, int act, hipStream_t s, int* status, int dact, const float* saved,
                             int64_t ld_saved) {
  *status = TFGNN_OK;
  const int bn = x3_tile_width(N);
  if (!bn || K < 64 || K % 4 || lda % 4 || ldb % 4 || ldc % 4 || stride_b % 4) return 0;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16) return 0;
  if (saved && (ld_saved % 4 || (uintptr_t)saved % 16)) return 0;
  X3Args g{};
  g.M = max_rows; g.N = N; g.K = K; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.saved = saved; g.ld_saved = ld_saved; g.dact = dact;
  g.act = act; g.splits = 1; g.k_chunk = ceil_div(K, X3_BK) * X3_BK;
  g.group_mode = 1; g.group_off = group_off; g.strideB = stride_b;
  g.n_tiles = (unsigned)(N / bn);
  dim3 grid((unsigned)(ceil_div(max_rows, X3_BM) * g.n_tiles), (unsigned)num_groups, 1);
  launch_bn(g, grid, bn, nprod, 0, trans_b, s);
  if (hipGetLastError() != hipSuccess) {
    set_error("bf16x3 grouped GEMM launch failed");
    *status = TFGNN_ERR_HIP;
  }
  return 1;
}

// C + g stride_c = A[rows g]^T @ B[rows g] on the pipelined kernel (+ deterministic split-K over the group's rows)
int gemm_x3_try_grouped_k(int nprod, int num_groups, const int32_t* group_off, int64_t max_rows, int64_t M, int64_t N,
                          const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                          int64_t stride_c, void* workspace, size_t workspace_bytes, hipStream_t s, int* status) {
  *status = TFGNN_OK;
  const int bn = x3_tile_width(N);
  if (!bn || M % 4 || lda % 4 || ldb % 4 || ldc % 4 || stride_c % 4 || max_rows < 64) return 0;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16) return 0;
  X3Args g{};
  g.M = M; g.N = N; g.K = max_rows; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.n_tiles = (unsigned)(N / bn);
  const int64_t tiles = ceil_div(M, X3_BM) * (int64_t)g.n_tiles * num_groups;
  int64_t splits = ceil_div(512, tiles > 0 ? tiles : 1);
  splits = std::min<int64_t>(splits, max_rows / 128);
  splits = std::max<int64_t>(splits, ceil_div(max_rows, 16384));  // accuracy: fp32 chains of <= 16384 rows (gemm.hip grouped_splits)
  splits = std::min<int64_t>(splits, 64);
  const size_t per_split = (size_t)num_groups * (size_t)M * (size_t)N * 4;
  if (splits > 1 && (!workspace || (uintptr_t)workspace % 16 || workspace_bytes < 2 * per_split)) splits = 1;
  if (splits > 1) splits = std::min<int64_t>(splits, (int64_t)(workspace_bytes / per_split));
  if (splits < 1) splits = 1;
  g.splits = (int)splits; g.k_chunk = 0; g.partial = (float*)workspace;
  g.group_mode = 2; g.group_off = group_off; g.strideC = stride_c;
  dim3 grid((unsigned)(ceil_div(M, X3_BM) * g.n_tiles), (unsigned)num_groups, (unsigned)splits);
  launch_bn(g, grid, bn, nprod, 1, 0, s);
  if (hipGetLastError() != hipSuccess) {
    set_error("bf16x3 grouped GEMM launch failed");
    *status = TFGNN_ERR_HIP;
    return 1;
  }
  if (splits > 1) {
    dim3 rgrid((unsigned)std::min<int64_t>(ceil_div(M * N, 256), 1024), (unsigned)num_groups);
    hipLaunchKernelGGL(x3_splitk_reduce_kernel, rgrid, dim3(256), 0, s, g);
  }
  return 1;
}

}  // namespace tfgnn
