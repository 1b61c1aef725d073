// fp32 GEMM on the fp16 matrix cores from PRE-SPLIT operands ("f16x2" / SP16 operand format).
//
// Numerics.  Every fp32 operand value x of a row (or of a block of a row) is scaled by a power of two 2^e chosen
// so that the block maximum lies in [2^14, 2^15), and the scaled value is split by two round-to-nearest
// conversions  xs = h + l + r,  h = fp16(xs),  l = fp16(xs - h)  (xs - h is exact in fp32).  |r| <= 2^-22 |xs| for
// every element within 2^-3 of the block maximum (l normal), and |r| <= 2^-25 (absolute, in units where the block
// maximum is 2^14..2^15, i.e. <= 2^-39 of the block maximum) below that.  A product a*b is evaluated as
// l_a h_b + h_a l_b + h_a h_b on v_mfma_f32_32x32x16_f16: each piece product has 11 x 11 significand bits and is
// exact in fp32, the accumulation is fp32, the dropped l_a l_b term is <= 2^-22 |a b|.  The power-of-two scales
// are removed exactly in the epilogue.  Measured against fp64 the result is in the error class of the fp32-MFMA
// kernel (tests/test_gpu_gemm_sp.py: same max error within 10 % on N(0,1), relu-sparse and wide-dynamic-range
// operands): the fp32 accumulation rounds more than the operand representation.  fp32 in, fp32 out.
//
// Operand format SP16 (the same number of bytes as the fp32 matrix, so it can be produced in place of it by the
// kernel that computes the operand - the gather writes its sums this way, sp_split_* converts anything else):
//   row r, column c  ->  byte  r * ld_bytes + (c / 16) * 64 + plane * 32 + (c % 16) * 2,   plane 0 = h, 1 = l
// i.e. per row and per group of 16 columns one 64-byte granule [16 x h | 16 x l].  Scales: inv[r][c / SB] = 2^-e
// (fp32), SB = columns per scale block (a multiple of 16; SB = C: one scale per row).
//
// NT kernel (both operands K-contiguous: activations [M, K] against weights kept as [N, K]):
//   128 x BN output tile per 256-thread workgroup (BN = 320 / 256 / 128), one wave per SIMD, each wave a
//   64 x BN/2 block of 32x32 MFMA tiles (160 accumulator registers at BN = 320).  K advances in steps of 16 (one
//   MFMA k-step = one granule per row); a ring of five or six LDS stages is filled by LDS-DMA only
//   (buffer_load_dwordx4 ... lds: no VALU, no registers), four or five steps ahead; the per-lane SOURCE address is
//   permuted so that the lane-linear LDS image is XOR-swizzled and every ds_read_b128 fragment read is
//   conflict-free.  Fragments of step s+1 are read while step s multiplies (two register sets), so the one
//   barrier per step only orders buffer reuse.  Epilogue: scales, bias, activation, gradient factors, through a
//   wave-private LDS patch so that every lane stores 16 contiguous bytes.
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "aux_jobs.hpp"
#include "common.hpp"
#include "sp16.hpp"

namespace tfgnn {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int SP_BM = 128;
constexpr int SP_NT = 256;

struct SpArgs {
  int64_t M, N, K;
  const uint8_t* A;   // SP16 [M rows]
  int64_t lda;        // bytes per row
  const float* a_inv; // 2^-e of (row r, scale block b) at a_inv[r * a_inv_ld + b], or NULL (all 1); a_inv_ld = 0: one
  int64_t a_inv_ld;   // scale for the whole tensor
  int a_nblk;
  int a_blk_steps;    // k16 steps per scale block
  const uint8_t* B;   // SP16 [N rows]
  int64_t ldb;
  const float* b_inv; // [N] or NULL
  float* C;
  int64_t ldc;
  const float* bias;
  int act;
  int accumulate;
  const float* mul;
  int64_t ld_mul;
  const float* saved;
  int64_t ld_saved;
  int dact;
  unsigned n_tiles;
  // OUT_SP kernels: the result also (or only, C == NULL) as an SP16 operand with one scale per row (N == tile width)
  uint8_t* out_sp;
  int64_t ld_out_sp;
  float* out_inv;  // [M]
  // dropout in the epilogue (drop_on = 1): the result - after bias, activation and the gradient factors - times the mask of
  // element index row * drop_ld + column (common.hpp dropout_mask_at: the mask tfgnn_dropout_forward draws for that seed).
  // drop_on = 2 (gradient products, relu): the saved tensor IS the dropped relu output - it is positive exactly where the
  // unit was kept and active, so relu'(saved) already carries the mask's zeros and only 1 / (1 - rate) is left to apply
  int drop_on;
  int64_t drop_ld;
  DropoutKey drop;
  float saved_scale;  // act'(saved * saved_scale): the saved tensor is a dropped activation (scaled by 1/(1-rate) where kept)
  // zero-block skipping (round 4): bit b of tile_kmask[row tile] = scale block b of A holds non-zeros in that tile; the other
  // blocks' k16 steps are not executed.  row_map: output row of product row r (the operand's rows are in pattern order)
  const uint8_t* tile_kmask;
  const int32_t* row_map;
  // operand rows through an index (round 5): product row r multiplies row a_rows[r] of A (and takes that row's scales) - the
  // by-source sums stay in node order for the weight-gradient product while the input-gradient product walks them in the
  // order of the by-source emptiness patterns (tile_kmask of THAT order) and writes node order again through row_map
  const int32_t* a_rows;
  int64_t a_src_rows;  // rows of the operand a_rows points into (its descriptor spans them); = M without a_rows
  // grouped rows (round 5; the per-relation products of RGIN / GNN_Edge_MLP over the non-empty (source, type) rows, grouped by
  // type): tile_table[tile] = {first row, rows (<= 128), group, -}; group gr multiplies B + gr * group_stride_b (its own
  // [N, K] weight operand) with column scales / bias at offset gr * N.  A row tile never straddles two groups.
  const int32_t* tile_table;
  int64_t group_stride_b;      // bytes between the weight operands of consecutive groups
  int64_t group_stride_scale;  // elements between their column scales / biases (0: shared)
  // K split inside the launch (round 5, few row tiles: a batch of some thousand nodes leaves most CUs idle while 56
  // workgroups each stream the whole weight operand).  ksplit = S > 1: the grid holds S workgroups per output tile, each
  // multiplies 1/S of the tile's k16 steps; splits 1 .. S-1 publish their raw accumulators (write-through stores into
  // ws_partial + a flag word), split 0 adds them IN SPLIT ORDER (bit-reproducible) and runs the one epilogue.
  int ksplit;
  float* ws_partial;    // [tiles][S - 1][4 waves x 2 x TNW x 16 x 64 floats]
  unsigned* ws_flags;   // [tiles][S - 1], zero when the kernel starts (split 0 clears what it consumed)
  int* ws_timeout;      // host-mapped: set to 1 if a reducer gave up waiting (never expected; the result is then wrong)
  // XCD-aware tile order (round 6, products over several column tiles: N = 512 / 1024).  Workgroup b runs on XCD b % 8; with
  // the column tile as the fast index the two (four) column tiles of a row tile sat on DIFFERENT XCDs and each fetched the
  // row tile's A rows over the fabric (counters, configs[4]: 2 132 MB fetched per grouped launch for 927 MB of A).  xcd_per
  // != 0: tile = (b % 8) * xcd_per + b / 8 - an XCD walks whole row tiles, their column tiles back to back - and workgroups
  // past xcd_total leave at once (the grid is padded to 8 xcd_per).  Set by launch_sp_nt (never together with ksplit).
  unsigned xcd_per, xcd_total;
};

// ------------------------------------------------------------------------------------------------------
// fp32 -> SP16 conversion
// ------------------------------------------------------------------------------------------------------
// (the conversion bodies live in aux_jobs.hpp: they also run as jobs of the merged small-pass launch, tfgnn_aux_launch)
__global__ void __launch_bounds__(256) sp_split_rows_kernel(const float* __restrict__ src, int64_t ld, int64_t seg_len,
                                                            int64_t seg_stride, int64_t R, int64_t C, int sb,
                                                            uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ inv,
                                                            const float* __restrict__ fixed_inv) {
  sp_split_rows_body(src, ld, seg_len, seg_stride, R, C, sb, dst, ld_dst, inv, fixed_inv, blockIdx.x);
}

// Plain row-major rows of 64 / 128 / 256 columns with one scale per row (node states, gate gradients: [10^6, 128]): C / 4 lanes
// per row, one float4 per lane held in registers between the row maximum and the store - one pass and full waves, where the
// general kernel above spends a half-empty wave and two passes per row (0.50 -> ~0.2 ms for [1.15 M, 128]).  Same scale
// function and store: bit-identical output.
template <int LPR>
__global__ void __launch_bounds__(256) sp_split_rows_narrow_kernel(const float* __restrict__ src, int64_t ld, int64_t R,
                                                                   uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ inv) {
  constexpr int RPB = 256 / LPR;  // rows per workgroup and round
  const int sub = threadIdx.x % LPR;
  for (int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; r < R; r += (int64_t)gridDim.x * RPB) {
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld + sub * 4);
    float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
    for (int o = LPR / 2; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float iv;
    const float sc = sp_scale_for_max(mx, &iv);
    if (inv && sub == 0) inv[r] = iv;
    sp_store4(dst + r * ld_dst, sub * 4, v, sc);
  }
}

// out[0] = max(out[0], scale * max |x|): the atomic maximum of non-negative floats through their bit patterns is
// independent of the order -> reproducible.  out must start at 0.
__global__ void __launch_bounds__(256) sp_absmax_kernel(const float* __restrict__ x, int64_t n, float scale, float* out) {
  float mx = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) mx = fmaxf(mx, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  __shared__ float wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)  // one atomic per workgroup: thousands on one address serialise (8192 of them cost 80 us)
    atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])) * scale));
}

__global__ void sp_inv_scale_kernel(const float* bound, float* inv) {
  float iv;
  (void)sp_scale_for_max(fabsf(bound[0]), &iv);
  inv[0] = iv;
}

__global__ void __launch_bounds__(256) sp_split_cols_kernel(const float* __restrict__ src, int64_t ld, int64_t K, int64_t N,
                                                            uint8_t* __restrict__ dst, int64_t ld_dst, float* __restrict__ inv) {
  sp_split_cols_body(src, ld, K, N, dst, ld_dst, inv, blockIdx.x, blockIdx.y, gridDim.y);
}

// Both operand forms of up to 16 stacked kernels [L, D, H] of the same shape in ONE launch (the per-step weight
// preparation of a layer stack: 2 x layers small launches otherwise, each latency-bound): blockIdx.y = kernel stack;
// workgroups [0, nc) do slice (x % ncx, x / ncx) of the transposed form [H, L D], the rest the stacked-rows form
// [D, L H] (row d = [W_0[d, :] | W_1[d, :] | ...]).
struct SpWeightJobs {
  const float* src[16];
  uint8_t* cols_sp[16];
  float* cols_inv[16];
  uint8_t* rows_sp[16];
  float* rows_inv[16];
  int64_t L, D, H;
  unsigned ncx, ncy;
};
__global__ void __launch_bounds__(256) sp_split_weights_kernel(SpWeightJobs j) {
  const unsigned w = blockIdx.y, nc = j.ncx * j.ncy;
  if (blockIdx.x < nc) {
    sp_split_cols_body(j.src[w], j.H, j.L * j.D, j.H, j.cols_sp[w], j.L * j.D * 4, j.cols_inv[w], blockIdx.x % j.ncx, blockIdx.x / j.ncx, j.ncy);
  } else {
    sp_split_rows_body(j.src[w], j.H, j.H, j.D * j.H, j.D, j.L * j.H, (int)(j.L * j.H), j.rows_sp[w], j.L * j.H * 4, j.rows_inv[w], nullptr,
                       blockIdx.x - nc);
  }
}

// ------------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------------
template <int TNW>
struct SpGeo {
  static constexpr int BN = 64 * TNW;
  static constexpr int ROWS = SP_BM + BN;      // rows per stage
  static constexpr int STG = ROWS * 64;        // bytes per stage
  static constexpr int ND_A = 2;               // DMA instructions per wave and step for A (8 x 16 rows / 4 waves)
  static constexpr int ND_B = TNW;             // ... for B (4 TNW x 16 rows / 4 waves)
  static constexpr int ND = ND_A + ND_B;
  static constexpr int PATCH_LD = 32 * TNW + 4;  // floats per patch row (wave-private epilogue patch: 32 rows)
  // LDS ring: one k16 step per stage, as many stages as 160 KB hold (at most 6).  The DMA of a step is issued NST-1
  // steps before the step and has to have landed two steps before it: NST - 3 steps (and the rest of the issuing
  // one) cover its latency - ~1 us under load, a step is ~0.45 us (tools/sp_ablate.py: with four stages the loop
  // waited for its DMAs).
  // BN = 128 (round 5): 4 stages = 64 KB, so that TWO workgroups share a CU (the kernel needs <= 228 registers there: two waves
  // per SIMD) and one tile's epilogue / ring fill runs under the other's main loop - the 10^6-row products of configs[3]
  // (K = 640 .. 2560, N = 128) spend a third of a tile's time outside the main loop.  TFGNN_SP_NT_NARROW_STAGES (build flag)
  // restores 6 for A/B runs.
#ifndef TFGNN_SP_NT_NARROW_STAGES
#define TFGNN_SP_NT_NARROW_STAGES 4
#endif
  static constexpr int NST_MAX = TNW == 2 ? TFGNN_SP_NT_NARROW_STAGES : 6;
  static constexpr int NST = (163840 / STG) < NST_MAX ? (163840 / STG) : NST_MAX;
  static constexpr int UNR = NST % 2 == 0 ? NST : 2 * NST;  // steps per unrolled loop body (register sets alternate)
  static constexpr int VMW = (NST - 3) * ND;                // DMAs that may still be in flight at the end of a step
  static constexpr int LDS_BYTES = NST * STG;
  static_assert(NST >= 4 && VMW < 64, "ring depth");
  static_assert(4 * 32 * PATCH_LD * 4 <= LDS_BYTES, "epilogue patch must fit the ring");
};

// Register state of the main loop.  The loop is pinned instruction by instruction with small volatile asm statements
// (hipcc's scheduler otherwise sinks the fragment reads to their uses and puts dependent MFMAs back to back); the
// member templates have every index as a template argument so that each operand is a fixed register.
template <int TNW>
struct SpLoop {
  using G = SpGeo<TNW>;
  // references to arrays that live in the kernel's frame (one alloca each: the optimizer's scalar replacement gives
  // up on one big aggregate with this many uses and would keep the whole state in scratch memory)
  half8 (&fa)[2][2][2];    // [set][row tile][plane]
  half8 (&fb)[2][TNW][2];  // [set][col tile][plane]
  floatx16 (&acc)[2][TNW];  // one accumulator for the three piece products (tools/mfma_acc_probe.hip, K = 1280: rms
                            // accumulation error 7.0e-7, fp32-MFMA chain 1.14e-6, bf16x3 9.8e-7; a second accumulator for
                            // the l*h + h*l terms would give 4.1e-7 but 320 accumulator registers exceed the 256 AGPRs)
  unsigned (&a_addr)[G::NST][2], (&b_addr)[G::NST][2];  // LDS byte address of this lane's fragment slot per stage / plane
  unsigned (&voff_a)[G::ND_A], (&voff_b)[G::ND_B];      // DMA source offsets of this lane
  uint4v rs_a, rs_b;       // buffer descriptors of this tile's A / B rows (wave-uniform)
  unsigned m0_a, m0_b;     // LDS byte address of this wave's first DMA slot in stage 0 (A part / B part)
  int nsteps;
  // zero-block skipping: logical step s = (position q in the tile's list of non-empty blocks, step r inside the block) runs
  // physical step blk[q] * bsteps + r (blk: 4 bits each).  bsteps == 0: identity.  brecip = ceil(2^16 / bsteps): s / bsteps
  // for s < 8 * bsteps, bsteps <= 64
  unsigned blkmap, bsteps, brecip;
  unsigned step0;  // first logical step of this workgroup's share of K (K split inside the launch; 0 otherwise)
  __device__ __forceinline__ unsigned phys_step(unsigned s) const {
    s += step0;
    if (!bsteps) return s;
    const unsigned q = (s * brecip) >> 16;
    return ((blkmap >> (4u * q)) & 15u) * bsteps + (s - q * bsteps);
  }
  __device__ __forceinline__ SpLoop(half8 (&fa_)[2][2][2], half8 (&fb_)[2][TNW][2], floatx16 (&acc_)[2][TNW],
                                    unsigned (&aa)[G::NST][2], unsigned (&ba)[G::NST][2], unsigned (&va)[G::ND_A],
                                    unsigned (&vb)[G::ND_B])
      : fa(fa_), fb(fb_), acc(acc_), a_addr(aa), b_addr(ba), voff_a(va), voff_b(vb) {}

  template <int I, int SET, int ST>
  __device__ __forceinline__ void read_one() {
    if constexpr (I < 4) {
      constexpr int t = I >> 1, p = I & 1;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[SET][t][p]) : "v"(a_addr[ST][p]), "n"(t * 2048) : "memory");
    } else {
      constexpr int c = (I - 4) >> 1, p = (I - 4) & 1;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[SET][c][p]) : "v"(b_addr[ST][p]), "n"(c * 2048) : "memory");
    }
  }
  // MFMA order: the three piece products (l*h, h*l, h*h) each over all 2 x TNW tiles - consecutive MFMAs never share
  // an accumulator; smallest terms first
  template <int I, int SET>
  __device__ __forceinline__ void mfma_one() {
    constexpr int prod = I / (2 * TNW), j = I % (2 * TNW), t = j / TNW, c = j % TNW;
    constexpr int pa = prod == 0 ? 1 : 0, pb = prod == 1 ? 1 : 0;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t][c]) : "v"(fa[SET][t][pa]), "v"(fb[SET][c][pb]));
  }
  // steps past the end of K re-load the last step into a stage nobody reads any more: the count of outstanding DMAs
  // per step stays constant, which is what the counted vmcnt waits rely on
  template <int I, int ST>
  __device__ __forceinline__ void dma_one(int step) {
    // written out (not the builtin): with the builtin hipcc keeps one SGPR per (stage, instruction) LDS address -
    // 35 of them at BN = 320 - runs out of SGPRs, parks the buffer descriptors in VGPRs and wraps every DMA in a
    // readfirstlane waterfall loop.  Here m0 is base + immediate: two SGPRs in all.
    unsigned so = phys_step((unsigned)(step < nsteps ? step : nsteps - 1)) * 64u;
    if constexpr (I < G::ND_A)
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                   :: "s"(m0_a), "n"(ST * G::STG + I * 1024), "v"(voff_a[I]), "s"(rs_a), "s"(so) : "memory");
    else
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                   :: "s"(m0_b), "n"(ST * G::STG + SP_BM * 64 + (I - G::ND_A) * 1024), "v"(voff_b[I - G::ND_A]), "s"(rs_b), "s"(so)
                   : "memory");
  }
  template <int I, int N, int ST>
  __device__ __forceinline__ void dma_all(int step) {
    if constexpr (I < N) {
      dma_one<I, ST>(step);
      dma_all<I + 1, N, ST>(step);
    }
  }
  template <int I, int N, int SET, int ST>
  __device__ __forceinline__ void read_all() {
    if constexpr (I < N) {
      read_one<I, SET, ST>();
      read_all<I + 1, N, SET, ST>();
    }
  }

  // block scaling of A: fragments of scale block b are multiplied by inv[row][b] / max_b inv[row][b] <= 1
  const float* a_inv;
  int a_nblk, a_blk_steps;
  int64_t a_row[2];
  float a_rmax[2];
  half2v afac[2];
  template <int SET>
  __device__ __forceinline__ void scale_frags(int step) {  // plain VALU on the freshly read A fragments
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int lstep = step + (int)step0;  // logical step of the tile (a K split starts inside a block)
    if (lstep % a_blk_steps == 0 || step == 0) {
      const int q = step < nsteps ? lstep / a_blk_steps : 0;
      const int b = bsteps ? (int)((blkmap >> (4u * (unsigned)q)) & 15u) : q;
      if (step < nsteps && b < a_nblk) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const _Float16 f = (_Float16)(a_inv[a_row[t] * a_nblk + b] / a_rmax[t]);  // (a_inv_ld == a_nblk here)
          afac[t] = half2v{f, f};
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        half2v* v = reinterpret_cast<half2v*>(&fa[SET][t][p]);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * afac[t];
      }
  }

  // one k16 step S (mod UNR; stage S % NST, register set S & 1):  MFMA i (i < NR) + fragment read i of the NEXT step
  // (other register set);  MFMA NR + i (i < ND) + LDS-DMA i of step s + NST - 1;  the remaining MFMAs bare;
  // s_waitcnt vmcnt(VMW) lgkmcnt(0) (step s+2 has landed, the fragments of s+1 are in registers);  barrier.
  static constexpr int NR = 4 + 2 * TNW;  // fragment reads per step
  static constexpr int NM = 6 * TNW;      // MFMAs per step
  static_assert(NR + G::ND <= NM, "issue pattern");
  template <int S, int I, bool ABLK>
  __device__ __forceinline__ void step_items(int sbase) {
    if constexpr (I < NM) {
      mfma_one<I, (S & 1)>();
      if constexpr (I < NR) read_one<I, ((S + 1) & 1), ((S + 1) % G::NST)>();
      else if constexpr (I < NR + G::ND) dma_one<I - NR, ((S + G::NST - 1) % G::NST)>(sbase + S + G::NST - 1);
      if constexpr (ABLK && I == NM - 1) scale_frags<((S + 1) & 1)>(sbase + S + 1);  // after the last read was issued
      step_items<S, I + 1, ABLK>(sbase);
    }
  }
  template <int S, bool ABLK>
  __device__ __forceinline__ void step(int sbase) {
    step_items<S, 0, ABLK>(sbase);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::VMW) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  template <int S, bool ABLK>
  __device__ __forceinline__ void steps(int sbase) {  // one unrolled loop body: up to UNR steps (wave-uniform guard)
    if constexpr (S < G::UNR) {
      if (sbase + S < nsteps) step<S, ABLK>(sbase);
      steps<S + 1, ABLK>(sbase);
    }
  }
  template <int J>
  __device__ __forceinline__ void dma_prologue() {  // steps 0 .. NST-2 in flight
    if constexpr (J < G::NST - 1) {
      dma_all<0, G::ND, J>(J);
      dma_prologue<J + 1>();
    }
  }
};

template <int TNW, bool ABLK, bool GRAD, bool OUT_SP = false>
__global__ void __launch_bounds__(SP_NT, (TNW == 2 ? 2 : 1)) gemm_sp_nt_kernel(SpArgs g) {  // (BN = 128: two workgroups per CU, see SpGeo::NST)
  using G = SpGeo<TNW>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // K split: the producers (splits S-1 .. 1) take the low block ids - they are dispatched first -, the reducers (split 0) last
  unsigned bid = blockIdx.x;
  if (g.xcd_per) {
    bid = (blockIdx.x & 7u) * g.xcd_per + (blockIdx.x >> 3);
    if (bid >= g.xcd_total) return;  // (the whole workgroup: no barrier has been met)
  }
  int split = 0;
  int S = g.ksplit;
  if (S > 1) {
    const unsigned nt_all = gridDim.x / (unsigned)S;
    if (bid < nt_all * (unsigned)(S - 1)) {
      split = 1 + (int)(bid / nt_all);
      bid -= (unsigned)(split - 1) * nt_all;
    } else {
      bid -= nt_all * (unsigned)(S - 1);
    }
  }
  const unsigned tile_n = bid % g.n_tiles;
  const unsigned tile_m = bid / g.n_tiles;
  int64_t row0 = (int64_t)tile_m * SP_BM, m_end = g.M;  // this tile's rows: [row0, min(row0 + 128, m_end))
  int64_t grp = 0;
  if (g.tile_table) {
    const int32_t* e = g.tile_table + 4 * (int64_t)tile_m;
    row0 = __builtin_amdgcn_readfirstlane(e[0]);
    m_end = row0 + __builtin_amdgcn_readfirstlane(e[1]);
    grp = __builtin_amdgcn_readfirstlane(e[2]);
  }
  const int64_t col0 = (int64_t)tile_n * G::BN;
  int nsteps = (int)(g.K >> 4);
  unsigned blkmap = 0, bsteps = 0;
  if (g.tile_kmask) {  // only the blocks of K that hold anything in this row tile (wave-uniform: one byte per tile)
    unsigned m = __builtin_amdgcn_readfirstlane((int)g.tile_kmask[tile_m]) & ((1u << g.a_nblk) - 1u);
    if (!m) m = 1u;  // an all-empty tile still runs one (zero) block: the epilogue needs defined accumulators and scales
    int n = 0;
    for (int b = 0; b < g.a_nblk; ++b)
      if (m & (1u << b)) blkmap |= (unsigned)b << (4 * n++);
    bsteps = (unsigned)g.a_blk_steps;
    nsteps = n * g.a_blk_steps;
  }
  unsigned step0 = 0;
  if (S > 1) {
    // This workgroup's share of K: the PHYSICAL k16 steps [p0, p1) - the same range with and without a tile mask, so that a
    // product that skips all-zero blocks groups its non-zero terms exactly like the one that multiplies them (bit-equal
    // results, tests/test_gpu_gemm_sp.py) - of which the tile's non-empty blocks hold a contiguous run of logical steps.
    const int P = (int)(g.K >> 4);
    const int per = (P + S - 1) / S;
    const int p0 = split * per, p1 = p0 + per < P ? p0 + per : P;
    if (!bsteps) {
      step0 = (unsigned)p0;
      nsteps = p1 > p0 ? p1 - p0 : 0;
    } else {
      int first = -1, cnt = 0;
      const int nb = nsteps / (int)bsteps;
      for (int q = 0; q < nb; ++q) {
        const int b = (int)((blkmap >> (4u * (unsigned)q)) & 15u);
        const int lo = p0 > b * (int)bsteps ? p0 : b * (int)bsteps;
        const int hi = p1 < (b + 1) * (int)bsteps ? p1 : (b + 1) * (int)bsteps;
        if (hi > lo) {
          if (first < 0) first = q * (int)bsteps + (lo - b * (int)bsteps);
          cnt += hi - lo;
        }
      }
      step0 = first < 0 ? 0u : (unsigned)first;
      nsteps = cnt;
    }
  }

  // ---- DMA setup: buffer descriptors over this tile's rows (rows past M / N read as zeros) -------------------
  const int64_t rows_a = m_end - row0 < SP_BM ? m_end - row0 : SP_BM;
  const int64_t rows_b = g.N - col0 < G::BN ? g.N - col0 : G::BN;
  // raw buffer descriptors: {base[31:0], base[47:32] (stride 0), bytes, 0x00020000}
  auto make_rsrc = [](const uint8_t* p, int64_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return uint4v{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
                  (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  // (a_rows: the descriptor spans the whole operand - its rows come from anywhere; host: M * lda < 2^32)
  const uint4v rs_a = g.a_rows ? make_rsrc(g.A, g.a_src_rows * g.lda) : make_rsrc(g.A + row0 * g.lda, rows_a * g.lda);
  const uint4v rs_b = make_rsrc(g.B + grp * g.group_stride_b + col0 * g.ldb, rows_b * g.ldb);
  half8 r_fa[2][2][2];
  half8 r_fb[2][TNW][2];
  floatx16 r_acc[2][TNW];
  unsigned r_aa[G::NST][2], r_ba[G::NST][2], r_va[G::ND_A], r_vb[G::ND_B];
  SpLoop<TNW> L(r_fa, r_fb, r_acc, r_aa, r_ba, r_va, r_vb);
  L.rs_a = rs_a;
  L.rs_b = rs_b;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void*)lds;
  L.m0_a = __builtin_amdgcn_readfirstlane(lds_base + wave * G::ND_A * 1024);
  L.m0_b = __builtin_amdgcn_readfirstlane(lds_base + wave * G::ND_B * 1024);
  L.nsteps = nsteps;
  L.blkmap = blkmap;
  L.step0 = step0;
  L.bsteps = bsteps;
  L.brecip = bsteps ? (65536u + bsteps - 1u) / bsteps : 0u;
  // lane j of a DMA instruction fills LDS slot j of 16 rows x 64 bytes: row j / 4, slot q' = j % 4 holds source
  // chunk q = q' ^ ((row >> 2) & 3)
  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((drow >> 2) & 3);
#pragma unroll
  for (int i = 0; i < G::ND_A; ++i) {
    const int tr = (wave * G::ND_A + i) * 16 + drow;  // row of the tile this lane fetches
    if (g.a_rows) {  // rows past M: an offset outside the descriptor reads zeros
      const int64_t pr = row0 + tr;
      L.voff_a[i] = pr < m_end ? (unsigned)((int64_t)g.a_rows[pr] * g.lda + dq * 16) : 0xfffffff0u;
    } else {
      L.voff_a[i] = (unsigned)(tr * g.lda + dq * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < G::ND_B; ++i) L.voff_b[i] = (unsigned)(((wave * G::ND_B + i) * 16 + drow) * g.ldb + dq * 16);

  // ---- fragment addresses (one base register per stage and operand plane; tiles are immediate offsets) ---------
  const int fi = lane & 31, kg = lane >> 5;
  const int sw = (fi >> 2) & 3;
#pragma unroll
  for (int st = 0; st < G::NST; ++st)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      L.a_addr[st][p] = (unsigned)(st * G::STG + (wm * 64 + fi) * 64 + (((p * 2 + kg) ^ sw) * 16));
      L.b_addr[st][p] = (unsigned)(st * G::STG + SP_BM * 64 + (wn * 32 * TNW + fi) * 64 + (((p * 2 + kg) ^ sw) * 16));
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < TNW; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) L.acc[t][c][r] = 0.f;

  L.a_inv = g.a_inv; L.a_nblk = g.a_nblk; L.a_blk_steps = g.a_blk_steps;
  L.afac[0] = L.afac[1] = half2v{(_Float16)1.f, (_Float16)1.f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int64_t r = row0 + wm * 64 + t * 32 + fi;
    L.a_row[t] = r < m_end ? r : m_end - 1;
    if (g.a_rows) L.a_row[t] = g.a_rows[L.a_row[t]];  // the scales of the operand row this product row reads
    float m = 1.f;
    if (ABLK) {
      m = 0.f;
      for (int b = 0; b < g.a_nblk; ++b) m = fmaxf(m, g.a_inv[L.a_row[t] * g.a_inv_ld + b]);
    } else if (g.a_inv) {
      m = g.a_inv[L.a_row[t] * g.a_inv_ld];
    }
    L.a_rmax[t] = m;  // the row's epilogue factor (lane fi holds row fi of row tile t)
  }
  float cfac[TNW], cbias[TNW];  // column factors / bias of this lane's accumulator columns
#pragma unroll
  for (int c = 0; c < TNW; ++c) {
    const int64_t col = col0 + wn * 32 * TNW + c * 32 + fi;
    cfac[c] = g.b_inv ? g.b_inv[grp * g.group_stride_scale + col] : 1.f;
    cbias[c] = g.bias ? g.bias[grp * g.group_stride_scale + col] : 0.f;
  }

  // ---- prologue: three steps in flight, fragments of step 0 in set 0 ----------------------------------------
  constexpr int NR = SpLoop<TNW>::NR;
  if (nsteps > 0) {  // (a late K split of a short tile has nothing to multiply: its accumulators stay zero)
    L.template dma_prologue<0>();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::VMW) : "memory");  // steps 0 and 1 have landed
    __builtin_amdgcn_s_barrier();
    L.template read_all<0, NR, 0, 0>();
    if (ABLK) L.template scale_frags<0>(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    for (int s = 0; s < nsteps; s += G::UNR) L.template steps<0, ABLK>(s);
  }
  // the re-loads of the last steps must not land in the patch; the accumulators of the last MFMAs must be
  // readable by plain VALU (the compiler does not see the MFMAs inside the asm statements)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- K split inside the launch: hand the raw accumulators to split 0 ------------------------------------------
  // (cdna_hip_programming.md guideline 16: write-through 16-byte stores, every storing wave drains, one barrier, ONE lane
  // publishes the flag with an agent-scope store; the reducer polls that one word relaxed, then reads the slab with sc1
  // loads.)  A slab is stored in the accumulator layout, lane-major - float4 q of accumulator (t, c) of lane l at
  // [wave][t][c][q][l] - so every store / load instruction moves 1 KB of consecutive bytes and split 0 adds register to
  // register.  Sum order: own share, then splits 1, 2, ..: fixed, so the result is bit-reproducible.
  if (S > 1) {
    constexpr int WAVE_F4 = 2 * TNW * 4 * 64;  // float4 per wave and slab
    const unsigned tile_lin = bid;
    auto make_rsrc_ws = [](const float* ptr, int bytes) {  // raw buffer over one wave's share of a slab
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ptr), (short)0, bytes, 0x00020000);
    };
    unsigned* flags = g.ws_flags + (size_t)tile_lin * (unsigned)(S - 1);
    if (split != 0) {
      float* slab = g.ws_partial + ((size_t)(tile_lin * (unsigned)(S - 1) + (unsigned)(split - 1)) * 4 + wave) * (WAVE_F4 * 4);
      const __amdgpu_buffer_rsrc_t rs = make_rsrc_ws(slab, WAVE_F4 * 16);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < TNW; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4v v{__float_as_uint(L.acc[t][c][4 * q]), __float_as_uint(L.acc[t][c][4 * q + 1]),
                           __float_as_uint(L.acc[t][c][4 * q + 2]), __float_as_uint(L.acc[t][c][4 * q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((((t * TNW + c) * 4 + q) * 64 + lane) * 16), 0, 16 /* sc1 */);
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + (split - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid < S - 1) {  // lane i of wave 0 polls the flag of split i + 1: one word each, relaxed; bounded (a producer that
      unsigned spins = 0;  // never arrives must not hang the device)
      while (__hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 24)) {
          if (g.ws_timeout) *g.ws_timeout = 1;
          break;
        }
      }
    }
    __syncthreads();
    for (int sp = 1; sp < S; ++sp) {
      const float* slab = g.ws_partial + ((size_t)(tile_lin * (unsigned)(S - 1) + (unsigned)(sp - 1)) * 4 + wave) * (WAVE_F4 * 4);
      const __amdgpu_buffer_rsrc_t rs = make_rsrc_ws(slab, WAVE_F4 * 16);
      // the whole slab in flight at once (2 TNW x 4 loads of 16 bytes per lane: the fragment registers are dead here) - a
      // first version waited for four loads at a time and spent ~25 us per launch on 30 dependent round trips
      uint4v v[2 * TNW * 4];
#pragma unroll
      for (int i = 0; i < 2 * TNW * 4; ++i)
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((i * 64 + lane) * 16), 0, 16 /* sc1 */);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < TNW; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint4v w = v[(t * TNW + c) * 4 + q];
            L.acc[t][c][4 * q] += __uint_as_float(w.x);
            L.acc[t][c][4 * q + 1] += __uint_as_float(w.y);
            L.acc[t][c][4 * q + 2] += __uint_as_float(w.z);
            L.acc[t][c][4 * q + 3] += __uint_as_float(w.w);
          }
    }
    // the flags this tile consumed go back to zero for the next launch (also the next replay of a captured step)
    __syncthreads();
    if (tid < S - 1) __hip_atomic_store(flags + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------
  // Scales, bias and activation are applied in the accumulator layout (their operands were fetched before the main
  // loop: column factors / bias per (lane, column tile), row factors by a lane exchange of a_rmax), then a 32-row
  // block goes through the wave's LDS patch so that every lane stores 16 contiguous bytes; the gradient factors and the
  // accumulate operand are fetched four float4 ahead of their use.
  float* patch = reinterpret_cast<float*>(lds) + wave * 32 * G::PATCH_LD;
  const int64_t wcol0 = col0 + wn * 32 * TNW;
  const DropoutKey dkey = g.drop_on == 1 ? dropout_resolve(g.drop) : g.drop;  // the mask of the current epoch (common.hpp)
  constexpr int C4 = 8 * TNW;             // float4 per patch row
  constexpr int NIT = 32 * C4 / 64;       // float4 per lane and row tile
  static_assert(NIT % 4 == 0, "epilogue chunking");
  auto epilogue_tile = [&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    float rf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rf[r] = __shfl(L.a_rmax[t], (r & 3) + 8 * (r >> 2) + 4 * kg, 64);
#pragma unroll
    for (int c = 0; c < TNW; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) L.acc[t][c][r] = L.acc[t][c][r] * (rf[r] * cfac[c]) + cbias[c];
    // one wave-uniform dispatch on the activation for the whole tile (a switch per element costs a taken branch each:
    // 320 of them were 40 us of a 120 us kernel)
    auto act_tile = [&](auto act_c) {
#pragma unroll
      for (int c = 0; c < TNW; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) L.acc[t][c][r] = act_apply(decltype(act_c)::value, L.acc[t][c][r]);
    };
    switch (g.act) {
      case TFGNN_ACT_RELU: act_tile(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
      case TFGNN_ACT_TANH: act_tile(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
      case TFGNN_ACT_LEAKY_RELU: act_tile(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
      case TFGNN_ACT_ELU: act_tile(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
      case TFGNN_ACT_SELU: act_tile(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
      case TFGNN_ACT_GELU: act_tile(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
      case TFGNN_ACT_SIGMOID: act_tile(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
      default: break;
    }
#pragma unroll
    for (int c = 0; c < TNW; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = (r & 3) + 8 * (r >> 2) + 4 * kg;
        patch[prow * G::PATCH_LD + c * 32 + fi] = L.acc[t][c][r];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int64_t wrow0 = row0 + wm * 64 + t * 32;
    const float* __restrict__ mulp = g.mul;
    const float* __restrict__ savp = g.saved;
    float* __restrict__ cptr = g.C;
    if constexpr (OUT_SP) {
      // The result as a split operand of the next product (one scale per row): 8 lanes own a row of this wave's
      // 32 TNW columns (TNW float4 each), four row groups per 32-row block; the row maximum of the FINAL values (after
      // the gradient factors) is reduced over the 8 lanes, exchanged with the wave that holds the other half of the
      // row through LDS, then every lane splits and stores its chunks (and the fp32 form when C is given).
      float* xmax = reinterpret_cast<float*>(lds) + 4 * 32 * G::PATCH_LD + t * 4 * 32;  // [tile][wave][32 rows]
      const int rgrp = lane >> 3, l8 = lane & 7;
      float4 w[4][TNW];
      float rmax[4];
      bool okq[4];
      int64_t rowq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pr = q * 8 + rgrp;
        const int64_t row = wrow0 + pr;
        okq[q] = row < m_end;
        rowq[q] = okq[q] ? row : m_end - 1;
        if (g.row_map) rowq[q] = g.row_map[rowq[q]];
        float4 m[TNW], sv[TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          const int c4 = l8 + 8 * j;
          const int64_t col = wcol0 + c4 * 4;
          w[q][j] = *reinterpret_cast<const float4*>(patch + pr * G::PATCH_LD + c4 * 4);
          if (GRAD) {
            m[j] = mulp ? *reinterpret_cast<const float4*>(mulp + rowq[q] * g.ld_mul + col) : float4{1.f, 1.f, 1.f, 1.f};
            sv[j] = savp ? *reinterpret_cast<const float4*>(savp + rowq[q] * g.ld_saved + col) : float4{0.f, 0.f, 0.f, 0.f};
          }
        }
        if (GRAD && savp) {
          if (g.saved_scale != 1.f) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) { sv[j].x *= g.saved_scale; sv[j].y *= g.saved_scale; sv[j].z *= g.saved_scale; sv[j].w *= g.saved_scale; }
          }
          auto dact_row = [&](auto act_c) {
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
              sv[j].x = act_grad(decltype(act_c)::value, sv[j].x); sv[j].y = act_grad(decltype(act_c)::value, sv[j].y);
              sv[j].z = act_grad(decltype(act_c)::value, sv[j].z); sv[j].w = act_grad(decltype(act_c)::value, sv[j].w);
            }
          };
          switch (g.dact) {
            case TFGNN_ACT_RELU: dact_row(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
            case TFGNN_ACT_TANH: dact_row(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
            case TFGNN_ACT_LEAKY_RELU: dact_row(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
            case TFGNN_ACT_ELU: dact_row(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
            case TFGNN_ACT_SELU: dact_row(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
            case TFGNN_ACT_GELU: dact_row(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
            case TFGNN_ACT_SIGMOID: dact_row(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
            default: dact_row(std::integral_constant<int, TFGNN_ACT_NONE>{}); break;
          }
        }
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          if (GRAD) {
            w[q][j].x *= m[j].x; w[q][j].y *= m[j].y; w[q][j].z *= m[j].z; w[q][j].w *= m[j].w;
            if (savp) { w[q][j].x *= sv[j].x; w[q][j].y *= sv[j].y; w[q][j].z *= sv[j].z; w[q][j].w *= sv[j].w; }
          }
          if (g.drop_on == 1) {
            const float4 dm = dropout_mask4(dkey, (uint64_t)(rowq[q] * g.drop_ld + wcol0 + (l8 + 8 * j) * 4));  // N % 4 == 0
            w[q][j].x *= dm.x; w[q][j].y *= dm.y; w[q][j].z *= dm.z; w[q][j].w *= dm.w;
          } else if (g.drop_on == 2) {  // the mask is in the saved tensor (see SpArgs)
            w[q][j].x *= g.drop.scale; w[q][j].y *= g.drop.scale; w[q][j].z *= g.drop.scale; w[q][j].w *= g.drop.scale;
          }
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(w[q][j].x), fabsf(w[q][j].y)), fmaxf(fabsf(w[q][j].z), fabsf(w[q][j].w))));
          if (w[q][j].x != w[q][j].x || w[q][j].y != w[q][j].y || w[q][j].z != w[q][j].z || w[q][j].w != w[q][j].w)
            mx = __builtin_inff();
        }
#pragma unroll
        for (int o = 4; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        rmax[q] = mx;
        if (l8 == 0) xmax[wave * 32 + pr] = mx;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // all four waves run the two row tiles in step
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pr = q * 8 + rgrp;
        const float full = fmaxf(rmax[q], xmax[(wave ^ 1) * 32 + pr]);  // the wave with the other column half: same wm
        float iv;
        const float sc = sp_scale_for_max(full, &iv);
        if (okq[q]) {
          if (wn == 0 && l8 == 0) g.out_inv[rowq[q] * g.n_tiles + tile_n] = iv;  // one scale per row and column tile
          uint8_t* drow = g.out_sp + rowq[q] * g.ld_out_sp;
#pragma unroll
          for (int j = 0; j < TNW; ++j) {
            const int64_t col = wcol0 + (l8 + 8 * j) * 4;
            if (cptr) *reinterpret_cast<float4*>(cptr + rowq[q] * g.ldc + col) = w[q][j];
            sp_store4(drow, col, w[q][j], sc);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      return;
    }
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += 4) {
      float4 v[4], m[4], sv[4], o[4];
      bool ok[4];
      int64_t coff[4], doff[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = lane + (it0 + j) * 64;
        const int pr = idx / C4, c4 = idx - pr * C4;
        const int64_t row = wrow0 + pr, col = wcol0 + c4 * 4;
        ok[j] = row < m_end;
        int64_t rr = ok[j] ? row : m_end - 1;
        if (g.row_map) rr = g.row_map[rr];
        coff[j] = rr * g.ldc + col;
        doff[j] = rr * g.drop_ld + col;
        v[j] = *reinterpret_cast<const float4*>(patch + pr * G::PATCH_LD + c4 * 4);
        if (GRAD) {
          m[j] = mulp ? *reinterpret_cast<const float4*>(mulp + rr * g.ld_mul + col) : float4{1.f, 1.f, 1.f, 1.f};
          if (savp) sv[j] = *reinterpret_cast<const float4*>(savp + rr * g.ld_saved + col);
        }
        if (g.accumulate) o[j] = *reinterpret_cast<const float4*>(cptr + coff[j]);
      }
      if (GRAD && savp) {  // sv <- act'(saved), one dispatch per four float4
        if (g.saved_scale != 1.f) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { sv[j].x *= g.saved_scale; sv[j].y *= g.saved_scale; sv[j].z *= g.saved_scale; sv[j].w *= g.saved_scale; }
        }
        auto dact_chunk = [&](auto act_c) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sv[j].x = act_grad(decltype(act_c)::value, sv[j].x); sv[j].y = act_grad(decltype(act_c)::value, sv[j].y);
            sv[j].z = act_grad(decltype(act_c)::value, sv[j].z); sv[j].w = act_grad(decltype(act_c)::value, sv[j].w);
          }
        };
        switch (g.dact) {
          case TFGNN_ACT_RELU: dact_chunk(std::integral_constant<int, TFGNN_ACT_RELU>{}); break;
          case TFGNN_ACT_TANH: dact_chunk(std::integral_constant<int, TFGNN_ACT_TANH>{}); break;
          case TFGNN_ACT_LEAKY_RELU: dact_chunk(std::integral_constant<int, TFGNN_ACT_LEAKY_RELU>{}); break;
          case TFGNN_ACT_ELU: dact_chunk(std::integral_constant<int, TFGNN_ACT_ELU>{}); break;
          case TFGNN_ACT_SELU: dact_chunk(std::integral_constant<int, TFGNN_ACT_SELU>{}); break;
          case TFGNN_ACT_GELU: dact_chunk(std::integral_constant<int, TFGNN_ACT_GELU>{}); break;
          case TFGNN_ACT_SIGMOID: dact_chunk(std::integral_constant<int, TFGNN_ACT_SIGMOID>{}); break;
          default: dact_chunk(std::integral_constant<int, TFGNN_ACT_NONE>{}); break;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 w = v[j];
        if (GRAD) {
          w.x *= m[j].x; w.y *= m[j].y; w.z *= m[j].z; w.w *= m[j].w;
          if (savp) { w.x *= sv[j].x; w.y *= sv[j].y; w.z *= sv[j].z; w.w *= sv[j].w; }
        }
        if (g.drop_on == 1) {
          const float4 dm = dropout_mask4(dkey, (uint64_t)doff[j]);
          w.x *= dm.x; w.y *= dm.y; w.z *= dm.z; w.w *= dm.w;
        } else if (g.drop_on == 2) {
          w.x *= g.drop.scale; w.y *= g.drop.scale; w.z *= g.drop.scale; w.w *= g.drop.scale;
        }
        if (g.accumulate) { w.x += o[j].x; w.y += o[j].y; w.z += o[j].z; w.w += o[j].w; }
        if (ok[j]) *reinterpret_cast<float4*>(cptr + coff[j]) = w;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  epilogue_tile(std::integral_constant<int, 0>{});
  epilogue_tile(std::integral_constant<int, 1>{});
}


// ------------------------------------------------------------------------------------------------------
// TN kernel: C[m, n] = sum_k A[k, m] B[k, n] - both operands stored with K as the ROW index (weight gradients
// dW = X^T G: K runs over the nodes).  Same tile, ring, pinned main loop and MFMA order as the NT kernel; what differs:
//   * a k16 step of the A tile is 16 operand rows x 128 columns (8 granules of 64 bytes per row), of the B tile
//     16 rows x BN columns.  LDS image per group of 4 rows and per pair of granules (32 columns): 8 containers of 64
//     bytes in the order (row & 3, granule & 1); containers of rows 2, 3 are stored l-plane first.  One DMA instruction
//     = 2 granule pairs x 4 rows: every row contributes 256 contiguous source bytes.
//   * an MFMA operand (8 consecutive k of one column per lane) is two ds_read_b64_tr_b16: a 16-lane group reads a
//     [4 rows] x [16 columns] block - lane j supplies the address of row j / 4, columns (j % 4) * 4 .. + 3 - and gets
//     column j.  With the image above the eight 32-byte pieces a half-wave touches lie in eight different bank
//     groups (conflict-free).  These reads go through the builtin (the two halves must land in adjacent registers,
//     which an asm operand cannot express); sched_barrier pins them behind their MFMA.
//   * scales: the operands carry per-ROW scales, i.e. per-k factors here: sp_tn_factors_kernel turns them into one fp16
//     factor <= 1 per (k, column block of A) that travels with the ring (one more 1 KB DMA slot per
//     stage) and are multiplied into the A fragments (16 v_pk_mul_f16 per step); the reduce pass multiplies the
//     block's reference scale back.
//   * split-K over blockIdx.y; partial tiles go to the workspace, sp_tn_reduce_kernel sums them in split order,
//     applies the two scales and writes C through (group, row, column) strides - dW comes out in the kernels' [L, D, H]
//     layout without a transpose pass.
// ------------------------------------------------------------------------------------------------------
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) short4v lds_short4;
__device__ __forceinline__ half4 sp_tr_read(unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(uintptr_t)lds_addr));
#else
  (void)lds_addr;
  return half4{};
#endif
}

struct SpTnArgs {
  int64_t M, N, K;
  const uint8_t* A;  // SP16, rows = k; already advanced to this product's first column
  int64_t lda;
  const uint8_t* B;
  int64_t ldb;
  const _Float16* F;  // [a_nblk][f_ld] per-k factors of the A blocks (sp_tn_factors_kernel), zero past K
  int64_t f_ld;
  int a_sb;           // columns per scale block of A
  int64_t a_col0;     // first column of A (for the block index of a column)
  int a_nblk;         // scale blocks per row of A
  // FIK kernels (factors in the kernel, round 4): the scales themselves; every workgroup normalises its own K range
  const float* inv_a;  // [K][a_nblk]
  const float* inv_b;  // [K] or NULL
  float* ref_split;    // [splits][a_nblk]: the reference scale of (split, block), multiplied back by the reduce pass
  int* spread_flag;
  float* partial;     // [splits][M][N]
  int64_t k_chunk;    // rows of K per split (a multiple of 16)
  unsigned n_tiles;
  unsigned tiles, splits, per_xcd;  // output tiles, K splits, workgroups per XCD (grid = 8 * per_xcd)
  // grouped K ranges (round 5): split z multiplies rows [split_table[2 z], + split_table[2 z + 1]) - the ranges of several
  // products over row groups of the same operands in ONE launch; the reduce pass sums each group's ranges
  const int32_t* split_table;
};

// Per-k factors of the weight-gradient product.  Both operands carry one power-of-two scale per ROW (A one per row and
// column block), i.e. per k:  C[m, n] = sum_k inv_a[k, blk(m)] inv_b[k] A^[k, m] B^[k, n].  With ref[b] = max_k of the
// product of the two scales, F[b][k] = inv_a[k, b] inv_b[k] / ref[b] <= 1 is a power of two that fp16 holds exactly down
// to 2^-24 (0 below); the kernel multiplies it into the A fragments (16 v_pk_mul_f16 per step, in the shadow of the
// MFMAs), the reduce pass multiplies ref[blk(m)] back.  A row whose scale product is 2^-j of the largest keeps all 22
// bits of its elements while j <= 13, 35 - j bits after that (absolute error 2^-25 of the largest row's elements) and
// drops out at j > 24 - far beyond the spread of the node states and gradients of a batch (tests: rows over 2^+-9
// keep the fp32 error class; over 2^+-18 the error grows to 1e-5 of sum |a||b|).  Scaling BOTH operands' fragments by
// their own factors would double that range but costs 40 more VALU instructions per step (measured 115 vs 92 us).
// SP_TN_FCHUNKS workgroups per block: each one takes the maximum over ALL k itself (the scales are 4 bytes per row, L2
// resident; loads issued eight at a time) and writes its own slice of the factors - no second launch, no atomics.
constexpr int SP_TN_FCHUNKS = 16;  // K up to 128k rows: every workgroup takes the maximum over all k itself (one launch)
// Longer operands (QM9-sized batches: 1.15M rows - 16 workgroups walking all of them took 225 us, 128 of them 359 us): the
// maxima of SP_TN_MAXCHUNKS slices first (sp_tn_slice_max_kernel), then the factor pass reads those instead of all k
constexpr int SP_TN_MAXCHUNKS = 128;
static int sp_tn_fchunks(int64_t K) { return K > 131072 ? SP_TN_MAXCHUNKS : SP_TN_FCHUNKS; }

__global__ void __launch_bounds__(1024) sp_tn_slice_max_kernel(const float* __restrict__ inv_a, int64_t ld_a, const float* __restrict__ inv_b,
                                                               int64_t ld_b, int64_t K, float* __restrict__ slice_max, int nchunks) {
  __shared__ float red[16];
  const int b = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  const int64_t per = (K + nchunks - 1) / nchunks;
  const int64_t k1 = (chunk + 1) * per < K ? (chunk + 1) * per : K;
  float mx = 0.f;
  for (int64_t k = chunk * per + threadIdx.x; k < k1; k += 1024) mx = fmaxf(mx, inv_a[k * ld_a + b] * (inv_b ? inv_b[k * ld_b] : 1.f));
#pragma unroll
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i]);
    slice_max[blockIdx.x] = m;
  }
}
__global__ void __launch_bounds__(1024) sp_tn_factors_kernel(const float* __restrict__ inv_a, int64_t ld_a, const float* __restrict__ inv_b,
                                                             int64_t ld_b, int64_t K, _Float16* __restrict__ F, int64_t f_ld,
                                                             float* __restrict__ ref, int* __restrict__ spread_flag, int nchunks,
                                                             const float* __restrict__ slice_max) {
  __shared__ float red[16];
  const int b = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
  float mx = 0.f;
  if (slice_max) {  // the slices' maxima were taken by sp_tn_slice_max_kernel
    for (int i = threadIdx.x; i < nchunks; i += 1024) mx = fmaxf(mx, slice_max[b * nchunks + i]);
  } else
  for (int64_t k0 = threadIdx.x; k0 < K; k0 += 8 * 1024) {  // (32 loads in flight per array measured 35 us instead of 13)
    float va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t k = k0 + u * 1024;
      va[u] = k < K ? inv_a[k * ld_a + b] : 0.f;
      vb[u] = (k < K && inv_b) ? inv_b[k * ld_b] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) mx = fmaxf(mx, va[u] * vb[u]);
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
  if (mx == 0.f) mx = 1.f;
  if (threadIdx.x == 0 && chunk == 0) ref[b] = mx;
  const float r = 1.f / mx;  // powers of two: exact
  const int64_t per = ((f_ld + nchunks - 1) / nchunks + 7) & ~7ll;
  const int64_t kend = (chunk + 1) * per < f_ld ? (chunk + 1) * per : f_ld;
  bool wide = false;
  for (int64_t k = chunk * per + threadIdx.x; k < kend; k += 1024) {
    const float ia = k < K ? inv_a[k * ld_a + b] : 0.f, ib = (k < K && inv_b) ? inv_b[k * ld_b] : 1.f;
    const float f = ia * ib * r;
    F[(int64_t)b * f_ld + k] = (_Float16)f;
    // a NON-ZERO row more than 2^20 below the block's largest scale product (all-zero rows carry the smallest normal
    // scale, 2^-126: their factor is 0 and they contribute nothing).  Below 2^-13 a row's elements start to keep fewer
    // than 22 bits relative to THEMSELVES (35 - j bits at 2^-j) while the absolute error stays <= 2^-39 of the largest
    // rows' elements - far below the fp32 rounding of the sum, which is why the benchmark's own gradient rows (hub-normalised,
    // 2^16 apart) give dW errors of 6e-7 of the largest entry; past 2^-20 fewer than 15 bits are left and at 2^-24 the
    // row drops out: reported, see tfgnn_sp_spread_flag
    wide |= sp_row_too_small(f, ia, ib);
  }
  if (spread_flag && __any(wide) && (threadIdx.x & 63) == 0) *spread_flag = 1;
}

// FIK = the per-k factors are computed by the workgroup itself (round 4).  The separate factor pass (sp_tn_factors_kernel:
// 13 us per product, six products per step) existed because the factors were normalised by the GLOBAL maximum of the scale
// products; a workgroup only ever multiplies its own K range, so it can normalise by the maximum over THAT range - the reduce
// pass multiplies every split's partial by the split's own reference.  The factors of the whole range are computed into an
// LDS table ([step][4 blocks][16 k] fp16, 128 bytes per step, up to SP_TN_FTAB_BYTES: k_chunk <= 2560) while the first ring
// stages are in flight; the ring loses the 1 KB factor slot per stage and wave 0 its extra DMA per step.
constexpr int SP_TN_FTAB_BYTES = 20480;
constexpr int64_t SP_TN_FIK_MAX_CHUNK = (SP_TN_FTAB_BYTES / 128 - 2) * 16;  // rows of K per split: one zeroed step + 128 B of scratch
// BSC (round 5, "wide range"): BOTH operands' fragments are scaled by per-k factors - instead of ONE combined factor on the A
// fragments.  A row keeps >= 16 bits while each of its two factors is >= 2^-22: the scale products of the per-relation weight
// gradients of an un-normalised RGIN stack spread over 2^25 (2^19 and 2^21 per operand) and trip the combined-factor guard.
// Round 5 gave each operand its own deficit (F_a[b][k] = inv_a[k, b] / max_k inv_a[., b], F_b[k] = inv_b[k] / max_k inv_b);
// since round 6 the deficit of the PAIR is split evenly between the two (see the table's computation in the kernel): 2^44 of
// spread in the scale products of a K range, wherever it comes from.
// 2 TNW more packed multiplies per fragment set and step; the factor table takes 160 bytes per step (k_chunk <= 2016).
constexpr int64_t SP_TN_BSC_MAX_CHUNK = (SP_TN_FTAB_BYTES / 160 - 2) * 16;
template <int TNW, bool FIK, bool BSC = false>
struct SpGeoTN : SpGeo<TNW> {
  using B0 = SpGeo<TNW>;
  static_assert(!BSC || FIK, "the two-factor form computes its factors in the kernel");
  static constexpr int FSTRIDE = BSC ? 160 : 128;   // bytes of factors per k16 step: [4 blocks][16 k] (+ [16 k] of B)
  static constexpr int FOFF = B0::STG;          // !FIK: per stage the factors of the step, [<= 4 blocks][16 k] fp16 (1 KB slot)
  static constexpr int STG = B0::STG + (FIK ? 0 : 1024);
  static constexpr int ND = B0::ND + (FIK ? 0 : 1);  // !FIK: wave 0 also fetches the factor slot; the other waves issue ND - 1 DMAs
  static constexpr int RING_MAX = 163840 - (FIK ? SP_TN_FTAB_BYTES : 0);
  static constexpr int NST = (RING_MAX / STG) < 6 ? (RING_MAX / STG) : 6;
  static constexpr int UNR = NST % 2 == 0 ? NST : 2 * NST;
  static constexpr int VMW = (NST - 3) * ND;        // DMAs that may still be in flight at the end of a step: wave 0 ...
  static constexpr int VMW1 = (NST - 3) * (FIK ? ND : ND - 1);  // ... and waves 1 - 3
  static constexpr int FTAB = NST * STG;            // FIK: LDS offset of the factor table
  static constexpr int LDS_BYTES = NST * STG + (FIK ? SP_TN_FTAB_BYTES : 0);
  static_assert(NST >= 4 && VMW < 64, "ring depth");
  static_assert(4 * 32 * B0::PATCH_LD * 4 <= NST * STG, "epilogue patch must fit the ring");
};

template <int TNW, bool FIK, bool BSC = false>
struct SpLoopTN {
  using G = SpGeoTN<TNW, FIK, BSC>;
  static constexpr int KGB_A = 2048, KGB_B = TNW * 1024;  // bytes per 4-row group: A (4 granule pairs), B (2 TNW pairs)
  half8 (&fa)[2][2][2];
  half8 (&fb)[2][TNW][2];
  floatx16 (&acc)[2][TNW];
  unsigned (&a_addr)[G::NST][2], (&b_addr)[G::NST][2];  // this lane's tr-read address per stage / plane (kb 0, tile 0)
  unsigned (&voff_a)[G::ND_A], (&voff_b)[G::ND_B];
  uint4v rs_a, rs_b, rs_f;
  unsigned m0_a, m0_b, m0_f, voff_f;
  unsigned step_bytes_a, step_bytes_b;  // 16 rows of the operand
  unsigned f_addr[2];                   // LDS address (stage 0) of this lane's 8 factors for row tile t
  int nsteps;
  bool wave0;                           // wave-uniform
  __device__ __forceinline__ SpLoopTN(half8 (&fa_)[2][2][2], half8 (&fb_)[2][TNW][2], floatx16 (&acc_)[2][TNW],
                                      unsigned (&aa)[G::NST][2], unsigned (&ba)[G::NST][2], unsigned (&va)[G::ND_A],
                                      unsigned (&vb)[G::ND_B])
      : fa(fa_), fb(fb_), acc(acc_), a_addr(aa), b_addr(ba), voff_a(va), voff_b(vb) {}

  template <int I, int SET, int ST>
  __device__ __forceinline__ void read_one() {
    if constexpr (I < 4) {
      constexpr int t = I >> 1, p = I & 1;
      const half4 lo = sp_tr_read(a_addr[ST][p] + t * 512);
      const half4 hi = sp_tr_read(a_addr[ST][p] + t * 512 + KGB_A);
      fa[SET][t][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      constexpr int c = (I - 4) >> 1, p = (I - 4) & 1;
      const half4 lo = sp_tr_read(b_addr[ST][p] + c * 512);
      const half4 hi = sp_tr_read(b_addr[ST][p] + c * 512 + KGB_B);
      fb[SET][c][p] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
  // fragments times the per-k factors of their operand (element j of a fragment is k = 8 kg + j).  The factors of a step
  // are read at its first item, every fragment is scaled a few MFMAs after its read was issued - VALU work in the
  // shadow of the matrix pipe (all of it after the last MFMA of the step cost 40 us of a 125 us launch).
  half8 fvA[2];
  half8 fvB;          // BSC: the factors of B's rows (this lane's 8 k)
  unsigned fb_addr;   // BSC: LDS address (step 0) of those
  template <int ST>
  __device__ __forceinline__ void load_factors(int step) {
#if defined(__HIP_DEVICE_COMPILE__)
    // FIK: the table entry of the step (FSTRIDE bytes per step; steps past the end read the zeroed tail); else the stage's slot
    const unsigned off = FIK ? (unsigned)step * (unsigned)G::FSTRIDE : (unsigned)(ST * G::STG);
    fvA[0] = *reinterpret_cast<const __attribute__((address_space(3))) half8*>((uintptr_t)(f_addr[0] + off));
    fvA[1] = *reinterpret_cast<const __attribute__((address_space(3))) half8*>((uintptr_t)(f_addr[1] + off));
    if constexpr (BSC) fvB = *reinterpret_cast<const __attribute__((address_space(3))) half8*>((uintptr_t)(fb_addr + off));
#else
    (void)step;
#endif
  }
  template <int I, int SET>
  __device__ __forceinline__ void scale_one() {  // I < 4: the A fragments; BSC: I - 4 = the B fragments
    if constexpr (I < 4) {
      constexpr int t = I >> 1, p = I & 1;
      fa[SET][t][p] = fa[SET][t][p] * fvA[t];
    } else {
      constexpr int c = (I - 4) >> 1, p = (I - 4) & 1;
      fb[SET][c][p] = fb[SET][c][p] * fvB;
    }
  }
  template <int I, int N, int SET>
  __device__ __forceinline__ void scale_all() {
    if constexpr (I < N) {
      scale_one<I, SET>();
      scale_all<I + 1, N, SET>();
    }
  }
  template <int I, int SET>
  __device__ __forceinline__ void mfma_one() {
    constexpr int prod = I / (2 * TNW), j = I % (2 * TNW), t = j / TNW, c = j % TNW;
    constexpr int pa = prod == 0 ? 1 : 0, pb = prod == 1 ? 1 : 0;
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[t][c]) : "v"(fa[SET][t][pa]), "v"(fb[SET][c][pb]) : "memory");
  }
  template <int I, int ST>
  __device__ __forceinline__ void dma_one(int step) {
    const unsigned sidx = (unsigned)(step < nsteps ? step : nsteps - 1);
    if constexpr (I < G::ND_A)
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                   :: "s"(m0_a), "n"(ST * G::STG + I * 1024), "v"(voff_a[I]), "s"(rs_a), "s"(sidx * step_bytes_a) : "memory");
    else if constexpr (I < G::ND_A + G::ND_B)
      asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                   :: "s"(m0_b), "n"(ST * G::STG + SP_BM * 64 + (I - G::ND_A) * 1024), "v"(voff_b[I - G::ND_A]), "s"(rs_b),
                   "s"(sidx * step_bytes_b) : "memory");
    else if constexpr (!FIK) {
      if (wave0)  // wave-uniform: one 1 KB slot per stage, fetched once
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                     :: "s"(m0_f), "n"(ST * G::STG + G::FOFF), "v"(voff_f), "s"(rs_f), "s"(sidx * 32u) : "memory");
    }
  }
  template <int I, int N, int ST>
  __device__ __forceinline__ void dma_all(int step) {
    if constexpr (I < N) {
      dma_one<I, ST>(step);
      dma_all<I + 1, N, ST>(step);
    }
  }
  template <int I, int N, int SET, int ST>
  __device__ __forceinline__ void read_all() {
    if constexpr (I < N) {
      read_one<I, SET, ST>();
      read_all<I + 1, N, SET, ST>();
    }
  }
  static constexpr int NR = 4 + 2 * TNW;  // fragments per step (two tr reads each)
  static constexpr int NM = 6 * TNW;
  static constexpr int SLAG0 = NM - 4 < 10 ? NM - 4 : 10;  // an A fragment (the first four reads) is scaled SLAG items
  static constexpr int SLAG = BSC ? (SLAG0 < NM - NR ? SLAG0 : NM - NR) : SLAG0;  // (BSC: every fragment is scaled)
  static_assert((BSC ? NR : 4) + SLAG <= NM, "issue pattern");  // (MFMAs) after its read was issued: the LDS round trip is over
  template <int S, int I>
  __device__ __forceinline__ void step_items(int sbase) {
    if constexpr (I < NM) {
      mfma_one<I, (S & 1)>();
      if constexpr (I == 0) load_factors<((S + 1) % G::NST)>(sbase + S + 1);
      if constexpr (I < NR) read_one<I, ((S + 1) & 1), ((S + 1) % G::NST)>();
      else if constexpr (I < NR + G::ND) dma_one<I - NR, ((S + G::NST - 1) % G::NST)>(sbase + S + G::NST - 1);
      if constexpr (I == NM - 1 && NM - NR < G::ND)  // narrow tiles: fewer bare MFMAs than DMAs - the rest goes last
        dma_all<NM - NR, G::ND, ((S + G::NST - 1) % G::NST)>(sbase + S + G::NST - 1);
      if constexpr (I >= SLAG && I < (BSC ? NR : 4) + SLAG) scale_one<I - SLAG, ((S + 1) & 1)>();
      __builtin_amdgcn_sched_barrier(0);
      step_items<S, I + 1>(sbase);
    }
  }
  template <int S>
  __device__ __forceinline__ void step(int sbase) {
    step_items<S, 0>(sbase);
    wait_landed();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void wait_landed() {
    if (FIK || wave0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::VMW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::VMW1) : "memory");
  }
  template <int S>
  __device__ __forceinline__ void steps(int sbase) {
    if constexpr (S < G::UNR) {
      if (sbase + S < nsteps) step<S>(sbase);
      steps<S + 1>(sbase);
    }
  }
  template <int J>
  __device__ __forceinline__ void dma_prologue() {
    if constexpr (J < G::NST - 1) {
      dma_all<0, G::ND, J>(J);
      dma_prologue<J + 1>();
    }
  }
};

template <int TNW, bool FIK, bool BSC = false>
__global__ void __launch_bounds__(SP_NT, 1) gemm_sp_tn_kernel(SpTnArgs g) {
  using G = SpGeoTN<TNW, FIK, BSC>;
  using LP = SpLoopTN<TNW, FIK, BSC>;
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  All tiles of one
  // K split read the same rows of both operands: lay the (split, tile) pairs out so that one XCD owns whole splits -
  // the B rows of a split are then fetched over the fabric once instead of once per XCD (counters: 500 MB -> the
  // operands' own bytes).
  const unsigned logical = (blockIdx.x & 7u) * g.per_xcd + (blockIdx.x >> 3);
  if (logical >= g.tiles * g.splits) return;
  const unsigned split = logical / g.tiles;
  const unsigned tile = logical - split * g.tiles;
  const unsigned tile_n = tile % g.n_tiles;
  const unsigned tile_m = tile / g.n_tiles;
  const int64_t row0 = (int64_t)tile_m * SP_BM;   // first column of A (= output row)
  const int64_t col0 = (int64_t)tile_n * G::BN;   // first column of B (= output column)
  int64_t k0 = (int64_t)split * g.k_chunk;
  int64_t krows = g.K - k0 < g.k_chunk ? g.K - k0 : g.k_chunk;
  if (g.split_table) {
    k0 = __builtin_amdgcn_readfirstlane(g.split_table[2 * split]);
    krows = __builtin_amdgcn_readfirstlane(g.split_table[2 * split + 1]);
  }
  const int nsteps = (int)((krows + 15) >> 4);

  auto make_rsrc = [](const uint8_t* p, int64_t bytes) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return uint4v{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
                  (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  half8 r_fa[2][2][2];
  half8 r_fb[2][TNW][2];
  floatx16 r_acc[2][TNW];
  unsigned r_aa[G::NST][2], r_ba[G::NST][2], r_va[G::ND_A], r_vb[G::ND_B];
  LP L(r_fa, r_fb, r_acc, r_aa, r_ba, r_va, r_vb);
  // rows past K read as zeros: the descriptors end with the last row of this split
  L.rs_a = make_rsrc(g.A + k0 * g.lda + row0 * 4, krows * g.lda - row0 * 4);
  L.rs_b = make_rsrc(g.B + k0 * g.ldb + col0 * 4, krows * g.ldb - col0 * 4);
  // factors: the scale blocks this tile's 128 columns of A touch (at most 4), 16 k = 32 bytes per block and step
  const int blk_first = (int)((g.a_col0 + row0) / g.a_sb);
  // (M is padded to a multiple of 128: columns past the operand's last block use that block's factors - their rows of
  // the partial result are never read)
  const int blk_last = min((int)((g.a_col0 + row0 + SP_BM - 1) / g.a_sb), g.a_nblk - 1);
  const int nb = blk_last - blk_first + 1;
  if constexpr (!FIK) {
    L.rs_f = make_rsrc(reinterpret_cast<const uint8_t*>(g.F + (int64_t)blk_first * g.f_ld + k0),
                       ((int64_t)(nb - 1) * g.f_ld + ((krows + 15) & ~15ll)) * 2);
    L.voff_f = lane < 2 * nb ? (unsigned)((lane >> 1) * g.f_ld * 2 + (lane & 1) * 16) : 0x7ffffff0u;  // other lanes: zero fill
  } else {
    L.rs_f = uint4v{0u, 0u, 0u, 0u};
    L.voff_f = 0u;
  }
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_void*)lds;
  L.m0_a = __builtin_amdgcn_readfirstlane(lds_base + wave * G::ND_A * 1024);
  L.m0_b = __builtin_amdgcn_readfirstlane(lds_base + wave * G::ND_B * 1024);
  L.m0_f = __builtin_amdgcn_readfirstlane(lds_base);
  L.step_bytes_a = (unsigned)(16 * g.lda);
  L.step_bytes_b = (unsigned)(16 * g.ldb);
  L.nsteps = nsteps;
  L.wave0 = wave == 0;
  // DMA source of lane j of instruction d: container ct = j / 4 -> granule pair ct / 8 of the instruction's two, row
  // (ct % 8) / 2, granule parity ct % 2; chunk j % 4 of the container, halves swapped for rows 2, 3
  {
    const int ct = lane >> 2, cc = lane & 3;
    const int pin = ct >> 3, r4 = (ct & 7) >> 1, gsel = ct & 1;
    const int src_chunk = cc ^ ((r4 >> 1) << 1);
#pragma unroll
    for (int i = 0; i < G::ND_A; ++i) {
      const int d = wave * G::ND_A + i;  // 0 .. 7: row group d / 2, pair pair d % 2
      const int kgp = d >> 1, pp = d & 1;
      L.voff_a[i] = (unsigned)((kgp * 4 + r4) * g.lda + ((pp * 2 + pin) * 2 + gsel) * 64 + src_chunk * 16);
    }
#pragma unroll
    for (int i = 0; i < G::ND_B; ++i) {
      const int d = wave * G::ND_B + i;  // 0 .. 4 TNW - 1: row group d / TNW, pair pair d % TNW
      const int kgp = d / TNW, pp = d % TNW;
      L.voff_b[i] = (unsigned)((kgp * 4 + r4) * g.ldb + ((pp * 2 + pin) * 2 + gsel) * 64 + src_chunk * 16);
    }
  }
  // tr-read addresses: lane -> (column i = lane & 31, k half kg = lane >> 5); piece j = lane & 15 of its 16-lane group
  const int fi = lane & 31, kg = lane >> 5;
  {
    const int j = lane & 15, r4 = j >> 2, gsel = (lane >> 4) & 1;
#pragma unroll
    for (int st = 0; st < G::NST; ++st)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned in_pair = (unsigned)((r4 * 2 + gsel) * 64 + 32 * (p ^ (r4 >> 1)) + (j & 3) * 8);
        L.a_addr[st][p] = lds_base + (unsigned)(st * G::STG + kg * 2 * LP::KGB_A + (wm * 2) * 512) + in_pair;
        L.b_addr[st][p] = lds_base + (unsigned)(st * G::STG + SP_BM * 64 + kg * 2 * LP::KGB_B + (wn * TNW) * 512) + in_pair;
      }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int blk = min((int)((g.a_col0 + row0 + wm * 64 + t * 32 + fi) / g.a_sb), blk_last) - blk_first;
      L.f_addr[t] = lds_base + (unsigned)((FIK ? G::FTAB : G::FOFF) + blk * 32 + kg * 16);
    }
    L.fb_addr = lds_base + (unsigned)(G::FTAB + 128 + kg * 16);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < TNW; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) L.acc[t][c][r] = 0.f;

  constexpr int NR = LP::NR;
  L.template dma_prologue<0>();
  if constexpr (FIK) {
    // ---- the factors of this workgroup's K range (while the first stages are in flight) --------------------------------
    // p[k][j] = inv_a[k0 + k, blk_first + j] * inv_b[k0 + k]; ref[j] = max_k p (1 if all zero); table[step][j][k % 16] =
    // fp16(p / ref[j]) - powers of two, exact down to 2^-24, 0 below and past the range's end
    _Float16* ftab = reinterpret_cast<_Float16*>(lds + G::FTAB);
    float (*fmx)[4] = reinterpret_cast<float (*)[4]>(lds + G::FTAB + SP_TN_FTAB_BYTES - 128);  // [4 waves][4 blocks]
    float* fref = reinterpret_cast<float*>(lds + G::FTAB + SP_TN_FTAB_BYTES - 64);            // 1 / reference of block j
    const int64_t kpad = (int64_t)(nsteps + 1) * 16;  // one zeroed step past the end: the loop reads the factors one step ahead
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    float mxb = 0.f;  // BSC: the largest scale of B's rows (all-zero rows carry the marker 2^-126: never the maximum of a range
                      // that holds anything)
    for (int64_t k = tid; k < krows; k += SP_NT) {
      const float ib = g.inv_b ? g.inv_b[k0 + k] : 1.f;
      if constexpr (BSC) mxb = fmaxf(mxb, ib);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nb) mx[j] = fmaxf(mx[j], g.inv_a[(k0 + k) * g.a_nblk + blk_first + j] * (BSC ? 1.f : ib));
    }
    float frefb = 1.f;
    if constexpr (BSC) {
      float* fmb = reinterpret_cast<float*>(lds + G::FTAB + SP_TN_FTAB_BYTES - 160);  // [4 waves]
#pragma unroll
      for (int o = 32; o; o >>= 1) mxb = fmaxf(mxb, __shfl_xor(mxb, o, 64));
      if (lane == 0) fmb[wave] = mxb;
      __syncthreads();
      mxb = fmaxf(fmaxf(fmb[0], fmb[1]), fmaxf(fmb[2], fmb[3]));
      if (mxb == 0.f) mxb = 1.f;
      frefb = 1.f / mxb;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 32; o; o >>= 1) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], o, 64));
      if (lane == 0) fmx[wave][j] = mx[j];
    }
    __syncthreads();
    if (tid < 4) {
      float m = fmaxf(fmaxf(fmx[0][tid], fmx[1][tid]), fmaxf(fmx[2][tid], fmx[3][tid]));
      if (m == 0.f) m = 1.f;
      fref[tid] = 1.f / m;  // a power of two: exact
      // every tile of the split that touches the block writes the same value (BSC: the product of the two references)
      if (tid < nb) g.ref_split[(int64_t)split * g.a_nblk + blk_first + tid] = BSC ? m * mxb : m;
    }
    __syncthreads();
    bool wide = false;
    constexpr int FH = G::FSTRIDE / 2;  // fp16 per step of the table
    for (int64_t k = tid; k < kpad; k += SP_NT) {
      const float ib = (k < krows && g.inv_b) ? g.inv_b[k0 + k] : 1.f;
      if constexpr (BSC) {
        // BALANCED factors (round 6): what counts is the product F_a[j][k] F_b[k] = 2^-(e_a + e_b), not who carries it.  With
        // each operand scaled by its OWN deficit a row 2^-25 below its operand's largest loses bits (and trips the guard at
        // 2^-22) even when the other operand's row needs no scaling at all - the attention-pooled gradients of configs[2] /
        // configs[3] are spread like that, and their weight gradients went back to the exact bf16x3 kernels.  The deficit of
        // the pair is split evenly instead: F_b = 2^-floor(e / 2), F_a[j] = 2^-(e_a[j] + e_b - floor(e / 2)) with e = e_b + the
        // smallest e_a of the tile's non-zero blocks at k - powers of two, exact; a pair 2^-26 below the range's largest keeps
        // every bit of its high pieces (2^-13 each), the absolute error of a row's term shrinks with the row (<= 2^-25 - e/2
        // of the largest term: a scaled low piece's rounding times the OTHER, scaled, operand) instead of staying 2^-25, and
        // the guard's 2^-22 per factor is reached at 2^-44 for the pair.
        float fa[4], famax = 0.f;
        bool anz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          fa[j] = 0.f;
          anz[j] = false;
          if (j < nb && k < krows) {
            const float ia = g.inv_a[(k0 + k) * g.a_nblk + blk_first + j];
            fa[j] = ia * fref[j];  // powers of two: exact
            anz[j] = ia > 1.2e-38f;  // (all-zero rows carry the marker 2^-126)
            if (anz[j]) famax = fmaxf(famax, fa[j]);
          }
        }
        const bool bnz = k < krows && ib > 1.2e-38f;
        const float fbv = k < krows ? ib * frefb : 0.f;
        float fb2 = fbv, up = 1.f;  // up = fbv / fb2: what B hands over to A
        if (bnz && famax > 0.f) {
          const unsigned ex = (__float_as_uint(famax * fbv) >> 23) & 0xffu;  // the pair's product 2^(ex - 127) <= 1
          if (ex > 0u && ex <= 127u) {
            const unsigned tb = (127u - ex) >> 1;
            fb2 = __uint_as_float((127u - tb) << 23);
            up = fbv * __uint_as_float((127u + tb) << 23);
          }
        }
        ftab[(k >> 4) * FH + 64 + (k & 15)] = (_Float16)fb2;
        // a pair of non-zero rows whose share of the deficit is more than 2^22 (>= 16 bits are left above that)
        wide |= bnz && famax > 0.f && fb2 < 2.384185791015625e-07f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f = fa[j] * up;
          ftab[(k >> 4) * FH + j * 16 + (k & 15)] = (_Float16)f;
          wide |= anz[j] && bnz && f < 2.384185791015625e-07f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float f = 0.f, ia = 0.f;
          if (j < nb && k < krows) {
            ia = g.inv_a[(k0 + k) * g.a_nblk + blk_first + j];
            f = ia * ib * fref[j];  // powers of two: exact
          }
          ftab[(k >> 4) * FH + j * 16 + (k & 15)] = (_Float16)f;
          // the spread guard (see sp_tn_factors_kernel), relative to the largest scale (product) of THIS K range
          wide |= sp_row_too_small(f, ia, ib);
        }
      }
    }
    if (g.spread_flag && __any(wide) && lane == 0) *g.spread_flag = 1;
  }
  L.wait_landed();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  L.template load_factors<0>(0);
  L.template read_all<0, NR, 0, 0>();
  L.template scale_all<0, (BSC ? NR : 4), 0>();
  __builtin_amdgcn_sched_barrier(0);
  for (int s = 0; s < nsteps; s += G::UNR) L.template steps<0>(s);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // partial tile -> workspace (raw sums; the reduce pass scales)
  float* patch = reinterpret_cast<float*>(lds) + wave * 32 * G::PATCH_LD;
  float* __restrict__ part = g.partial + (int64_t)split * g.M * g.N;
  const int64_t wcol0 = col0 + wn * 32 * TNW;
  constexpr int C4 = 8 * TNW;
  constexpr int NIT = 32 * C4 / 64;
  auto store_tile = [&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
#pragma unroll
    for (int c = 0; c < TNW; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int prow = (r & 3) + 8 * (r >> 2) + 4 * kg;
        patch[prow * G::PATCH_LD + c * 32 + fi] = L.acc[t][c][r];
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int64_t wrow0 = row0 + wm * 64 + t * 32;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = lane + it * 64;
      const int pr = idx / C4, c4 = idx - pr * C4;
      const float4 v = *reinterpret_cast<const float4*>(patch + pr * G::PATCH_LD + c4 * 4);
      *reinterpret_cast<float4*>(part + (wrow0 + pr) * g.N + wcol0 + c4 * 4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  store_tile(std::integral_constant<int, 0>{});
  store_tile(std::integral_constant<int, 1>{});
}

__global__ void __launch_bounds__(256) sp_tn_reduce_kernel(AuxTnReduce a) { sp_tn_reduce_body(a, blockIdx.x, gridDim.x); }
// blockIdx.y = row group: the K ranges [split_off[g], split_off[g + 1]) of the grouped product belong to output g
__global__ void __launch_bounds__(256) sp_tn_reduce_grouped_kernel(AuxTnReduce a, const int32_t* __restrict__ split_off,
                                                                   int64_t c_group_stride) {
  const int gq = blockIdx.y;
  const int s0 = split_off[gq], s1 = split_off[gq + 1];
  a.partial += (int64_t)s0 * a.slab;
  a.ref += (int64_t)s0 * a.ref_ld;
  a.splits = s1 - s0;
  a.C += (int64_t)gq * c_group_stride;
  sp_tn_reduce_body(a, blockIdx.x, gridDim.x);
}

static int sp_tn_splits(int64_t M, int64_t N, int64_t K, int bn) {
  const int64_t tiles = (M / SP_BM) * (N / bn);
  const int64_t steps = (K + 15) / 16;
  int64_t splits = std::max<int64_t>(1, std::min<int64_t>(250 / std::max<int64_t>(tiles, 1), steps / 16));
  return (int)std::max<int64_t>(1, splits);
}

template <int TNW>
static void launch_sp_nt(const SpArgs& g_in, dim3 grid, hipStream_t s) {
  using G = SpGeo<TNW>;
  SpArgs g = g_in;
  g.xcd_per = g.xcd_total = 0;
  static const bool xcd_order = [] { const char* e = getenv("TFGNN_SP_NT_XCD"); return !(e && atoi(e) == 0); }();
  if (xcd_order && g.n_tiles > 1 && g.ksplit <= 1 && grid.y == 1 && grid.z == 1 && grid.x >= 16) {
    g.xcd_total = grid.x;
    g.xcd_per = (grid.x + 7u) / 8u;
    grid.x = g.xcd_per * 8u;
  }
  const bool ablk = g.a_inv && g.a_nblk > 1;
  const bool grad = g.mul || g.saved;
  count_launch(TFGNN_KFAM_SP_NT);
#define SP_LAUNCH(AB, GR, OS)                                                                                      \
  do {                                                                                                             \
    static bool attr_set = false;                                                                                  \
    if (!attr_set) {                                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_sp_nt_kernel<TNW, AB, GR, OS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                G::LDS_BYTES);                                                                     \
      attr_set = true;                                                                                             \
    }                                                                                                              \
    hipLaunchKernelGGL((gemm_sp_nt_kernel<TNW, AB, GR, OS>), grid, dim3(SP_NT), G::LDS_BYTES, s, g);               \
  } while (0)
  if (g.out_sp) {
    if (ablk) {
      if (grad) SP_LAUNCH(true, true, true);
      else SP_LAUNCH(true, false, true);
    } else {
      if (grad) SP_LAUNCH(false, true, true);
      else SP_LAUNCH(false, false, true);
    }
  } else if (ablk) {
    if (grad) SP_LAUNCH(true, true, false);
    else SP_LAUNCH(true, false, false);
  } else {
    if (grad) SP_LAUNCH(false, true, false);
    else SP_LAUNCH(false, false, false);
  }
#undef SP_LAUNCH
}

// Workspace of the in-launch K split of the NT product (tfgnn_sp_gemm_nt_set_splitk_workspace): [flags 64 KB][slabs].
constexpr size_t kSplitkFlagBytes = 65536;
// ONE workspace per process: it belongs to the device that was current when it was registered, and its flags and
// slabs are indexed by tile only - so a split launch (a) never happens on another device, (b) on another STREAM than the
// previous split launch first waits (host side) for that stream (a stream that is being captured cannot wait: its owner
// synchronises the device before the capture; while ANOTHER stream is capturing split products, this one stays unsplit),
// (c) after a reducer timeout (a producer never arrived: the product that timed out is WRONG) the next product call fails
// loudly, zeroes the flags - a late producer may have left one set - and reports through tfgnn_sp_gemm_nt_splitk_status.
struct SplitkWorkspace {
  void* base = nullptr;
  size_t bytes = 0;
  bool enabled = true;
  long long split_launches = 0;
  long long timeouts = 0;
  int device = -1;
  hipStream_t last_stream = nullptr;
  bool used = false;
  int* timeout_host = nullptr;
  int* timeout_dev = nullptr;
};
static SplitkWorkspace g_splitk;

static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return st != hipStreamCaptureStatusNone;
}

// May THIS product split K through the shared workspace?  <0: error (a reducer timed out earlier), 0: no, 1: yes.
static int splitk_admit(hipStream_t s) {
  if (!(g_splitk.enabled && g_splitk.base)) return 0;
  if (g_splitk.timeout_host && __atomic_load_n(g_splitk.timeout_host, __ATOMIC_RELAXED)) {
    ++g_splitk.timeouts;
    __atomic_store_n(g_splitk.timeout_host, 0, __ATOMIC_RELAXED);
    (void)hipDeviceSynchronize();  // nothing in flight may still touch the flags
    (void)hipMemset(g_splitk.base, 0, kSplitkFlagBytes);
    set_error("tfgnn_sp_gemm_nt: a reducer of an earlier K-split product gave up waiting for its producers - that product's "
              "result is incomplete; the workspace flags were cleared (tfgnn_sp_gemm_nt_splitk_status reports the count)");
    return -1;
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != g_splitk.device) {
    (void)hipGetLastError();
    return 0;  // the workspace lives on another device: this product runs unsplit
  }
  if (g_splitk.used && s != g_splitk.last_stream) {
    if (stream_is_capturing(g_splitk.last_stream)) return 0;  // somebody else's capture is recording split products
    if (!stream_is_capturing(s)) {
      // two split products never share the flags in flight: wait for the stream that launched the previous one
      if (hipStreamSynchronize(g_splitk.last_stream) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
      }
    }
    // (a CAPTURING stream cannot wait for another one; whoever captures a step synchronises the device first -
    //  capture.CapturedStep does - and replays then run on the stream the caller launches them on, like any product)
  }
  g_splitk.last_stream = s;
  g_splitk.used = true;
  return 1;
}
// (Helper workgroups for the heavy tiles of a masked product - round 5, measured a loss: 112 / 101 us against 90 us per forward
//  product, NOTEBOOK 10 - were removed in round 6 together with their export.)

// row slabs of the column-maxima pass of a long [K, N] kernel stack: ~160 rows each, at most 256
static int sp_colmax_parts(int64_t K) { return (int)std::max<int64_t>(1, std::min<int64_t>(256, K / 160)); }

static int sp_tile_width(int64_t N) { return N % 320 == 0 ? 320 : (N % 256 == 0 ? 256 : (N % 128 == 0 ? 128 : 0)); }

// one int in host memory mapped into the device's address space: the factor pass of the weight-gradient product stores 1
// there when an operand's row scales spread over more than 2^20 (plain store, no atomic: the value only ever becomes 1);
// the host reads it without synchronising with any stream
static int* g_spread_host = nullptr;
static int* g_spread_dev = nullptr;
static int* sp_spread_flag_device() {
  if (!g_spread_host) {
    int* h = nullptr;
    if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    *h = 0;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostFree(h);
      return nullptr;
    }
    g_spread_host = h;
    g_spread_dev = (int*)d;
  }
  return g_spread_dev;
}

}  // namespace tfgnn

using namespace tfgnn;

extern "C" {

size_t tfgnn_sp_bytes(int64_t rows, int64_t cols) { return (size_t)rows * (size_t)cols * 4; }

int tfgnn_sp_spread_flag(int reset) {
  if (!g_spread_host) return 0;
  const int v = __atomic_load_n(g_spread_host, __ATOMIC_RELAXED);
  if (reset) __atomic_store_n(g_spread_host, 0, __ATOMIC_RELAXED);
  return v;
}

int tfgnn_sp_split_rows(const float* d_src, int64_t ld, int64_t seg_len, int64_t seg_stride, int64_t rows, int64_t cols,
                        int scale_block, void* d_sp, int64_t ld_sp_bytes, float* d_inv_scale,
                        const float* d_fixed_inv_scale, void* stream) {
  TFGNN_REQUIRE(d_src && d_sp, "tfgnn_sp_split_rows: null pointer");
  TFGNN_REQUIRE(rows >= 0 && cols > 0 && cols % 16 == 0, "tfgnn_sp_split_rows: cols must be a positive multiple of 16");
  if (scale_block <= 0) scale_block = (int)cols;
  if (seg_len <= 0) { seg_len = cols; seg_stride = 0; }
  TFGNN_REQUIRE(scale_block % 16 == 0 && cols % scale_block == 0, "tfgnn_sp_split_rows: scale_block must divide cols and be a multiple of 16");
  TFGNN_REQUIRE(seg_len % 4 == 0 && cols % seg_len == 0 && ld % 4 == 0 && seg_stride % 4 == 0 && (uintptr_t)d_src % 16 == 0,
                "tfgnn_sp_split_rows: source must be 16-byte aligned with segment length / strides multiples of 4");
  TFGNN_REQUIRE(ld_sp_bytes >= cols * 4 && ld_sp_bytes % 64 == 0 && (uintptr_t)d_sp % 64 == 0, "tfgnn_sp_split_rows: bad SP16 leading dimension / alignment");
  if (rows == 0) return TFGNN_OK;
  if (!d_fixed_inv_scale && scale_block == cols && (seg_len <= 0 || seg_len >= cols) && rows >= 4096 &&
      (cols == 64 || cols == 128 || cols == 256)) {
    const unsigned rpb = (unsigned)(256 / (cols / 4));
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(rows, rpb), 256 * 32);
    hipStream_t s = (hipStream_t)stream;
    if (cols == 64) hipLaunchKernelGGL(sp_split_rows_narrow_kernel<16>, dim3(grid), dim3(256), 0, s, d_src, ld, rows, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale);
    else if (cols == 128) hipLaunchKernelGGL(sp_split_rows_narrow_kernel<32>, dim3(grid), dim3(256), 0, s, d_src, ld, rows, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale);
    else hipLaunchKernelGGL(sp_split_rows_narrow_kernel<64>, dim3(grid), dim3(256), 0, s, d_src, ld, rows, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale);
    TFGNN_LAUNCH_CHECK();
    return TFGNN_OK;
  }
  const int64_t items = rows * (cols / scale_block);
  hipLaunchKernelGGL(sp_split_rows_kernel, dim3((unsigned)ceil_div(items, 4)), dim3(256), 0, (hipStream_t)stream, d_src, ld,
                     seg_len, seg_stride, rows, cols, scale_block, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale, d_fixed_inv_scale);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int tfgnn_sp_split_cols(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                        float* d_inv_scale, void* stream) {
  TFGNN_REQUIRE(d_src && d_sp, "tfgnn_sp_split_cols: null pointer");
  TFGNN_REQUIRE(K > 0 && N > 0 && K % 16 == 0 && N % 4 == 0 && ld % 4 == 0 && (uintptr_t)d_src % 16 == 0,
                "tfgnn_sp_split_cols: K must be a multiple of 16, N and ld multiples of 4");
  TFGNN_REQUIRE(ld_sp_bytes >= K * 4 && ld_sp_bytes % 64 == 0 && (uintptr_t)d_sp % 64 == 0, "tfgnn_sp_split_cols: bad SP16 leading dimension / alignment");
  hipLaunchKernelGGL(sp_split_cols_kernel, dim3((unsigned)ceil_div(N, 16), (unsigned)std::max<int64_t>(1, std::min<int64_t>(8, K / 128))), dim3(256), 0, (hipStream_t)stream, d_src, ld, K, N,
                     (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int tfgnn_sp_split_weights(int count, const float* const* h_src, int64_t L, int64_t D, int64_t H, void* const* h_cols_sp,
                           float* const* h_cols_inv_scale, void* const* h_rows_sp, float* const* h_rows_inv_scale, void* stream) {
  TFGNN_REQUIRE(count >= 1 && count <= 16 && h_src && h_cols_sp && h_cols_inv_scale && h_rows_sp && h_rows_inv_scale,
                "tfgnn_sp_split_weights: 1 .. 16 kernel stacks");
  TFGNN_REQUIRE(L > 0 && D > 0 && H > 0 && (L * D) % 16 == 0 && (L * H) % 16 == 0 && H % 4 == 0,
                "tfgnn_sp_split_weights: L D and L H must be multiples of 16, H a multiple of 4");
  SpWeightJobs j{};
  for (int i = 0; i < count; ++i) {
    TFGNN_REQUIRE(h_src[i] && h_cols_sp[i] && h_cols_inv_scale[i] && h_rows_sp[i] && h_rows_inv_scale[i] &&
                      (uintptr_t)h_src[i] % 16 == 0 && (uintptr_t)h_cols_sp[i] % 64 == 0 && (uintptr_t)h_rows_sp[i] % 64 == 0,
                  "tfgnn_sp_split_weights: null or unaligned pointer");
    j.src[i] = h_src[i]; j.cols_sp[i] = (uint8_t*)h_cols_sp[i]; j.cols_inv[i] = h_cols_inv_scale[i];
    j.rows_sp[i] = (uint8_t*)h_rows_sp[i]; j.rows_inv[i] = h_rows_inv_scale[i];
  }
  j.L = L; j.D = D; j.H = H;
  j.ncx = (unsigned)ceil_div(H, 16);
  j.ncy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8, L * D / 128));
  const unsigned nr = (unsigned)ceil_div(D, 4);  // one wave per row of the stacked-rows form
  hipLaunchKernelGGL(sp_split_weights_kernel, dim3(j.ncx * j.ncy + nr, (unsigned)count), dim3(256), 0, (hipStream_t)stream, j);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

static int sp_gemm_nt_impl(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                           int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                           int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                           int act_of_saved, const float* d_saved, int64_t ld_saved, void* d_out_sp, int64_t ld_out_sp_bytes,
                           float* d_out_inv_scale, void* stream, float dropout_rate = 0.f, uint64_t dropout_seed = 0,
                           float saved_scale = 1.f, const uint8_t* d_tile_kmask = nullptr, const int32_t* d_row_map = nullptr,
                           const int32_t* d_a_rows = nullptr, int64_t a_src_rows = 0, const int32_t* d_tile_table = nullptr,
                           int64_t num_tiles_m = 0, int num_groups = 1, int64_t b_group_stride_bytes = -1,
                           int64_t b_scale_group_stride = -1) {
  TFGNN_REQUIRE(d_A_sp && d_B_sp && (d_C || d_out_sp), "tfgnn_sp_gemm_nt: null pointer");
  if (!d_a_rows || a_src_rows <= 0) a_src_rows = M;
  TFGNN_REQUIRE(!d_a_rows || a_src_rows * lda_bytes < (1ll << 32) - 65536, "tfgnn_sp_gemm_nt_rows: the operand read through a row index must be below 4 GB");
  TFGNN_REQUIRE(!d_tile_table || (num_tiles_m > 0 && num_groups >= 1 && !d_tile_kmask && !d_row_map),
                "tfgnn_sp_gemm_nt_grouped: a tile table needs its tile count and excludes the tile mask / row map");
  TFGNN_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "tfgnn_sp_gemm_nt: dropout rate must be in [0, 1), got %f", (double)dropout_rate);
  TFGNN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "tfgnn_sp_gemm_nt: K must be a positive multiple of 16");
  const int bn = sp_tile_width(N);
  if (!bn) {
    set_error("tfgnn_sp_gemm_nt: N = %lld is not a multiple of 128", (long long)N);
    return TFGNN_ERR_UNSUPPORTED;
  }
  TFGNN_REQUIRE(lda_bytes % 64 == 0 && ldb_bytes % 64 == 0 && lda_bytes >= K * 4 && ldb_bytes >= K * 4 &&
                    (uintptr_t)d_A_sp % 64 == 0 && (uintptr_t)d_B_sp % 64 == 0,
                "tfgnn_sp_gemm_nt: SP16 operands must be 64-byte aligned with leading dimensions >= 4 K bytes");
  TFGNN_REQUIRE((!d_C || (ldc % 4 == 0 && (uintptr_t)d_C % 16 == 0)) && (!d_bias || (uintptr_t)d_bias % 16 == 0) &&
                    (!d_b_inv_scale || (uintptr_t)d_b_inv_scale % 16 == 0) && (!d_mul || (ld_mul % 4 == 0 && (uintptr_t)d_mul % 16 == 0)) &&
                    (!d_saved || (ld_saved % 4 == 0 && (uintptr_t)d_saved % 16 == 0)),
                "tfgnn_sp_gemm_nt: C / bias / scale / factor operands must be 16-byte aligned with ld %% 4 == 0");
  TFGNN_REQUIRE(128 * lda_bytes < (1ll << 31) && 320 * ldb_bytes < (1ll << 31), "tfgnn_sp_gemm_nt: K too large");
  SpArgs g{};
  g.M = M; g.N = N; g.K = K;
  g.A = (const uint8_t*)d_A_sp; g.lda = lda_bytes; g.a_inv = d_a_inv_scale;
  const bool a_uniform = a_scale_block < 0;  // one scale for the whole tensor
  if (a_scale_block <= 0 || a_scale_block >= K) a_scale_block = (int)K;
  TFGNN_REQUIRE(a_scale_block % 16 == 0 && K % a_scale_block == 0, "tfgnn_sp_gemm_nt: a_scale_block must divide K and be a multiple of 16");
  g.a_nblk = (int)(K / a_scale_block);
  g.a_inv_ld = a_uniform ? 0 : g.a_nblk;
  g.a_blk_steps = a_scale_block / 16;
  g.B = (const uint8_t*)d_B_sp; g.ldb = ldb_bytes; g.b_inv = d_b_inv_scale;
  g.C = d_C; g.ldc = ldc; g.bias = d_bias; g.act = act; g.accumulate = accumulate;
  g.mul = d_mul; g.ld_mul = ld_mul; g.saved = d_saved; g.ld_saved = ld_saved; g.dact = act_of_saved;
  g.saved_scale = saved_scale;
  if (d_tile_kmask) {
    TFGNN_REQUIRE(g.a_nblk >= 1 && g.a_nblk <= 8 && g.a_blk_steps <= 64 && g.a_blk_steps * g.a_nblk == (int)(K >> 4) && g.a_inv &&
                      !a_uniform,
                  "tfgnn_sp_gemm_nt: the tile mask needs 1..8 scale blocks of A (of at most 1024 columns) that tile K");
    g.tile_kmask = d_tile_kmask;
  }
  g.row_map = d_row_map;
  g.a_rows = d_a_rows;
  g.a_src_rows = a_src_rows;
  g.tile_table = d_tile_table;
  g.group_stride_b = b_group_stride_bytes >= 0 ? b_group_stride_bytes : N * ldb_bytes;
  g.group_stride_scale = b_scale_group_stride >= 0 ? b_scale_group_stride : N;
  TFGNN_REQUIRE(g.group_stride_b % 64 == 0, "tfgnn_sp_gemm_nt_grouped: group stride of B must be a multiple of 64 bytes");
  g.drop_on = dropout_rate > 0.f ? (dropout_seed == ~0ull ? 2 : 1) : 0;
  if (g.drop_on == 2)
    TFGNN_REQUIRE(d_saved && act_of_saved == TFGNN_ACT_RELU, "tfgnn_sp_gemm_nt_dropout: the mask-from-saved form needs a relu saved tensor");
  g.drop_ld = N;
  g.drop = dropout_key(dropout_seed, dropout_rate);
  TFGNN_REQUIRE(dropout_rate <= 0.f || g.drop.epoch, "tfgnn_sp_gemm_nt_dropout: no device memory for this device's epoch word");
  g.n_tiles = (unsigned)(N / bn);
  if (d_out_sp) {
    // one scale per row and COLUMN TILE (N = bn: one per row; N = 512 = 2 x 256: scale blocks of 256 columns - round 5)
    TFGNN_REQUIRE(!accumulate && d_out_inv_scale, "tfgnn_sp_gemm_nt_sp: the split result needs no accumulation and an inverse-scale array");
    TFGNN_REQUIRE(ld_out_sp_bytes >= N * 4 && ld_out_sp_bytes % 64 == 0 && (uintptr_t)d_out_sp % 64 == 0,
                  "tfgnn_sp_gemm_nt_sp: SP16 result rows must be 64-byte aligned and at least 4 N bytes");
    g.out_sp = (uint8_t*)d_out_sp; g.ld_out_sp = ld_out_sp_bytes; g.out_inv = d_out_inv_scale;
  }
  const int64_t tiles = (d_tile_table ? num_tiles_m : ceil_div(M, SP_BM)) * g.n_tiles;
  TFGNN_REQUIRE(tiles <= 0x7fffffff, "tfgnn_sp_gemm_nt: too many tiles");
  // K split inside the launch (SpArgs::ksplit): only where one wave of workgroups leaves most of the chip idle - every
  // workgroup of a split launch must be RESIDENT (the reducers wait for the producers), one per CU
  g.ksplit = 1;
  {
    const int64_t steps = K >> 4;
    int S = (int)std::min<int64_t>(4, steps / 15);  // (K = 320 split in two measured slower: 52 vs 51 us at 56 tiles - the hand-off costs what 10 steps do)
    while (S > 1 && tiles * S > 232) --S;
    if (g_splitk.enabled && g_splitk.base && S > 1 && tiles <= 112 && !d_tile_table) {
      const size_t slab = (size_t)SP_BM * bn * 4;
      const size_t need = kSplitkFlagBytes + (size_t)tiles * (S - 1) * slab;
      const int admit = (need <= g_splitk.bytes && (size_t)tiles * (S - 1) * 4 <= kSplitkFlagBytes) ? splitk_admit((hipStream_t)stream) : 0;
      if (admit < 0) return TFGNN_ERR_HIP;
      if (admit > 0) {
        g.ksplit = S;
        ++g_splitk.split_launches;
        g.ws_flags = (unsigned*)g_splitk.base;
        g.ws_partial = (float*)((char*)g_splitk.base + kSplitkFlagBytes);
        g.ws_timeout = g_splitk.timeout_dev;
      }
    }
  }
  dim3 grid((unsigned)(tiles * g.ksplit));
  hipStream_t s = (hipStream_t)stream;
  if (bn == 320) launch_sp_nt<5>(g, grid, s);
  else if (bn == 256) launch_sp_nt<4>(g, grid, s);
  else launch_sp_nt<2>(g, grid, s);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}


int tfgnn_sp_gemm_nt_set_splitk_workspace(void* d_workspace, size_t bytes) {
  // (waits for the device: nothing in flight may still be using the previous workspace)
  TFGNN_HIP_CHECK(hipDeviceSynchronize());
  if (!d_workspace || bytes <= kSplitkFlagBytes) {
    g_splitk.base = nullptr;
    g_splitk.bytes = 0;
    return TFGNN_OK;
  }
  TFGNN_REQUIRE((uintptr_t)d_workspace % 256 == 0, "tfgnn_sp_gemm_nt_set_splitk_workspace: the workspace must be 256-byte aligned");
  if (!g_splitk.timeout_host) {
    int* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
      *h = 0;
      g_splitk.timeout_host = h;
      g_splitk.timeout_dev = (int*)d;
    } else {
      (void)hipGetLastError();
    }
  }
  TFGNN_HIP_CHECK(hipMemset(d_workspace, 0, kSplitkFlagBytes));  // the flags are zero between launches from here on
  TFGNN_HIP_CHECK(hipGetDevice(&g_splitk.device));
  g_splitk.base = d_workspace;
  g_splitk.bytes = bytes;
  g_splitk.used = false;
  g_splitk.last_stream = nullptr;
  return TFGNN_OK;
}

int tfgnn_sp_gemm_nt_splitk_status(int enable, int* timed_out, int64_t* split_launches) {
  if (enable >= 0) g_splitk.enabled = enable != 0;
  if (split_launches) *split_launches = g_splitk.split_launches;
  if (timed_out)  // timeouts already turned into an error + the flag of one nobody has seen yet
    *timed_out = (int)g_splitk.timeouts + (g_splitk.timeout_host ? __atomic_load_n(g_splitk.timeout_host, __ATOMIC_RELAXED) : 0);
  return (g_splitk.enabled && g_splitk.base) ? 1 : 0;
}

int tfgnn_sp_gemm_nt(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                     int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                     int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                     int act_of_saved, const float* d_saved, int64_t ld_saved, void* stream) {
  TFGNN_REQUIRE(d_C, "tfgnn_sp_gemm_nt: null pointer");
  return sp_gemm_nt_impl(M, N, K, d_A_sp, lda_bytes, d_a_inv_scale, a_scale_block, d_B_sp, ldb_bytes, d_b_inv_scale, d_C, ldc, d_bias,
                         act, accumulate, d_mul, ld_mul, act_of_saved, d_saved, ld_saved, nullptr, 0, nullptr, stream);
}

int tfgnn_sp_gemm_nt_sp(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                        int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                        int64_t ldc, const float* d_bias, int act, const float* d_mul, int64_t ld_mul, int act_of_saved,
                        const float* d_saved, int64_t ld_saved, void* d_out_sp, int64_t ld_out_sp_bytes,
                        float* d_out_inv_scale, void* stream) {
  TFGNN_REQUIRE(d_out_sp, "tfgnn_sp_gemm_nt_sp: null pointer");
  return sp_gemm_nt_impl(M, N, K, d_A_sp, lda_bytes, d_a_inv_scale, a_scale_block, d_B_sp, ldb_bytes, d_b_inv_scale, d_C, ldc, d_bias,
                         act, 0, d_mul, ld_mul, act_of_saved, d_saved, ld_saved, d_out_sp, ld_out_sp_bytes, d_out_inv_scale, stream);
}

int tfgnn_sp_gemm_nt_dropout(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale, float* d_C,
                             int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                             int act_of_saved, const float* d_saved, int64_t ld_saved, float saved_scale, void* d_out_sp,
                             int64_t ld_out_sp_bytes, float* d_out_inv_scale, float dropout_rate, uint64_t dropout_seed,
                             const uint8_t* d_tile_kmask, const int32_t* d_row_map, void* stream) {
  return sp_gemm_nt_impl(M, N, K, d_A_sp, lda_bytes, d_a_inv_scale, a_scale_block, d_B_sp, ldb_bytes, d_b_inv_scale, d_C, ldc, d_bias,
                         act, accumulate, d_mul, ld_mul, act_of_saved, d_saved, ld_saved, d_out_sp, ld_out_sp_bytes, d_out_inv_scale,
                         stream, dropout_rate, dropout_seed, saved_scale, d_tile_kmask, d_row_map);
}

int tfgnn_sp_gemm_nt_rows(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                          int a_scale_block, const int32_t* d_a_rows, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale,
                          float* d_C, int64_t ldc, const float* d_bias, int act, int accumulate, const float* d_mul, int64_t ld_mul,
                          int act_of_saved, const float* d_saved, int64_t ld_saved, float saved_scale, void* d_out_sp,
                          int64_t ld_out_sp_bytes, float* d_out_inv_scale, float dropout_rate, uint64_t dropout_seed,
                          const uint8_t* d_tile_kmask, const int32_t* d_row_map, void* stream) {
  return sp_gemm_nt_impl(M, N, K, d_A_sp, lda_bytes, d_a_inv_scale, a_scale_block, d_B_sp, ldb_bytes, d_b_inv_scale, d_C, ldc, d_bias,
                         act, accumulate, d_mul, ld_mul, act_of_saved, d_saved, ld_saved, d_out_sp, ld_out_sp_bytes, d_out_inv_scale,
                         stream, dropout_rate, dropout_seed, saved_scale, d_tile_kmask, d_row_map, d_a_rows);
}

int tfgnn_sp_gemm_nt_grouped(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int a_scale_block, const int32_t* d_a_rows, int64_t a_src_rows, const int32_t* d_tile_table,
                             int64_t num_tiles, int num_groups, const void* d_B_sp, int64_t ldb_bytes, int64_t b_group_stride_bytes,
                             const float* d_b_inv_scale, int64_t b_scale_group_stride, float* d_C, int64_t ldc, const float* d_bias,
                             int act, const float* d_mul, int64_t ld_mul, int act_of_saved, const float* d_saved, int64_t ld_saved,
                             void* d_out_sp, int64_t ld_out_sp_bytes, float* d_out_inv_scale, void* stream) {
  TFGNN_REQUIRE(d_tile_table && num_tiles > 0 && num_groups >= 1, "tfgnn_sp_gemm_nt_grouped: tile table missing");
  return sp_gemm_nt_impl(M, N, K, d_A_sp, lda_bytes, d_a_inv_scale, a_scale_block, d_B_sp, ldb_bytes, d_b_inv_scale, d_C, ldc, d_bias,
                         act, 0, d_mul, ld_mul, act_of_saved, d_saved, ld_saved, d_out_sp, ld_out_sp_bytes, d_out_inv_scale, stream,
                         0.f, 0, 1.f, nullptr, nullptr, d_a_rows, a_src_rows, d_tile_table, num_tiles, num_groups,
                         b_group_stride_bytes, b_scale_group_stride);
}

static size_t sp_gemm_tn_ws_bytes(int64_t M, int64_t N, int64_t K, int64_t a_total_cols, int a_scale_block, bool wide) {
  const int bn = sp_tile_width(N);
  if (!bn || M <= 0 || a_scale_block <= 0) return 0;
  const int64_t Mp = ceil_div(M, SP_BM) * SP_BM;
  const int64_t nblk = ceil_div(a_total_cols, a_scale_block), kpad = (K + 15) & ~15ll;
  const size_t factors = (((size_t)nblk * kpad * 2 + 255) & ~(size_t)255) + (((size_t)nblk * (1 + SP_TN_MAXCHUNKS + 512) * 4 + 255) & ~(size_t)255);
  int64_t splits = sp_tn_splits(Mp, N, K, bn);
  if (wide) splits = std::max<int64_t>(splits, ceil_div(K, SP_TN_BSC_MAX_CHUNK));
  return factors + (size_t)splits * (size_t)Mp * (size_t)N * 4;
}

size_t tfgnn_sp_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t a_total_cols, int a_scale_block) {
  return sp_gemm_tn_ws_bytes(M, N, K, a_total_cols, a_scale_block, false);
}

size_t tfgnn_sp_gemm_tn_wide_workspace_bytes(int64_t M, int64_t N, int64_t K, int64_t a_total_cols, int a_scale_block) {
  return sp_gemm_tn_ws_bytes(M, N, K, a_total_cols, a_scale_block, true);
}

static int sp_gemm_tn_impl(int phases, int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                           const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                           int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                           int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                           size_t workspace_bytes, void* stream, tfgnn_aux_job* reduce_job = nullptr,
                           tfgnn_aux_job* factors_job = nullptr, bool wide = false) {
  TFGNN_REQUIRE(d_A_sp && d_B_sp && d_C && d_a_inv_scale, "tfgnn_sp_gemm_tn: null pointer");
  TFGNN_REQUIRE(M > 0 && N > 0 && K > 0, "tfgnn_sp_gemm_tn: empty product");
  const int bn = sp_tile_width(N);
  if (!bn || M % 16 || a_scale_block < 32) {
    set_error("tfgnn_sp_gemm_tn: M = %lld must be a multiple of 16, N = %lld of 128, scale blocks of A >= 32 columns",
              (long long)M, (long long)N);
    return TFGNN_ERR_UNSUPPORTED;
  }
  // row tiles are 128 columns of A: the last one may run past M (it reads the neighbouring bytes of the operand rows,
  // zeros past the operand's end; its surplus result rows stay in the workspace)
  const int64_t Mp = ceil_div(M, SP_BM) * SP_BM;
  TFGNN_REQUIRE(lda_bytes % 64 == 0 && ldb_bytes % 64 == 0 && (uintptr_t)d_A_sp % 64 == 0 && (uintptr_t)d_B_sp % 64 == 0 &&
                    a_first_col % 16 == 0 && b_first_col % 16 == 0 && a_first_col >= 0 && b_first_col >= 0 &&
                    lda_bytes >= (a_first_col + M) * 4 && ldb_bytes >= (b_first_col + N) * 4 && a_total_cols >= a_first_col + M &&
                    a_total_cols % a_scale_block == 0,
                "tfgnn_sp_gemm_tn: SP16 operands must be 64-byte aligned, first columns multiples of 16, rows wide enough");
  TFGNN_REQUIRE(group_rows > 0, "tfgnn_sp_gemm_tn: group_rows must be positive");
  TFGNN_REQUIRE(a_scale_block >= 48 || a_first_col % a_scale_block == 0,
                "tfgnn_sp_gemm_tn: 32-column scale blocks need a block-aligned first column (at most 4 blocks per 128-column tile)");
  int splits = sp_tn_splits(Mp, N, K, bn);
  if (wide) {  // two-factor form: factors in the kernel, K ranges of at most SP_TN_BSC_MAX_CHUNK rows
    splits = (int)std::max<int64_t>(splits, ceil_div(K, SP_TN_BSC_MAX_CHUNK));
    TFGNN_REQUIRE(splits <= 512 && (phases == (1 | 2 | 4) || phases == (1 | 2 | 8)) && d_b_inv_scale,
                  "tfgnn_sp_gemm_tn_wide: K too large (more than 512 ranges of 2016 rows), or no scales of B");
  }
  const int64_t nblk = a_total_cols / a_scale_block, kpad = (K + 15) & ~15ll;
  const size_t f_bytes = ((size_t)nblk * kpad * 2 + 255) & ~(size_t)255, r_bytes = ((size_t)nblk * (1 + SP_TN_MAXCHUNKS + 512) * 4 + 255) & ~(size_t)255;
  const size_t need = f_bytes + r_bytes + (size_t)splits * (size_t)Mp * (size_t)N * 4;
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= need && (uintptr_t)d_workspace % 256 == 0,
                "tfgnn_sp_gemm_tn: workspace too small or unaligned (need %zu bytes)", need);
  hipStream_t s = (hipStream_t)stream;
  _Float16* F = (_Float16*)d_workspace;
  float* ref = (float*)((uint8_t*)d_workspace + f_bytes);
  // factors in the kernel (FIK): every workgroup normalises its own K range - no factor pass at all
  static const bool fik_env = [] { const char* e = getenv("TFGNN_TN_FIK"); return !e || atoi(e) != 0; }();
  const int64_t steps_all = (K + 15) / 16;
  const int64_t k_chunk_all = ((steps_all + splits - 1) / splits) * 16;
  const bool fik = wide || (fik_env && k_chunk_all <= SP_TN_FIK_MAX_CHUNK && splits <= 512);
  TFGNN_REQUIRE(!wide || k_chunk_all <= SP_TN_BSC_MAX_CHUNK, "tfgnn_sp_gemm_tn_wide: K range too long");
  float* ref_split = ref + nblk * (1 + SP_TN_MAXCHUNKS);  // [splits][nblk]
  if (fik) phases &= ~(1 | 16);
  if ((phases & 16) && factors_job) factors_job->kind = 0, factors_job->num_blocks = 0;
  if (phases & 16) {  // the factor pass as a job of a merged launch (operands of up to 128k rows: one-stage factor pass)
    TFGNN_REQUIRE(factors_job != nullptr && sp_tn_fchunks(K) == SP_TN_FCHUNKS, "tfgnn_sp_gemm_tn: factors job needs K <= 131072");
    AuxTnFactors fa{d_a_inv_scale, nblk, d_b_inv_scale, 1, K, F, kpad, ref, sp_spread_flag_device(), SP_TN_FCHUNKS};
    aux_job_set(factors_job, AUX_TN_FACTORS, (unsigned)(nblk * SP_TN_FCHUNKS), fa);
  }
  if (phases & 1) {
    const int fch = sp_tn_fchunks(K);
    float* slice_max = nullptr;
    if (fch > SP_TN_FCHUNKS) {  // two stages; the slice maxima live behind the reference scales (r_bytes reserves nblk * SP_TN_MAXCHUNKS floats for them)
      slice_max = ref + nblk;
      hipLaunchKernelGGL(sp_tn_slice_max_kernel, dim3((unsigned)nblk * fch), dim3(1024), 0, s, d_a_inv_scale, nblk, d_b_inv_scale, (int64_t)1, K,
                         slice_max, fch);
    }
    hipLaunchKernelGGL(sp_tn_factors_kernel, dim3((unsigned)nblk * fch), dim3(1024), 0, s, d_a_inv_scale, nblk, d_b_inv_scale, (int64_t)1, K, F,
                       kpad, ref, sp_spread_flag_device(), fch, (const float*)slice_max);
    TFGNN_LAUNCH_CHECK();
  }
  SpTnArgs g{};
  g.M = Mp; g.N = N; g.K = K;
  g.A = (const uint8_t*)d_A_sp + a_first_col * 4; g.lda = lda_bytes;
  g.B = (const uint8_t*)d_B_sp + b_first_col * 4; g.ldb = ldb_bytes;
  g.F = F; g.f_ld = kpad; g.a_sb = a_scale_block; g.a_col0 = a_first_col; g.a_nblk = (int)nblk;
  g.inv_a = d_a_inv_scale; g.inv_b = d_b_inv_scale; g.ref_split = ref_split; g.spread_flag = sp_spread_flag_device();
  g.partial = (float*)((uint8_t*)d_workspace + f_bytes + r_bytes);
  const int64_t steps = (K + 15) / 16;
  g.k_chunk = ((steps + splits - 1) / splits) * 16;
  const int splits_used = (int)ceil_div(K, g.k_chunk);  // rounding the chunk up to whole steps can leave the last splits empty
  g.n_tiles = (unsigned)(N / bn);
  TFGNN_REQUIRE(g.k_chunk * std::max(lda_bytes, ldb_bytes) < (1ll << 31) && nblk * kpad * 2 < (1ll << 31), "tfgnn_sp_gemm_tn: K chunk too large");
  g.tiles = (unsigned)((Mp / SP_BM) * g.n_tiles);
  g.splits = (unsigned)splits_used;
  g.per_xcd = (g.tiles * g.splits + 7) / 8;
  dim3 grid(8 * g.per_xcd);
#define SP_LAUNCH_TN(T, FK, ...)                                                                                   \
  do {                                                                                                             \
    static bool attr_set = false;                                                                                  \
    using TnGeo = SpGeoTN<T, FK, ##__VA_ARGS__>;                                                                   \
    constexpr int tn_lds = TnGeo::LDS_BYTES;                                                                       \
    if (!attr_set) {                                                                                               \
      (void)hipFuncSetAttribute((const void*)gemm_sp_tn_kernel<T, FK, ##__VA_ARGS__>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, tn_lds);                               \
      attr_set = true;                                                                                             \
    }                                                                                                              \
    hipLaunchKernelGGL((gemm_sp_tn_kernel<T, FK, ##__VA_ARGS__>), grid, dim3(SP_NT), tn_lds, s, g);                \
  } while (0)
  if (phases & 2) {
    count_launch(TFGNN_KFAM_SP_TN);
    if (wide) {
      if (bn == 320) SP_LAUNCH_TN(5, true, true);
      else if (bn == 256) SP_LAUNCH_TN(4, true, true);
      else SP_LAUNCH_TN(2, true, true);
    } else if (fik) {
      if (bn == 320) SP_LAUNCH_TN(5, true);
      else if (bn == 256) SP_LAUNCH_TN(4, true);
      else SP_LAUNCH_TN(2, true);
    } else {
      if (bn == 320) SP_LAUNCH_TN(5, false);
      else if (bn == 256) SP_LAUNCH_TN(4, false);
      else SP_LAUNCH_TN(2, false);
    }
    TFGNN_LAUNCH_CHECK();
  }
#undef SP_LAUNCH_TN
  if (!(phases & 12)) return TFGNN_OK;
  const int64_t total = M * N;
  AuxTnReduce ra{};
  ra.partial = g.partial; ra.splits = splits_used; ra.M = M; ra.N = N; ra.ref = fik ? ref_split : ref; ra.a_col0 = a_first_col; ra.a_sb = a_scale_block;
  ra.ref_ld = fik ? (int)nblk : 0;
  ra.C = d_C; ra.group_rows = group_rows; ra.stride_group = stride_group; ra.stride_row = stride_row; ra.stride_col = stride_col;
  ra.accumulate = accumulate; ra.slab = Mp * N;
  const unsigned rblocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 2048);
  if (phases & 8) {  // the reduction as a job of a later merged launch (tfgnn_aux_launch) instead of a launch of its own
    TFGNN_REQUIRE(reduce_job != nullptr, "tfgnn_sp_gemm_tn: reduce_job is NULL");
    aux_job_set(reduce_job, AUX_TN_REDUCE, rblocks, ra);
    return TFGNN_OK;
  }
  hipLaunchKernelGGL(sp_tn_reduce_kernel, dim3(rblocks), dim3(256), 0, s, ra);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int tfgnn_sp_gemm_tn(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                     const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                     int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                     int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                     size_t workspace_bytes, void* stream) {
  return sp_gemm_tn_impl(7, M, N, K, d_A_sp, lda_bytes, a_first_col, d_a_inv_scale, a_total_cols, a_scale_block, d_B_sp, ldb_bytes,
                         b_first_col, d_b_inv_scale, d_C, group_rows, stride_group, stride_row, stride_col, accumulate, d_workspace,
                         workspace_bytes, stream);
}

int tfgnn_sp_gemm_tn_wide(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                          const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                          int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                          int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                          size_t workspace_bytes, tfgnn_aux_job* reduce_job, void* stream) {
  return sp_gemm_tn_impl(reduce_job ? (1 | 2 | 8) : 7, M, N, K, d_A_sp, lda_bytes, a_first_col, d_a_inv_scale, a_total_cols, a_scale_block,
                         d_B_sp, ldb_bytes, b_first_col, d_b_inv_scale, d_C, group_rows, stride_group, stride_row, stride_col, accumulate,
                         d_workspace, workspace_bytes, stream, reduce_job, nullptr, true);
}

int tfgnn_sp_gemm_tn_grouped(int64_t M, int64_t N, const void* d_A_sp, int64_t lda_bytes, const float* d_a_inv_scale,
                             int64_t a_total_cols, int a_scale_block, const void* d_B_sp, int64_t ldb_bytes, const float* d_b_inv_scale,
                             int num_groups, const int32_t* d_group_split_offsets, int num_splits, const int32_t* d_split_table,
                             float* d_C, int64_t c_group_stride, int64_t stride_row, int64_t stride_col, void* d_workspace,
                             size_t workspace_bytes, void* stream) {
  TFGNN_REQUIRE(d_A_sp && d_B_sp && d_C && d_a_inv_scale && d_b_inv_scale && d_group_split_offsets && d_split_table,
                "tfgnn_sp_gemm_tn_grouped: null pointer");
  TFGNN_REQUIRE(M > 0 && N > 0 && num_groups >= 1 && num_splits >= 0 && num_splits <= 512, "tfgnn_sp_gemm_tn_grouped: 1.. groups, at most 512 K ranges");
  const int bn = sp_tile_width(N);
  if (!bn || M % 16 || a_scale_block < 32) {
    set_error("tfgnn_sp_gemm_tn_grouped: M = %lld must be a multiple of 16, N = %lld of 128, scale blocks of A >= 32 columns",
              (long long)M, (long long)N);
    return TFGNN_ERR_UNSUPPORTED;
  }
  const int64_t Mp = ceil_div(M, SP_BM) * SP_BM;
  TFGNN_REQUIRE(lda_bytes % 64 == 0 && ldb_bytes % 64 == 0 && (uintptr_t)d_A_sp % 64 == 0 && (uintptr_t)d_B_sp % 64 == 0 &&
                    lda_bytes >= M * 4 && ldb_bytes >= N * 4 && a_total_cols >= M && a_total_cols % a_scale_block == 0,
                "tfgnn_sp_gemm_tn_grouped: SP16 operands must be 64-byte aligned with rows wide enough");
  const int64_t nblk = a_total_cols / a_scale_block;
  const size_t r_bytes = ((size_t)nblk * 512 * 4 + 255) & ~(size_t)255;
  const size_t need = r_bytes + (size_t)num_splits * (size_t)Mp * (size_t)N * 4;
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= need && (uintptr_t)d_workspace % 256 == 0,
                "tfgnn_sp_gemm_tn_grouped: workspace too small or unaligned (need %zu bytes)", need);
  hipStream_t s = (hipStream_t)stream;
  float* ref_split = (float*)d_workspace;  // [num_splits][nblk]
  SpTnArgs g{};
  g.M = Mp; g.N = N; g.K = 0;
  g.A = (const uint8_t*)d_A_sp; g.lda = lda_bytes;
  g.B = (const uint8_t*)d_B_sp; g.ldb = ldb_bytes;
  g.a_sb = a_scale_block; g.a_col0 = 0; g.a_nblk = (int)nblk;
  g.inv_a = d_a_inv_scale; g.inv_b = d_b_inv_scale; g.ref_split = ref_split; g.spread_flag = sp_spread_flag_device();
  g.partial = (float*)((uint8_t*)d_workspace + r_bytes);
  g.k_chunk = SP_TN_BSC_MAX_CHUNK;
  g.split_table = d_split_table;
  g.n_tiles = (unsigned)(N / bn);
  g.tiles = (unsigned)((Mp / SP_BM) * g.n_tiles);
  g.splits = (unsigned)num_splits;
  g.per_xcd = (g.tiles * g.splits + 7) / 8;
  if (num_splits > 0) {
    dim3 grid(8 * g.per_xcd);
    count_launch(TFGNN_KFAM_SP_TN);
#define SP_LAUNCH_TNG(T)                                                                                             \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    using TnGeo = SpGeoTN<T, true, true>;                                                                            \
    constexpr int tn_lds = TnGeo::LDS_BYTES;                                                                         \
    if (!attr_set) {                                                                                                 \
      (void)hipFuncSetAttribute((const void*)gemm_sp_tn_kernel<T, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, tn_lds); \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    hipLaunchKernelGGL((gemm_sp_tn_kernel<T, true, true>), grid, dim3(SP_NT), tn_lds, s, g);                         \
  } while (0)
    if (bn == 320) SP_LAUNCH_TNG(5);
    else if (bn == 256) SP_LAUNCH_TNG(4);
    else SP_LAUNCH_TNG(2);
#undef SP_LAUNCH_TNG
    TFGNN_LAUNCH_CHECK();
  }
  AuxTnReduce ra{};
  ra.partial = g.partial; ra.splits = 0; ra.M = M; ra.N = N; ra.ref = ref_split; ra.a_col0 = 0; ra.a_sb = a_scale_block;
  ra.ref_ld = (int)nblk;
  ra.C = d_C; ra.group_rows = M; ra.stride_group = 0; ra.stride_row = stride_row; ra.stride_col = stride_col;
  ra.accumulate = 0; ra.slab = Mp * N;
  const unsigned rblocks = (unsigned)std::min<int64_t>(ceil_div(M * N, 256), 512);
  hipLaunchKernelGGL(sp_tn_reduce_grouped_kernel, dim3(rblocks, (unsigned)num_groups), dim3(256), 0, s, ra, d_group_split_offsets,
                     c_group_stride);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int tfgnn_sp_gemm_tn_phase(int phases, int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                           const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                           int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                           int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                           size_t workspace_bytes, void* stream) {
  TFGNN_REQUIRE(phases >= 1 && phases <= 7, "tfgnn_sp_gemm_tn_phase: phases is a mask of 1 (factors), 2 (product), 4 (reduce)");
  return sp_gemm_tn_impl(phases, M, N, K, d_A_sp, lda_bytes, a_first_col, d_a_inv_scale, a_total_cols, a_scale_block, d_B_sp, ldb_bytes,
                         b_first_col, d_b_inv_scale, d_C, group_rows, stride_group, stride_row, stride_col, accumulate, d_workspace,
                         workspace_bytes, stream);
}

/* factors (if asked for) + product now, the split reduction as a job for tfgnn_aux_launch */
int tfgnn_sp_gemm_tn_deferred(int with_factors, int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                              const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                              int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                              int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                              size_t workspace_bytes, tfgnn_aux_job* reduce_job, void* stream) {
  TFGNN_REQUIRE(reduce_job != nullptr, "tfgnn_sp_gemm_tn_deferred: reduce_job is NULL");
  return sp_gemm_tn_impl((with_factors ? 1 : 0) | 2 | 8, M, N, K, d_A_sp, lda_bytes, a_first_col, d_a_inv_scale, a_total_cols,
                         a_scale_block, d_B_sp, ldb_bytes, b_first_col, d_b_inv_scale, d_C, group_rows, stride_group, stride_row,
                         stride_col, accumulate, d_workspace, workspace_bytes, stream, reduce_job);
}

/* nothing is launched: the factor pass and the split reduction of tfgnn_sp_gemm_tn as jobs; the product itself is
 * tfgnn_sp_gemm_tn_phase(2, ...) with the same arguments, after the factors job and before the reduce job have run */
int tfgnn_sp_gemm_tn_jobs(int64_t M, int64_t N, int64_t K, const void* d_A_sp, int64_t lda_bytes, int64_t a_first_col,
                          const float* d_a_inv_scale, int64_t a_total_cols, int a_scale_block, const void* d_B_sp,
                          int64_t ldb_bytes, int64_t b_first_col, const float* d_b_inv_scale, float* d_C, int64_t group_rows,
                          int64_t stride_group, int64_t stride_row, int64_t stride_col, int accumulate, void* d_workspace,
                          size_t workspace_bytes, tfgnn_aux_job* factors_job, tfgnn_aux_job* reduce_job) {
  TFGNN_REQUIRE(reduce_job != nullptr && factors_job != nullptr, "tfgnn_sp_gemm_tn_jobs: NULL job");
  if (K > 131072) {
    set_error("tfgnn_sp_gemm_tn_jobs: operands of more than 131072 rows take the two-stage factor pass (tfgnn_sp_gemm_tn_deferred)");
    return TFGNN_ERR_UNSUPPORTED;
  }
  return sp_gemm_tn_impl(16 | 8, M, N, K, d_A_sp, lda_bytes, a_first_col, d_a_inv_scale, a_total_cols, a_scale_block, d_B_sp,
                         ldb_bytes, b_first_col, d_b_inv_scale, d_C, group_rows, stride_group, stride_row, stride_col, accumulate,
                         d_workspace, workspace_bytes, nullptr, reduce_job, factors_job);
}

int tfgnn_sp_split_rows_job(const float* d_src, int64_t ld, int64_t seg_len, int64_t seg_stride, int64_t rows, int64_t cols,
                            int scale_block, void* d_sp, int64_t ld_sp_bytes, float* d_inv_scale,
                            const float* d_fixed_inv_scale, tfgnn_aux_job* job) {
  TFGNN_REQUIRE(d_src && d_sp && job, "tfgnn_sp_split_rows_job: null pointer");
  TFGNN_REQUIRE(rows > 0 && cols > 0 && cols % 16 == 0, "tfgnn_sp_split_rows_job: cols must be a positive multiple of 16");
  if (scale_block <= 0) scale_block = (int)cols;
  if (seg_len <= 0) { seg_len = cols; seg_stride = 0; }
  TFGNN_REQUIRE(scale_block % 16 == 0 && cols % scale_block == 0, "tfgnn_sp_split_rows_job: scale_block must divide cols and be a multiple of 16");
  TFGNN_REQUIRE(seg_len % 4 == 0 && cols % seg_len == 0 && ld % 4 == 0 && seg_stride % 4 == 0 && (uintptr_t)d_src % 16 == 0,
                "tfgnn_sp_split_rows_job: source must be 16-byte aligned with segment length / strides multiples of 4");
  TFGNN_REQUIRE(ld_sp_bytes >= cols * 4 && ld_sp_bytes % 64 == 0 && (uintptr_t)d_sp % 64 == 0, "tfgnn_sp_split_rows_job: bad SP16 leading dimension / alignment");
  AuxSplitRows a{d_src, ld, seg_len, seg_stride, rows, cols, scale_block, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale, d_fixed_inv_scale};
  const int64_t items = rows * (cols / scale_block);
  TFGNN_REQUIRE(ceil_div(items, 4) < (1ll << 31), "tfgnn_sp_split_rows_job: too many rows");
  aux_job_set(job, AUX_SPLIT_ROWS, (unsigned)ceil_div(items, 4), a);
  return TFGNN_OK;
}

int tfgnn_sp_split_cols_job(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                            float* d_inv_scale, tfgnn_aux_job* job) {
  TFGNN_REQUIRE(d_src && d_sp && job, "tfgnn_sp_split_cols_job: null pointer");
  TFGNN_REQUIRE(K > 0 && N > 0 && K % 16 == 0 && N % 4 == 0 && ld % 4 == 0 && (uintptr_t)d_src % 16 == 0,
                "tfgnn_sp_split_cols_job: K must be a multiple of 16, N and ld multiples of 4");
  TFGNN_REQUIRE(ld_sp_bytes >= K * 4 && ld_sp_bytes % 64 == 0 && (uintptr_t)d_sp % 64 == 0, "tfgnn_sp_split_cols_job: bad SP16 leading dimension / alignment");
  AuxSplitCols a{d_src, ld, K, N, (uint8_t*)d_sp, ld_sp_bytes, d_inv_scale, (unsigned)ceil_div(N, 16),
                 (unsigned)std::max<int64_t>(1, std::min<int64_t>(8, K / 128)), nullptr, 0};
  aux_job_set(job, AUX_SPLIT_COLS, a.ncx * a.ncy, a);
  return TFGNN_OK;
}

size_t tfgnn_sp_split_cols_two_pass_bytes(int64_t K, int64_t N) {
  // worth it where the one-pass conversion cannot keep a column strip in registers (K > 1280) and re-reads it per K slice
  if (K <= 1280 || N <= 0 || N % 4) return 0;
  return (size_t)sp_colmax_parts(K) * (size_t)N * 4;
}

int tfgnn_sp_split_cols_jobs(const float* d_src, int64_t ld, int64_t K, int64_t N, void* d_sp, int64_t ld_sp_bytes,
                             float* d_inv_scale, float* d_colmax_workspace, size_t workspace_bytes, tfgnn_aux_job* maxima_job,
                             tfgnn_aux_job* split_job) {
  TFGNN_REQUIRE(maxima_job && split_job, "tfgnn_sp_split_cols_jobs: null job");
  int rc = tfgnn_sp_split_cols_job(d_src, ld, K, N, d_sp, ld_sp_bytes, d_inv_scale, split_job);
  if (rc) return rc;
  maxima_job->kind = AUX_NONE;
  maxima_job->num_blocks = 0;
  const size_t need = tfgnn_sp_split_cols_two_pass_bytes(K, N);
  if (!need) return TFGNN_OK;  // the one-pass job
  TFGNN_REQUIRE(d_colmax_workspace && workspace_bytes >= need && (uintptr_t)d_colmax_workspace % 16 == 0,
                "tfgnn_sp_split_cols_jobs: the maxima need %zu bytes of 16-byte aligned workspace", need);
  const int nparts = sp_colmax_parts(K);
  AuxColAbsmax m{d_src, ld, K, N, d_colmax_workspace, nparts, (unsigned)ceil_div(N, 1024)};
  aux_job_set(maxima_job, AUX_COL_ABSMAX, (unsigned)nparts * m.nchunks, m);
  AuxSplitCols a{};
  memcpy(&a, split_job->payload, sizeof(a));
  a.colmax_parts = d_colmax_workspace;
  a.nparts = nparts;
  aux_job_set(split_job, AUX_SPLIT_COLS, a.ncx * a.ncy, a);
  return TFGNN_OK;
}

int tfgnn_absmax(const float* d_x, int64_t n, float scale, float* d_out, void* stream) {
  TFGNN_REQUIRE(d_x && d_out && n >= 0 && scale > 0.f && (uintptr_t)d_x % 16 == 0, "tfgnn_absmax: bad arguments");
  if (n == 0) return TFGNN_OK;
  hipLaunchKernelGGL(sp_absmax_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 1024), 512)), dim3(256), 0, (hipStream_t)stream,
                     d_x, n, scale, d_out);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int tfgnn_sp_inv_scale_from_bound(const float* d_bound, float* d_inv_scale, void* stream) {
  TFGNN_REQUIRE(d_bound && d_inv_scale, "tfgnn_sp_inv_scale_from_bound: null pointer");
  hipLaunchKernelGGL(sp_inv_scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, d_bound, d_inv_scale);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

}  // extern "C"
