// Probe (not product code): how fast can the PRODUCER side of a gather-fused RGCN product run?
//
// One workgroup per 128-node tile.  For every (edge type l, 64-column chunk c) the four waves sum the source rows of
// the tile's 128 buckets (node, l) - 16 lanes per bucket, one float4 per lane and edge, UNR edges in flight - scale by
// 1 / (count + 1e-7), split the 64 sums into fp16 pairs with a per-(row, chunk) power-of-two scale and store them in an
// LDS chunk buffer in the SP16 granule layout; one workgroup barrier per chunk stands in for the hand-over to the
// multiplying waves.  Buckets longer than `long_threshold` read ONE pre-aggregated row instead (the long-row plan of
// the existing gather would produce that side table).  Nothing is multiplied: the kernel time is the floor the gather
// imposes on a fused kernel; compare with ~45-60 us of MFMA time per tile.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sp16.hpp"

using namespace tfgnn;

struct ProbeArgs {
  const int32_t* tile_nodes;  // [tiles][128], -1 = padding
  const int32_t* rowptr;      // [V * L + 1], bucket = node * L + type
  const int32_t* col;         // [E] source node
  const float* invdeg;        // [V * L]
  const float* X;             // [V][ldx]
  int64_t ldx;
  const float* side;          // [V * L][ldx] pre-aggregated (already scaled) rows; read for long buckets
  int L, D;                   // D = columns of X (multiple of 64)
  int long_threshold;
  int mode;                   // 0 full, 1 no loads of X (index walk only), 2 no split / LDS stores
  float* out;                 // [tiles] checksum
};

template <int UNR>
__global__ void __launch_bounds__(256, 1) gather_chunks_probe(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];  // [NBUF][128 rows][256 B] + scales
  constexpr int NBUF = 3;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int grp = lane >> 4, gl = lane & 15;
  const int32_t* nodes = a.tile_nodes + (int64_t)blockIdx.x * 128;
  float check = 0.f;
  const int chunks = a.D / 64;
  int it = 0;
  for (int l = 0; l < a.L; ++l) {
    for (int c = 0; c < chunks; ++c, ++it) {
      uint8_t* buf = lds + (it % NBUF) * (128 * 256);
      float* scales = reinterpret_cast<float*>(lds + NBUF * 128 * 256) + (it % NBUF) * 128;
#pragma unroll 1
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 32 + rr * 4 + grp;
        const int v = nodes[r];
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        if (v >= 0) {
          const int64_t bucket = (int64_t)v * a.L + l;
          const int32_t beg = a.rowptr[bucket], end = a.rowptr[bucket + 1];
          const int f0 = c * 64 + gl * 4;
          if (end - beg > a.long_threshold) {
            acc = *reinterpret_cast<const float4*>(a.side + bucket * a.ldx + f0);
          } else {
            for (int32_t e = beg; e < end; e += UNR) {
              int32_t idx[UNR];
              bool ok[UNR];
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                ok[u] = e + u < end;
                idx[u] = a.col[ok[u] ? e + u : end - 1];
              }
              float4 x[UNR];
#pragma unroll
              for (int u = 0; u < UNR; ++u) {
                if (a.mode == 1) x[u] = float4{(float)idx[u], 0.f, 0.f, 0.f};
                else x[u] = *reinterpret_cast<const float4*>(a.X + (int64_t)idx[u] * a.ldx + f0);
              }
#pragma unroll
              for (int u = 0; u < UNR; ++u)
                if (ok[u]) { acc.x += x[u].x; acc.y += x[u].y; acc.z += x[u].z; acc.w += x[u].w; }
            }
            const float s = a.invdeg[bucket];
            acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
          }
        }
        if (a.mode == 2) {
          check += acc.x + acc.y + acc.z + acc.w;
        } else {
          float mx = fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w)));
#pragma unroll
          for (int o = 8; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
          float iv;
          const float sc = sp_scale_for_max(mx, &iv);
          sp_store4(buf + r * 256, gl * 4, acc, sc);
          if (gl == 0) scales[r] = iv;
        }
      }
      __syncthreads();
      if (a.mode != 2) check += reinterpret_cast<const float*>(buf)[tid] + scales[tid & 127];
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) check += __shfl_xor(check, o, 64);
  if (lane == 0) atomicAdd(a.out + blockIdx.x, check);
}

extern "C" int fused_probe_launch(const int32_t* tile_nodes, int tiles, const int32_t* rowptr, const int32_t* col, const float* invdeg,
                                  const float* X, int64_t ldx, const float* side, int L, int D, int long_threshold, int mode, int unr,
                                  float* out, void* stream) {
  ProbeArgs a{tile_nodes, rowptr, col, invdeg, X, ldx, side, L, D, long_threshold, mode, out};
  const size_t lds_bytes = 3 * 128 * 256 + 3 * 128 * 4;
  hipStream_t s = (hipStream_t)stream;
  if (unr == 8) {
    (void)hipFuncSetAttribute((const void*)gather_chunks_probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(gather_chunks_probe<8>, dim3(tiles), dim3(256), lds_bytes, s, a);
  } else if (unr == 4) {
    (void)hipFuncSetAttribute((const void*)gather_chunks_probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(gather_chunks_probe<4>, dim3(tiles), dim3(256), lds_bytes, s, a);
  } else {
    (void)hipFuncSetAttribute((const void*)gather_chunks_probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(gather_chunks_probe<2>, dim3(tiles), dim3(256), lds_bytes, s, a);
  }
  return (int)hipGetLastError();
}
