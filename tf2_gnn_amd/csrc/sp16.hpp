// SP16 operand format helpers shared by the product kernels (gemm_sp.hip) and the kernels that PRODUCE split operands
// (the gather of spmm.hip writes its sums this way).  Format and numerics: see the head of gemm_sp.hip / tfgnn.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace tfgnn {

// power of two s with mx * s in [2^14, 2^15) and *inv = 1 / s
__device__ __forceinline__ float sp_scale_for_max(float mx, float* inv) {
  const unsigned b = __float_as_uint(mx);
  int ex = (int)((b >> 23) & 0xffu);
  if (ex == 255) {  // inf / nan in the block: no scaling, the result is inf / nan anyway
    *inv = 1.f;
    return 1.f;
  }
  if (mx == 0.f) {  // all-zero block: any scale represents it; the smallest one, so that among the blocks of a row (the
    *inv = 1.1754943508222875e-38f;  // kernel normalises them to the LARGEST 2^-e) it never is the reference
    return 1.f;
  }
  if (ex == 0) ex = 1;  // subnormal maximum: scale as the smallest normal exponent
  int e = 14 - (ex - 127);
  e = e > 126 ? 126 : e;
  *inv = __uint_as_float((unsigned)(127 - e) << 23);
  return __uint_as_float((unsigned)(127 + e) << 23);
}

// the spread guard of the weight-gradient product: is a row's factor f = inv_a * inv_b / (largest such product) more than 2^20
// below the reference although the row holds something?  All-zero rows are recognised by THEIR OWN scale - the marker
// 2^-126 sp_scale_for_max gives them (also rows whose largest entry is below 2^-112: nothing) - not by the size of f: with
// small operands (a loss gradient of 1e-9 against unit activations) the marker divided by the reference is no longer "tiny"
// and every empty bucket used to trip the guard (round 4: the ppi workload lost the split-operand path on its first step).
__device__ __forceinline__ bool sp_row_too_small(float f, float inv_a, float inv_b) {
  return f < 9.5367431640625e-07f && inv_a > 1.2e-38f && inv_b > 1.2e-38f;
}

__device__ __forceinline__ void sp_split(float xs, _Float16& h, _Float16& l) {
  h = (_Float16)xs;  // v_cvt_f16_f32: round to nearest even
  const float r = xs - (float)h;
  // inf / nan: keep the class in h, nothing in l (inf - inf would make l a NaN)
  l = (__float_as_uint(xs) & 0x7f800000u) == 0x7f800000u ? (_Float16)0.f : (_Float16)r;
}

// columns c .. c+3 (c % 4 == 0) of an SP16 row
__device__ __forceinline__ void sp_store4(uint8_t* row, int64_t c, float4 v, float s) {
  _Float16 h[4], l[4];
  sp_split(v.x * s, h[0], l[0]);
  sp_split(v.y * s, h[1], l[1]);
  sp_split(v.z * s, h[2], l[2]);
  sp_split(v.w * s, h[3], l[3]);
  uint8_t* g = row + (c >> 4) * 64 + (c & 15) * 2;
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<half4*>(g) = half4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<half4*>(g + 32) = half4{l[0], l[1], l[2], l[3]};
}

}  // namespace tfgnn
