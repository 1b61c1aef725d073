// How accurately do the f16 / bf16 / f32 MFMAs ACCUMULATE?  One wave per 32 x 32 output tile, K = 1280,
// A ~ N(0,1), B ~ 0.05 N(0,1); every variant is compared with the exact (fp64) value of the sum it is asked for, so
// that only the accumulation error is left, and with the fp64 product of the fp32 operands (total error).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_acc_probe.hip -o tools/_probe/mfma_acc_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int K = 1280, TILES = 256;

__device__ inline void split16(float x, _Float16& h, _Float16& l) { h = (_Float16)x; l = (_Float16)(x - (float)h); }
__device__ inline __bf16 tobf(float x) { unsigned u = __float_as_uint(x) & 0xffff0000u; return __builtin_bit_cast(__bf16, (unsigned short)(u >> 16)); }
__device__ inline float bff(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }

// variant: 0 = f16 hh only (operands rounded to fp16)      1 = bf16 hh only (operands truncated to bf16)
//          2 = f32 MFMA on fp16-rounded operands           3 = f16x2 three products, one accumulator (lh, hl, hh)
//          4 = f16x2, cross terms in a second accumulator  5 = f16x2, accumulator flushed into a master every 20 steps
//          6 = f16x2 order hh, hl, lh                       7 = bf16x3 six products one accumulator
//          8 = f32 MFMA on the fp32 operands
template <int V>
__global__ void __launch_bounds__(64) probe(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
  const float* a = A + ((size_t)blockIdx.x * 32 + i) * K;
  const float* b = B + ((size_t)blockIdx.x * 32 + i) * K;
  floatx16 acc = {}, acc2 = {}, master = {};
  for (int s = 0; s < K / 16; ++s) {
    float av[8], bv[8];
    for (int j = 0; j < 8; ++j) { av[j] = a[s * 16 + kg * 8 + j]; bv[j] = b[s * 16 + kg * 8 + j]; }
    if (V == 2 || V == 8) {
      for (int kk = 0; kk < 8; ++kk) {
        // 32x32x2 f32: lane (i, k = lane>>5): two k per instruction -> 8 instructions cover this lane pair's 16 k
        // lane kg=0 supplies k = 2 kk', kg=1 supplies k = 2kk'+1 : re-read so that the k order is 0..15
        const int k = s * 16 + kk * 2 + kg;
        float x = a[k], y = b[k];
        if (V == 2) { x = (float)(_Float16)x; y = (float)(_Float16)y; }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
      }
      continue;
    }
    if (V == 1 || V == 7) {
      bf8 ah, am, al, bh, bm, bl;
      for (int j = 0; j < 8; ++j) {
        float x = av[j], h = bff(x), r1 = x - h, m = bff(r1), l = r1 - m;
        ah[j] = tobf(h); am[j] = tobf(m); al[j] = tobf(l);
        x = bv[j]; h = bff(x); r1 = x - h; m = bff(r1); l = r1 - m;
        bh[j] = tobf(h); bm[j] = tobf(m); bl[j] = tobf(l);
      }
      if (V == 7) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      continue;
    }
    half8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) { _Float16 h, l; split16(av[j], h, l); ah[j] = h; al[j] = l; split16(bv[j], h, l); bh[j] = h; bl[j] = l; }
    if (V == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    if (V == 3 || V == 5) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      if (V == 5 && s % 20 == 19) { for (int r = 0; r < 16; ++r) { master[r] += acc[r]; acc[r] = 0.f; } }
    }
    if (V == 4) {
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    if (V == 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
    float v = acc[r];
    if (V == 4) v += acc2[r];
    if (V == 5) v += master[r];
    C[((size_t)blockIdx.x * 32 + row) * 32 + i] = v;
  }
}

int main() {
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A((size_t)TILES * 32 * K), B(A.size());
  for (auto& x : A) x = nd(rng);
  for (auto& x : B) x = 0.05f * nd(rng);
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)TILES * 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> C((size_t)TILES * 1024);
  auto bf = [](float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u); };
  const char* names[9] = {"f16 hh only", "bf16 hh only", "f32 MFMA on fp16-rounded", "f16x2 lh,hl,hh one acc", "f16x2 cross terms in 2nd acc",
                          "f16x2 flush every 20 steps", "f16x2 hh,hl,lh one acc", "bf16x3 six products", "f32 MFMA on fp32 operands"};
  for (int v = 0; v < 9; ++v) {
    switch (v) {
      case 0: probe<0><<<TILES, 64>>>(dA, dB, dC); break; case 1: probe<1><<<TILES, 64>>>(dA, dB, dC); break;
      case 2: probe<2><<<TILES, 64>>>(dA, dB, dC); break; case 3: probe<3><<<TILES, 64>>>(dA, dB, dC); break;
      case 4: probe<4><<<TILES, 64>>>(dA, dB, dC); break; case 5: probe<5><<<TILES, 64>>>(dA, dB, dC); break;
      case 6: probe<6><<<TILES, 64>>>(dA, dB, dC); break; case 7: probe<7><<<TILES, 64>>>(dA, dB, dC); break;
      case 8: probe<8><<<TILES, 64>>>(dA, dB, dC); break;
    }
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double max_acc = 0, max_tot = 0, sum_acc = 0, sum_signed = 0;
    for (int t = 0; t < TILES; ++t)
      for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c) {
          const float* a = &A[((size_t)t * 32 + r) * K];
          const float* b = &B[((size_t)t * 32 + c) * K];
          double asked = 0, full = 0;
          for (int k = 0; k < K; ++k) {
            full += (double)a[k] * (double)b[k];
            if (v == 0 || v == 2) asked += (double)(float)(_Float16)a[k] * (double)(float)(_Float16)b[k];
            else if (v == 1) asked += (double)bf(a[k]) * (double)bf(b[k]);
            else if (v == 8) asked += (double)a[k] * (double)b[k];
            else if (v == 7) {
              float ah = bf(a[k]), am = bf(a[k] - ah), al = a[k] - ah - am, bh = bf(b[k]), bm = bf(b[k] - bh), bl = b[k] - bh - bm;
              asked += (double)al * bh + (double)ah * bl + (double)am * bm + (double)am * bh + (double)ah * bm + (double)ah * bh;
            } else {
              float ah = (float)(_Float16)a[k], al = (float)(_Float16)(a[k] - ah), bh = (float)(_Float16)b[k], bl = (float)(_Float16)(b[k] - bh);
              asked += (double)al * bh + (double)ah * bl + (double)ah * bh;
            }
          }
          const double got = C[((size_t)t * 32 + r) * 32 + c];
          max_acc = fmax(max_acc, fabs(got - asked)); max_tot = fmax(max_tot, fabs(got - full));
          sum_acc += (got - asked) * (got - asked); sum_signed += got - asked;
        }
    const double n = (double)TILES * 1024;
    printf("%-32s accumulation error: max %.3e rms %.3e mean %+.3e | vs fp64 of the fp32 operands: max %.3e\n", names[v], max_acc,
           sqrt(sum_acc / n), sum_signed / n, max_tot);
  }
  return 0;
}
