// Probe: cycles per v_mfma_f32_32x32x16_bf16 as a function of the distance between two MFMAs on the same accumulator
// (1 = back to back, 2 = the split GEMM's alternating pair, 4, 10), one wave per SIMD and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_dep_probe.hip -o tools/mfma_dep_probe && tools/mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(512) mfma_loop(float* out, int iters, long long* clocks, int random_bits) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int r = 0; r < 8; ++r) {
    if (random_bits) {  // operands with random significands (the toggle rate of real data)
      unsigned h = (threadIdx.x * 8 + r) * 2654435761u + blockIdx.x * 40503u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      a[r] = (__bf16)(((h & 0xffff) / 32768.0f) - 1.0f);
      b[r] = (__bf16)(((h >> 16) / 32768.0f) - 1.0f);
    } else {
      a[r] = (__bf16)(threadIdx.x * 1e-3f + r);
      b[r] = (__bf16)(blockIdx.x * 1e-3f + 1.f);
    }
  }
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 20 / NACC; ++rep)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clocks[0] = c1 - c0;
}

template <int NACC>
static void run(float* out, long long* clocks, int waves, int random_bits) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(256), dim3(64 * waves), 0, 0, out, iters, clocks, random_bits);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  long long h;
  (void)hipMemcpy(&h, clocks, 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * (20 / NACC) * NACC;
  const double per_wave = (double)h / n_mfma;
  const double flops = n_mfma * 32768.0 * waves * 256;
  printf("%s operands  waves/block=%d  distance %2d: %.1f clock64 cycles per MFMA of one wave; %.2f ms -> %.0f TFLOP/s, %.2f ns per MFMA of one wave\n",
         random_bits ? "random" : "smooth", waves, NACC, per_wave, ms, flops / ms / 1e9, ms * 1e6 / n_mfma);
}

int main() {
  float* out;
  long long* clocks;
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clocks, 16);
  for (int random_bits : {0, 1})
    for (int waves : {4, 8}) {
      run<1>(out, clocks, waves, random_bits);
      run<2>(out, clocks, waves, random_bits);
      run<5>(out, clocks, waves, random_bits);
      run<10>(out, clocks, waves, random_bits);
    }
  return 0;
}
