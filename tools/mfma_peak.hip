// Probe: sustained v_mfma_f32_32x32x2_f32 rate and shader clock on this box.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(512) mfma_loop(float* out, int iters, long long* clocks) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clocks[0] = c1 - c0;
    clocks[1] = w1 - w0;
  }
}

int main() {
  float* out;
  long long* clocks;
  hipMalloc(&out, 1024 * 512 * 4);
  hipMalloc(&clocks, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves : {4, 8}) {
    for (int blocks : {235, 256, 512}) {
      const int iters = 8000;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<5>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, clocks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long h[2];
      hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost);
      double flops = 2.0 * 32 * 32 * 2 * 5.0 * iters * waves * blocks;
      printf("waves/block=%d blocks=%d: %.1f us  %.1f TFLOP/s  shader clock %.0f MHz (clock64/wall_clock64 @100MHz)\n", waves,
             blocks, ms * 1e3, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
    }
  }
  return 0;
}
