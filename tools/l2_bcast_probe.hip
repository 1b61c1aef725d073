// Probe: how fast can 235 workgroups each stream the same 1.6 MB weight matrix out of L2, in the access pattern of the NT
// split GEMM's staging waves (per K tile: 320 rows x 64 bytes, rows 5 KB apart), all in the same K order or each
// starting at its own K tile?  Also with the tile contiguous (20 KB per K tile).
//   hipcc --offload-arch=gfx950 -O3 tools/l2_bcast_probe.hip -o tools/l2_bcast_probe && tools/l2_bcast_probe
#include <hip/hip_runtime.h>
#include <cstdio>

// 256 threads: thread -> (row = id >> 2, kq = id & 3), 5 rows-slots per thread (320 rows), float4 each
template <int MODE>  // 0 strided same order, 1 strided rotated, 2 contiguous tile same order, 3 contiguous rotated
__global__ void __launch_bounds__(256) stream_b(const float* __restrict__ B, int K, int T, float* out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int tid = threadIdx.x;
  const int start = (MODE & 1) ? (int)((blockIdx.x * 7) % T) : 0;
  for (int tt = 0; tt < T; ++tt) {
    int t = tt + start;
    if (t >= T) t -= T;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = tid + 256 * i;
      const float* p = (MODE & 2) ? B + (size_t)t * (320 * 16) + id * 4 : B + (size_t)(id >> 2) * K + t * 16 + (id & 3) * 4;
      const float4 v = *reinterpret_cast<const float4*>(p);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x == 12345.f) out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static void run(const float* B, float* out, int K, const char* name) {
  const int T = K / 16;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_b<MODE>, dim3(235), dim3(256), 0, 0, B, K, T, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double us = ms * 100.0, bytes = 235.0 * 320 * K * 4;
  printf("%-34s %7.1f us per pass over B by 235 workgroups  (%.2f TB/s out of L2, %.2f us per K tile)\n", name, us, bytes / us / 1e6,
         us / T);
}

int main() {
  const int K = 1280;
  float *B, *out;
  (void)hipMalloc(&B, 320 * K * 4);
  (void)hipMemset(B, 0, 320 * K * 4);
  (void)hipMalloc(&out, 235 * 256 * 4);
  run<0>(B, out, K, "strided rows, same K order");
  run<1>(B, out, K, "strided rows, rotated K order");
  run<2>(B, out, K, "contiguous K tiles, same order");
  run<3>(B, out, K, "contiguous K tiles, rotated");
  return 0;
}
