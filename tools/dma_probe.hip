// Semantics probe for global_load_lds_dwordx4 on gfx950: where do a wave's 64 x 16 bytes land in LDS?
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/dma_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__global__ void k(const uint4v* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 256];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint4v* g = src + wave * 64 + (lane ^ 1);  // lane i fetches element (i ^ 1) of its wave's 64
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, (const uint4v*)d, o);
  std::vector<unsigned> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 4; ++c) {
        const unsigned expect = (w * 64 + (l ^ 1)) * 4 + c;  // LDS slot l of wave w holds what lane l fetched
        if (r[w * 256 + l * 4 + c] != expect) ++bad;
      }
  printf("global_load_lds_dwordx4: LDS[base + lane*16] <- lane's 16 bytes: %s (%d mismatches); first words %u %u %u %u %u\n",
         bad ? "NO" : "yes", bad, r[0], r[1], r[4], r[5], r[256]);
  return 0;
}
