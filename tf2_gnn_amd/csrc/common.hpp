// Shared helpers for the tfgnn HIP library (gfx950 only).
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tfgnn.h"

namespace tfgnn {

void set_error(const char* fmt, ...);
// host-side launch counters per product-kernel family (tfgnn_launch_counts: the parity tests assert that a GEMM mode
// really ran the kernels it names)
void count_launch(int family);

#define TFGNN_HIP_CHECK(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ::tfgnn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                         __LINE__);                                                        \
      return TFGNN_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

#define TFGNN_REQUIRE(cond, ...)              \
  do {                                        \
    if (!(cond)) {                            \
      ::tfgnn::set_error(__VA_ARGS__);        \
      return TFGNN_ERR_INVALID_ARGUMENT;      \
    }                                         \
  } while (0)

#define TFGNN_LAUNCH_CHECK()                                                               \
  do {                                                                                     \
    hipError_t _e = hipGetLastError();                                                     \
    if (_e != hipSuccess) {                                                                \
      ::tfgnn::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),        \
                         __FILE__, __LINE__);                                              \
      return TFGNN_ERR_HIP;                                                                \
    }                                                                                      \
  } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Split-K count of a product whose output grid has `tiles` workgroups (before the limits of K and of the workspace):
// one round of workgroups on the 256 CUs, at most 64 splits - more only add rounds on the hot-path shapes (12 tiles x 43
// splits ran as 3 rounds of 22 K tiles, x 21 as one round of 45).  Skinny outputs under a very long K (QM9-sized weight
// gradients: 1 - 8 tiles, K ~ 10^6 rows) are the exception: 64 workgroups leave three quarters of the chip idle, so
// they get two workgroups per CU with chunks of at least 512 rows.
static inline int64_t splitk_want(int64_t tiles, int64_t K) {
  if (tiles < 1) tiles = 1;
  static const bool long_k = [] { const char* e = getenv("TFGNN_LONG_K_SPLITS"); return !e || atoi(e) != 0; }();  // 0: A/B probe
  if (long_k && tiles <= 8 && K >= 131072) {
    const int64_t want = std::min<int64_t>(512 / tiles, K / 512);
    return std::min<int64_t>(want, 512);
  }
  return std::max<int64_t>(1, std::min<int64_t>(256 / tiles, 64));
}

// ---- dropout masks (gnn.py:285-288, [ext] tf.nn.dropout: keep with probability 1 - rate, scale kept values by 1/(1-rate)) ----
// Counter-based: the decision of element i of a call is a pure function of (seed, i), so the kernel that PRODUCES a tensor can
// drop it in its epilogue, the stand-alone kernel gives the same mask, and the backward pass RECOMPUTES the mask instead of
// reading one back (no mask tensor exists on the fused path).  32-bit arithmetic (two v_mul_lo_u32 per element; the 64-bit
// splitmix finaliser of rounds 1-3 cost ~200 VALU cycles per element - fine in a bandwidth-bound kernel, not in a product's
// epilogue): the host turns the call's seed into two well-mixed words (splitmix64), the element hashes its index xor the first
// with the "lowbias32" finaliser and xors the second; 24 bits of the word are the uniform number compared with rate * 2^24.
//
// Epoch (round 5): a step REPLAYED from a captured hipGraph launches the same kernels with the same arguments - the same
// seeds - every time.  The mask is therefore a function of (seed, i, epoch), where the epoch is a word in DEVICE memory that
// every kernel drawing a mask reads once at its start (dropout_resolve); tfgnn_dropout_epoch_advance - a one-thread kernel,
// the first node of a captured step - bumps it per replay.  Epoch 0 (never advanced: every eager caller) leaves the key
// untouched, so the masks of rounds 1-4 are unchanged; forward and backward kernels of one step see the same epoch.
struct DropoutKey {
  uint32_t s0, s1;
  uint32_t threshold;  // keep iff u24 >= threshold, threshold = ceil(rate * 2^24)
  float scale;         // 1 / (1 - rate)
  const uint32_t* epoch;  // device word (elementwise.hip dropout_epoch_word), or NULL
};
uint32_t* dropout_epoch_word();  // the library's epoch word in device memory, allocated (zero) at first use; NULL on failure
static inline DropoutKey dropout_key(uint64_t seed, float rate) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  DropoutKey k;
  k.s0 = (uint32_t)z;
  k.s1 = (uint32_t)(z >> 32);
  const double t = (double)rate * 16777216.0;
  uint32_t th = (uint32_t)t;
  if ((double)th < t) ++th;
  k.threshold = th;
  k.scale = 1.f / (1.f - rate);
  k.epoch = rate > 0.f ? dropout_epoch_word() : nullptr;
  return k;
}
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// the key of the current epoch: once per kernel, before the first mask is drawn
__device__ __forceinline__ DropoutKey dropout_resolve(DropoutKey k) {
  if (k.epoch) {
    const uint32_t e = *k.epoch;
    if (e) {
      k.s0 ^= lowbias32(e * 0x9E3779B9u + 0x7F4A7C15u);
      k.s1 += e * 0x85EBCA6Bu;
    }
  }
  return k;
}
// One hash serves the FOUR elements of an aligned group (idx / 4): element e takes the word rotated left by 8 e bits, i.e. its
// decision hangs on a byte of its own (the top byte of its 24-bit number; the lower 16 bits are shared with neighbours and
// matter only within 2^-8 of the threshold).  Every element's number is uniform; the hash (two quarter-rate multiplies)
// costs half a multiply per element - an epilogue of 160 elements per thread spent ~12 us on one hash per element.
__device__ __forceinline__ uint32_t dropout_word(const DropoutKey& k, uint64_t group) {
  uint32_t s = k.s0;
  const uint32_t hi = (uint32_t)(group >> 32);
  if (hi) s ^= lowbias32(hi * 0x9E3779B9u + 1u);  // tensors of more than 2^34 elements
  return lowbias32((uint32_t)group ^ s) ^ k.s1;
}
__device__ __forceinline__ float dropout_from_word(const DropoutKey& k, uint32_t w, int e) {
  const uint32_t u24 = __builtin_rotateleft32(w, 8u * (unsigned)e) >> 8;
  return u24 >= k.threshold ? k.scale : 0.f;
}
// mask value (0 or 1/(1-rate)) of flat element index idx
__device__ __forceinline__ float dropout_mask_at(const DropoutKey& k, uint64_t idx) {
  return dropout_from_word(k, dropout_word(k, idx >> 2), (int)(idx & 3));
}
// the masks of elements idx .. idx + 3 (idx % 4 == 0)
__device__ __forceinline__ float4 dropout_mask4(const DropoutKey& k, uint64_t idx) {
  const uint32_t w = dropout_word(k, idx >> 2);
  return make_float4(dropout_from_word(k, w, 0), dropout_from_word(k, w, 1), dropout_from_word(k, w, 2), dropout_from_word(k, w, 3));
}

constexpr float kSmallNumber = 1e-7f;  // tf2_gnn/utils/constants.py:2
constexpr float kFloatLowest = -3.402823466e+38f;

// ---- activations (tf2_gnn/utils/param_helpers.py:25-33, utils/activation.py:7-14) -------------
// tanh in two pieces, selected without a branch (the epilogue of a product applies it to 160 elements per thread):
//   |x| <  0.625: x + x u q(u), u = x^2, q = degree-4 minimax fit of (tanh(sqrt u) / sqrt u - 1) / u on [0, 0.625^2]
//                 (fit error 4e-9 relative; evaluated in fp32 with fma: <= 1.5 ulp of tanh, tanh(x) -> x exactly for tiny x);
//   |x| >= 0.625: sign(x) (1 - 2 / (e^{2|x|} + 1)) on the hardware exponential with an IEEE division.  The subtraction
//                 cancels at most one bit there (the subtracted term is <= 0.78), and the exponential's relative error d
//                 reaches the result as d / sinh(2|x|) <= 0.62 d: <= 3 ulp of tanh with a 1-ulp v_exp_f32.
// Rounds 1-4 used the second form for every x: absolutely accurate (1.4e-7) but not RELATIVELY - 1 - 2/(t+1) cancels for
// small |x| (relative error 1e-7 / |x|, tanh(1e-8) = 0), which TensorFlow's tanh does not do.
// tests/test_gpu_ops.py::test_tanh_relative_accuracy: log-spaced 1e-7 .. 10, both signs, <= 4 ulp.
// Why not tanhf: its ~40 instructions with branches were 17 of 51 us in the epilogue of a K = 320 product
// (tools/nt_epilogue_probe.py).  Saturates correctly: e^{2|x|} = inf -> 1.
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float u = x * x;
  float q = fmaf(u, -5.6599969257e-03f, 2.0590996108e-02f);
  q = fmaf(u, q, -5.3721070702e-02f);
  q = fmaf(u, q, 1.3331127574e-01f);
  q = fmaf(u, q, -3.3333260529e-01f);
  const float small = fmaf(x * u, q, x);
  const float t = __expf(2.f * ax);
  const float big = copysignf(1.f - 2.f / (t + 1.f), x);
  return ax < 0.625f ? small : big;
}

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case TFGNN_ACT_RELU:
      return x > 0.f ? x : 0.f;
    case TFGNN_ACT_TANH:
      return fast_tanh(x);
    case TFGNN_ACT_LEAKY_RELU:
      return x > 0.f ? x : 0.2f * x;
    case TFGNN_ACT_ELU:
      return x > 0.f ? x : expm1f(x);
    case TFGNN_ACT_SELU: {
      const float scale = 1.0507009873554804934193349852946f;
      const float alpha = 1.6732632423543772848170429916717f;
      return x > 0.f ? scale * x : scale * alpha * expm1f(x);
    }
    case TFGNN_ACT_GELU: {
      const float c = 0.7978845608028654f;  // sqrt(2/pi)
      float cdf = 0.5f * (1.0f + fast_tanh(c * (x + 0.044715f * x * x * x)));
      return x * cdf;
    }
    case TFGNN_ACT_SIGMOID:
      return 1.0f / (1.0f + expf(-x));
    default:
      return x;
  }
}

// derivative given the saved tensor value s (output y, except gelu: input x)
__device__ __forceinline__ float act_grad(int act, float s) {
  switch (act) {
    case TFGNN_ACT_RELU:
      return s > 0.f ? 1.f : 0.f;
    case TFGNN_ACT_TANH:
      return 1.f - s * s;
    case TFGNN_ACT_LEAKY_RELU:
      return s > 0.f ? 1.f : 0.2f;
    case TFGNN_ACT_ELU:
      return s > 0.f ? 1.f : s + 1.f;
    case TFGNN_ACT_SELU: {
      const float scale = 1.0507009873554804934193349852946f;
      const float alpha = 1.6732632423543772848170429916717f;
      return s > 0.f ? scale : s + scale * alpha;
    }
    case TFGNN_ACT_GELU: {
      const float c = 0.7978845608028654f;
      float x = s;
      float u = c * (x + 0.044715f * x * x * x);
      float t = fast_tanh(u);
      float du = c * (1.f + 3.f * 0.044715f * x * x);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
    }
    case TFGNN_ACT_SIGMOID:
      return s * (1.f - s);
    default:
      return 1.f;
  }
}

}  // namespace tfgnn
