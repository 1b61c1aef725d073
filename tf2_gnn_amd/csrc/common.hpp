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

constexpr float kSmallNumber = 1e-7f;  // tf2_gnn/utils/constants.py:2
constexpr float kFloatLowest = -3.402823466e+38f;

// ---- activations (tf2_gnn/utils/param_helpers.py:25-33, utils/activation.py:7-14) -------------
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case TFGNN_ACT_RELU:
      return x > 0.f ? x : 0.f;
    case TFGNN_ACT_TANH:
      return tanhf(x);
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
      float cdf = 0.5f * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
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
      float t = tanhf(u);
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
