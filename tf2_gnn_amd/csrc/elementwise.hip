// Element-wise kernels of the hot path: activations (tf2_gnn/utils/param_helpers.py:21-39,
// applied at message_passing.py:169-177) and their gradients.
#include <algorithm>

#include "common.hpp"
#include "sp16.hpp"

namespace tfgnn {

__global__ void __launch_bounds__(256)
act_forward_kernel(int act, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i + 3 < n; i += stride * 4) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    v.x = act_apply(act, v.x);
    v.y = act_apply(act, v.y);
    v.z = act_apply(act, v.z);
    v.w = act_apply(act, v.w);
    *reinterpret_cast<float4*>(y + i) = v;
  }
  for (; i < n; ++i) {  // at most one thread lands here with a partial tail
    y[i] = act_apply(act, x[i]);
  }
}

__global__ void __launch_bounds__(256)
act_backward_kernel(int act, const float* __restrict__ dy, const float* __restrict__ saved,
                    float* __restrict__ dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i + 3 < n; i += stride * 4) {
    float4 g = *reinterpret_cast<const float4*>(dy + i);
    float4 s = *reinterpret_cast<const float4*>(saved + i);
    g.x *= act_grad(act, s.x);
    g.y *= act_grad(act, s.y);
    g.z *= act_grad(act, s.z);
    g.w *= act_grad(act, s.w);
    *reinterpret_cast<float4*>(dx + i) = g;
  }
  for (; i < n; ++i) dx[i] = dy[i] * act_grad(act, saved[i]);
}

// dx = dy * mul * act'(saved): the dropout mask of a layer input and the activation derivative of the layer below in one pass
__global__ void __launch_bounds__(256)
act_backward_mul_kernel(int act, const float* __restrict__ dy, const float* __restrict__ saved, const float* __restrict__ mul,
                        float* __restrict__ dx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (; i + 3 < n; i += stride * 4) {
    float4 g = *reinterpret_cast<const float4*>(dy + i);
    const float4 s = *reinterpret_cast<const float4*>(saved + i);
    const float4 m = *reinterpret_cast<const float4*>(mul + i);
    g.x = (g.x * m.x) * act_grad(act, s.x);  // (the order of tfgnn_mul followed by tfgnn_activation_backward: same bits)
    g.y = (g.y * m.y) * act_grad(act, s.y);
    g.z = (g.z * m.z) * act_grad(act, s.z);
    g.w = (g.w * m.w) * act_grad(act, s.w);
    *reinterpret_cast<float4*>(dx + i) = g;
  }
  for (; i < n; ++i) dx[i] = (dy[i] * mul[i]) * act_grad(act, saved[i]);
}

__global__ void act_forward_scalar_kernel(int act, const float* x, float* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = act_apply(act, x[i]);
}
__global__ void act_backward_scalar_kernel(int act, const float* dy, const float* saved, float* dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * act_grad(act, saved[i]);
}

// ---- GRUCell gates ([ext] tf.keras.layers.GRUCell, reset_after=True; ggnn.py:84-87) -----------
// mx = x @ kernel + bias[0], mh = h @ recurrent_kernel + bias[1]  (both [V, 3H], gate order z|r|h)
//   z = sigmoid(mx_z + mh_z); r = sigmoid(mx_r + mh_r); c = tanh(mx_h + r * mh_h); h' = z*h + (1-z)*c
// gates (optional) receives z|r|c for the backward pass.
__global__ void __launch_bounds__(256)
gru_gates_forward_kernel(const float* __restrict__ mx, const float* __restrict__ mh, const float* __restrict__ h,
                         float* __restrict__ h_new, float* __restrict__ gates, int64_t V, int H) {
  const int64_t total = V * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / H;
    const int j = (int)(i - v * H);
    const float* px = mx + v * 3 * H;
    const float* ph = mh + v * 3 * H;
    const float z = 1.f / (1.f + expf(-(px[j] + ph[j])));
    const float r = 1.f / (1.f + expf(-(px[H + j] + ph[H + j])));
    const float c = tanhf(px[2 * H + j] + r * ph[2 * H + j]);
    const float hv = h[i];
    h_new[i] = z * hv + (1.f - z) * c;
    if (gates) {
      float* pg = gates + v * 3 * H;
      pg[j] = z;
      pg[H + j] = r;
      pg[2 * H + j] = c;
    }
  }
}

// dmx, dmh [V,3H], dh_direct [V,H] = dh_new * z
__global__ void __launch_bounds__(256)
gru_gates_backward_kernel(const float* __restrict__ dh_new, const float* __restrict__ gates,
                          const float* __restrict__ mh, const float* __restrict__ h, float* __restrict__ dmx,
                          float* __restrict__ dmh, float* __restrict__ dh_direct, int64_t V, int H) {
  const int64_t total = V * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / H;
    const int j = (int)(i - v * H);
    const float* pg = gates + v * 3 * H;
    const float z = pg[j], r = pg[H + j], c = pg[2 * H + j];
    const float hh = mh[v * 3 * H + 2 * H + j];
    const float g = dh_new[i];
    const float dc = g * (1.f - z);
    const float dz = g * (h[i] - c);
    const float dpc = dc * (1.f - c * c);
    const float dpz = dz * z * (1.f - z);
    const float dpr = dpc * hh * r * (1.f - r);
    float* ox = dmx + v * 3 * H;
    float* oh = dmh + v * 3 * H;
    ox[j] = dpz;
    ox[H + j] = dpr;
    ox[2 * H + j] = dpc;
    oh[j] = dpz;
    oh[H + j] = dpr;
    oh[2 * H + j] = dpc * r;
    dh_direct[i] = g * z;
  }
}

// The same gate gradients written as the SPLIT OPERANDS of the f16x2 products that consume them (csrc/gemm_sp.hip: dmx, dmh
// [V, 3H] as SP16 rows with one power-of-two scale per row), the bias gradients (column sums of dmx and dmh) folded in as
// per-wave partial sums: no fp32 dmx / dmh is ever written or re-read.  One wave per row, lane j owns units j, j + 64, ...
// (H % 64 == 0); a wave walks rows w, w + W, ... in order and keeps the six column sums of each of its units in registers
// -> partial[w][6H] (x gates z, r, c then h gates z, r, c), added over w in order by colsum_final_kernel.
template <int UPL>  // units per lane = H / 64
__global__ void __launch_bounds__(256)
gru_gates_backward_sp_kernel(const float* __restrict__ dh_new, const float* __restrict__ gates, const float* __restrict__ mh,
                             const float* __restrict__ h, uint8_t* __restrict__ dmx_sp, float* __restrict__ dmx_inv,
                             uint8_t* __restrict__ dmh_sp, float* __restrict__ dmh_inv, float* __restrict__ dh_direct,
                             const float* __restrict__ out_mul, float* __restrict__ partial, int64_t V, int drop_on,
                             DropoutKey drop_arg) {
  constexpr int H = 64 * UPL;
  const DropoutKey drop = drop_on ? dropout_resolve(drop_arg) : drop_arg;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), W = (int64_t)gridDim.x * 4;
  float cs[6][UPL];
#pragma unroll
  for (int g = 0; g < 6; ++g)
#pragma unroll
    for (int u = 0; u < UPL; ++u) cs[g][u] = 0.f;
  for (int64_t v = w; v < V; v += W) {
    const float* pg = gates + v * 3 * H;
    const float* pm = mh + v * 3 * H + 2 * H;
    float x[3][UPL], y[UPL];  // dmx gates; dmh differs in the candidate gate only
    float mxx = 0.f, mxh = 0.f;
#pragma unroll
    for (int u = 0; u < UPL; ++u) {
      const int j = lane + 64 * u;
      const float z = pg[j], r = pg[H + j], c = pg[2 * H + j];
      const float hh = pm[j];
      const float g = dh_new[v * H + j];
      const float dc = g * (1.f - z);
      const float dz = g * (h[v * H + j] - c);
      const float dpc = dc * (1.f - c * c);
      const float dpz = dz * z * (1.f - z);
      const float dpr = dpc * hh * r * (1.f - r);
      x[0][u] = dpz; x[1][u] = dpr; x[2][u] = dpc; y[u] = dpc * r;
      // the factor of d h: a stored mask, or the mask tfgnn_dropout_forward draws for (seed, rate) at element v H + j recomputed
      float keep = out_mul ? out_mul[v * H + j] : 1.f;
      if (drop_on) keep = dropout_mask_at(drop, (uint64_t)(v * H + j));
      dh_direct[v * H + j] = (out_mul || drop_on) ? g * z * keep : g * z;
      cs[0][u] += dpz; cs[1][u] += dpr; cs[2][u] += dpc;
      cs[3][u] += dpz; cs[4][u] += dpr; cs[5][u] += dpc * r;
      const float m2 = fmaxf(fabsf(dpz), fabsf(dpr));
      mxx = fmaxf(mxx, fmaxf(m2, fabsf(dpc)));
      mxh = fmaxf(mxh, fmaxf(m2, fabsf(dpc * r)));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
      mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64));
      mxh = fmaxf(mxh, __shfl_xor(mxh, o, 64));
    }
    float ivx, ivh;
    const float sx = sp_scale_for_max(mxx, &ivx), sh = sp_scale_for_max(mxh, &ivh);
    if (lane == 0) {
      dmx_inv[v] = ivx;
      dmh_inv[v] = ivh;
    }
    uint8_t* rx = dmx_sp + v * (int64_t)(3 * H * 4);
    uint8_t* rh = dmh_sp + v * (int64_t)(3 * H * 4);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int u = 0; u < UPL; ++u) {
        const int c = g * H + lane + 64 * u;  // column of the [V, 3H] operand: granule c / 16, position c % 16
        _Float16 hi, lo;
        sp_split(x[g][u] * sx, hi, lo);
        _Float16* px = reinterpret_cast<_Float16*>(rx + (c >> 4) * 64) + (c & 15);
        px[0] = hi;
        px[16] = lo;
        sp_split((g == 2 ? y[u] : x[g][u]) * sh, hi, lo);
        _Float16* ph = reinterpret_cast<_Float16*>(rh + (c >> 4) * 64) + (c & 15);
        ph[0] = hi;
        ph[16] = lo;
      }
  }
  float* out = partial + w * 6 * H;
#pragma unroll
  for (int g = 0; g < 6; ++g)
#pragma unroll
    for (int u = 0; u < UPL; ++u) out[g * H + lane + 64 * u] = cs[g][u];
}

// out[n] = sum_m in[m, n]   (bias gradients).  Two deterministic stages: gridDim.y row slabs each
// reduce to partial[slab, n] (4 waves x 64 columns per workgroup), then the slabs are summed in order.
__global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ in, int64_t M, int N, int64_t ld, float* __restrict__ out, int64_t rows_per_slab) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t m1 = m0 + rows_per_slab < M ? m0 + rows_per_slab : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    int64_t m = m0 + w;
    for (; m + 28 < m1; m += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += in[(m + 4 * u) * ld + c];
    }
    for (; m < m1; m += 4) s[0] += in[m * ld + c];
  }
  red[w][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (w == 0 && c < N)
    out[(int64_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the same for 16-byte aligned rows: a lane owns four columns (a wave reads 1 KB of a row), eight rows in flight per lane
__global__ void __launch_bounds__(256)
colsum4_kernel(const float* __restrict__ in, int64_t M, int N, int64_t ld, float* __restrict__ out, int64_t rows_per_slab) {
  __shared__ float4 red[4][64];
  const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int w = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t m1 = m0 + rows_per_slab < M ? m0 + rows_per_slab : M;
  float4 s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
    int64_t m = m0 + w;
    for (; m + 28 < m1; m += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(in + (m + 4 * u) * ld + c);
        s[u].x += v.x; s[u].y += v.y; s[u].z += v.z; s[u].w += v.w;
      }
    }
    for (; m < m1; m += 4) {
      const float4 v = *reinterpret_cast<const float4*>(in + m * ld + c);
      s[0].x += v.x; s[0].y += v.y; s[0].z += v.z; s[0].w += v.w;
    }
  }
  float4 t;
  t.x = ((s[0].x + s[1].x) + (s[2].x + s[3].x)) + ((s[4].x + s[5].x) + (s[6].x + s[7].x));
  t.y = ((s[0].y + s[1].y) + (s[2].y + s[3].y)) + ((s[4].y + s[5].y) + (s[6].y + s[7].y));
  t.z = ((s[0].z + s[1].z) + (s[2].z + s[3].z)) + ((s[4].z + s[5].z) + (s[6].z + s[7].z));
  t.w = ((s[0].w + s[1].w) + (s[2].w + s[3].w)) + ((s[4].w + s[5].w) + (s[6].w + s[7].w));
  red[w][threadIdx.x & 63] = t;
  __syncthreads();
  if (w == 0 && c < N) {
    const float4 a = red[0][threadIdx.x], b = red[1][threadIdx.x], e = red[2][threadIdx.x], f = red[3][threadIdx.x];
    float* o = out + (int64_t)blockIdx.y * N + c;
    o[0] = (a.x + b.x) + (e.x + f.x); o[1] = (a.y + b.y) + (e.y + f.y);
    o[2] = (a.z + b.z) + (e.z + f.z); o[3] = (a.w + b.w) + (e.w + f.w);
  }
}

// stage 2: 64 columns per workgroup; wave w adds slabs w, w+4, ... (eight loads in flight), then the four waves in order
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int slabs, int N, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    int k = w;
    for (; k + 28 < slabs; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(int64_t)(k + 4 * u) * N + c];
    }
    for (; k < slabs; k += 4) s[0] += partial[(int64_t)k * N + c];
  }
  red[w][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (w == 0 && c < N) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// y = a * x + b * y' style helpers used by the layer stack (gnn.py:291-296 residual averaging):
// out = alpha * (x + y)
__global__ void __launch_bounds__(256)
add_scale_kernel(const float* __restrict__ x, const float* __restrict__ y, float alpha, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = alpha * (x[i] + y[i]);
}

// ---- dropout (gnn.py:285-288, [ext] tf.nn.dropout): masks are counter-based (common.hpp dropout_mask_at): reproducible for
// a given seed, independent of the launch geometry, and identical in every kernel that draws them (this file, the epilogue of
// the split-operand product).  The mask (0 or 1/(1-rate)) is stored for the backward pass when the caller asks for it.
__global__ void __launch_bounds__(256)
dropout_forward_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mask, int64_t n, DropoutKey key_arg) {
  const DropoutKey key = dropout_resolve(key_arg);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = dropout_mask_at(key, (uint64_t)i);
    if (mask) mask[i] = m;
    if (y) y[i] = x[i] * m;
  }
}

// dropout + the SP16 form of the result (the split operand of the weight-gradient product downstream: saves the separate
// split pass over the layer input).  16 lanes per row, a row's float4s held in registers between the mask pass and the
// split pass; the random number of element (r, c) is that of the flat kernel at index r * cols + c, so both kernels
// produce the same y and mask.
template <int VPL>
__global__ void __launch_bounds__(256)
dropout_forward_sp_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mask, int64_t rows, int cols,
                          DropoutKey key_arg, uint8_t* __restrict__ out_sp, int64_t ld_sp, float* __restrict__ inv_out) {
  const DropoutKey key = dropout_resolve(key_arg);
  const int sub = threadIdx.x & 15;
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (r >= rows) return;  // whole 16-lane groups leave together; the shuffles below stay inside a group
  const int64_t base = r * cols;
  float4 v[VPL];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (sub + 16 * j) * 4;
    if (c < cols) {
      const float4 xv = *reinterpret_cast<const float4*>(x + base + c);
      const float4 m4 = dropout_mask4(key, (uint64_t)(base + c));  // cols % 16 == 0: base + c is a multiple of 4
      const float m[4] = {m4.x, m4.y, m4.z, m4.w};
      v[j] = make_float4(xv.x * m[0], xv.y * m[1], xv.z * m[2], xv.w * m[3]);
      if (mask) *reinterpret_cast<float4*>(mask + base + c) = m4;
      *reinterpret_cast<float4*>(y + base + c) = v[j];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[j].x), fabsf(v[j].y)), fmaxf(fabsf(v[j].z), fabsf(v[j].w))));
      if (v[j].x != v[j].x || v[j].y != v[j].y || v[j].z != v[j].z || v[j].w != v[j].w) mx = __builtin_inff();
    }
  }
#pragma unroll
  for (int o = 8; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float inv;
  const float sc = sp_scale_for_max(mx, &inv);
  if (sub == 0) inv_out[r] = inv;
  uint8_t* row = out_sp + r * ld_sp;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c = (sub + 16 * j) * 4;
    if (c < cols) sp_store4(row, c, v[j], sc);
  }
}

__global__ void __launch_bounds__(256)
mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] * b[i];
}

// ---- LayerNormalization ([ext] tf.keras.layers.LayerNormalization defaults: axis=-1, eps=1e-3;
// gnn.py:157-161,318-321).  One wave per row.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

__global__ void __launch_bounds__(256)
layernorm_forward_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                         float eps, int64_t rows, int H, float* __restrict__ y, float* __restrict__ mean_out,
                         float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * H;
  float s = 0.f;
  for (int j = lane; j < H; j += 64) s += xr[j];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
  for (int j = lane; j < H; j += 64) {
    const float d = xr[j] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  float* yr = y + row * H;
  for (int j = lane; j < H; j += 64) yr[j] = (xr[j] - mean) * rstd * gamma[j] + beta[j];
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dy_xhat = dy * xhat (for d gamma)
__global__ void __launch_bounds__(256)
layernorm_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                          const float* __restrict__ mean_in, const float* __restrict__ rstd_in, int64_t rows, int H,
                          float* __restrict__ dx, float* __restrict__ dy_xhat) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mean = mean_in[row], rstd = rstd_in[row];
  const float* xr = x + row * H;
  const float* dr = dy + row * H;
  float sg = 0.f, sgx = 0.f;
  for (int j = lane; j < H; j += 64) {
    const float xh = (xr[j] - mean) * rstd;
    const float g = dr[j] * gamma[j];
    sg += g;
    sgx += g * xh;
  }
  sg = wave_sum(sg) / (float)H;
  sgx = wave_sum(sgx) / (float)H;
  for (int j = lane; j < H; j += 64) {
    const float xh = (xr[j] - mean) * rstd;
    const float g = dr[j] * gamma[j];
    dx[row * H + j] = rstd * (g - sg - xh * sgx);
    dy_xhat[row * H + j] = dr[j] * xh;
  }
}

static unsigned ew_blocks(int64_t n_vec) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n_vec, 256), 256 * 16));
}

}  // namespace tfgnn

extern "C" int tfgnn_activation_forward(int act, const float* d_x, float* d_y, int64_t n, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_y, "NULL pointer");
  TFGNN_REQUIRE(act >= TFGNN_ACT_NONE && act <= TFGNN_ACT_SIGMOID, "unknown activation %d", act);
  hipStream_t s = (hipStream_t)stream;
  if ((((uintptr_t)d_x | (uintptr_t)d_y) & 15) == 0)
    hipLaunchKernelGGL(act_forward_kernel, dim3(ew_blocks(ceil_div(n, 4))), dim3(256), 0, s, act, d_x, d_y, n);
  else
    hipLaunchKernelGGL(act_forward_scalar_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, act, d_x, d_y, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_activation_backward(int act, const float* d_dy, const float* d_saved, float* d_dx,
                                         int64_t n, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dy && d_saved && d_dx, "NULL pointer");
  TFGNN_REQUIRE(act >= TFGNN_ACT_NONE && act <= TFGNN_ACT_SIGMOID, "unknown activation %d", act);
  hipStream_t s = (hipStream_t)stream;
  if ((((uintptr_t)d_dy | (uintptr_t)d_saved | (uintptr_t)d_dx) & 15) == 0)
    hipLaunchKernelGGL(act_backward_kernel, dim3(ew_blocks(ceil_div(n, 4))), dim3(256), 0, s, act, d_dy, d_saved, d_dx, n);
  else
    hipLaunchKernelGGL(act_backward_scalar_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, act, d_dy, d_saved, d_dx, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_activation_backward_mul(int act, const float* d_dy, const float* d_saved, const float* d_mul, float* d_dx,
                                             int64_t n, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dy && d_saved && d_mul && d_dx, "NULL pointer");
  TFGNN_REQUIRE(act >= TFGNN_ACT_NONE && act <= TFGNN_ACT_SIGMOID, "unknown activation %d", act);
  TFGNN_REQUIRE((((uintptr_t)d_dy | (uintptr_t)d_saved | (uintptr_t)d_mul | (uintptr_t)d_dx) & 15) == 0,
                "tfgnn_activation_backward_mul: operands must be 16-byte aligned");
  hipLaunchKernelGGL(act_backward_mul_kernel, dim3(ew_blocks(ceil_div(n, 4))), dim3(256), 0, (hipStream_t)stream, act, d_dy, d_saved,
                     d_mul, d_dx, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_gru_gates_forward(const float* d_mx, const float* d_mh, const float* d_h, float* d_h_new,
                                       float* d_gates, int64_t V, int H, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && H >= 0, "negative size");
  if (V == 0 || H == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_mx && d_mh && d_h && d_h_new, "NULL pointer");
  hipLaunchKernelGGL(gru_gates_forward_kernel, dim3(ew_blocks(V * H)), dim3(256), 0, (hipStream_t)stream, d_mx, d_mh,
                     d_h, d_h_new, d_gates, V, H);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_gru_gates_backward(const float* d_dh_new, const float* d_gates, const float* d_mh,
                                        const float* d_h, float* d_dmx, float* d_dmh, float* d_dh_direct,
                                        int64_t V, int H, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && H >= 0, "negative size");
  if (V == 0 || H == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dh_new && d_gates && d_mh && d_h && d_dmx && d_dmh && d_dh_direct, "NULL pointer");
  hipLaunchKernelGGL(gru_gates_backward_kernel, dim3(ew_blocks(V * H)), dim3(256), 0, (hipStream_t)stream, d_dh_new,
                     d_gates, d_mh, d_h, d_dmx, d_dmh, d_dh_direct, V, H);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

static int gates_sp_waves(int64_t V) { return (int)(4 * std::max<int64_t>(1, std::min<int64_t>(512, (V + 255) / 256))); }

extern "C" size_t tfgnn_gru_gates_backward_sp_workspace_bytes(int64_t V, int H) {
  if (V <= 0 || H <= 0) return 0;
  return (size_t)gates_sp_waves(V) * 6 * (size_t)H * 4;
}

static int gru_gates_backward_sp_impl(const float* d_dh_new, const float* d_gates, const float* d_mh, const float* d_h,
                                      void* d_dmx_sp, float* d_dmx_inv_scale, void* d_dmh_sp, float* d_dmh_inv_scale,
                                      float* d_dh_direct, const float* d_out_mul, float* d_bias_grad, int64_t V, int H,
                                      void* d_workspace, size_t workspace_bytes, void* stream, float dropout_rate,
                                      uint64_t dropout_seed) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V >= 0 && H >= 0, "negative size");
  TFGNN_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f && !(dropout_rate > 0.f && d_out_mul),
                "tfgnn_gru_gates_backward_sp: a stored mask or a recomputed one, rate in [0, 1)");
  const int drop_on = dropout_rate > 0.f ? 1 : 0;
  if (H % 64 != 0 || H > 512) return TFGNN_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (V == 0) {
    if (d_bias_grad) TFGNN_HIP_CHECK(hipMemsetAsync(d_bias_grad, 0, (size_t)6 * H * 4, s));
    return TFGNN_OK;
  }
  TFGNN_REQUIRE(d_dh_new && d_gates && d_mh && d_h && d_dmx_sp && d_dmx_inv_scale && d_dmh_sp && d_dmh_inv_scale && d_dh_direct &&
                    d_bias_grad, "NULL pointer");
  TFGNN_REQUIRE(((uintptr_t)d_dmx_sp | (uintptr_t)d_dmh_sp) % 64 == 0, "SP16 operands must be 64-byte aligned");
  const int waves = gates_sp_waves(V);
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= (size_t)waves * 6 * H * 4, "workspace too small: need %zu bytes",
                (size_t)waves * 6 * H * 4);
  float* partial = (float*)d_workspace;
  const DropoutKey drop = dropout_key(dropout_seed, dropout_rate);
  TFGNN_REQUIRE(dropout_rate <= 0.f || drop.epoch, "tfgnn_gru_gates_backward_sp: no device memory for this device's epoch word");
  const dim3 grid(waves / 4), block(256);
#define GATES_SP(U)                                                                                                     \
  hipLaunchKernelGGL(gru_gates_backward_sp_kernel<U>, grid, block, 0, s, d_dh_new, d_gates, d_mh, d_h, (uint8_t*)d_dmx_sp, \
                     d_dmx_inv_scale, (uint8_t*)d_dmh_sp, d_dmh_inv_scale, d_dh_direct, d_out_mul, partial, V, drop_on, drop)
  switch (H / 64) {
    case 1: GATES_SP(1); break;
    case 2: GATES_SP(2); break;
    case 3: GATES_SP(3); break;
    case 4: GATES_SP(4); break;
    case 5: GATES_SP(5); break;
    case 6: GATES_SP(6); break;
    case 7: GATES_SP(7); break;
    default: GATES_SP(8); break;
  }
#undef GATES_SP
  TFGNN_LAUNCH_CHECK();
  // bias gradients [2, 3H]: the six column sums, waves added in order
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(6 * H, 64)), dim3(256), 0, s, partial, waves, 6 * H, d_bias_grad);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_gru_gates_backward_sp(const float* d_dh_new, const float* d_gates, const float* d_mh, const float* d_h,
                                           void* d_dmx_sp, float* d_dmx_inv_scale, void* d_dmh_sp, float* d_dmh_inv_scale,
                                           float* d_dh_direct, const float* d_out_mul, float* d_bias_grad, int64_t V, int H,
                                           void* d_workspace, size_t workspace_bytes, void* stream) {
  return gru_gates_backward_sp_impl(d_dh_new, d_gates, d_mh, d_h, d_dmx_sp, d_dmx_inv_scale, d_dmh_sp, d_dmh_inv_scale, d_dh_direct,
                                    d_out_mul, d_bias_grad, V, H, d_workspace, workspace_bytes, stream, 0.f, 0);
}

/* tfgnn_gru_gates_backward_sp with the factor of d h = the dropout mask tfgnn_dropout_forward draws for (seed, rate) over the
 * [V, H] layer input, recomputed per element instead of read (no mask tensor exists when the layer input was dropped without
 * storing one) */
extern "C" int tfgnn_gru_gates_backward_sp_dropout(const float* d_dh_new, const float* d_gates, const float* d_mh, const float* d_h,
                                                   void* d_dmx_sp, float* d_dmx_inv_scale, void* d_dmh_sp, float* d_dmh_inv_scale,
                                                   float* d_dh_direct, float dropout_rate, uint64_t dropout_seed, float* d_bias_grad,
                                                   int64_t V, int H, void* d_workspace, size_t workspace_bytes, void* stream) {
  return gru_gates_backward_sp_impl(d_dh_new, d_gates, d_mh, d_h, d_dmx_sp, d_dmx_inv_scale, d_dmh_sp, d_dmh_inv_scale, d_dh_direct,
                                    nullptr, d_bias_grad, V, H, d_workspace, workspace_bytes, stream, dropout_rate, dropout_seed);
}

// row slabs of stage 1: 64 rows and more each (16 per wave), so that a batch of a few thousand nodes already fills the chip
// (one slab per 512 rows left a [7110, 121] sum at 28 workgroups and 23 us)
static int64_t colsum_slabs(int64_t M) { return M <= 256 ? 1 : std::min<int64_t>(2048, (M + 63) / 64); }

extern "C" size_t tfgnn_colsum_workspace_bytes(int64_t M, int N) {
  if (N <= 0 || colsum_slabs(M) <= 1) return 0;
  return (size_t)colsum_slabs(M) * N * 4;
}

extern "C" int tfgnn_colsum(const float* d_in, int64_t M, int N, int64_t ld, float* d_out, void* d_workspace,
                            size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(M >= 0 && N >= 0, "negative size");
  if (N == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_out && (M == 0 || d_in) && ld >= N, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  int64_t slabs = colsum_slabs(M);
  if (!d_workspace || workspace_bytes < (size_t)slabs * N * 4) slabs = 1;
  const int64_t rows_per_slab = slabs > 1 ? ceil_div(M, slabs) : (M > 0 ? M : 1);
  float* stage1 = slabs > 1 ? (float*)d_workspace : d_out;
  if (N % 4 == 0 && ld % 4 == 0 && (uintptr_t)d_in % 16 == 0)
    hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)slabs), dim3(256), 0, s, d_in, M, N, ld,
                       stage1, rows_per_slab);
  else
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(N, 64), (unsigned)slabs), dim3(256), 0, s, d_in, M, N, ld, stage1,
                       rows_per_slab);
  TFGNN_LAUNCH_CHECK();
  if (slabs > 1) {
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)ceil_div(N, 64)), dim3(256), 0, s, stage1, (int)slabs, N, d_out);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}

extern "C" int tfgnn_add_scale(const float* d_x, const float* d_y, float alpha, float* d_out, int64_t n, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_y && d_out, "NULL pointer");
  hipLaunchKernelGGL(add_scale_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, d_x, d_y, alpha, d_out, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- rows of an SP16 operand through an index (round 5): dst row r = src row index[r], with its scales.  The expanded operand
// of the per-relation weight-gradient products (the K dimension of a TN product cannot be read through an index: its rows
// change every k step).  16 bytes per lane, a row of `row_bytes` (multiple of 64) by row_bytes / 16 consecutive lanes.
namespace tfgnn {
__global__ void __launch_bounds__(256) sp_gather_rows_kernel(const uint8_t* __restrict__ src, int64_t ld_src, const float* __restrict__ src_inv,
                                                             int inv_per_row, const int32_t* __restrict__ index, int64_t rows,
                                                             int64_t src_rows, int row_bytes, uint8_t* __restrict__ dst, int64_t ld_dst,
                                                             float* __restrict__ dst_inv) {
  const int lanes_per_row = row_bytes >> 4;
  const int64_t total = rows * lanes_per_row;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / lanes_per_row;
    const int c = (int)(i - r * lanes_per_row);
    const int64_t sr = index[r];
    const bool ok = sr >= 0 && sr < src_rows;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (ok) v = *reinterpret_cast<const uint4*>(src + sr * ld_src + (int64_t)c * 16);
    *reinterpret_cast<uint4*>(dst + r * ld_dst + (int64_t)c * 16) = v;
    if (c < inv_per_row) dst_inv[r * inv_per_row + c] = ok ? src_inv[sr * inv_per_row + c] : 1.1754943508222875e-38f;
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_sp_gather_rows(const void* d_src_sp, int64_t ld_src_bytes, const float* d_src_inv_scale, int scales_per_row,
                                    const int32_t* d_index, int64_t rows, int64_t src_rows, int64_t cols, void* d_dst_sp,
                                    int64_t ld_dst_bytes, float* d_dst_inv_scale, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(rows >= 0 && src_rows >= 0 && cols > 0 && cols % 16 == 0 && scales_per_row >= 1 && scales_per_row <= cols / 16,
                "tfgnn_sp_gather_rows: cols must be a positive multiple of 16");
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_src_sp && d_src_inv_scale && d_index && d_dst_sp && d_dst_inv_scale, "NULL pointer");
  TFGNN_REQUIRE(ld_src_bytes >= cols * 4 && ld_dst_bytes >= cols * 4 && ld_src_bytes % 16 == 0 && ld_dst_bytes % 16 == 0 &&
                    (uintptr_t)d_src_sp % 16 == 0 && (uintptr_t)d_dst_sp % 16 == 0,
                "tfgnn_sp_gather_rows: SP16 rows must be 16-byte aligned");
  const int64_t total = rows * (cols / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(total, 256), 256 * 64);
  hipLaunchKernelGGL(sp_gather_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_src_sp, ld_src_bytes,
                     d_src_inv_scale, scales_per_row, d_index, rows, src_rows, (int)(cols * 4), (uint8_t*)d_dst_sp, ld_dst_bytes,
                     d_dst_inv_scale);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- the dropout epoch (common.hpp DropoutKey): one word of device memory, bumped by a kernel so that the bump can be a
// node of a captured hipGraph ----
namespace tfgnn {
// One word PER DEVICE (a process-global word lived on whichever device drew the first mask; kernels on another
// device dereferenced a foreign pointer).  Allocated when a batch handle is created on the device (tfgnn_graph_create: never
// inside a stream capture) or at the first mask; a failed allocation is an ERROR for every caller with rate > 0
// (dropout_key_checked) - a NULL word would silently repeat the masks of a replayed step.
constexpr int kMaxDevices = 64;
static uint32_t* g_dropout_epoch[kMaxDevices] = {};
uint32_t* dropout_epoch_word() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (!g_dropout_epoch[dev]) {
    uint32_t* p = nullptr;
    if (hipMalloc((void**)&p, 256) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    if (hipMemset(p, 0, 256) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(p);
      return nullptr;
    }
    g_dropout_epoch[dev] = p;
  }
  return g_dropout_epoch[dev];
}
__global__ void dropout_epoch_kernel(uint32_t* word, uint32_t value, int add) { *word = add ? *word + value : value; }
}  // namespace tfgnn

extern "C" int tfgnn_dropout_epoch_advance(void* stream) {
  using namespace tfgnn;
  uint32_t* w = dropout_epoch_word();
  TFGNN_REQUIRE(w, "tfgnn_dropout_epoch_advance: no device memory for the epoch word");
  hipLaunchKernelGGL(dropout_epoch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, 1u, 1);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_dropout_epoch_set(uint32_t value, void* stream) {
  using namespace tfgnn;
  uint32_t* w = dropout_epoch_word();
  TFGNN_REQUIRE(w, "tfgnn_dropout_epoch_set: no device memory for the epoch word");
  hipLaunchKernelGGL(dropout_epoch_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, value, 0);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_dropout_epoch_get(uint32_t* h_value, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(h_value, "NULL pointer");
  uint32_t* w = dropout_epoch_word();
  TFGNN_REQUIRE(w, "tfgnn_dropout_epoch_get: no device memory for the epoch word");
  TFGNN_HIP_CHECK(hipMemcpyAsync(h_value, w, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
  TFGNN_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return TFGNN_OK;
}

extern "C" int tfgnn_dropout_forward(const float* d_x, float* d_y, float* d_mask, int64_t n, float rate,
                                     uint64_t seed, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  TFGNN_REQUIRE(rate >= 0.f && rate < 1.f, "dropout rate must be in [0, 1), got %f", (double)rate);
  if (n == 0) return TFGNN_OK;
  // d_y == NULL: only the mask of (seed, rate) is (re)generated - the fused producers draw the same one in their epilogues
  TFGNN_REQUIRE((d_y == nullptr || d_x != nullptr) && (d_y || d_mask), "NULL pointer");
  const DropoutKey key = dropout_key(seed, rate);
  TFGNN_REQUIRE(rate <= 0.f || key.epoch, "tfgnn_dropout_forward: no device memory for this device's epoch word");
  hipLaunchKernelGGL(dropout_forward_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, d_x, d_y, d_mask,
                     n, key);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_dropout_forward_sp(const float* d_x, float* d_y, float* d_mask, int64_t rows, int64_t cols, float rate,
                                        uint64_t seed, void* d_out_sp, int64_t ld_out_sp_bytes, float* d_inv_scale,
                                        void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(rows >= 0 && cols > 0, "bad sizes");
  TFGNN_REQUIRE(rate >= 0.f && rate < 1.f, "dropout rate must be in [0, 1), got %f", (double)rate);
  if (cols % 16 != 0 || cols > 512) {
    set_error("tfgnn_dropout_forward_sp: cols = %lld must be a multiple of 16 and at most 512", (long long)cols);
    return TFGNN_ERR_UNSUPPORTED;
  }
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_y && d_out_sp && d_inv_scale, "NULL pointer");  // d_mask may be NULL: it can be regenerated
  TFGNN_REQUIRE(ld_out_sp_bytes >= cols * 4 && ld_out_sp_bytes % 64 == 0 && (uintptr_t)d_out_sp % 64 == 0 &&
                    (uintptr_t)d_x % 16 == 0 && (uintptr_t)d_y % 16 == 0 && (uintptr_t)d_mask % 16 == 0,
                "tfgnn_dropout_forward_sp: SP16 rows must be 64-byte aligned, fp32 tensors 16-byte aligned");
  const dim3 grid((unsigned)ceil_div(rows, 16));
  const int vpl = (int)ceil_div(cols, 64);
  const DropoutKey key = dropout_key(seed, rate);
  TFGNN_REQUIRE(rate <= 0.f || key.epoch, "tfgnn_dropout_forward_sp: no device memory for this device's epoch word");
#define DROP_SP(V)                                                                                                      \
  hipLaunchKernelGGL((dropout_forward_sp_kernel<V>), grid, dim3(256), 0, (hipStream_t)stream, d_x, d_y, d_mask, rows, (int)cols, \
                     key, (uint8_t*)d_out_sp, ld_out_sp_bytes, d_inv_scale)
  if (vpl <= 2) DROP_SP(2);
  else if (vpl <= 4) DROP_SP(4);
  else if (vpl <= 5) DROP_SP(5);
  else DROP_SP(8);
#undef DROP_SP
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_mul(const float* d_a, const float* d_b, float* d_out, int64_t n, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_a && d_b && d_out, "NULL pointer");
  hipLaunchKernelGGL(mul_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, d_out, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_layernorm_forward(const float* d_x, const float* d_gamma, const float* d_beta, float eps,
                                       int64_t rows, int H, float* d_y, float* d_mean, float* d_rstd, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(rows >= 0 && H > 0, "bad sizes");
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_gamma && d_beta && d_y && d_mean && d_rstd, "NULL pointer");
  hipLaunchKernelGGL(layernorm_forward_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     d_x, d_gamma, d_beta, eps, rows, H, d_y, d_mean, d_rstd);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_layernorm_backward(const float* d_dy, const float* d_x, const float* d_gamma, const float* d_mean,
                                        const float* d_rstd, int64_t rows, int H, float* d_dx, float* d_dy_xhat,
                                        void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(rows >= 0 && H > 0, "bad sizes");
  if (rows == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dy && d_x && d_gamma && d_mean && d_rstd && d_dx && d_dy_xhat, "NULL pointer");
  hipLaunchKernelGGL(layernorm_backward_kernel, dim3((unsigned)ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     d_dy, d_x, d_gamma, d_mean, d_rstd, rows, H, d_dx, d_dy_xhat);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// dst[b, a, :] = src[a, b, :]  ([A, B, C] -> [B, A, C]).  Re-packs the per-edge-type kernels
// [L, D, H] (vertical stack, forward operand) into [D, L*H] (horizontal stack) so that the backward
// pass of a layer is two large GEMMs instead of 2*L small ones, and back for the gradients.
namespace tfgnn {
__global__ void __launch_bounds__(256)
permute_021_kernel(const float* __restrict__ src, int64_t A, int64_t B, int64_t C, float* __restrict__ dst) {
  const int64_t total = A * B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % C;
    const int64_t ab = i / C;
    const int64_t b = ab % B, a = ab / B;
    dst[(b * A + a) * C + c] = src[i];
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_permute_021(const float* d_src, int64_t A, int64_t B, int64_t C, float* d_dst, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(A >= 0 && B >= 0 && C >= 0, "negative size");
  if (A * B * C == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_src && d_dst, "NULL pointer");
  hipLaunchKernelGGL(permute_021_kernel, dim3(ew_blocks(A * B * C)), dim3(256), 0, (hipStream_t)stream, d_src, A, B, C, d_dst);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ---- batched 2-D transpose -----------------------------------------------------------------------
namespace tfgnn {
__global__ void __launch_bounds__(256)
transpose_batched_kernel(const float* __restrict__ src, int64_t rows, int64_t cols, float* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const float* s = src + b * rows * cols;
  float* d = dst + b * rows * cols;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * i][tx] = s[r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) d[c * rows + r] = tile[tx][ty + 8 * i];
  }
}
// tf.maximum(x, lo) / tf.minimum(x, hi) and their gradient (the gradient flows where lo <= x <= hi, TensorFlow's
// MaximumMinimumGrad: x >= lo for the maximum, x <= hi for the minimum) - nodes_to_graph_representation.py:194-197
__global__ void __launch_bounds__(256) clip_kernel(const float* __restrict__ x, int64_t n, float lo, float hi, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}
__global__ void __launch_bounds__(256) clip_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t n, float lo,
                                                            float hi, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    dx[i] = (v >= lo && v <= hi) ? dy[i] : 0.f;
  }
}

}  // namespace tfgnn

extern "C" int tfgnn_transpose_batched(const float* d_src, int64_t batch, int64_t rows, int64_t cols, float* d_dst,
                                       void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0, "negative size");
  if (batch * rows * cols == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_src && d_dst, "NULL pointer");
  TFGNN_REQUIRE(batch < 65536 && ceil_div(rows, 32) < 65536, "transpose grid too large");
  dim3 grid((unsigned)ceil_div(cols, 32), (unsigned)ceil_div(rows, 32), (unsigned)batch);
  hipLaunchKernelGGL(transpose_batched_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_src, rows, cols, d_dst);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

/* lower / upper: the bounds, or -inf / +inf for "no bound" */
extern "C" int tfgnn_clip(const float* d_x, int64_t n, float lower, float upper, float* d_y, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_x && d_y, "NULL pointer");
  hipLaunchKernelGGL(clip_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 16384)), dim3(256), 0, (hipStream_t)stream, d_x, n,
                     lower, upper, d_y);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_clip_backward(const float* d_dy, const float* d_x, int64_t n, float lower, float upper, float* d_dx, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(n >= 0, "negative size");
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE(d_dy && d_x && d_dx, "NULL pointer");
  hipLaunchKernelGGL(clip_backward_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(n, 256), 16384)), dim3(256), 0, (hipStream_t)stream,
                     d_dy, d_x, n, lower, upper, d_dx);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
