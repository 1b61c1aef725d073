// Task metrics at the end of the path: the losses the reference differentiates and the
// numbers its epoch loop aggregates.
//   tfgnn_sigmoid_ce_metrics: NodeMulticlassTask._fast_task_metrics + micro_f1 (tf2_gnn/models/node_multiclass_task.py:10-23,62-70)
//   tfgnn_regression_metrics: tf.losses.mean_squared_error / mean_absolute_error of the per-graph outputs
//                             (tf2_gnn/models/graph_regression_task.py:157-158, tf2_gnn/models/qm9_regression.py:122-123)
// Both are one streaming pass (HBM-bound, 8 - 12 bytes per element) with a two-stage, fixed-order reduction: stage 1 leaves
// one partial per workgroup, stage 2 (one workgroup) adds them in index order, so the loss is reproducible run to run.
#include <algorithm>

#include "common.hpp"

namespace tfgnn {

constexpr int TASK_BLOCKS = 1024;

struct TaskPartial {
  double a;              // CE: sum of element losses | regression: sum of squared errors
  double b;              // regression: sum of absolute errors
  long long tp, fp, fn;  // CE: micro-F1 counts
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void block_reduce_store(TaskPartial mine, TaskPartial* out) {
  __shared__ TaskPartial sh[4];
  mine.a = wave_sum(mine.a);
  mine.b = wave_sum(mine.b);
  mine.tp = wave_sum(mine.tp);
  mine.fp = wave_sum(mine.fp);
  mine.fn = wave_sum(mine.fn);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    TaskPartial t = sh[0];
    for (int w = 1; w < 4; ++w) {
      t.a += sh[w].a; t.b += sh[w].b; t.tp += sh[w].tp; t.fp += sh[w].fp; t.fn += sh[w].fn;
    }
    *out = t;
  }
}

// element (v, c): x = logits, z = labels.
//   loss  max(x, 0) - x z + log1p(exp(-|x|))            [ext] tf.nn.sigmoid_cross_entropy_with_logits
//   pred  round(sigmoid(x)) as int32, label int32(z)     node_multiclass_task.py:12-14
//   grad  (sigmoid(x) - z) / V                           d mean_v(sum_c loss) / dx
__global__ void __launch_bounds__(256)
sigmoid_ce_partial_kernel(const float* __restrict__ logits, int64_t ld_x, const float* __restrict__ labels, int64_t ld_z,
                          int64_t V, int64_t C, float inv_v, float* __restrict__ dlogits, TaskPartial* __restrict__ partials) {
  TaskPartial mine = {0.0, 0.0, 0, 0, 0};
  const int64_t total = V * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / C, c = i - v * C;
    const float x = logits[v * ld_x + c], z = labels[v * ld_z + c];
    mine.a += (double)(fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))));
    const float p = 1.0f / (1.0f + expf(-x));
    const int pred = (int)rintf(p), lab = (int)z;
    mine.tp += (pred * lab) != 0;
    mine.fp += (pred * (lab - 1)) != 0;
    mine.fn += ((pred - 1) * lab) != 0;
    if (dlogits) dlogits[i] = (p - z) * inv_v;
  }
  block_reduce_store(mine, partials + blockIdx.x);
}

__global__ void __launch_bounds__(256)
regression_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, int64_t G, float two_over_g,
                          float* __restrict__ dpred, TaskPartial* __restrict__ partials) {
  TaskPartial mine = {0.0, 0.0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i];
    mine.a += (double)d * d;
    mine.b += (double)fabsf(d);
    if (dpred) dpred[i] = two_over_g * d;  // d mse / d pred
  }
  block_reduce_store(mine, partials + blockIdx.x);
}

// kind 0: metrics = {mean per-node loss, micro-F1}, counts = {tp, fp, fn};  kind 1: metrics = {mse, mae}
__global__ void __launch_bounds__(256)
task_final_kernel(const TaskPartial* __restrict__ partials, int n, int kind, double denom, float* __restrict__ metrics,
                  long long* __restrict__ counts) {
  TaskPartial mine = {0.0, 0.0, 0, 0, 0};
  // fixed order: thread t adds partials t, t+256, ...; the tree over threads is fixed as well
  for (int i = threadIdx.x; i < n; i += 256) {
    const TaskPartial p = partials[i];
    mine.a += p.a; mine.b += p.b; mine.tp += p.tp; mine.fp += p.fp; mine.fn += p.fn;
  }
  __shared__ TaskPartial total;
  block_reduce_store(mine, &total);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (kind == 0) {
      metrics[0] = (float)(total.a / denom);
      // int64 / int64 -> float64 in the reference (node_multiclass_task.py:20-23); 0/0 -> nan as there
      const double tp = (double)total.tp, fp = (double)total.fp, fn = (double)total.fn;
      const double precision = tp / (tp + fp), recall = tp / (tp + fn);
      metrics[1] = (float)((2.0 * precision * recall) / (precision + recall));
      if (counts) { counts[0] = total.tp; counts[1] = total.fp; counts[2] = total.fn; }
    } else {
      metrics[0] = (float)(total.a / denom);
      metrics[1] = (float)(total.b / denom);
    }
  }
}

static int task_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(TASK_BLOCKS, ceil_div(n, 256 * 4))); }

}  // namespace tfgnn

extern "C" size_t tfgnn_task_metrics_workspace_bytes(void) { return sizeof(tfgnn::TaskPartial) * tfgnn::TASK_BLOCKS; }

extern "C" int tfgnn_sigmoid_ce_metrics(const float* d_logits, int64_t ld_logits, const float* d_labels, int64_t ld_labels,
                                        int64_t V, int64_t C, float* d_metrics, int64_t* d_counts, float* d_dlogits,
                                        void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(V > 0 && C > 0, "sigmoid_ce_metrics: empty batch (the reference's reduce_mean gives nan)");
  TFGNN_REQUIRE(d_logits && d_labels && d_metrics, "NULL pointer");
  TFGNN_REQUIRE(ld_logits >= C && ld_labels >= C, "bad leading dimension");
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= tfgnn_task_metrics_workspace_bytes() && (uintptr_t)d_workspace % 8 == 0,
                "workspace of tfgnn_task_metrics_workspace_bytes() bytes required");
  hipStream_t s = (hipStream_t)stream;
  TaskPartial* partials = (TaskPartial*)d_workspace;
  const int grid = task_grid(V * C);
  hipLaunchKernelGGL(sigmoid_ce_partial_kernel, dim3(grid), dim3(256), 0, s, d_logits, ld_logits, d_labels, ld_labels, V, C,
                     1.0f / (float)V, d_dlogits, partials);
  TFGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(task_final_kernel, dim3(1), dim3(256), 0, s, partials, grid, 0, (double)V, d_metrics, (long long*)d_counts);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_regression_metrics(const float* d_pred, const float* d_target, int64_t G, float* d_metrics, float* d_dpred,
                                        void* d_workspace, size_t workspace_bytes, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(G > 0, "regression_metrics: empty batch (the reference's mean gives nan)");
  TFGNN_REQUIRE(d_pred && d_target && d_metrics, "NULL pointer");
  TFGNN_REQUIRE(d_workspace && workspace_bytes >= tfgnn_task_metrics_workspace_bytes() && (uintptr_t)d_workspace % 8 == 0,
                "workspace of tfgnn_task_metrics_workspace_bytes() bytes required");
  hipStream_t s = (hipStream_t)stream;
  TaskPartial* partials = (TaskPartial*)d_workspace;
  const int grid = task_grid(G);
  hipLaunchKernelGGL(regression_partial_kernel, dim3(grid), dim3(256), 0, s, d_pred, d_target, G, 2.0f / (float)G, d_dpred, partials);
  TFGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(task_final_kernel, dim3(1), dim3(256), 0, s, partials, grid, 1, (double)G, d_metrics, (long long*)nullptr);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}
