// tfgnn_aux_launch: up to 8 small passes in one launch (aux_jobs.hpp).
#include <algorithm>

#include "aux_jobs.hpp"

namespace tfgnn {

constexpr int AUX_MAX_JOBS = 8;
struct AuxJobTable {
  tfgnn_aux_job j[AUX_MAX_JOBS];
  int n;
};

typedef const __attribute__((address_space(4))) tfgnn_aux_job* aux_job_cptr;

template <class P>
__device__ __forceinline__ P aux_payload(aux_job_cptr j) {
  // word-wise from the constant address space (scalar loads from the argument segment); payload sits at offset 8 of a job
  constexpr int W = (int)((sizeof(P) + 7) / 8);
  union {
    P p;
    unsigned long long w[W];
  } u;
  const __attribute__((address_space(4))) unsigned long long* src =
      (const __attribute__((address_space(4))) unsigned long long*)&j->payload[0];
#pragma unroll
  for (int i = 0; i < W; ++i) u.w[i] = src[i];
  return u.p;
}

__global__ void __launch_bounds__(256) aux_jobs_kernel(AuxJobTable table) {
  // the table is the first (only) explicit argument: offset 0 of the kernel-argument segment.  Indexing the by-value
  // parameter with a run-time job index would make the compiler copy all 2 KB of it to scratch memory first.
  aux_job_cptr jobs = (aux_job_cptr)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned b = blockIdx.x;
  int k = 0;
  const int n = table.n;
  while (k < n - 1 && b >= jobs[k].num_blocks) {
    b -= jobs[k].num_blocks;
    ++k;
  }
  aux_job_cptr j = jobs + k;
  const unsigned nb = j->num_blocks;
  if (b >= nb) return;
  switch (j->kind) {
    case AUX_SPLIT_ROWS: {
      const AuxSplitRows a = aux_payload<AuxSplitRows>(j);
      sp_split_rows_body(a.src, a.ld, a.seg_len, a.seg_stride, a.R, a.C, a.sb, a.dst, a.ld_dst, a.inv, a.fixed_inv, b);
      break;
    }
    case AUX_SPLIT_COLS: {
      const AuxSplitCols a = aux_payload<AuxSplitCols>(j);
      sp_split_cols_body(a.src, a.ld, a.K, a.N, a.dst, a.ld_dst, a.inv, b % a.ncx, b / a.ncx, a.ncy, a.colmax_parts, a.nparts);
      break;
    }
    case AUX_COL_ABSMAX: {
      const AuxColAbsmax a = aux_payload<AuxColAbsmax>(j);
      col_absmax_body(a, b);
      break;
    }
    case AUX_TN_REDUCE: {
      const AuxTnReduce a = aux_payload<AuxTnReduce>(j);
      sp_tn_reduce_body(a, b, nb);
      break;
    }
    case AUX_TN_FACTORS: {
      const AuxTnFactors a = aux_payload<AuxTnFactors>(j);
      sp_tn_factors_body(a, b);
      break;
    }
    case AUX_COMBINE_SP: {
      const AuxCombineSp a = aux_payload<AuxCombineSp>(j);
      combine_sp_body(a, b);
      break;
    }
    default:
      break;
  }
}

}  // namespace tfgnn

extern "C" int tfgnn_aux_launch(const tfgnn_aux_job* jobs, int num_jobs, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(num_jobs >= 0 && (num_jobs == 0 || jobs != nullptr), "tfgnn_aux_launch: bad arguments");
  int i = 0;
  while (i < num_jobs) {
    AuxJobTable t{};
    uint64_t blocks = 0;
    while (i < num_jobs && t.n < AUX_MAX_JOBS) {
      const tfgnn_aux_job& j = jobs[i++];
      if (j.kind == AUX_NONE || j.num_blocks == 0) continue;
      TFGNN_REQUIRE(j.kind > AUX_NONE && j.kind < AUX_KIND_END, "tfgnn_aux_launch: unknown job kind %d", j.kind);
      t.j[t.n++] = j;
      blocks += j.num_blocks;
    }
    if (t.n == 0) continue;
    TFGNN_REQUIRE(blocks < (1ull << 31), "tfgnn_aux_launch: too many workgroups");
    hipLaunchKernelGGL(aux_jobs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}
