// Graph handle: bucket the edges of one batch by (target, edge_type) and by (source, edge_type).
//
// Replaces, once per batch instead of once per layer and pass, the index work the reference does
// inside MessagePassing.call (tf2_gnn/layers/message_passing/message_passing.py:166-167,195-206,
// 230-263): slicing the src/dst columns of every adjacency list, counting incoming edges per
// (type, node) with scatter_nd, gathering those counts per edge, and concatenating the per-type
// target lists.  Row r = node * L + edge_type; the in-degree c[l, v] is the length of row v*L+l.
//
// Pipeline (all on the caller's stream; deterministic result):
//   count keys (int atomics) -> exclusive scan -> scatter (col<<32 | edge_id) with an atomic cursor
//   -> per-row sort of the 64-bit composites (canonical order) -> unpack + derived arrays.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "graph.hpp"

namespace tfgnn {

// ------------------------------------------------------------------------------------------
// error plumbing (thread-local)
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int64_t g_launch_counts[TFGNN_KFAM_COUNT] = {0};
void count_launch(int family) {
  if (family >= 0 && family < TFGNN_KFAM_COUNT) __atomic_fetch_add(&g_launch_counts[family], 1, __ATOMIC_RELAXED);
}

}  // namespace tfgnn

extern "C" int tfgnn_launch_counts(int64_t* out_counts, int n) {
  if (!out_counts || n < 0) return TFGNN_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i)
    out_counts[i] = i < TFGNN_KFAM_COUNT ? __atomic_load_n(&tfgnn::g_launch_counts[i], __ATOMIC_RELAXED) : 0;
  return TFGNN_OK;
}

extern "C" const char* tfgnn_last_error(void) { return tfgnn::g_err; }
extern "C" const char* tfgnn_version(void) { return "tfgnn 0.1 gfx950"; }

namespace tfgnn {

// ------------------------------------------------------------------------------------------
// exclusive scan of int32 (device wide, 3-phase, recursive on block totals)
// ------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// out[i] = exclusive prefix within the tile; block_sums[b] = tile total
__global__ void __launch_bounds__(SCAN_THREADS)
scan_tiles_kernel(const int32_t* in, int32_t* out, int32_t* block_sums, int64_t n) {  // in may alias out
  __shared__ int32_t wave_tot[SCAN_THREADS / 64];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)tid * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    v[i] = idx < n ? in[idx] : 0;
    sum += v[i];
  }
  // inclusive scan of `sum` across the wave
  const int lane = tid & 63;
  int32_t incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  int32_t wave_off = 0;
  for (int w = 0; w < (tid >> 6); ++w) wave_off += wave_tot[w];
  int32_t excl = wave_off + incl - sum;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    int64_t idx = base + i;
    if (idx < n) out[idx] = excl;
    excl += v[i];
  }
  if (tid == SCAN_THREADS - 1 && block_sums) block_sums[blockIdx.x] = wave_off + incl;
}

__global__ void __launch_bounds__(SCAN_THREADS)
scan_add_offsets_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ block_offs, int64_t n) {
  const int32_t off = block_offs[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  for (int i = threadIdx.x; i < SCAN_TILE; i += SCAN_THREADS) {
    int64_t idx = base + i;
    if (idx < n) out[idx] += off;
  }
}

static size_t scan_scratch_elems(int64_t n) {
  size_t total = 0;
  while (n > SCAN_TILE) {
    n = ceil_div(n, SCAN_TILE);
    total += (size_t)n;
  }
  return total + 1;
}

// in/out may alias.  scratch: scan_scratch_elems(n) int32.
static int exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int32_t* scratch,
                              hipStream_t s) {
  if (n <= 0) return TFGNN_OK;
  int64_t nb = ceil_div(n, SCAN_TILE);
  if (nb == 1) {
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, (int32_t*)nullptr, n);
    TFGNN_LAUNCH_CHECK();
    return TFGNN_OK;
  }
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, scratch, n);
  TFGNN_LAUNCH_CHECK();
  int rc = exclusive_scan_i32(scratch, scratch, nb, scratch + nb, s);
  if (rc) return rc;
  hipLaunchKernelGGL(scan_add_offsets_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, out, scratch, n);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
struct EdgeLists {
  const int32_t* const* adj;  // device array of L device pointers
  const int64_t* edge_off;    // device [L+1]
  int L;
};

__device__ __forceinline__ int find_type(const int64_t* __restrict__ edge_off, int L, int64_t g) {
  int lo = 0, hi = L;  // edge_off[lo] <= g < edge_off[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (edge_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// ---- bucketing = one LSD radix sort per CSR (no global atomics, deterministic) -------------------
// composite key of an edge: (row << sec_bits) | sec, row = node * L + type (the bucket), sec = the
// node at the other end; payload = position of the edge in the concatenated adjacency lists.  Sorting
// by the composite puts edges in bucket order with ascending columns inside a bucket (canonical).
// Global memory atomics cost ~10 ns each on this chip (they execute at the memory side: the XCD L2s
// are not coherent), which made a counting sort with 4 atomics per edge take ~1 ms per batch.
constexpr int RS_THREADS = 256;
constexpr int RS_ROUNDS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;  // 4096 keys per workgroup
constexpr int RS_RADIX = 256;

__global__ void fill_keys_kernel(EdgeLists el, int64_t E, int64_t V, int sec_bits, uint64_t* __restrict__ comp_d,
                                 uint64_t* __restrict__ comp_s, uint32_t* __restrict__ pay,
                                 int32_t* __restrict__ err_flag) {
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < E; g += (int64_t)gridDim.x * blockDim.x) {
    const int l = find_type(el.edge_off, el.L, g);
    const int32_t* a = el.adj[l] + 2 * (g - el.edge_off[l]);
    int64_t src = a[0], dst = a[1];
    if (src < 0 || src >= V || dst < 0 || dst >= V) {
      atomicOr(err_flag, 1);  // rare path; the build is rejected in tfgnn_graph_wait
      src = 0;
      dst = 0;
    }
    comp_d[g] = ((uint64_t)(dst * el.L + l) << sec_bits) | (uint64_t)src;
    comp_s[g] = ((uint64_t)(src * el.L + l) << sec_bits) | (uint64_t)dst;
    pay[g] = (uint32_t)g;
  }
}

// hist[digit * nblocks + block] = number of keys of this workgroup's tile with that digit
__global__ void __launch_bounds__(RS_THREADS)
rs_hist_kernel(const uint64_t* __restrict__ comp, int64_t n, int shift, int32_t* __restrict__ hist, int nblocks) {
  __shared__ int32_t h[RS_RADIX];
  const int tid = threadIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t idx = base + r * RS_THREADS + tid;
    if (idx < n) atomicAdd(&h[(int)((comp[idx] >> shift) & (RS_RADIX - 1))], 1);  // LDS atomic
  }
  __syncthreads();
  hist[(int64_t)tid * nblocks + blockIdx.x] = h[tid];
}

// stable scatter: keys keep their input order inside a digit (tile order, then round, wave, lane)
__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(const uint64_t* __restrict__ comp_in, const uint32_t* __restrict__ pay_in, int64_t n, int shift,
                  const int32_t* __restrict__ offs, int nblocks, uint64_t* __restrict__ comp_out,
                  uint32_t* __restrict__ pay_out) {
  __shared__ int32_t base[RS_RADIX];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  base[tid] = offs[(int64_t)tid * nblocks + blockIdx.x];
  __syncthreads();
  const int64_t tile = (int64_t)blockIdx.x * RS_TILE;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t idx = tile + r * RS_THREADS + tid;
    const bool valid = idx < n;
    const uint64_t c = valid ? comp_in[idx] : 0;
    const uint32_t pl = valid ? pay_in[idx] : 0;
    const int d = (int)((c >> shift) & (RS_RADIX - 1));
    // lanes of this wave holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int cnt = __popcll(peers);
    int mybase = 0;
    for (int w = 0; w < RS_THREADS / 64; ++w) {  // waves take their turn: keeps the order stable
      if (wave == w && valid && rank == 0) {
        mybase = base[d];
        base[d] = mybase + cnt;
      }
      __syncthreads();
    }
    const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
    const int pos = __shfl(mybase, leader, 64) + rank;
    if (valid) {
      comp_out[pos] = c;
      pay_out[pos] = pl;
    }
  }
}

// rowptr[r] = first sorted position whose bucket is >= r
__global__ void rowptr_from_sorted_kernel(const uint64_t* __restrict__ comp, int64_t E, int sec_bits, int64_t R,
                                          int32_t* __restrict__ rowptr) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R; r += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = E;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)(comp[mid] >> sec_bits) < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
  }
}

// work items for rows longer than LONG_ROW_THRESHOLD (see graph.hpp / spmm.hip); counters = {items,
// multi-item rows, partial slots}.  The order in which rows claim their slots is irrelevant: a row's
// items are contiguous and are always combined in chunk order.
__global__ void plan_rows_kernel(const int32_t* __restrict__ rowptr, int64_t R, int long_threshold, int chunk_edges,
                                 int32_t* __restrict__ counters,
                                 int32_t* __restrict__ item_row, int32_t* __restrict__ item_chunk,
                                 int32_t* __restrict__ item_slot, int32_t* __restrict__ multi_row,
                                 int32_t* __restrict__ multi_base, int32_t* __restrict__ multi_n) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t len = rowptr[r + 1] - rowptr[r];
    if (len <= long_threshold) continue;
    const int32_t n = (len + chunk_edges - 1) / chunk_edges;
    const int32_t base = atomicAdd(&counters[0], n);
    int32_t pb = -1;
    if (n > 1) {
      const int32_t m = atomicAdd(&counters[1], 1);
      pb = atomicAdd(&counters[2], n);
      multi_row[m] = (int32_t)r;
      multi_base[m] = pb;
      multi_n[m] = n;
    }
    for (int32_t i = 0; i < n; ++i) {
      item_row[base + i] = (int32_t)r;
      item_chunk[base + i] = i;
      item_slot[base + i] = n > 1 ? pb + i : -1;
    }
  }
}

// ---- short rows ordered by length (CsrPlan::short_rows) ------------------------------------------
__global__ void __launch_bounds__(256)
short_hist_kernel(const int32_t* __restrict__ rowptr, int64_t R, int thr, int32_t* __restrict__ bin_count,
                  int32_t* __restrict__ num_short) {
  __shared__ int h[SHORT_BINS];
  __shared__ int n_short;
  for (int i = threadIdx.x; i <= thr; i += 256) h[i] = 0;
  if (threadIdx.x == 0) n_short = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const int32_t len = rowptr[r + 1] - rowptr[r];
    if (len <= thr) {
      atomicAdd(&h[len], 1);
      ++mine;
    }
  }
  if (mine) atomicAdd(&n_short, mine);
  __syncthreads();
  for (int i = threadIdx.x; i <= thr; i += 256)
    if (h[i]) atomicAdd(&bin_count[i], h[i]);
  if (threadIdx.x == 0 && n_short) atomicAdd(num_short, n_short);
}

// short_rows = rows with len <= thr in order of descending length (the order inside one length is arbitrary and
// has no effect on any result: every row is reduced on its own)
constexpr int SHORT_TILE_ROWS = 8;  // rows per thread and tile: one global atomic per (tile, length) pair
__global__ void __launch_bounds__(256)
short_scatter_kernel(const int32_t* __restrict__ rowptr, int64_t R, int thr, const int32_t* __restrict__ bin_count,
                     int32_t* __restrict__ bin_cursor, int32_t* __restrict__ short_rows) {
  __shared__ int h[SHORT_BINS], base[SHORT_BINS], start[SHORT_BINS];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int i = thr; i >= 0; --i) {
      start[i] = acc;
      acc += bin_count[i];
    }
  }
  constexpr int TILE = 256 * SHORT_TILE_ROWS;
  for (int64_t tile = blockIdx.x; tile * TILE < R; tile += gridDim.x) {
    for (int i = threadIdx.x; i <= thr; i += 256) h[i] = 0;
    __syncthreads();
    int32_t len[SHORT_TILE_ROWS];
    int rank[SHORT_TILE_ROWS];
#pragma unroll
    for (int k = 0; k < SHORT_TILE_ROWS; ++k) {
      const int64_t r = tile * TILE + k * 256 + threadIdx.x;
      len[k] = r < R ? rowptr[r + 1] - rowptr[r] : -1;
      rank[k] = (len[k] >= 0 && len[k] <= thr) ? atomicAdd(&h[len[k]], 1) : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= thr; i += 256)
      if (h[i]) base[i] = atomicAdd(&bin_cursor[i], h[i]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SHORT_TILE_ROWS; ++k)
      if (len[k] >= 0 && len[k] <= thr)
        short_rows[start[len[k]] + base[len[k]] + rank[k]] = (int32_t)(tile * TILE + k * 256 + threadIdx.x);
    __syncthreads();
  }
}

// ---- non-empty buckets, type-major (CompactBuckets) ---------------------------------------------
// flags in type-major layout: i = l * V + v ; flags[R] = 0 ; node_cnt[v] = non-empty buckets of v
__global__ void nz_flags_kernel(const int32_t* __restrict__ rowptr, int64_t V, int L, int32_t* __restrict__ flags,
                                int32_t* __restrict__ node_cnt) {
  const int64_t R = V * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= R; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < R) {
      const int64_t l = i / V, v = i - l * V;
      const int64_t r = v * L + l;
      flags[i] = rowptr[r + 1] > rowptr[r] ? 1 : 0;
    } else {
      flags[i] = 0;
    }
    if (i <= V) {
      int32_t c = 0;
      if (i < V)
        for (int l = 0; l < L; ++l) c += rowptr[i * L + l + 1] > rowptr[i * L + l] ? 1 : 0;
      node_cnt[i] = c;
    }
  }
}

__global__ void nz_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ scan_t, int64_t V,
                               int L, int32_t* __restrict__ cpos, int32_t* __restrict__ nzrow,
                               int32_t* __restrict__ nz_node, int32_t* __restrict__ nz_off) {
  const int64_t R = V * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t l = i / V, v = i - l * V;
    const int64_t r = v * L + l;
    const int32_t c = scan_t[i];
    if (rowptr[r + 1] > rowptr[r]) {
      cpos[r] = c;
      nzrow[c] = (int32_t)r;
      nz_node[c] = (int32_t)v;
    } else {
      cpos[r] = -1;
    }
    if (v == 0) nz_off[l] = c;
    if (i == R - 1) nz_off[L] = scan_t[R];
  }
}

__global__ void nz_cols_kernel(const int32_t* __restrict__ cpos, const int32_t* __restrict__ nodeptr_nz, int64_t V,
                               int L, int32_t* __restrict__ col_nz) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (int64_t)gridDim.x * blockDim.x) {
    int32_t j = nodeptr_nz[v];
    for (int l = 0; l < L; ++l) {
      const int32_t c = cpos[v * L + l];
      if (c >= 0) col_nz[j++] = c;
    }
  }
}

__global__ void invdeg_rows_kernel(const int32_t* __restrict__ rowptr_d, int64_t R, int L,
                                   float* __restrict__ invdeg_d, int32_t* __restrict__ nodeptr_d,
                                   const int32_t* __restrict__ rowptr_s, int32_t* __restrict__ nodeptr_s) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R;
       r += (int64_t)gridDim.x * blockDim.x) {
    if (r < R) {
      int32_t len = rowptr_d[r + 1] - rowptr_d[r];
      // gnn_edge_mlp.py:102-106: 1.0 / (num_incoming + SMALL_NUMBER), evaluated in fp32 like TF
      invdeg_d[r] = len > 0 ? 1.0f / ((float)len + kSmallNumber) : 0.f;
    }
    if (r % L == 0) {
      nodeptr_d[r / L] = rowptr_d[r];
      nodeptr_s[r / L] = rowptr_s[r];
    }
  }
}

__global__ void unpack_kernel(const uint64_t* __restrict__ comp, const uint32_t* __restrict__ pay, int sec_bits,
                              int64_t E, int L, int32_t* __restrict__ col, int32_t* __restrict__ eid,
                              int32_t* __restrict__ coll, const float* __restrict__ invdeg_d,
                              float* __restrict__ invdeg_edge, int by_src,
                              int32_t* __restrict__ eid_to_pos, int32_t* __restrict__ row_node) {
  const uint64_t sec_mask = ((uint64_t)1 << sec_bits) - 1;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E;
       p += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t c = comp[p];
    const int32_t other = (int32_t)(c & sec_mask);
    const int32_t id = (int32_t)pay[p];
    const int32_t row = (int32_t)(c >> sec_bits);
    const int l = row % L;
    col[p] = other;
    eid[p] = id;
    const int64_t cl = (int64_t)other * L + l;
    coll[p] = (int32_t)cl;
    if (row_node) row_node[p] = row / L;
    // the degree that normalises an edge is always the in-degree of its TARGET for its type
    invdeg_edge[p] = by_src ? invdeg_d[cl] : invdeg_d[row];
    if (eid_to_pos) eid_to_pos[id] = (int32_t)p;
  }
}

__global__ void src2dst_kernel(const int32_t* __restrict__ eid_s, const int32_t* __restrict__ eid_to_pos_d,
                               int64_t E, int32_t* __restrict__ src2dst, int32_t* __restrict__ dst2src) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int32_t q = eid_to_pos_d[eid_s[p]];
    src2dst[p] = q;
    dst2src[q] = (int32_t)p;  // the inverse permutation (RGAT writes its attention weights in both edge orders)
  }
}

}  // namespace tfgnn

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------

namespace tfgnn {
// Small cache of device / pinned-host blocks so that building a graph per batch does not pay a
// hipMalloc + hipFree (both synchronise the device) every step.
struct PoolEntry {
  void* ptr;
  size_t bytes;
  hipEvent_t ready;  // nullable: work that may still be using the block (recorded at release)
};
static std::mutex g_pool_mutex;
static std::vector<PoolEntry> g_dev_pool, g_pinned_pool;
constexpr size_t POOL_MAX_ENTRIES = 12;

static void* pool_take(std::vector<PoolEntry>& pool, size_t bytes, size_t* got, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  int best = -1;
  for (int i = 0; i < (int)pool.size(); ++i)
    if (pool[i].bytes >= bytes && pool[i].bytes <= 2 * bytes + 4096 && (best < 0 || pool[i].bytes < pool[best].bytes)) best = i;
  if (best < 0) return nullptr;
  void* p = pool[best].ptr;
  *got = pool[best].bytes;
  if (pool[best].ready) {  // the new owner's stream waits for the previous owner's last use
    (void)hipStreamWaitEvent(s, pool[best].ready, 0);
    (void)hipEventDestroy(pool[best].ready);
  }
  pool.erase(pool.begin() + best);
  return p;
}

static hipError_t dev_alloc(void** p, size_t bytes, size_t* got, hipStream_t s) {
  if (bytes == 0) bytes = 256;
  *p = pool_take(g_dev_pool, bytes, got, s);
  if (*p) return hipSuccess;
  *got = bytes;
  return hipMalloc(p, bytes);
}
// last_use: stream on which the block may still be in use (nullptr + sync=false: known idle)
static void dev_release(void* p, size_t bytes, hipStream_t last_use = nullptr, bool in_use = false) {
  if (!p) return;
  hipEvent_t ev = nullptr;
  if (in_use) {
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, last_use) != hipSuccess) {
      (void)hipStreamSynchronize(last_use);
      if (ev) (void)hipEventDestroy(ev);
      ev = nullptr;
    }
  }
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_dev_pool.size() < POOL_MAX_ENTRIES) {
      g_dev_pool.push_back({p, bytes, ev});
      return;
    }
  }
  if (ev) (void)hipEventDestroy(ev);
  (void)hipFree(p);  // synchronises
}
static hipError_t pinned_alloc(void** p, size_t bytes, size_t* got) {
  if (bytes < 4096) bytes = 4096;
  *p = pool_take(g_pinned_pool, bytes, got, nullptr);
  if (*p) return hipSuccess;
  *got = bytes;
  return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
static void pinned_release(void* p, size_t bytes) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (g_pinned_pool.size() < POOL_MAX_ENTRIES) {
      g_pinned_pool.push_back({p, bytes, nullptr});
      return;
    }
  }
  (void)hipHostFree(p);
}
}  // namespace tfgnn

namespace {
struct SlabPlan {
  size_t total = 0;
  size_t take(size_t bytes) {
    size_t off = total;
    total += (bytes + 255) & ~(size_t)255;
    return off;
  }
};
}  // namespace

static void graph_release_all(tfgnn_graph* g, hipStream_t last_use = nullptr, bool in_use = false) {
  using namespace tfgnn;
  if (!g) return;
  if (g->event) (void)hipEventDestroy((hipEvent_t)g->event);
  dev_release(g->scratch, g->scratch_bytes, last_use, in_use);
  dev_release(g->slab, g->slab_bytes, last_use, in_use);
  pinned_release(g->pinned, g->pinned_bytes);
  delete g;
}

extern "C" int tfgnn_graph_create_async(int num_edge_types, int64_t num_nodes,
                                        const int32_t* const* d_adjacency, const int64_t* num_edges,
                                        void* stream, tfgnn_graph** out_graph) {
  using namespace tfgnn;
  TFGNN_REQUIRE(out_graph != nullptr, "out_graph is NULL");
  *out_graph = nullptr;
  TFGNN_REQUIRE(num_edge_types >= 0 && num_nodes >= 0, "negative sizes");
  TFGNN_REQUIRE(num_edge_types == 0 || (d_adjacency && num_edges), "adjacency arrays are NULL");
  hipStream_t s = (hipStream_t)stream;
  const int L = num_edge_types;
  const int64_t V = num_nodes;
  std::vector<int64_t> edge_off(L + 1, 0);
  for (int l = 0; l < L; ++l) {
    TFGNN_REQUIRE(num_edges[l] >= 0, "num_edges[%d] < 0", l);
    TFGNN_REQUIRE(num_edges[l] == 0 || d_adjacency[l] != nullptr, "adjacency list %d is NULL", l);
    edge_off[l + 1] = edge_off[l] + num_edges[l];
  }
  const int64_t E = edge_off[L];
  const int64_t R = V * (int64_t)L;
  TFGNN_REQUIRE(R < ((int64_t)1 << 31) - 1 && E < ((int64_t)1 << 31) - 1,
                "graph too large for int32 indexing (V*L=%lld, E=%lld)", (long long)R, (long long)E);

  tfgnn_graph* g = new tfgnn_graph();
  g->L = L;
  g->V = V;
  g->E = E;
  g->R = R;

  // persistent arrays
  SlabPlan plan;
  const size_t o_rowptr_d = plan.take((R + 1) * 4), o_rowptr_s = plan.take((R + 1) * 4);
  const size_t o_col_d = plan.take(E * 4), o_eid_d = plan.take(E * 4), o_coll_d = plan.take(E * 4);
  const size_t o_col_s = plan.take(E * 4), o_eid_s = plan.take(E * 4), o_coll_s = plan.take(E * 4);
  const size_t o_nodeptr_d = plan.take((V + 1) * 4), o_nodeptr_s = plan.take((V + 1) * 4);
  const size_t o_src2dst = plan.take(E * 4), o_dst2src = plan.take(E * 4);
  const size_t o_tgt_d = plan.take(E * 4);
  const size_t o_invdeg_d = plan.take((R + 1) * 4);
  const size_t o_invdeg_es = plan.take(E * 4), o_invdeg_ed = plan.take(E * 4);
  size_t o_cb[2][6];
  for (int side = 0; side < 2; ++side) {
    o_cb[side][0] = plan.take((R + 1) * 4);            // cpos
    o_cb[side][1] = plan.take((R + 1) * 4);            // nzrow
    o_cb[side][2] = plan.take((R + 1) * 4);            // nz_node
    o_cb[side][3] = plan.take((size_t)(L + 1) * 4);    // nz_off
    o_cb[side][4] = plan.take((V + 1) * 4);            // nodeptr_nz
    o_cb[side][5] = plan.take((R + 1) * 4);            // col_nz
  }
  // long-row plan parameters per view (tools/gather_probe.py sweeps at cfg-2, rows ordered by length): typed
  // views 91 us at (48, 512) vs 104 us at (16, 192) (134 us in natural row order at (16, 128)); the node views -
  // all edge types of a node in one row - 121 us at (32, 512) vs 128 us at (64, 512) (138 us in natural order).  TFGNN_LONG_ROW / TFGNN_ITEM_CHUNK override the typed views, TFGNN_LONG_ROW_NODE /
  // TFGNN_ITEM_CHUNK_NODE the node views, for probing.
  static const int env_long = [] { const char* e = getenv("TFGNN_LONG_ROW"); return e ? atoi(e) : 0; }();
  static const int env_chunk = [] { const char* e = getenv("TFGNN_ITEM_CHUNK"); return e ? atoi(e) : 0; }();
  static const int env_long_n = [] { const char* e = getenv("TFGNN_LONG_ROW_NODE"); return e ? atoi(e) : 0; }();
  static const int env_chunk_n = [] { const char* e = getenv("TFGNN_ITEM_CHUNK_NODE"); return e ? atoi(e) : 0; }();
  int view_long[4] = {LONG_ROW_THRESHOLD_TYPED, LONG_ROW_THRESHOLD, LONG_ROW_THRESHOLD_TYPED, LONG_ROW_THRESHOLD};
  int view_chunk[4] = {ITEM_CHUNK_TYPED, ITEM_CHUNK, ITEM_CHUNK_TYPED, ITEM_CHUNK};
  for (int v = 0; v < 4; ++v) {
    const bool node_view = v & 1;
    if ((node_view ? env_long_n : env_long) > 0) view_long[v] = node_view ? env_long_n : env_long;
    if ((node_view ? env_chunk_n : env_chunk) > 0) view_chunk[v] = node_view ? env_chunk_n : env_chunk;
    if (view_long[v] > SHORT_BINS - 1) view_long[v] = SHORT_BINS - 1;
    if (view_chunk[v] < view_long[v]) view_chunk[v] = view_long[v];
  }
  const int min_long = std::min(std::min(view_long[0], view_long[1]), std::min(view_long[2], view_long[3]));
  const int min_chunk = std::min(std::min(view_chunk[0], view_chunk[1]), std::min(view_chunk[2], view_chunk[3]));
  const size_t max_items = (size_t)(E / min_long + 1), max_multi = (size_t)(E / min_chunk + 1);
  size_t o_item[4][3], o_multi[4][3];
  for (int v = 0; v < 4; ++v) {
    for (int k = 0; k < 3; ++k) o_item[v][k] = plan.take(max_items * 4);
    for (int k = 0; k < 3; ++k) o_multi[v][k] = plan.take(max_multi * 4);
  }
  size_t o_short[4];
  for (int v = 0; v < 4; ++v) o_short[v] = plan.take((size_t)((v & 1) ? V : R) * 4 + 4);
  const size_t persistent = plan.total;
  // build-time scratch (freed with a second allocation)
  SlabPlan tmp;
  const int rs_blocks = (int)ceil_div(E > 0 ? E : 1, RS_TILE);
  const size_t t_comp_d = tmp.take(E * 8), t_comp_s = tmp.take(E * 8);
  const size_t t_comp_alt_d = tmp.take(E * 8), t_comp_alt_s = tmp.take(E * 8);
  const size_t t_pay_a = tmp.take(E * 4), t_pay_d0 = tmp.take(E * 4), t_pay_d1 = tmp.take(E * 4);
  const size_t t_pay_s0 = tmp.take(E * 4), t_pay_s1 = tmp.take(E * 4);
  const size_t t_hist = tmp.take((size_t)RS_RADIX * rs_blocks * 4 + 16);
  const size_t t_counters = tmp.take(64 * 4);
  const size_t t_bins = tmp.take((size_t)4 * 2 * SHORT_BINS * 4);
  const size_t t_scan = tmp.take((scan_scratch_elems(R + 1) + scan_scratch_elems((int64_t)RS_RADIX * rs_blocks)) * 4 + 16);
  const size_t t_eid2pos = tmp.take(E * 4);
  const size_t t_ptrs = tmp.take((size_t)(L + 1) * 8), t_off = tmp.take((size_t)(L + 1) * 8);
  const size_t t_flags = tmp.take((R + 2) * 4);

  char* slab = nullptr;
  char* scratch = nullptr;
  TFGNN_REQUIRE(L <= 256, "at most 256 edge types are supported (got %d)", L);
  const size_t pinned_need = 256 + (size_t)(L + 1) * 16 + (size_t)(L + 1) * 8;
  hipError_t he = dev_alloc((void**)&slab, persistent, &g->slab_bytes, s);
  if (he == hipSuccess) {
    g->slab = slab;
    he = dev_alloc((void**)&scratch, tmp.total, &g->scratch_bytes, s);
  }
  if (he == hipSuccess) {
    g->scratch = scratch;
    he = pinned_alloc(&g->pinned, pinned_need, &g->pinned_bytes);
  }
  if (he == hipSuccess) {
    hipEvent_t ev;
    he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (he == hipSuccess) g->event = ev;
  }
  if (he != hipSuccess) {
    set_error("graph allocation failed (%zu + %zu bytes): %s", persistent, tmp.total, hipGetErrorString(he));
    graph_release_all(g);
    return TFGNN_ERR_HIP;
  }
  g->slab = slab;
  g->rowptr_d = (int32_t*)(slab + o_rowptr_d);
  g->rowptr_s = (int32_t*)(slab + o_rowptr_s);
  g->col_d = (int32_t*)(slab + o_col_d);
  g->eid_d = (int32_t*)(slab + o_eid_d);
  g->coll_d = (int32_t*)(slab + o_coll_d);
  g->col_s = (int32_t*)(slab + o_col_s);
  g->eid_s = (int32_t*)(slab + o_eid_s);
  g->coll_s = (int32_t*)(slab + o_coll_s);
  g->nodeptr_d = (int32_t*)(slab + o_nodeptr_d);
  g->nodeptr_s = (int32_t*)(slab + o_nodeptr_s);
  g->src2dst = (int32_t*)(slab + o_src2dst);
  g->dst2src = (int32_t*)(slab + o_dst2src);
  g->tgt_d = (int32_t*)(slab + o_tgt_d);
  g->invdeg_d = (float*)(slab + o_invdeg_d);
  g->invdeg_edge_s = (float*)(slab + o_invdeg_es);
  g->invdeg_edge_d = (float*)(slab + o_invdeg_ed);
  for (int side = 0; side < 2; ++side) {
    CompactBuckets& cb = g->compact[side];
    cb.cpos = (int32_t*)(slab + o_cb[side][0]);
    cb.nzrow = (int32_t*)(slab + o_cb[side][1]);
    cb.nz_node = (int32_t*)(slab + o_cb[side][2]);
    cb.nz_off = (int32_t*)(slab + o_cb[side][3]);
    cb.nodeptr_nz = (int32_t*)(slab + o_cb[side][4]);
    cb.col_nz = (int32_t*)(slab + o_cb[side][5]);
  }

  uint64_t* comp_d = (uint64_t*)(scratch + t_comp_d);
  uint64_t* comp_s = (uint64_t*)(scratch + t_comp_s);
  uint64_t* comp_alt_d = (uint64_t*)(scratch + t_comp_alt_d);
  uint64_t* comp_alt_s = (uint64_t*)(scratch + t_comp_alt_s);
  uint32_t* pay_a = (uint32_t*)(scratch + t_pay_a);
  uint32_t* pay_d0 = (uint32_t*)(scratch + t_pay_d0);
  uint32_t* pay_d1 = (uint32_t*)(scratch + t_pay_d1);
  uint32_t* pay_s0 = (uint32_t*)(scratch + t_pay_s0);
  uint32_t* pay_s1 = (uint32_t*)(scratch + t_pay_s1);
  int32_t* hist = (int32_t*)(scratch + t_hist);
  int32_t* counters = (int32_t*)(scratch + t_counters);
  int32_t* scan_tmp = (int32_t*)(scratch + t_scan);
  int32_t* eid2pos = (int32_t*)(scratch + t_eid2pos);
  const int32_t** d_ptrs = (const int32_t**)(scratch + t_ptrs);
  int64_t* d_off = (int64_t*)(scratch + t_off);

  int rc = TFGNN_OK;
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(s);
    graph_release_all(g);
    return code;
  };
#define G_CHECK(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);    \
      return fail(TFGNN_ERR_HIP);                                                              \
    }                                                                                          \
  } while (0)

  G_CHECK(hipMemsetAsync(counters, 0, 64 * 4, s));
  int32_t* bins = (int32_t*)(scratch + t_bins);
  G_CHECK(hipMemsetAsync(bins, 0, (size_t)4 * 2 * SHORT_BINS * 4, s));
  // pointer / offset tables go through the handle's pinned staging block: no host synchronisation
  {
    char* hp = (char*)g->pinned + 256;
    const int32_t** h_ptrs = (const int32_t**)hp;
    int64_t* h_off = (int64_t*)(hp + (size_t)(L + 1) * 8);
    for (int l = 0; l < L; ++l) h_ptrs[l] = d_adjacency[l];
    for (int l = 0; l <= L; ++l) h_off[l] = edge_off[l];
    if (L > 0) G_CHECK(hipMemcpyAsync(d_ptrs, h_ptrs, (size_t)L * 8, hipMemcpyHostToDevice, s));
    G_CHECK(hipMemcpyAsync(d_off, h_off, (size_t)(L + 1) * 8, hipMemcpyHostToDevice, s));
  }

  EdgeLists el{d_ptrs, d_off, L};
  const int threads = 256;
  auto blocks_for = [&](int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, threads), 8192)); };

  // sort the edges into (bucket, column) order, once per bucketing
  int sec_bits = 1;
  while (((int64_t)1 << sec_bits) < V) ++sec_bits;
  int row_bits = 1;
  while (((int64_t)1 << row_bits) < (R > 0 ? R : 1)) ++row_bits;
  const uint64_t* sorted_d = comp_d;
  const uint64_t* sorted_s = comp_s;
  const uint32_t* pay_d = pay_a;
  const uint32_t* pay_s = pay_a;
  if (E > 0) {
    hipLaunchKernelGGL(fill_keys_kernel, dim3(blocks_for(E)), dim3(threads), 0, s, el, E, V, sec_bits, comp_d, comp_s,
                       pay_a, counters + 2);
    const int total_bits = sec_bits + row_bits;
    for (int side = 0; side < 2; ++side) {
      // private ping-pong buffers per side; both sorts start from the shared identity payload pay_a
      uint64_t* cbuf[2] = {side == 0 ? comp_d : comp_s, side == 0 ? comp_alt_d : comp_alt_s};
      uint32_t* pbuf[2] = {side == 0 ? pay_d0 : pay_s0, side == 0 ? pay_d1 : pay_s1};
      const uint64_t* cin = cbuf[0];
      const uint32_t* pin = pay_a;
      int pass = 0;
      for (int shift = 0; shift < total_bits; shift += 8, ++pass) {
        uint64_t* cout = cbuf[(pass + 1) & 1];
        uint32_t* pout = pbuf[pass & 1];
        hipLaunchKernelGGL(rs_hist_kernel, dim3(rs_blocks), dim3(RS_THREADS), 0, s, cin, E, shift, hist, rs_blocks);
        rc = exclusive_scan_i32(hist, hist, (int64_t)RS_RADIX * rs_blocks, scan_tmp, s);
        if (rc) return fail(rc);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(rs_blocks), dim3(RS_THREADS), 0, s, cin, pin, E, shift, hist,
                           rs_blocks, cout, pout);
        cin = cout;
        pin = pout;
      }
      if (side == 0) {
        sorted_d = cin;
        pay_d = pin;
      } else {
        sorted_s = cin;
        pay_s = pin;
      }
    }
    hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(blocks_for(R + 1)), dim3(threads), 0, s, sorted_d, E, sec_bits,
                       R, g->rowptr_d);
    hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(blocks_for(R + 1)), dim3(threads), 0, s, sorted_s, E, sec_bits,
                       R, g->rowptr_s);
  } else {
    G_CHECK(hipMemsetAsync(g->rowptr_d, 0, (R + 1) * 4, s));
    G_CHECK(hipMemsetAsync(g->rowptr_s, 0, (R + 1) * 4, s));
  }
  if (L > 0) {
    hipLaunchKernelGGL(invdeg_rows_kernel, dim3(blocks_for(R + 1)), dim3(threads), 0, s, g->rowptr_d, R, L,
                       g->invdeg_d, g->nodeptr_d, g->rowptr_s, g->nodeptr_s);
  } else {
    G_CHECK(hipMemsetAsync(g->nodeptr_d, 0, (V + 1) * 4, s));
    G_CHECK(hipMemsetAsync(g->nodeptr_s, 0, (V + 1) * 4, s));
  }
  if (E > 0) {
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(E)), dim3(threads), 0, s, sorted_d, pay_d, sec_bits, E, L,
                       g->col_d, g->eid_d, g->coll_d, g->invdeg_d, g->invdeg_edge_d, 0, eid2pos, g->tgt_d);
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks_for(E)), dim3(threads), 0, s, sorted_s, pay_s, sec_bits, E, L,
                       g->col_s, g->eid_s, g->coll_s, g->invdeg_d, g->invdeg_edge_s, 1, (int32_t*)nullptr,
                       (int32_t*)nullptr);
    hipLaunchKernelGGL(src2dst_kernel, dim3(blocks_for(E)), dim3(threads), 0, s, g->eid_s, eid2pos, E,
                       g->src2dst, g->dst2src);
  }
  // non-empty buckets in type-major order, for both bucketings
  {
    int32_t* flags = (int32_t*)(scratch + t_flags);
    for (int side = 0; side < 2; ++side) {
      CompactBuckets& cb = g->compact[side];
      const int32_t* rp = side == 0 ? g->rowptr_d : g->rowptr_s;
      if (L > 0) {
        hipLaunchKernelGGL(nz_flags_kernel, dim3(blocks_for(R + 1)), dim3(threads), 0, s, rp, V, L, flags, cb.nodeptr_nz);
        rc = exclusive_scan_i32(flags, flags, R + 1, scan_tmp, s);
        if (rc) return fail(rc);
        rc = exclusive_scan_i32(cb.nodeptr_nz, cb.nodeptr_nz, V + 1, scan_tmp, s);
        if (rc) return fail(rc);
        if (R > 0) {
          hipLaunchKernelGGL(nz_fill_kernel, dim3(blocks_for(R)), dim3(threads), 0, s, rp, flags, V, L, cb.cpos, cb.nzrow,
                             cb.nz_node, cb.nz_off);
          hipLaunchKernelGGL(nz_cols_kernel, dim3(blocks_for(V)), dim3(threads), 0, s, cb.cpos, cb.nodeptr_nz, V, L,
                             cb.col_nz);
        } else {
          G_CHECK(hipMemsetAsync(cb.nz_off, 0, (size_t)(L + 1) * 4, s));
        }
      } else {
        G_CHECK(hipMemsetAsync(cb.nz_off, 0, (size_t)(L + 1) * 4, s));
        G_CHECK(hipMemsetAsync(cb.nodeptr_nz, 0, (V + 1) * 4, s));
      }
      G_CHECK(hipMemcpyAsync((char*)g->pinned + 256 + (size_t)(L + 1) * 16 + (size_t)side * (L + 1) * 4, cb.nz_off,
                             (size_t)(L + 1) * 4, hipMemcpyDeviceToHost, s));
    }
  }
  // gather views + long-row plans (tfgnn_graph_view order)
  g->views[0].rowptr = g->rowptr_d;  g->views[0].num_rows = R; g->views[0].col = g->col_d;
  g->views[1].rowptr = g->nodeptr_d; g->views[1].num_rows = V; g->views[1].col = g->coll_d;
  g->views[2].rowptr = g->rowptr_s;  g->views[2].num_rows = R; g->views[2].col = g->col_s;
  g->views[3].rowptr = g->nodeptr_s; g->views[3].num_rows = V; g->views[3].col = g->coll_s;
  for (int v = 0; v < 4; ++v) {
    CsrPlan& pl = g->views[v].plan;
    pl.item_row = (int32_t*)(slab + o_item[v][0]);
    pl.item_chunk = (int32_t*)(slab + o_item[v][1]);
    pl.item_slot = (int32_t*)(slab + o_item[v][2]);
    pl.multi_row = (int32_t*)(slab + o_multi[v][0]);
    pl.multi_base = (int32_t*)(slab + o_multi[v][1]);
    pl.multi_n = (int32_t*)(slab + o_multi[v][2]);
    pl.long_threshold = view_long[v];
    pl.item_chunk_edges = view_chunk[v];
    pl.short_rows = (int32_t*)(slab + o_short[v]);
    if (g->views[v].num_rows > 0) {
      int32_t* bc = bins + (size_t)v * 2 * SHORT_BINS;
      const int64_t nr = g->views[v].num_rows;
      const unsigned hist_blocks = (unsigned)std::min<int64_t>(ceil_div(nr, 256), 1024);
      const unsigned scat_blocks = (unsigned)std::min<int64_t>(ceil_div(nr, 256 * SHORT_TILE_ROWS), 4096);
      hipLaunchKernelGGL(short_hist_kernel, dim3(hist_blocks), dim3(256), 0, s, g->views[v].rowptr, nr, view_long[v], bc,
                         counters + 16 + 4 * v + 3);
      hipLaunchKernelGGL(short_scatter_kernel, dim3(scat_blocks), dim3(256), 0, s, g->views[v].rowptr, nr, view_long[v],
                         bc, bc + SHORT_BINS, pl.short_rows);
    }
    if (E > 0 && g->views[v].num_rows > 0) {
      hipLaunchKernelGGL(plan_rows_kernel, dim3(blocks_for(g->views[v].num_rows)), dim3(threads), 0, s,
                         g->views[v].rowptr, g->views[v].num_rows, view_long[v], view_chunk[v], counters + 16 + 4 * v, pl.item_row,
                         pl.item_chunk, pl.item_slot, pl.multi_row, pl.multi_base, pl.multi_n);
    }
  }
  {
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
      set_error("graph build kernel launch failed: %s", hipGetErrorString(le));
      return fail(TFGNN_ERR_HIP);
    }
  }
  G_CHECK(hipMemcpyAsync(g->pinned, counters, 64 * 4, hipMemcpyDeviceToHost, s));
  G_CHECK(hipEventRecord((hipEvent_t)g->event, s));
#undef G_CHECK
  g->pending = true;
  *out_graph = g;
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_wait(tfgnn_graph* g) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  if (!g->pending) return TFGNN_OK;
  TFGNN_HIP_CHECK(hipEventSynchronize((hipEvent_t)g->event));
  g->pending = false;
  const int32_t* h_counters = (const int32_t*)g->pinned;
  for (int v = 0; v < 4; ++v) {
    g->views[v].plan.num_items = h_counters[16 + 4 * v + 0];
    g->views[v].plan.num_multi = h_counters[16 + 4 * v + 1];
    g->views[v].plan.num_partials = h_counters[16 + 4 * v + 2];
    g->views[v].plan.num_short = h_counters[16 + 4 * v + 3];
  }
  for (int side = 0; side < 2; ++side) {
    const int32_t* h = (const int32_t*)((const char*)g->pinned + 256 + (size_t)(g->L + 1) * 16 + (size_t)side * (g->L + 1) * 4);
    for (int l = 0; l <= g->L; ++l) g->compact[side].h_nz_off[l] = h[l];
    g->compact[side].num_nz = h[g->L];
  }
  const bool bad_index = h_counters[2] != 0;
  dev_release(g->scratch, g->scratch_bytes);  // build-time scratch is no longer needed
  g->scratch = nullptr;
  if (bad_index) {
    set_error("adjacency list contains a node index outside [0, %lld)", (long long)g->V);
    return TFGNN_ERR_OUT_OF_RANGE;
  }
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_create(int num_edge_types, int64_t num_nodes,
                                  const int32_t* const* d_adjacency, const int64_t* num_edges,
                                  void* stream, tfgnn_graph** out_graph) {
  int rc = tfgnn_graph_create_async(num_edge_types, num_nodes, d_adjacency, num_edges, stream, out_graph);
  if (rc) return rc;
  rc = tfgnn_graph_wait(*out_graph);
  if (rc) {
    graph_release_all(*out_graph);
    *out_graph = nullptr;
  }
  return rc;
}


namespace tfgnn {
// mode: 0 = sum, 1 = mean (1/max(N_v,1)), 2 = sqrt_n (1/sqrt(max(N_v,1))); N_v = all in-edges of v
__device__ __forceinline__ float node_mult(const int32_t* __restrict__ nodeptr_d, int64_t v, int mode) {
  if (mode == 0) return 1.f;
  float n = (float)(nodeptr_d[v + 1] - nodeptr_d[v]);
  n = n < 1.f ? 1.f : n;
  return mode == 1 ? 1.f / n : 1.f / sqrtf(n);
}

__global__ void graph_scales_kernel(const int32_t* __restrict__ nodeptr_d, const float* __restrict__ invdeg_d,
                                    const float* __restrict__ invdeg_edge_s, const float* __restrict__ invdeg_edge_d,
                                    const int32_t* __restrict__ col_s, int64_t V, int L, int64_t E,
                                    int normalize, int mode, float* __restrict__ row_scale,
                                    float* __restrict__ node_scale, float* __restrict__ ew_s,
                                    float* __restrict__ ew_d) {
  const int64_t R = V * L;
  const int64_t n = R > E ? R : E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < R) {
      const int64_t v = i / L;
      const float m = node_mult(nodeptr_d, v, mode);
      row_scale[i] = (normalize ? invdeg_d[i] : 1.f) * m;
      if (i % L == 0) node_scale[v] = m;
    }
    if (i < E) {
      ew_s[i] = (normalize ? invdeg_edge_s[i] : 1.f) * node_mult(nodeptr_d, col_s[i], mode);
      ew_d[i] = normalize ? invdeg_edge_d[i] : 1.f;
    }
  }
}

__global__ void target_multiplier_kernel(const int32_t* __restrict__ rowptr_d, const float* __restrict__ row_scale,
                                         int64_t R, int L, float* __restrict__ k, int32_t* __restrict__ ident_ptr,
                                         int32_t* __restrict__ node_of_row) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R; r += (int64_t)gridDim.x * blockDim.x) {
    ident_ptr[r] = (int32_t)r;
    if (r < R) {
      const float c = (float)(rowptr_d[r + 1] - rowptr_d[r]);
      k[r] = row_scale ? c * row_scale[r] : c;
      node_of_row[r] = (int32_t)(r / L);
    }
  }
}
}  // namespace tfgnn

extern "C" int tfgnn_graph_scales(const tfgnn_graph* g, int normalize_by_num_incoming, int aggregation_mode,
                                  float* d_row_scale, float* d_node_scale, float* d_edge_weight_by_src,
                                  float* d_edge_weight_by_dst, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  TFGNN_REQUIRE(aggregation_mode >= 0 && aggregation_mode <= 2, "unknown aggregation mode %d", aggregation_mode);
  const int64_t R = g->R, E = g->E;
  const int64_t n = R > E ? R : E;
  if (n == 0) return TFGNN_OK;
  TFGNN_REQUIRE((R == 0 || (d_row_scale && d_node_scale)) && (E == 0 || (d_edge_weight_by_src && d_edge_weight_by_dst)),
                "NULL output");
  unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256), 8192);
  hipLaunchKernelGGL(graph_scales_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g->nodeptr_d,
                     g->invdeg_d, g->invdeg_edge_s, g->invdeg_edge_d, g->col_s, g->V, g->L, E,
                     normalize_by_num_incoming, aggregation_mode, d_row_scale, d_node_scale,
                     d_edge_weight_by_src, d_edge_weight_by_dst);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_target_multiplier(const tfgnn_graph* g, const float* d_row_scale, float* d_k,
                                             int32_t* d_ident_ptr, int32_t* d_node_of_row, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr && d_ident_ptr != nullptr, "NULL argument");
  TFGNN_REQUIRE(g->R == 0 || (d_k && d_node_of_row), "NULL output");
  unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(g->R + 1, 256), 8192);
  hipLaunchKernelGGL(target_multiplier_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g->rowptr_d,
                     d_row_scale, g->R, g->L > 0 ? g->L : 1, d_k, d_ident_ptr, d_node_of_row);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_destroy(tfgnn_graph* graph) {
  if (!graph) return TFGNN_OK;
  // no stream given: anything may still be reading the arrays -> wait for the whole device
  (void)hipDeviceSynchronize();
  graph_release_all(graph);
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_destroy_async(tfgnn_graph* graph, void* last_use_stream) {
  if (!graph) return TFGNN_OK;
  if (graph->pending) (void)hipEventSynchronize((hipEvent_t)graph->event);
  graph_release_all(graph, (hipStream_t)last_use_stream, true);
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_nonempty_offsets(const tfgnn_graph* g, int by_src, int32_t* h_offsets) {
  TFGNN_REQUIRE(g != nullptr && h_offsets != nullptr, "NULL argument");
  TFGNN_REQUIRE(!g->pending, "tfgnn_graph_wait has not been called");
  for (int l = 0; l <= g->L; ++l) h_offsets[l] = g->compact[by_src ? 1 : 0].h_nz_off[l];
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_dims(const tfgnn_graph* g, int64_t* num_nodes, int* num_edge_types,
                                int64_t* num_edges) {
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  if (num_nodes) *num_nodes = g->V;
  if (num_edge_types) *num_edge_types = g->L;
  if (num_edges) *num_edges = g->E;
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_array(const tfgnn_graph* g, int array_id, const void** d_ptr, int64_t* count) {
  TFGNN_REQUIRE(g != nullptr && d_ptr != nullptr && count != nullptr, "NULL argument");
  switch (array_id) {
    case TFGNN_G_ROWPTR_BY_DST: *d_ptr = g->rowptr_d; *count = g->R + 1; break;
    case TFGNN_G_COL_BY_DST: *d_ptr = g->col_d; *count = g->E; break;
    case TFGNN_G_EID_BY_DST: *d_ptr = g->eid_d; *count = g->E; break;
    case TFGNN_G_COLL_BY_DST: *d_ptr = g->coll_d; *count = g->E; break;
    case TFGNN_G_ROWPTR_BY_SRC: *d_ptr = g->rowptr_s; *count = g->R + 1; break;
    case TFGNN_G_COL_BY_SRC: *d_ptr = g->col_s; *count = g->E; break;
    case TFGNN_G_EID_BY_SRC: *d_ptr = g->eid_s; *count = g->E; break;
    case TFGNN_G_COLL_BY_SRC: *d_ptr = g->coll_s; *count = g->E; break;
    case TFGNN_G_INVDEG_BY_DST: *d_ptr = g->invdeg_d; *count = g->R; break;
    case TFGNN_G_INVDEG_EDGE_BY_SRC: *d_ptr = g->invdeg_edge_s; *count = g->E; break;
    case TFGNN_G_NODEPTR_BY_DST: *d_ptr = g->nodeptr_d; *count = g->V + 1; break;
    case TFGNN_G_NODEPTR_BY_SRC: *d_ptr = g->nodeptr_s; *count = g->V + 1; break;
    case TFGNN_G_INVDEG_EDGE_BY_DST: *d_ptr = g->invdeg_edge_d; *count = g->E; break;
    case TFGNN_G_SRC2DST_POS: *d_ptr = g->src2dst; *count = g->E; break;
    case TFGNN_G_TARGET_BY_DST: *d_ptr = g->tgt_d; *count = g->E; break;
    case TFGNN_G_NZ_CPOS_BY_DST: case TFGNN_G_NZ_CPOS_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].cpos; *count = g->R; break;
    case TFGNN_G_NZ_ROW_BY_DST: case TFGNN_G_NZ_ROW_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].nzrow; *count = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].num_nz; break;
    case TFGNN_G_NZ_NODE_BY_DST: case TFGNN_G_NZ_NODE_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].nz_node; *count = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].num_nz; break;
    case TFGNN_G_NZ_OFF_BY_DST: case TFGNN_G_NZ_OFF_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].nz_off; *count = g->L + 1; break;
    case TFGNN_G_NZ_NODEPTR_BY_DST: case TFGNN_G_NZ_NODEPTR_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].nodeptr_nz; *count = g->V + 1; break;
    case TFGNN_G_NZ_COL_BY_DST: case TFGNN_G_NZ_COL_BY_SRC:
      *d_ptr = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].col_nz; *count = g->compact[array_id >= TFGNN_G_NZ_CPOS_BY_SRC].num_nz; break;
    default:
      tfgnn::set_error("unknown graph array id %d", array_id);
      return TFGNN_ERR_INVALID_ARGUMENT;
  }
  return TFGNN_OK;
}
