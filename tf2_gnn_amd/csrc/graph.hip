// Graph handle: bucket the edges of one batch by (target, edge_type) and by (source, edge_type).
//
// Replaces, once per batch instead of once per layer and pass, the index work the reference does
// inside MessagePassing.call (tf2_gnn/layers/message_passing/message_passing.py:166-167,195-206,
// 230-263): slicing the src/dst columns of every adjacency list, counting incoming edges per
// (type, node) with scatter_nd, gathering those counts per edge, and concatenating the per-type
// target lists.  Row r = node * L + edge_type; the in-degree c[l, v] is the length of row v*L+l.
//
// Pipeline (all on the caller's stream; deterministic result): one stable LSD radix sort of composite keys (bucket | column)
// over the bucket bits for both bucketings at once -> row pointers by binary search -> unpack + derived arrays.  Edges of a
// bucket keep the order of the adjacency lists.
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

namespace tfgnn {
int graph_require_parts(const tfgnn_graph* g, unsigned need, const char* who) {
  const unsigned missing = need & ~g->parts;
  if (!missing) return TFGNN_OK;
  set_error("%s: the graph handle lacks part%s%s%s%s (requested 0x%x at creation): tfgnn_graph_ensure(graph, parts, stream) builds it",
            who, (missing & TFGNN_GRAPH_PART_PLAN_TYPED) ? " PLAN_TYPED" : "", (missing & TFGNN_GRAPH_PART_PLAN_NODE) ? " PLAN_NODE" : "",
            (missing & TFGNN_GRAPH_PART_COMPACT) ? " COMPACT" : "",
            (missing & (TFGNN_GRAPH_PART_EDGE_MAPS | TFGNN_GRAPH_PART_EDGE_IDS | TFGNN_GRAPH_PART_DST_PATTERN)) ? " EDGE_MAPS / EDGE_IDS / DST_PATTERN" : "", g->parts);
  return TFGNN_ERR_INVALID_ARGUMENT;
}
}  // namespace tfgnn

extern "C" const char* tfgnn_last_error(void) { return tfgnn::g_err; }
extern "C" const char* tfgnn_version(void) { return "tfgnn 0.2 gfx950"; }
extern "C" int tfgnn_abi_version(void) { return TFGNN_ABI_VERSION; }

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
constexpr int EL_INLINE = 8;  // edge-list tables of up to this many types travel in the kernel arguments (no copy commands)
struct EdgeLists {
  const int32_t* const* adj;  // device array of L device pointers (L > EL_INLINE)
  const int64_t* edge_off;    // device [L+1]
  const int32_t* adj_v[EL_INLINE];
  int64_t off_v[EL_INLINE + 1];
  int L;
  int inline_tables;
};

__device__ __forceinline__ int find_type(const int64_t* __restrict__ edge_off, int L, int64_t g) {
  int lo = 0, hi = L;  // edge_off[lo] <= g < edge_off[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (edge_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// (source, target, type) of edge g of the concatenated adjacency lists
__device__ __forceinline__ void load_edge(const EdgeLists& el, int64_t g, int64_t& src, int64_t& dst, int& l) {
  const int32_t* a;
  if (el.inline_tables) {
    l = 0;
#pragma unroll
    for (int i = 1; i < EL_INLINE; ++i) l += (i < el.L && el.off_v[i] <= g) ? 1 : 0;
    a = el.adj_v[l] + 2 * (g - el.off_v[l]);
  } else {
    l = find_type(el.edge_off, el.L, g);
    a = el.adj[l] + 2 * (g - el.edge_off[l]);
  }
  src = a[0];
  dst = a[1];
}

// ---- bucketing = one LSD radix sort for BOTH CSRs (no global atomics, deterministic) -----------------
// composite key of an edge: (row << sec_bits) | sec, row = node * L + type (the bucket), sec = the
// node at the other end; payload = position of the edge in the concatenated adjacency lists.  Sorting
// by the composite puts edges in bucket order with ascending columns inside a bucket (canonical).
// Global memory atomics cost ~10 ns each on this chip (they execute at the memory side: the XCD L2s
// are not coherent), which made a counting sort with 4 atomics per edge take ~1 ms per batch.
//
// Round 4: the by-target and the by-source sort run in the SAME launches (blockIdx.y = side); keys are 32 bits wide when
// row and column bits fit (cfg-2: 17 + 15), else 64; there is no scan launch - a scatter workgroup takes the column prefix
// of its tile and the digit bases from the [digit][tile] table itself (256 threads x tiles/4 int4 loads, L2 resident); a
// wave owns 1024 CONSECUTIVE keys of its tile and its own running digit bases, so the scatter rounds need no workgroup
// barrier (two per tile in all, where the waves used to take turns: four per round).  Per digit: one histogram launch and
// one scatter launch for both sides (the first histogram comes out of the key-building kernel): 8 launches for 32-bit keys
// where the two sorts took 40.
constexpr int RS_THREADS = 512;
constexpr int RS_ROUNDS = 16;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_WAVE_KEYS = 64 * RS_ROUNDS;     // consecutive keys per wave
constexpr int RS_TILE = RS_THREADS * RS_ROUNDS;  // 8192 keys per workgroup
constexpr int RS_DIGIT_BITS = 9;
constexpr int RS_RADIX = 1 << RS_DIGIT_BITS;     // 512 = one digit per thread
static_assert(RS_RADIX == RS_THREADS, "one digit per thread");

// keys of both sides + the first-digit histogram of every tile: hist[(side * RS_RADIX + digit) * nblk_ld + tile]
template <typename KeyT>
__global__ void __launch_bounds__(RS_THREADS)
fill_keys_kernel(EdgeLists el, int64_t E, int64_t V, int sec_bits, KeyT* __restrict__ comp_d, KeyT* __restrict__ comp_s,
                 int32_t* __restrict__ hist, int nblk_ld, int32_t* __restrict__ err_flag) {
  __shared__ int32_t h[2][RS_RADIX];
  const int tid = threadIdx.x;
  h[0][tid] = 0;
  h[1][tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t g = base + r * RS_THREADS + tid;
    if (g < E) {
      int64_t src, dst;
      int l;
      load_edge(el, g, src, dst, l);
      if (src < 0 || src >= V || dst < 0 || dst >= V) {
        atomicOr(err_flag, 1);  // rare path; the build is rejected in tfgnn_graph_wait
        src = 0;
        dst = 0;
      }
      const KeyT kd = (KeyT)(((uint64_t)(dst * el.L + l) << sec_bits) | (uint64_t)src);
      const KeyT ks = (KeyT)(((uint64_t)(src * el.L + l) << sec_bits) | (uint64_t)dst);
      comp_d[g] = kd;
      comp_s[g] = ks;
      atomicAdd(&h[0][(int)((kd >> sec_bits) & (RS_RADIX - 1))], 1);  // LDS atomics
      atomicAdd(&h[1][(int)((ks >> sec_bits) & (RS_RADIX - 1))], 1);
    }
  }
  __syncthreads();
  hist[(int64_t)tid * nblk_ld + blockIdx.x] = h[0][tid];
  hist[(int64_t)(RS_RADIX + tid) * nblk_ld + blockIdx.x] = h[1][tid];
}

// hist[(side * 256 + digit) * nblk_ld + tile] = number of keys of the tile with that digit; grid (tiles, 2 sides)
template <typename KeyT>
__global__ void __launch_bounds__(RS_THREADS)
rs_hist_kernel(const KeyT* __restrict__ comp_d, const KeyT* __restrict__ comp_s, int64_t n, int shift,
               int32_t* __restrict__ hist, int nblk_ld) {
  __shared__ int32_t h[RS_RADIX];
  const int tid = threadIdx.x;
  const int side = blockIdx.y;
  const KeyT* __restrict__ comp = side ? comp_s : comp_d;
  h[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t idx = base + r * RS_THREADS + tid;
    if (idx < n) atomicAdd(&h[(int)((comp[idx] >> shift) & (RS_RADIX - 1))], 1);
  }
  __syncthreads();
  hist[(int64_t)(side * RS_RADIX + tid) * nblk_ld + blockIdx.x] = h[tid];
}

// stable scatter of one digit, both sides: keys keep their input order inside a digit (tile, then wave, round, lane =
// index order).  PAY: a 32-bit payload travels with every key (the edge id; pay_in == NULL: the key's index, first pass).
template <typename KeyT, bool PAY>
__global__ void __launch_bounds__(RS_THREADS)
rs_scatter_kernel(const KeyT* __restrict__ in_d, const KeyT* __restrict__ in_s, const uint32_t* __restrict__ pin_d,
                  const uint32_t* __restrict__ pin_s, int64_t n, int shift, const int32_t* __restrict__ hist, int nblk,
                  int nblk_ld, KeyT* __restrict__ out_d, KeyT* __restrict__ out_s, uint32_t* __restrict__ pout_d,
                  uint32_t* __restrict__ pout_s) {
  __shared__ int32_t wbase[RS_WAVES][RS_RADIX];  // per wave: digit counts, then running output positions
  __shared__ int32_t wtot[RS_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int side = blockIdx.y;
  const KeyT* __restrict__ kin = side ? in_s : in_d;
  const uint32_t* __restrict__ pin = side ? pin_s : pin_d;
  KeyT* __restrict__ kout = side ? out_s : out_d;
  uint32_t* __restrict__ pout = side ? pout_s : pout_d;
  volatile int32_t* mine_base = &wbase[wave][0];
#pragma unroll
  for (int i = 0; i < RS_RADIX / 64; ++i) mine_base[lane + 64 * i] = 0;
  __builtin_amdgcn_wave_barrier();
  // 1. keys into registers; digit counts of this wave (one leader lane per digit and round adds the group size)
  const int64_t wave0 = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * RS_WAVE_KEYS;
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  KeyT key[RS_ROUNDS];
  uint32_t pl[PAY ? RS_ROUNDS : 1];
  int32_t info[RS_ROUNDS];  // rank | group size << 8 | valid << 16
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t idx = wave0 + r * 64 + lane;
    const bool valid = idx < n;
    key[r] = valid ? kin[idx] : (KeyT)0;
    if (PAY) pl[r] = valid ? (pin ? pin[idx] : (uint32_t)idx) : 0u;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const bool valid = wave0 + r * 64 + lane < n;
    const int d = (int)((key[r] >> shift) & (RS_RADIX - 1));
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_DIGIT_BITS; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt_mask);
    const int cnt = __popcll(peers);
    info[r] = valid ? (rank | (cnt << 8) | (1 << 16)) : 0;
    if (valid && rank == 0) mine_base[d] = mine_base[d] + cnt;  // leaders of one round hold distinct digits
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // 2. thread d: keys with digit d in the tiles before this one + the digit's global base -> the waves' first positions
  int32_t tot = 0, before = 0;
  {
    const int32_t* __restrict__ row = hist + (int64_t)(side * RS_RADIX + tid) * nblk_ld;
    const int me = (int)blockIdx.x;
    for (int b0 = 0; b0 < nblk; b0 += 16) {
      int4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = b0 + 4 * u < nblk ? *reinterpret_cast<const int4*>(row + b0 + 4 * u) : make_int4(0, 0, 0, 0);  // nblk_ld % 4 == 0
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 4 * u;
        const int e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (b + k < nblk) {
            tot += e[k];
            if (b + k < me) before += e[k];
          }
      }
    }
  }
  // exclusive scan of tot over the digits: inside the wave, then over the waves
  int32_t incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  {
    int32_t off = 0;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    int32_t running = off + incl - tot + before;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const int32_t c = wbase[w][tid];
      wbase[w][tid] = running;
      running += c;
    }
  }
  __syncthreads();
  // 3. scatter: every wave on its own (LDS accesses of one wave execute in program order)
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const bool valid = (info[r] >> 16) & 1;
    const int d = (int)((key[r] >> shift) & (RS_RADIX - 1));
    const int rank = info[r] & 255, cnt = (info[r] >> 8) & 255;
    const int32_t b = valid ? mine_base[d] : 0;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) mine_base[d] = b + cnt;
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      kout[b + rank] = key[r];
      if (PAY) pout[b + rank] = pl[r];
    }
  }
}

// rowptr[r] = first sorted position whose bucket is >= r; grid.y = side
template <typename KeyT>
__global__ void rowptr_from_sorted_kernel(const KeyT* __restrict__ comp_d, const KeyT* __restrict__ comp_s, int64_t E,
                                          int sec_bits, int64_t R, int32_t* __restrict__ rowptr_d,
                                          int32_t* __restrict__ rowptr_s) {
  const KeyT* __restrict__ comp = blockIdx.y ? comp_s : comp_d;
  int32_t* __restrict__ rowptr = blockIdx.y ? rowptr_s : rowptr_d;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R; r += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = E;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)((uint64_t)comp[mid] >> sec_bits) < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = (int32_t)lo;
  }
}

// work items for rows longer than LONG_ROW_THRESHOLD (see graph.hpp / spmm.hip); counters = {items,
// multi-item rows, partial slots}.  The order in which rows claim their slots is irrelevant: a row's
// items are contiguous and are always combined in chunk order.
// One job per view; a launch takes the two views of a pair (typed: by target / by source; node: likewise) side by side.
struct PlanJob {
  const int32_t* rowptr;
  int64_t R;
  int thr, chunk;
  int32_t* counters;  // {items, multi-item rows, partial slots, short rows}
  int32_t *item_row, *item_chunk, *item_slot, *multi_row, *multi_base, *multi_n;
  int32_t *bin_count, *bin_cursor, *short_rows;
  int want_items;  // E > 0
};
struct PlanJobs {
  PlanJob j[2];
};

__device__ __forceinline__ void plan_rows_body(const int32_t* __restrict__ rowptr, int64_t R, int long_threshold, int chunk_edges,
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
__device__ __forceinline__ void
short_hist_body(const int32_t* __restrict__ rowptr, int64_t R, int thr, int32_t* __restrict__ bin_count,
                int32_t* __restrict__ num_short, int* h, int& n_short) {
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

// blockIdx.y = view of the pair + 2 * job: job 0 = length histogram of the short rows, job 1 = items of the long rows
__global__ void __launch_bounds__(256) plan_hist_kernel(PlanJobs jobs) {
  __shared__ int h[SHORT_BINS];
  __shared__ int n_short;
  const PlanJob& j = jobs.j[blockIdx.y & 1];
  if (j.R <= 0) return;
  if ((blockIdx.y >> 1) == 0) {
    short_hist_body(j.rowptr, j.R, j.thr, j.bin_count, j.counters + 3, h, n_short);
  } else if (j.want_items) {
    plan_rows_body(j.rowptr, j.R, j.thr, j.chunk, j.counters, j.item_row, j.item_chunk, j.item_slot, j.multi_row, j.multi_base,
                   j.multi_n);
  }
}

// short_rows = rows with len <= thr in order of descending length (the order inside one length is arbitrary and
// has no effect on any result: every row is reduced on its own)
constexpr int SHORT_TILE_ROWS = 8;  // rows per thread and tile: one global atomic per (tile, length) pair
__global__ void __launch_bounds__(256) short_scatter_kernel(PlanJobs jobs) {
  __shared__ int h[SHORT_BINS], base[SHORT_BINS], start[SHORT_BINS];
  const PlanJob& j = jobs.j[blockIdx.y];
  const int32_t* __restrict__ rowptr = j.rowptr;
  const int64_t R = j.R;
  const int thr = j.thr;
  const int32_t* __restrict__ bin_count = j.bin_count;
  int32_t* __restrict__ bin_cursor = j.bin_cursor;
  int32_t* __restrict__ short_rows = j.short_rows;
  if (R <= 0) return;
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

// ---- nodes grouped by which of their (node, type) buckets are empty (part DST_PATTERN, round 4) ------------------------------
// 45 % of the buckets of an R-MAT batch are empty: the 320-column block of such a bucket in [A_0|..|A_{L-1}] is zeros the
// product multiplies for nothing.  With the rows of a product tile sharing one emptiness pattern the product skips the empty
// blocks of the whole tile: the gather writes its rows at pos[node] (nodes ordered by pattern), the product walks a per-tile
// list of K blocks and writes its output rows back to node order.  L <= 8 (one mask byte per node / tile).
__device__ __forceinline__ int node_pattern(const int32_t* __restrict__ rowptr, int64_t v, int L) {
  int m = 0;
  for (int l = 0; l < L; ++l) m |= (rowptr[v * L + l + 1] > rowptr[v * L + l]) ? (1 << l) : 0;
  return m;
}
__global__ void __launch_bounds__(256) pattern_count_kernel(const int32_t* __restrict__ rowptr, int64_t V, int L,
                                                            int32_t* __restrict__ bins /* [256], zeroed */) {
  __shared__ int h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (int64_t)gridDim.x * 256) atomicAdd(&h[node_pattern(rowptr, v, L)], 1);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
}
// pos[v] = first position of v's pattern + a slot claimed inside the pattern (the order inside a pattern is arbitrary and has
// no effect on any result: every row is computed on its own, only its place in the operand differs)
__global__ void __launch_bounds__(256) pattern_place_kernel(const int32_t* __restrict__ rowptr, int64_t V, int L,
                                                            const int32_t* __restrict__ bins, int32_t* __restrict__ cursor /* [256], zeroed */,
                                                            int32_t* __restrict__ pos, int32_t* __restrict__ node_at,
                                                            int32_t* __restrict__ rowmap) {
  __shared__ int base[256], h[256], slot0[256];
  {  // patterns with many non-empty buckets first (any fixed order would do): order key (8 - popcount, pattern)
    h[threadIdx.x] = bins[threadIdx.x];
    __syncthreads();
    const int key = ((8 - __popc((int)threadIdx.x)) << 8) | (int)threadIdx.x;
    int acc = 0;
    for (int m = 0; m < 256; ++m) acc += ((((8 - __popc(m)) << 8) | m) < key) ? h[m] : 0;
    base[threadIdx.x] = acc;
    __syncthreads();
  }
  for (int64_t v0 = (int64_t)blockIdx.x * 256; v0 < V; v0 += (int64_t)gridDim.x * 256) {
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t v = v0 + threadIdx.x;
    int m = 0, rank = 0;
    if (v < V) {
      m = node_pattern(rowptr, v, L);
      rank = atomicAdd(&h[m], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) slot0[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (v < V) {
      const int32_t p = base[m] + slot0[m] + rank;
      if (pos) pos[v] = p;
      node_at[p] = (int32_t)v;
      if (rowmap)
        for (int l = 0; l < L; ++l) rowmap[v * L + l] = p * L + l;
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(128) pattern_tilemask_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ node_at,
                                                               int64_t V, int L, uint8_t* __restrict__ tilemask) {
  __shared__ int m;
  if (threadIdx.x == 0) m = 0;
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (p < V) atomicOr(&m, node_pattern(rowptr, node_at[p], L));
  __syncthreads();
  if (threadIdx.x == 0) tilemask[blockIdx.x] = (uint8_t)m;
}

// blockIdx.y = 0 / 1: unpack the sorted keys of the by-target / by-source side; blockIdx.y = 2: per-row arrays
// (1 / in-degree, node pointers).  The degree that normalises an edge is always the in-degree of its TARGET for its type,
// i.e. the length of by-target row (target, type) - read from rowptr_d directly, so the three jobs share one launch.
template <typename KeyT, bool PAY>
__global__ void unpack_kernel(const KeyT* __restrict__ comp_d, const KeyT* __restrict__ comp_s, const uint32_t* __restrict__ pay_d,
                              const uint32_t* __restrict__ pay_s, int sec_bits, int64_t E, int L, int64_t R,
                              const int32_t* __restrict__ rowptr_d, const int32_t* __restrict__ rowptr_s,
                              int32_t* __restrict__ col_d, int32_t* __restrict__ eid_d, int32_t* __restrict__ coll_d,
                              float* __restrict__ invdeg_edge_d, int32_t* __restrict__ eid_to_pos, int32_t* __restrict__ row_node,
                              int32_t* __restrict__ col_s, int32_t* __restrict__ eid_s, int32_t* __restrict__ coll_s,
                              float* __restrict__ invdeg_edge_s, float* __restrict__ invdeg_d, int32_t* __restrict__ nodeptr_d,
                              int32_t* __restrict__ nodeptr_s, int first_job) {
  const int job = (int)blockIdx.y + first_job;
  if (job == 2) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R; r += (int64_t)gridDim.x * blockDim.x) {
      if (r < R) {
        const int32_t len = rowptr_d[r + 1] - rowptr_d[r];
        // gnn_edge_mlp.py:102-106: 1.0 / (num_incoming + SMALL_NUMBER), evaluated in fp32 like TF
        invdeg_d[r] = len > 0 ? 1.0f / ((float)len + kSmallNumber) : 0.f;
      }
      if (r % L == 0) {
        nodeptr_d[r / L] = rowptr_d[r];
        nodeptr_s[r / L] = rowptr_s[r];
      }
    }
    return;
  }
  const int by_src = job;
  const KeyT* __restrict__ comp = by_src ? comp_s : comp_d;
  const uint32_t* __restrict__ pay = by_src ? pay_s : pay_d;
  int32_t* __restrict__ col = by_src ? col_s : col_d;
  int32_t* __restrict__ eid = by_src ? eid_s : eid_d;
  int32_t* __restrict__ coll = by_src ? coll_s : coll_d;
  float* __restrict__ invdeg_edge = by_src ? invdeg_edge_s : invdeg_edge_d;
  const uint64_t sec_mask = ((uint64_t)1 << sec_bits) - 1;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t c = (uint64_t)comp[p];
    const int32_t other = (int32_t)(c & sec_mask);
    const int32_t row = (int32_t)(c >> sec_bits);
    const int l = row % L;
    col[p] = other;
    if (PAY) eid[p] = (int32_t)pay[p];
    const int64_t cl = (int64_t)other * L + l;
    coll[p] = (int32_t)cl;
    const int64_t trow = by_src ? cl : (int64_t)row;  // the by-target bucket of this edge
    const int32_t len = rowptr_d[trow + 1] - rowptr_d[trow];
    invdeg_edge[p] = len > 0 ? 1.0f / ((float)len + kSmallNumber) : 0.f;
    if (!by_src) {
      row_node[p] = row / L;
      if (PAY) eid_to_pos[pay[p]] = (int32_t)p;
    }
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

namespace {
using namespace tfgnn;

constexpr unsigned kPartsAll = TFGNN_GRAPH_PART_PLAN_TYPED | TFGNN_GRAPH_PART_PLAN_NODE | TFGNN_GRAPH_PART_COMPACT |
                               TFGNN_GRAPH_PART_EDGE_MAPS | TFGNN_GRAPH_PART_EDGE_IDS | TFGNN_GRAPH_PART_DST_PATTERN;

// long-row plan parameters per view (tools/gather_probe.py sweeps at cfg-2, rows ordered by length): typed
// views 91 us at (48, 512) vs 104 us at (16, 192) (134 us in natural row order at (16, 128)); the node views -
// all edge types of a node in one row - 121 us at (32, 512) vs 128 us at (64, 512) (138 us in natural order).  TFGNN_LONG_ROW / TFGNN_ITEM_CHUNK override the typed views, TFGNN_LONG_ROW_NODE /
// TFGNN_ITEM_CHUNK_NODE the node views, for probing.
void view_plan_parameters(int view_long[4], int view_chunk[4]) {
  static const int env_long = [] { const char* e = getenv("TFGNN_LONG_ROW"); return e ? atoi(e) : 0; }();
  static const int env_chunk = [] { const char* e = getenv("TFGNN_ITEM_CHUNK"); return e ? atoi(e) : 0; }();
  static const int env_long_n = [] { const char* e = getenv("TFGNN_LONG_ROW_NODE"); return e ? atoi(e) : 0; }();
  static const int env_chunk_n = [] { const char* e = getenv("TFGNN_ITEM_CHUNK_NODE"); return e ? atoi(e) : 0; }();
  const int dl[4] = {LONG_ROW_THRESHOLD_TYPED, LONG_ROW_THRESHOLD, LONG_ROW_THRESHOLD_TYPED, LONG_ROW_THRESHOLD};
  const int dc[4] = {ITEM_CHUNK_TYPED, ITEM_CHUNK, ITEM_CHUNK_TYPED, ITEM_CHUNK};
  for (int v = 0; v < 4; ++v) {
    const bool node_view = v & 1;
    view_long[v] = dl[v];
    view_chunk[v] = dc[v];
    if ((node_view ? env_long_n : env_long) > 0) view_long[v] = node_view ? env_long_n : env_long;
    if ((node_view ? env_chunk_n : env_chunk) > 0) view_chunk[v] = node_view ? env_chunk_n : env_chunk;
    if (view_long[v] > SHORT_BINS - 1) view_long[v] = SHORT_BINS - 1;
    if (view_chunk[v] < view_long[v]) view_chunk[v] = view_long[v];
  }
}

unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256), 8192)); }

// the plans of one pair of views (node_views = 0: views 0 and 2, 1: views 1 and 3): two launches.
// counters: the 64-int counter block (zeroed), bins: [4 views][2][SHORT_BINS] (zeroed)
int build_plans(tfgnn_graph* g, int node_views, int32_t* counters, int32_t* bins, hipStream_t s) {
  PlanJobs jobs{};
  int64_t max_rows = 0;
  for (int k = 0; k < 2; ++k) {
    const int v = 2 * k + node_views;
    CsrPlan& pl = g->views[v].plan;
    PlanJob& j = jobs.j[k];
    j.rowptr = g->views[v].rowptr;
    j.R = g->views[v].num_rows;
    j.thr = pl.long_threshold;
    j.chunk = pl.item_chunk_edges;
    j.counters = counters + 16 + 4 * v;
    j.item_row = pl.item_row; j.item_chunk = pl.item_chunk; j.item_slot = pl.item_slot;
    j.multi_row = pl.multi_row; j.multi_base = pl.multi_base; j.multi_n = pl.multi_n;
    j.bin_count = bins + (size_t)v * 2 * SHORT_BINS;
    j.bin_cursor = j.bin_count + SHORT_BINS;
    j.short_rows = pl.short_rows;
    j.want_items = g->E > 0;
    max_rows = std::max(max_rows, j.R);
  }
  if (max_rows <= 0) return TFGNN_OK;
  hipLaunchKernelGGL(plan_hist_kernel, dim3(blocks_for(max_rows), 4), dim3(256), 0, s, jobs);
  TFGNN_LAUNCH_CHECK();
  const unsigned scat_blocks = (unsigned)std::min<int64_t>(ceil_div(max_rows, 256 * SHORT_TILE_ROWS), 4096);
  hipLaunchKernelGGL(short_scatter_kernel, dim3(scat_blocks, 2), dim3(256), 0, s, jobs);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

size_t compact_scratch_bytes(int64_t R, int64_t V) {
  return (size_t)(R + 2) * 4 + 256 + (scan_scratch_elems(R + 1) + scan_scratch_elems(V + 1)) * 4 + 64;
}

// non-empty buckets in type-major order, for both bucketings; nz_off -> the pinned staging block
int build_compact(tfgnn_graph* g, char* scratch, hipStream_t s) {
  const int L = g->L;
  const int64_t V = g->V, R = g->R;
  int32_t* flags = (int32_t*)scratch;
  int32_t* scan_tmp = (int32_t*)(scratch + (((size_t)(R + 2) * 4 + 255) & ~(size_t)255));
  for (int side = 0; side < 2; ++side) {
    CompactBuckets& cb = g->compact[side];
    const int32_t* rp = side == 0 ? g->rowptr_d : g->rowptr_s;
    if (L > 0) {
      hipLaunchKernelGGL(nz_flags_kernel, dim3(blocks_for(R + 1)), dim3(256), 0, s, rp, V, L, flags, cb.nodeptr_nz);
      int rc = exclusive_scan_i32(flags, flags, R + 1, scan_tmp, s);
      if (rc) return rc;
      rc = exclusive_scan_i32(cb.nodeptr_nz, cb.nodeptr_nz, V + 1, scan_tmp, s);
      if (rc) return rc;
      if (R > 0) {
        hipLaunchKernelGGL(nz_fill_kernel, dim3(blocks_for(R)), dim3(256), 0, s, rp, flags, V, L, cb.cpos, cb.nzrow,
                           cb.nz_node, cb.nz_off);
        hipLaunchKernelGGL(nz_cols_kernel, dim3(blocks_for(V)), dim3(256), 0, s, cb.cpos, cb.nodeptr_nz, V, L, cb.col_nz);
      } else {
        TFGNN_HIP_CHECK(hipMemsetAsync(cb.nz_off, 0, (size_t)(L + 1) * 4, s));
      }
    } else {
      TFGNN_HIP_CHECK(hipMemsetAsync(cb.nz_off, 0, (size_t)(L + 1) * 4, s));
      TFGNN_HIP_CHECK(hipMemsetAsync(cb.nodeptr_nz, 0, (V + 1) * 4, s));
    }
    TFGNN_HIP_CHECK(hipMemcpyAsync((char*)g->pinned + 256 + (size_t)(L + 1) * 16 + (size_t)side * (L + 1) * 4, cb.nz_off,
                                   (size_t)(L + 1) * 4, hipMemcpyDeviceToHost, s));
  }
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// nodes ordered by the emptiness pattern of their by-target buckets (kernels above); pat = the 512 zeroed ints behind the bins
int build_dst_pattern(tfgnn_graph* g, int32_t* pat, hipStream_t s) {
  if (g->V <= 0 || g->L <= 0 || g->L > 8) return TFGNN_OK;  // (L > 8: the part stays empty, callers fall back)
  const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(g->V, 256), 1024);
  hipLaunchKernelGGL(pattern_count_kernel, dim3(blocks), dim3(256), 0, s, (const int32_t*)g->rowptr_d, g->V, g->L, pat);
  hipLaunchKernelGGL(pattern_place_kernel, dim3(blocks), dim3(256), 0, s, (const int32_t*)g->rowptr_d, g->V, g->L, (const int32_t*)pat,
                     pat + 256, g->pat_pos_d, g->pat_node_d, g->pat_rowmap_d);
  hipLaunchKernelGGL(pattern_tilemask_kernel, dim3((unsigned)ceil_div(g->V, 128)), dim3(128), 0, s, (const int32_t*)g->rowptr_d,
                     (const int32_t*)g->pat_node_d, g->V, g->L, g->pat_tilemask_d);
  // the same for the by-SOURCE buckets (round 5: the input-gradient product over [G_0|..|G_{L-1}] reads its rows through
  // node_at and skips the all-zero blocks of a tile; the gather keeps writing node order, which the weight-gradient
  // product needs) - no position / row-map arrays on this side
  int32_t* pat_s = pat + 512;
  hipLaunchKernelGGL(pattern_count_kernel, dim3(blocks), dim3(256), 0, s, (const int32_t*)g->rowptr_s, g->V, g->L, pat_s);
  hipLaunchKernelGGL(pattern_place_kernel, dim3(blocks), dim3(256), 0, s, (const int32_t*)g->rowptr_s, g->V, g->L, (const int32_t*)pat_s,
                     pat_s + 256, (int32_t*)nullptr, g->pat_node_s, (int32_t*)nullptr);
  hipLaunchKernelGGL(pattern_tilemask_kernel, dim3((unsigned)ceil_div(g->V, 128)), dim3(128), 0, s, (const int32_t*)g->rowptr_s,
                     (const int32_t*)g->pat_node_s, g->V, g->L, g->pat_tilemask_s);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

int build_edge_maps(tfgnn_graph* g, hipStream_t s) {
  if (g->E > 0) {
    hipLaunchKernelGGL(src2dst_kernel, dim3(blocks_for(g->E)), dim3(256), 0, s, g->eid_s, g->eid2pos, g->E, g->src2dst, g->dst2src);
    TFGNN_LAUNCH_CHECK();
  }
  return TFGNN_OK;
}

// the sort of both sides + everything derived per edge and per row; KeyT = uint32_t when row and column bits fit.
// The sort runs over the ROW bits of the composite only (9-bit digits from bit sec_bits up): the column rides in the low bits
// of the key - edges of a bucket keep the order of the adjacency list (stable), which is all the sums need to be reproducible.
// PAY: the edge ids travel along as a 32-bit payload (parts that map edges back to the caller's lists).
template <typename KeyT, bool PAY>
int build_core(tfgnn_graph* g, const EdgeLists& el, char* keys, char* pays, int32_t* hist, int nblk, int nblk_ld,
               int32_t* counters, hipStream_t s) {
  const int64_t E = g->E, R = g->R, V = g->V;
  const int L = g->L;
  const int sec_bits = g->sec_bits, total_bits = g->total_bits;
  KeyT* kbuf[2][2];  // [side][ping-pong]
  uint32_t* pbuf[2][2];
  for (int side = 0; side < 2; ++side)
    for (int b = 0; b < 2; ++b) {
      kbuf[side][b] = (KeyT*)(keys + (size_t)(side * 2 + b) * (size_t)E * 8);
      pbuf[side][b] = PAY ? (uint32_t*)(pays + (size_t)(side * 2 + b) * (size_t)E * 4) : nullptr;
    }
  hipLaunchKernelGGL((fill_keys_kernel<KeyT>), dim3(nblk), dim3(RS_THREADS), 0, s, el, E, V, sec_bits, kbuf[0][0], kbuf[1][0],
                     hist, nblk_ld, counters + 2);
  int cur = 0;
  bool first = true;
  for (int shift = sec_bits; shift < total_bits; shift += RS_DIGIT_BITS) {
    if (!first)
      hipLaunchKernelGGL((rs_hist_kernel<KeyT>), dim3(nblk, 2), dim3(RS_THREADS), 0, s, (const KeyT*)kbuf[0][cur],
                         (const KeyT*)kbuf[1][cur], E, shift, hist, nblk_ld);
    hipLaunchKernelGGL((rs_scatter_kernel<KeyT, PAY>), dim3(nblk, 2), dim3(RS_THREADS), 0, s, (const KeyT*)kbuf[0][cur],
                       (const KeyT*)kbuf[1][cur], first ? (const uint32_t*)nullptr : (const uint32_t*)pbuf[0][cur],
                       first ? (const uint32_t*)nullptr : (const uint32_t*)pbuf[1][cur], E, shift, (const int32_t*)hist, nblk, nblk_ld,
                       kbuf[0][cur ^ 1], kbuf[1][cur ^ 1], pbuf[0][cur ^ 1], pbuf[1][cur ^ 1]);
    cur ^= 1;
    first = false;
  }
  hipLaunchKernelGGL((rowptr_from_sorted_kernel<KeyT>), dim3(blocks_for(R + 1), 2), dim3(256), 0, s, (const KeyT*)kbuf[0][cur],
                     (const KeyT*)kbuf[1][cur], E, sec_bits, R, g->rowptr_d, g->rowptr_s);
  hipLaunchKernelGGL((unpack_kernel<KeyT, PAY>), dim3(blocks_for(std::max(E, R + 1)), 3), dim3(256), 0, s, (const KeyT*)kbuf[0][cur],
                     (const KeyT*)kbuf[1][cur], (const uint32_t*)pbuf[0][cur], (const uint32_t*)pbuf[1][cur], sec_bits, E, L, R,
                     (const int32_t*)g->rowptr_d, (const int32_t*)g->rowptr_s, g->col_d, g->eid_d, g->coll_d, g->invdeg_edge_d,
                     g->eid2pos, g->tgt_d, g->col_s, g->eid_s, g->coll_s, g->invdeg_edge_s, g->invdeg_d, g->nodeptr_d,
                     g->nodeptr_s, 0);
  TFGNN_LAUNCH_CHECK();
  return TFGNN_OK;
}

// scratch of one run of the sort
struct CoreScratch {
  size_t keys, pays, hist, counters, ptrs, off, total;
  int nblk, nblk_ld;
};
CoreScratch core_scratch(int64_t E, int L, bool pay) {
  CoreScratch c{};
  SlabPlan tmp;
  c.nblk = (int)ceil_div(E > 0 ? E : 1, RS_TILE);
  c.nblk_ld = (c.nblk + 3) & ~3;
  c.keys = tmp.take((size_t)E * 8 * 4);
  c.pays = tmp.take(pay ? (size_t)E * 4 * 4 : 0);
  c.hist = tmp.take((size_t)2 * RS_RADIX * c.nblk_ld * 4 + 16);
  c.counters = tmp.take((size_t)(64 + 4 * 2 * SHORT_BINS + 1024) * 4);  // counters, the bins, the pattern bins: one memset
  c.ptrs = tmp.take((size_t)(L + 1) * 8);
  c.off = tmp.take((size_t)(L + 1) * 8);
  c.total = tmp.total;
  return c;
}

// E > 0: keys, sort, row pointers, per-edge arrays of both sides.  `scratch` laid out by core_scratch; the counters block
// (zeroed by the caller) receives the bad-index flag at [2].  Reads the adjacency lists the handle was created from.
int run_core(tfgnn_graph* g, bool pay, char* scratch, const CoreScratch& cs, hipStream_t s) {
  const int L = g->L;
  EdgeLists el{};
  el.L = L;
  el.inline_tables = L <= EL_INLINE;
  if (el.inline_tables) {
    for (int l = 0; l < EL_INLINE; ++l) el.adj_v[l] = l < L ? g->h_adj[l] : nullptr;
    for (int l = 0; l <= EL_INLINE; ++l) el.off_v[l] = l <= L ? g->h_off[l] : g->E;
  } else {
    // pointer / offset tables go through the handle's pinned staging block: no host synchronisation
    const int32_t** d_ptrs = (const int32_t**)(scratch + cs.ptrs);
    int64_t* d_off = (int64_t*)(scratch + cs.off);
    char* hp = (char*)g->pinned + 256;
    const int32_t** h_ptrs = (const int32_t**)hp;
    int64_t* h_off = (int64_t*)(hp + (size_t)(L + 1) * 8);
    for (int l = 0; l < L; ++l) h_ptrs[l] = g->h_adj[l];
    for (int l = 0; l <= L; ++l) h_off[l] = g->h_off[l];
    TFGNN_HIP_CHECK(hipMemcpyAsync(d_ptrs, h_ptrs, (size_t)L * 8, hipMemcpyHostToDevice, s));
    TFGNN_HIP_CHECK(hipMemcpyAsync(d_off, h_off, (size_t)(L + 1) * 8, hipMemcpyHostToDevice, s));
    el.adj = d_ptrs;
    el.edge_off = d_off;
  }
  char* keys = scratch + cs.keys;
  char* pays = scratch + cs.pays;
  int32_t* hist = (int32_t*)(scratch + cs.hist);
  int32_t* counters = (int32_t*)(scratch + cs.counters);
  if (g->total_bits <= 32)
    return pay ? build_core<uint32_t, true>(g, el, keys, pays, hist, cs.nblk, cs.nblk_ld, counters, s)
               : build_core<uint32_t, false>(g, el, keys, pays, hist, cs.nblk, cs.nblk_ld, counters, s);
  return pay ? build_core<uint64_t, true>(g, el, keys, pays, hist, cs.nblk, cs.nblk_ld, counters, s)
             : build_core<uint64_t, false>(g, el, keys, pays, hist, cs.nblk, cs.nblk_ld, counters, s);
}
}  // namespace

static int graph_create_impl(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                             const int64_t* num_edges, unsigned parts, void* stream, tfgnn_graph** out_graph) {
  using namespace tfgnn;
  TFGNN_REQUIRE(out_graph != nullptr, "out_graph is NULL");
  *out_graph = nullptr;
  TFGNN_REQUIRE(num_edge_types >= 0 && num_nodes >= 0, "negative sizes");
  TFGNN_REQUIRE(num_edge_types == 0 || (d_adjacency && num_edges), "adjacency arrays are NULL");
  TFGNN_REQUIRE((parts & ~kPartsAll) == 0, "unknown graph part bits 0x%x", parts);
  hipStream_t s = (hipStream_t)stream;
  // handle initialisation is where this device's per-process state is allocated (the dropout epoch word, common.hpp): never
  // inside a stream capture, and a failure is reported by the first call that needs the word
  (void)dropout_epoch_word();
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
  TFGNN_REQUIRE(L <= 256, "at most 256 edge types are supported (got %d)", L);

  tfgnn_graph* g = new tfgnn_graph();
  g->L = L;
  g->V = V;
  g->E = E;
  g->R = R;

  // persistent arrays (all parts have their place in the slab whether they are built now or later)
  SlabPlan plan;
  const size_t o_rowptr_d = plan.take((R + 1) * 4), o_rowptr_s = plan.take((R + 1) * 4);
  const size_t o_col_d = plan.take(E * 4), o_eid_d = plan.take(E * 4), o_coll_d = plan.take(E * 4);
  const size_t o_col_s = plan.take(E * 4), o_eid_s = plan.take(E * 4), o_coll_s = plan.take(E * 4);
  const size_t o_nodeptr_d = plan.take((V + 1) * 4), o_nodeptr_s = plan.take((V + 1) * 4);
  const size_t o_src2dst = plan.take(E * 4), o_dst2src = plan.take(E * 4);
  const size_t o_tgt_d = plan.take(E * 4), o_eid2pos = plan.take(E * 4);
  const size_t o_pat_pos = plan.take((V + 1) * 4), o_pat_node = plan.take((V + 1) * 4), o_pat_rowmap = plan.take((R + 1) * 4);
  const size_t o_pat_mask = plan.take((size_t)ceil_div(V > 0 ? V : 1, 128) + 16);
  const size_t o_pat_node_s = plan.take((V + 1) * 4), o_pat_mask_s = plan.take((size_t)ceil_div(V > 0 ? V : 1, 128) + 16);
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
  int view_long[4], view_chunk[4];
  view_plan_parameters(view_long, view_chunk);
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
  // build-time scratch (freed by tfgnn_graph_wait)
  if (parts & TFGNN_GRAPH_PART_EDGE_MAPS) parts |= TFGNN_GRAPH_PART_EDGE_IDS;  // the maps are built from the edge ids
  const bool with_ids = (parts & TFGNN_GRAPH_PART_EDGE_IDS) != 0;
  const CoreScratch cs = core_scratch(E, L, with_ids);
  const size_t t_compact = (cs.total + 255) & ~(size_t)255;
  const size_t scratch_total = t_compact + ((parts & TFGNN_GRAPH_PART_COMPACT) ? compact_scratch_bytes(R, V) : 0);
  g->h_adj.assign(d_adjacency, d_adjacency + L);
  g->h_off = edge_off;
  {
    int sec_bits = 1;
    while (((int64_t)1 << sec_bits) < V) ++sec_bits;
    int row_bits = 1;
    while (((int64_t)1 << row_bits) < (R > 0 ? R : 1)) ++row_bits;
    g->sec_bits = sec_bits;
    g->total_bits = sec_bits + row_bits;
  }

  char* slab = nullptr;
  char* scratch = nullptr;
  const size_t pinned_need = 256 + (size_t)(L + 1) * 16 + (size_t)(L + 1) * 8;
  hipError_t he = dev_alloc((void**)&slab, persistent, &g->slab_bytes, s);
  if (he == hipSuccess) {
    g->slab = slab;
    he = dev_alloc((void**)&scratch, scratch_total, &g->scratch_bytes, s);
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
    set_error("graph allocation failed (%zu + %zu bytes): %s", persistent, scratch_total, hipGetErrorString(he));
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
  g->eid2pos = (int32_t*)(slab + o_eid2pos);
  g->pat_pos_d = (int32_t*)(slab + o_pat_pos);
  g->pat_node_d = (int32_t*)(slab + o_pat_node);
  g->pat_rowmap_d = (int32_t*)(slab + o_pat_rowmap);
  g->pat_tilemask_d = (uint8_t*)(slab + o_pat_mask);
  g->pat_node_s = (int32_t*)(slab + o_pat_node_s);
  g->pat_tilemask_s = (uint8_t*)(slab + o_pat_mask_s);
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
  // gather views (tfgnn_graph_view order) and the places of their long-row plans
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
  }

  int32_t* counters = (int32_t*)(scratch + cs.counters);
  int32_t* bins = counters + 64;

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

  G_CHECK(hipMemsetAsync(counters, 0, (size_t)(64 + 4 * 2 * SHORT_BINS + 1024) * 4, s));
  // sort the edges into bucket order for both bucketings, then everything per edge and per row
  if (E > 0) {
    rc = run_core(g, with_ids, scratch, cs, s);
    if (rc) return fail(rc);
  } else {
    G_CHECK(hipMemsetAsync(g->rowptr_d, 0, (R + 1) * 4, s));
    G_CHECK(hipMemsetAsync(g->rowptr_s, 0, (R + 1) * 4, s));
    if (L > 0) {
      hipLaunchKernelGGL((unpack_kernel<uint32_t, false>), dim3(blocks_for(R + 1), 1), dim3(256), 0, s, (const uint32_t*)nullptr,
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, (int64_t)0, L, R,
                         (const int32_t*)g->rowptr_d, (const int32_t*)g->rowptr_s, g->col_d, g->eid_d, g->coll_d, g->invdeg_edge_d,
                         g->eid2pos, g->tgt_d, g->col_s, g->eid_s, g->coll_s, g->invdeg_edge_s, g->invdeg_d, g->nodeptr_d,
                         g->nodeptr_s, 2);
    } else {
      G_CHECK(hipMemsetAsync(g->nodeptr_d, 0, (V + 1) * 4, s));
      G_CHECK(hipMemsetAsync(g->nodeptr_s, 0, (V + 1) * 4, s));
    }
  }
  if (parts & TFGNN_GRAPH_PART_PLAN_TYPED) {
    rc = build_plans(g, 0, counters, bins, s);
    if (rc) return fail(rc);
  }
  if (parts & TFGNN_GRAPH_PART_PLAN_NODE) {
    rc = build_plans(g, 1, counters, bins, s);
    if (rc) return fail(rc);
  }
  if (parts & TFGNN_GRAPH_PART_EDGE_MAPS) {
    rc = build_edge_maps(g, s);
    if (rc) return fail(rc);
  }
  if (parts & TFGNN_GRAPH_PART_DST_PATTERN) {
    rc = build_dst_pattern(g, bins + 4 * 2 * SHORT_BINS, s);
    if (rc) return fail(rc);
  }
  if (parts & TFGNN_GRAPH_PART_COMPACT) {
    rc = build_compact(g, scratch + t_compact, s);
    if (rc) return fail(rc);
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
  g->parts = parts;
  g->pending = true;
  *out_graph = g;
  return TFGNN_OK;
}

extern "C" int tfgnn_graph_create_async(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                                        const int64_t* num_edges, void* stream, tfgnn_graph** out_graph) {
  return graph_create_impl(num_edge_types, num_nodes, d_adjacency, num_edges, TFGNN_GRAPH_PARTS_DEFAULT, stream, out_graph);
}

extern "C" int tfgnn_graph_create_parts_async(int num_edge_types, int64_t num_nodes, const int32_t* const* d_adjacency,
                                              const int64_t* num_edges, unsigned parts, void* stream, tfgnn_graph** out_graph) {
  return graph_create_impl(num_edge_types, num_nodes, d_adjacency, num_edges, parts, stream, out_graph);
}

static void graph_read_counters(tfgnn_graph* g, unsigned parts) {
  const int32_t* h_counters = (const int32_t*)g->pinned;
  for (int v = 0; v < 4; ++v) {
    if (!(parts & ((v & 1) ? TFGNN_GRAPH_PART_PLAN_NODE : TFGNN_GRAPH_PART_PLAN_TYPED))) continue;
    g->views[v].plan.num_items = h_counters[16 + 4 * v + 0];
    g->views[v].plan.num_multi = h_counters[16 + 4 * v + 1];
    g->views[v].plan.num_partials = h_counters[16 + 4 * v + 2];
    g->views[v].plan.num_short = h_counters[16 + 4 * v + 3];
  }
  if (parts & TFGNN_GRAPH_PART_COMPACT)
    for (int side = 0; side < 2; ++side) {
      const int32_t* h = (const int32_t*)((const char*)g->pinned + 256 + (size_t)(g->L + 1) * 16 + (size_t)side * (g->L + 1) * 4);
      for (int l = 0; l <= g->L; ++l) g->compact[side].h_nz_off[l] = h[l];
      g->compact[side].num_nz = h[g->L];
    }
}

// Build the parts that were not requested at creation (tfgnn_graph_create_parts_async) on `stream`; blocks the host until
// they are there (their sizes come back from the device).  A no-op for parts the handle already has.
extern "C" int tfgnn_graph_ensure(tfgnn_graph* g, unsigned parts, void* stream) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  TFGNN_REQUIRE((parts & ~kPartsAll) == 0, "unknown graph part bits 0x%x", parts);
  if (parts & TFGNN_GRAPH_PART_EDGE_MAPS) parts |= TFGNN_GRAPH_PART_EDGE_IDS;
  const unsigned missing = parts & ~g->parts;
  if (!missing) return TFGNN_OK;
  TFGNN_REQUIRE(!g->pending, "tfgnn_graph_wait has not been called");
  hipStream_t s = (hipStream_t)stream;
  const bool redo_core = (missing & TFGNN_GRAPH_PART_EDGE_IDS) && g->E > 0;
  // the edge ids were not carried through the sort: run it again with the payload (same order, same arrays - the adjacency
  // lists of tfgnn_graph_create* are read again and must still be alive)
  const CoreScratch cs = core_scratch(redo_core ? g->E : 0, g->L, true);
  const size_t c_bytes = (size_t)(64 + 4 * 2 * SHORT_BINS + 1024) * 4;
  const size_t core_take = (cs.total + 255) & ~(size_t)255;
  const size_t need = core_take + ((missing & TFGNN_GRAPH_PART_COMPACT) ? compact_scratch_bytes(g->R, g->V) : 0);
  char* scratch = nullptr;
  size_t got = 0;
  hipError_t he = dev_alloc((void**)&scratch, need, &got, s);
  if (he != hipSuccess) {
    set_error("tfgnn_graph_ensure: allocation of %zu bytes failed: %s", need, hipGetErrorString(he));
    return TFGNN_ERR_HIP;
  }
  int32_t* counters = (int32_t*)(scratch + cs.counters);
  int rc = TFGNN_OK;
  hipError_t e = hipMemsetAsync(counters, 0, c_bytes, s);
  if (e == hipSuccess && redo_core) rc = run_core(g, true, scratch, cs, s);
  if (e == hipSuccess && !rc && (missing & TFGNN_GRAPH_PART_PLAN_TYPED)) rc = build_plans(g, 0, counters, counters + 64, s);
  if (e == hipSuccess && !rc && (missing & TFGNN_GRAPH_PART_PLAN_NODE)) rc = build_plans(g, 1, counters, counters + 64, s);
  if (e == hipSuccess && !rc && (missing & TFGNN_GRAPH_PART_EDGE_MAPS)) rc = build_edge_maps(g, s);
  if (e == hipSuccess && !rc && (missing & TFGNN_GRAPH_PART_DST_PATTERN)) rc = build_dst_pattern(g, counters + 64 + 4 * 2 * SHORT_BINS, s);
  if (e == hipSuccess && !rc && (missing & TFGNN_GRAPH_PART_COMPACT)) rc = build_compact(g, scratch + core_take, s);
  if (e == hipSuccess && !rc) e = hipMemcpyAsync(g->pinned, counters, 64 * 4, hipMemcpyDeviceToHost, s);
  hipError_t e2 = hipStreamSynchronize(s);
  dev_release(scratch, got);
  if (rc) return rc;
  if (e != hipSuccess || e2 != hipSuccess) {
    set_error("tfgnn_graph_ensure failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return TFGNN_ERR_HIP;
  }
  graph_read_counters(g, missing);
  g->parts |= missing;
  return TFGNN_OK;
}

extern "C" unsigned tfgnn_graph_parts(const tfgnn_graph* g) { return g ? g->parts : 0u; }

extern "C" int tfgnn_graph_wait(tfgnn_graph* g) {
  using namespace tfgnn;
  TFGNN_REQUIRE(g != nullptr, "graph is NULL");
  if (!g->pending) return TFGNN_OK;
  TFGNN_HIP_CHECK(hipEventSynchronize((hipEvent_t)g->event));
  g->pending = false;
  const int32_t* h_counters = (const int32_t*)g->pinned;
  graph_read_counters(g, g->parts);
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
  {
    const int rc = tfgnn::graph_require_parts(g, TFGNN_GRAPH_PART_COMPACT, "tfgnn_graph_nonempty_offsets");
    if (rc) return rc;
  }
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
  {
    unsigned need = 0;
    if (array_id == TFGNN_G_SRC2DST_POS) need = TFGNN_GRAPH_PART_EDGE_MAPS;
    if (array_id == TFGNN_G_EID_BY_DST || array_id == TFGNN_G_EID_BY_SRC) need = TFGNN_GRAPH_PART_EDGE_IDS;
    if (array_id >= TFGNN_G_PATTERN_POS_BY_DST && array_id <= TFGNN_G_PATTERN_TILEMASK_BY_SRC) need = TFGNN_GRAPH_PART_DST_PATTERN;
    if (array_id >= TFGNN_G_NZ_CPOS_BY_DST && array_id <= TFGNN_G_NZ_COL_BY_SRC) need = TFGNN_GRAPH_PART_COMPACT;
    if (need) {
      const int rc = tfgnn::graph_require_parts(g, need, "tfgnn_graph_array");
      if (rc) return rc;
    }
  }
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
    case TFGNN_G_PATTERN_POS_BY_DST: *d_ptr = g->L <= 8 ? g->pat_pos_d : nullptr; *count = g->L <= 8 ? g->V : 0; break;
    case TFGNN_G_PATTERN_NODE_BY_DST: *d_ptr = g->L <= 8 ? g->pat_node_d : nullptr; *count = g->L <= 8 ? g->V : 0; break;
    case TFGNN_G_PATTERN_TILEMASK_BY_DST: *d_ptr = g->L <= 8 ? (const void*)g->pat_tilemask_d : nullptr; *count = g->L <= 8 ? ceil_div(g->V, 128) : 0; break;
    case TFGNN_G_PATTERN_NODE_BY_SRC: *d_ptr = g->L <= 8 ? g->pat_node_s : nullptr; *count = g->L <= 8 ? g->V : 0; break;
    case TFGNN_G_PATTERN_TILEMASK_BY_SRC: *d_ptr = g->L <= 8 ? (const void*)g->pat_tilemask_s : nullptr; *count = g->L <= 8 ? ceil_div(g->V, 128) : 0; break;
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
