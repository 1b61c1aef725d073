// Internal layout of the graph handle (shared by graph.hip and spmm.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace tfgnn {

// Rows longer than LONG_ROW_THRESHOLD edges are not walked by a single lane group: they are cut
// into items of ITEM_CHUNK consecutive edges, one workgroup per item (deterministic partial sums,
// combined in item order).  Keeps the tail of the gather kernel bounded on skewed (R-MAT) graphs.
constexpr int LONG_ROW_THRESHOLD = 32;        // node views (all edge types of a node in one row)
constexpr int ITEM_CHUNK = 512;
constexpr int LONG_ROW_THRESHOLD_TYPED = 48;  // typed views (one row per (node, type) bucket)
constexpr int ITEM_CHUNK_TYPED = 512;

struct CsrPlan {
  int32_t long_threshold = LONG_ROW_THRESHOLD;  // values the plan was built with (env-tunable for probes)
  int32_t item_chunk_edges = ITEM_CHUNK;
  int32_t num_items = 0;     // host copies of the device counters
  int32_t num_multi = 0;     // rows with more than one item
  int32_t num_partials = 0;  // items belonging to multi-item rows (scratch slots)
  int32_t* item_row = nullptr;    // [num_items]
  int32_t* item_chunk = nullptr;  // [num_items] chunk index within the row
  int32_t* item_slot = nullptr;   // [num_items] scratch slot, or -1: the item is the whole row
  int32_t* multi_row = nullptr;   // [num_multi]
  int32_t* multi_base = nullptr;  // [num_multi] first scratch slot
  int32_t* multi_n = nullptr;     // [num_multi] number of items
  // rows of at most long_threshold edges, longest first: the lane groups of a wave get rows of (almost) equal
  // length, so a wave is not held up by its longest row (65 % -> ~100 % lane use on an R-MAT batch)
  int32_t* short_rows = nullptr;  // [num_short]
  int32_t num_short = 0;
};
constexpr int SHORT_BINS = 257;  // long_threshold <= 256

// Non-empty buckets in type-major order (all non-empty (node, type 0) rows, then type 1, ...): lets the
// dense per-relation multiply run only over buckets that received at least one edge.
struct CompactBuckets {
  int32_t num_nz = 0;
  int32_t* cpos = nullptr;        // [R]   bucket row -> compact index, or -1
  int32_t* nzrow = nullptr;       // [nz]  compact index -> bucket row (node * L + type)
  int32_t* nz_node = nullptr;     // [nz]  compact index -> node
  int32_t* nz_off = nullptr;      // [L+1] device: first compact index of every type
  int32_t* nodeptr_nz = nullptr;  // [V+1] per node: range of its non-empty buckets in col_nz
  int32_t* col_nz = nullptr;      // [nz]  compact indices grouped by node (type ascending)
  int32_t h_nz_off[257] = {0};    // host copy (L <= 256), filled by tfgnn_graph_wait
};

struct GraphView {
  const int32_t* rowptr = nullptr;
  int64_t num_rows = 0;
  const int32_t* col = nullptr;
  CsrPlan plan;
};

}  // namespace tfgnn

struct tfgnn_graph {
  int L = 0;
  int64_t V = 0, E = 0, R = 0;
  void* slab = nullptr;
  int32_t *rowptr_d = nullptr, *col_d = nullptr, *eid_d = nullptr, *coll_d = nullptr;
  int32_t *rowptr_s = nullptr, *col_s = nullptr, *eid_s = nullptr, *coll_s = nullptr;
  int32_t *nodeptr_d = nullptr, *nodeptr_s = nullptr, *src2dst = nullptr, *dst2src = nullptr, *tgt_d = nullptr;
  int32_t* eid2pos = nullptr;  // edge id -> position in the by-target order
  // part DST_PATTERN: nodes ordered by which of their by-target buckets are empty (graph.hip)
  int32_t *pat_pos_d = nullptr, *pat_node_d = nullptr, *pat_rowmap_d = nullptr;
  uint8_t* pat_tilemask_d = nullptr;
  int32_t* pat_node_s = nullptr;  // ... and of their by-source buckets (position -> node, tile masks; round 5)
  uint8_t* pat_tilemask_s = nullptr;
  unsigned parts = 0;          // TFGNN_GRAPH_PART_* bits that have been built (the sort and the per-edge arrays always are)
  int sec_bits = 0, total_bits = 0;        // composite key: (bucket << sec_bits) | column
  std::vector<const int32_t*> h_adj;       // the adjacency lists of the creation call (device pointers) and their
  std::vector<int64_t> h_off;              // offsets in the concatenation: tfgnn_graph_ensure(EDGE_IDS) sorts again
  float *invdeg_d = nullptr, *invdeg_edge_s = nullptr, *invdeg_edge_d = nullptr;
  tfgnn::GraphView views[4];  // tfgnn_graph_view order
  tfgnn::CompactBuckets compact[2];  // 0: by target, 1: by source
  // asynchronous build state (tfgnn_graph_create_async / tfgnn_graph_wait)
  size_t slab_bytes = 0;
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* pinned = nullptr;  // host staging: [0,256) counters read back, then pointer / offset tables
  size_t pinned_bytes = 0;
  void* event = nullptr;   // hipEvent_t recorded after the last build command
  bool pending = false;
};

namespace tfgnn {
// TFGNN_OK, or TFGNN_ERR_INVALID_ARGUMENT (error text set) when the handle lacks one of the parts in `need`
int graph_require_parts(const tfgnn_graph* g, unsigned need, const char* who);
}  // namespace tfgnn
