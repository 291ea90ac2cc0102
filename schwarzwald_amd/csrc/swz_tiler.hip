// swz_tiler.hip -- the multi-batch tiler (BASELINE config 5's single-GPU shape).
//
// The reference tiles a data set larger than internal_cache_size (executable/main.cpp:233-236, default 10 M points)
// batch by batch on ONE TilingAlgorithm object: every node a batch reaches re-reads the points earlier batches
// persisted under its name, re-keys them relative to the node (read_pnts_from_disk, core/tiling/TilingAlgorithms.cpp:
// 50-109), merges them with the new points (merge_node_data_sorted, core/tiling/Node.cpp:4-22: std::merge, new
// points first on equal keys), samples the union with AlwaysAdhereToMinSpacing (:272-275), REPLACES the node's file
// with the taken points (BinaryPersistence.h:46-57) and hands the rest -- new and displaced old points alike -- to
// the children.  Nodes no point of the batch reaches are not touched.
//
// Here the "files" are a device-resident node store: per octree level the (key, point id) entries of its nodes' files,
// each file contiguous and in file order, found through a node table (StoreLevel); positions (and attribute columns) of
// all points ever added live in pools indexed by point id (= running index over all batches).  A batch runs the same
// level-synchronous loop as a single batch (swz_level.hip); per level the files of the nodes the active set touches are
// copied out and merged in, and the level's taken points are appended as those nodes' new files -- carrying the key the
// reference computes when it reads a file back, so that the re-keying is paid once per stored entry, not per pull.  FAST (TilingAlgorithmV3) later iterations
// (:1362-1453, 1620-1659: per-thread chunks sorted, split at the start level, k-way merged, earlier chunk first on
// ties) are the stable sort of the batch; finalize (:1661-1784) rebuilds the skipped levels from the store.
//
// Staging (BASELINE config 5): swz_tiler_stage_batch copies the NEXT batch from pinned host memory straight into
// its final place in the pools with hipMemcpyAsync on a copy stream while the current batch is tiled.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

static const uint32_t TILER_ATTR_BYTES[SWZ_ATTR_COUNT] = {3, 12, 2, 1, 1, 8, 1, 1, 2, 1, 1, 1};

// The files of one octree level.  Two forms:
//   linear: side `cur` holds exactly the `cnt` live entries, node after node in node order (what every reader but the
//           level loop wants: export, node table, finalize, re-rooting);
//   log:    a batch must not move the files of the nodes it does not reach (batches of a real data set -- LAS tiles --
//           reach a small part of the tree), so the level loop only APPENDS the new versions of the files it rewrites at
//           `end` and keeps a node table {node key, offset, count} that says where each node's current file lies; the
//           old versions stay behind as garbage until the side is full, then the live files are gathered into the other
//           side (store_compact).  Per batch and level the store costs what the batch pulls and writes, not what it holds.
// store_table() / store_linearize() convert between the two.
struct StoreLevel {
  uint64_t* key[2] = {nullptr, nullptr};
  uint32_t* gid[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  int cur = 0;
  uint32_t cnt = 0;          // live entries
  uint32_t end = 0;          // entries of side `cur` in use (live + garbage)
  bool linear = true;
  bool table_valid = false;
  uint64_t* nkey[2] = {nullptr, nullptr};  // node table, ascending by node key (the key with the bits below the node cleared)
  uint64_t* noff[2] = {nullptr, nullptr};
  uint32_t* ncnt[2] = {nullptr, nullptr};
  int ncur = 0;
  uint32_t nn = 0;
  // every entry carries the key read_pnts_from_disk would give it (relative to its NODE's bounds): files written by the
  // level loop do (TakeStoreG), files written by finalize / re-rooting do not and are re-keyed when they are pulled
  bool rekeyed = true;
};

// One batch on its way through the levels.  `as` (kept beside it) is the active set handed down -- new points and
// displaced old ones --, Morton sorted.
struct BatchWork {
  uint32_t n = 0;          // points of the batch
  uint32_t wused = 0;      // working-pool entries in use
  uint32_t wcap = 0;
  double *wx = nullptr, *wy = nullptr, *wz = nullptr;  // positions by working index -- filled on demand, see work_need_positions
  bool have_pos = false;
  int8_t* wlevel = nullptr;
  uint32_t* wgid = nullptr;
  uint64_t* surv_key[2] = {nullptr, nullptr};
  uint32_t* surv_idx[2] = {nullptr, nullptr};
  int which = 0;
};

// what a shard of a multi-GPU batch knows about the other shards (root node only)
struct ShardRoot {
  bool active = false;
  bool sample = false;        // the root samples (global counts), else it takes everything
  const double* ghost_xyz = nullptr;
  uint32_t ghosts = 0;
};

}  // namespace swz

struct swz_tiler {
  swz_ctx* c = nullptr;
  double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  swz_tile_params p{};
  // pools by point id
  double* pool_xyz = nullptr;
  void* pool_attr[SWZ_ATTR_COUNT] = {nullptr};
  uint32_t attr_mask = 0;   // attribute columns the pools hold (fixed by the first staged batch)
  size_t pool_cap = 0;      // points
  uint32_t total = 0;       // points tiled so far
  uint32_t staged_total = 0;  // points copied (or being copied) into the pools
  std::vector<uint32_t> staged_sizes;  // batches staged and not yet tiled (at most 2)
  std::vector<hipEvent_t> staged_events;
  hipStream_t copy_stream = nullptr;
  swz::StoreLevel lv[22];  // index = node level + 1
  int fast_start = -1;
  bool finalized = false;
  uint64_t batches = 0;
  uint64_t rekey_inversions = 0;
  uint64_t staged_bytes = 0;
  double staged_wait_ms = 0.0;  // time swz_tiler_tile_staged had to WAIT for its copy (0 when fully overlapped)
  // a batch between swz_tiler_shard_begin_device and swz_tiler_shard_finish
  bool batch_open = false;
  // A batch that fails part-way leaves levels of the node store merged and its survivors lost: the tiler is poisoned
  // and every later call reports SWZ_ERR_TILER_FAILED (the store must not be read or extended any more).
  bool failed = false;
  std::string failed_why;
  bool shard_fast = false;  // a FAST batch of a sharded data set is open: the start level comes from the driver
  swz::BatchWork bw;
  swz::ActiveSet as;
  int next_level = -1;
  uint64_t acc_visited = 0, acc_nodes = 0;
  uint32_t acc_rounds = 0, acc_levels = 0;
  int acc_max_level = -1;
};

static int tiler_guard(swz_tiler* t) {
  // (every call of the tiler's API starts a scratch epoch: what earlier calls asked for and nothing holds on to -- level
  // scratch, and the per-batch "tl_*" buffers once no batch is open -- may be freed when the device runs out of memory)
  if (!t->failed) {
    t->c->next_scratch_epoch();
    return SWZ_OK;
  }
  return t->c->fail(SWZ_ERR_TILER_FAILED, "swz_tiler: an earlier batch failed part-way (" + t->failed_why +
                                            "); the node store is incomplete -- destroy the tiler");
}
// st != SWZ_OK after the tiler started to change its state: remember it
static int tiler_poison(swz_tiler* t, int st) {
  if (st != SWZ_OK && !t->failed) {
    t->failed = true;
    t->failed_why = t->c->err;
    t->batch_open = false;
  }
  return st;
}

namespace swz {

// ---------------------------------------------------------------------------------------------- kernels
__global__ __launch_bounds__(256) void tl_wgid_kernel(const uint32_t* __restrict__ perm, uint32_t n, uint32_t base,
                                                      uint32_t* __restrict__ wgid) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) wgid[i] = base + perm[i];
}

// store entry j lies in a node the active set reaches <=> some active key has the same node prefix
// (the entries of a workgroup are consecutive store entries, sorted by node prefix: the searches of its first and last
// entry bracket all others, as in tl_merge_rank_kernel)
__global__ __launch_bounds__(256) void tl_touch_kernel(const uint64_t* __restrict__ skey, uint32_t cnt,
                                                       const uint64_t* __restrict__ akey, uint32_t m, uint32_t nsh,
                                                       uint8_t* __restrict__ touch) {
  __shared__ uint32_t s_lo, s_hi;
  const uint32_t j0 = blockIdx.x * 256u, j = j0 + threadIdx.x;
  const uint32_t last = (cnt - j0) > 256u ? j0 + 255u : cnt - 1u;
  auto lower = [&](uint32_t lo, uint32_t hi, uint64_t prefix) {
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if ((akey[mid] >> nsh) < prefix) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  if (threadIdx.x < 2) {  // (two lanes of one wavefront: see tl_merge_rank_kernel)
    const uint32_t r = lower(0u, m, skey[threadIdx.x ? last : j0] >> nsh);
    if (threadIdx.x) s_hi = r; else s_lo = r;
  }
  __syncthreads();
  if (j >= cnt) return;
  const uint64_t prefix = skey[j] >> nsh;
  const uint32_t lo = lower(s_lo, s_hi, prefix);
  touch[j] = (lo < m && (akey[lo] >> nsh) == prefix) ? 1 : 0;
}
struct TouchF {
  const uint8_t* touch;
  __device__ uint32_t operator()(uint32_t i) const { return touch[i]; }
};
struct SplitG {  // touched entries -> cached set, the others stay
  const uint64_t* skey;
  const uint32_t* sgid;
  uint64_t* ckey;
  uint32_t* cgid;
  uint64_t* rkey;
  uint32_t* rgid;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t t) const {
    if (t) {
      ckey[excl] = skey[i];
      cgid[excl] = sgid[i];
    } else {
      rkey[i - excl] = skey[i];
      rgid[i - excl] = sgid[i];
    }
  }
};

// static_cast<uint64_t>(double) the way x86-64 gcc compiles it for the reference (cvttsd2si): values in (-1, 0)
// give 0, values <= -1 wrap to huge numbers (which std::min then turns into 2^21 - 1); formally undefined, but it is
// what calculate_morton_index (OctreeAlgorithms.h:76-79) does for a point outside the box it is indexed against.
__device__ __forceinline__ uint64_t cvt_u64_like_x86(double v) {
  return v < 0.0 ? (uint64_t)(int64_t)v : (uint64_t)v;
}
// calculate_morton_index<21>(p, box) without clamping the position -- OctreeAlgorithms.h:64-87
__device__ __forceinline__ uint64_t morton_in_box(double x, double y, double z, const Box& b) {
  const double two21 = 2097152.0;
  const double sx = two21 / (b.maxx - b.minx), sy = two21 / (b.maxy - b.miny), sz = two21 / (b.maxz - b.minz);
  const double nx = (x - b.minx) * sx, ny = (y - b.miny) * sy, nz = (z - b.minz) * sz;
  const uint64_t lim = (1ull << 21) - 1ull;
  uint64_t bx = cvt_u64_like_x86(nx), by = cvt_u64_like_x86(ny), bz = cvt_u64_like_x86(nz);
  bx = bx < lim ? bx : lim;
  by = by < lim ? by : lim;
  bz = bz < lim ? bz : lim;
  return expand_bits_by_3(bz) | (expand_bits_by_3(by) << 1) | (expand_bits_by_3(bx) << 2);
}

// read_pnts_from_disk, TilingAlgorithms.cpp:80-99: idx = node.morton_index; levels node.level+1 .. 20 are levels
// 0 .. of the index of the position inside node.bounds (bounds by descending octant by octant from the root).
// The result depends on the point's position and on the NODE (the top level+1 digits of `old`) only, not on the lower
// digits of `old`: re-keying a re-keyed entry of the same node changes nothing.
__device__ __forceinline__ uint64_t rekey_one(uint64_t old, const double* __restrict__ pool, size_t g, const Box& root, int level) {
  const Box nb = bounds_from_key(old, root, level + 1);
  const uint64_t rel = morton_in_box(pool[3 * g], pool[3 * g + 1], pool[3 * g + 2], nb);
  const uint32_t start_level = (uint32_t)(level + 1);
  const uint64_t prefix = start_level == 0 ? 0ull : ((old >> level_shift(level)) << level_shift(level));
  return prefix | (rel >> (3u * start_level));
}
__global__ __launch_bounds__(256) void tl_rekey_kernel(uint64_t* __restrict__ ckey, const uint32_t* __restrict__ cgid,
                                                       uint32_t nc, const double* __restrict__ pool, Box root,
                                                       int level) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nc) return;
  ckey[j] = rekey_one(ckey[j], pool, cgid[j], root, level);
}

// index_points<21>(root bounds, ClampToBounds) on a COPY of the positions (reconstruct_single_node :1682-1688)
__global__ __launch_bounds__(256) void tl_reencode_kernel(const uint32_t* __restrict__ gid, uint32_t n,
                                                          const double* __restrict__ pool, Box b,
                                                          uint64_t* __restrict__ keys) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const size_t g = gid[j];
  double x = pool[3 * g], y = pool[3 * g + 1], z = pool[3 * g + 2];
  const bool inside = (x >= b.minx && x <= b.maxx && y >= b.miny && y <= b.maxy && z >= b.minz && z <= b.maxz);
  if (!inside) {
    x = (b.minx < x) ? x : b.minx;
    x = (x < b.maxx) ? x : b.maxx;
    y = (b.miny < y) ? y : b.miny;
    y = (y < b.maxy) ? y : b.maxy;
    z = (b.minz < z) ? z : b.minz;
    z = (z < b.maxz) ? z : b.maxz;
  }
  keys[j] = morton_in_box(x, y, z, b);
}

// pairs of neighbours inside one node (same prefix >> nsh) whose keys descend
__global__ __launch_bounds__(256) void tl_inversion_kernel(const uint64_t* __restrict__ key, uint32_t n, uint32_t nsh,
                                                           uint32_t* __restrict__ count) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  const bool bad = j > 0 && j < n && (key[j] >> nsh) == (key[j - 1] >> nsh) && key[j] < key[j - 1];
  const uint64_t b = __ballot(bad);
  if (lane_id() == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
}

// point ids of the pulled points into the working pool, behind the batch's own points -- and their positions (SoA) when
// the working pool keeps positions (X != null; see work_need_positions)
__global__ __launch_bounds__(256) void tl_fill_kernel(const uint32_t* __restrict__ cgid, uint32_t nc,
                                                      const double* __restrict__ pool, double* __restrict__ X,
                                                      double* __restrict__ Y, double* __restrict__ Z,
                                                      uint32_t* __restrict__ wgid) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nc) return;
  const size_t g = cgid[j];
  if (wgid) wgid[j] = (uint32_t)g;
  if (!X || g == 0xFFFFFFFFu) return;  // (the ghosts of a sharded root have no id: their positions were written with them)
  X[j] = pool[3 * g];
  Y[j] = pool[3 * g + 1];
  Z[j] = pool[3 * g + 2];
}

// std::merge(first, second, comp = key <): elements of `first` precede equal elements of `second`.
// Keys are compared after >> sh (sh = node shift merges by node only: merge_node_data_unsorted's "new ++ cached").
// Stable merge of two sorted runs by rank: an element's place is its own index plus the number of elements of the other
// run that go before it (first run: strictly smaller keys; second run: smaller or equal -- the first run wins ties, like
// std::merge).  The 256 consecutive elements of a workgroup are sorted, so the ranks of its first and last element bracket
// all others: two searches over the whole other run per workgroup, then every thread searches that bracket only -- out of
// LDS when it holds at most 1024 keys (runs of similar length interleave: a few hundred), instead of ~25 dependent probes
// all over a run of tens of millions of keys per element.
constexpr uint32_t TL_MERGE_LDS = 2048;
constexpr uint32_t TL_MERGE_IPT = 4;                    // elements per thread: the two searches over the whole other run that
constexpr uint32_t TL_MERGE_TILE = 256 * TL_MERGE_IPT;  // open a workgroup (~23 dependent loads each) serve 1024 elements
template <bool UPPER>
__device__ __forceinline__ uint32_t tl_rank(const uint64_t* __restrict__ k, uint32_t lo, uint32_t hi, uint64_t ks, uint32_t sh) {
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    const uint64_t v = k[mid] >> sh;
    if (UPPER ? (v <= ks) : (v < ks)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// run `a` (n_a elements, values va or `base + index`) against the other run `b`; UPPER: a is the second run
template <bool UPPER>
__global__ __launch_bounds__(256) void tl_merge_rank_kernel(const uint64_t* __restrict__ ka, const uint32_t* __restrict__ va, uint32_t na,
                                                            const uint64_t* __restrict__ kb, uint32_t nb, uint32_t sh, uint32_t base,
                                                            uint64_t* __restrict__ ok, uint32_t* __restrict__ ov) {
  __shared__ uint32_t s_lo, s_hi;
  __shared__ uint64_t sk[TL_MERGE_LDS];
  const uint32_t tid = threadIdx.x;
  const uint32_t i0 = blockIdx.x * TL_MERGE_TILE;
  const uint32_t last = (na - i0) > TL_MERGE_TILE ? i0 + TL_MERGE_TILE - 1u : na - 1u;
  // (both searches by two lanes of the first wavefront.  With the second one on thread 64 -- alone in its wavefront, so
  // hipcc 7.2 turns its search into scalar loads -- and more than one element per thread, the shift count of the loops
  // below came out of a register that only some wavefronts had set: wrong ranks for threads 128-255.  Found with a
  // stand-alone copy of this kernel against std::merge.)
  if (tid < 2) {
    const uint32_t r = tl_rank<UPPER>(kb, 0u, nb, ka[tid ? last : i0] >> sh, sh);
    if (tid) s_hi = r; else s_lo = r;
  }
  __syncthreads();
  const uint32_t lo = s_lo, hi = s_hi;
  const bool in_lds = hi - lo <= TL_MERGE_LDS;
  if (in_lds)
    for (uint32_t j = tid; j < hi - lo; j += 256u) sk[j] = kb[lo + j] >> sh;
  __syncthreads();
  for (uint32_t q = 0; q < TL_MERGE_IPT; ++q) {
    const uint32_t i = i0 + q * 256u + tid;
    if (i >= na) break;
    const uint64_t k = ka[i];
    const uint64_t ks = k >> sh;
    uint32_t r;
    if (in_lds) r = lo + tl_rank<UPPER>(sk, 0u, hi - lo, ks, 0u);
    else r = tl_rank<UPPER>(kb, lo, hi, ks, sh);
    ok[i + r] = k;
    ov[i + r] = va ? va[i] : base + i;
  }
}

struct TakenF {
  const uint8_t* taken;
  __device__ uint32_t operator()(uint32_t i) const { return taken[i] ? 1u : 0u; }
};
struct TakeG {  // the node's new file content: taken points in the order of the merged range
  const uint64_t* mkey;
  const uint32_t* midx;
  const uint32_t* wgid;
  uint64_t* tkey;
  uint32_t* tgid;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t t) const {
    if (!t) return;
    tkey[excl] = mkey[i];
    tgid[excl] = wgid[midx ? midx[i] : i];
  }
};

struct HeadF {
  const uint64_t* key;
  uint32_t nsh;
  __device__ uint32_t operator()(uint32_t i) const { return (i == 0 || (key[i] >> nsh) != (key[i - 1] >> nsh)) ? 1u : 0u; }
};
struct HeadG {
  const uint64_t* key;
  uint32_t nsh;
  uint32_t* head_pos;
  uint64_t* head_key;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t h) const {
    if (!h) return;
    head_pos[excl] = i;
    head_key[excl] = nsh >= 63 ? 0ull : ((key[i] >> nsh) << nsh);
  }
};

// The level loop's store step: the taken points of the merged range become the new files of their nodes, written
// straight behind the files the side already holds.  An entry that did not come out of this level's files (a point of
// the batch, or one an ancestor handed down) gets the key the reference would compute when it reads the file back
// (rekey_one) -- once, here, instead of with every later batch that pulls the file; entries pulled from this level's
// files carry that key already.  The first `ghosts` taken entries are a sharded root's ghosts: not part of the file.
struct TakeStoreG {
  const uint64_t* mkey;
  const uint32_t* midx;
  const uint32_t* wgid;
  uint32_t pull_lo, pull_hi;  // working indices of what this level pulled
  uint32_t ghosts;
  const double* pool;
  Box root;
  int level;
  uint64_t* okey;
  uint32_t* ogid;
  uint32_t nsh;
  uint32_t* texcl;  // [i]: taken entries in front of merged entry i, written where a node starts (-> the heads of the new files)
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t t) const {
    if (i == 0 || (mkey[i] >> nsh) != (mkey[i - 1] >> nsh)) texcl[i] = excl;
    if (!t || excl < ghosts) return;
    const uint32_t w = midx ? midx[i] : i;
    const uint32_t g = wgid[w];
    uint64_t k = mkey[i];
    if ((w < pull_lo || w >= pull_hi) && g != 0xFFFFFFFFu) k = rekey_one(k, pool, g, root, level);
    okey[excl - ghosts] = k;
    ogid[excl - ghosts] = g;
  }
};

// ---- node table of a level store (log form)
__global__ __launch_bounds__(256) void tl_table_build_kernel(const uint64_t* __restrict__ hk, const uint32_t* __restrict__ hp,
                                                             uint32_t heads, uint32_t cnt, uint64_t* __restrict__ nkey,
                                                             uint64_t* __restrict__ noff, uint32_t* __restrict__ ncnt) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= heads) return;
  nkey[j] = hk[j];
  noff[j] = hp[j];
  ncnt[j] = (j + 1 < heads ? hp[j + 1] : cnt) - hp[j];
}
__device__ __forceinline__ uint32_t tl_lower_u64(const uint64_t* __restrict__ a, uint32_t n, uint64_t k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (a[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the files of the nodes the active set reaches: head j of the active set -> its node's file (if it has one)
struct PullCntF {
  const uint64_t* hk;
  const uint64_t* nkey;
  const uint32_t* ncnt;
  uint32_t nn;
  __device__ uint32_t find(uint64_t k) const {
    const uint32_t r = tl_lower_u64(nkey, nn, k);
    return (r < nn && nkey[r] == k) ? r : 0xFFFFFFFFu;
  }
  __device__ uint32_t operator()(uint32_t j) const {
    const uint32_t r = find(hk[j]);
    return r == 0xFFFFFFFFu ? 0u : ncnt[r];
  }
};
struct PullSegG {
  PullCntF f;
  const uint64_t* noff;
  uint32_t* poff;
  uint64_t* psrc;
  uint8_t* touched;
  __device__ void operator()(uint32_t j, uint32_t excl, uint32_t) const {
    const uint32_t r = f.find(f.hk[j]);
    poff[j] = excl;
    psrc[j] = r == 0xFFFFFFFFu ? 0ull : noff[r];
    if (r != 0xFFFFFFFFu) touched[r] = 1;
  }
};
// segments j = 0 .. segs-1 of a source array, segment j = [psrc[j], +len_j) with len_j = poff[j+1] - poff[j]
// (poff[segs] = total), copied one behind the other: output element e belongs to the last segment that starts at or
// before e.  A workgroup's 256 consecutive outputs lie in consecutive segments: two searches over all of poff bracket
// them, every thread then searches the bracket (out of LDS when it is short).
constexpr uint32_t TL_SEG_LDS = 1024;
__device__ __forceinline__ uint32_t tl_upper_u32(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t k) {
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (a[mid] <= k) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__global__ __launch_bounds__(256) void tl_gather_files_kernel(const uint32_t* __restrict__ poff, const uint64_t* __restrict__ psrc,
                                                              uint32_t segs, uint32_t total, const uint64_t* __restrict__ skey,
                                                              const uint32_t* __restrict__ sgid, uint64_t* __restrict__ okey,
                                                              uint32_t* __restrict__ ogid) {
  __shared__ uint32_t s_lo, s_hi;
  __shared__ uint32_t so[TL_SEG_LDS];
  const uint32_t tid = threadIdx.x;
  const uint32_t e0 = blockIdx.x * 256u, e = e0 + tid;
  const uint32_t last = (total - e0) > 256u ? e0 + 255u : total - 1u;
  if (tid < 2) {  // (two lanes of one wavefront: see tl_merge_rank_kernel)
    const uint32_t r = tl_upper_u32(poff, 0u, segs, tid ? last : e0) - 1u;
    if (tid) s_hi = r; else s_lo = r;
  }
  __syncthreads();
  const uint32_t lo = s_lo, span = s_hi - s_lo + 1u;
  const bool in_lds = span <= TL_SEG_LDS;
  if (in_lds)
    for (uint32_t j = tid; j < span; j += 256u) so[j] = poff[lo + j];
  __syncthreads();
  if (e >= total) return;
  const uint32_t j = in_lds ? lo + tl_upper_u32(so, 0u, span, e) - 1u : tl_upper_u32(poff, lo, lo + span, e) - 1u;
  const uint64_t src = psrc[j] + (e - poff[j]);
  okey[e] = skey[src];
  ogid[e] = sgid[src];
}
struct UntouchedF {
  const uint8_t* touched;  // null: every node counts
  __device__ uint32_t operator()(uint32_t i) const { return (touched && touched[i]) ? 0u : 1u; }
};
struct TableFilterG {
  const uint64_t* nkey;
  const uint64_t* noff;
  const uint32_t* ncnt;
  uint64_t* fkey;
  uint64_t* foff;
  uint32_t* fcnt;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t keep) const {
    if (!keep) return;
    fkey[excl] = nkey[i];
    foff[excl] = noff[i];
    fcnt[excl] = ncnt[i];
  }
};
struct SegCntF {
  const uint32_t* cnt;
  __device__ uint32_t operator()(uint32_t i) const { return cnt[i]; }
};
struct SegMoveG {  // the files of a table, gathered one behind the other: where each goes, where it came from
  uint64_t* off;
  uint32_t* poff;
  uint64_t* psrc;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t) const {
    poff[i] = excl;
    psrc[i] = off[i];
    off[i] = excl;
  }
};
// the heads of the files a level step has just written: node j of the merged range starts at merged entry nstart[j], its
// file at the number of taken entries in front of that (texcl, TakeStoreG) -- no scan over the new files
__global__ __launch_bounds__(256) void tl_new_heads_kernel(const uint64_t* __restrict__ mkey, const uint32_t* __restrict__ nstart,
                                                           const uint32_t* __restrict__ texcl, uint32_t nodes, uint32_t nsh, uint32_t ghosts,
                                                           uint64_t* __restrict__ hk, uint32_t* __restrict__ hp) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nodes) return;
  const uint32_t i = nstart[j];
  hk[j] = nsh >= 63u ? 0ull : ((mkey[i] >> nsh) << nsh);
  const uint32_t e = texcl[i];
  hp[j] = e > ghosts ? e - ghosts : 0u;
}
// two node tables with disjoint keys, both ascending, into one: the entries the batch left alone (f*) and the heads of
// the files it wrote (keys hk at positions hp of the `added` entries appended at `base`)
__global__ __launch_bounds__(256) void tl_table_merge_kernel(const uint64_t* __restrict__ fkey, const uint64_t* __restrict__ foff,
                                                             const uint32_t* __restrict__ fcnt, uint32_t nf,
                                                             const uint64_t* __restrict__ hk, const uint32_t* __restrict__ hp,
                                                             uint32_t heads, uint32_t added, uint64_t base,
                                                             uint64_t* __restrict__ okey, uint64_t* __restrict__ ooff,
                                                             uint32_t* __restrict__ ocnt) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < nf) {
    const uint32_t o = i + tl_lower_u64(hk, heads, fkey[i]);
    okey[o] = fkey[i];
    ooff[o] = foff[i];
    ocnt[o] = fcnt[i];
  } else if (i < nf + heads) {
    const uint32_t j = i - nf;
    const uint32_t o = j + tl_lower_u64(fkey, nf, hk[j]);
    okey[o] = hk[j];
    ooff[o] = base + hp[j];
    ocnt[o] = (j + 1 < heads ? hp[j + 1] : added) - hp[j];
  }
}

// ---- re-rooting (tile_node, TilingAlgorithms.cpp:444-483)
// calculate_morton_index<21>(position, new_root.bounds), no clamp (:470-473)
__global__ __launch_bounds__(256) void rr_encode_kernel(const uint32_t* __restrict__ idx, uint32_t m,
                                                        const double* __restrict__ X, const double* __restrict__ Y,
                                                        const double* __restrict__ Z, Box b, uint64_t* __restrict__ keys) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t p = idx[i];
  keys[i] = morton_in_box(X[p], Y[p], Z[p], b);
}
// partition_points_into_child_octants (OctreeAlgorithms.h:240-265): the range of octant o ends at the first element
// at or behind its start whose octant at the given level is > o (std::find_if) -- on re-rooted keys split at the
// ABSOLUTE child level (TilingAlgorithms.cpp:124-125) the octants do not ascend, so this is not a digit histogram
__global__ __launch_bounds__(256) void rr_split_kernel(const uint64_t* __restrict__ keys, uint32_t m, uint32_t shift,
                                                       uint32_t octant, uint32_t* __restrict__ bounds) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t start = bounds[octant];
  const bool hit = i < m && i >= start && (uint32_t)((keys[i] >> shift) & 7u) > octant;
  const uint64_t b = __ballot(hit);
  if (b && lane_id() == (uint32_t)(__ffsll((unsigned long long)b) - 1)) atomicMin(&bounds[octant + 1], i);
}
__global__ void rr_split_init_kernel(uint32_t* bounds, uint32_t m) {
  if (threadIdx.x < 9) bounds[threadIdx.x] = threadIdx.x == 0 ? 0u : m;
}
// bounds[o + 1] must not lie before bounds[o] when nothing was found behind it (it stays m) -- nothing to fix up
__global__ __launch_bounds__(256) void rr_iota_kernel(uint32_t* __restrict__ out, uint32_t m) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] = i;
}
struct TakeNodeG {  // taken points of ONE node in the order of the range; every entry carries the node's key
  const uint32_t* idx;
  const uint32_t* wgid;
  uint64_t node_key;
  uint64_t* tkey;
  uint32_t* tgid;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t t) const {
    if (!t) return;
    tkey[excl] = node_key;
    tgid[excl] = wgid[idx[i]];
  }
};
struct AllF {
  __device__ uint32_t operator()(uint32_t) const { return 1u; }
};

__global__ __launch_bounds__(256) void tl_fill_level_kernel(int8_t* __restrict__ out, uint32_t n, int8_t v) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

// ---------------------------------------------------------------------------------------------- host helpers
static Box root_box(const swz_tiler* t) { return Box{t->bmin[0], t->bmin[1], t->bmin[2], t->bmax[0], t->bmax[1], t->bmax[2]}; }

// The node store and the pools live in the context's grow-only workspace under fixed names, so a tiler created after
// another one on the same context reuses the memory (hipMalloc / hipFree of multi-GB blocks were measured to stall for
// seconds now and then).  One tiler per context at a time.
static int grow_preserving(swz_ctx* c, const char* name, size_t bytes, size_t keep, void** out, bool spill);
static int store_reserve(swz_ctx* c, StoreLevel& s, int level_index, int which, size_t count) {
  if (s.cap[which] >= count && s.key[which]) return SWZ_OK;
  const size_t want = count + count / 4 + 1024;
  const std::string kn = "tiler_store_key_" + std::to_string(level_index) + "_" + std::to_string(which);
  const std::string gn = "tiler_store_gid_" + std::to_string(level_index) + "_" + std::to_string(which);
  // (the side being written holds nothing that is still needed: nothing is kept.  Like the pools, a store side that finds
  // no device memory -- or would push the workspace over SWZ_TILER_DEVICE_BUDGET_MB -- is placed in mapped pinned host
  // memory: the merges then stream through it over the host link, slowly, but a data set whose node store outgrows the
  // device still tiles.  The growth policy of the workspace (twice the old capacity) applies here as well.)
  const size_t old_k = c->bufs[kn].cap / sizeof(uint64_t);
  const size_t grown = std::max(want, std::min<size_t>(2 * old_k, want + (size_t(1) << 27)));
  void *pk = nullptr, *pg = nullptr;
  SWZ_TRY(grow_preserving(c, kn.c_str(), grown * sizeof(uint64_t), 0, &pk, true));
  SWZ_TRY(grow_preserving(c, gn.c_str(), grown * sizeof(uint32_t), 0, &pg, true));
  s.key[which] = static_cast<uint64_t*>(pk);
  s.gid[which] = static_cast<uint32_t*>(pg);
  s.cap[which] = std::min(c->bufs[kn].cap / sizeof(uint64_t), c->bufs[gn].cap / sizeof(uint32_t));
  return SWZ_OK;
}

static void store_free(StoreLevel& s) { s = StoreLevel{}; }

// side `which` has just been written as a whole: `cnt` entries, node after node
static void store_written_linear(StoreLevel& s, int which, uint32_t cnt, bool rekeyed) {
  s.cur = which;
  s.cnt = s.end = cnt;
  s.linear = true;
  s.table_valid = false;
  s.nn = 0;
  s.rekeyed = rekeyed || cnt == 0;
}

static int read_u32(swz_ctx* c, const uint32_t* d, uint32_t* h);
static int table_reserve(swz_ctx* c, StoreLevel& s, int level_index, int which, size_t count) {
  // (part of the store: placed like its sides -- SWZ_TILER_SPILL=host leaves nothing of a tiler on the device)
  const std::string sfx = std::to_string(level_index) + "_" + std::to_string(which);
  auto one = [&](const std::string& name, size_t elem, void** out) -> int {
    const size_t have = c->bufs[name].cap / elem;
    const size_t want = have >= count && c->bufs[name].ptr ? have : count + count / 2 + 1024;
    return grow_preserving(c, name.c_str(), want * elem, 0, out, true);
  };
  void *pk = nullptr, *po = nullptr, *pc = nullptr;
  SWZ_TRY(one("tiler_store_tab_key_" + sfx, 8, &pk));
  SWZ_TRY(one("tiler_store_tab_off_" + sfx, 8, &po));
  SWZ_TRY(one("tiler_store_tab_cnt_" + sfx, 4, &pc));
  s.nkey[which] = static_cast<uint64_t*>(pk);
  s.noff[which] = static_cast<uint64_t*>(po);
  s.ncnt[which] = static_cast<uint32_t*>(pc);
  return SWZ_OK;
}
// the node table of a level in linear form: the runs of equal node prefix
static int store_table(swz_ctx* c, StoreLevel& s, int level_index) {
  if (s.table_valid) return SWZ_OK;
  if (!s.linear) return c->fail(SWZ_ERR_INTERNAL, "node store: neither linear nor indexed");
  s.nn = 0;
  if (s.cnt) {
    const int level = level_index - 1;
    const uint32_t nsh = level < 0 ? 63u : level_shift(level);
    uint32_t *counters = nullptr, *hp = nullptr;
    uint64_t* hk = nullptr;
    SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
    SWZ_TRY(c->get("tl_head_pos", (size_t)s.cnt, &hp));
    SWZ_TRY(c->get("tl_head_key", (size_t)s.cnt, &hk));
    SWZ_TRY(fused_scan(c, HeadF{s.key[s.cur], nsh}, HeadG{s.key[s.cur], nsh, hp, hk}, s.cnt, counters + 3, "tl"));
    uint32_t heads = 0;
    SWZ_TRY(read_u32(c, counters + 3, &heads));
    SWZ_TRY(table_reserve(c, s, level_index, s.ncur, heads));
    hipLaunchKernelGGL(tl_table_build_kernel, dim3(div_up(heads, 256)), dim3(256), 0, c->stream, hk, hp, heads, s.cnt,
                       s.nkey[s.ncur], s.noff[s.ncur], s.ncnt[s.ncur]);
    SWZ_LAUNCH_CHECK(c);
    s.nn = heads;
  }
  s.table_valid = true;
  return SWZ_OK;
}
// Gathers the files a table lists (ntab entries, `live` entries in all, in table order) into the other side, which gets
// room for `room` entries, and makes it the current one; off[] (device) is rewritten to the new places.
static int store_compact(swz_ctx* c, StoreLevel& s, int level_index, uint64_t* off, const uint32_t* cnt, uint32_t ntab,
                         uint32_t live, size_t room) {
  const int dst = s.cur ^ 1;
  SWZ_TRY(store_reserve(c, s, level_index, dst, std::max<size_t>(room, live)));
  if (ntab && live) {
    uint32_t *poff = nullptr, *counters = nullptr;
    uint64_t* psrc = nullptr;
    SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
    SWZ_TRY(c->get("tl_poff", (size_t)ntab, &poff));
    SWZ_TRY(c->get("tl_psrc", (size_t)ntab, &psrc));
    SWZ_TRY(fused_scan(c, SegCntF{cnt}, SegMoveG{off, poff, psrc}, ntab, counters + 2, "tl"));
    hipLaunchKernelGGL(tl_gather_files_kernel, dim3(div_up(live, 256)), dim3(256), 0, c->stream, poff, psrc, ntab, live,
                       s.key[s.cur], s.gid[s.cur], s.key[dst], s.gid[dst]);
    SWZ_LAUNCH_CHECK(c);
  }
  s.cur = dst;
  s.end = live;
  return SWZ_OK;
}
// log form -> linear form (the table stays valid)
static int store_linearize(swz_ctx* c, StoreLevel& s, int level_index) {
  if (s.linear) return SWZ_OK;
  if (!s.table_valid) return c->fail(SWZ_ERR_INTERNAL, "node store: log without a table");
  SWZ_TRY(store_compact(c, s, level_index, s.noff[s.ncur], s.ncnt[s.ncur], s.nn, s.cnt, s.cnt));
  s.linear = true;
  return SWZ_OK;
}

// Grows a named workspace buffer keeping its first `keep` bytes.  A POOL (spill == true) that finds no device memory
// -- hipMalloc out of memory, or the workspace above SWZ_TILER_DEVICE_BUDGET_MB -- moves to page-locked host memory mapped
// into the device's address space and stays there: the pools hold 24 bytes + the attribute rows of EVERY point of the
// data set, the bulk of a tiler's memory, and the tiling touches them lightly -- a batch's own points once, in order
// (clamp + index), the cached points a batch pulls in by id (re-key), MIN_DISTANCE's rare exact compares; the
// attribute columns not at all until the files are exported.  The kernels read and write them in place over the host
// link.  SWZ_TILER_SPILL: "auto" (default), "host" (pools on the host from the start), "off".
static int grow_preserving(swz_ctx* c, const char* name, size_t bytes, size_t keep, void** out, bool spill) {
  swz::DevBuf& b = c->bufs[name];
  if (b.cap < bytes) {
    void* np = nullptr;
    const size_t want = (bytes + 255) & ~size_t(255);
    int policy = 1;
    if (const char* e = c->opt("SWZ_TILER_SPILL")) policy = strcmp(e, "off") == 0 ? 0 : (strcmp(e, "host") == 0 ? 2 : 1);
    if (!spill) policy = 0;
    hipError_t e = hipErrorOutOfMemory;
    if (policy != 2 && !b.host) {  // (a pool that has moved to the host does not come back)
      bool over_budget = false;
      if (const char* bm = c->opt("SWZ_TILER_DEVICE_BUDGET_MB"))
        over_budget = policy != 0 && (c->held_bytes() + want) > (uint64_t)atoll(bm) * 1048576ull;
      if (!over_budget) e = hipMalloc(&np, want);
      if (const char* fa = c->opt("SWZ_FAIL_ALLOC"))
        if (e == hipSuccess && strcmp(fa, name) == 0) {
          (void)hipFree(np);
          np = nullptr;
          e = hipErrorOutOfMemory;
        }
    }
    bool host = false;
    if (e == hipErrorOutOfMemory && policy != 0) {
      (void)hipGetLastError();
      e = hipHostMalloc(&np, want, hipHostMallocMapped | hipHostMallocPortable);
      host = e == hipSuccess;
    }
    if (e != hipSuccess) return c->fail(SWZ_ERR_HIP, std::string("hipMalloc(") + name + "): " + hipGetErrorString(e));
    if (b.ptr && keep) SWZ_HIP(c, hipMemcpy(np, b.ptr, keep, hipMemcpyDefault));
    if (const char* e = c->opt("SWZ_POISON")) {  // (like swz_ctx::get: what nobody has written yet must not read as zeros)
      const char* only = c->opt("SWZ_POISON_ONLY");
      if (!only || strstr(name, only)) {
        if (host) {
          memset((char*)np + keep, atoi(e), want - keep);
        } else {  // (complete before anybody's stream writes into the buffer: hipMemset may return early)
          SWZ_HIP(c, hipMemset((char*)np + keep, atoi(e), want - keep));
          SWZ_HIP(c, hipDeviceSynchronize());
        }
      }
    }
    c->free_buf(b);
    b.ptr = np;
    b.cap = want;
    b.host = host;
  }
  *out = b.ptr;
  return SWZ_OK;
}

// makes room for `points` points in the pools (positions and the attribute columns in use); keeps the content
static int pool_reserve(swz_tiler* t, size_t points) {
  swz_ctx* c = t->c;
  if (points <= t->pool_cap && t->pool_xyz) return SWZ_OK;
  // nothing may still be writing into or reading from the old pools
  if (t->copy_stream) SWZ_HIP(c, hipStreamSynchronize(t->copy_stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  size_t have = c->bufs["tiler_pool_xyz"].cap / 24;  // what an earlier tiler of this context left behind
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (t->attr_mask & (1u << a)) have = std::min(have, c->bufs["tiler_pool_attr" + std::to_string(a)].cap / TILER_ATTR_BYTES[a]);
  const size_t want = points <= have ? have : std::max(points, t->pool_cap + t->pool_cap / 2);
  const size_t used = t->staged_total;
  void* px = nullptr;
  SWZ_TRY(grow_preserving(c, "tiler_pool_xyz", want * 24, used * 24, &px, true));
  t->pool_xyz = static_cast<double*>(px);
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a) {
    if (!(t->attr_mask & (1u << a))) continue;
    const std::string name = "tiler_pool_attr" + std::to_string(a);
    SWZ_TRY(grow_preserving(c, name.c_str(), want * TILER_ATTR_BYTES[a], used * TILER_ATTR_BYTES[a], &t->pool_attr[a], true));
  }
  t->pool_cap = want;
  return SWZ_OK;
}

static int merge_pairs(swz_ctx* c, const uint64_t* k1, const uint32_t* v1, uint32_t n1, const uint64_t* k2,
                       const uint32_t* v2, uint32_t n2, uint32_t sh, uint32_t base2, uint64_t* ok, uint32_t* ov) {
  if (n1) {
    hipLaunchKernelGGL(tl_merge_rank_kernel<false>, dim3(div_up(n1, TL_MERGE_TILE)), dim3(256), 0, c->stream, k1, v1, n1, k2, n2, sh, 0u, ok, ov);
    SWZ_LAUNCH_CHECK(c);
  }
  if (n2) {
    hipLaunchKernelGGL(tl_merge_rank_kernel<true>, dim3(div_up(n2, TL_MERGE_TILE)), dim3(256), 0, c->stream, k2, v2, n2, k1, n1, sh, base2, ok, ov);
    SWZ_LAUNCH_CHECK(c);
  }
  return SWZ_OK;
}

static int read_u32(swz_ctx* c, const uint32_t* d, uint32_t* h) {
  SWZ_HIP(c, hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

// (key, gid) ascending by key when the re-keyed order is not (the reference only sorts for a lossy persistence,
// :103-106 / :1690-1692; counted in rekey_inversions)
static int sort_pairs_by_key(swz_ctx* c, uint64_t* key, uint32_t* gid, uint32_t n) {
  uint64_t* kb = nullptr;
  uint32_t* vb = nullptr;
  SWZ_TRY(c->get("tl_sort_k", (size_t)n, &kb));
  SWZ_TRY(c->get("tl_sort_v", (size_t)n, &vb));
  if (radix_result_in_second()) {
    SWZ_HIP(c, hipMemcpyAsync(kb, key, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_HIP(c, hipMemcpyAsync(vb, gid, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
    SWZ_TRY(radix_sort_pairs(c, kb, vb, key, gid, n, false));
  } else {
    SWZ_TRY(radix_sort_pairs(c, key, gid, kb, vb, n, false));
  }
  return SWZ_OK;
}

__global__ __launch_bounds__(256) void tl_rows_kernel(const uint32_t* __restrict__ gid, uint32_t n, const double* __restrict__ pool,
                                                      double* __restrict__ out) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t g = gid[i];
  out[3 * (size_t)i] = pool[3 * g];
  out[3 * (size_t)i + 1] = pool[3 * g + 1];
  out[3 * (size_t)i + 2] = pool[3 * g + 2];
}
__global__ __launch_bounds__(256) void tl_iota_base_kernel(uint32_t* __restrict__ out, uint32_t n, uint32_t base) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = base + i;
}
__global__ __launch_bounds__(256) void tl_count_untaken_kernel(const uint8_t* __restrict__ taken, uint32_t n,
                                                               uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint64_t b = __ballot(i < n && !taken[i]);
  if (lane_id() == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
}

// sr != nullptr: the ROOT level of a sharded batch (the node spans all shards: its take-all / sample decision is
// the global one, its whole local file takes part even without new local points, and for MIN_DISTANCE what the root
// took on lower shards in this batch sorts first as ghosts -- swz_tiler_shard_begin_device)
// The working pool keeps positions (SoA, by working index) only from the first level on that reads them: RANDOM_GRID never
// does, MIN_DISTANCE / GRID_CENTER / JITTERED decide on key coordinates and look up the position pool through the point ids
// (level_decides_on_keys) -- a batch of a usual data set never fills them.  Levels that do read them (bounds that are no
// cube, spacings of fewer than 64 key cells, re-rooted nodes, the ghosts of a sharded root) fill everything that is in the
// working pool by then, from the ids; entries pulled in later are filled as they come.
static int work_need_positions(swz_tiler* t, BatchWork& w) {
  if (w.have_pos) return SWZ_OK;
  swz_ctx* c = t->c;
  // (24 bytes per working-pool entry -- batch + everything stored -- that most data sets never touch: allocated here, not
  // with the batch.  At 2.4 B stored points the three arrays were 78 of the 286 GB that ran the device out of memory.)
  SWZ_TRY(c->get("tl_wx", (size_t)w.wcap, &w.wx));
  SWZ_TRY(c->get("tl_wy", (size_t)w.wcap, &w.wy));
  SWZ_TRY(c->get("tl_wz", (size_t)w.wcap, &w.wz));
  if (w.wused) {
    hipLaunchKernelGGL(tl_fill_kernel, dim3(div_up(w.wused, 256)), dim3(256), 0, c->stream, w.wgid, w.wused, t->pool_xyz, w.wx, w.wy, w.wz,
                       (uint32_t*)nullptr);
    SWZ_LAUNCH_CHECK(c);
  }
  w.have_pos = true;
  return SWZ_OK;
}

static int tiler_level(swz_tiler* t, BatchWork& w, const LevelPlan& plan_in, ActiveSet& as, LevelResult* res,
                       uint32_t* merged_out, const ShardRoot* sr = nullptr) {
  swz_ctx* c = t->c;
  LevelPlan plan = plan_in;
  StoreLevel& st = t->lv[plan.level + 1];
  const uint32_t nsh = plan.node_shift;
  uint32_t* counters = nullptr;
  SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
  const uint32_t ng = sr ? sr->ghosts : 0u;
  if (sr) {
    if (sr->sample) plan.force_sample = true; else plan.max_points = ~0ull;
  }

  // ---- pull the files of the nodes this level's active set reaches
  uint64_t* ckey = nullptr;
  uint32_t* cgid = nullptr;
  uint32_t nc = 0;
  uint8_t* touched = nullptr;  // per node of the level's table: the batch rewrites its file
  bool all_touched = false;
  const int lvi = plan.level + 1;
  if (st.cnt && sr) {  // the root is reached by the batch as a whole: all of its local file
    SWZ_TRY(store_linearize(c, st, lvi));
    SWZ_TRY(store_table(c, st, lvi));
    SWZ_TRY(c->get("tl_ckey", (size_t)st.cnt, &ckey));
    SWZ_TRY(c->get("tl_cgid", (size_t)st.cnt, &cgid));
    SWZ_HIP(c, hipMemcpyAsync(ckey, st.key[st.cur], (size_t)st.cnt * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_HIP(c, hipMemcpyAsync(cgid, st.gid[st.cur], (size_t)st.cnt * 4, hipMemcpyDeviceToDevice, c->stream));
    nc = st.cnt;
    all_touched = true;
  } else if (st.cnt) {
    // (profile class "tiler_pull": the nodes of the active set, their files looked up in the level's node table and
    // copied out -- what the batch reaches, not what the level holds)
    SWZ_TRY(store_table(c, st, lvi));
    uint32_t* hp = nullptr;
    uint64_t* hk = nullptr;
    SWZ_TRY(c->get("tl_head_pos", (size_t)as.m, &hp));
    SWZ_TRY(c->get("tl_head_key", (size_t)as.m, &hk));
    SWZ_TRY(c->get("tl_ntouch", (size_t)st.nn, &touched));
    uint32_t heads = 0;
    {
      ProfScope ps(c, "tiler_pull", (uint64_t)as.m * 8ull, 2);
      SWZ_TRY(fused_scan(c, HeadF{as.akey, nsh}, HeadG{as.akey, nsh, hp, hk}, as.m, counters + 3, "tl"));
      SWZ_HIP(c, hipMemsetAsync(touched, 0, (size_t)st.nn, c->stream));
    }
    SWZ_TRY(read_u32(c, counters + 3, &heads));
    uint32_t* poff = nullptr;
    uint64_t* psrc = nullptr;
    SWZ_TRY(c->get("tl_poff", (size_t)heads, &poff));
    SWZ_TRY(c->get("tl_psrc", (size_t)heads, &psrc));
    const PullCntF pf{hk, st.nkey[st.ncur], st.ncnt[st.ncur], st.nn};
    SWZ_TRY(fused_scan(c, pf, PullSegG{pf, st.noff[st.ncur], poff, psrc, touched}, heads, counters, "tl"));
    SWZ_TRY(read_u32(c, counters, &nc));
    if (nc == st.cnt && st.linear) {
      // the batch reaches every node of the level and the side holds the files in node order (a batch cut out of the whole
      // cloud): what would be copied out is the side itself.  Everything on it is rewritten by this batch -- behind its end
      // or on the other side --, so the merge may read it in place (and a re-sort after an inversion may reorder it).
      ckey = st.key[st.cur];
      cgid = st.gid[st.cur];
    } else if (nc) {
      ProfScope ps(c, "tiler_pull", (uint64_t)nc * 24ull, 1);
      SWZ_TRY(c->get("tl_ckey", (size_t)nc, &ckey));
      SWZ_TRY(c->get("tl_cgid", (size_t)nc, &cgid));
      hipLaunchKernelGGL(tl_gather_files_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, poff, psrc, heads, nc,
                         st.key[st.cur], st.gid[st.cur], ckey, cgid);
      SWZ_LAUNCH_CHECK(c);
    }
  } else {
    SWZ_TRY(store_table(c, st, lvi));  // (an empty level: an empty table)
  }
  const uint32_t pull_lo = w.wused;  // working indices of the pulled entries, below
  ActiveSet ms = as;
  if (ng) SWZ_TRY(work_need_positions(t, w));  // (the ghosts bring their positions: the working pool holds them from here on)
  if (nc || ng) {
    if (nc && !st.rekeyed) {
      // ("tiler_rekey": the pulled points' keys against their NODE's bounds -- a random 24-byte read per point from the
      // pool; only for files the level loop did not write itself, see TakeStoreG)
      ProfScope ps(c, "tiler_rekey", (uint64_t)nc * 44ull, 1);
      hipLaunchKernelGGL(tl_rekey_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, ckey, cgid, nc, t->pool_xyz,
                         root_box(t), plan.level);
      SWZ_LAUNCH_CHECK(c);
    }
    // (needed for files the level loop wrote as well: TakeStoreG stores the key against the NODE's bounds, which may order two
    // points the other way round than the key they were sorted by -- tests/test_multibatch.py constructs such a pair)
    if (nc && !plan.terminal) {
      SWZ_HIP(c, hipMemsetAsync(counters + 1, 0, 4, c->stream));
      hipLaunchKernelGGL(tl_inversion_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, ckey, nc, nsh, counters + 1);
      SWZ_LAUNCH_CHECK(c);
      uint32_t inv = 0;
      SWZ_TRY(read_u32(c, counters + 1, &inv));
      if (inv) {
        t->rekey_inversions += inv;
        SWZ_TRY(sort_pairs_by_key(c, ckey, cgid, nc));
      }
    }
    if (w.wused + nc + ng > w.wcap) return c->fail(SWZ_ERR_INTERNAL, "working pool overflow");
    if (nc) {
      hipLaunchKernelGGL(tl_fill_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, cgid, nc, t->pool_xyz,
                         w.have_pos ? w.wx + w.wused : nullptr, w.have_pos ? w.wy + w.wused : nullptr,
                         w.have_pos ? w.wz + w.wused : nullptr, w.wgid + w.wused);
      SWZ_LAUNCH_CHECK(c);
    }
    uint64_t* mkey = nullptr;
    uint32_t* midx = nullptr;
    SWZ_TRY(c->get("tl_mkey", (size_t)as.m + nc + ng, &mkey));
    SWZ_TRY(c->get("tl_midx", (size_t)as.m + nc + ng, &midx));
    // tile_node :421-442: terminal nodes append (new ++ cached), the others std::merge by key; ghosts lie in lower
    // octants, so their keys are smaller than every local key: sorted ghosts ++ merged locals is the merged whole
    {
      ProfScope ps(c, "tiler_merge", ((uint64_t)as.m + nc) * 24ull, 2);
      SWZ_TRY(merge_pairs(c, as.akey, as.aidx, as.m, ckey, nullptr, nc, plan.terminal ? nsh : 0u, w.wused, mkey + ng, midx + ng));
    }
    w.wused += nc;
    if (ng) {
      uint64_t *gk = nullptr, *gkb = nullptr;
      uint32_t *gp = nullptr, *gpb = nullptr;
      SWZ_TRY(c->get("tl_gkey", (size_t)ng, &gk));
      SWZ_TRY(c->get("tl_gkey_b", (size_t)ng, &gkb));
      SWZ_TRY(c->get("tl_gperm", (size_t)ng, &gp));
      SWZ_TRY(c->get("tl_gperm_b", (size_t)ng, &gpb));
      double* gx = const_cast<double*>(sr->ghost_xyz);  // inside the bounds already: the clamp of the encode is a no-op
      uint64_t* sorted_k = gk;
      uint32_t* sorted_p = gp;
      if (radix_result_in_second()) {
        SWZ_TRY(encode_device(c, gx, ng, t->bmin, t->bmax, gkb));
        SWZ_TRY(radix_sort_pairs(c, gkb, gpb, gk, gp, ng, true));
      } else {
        SWZ_TRY(encode_device(c, gx, ng, t->bmin, t->bmax, gk));
        SWZ_TRY(radix_sort_pairs(c, gk, gp, gkb, gpb, ng, true));
      }
      SWZ_TRY(gather_positions(c, sr->ghost_xyz, sorted_p, ng, w.wx + w.wused, w.wy + w.wused, w.wz + w.wused));
      SWZ_HIP(c, hipMemsetAsync(w.wgid + w.wused, 0xFF, (size_t)ng * 4, c->stream));
      SWZ_HIP(c, hipMemcpyAsync(mkey, sorted_k, (size_t)ng * 8, hipMemcpyDeviceToDevice, c->stream));
      hipLaunchKernelGGL(tl_iota_base_kernel, dim3(div_up(ng, 256)), dim3(256), 0, c->stream, midx, ng, w.wused);
      SWZ_LAUNCH_CHECK(c);
      w.wused += ng;
    }
    ms.akey = mkey;
    ms.aidx = midx;
    ms.m = as.m + nc + ng;
    ms.ckey = sr ? nullptr : ckey;  // a sharded root decides from the global counts
    ms.nc = sr ? 0u : nc;
    if (!sr && !ng && nc) {
      ms.old_lo = pull_lo;
      ms.old_hi = pull_lo + nc;
      ms.new_key = as.akey;
      ms.new_m = as.m;
      if (plan.sampler == SWZ_MIN_DISTANCE && !plan.terminal && plan.level + 2 < 22) {
        StoreLevel& below = t->lv[plan.level + 2];
        if (below.cnt) {  // (the table this batch's next level asks for anyway)
          SWZ_TRY(store_table(c, below, plan.level + 2));
          ms.child_nkey = below.nkey[below.ncur];
          ms.child_nn = below.nn;
        }
      }
    }
  }
  if (ms.m == 0) {  // a shard without new points and without a root file
    res->remaining = 0;
    *merged_out = 0;
    return SWZ_OK;
  }
  *merged_out = ms.m;

  // ---- sample / take all, compact the survivors
  SWZ_TRY(c->get(w.which ? "tl_surv_key_1" : "tl_surv_key_0", (size_t)ms.m, &w.surv_key[w.which]));
  SWZ_TRY(c->get(w.which ? "tl_surv_idx_1" : "tl_surv_idx_0", (size_t)ms.m, &w.surv_idx[w.which]));
  LevelBuffers lb;
  SWZ_TRY(alloc_level_buffers(c, ms.m, &lb));
  // (exact positions for MIN_DISTANCE on key coordinates, swz_mdkeys.hip: working index -> point id -> position pool;
  // ghosts of a sharded root lie outside the pool)
  SortedPoints sp{nullptr, nullptr, nullptr, ng ? nullptr : t->pool_xyz, w.wgid};
  if (!level_decides_on_keys(c, plan, sp)) SWZ_TRY(work_need_positions(t, w));
  if (w.have_pos) {
    sp.X = w.wx;
    sp.Y = w.wy;
    sp.Z = w.wz;
  }
  SWZ_TRY(level_step(c, plan, ms, sp, lb, w.wlevel, w.surv_key[w.which], w.surv_idx[w.which], res));

  // ---- the nodes' new files: appended behind what the side holds, the node table points at them
  const uint32_t nt = ms.m - res->remaining - ng;  // (the ghosts lead the merged range and are all taken again)
  const uint32_t rest = st.cnt - nc;               // entries of the files the batch did not reach
  uint64_t *fkey = nullptr, *foff = nullptr;
  uint32_t* fcnt = nullptr;
  uint32_t nf = 0;
  bool nf_known = true;
  if (st.nn && !all_touched && rest) {
    ProfScope ps(c, "tiler_store", (uint64_t)st.nn * 21ull, 2);
    SWZ_TRY(c->get("tl_ftab_key", (size_t)st.nn, &fkey));
    SWZ_TRY(c->get("tl_ftab_off", (size_t)st.nn, &foff));
    SWZ_TRY(c->get("tl_ftab_cnt", (size_t)st.nn, &fcnt));
    SWZ_TRY(fused_scan(c, UntouchedF{touched}, TableFilterG{st.nkey[st.ncur], st.noff[st.ncur], st.ncnt[st.ncur], fkey, foff, fcnt},
                       st.nn, counters + 1, "tl"));
    nf_known = false;
  }
  if (!st.key[st.cur] || (uint64_t)st.end + nt > st.cap[st.cur]) {
    // the side is full: the files the batch left alone move to the other side, the old versions of the rest stay behind
    if (!nf_known) {
      SWZ_TRY(read_u32(c, counters + 1, &nf));
      nf_known = true;
    }
    ProfScope ps(c, "tiler_store", (uint64_t)rest * 24ull, 2);
    const size_t room = (size_t)rest + nt;
    SWZ_TRY(store_compact(c, st, lvi, foff, fcnt, nf, rest, room + room / 4));
  }
  const uint32_t at = st.end;
  uint32_t* texcl = nullptr;
  SWZ_TRY(c->get("tl_texcl", (size_t)ms.m, &texcl));
  {
    ProfScope ps(c, "tiler_store", (uint64_t)ms.m * 14ull + (uint64_t)nt * 12ull, 2);
    SWZ_TRY(fused_scan(c, TakenF{lb.taken},
                       TakeStoreG{ms.akey, ms.aidx, w.wgid, pull_lo, pull_lo + nc, ng, t->pool_xyz, root_box(t), plan.level,
                                  st.key[st.cur] + at, st.gid[st.cur] + at, nsh, texcl},
                       ms.m, counters + 2, "tl"));
  }
  if (ng) {  // the ghosts are not part of the local file
    SWZ_HIP(c, hipMemsetAsync(counters, 0, 4, c->stream));
    hipLaunchKernelGGL(tl_count_untaken_kernel, dim3(div_up(ng, 256)), dim3(256), 0, c->stream, lb.taken, ng, counters);
    SWZ_LAUNCH_CHECK(c);
    uint32_t lost = 0;
    SWZ_TRY(read_u32(c, counters, &lost));
    if (lost) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler: " + std::to_string(lost) + " ghost points were not taken again "
                                              "(they must be what the root took on LOWER shards in this batch)");
  }
  {
    ProfScope ps(c, "tiler_store", (uint64_t)nt * 8ull + (uint64_t)st.nn * 20ull, 2);
    // the nodes of the merged range are the nodes of the new files (every node takes at least one point; only a sharded
    // root whose local points all fell to the ghosts writes nothing): their number is the level step's, no read-back
    uint32_t* hp = nullptr;
    uint64_t* hk = nullptr;
    const uint32_t heads = nt ? res->num_nodes : 0u;
    if (heads) {
      SWZ_TRY(c->get("tl_head_pos", (size_t)heads, &hp));
      SWZ_TRY(c->get("tl_head_key", (size_t)heads, &hk));
      hipLaunchKernelGGL(tl_new_heads_kernel, dim3(div_up(heads, 256)), dim3(256), 0, c->stream, ms.akey, lb.nstart, texcl, heads, nsh, ng, hk, hp);
      SWZ_LAUNCH_CHECK(c);
    }
    if (!nf_known) SWZ_TRY(read_u32(c, counters + 1, &nf));
    const int nd = st.ncur ^ 1;
    SWZ_TRY(table_reserve(c, st, lvi, nd, (size_t)nf + heads));
    if (nf + heads) {
      hipLaunchKernelGGL(tl_table_merge_kernel, dim3(div_up(nf + heads, 256)), dim3(256), 0, c->stream, fkey, foff, fcnt, nf, hk, hp,
                         heads, nt, (uint64_t)at, st.nkey[nd], st.noff[nd], st.ncnt[nd]);
      SWZ_LAUNCH_CHECK(c);
    }
    st.ncur = nd;
    st.nn = nf + heads;
  }
  st.end = at + nt;
  st.cnt = rest + nt;
  st.linear = at == 0;  // (nothing in front of the new files: they are the level)
  if (rest == 0) st.rekeyed = true;

  as = ActiveSet{w.surv_key[w.which], w.surv_idx[w.which], res->remaining};
  as.parent_prefix = res->node_prefix;
  as.parents = res->node_prefix ? res->num_nodes : 0u;
  w.which ^= 1;
  return SWZ_OK;
}

// ---------------------------------------------------------------------------------------------- re-rooting
// A node whose sampler needs more than 21 key levels below the root becomes the root of a new 21-level index
// (tile_node, TilingAlgorithms.cpp:444-483): all its points (new ++ cached, unsorted) are re-indexed against the
// NODE's bounds, sorted, and sampled as that root's level -1 with the node's own max_spacing.  Its children inherit
// the new root, so every node below re-roots again until the levels run out (level >= min(20, max_depth): terminal).
// Such nodes are rare (> max_points_per_node points inside a cell 2^-15 of the root's extent at d = 250) and handled
// one node at a time by the host, every step on the device.  Literal, including that the children
// are split at the ABSOLUTE level of the re-rooted keys (:124-125 via :479-482) and that a point outside the box it
// is re-indexed against goes through static_cast<uint64_t> of a negative double as x86-64 evaluates it.
struct RrNode {
  int level;
  uint64_t key;
  double bmin[3], bmax[3];
  float max_spacing;
};
struct RrTotals {
  uint64_t nodes = 0, visited = 0;
  int max_level = -1;
};

static int rr_store_node(swz_tiler* t, int level, const uint64_t* rkey, const uint32_t* rgid, uint32_t nr,
                         const uint64_t* tkey, const uint32_t* tgid, uint32_t nt) {
  swz_ctx* c = t->c;
  StoreLevel& st = t->lv[level + 1];
  const int dst = st.cur ^ 1;
  SWZ_TRY(store_reserve(c, st, level + 1, dst, (size_t)nr + nt));
  SWZ_TRY(merge_pairs(c, rkey, rgid, nr, tkey, tgid, nt, level < 0 ? 63u : level_shift(level), 0u, st.key[dst], st.gid[dst]));
  store_written_linear(st, dst, nr + nt, false);
  return SWZ_OK;
}

static int rr_node(swz_tiler* t, BatchWork& w, const RrNode& node, double root_ext_x, float root_max_spacing,
                   const uint32_t* d_idx, uint32_t cnt, int depth, RrTotals& tot) {
  swz_ctx* c = t->c;
  if (depth > 24) return c->fail(SWZ_ERR_INTERNAL, "re-rooting recursed too deep");
  SWZ_TRY(work_need_positions(t, w));  // re-indexing against the node's box reads the positions
  const uint32_t nsh = level_shift(node.level);
  uint32_t* counters = nullptr;
  SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
  // ---- the node's file (read_pnts_from_disk; the re-keying is irrelevant: everything is re-indexed or appended)
  StoreLevel& st = t->lv[node.level + 1];
  SWZ_TRY(store_linearize(c, st, node.level + 1));
  uint64_t *ckey = nullptr, *rkey = nullptr;
  uint32_t *cgid = nullptr, *rgid = nullptr;
  uint32_t nc = 0, nr = 0;
  if (st.cnt) {
    uint8_t* touch = nullptr;
    uint64_t* d_nodekey = nullptr;
    SWZ_TRY(c->get("tl_touch", (size_t)st.cnt, &touch));
    SWZ_TRY(c->get("tl_ckey", (size_t)st.cnt, &ckey));
    SWZ_TRY(c->get("tl_cgid", (size_t)st.cnt, &cgid));
    SWZ_TRY(c->get("tl_rkey", (size_t)st.cnt, &rkey));
    SWZ_TRY(c->get("tl_rgid", (size_t)st.cnt, &rgid));
    SWZ_TRY(c->get("rr_nodekey", (size_t)1, &d_nodekey));
    SWZ_HIP(c, hipMemcpyAsync(d_nodekey, &node.key, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(tl_touch_kernel, dim3(div_up(st.cnt, 256)), dim3(256), 0, c->stream, st.key[st.cur], st.cnt,
                       d_nodekey, 1u, nsh, touch);
    SWZ_LAUNCH_CHECK(c);
    SWZ_TRY(fused_scan(c, TouchF{touch}, SplitG{st.key[st.cur], st.gid[st.cur], ckey, cgid, rkey, rgid}, st.cnt, counters, "tl"));
    SWZ_TRY(read_u32(c, counters, &nc));
    nr = st.cnt - nc;
  }
  const uint32_t m = cnt + nc;
  tot.nodes += 1;
  tot.visited += m;
  tot.max_level = std::max(tot.max_level, node.level);
  // all = node_data ++ cached (merge_node_data_unsorted), as working-pool indices
  const std::string sfx = std::to_string(depth);
  uint32_t* all = nullptr;
  SWZ_TRY(c->get(("rr_all_" + sfx).c_str(), (size_t)m, &all));
  SWZ_HIP(c, hipMemcpyAsync(all, d_idx, (size_t)cnt * 4, hipMemcpyDeviceToDevice, c->stream));
  if (nc) {
    if (w.wused + nc > w.wcap) return c->fail(SWZ_ERR_INTERNAL, "working pool overflow");
    hipLaunchKernelGGL(tl_fill_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, cgid, nc, t->pool_xyz,
                       w.wx + w.wused, w.wy + w.wused, w.wz + w.wused, w.wgid + w.wused);
    SWZ_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(rr_iota_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, all + cnt, nc);  // 0..nc-1
    SWZ_LAUNCH_CHECK(c);
    // shift to the pool positions just filled
    hipLaunchKernelGGL(tl_wgid_kernel, dim3(div_up(nc, 256)), dim3(256), 0, c->stream, all + cnt, nc, w.wused, all + cnt);
    SWZ_LAUNCH_CHECK(c);
    w.wused += nc;
  }
  uint64_t* tkey = nullptr;
  uint32_t* tgid = nullptr;
  SWZ_TRY(c->get("tl_tkey", (size_t)m, &tkey));
  SWZ_TRY(c->get("tl_tgid", (size_t)m, &tgid));

  const int req = required_depth_host(t->p.sampler, node.level, root_ext_x, root_max_spacing);
  const int max_level = (int)std::min<uint32_t>(MAX_LEVELS - 1, t->p.max_depth);
  if (req <= node.level) return c->fail(SWZ_ERR_INTERNAL, "re-rooted subtree reached a node that needs no deeper index");
  if (node.level >= max_level) {  // tile_terminal_node (:436-442): everything, in this order
    LevelBuffers lb;
    SWZ_TRY(alloc_level_buffers(c, m, &lb));
    SWZ_TRY(fused_scan(c, AllF{}, TakeNodeG{all, w.wgid, node.key, tkey, tgid}, m, counters + 2, "tl"));
    return rr_store_node(t, node.level, rkey, rgid, nr, tkey, tgid, m);
  }
  if (req < (int)MAX_LEVELS) return c->fail(SWZ_ERR_INTERNAL, "re-rooted subtree reached a node that needs no re-rooting");

  // ---- re-index against the node's box, sort, sample as the new root's level -1
  uint64_t *keys = nullptr, *okey = nullptr;
  uint32_t* oidx = nullptr;
  SWZ_TRY(c->get(("rr_keys_" + sfx).c_str(), (size_t)m, &keys));
  SWZ_TRY(c->get(("rr_okey_" + sfx).c_str(), (size_t)m, &okey));
  SWZ_TRY(c->get(("rr_oidx_" + sfx).c_str(), (size_t)m, &oidx));
  const Box nb{node.bmin[0], node.bmin[1], node.bmin[2], node.bmax[0], node.bmax[1], node.bmax[2]};
  hipLaunchKernelGGL(rr_encode_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, all, m, w.wx, w.wy, w.wz, nb, keys);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(sort_pairs_by_key(c, keys, all, m));  // stable: ties keep the order of `all`
  LevelPlan plan = make_plan(-1, t->p.sampler, t->p.max_points_per_node, node.max_spacing, t->p.max_depth, node.bmin,
                             node.bmax, nc > 0, false);
  LevelBuffers lb;
  SWZ_TRY(alloc_level_buffers(c, m, &lb));
  LevelResult r;
  SWZ_TRY(level_step(c, plan, ActiveSet{keys, all, m}, SortedPoints{w.wx, w.wy, w.wz}, lb, w.wlevel, okey, oidx, &r));
  const uint32_t nt = m - r.remaining;
  SWZ_TRY(fused_scan(c, TakenF{lb.taken}, TakeNodeG{all, w.wgid, node.key, tkey, tgid}, m, counters + 2, "tl"));
  SWZ_TRY(rr_store_node(t, node.level, rkey, rgid, nr, tkey, tgid, nt));
  if (r.remaining == 0) return SWZ_OK;

  // ---- children (split_range_into_child_nodes :116-162)
  const int child_level = node.level + 1;
  uint32_t* d_bounds = nullptr;
  SWZ_TRY(c->get(("rr_bounds_" + sfx).c_str(), (size_t)16, &d_bounds));
  hipLaunchKernelGGL(rr_split_init_kernel, dim3(1), dim3(64), 0, c->stream, d_bounds, r.remaining);
  for (uint32_t o = 0; o < 8; ++o)
    hipLaunchKernelGGL(rr_split_kernel, dim3(div_up(r.remaining, 256)), dim3(256), 0, c->stream, okey, r.remaining,
                       level_shift(child_level), o, d_bounds);
  SWZ_LAUNCH_CHECK(c);
  uint32_t b[9];
  SWZ_HIP(c, hipMemcpyAsync(b, d_bounds, 36, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  const double child_root_ext = node.bmax[0] - node.bmin[0];
  for (uint32_t o = 0; o < 8; ++o) {
    if (b[o + 1] <= b[o]) continue;
    RrNode child;
    child.level = child_level;
    child.key = node.key | ((uint64_t)o << level_shift(child_level));
    for (int ax = 0; ax < 3; ++ax) {  // get_octant_bounds
      const double e = node.bmax[ax] - node.bmin[ax];
      const uint32_t bit = ax == 0 ? (o >> 2) & 1u : (ax == 1 ? (o >> 1) & 1u : o & 1u);
      child.bmin[ax] = bit ? node.bmin[ax] + e / 2 : node.bmin[ax];
      child.bmax[ax] = child.bmin[ax] + e / 2;
    }
    child.max_spacing = node.max_spacing / 2;
    SWZ_TRY(rr_node(t, w, child, child_root_ext, node.max_spacing, oidx + b[o], b[o + 1] - b[o], depth + 1, tot));
  }
  return SWZ_OK;
}

// every node of the level the active set has reached needs re-rooting
static int tiler_reroot_level(swz_tiler* t, BatchWork& w, const LevelPlan& plan, const ActiveSet& as, RrTotals& tot) {
  swz_ctx* c = t->c;
  uint32_t* counters = nullptr;
  SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
  uint32_t* hp = nullptr;
  uint64_t* hk = nullptr;
  SWZ_TRY(c->get("tl_head_pos", (size_t)as.m, &hp));
  SWZ_TRY(c->get("tl_head_key", (size_t)as.m, &hk));
  SWZ_TRY(fused_scan(c, HeadF{as.akey, plan.node_shift}, HeadG{as.akey, plan.node_shift, hp, hk}, as.m, counters + 3, "tl"));
  uint32_t heads = 0;
  SWZ_TRY(read_u32(c, counters + 3, &heads));
  std::vector<uint32_t> pos(heads);
  std::vector<uint64_t> key(heads);
  SWZ_HIP(c, hipMemcpyAsync(pos.data(), hp, (size_t)heads * 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipMemcpyAsync(key.data(), hk, (size_t)heads * 8, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  const uint32_t* idx = as.aidx;
  if (!idx) {
    uint32_t* iota = nullptr;
    SWZ_TRY(c->get("rr_iota", (size_t)as.m, &iota));
    hipLaunchKernelGGL(rr_iota_kernel, dim3(div_up(as.m, 256)), dim3(256), 0, c->stream, iota, as.m);
    SWZ_LAUNCH_CHECK(c);
    idx = iota;
  }
  for (uint32_t j = 0; j < heads; ++j) {
    RrNode node;
    node.level = plan.level;
    node.key = key[j];
    swz_node_bounds((int8_t)plan.level, key[j], t->bmin, t->bmax, node.bmin, node.bmax);
    node.max_spacing = t->p.spacing_at_root;
    for (int l = 0; l <= plan.level; ++l) node.max_spacing /= 2;  // child_node.max_spacing /= 2 per level (:138)
    const uint32_t end = j + 1 < heads ? pos[j + 1] : as.m;
    SWZ_TRY(rr_node(t, w, node, t->bmax[0] - t->bmin[0], t->p.spacing_at_root, idx + pos[j], end - pos[j], 0, tot));
  }
  return SWZ_OK;
}

static void zero_stats(swz_tile_stats* s) {
  if (!s) return;
  std::memset(s, 0, sizeof(*s));
  s->max_level = -1;
  s->fast_start_levels = -1;
}

static uint64_t stored_total(const swz_tiler* t) {
  uint64_t s = 0;
  for (int l = 0; l < 22; ++l) s += t->lv[l].cnt;
  return s;
}

// d_xyz: the batch inside the position pool (already there) or anywhere else on the device (copied in).
// Index + sort + positions into Morton order (K1, K2, gather), exactly like a single batch; leaves the batch open.
static int tiler_batch_prepare(swz_tiler* t, double* d_xyz, uint32_t n, uint32_t extra_pool) {
  swz_ctx* c = t->c;
  c->tiler_scratch_dead = false;  // (the batch's scratch lives until tiler_batch_close)
  if (t->finalized) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler: batches cannot be added after finalize");
  if (t->batch_open) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler: the previous batch is still open (swz_tiler_shard_finish)");
  // parallel::scatter throws for a batch with fewer points than indexing threads (util/threading/Parallel.h:181-186)
  if (t->p.strategy == SWZ_FAST && n < t->p.fast_concurrency && !t->shard_fast)  // (a sharded batch: its driver checks the whole batch)
    return c->fail(SWZ_ERR_BAD_ARG, "FAST: a batch needs at least fast_concurrency points");
  const uint32_t base = t->total;
  uint64_t* keys = nullptr;
  uint32_t* perm = nullptr;
  if (n) {
    double* slot = t->pool_xyz + (size_t)base * 3;
    const bool in_pool = d_xyz == slot;
    uint64_t* keys_b = nullptr;
    uint32_t* vals_b = nullptr;
    SWZ_TRY(c->get("tl_keys", (size_t)n, &keys));
    SWZ_TRY(c->get("tl_perm", (size_t)n, &perm));
    SWZ_TRY(c->get("sort_keys_b", (size_t)n, &keys_b));
    SWZ_TRY(c->get("sort_vals_b", (size_t)n, &vals_b));
    if (radix_result_in_second()) {
      SWZ_TRY(encode_device(c, d_xyz, n, t->bmin, t->bmax, keys_b));
      SWZ_TRY(radix_sort_pairs(c, keys_b, vals_b, keys, perm, n, true));
    } else {
      SWZ_TRY(encode_device(c, d_xyz, n, t->bmin, t->bmax, keys));
      SWZ_TRY(radix_sort_pairs(c, keys, perm, keys_b, vals_b, n, true));
    }
    if (!in_pool)  // clamped positions (index_point clamps in place, OctreeAlgorithms.h:167-169) into the pool
      SWZ_HIP(c, hipMemcpyAsync(slot, d_xyz, (size_t)n * 24, hipMemcpyDefault, c->stream));
  }
  BatchWork& w = t->bw;
  w = BatchWork{};
  w.n = n;
  const uint64_t wcap64 = (uint64_t)n + stored_total(t) + extra_pool;
  if (wcap64 > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "batch + stored points exceed 2^32-65536");
  w.wcap = (uint32_t)wcap64;
  SWZ_TRY(c->get("tl_wlevel", (size_t)w.wcap, &w.wlevel));
  SWZ_TRY(c->get("tl_wgid", (size_t)w.wcap, &w.wgid));
  if (n) {  // (positions: on demand, work_need_positions)
    hipLaunchKernelGGL(tl_wgid_kernel, dim3(div_up(n, 256)), dim3(256), 0, c->stream, perm, n, base, w.wgid);
    SWZ_LAUNCH_CHECK(c);
  }
  SWZ_HIP(c, hipMemsetAsync(w.wlevel, 0x80, (size_t)std::max<uint32_t>(w.wcap, 1u), c->stream));
  w.wused = n;
  t->next_level = -1;
  if (t->p.strategy == SWZ_FAST) {
    if (t->fast_start < 0 && !t->shard_fast) SWZ_TRY(fast_start_level(c, keys, n, t->p.fast_concurrency, &t->fast_start));
    t->next_level = t->fast_start - 1;  // (a sharded batch: set again once the driver has told the start level)
  }
  t->as = ActiveSet{keys, nullptr, n};
  t->acc_visited = t->acc_nodes = 0;
  t->acc_rounds = t->acc_levels = 0;
  t->acc_max_level = -1;
  t->batch_open = true;
  return SWZ_OK;
}

// the level loop from t->next_level to last_level (sr: the root level of a sharded batch)
static int tiler_batch_run(swz_tiler* t, int last_level, const ShardRoot* sr) {
  swz_ctx* c = t->c;
  BatchWork& w = t->bw;
  ActiveSet& as = t->as;
  as.parent_prefix = nullptr;  // (known inside one call only, see session_run_levels)
  as.parents = 0;
  for (int level = t->next_level; level <= last_level; ++level) {
    const bool shard_root = sr && sr->active && level == -1;
    if (as.m == 0 && !shard_root) break;
    if (level > 20) return c->fail(SWZ_ERR_INTERNAL, "level loop ran past level 20");
    LevelPlan plan = make_plan(level, t->p.sampler, t->p.max_points_per_node, t->p.spacing_at_root, t->p.max_depth,
                               t->bmin, t->bmax, false, true);
    plan.md_property = (t->p.flags & SWZ_FLAG_MIN_DISTANCE_PROPERTY) != 0 && !shard_root;  // (a sharded root is sampled exactly)
    if (plan.reroot && !plan.terminal && t->p.sampler != SWZ_MIN_DISTANCE) {
      RrTotals tot;
      SWZ_TRY(tiler_reroot_level(t, w, plan, as, tot));
      t->acc_visited += tot.visited;
      t->acc_nodes += tot.nodes;
      t->acc_max_level = std::max(t->acc_max_level, tot.max_level);
      ++t->acc_levels;
      as.m = 0;
      t->next_level = 21;
      return SWZ_OK;
    }
    LevelResult r;
    uint32_t merged = 0;
    SWZ_TRY(tiler_level(t, w, plan, as, &r, &merged, shard_root ? sr : nullptr));
    t->acc_visited += merged;
    t->acc_nodes += r.num_nodes;
    t->acc_rounds += r.md_rounds;
    if (merged) t->acc_max_level = level;
    ++t->acc_levels;
    t->next_level = level + 1;
  }
  return SWZ_OK;
}

static void tiler_batch_close(swz_tiler* t, swz_tile_stats* stats) {
  t->c->tiler_scratch_dead = true;
  t->total += t->bw.n;
  if (t->staged_total < t->total) t->staged_total = t->total;
  ++t->batches;
  t->batch_open = false;
  if (stats) {
    stats->num_nodes = t->acc_nodes;
    stats->points_visited = t->acc_visited;
    stats->max_level = t->acc_max_level;
    stats->fast_start_levels = t->fast_start;
    stats->num_levels = t->acc_levels;
    stats->min_distance_rounds = t->acc_rounds;
  }
}

static int tiler_add_batch(swz_tiler* t, double* d_xyz, uint32_t n, swz_tile_stats* stats) {
  zero_stats(stats);
  if (n == 0) {
    if (t->finalized) return t->c->fail(SWZ_ERR_BAD_ARG, "swz_tiler: batches cannot be added after finalize");
    ++t->batches;
    return SWZ_OK;
  }
  int st = tiler_batch_prepare(t, d_xyz, n, 0);
  if (st == SWZ_OK) st = tiler_batch_run(t, 20, nullptr);
  if (st != SWZ_OK) {
    t->batch_open = false;
    return st;
  }
  tiler_batch_close(t, stats);
  return SWZ_OK;
}

// TilingAlgorithmV3::finalize -> reconstruct_left_out_nodes (:1717-1784): every ancestor of a start node samples the
// points its children hold (children in octant order = the store order of the level below), re-indexed against the
// root bounds, with AlwaysAdhereToMinSpacing (reconstruct_single_node :1661-1715).
// While a call outside a batch holds "tl_*" buffers across level_step (whose MIN_DISTANCE levels open new scratch
// epochs and may run out of memory), the workspace must not count them as dead scratch: get() would free the very
// arrays the level is reading.
struct TilerScratchLive {
  swz_ctx* c;
  bool was;
  explicit TilerScratchLive(swz_ctx* ctx) : c(ctx), was(ctx->tiler_scratch_dead) { c->tiler_scratch_dead = false; }
  ~TilerScratchLive() { c->tiler_scratch_dead = was; }
};

static int tiler_finalize(swz_tiler* t, swz_tile_stats* stats, int lowest_children = 0) {
  swz_ctx* c = t->c;
  zero_stats(stats);
  TilerScratchLive live(c);
  if (t->finalized) return SWZ_OK;
  if (!t->staged_sizes.empty()) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_finalize: staged batches have not been tiled");
  t->finalized = lowest_children == 0;  // (a shard stops below the root: swz_tiler_shard_fast_set_root ends the data set)
  if (t->p.strategy != SWZ_FAST || t->fast_start <= 0) return SWZ_OK;
  const int S = t->fast_start;
  uint64_t nodes = 0;
  uint32_t rounds = 0;
  uint32_t* counters = nullptr;
  SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
  for (int lv = S - 1; lv >= lowest_children; --lv) {  // children at node level lv, parents at lv - 1
    StoreLevel& src = t->lv[lv + 1];
    StoreLevel& dst = t->lv[lv];
    const uint32_t m = src.cnt;
    if (m == 0) continue;
    SWZ_TRY(store_linearize(c, src, lv + 1));
    uint64_t* keys = nullptr;
    uint32_t *gid = nullptr, *wgid = nullptr;
    double *wx = nullptr, *wy = nullptr, *wz = nullptr;
    SWZ_TRY(c->get("tl_keys", (size_t)m, &keys));
    SWZ_TRY(c->get("tl_perm", (size_t)m, &gid));
    SWZ_TRY(c->get("tl_wgid", (size_t)m, &wgid));
    LevelPlan plan = make_plan(lv - 1, t->p.sampler, t->p.max_points_per_node, t->p.spacing_at_root, t->p.max_depth,
                               t->bmin, t->bmax, true, false);
    plan.md_property = (t->p.flags & SWZ_FLAG_MIN_DISTANCE_PROPERTY) != 0;
    // Like the levels of a batch (tiler_level), the rebuilt levels are decided on key coordinates with the exact positions
    // looked up in the pool through the point ids (round 5: they gathered the positions and ran the sweep on positions --
    // 253 of the 1 073 ms of the default operating point's data set, 1 B points in 100 tile batches); positions in
    // Morton order only for what cannot be decided on keys.
    SortedPoints sp{nullptr, nullptr, nullptr, t->pool_xyz, wgid};
    const bool on_keys = level_decides_on_keys(c, plan, sp);
    if (!on_keys) {
      SWZ_TRY(c->get("tl_wx", (size_t)m, &wx));
      SWZ_TRY(c->get("tl_wy", (size_t)m, &wy));
      SWZ_TRY(c->get("tl_wz", (size_t)m, &wz));
      sp.X = wx;
      sp.Y = wy;
      sp.Z = wz;
    }
    SWZ_HIP(c, hipMemcpyAsync(gid, src.gid[src.cur], (size_t)m * 4, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(tl_reencode_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, gid, m, t->pool_xyz, root_box(t), keys);
    SWZ_LAUNCH_CHECK(c);
    // inside one PARENT the appended children must ascend (reconstruct_single_node does not sort a lossless store)
    const uint32_t psh = lv == 0 ? 63u : level_shift(lv - 1);
    SWZ_HIP(c, hipMemsetAsync(counters + 1, 0, 4, c->stream));
    hipLaunchKernelGGL(tl_inversion_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, keys, m, psh, counters + 1);
    SWZ_LAUNCH_CHECK(c);
    uint32_t inv = 0;
    SWZ_TRY(read_u32(c, counters + 1, &inv));
    if (inv) {
      t->rekey_inversions += inv;
      SWZ_TRY(sort_pairs_by_key(c, keys, gid, m));
    }
    hipLaunchKernelGGL(tl_fill_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, gid, m, t->pool_xyz, wx, wy, wz, wgid);
    SWZ_LAUNCH_CHECK(c);
    LevelBuffers lb;
    SWZ_TRY(alloc_level_buffers(c, m, &lb));
    ActiveSet as{keys, nullptr, m};
    LevelResult r;
    SWZ_TRY(level_step(c, plan, as, sp, lb, nullptr, nullptr, nullptr, &r));
    // count the taken points, then write them as the parents' files
    uint64_t* tkey = nullptr;
    uint32_t* tgid = nullptr;
    SWZ_TRY(c->get("tl_tkey", (size_t)m, &tkey));
    SWZ_TRY(c->get("tl_tgid", (size_t)m, &tgid));
    SWZ_TRY(fused_scan(c, TakenF{lb.taken}, TakeG{keys, nullptr, wgid, tkey, tgid}, m, counters + 2, "tl"));
    uint32_t nt = 0;
    SWZ_TRY(read_u32(c, counters + 2, &nt));
    const int w = dst.cur ^ 1;
    SWZ_TRY(store_reserve(c, dst, lv, w, nt));
    SWZ_HIP(c, hipMemcpyAsync(dst.key[w], tkey, (size_t)nt * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_HIP(c, hipMemcpyAsync(dst.gid[w], tgid, (size_t)nt * 4, hipMemcpyDeviceToDevice, c->stream));
    store_written_linear(dst, w, nt, false);
    nodes += r.num_nodes;
    rounds += r.md_rounds;
  }
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (stats) {
    stats->num_nodes = nodes;
    stats->fast_start_levels = S;
    stats->min_distance_rounds = rounds;
  }
  return SWZ_OK;
}

// node table of the store: per level the runs of equal node prefix
static int tiler_node_table(swz_tiler* t, std::vector<int8_t>* nl, std::vector<uint64_t>* nk, std::vector<uint64_t>* no,
                            std::vector<uint64_t>* nc, uint64_t* num_nodes) {
  swz_ctx* c = t->c;
  uint64_t offset = 0, nn = 0;
  uint32_t* counters = nullptr;
  SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
  for (int l = 0; l < 22; ++l) {
    StoreLevel& s = t->lv[l];
    if (!s.cnt) continue;
    if (!nl && s.table_valid) {  // only the count is asked for
      nn += s.nn;
      offset += s.cnt;
      continue;
    }
    SWZ_TRY(store_linearize(c, s, l));
    const int level = l - 1;
    const uint32_t nsh = level < 0 ? 63u : level_shift(level);
    uint32_t* hp = nullptr;
    uint64_t* hk = nullptr;
    SWZ_TRY(c->get("tl_head_pos", (size_t)s.cnt, &hp));
    SWZ_TRY(c->get("tl_head_key", (size_t)s.cnt, &hk));
    SWZ_TRY(fused_scan(c, HeadF{s.key[s.cur], nsh}, HeadG{s.key[s.cur], nsh, hp, hk}, s.cnt, counters + 3, "tl"));
    uint32_t heads = 0;
    SWZ_TRY(read_u32(c, counters + 3, &heads));
    if (nl) {
      std::vector<uint32_t> pos(heads);
      std::vector<uint64_t> key(heads);
      SWZ_HIP(c, hipMemcpyAsync(pos.data(), hp, (size_t)heads * 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipMemcpyAsync(key.data(), hk, (size_t)heads * 8, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      for (uint32_t k = 0; k < heads; ++k) {
        nl->push_back((int8_t)level);
        nk->push_back(key[k]);
        no->push_back(offset + pos[k]);
        nc->push_back((k + 1 < heads ? pos[k + 1] : s.cnt) - pos[k]);
      }
    }
    nn += heads;
    offset += s.cnt;
  }
  *num_nodes = nn;
  return SWZ_OK;
}

}  // namespace swz

using namespace swz;

// ================================================================================================== C ABI
extern "C" {

int swz_tiler_create(swz_ctx* c, const double bmin[3], const double bmax[3], const swz_tile_params* params,
                     uint64_t capacity_hint, swz_tiler** out) {
  if (!c || !out) return SWZ_ERR_BAD_ARG;
  *out = nullptr;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (!bmin || !bmax || !params) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_create: NULL argument");
  for (int a = 0; a < 3; ++a)
    if (!(bmax[a] > bmin[a])) return c->fail(SWZ_ERR_BAD_ARG, "bounds must have positive extent on every axis");
  if (params->sampler < SWZ_RANDOM_GRID || params->sampler > SWZ_JITTERED) return c->fail(SWZ_ERR_BAD_ARG, "unknown sampler");
  if (params->strategy != SWZ_ACCURATE && params->strategy != SWZ_FAST) return c->fail(SWZ_ERR_BAD_ARG, "unknown strategy");
  if (!(params->spacing_at_root > 0.f)) return c->fail(SWZ_ERR_BAD_ARG, "spacing_at_root must be > 0");
  if (params->strategy == SWZ_FAST && params->fast_concurrency == 0)
    return c->fail(SWZ_ERR_BAD_ARG, "FAST needs fast_concurrency >= 1");
  if (capacity_hint > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points per tiler");
  if (c->tiler_active) return c->fail(SWZ_ERR_BAD_ARG, "this context already has a tiler (one per context; use one context per data set)");
  swz_tiler* t = new swz_tiler();
  t->c = c;
  for (int a = 0; a < 3; ++a) {
    t->bmin[a] = bmin[a];
    t->bmax[a] = bmax[a];
  }
  t->p = *params;
  hipError_t e = hipStreamCreateWithFlags(&t->copy_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete t;
    return c->fail(SWZ_ERR_HIP, std::string("hipStreamCreate(copy stream): ") + hipGetErrorString(e));
  }
  if (capacity_hint) {
    const int st = pool_reserve(t, (size_t)capacity_hint);
    if (st != SWZ_OK) {
      (void)hipStreamDestroy(t->copy_stream);
      delete t;
      return st;
    }
  }
  c->tiler_active = true;
  *out = t;
  return SWZ_OK;
}

int swz_tiler_destroy(swz_tiler* t) {
  if (!t) return SWZ_OK;
  (void)hipSetDevice(t->c->device);
  if (t->copy_stream) (void)hipStreamSynchronize(t->copy_stream);
  (void)hipStreamSynchronize(t->c->stream);
  for (hipEvent_t e : t->staged_events) (void)hipEventDestroy(e);
  for (int l = 0; l < 22; ++l) store_free(t->lv[l]);  // the memory stays in the context's workspace (swz_release_workspace)
  t->c->tiler_active = false;
  if (t->copy_stream) (void)hipStreamDestroy(t->copy_stream);
  delete t;
  return SWZ_OK;
}

int swz_tiler_add_batch_device(swz_tiler* t, double* d_xyz, uint64_t n, swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (!t->staged_sizes.empty()) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_add_batch_device: staged batches are pending");
  if (n && !d_xyz) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_add_batch_device: NULL buffer");
  if ((uint64_t)t->total + n > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points per tiler");
  if (t->attr_mask) return c->fail(SWZ_ERR_BAD_ARG, "this tiler carries attribute columns: use swz_tiler_stage_batch");
  SWZ_TRY(pool_reserve(t, (size_t)t->total + n));
  const int st = tiler_add_batch(t, d_xyz, (uint32_t)n, stats);
  const hipError_t e = hipStreamSynchronize(c->stream);
  c->prof_collect();
  if (st != SWZ_OK) return tiler_poison(t, st);
  if (e != hipSuccess) return tiler_poison(t, c->hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
  return SWZ_OK;
}

// ---- a tiler per GPU of a multi-GPU run: the shard owns the subtrees of its level-0 octants and ITS part of the root's
// file; the root node itself spans the shards (swz_shard_* for the single-batch form)
int swz_tiler_shard_begin_device(swz_tiler* t, double* d_xyz, uint64_t n, const swz_attribute_columns* d_attrs,
                                 const swz_tiler_shard_info* info, uint64_t* root_file_count_out) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (!info) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_begin_device: NULL shard info");
  const bool fast = t->p.strategy == SWZ_FAST;
  if (!t->staged_sizes.empty()) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_begin_device: staged batches are pending");
  if ((n && !d_xyz) || (info->num_ghosts && !info->d_ghost_xyz)) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_begin_device: NULL buffer");
  if ((uint64_t)t->total + n > 0xFFFF0000ull || info->num_ghosts > 0x7FFFFFFFull)
    return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points per tiler");
  uint32_t mask = 0;
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (d_attrs && d_attrs->column[a]) mask |= 1u << a;
  if (t->total == 0 && t->batches == 0) {
    t->attr_mask = mask;
    if (mask && t->pool_cap) {
      const size_t cap = t->pool_cap;
      t->pool_cap = 0;
      t->pool_xyz = nullptr;
      SWZ_TRY(pool_reserve(t, cap));
    }
  } else if (mask != t->attr_mask && n) {
    return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_begin_device: every batch must carry the same attribute columns");
  }
  SWZ_TRY(pool_reserve(t, (size_t)t->total + n));
  for (int a = 0; a < SWZ_ATTR_COUNT && n; ++a) {
    if (!(t->attr_mask & (1u << a))) continue;
    const size_t rb = TILER_ATTR_BYTES[a];
    SWZ_HIP(c, hipMemcpyAsync((char*)t->pool_attr[a] + (size_t)t->total * rb, d_attrs->column[a], (size_t)n * rb,
                              hipMemcpyDefault, c->stream));
  }
  t->shard_fast = fast;
  int st = tiler_batch_prepare(t, d_xyz, (uint32_t)n, (uint32_t)info->num_ghosts);
  t->shard_fast = false;
  // FAST (TilingAlgorithmV3) skips the levels above its start nodes until the data set ends: no root step per batch
  if (st == SWZ_OK && info->global_new_points > 0 && !fast) {
    ShardRoot sr;
    sr.active = true;
    // tile_internal_node :272-275 with the counts of the WHOLE root: cached points force sampling, else count <= max takes all
    sr.sample = info->global_root_stored > 0 || info->global_new_points + info->global_root_stored > t->p.max_points_per_node;
    sr.ghost_xyz = info->d_ghost_xyz;
    sr.ghosts = sr.sample ? (uint32_t)info->num_ghosts : 0u;
    st = tiler_batch_run(t, -1, &sr);
  }
  const hipError_t e = hipStreamSynchronize(c->stream);
  c->prof_collect();
  if (st != SWZ_OK) return tiler_poison(t, st);
  if (e != hipSuccess) return tiler_poison(t, c->hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
  if (root_file_count_out) *root_file_count_out = t->lv[0].cnt;
  return SWZ_OK;
}

int swz_tiler_shard_finish(swz_tiler* t, swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  zero_stats(stats);
  if (!t->batch_open) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_finish: no batch is open");
  if (t->p.strategy == SWZ_FAST) {
    if (t->fast_start < 1) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_finish: FAST needs swz_tiler_shard_set_start_level first");
    t->next_level = t->fast_start - 1;
  } else if (t->next_level < 0) {
    t->next_level = 0;  // the root was skipped (an empty batch)
  }
  const int st = tiler_batch_run(t, 20, nullptr);
  const hipError_t e = hipStreamSynchronize(c->stream);
  c->prof_collect();
  if (st != SWZ_OK) return tiler_poison(t, st);
  if (e != hipSuccess) return tiler_poison(t, c->hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
  tiler_batch_close(t, stats);
  return SWZ_OK;
}

// ---- FAST on a sharded data set.  The start level of TilingAlgorithmV3 comes from the distribution of the FIRST batch
// (TilingAlgorithms.cpp:1473-1535, kept afterwards :1230-1236): every shard reports the counts of its part per 6-octant
// prefix, the driver sums them and tells every shard the level.  Start nodes lie at level >= 2, inside one shard's
// octants, so batches need no root step; when the data set ends every shard rebuilds the skipped levels of its own
// octants down to level 0 and the root is rebuilt from the level-0 files of all shards by the driver.
int swz_tiler_shard_fast_histogram(swz_tiler* t, uint32_t* counts_out) {
  if (!t || !counts_out) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (!t->batch_open) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_fast_histogram: no batch is open");
  return tiler_poison(t, fast_prefix_counts(c, t->as.akey, t->as.m, counts_out));
}
int swz_fast_start_level_from_counts(const uint64_t* counts, uint32_t fast_concurrency, int32_t* start_level_out) {
  if (!counts || !start_level_out || !fast_concurrency) return SWZ_ERR_BAD_ARG;
  *start_level_out = fast_start_level_from_counts(counts, fast_concurrency);
  return SWZ_OK;
}
int swz_tiler_shard_set_start_level(swz_tiler* t, int32_t start_level) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_TRY(tiler_guard(t));
  if (t->p.strategy != SWZ_FAST || start_level < 1 || start_level > 6) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_set_start_level: FAST tilers, levels 1..6");
  if (t->fast_start >= 0 && t->fast_start != start_level) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_set_start_level: the start level is fixed by the first batch");
  t->fast_start = start_level;
  return SWZ_OK;
}
int swz_tiler_shard_fast_finalize_local(swz_tiler* t, swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (t->p.strategy != SWZ_FAST) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_fast_finalize_local: not a FAST tiler");
  const int st = tiler_finalize(t, stats, 1);  // parents down to level 0; the root spans the shards
  c->prof_collect();
  return tiler_poison(t, st);
}
// d_taken: one flag per entry of this shard's level-0 files, in file order: the entry belongs to the rebuilt root
int swz_tiler_shard_fast_set_root(swz_tiler* t, const uint8_t* d_taken) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (t->p.strategy != SWZ_FAST || t->finalized) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_fast_set_root: an open FAST tiler is needed");
  StoreLevel& src = t->lv[1];
  StoreLevel& dst = t->lv[0];
  const uint32_t m = src.cnt;
  t->finalized = true;
  if (m == 0) return SWZ_OK;
  if (!d_taken) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_shard_fast_set_root: NULL flags");
  TilerScratchLive live(c);
  auto body = [&]() -> int {
    SWZ_TRY(store_linearize(c, src, 1));
    uint64_t *keys = nullptr, *tkey = nullptr;
    uint32_t *tgid = nullptr, *counters = nullptr;
    SWZ_TRY(c->get("tl_keys", (size_t)m, &keys));
    SWZ_TRY(c->get("tl_tkey", (size_t)m, &tkey));
    SWZ_TRY(c->get("tl_tgid", (size_t)m, &tgid));
    SWZ_TRY(c->get("tl_counters", (size_t)4, &counters));
    hipLaunchKernelGGL(tl_reencode_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, src.gid[src.cur], m, t->pool_xyz, root_box(t), keys);
    SWZ_LAUNCH_CHECK(c);
    SWZ_TRY(fused_scan(c, TakenF{d_taken}, TakeG{keys, nullptr, src.gid[src.cur], tkey, tgid}, m, counters + 2, "tl"));
    uint32_t nt = 0;
    SWZ_TRY(read_u32(c, counters + 2, &nt));
    const int w = dst.cur ^ 1;
    SWZ_TRY(store_reserve(c, dst, 0, w, nt));
    SWZ_HIP(c, hipMemcpyAsync(dst.key[w], tkey, (size_t)nt * 8, hipMemcpyDeviceToDevice, c->stream));
    SWZ_HIP(c, hipMemcpyAsync(dst.gid[w], tgid, (size_t)nt * 4, hipMemcpyDeviceToDevice, c->stream));
    store_written_linear(dst, w, nt, false);
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    return SWZ_OK;
  };
  return tiler_poison(t, body());
}

int swz_tiler_poison(swz_tiler* t, const char* why) {
  if (!t) return SWZ_ERR_BAD_ARG;
  if (!t->failed) {
    t->failed = true;
    t->failed_why = why ? why : "poisoned by the caller";
    t->batch_open = false;
  }
  return SWZ_OK;
}

int swz_tiler_level_count(swz_tiler* t, int level, uint64_t* count_out) {
  if (!t || !count_out || level < -1 || level > 20) return SWZ_ERR_BAD_ARG;
  SWZ_TRY(tiler_guard(t));
  *count_out = t->lv[level + 1].cnt;
  return SWZ_OK;
}

int swz_tiler_level_positions_device(swz_tiler* t, int level, double* d_xyz_out) {
  if (!t || level < -1 || level > 20) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  StoreLevel& s = t->lv[level + 1];
  if (!s.cnt) return SWZ_OK;
  if (!d_xyz_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_level_positions_device: NULL buffer");
  SWZ_TRY(store_linearize(c, s, level + 1));
  hipLaunchKernelGGL(tl_rows_kernel, dim3(div_up(s.cnt, 256)), dim3(256), 0, c->stream, s.gid[s.cur], s.cnt, t->pool_xyz, d_xyz_out);
  SWZ_LAUNCH_CHECK(c);
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

int swz_tiler_stage_batch(swz_tiler* t, const double* xyz_host, uint64_t n, const swz_attribute_columns* attrs_host) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (t->finalized) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler: batches cannot be added after finalize");
  if (t->staged_sizes.size() >= 2) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_stage_batch: two batches are already staged");
  if (n && !xyz_host) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_stage_batch: NULL buffer");
  if ((uint64_t)t->staged_total + n > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points per tiler");
  uint32_t mask = 0;
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (attrs_host && attrs_host->column[a]) mask |= 1u << a;
  if (t->staged_total == 0 && t->batches == 0) {
    t->attr_mask = mask;
    if (mask && t->pool_cap) {  // the pools were presized before the columns were known
      const size_t cap = t->pool_cap;
      t->pool_cap = 0;
      t->pool_xyz = nullptr;
      SWZ_TRY(pool_reserve(t, cap));
    }
  } else if (mask != t->attr_mask) {
    return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_stage_batch: every batch must carry the same attribute columns");
  }
  SWZ_TRY(pool_reserve(t, (size_t)t->staged_total + n));
  const size_t at = t->staged_total;
  if (n) {
    SWZ_HIP(c, hipMemcpyAsync(t->pool_xyz + at * 3, xyz_host, (size_t)n * 24, hipMemcpyDefault, t->copy_stream));
    t->staged_bytes += n * 24;
    for (int a = 0; a < SWZ_ATTR_COUNT; ++a) {
      if (!(mask & (1u << a))) continue;
      const size_t rb = TILER_ATTR_BYTES[a];
      SWZ_HIP(c, hipMemcpyAsync((char*)t->pool_attr[a] + at * rb, attrs_host->column[a], (size_t)n * rb,
                                hipMemcpyDefault, t->copy_stream));
      t->staged_bytes += n * rb;
    }
  }
  hipEvent_t ev = nullptr;
  SWZ_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  SWZ_HIP(c, hipEventRecord(ev, t->copy_stream));
  t->staged_events.push_back(ev);
  t->staged_sizes.push_back((uint32_t)n);
  t->staged_total += (uint32_t)n;
  return SWZ_OK;
}

int swz_tiler_tile_staged(swz_tiler* t, swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (t->staged_sizes.empty()) return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_tile_staged: nothing is staged");
  const uint32_t n = t->staged_sizes.front();
  hipEvent_t ev = t->staged_events.front();
  t->staged_sizes.erase(t->staged_sizes.begin());
  t->staged_events.erase(t->staged_events.begin());
  // this batch needs its copy complete; the copy of the NEXT staged batch keeps running beside the kernels
  const auto t0 = std::chrono::steady_clock::now();
  SWZ_HIP(c, hipEventSynchronize(ev));
  t->staged_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  const int st = tiler_add_batch(t, t->pool_xyz + (size_t)t->total * 3, n, stats);
  const hipError_t e = hipStreamSynchronize(c->stream);
  (void)hipEventDestroy(ev);
  c->prof_collect();
  // (a failure also leaves the pools one staged batch ahead of the ids the next batch would use)
  if (st != SWZ_OK) return tiler_poison(t, st);
  if (e != hipSuccess) return tiler_poison(t, c->hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__));
  return SWZ_OK;
}

int swz_tiler_add_batch(swz_tiler* t, const double* xyz_host, uint64_t n, const swz_attribute_columns* attrs_host,
                        swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  if (!t->staged_sizes.empty()) return t->c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_add_batch: staged batches are pending");
  SWZ_TRY(swz_tiler_stage_batch(t, xyz_host, n, attrs_host));
  return swz_tiler_tile_staged(t, stats);
}

int swz_tiler_finalize(swz_tiler* t, swz_tile_stats* stats) {
  if (!t) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(t->c, hipSetDevice(t->c->device));
  SWZ_TRY(tiler_guard(t));
  const int st = tiler_finalize(t, stats);
  t->c->prof_collect();
  return tiler_poison(t, st);
}

int swz_tiler_get_info(swz_tiler* t, swz_tiler_info* info) {
  if (!t || !info) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(t->c, hipSetDevice(t->c->device));
  std::memset(info, 0, sizeof(*info));
  info->num_points = t->total;
  info->num_stored = stored_total(t);
  info->num_batches = t->batches;
  info->rekey_inversions = t->rekey_inversions;
  info->fast_start_levels = t->fast_start;
  info->staged_bytes = t->staged_bytes;
  info->staged_wait_ms = t->staged_wait_ms;
  uint64_t nn = 0;
  SWZ_TRY(tiler_node_table(t, nullptr, nullptr, nullptr, nullptr, &nn));
  info->num_nodes = nn;
  return SWZ_OK;
}

int swz_tiler_pool_residency(swz_tiler* t, uint64_t* device_bytes_out, uint64_t* host_bytes_out) {
  if (!t) return SWZ_ERR_BAD_ARG;
  uint64_t dev = 0, host = 0;
  auto add = [&](const std::string& name) {
    auto it = t->c->bufs.find(name);
    if (it == t->c->bufs.end()) return;
    (it->second.host ? host : dev) += it->second.cap;
  };
  add("tiler_pool_xyz");
  for (int a = 0; a < SWZ_ATTR_COUNT; ++a)
    if (t->attr_mask & (1u << a)) add("tiler_pool_attr" + std::to_string(a));
  if (device_bytes_out) *device_bytes_out = dev;
  if (host_bytes_out) *host_bytes_out = host;
  return SWZ_OK;
}

int swz_tiler_reserve(swz_tiler* t, uint64_t total_points) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  if (total_points > 0xFFFF0000ull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "more than 2^32-65536 points per tiler");
  if (total_points <= t->pool_cap && t->pool_xyz) return SWZ_OK;
  return pool_reserve(t, (size_t)total_points);
}

int swz_tiler_store_residency(swz_tiler* t, uint64_t* device_bytes_out, uint64_t* host_bytes_out) {
  if (!t) return SWZ_ERR_BAD_ARG;
  uint64_t dev = 0, host = 0;
  for (const auto& kv : t->c->bufs)
    if (kv.first.compare(0, 11, "tiler_store") == 0) (kv.second.host ? host : dev) += kv.second.cap;
  if (device_bytes_out) *device_bytes_out = dev;
  if (host_bytes_out) *host_bytes_out = host;
  return SWZ_OK;
}

int swz_tiler_node_table(swz_tiler* t, uint64_t max_nodes, int8_t* node_level_out, uint64_t* node_key_out,
                         uint64_t* node_offset_out, uint64_t* node_count_out, uint64_t* num_nodes_out) {
  if (!t || !num_nodes_out) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  std::vector<int8_t> nl;
  std::vector<uint64_t> nk, no, nc;
  uint64_t nn = 0;
  SWZ_TRY(tiler_node_table(t, &nl, &nk, &no, &nc, &nn));
  *num_nodes_out = nn;
  if (nn > max_nodes) return c->fail(SWZ_ERR_BAD_ARG, "max_nodes too small");
  if (!node_level_out || !node_key_out || !node_offset_out || !node_count_out)
    return c->fail(SWZ_ERR_BAD_ARG, "swz_tiler_node_table: NULL buffer");
  std::memcpy(node_level_out, nl.data(), nn);
  std::memcpy(node_key_out, nk.data(), nn * 8);
  std::memcpy(node_offset_out, no.data(), nn * 8);
  std::memcpy(node_count_out, nc.data(), nn * 8);
  return SWZ_OK;
}

int swz_tiler_export_device(swz_tiler* t, uint64_t* d_keys_out, uint32_t* d_ids_out, int8_t* d_level_out) {
  if (!t) return SWZ_ERR_BAD_ARG;
  swz_ctx* c = t->c;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_TRY(tiler_guard(t));
  size_t off = 0;
  for (int l = 0; l < 22; ++l) {
    StoreLevel& s = t->lv[l];
    if (!s.cnt) continue;
    SWZ_TRY(store_linearize(c, s, l));
    if (d_keys_out) SWZ_HIP(c, hipMemcpyAsync(d_keys_out + off, s.key[s.cur], (size_t)s.cnt * 8, hipMemcpyDeviceToDevice, c->stream));
    if (d_ids_out) SWZ_HIP(c, hipMemcpyAsync(d_ids_out + off, s.gid[s.cur], (size_t)s.cnt * 4, hipMemcpyDeviceToDevice, c->stream));
    if (d_level_out) {
      hipLaunchKernelGGL(tl_fill_level_kernel, dim3(div_up(s.cnt, 256)), dim3(256), 0, c->stream, d_level_out + off, s.cnt,
                         (int8_t)(l - 1));
      SWZ_LAUNCH_CHECK(c);
    }
    off += s.cnt;
  }
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

// One batch as node files (see the header): a tiler of its own for the batch, closed by the second call.
int swz_tile_nodes_begin_device(swz_ctx* c, double* d_xyz, uint64_t n, const double bmin[3], const double bmax[3],
                                const swz_tile_params* params, uint64_t* num_stored_out, uint64_t* num_nodes_out, swz_tile_stats* stats) {
  if (!c) return SWZ_ERR_BAD_ARG;
  if (!num_stored_out || !num_nodes_out) return c->fail(SWZ_ERR_BAD_ARG, "swz_tile_nodes_begin_device: NULL argument");
  if (c->nodes_tiler) return c->fail(SWZ_ERR_BAD_ARG, "swz_tile_nodes_begin_device: the call before has not been closed (swz_tile_nodes_end_device)");
  swz_tiler* t = nullptr;
  SWZ_TRY(swz_tiler_create(c, bmin, bmax, params, n, &t));
  swz_tile_stats batch{}, fin{};
  int st = swz_tiler_add_batch_device(t, d_xyz, n, &batch);
  if (st == SWZ_OK) st = swz_tiler_finalize(t, &fin);
  swz_tiler_info info{};
  if (st == SWZ_OK) st = swz_tiler_get_info(t, &info);
  if (st != SWZ_OK) {
    const std::string why = c->err;  // (destroying the tiler must not lose the reason)
    (void)swz_tiler_destroy(t);
    return c->fail(st, why);
  }
  if (stats) {
    *stats = batch;
    stats->num_nodes = (uint32_t)info.num_nodes;
    stats->points_visited += fin.points_visited;
    stats->num_levels += fin.num_levels;
    stats->min_distance_rounds += fin.min_distance_rounds;
    stats->max_level = std::max(batch.max_level, fin.max_level);
  }
  *num_stored_out = info.num_stored;
  *num_nodes_out = info.num_nodes;
  c->nodes_tiler = t;
  return SWZ_OK;
}

int swz_tile_nodes_end_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_ids_out, int8_t* d_level_out, uint64_t max_nodes,
                              int8_t* node_level_out, uint64_t* node_key_out, uint64_t* node_offset_out, uint64_t* node_count_out) {
  if (!c) return SWZ_ERR_BAD_ARG;
  swz_tiler* t = c->nodes_tiler;
  if (!t) return c->fail(SWZ_ERR_BAD_ARG, "swz_tile_nodes_end_device: no call is open (swz_tile_nodes_begin_device)");
  int st = SWZ_OK;
  if (d_keys_out || d_ids_out || d_level_out) st = swz_tiler_export_device(t, d_keys_out, d_ids_out, d_level_out);
  if (st == SWZ_OK && (max_nodes || node_level_out || node_key_out || node_offset_out || node_count_out)) {
    uint64_t nn = 0;
    st = swz_tiler_node_table(t, max_nodes, node_level_out, node_key_out, node_offset_out, node_count_out, &nn);
  }
  const std::string why = c->err;
  c->nodes_tiler = nullptr;
  (void)swz_tiler_destroy(t);
  return st == SWZ_OK ? SWZ_OK : c->fail(st, why);
}

int swz_tiler_pools_device(swz_tiler* t, const double** d_xyz_out, swz_attribute_columns* d_attrs_out) {
  if (!t) return SWZ_ERR_BAD_ARG;
  if (d_xyz_out) *d_xyz_out = t->pool_xyz;
  if (d_attrs_out)
    for (int a = 0; a < SWZ_ATTR_COUNT; ++a) d_attrs_out->column[a] = (t->attr_mask & (1u << a)) ? t->pool_attr[a] : nullptr;
  return SWZ_OK;
}

int swz_host_alloc_pinned(uint64_t bytes, void** out) {
  if (!out) return SWZ_ERR_BAD_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? SWZ_OK : SWZ_ERR_HIP;
}
int swz_host_free_pinned(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? SWZ_OK : SWZ_ERR_HIP; }

int swz_device_alloc(uint64_t bytes, void** d_out) {
  if (!d_out) return SWZ_ERR_BAD_ARG;
  *d_out = nullptr;
  return hipMalloc(d_out, bytes ? bytes : 1) == hipSuccess ? SWZ_OK : SWZ_ERR_HIP;
}
int swz_device_alloc_on(swz_ctx* c, uint64_t bytes, void** d_out) {
  if (!c || !d_out) return SWZ_ERR_BAD_ARG;
  *d_out = nullptr;
  SWZ_HIP(c, hipSetDevice(c->device));
  SWZ_HIP(c, hipMalloc(d_out, bytes ? bytes : 1));
  return SWZ_OK;
}
int swz_device_free(void* d_ptr) { return (!d_ptr || hipFree(d_ptr) == hipSuccess) ? SWZ_OK : SWZ_ERR_HIP; }
// (a pointer the library hands out may be a spilled pool: page-locked HOST memory mapped into the device's address space)
static bool is_mapped_host_memory(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}
int swz_copy_to_host(swz_ctx* c, void* dst_host, const void* d_src, uint64_t bytes) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (bytes && is_mapped_host_memory(d_src)) {
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(dst_host, d_src, bytes);
    return SWZ_OK;
  }
  if (bytes) SWZ_HIP(c, hipMemcpyAsync(dst_host, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}
int swz_copy_to_device(swz_ctx* c, void* d_dst, const void* src_host, uint64_t bytes) {
  if (!c) return SWZ_ERR_BAD_ARG;
  SWZ_HIP(c, hipSetDevice(c->device));
  if (bytes && is_mapped_host_memory(d_dst)) {
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(d_dst, src_host, bytes);
    return SWZ_OK;
  }
  if (bytes) SWZ_HIP(c, hipMemcpyAsync(d_dst, src_host, bytes, hipMemcpyHostToDevice, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

}  // extern "C"
