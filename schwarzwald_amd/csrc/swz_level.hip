// swz_level.hip -- level-synchronous octree tiling (K3, K4a/b/d, K5) and the tiler drivers.
//
// The reference recurses top-down per node (TilingAlgorithmBase::do_tiling_for_node /
// tile_node / tile_internal_node, core/tiling/TilingAlgorithms.cpp:499-561, 351-492, 247-349): sample
// a node's Morton-sorted points (stable partition), persist the taken ones, split the rest into the
// <= 8 child octants (:116-162) and recurse.  Here all nodes of one level are processed together:
// the "active set" is the Morton-sorted array of points not yet taken; a node is a run of equal key
// prefix, a sampling-grid cell a run of a longer prefix; taken points get their level recorded and
// the survivors are stream-compacted (stable) into the next level's active set.
#include <algorithm>
#include <cmath>
#include <vector>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

#define SWZ_JITTER_TABLE(W) __constant__ uint8_t PERMUTATIONS_##W[16 * W]
#include "jitter_tables.inc"
#undef SWZ_JITTER_TABLE

__device__ __forceinline__ uint32_t spos_of(const uint32_t* aidx, uint32_t i) { return aidx ? aidx[i] : i; }

// ----------------------------------------------------------------------------- node segmentation
// fused form: node-head flag computed from the keys inside the scan, node id / node start written by it
struct NodeHeadF {
  const uint64_t* akey;
  uint32_t nsh;
  __device__ uint32_t operator()(uint32_t i) const {
    return (i == 0) ? 1u : (uint32_t)((akey[i] >> nsh) != (akey[i - 1] >> nsh));
  }
};
struct NodeAssignG {
  uint32_t* nid;
  uint32_t* nstart;
  uint32_t m;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t head) const {
    const uint32_t id = excl + head - 1u;
    nid[i] = id;
    if (head) nstart[id] = i;
    if (i == m - 1) nstart[id + 1] = m;
  }
};
// ---- segmentation by search: the nodes of a level are children of the nodes of the level above, and the active set is
// sorted by node prefix, so child o of parent P starts at the lower bound of (P, o) among the keys.  Nine searches per
// parent node instead of two passes over the keys of every point (20 bytes per point: the segmentation was as expensive
// as the sort's histogram passes); the node id per point, which the samplers look up, is then filled from the node starts.
__global__ __launch_bounds__(256) void node_child_bounds_kernel(const uint64_t* __restrict__ akey, uint32_t m, uint32_t nsh,
                                                                const uint64_t* __restrict__ pprefix, uint32_t parents,
                                                                uint32_t* __restrict__ cb) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t j = t >> 4, o = t & 15u;
  if (j >= parents || o > 8u) return;
  const uint64_t target = (pprefix[j] >> nsh) + o;
  uint32_t lo = 0, hi = m;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if ((akey[mid] >> nsh) < target) lo = mid + 1; else hi = mid;
  }
  cb[j * 9u + o] = lo;
}
struct ChildExistsF {
  const uint32_t* cb;
  uint32_t parents, m;
  uint32_t* counters;
  __device__ uint32_t operator()(uint32_t e) const {
    const uint32_t j = e >> 3, o = e & 7u;
    if (o == 0) {  // the parents' ranges must tile the active set: a point under none of them would belong to no node
      const uint32_t b = cb[j * 9u], en = cb[j * 9u + 8u];
      const bool ok = (j == 0 ? b == 0u : b == cb[(j - 1u) * 9u + 8u]) && (j + 1u < parents || en == m);
      if (!ok) atomicMax(&counters[CTR_ERROR], (uint32_t)SWZ_ERR_INTERNAL);
    }
    return cb[j * 9u + o + 1u] > cb[j * 9u + o] ? 1u : 0u;
  }
};
struct ChildStartG {
  const uint32_t* cb;
  uint32_t* nstart;
  __device__ void operator()(uint32_t e, uint32_t excl, uint32_t exists) const {
    if (!exists) return;
    const uint32_t j = e >> 3, o = e & 7u;
    nstart[excl] = cb[j * 9u + o];
    nstart[excl + 1u] = cb[j * 9u + o + 1u];  // (the next child writes the same value; the last one closes the table)
  }
};
constexpr uint32_t NF_TILE = 1024;  // points per workgroup: four per thread, one 16-byte store
// (lazy: the grid samplers look a point's node up only on levels that have take-all nodes -- the counters node_mode_kernel
// has just written say so --, and most levels of a large batch have none)
__global__ __launch_bounds__(256) void node_fill_kernel(const uint32_t* __restrict__ nstart, const uint32_t* __restrict__ counters,
                                                        uint32_t m, uint32_t* __restrict__ nid, int lazy) {
  __shared__ uint32_t s_lo, s_hi;
  __shared__ uint32_t ss[NF_TILE + 1];
  const uint32_t nn = counters[CTR_NUM_NODES];
  if (lazy && counters[CTR_SAMPLE_NODES] == nn) return;
  // (an inconsistent segmentation -- ChildExistsF raises CTR_ERROR -- must not be walked: the host reads the counter later)
  if (counters[CTR_ERROR] != 0u || nn == 0u) return;
  const uint32_t tid = threadIdx.x;
  const uint32_t i0 = blockIdx.x * NF_TILE;
  const uint32_t last = (m - i0) > NF_TILE ? i0 + NF_TILE - 1u : m - 1u;
  auto node_of = [&](uint32_t i) {  // the last node that starts at or before i
    uint32_t lo = 0, hi = nn;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (nstart[mid] <= i) lo = mid + 1; else hi = mid;
    }
    return lo - 1u;
  };
  if (tid < 2) {  // (two lanes of one wavefront: see tl_merge_rank_kernel, swz_tiler.hip)
    const uint32_t r = node_of(tid ? last : i0);
    if (tid) s_hi = r; else s_lo = r;
  }
  __syncthreads();
  const uint32_t lo = s_lo, span = min(s_hi - s_lo + 1u, (uint32_t)NF_TILE);  // (at most one node starts per point: span <= NF_TILE)
  for (uint32_t k = tid; k <= span; k += 256u) ss[k] = (lo + k < nn) ? nstart[lo + k] : 0xFFFFFFFFu;  // ss[k]: start of node lo + k
  __syncthreads();
  const uint32_t i = i0 + tid * 4u;
  if (i >= m) return;
  uint32_t a = 0, b = span;  // the last k < span with ss[k] <= i
  while (a < b) {
    const uint32_t mid = a + (b - a) / 2;
    if (ss[mid] <= i) a = mid + 1; else b = mid;
  }
  uint32_t k = a - 1u;
  uint32_t v[4];
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q) {
    while (k + 1u <= span && ss[k + 1u] <= i + q) ++k;
    v[q] = lo + k;
  }
  if (i + 4u <= m) {
    *reinterpret_cast<uint4*>(nid + i) = make_uint4(v[0], v[1], v[2], v[3]);
  } else {
    for (uint32_t q = 0; i + q < m; ++q) nid[i + q] = v[q];
  }
}
__global__ __launch_bounds__(256) void node_prefix_kernel(const uint32_t* __restrict__ nstart, const uint64_t* __restrict__ akey,
                                                          uint32_t nsh, uint32_t nn, uint64_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < nn) out[j] = nsh >= 63u ? 0ull : ((akey[nstart[j]] >> nsh) << nsh);
}
__global__ void single_node_kernel(uint32_t* __restrict__ nstart, uint32_t* __restrict__ num_nodes, uint32_t m) {
  nstart[0] = 0;
  nstart[1] = m;
  *num_nodes = 1;
}
// fused stable compaction: survivors move to the next level's active set, taken points get their level
struct KeepF {
  const uint8_t* taken;
  __device__ uint32_t operator()(uint32_t i) const { return taken[i] ? 0u : 1u; }
};
template <>
struct FsCountsZeroBytes<KeepF> {
  static constexpr bool value = true;
};
struct CompactG {
  const uint64_t* akey;
  const uint32_t* aidx;
  int8_t level;
  int8_t* level_out;
  uint64_t* okey;
  uint32_t* oidx;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t keep) const {
    const uint32_t p = aidx ? aidx[i] : i;
    if (keep) {
      okey[excl] = akey[i];
      oidx[excl] = p;
    } else {
      level_out[p] = level;
    }
  }
};

__global__ __launch_bounds__(256) void node_head_kernel(const uint64_t* __restrict__ akey, uint32_t m,
                                                        uint32_t nsh, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  flags[i] = (i == 0) ? 1u : (uint32_t)((akey[i] >> nsh) != (akey[i - 1] >> nsh));
}

__global__ __launch_bounds__(256) void node_finish_kernel(const uint32_t* __restrict__ flags,
                                                          uint32_t* __restrict__ nid, uint32_t m,
                                                          uint32_t* __restrict__ nstart) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t f = flags[i];
  const uint32_t id = nid[i] + f - 1u;  // exclusive count of heads before i -> node index of i
  nid[i] = id;
  if (f) nstart[id] = i;
  if (i == m - 1) nstart[id + 1] = m;
}

// tile_node / tile_internal_node decisions per node: terminal nodes and nodes with <= max_points
// points (SamplingBehaviour::TakeAllWhenCountBelowMaxPoints, Sampling.h:201-208) keep everything.
__global__ __launch_bounds__(256) void node_mode_kernel(const uint32_t* __restrict__ nstart,
                                                        uint8_t* __restrict__ nmode, uint32_t* __restrict__ counters,
                                                        uint64_t max_points, int force_sample, int terminal,
                                                        int reroot, const uint64_t* __restrict__ akey, uint32_t nsh,
                                                        const uint64_t* __restrict__ ckey, uint32_t nc) {
  // grid-stride over the nodes: their number is only known on the device, and it is small next to the points
  const uint32_t nnodes = counters[CTR_NUM_NODES];
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < nnodes; j += gridDim.x * 256u) {
    const uint32_t cnt = nstart[j + 1] - nstart[j];
    bool cached = false;  // previously_taken_points_count > 0 (TilingAlgorithms.cpp:272-275)
    if (nc) {
      const uint64_t prefix = akey[nstart[j]] >> nsh;
      uint32_t lo = 0, hi = nc;
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if ((ckey[mid] >> nsh) < prefix) lo = mid + 1; else hi = mid;
      }
      cached = lo < nc && (ckey[lo] >> nsh) == prefix;
    }
    const bool sample = !terminal && (force_sample || cached || (uint64_t)cnt > max_points);
    nmode[j] = sample ? MODE_SAMPLE : MODE_TAKE_ALL;
    if (sample) {
      if (reroot) atomicMax(&counters[CTR_ERROR], (uint32_t)SWZ_ERR_REROOT_UNSUPPORTED);
      atomicAdd(&counters[CTR_SAMPLE_NODES], 1u);
      atomicAdd(&counters[CTR_SAMPLE_POINTS], cnt);
    }
  }
}

// ----------------------------------------------------------------------------- RANDOM_GRID (K4a)
// RandomSortedGridSampling::sample_points, Sampling.h:187-308: the first point of every run of equal
// truncate_to_level(candidate_level) is taken.  candidate_level == -1 takes the first point only.
// (when every node of the level is sampled -- the counters of node_mode_kernel say so -- nobody looks at nid / nmode)
// Four consecutive points per thread: two 16-byte key loads and ONE 4-byte store of the four flags (a wavefront's byte
// stores fill 64 bytes of a line each).
constexpr uint32_t RG_IPT = 4;
__global__ __launch_bounds__(256) void random_grid_kernel(const uint64_t* __restrict__ akey, uint32_t m,
                                                          const uint32_t* __restrict__ nid,
                                                          const uint8_t* __restrict__ nmode, uint32_t csh,
                                                          uint8_t* __restrict__ taken, const uint32_t* __restrict__ counters) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * RG_IPT;
  if (i0 >= m) return;
  const bool all_sampled = counters[CTR_SAMPLE_NODES] == counters[CTR_NUM_NODES];
  if (i0 + RG_IPT <= m) {
    const ulonglong2 ka = *reinterpret_cast<const ulonglong2*>(akey + i0);
    const ulonglong2 kb = *reinterpret_cast<const ulonglong2*>(akey + i0 + 2);
    const uint64_t prev = i0 ? akey[i0 - 1] : 0ull;
    const uint64_t k[RG_IPT + 1] = {prev >> csh, ka.x >> csh, ka.y >> csh, kb.x >> csh, kb.y >> csh};
    uint32_t packed = 0;
#pragma unroll
    for (uint32_t j = 0; j < RG_IPT; ++j) {
      uint32_t t = 1;
      if (all_sampled || nmode[nid[i0 + j]] == MODE_SAMPLE) t = (i0 + j == 0) || (k[j + 1] != k[j]);
      packed |= t << (8u * j);
    }
    *reinterpret_cast<uint32_t*>(taken + i0) = packed;
    return;
  }
  for (uint64_t i = i0; i < m; ++i) {  // the last thread's partial group
    uint8_t t = 1;
    if (all_sampled || nmode[nid[i]] == MODE_SAMPLE) t = (i == 0) || ((akey[i] >> csh) != (akey[i - 1] >> csh));
    taken[i] = t;
  }
}

__global__ __launch_bounds__(256) void take_all_kernel(uint32_t m, const uint32_t* __restrict__ nid,
                                                       const uint8_t* __restrict__ nmode,
                                                       uint8_t* __restrict__ taken) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  if (nmode[nid[i]] == MODE_TAKE_ALL) taken[i] = 1;
}

// ----------------------------------------------------------------------------- GRID_CENTER / JITTERED (K4b, K4d)
// Both pick, per run of equal grid-cell prefix, the first point with the smallest squared distance
// to a per-cell target (std::min_element, Sampling.h:392-403 / :741-750): a segmented arg-min.
#ifndef SWZ_GA_THREADS
#define SWZ_GA_THREADS 256
#endif
#ifndef SWZ_GA_IPT
#define SWZ_GA_IPT 2
#endif
constexpr int GA_THREADS = SWZ_GA_THREADS;
constexpr int GA_IPT = SWZ_GA_IPT;
constexpr int GA_TILE = GA_THREADS * GA_IPT;
// the kernel that decides on keys (grid_argmin_keys_kernel) takes four points per thread: its loads are the keys alone, and
// the segmented scan across the lanes -- a third of its instructions -- is paid per thread (measured at 1 B points,
// GRID_CENTER / JITTERED sampling per step: 26.0 / 27.9 ms with two, 24.0 / 23.5 ms with four)
#ifndef SWZ_GAK_IPT
#define SWZ_GAK_IPT 4
#endif
constexpr int GAK_IPT = SWZ_GAK_IPT;
constexpr int GAK_TILE = GA_THREADS * GAK_IPT;
constexpr uint32_t NONE = 0xFFFFFFFFu;

struct Agg {
  double d;    // smallest squared distance since the last run start (or since the range began)
  uint32_t i;  // active index of the first point attaining it
  uint32_t f;  // 1 when a run start lies inside the covered range
};
__device__ __forceinline__ bool agg_less(double d1, uint32_t i1, double d2, uint32_t i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}
__device__ __forceinline__ Agg agg_combine(Agg a, Agg b) {  // a covers earlier points than b
  const bool take_b = b.f != 0 || agg_less(b.d, b.i, a.d, a.i);  // selects only: no branches, nothing on the stack
  Agg r;
  r.d = take_b ? b.d : a.d;
  r.i = take_b ? b.i : a.i;
  r.f = a.f | b.f;
  return r;
}
__device__ __forceinline__ Agg agg_shfl_up(Agg a, int delta) {
  Agg r;
  r.d = __shfl_up(a.d, delta, WAVE);
  r.i = __shfl_up(a.i, delta, WAVE);
  r.f = __shfl_up(a.f, delta, WAVE);
  return r;
}
// One step of the wave's inclusive scan over DPP: the aggregate of the lanes the control word names (the identity
// where it names none) combined in front of the lane's own.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Agg agg_dpp_step(Agg v) {
  const uint64_t db = (uint64_t)__double_as_longlong(v.d);
  const uint64_t inf = 0x7FF0000000000000ull;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)inf, (int)(uint32_t)db, CTRL, ROW_MASK, 0xF, false);
  const uint32_t hi =
    (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(inf >> 32), (int)(uint32_t)(db >> 32), CTRL, ROW_MASK, 0xF, false);
  Agg o;
  o.d = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
  o.i = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v.i, CTRL, ROW_MASK, 0xF, false);
  o.f = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.f, CTRL, ROW_MASK, 0xF, false);
  return agg_combine(o, v);
}
__device__ __forceinline__ Agg agg_wave_incl_scan(Agg v) {
  v = agg_dpp_step<0x111, 0xF>(v);  // row_shr:1
  v = agg_dpp_step<0x112, 0xF>(v);  // row_shr:2
  v = agg_dpp_step<0x114, 0xF>(v);  // row_shr:4
  v = agg_dpp_step<0x118, 0xF>(v);  // row_shr:8
  v = agg_dpp_step<0x142, 0xA>(v);  // row_bcast:15 -> rows 1, 3
  v = agg_dpp_step<0x143, 0xC>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

struct TileSummary {
  double head_d;  // leading partial run (continues a run of the previous tile), if the first point is no start
  double tail_d;  // trailing run that starts in this tile and continues into the next one
  uint32_t head_i;
  uint32_t tail_i;
  uint32_t has_start;  // some run starts inside this tile
  uint32_t last_open;  // the last run continues into the next tile
};

struct GridParams {
  Box root;
  int level;              // node level
  int sampler;            // SWZ_GRID_CENTER or SWZ_JITTERED
  int cand;               // GRID_CENTER candidate level (>= 0 here)
  double spacing_node;    // JITTERED
  uint32_t jitter_start;  // JITTERED
  // bounds of every octree cell at depth table_depth, indexed by the key's first table_depth octants (0: no table).
  // The points of a cell all walk the same halving chain; its first table_depth steps are looked up instead (the
  // table is small enough to stay in the caches, and neighbouring lanes read the same entry).
  const Box* box_table;
  int table_depth;
  const struct JitNode* jit_table;  // JITTERED: what the sampler derives from a node's bounds, per node prefix (or null)
};
// JitteredSampling's per-node quantities (Sampling.h:621-668): every point of a node derives the same ones
struct alignas(16) JitNode {
  double minx, miny, minz;  // the node's bounds_from_key minimum
  double cell_size, perm_size;
  uint32_t cells, levels;
  int32_t err;  // SWZ_ERR_JITTER_* or 0
  uint32_t pad;
};
constexpr int GRID_TABLE_MAX_DEPTH = 6;  // 8^6 boxes of 48 bytes = 12.6 MB (deeper tables were measured: no faster)
__global__ __launch_bounds__(256) void grid_box_table_kernel(Box root, int depth, Box* __restrict__ table) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (1u << (3 * depth))) return;
  table[t] = bounds_from_key((uint64_t)t << level_shift(depth - 1), root, depth);
}

// get_prev_power_of_two -- core/util/stuff.cpp:340-349
__device__ __forceinline__ uint32_t prev_pow2(uint32_t x) {
  x = x | (x >> 1);
  x = x | (x >> 2);
  x = x | (x >> 4);
  x = x | (x >> 8);
  x = x | (x >> 16);
  return x - (x >> 1);
}

// Depth of the bounds chain a point's target starts from: the candidate cell (GRID_CENTER) or the node (JITTERED).
__device__ __forceinline__ int cell_box_depth(const GridParams& g) {
  return g.sampler == SWZ_GRID_CENTER ? g.cand + 1 : g.level + 1;
}
// GridCenterSampling, Sampling.h:387-390: centre of kb = get_bounds_from_morton_index(key, root, cand + 1)
__device__ __forceinline__ void grid_center_target(const Box& kb, double& tx, double& ty, double& tz) {
  tx = kb.minx + (kb.maxx - kb.minx) / 2;  // AABB::getCenter, AABB.h:70
  ty = kb.miny + (kb.maxy - kb.miny) / 2;
  tz = kb.minz + (kb.maxz - kb.minz) / 2;
}
// JitteredSampling, Sampling.h:621-668: the grid of a node with bounds nb = bounds_from_key(key, root, level + 1)
__device__ __forceinline__ JitNode jitter_node(const Box& nb, double spacing_node, int level) {
  JitNode n;
  n.minx = nb.minx;
  n.miny = nb.miny;
  n.minz = nb.minz;
  n.pad = 0;
  n.err = 0;
  const double ext_x = nb.maxx - nb.minx;
  const double perfect = ext_x / spacing_node;
  const uint32_t perfect_u = perfect >= 4294967295.0 ? 4294967295u : (uint32_t)perfect;
  n.cells = prev_pow2(perfect_u);
  n.levels = n.cells ? 31u - (uint32_t)__clz((int)n.cells) : 0u;  // (uint32_t)std::log2(power of two)
  if (n.cells < 16) n.err = SWZ_ERR_JITTER_GRID_TOO_SMALL;
  else if ((uint32_t)level + n.levels >= MAX_LEVELS) n.err = SWZ_ERR_JITTER_NODE_TOO_DEEP;
  // ext_x / cells and cell_size / cells: cells = 2^levels, so the quotients are the scaled operands (ldexp rounds a
  // result that underflows once, like the division)
  n.cell_size = ldexp(ext_x, -(int)n.levels);
  n.perm_size = ldexp(n.cell_size, -(int)n.levels);
  return n;
}
// Sampling.h:669-739: cell prefix shift and jittered target of the grid cell `key` falls in (n.err == 0)
__device__ __forceinline__ void jitter_target(const GridParams& g, uint64_t key, const JitNode& n, uint32_t& csh, double& tx,
                                              double& ty, double& tz) {
  const uint32_t cells = n.cells, levels = n.levels;
  csh = level_shift((int)((uint32_t)g.level + levels));
  const uint64_t rel = (key >> csh) & ((1ull << (3u * levels)) - 1ull);
  const uint64_t mask = (1ull << levels) - 1ull;
  uint32_t gx, gy, gz;  // OctreeNodeIndex64::to_grid_index, OctreeNodeIndex.h:357-363 (below 2^levels <= 2^20)
  if (levels <= 10u) {  // the usual case (grids up to 1024 cells a side): rel has at most 30 bits, half the instructions
    const uint32_t r = (uint32_t)rel, m32 = (uint32_t)mask;
    gz = contract_bits_by_3_u32(r) & m32;
    gy = contract_bits_by_3_u32(r >> 1) & m32;
    gx = contract_bits_by_3_u32(r >> 2) & m32;
  } else {
    gz = (uint32_t)(contract_bits_by_3(rel) & mask);
    gy = (uint32_t)(contract_bits_by_3(rel >> 1) & mask);
    gx = (uint32_t)(contract_bits_by_3(rel >> 2) & mask);
  }
  const uint8_t* table;
  uint32_t width;
  if (cells <= 16) {
    table = PERMUTATIONS_16;
    width = 16;
  } else if (cells <= 32) {
    table = PERMUTATIONS_32;
    width = 32;
  } else {
    table = PERMUTATIONS_64;
    width = 64;
  }
  // length of the permutation in use: min(cells, 64), a power of two like cells -- "% plen" is a mask (the 64-bit
  // remainder the expression would otherwise compile to costs more than the rest of the function).  (The three rows in
  // use copied to LDS instead of three dependent byte loads from memory: measured, no faster.)
  const uint32_t plen_mask = (cells < 64 ? cells : 64) - 1u;
  const uint32_t s0 = g.jitter_start, s1 = (g.jitter_start + 1) % 16, s2 = (g.jitter_start + 2) % 16;
  const uint32_t px = (uint32_t)table[s0 * width + ((gy + gz) & plen_mask)] - 1u;
  const uint32_t py = (uint32_t)table[s1 * width + ((gx + gz) & plen_mask)] - 1u;
  const uint32_t pz = (uint32_t)table[s2 * width + ((gx + gy) & plen_mask)] - 1u;
  tx = n.minx + ((double)gx * n.cell_size + (double)px * n.perm_size);
  ty = n.miny + ((double)gy * n.cell_size + (double)py * n.perm_size);
  tz = n.minz + ((double)gz * n.cell_size + (double)pz * n.perm_size);
}
__global__ __launch_bounds__(256) void jitter_node_table_kernel(Box root, int level, double spacing_node,
                                                                JitNode* __restrict__ table) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (1u << (3 * (level + 1)))) return;
  const Box nb = level < 0 ? root : bounds_from_key((uint64_t)t << level_shift(level), root, level + 1);
  table[t] = jitter_node(nb, spacing_node, level);
}

#ifndef SWZ_GA_MINW
#define SWZ_GA_MINW 1
#endif
__global__ __launch_bounds__(GA_THREADS, SWZ_GA_MINW) void grid_argmin_kernel(
  const uint64_t* __restrict__ akey, const uint32_t* __restrict__ aidx, uint32_t m, const uint32_t* __restrict__ nid,
  const uint8_t* __restrict__ nmode, const double* __restrict__ X, const double* __restrict__ Y,
  const double* __restrict__ Z, GridParams g, uint32_t node_shift, uint8_t* __restrict__ taken,
  TileSummary* __restrict__ summaries, uint32_t* __restrict__ counters) {
  __shared__ Agg wave_tot[GA_THREADS / WAVE];
  const uint32_t tid = threadIdx.x, w = tid / WAVE, l = lane_id();
  const uint32_t tile_base = blockIdx.x * GA_TILE;
  const uint32_t tile_end = (m - tile_base) < (uint32_t)GA_TILE ? m : tile_base + GA_TILE;
  const uint32_t last_valid = tile_end - 1;
  const uint32_t first = tile_base + tid * GA_IPT;
  const bool all_sampled = counters[CTR_SAMPLE_NODES] == counters[CTR_NUM_NODES];  // then nobody looks at nid / nmode

  // Every load an item needs is issued before the arithmetic starts: the bounds chain below is a loop of dependent
  // f64 operations, and the loads of the next item must not queue up behind it.
  uint64_t key[GA_IPT];
  uint32_t spos[GA_IPT];
  bool sample[GA_IPT];
  double px[GA_IPT], py[GA_IPT], pz[GA_IPT];
  uint64_t prev_key = 0;
  bool have_prev = false;
  if (first < tile_end && first > 0) {
    prev_key = akey[first - 1];
    have_prev = true;
  }
  if (GA_IPT == 2 && first + 2 <= tile_end) {  // the usual case: two-item vector loads
    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(akey + first);
    key[0] = k2.x;
    key[GA_IPT - 1] = k2.y;
    if (aidx) {
      const uint2 p2 = *reinterpret_cast<const uint2*>(aidx + first);
      spos[0] = p2.x;
      spos[GA_IPT - 1] = p2.y;
    } else {
      spos[0] = first;
      spos[GA_IPT - 1] = first + 1;
    }
  } else {
#pragma unroll
    for (int j = 0; j < GA_IPT; ++j) {
      const uint32_t gc = first + j < tile_end ? first + j : last_valid;
      key[j] = akey[gc];
      spos[j] = spos_of(aidx, gc);
    }
  }
#pragma unroll
  for (int j = 0; j < GA_IPT; ++j) {
    const uint32_t gc = first + j < tile_end ? first + j : last_valid;
    sample[j] = all_sampled || nmode[nid[gc]] == MODE_SAMPLE;
  }
  if (GA_IPT == 2 && first + 2 <= tile_end && !aidx) {
    const double2 x2 = *reinterpret_cast<const double2*>(X + first);
    const double2 y2 = *reinterpret_cast<const double2*>(Y + first);
    const double2 z2 = *reinterpret_cast<const double2*>(Z + first);
    px[0] = x2.x, px[GA_IPT - 1] = x2.y;
    py[0] = y2.x, py[GA_IPT - 1] = y2.y;
    pz[0] = z2.x, pz[GA_IPT - 1] = z2.y;
  } else {
#pragma unroll
    for (int j = 0; j < GA_IPT; ++j) {
      px[j] = X[spos[j]];
      py[j] = Y[spos[j]];
      pz[j] = Z[spos[j]];
    }
  }
  Box kb[GA_IPT];
  JitNode jn[GA_IPT];
  if (g.jit_table) {  // JITTERED with a table: nothing of the node is computed here
    const uint32_t tsh = g.level < 0 ? 63u : level_shift(g.level);
#pragma unroll
    for (int j = 0; j < GA_IPT; ++j) jn[j] = g.jit_table[key[j] >> tsh];
  } else {
    if (g.table_depth > 0) {
      const uint32_t tsh = level_shift(g.table_depth - 1);
#pragma unroll
      for (int j = 0; j < GA_IPT; ++j) kb[j] = g.box_table[key[j] >> tsh];
    } else {
#pragma unroll
      for (int j = 0; j < GA_IPT; ++j) kb[j] = g.root;
    }
    bounds_from_keys<GA_IPT>(key, g.table_depth, cell_box_depth(g), kb);
    if (g.sampler != SWZ_GRID_CENTER) {
#pragma unroll
      for (int j = 0; j < GA_IPT; ++j) jn[j] = jitter_node(kb[j], g.spacing_node, g.level);
    }
  }

  double dist[GA_IPT];
  bool head[GA_IPT];
  uint32_t last_csh = node_shift;  // shift of the last valid item (for the last_open test)
  uint64_t last_key = 0;
  bool any_head = false;
#pragma unroll
  for (int j = 0; j < GA_IPT; ++j) {
    const uint32_t gi = first + j;
    dist[j] = __builtin_inf();
    head[j] = false;
    if (gi < tile_end) {
      uint32_t csh = node_shift;
      if (sample[j]) {
        double tx = 0, ty = 0, tz = 0;
        int err = 0;
        if (g.sampler == SWZ_GRID_CENTER) {
          csh = level_shift(g.cand);
          grid_center_target(kb[j], tx, ty, tz);
        } else {
          err = jn[j].err;
          if (!err) jitter_target(g, key[j], jn[j], csh, tx, ty, tz);
        }
        if (err) {
          atomicMax(&counters[CTR_ERROR], (uint32_t)err);
          csh = node_shift;
        } else {
          dist[j] = sq_dist(px[j], py[j], pz[j], tx, ty, tz);
        }
      } else {
        taken[gi] = 1;  // take-all node
      }
      head[j] = !have_prev || ((key[j] >> csh) != (prev_key >> csh));
      any_head |= head[j];
      prev_key = key[j];
      have_prev = true;
      last_csh = csh;
      last_key = key[j];
    }
  }

  // thread aggregate over its items, then block-wide exclusive segmented scan
  Agg a{__builtin_inf(), NONE, 0};
#pragma unroll
  for (int j = 0; j < GA_IPT; ++j) {
    const uint32_t gi = first + j;
    if (gi < tile_end) {
      if (head[j]) {
        a.d = dist[j];
        a.i = gi;
        a.f = 1;
      } else if (agg_less(dist[j], gi, a.d, a.i)) {
        a.d = dist[j];
        a.i = gi;
      }
    }
  }
  const Agg incl = agg_wave_incl_scan(a);
  if (l == WAVE - 1) wave_tot[w] = incl;
  const Agg up = agg_shfl_up(incl, 1);
  Agg excl;
  excl.d = l == 0 ? __builtin_inf() : up.d;
  excl.i = l == 0 ? NONE : up.i;
  excl.f = l == 0 ? 0u : up.f;
  const int tile_has_start = __syncthreads_or(any_head ? 1 : 0);
  Agg carry{__builtin_inf(), NONE, 0};
#pragma unroll
  for (uint32_t i = 0; i + 1 < (uint32_t)(GA_THREADS / WAVE); ++i) {
    const Agg t = wave_tot[i];
    const Agg cc = agg_combine(carry, t);
    carry.d = i < w ? cc.d : carry.d;
    carry.i = i < w ? cc.i : carry.i;
    carry.f = i < w ? cc.f : carry.f;
  }
  carry = agg_combine(carry, excl);

  // second pass: close runs, emit winners / partial aggregates
  bool started = carry.f != 0;
  double rd = carry.d;
  uint32_t ri = carry.i;
  TileSummary* sum = &summaries[blockIdx.x];
#pragma unroll
  for (int j = 0; j < GA_IPT; ++j) {
    const uint32_t gi = first + j;
    if (gi < tile_end) {
      if (head[j]) {
        if (gi != tile_base) {  // the run ending at gi-1 closes inside this tile
          if (started) {
            if (ri != NONE) taken[ri] = 1;
          } else {
            sum->head_d = rd;
            sum->head_i = ri;
          }
        }
        rd = dist[j];
        ri = gi;
        started = true;
      } else if (agg_less(dist[j], gi, rd, ri)) {
        rd = dist[j];
        ri = gi;
      }
      if (gi == last_valid) {
        const bool last_open = (tile_end < m) && ((akey[tile_end] >> last_csh) == (last_key >> last_csh));
        if (!last_open) {
          if (started) {
            if (ri != NONE) taken[ri] = 1;
          } else {
            sum->head_d = rd;
            sum->head_i = ri;
          }
        } else if (started) {
          sum->tail_d = rd;
          sum->tail_i = ri;
        } else {
          sum->head_d = rd;
          sum->head_i = ri;
        }
        sum->has_start = (uint32_t)tile_has_start;
        sum->last_open = last_open ? 1u : 0u;
      }
    }
  }
}

// runs that cross tile borders: the thread of the tile in which the run starts walks forward
__global__ __launch_bounds__(256) void grid_resolve_kernel(const TileSummary* __restrict__ summaries,
                                                           uint32_t ntiles, uint8_t* __restrict__ taken) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  const TileSummary s = summaries[t];
  if (!(s.has_start && s.last_open)) return;
  double d = s.tail_d;
  uint32_t i = s.tail_i;
  for (uint32_t u = t + 1; u < ntiles; ++u) {
    const TileSummary h = summaries[u];
    if (agg_less(h.head_d, h.head_i, d, i)) {
      d = h.head_d;
      i = h.head_i;
    }
    if (h.has_start || !h.last_open) break;
  }
  if (i != NONE) taken[i] = 1;
}

// ----------------------------------------------------------------------------- GRID_CENTER / JITTERED on key coordinates
// The arg-min above reads every point's position (24 bytes, in Morton order: a gather of the whole batch after the sort).
// But the Morton key IS the position, quantised to 2^-21 of the bounds per axis (calculate_morton_index,
// OctreeAlgorithms.h:64-87): a point with key coordinate i lies in [i, i + 1] key cells, so its distance to a target is
// known to +- half a cell per axis from the key alone.  Per grid cell the kernel below keeps the point with the smallest
// UPPER bound of that distance, that point's lower bound, and the smallest lower bound among all the others: when even
// that exceeds the leader's upper bound the leader is the arg-min whatever the exact positions are (and the first one:
// equal distances would overlap).  Otherwise -- two points whose distances to the target differ by less than the
// quantisation -- the run goes on a list and a second kernel repeats it with the reference's own arithmetic (target from
// the halving chain of the bounds, sq_dist in double on the ORIGINAL positions, read through the permutation;
// Sampling.h:387-403 / :741-750).  No position is moved; runs of one point (most runs of the deeper levels) never need it.
// hk = 0.5 + slack: the slack covers the rounding of the encoder's (p - min) * scale (1e-9 cells) and the difference
// between the ideal target and the reference's, computed from bounds that went through up to 21 halvings (make_grid_keys).
struct KAgg {
  float ub, lb;    // leader: upper / lower bound of its squared distance (in units of the widest key cell, squared)
  float m2;        // smallest lower bound among the run's other points
  uint32_t i;      // leader (first one with the smallest upper bound)
  uint32_t start;  // the run's first point, NONE when it lies before the covered range
  uint32_t f;      // 1 when a run start lies inside the covered range
};
__device__ __forceinline__ KAgg kagg_combine(KAgg a, KAgg b) {  // a covers earlier points than b; selects only
  const bool bwin = b.ub < a.ub || (b.ub == a.ub && b.i < a.i);
  const bool bf = b.f != 0;
  const float l_lb = bwin ? a.lb : b.lb;
  KAgg r;
  r.ub = (bf || bwin) ? b.ub : a.ub;
  r.lb = (bf || bwin) ? b.lb : a.lb;
  r.i = (bf || bwin) ? b.i : a.i;
  r.m2 = bf ? b.m2 : fminf(fminf(a.m2, b.m2), l_lb);
  r.start = bf ? b.start : a.start;
  r.f = a.f | b.f;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ KAgg kagg_dpp_step(KAgg v) {
  const int inf = 0x7F800000;
  KAgg o;
  o.ub = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v.ub), CTRL, ROW_MASK, 0xF, false));
  o.lb = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v.lb), CTRL, ROW_MASK, 0xF, false));
  o.m2 = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(v.m2), CTRL, ROW_MASK, 0xF, false));
  o.i = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v.i, CTRL, ROW_MASK, 0xF, false);
  o.start = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v.start, CTRL, ROW_MASK, 0xF, false);
  o.f = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.f, CTRL, ROW_MASK, 0xF, false);
  return kagg_combine(o, v);
}
__device__ __forceinline__ KAgg kagg_wave_incl_scan(KAgg v) {
  v = kagg_dpp_step<0x111, 0xF>(v);
  v = kagg_dpp_step<0x112, 0xF>(v);
  v = kagg_dpp_step<0x114, 0xF>(v);
  v = kagg_dpp_step<0x118, 0xF>(v);
  v = kagg_dpp_step<0x142, 0xA>(v);
  v = kagg_dpp_step<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ KAgg kagg_identity() { return KAgg{__builtin_inff(), __builtin_inff(), __builtin_inff(), NONE, NONE, 0u}; }

struct KTileSummary {
  KAgg head;          // leading partial run (continues a run of the previous tile), if the first point is no start
  KAgg tail;          // trailing run that starts in this tile and continues into the next one (tail.start: where)
  uint32_t head_end;  // where the leading partial run ends: the tile's first run start, or the tile's end
  uint32_t has_start;
  uint32_t last_open;
  uint32_t pad;
};

struct GridKeys {
  float w[3];   // key cell width per axis relative to the widest one
  double hk;    // half a key cell plus the slack (see above)
  // the same for the kernel's single-precision bounds, rounded to the safe side: hk up, the widths down for the lower and
  // up for the upper bound (grid_argmin_keys_kernel)
  float hk_f, w_lo[3], w_hi[3];
  uint2* amb;   // runs the keys cannot decide: {first, end} active index
  uint32_t* amb_count;
};

// a run is closed: the leader is taken, or the run goes to the exact pass
__device__ __forceinline__ void kagg_close(const KAgg& r, uint32_t end, const GridKeys& gk, uint8_t* __restrict__ taken) {
  if (r.i == NONE) return;
  if (r.m2 <= r.ub && r.ub < __builtin_inff()) {
    const uint32_t at = atomicAdd(gk.amb_count, 1u);
    gk.amb[at] = make_uint2(r.start, end);
  } else {
    taken[r.i] = 1;
  }
}

__global__ __launch_bounds__(GA_THREADS) void grid_argmin_keys_kernel(
  const uint64_t* __restrict__ akey, uint32_t m, const uint32_t* __restrict__ nid, const uint8_t* __restrict__ nmode, GridParams g,
  GridKeys gk, uint32_t node_shift, uint8_t* __restrict__ taken, KTileSummary* __restrict__ summaries, uint32_t* __restrict__ counters) {
  __shared__ KAgg wave_tot[GA_THREADS / WAVE];
  const uint32_t tid = threadIdx.x, w = tid / WAVE, l = lane_id();
  const uint32_t tile_base = blockIdx.x * GAK_TILE;
  const uint32_t tile_end = (m - tile_base) < (uint32_t)GAK_TILE ? m : tile_base + GAK_TILE;
  const uint32_t last_valid = tile_end - 1;
  const uint32_t first = tile_base + tid * GAK_IPT;
  const bool all_sampled = counters[CTR_SAMPLE_NODES] == counters[CTR_NUM_NODES];

  uint64_t key[GAK_IPT];
  bool sample[GAK_IPT];
  uint64_t prev_key = 0;
  bool have_prev = false;
  if (first < tile_end && first > 0) {
    prev_key = akey[first - 1];
    have_prev = true;
  }
#pragma unroll
  for (int j = 0; j < GAK_IPT; ++j) {
    const uint32_t gc = first + j < tile_end ? first + j : last_valid;
    key[j] = akey[gc];
    sample[j] = all_sampled || nmode[nid[gc]] == MODE_SAMPLE;
  }
  JitNode jn[GAK_IPT];
  if (g.sampler != SWZ_GRID_CENTER) {  // what JITTERED derives from the node's box: grid size, levels, error (as above)
    if (g.jit_table) {
      const uint32_t tsh = g.level < 0 ? 63u : level_shift(g.level);
#pragma unroll
      for (int j = 0; j < GAK_IPT; ++j) jn[j] = g.jit_table[key[j] >> tsh];
    } else {
      Box kb[GAK_IPT];
      if (g.table_depth > 0) {
        const uint32_t tsh = level_shift(g.table_depth - 1);
#pragma unroll
        for (int j = 0; j < GAK_IPT; ++j) kb[j] = g.box_table[key[j] >> tsh];
      } else {
#pragma unroll
        for (int j = 0; j < GAK_IPT; ++j) kb[j] = g.root;
      }
      bounds_from_keys<GAK_IPT>(key, g.table_depth, g.level + 1, kb);
#pragma unroll
      for (int j = 0; j < GAK_IPT; ++j) jn[j] = jitter_node(kb[j], g.spacing_node, g.level);
    }
  }

  float ub[GAK_IPT], lb[GAK_IPT];
  bool head[GAK_IPT];
  uint32_t last_csh = node_shift;
  uint64_t last_key = 0;
  bool any_head = false;
#pragma unroll
  for (int j = 0; j < GAK_IPT; ++j) {
    const uint32_t gi = first + j;
    ub[j] = __builtin_inff();
    lb[j] = __builtin_inff();
    head[j] = false;
    if (gi < tile_end) {
      uint32_t csh = node_shift;
      if (sample[j]) {
        int err = 0;
        // Offset of the point's key cell centre from the target, per axis, in key cells -- in single precision, exactly:
        // a half-integer below 2^21 (GRID_CENTER), or a multiple of the permutation step 2^(sbits - levels) >= 2^-6 below
        // 2^sbits with levels <= 6 (JITTERED): at most 22 significant bits either way.
        float ox = 0.f, oy = 0.f, oz = 0.f;
        uint32_t ix, iy, iz;
        key_coords_u32(key[j], ix, iy, iz);
        if (g.sampler == SWZ_GRID_CENTER) {
          csh = level_shift(g.cand);
          const uint32_t sbits = csh / 3u, mask = (1u << sbits) - 1u;
          const float half = ldexpf(1.0f, (int)sbits - 1);  // (0.5 for a cell one key cell wide)
          ox = (float)(ix & mask) + 0.5f - half;
          oy = (float)(iy & mask) + 0.5f - half;
          oz = (float)(iz & mask) + 0.5f - half;
        } else {
          err = jn[j].err;
          if (!err) {
            const uint32_t levels = jn[j].levels, cells = jn[j].cells;
            csh = level_shift((int)((uint32_t)g.level + levels));
            const uint32_t sbits = csh / 3u, mask = (1u << sbits) - 1u, gmask = cells - 1u;
            const uint32_t gx = (ix >> sbits) & gmask, gy = (iy >> sbits) & gmask, gz = (iz >> sbits) & gmask;  // to_grid_index
            const uint8_t* table;
            uint32_t width;
            if (cells <= 16) {
              table = PERMUTATIONS_16;
              width = 16;
            } else if (cells <= 32) {
              table = PERMUTATIONS_32;
              width = 32;
            } else {
              table = PERMUTATIONS_64;
              width = 64;
            }
            const uint32_t plen_mask = (cells < 64 ? cells : 64) - 1u;
            const uint32_t s0 = g.jitter_start, s1 = (g.jitter_start + 1) % 16, s2 = (g.jitter_start + 2) % 16;
            const uint32_t px = (uint32_t)table[s0 * width + ((gy + gz) & plen_mask)] - 1u;
            const uint32_t py = (uint32_t)table[s1 * width + ((gx + gz) & plen_mask)] - 1u;
            const uint32_t pz = (uint32_t)table[s2 * width + ((gx + gy) & plen_mask)] - 1u;
            const float perm = ldexpf(1.0f, (int)sbits - (int)levels);  // perm_size = cell_size / cells, in key cells
            ox = (float)(ix & mask) + 0.5f - (float)px * perm;
            oy = (float)(iy & mask) + 0.5f - (float)py * perm;
            oz = (float)(iz & mask) + 0.5f - (float)pz * perm;
          }
        }
        if (err) {
          atomicMax(&counters[CTR_ERROR], (uint32_t)err);
          csh = node_shift;
        } else {
          // Bounds of the squared distance, rounded outwards.  hk_f >= hk and w_lo <= w <= w_hi are rounded to the safe side
          // already; what is left are the roundings of this arithmetic on non-negative terms -- the sum / difference with
          // hk_f, the product with the width, the square, two additions: five at 2^-24 relative each along any path --, which
          // the factors 1 -+ 2^-20 cover several times over.  (Until round 4 this ran in double: half the rate and twice the
          // registers for bounds that end up as floats.)
          const float ax = fabsf(ox), ay = fabsf(oy), az = fabsf(oz);
          const float lx = fmaxf(ax - gk.hk_f, 0.f) * gk.w_lo[0], ly = fmaxf(ay - gk.hk_f, 0.f) * gk.w_lo[1], lz = fmaxf(az - gk.hk_f, 0.f) * gk.w_lo[2];
          const float ux = (ax + gk.hk_f) * gk.w_hi[0], uy = (ay + gk.hk_f) * gk.w_hi[1], uz = (az + gk.hk_f) * gk.w_hi[2];
          lb[j] = (lx * lx + ly * ly + lz * lz) * (1.0f - 0x1.0p-20f);
          ub[j] = (ux * ux + uy * uy + uz * uz) * (1.0f + 0x1.0p-20f);
        }
      } else {
        taken[gi] = 1;  // take-all node
      }
      head[j] = !have_prev || ((key[j] >> csh) != (prev_key >> csh));
      any_head |= head[j];
      prev_key = key[j];
      have_prev = true;
      last_csh = csh;
      last_key = key[j];
    }
  }

  // thread aggregate over its items, then block-wide exclusive segmented scan
  KAgg a = kagg_identity();
#pragma unroll
  for (int j = 0; j < GAK_IPT; ++j) {
    const uint32_t gi = first + j;
    if (gi < tile_end) {
      KAgg it{ub[j], lb[j], __builtin_inff(), gi, head[j] ? gi : NONE, head[j] ? 1u : 0u};
      a = kagg_combine(a, it);
    }
  }
  const KAgg incl = kagg_wave_incl_scan(a);
  if (l == WAVE - 1) wave_tot[w] = incl;
  KAgg excl;
  excl.ub = __shfl_up(incl.ub, 1, WAVE);
  excl.lb = __shfl_up(incl.lb, 1, WAVE);
  excl.m2 = __shfl_up(incl.m2, 1, WAVE);
  excl.i = __shfl_up(incl.i, 1, WAVE);
  excl.start = __shfl_up(incl.start, 1, WAVE);
  excl.f = __shfl_up(incl.f, 1, WAVE);
  if (l == 0) excl = kagg_identity();
  const int tile_has_start = __syncthreads_or(any_head ? 1 : 0);
  KAgg carry = kagg_identity();
#pragma unroll
  for (uint32_t i = 0; i + 1 < (uint32_t)(GA_THREADS / WAVE); ++i) {
    const KAgg cc = kagg_combine(carry, wave_tot[i]);
    if (i < w) carry = cc;
  }
  carry = kagg_combine(carry, excl);

  // second pass: close runs, emit winners / undecided runs / partial aggregates
  KAgg run = carry;
  KTileSummary* sum = &summaries[blockIdx.x];
#pragma unroll
  for (int j = 0; j < GAK_IPT; ++j) {
    const uint32_t gi = first + j;
    if (gi < tile_end) {
      if (head[j] && gi != tile_base) {  // the run ending at gi - 1 closes inside this tile
        if (run.f) {
          kagg_close(run, gi, gk, taken);
        } else {
          sum->head = run;
          sum->head_end = gi;
        }
      }
      KAgg it{ub[j], lb[j], __builtin_inff(), gi, head[j] ? gi : NONE, head[j] ? 1u : 0u};
      run = kagg_combine(run, it);
      if (gi == last_valid) {
        const bool last_open = (tile_end < m) && ((akey[tile_end] >> last_csh) == (last_key >> last_csh));
        if (!last_open) {
          if (run.f) {
            kagg_close(run, tile_end, gk, taken);
          } else {
            sum->head = run;
            sum->head_end = tile_end;
          }
        } else if (run.f) {
          sum->tail = run;
        } else {
          sum->head = run;
          sum->head_end = tile_end;
        }
        sum->has_start = (uint32_t)tile_has_start;
        sum->last_open = last_open ? 1u : 0u;
      }
    }
  }
}

// runs that cross tile borders: the thread of the tile in which the run starts walks forward
__global__ __launch_bounds__(256) void grid_resolve_keys_kernel(const KTileSummary* __restrict__ summaries, uint32_t ntiles, GridKeys gk,
                                                                uint8_t* __restrict__ taken) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  if (!(summaries[t].has_start && summaries[t].last_open)) return;
  KAgg run = summaries[t].tail;
  uint32_t end = 0;
  for (uint32_t u = t + 1; u < ntiles; ++u) {
    const KAgg h = summaries[u].head;  // (no run start inside: it continues this run)
    run = kagg_combine(run, h);
    end = summaries[u].head_end;
    if (summaries[u].has_start || !summaries[u].last_open) break;
  }
  kagg_close(run, end, gk, taken);
}

// the runs the keys could not decide, one wavefront each, with the reference's arithmetic on the original positions
__global__ __launch_bounds__(256) void grid_exact_runs_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ aidx,
                                                              SortedPoints sp, GridParams g, GridKeys gk, uint8_t* __restrict__ taken) {
  const uint32_t l = lane_id();
  const uint32_t nruns = *gk.amb_count;
  for (uint32_t r = blockIdx.x * (256u / WAVE) + threadIdx.x / WAVE; r < nruns; r += gridDim.x * (256u / WAVE)) {
    const uint2 se = gk.amb[r];
    double best = __builtin_inf();
    uint32_t besti = NONE;
    for (uint32_t i = se.x + l; i < se.y; i += WAVE) {
      const uint64_t key = akey[i];
      const double* pp = sorted_point_xyz(sp.xyz, sp.perm, sp.ghost_xyz, sp.ghosts, aidx ? aidx[i] : i);
      const double px = pp[0], py = pp[1], pz = pp[2];
      const Box kb = bounds_from_key(key, g.root, cell_box_depth(g));
      double tx = 0, ty = 0, tz = 0;
      if (g.sampler == SWZ_GRID_CENTER) {
        grid_center_target(kb, tx, ty, tz);
      } else {
        const JitNode n = jitter_node(kb, g.spacing_node, g.level);
        uint32_t csh;
        jitter_target(g, key, n, csh, tx, ty, tz);
      }
      const double d = sq_dist(px, py, pz, tx, ty, tz);
      if (agg_less(d, i, best, besti)) {
        best = d;
        besti = i;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = __shfl_xor(best, off, WAVE);
      const uint32_t oi = (uint32_t)__shfl_xor((int)besti, off, WAVE);
      if (agg_less(od, oi, best, besti)) {
        best = od;
        besti = oi;
      }
    }
    if (l == 0 && besti != NONE) taken[besti] = 1;
  }
}

// ----------------------------------------------------------------------------- compaction (K5)
__global__ __launch_bounds__(256) void keep_flags_kernel(const uint8_t* __restrict__ taken, uint32_t m,
                                                         uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) flags[i] = taken[i] ? 0u : 1u;
}

__global__ __launch_bounds__(256) void compact_kernel(const uint64_t* __restrict__ akey,
                                                      const uint32_t* __restrict__ aidx, uint32_t m,
                                                      const uint8_t* __restrict__ taken,
                                                      const uint32_t* __restrict__ pos, int8_t level,
                                                      int8_t* __restrict__ level_out, uint64_t* __restrict__ okey,
                                                      uint32_t* __restrict__ oidx) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t p = spos_of(aidx, i);
  if (taken[i]) {
    level_out[p] = level;
  } else {
    const uint32_t o = pos[i];
    okey[o] = akey[i];
    oidx[o] = p;
  }
}

// ----------------------------------------------------------------------------- host: level plans
// candidate_level_in_octree -- Sampling.h:210-229 (std::log2f on the double ratio narrowed to float)
static int candidate_level_host(double root_extent_x, float spacing_at_root, int node_level) {
  const auto spacing_at_this_node = spacing_at_root / std::pow(2, node_level + 1);
  return std::max(-1, (int)std::floor(std::log2f(root_extent_x / spacing_at_this_node)) - 1);
}
// get_node_level_to_sample_from / first_node_level_obeying_spacing -- core/tiling/Node.cpp:37-57
static int node_level_to_sample_from_host(double root_extent_x, float root_max_spacing, int node_level) {
  const auto spacing_at_target_node = root_max_spacing / std::pow(2, node_level + 1);
  const float target_spacing = (float)spacing_at_target_node;
  return std::max(-1, (int)std::floor(std::log2f(root_extent_x / target_spacing)) - 1);
}
static uint32_t prev_pow2_host(uint32_t x) {
  x = x | (x >> 1);
  x = x | (x >> 2);
  x = x | (x >> 4);
  x = x | (x >> 8);
  x = x | (x >> 16);
  return x - (x >> 1);
}
// required_morton_index_depth -- core/tiling/Sampling.cpp:29-62
int required_depth_host(int sampler, int node_level, double root_extent_x, float root_max_spacing) {
  switch (sampler) {
    case SWZ_RANDOM_GRID:
    case SWZ_GRID_CENTER:
      return node_level_to_sample_from_host(root_extent_x, root_max_spacing, node_level);
    case SWZ_MIN_DISTANCE:
      return node_level;
    default: {
      const auto spacing_at_this_node = root_max_spacing / std::pow(2, node_level + 1);
      const auto perfect_cell_count = (root_extent_x / std::pow(2, node_level + 1)) / spacing_at_this_node;
      const double clamped = perfect_cell_count >= 4294967295.0 ? 4294967295.0 : perfect_cell_count;
      const auto actual_cell_count = prev_pow2_host(static_cast<uint32_t>(clamped));
      const uint32_t levels = static_cast<uint32_t>(std::log2(actual_cell_count));
      return static_cast<int32_t>(static_cast<uint32_t>(node_level + levels));
    }
  }
}

}  // namespace swz
extern "C" int32_t swz_required_morton_index_depth(int sampler, int32_t node_level, const double root_min[3],
                                                   const double root_max[3], float spacing_at_root) {
  if (!root_min || !root_max) return INT32_MIN;
  return swz::required_depth_host(sampler, node_level, root_max[0] - root_min[0], spacing_at_root);
}
namespace swz {

LevelPlan make_plan(int level, int sampler, uint64_t max_points, float spacing_at_root, uint32_t max_depth,
                           const double bmin[3], const double bmax[3], bool force_sample, bool tiler_rules) {
  LevelPlan p;
  p.level = level;
  p.node_shift = level < 0 ? 63u : level_shift(level);
  p.sampler = sampler;
  p.max_points = max_points;
  p.force_sample = force_sample;
  p.root = Box{bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2]};
  const double ext_x = bmax[0] - bmin[0];
  if (tiler_rules) {
    // tile_node, TilingAlgorithms.cpp:408-444
    const int req = required_depth_host(sampler, level, ext_x, spacing_at_root);
    const bool deeper = req > level;
    const int max_level = (int)std::min<uint32_t>(MAX_LEVELS - 1, max_depth);
    if (!deeper) {
      p.terminal = req >= max_level;
    } else {
      p.terminal = level >= max_level;
      p.reroot = !p.terminal && req >= (int)MAX_LEVELS;
    }
  }
  p.cand = candidate_level_host(ext_x, spacing_at_root, level);
  p.spacing_node = spacing_at_root / std::pow(2, level + 1);
  p.jitter_start = (3u * static_cast<uint32_t>(level + 1)) % 16u;
  const float sf = static_cast<float>(p.spacing_node);  // PoissonDiskSampling, Sampling.h:448-449
  const float sq = sf * sf;                             // SparseGrid::SparseGrid, SparseGrid.cpp:13
  p.sq_spacing = (double)sq;                            // widened at the compare, GridCell.cpp:44,52
  // finest subdivision of a node whose cells are still at least one spacing wide on every axis (with
  // a 2^-20 relative margin so that quantisation of the key never lets two points closer than the
  // spacing sit in non-adjacent cells); node extent / spacing is the same at every level
  double min_ext = std::min(bmax[0] - bmin[0], std::min(bmax[1] - bmin[1], bmax[2] - bmin[2]));
  const double need = (double)spacing_at_root * (1.0 + 0x1.0p-20);
  int mg = 0;
  while (mg < 20 && min_ext / 2 >= need) {
    min_ext /= 2;
    ++mg;
  }
  p.cell_levels_geo = std::min(mg, 20 - level);
  return p;
}

// GRID_CENTER / JITTERED: can this level be decided on key coordinates?  Needs the original positions and the
// permutation for the undecided runs; JITTERED additionally cubic bounds (its grid cells are cubes of the node's x-extent
// along every axis, Sampling.h:621-668: with other bounds its targets do not sit where the key cells put them).
// SWZ_GRID_KEYS=0 switches it off; SWZ_GRID_KEYS_SLACK adds to the slack (tests: a huge one sends every run of more than
// one point through the exact pass, a negative one must change results).
bool grid_level_uses_keys(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp, GridKeys* out) {
  if (plan.sampler != SWZ_GRID_CENTER && plan.sampler != SWZ_JITTERED) return false;
  if (!sp.xyz || !sp.perm) return false;
  if (const char* e = c->opt("SWZ_GRID_KEYS"))
    if (atoi(e) == 0) return false;
  const double ext[3] = {plan.root.maxx - plan.root.minx, plan.root.maxy - plan.root.miny, plan.root.maxz - plan.root.minz};
  if (!(ext[0] > 0.0) || !(ext[1] > 0.0) || !(ext[2] > 0.0)) return false;
  if (plan.sampler == SWZ_JITTERED && !(ext[0] == ext[1] && ext[1] == ext[2])) return false;
  const double wmax = std::max(ext[0], std::max(ext[1], ext[2])), wmin = std::min(ext[0], std::min(ext[1], ext[2]));
  const double max_abs = std::max(std::max(std::max(std::fabs(plan.root.minx), std::fabs(plan.root.maxx)),
                                           std::max(std::fabs(plan.root.miny), std::fabs(plan.root.maxy))),
                                  std::max(std::fabs(plan.root.minz), std::fabs(plan.root.maxz)));
  // The reference's target comes out of bounds that went through up to 21 halvings and a few more operations, each
  // rounding at the magnitude of the coordinates: 128 ulp of the largest one, in key cells of the narrowest axis; plus
  // the rounding of the encoder's (p - min) * scale.
  double slack = 1e-6 + 128.0 * 0x1.0p-52 * max_abs / (wmin / 2097152.0);
  if (const char* e = c->opt("SWZ_GRID_KEYS_SLACK")) slack += atof(e);
  if (!(slack < 0.25) && !c->opt("SWZ_GRID_KEYS_SLACK")) return false;  // bounds far from the origin relative to their size
  if (out) {
    for (int a = 0; a < 3; ++a) {
      out->w[a] = (float)(ext[a] / wmax);
      const double wd = ext[a] / wmax;
      float lo = (float)wd, hi = (float)wd;
      if ((double)lo > wd) lo = std::nextafterf(lo, 0.f);
      if ((double)hi < wd) hi = std::nextafterf(hi, INFINITY);
      out->w_lo[a] = lo;
      out->w_hi[a] = hi;
    }
    out->hk = 0.5 + slack;
    out->hk_f = (float)out->hk;
    if ((double)out->hk_f < out->hk) out->hk_f = std::nextafterf(out->hk_f, INFINITY);
    out->amb = nullptr;
    out->amb_count = nullptr;
  }
  return true;
}

bool level_decides_on_keys(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp) {
  if (plan.sampler == SWZ_RANDOM_GRID) return true;
  if (plan.sampler == SWZ_MIN_DISTANCE) return min_distance_level_uses_keys(c, plan, sp);
  return grid_level_uses_keys(c, plan, sp, nullptr);
}

// what a kernel of the level raised in CTR_ERROR, in words
static const char* level_error_message(int code) {
  switch (code) {
    case SWZ_ERR_JITTER_GRID_TOO_SMALL: return "Grids smaller than 16x16 are not supported currently!";
    case SWZ_ERR_JITTER_NODE_TOO_DEEP: return "node is too small to be sampled with JITTERED";
    case SWZ_ERR_REROOT_UNSUPPORTED: return "a node needs Morton re-rooting, which this call's per-point outputs cannot express: use swz_tile_nodes_begin_device / _end_device (one batch as node files) or a swz_tiler";
    case SWZ_ERR_INTERNAL: return "level segmentation inconsistent, or a MIN_DISTANCE sweep / a peer shard failed";
    default: return "a kernel of the level raised an error";
  }
}

// ----------------------------------------------------------------------------- one level
// Samples every node of the level.  When okey/oidx are given the survivors are compacted into them
// and level_out receives plan.level for the taken points; otherwise only lb.taken is produced.
int level_step(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                      const LevelBuffers& lb, int8_t* level_out, uint64_t* okey, uint32_t* oidx,
                      LevelResult* res) {
  const uint32_t m = as.m;
  const uint32_t nb = div_up(m, 256);
  SWZ_HIP(c, hipMemsetAsync(lb.counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
  {
    ProfScope ps(c, "level_nodes", (uint64_t)m * 8ull, 3);
    bool fill_after_modes = false;
    if (plan.node_shift >= 63u && m > 0) {
      // the root (Morton keys have 63 bits): one node, nothing to segment -- two passes over the keys saved
      // (node ids: zeros -- unless the one node is going to be sampled and the sampler is one that then never looks, see
      // node_fill_kernel)
      const bool sampled_for_sure = !plan.terminal && (plan.force_sample || (uint64_t)m > plan.max_points);
      if (plan.sampler == SWZ_MIN_DISTANCE || !sampled_for_sure)
        SWZ_HIP(c, hipMemsetAsync(lb.nid, 0, (size_t)m * sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(single_node_kernel, dim3(1), dim3(1), 0, c->stream, lb.nstart, lb.counters + CTR_NUM_NODES, m);
      SWZ_LAUNCH_CHECK(c);
    } else if (as.parent_prefix && as.parents && m > 0 && !c->opt("SWZ_LEVEL_NODES_SCAN")) {
      uint32_t* cb = nullptr;
      SWZ_TRY(c->get("lvl_child_bounds", (size_t)as.parents * 9u, &cb));
      hipLaunchKernelGGL(node_child_bounds_kernel, dim3(div_up(as.parents * 16u, 256u)), dim3(256), 0, c->stream, as.akey, m,
                         plan.node_shift, as.parent_prefix, as.parents, cb);
      SWZ_LAUNCH_CHECK(c);
      SWZ_TRY(fused_scan(c, ChildExistsF{cb, as.parents, m, lb.counters}, ChildStartG{cb, lb.nstart}, as.parents * 8u,
                         lb.counters + CTR_NUM_NODES, "lvl"));
      fill_after_modes = true;
    } else {
      SWZ_TRY(fused_scan(c, NodeHeadF{as.akey, plan.node_shift}, NodeAssignG{lb.nid, lb.nstart, m}, m,
                         lb.counters + CTR_NUM_NODES, "lvl"));
    }
    hipLaunchKernelGGL(node_mode_kernel, dim3(std::min(nb, 2048u)), dim3(256), 0, c->stream, lb.nstart, lb.nmode, lb.counters,
                       plan.max_points, plan.force_sample ? 1 : 0, plan.terminal ? 1 : 0, plan.reroot ? 1 : 0, as.akey,
                       plan.node_shift, as.ckey, as.nc);
    SWZ_LAUNCH_CHECK(c);
    if (fill_after_modes) {  // the node id per point, from the node starts (MIN_DISTANCE reads it on every level)
      hipLaunchKernelGGL(node_fill_kernel, dim3(div_up(m, NF_TILE)), dim3(256), 0, c->stream, lb.nstart, lb.counters, m, lb.nid,
                         plan.sampler == SWZ_MIN_DISTANCE ? 0 : 1);
      SWZ_LAUNCH_CHECK(c);
    }
  }

  const bool first_only = (plan.sampler == SWZ_RANDOM_GRID || plan.sampler == SWZ_GRID_CENTER) && plan.cand < 0;
  if (plan.sampler == SWZ_RANDOM_GRID || first_only) {
    // candidate level -1: "just take the first point" (Sampling.h:290-298, :346-348)
    const uint32_t csh = first_only ? plan.node_shift : level_shift(plan.cand);
    if (!first_only && plan.cand >= (int)MAX_LEVELS) return c->fail(SWZ_ERR_REROOT_UNSUPPORTED, level_error_message(SWZ_ERR_REROOT_UNSUPPORTED));
    ProfScope ps(c, "sample_random_grid", (uint64_t)m * 9ull);
    hipLaunchKernelGGL(random_grid_kernel, dim3(div_up(m, 256u * RG_IPT)), dim3(256), 0, c->stream, as.akey, m, lb.nid, lb.nmode, csh,
                       lb.taken, lb.counters);
    SWZ_LAUNCH_CHECK(c);
  } else if (plan.sampler == SWZ_GRID_CENTER || plan.sampler == SWZ_JITTERED) {
    if (plan.sampler == SWZ_GRID_CENTER && plan.cand >= (int)MAX_LEVELS)
      return c->fail(SWZ_ERR_REROOT_UNSUPPORTED, level_error_message(SWZ_ERR_REROOT_UNSUPPORTED));
    const uint32_t ntiles = div_up(m, GA_TILE);
    TileSummary* d_sum = nullptr;
    SWZ_TRY(c->get("grid_summaries", (size_t)ntiles, &d_sum));
    SWZ_HIP(c, hipMemsetAsync(lb.taken, 0, m, c->stream));
    GridParams g;
    g.root = plan.root;
    g.level = plan.level;
    g.sampler = plan.sampler;
    g.cand = plan.cand;
    g.spacing_node = plan.spacing_node;
    g.jitter_start = plan.jitter_start;
    g.box_table = nullptr;
    g.table_depth = 0;
    g.jit_table = nullptr;
    ProfScope ps(c, plan.sampler == SWZ_GRID_CENTER ? "sample_grid_center" : "sample_jittered", (uint64_t)m * 33ull,
                 2);
    {  // all but the last three steps of the bounds chain from a table (worth it from a few thousand points per entry on)
      const int chain = plan.sampler == SWZ_GRID_CENTER ? plan.cand + 1 : plan.level + 1;
      int td = std::min(chain - 3, GRID_TABLE_MAX_DEPTH);
      if (const char* e = c->opt("SWZ_GRID_TABLE_DEPTH")) td = std::min(std::min(atoi(e), chain), GRID_TABLE_MAX_DEPTH);
      while (td > 0 && ((uint64_t)1 << (3 * td)) * 64u > (uint64_t)m) --td;
      const char* jt = c->opt("SWZ_JITTER_TABLE");
      if (plan.sampler == SWZ_JITTERED && chain <= GRID_TABLE_MAX_DEPTH && !(jt && atoi(jt) == 0)) {
        // one entry per node prefix: the node's box and everything the sampler derives from it
        JitNode* d_nodes = nullptr;
        const uint32_t entries = 1u << (3 * chain);
        SWZ_TRY(c->get("grid_jitter_nodes", (size_t)entries, &d_nodes));
        hipLaunchKernelGGL(jitter_node_table_kernel, dim3(div_up(entries, 256)), dim3(256), 0, c->stream, plan.root, plan.level,
                           plan.spacing_node, d_nodes);
        SWZ_LAUNCH_CHECK(c);
        g.jit_table = d_nodes;
        td = 0;
      }
      if (td > 0) {
        Box* d_table = nullptr;
        SWZ_TRY(c->get("grid_boxes", (size_t)1 << (3 * td), &d_table));
        hipLaunchKernelGGL(grid_box_table_kernel, dim3(div_up(1u << (3 * td), 256)), dim3(256), 0, c->stream, plan.root, td, d_table);
        SWZ_LAUNCH_CHECK(c);
        g.box_table = d_table;
        g.table_depth = td;
      }
    }
    GridKeys gk;
    if (grid_level_uses_keys(c, plan, sp, &gk)) {
      // decided on the key coordinates; the runs they cannot decide repeated on the original positions
      KTileSummary* d_ksum = nullptr;
      const uint32_t nktiles = div_up(m, GAK_TILE);
      SWZ_TRY(c->get("grid_key_summaries", (size_t)nktiles, &d_ksum));
      SWZ_TRY(c->get("grid_key_undecided", (size_t)m / 2 + 1024, &gk.amb));  // (a run of one point is always decided)
      gk.amb_count = lb.counters + CTR_NUM_CELLS;
      hipLaunchKernelGGL(grid_argmin_keys_kernel, dim3(nktiles), dim3(GA_THREADS), 0, c->stream, as.akey, m, lb.nid, lb.nmode, g, gk,
                         plan.node_shift, lb.taken, d_ksum, lb.counters);
      SWZ_LAUNCH_CHECK(c);
      hipLaunchKernelGGL(grid_resolve_keys_kernel, dim3(div_up(nktiles, 256)), dim3(256), 0, c->stream, d_ksum, nktiles, gk, lb.taken);
      SWZ_LAUNCH_CHECK(c);
      hipLaunchKernelGGL(grid_exact_runs_kernel, dim3(std::min<uint32_t>(div_up(m, 2048u), 4096u)), dim3(256), 0, c->stream, as.akey, as.aidx,
                         sp, g, gk, lb.taken);
      SWZ_LAUNCH_CHECK(c);
    } else {
      if (!sp.X) return c->fail(SWZ_ERR_INTERNAL, "GRID_CENTER / JITTERED: this level needs the positions in Morton order");
      hipLaunchKernelGGL(grid_argmin_kernel, dim3(ntiles), dim3(GA_THREADS), 0, c->stream, as.akey, as.aidx, m, lb.nid,
                         lb.nmode, sp.X, sp.Y, sp.Z, g, plan.node_shift, lb.taken, d_sum, lb.counters);
      SWZ_LAUNCH_CHECK(c);
      hipLaunchKernelGGL(grid_resolve_kernel, dim3(div_up(ntiles, 256)), dim3(256), 0, c->stream, d_sum, ntiles,
                         lb.taken);
      SWZ_LAUNCH_CHECK(c);
    }
  } else {  // MIN_DISTANCE
    SWZ_HIP(c, hipMemsetAsync(lb.taken, 0, m, c->stream));
    uint32_t h[CTR_COUNT];
    SWZ_HIP(c, hipMemcpyAsync(h, lb.counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (h[CTR_ERROR]) return c->fail((int)h[CTR_ERROR], level_error_message((int)h[CTR_ERROR]));
    if (h[CTR_SAMPLE_NODES] < h[CTR_NUM_NODES]) {  // only levels that have take-all nodes pay for the pass
      hipLaunchKernelGGL(take_all_kernel, dim3(nb), dim3(256), 0, c->stream, m, lb.nid, lb.nmode, lb.taken);
      SWZ_LAUNCH_CHECK(c);
    }
    if (h[CTR_SAMPLE_NODES] > 0) {
      if (plan.md_property)
        SWZ_TRY(min_distance_property_level(c, plan, as, sp, lb, h[CTR_NUM_NODES], h[CTR_SAMPLE_NODES],
                                            h[CTR_SAMPLE_POINTS], &res->md_rounds));
      else
        SWZ_TRY(min_distance_level(c, plan, as, sp, lb, h[CTR_NUM_NODES], h[CTR_SAMPLE_NODES], h[CTR_SAMPLE_POINTS],
                                   &res->md_rounds));
    }
  }

  if (okey) {
    ProfScope ps(c, "level_compact", (uint64_t)m * 14ull, 2);
    SWZ_TRY(fused_scan(c, KeepF{lb.taken}, CompactG{as.akey, as.aidx, (int8_t)plan.level, level_out, okey, oidx}, m,
                       lb.counters + CTR_REMAINING, "lvl"));
  }
  uint32_t h[CTR_COUNT];
  SWZ_HIP(c, hipMemcpyAsync(h, lb.counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  c->prof_collect();
  if (h[CTR_ERROR]) {
    const int code = (int)h[CTR_ERROR];
    return c->fail(code, level_error_message(code));
  }
  res->remaining = h[CTR_REMAINING];
  res->num_nodes = h[CTR_NUM_NODES];
  res->node_prefix = nullptr;
  if (okey && res->remaining && res->num_nodes) {  // for the next level's segmentation (two buffers: the one of the level above is still read)
    uint64_t* np = nullptr;
    SWZ_TRY(c->get((plan.level & 1) ? "lvl_node_prefix_1" : "lvl_node_prefix_0", (size_t)res->num_nodes, &np));
    hipLaunchKernelGGL(node_prefix_kernel, dim3(div_up(res->num_nodes, 256)), dim3(256), 0, c->stream, lb.nstart, as.akey,
                       plan.node_shift, res->num_nodes, np);
    SWZ_LAUNCH_CHECK(c);
    res->node_prefix = np;
  }
  return SWZ_OK;
}

int alloc_level_buffers(swz_ctx* c, uint32_t m, LevelBuffers* lb) {
  SWZ_TRY(c->get("lvl_flags", (size_t)m, &lb->flags));
  SWZ_TRY(c->get("lvl_nid", (size_t)m, &lb->nid));
  SWZ_TRY(c->get("lvl_nstart", (size_t)m + 1, &lb->nstart));
  SWZ_TRY(c->get("lvl_nmode", (size_t)m, &lb->nmode));
  SWZ_TRY(c->get("lvl_taken", (size_t)m, &lb->taken));
  SWZ_TRY(c->get("lvl_counters", (size_t)CTR_COUNT, &lb->counters));
  return SWZ_OK;
}

// ----------------------------------------------------------------------------- drivers
// State of one batch between "indexed + sorted" and "all levels done".  Held in the context while a
// sharded batch waits for its neighbours' root samples (swz_shard_begin / swz_shard_finish).
struct TileSession {
  uint32_t n = 0;
  double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
  swz_tile_params params{};
  uint64_t* keys = nullptr;  // sorted keys
  uint32_t* perm = nullptr;  // original index per sorted position
  int8_t* level = nullptr;
  uint32_t* dup = nullptr;
  const double* xyz_in = nullptr;  // the caller's positions (clamped by the encode)
  SortedPoints sp;
  LevelBuffers lb;
  uint64_t* key_buf[2] = {nullptr, nullptr};
  uint32_t* idx_buf[2] = {nullptr, nullptr};
  int which = 0;
  ActiveSet as;
  int next_level = -1;
  uint64_t visited = 0, nodes = 0;
  uint32_t rounds = 0, nlevels = 0;
  int max_level = -1;
  int fast_start = -1;
  uint32_t ghosts = 0;  // leading points that belong to other shards (sharded batches only)
  uint32_t front = 0;   // entries kept free in front of the per-position arrays (sharded batches)
};

// positions of the session's points into Morton order (once)
static int session_gather_positions(swz_ctx* c, TileSession& t) {
  if (t.sp.X) return SWZ_OK;
  double *X = nullptr, *Y = nullptr, *Z = nullptr;
  SWZ_TRY(c->get("sorted_x", (size_t)t.n + t.front, &X));
  SWZ_TRY(c->get("sorted_y", (size_t)t.n + t.front, &Y));
  SWZ_TRY(c->get("sorted_z", (size_t)t.n + t.front, &Z));
  SWZ_STAGE(c, "sort");
  const uint32_t g = t.sp.ghosts;
  if (g) {
    // ghosts are attached already (a sharded batch whose earlier levels were decided on keys): they lead the sorted
    // order and their perm entries index the ghost array.  They matter at the root level only -- later the caller's
    // ghost array may be gone, and nobody reads those entries any more.
    if (t.next_level <= -1) SWZ_TRY(gather_positions(c, t.sp.ghost_xyz, t.perm, g, X, Y, Z));
    SWZ_TRY(gather_positions(c, t.xyz_in, t.perm + g, t.n - g, X + g, Y + g, Z + g));
  } else {
    X += t.front;
    Y += t.front;
    Z += t.front;
    SWZ_TRY(gather_positions(c, t.xyz_in, t.perm, t.n, X, Y, Z));
  }
  SWZ_STAGE(c, "gather");
  t.sp.X = X;
  t.sp.Y = Y;
  t.sp.Z = Z;
  return SWZ_OK;
}
// a level that cannot be decided on keys needs them
static int session_need_positions(swz_ctx* c, TileSession& t, const LevelPlan& plan) {
  if (t.sp.X || plan.sampler == SWZ_RANDOM_GRID) return SWZ_OK;
  if (plan.sampler == SWZ_MIN_DISTANCE ? min_distance_level_uses_keys(c, plan, t.sp) : grid_level_uses_keys(c, plan, t.sp, nullptr)) return SWZ_OK;
  return session_gather_positions(c, t);
}

// K1 + K2 + gather: index, sort, positions into Morton order
// `front`: entries kept free in FRONT of every per-sorted-position array (sharded batches prepend ghosts).
static int session_prepare(swz_ctx* c, TileSession& t, double* d_xyz, uint32_t n, const double bmin[3],
                           const double bmax[3], const swz_tile_params& p, const TileDeviceOut& out,
                           uint32_t front = 0) {
  t = TileSession{};
  t.n = n;
  for (int a = 0; a < 3; ++a) {
    t.bmin[a] = bmin[a];
    t.bmax[a] = bmax[a];
  }
  t.params = p;
  t.keys = out.keys;
  t.perm = out.perm;
  t.level = out.level;
  t.dup = out.dup;
  t.xyz_in = d_xyz;
  uint64_t* keys_b = nullptr;
  uint32_t* vals_b = nullptr;
  SWZ_TRY(c->get("sort_keys_b", (size_t)n, &keys_b));
  SWZ_TRY(c->get("sort_vals_b", (size_t)n, &vals_b));
  if (radix_result_in_second()) {  // place the input so that the sorted result lands in the output buffers
    SWZ_TRY(encode_device(c, d_xyz, n, bmin, bmax, keys_b));
    SWZ_STAGE(c, "encode");
    SWZ_TRY(radix_sort_pairs(c, keys_b, vals_b, out.keys, out.perm, n, true));
  } else {
    SWZ_TRY(encode_device(c, d_xyz, n, bmin, bmax, out.keys));
    SWZ_TRY(radix_sort_pairs(c, out.keys, out.perm, keys_b, vals_b, n, true));
  }
  // The positions in Morton order (SoA).  RANDOM_GRID decides on the keys alone.  MIN_DISTANCE decides on the key
  // coordinates and looks up the pairs inside the quantisation band through the permutation (swz_mdkeys.hip): there
  // the gather is put off until a level asks for it (session_need_positions) -- for cubic bounds and exact mode that is
  // a level so deep that its spacing spans fewer than 64 key cells, which few clouds reach.  Sharded batches that
  // prepend ghosts (front > 0) look up two position arrays: the sorted positions in front are the ghosts
  // (shard_attach_ghosts; sorted_point_xyz).
  t.sp = SortedPoints{nullptr, nullptr, nullptr, d_xyz, out.perm};
  t.front = front;
  if (p.sampler != SWZ_RANDOM_GRID) {
    const LevelPlan top = make_plan(-1, p.sampler, p.max_points_per_node, p.spacing_at_root, p.max_depth, bmin, bmax, false, true);
    const bool on_keys = p.sampler == SWZ_MIN_DISTANCE ? key_metric(c, top, t.sp).ok
                                                       : grid_level_uses_keys(c, top, t.sp, nullptr);
    if (!on_keys) SWZ_TRY(session_gather_positions(c, t));
  }
  if (out.dup) SWZ_HIP(c, hipMemsetAsync(out.dup, 0, (size_t)n * 4, c->stream));
  SWZ_HIP(c, hipMemsetAsync(out.level, 0x80, (size_t)n, c->stream));  // -128 = not persisted yet
  SWZ_TRY(alloc_level_buffers(c, n + front, &t.lb));
  // survivors ping-pong between the sort's secondary buffers and one extra pair
  t.key_buf[0] = keys_b;
  t.idx_buf[0] = vals_b;
  t.as = ActiveSet{out.keys, nullptr, n};
  return SWZ_OK;
}

// Runs the level loop from t.next_level while points remain and level <= last_level.
// root_mode: -1 = decide per node from its count; 0/1 force take-all/sample for the FIRST level run
// (sharded batches decide the root from the global point count).
static int session_run_levels(swz_ctx* c, TileSession& t, int last_level, int first_mode) {
  const uint64_t* pprefix = nullptr;
  uint32_t parents = 0;
  for (int level = t.next_level; t.as.m > 0 && level <= last_level; ++level) {
    if (level > 20) return c->fail(SWZ_ERR_INTERNAL, "level loop ran past level 20");
    if (!t.key_buf[t.which]) {
      SWZ_TRY(c->get("active_keys_2", (size_t)t.as.m, &t.key_buf[t.which]));
      SWZ_TRY(c->get("active_idx_2", (size_t)t.as.m, &t.idx_buf[t.which]));
    }
    LevelPlan plan = make_plan(level, t.params.sampler, t.params.max_points_per_node, t.params.spacing_at_root,
                               t.params.max_depth, t.bmin, t.bmax, false, true);
    plan.md_property = (t.params.flags & SWZ_FLAG_MIN_DISTANCE_PROPERTY) != 0;
    // (the root of a sharded batch spans the shards: it is sampled exactly -- the lower shards' samples as ghosts, or all
    // shards sweeping together -- which has the property a fortiori; the flag decides the levels below)
    if (first_mode >= 0 && level == t.next_level) plan.md_property = false;
    if (first_mode >= 0 && level == t.next_level && !plan.terminal) {
      if (first_mode == 1) {
        plan.force_sample = true;
      } else {
        plan.max_points = ~0ull;
      }
    }
    SWZ_TRY(session_need_positions(c, t, plan));
    LevelResult r;
    t.as.parent_prefix = pprefix;
    t.as.parents = pprefix ? parents : 0u;
    SWZ_TRY(level_step(c, plan, t.as, t.sp, t.lb, t.level, t.key_buf[t.which], t.idx_buf[t.which], &r));
    t.visited += t.as.m;
    t.nodes += r.num_nodes;
    t.rounds += r.md_rounds;
    t.max_level = level;
    ++t.nlevels;
    t.as = ActiveSet{t.key_buf[t.which], t.idx_buf[t.which], r.remaining};
    // (the nodes of this level for the next one's segmentation -- inside this call only: between two calls of a sharded
    // batch other work of the context may reuse the buffer)
    pprefix = r.node_prefix;
    parents = r.num_nodes;
    t.which ^= 1;
    t.next_level = level + 1;
  }
  return SWZ_OK;
}

static void session_stats(const TileSession& t, swz_tile_stats* stats) {
  if (!stats) return;
  stats->num_nodes = t.nodes;
  stats->points_visited = t.visited;
  stats->max_level = t.max_level;
  stats->fast_start_levels = t.fast_start;
  stats->num_levels = t.nlevels;
  stats->min_distance_rounds = t.rounds;
}

// ---- FAST (TilingAlgorithmV3) -------------------------------------------------------------------
// first index whose 6-octant prefix is >= bin, for every bin of the 8^6 grid (+ the end sentinel)
__global__ __launch_bounds__(256) void prefix_bounds_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                            uint32_t* __restrict__ starts, uint32_t nbins) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b > nbins) return;
  if (b == nbins) {
    starts[b] = n;
    return;
  }
  const uint64_t target = (uint64_t)b << 45;  // 63 - 6*3
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  starts[b] = lo;
}

// estimate_start_node_level_in_octree -- TilingAlgorithms.cpp:1473-1535, from the 6-level prefix counts
static size_t estimate_start_level_host(const std::vector<uint32_t>& starts6, size_t concurrency) {
  constexpr uint32_t MIN_LEVEL = 3, MAX_LEVEL = 6;
  constexpr float MIN_SCORE = 1.f;
  for (uint32_t level = 0; level < MAX_LEVEL; ++level) {
    const uint32_t digits = level + 1;
    const uint32_t group = 1u << (3 * (6 - digits));  // 6-digit bins per range at this level
    size_t ranges = 0, large = 0;
    for (uint32_t b = 0; b < (1u << 18); b += group) {
      const uint32_t cnt = starts6[b + group] - starts6[b];
      if (cnt > 0) ++ranges;
      if (cnt >= 100000) ++large;
    }
    float score = 0.f;
    if (!(ranges <= concurrency / 2)) score = static_cast<float>(large) / static_cast<float>(concurrency);
    if (score >= MIN_SCORE) return std::max(level + 1, MIN_LEVEL);
  }
  return MAX_LEVEL;
}

// children's persisted points of the nodes being reconstructed: taken at level S-1 (start nodes) or
// flagged as stored in the reconstructed node one level below
__global__ __launch_bounds__(256) void recon_select_kernel(const int8_t* __restrict__ level,
                                                           const uint32_t* __restrict__ dup, uint32_t n,
                                                           int start_node_level, uint32_t child_bit,
                                                           uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  flags[i] = child_bit ? ((dup[i] & child_bit) ? 1u : 0u) : (level[i] == (int8_t)start_node_level ? 1u : 0u);
}
__global__ __launch_bounds__(256) void recon_gather_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                           const uint32_t* __restrict__ flags_in_scanned,
                                                           const int8_t* __restrict__ level,
                                                           const uint32_t* __restrict__ dup, int start_node_level,
                                                           uint32_t child_bit, uint64_t* __restrict__ okey,
                                                           uint32_t* __restrict__ oidx) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool sel = child_bit ? ((dup[i] & child_bit) != 0) : (level[i] == (int8_t)start_node_level);
  if (sel) {
    const uint32_t o = flags_in_scanned[i];
    okey[o] = keys[i];
    oidx[o] = i;
  }
}
__global__ __launch_bounds__(256) void recon_mark_kernel(const uint32_t* __restrict__ aidx, uint32_t m,
                                                         const uint8_t* __restrict__ taken, uint32_t bit,
                                                         uint32_t* __restrict__ dup) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m && taken[i]) dup[aidx[i]] |= bit;
}

// points per 6-octant prefix of a sorted batch (what the start-level estimate looks at), host array of 2^18 counts
int fast_prefix_counts(swz_ctx* c, const uint64_t* d_keys_sorted, uint32_t n, uint32_t* counts_host) {
  const uint32_t nbins = 1u << 18;
  uint32_t* d_starts = nullptr;
  SWZ_TRY(c->get("fast_starts", (size_t)nbins + 1, &d_starts));
  std::vector<uint32_t> starts(nbins + 1, 0);
  if (n) {
    hipLaunchKernelGGL(prefix_bounds_kernel, dim3(div_up(nbins + 1, 256)), dim3(256), 0, c->stream, d_keys_sorted, n, d_starts, nbins);
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipMemcpyAsync(starts.data(), d_starts, (nbins + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
  }
  for (uint32_t b = 0; b < nbins; ++b) counts_host[b] = starts[b + 1] - starts[b];
  return SWZ_OK;
}
// the estimate from counts that may be the sum over the shards of a batch (each below 2^32 in total)
int fast_start_level_from_counts(const uint64_t* counts, uint32_t concurrency) {
  std::vector<uint32_t> starts((1u << 18) + 1, 0);
  uint64_t run = 0;
  for (uint32_t b = 0; b < (1u << 18); ++b) {
    starts[b] = (uint32_t)std::min<uint64_t>(run, 0xFFFFFFFFull);
    run += counts[b];
  }
  starts[1u << 18] = (uint32_t)std::min<uint64_t>(run, 0xFFFFFFFFull);
  return (int)estimate_start_level_host(starts, concurrency);
}

int fast_start_level(swz_ctx* c, const uint64_t* d_keys_sorted, uint32_t n, uint32_t concurrency, int* start_level) {
  const uint32_t nbins = 1u << 18;
  uint32_t* d_starts = nullptr;
  SWZ_TRY(c->get("fast_starts", (size_t)nbins + 1, &d_starts));
  hipLaunchKernelGGL(prefix_bounds_kernel, dim3(div_up(nbins + 1, 256)), dim3(256), 0, c->stream, d_keys_sorted, n,
                     d_starts, nbins);
  SWZ_LAUNCH_CHECK(c);
  std::vector<uint32_t> starts(nbins + 1);
  SWZ_HIP(c, hipMemcpyAsync(starts.data(), d_starts, (nbins + 1) * 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  *start_level = (int)estimate_start_level_host(starts, concurrency);
  return SWZ_OK;
}

// FAST: reconstruct the skipped levels S-1 .. lowest_lv, deepest first: a node with lv octants samples the points
// persisted by its (up to) 8 children with AlwaysAdhereToMinSpacing (reconstruct_single_node :1661-1715).
// lowest_lv = 0 includes the root; a shard of a sharded batch stops at 1 (the root's children lie on several shards).
static int session_fast_reconstruct(swz_ctx* c, TileSession& t, const swz_tile_params& p, int S, int lowest_lv) {
  const uint32_t n = t.n;
  const double* bmin = t.bmin;
  const double* bmax = t.bmax;
  uint64_t* rkey = nullptr;
  uint32_t* ridx = nullptr;
  SWZ_TRY(c->get("recon_keys", (size_t)n, &rkey));
  SWZ_TRY(c->get("recon_idx", (size_t)n, &ridx));
  const uint32_t nb = div_up(n, 256);
  for (int lv = S - 1; lv >= lowest_lv; --lv) {
    const uint32_t child_bit = (lv + 1 == S) ? 0u : (1u << (lv + 1));
    hipLaunchKernelGGL(recon_select_kernel, dim3(nb), dim3(256), 0, c->stream, t.level, t.dup, n, S - 1, child_bit,
                       t.lb.flags);
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipMemsetAsync(t.lb.counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
    SWZ_TRY(scan_exclusive_u32(c, t.lb.flags, t.lb.flags, n, t.lb.counters + CTR_REMAINING, "rec"));
    hipLaunchKernelGGL(recon_gather_kernel, dim3(nb), dim3(256), 0, c->stream, t.keys, n, t.lb.flags, t.level, t.dup,
                       S - 1, child_bit, rkey, ridx);
    SWZ_LAUNCH_CHECK(c);
    uint32_t m = 0;
    SWZ_HIP(c, hipMemcpyAsync(&m, t.lb.counters + CTR_REMAINING, 4, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (m == 0) continue;
    LevelPlan plan = make_plan(lv - 1, p.sampler, p.max_points_per_node, p.spacing_at_root, p.max_depth, bmin,
                               bmax, true, false);
    plan.md_property = (p.flags & SWZ_FLAG_MIN_DISTANCE_PROPERTY) != 0;
    ActiveSet as{rkey, ridx, m};
    SWZ_TRY(session_need_positions(c, t, plan));
    LevelResult r;
    SWZ_TRY(level_step(c, plan, as, t.sp, t.lb, nullptr, nullptr, nullptr, &r));
    hipLaunchKernelGGL(recon_mark_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, ridx, m, t.lb.taken,
                       1u << lv, t.dup);
    SWZ_LAUNCH_CHECK(c);
    t.nodes += r.num_nodes;
    t.rounds += r.md_rounds;
  }
  return SWZ_OK;
}

int tile_device(swz_ctx* c, double* d_xyz, uint32_t n, const double bmin[3], const double bmax[3],
                const swz_tile_params& p, const TileDeviceOut& out_in, swz_tile_stats* stats) {
  TileDeviceOut out = out_in;
  if (p.strategy == SWZ_FAST && !out.dup) SWZ_TRY(c->get("fast_dup", (size_t)n, &out.dup));
  TileSession t;
  SWZ_TRY(session_prepare(c, t, d_xyz, n, bmin, bmax, p, out));
  if (p.strategy == SWZ_ACCURATE) {
    SWZ_TRY(session_run_levels(c, t, 20, -1));
    session_stats(t, stats);
    return SWZ_OK;
  }
  // ---- FAST: TilingAlgorithmV3 first iteration (:1250-1360) + finalize (:1717-1784)
  int S = 0;
  SWZ_TRY(fast_start_level(c, t.keys, n, p.fast_concurrency, &S));
  t.fast_start = S;
  // every point starts in the node made of its first S octants (split_indexed_points_into_subranges)
  t.next_level = S - 1;
  SWZ_TRY(session_run_levels(c, t, 20, -1));
  SWZ_TRY(session_fast_reconstruct(c, t, p, S, 0));
  session_stats(t, stats);
  return SWZ_OK;
}

// ---- sharded batches ------------------------------------------------------------------------------
struct ShardState {
  TileSession t;
  uint32_t n_local = 0;
  bool open = false;
  // swz_shard_presort_device ran: the local points are indexed and sorted, `front` entries are free in front
  bool presorted = false;
  uint32_t front = 0;
  const double* xyz_local = nullptr;
  bool perm_local = false;  // perm of the local points counts from the first LOCAL point
  bool empty = false;       // the open batch has no local points
  bool fast = false;        // the open batch runs the FAST strategy (swz_shard_fast_*)
  uint32_t fast_candidates = 0;  // points of this shard's level-0 nodes: what the root is reconstructed from
};

static ShardState* shard_state(swz_ctx* c) {
  if (!c->shard) c->shard = new ShardState();
  return static_cast<ShardState*>(c->shard);
}
int shard_begin_empty(swz_ctx* c) {
  ShardState* s = shard_state(c);
  s->presorted = false;
  s->fast = false;
  s->t = TileSession{};
  s->n_local = 0;
  s->empty = true;
  s->open = true;
  return SWZ_OK;
}
void shard_free(swz_ctx* c) {
  delete static_cast<ShardState*>(c->shard);
  c->shard = nullptr;
}

__global__ __launch_bounds__(256) void root_taken_count_kernel(const int8_t* __restrict__ level, uint32_t first,
                                                               uint32_t n, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[i] = (i >= first && level[i] == (int8_t)-1) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void root_taken_gather_kernel(const int8_t* __restrict__ level, uint32_t first,
                                                                uint32_t n, const uint32_t* __restrict__ pos,
                                                                SortedPoints sp, double* __restrict__ out_xyz) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || i < first || level[i] != (int8_t)-1) return;
  const uint64_t o = pos[i];
  if (sp.X) {
    out_xyz[3 * o] = sp.X[i];
    out_xyz[3 * o + 1] = sp.Y[i];
    out_xyz[3 * o + 2] = sp.Z[i];
  } else {  // the positions were never brought into Morton order (the samplers decided on keys)
    const double* p = sorted_point_xyz(sp.xyz, sp.perm, sp.ghost_xyz, sp.ghosts, i);
    out_xyz[3 * o] = p[0];
    out_xyz[3 * o + 1] = p[1];
    out_xyz[3 * o + 2] = p[2];
  }
}
__global__ __launch_bounds__(256) void shard_strip_kernel(const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ perm,
                                                          const int8_t* __restrict__ level, uint32_t ghosts,
                                                          uint32_t perm_base, uint32_t n_local,
                                                          uint64_t* __restrict__ okeys, uint32_t* __restrict__ operm,
                                                          int8_t* __restrict__ olevel) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_local) return;
  okeys[i] = keys[ghosts + i];
  operm[i] = perm[ghosts + i] - perm_base;
  olevel[i] = level[ghosts + i];
}

// Everything of swz_shard_begin_device that does not depend on the ghosts: index + sort + gather of the local
// points, with room for up to ghost_capacity ghosts in front of every array.  All shards can do this at the
// same time, so that only the root node itself is left in the chain that passes the ghosts from shard to shard.
int shard_presort_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3], const double bmax[3],
                         const swz_tile_params& p, uint32_t ghost_capacity) {
  if (p.strategy != SWZ_ACCURATE) return c->fail(SWZ_ERR_BAD_ARG, "this call runs the ACCURATE strategy of a sharded batch (FAST: swz_shard_fast_*)");
  if ((uint64_t)n + ghost_capacity > 0xFFFFFFFEull) return c->fail(SWZ_ERR_TOO_MANY_POINTS, "shard + ghosts exceed 2^32-2 points");
  ShardState* s = shard_state(c);
  s->open = false;
  s->presorted = false;
  s->fast = false;
  TileDeviceOut out{};
  const size_t cap = (size_t)n + ghost_capacity;
  SWZ_TRY(c->get("shard_keys", cap, &out.keys));
  SWZ_TRY(c->get("shard_perm", cap, &out.perm));
  SWZ_TRY(c->get("shard_level", cap, &out.level));
  out.keys += ghost_capacity;
  out.perm += ghost_capacity;
  out.level += ghost_capacity;
  SWZ_TRY(session_prepare(c, s->t, const_cast<double*>(d_xyz_local), n, bmin, bmax, p, out, ghost_capacity));
  s->n_local = n;
  s->front = ghost_capacity;
  s->xyz_local = d_xyz_local;
  s->perm_local = true;
  s->presorted = true;
  return SWZ_OK;
}

// ghosts lie in lower octants, so their keys are smaller than every local key: sorted ghosts ++ sorted locals
// is the sorted whole.  Writes the g ghosts into the free entries in front of the presorted arrays.
static int shard_attach_ghosts(swz_ctx* c, ShardState* s, const double* d_ghost_xyz, uint32_t g) {
  TileSession& t = s->t;
  if (g) {
    uint64_t* tmpk = nullptr;
    uint32_t* tmpv = nullptr;
    SWZ_TRY(c->get("ghost_keys", (size_t)g, &tmpk));
    SWZ_TRY(c->get("ghost_vals", (size_t)g, &tmpv));
    uint64_t* gk = t.keys - g;
    uint32_t* gp = t.perm - g;
    double* gx = const_cast<double*>(d_ghost_xyz);  // inside the bounds already: the clamp of the encode is a no-op
    if (radix_result_in_second()) {
      SWZ_TRY(encode_device(c, gx, g, t.bmin, t.bmax, tmpk));
      SWZ_TRY(radix_sort_pairs(c, tmpk, tmpv, gk, gp, g, true));
    } else {
      SWZ_TRY(encode_device(c, gx, g, t.bmin, t.bmax, gk));
      SWZ_TRY(radix_sort_pairs(c, gk, gp, tmpk, tmpv, g, true));
    }
    if (t.sp.X) SWZ_TRY(gather_positions(c, d_ghost_xyz, gp, g, const_cast<double*>(t.sp.X) - g, const_cast<double*>(t.sp.Y) - g,
                                         const_cast<double*>(t.sp.Z) - g));
    SWZ_HIP(c, hipMemsetAsync(t.level - g, 0x80, (size_t)g, c->stream));
    t.keys -= g;
    t.perm -= g;
    t.level -= g;
    if (t.sp.X) {
      t.sp.X -= g;
      t.sp.Y -= g;
      t.sp.Z -= g;
    }
    t.sp.perm = t.perm;  // (now starts with the ghosts' entries, which index the ghost array)
    t.sp.ghost_xyz = d_ghost_xyz;
    t.sp.ghosts = g;
    t.n += g;
    t.as = ActiveSet{t.keys, nullptr, t.n};
  }
  t.ghosts = g;
  return SWZ_OK;
}

int shard_begin_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3],
                       const double bmax[3], const swz_tile_params& p, uint64_t global_points,
                       const double* d_ghost_xyz, uint32_t ghosts, uint64_t* num_root_taken) {
  if (p.strategy != SWZ_ACCURATE) return c->fail(SWZ_ERR_BAD_ARG, "this call runs the ACCURATE strategy of a sharded batch (FAST: swz_shard_fast_*)");
  ShardState* s = shard_state(c);
  s->open = false;
  s->empty = false;
  s->fast = false;
  const uint32_t total = n + ghosts;
  const bool fast = s->presorted && s->xyz_local == d_xyz_local && s->n_local == n && ghosts <= s->front;
  s->presorted = false;
  if (fast) {
    SWZ_TRY(shard_attach_ghosts(c, s, d_ghost_xyz, ghosts));
  } else {
  s->perm_local = false;
  double* xyz = nullptr;
  if (ghosts == 0) {
    xyz = const_cast<double*>(d_xyz_local);  // already inside the bounds (it was encoded before the exchange)
  } else if (d_ghost_xyz + (size_t)ghosts * 3 == d_xyz_local) {
    xyz = const_cast<double*>(d_ghost_xyz);  // caller laid the ghosts out right in front of its points
  } else {
    SWZ_TRY(c->get("shard_xyz", (size_t)total * 3, &xyz));
    SWZ_HIP(c, hipMemcpyAsync(xyz, d_ghost_xyz, (size_t)ghosts * 24, hipMemcpyDeviceToDevice, c->stream));
    SWZ_HIP(c, hipMemcpyAsync(xyz + (size_t)ghosts * 3, d_xyz_local, (size_t)n * 24, hipMemcpyDeviceToDevice,
                              c->stream));
  }
  TileDeviceOut out{};
  SWZ_TRY(c->get("shard_keys", (size_t)total, &out.keys));
  SWZ_TRY(c->get("shard_perm", (size_t)total, &out.perm));
  SWZ_TRY(c->get("shard_level", (size_t)total, &out.level));
  SWZ_TRY(session_prepare(c, s->t, xyz, total, bmin, bmax, p, out));
  s->t.ghosts = ghosts;
  s->n_local = n;
  }
  // the root node spans all shards: its take-all / sample decision uses the global point count
  const LevelPlan root_plan =
    make_plan(-1, p.sampler, p.max_points_per_node, p.spacing_at_root, p.max_depth, bmin, bmax, false, true);
  if ((p.sampler == SWZ_RANDOM_GRID || p.sampler == SWZ_GRID_CENTER) && root_plan.cand < 0 &&
      global_points > p.max_points_per_node)
    return c->fail(SWZ_ERR_BAD_ARG, "sharded root with candidate level -1 (spacing >= half the extent) is unsupported");
  SWZ_TRY(session_run_levels(c, s->t, -1, global_points > p.max_points_per_node ? 1 : 0));
  // how many LOCAL points the root took (their positions become the next shard's ghosts)
  const uint32_t nb = div_up(total, 256);
  SWZ_HIP(c, hipMemsetAsync(s->t.lb.counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(root_taken_count_kernel, dim3(nb), dim3(256), 0, c->stream, s->t.level, ghosts, total,
                     s->t.lb.flags);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, s->t.lb.flags, s->t.lb.flags, total, s->t.lb.counters + CTR_REMAINING, "shr"));
  uint32_t cnt = 0;
  SWZ_HIP(c, hipMemcpyAsync(&cnt, s->t.lb.counters + CTR_REMAINING, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (num_root_taken) *num_root_taken = cnt;
  s->open = true;
  return SWZ_OK;
}

int shard_root_taken_device(swz_ctx* c, double* d_xyz_out) {
  ShardState* s = shard_state(c);
  if (!s->open) return c->fail(SWZ_ERR_BAD_ARG, "no sharded batch is open");
  if (s->empty) return SWZ_OK;
  const uint32_t total = s->t.n;
  // lb.flags still holds the exclusive scan of the root-taken flags of swz_shard_begin
  hipLaunchKernelGGL(root_taken_gather_kernel, dim3(div_up(total, 256)), dim3(256), 0, c->stream, s->t.level,
                     s->t.ghosts, total, s->t.lb.flags, s->t.sp, d_xyz_out);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

int shard_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out,
                        swz_tile_stats* stats) {
  ShardState* s = shard_state(c);
  if (!s->open) return c->fail(SWZ_ERR_BAD_ARG, "no sharded batch is open");
  s->open = false;
  if (s->empty) {
    s->empty = false;
    session_stats(s->t, stats);
    return SWZ_OK;
  }
  SWZ_TRY(session_run_levels(c, s->t, 20, -1));
  hipLaunchKernelGGL(shard_strip_kernel, dim3(div_up(s->n_local, 256)), dim3(256), 0, c->stream, s->t.keys, s->t.perm,
                     s->t.level, s->t.ghosts, s->perm_local ? 0u : s->t.ghosts, s->n_local, d_keys_out, d_perm_out, d_level_out);
  SWZ_LAUNCH_CHECK(c);
  session_stats(s->t, stats);
  return SWZ_OK;
}

// ---- FAST (TilingAlgorithmV3, the reference's default) on a sharded batch.  The start level comes from the distribution
// of the WHOLE batch (:1473-1535): every shard reports the counts of its part per 6-octant prefix, the driver sums them
// and tells every shard the level.  Start nodes lie at level >= 2, inside one shard's octants, so the levels from there
// down and the reconstruction of the skipped levels down to level 0 (:1717-1784) are local; the root is reconstructed
// from what the level-0 nodes of ALL shards hold -- in octant order, which is shard order --, so the driver collects
// those candidates (swz_shard_fast_root_candidates_device), samples them in one place (swz_sample_points_device with
// AlwaysAdhereToMinSpacing at node level -1) and hands every shard the flags of its part.
__global__ __launch_bounds__(256) void shard_fast_cand_kernel(const uint64_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ pos,
                                                              const int8_t* __restrict__ level, const uint32_t* __restrict__ dup,
                                                              int start_node_level, uint32_t child_bit, SortedPoints sp,
                                                              uint64_t* __restrict__ okeys, double* __restrict__ oxyz) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool sel = child_bit ? ((dup[i] & child_bit) != 0) : (level[i] == (int8_t)start_node_level);
  if (!sel) return;
  const uint64_t o = pos[i];
  okeys[o] = keys[i];
  if (sp.X) {
    oxyz[3 * o] = sp.X[i];
    oxyz[3 * o + 1] = sp.Y[i];
    oxyz[3 * o + 2] = sp.Z[i];
  } else {
    const double* q = sorted_point_xyz(sp.xyz, sp.perm, sp.ghost_xyz, sp.ghosts, i);
    oxyz[3 * o] = q[0];
    oxyz[3 * o + 1] = q[1];
    oxyz[3 * o + 2] = q[2];
  }
}
__global__ __launch_bounds__(256) void shard_fast_mark_root_kernel(uint32_t n, const uint32_t* __restrict__ pos, const int8_t* __restrict__ level,
                                                                   uint32_t* __restrict__ dup, int start_node_level, uint32_t child_bit,
                                                                   const uint8_t* __restrict__ taken) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool sel = child_bit ? ((dup[i] & child_bit) != 0) : (level[i] == (int8_t)start_node_level);
  if (sel && taken[pos[i]]) dup[i] |= 1u;
}
__global__ __launch_bounds__(256) void shard_fast_strip_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                                                               const int8_t* __restrict__ level, const uint32_t* __restrict__ dup, uint32_t n,
                                                               uint64_t* __restrict__ okeys, uint32_t* __restrict__ operm,
                                                               int8_t* __restrict__ olevel, uint32_t* __restrict__ odup) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  okeys[i] = keys[i];
  operm[i] = perm[i];
  olevel[i] = level[i];
  odup[i] = dup[i];
}

int shard_fast_begin_device(swz_ctx* c, const double* d_xyz_local, uint32_t n, const double bmin[3], const double bmax[3],
                            const swz_tile_params& p, uint32_t* counts_host) {
  if (p.strategy != SWZ_FAST) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_begin_device: not the FAST strategy");
  ShardState* s = shard_state(c);
  s->open = false;
  s->presorted = false;
  s->fast = true;
  s->fast_candidates = 0;
  s->empty = n == 0;
  s->n_local = n;
  if (n == 0) {
    s->t = TileSession{};
    s->t.params = p;
    for (uint32_t b = 0; b < (1u << 18); ++b) counts_host[b] = 0;
    s->open = true;
    return SWZ_OK;
  }
  TileDeviceOut out{};
  SWZ_TRY(c->get("shard_keys", (size_t)n, &out.keys));
  SWZ_TRY(c->get("shard_perm", (size_t)n, &out.perm));
  SWZ_TRY(c->get("shard_level", (size_t)n, &out.level));
  SWZ_TRY(c->get("shard_dup", (size_t)n, &out.dup));
  SWZ_TRY(session_prepare(c, s->t, const_cast<double*>(d_xyz_local), n, bmin, bmax, p, out));
  s->perm_local = true;
  SWZ_TRY(fast_prefix_counts(c, s->t.keys, n, counts_host));
  s->open = true;
  return SWZ_OK;
}

// the levels from the start level down, the local reconstruction, and how many points this shard's level-0 nodes hold
int shard_fast_run_device(swz_ctx* c, int start_level, uint64_t* num_root_candidates) {
  ShardState* s = shard_state(c);
  if (!s->open || !s->fast) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_run: no FAST sharded batch is open");
  if (start_level < 1 || start_level > 6) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_run: start levels 1..6");
  *num_root_candidates = 0;
  TileSession& t = s->t;
  t.fast_start = start_level;
  if (s->empty) return SWZ_OK;
  t.next_level = start_level - 1;
  SWZ_TRY(session_run_levels(c, t, 20, -1));
  SWZ_TRY(session_fast_reconstruct(c, t, t.params, start_level, 1));
  // what the root's children hold (the selection of reconstruct level 0)
  const uint32_t child_bit = (1 == start_level) ? 0u : 2u;
  uint32_t* pos = nullptr;
  SWZ_TRY(c->get("shard_fast_pos", (size_t)t.n, &pos));
  hipLaunchKernelGGL(recon_select_kernel, dim3(div_up(t.n, 256)), dim3(256), 0, c->stream, t.level, t.dup, t.n, start_level - 1, child_bit, pos);
  SWZ_LAUNCH_CHECK(c);
  SWZ_HIP(c, hipMemsetAsync(t.lb.counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
  SWZ_TRY(scan_exclusive_u32(c, pos, pos, t.n, t.lb.counters + CTR_REMAINING, "rec"));
  uint32_t m = 0;
  SWZ_HIP(c, hipMemcpyAsync(&m, t.lb.counters + CTR_REMAINING, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  s->fast_candidates = m;
  *num_root_candidates = m;
  return SWZ_OK;
}

int shard_fast_root_candidates_device(swz_ctx* c, uint64_t* d_keys_out, double* d_xyz_out) {
  ShardState* s = shard_state(c);
  if (!s->open || !s->fast) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_root_candidates_device: no FAST sharded batch is open");
  if (s->empty || !s->fast_candidates) return SWZ_OK;
  TileSession& t = s->t;
  uint32_t* pos = nullptr;
  SWZ_TRY(c->get("shard_fast_pos", (size_t)t.n, &pos));
  hipLaunchKernelGGL(shard_fast_cand_kernel, dim3(div_up(t.n, 256)), dim3(256), 0, c->stream, t.keys, t.n, pos, t.level, t.dup,
                     t.fast_start - 1, (1 == t.fast_start) ? 0u : 2u, t.sp, d_keys_out, d_xyz_out);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

int shard_fast_set_root_device(swz_ctx* c, const uint8_t* d_taken) {
  ShardState* s = shard_state(c);
  if (!s->open || !s->fast) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_set_root_device: no FAST sharded batch is open");
  if (s->empty || !s->fast_candidates) return SWZ_OK;
  TileSession& t = s->t;
  uint32_t* pos = nullptr;
  SWZ_TRY(c->get("shard_fast_pos", (size_t)t.n, &pos));
  hipLaunchKernelGGL(shard_fast_mark_root_kernel, dim3(div_up(t.n, 256)), dim3(256), 0, c->stream, t.n, pos, t.level, t.dup,
                     t.fast_start - 1, (1 == t.fast_start) ? 0u : 2u, d_taken);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

int shard_fast_finish_device(swz_ctx* c, uint64_t* d_keys_out, uint32_t* d_perm_out, int8_t* d_level_out, uint32_t* d_dup_out,
                             swz_tile_stats* stats) {
  ShardState* s = shard_state(c);
  if (!s->open || !s->fast) return c->fail(SWZ_ERR_BAD_ARG, "swz_shard_fast_finish_device: no FAST sharded batch is open");
  s->open = false;
  s->fast = false;
  if (!s->empty) {
    hipLaunchKernelGGL(shard_fast_strip_kernel, dim3(div_up(s->n_local, 256)), dim3(256), 0, c->stream, s->t.keys, s->t.perm, s->t.level,
                       s->t.dup, s->n_local, d_keys_out, d_perm_out, d_level_out, d_dup_out);
    SWZ_LAUNCH_CHECK(c);
  }
  s->empty = false;
  session_stats(s->t, stats);
  return SWZ_OK;
}

__global__ __launch_bounds__(256) void count_taken_kernel(const uint8_t* __restrict__ taken, uint32_t n,
                                                          uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool t = (i < n) && taken[i];
  const uint64_t b = __ballot(t);
  if (lane_id() == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
}

// every key of the range must lie in the node: the reference takes the node's bounds from node_key
// (Sampling.h:441, 622), this implementation from the keys' own prefix -- the two agree exactly then
__global__ __launch_bounds__(256) void node_key_check_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t nsh,
                                                             uint64_t prefix, uint32_t* __restrict__ bad) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool b = i < n && (keys[i] >> nsh) != prefix;
  const uint64_t m = __ballot(b);
  if (lane_id() == 0 && m) atomicAdd(bad, (uint32_t)__popcll(m));
}

int sample_points_device(swz_ctx* c, int sampler, uint64_t max_points, const uint64_t* d_keys, const uint32_t* d_idx,
                         uint32_t n, const double* d_xyz, uint64_t node_key, int32_t node_level,
                         const double rmin[3], const double rmax[3], float spacing, int behaviour, uint8_t* d_taken,
                         uint64_t* num_taken) {
  // RANDOM_GRID and GRID_CENTER never look at node_key: the whole range is "the node" (count, candidate level from
  // node_level; the reference's own test samples a range spanning all octants at node level 0,
  // test/TestOctreeIndexing.cpp:169-252).  MIN_DISTANCE and JITTERED take the node's box from node_key.
  const bool uses_node_key = sampler == SWZ_MIN_DISTANCE || sampler == SWZ_JITTERED;
  if (node_level >= 0 && uses_node_key) {
    uint32_t* d_bad = nullptr;
    SWZ_TRY(c->get("lvl_counters", (size_t)CTR_COUNT, &d_bad));
    SWZ_HIP(c, hipMemsetAsync(d_bad, 0, sizeof(uint32_t), c->stream));
    const uint32_t nsh = level_shift(node_level);
    hipLaunchKernelGGL(node_key_check_kernel, dim3(div_up(n, 256)), dim3(256), 0, c->stream, d_keys, n, nsh,
                       node_key >> nsh, d_bad);
    SWZ_LAUNCH_CHECK(c);
    uint32_t bad = 0;
    SWZ_HIP(c, hipMemcpyAsync(&bad, d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (bad) return c->fail(SWZ_ERR_BAD_ARG, "swz_sample_points: " + std::to_string(bad) + " keys of the range do not lie in node_key's node");
  }
  double *X = nullptr, *Y = nullptr, *Z = nullptr;
  SWZ_TRY(c->get("sorted_x", (size_t)n, &X));
  SWZ_TRY(c->get("sorted_y", (size_t)n, &Y));
  SWZ_TRY(c->get("sorted_z", (size_t)n, &Z));
  SWZ_TRY(gather_positions(c, d_xyz, d_idx, n, X, Y, Z));
  LevelBuffers lb;
  SWZ_TRY(alloc_level_buffers(c, n, &lb));
  LevelPlan plan = make_plan(node_level, sampler, max_points, spacing, 100, rmin, rmax,
                             behaviour == SWZ_ALWAYS_ADHERE_TO_MIN_SPACING, false);
  if (!uses_node_key) plan.node_shift = 63;  // one node: the range
  ActiveSet as{d_keys, nullptr, n};
  SortedPoints sp{X, Y, Z, d_xyz, d_idx};
  LevelResult r;
  SWZ_TRY(level_step(c, plan, as, sp, lb, nullptr, nullptr, nullptr, &r));
  SWZ_HIP(c, hipMemcpyAsync(d_taken, lb.taken, n, hipMemcpyDeviceToDevice, c->stream));
  if (num_taken) {
    SWZ_HIP(c, hipMemsetAsync(lb.counters, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(count_taken_kernel, dim3(div_up(n, 256)), dim3(256), 0, c->stream, lb.taken, n, lb.counters);
    SWZ_LAUNCH_CHECK(c);
    uint32_t cnt = 0;
    SWZ_HIP(c, hipMemcpyAsync(&cnt, lb.counters, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    *num_taken = cnt;
  }
  return SWZ_OK;
}

}  // namespace swz
