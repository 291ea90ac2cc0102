// swz_level.h -- per-level state shared between the level-synchronous tiler (swz_level.hip) and the
// MIN_DISTANCE dependency sweep (swz_mindist.hip).
#pragma once
#include "swz_device.h"
#include "swz_internal.h"

namespace swz {

enum : uint8_t { MODE_TAKE_ALL = 0, MODE_SAMPLE = 1 };

// device counters of one level iteration
enum {
  CTR_NUM_NODES = 0,     // nodes at this level
  CTR_SAMPLE_NODES = 1,  // nodes that run the sampler (count > max_points or forced)
  CTR_SAMPLE_POINTS = 2, // points inside those nodes
  CTR_ERROR = 3,         // SWZ_ERR_* raised by a kernel (0 = none)
  CTR_REMAINING = 4,     // points handed to the next level
  CTR_NUM_CELLS = 5,     // MIN_DISTANCE: cells
  CTR_DONE_CELLS = 6,    // MIN_DISTANCE: cells finished
  CTR_Q0 = 8,            // MIN_DISTANCE: three rotating queue counters
  // MIN_DISTANCE sweep, builds with -DSWZ_MD_STATS only (printed with SWZ_DEBUG=1): timings of a sample of the
  // activations in 10 ns ticks
  CTR_DBG_TIME = 20,     // total
  CTR_DBG_TMAX = 21,     // longest
  CTR_DBG_HIST = 22,     // +0 activations sampled, +1 prologue, +2 chunk loads and rejection tests, +3 blocker scans,
                         // +4 chunks, +5 cells scanned
  CTR_COUNT = 40
};

// The active set of one level: Morton-sorted survivors.  aidx == nullptr means identity (level -1).
struct ActiveSet {
  const uint64_t* akey = nullptr;  // key of every active point
  const uint32_t* aidx = nullptr;  // its position in the fully sorted arrays (X/Y/Z/level)
  uint32_t m = 0;
  // multi-batch tiling (swz_tiler.hip): the keys of the points earlier batches persisted in the nodes of this level
  // that the active set touches, ascending by node prefix.  A node that has some is sampled with
  // SamplingBehaviour::AlwaysAdhereToMinSpacing (tile_internal_node, TilingAlgorithms.cpp:272-275).
  const uint64_t* ckey = nullptr;
  uint32_t nc = 0;
  // ... and where they sit among the active points: aidx values in [old_lo, old_hi) are pulled entries.  The entries of a
  // file that holds more than max_points were the OUTPUT of this sampler at this spacing, so they are pairwise at least
  // one spacing apart, and MIN_DISTANCE only has to look at what the new points can change (swz_mdblock.hip).
  uint32_t old_lo = 0, old_hi = 0;
  const uint64_t* new_key = nullptr;  // the batch's own points before the merge, ascending
  uint32_t new_m = 0;
  // the nodes that hold files one level further down (their keys with the bits below the node cleared, ascending): a node with a
  // child there has handed points down, i.e. it has been sampled, and its file -- rewritten by every visit since -- is a sampler's
  // output whatever its size
  const uint64_t* child_nkey = nullptr;
  uint32_t child_nn = 0;
  // the nodes of the level above (LevelResult::node_prefix of the step whose survivors these are), when the caller has
  // them: every node of this level is a child of one of them, so its first point is found by searching the sorted keys
  // instead of by a scan over all points (level_step; null: scan)
  const uint64_t* parent_prefix = nullptr;
  uint32_t parents = 0;
};

struct SortedPoints {
  const double* X = nullptr;  // positions in Morton order, SoA (null when the sampler decides on the keys: see xyz / perm)
  const double* Y = nullptr;
  const double* Z = nullptr;
  // The clamped input positions (AoS, caller's order) and the sort's permutation: sorted position s is point perm[s].
  // MIN_DISTANCE decides almost every pair on the coordinates its keys already hold (swz_mdkeys.hip) and looks up the
  // few pairs inside the quantisation band here, so the positions are never brought into Morton order.
  const double* xyz = nullptr;
  const uint32_t* perm = nullptr;
  // A sharded batch: the first `ghosts` sorted positions are points of lower shards (they sort first: lower octants); their
  // perm entries index ghost_xyz instead.
  const double* ghost_xyz = nullptr;
  uint32_t ghosts = 0;
};
// exact position of sorted position s
__host__ __device__ inline const double* sorted_point_xyz(const double* xyz, const uint32_t* perm, const double* ghost_xyz, uint32_t ghosts,
                                                          uint32_t s) {
  return (s < ghosts ? ghost_xyz : xyz) + (size_t)perm[s] * 3;
}

// What the host decides once per level (all float/libm corner cases of the reference live here,
// evaluated with the host's glibc exactly like the reference evaluates them).
struct LevelPlan {
  int level = -1;           // node level (-1 = root)
  uint32_t node_shift = 63; // key >> node_shift == node prefix
  int sampler = 0;
  uint64_t max_points = 0;
  bool force_sample = false; // SamplingBehaviour::AlwaysAdhereToMinSpacing
  bool terminal = false;     // tile_terminal_node: every node of this level keeps all its points
  bool reroot = false;       // sampling this level would need Morton re-rooting
  bool md_property = false;  // MIN_DISTANCE: SWZ_FLAG_MIN_DISTANCE_PROPERTY (swz_mdprop.hip)
  Box root;
  // RANDOM_GRID / GRID_CENTER: candidate_level_in_octree (Sampling.h:223-229); -1 = first point only
  int cand = -1;
  // JITTERED / MIN_DISTANCE
  double spacing_node = 0.0; // spacing_at_root / pow(2, level + 1)
  uint32_t jitter_start = 0; // (3 * (level + 1)) % 16
  // MIN_DISTANCE
  double sq_spacing = 0.0;   // (double)((float)spacing_node * (float)spacing_node)
  int cell_levels_geo = 0;   // finest cell subdivision (levels below the node) whose cells are >= spacing
};

struct LevelBuffers {
  uint32_t* flags = nullptr;   // m
  uint32_t* nid = nullptr;     // m
  uint32_t* nstart = nullptr;  // m + 1
  uint8_t* nmode = nullptr;    // m
  uint8_t* taken = nullptr;    // m
  uint32_t* counters = nullptr; // CTR_COUNT (device)
};

struct LevelResult {
  uint32_t remaining = 0;
  uint32_t num_nodes = 0;
  uint32_t md_rounds = 0;
  // the key prefixes of the level's nodes, ascending (device; only when the step compacted survivors: for the next
  // level's ActiveSet::parent_prefix).  Valid until the step after the next one on this context.
  const uint64_t* node_prefix = nullptr;
};

// What the host decides for the nodes of one level (tiler_rules: the terminal / re-root tests of tile_node,
// TilingAlgorithms.cpp:408-444; off for a bare sample_points call).
LevelPlan make_plan(int level, int sampler, uint64_t max_points, float spacing_at_root, uint32_t max_depth,
                    const double bmin[3], const double bmax[3], bool force_sample, bool tiler_rules);
int alloc_level_buffers(swz_ctx* c, uint32_t m, LevelBuffers* lb);
// required_morton_index_depth -- core/tiling/Sampling.cpp:29-62, for a root node with this x extent and max_spacing
int required_depth_host(int sampler, int node_level, double root_extent_x, float root_max_spacing);
// Samples every node of the level.  When okey/oidx are given the survivors are compacted into them and level_out
// receives plan.level for the taken points; otherwise only lb.taken is produced.
int level_step(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
               int8_t* level_out, uint64_t* okey, uint32_t* oidx, LevelResult* res);
// estimate_start_node_level_in_octree (TilingAlgorithms.cpp:1473-1535) of a sorted batch
int fast_start_level(swz_ctx* c, const uint64_t* d_keys_sorted, uint32_t n, uint32_t concurrency, int* start_level);
// the same in two steps for a batch that is spread over several GPUs: counts per 6-octant prefix (2^18, host), summed by
// the driver, then the estimate
int fast_prefix_counts(swz_ctx* c, const uint64_t* d_keys_sorted, uint32_t n, uint32_t* counts_host);
int fast_start_level_from_counts(const uint64_t* counts, uint32_t concurrency);

// MIN_DISTANCE for one level; fills lb.taken for the points of MODE_SAMPLE nodes (take-all points
// are flagged by the caller).  rounds_out accumulates the dependency rounds executed.
int min_distance_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                       const LevelBuffers& lb, uint32_t num_nodes, uint32_t sample_nodes,
                       uint32_t sample_points, uint32_t* rounds_out);

// SWZ_FLAG_MIN_DISTANCE_PROPERTY: coloured cell phases instead of the exact Morton-order greedy (swz_mdprop.hip)
int min_distance_property_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                                const LevelBuffers& lb, uint32_t num_nodes, uint32_t sample_nodes,
                                uint32_t sample_points, uint32_t* phases_out);

// ---- the MIN_DISTANCE root of a batch sharded over several GPUs of ONE process (swz_group)
// Every shard sweeps the root cells of its own octants at the same time.  Cells only depend on EARLIER adjacent cells,
// i.e. on cells of the same or of a lower shard: a cell at the face of a lower octant reads the records, key coordinates
// and state bytes of that shard's adjacent cells through peer access (the halo), as far as that shard's completed rounds
// have published them, and polls when it has to wait.  Nothing else is exchanged and no shard waits for another's whole
// root.  What a shard publishes about its root level:
struct MdPeerView {
  const uint4* rec = nullptr;        // cell records, [cell][2][rg]
  const uint64_t* qpos = nullptr;    // key coordinates of its active points
  const uint8_t* state = nullptr;
  const float4* ovf = nullptr;
  const uint32_t* gridmap = nullptr; // [cell code of the root node] -> cell
  const uint32_t* round_word = nullptr;  // the round its sweep is in: records stamped with an earlier round are complete
  const uint32_t* perm = nullptr;    // exact positions of its points: xyz[3 * perm[aidx ? aidx[i] : i]]
  const double* xyz = nullptr;
  const uint32_t* aidx = nullptr;    // null: the active points are the sorted points (the root of a single batch); a tiler's root
                                     // level -- batch + cached root file, merged -- has an index into its working arrays
  uint32_t ncells = 0, rg = 0, cell_shift = 0;
  uint32_t npoints = 0;              // points of its root level (the readers size their round limit by the lower shards' work too)
  int status = 0;                    // SWZ_OK, or why this shard cannot take part
  int entered = 0;                   // the shard's sweep has met the others at the barrier (else its driver does so for it)
};
struct MdShardRoot {
  int shard = 0, shards = 1;
  MdPeerView* views = nullptr;       // [shards], shared by the group's shards
  void (*barrier)(void*) = nullptr;  // all shards of the group meet
  void* barrier_arg = nullptr;
};

// ---- MIN_DISTANCE on key coordinates (swz_mdkeys.hip) ----------------------------------------------------------
// The key of a point is its position quantised to 2^-21 of the (cubic) bounds: the integer coordinates of two points
// bound their distance to +-sqrt(3) key cells, so a compare against the spacing is decided on the keys alone unless the
// integer distance lies within that band around it; those pairs -- a few in ten thousand at the root, a few per cent of
// the near pairs at level 3 -- are evaluated on the exact positions with the reference's arithmetic.  The result is
// therefore the exact one.  KeyMetric holds the thresholds in key cells.
struct KeyMetric {
  bool ok = false;     // the level can be decided on keys (cubic bounds, spacing of at least key_min_cells key cells)
  double T = 0.0;      // spacing in key cells
  float f_lo = 0.f;    // float squared integer distance <  f_lo: closer than the spacing for sure
  float f_hi = 0.f;    //                                >= f_hi: at least the spacing apart for sure
};
KeyMetric key_metric(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp);
// The caller's index of every active point: perm[aidx[i]] in one array (a streaming pass: aidx ascends), so that the exact
// compare of a pair costs two dependent loads per point instead of three.  At the root this is perm itself.
int key_point_ids(swz_ctx* c, const ActiveSet& as, const SortedPoints& sp, const uint32_t** ids);
// true when min_distance_level will not need sp.X / sp.Y / sp.Z for this level
bool min_distance_level_uses_keys(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp);
// no sampler of this level reads sp.X / Y / Z: RANDOM_GRID never does, the others decide on key coordinates and look up
// sp.xyz through sp.perm (swz_mdkeys.hip, grid_argmin_keys_kernel)
bool level_decides_on_keys(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp);
// Frontier sweep on key coordinates for a dense level; *used = false when the level does not qualify.
// cl: cell levels below the node chosen by the caller; typical_pop: points-weighted mean cell population.
int min_distance_keys_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                            const LevelBuffers& lb, uint32_t num_nodes, uint32_t sample_nodes, uint32_t sample_points,
                            const uint32_t* snode_of, int cl, double typical_pop, uint32_t* rounds_out, bool* used,
                            const MdShardRoot* shard_root = nullptr);

// Property mode on key coordinates in data-parallel rounds (swz_mdrounds.hip): candidates per cell, winners by a hashed
// priority, a kill pass; *used = false when the level does not qualify (then nothing has been decided).
int min_distance_rounds_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                              uint32_t num_nodes, uint32_t sample_nodes, uint32_t sample_points, const uint32_t* snode_of, uint32_t* rounds_out,
                              bool* used);

// Thread-per-point variant for sparse levels (swz_mdsparse.hip); *used = false when the level does not
// qualify.  snode_of: node -> index among the sampled nodes; occupied[cl]: occupied cells at cell level cl.
int min_distance_sparse_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                              const LevelBuffers& lb, const uint32_t* snode_of, bool all_sampled, uint32_t num_nodes,
                              uint32_t sample_nodes, uint32_t sample_points, const uint32_t occupied[12], uint32_t* rounds_out,
                              bool* used);

// The same set by blocks of 8^3 cells staged in LDS, blocks in Morton order, decisions in the same launch
// (swz_mdblock.hip, round 6); *done = false: the level does not qualify or a block did not fit, nothing is lost.
int min_distance_block_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                             const uint32_t* snode_of, bool all_sampled, uint32_t num_nodes, uint32_t sample_nodes,
                             uint32_t sample_points, const uint32_t occupied[12], const KeyMetric& km, bool* done);

}  // namespace swz
