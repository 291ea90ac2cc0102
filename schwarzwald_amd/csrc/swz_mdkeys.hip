// swz_mdkeys.hip -- MIN_DISTANCE frontier sweep on KEY COORDINATES: the exact greedy set without moving positions.
//
// Reference: PoissonDiskSampling::sample_points (core/tiling/Sampling.h:421-471) + SparseGrid::add
// (core/datastructures/SparseGrid.cpp:116-146) + GridCell::isDistant (GridCell.cpp:43-58); the result is the
// lexicographically-first maximal independent set in Morton order (see swz_mindist.hip for the derivation and the
// frontier rules (R) / (A), which are the same here).
//
// What is different from swz_mindist.hip:
//  * A Morton key IS the position, quantised to 2^-21 of the (cubic) bounds.  For two points with integer key
//    coordinates i, j the true distance in key cells lies within sqrt(3) of |i - j| (each coordinate is somewhere in its
//    cell), so "closer than the spacing?" is decided on the keys unless |i - j| falls into a band of +-1.75 cells around
//    the spacing (14 530 cells at the root for spacing = diagonal / 250).  Only those pairs are evaluated on the exact
//    positions, with the reference's arithmetic, through the sort's permutation.  Blocker scans need no exact answer at
//    all: "possibly closer" keeps a cell waiting, which is always allowed.  Nothing is gathered: a cell's points are a
//    run of the level's sorted keys (8 coalesced bytes per point instead of 24 gathered ones).
//  * A point that is known to be rejected (closer than the spacing to an accepted earlier point) gets a state byte, so a
//    cell tests each of its points against the accepted points of its neighbourhood once -- not again on every
//    activation -- and blocker scans skip dead points without testing them.
//  * ONE launch per round.  A cell's record (frontier, accepted points) is double buffered and stamped with the round
//    that wrote it: an activation reads, of every adjacent cell, the newer record written BEFORE this round (stable: its
//    owner writes the other one) and writes its own new record itself -- no second launch that publishes.  Sleeping and
//    waking without that launch: a cell that stalls writes {point it waits for, round} into the blocking cell's slot for
//    its direction and queues itself once more ("confirm": one round later the blocker's record of THIS round is stable
//    and says whether the point has been passed meanwhile).  A cell whose frontier moves claims the slots it has passed
//    with a compare-and-swap and queues their owners -- except slots stamped with the current round, whose owners
//    confirm themselves.  Whoever wins the compare-and-swap activates the sleeper, so a cell is never queued twice.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

constexpr uint32_t QNONE = 0xFFFFFFFFu;
constexpr unsigned long long QEMPTY = ~0ull;
constexpr uint32_t MQ_WOKEN = 0x80000000u;  // queue entry: activated by a claimed slot (or never slept): no confirm step
constexpr int MQ_LIST_CAP = 128;            // accepted points of the neighbourhood in LDS (window)
constexpr int MQ_FRESH_CAP = 32;            // points a cell may accept per activation
#ifndef MQ_MINW4
#define MQ_MINW4 4
#endif
#ifndef MQ_MINW1
#define MQ_MINW1 6
#endif
constexpr uint32_t MQ_FIRST_ROUND = 2;      // record buffers start with stamps 0 and 1
constexpr uint32_t MQ_ROUND_DONE = 0xFFFFFFF0u;    // a shard's round word after its sweep: every record is complete
constexpr uint32_t MQ_ROUND_FAILED = 0xFFFFFFF1u;  // ... after it has given up: whoever waits for its cells gives up too

enum : uint8_t { QS_OPEN = 0, QS_TAKEN = 1, QS_DEAD = 2 };

// ----------------------------------------------------------------------------- thresholds in key cells
KeyMetric key_metric(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp) {
  KeyMetric k;
  if (!sp.xyz || !sp.perm) return k;
  if (const char* e = c->opt("SWZ_MD_KEYS"))
    if (atoi(e) == 0) return k;
  const double ex = plan.root.maxx - plan.root.minx, ey = plan.root.maxy - plan.root.miny, ez = plan.root.maxz - plan.root.minz;
  if (!(ex > 0.0) || ex != ey || ey != ez) return k;  // one key cell must be a cube (the Tiler's bounds are)
  const double cell = ex / 2097152.0;
  const double T = std::sqrt(plan.sq_spacing) / cell;
  double tmin = 64.0;  // below that most near pairs fall into the band (levels >= 7 at spacing = diagonal / 250)
  if (const char* e = c->opt("SWZ_MD_KEYS_MIN_CELLS")) tmin = atof(e);
  if (!(T >= tmin) || !(T < 4.0e6)) return k;
  // A coordinate u = (p - min) * scale (calculate_morton_index, OctreeAlgorithms.h:64-87) has key coordinate
  // i = min(trunc(fl(u)), 2^21 - 1): u is within [i, i + 1] up to the rounding of fl (1e-9 cells).  Per axis
  // |du - di| <= 1 + 2e-9, so | |du| - |di| | <= sqrt(3) (1 + 2e-9) < 1.7321.  The squared integer distance is evaluated
  // in float (exact differences, three roundings: relative 3 * 2^-24, i.e. T * 2^-23 cells of distance near the
  // spacing).  Band: 1.75 + T * 2^-20 cells -- the reference's own rounding (1e-16 relative) disappears in the slack.
  double band = 1.75 + T * 0x1.0p-20;
  if (const char* e = c->opt("SWZ_MD_KEYS_BAND")) band += atof(e);  // tests: a wide band sends many / all pairs to the exact path
  const double lo = T - band, hi = T + band;
  float f_lo = 0.f;
  if (lo > 0.0) {
    f_lo = (float)(lo * lo);
    while ((double)f_lo > lo * lo) f_lo = std::nextafterf(f_lo, 0.f);
    f_lo = std::nextafterf(f_lo, 0.f);
  }
  float f_hi = (float)(hi * hi);
  while ((double)f_hi < hi * hi) f_hi = std::nextafterf(f_hi, INFINITY);
  f_hi = std::nextafterf(f_hi, INFINITY);
  k.T = T;
  k.f_lo = f_lo;
  k.f_hi = f_hi;
  k.ok = true;
  return k;
}

// (with ghosts in front -- a sharded batch --: ghost g keeps its index into the ghost array, local point p becomes
// ghosts + p: one id space, see SpArgs)
__global__ __launch_bounds__(256) void mq_point_ids_kernel(const uint32_t* __restrict__ aidx, const uint32_t* __restrict__ perm, uint32_t m,
                                                           uint32_t ghosts, uint32_t* __restrict__ ids) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t s = aidx ? aidx[i] : i;
  ids[i] = s < ghosts ? perm[s] : ghosts + perm[s];
}
int key_point_ids(swz_ctx* c, const ActiveSet& as, const SortedPoints& sp, const uint32_t** ids) {
  if (!as.aidx && !sp.ghosts) {
    *ids = sp.perm;
    return SWZ_OK;
  }
  uint32_t* d = nullptr;
  SWZ_TRY(c->get("md_qids", (size_t)as.m, &d));
  hipLaunchKernelGGL(mq_point_ids_kernel, dim3(div_up(as.m, 256)), dim3(256), 0, c->stream, as.aidx, sp.perm, as.m, sp.ghosts, d);
  SWZ_LAUNCH_CHECK(c);
  *ids = d;
  return SWZ_OK;
}

// ----------------------------------------------------------------------------- device helpers
// the root of a batch sharded over the GPUs of one process: what the lower shards publish (swz_level.h, MdPeerView)
constexpr uint32_t MQ_PEER_SHIFT = 28;          // neighbour id = (shard + 1) << 28 | cell for a cell of another shard
constexpr uint32_t MQ_ID_MASK = (1u << MQ_PEER_SHIFT) - 1u;
struct MqPeers {
  const uint4* rec[8];
  const uint64_t* qpos[8];
  const uint8_t* state[8];
  const float4* ovf[8];
  const uint32_t* gridmap[8];
  const uint32_t* round_word[8];
  const uint32_t* perm[8];
  const double* xyz[8];
  const uint32_t* aidx[8];
  uint32_t shard, shards;
};

struct MqArgs {
  const uint64_t* akey;
  const uint32_t* aidx;
  uint32_t m;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* nstart;
  const double* xyz;      // exact positions: point perm[aidx ? aidx[i] : i] of the caller's array (pairs inside the band are
                          // rare enough on dense levels that the three dependent loads do not matter)
  const uint32_t* perm;
  const double* gxyz;     // a sharded batch: the first ng sorted positions are ghosts, their perm entries index gxyz
  uint32_t ng;
  uint8_t* taken;
  uint32_t* counters;
  uint64_t* qpos;         // [m] key coordinates x | y << 21 | z << 42 of the active points
  uint8_t* state;         // [m] QS_*
  uint2* cinfo;           // [cell] {start, end}
  uint32_t* crel;         // [cell] cell code inside its node (build time)
  uint32_t* csnode;       // [cell] index of its node among the sampled nodes
  uint32_t* gridmap;      // [sample node][cell code] -> cell (build time)
  uint32_t* qnbr;         // [cell][32]: adjacent cell in direction k = (dx+1)*9 + (dy+1)*3 + (dz+1), 13 = the cell itself
  uint4* rec;             // [cell][2][rg]: header {frontier, accepted, stamp, end} + the first rg-1 accepted points
  float4* ovf;            // accepted point j >= rg-1 of the cell ending at `end`: ovf[end - 1 - (j - (rg-1))]
  unsigned long long* slot;  // [cell][32]: {round written << 32 | point waited for} of the sleeper in direction k
  uint4* qst;             // [cell]: {stalled candidate or NONE, blocker direction, point waited for, round of the stall}
  uint32_t* queue[2];     // [nseg][segcap]: the round's cells, in nseg segments so that no single counter takes every push
  uint32_t* qctr;         // [3 rotating rounds][nseg] segment fill counters, one per 128-byte line
  uint32_t* qtotal;       // [3]: entries the round started with (what the host polls: 0 = the level is done)
  uint32_t* qhead;        // [3 rotating rounds][nseg] tickets: the next entry of the segment nobody has taken yet
  uint32_t nseg, nseg_shift, segcap;
  const uint32_t* snode_of;
  uint32_t cell_shift;    // key >> cell_shift = node prefix + cell code
  uint64_t cells_per_node;
  uint32_t cell_bits;     // cell_shift / 3: a cell is 2^cell_bits key cells wide
  uint32_t cell_levels;   // octree levels between node and cell
  uint32_t rg, rg2_shift; // granules per record buffer (4 or 8); log2(2 * rg)
  float f_lo, f_hi;
  double sq_spacing;
  uint32_t patient;
  float lazy_frac;
  uint32_t all_sampled;
  uint32_t group, groups;
  uint32_t no_dead_test;  // debugging / tests: blocker scans do not test for dead points
  uint32_t stats;         // SWZ_DEBUG: count activations by kind in counters[CTR_DBG_HIST ...]
  uint32_t chain;         // cells a wavefront may run one after the other in a launch, each woken by the one before (1: none)
  // dense cells (thousands of points: the blobs and sheets of a clustered cloud), levels of large cells only -- see
  // mq_dense_reject_kernel
  uint32_t dense_min;     // a cell with at least this many points is dense (0: the level has none / the feature is off)
  uint8_t* dirty;         // [cell] set when the cell or an adjacent one has published accepted points since its last reject pass
  uint32_t* bulk_round;   // [cell] round of the cell's last reject pass + 1
  const uint32_t* dlist;  // the dense cells
  uint32_t* round_word;   // sharded root: the round this shard's sweep is in, for the shards that read its records
  MqPeers peers;
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t qb_u32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float qb_f32(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ unsigned long long qb_u64(unsigned long long v, int src) {
  const uint32_t lo = qb_u32((uint32_t)v, src), hi = qb_u32((uint32_t)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}
template <typename Op>
__device__ __forceinline__ uint32_t mq_wave_scan(uint32_t v, Op op, uint32_t identity) {
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31
  return v;
}
struct MqAdd {
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; }
};
struct MqMax {
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};
struct MqMin {
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a < b ? a : b; }
};

__device__ __forceinline__ uint64_t mq_pack(uint64_t key) {
  // octant = x << 2 | y << 1 | z (MortonIndex.h:62-79): bit 3j+2 of the key is bit j of x
  uint32_t x, y, z;
  key_coords_u32(key, x, y, z);
  return (uint64_t)x | ((uint64_t)y << 21) | ((uint64_t)z << 42);
}
__device__ __forceinline__ void mq_unpack(uint64_t v, float& x, float& y, float& z) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  x = (float)(lo & 0x1FFFFFu);
  y = (float)(((lo >> 21) | (hi << 11)) & 0x1FFFFFu);
  z = (float)(hi >> 10);
}
// squared distance of two points in key cells: the differences of integers below 2^21 are exact in float, the three
// roundings of the sum are part of the band (key_metric)
__device__ __forceinline__ float mq_d2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
// the reference's compare on the exact positions (GridCell.cpp:52) for active points i and j
__device__ __forceinline__ bool mq_exact_near(const MqArgs& a, uint32_t i, uint32_t j) {
  const double* p = sorted_point_xyz(a.xyz, a.perm, a.gxyz, a.ng, a.aidx ? a.aidx[i] : i);
  const double* q = sorted_point_xyz(a.xyz, a.perm, a.gxyz, a.ng, a.aidx ? a.aidx[j] : j);
  return sq_dist(p[0], p[1], p[2], q[0], q[1], q[2]) < a.sq_spacing;
}

#ifdef SWZ_MQ_STATS
#define MQ_T(var) const uint64_t var = wall_clock64()
#define MQ_TACC(idx, t0, t1) do { if (l == 0 && (c & 31u) == 0u) atomicAdd(&a.counters[CTR_DBG_HIST + 8 + (idx)], (uint32_t)((t1) - (t0))); } while (0)
#else
#define MQ_T(var) do { } while (0)
#define MQ_TACC(idx, t0, t1) do { } while (0)
#endif
#define MQ_STAT(idx) do { if (a.stats && l == 0) atomicAdd(&a.counters[CTR_DBG_HIST + (idx)], 1u); } while (0)

// Loads that see what another GPU's kernel has written back by now (nothing cached on this side): records and overflow
// entries of a lower shard's cells.  Two 8-byte halves; tearing is caught by bracketing the reads with that shard's
// round word (see mq_activate).
__device__ __forceinline__ uint4 mq_ld_sys(const uint4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
// ... and what another wavefront of THIS GPU -- or this one, earlier in the launch -- has written: past the CU's L1
__device__ __forceinline__ uint4 mq_ld_agent(const uint4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ uint32_t mq_ld_sys(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// ... for a local point i and an accepted point j of shard `tag - 1` (0: this shard)
__device__ __forceinline__ bool mq_exact_near_peer(const MqArgs& a, uint32_t i, uint32_t j, uint32_t tag) {
  const double* p = sorted_point_xyz(a.xyz, a.perm, a.gxyz, a.ng, a.aidx ? a.aidx[i] : i);
  const double* q = tag ? a.peers.xyz[tag - 1u] + (size_t)a.peers.perm[tag - 1u][a.peers.aidx[tag - 1u] ? a.peers.aidx[tag - 1u][j] : j] * 3
                        : sorted_point_xyz(a.xyz, a.perm, a.gxyz, a.ng, a.aidx ? a.aidx[j] : j);
  return sq_dist(p[0], p[1], p[2], q[0], q[1], q[2]) < a.sq_spacing;
}

struct MqLds {
  uint4* stage;   // [27][2][rg]: both record buffers of the 27 cells of the neighbourhood
  float4* list;   // [MQ_LIST_CAP]: accepted points of the earlier adjacent cells and of this cell (window)
  float4* fresh;  // [MQ_FRESH_CAP]: accepted in this activation
  uint8_t* tag;   // [MQ_LIST_CAP]: sharded root: which shard a list entry comes from (0 = this one)
  uint32_t* trust;  // [MQ_CHAIN]: the cells this wavefront has run earlier in the launch, one after the other (see mq_sweep_kernel)
  uint32_t* sid;    // [32]: the adjacent cells whose records are staged (the earlier ones and the cell itself: at most mq_staged()), by rank
};
constexpr uint32_t MQ_CHAIN = 8;
// Records staged per activation.  A cell has at most 19 adjacent cells before it in Morton order: the order of a cell and its
// neighbour is decided by the axis whose coordinate changes at the highest bit, a step of -1 changes a higher bit than a step of
// +1 only along axes where the coordinate is even, so the worst case is the cell with three even coordinates, where every offset
// with a -1 in it (27 - 8) leads to an earlier cell.  Plus the cell itself.  Records of other shards are staged whatever their
// place in the order: all 27 there.
__host__ __device__ constexpr uint32_t mq_staged(bool peers) { return peers ? 27u : 20u; }
static inline size_t mq_lds_bytes(uint32_t rg, bool peers) {
  return (size_t)(mq_staged(peers) * 2u * rg + MQ_LIST_CAP + MQ_FRESH_CAP) * 16u + MQ_LIST_CAP + MQ_CHAIN * 4u + 128u;
}

__device__ __forceinline__ bool mq_is_head(const MqArgs& a, uint32_t i) {
  if (!a.all_sampled && a.nmode[a.nid[i]] != MODE_SAMPLE) return false;
  return i == 0 || ((a.akey[i] >> a.cell_shift) != (a.akey[i - 1] >> a.cell_shift));
}

// ----------------------------------------------------------------------------- build
__global__ __launch_bounds__(256) void mq_pack_kernel(const uint64_t* __restrict__ akey, uint32_t m, uint64_t* __restrict__ qpos,
                                                      uint8_t* __restrict__ state) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  qpos[i] = mq_pack(akey[i]);
  state[i] = QS_OPEN;
}

struct MqHeadF {
  MqArgs a;
  __device__ uint32_t operator()(uint32_t i) const { return mq_is_head(a, i) ? 1u : 0u; }
};
struct MqCellBuildG {
  MqArgs a;
  __device__ void operator()(uint32_t i, uint32_t c, uint32_t head) const {
    const uint64_t key = a.akey[i];  // (the scan's head functor has just read it: a cache hit)
    a.qpos[i] = mq_pack(key);
    a.state[i] = QS_OPEN;
    if (!head) return;
    a.cinfo[c].x = i;
    a.crel[c] = (uint32_t)((key >> a.cell_shift) & (a.cells_per_node - 1ull));
    a.csnode[c] = a.all_sampled ? a.nid[i] : a.snode_of[a.nid[i]];
  }
};

__global__ __launch_bounds__(256) void mq_cell_end_kernel(MqArgs a, uint32_t ncells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t s = a.cinfo[c].x;
  const uint32_t node_end = a.nstart[a.nid[s] + 1];
  const uint32_t next = (c + 1 < ncells) ? a.cinfo[c + 1].x : a.m;
  const uint32_t e = next < node_end ? next : node_end;
  a.cinfo[c].y = e;
  a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + a.crel[c]] = c;
  uint4* r = a.rec + ((size_t)c << a.rg2_shift);
  r[0] = make_uint4(s, 0u, 0u, e);
  r[a.rg] = make_uint4(s, 0u, 1u, e);
  a.qst[c] = make_uint4(QNONE, 0u, 0u, 0u);
}

// append `value` of every lane with want == true to the queue: one atomic per wavefront
__device__ __forceinline__ void mq_wave_push(bool want, uint32_t value, uint32_t* qout, uint32_t* cout) {
  const uint64_t m = __ballot(want);
  if (!m) return;
  const int leader = __ffsll((unsigned long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane_id() == leader) base = atomicAdd(cout, (uint32_t)__popcll(m));
  base = __shfl(base, leader, WAVE);
  if (want) qout[base + (uint32_t)__popcll(m & lanemask_lt())] = value;
}

// where a level starts: the first round's queue of every node group
struct MqStart {
  uint32_t* queue[8];
  uint32_t* qctr[8];
  uint32_t groups;
  uint32_t lazy;
};

// The 26 adjacent cells by direction (codes by arithmetic on the dilated coordinates, as md_nbr_build_kernel does), 32
// lanes per cell -- and the start of the level, from the same lanes: not lazy, every cell is queued; lazy, only the
// cells without an earlier adjacent cell are, and every other cell sleeps on its latest earlier neighbour until that one
// has decided the given fraction of its points (sleeping on a cell that is not the real blocker is always safe: the cell
// looks again when it wakes up).
__global__ __launch_bounds__(256) void mq_nbr_build_kernel(MqArgs a, uint32_t ncells, MqStart st) {
  for (uint64_t cbase = (uint64_t)blockIdx.x * 8u; cbase < ncells; cbase += (uint64_t)gridDim.x * 8u) {
    const uint32_t c = (uint32_t)cbase + threadIdx.x / 32u;
    const uint32_t k = threadIdx.x & 31u;
    const bool in = c < ncells;
    uint32_t nb = QNONE;
    if (in && k == 13u) {
      nb = c;
    } else if (in && k < 27u) {
      const uint32_t rel = a.crel[c];
      const uint32_t all = (uint32_t)(a.cells_per_node - 1ull);
      const uint32_t mz = all & 0x09249249u, my = mz << 1, mx = mz << 2;
      const uint32_t v[3] = {rel & mx, rel & my, rel & mz};
      const uint32_t mk[3] = {mx, my, mz};
      const uint32_t d[3] = {k / 9u, (k / 3u) % 3u, k % 3u};  // 0: minus one, 1: same, 2: plus one
      bool inside = true;
      uint32_t nrel = 0;
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        uint32_t w = v[ax];
        if (d[ax] == 0u) {
          inside &= w != 0u;
          w = (w - 1u) & mk[ax];
        } else if (d[ax] == 2u) {
          inside &= w != mk[ax];
          w = ((w | ~mk[ax]) + 1u) & mk[ax];
        }
        nrel |= w;
      }
      if (inside) {
        if (a.peers.shards > 1u) {
          // the root of a sharded batch: the cell code's first octant names the owner; cells of higher shards are later
          // ones, nobody here looks at them
          const uint32_t owner = (nrel >> (3u * (a.cell_levels - 1u))) * a.peers.shards / 8u;
          if (owner == a.peers.shard) {
            nb = a.gridmap[nrel];
          } else if (owner < a.peers.shard && a.peers.gridmap[owner]) {
            const uint32_t id = a.peers.gridmap[owner][nrel];
            if (id != QNONE) nb = ((owner + 1u) << MQ_PEER_SHIFT) | id;
          }
        } else {
          nb = a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + nrel];
        }
      }
    }
    if (in) a.qnbr[(size_t)c * 32 + k] = nb;
    // ---- the start: the latest earlier adjacent cell of this shard (the maximum over the cell's 32 lanes)
    uint32_t best = 0;  // id + 1
    if (st.lazy && in && k < 27u && nb != QNONE && (a.peers.shards <= 1u || (nb >> MQ_PEER_SHIFT) == 0u) && nb < c) best = nb + 1u;
    if (st.lazy) {
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, off, 32));
      if (in && best && nb + 1u == best) {  // (the one lane that holds it: the ids of adjacent cells are distinct)
        const uint2 o = a.cinfo[best - 1u];
        const uint32_t bq = o.x + (uint32_t)((float)(o.y - 1u - o.x) * a.lazy_frac);
        a.slot[(size_t)(best - 1u) * 32 + (26u - k)] = (unsigned long long)bq;  // stamp 0: written before the first round
        a.qst[c] = make_uint4(a.cinfo[c].x, k, bq, 0u);
      }
    }
    const bool push = in && k == 13u && best == 0u;
    const uint32_t seg = (uint32_t)(cbase >> 3) & (a.nseg - 1u);  // (eight entries per step, spread evenly: a segment cannot overflow)
    if (st.groups == 1u) {
      mq_wave_push(push, c | MQ_WOKEN, st.queue[0] + (size_t)seg * a.segcap, st.qctr[0] + (size_t)seg * 32u);
    } else {
      const uint32_t g = push ? a.csnode[c] % st.groups : 0u;
      for (uint32_t gg = 0; gg < st.groups; ++gg)
        mq_wave_push(push && g == gg, c | MQ_WOKEN, st.queue[gg] + (size_t)seg * a.segcap, st.qctr[gg] + (size_t)seg * 32u);
    }
  }
}

// ----------------------------------------------------------------------------- the sweep
// First point of [qs, qe) that may keep a candidate at (cx, cy, cz) waiting: not known dead and possibly closer than the
// spacing (inside the band counts: waiting is always allowed), or NONE.  A point that is closer than the spacing for sure
// to one of the live_wn accepted points in LDS is dead as well (its cell has not looked at it since).
template <int SU>
__device__ __forceinline__ uint32_t mq_scan(const MqArgs& a, const uint64_t* __restrict__ qpos, const uint8_t* __restrict__ state,
                                            const MqLds& lds, uint32_t live_wn, uint32_t qs, uint32_t qe, float cx, float cy, float cz) {
  const uint32_t l = lane_id();
  for (uint32_t q0 = qs; q0 < qe; q0 += (uint32_t)SU * WAVE) {
    uint64_t v[SU];
    uint8_t s[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const uint32_t q = q0 + (uint32_t)u * WAVE + l;
      const bool inb = q < qe;
      v[u] = inb ? qpos[q] : 0ull;
      s[u] = inb ? state[q] : (uint8_t)QS_DEAD;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      float x, y, z;
      mq_unpack(v[u], x, y, z);
      bool hit = s[u] != QS_DEAD && mq_d2(x, y, z, cx, cy, cz) < a.f_hi;
      uint64_t hb = __ballot(hit);
      if (hb && live_wn) {
        if (hit) {
          for (uint32_t i = 0; i < live_wn; ++i) {
            const float4 e = lds.list[i];
            if (mq_d2(x, y, z, e.x, e.y, e.z) < a.f_lo) {
              hit = false;
              break;
            }
          }
        }
        hb = __ballot(hit);
      }
      if (hb) return q0 + (uint32_t)u * WAVE + (uint32_t)__ffsll((unsigned long long)hb) - 1u;
    }
  }
  return QNONE;
}

// First point of [q0, qe) that is not known dead, by the state bytes alone: 1024 per look (16 bytes per lane).  qe when
// there is none.
__device__ __forceinline__ uint32_t mq_first_open(const uint8_t* __restrict__ state, uint32_t q0, uint32_t qe) {
  const uint32_t l = lane_id();
  while (q0 < qe) {
    const uint32_t base = q0 & ~15u;
    const uint32_t at = base + l * 16u;
    uint4 sb = make_uint4(0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u);  // QS_DEAD
    if (at < qe) sb = *reinterpret_cast<const uint4*>(state + at);
    const uint32_t w[4] = {sb.x, sb.y, sb.z, sb.w};
    uint32_t first = 16u;  // first byte of this lane's 16 that is not dead and lies in [q0, qe)
#pragma unroll
    for (int q = 3; q >= 0; --q)
#pragma unroll
      for (int bt = 3; bt >= 0; --bt) {
        const uint32_t idx = at + (uint32_t)q * 4u + (uint32_t)bt;
        if (((w[q] >> (8 * bt)) & 0xFFu) != (uint32_t)QS_DEAD && idx >= q0 && idx < qe) first = (uint32_t)q * 4u + (uint32_t)bt;
      }
    const uint64_t hit = __ballot(first < 16u);
    if (hit) {
      const int hl = __ffsll((unsigned long long)hit) - 1;
      return base + (uint32_t)hl * 16u + qb_u32(first, hl);
    }
    q0 = base + 1024u;
  }
  return qe;
}

// mq_scan over the undecided points of a DENSE adjacent cell: most of them have been killed by the cell's reject passes
// (mq_dense_reject_kernel), so the walk jumps from one point that is not dead to the next on the state bytes.
template <int SU>
__device__ __forceinline__ uint32_t mq_scan_dense(const MqArgs& a, const uint64_t* __restrict__ qpos, const uint8_t* __restrict__ state,
                                                  const MqLds& lds, uint32_t live_wn, uint32_t qs, uint32_t qe, float cx, float cy, float cz) {
  for (uint32_t q0 = mq_first_open(state, qs, qe); q0 < qe;) {
    const uint32_t wend = (qe - q0) > (uint32_t)SU * WAVE ? q0 + (uint32_t)SU * WAVE : qe;
    const uint32_t hq = mq_scan<SU>(a, qpos, state, lds, live_wn, q0, wend, cx, cy, cz);
    if (hq != QNONE) return hq;
    q0 = mq_first_open(state, wend, qe);
  }
  return QNONE;
}

enum : uint32_t { QO_FINISHED = 0, QO_STALLED = 1, QO_YIELD = 2 };

// One wavefront advances one cell as far as it can.  U: chunks of 64 points held in registers at a time.
template <int U, bool PEERS>
__device__ void mq_activate(const MqArgs& a, uint32_t round, uint32_t qentry, const MqLds& lds, uint32_t* qout, uint32_t* cout, uint32_t seg0,
                            uint32_t ntrust, bool may_chain, uint32_t& chain_next) {
  chain_next = QNONE;
  const uint32_t l = lane_id();
  const bool woken = (qentry & MQ_WOKEN) != 0u;
  const uint32_t c = qentry & ~MQ_WOKEN;
  const uint32_t rg = a.rg, rg2s = a.rg2_shift, cap = rg - 1u;
  const float f_lo = a.f_lo, f_hi = a.f_hi;

  MQ_T(t_begin);
  // ---- first round trip: everything that hangs on the cell index alone
  const uint32_t nbv = a.qnbr[(size_t)c * 32 + (l & 31u)];
  const unsigned long long myslot = a.slot[(size_t)c * 32 + (l & 31u)];
  const uint2 ci = a.cinfo[c];
  const uint4 st = a.qst[c];
  uint4* myrec = a.rec + ((size_t)c << rg2s);
  const uint4 h0 = myrec[0], h1 = myrec[rg];
  const uint32_t sbuf = h1.z > h0.z ? 1u : 0u;  // the newer of this cell's own records (both are from earlier rounds)
  const uint32_t e = ci.y, P = sbuf ? h1.x : h0.x, CNT = sbuf ? h1.y : h0.y;
  if (P >= e) return;  // (a finished cell is never queued)
  // adjacent cell of lane k: on this shard, or -- the root of a sharded batch -- on a lower one
  const bool nb_have = nbv != QNONE;
  const uint32_t nb_peer = (PEERS && nb_have) ? (nbv >> MQ_PEER_SHIFT) : 0u;  // 0: this shard, p + 1: shard p
  const uint32_t nb_id = PEERS ? (nbv & MQ_ID_MASK) : nbv;
  // this cell once more into the next round's queue, unchanged (a look at another shard that came too early or too late)
  auto push_again = [&](uint32_t entry) {
    uint32_t base = 0, seg = (seg0 + c * 0x9E3779B1u + (c >> 7)) & (a.nseg - 1u);
    if (l == 0) {
      uint32_t tries = 0;
      for (; tries < a.nseg; ++tries) {
        base = atomicAdd(cout + (size_t)seg * 32u, 1u);
        if (base + 1u <= a.segcap) break;
        seg = (seg + 1u) & (a.nseg - 1u);
      }
      if (tries == a.nseg) atomicMax(&a.counters[CTR_ERROR], (uint32_t)SWZ_ERR_INTERNAL);
      else qout[(size_t)seg * a.segcap + base] = entry;
    }
  };
  bool claimed = woken;  // the next activation must not go through the confirm step again

  // ---- confirm: this cell stalled last time it ran and nobody has claimed its slot since
  if (!woken) {
    // (An entry without the WOKEN flag was queued by a cell that stalled and is good for THAT stall only.  When the cell
    // has been woken and run since by the wavefront that passed its blocking point -- in this very launch, see
    // mq_sweep_kernel --, there is no stall on record any more, or one of this round: the entry is void.)
    if (st.x == QNONE || st.w == round) return;
    const uint32_t bk = uni(st.y);
    const uint32_t Bv = qb_u32(nbv, (int)bk);
    const uint32_t Bp = PEERS ? (Bv >> MQ_PEER_SHIFT) : 0u, B = PEERS ? (Bv & MQ_ID_MASK) : Bv;
    if (PEERS && Bp) {
      // a cell of a lower shard: nobody wakes this one, it looks again every round
      const uint32_t rb = mq_ld_sys(a.peers.round_word[Bp - 1u]);
      if (rb == MQ_ROUND_FAILED) {  // that shard's sweep has failed: nobody will ever pass the point this cell waits for
        if (l == 0) atomicMax(&a.counters[CTR_ERROR], (uint32_t)SWZ_ERR_INTERNAL);
        return;
      }
      const uint4* brec = a.peers.rec[Bp - 1u] + ((size_t)B << rg2s);
      const uint4 b0 = mq_ld_sys(brec), b1 = mq_ld_sys(brec + rg);
      const uint32_t ra = mq_ld_sys(a.peers.round_word[Bp - 1u]);
      const bool use1 = b1.z < rb && (b0.z >= rb || b1.z > b0.z);
      const uint4 hb = use1 ? b1 : b0;
      const bool seen = ra == rb && (use1 || b0.z < rb);
      if (!seen || !(hb.x > st.z || hb.x >= hb.w)) {
        push_again(c);
        return;
      }
      claimed = true;
    } else {
    const uint4* brec = a.rec + ((size_t)B << rg2s);
    const uint4 b0 = brec[0], b1 = brec[rg];
    const bool use1 = b1.z < round && (b0.z >= round || b1.z > b0.z);
    const uint4 hb = use1 ? b1 : b0;
    const bool passed = hb.x > st.z || hb.x >= hb.w;
    if (!passed) {
      MQ_STAT(1);
      return;  // still asleep: the blocker wakes this cell when it gets there
    }
    const unsigned long long expect = ((unsigned long long)st.w << 32) | st.z;
    unsigned long long old = expect;
    if (l == 0) old = atomicCAS(&a.slot[(size_t)B * 32 + (26u - bk)], expect, QEMPTY);
    old = qb_u64(old, 0);
    if (old != expect) {
      MQ_STAT(2);
      return;  // the blocker got there first and has queued this cell for the next round
    }
    claimed = true;
    }
  }

  MQ_STAT(0);
  MQ_T(t_rt1);
  MQ_TACC(0, t_begin, t_rt1);
  // ---- second round trip: the records of the neighbourhood (staged in LDS), the window of own points at the frontier
  uint64_t pv[U];
  uint8_t ps[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t p = P + (uint32_t)u * WAVE + l;
    const bool inb = p < e;
    pv[u] = inb ? a.qpos[p] : 0ull;
    ps[u] = inb ? a.state[p] : (uint8_t)QS_DEAD;
  }
  // Only the earlier adjacent cells and the cell itself are looked at: their records are staged side by side, by rank.
  const bool valid = l < 27u && nb_have && (nb_peer != 0u || nb_id <= c);
  const uint32_t vmask = (uint32_t)__ballot(valid);
  const uint32_t srank = (uint32_t)__popc(vmask & ((1u << (l & 31u)) - 1u));
  const uint32_t rank13 = (uint32_t)__popc(vmask & ((1u << 13) - 1u));
  if (valid) lds.sid[srank] = nbv;
  __builtin_amdgcn_wave_barrier();
  // (records of another shard: bracketed by that shard's round word -- a record stamped before the round it is in NOW is
  // complete, and it stays untouched while that round lasts; when the round has moved on meanwhile the look is repeated)
  uint32_t r_before = round;
  if (PEERS && nb_peer && l < 27u) r_before = mq_ld_sys(a.peers.round_word[nb_peer - 1u]);
  {
    const uint32_t ngran = (uint32_t)__popc(vmask) << rg2s;
    constexpr int NT = (int)(mq_staged(PEERS) * 16u + WAVE - 1u) / (int)WAVE;  // (records of 8 granules per buffer at most)
    uint4 tmp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t g = (uint32_t)j * WAVE + l;
      tmp[j] = make_uint4(0u, 0u, 0u, 0u);
      if (g < ngran) {
        const uint32_t nb = lds.sid[g >> rg2s], part = g & ((1u << rg2s) - 1u);
        if (PEERS) {
          const uint32_t np = nb >> MQ_PEER_SHIFT, ni = nb & MQ_ID_MASK;
          if (np) tmp[j] = mq_ld_sys(a.peers.rec[np - 1u] + ((size_t)ni << rg2s) + part);
          else tmp[j] = ntrust ? mq_ld_agent(a.rec + ((size_t)ni << rg2s) + part) : a.rec[((size_t)ni << rg2s) + part];
        } else {
          // (a chained activation: the record its predecessor in the chain has just written must not come from this CU's L1)
          tmp[j] = ntrust ? mq_ld_agent(a.rec + ((size_t)nb << rg2s) + part) : a.rec[((size_t)nb << rg2s) + part];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const uint32_t g = (uint32_t)j * WAVE + l;
      if (g < ngran) lds.stage[g] = tmp[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (PEERS) {
    bool late = false;
    if (nb_peer && l < 27u) late = mq_ld_sys(a.peers.round_word[nb_peer - 1u]) != r_before;
    if (__ballot(PEERS && nb_peer && l < 27u && r_before == MQ_ROUND_FAILED)) {  // a lower shard has given up
      if (l == 0) atomicMax(&a.counters[CTR_ERROR], (uint32_t)SWZ_ERR_INTERNAL);
      return;
    }
    if (__ballot(late)) {
      push_again(claimed ? (c | MQ_WOKEN) : c);
      return;
    }
  }

  MQ_T(t_rt2);
  MQ_TACC(1, t_rt1, t_rt2);
  // lane k < 27: adjacent cell k (13: this cell).  Of its two records the newer one written before this round.
  const bool earlier = valid && l != 13u;
  uint32_t n_pos = 0, n_cnt = 0, n_end = 0, pick = 0;
  if (valid) {
    const uint4 a0 = lds.stage[srank << rg2s], a1 = lds.stage[(srank << rg2s) + rg];
    uint32_t rr = (PEERS && nb_peer) ? r_before : round;  // stamps count in the owner's rounds
    if (ntrust && !(PEERS && nb_peer)) {
      // a record this very wavefront wrote earlier in the launch is complete, whatever its stamp says to the others
      bool mine = false;
      for (uint32_t j = 0; j < ntrust; ++j) mine |= lds.trust[j] == nb_id;
      if (mine) rr = round + 1u;
    }
    pick = (a1.z < rr && (a0.z >= rr || a1.z > a0.z)) ? 1u : 0u;
    const uint4 hd = pick ? a1 : a0;
    n_pos = hd.x;
    n_cnt = hd.y;
    n_end = hd.w;
  }
  // The flattened list of accepted points holds first what is NEW to this cell: the entries of adjacent cells whose record
  // was written in or after the round this cell last ran in (its own newest record carries that round) -- it could not
  // see that record then.  The points this cell tested last time (those below `tested`) only meet the new entries.
  const uint32_t my_last = sbuf ? h1.z : h0.z;
  const bool k_new = valid && l != 13u && (nb_peer != 0u || (pick ? lds.stage[(srank << rg2s) + rg].z : lds.stage[srank << rg2s].z) >= my_last);
  const uint32_t incl_new = mq_wave_scan(k_new ? n_cnt : 0u, MqAdd{}, 0u);
  const uint32_t incl_old = mq_wave_scan(k_new ? 0u : n_cnt, MqAdd{}, 0u);
  const uint32_t Tnew = qb_u32(incl_new, WAVE - 1);
  const uint32_t T = Tnew + qb_u32(incl_old, WAVE - 1);  // accepted points of the neighbourhood, own committed ones included
  const uint32_t off = k_new ? incl_new - n_cnt : Tnew + incl_old - n_cnt;
  const uint32_t maxcnt = qb_u32(mq_wave_scan(n_cnt, MqMax{}, 0u), WAVE - 1);
  const uint32_t tested = (st.x != QNONE && my_last >= MQ_FIRST_ROUND) ? st.x : P;
  // entries [base, base + MQ_LIST_CAP) of the flattened list -> LDS (lane k copies what adjacent cell k contributes;
  // the ones beyond the record's inline capacity are fetched four at a time)
  auto fill = [&](uint32_t base) -> uint32_t {
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = 0; j < maxcnt && j < cap; ++j) {
      const uint32_t ti = off + j;
      if (j < n_cnt && ti >= base && ti < base + (uint32_t)MQ_LIST_CAP) {
        *reinterpret_cast<uint4*>(&lds.list[ti - base]) = lds.stage[(srank << rg2s) + pick * rg + 1u + j];
        if (PEERS) lds.tag[ti - base] = (uint8_t)nb_peer;
      }
    }
    const float4* ovf = a.ovf;
    if (PEERS) {
      if (nb_peer) ovf = a.peers.ovf[(nb_peer - 1u) & 7u];
    }
    for (uint32_t j0 = cap; j0 < maxcnt; j0 += 2u) {  // (two loads in flight; no arrays: they would live in scratch)
      const uint32_t ja = j0, jb = j0 + 1u, ta = off + ja, tb = off + jb;
      const bool ma = ja < n_cnt && ta >= base && ta < base + (uint32_t)MQ_LIST_CAP;
      const bool mb = jb < n_cnt && tb >= base && tb < base + (uint32_t)MQ_LIST_CAP;
      uint4 sa = make_uint4(0u, 0u, 0u, 0u), sb = sa;
      if (PEERS && nb_peer) {
        if (ma) sa = mq_ld_sys(reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (ja - cap))));
        if (mb) sb = mq_ld_sys(reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (jb - cap))));
      } else if (ntrust) {
        if (ma) sa = mq_ld_agent(reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (ja - cap))));
        if (mb) sb = mq_ld_agent(reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (jb - cap))));
      } else {
        if (ma) sa = *reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (ja - cap)));
        if (mb) sb = *reinterpret_cast<const uint4*>(ovf + (n_end - 1u - (jb - cap)));
      }
      if (ma) *reinterpret_cast<uint4*>(&lds.list[ta - base]) = sa;
      if (mb) *reinterpret_cast<uint4*>(&lds.list[tb - base]) = sb;
      if (PEERS) {
        if (ma) lds.tag[ta - base] = (uint8_t)nb_peer;
        if (mb) lds.tag[tb - base] = (uint8_t)nb_peer;
      }
    }
    __builtin_amdgcn_wave_barrier();
    return (T - base) < (uint32_t)MQ_LIST_CAP ? (T - base) : (uint32_t)MQ_LIST_CAP;
  };
  const uint32_t wn0 = T ? fill(0u) : 0u;
  uint32_t resident = 0;  // first entry of the list window LDS holds
  const uint32_t live_wn = (T <= (uint32_t)MQ_LIST_CAP && !a.no_dead_test) ? wn0 : 0u;  // the whole list is resident: scans can tell dead points

  MQ_T(t_fill);
  MQ_TACC(2, t_rt2, t_fill);
  uint32_t fresh = 0;
  uint32_t out_pos = e, out_status = QO_FINISHED, b_k = 0, b_q = 0;
  uint32_t tested_now = P;  // the end of the last window whose points have met the whole list
  const uint32_t S = 1u << a.cell_bits;

  for (uint32_t W0 = P; W0 < e; W0 += (uint32_t)U * WAVE) {
    float fx[U], fy[U], fz[U];
    uint32_t alive = 0;  // bit u: this lane's point of chunk u is open (neither decided nor known dead)
    if (W0 != P) {
      // Dead stretches are skipped on the state bytes alone, 1024 points per look (a cell of thousands of points -- a
      // dense blob of a clustered cloud -- is mostly dead after its first pass; window by window that was a memory round
      // trip per 64 * U points).
      while (W0 < e) {
        const uint32_t base = W0 & ~15u;
        const uint32_t at = base + l * 16u;
        uint4 sb = make_uint4(0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u);  // QS_DEAD
        if (at < e) sb = *reinterpret_cast<const uint4*>(a.state + at);
        const uint32_t w[4] = {sb.x, sb.y, sb.z, sb.w};
        uint32_t first = 16u;  // first open byte of this lane's 16 that lies in [W0, e)
#pragma unroll
        for (int q = 3; q >= 0; --q)
#pragma unroll
          for (int bt = 3; bt >= 0; --bt) {
            const uint32_t idx = at + (uint32_t)q * 4u + (uint32_t)bt;
            if (((w[q] >> (8 * bt)) & 0xFFu) == (uint32_t)QS_OPEN && idx >= W0 && idx < e) first = (uint32_t)q * 4u + (uint32_t)bt;
          }
        const uint64_t hit = __ballot(first < 16u);
        if (hit) {
          const int hl = __ffsll((unsigned long long)hit) - 1;
          W0 = base + (uint32_t)hl * 16u + qb_u32(first, hl);
          break;
        }
        W0 = base + 1024u;
      }
      if (W0 >= e) break;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = W0 + (uint32_t)u * WAVE + l;
        const bool inb = p < e;
        pv[u] = inb ? a.qpos[p] : 0ull;
        ps[u] = inb ? a.state[p] : (uint8_t)QS_DEAD;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      mq_unpack(pv[u], fx[u], fy[u], fz[u]);
      alive |= (ps[u] == QS_OPEN ? 1u : 0u) << u;
    }
    tested_now = W0 + (uint32_t)U * WAVE;
    if (!__ballot(alive != 0u)) continue;  // a window of dead points

    // (R) against the committed accepted points of the neighbourhood, window by window, then against the points
    // accepted earlier in this activation (they matter for the windows after the first).  The kernel is bound by its
    // vector ALU work here (a root cell's first activation: 478 points x 40 accepted points), so every chunk first
    // finds the list entries that can matter to it at all -- one lane per ENTRY: is it within reach of the bounding
    // box of the chunk's open points? -- and tests only those.  64 consecutive points in Morton order fill a small
    // block, and on later activations the few open points of a cell a tiny one: most entries drop out.
    // Per chunk the smallest squared key distance to a tested entry decides: below f_lo some accepted point is closer
    // than the spacing for sure; below f_hi at least one pair lies inside the band and goes to the exact compare.
    {
      // Levels of small cells (one chunk per window) first find the entries that can matter at all -- one lane per
      // ENTRY: is it within reach of the bounding box of the open points? -- and test only those: a coarsened cell of
      // level 1 sees ~120 accepted points around it, and a point can be close to a handful.  Cells of hundreds of
      // points (several chunks per window) gain nothing from that on their first activation; they save on the later
      // ones by testing the points they have tested before against the new entries only.
      float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f, bz0 = 0.f, bz1 = 0.f;
      if (U == 1) {
        const bool al = alive & 1u;
        const uint32_t hi_id = 0xFFFFFFFFu;  // coordinates are non-negative floats: their bit patterns order like they do
        bx0 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fx[0]) : hi_id, MqMin{}, hi_id), WAVE - 1));
        by0 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fy[0]) : hi_id, MqMin{}, hi_id), WAVE - 1));
        bz0 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fz[0]) : hi_id, MqMin{}, hi_id), WAVE - 1));
        bx1 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fx[0]) : 0u, MqMax{}, 0u), WAVE - 1));
        by1 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fy[0]) : 0u, MqMax{}, 0u), WAVE - 1));
        bz1 = __uint_as_float(qb_u32(mq_wave_scan(al ? __float_as_uint(fz[0]) : 0u, MqMax{}, 0u), WAVE - 1));
      }
      // chunks whose points were all tested by the last activation (uniform)
      uint32_t oldc = 0;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (W0 + (uint32_t)(u + 1) * WAVE <= tested) oldc |= 1u << u;
      uint32_t nearb = 0;
      for (uint32_t base = 0;; base += (uint32_t)MQ_LIST_CAP) {
        const bool last = base >= T;  // the extra pass: the points accepted earlier in this activation
        uint32_t wn = fresh;
        if (!last) {
          if (oldc == (1u << U) - 1u && base >= Tnew) {  // nothing in this window of the list is new to any chunk
            base = T;  // (on to the pass over the points accepted in this activation)
            continue;
          }
          if (resident != base) {
            (void)fill(base);
            resident = base;
          }
          wn = (T - base) < (uint32_t)MQ_LIST_CAP ? (T - base) : (uint32_t)MQ_LIST_CAP;
        }
        const float4* lst = last ? lds.fresh : lds.list;
        float dmin[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dmin[u] = __builtin_inff();
        if (U == 1) {
          const uint32_t lim = (last || !(oldc & 1u)) ? wn : (Tnew > base ? ((Tnew - base) < wn ? (Tnew - base) : wn) : 0u);
          for (uint32_t b0 = 0; b0 < lim; b0 += WAVE) {
            bool rel = false;  // lane i: is entry b0 + i within reach of the box?
            if (b0 + l < lim) {
              const float4 en = lst[b0 + l];
              const float gx = fmaxf(fmaxf(bx0 - en.x, en.x - bx1), 0.f);
              const float gy = fmaxf(fmaxf(by0 - en.y, en.y - by1), 0.f);
              const float gz = fmaxf(fmaxf(bz0 - en.z, en.z - bz1), 0.f);
              rel = __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx)) < f_hi;
            }
            uint64_t rm = __ballot(rel);
            while (rm) {
              const int bit = __ffsll((unsigned long long)rm) - 1;
              rm &= rm - 1ull;
              const float4 en = lst[b0 + (uint32_t)bit];
              dmin[0] = fminf(dmin[0], mq_d2(fx[0], fy[0], fz[0], en.x, en.y, en.z));
            }
          }
        } else {
          // every entry against every chunk -- only the new ones when all chunks met the others last time
          const uint32_t nnew = last ? wn : (Tnew > base ? ((Tnew - base) < wn ? (Tnew - base) : wn) : 0u);
          const uint32_t lim = oldc == (1u << U) - 1u ? nnew : wn;
          for (uint32_t i = 0; i < lim; ++i) {
            const float4 en = lst[i];
#pragma unroll
            for (int u = 0; u < U; ++u) dmin[u] = fminf(dmin[u], mq_d2(fx[u], fy[u], fz[u], en.x, en.y, en.z));
          }
        }
        uint32_t pend = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          nearb |= (dmin[u] < f_lo ? 1u : 0u) << u;
          pend |= (dmin[u] >= f_lo && dmin[u] < f_hi ? 1u : 0u) << u;
        }
        pend &= alive & ~nearb;
        if (__ballot(pend != 0u)) {  // pairs inside the band: the exact compare on the original positions
          for (uint32_t i = 0; i < wn; ++i) {
            const float4 en = lst[i];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if ((pend >> u) & 1u) {
                const float d2 = mq_d2(fx[u], fy[u], fz[u], en.x, en.y, en.z);
                if (d2 >= f_lo && d2 < f_hi &&
                    ((PEERS && !last) ? mq_exact_near_peer(a, W0 + (uint32_t)u * WAVE + l, __float_as_uint(en.w), lds.tag[i])
                                      : mq_exact_near(a, W0 + (uint32_t)u * WAVE + l, __float_as_uint(en.w)))) {
                  nearb |= 1u << u;
                  pend &= ~(1u << u);
                }
              }
            }
          }
        }
        if (last) break;
      }
      const uint32_t died = alive & nearb;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((died >> u) & 1u) a.state[W0 + (uint32_t)u * WAVE + l] = QS_DEAD;
      alive &= ~nearb;
    }

    MQ_T(t_r);
    MQ_TACC(3, t_fill, t_r);
    // (A) surviving points in order: accept, or stall on a possibly undecided earlier point
    bool stop = false;
    for (;;) {
      int cu = -1;
      uint64_t bm = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t b = __ballot((alive >> u) & 1u);
        if (cu < 0 && b) {
          cu = u;
          bm = b;
        }
      }
      if (cu < 0) break;
      const int j = __ffsll((unsigned long long)bm) - 1;
      const uint32_t cand = W0 + (uint32_t)cu * WAVE + (uint32_t)j;
      float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u == cu) {
          cx = qb_f32(fx[u], j);
          cy = qb_f32(fy[u], j);
          cz = qb_f32(fz[u], j);
        }
      }
      // earlier adjacent cells that still hold undecided points and lie within reach of the candidate (gap between its
      // key cell and the adjacent cell, per axis: a point over there differs by more than that many cells)
      uint32_t need;
      {
        const uint32_t lx = (uint32_t)cx & (S - 1u), ly = (uint32_t)cy & (S - 1u), lz = (uint32_t)cz & (S - 1u);
        const uint32_t dxk = l / 9u, dyk = (l / 3u) % 3u, dzk = l % 3u;
        const float gx = dxk == 0u ? (float)lx : (dxk == 2u ? (float)(S - 1u - lx) : 0.f);
        const float gy = dyk == 0u ? (float)ly : (dyk == 2u ? (float)(S - 1u - ly) : 0.f);
        const float gz = dzk == 0u ? (float)lz : (dzk == 2u ? (float)(S - 1u - lz) : 0.f);
        const float gap2 = __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx));
        need = (uint32_t)__ballot(earlier && n_pos < n_end && gap2 < f_hi);
      }
      bool blocked = false;
      // (Measured and dropped, both on the 100 M clustered cloud: not scanning an adjacent cell that still has thousands of
      // undecided points within reach but waiting for all of it -- root 105 -> 140 .. 180 ms: dense cells overlap, a cell
      // moves on as soon as the one point that blocks it has been passed --, and the reverse, cutting long scans short.)
      while (need && !blocked) {
        const int k = __ffs((int)need) - 1;
        need &= need - 1u;
        const uint32_t qs = qb_u32(n_pos, k), qe = qb_u32(n_end, k);
        MQ_STAT(4);
        const uint32_t kp = PEERS ? qb_u32(nb_peer, k) : 0u;
        // (Measured and dropped: a scan that reads the state bytes first and fetches the coordinates of the open points
        // only.  Where it would pay -- the dense blob of a clustered cloud -- the root went 107 -> 101 ms; on uniform data,
        // whose neighbours' points are mostly still open when they are scanned, 73 -> 113 ms.)
        uint32_t hq;
        if (U > 1 && a.dense_min && !kp && qe - qs >= a.dense_min)
          hq = mq_scan_dense<4>(a, a.qpos, a.state, lds, live_wn, qs, qe, cx, cy, cz);
        else
          hq = mq_scan<(U > 1 ? 4 : 1)>(a, kp ? a.peers.qpos[kp - 1u] : a.qpos, kp ? a.peers.state[kp - 1u] : a.state, lds, live_wn, qs, qe,
                                        cx, cy, cz);
        if (hq != QNONE) {
          blocked = true;
          b_k = (uint32_t)k;
          b_q = a.patient ? qe - 1u : hq;
        }
      }
      if (blocked) {
        MQ_STAT(3);
        out_pos = cand;
        out_status = QO_STALLED;
        stop = true;
        break;
      }
      // accepted
      if ((int)l == j) {
        a.taken[cand] = 1;
        a.state[cand] = QS_TAKEN;
        alive &= ~(1u << cu);
      }
      if (l == 0) lds.fresh[fresh] = make_float4(cx, cy, cz, __uint_as_float(cand));
      ++fresh;
      // the points it rejects (all open points of the window are later ones)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((alive >> u) & 1u) {
          const float d2 = mq_d2(fx[u], fy[u], fz[u], cx, cy, cz);
          bool nr = d2 < f_lo;
          if (!nr && d2 < f_hi) nr = mq_exact_near(a, W0 + (uint32_t)u * WAVE + l, cand);
          if (nr) {
            a.state[W0 + (uint32_t)u * WAVE + l] = QS_DEAD;
            alive &= ~(1u << u);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (fresh == (uint32_t)MQ_FRESH_CAP) {  // LDS list full: publish and go on in the next round
        uint32_t next = W0 + (uint32_t)U * WAVE;
#pragma unroll
        for (int u = U - 1; u >= 0; --u) {
          const uint64_t b = __ballot((alive >> u) & 1u);
          if (b) next = W0 + (uint32_t)u * WAVE + (uint32_t)__ffsll((unsigned long long)b) - 1u;
        }
        if (next < e) {
          out_pos = next;
          out_status = QO_YIELD;
          stop = true;
        }
        break;
      }
    }
    if (stop) break;
    // (Measured and dropped: cells of thousands of points taking their turn in slices of 1024 .. 4096 points, so that a
    // launch does not last as long as its longest activation -- the clustered root went 105 -> 120 .. 153 ms.)
    // (... or a dense cell that has accepted a point: what that point rejects in the thousands of points behind it is
    // killed by a whole workgroup between the launches, mq_dense_reject_kernel, and the walk goes on next round)
    if ((fresh == (uint32_t)MQ_FRESH_CAP || (U > 1 && fresh && a.dense_min && e - ci.x >= a.dense_min)) &&
        W0 + (uint32_t)U * WAVE < e) {  // (full exactly at the end of a window)
      out_pos = W0 + (uint32_t)U * WAVE;
      out_status = QO_YIELD;
      break;
    }
  }

  MQ_T(t_cand);
  MQ_TACC(4, t_fill, t_cand);
  // ---- publish: the new record goes into the buffer that does NOT hold the newest one
  const uint32_t wb = sbuf ^ 1u;
  const uint32_t ncnt = CNT + fresh;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = l; j < ncnt && j < cap; j += WAVE) {
    uint4 src;
    if (j < CNT) src = lds.stage[(rank13 << rg2s) + sbuf * rg + 1u + j];
    else src = *reinterpret_cast<const uint4*>(&lds.fresh[j - CNT]);
    myrec[wb * rg + 1u + j] = src;
  }
  for (uint32_t jj = l; jj < fresh; jj += WAVE) {
    const uint32_t j = CNT + jj;
    if (j >= cap) a.ovf[e - 1u - (j - cap)] = lds.fresh[jj];
  }
  const bool fin = out_pos >= e;
  if (l == 0) {
    myrec[wb * rg] = make_uint4(fin ? e : out_pos, ncnt, round, e);
    a.qst[c] = out_status == QO_STALLED ? make_uint4(tested_now, b_k, b_q, round) : make_uint4(QNONE, 0u, 0u, 0u);
  }
  if (U > 1 && a.dirty && fresh && l < 27u && nb_have && nb_peer == 0u) a.dirty[nb_id] = 1;  // (k = 13: the cell itself)
  // ---- wake the later adjacent cells that sleep on a point the frontier has passed (slots written in THIS round belong
  // to cells that confirm themselves next round); go to sleep / come back next round
  bool won = false;
  if (l < 27u && nb_have && nb_peer == 0u && nb_id > c && myslot != QEMPTY && (uint32_t)(myslot >> 32) != round &&
      (fin || (uint32_t)myslot < out_pos)) {
    won = atomicCAS(&a.slot[(size_t)c * 32 + l], myslot, QEMPTY) == myslot;
  }
  if (out_status == QO_STALLED && l == 0) {
    const uint32_t Bv = qb_u32(nbv, (int)uni(b_k));
    // (a blocker on a lower shard has no slot for this cell: it confirms itself every round until the point is passed)
    if (!(PEERS && (Bv >> MQ_PEER_SHIFT))) a.slot[(size_t)(PEERS ? (Bv & MQ_ID_MASK) : Bv) * 32 + (26u - b_k)] = ((unsigned long long)round << 32) | b_q;
  }
  uint64_t wm = __ballot(won);
  if (may_chain && wm) {
    // the first sleeper this activation has woken is run by this wavefront right away, in this launch: it will read the
    // record just written (complete: same wavefront, program order) instead of waiting a round for it
    const int first = __ffsll((unsigned long long)wm) - 1;
    chain_next = qb_u32(nb_id, first);
    wm &= wm - 1ull;
    if ((int)l == first) won = false;
  }
  const uint32_t self = fin ? 0u : 1u;
  const uint32_t total = (uint32_t)__popcll(wm) + self;
  if (total) {
    // into one segment of the next round's queue -- chosen by the cell so that the work spreads over all workgroups
    // whatever this one read -- or, that one full, into the next that has room (the segments hold twice the cells
    // between them and a round never has more entries than cells)
    uint32_t base = 0, seg = (seg0 + c * 0x9E3779B1u + (c >> 7)) & (a.nseg - 1u);
    if (l == 0) {
      uint32_t tries = 0;
      for (; tries < a.nseg; ++tries) {
        base = atomicAdd(cout + (size_t)seg * 32u, total);
        if (base + total <= a.segcap) break;
        seg = (seg + 1u) & (a.nseg - 1u);
      }
      if (tries == a.nseg) {
        atomicMax(&a.counters[CTR_ERROR], (uint32_t)SWZ_ERR_INTERNAL);
        base = QNONE;
      }
    }
    base = qb_u32(base, 0);
    seg = qb_u32(seg, 0);
    if (base != QNONE) {
      uint32_t* q = qout + (size_t)seg * a.segcap + base;
      if (won) q[(uint32_t)__popcll(wm & lanemask_lt())] = nb_id | MQ_WOKEN;
      if (self && l == 0) q[(uint32_t)__popcll(wm)] = out_status == QO_YIELD ? (c | MQ_WOKEN) : c;
    }
  }
  MQ_T(t_end);
  MQ_TACC(5, t_cand, t_end);
  MQ_TACC(6, t_begin, t_end);
#ifdef SWZ_MQ_STATS
  if (l == 0 && (c & 31u) == 0u) atomicAdd(&a.counters[CTR_DBG_HIST + 8 + 7], 1u);
#endif
}

template <int U, bool PEERS>
__global__ __launch_bounds__(WAVE, (U == 1 && !PEERS) ? MQ_MINW1 : ((U == 4 && !PEERS) ? MQ_MINW4 : 4)) void mq_sweep_kernel(MqArgs a, uint32_t round) {
  extern __shared__ uint4 mq_smem[];
  MqLds lds;
  lds.stage = mq_smem;
  lds.list = reinterpret_cast<float4*>(mq_smem + mq_staged(PEERS) * 2u * a.rg);
  lds.fresh = lds.list + MQ_LIST_CAP;
  lds.tag = reinterpret_cast<uint8_t*>(lds.fresh + MQ_FRESH_CAP);
  lds.trust = reinterpret_cast<uint32_t*>(lds.tag + MQ_LIST_CAP);
  lds.sid = lds.trust + MQ_CHAIN;
  const uint32_t r0 = round - MQ_FIRST_ROUND;
  const uint32_t ci = r0 % 3u, co = (r0 + 1u) % 3u, cz = (r0 + 2u) % 3u;
  if (blockIdx.x == 0) {  // the counters the round after the next will fill; this round's total for the host
    const uint32_t l = lane_id();
    uint32_t mine = 0;
    for (uint32_t sg = l; sg < a.nseg; sg += WAVE) {
      a.qctr[((size_t)cz * a.nseg + sg) * 32u] = 0;
      a.qhead[((size_t)cz * a.nseg + sg) * 32u] = 0;
      const uint32_t n = a.qctr[((size_t)ci * a.nseg + sg) * 32u];
      mine += n < a.segcap ? n : a.segcap;
    }
    const uint32_t tot = qb_u32(mq_wave_scan(mine, MqAdd{}, 0u), WAVE - 1);
    if (l == 0) a.qtotal[ci] = tot;
  }
  const uint32_t seg = blockIdx.x & (a.nseg - 1u);
  uint32_t nq = a.qctr[((size_t)ci * a.nseg + seg) * 32u];
  nq = nq < a.segcap ? nq : a.segcap;  // (a counter overshoots when a push found the segment full and went elsewhere)
  const uint32_t* qin = a.queue[r0 & 1u] + (size_t)seg * a.segcap;
  uint32_t* qout = a.queue[(r0 + 1u) & 1u];
  uint32_t* cout = a.qctr + (size_t)co * a.nseg * 32u;
  // The workgroups of a segment draw its entries by ticket: activations differ in cost by an order of magnitude (a
  // confirm that finds its cell still asleep, a first activation of 500 points), and with a fixed stride the launch
  // lasts as long as its unluckiest workgroup.  The next ticket is requested before the current entry is worked on.
  uint32_t* head = a.qhead + ((size_t)ci * a.nseg + seg) * 32u;
  uint32_t i = blockIdx.x >> a.nseg_shift;  // the first entries go by position: no ticket needed
  const uint32_t wgs = gridDim.x >> a.nseg_shift;
  uint32_t entry = qin[i < a.segcap ? i : 0u];  // (requested together with the segment's count, not after it)
  while (i < nq) {
    uint32_t next = 0;
    if (lane_id() == 0) next = atomicAdd(head, 1u);  // (the result is first looked at after the activation)
    // the entry, then -- one after the other -- a chain of cells each woken by the one before: a dependency that this
    // wavefront resolves itself does not cost a round
    uint32_t cur = uni(entry), ntrust = 0;
    for (;;) {
      uint32_t woke = QNONE;
      mq_activate<U, PEERS>(a, round, cur, lds, qout, cout, seg, ntrust, ntrust + 1u < a.chain, woke);
      woke = uni(woke);
      if (woke == QNONE) break;
      if (lane_id() == 0) lds.trust[ntrust] = cur & ~MQ_WOKEN;
      ++ntrust;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // what the cell has published has reached the L2 the next one reads from
      __builtin_amdgcn_wave_barrier();
      cur = woke | MQ_WOKEN;
    }
    i = wgs + qb_u32(next, 0);
    if (i < nq) entry = qin[i];
  }
}

// Sharded root: the round this shard's sweep is in, for the shards that read its records.  A launch of its own BETWEEN
// two rounds: when a reader sees round R here, every record stamped before R is complete (round R - 1 has finished and
// written back) and round R only ever writes the OLDER buffer of a cell -- a store from inside the sweep would become
// visible some time into the round, when that is no longer true.
// ---- dense cells
// A cell of thousands of points (a blob or a sheet of a clustered cloud) is walked by ONE wavefront, 256 points per
// memory round trip, and a launch lasts as long as its longest activation.  Almost all of those points are rejected by
// the handful of accepted points around them, so that part is taken out of the walk: a dense cell yields after it has
// accepted a point, every activation that publishes accepted points marks the cells around it, and between two launches
// one WORKGROUP per marked dense cell tests the cell's undecided points against the accepted points of the earlier
// adjacent cells (and its own) that are new since its last pass, and marks what they reject dead.  That is all it does
// -- a dead point is one that some published accepted earlier point rejects, which is what the state means anyway --, so
// the walk and the blocker scans of the others then jump over the killed stretches on the state bytes, 1024 points per
// look.  Pairs inside the band are left to the walk (it has the exact compare).
__global__ __launch_bounds__(256) void mq_dense_list_kernel(MqArgs a, uint32_t ncells, uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint2 ci = a.cinfo[c];
  if (ci.y - ci.x >= a.dense_min) list[atomicAdd(count, 1u)] = c;
}

constexpr uint32_t MQ_DENSE_LIST = 256;  // accepted points a reject pass looks at (the rest waits for the walk)
__global__ __launch_bounds__(256) void mq_dense_reject_kernel(MqArgs a, uint32_t round) {
  __shared__ float4 lst[MQ_DENSE_LIST];
  __shared__ uint32_t cnt;
  const uint32_t c = a.dlist[blockIdx.x];
  if (!a.dirty[c]) return;  // (the same byte for the whole workgroup; cleared below, after everybody has read it)
  const uint32_t tid = threadIdx.x;
  const uint32_t rg = a.rg, rg2s = a.rg2_shift, cap = rg - 1u;
  const uint4* myrec = a.rec + ((size_t)c << rg2s);
  const uint4 h0 = myrec[0], h1 = myrec[rg];
  const uint4 mine = h1.z > h0.z ? h1 : h0;  // (no launch is running: both records are complete, the newer one counts)
  const uint32_t P = mine.x, e = a.cinfo[c].y;
  const uint32_t since = a.bulk_round[c];
  if (tid == 0) cnt = 0;
  __syncthreads();
  if (tid == 0) {
    a.dirty[c] = 0;
    a.bulk_round[c] = round + 1u;
  }
  if (P >= e) return;
  // the accepted points of the earlier adjacent cells and of the cell itself whose record is new since the last pass
  if (tid < 27u) {
    const uint32_t nb = a.qnbr[(size_t)c * 32 + tid];
    if (nb != QNONE && nb <= c) {
      const uint4* r = a.rec + ((size_t)nb << rg2s);
      const uint4 a0 = r[0], a1 = r[rg];
      const uint32_t pick = a1.z > a0.z ? 1u : 0u;
      const uint4 hd = pick ? a1 : a0;
      if (hd.z >= since && hd.y) {
        const uint32_t at = atomicAdd(&cnt, hd.y);
        for (uint32_t j = 0; j < hd.y && at + j < MQ_DENSE_LIST; ++j) {
          uint4 en;
          if (j < cap) en = r[pick * rg + 1u + j];
          else en = *reinterpret_cast<const uint4*>(a.ovf + (hd.w - 1u - (j - cap)));
          lst[at + j] = make_float4(__uint_as_float(en.x), __uint_as_float(en.y), __uint_as_float(en.z), 0.f);
        }
      }
    }
  }
  __syncthreads();
  const uint32_t n = cnt < MQ_DENSE_LIST ? cnt : MQ_DENSE_LIST;
  if (!n) return;
  for (uint32_t p = P + tid; p < e; p += 256u) {
    if (a.state[p] != QS_OPEN) continue;
    float x, y, z;
    mq_unpack(a.qpos[p], x, y, z);
    bool dead = false;
    for (uint32_t i = 0; i < n && !dead; ++i) dead = mq_d2(x, y, z, lst[i].x, lst[i].y, lst[i].z) < a.f_lo;
    if (dead) a.state[p] = QS_DEAD;
  }
}

__global__ void mq_round_word_kernel(uint32_t* word, uint32_t value) {
  __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// after the last round: every cell must have reached its end (a protocol error would otherwise go unnoticed)
__global__ __launch_bounds__(256) void mq_verify_kernel(MqArgs a, uint32_t ncells, uint32_t* __restrict__ open_cells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  bool open = false;
  if (c < ncells) {
    const uint4* r = a.rec + ((size_t)c << a.rg2_shift);
    const uint4 h0 = r[0], h1 = r[a.rg];
    const uint4 h = h1.z > h0.z ? h1 : h0;
    open = h.x < h.w;
  }
  const uint64_t bm = __ballot(open);
  if (bm && lane_id() == 0) atomicAdd(open_cells, (uint32_t)__popcll(bm));
}

// ----------------------------------------------------------------------------- host
bool min_distance_level_uses_keys(const swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp) {
  return key_metric(c, plan, sp).ok;  // (property mode decides on keys as well: swz_mdrounds.hip, or the exact set)
}

int min_distance_keys_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                            uint32_t nnodes, uint32_t sample_nodes, uint32_t sample_points, const uint32_t* snode_of, int cl,
                            double typical_pop, uint32_t* rounds_out, bool* used, const MdShardRoot* shard_root) {
  *used = false;
  const KeyMetric km = key_metric(c, plan, sp);
  if (!km.ok) return SWZ_OK;
  const bool sharded = shard_root != nullptr && shard_root->shards > 1;
  const std::string sfx = sharded ? "_sr" : "";  // the sharded root's arrays are read by other shards until the batch ends
  const uint32_t m = as.m;
  MqArgs a{};
  a.akey = as.akey;
  a.aidx = as.aidx;
  a.m = m;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.nstart = lb.nstart;
  a.xyz = sp.xyz;
  a.perm = sp.perm;
  a.gxyz = sp.ghost_xyz;
  a.ng = sp.ghosts;
  a.taken = lb.taken;
  a.counters = lb.counters;
  a.snode_of = snode_of;
  a.cells_per_node = 1ull << (3 * cl);
  a.cell_levels = (uint32_t)cl;
  a.peers.shards = sharded ? (uint32_t)shard_root->shards : 1u;
  a.peers.shard = sharded ? (uint32_t)shard_root->shard : 0u;
  a.cell_shift = (plan.node_shift == 63u ? 63u : plan.node_shift) - 3u * (uint32_t)cl;
  a.cell_bits = a.cell_shift / 3u;
  a.f_lo = km.f_lo;
  a.f_hi = km.f_hi;
  a.sq_spacing = plan.sq_spacing;
  a.all_sampled = sample_nodes == nnodes ? 1u : 0u;
  a.no_dead_test = (c->opt("SWZ_MD_ABLATE") && (atoi(c->opt("SWZ_MD_ABLATE")) & 8)) ? 1u : 0u;
  a.stats = c->opt("SWZ_MD_STATS") ? 1u : 0u;
  // (measured at 1 B points: 2 cells per chain root / level 0 / level 1 78 / 69 / 96 -> 73 / 62 / 91 ms; longer chains make
  // launches as long as their longest chain and lose again)
  a.chain = c->opt("SWZ_MD_CHAIN") ? std::min<uint32_t>(MQ_CHAIN, (uint32_t)std::max(1, atoi(c->opt("SWZ_MD_CHAIN")))) : 2u;
  if (a.cell_bits > 21u) return SWZ_OK;

  ProfScope ps(c, "sample_min_distance", (uint64_t)sample_points * 33ull, 1);
  // cells = runs of the cell prefix inside sampled nodes
  uint32_t* d_cell_sums = nullptr;
  SWZ_TRY(fused_scan_sums(c, MqHeadF{a}, m, lb.counters + CTR_NUM_CELLS, "mqc", &d_cell_sums));
  uint32_t ncells = 0;
  SWZ_HIP(c, hipMemcpyAsync(&ncells, lb.counters + CTR_NUM_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  *used = true;
  if (ncells == 0) return SWZ_OK;
  if (ncells >= 0x7FFFFFFFu) {
    *used = false;
    return SWZ_OK;
  }
  // A cell r spacings wide ends with about 0.75 r^3 accepted points: records of 3 or 7 inline ones, the rest in the
  // overflow.  (Small records win well beyond their capacity -- level 1 at 1 B points, cells two spacings wide with ~6
  // accepted points each: 78 ms with records of 3, 84 with records of 7: less to stage, more wavefronts per CU.)
  const double r_cell = std::ldexp(1.0, (int)a.cell_bits) / km.T;
  a.rg = (0.75 * r_cell * r_cell * r_cell <= 12.0) ? 4u : 8u;
  if (const char* e = c->opt("SWZ_MD_KEYS_RG")) a.rg = atoi(e) >= 8 ? 8u : 4u;
  a.rg2_shift = a.rg == 4u ? 3u : 4u;

  SWZ_TRY(c->get(("md_qpos" + sfx).c_str(), (size_t)m, &a.qpos));
  SWZ_TRY(c->get(("md_qstate" + sfx).c_str(), (size_t)m, &a.state));
  SWZ_TRY(c->get(("md_qovf" + sfx).c_str(), (size_t)m, &a.ovf));
  SWZ_TRY(c->get(("md_qcinfo" + sfx).c_str(), (size_t)ncells, &a.cinfo));
  uint32_t* cellbuf = nullptr;
  SWZ_TRY(c->get(("md_qcells" + sfx).c_str(), (size_t)ncells * 2, &cellbuf));
  a.crel = cellbuf;
  a.csnode = cellbuf + ncells;
  SWZ_TRY(c->get(("md_qnbr" + sfx).c_str(), (size_t)ncells * 32, &a.qnbr));
  SWZ_TRY(c->get(("md_qrec" + sfx).c_str(), (size_t)ncells * 2 * a.rg, &a.rec));
  SWZ_TRY(c->get(("md_qslot" + sfx).c_str(), (size_t)ncells * 32, &a.slot));
  SWZ_TRY(c->get(("md_qst" + sfx).c_str(), (size_t)ncells, &a.qst));
  const uint64_t grid_entries = (uint64_t)sample_nodes * a.cells_per_node;
  SWZ_TRY(c->get(("md_gridmap" + sfx).c_str(), (size_t)grid_entries, &a.gridmap));
  SWZ_HIP(c, memset_large(a.gridmap, 0xFF, (size_t)grid_entries * 4, c->stream));
  SWZ_HIP(c, memset_large(a.slot, 0xFF, (size_t)ncells * 32 * sizeof(unsigned long long), c->stream));

  SWZ_TRY(fused_scan_apply(c, MqHeadF{a}, MqCellBuildG{a}, m, d_cell_sums));
  const uint32_t cb = div_up(ncells, 256);
  hipLaunchKernelGGL(mq_cell_end_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  // From the moment its views are published, a shard that leaves this function early says so in its round word: the
  // higher shards' face cells poll that word and would otherwise spin until their own limits.
  struct WordGuard {
    swz_ctx* c;
    uint32_t* word = nullptr;
    bool done = false;
    ~WordGuard() {
      if (word && !done) {
        hipLaunchKernelGGL(mq_round_word_kernel, dim3(1), dim3(1), 0, c->stream, word, MQ_ROUND_FAILED);
        (void)hipStreamSynchronize(c->stream);
      }
    }
  } word_guard{c};
  if (sharded) {
    // publish this shard's root level, meet the others, take the lower shards' views
    MdPeerView& mine = shard_root->views[shard_root->shard];
    int my_status = SWZ_OK;
    if (cl < 1 || ncells >= (1u << MQ_PEER_SHIFT) || sample_nodes != 1) my_status = SWZ_ERR_INTERNAL;
    // (a tiler's root level -- the batch merged with the cached root file -- indexes its working arrays through aidx, a
    // buffer the next level reuses while higher shards may still look up exact positions here: a copy that stays)
    const uint32_t* aidx_sr = nullptr;
    if (as.aidx) {
      uint32_t* cp = nullptr;
      SWZ_TRY(c->get("md_qaidx_sr", (size_t)m, &cp));
      SWZ_HIP(c, hipMemcpyAsync(cp, as.aidx, (size_t)m * 4, hipMemcpyDeviceToDevice, c->stream));
      aidx_sr = cp;
    }
    SWZ_TRY(c->get("md_qround_sr", (size_t)32, &a.round_word));
    hipLaunchKernelGGL(mq_round_word_kernel, dim3(1), dim3(1), 0, c->stream, a.round_word, MQ_FIRST_ROUND);
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    mine.rec = a.rec;
    mine.qpos = a.qpos;
    mine.state = a.state;
    mine.ovf = a.ovf;
    mine.gridmap = a.gridmap;
    mine.round_word = a.round_word;
    mine.perm = a.perm;
    mine.xyz = a.xyz;
    mine.aidx = aidx_sr;
    mine.ncells = ncells;
    mine.npoints = m;
    mine.rg = a.rg;
    mine.cell_shift = a.cell_shift;
    mine.status = my_status;
    mine.entered = 1;
    word_guard.word = a.round_word;
    shard_root->barrier(shard_root->barrier_arg);
    for (int p = 0; p < shard_root->shards; ++p) {
      const MdPeerView& v = shard_root->views[p];
      if (v.status != SWZ_OK || (v.ncells && (v.rg != a.rg || v.cell_shift != a.cell_shift)))
        return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE root of a sharded batch: shard " + std::to_string(p) +
                                           " cannot take part in the joint sweep (cell levels, cell count or record size differ)");
      if (p < shard_root->shard && v.ncells) {
        a.peers.rec[p] = v.rec;
        a.peers.qpos[p] = v.qpos;
        a.peers.state[p] = v.state;
        a.peers.ovf[p] = v.ovf;
        a.peers.gridmap[p] = v.gridmap;
        a.peers.round_word[p] = v.round_word;
        a.peers.perm[p] = v.perm;
        a.peers.xyz[p] = v.xyz;
        a.peers.aidx[p] = v.aidx;
      }
    }
  }
  // scheduling: the same rules as the sweep on positions (swz_mindist.hip)
  const bool many_small = ncells >= (4u << 20) && (double)sample_points / (double)ncells <= 128.0;
  a.patient = many_small ? 1u : 0u;
  if (const char* e = c->opt("SWZ_MD_PATIENT")) a.patient = (uint32_t)atoi(e);
  bool lazy = true;  // (on keys also for roots of thousands of points per cell: clustered root 99 -> 89 ms)
  if (const char* e = c->opt("SWZ_MD_LAZY")) lazy = atoi(e) != 0;
  a.lazy_frac = c->opt("SWZ_MD_LAZY_FRAC") ? (float)atof(c->opt("SWZ_MD_LAZY_FRAC")) : (many_small ? 0.5f : 0.0f);
  // (the build for large cells from a few hundred points per typical cell on: at 113 -- level 1 of the clustered cloud --
  // the small-cell build is still ahead, 48 against 55 ms; at 478 and 710 the large-cell one, 72 / 81 against 100+)
  bool big_cells = typical_pop > 256.0;
  if (const char* e = c->opt("SWZ_MD_BIG")) big_cells = atoi(e) != 0;
  uint32_t groups = 1;
  if (sample_nodes >= 2 && !big_cells) groups = 2;
  if (const char* e = c->opt("SWZ_MD_GROUPS")) groups = (uint32_t)std::max(1, std::min(8, atoi(e)));
  groups = std::min(groups, sample_nodes);
  if (sharded) groups = 1;
  // dense cells: levels of large cells in one node group (mq_dense_reject_kernel)
  uint32_t ndense = 0;
  a.dense_min = 0;
  a.dirty = nullptr;
  if (big_cells && groups == 1 && !sharded) {
    uint32_t dense_min = 4096;  // (100 M clustered points, root: 2048 -> 99 ms, 4096 -> 95, 1024 -> 101, off -> 106; a sheet alone loses 4 ms of 34 to the extra rounds)
    if (const char* e = c->opt("SWZ_MD_DENSE_MIN")) dense_min = (uint32_t)std::max(0, atoi(e));
    if (dense_min && m >= dense_min) {
      uint32_t* dl = nullptr;
      uint32_t* dcount = nullptr;
      SWZ_TRY(c->get("md_qdense", (size_t)(m / dense_min + 1u), &dl));
      SWZ_TRY(c->get("md_qdense_count", (size_t)4, &dcount));
      SWZ_HIP(c, hipMemsetAsync(dcount, 0, 4, c->stream));
      a.dense_min = dense_min;
      hipLaunchKernelGGL(mq_dense_list_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells, dl, dcount);
      SWZ_LAUNCH_CHECK(c);
      SWZ_HIP(c, hipMemcpyAsync(&ndense, dcount, 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      if (ndense) {
        SWZ_TRY(c->get("md_qdirty", (size_t)ncells, &a.dirty));
        SWZ_TRY(c->get("md_qbulk", (size_t)ncells, &a.bulk_round));
        SWZ_HIP(c, hipMemsetAsync(a.dirty, 0, (size_t)ncells, c->stream));
        SWZ_HIP(c, hipMemsetAsync(a.bulk_round, 0, (size_t)ncells * 4, c->stream));
        a.dlist = dl;
      } else {
        a.dense_min = 0;
      }
    }
  }
  // the round's queue in segments with a counter each (a single counter word takes ~90 atomics per microsecond, and
  // every activation pushes): workgroup b reads segment b % nseg and pushes into it
  a.nseg_shift = 5;
  if (const char* e = c->opt("SWZ_MD_KEYS_SEGS")) a.nseg_shift = (uint32_t)std::max(0, std::min(8, atoi(e)));
  a.nseg = 1u << a.nseg_shift;
  a.segcap = (uint32_t)std::min<uint64_t>(0x7FFFFFF0ull, 2ull * ncells / a.nseg + 4096ull);
  std::vector<MqArgs> ga(groups, a);
  std::vector<hipStream_t> gs(groups, c->stream);
  for (uint32_t g = 0; g < groups; ++g) {
    ga[g].group = g;
    ga[g].groups = groups;
    const std::string tag = std::to_string(g);
    SWZ_TRY(c->get(("md_qqueue0_g" + tag).c_str(), (size_t)a.nseg * a.segcap, &ga[g].queue[0]));
    SWZ_TRY(c->get(("md_qqueue1_g" + tag).c_str(), (size_t)a.nseg * a.segcap, &ga[g].queue[1]));
    SWZ_TRY(c->get(("md_qctr_g" + tag).c_str(), (size_t)6 * a.nseg * 32 + 32, &ga[g].qctr));
    ga[g].qhead = ga[g].qctr + (size_t)3 * a.nseg * 32;
    ga[g].qtotal = ga[g].qctr + (size_t)6 * a.nseg * 32;
    SWZ_HIP(c, hipMemsetAsync(ga[g].qctr, 0, ((size_t)6 * a.nseg * 32 + 32) * sizeof(uint32_t), c->stream));
    if (g > 0) {
      SWZ_TRY(c->get(("md_counters_g" + tag).c_str(), (size_t)CTR_COUNT, &ga[g].counters));
      SWZ_HIP(c, hipMemsetAsync(ga[g].counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
      while (c->aux_streams.size() < g) {
        hipStream_t stn = nullptr;
        SWZ_HIP(c, hipStreamCreateWithFlags(&stn, hipStreamNonBlocking));
        c->aux_streams.push_back(stn);
      }
      gs[g] = c->aux_streams[g - 1];
    }
  }
  {
    // the adjacent cells of every cell and the first round's queues, in one pass
    MqStart start{};
    start.groups = groups;
    start.lazy = lazy ? 1u : 0u;
    for (uint32_t g = 0; g < groups; ++g) {
      start.queue[g] = ga[g].queue[0];
      start.qctr[g] = ga[g].qctr;
    }
    hipLaunchKernelGGL(mq_nbr_build_kernel, dim3(std::min<uint32_t>(div_up(ncells, 8), 1u << 20)), dim3(256), 0, c->stream, ga[0], ncells, start);
    SWZ_LAUNCH_CHECK(c);
    SWZ_STAGE(c, "mq tables");
  }
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[swz] MIN_DISTANCE level %d on keys: %u pts in %u nodes, %u cells (cell levels %d), spacing %.1f key cells, band "
                    "[%.0f, %.0f), records of %u, typical cell %.0f pts, lazy %d (%.2f) patient %u groups %u big %d dense cells %u\n",
            plan.level, sample_points, sample_nodes, ncells, cl, km.T, (double)km.f_lo, (double)km.f_hi, a.rg - 1u, typical_pop, (int)lazy,
            (double)a.lazy_frac, a.patient, groups, (int)big_cells, ndense);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (dbg) {
    ev0 = c->take_event();
    ev1 = c->take_event();
    (void)hipEventRecord(ev0, c->stream);
  }
  // one wavefront per workgroup, as many as are resident at once (registers and LDS decide), a multiple of the segments
  const size_t lds_bytes = mq_lds_bytes(a.rg, sharded);
  uint32_t sweep_grid = 0;
  {
    int dev = 0, cus = 0, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const hipError_t oe = big_cells ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mq_sweep_kernel<4, false>, WAVE, lds_bytes)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mq_sweep_kernel<1, false>, WAVE, lds_bytes);
    if (oe != hipSuccess || per_cu < 1) per_cu = 8;
    if (cus < 1) cus = 256;
    uint32_t cap = (uint32_t)cus * (uint32_t)per_cu;
    if (const char* e = c->opt("SWZ_MD_GRID")) cap = (uint32_t)std::max(1, atoi(e)) * 4u;
    sweep_grid = std::max(a.nseg, (std::min<uint32_t>(cap / groups, ncells) + a.nseg - 1u) / a.nseg * a.nseg);
  }
  uint32_t batch = 32, batches_done = 0;
  const bool fixed_batch = c->opt("SWZ_MD_BATCH") != nullptr;
  if (fixed_batch) batch = std::max(1u, (uint32_t)atoi(c->opt("SWZ_MD_BATCH")));
  const auto wall0 = std::chrono::steady_clock::now();
  double wall_limit = 900.0;
  if (const char* e = c->opt("SWZ_MD_TIME_LIMIT")) wall_limit = atof(e);
  uint64_t max_rounds = 8ull * m + 1024;
  if (sharded) {
    // A cell at the face of a lower shard looks again every round until that shard's sweep has got there, and a level only
    // ends with an empty queue: a shard of a few points beside a lower shard of 1e8 spends (cheap, nearly empty) rounds
    // for as long as the lower sweeps take.  Its limit therefore covers their work as well -- an empty round is ~10 us
    // against the >= 20 us of a working one, so their rounds count several times.
    uint64_t below = 0;
    for (int p = 0; p < shard_root->shard; ++p) below += shard_root->views[p].npoints;
    max_rounds += 64ull * below + 65536ull;
  }
  if (const char* e = c->opt("SWZ_MD_ROUND_LIMIT")) max_rounds = (uint64_t)atoll(e);
  hipEvent_t fork = nullptr;
  if (groups > 1) {
    fork = c->take_event();
    SWZ_HIP(c, hipEventRecord(fork, c->stream));
    for (uint32_t g = 1; g < groups; ++g) SWZ_HIP(c, hipStreamWaitEvent(gs[g], fork, 0));
  }
  // a level is done when a round starts with an empty queue: only running cells wake sleeping ones
  std::vector<uint32_t> gleft(groups, 1);
  // The grid of a round follows its queue.  The workgroups of a segment draw its entries by ticket, so any multiple of the
  // segment count works; a round of a small level -- a 10 M batch of the multi-batch tiler holds a few hundred cells per
  // round -- then starts a few hundred single-wavefront workgroups instead of the ~6000 that are resident at once, all but
  // a few of which would find their segment empty (an empty launch of the full grid costs ~15 us, and such a level runs
  // ~2000 rounds).  The host knows the size of the last queue it looked at; the queue of a round grows slowly (a front
  // moving through the cells), so four times that size, at least eight workgroups per segment, covers the rounds until
  // the next look -- a grid that turns out small only makes its workgroups draw more tickets.
  // (Measured, round 5: no gain where it was meant to help -- 100 M points in 10 batches 372 vs 374 ms: a small round's 32 us
  // are the latency of its activations, not the launch of empty workgroups -- and 3 ms lost at level 1 of the 1 B run.  Off
  // unless SWZ_MD_ADAPT_GRID=1.)
  const bool adapt_grid = c->opt("SWZ_MD_ADAPT_GRID") && atoi(c->opt("SWZ_MD_ADAPT_GRID")) != 0 && !c->opt("SWZ_MD_GRID");
  std::vector<uint32_t> ggrid(groups, sweep_grid);
  auto grid_for = [&](uint32_t queued) {
    const uint64_t want = std::max<uint64_t>(8ull * a.nseg, 4ull * queued);
    return (uint32_t)std::min<uint64_t>(sweep_grid, (want + a.nseg - 1u) / a.nseg * a.nseg);
  };
  if (adapt_grid) {  // the first rounds: what mq_nbr_build_kernel has queued
    std::vector<uint32_t> h((size_t)a.nseg * 32u);
    for (uint32_t g = 0; g < groups; ++g) {
      SWZ_HIP(c, hipMemcpyAsync(h.data(), ga[g].qctr, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      uint64_t queued = 0;
      for (uint32_t sgm = 0; sgm < a.nseg; ++sgm) queued += std::min(h[(size_t)sgm * 32u], a.segcap);
      ggrid[g] = grid_for((uint32_t)std::min<uint64_t>(queued, 0xFFFFFFFFull));
    }
  }
  uint32_t round = MQ_FIRST_ROUND;
  bool running = true;
  while (running) {
    for (uint32_t b = 0; b < batch; ++b, ++round) {
      for (uint32_t g = 0; g < groups; ++g) {
        if (sharded) {
          hipLaunchKernelGGL(mq_round_word_kernel, dim3(1), dim3(1), 0, gs[g], a.round_word, round);
          if (big_cells)
            hipLaunchKernelGGL((mq_sweep_kernel<4, true>), dim3(ggrid[g]), dim3(WAVE), lds_bytes, gs[g], ga[g], round);
          else
            hipLaunchKernelGGL((mq_sweep_kernel<1, true>), dim3(ggrid[g]), dim3(WAVE), lds_bytes, gs[g], ga[g], round);
        } else if (big_cells) {
          hipLaunchKernelGGL((mq_sweep_kernel<4, false>), dim3(ggrid[g]), dim3(WAVE), lds_bytes, gs[g], ga[g], round);
          if (ndense) hipLaunchKernelGGL(mq_dense_reject_kernel, dim3(ndense), dim3(256), 0, gs[g], ga[g], round);
        } else {
          hipLaunchKernelGGL((mq_sweep_kernel<1, false>), dim3(ggrid[g]), dim3(WAVE), lds_bytes, gs[g], ga[g], round);
        }
      }
    }
    SWZ_LAUNCH_CHECK(c);
    if (!fixed_batch && ++batches_done % 4u == 0u && batch < 128u) batch *= 2u;
    for (uint32_t g = 0; g < groups; ++g)  // what the last launched round started with
      SWZ_HIP(c, hipMemcpyAsync(&gleft[g], ga[g].qtotal + (round - 1u - MQ_FIRST_ROUND) % 3u, 4, hipMemcpyDeviceToHost, gs[g]));
    uint32_t peer_err = 0;
    if (sharded) SWZ_HIP(c, hipMemcpyAsync(&peer_err, ga[0].counters + CTR_ERROR, 4, hipMemcpyDeviceToHost, gs[0]));
    running = false;
    for (uint32_t g = 0; g < groups; ++g) {
      SWZ_HIP(c, hipStreamSynchronize(gs[g]));
      running |= gleft[g] != 0u;
      if (adapt_grid) ggrid[g] = grid_for(gleft[g]);
    }
    if (peer_err) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE root of a sharded batch: a lower shard's sweep failed (or a queue overflowed); this shard gives up as well");
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > wall_limit) {
      char msg[200];
      snprintf(msg, sizeof(msg), "MIN_DISTANCE sweep on keys exceeded %.0f s: level %d, %u cells, %u rounds", wall_limit, plan.level,
               ncells, round);
      return c->fail(SWZ_ERR_INTERNAL, msg);
    }
    if (round > max_rounds) {
      char msg[160];
      snprintf(msg, sizeof(msg), "MIN_DISTANCE sweep on keys did not terminate: level %d, %u cells, %u rounds", plan.level, ncells, round);
      return c->fail(SWZ_ERR_INTERNAL, msg);
    }
  }
  if (fork) c->event_pool.push_back(fork);
  if (sharded) {  // everything this shard has written is complete now
    hipLaunchKernelGGL(mq_round_word_kernel, dim3(1), dim3(1), 0, c->stream, a.round_word, MQ_ROUND_DONE);
    word_guard.done = true;
    SWZ_LAUNCH_CHECK(c);
  }
  {  // every cell at its end, no queue segment ever full everywhere
    uint32_t* d_open = nullptr;
    SWZ_TRY(c->get("md_qopen", (size_t)4, &d_open));
    SWZ_HIP(c, hipMemsetAsync(d_open, 0, 4, c->stream));
    hipLaunchKernelGGL(mq_verify_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells, d_open);
    SWZ_LAUNCH_CHECK(c);
    uint32_t open_cells = 0, err = 0;
    SWZ_HIP(c, hipMemcpyAsync(&open_cells, d_open, 4, hipMemcpyDeviceToHost, c->stream));
    for (uint32_t g = 0; g < groups; ++g) {
      uint32_t eg = 0;
      SWZ_HIP(c, hipMemcpyAsync(&eg, ga[g].counters + CTR_ERROR, 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      err |= eg;
    }
    if (open_cells || err) {
      char msg[200];
      snprintf(msg, sizeof(msg), "MIN_DISTANCE sweep on keys ended with %u of %u cells undecided at level %d after %u rounds (queue overflow: %s)",
               open_cells, ncells, plan.level, round - MQ_FIRST_ROUND, err ? "yes" : "no");
      return c->fail(SWZ_ERR_INTERNAL, msg);
    }
  }
  if (const char* dc = c->opt("SWZ_MD_DUMP_CELL")) {  // debugging: the final state of one cell
    const uint32_t cell = (uint32_t)atoll(dc);
    if (cell < ncells) {
      uint2 ci;
      SWZ_HIP(c, hipMemcpy(&ci, a.cinfo + cell, sizeof(ci), hipMemcpyDeviceToHost));
      std::vector<uint8_t> st(ci.y - ci.x), tk(ci.y - ci.x);
      SWZ_HIP(c, hipMemcpy(st.data(), a.state + ci.x, st.size(), hipMemcpyDeviceToHost));
      SWZ_HIP(c, hipMemcpy(tk.data(), a.taken + ci.x, tk.size(), hipMemcpyDeviceToHost));
      std::vector<uint4> r(2 * a.rg);
      SWZ_HIP(c, hipMemcpy(r.data(), a.rec + ((size_t)cell << a.rg2_shift), r.size() * sizeof(uint4), hipMemcpyDeviceToHost));
      uint32_t nbr[32];
      SWZ_HIP(c, hipMemcpy(nbr, a.qnbr + (size_t)cell * 32, sizeof(nbr), hipMemcpyDeviceToHost));
      fprintf(stderr, "[swz] cell %u: [%u, %u)\n  state:", cell, ci.x, ci.y);
      for (size_t i = 0; i < st.size(); ++i) fprintf(stderr, "%s%u", i % 64 == 0 ? "\n   " : "", (unsigned)st[i]);
      fprintf(stderr, "\n  taken:");
      for (size_t i = 0; i < tk.size(); ++i) fprintf(stderr, "%s%u", i % 64 == 0 ? "\n   " : "", (unsigned)tk[i]);
      for (uint32_t b = 0; b < 2; ++b) {
        const uint4 h = r[b * a.rg];
        fprintf(stderr, "\n  record %u: pos %u cnt %u stamp %u end %u:", b, h.x, h.y, h.z, h.w);
        for (uint32_t j = 0; j + 1 < a.rg; ++j) {
          float4 en;
          memcpy(&en, &r[b * a.rg + 1 + j], 16);
          uint32_t id;
          memcpy(&id, &en.w, 4);
          fprintf(stderr, " (%.0f %.0f %.0f #%u)", en.x, en.y, en.z, id);
        }
      }
      fprintf(stderr, "\n  neighbours:");
      for (int k = 0; k < 27; ++k) fprintf(stderr, " %d", (int)nbr[k]);
      fprintf(stderr, "\n");
    }
  }
  if (rounds_out) *rounds_out += round - MQ_FIRST_ROUND;
  if (dbg) {
    float ms = 0.f;
    (void)hipEventRecord(ev1, c->stream);
    (void)hipEventSynchronize(ev1);
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    c->event_pool.push_back(ev0);
    c->event_pool.push_back(ev1);
    fprintf(stderr, "[swz] MIN_DISTANCE level %d on keys: sweep %.2f ms, %u rounds\n", plan.level, ms, round - MQ_FIRST_ROUND);
    for (uint32_t g = 0; g < groups; ++g) {
      uint32_t h[CTR_COUNT];
      SWZ_HIP(c, hipMemcpy(h, ga[g].counters, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[swz]   group %u: %u full activations, %u confirms still asleep, %u confirms that lost the claim, %u stalls, %u scans\n", g,
              h[CTR_DBG_HIST], h[CTR_DBG_HIST + 1], h[CTR_DBG_HIST + 2], h[CTR_DBG_HIST + 3], h[CTR_DBG_HIST + 4]);
#ifdef SWZ_MQ_STATS
      {
        const double ns = std::max(1u, h[CTR_DBG_HIST + 15]) * 100.0;  // 100 MHz ticks -> us
        fprintf(stderr, "[swz]   timed %u full activations (us): first loads %.2f, records + window %.2f, list %.2f, first window (R) %.2f, "
                        "windows + candidates + scans %.2f, publish + wake %.2f, total %.2f\n", h[CTR_DBG_HIST + 15], h[CTR_DBG_HIST + 8] / ns,
                h[CTR_DBG_HIST + 9] / ns, h[CTR_DBG_HIST + 10] / ns, h[CTR_DBG_HIST + 11] / ns, h[CTR_DBG_HIST + 12] / ns, h[CTR_DBG_HIST + 13] / ns,
                h[CTR_DBG_HIST + 14] / ns);
      }
#endif
    }
  }
  return SWZ_OK;
}

}  // namespace swz
