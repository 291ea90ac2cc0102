// swz_mindist.hip -- MIN_DISTANCE (K4c): exact greedy minimum-distance sampling on the GPU.
//
// Reference: PoissonDiskSampling::sample_points (core/tiling/Sampling.h:421-471) feeds the node's
// Morton-sorted points one by one into SparseGrid::add (core/datastructures/SparseGrid.cpp:116-146),
// which accepts a point iff no previously accepted point is closer than the spacing
// (GridCell::isDistant, GridCell.cpp:43-58: squared double distance < float-squared spacing).  The
// hash grid there is only an accelerator; the result is the lexicographically-first maximal
// independent set in Morton order:  accept(i) <=> for all accepted j < i : d2(i, j) >= s2.
//
// Exact parallel form used here ("frontier sweep").  Every sampled node is cut into octree cells at
// least one spacing wide, so a point can only conflict with points of its own and the 26 adjacent
// cells; cells are runs of the sorted keys and all points of a cell with a smaller Morton code
// precede all points of a cell with a larger one.  Each cell owns a frontier `pos`: its points before
// pos are decided.  In every round a wavefront advances the frontier of an active cell:
//   (R) a point closer than the spacing to an already accepted earlier point is rejected;
//   (A) a surviving point is accepted once no possibly-undecided earlier point (a point at or behind
//       the frontier of an earlier adjacent cell) is closer than the spacing -- otherwise the cell
//       stalls on that point and sleeps on the blocking cell (one of its 27 sleeper slots) until its frontier passed it.
// Rounds read only state committed by earlier launches (pos/acc_cnt are published by the second launch of a round),
// so a stale view is merely conservative.  Decisions follow exactly the reference's distance compares,
// hence the accepted set is bit-identical; the number of rounds is the true dependency depth of the
// greedy sweep instead of the length of the cell adjacency chains.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

constexpr uint32_t NONE32 = 0xFFFFFFFFu;
constexpr int MD_THREADS = 64;  // one wavefront per workgroup: wavefronts of a larger workgroup would wait for its slowest
constexpr int MD_WAVES = MD_THREADS / WAVE;
constexpr int MD_EXT_CAP = 128;   // accepted points of the neighbourhood cached in LDS per wave (window)
constexpr int MD_FRESH_CAP = 64;  // points a cell may accept per activation

enum : uint32_t { ST_STALLED = 0, ST_FINISHED = 1, ST_YIELD = 2 };

// Everything about a cell that an activation -- its own or an adjacent cell's -- looks at, in ONE 128-byte line (a
// scattered load costs a 128-byte line of HBM traffic whatever its size, tools/fetch_calib.hip): the record the other
// cells read, the owner's bookkeeping, and the cell's first accepted points.  A cell as wide as the spacing rarely
// accepts more than four points, so the accepted points of the neighbourhood usually arrive with the records.
constexpr uint32_t MD_INLINE = 4;
struct alignas(128) MdCell {
  uint4 rec;   // {start, end, pos, cnt}: point range, committed frontier (active index), committed number of accepted points
  uint4 cst;   // {pending frontier written by the sweep kernel (what a cell that goes to sleep compares with), stalled
               //  candidate, its blocker's rank << 8 | slot, where the blocker scan stopped}
  double acc[MD_INLINE][3];  // the first accepted points; the others are in MdArgs::acc_xyz
};
static_assert(sizeof(MdCell) == 128, "one cache line");

struct MdArgs {
  const uint64_t* akey;
  const uint32_t* aidx;
  uint32_t m;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* nstart;
  const double* X;     // positions in ACTIVE order (X[i] belongs to active point i)
  const double* Y;
  const double* Z;
  uint8_t* taken;
  uint32_t* counters;
  // cells
  MdCell* cells;
  uint32_t* crel;
  uint32_t* csnode;
  uint4* result;       // [queue slot][2]: what the activation in that slot of the round's queue ended with --
                       // {cell, new frontier, new count, status | moved << 8}, {blocking cell, blocking point, its slot, -}
  uint2* sleeper;      // [cell][28], 27 used: {the adjacent cell in direction k that sleeps on this cell (NONE32 = nobody),
                       // the point it waits for}; a cell is the only writer of its entry and the cell it sleeps on the
                       // only one who clears it
  double* acc_xyz;     // per cell: positions of its accepted points from the fifth on, 3 doubles each, at slots [start, start+cnt)
  uint32_t* gridmap;   // [sample node][cell code] -> cell index (build time only)
  // per cell, built once: its earlier adjacent cells, latest (largest Morton code) first
  uint32_t* nbr_id;    // [cell][27] cell index per rank
  uint8_t* nbr_slot;   // [cell][32]: direction slot (0..26) per rank, byte 31 = number of earlier adjacent cells
  uint32_t* queue[2];
  const uint32_t* snode_of;  // node -> compact index among sampled nodes
  uint32_t cell_shift;
  uint32_t cell_levels;      // octree levels between node and cell (cells per axis = 2^cell_levels)
  uint64_t cells_per_node;   // 8^cell_levels
  double sq_spacing;
  // blocker-scan culling: a cell is cut into 2^sub_levels slabs per axis; usq[a] = squared slab width
  uint32_t sub_levels;
  uint32_t batch_blockers;   // very sparse level: test all surviving lanes in one pass over the neighbours
  uint32_t latest_first;     // blocker scans visit the latest adjacent cell first (else the earliest)
  uint32_t patient;          // 1 = a stalled cell sleeps until the blocking CELL is finished, not just the blocking point
  float lazy_frac;           // lazy start: sleep until this fraction of the latest earlier neighbour is decided
  uint32_t ff_min;           // cells with more remaining points than this try the fast-forward first
  uint32_t all_sampled;      // every node of the level is sampled (the usual case): cell heads need no look at nid / nmode
  uint32_t group, groups;    // the sampled nodes are dealt to `groups` independent sets of cells (node % groups); this is set `group`
  uint32_t xcd_chunks;       // 1 = each XCD sweeps a contiguous eighth of the queue
  uint32_t ablate;           // debugging only (SWZ_MD_ABLATE): 1 = never blocked, 2 = no rejection tests
  double usq[3];
  double cull_sq;            // sq_spacing with a safety margin
};

__device__ __forceinline__ uint32_t md_spos(const uint32_t* aidx, uint32_t i) { return aidx ? aidx[i] : i; }

__global__ __launch_bounds__(256) void md_node_flag_kernel(const uint8_t* __restrict__ nmode, uint32_t nnodes,
                                                           uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < nnodes) out[j] = nmode[j] == MODE_SAMPLE ? 1u : 0u;
}

__device__ __forceinline__ bool md_is_head(const MdArgs& a, uint32_t i) {
  if (!a.all_sampled && a.nmode[a.nid[i]] != MODE_SAMPLE) return false;
  return i == 0 || ((a.akey[i] >> a.cell_shift) != (a.akey[i - 1] >> a.cell_shift));
}

// How many cells are OCCUPIED at every candidate cell level: a point that is not the first of its node is a cell
// head at cell level cl exactly when its key differs from its predecessor's within the first cl digits below the
// node prefix.  hist[0] counts the firsts of the sampled nodes, hist[b] the points whose first differing digit is
// digit b (1-based); occupied(cl) = hist[0] + ... + hist[cl].
__global__ __launch_bounds__(256) void md_cell_hist_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ nid,
                                                           const uint8_t* __restrict__ nmode, uint32_t m, uint32_t node_shift,
                                                           uint32_t cl_geo, uint32_t skip, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[16];
  if (threadIdx.x < 16) lh[threadIdx.x] = 0;
  __syncthreads();
  uint32_t mine = 0;  // lane b accumulates the wavefront's count of bin b
  // every skip-th tile of 256 points (the counts only steer the choice of cell size and algorithm: on large levels a
  // sample of some million points says the same as all of them and saves a pass over the keys)
  for (uint64_t i0 = (uint64_t)blockIdx.x * skip * 256u; i0 < m; i0 += (uint64_t)gridDim.x * skip * 256u) {
    const uint32_t i = (uint32_t)i0 + threadIdx.x;
    uint32_t bin = 0xFFu;
    if (i0 + threadIdx.x < m && nmode[nid[i]] == MODE_SAMPLE) {
      if (i == 0 || nid[i - 1] != nid[i]) {
        bin = 0;
      } else if (cl_geo) {
        const uint64_t diff = ((akey[i] ^ akey[i - 1]) >> (node_shift - 3u * cl_geo)) & ((1ull << (3u * cl_geo)) - 1ull);
        if (diff) bin = cl_geo - (uint32_t)(63 - __clzll((unsigned long long)diff)) / 3u;  // 1 .. cl_geo
      }
    }
    for (uint32_t b = 0; b <= cl_geo; ++b) {
      const uint32_t cnt = (uint32_t)__popcll(__ballot(bin == b));
      if (lane_id() == b) mine += cnt;
    }
  }
  if (lane_id() <= cl_geo && mine) atomicAdd(&lh[lane_id()], mine);
  __syncthreads();
  if (threadIdx.x < 16 && lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

// Points-weighted mean cell population at the cell levels cl_geo, cl_geo-1, -2, -3 (out[0..3] = sums over the
// samples, out[4] = samples): the population of the cell of every MD_POP_SAMPLES-th point, found by binary search for the
// cell's run in the sorted keys.  Tells whether the TYPICAL point would sit in an oversized cell after coarsening,
// which the plain average over cells does not (a dense blob in a sparse background).
constexpr uint32_t MD_POP_SAMPLES = 1u << 16;
__global__ __launch_bounds__(256) void md_cell_pop_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ nid,
                                                          const uint8_t* __restrict__ nmode, uint32_t m, uint32_t node_shift,
                                                          uint32_t cl_geo, unsigned long long* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= MD_POP_SAMPLES) return;
  const uint32_t i = (uint32_t)(((uint64_t)t * m) / MD_POP_SAMPLES);
  if (i >= m || nmode[nid[i]] != MODE_SAMPLE) return;
  const uint64_t key = akey[i];
  for (uint32_t k = 0; k <= 3u && k <= cl_geo; ++k) {
    const uint32_t sh = node_shift - 3u * (cl_geo - k);
    const uint64_t pre = key >> sh;
    uint32_t lo = 0, hi = i;  // first index with prefix >= pre
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if ((akey[mid] >> sh) < pre) lo = mid + 1u; else hi = mid;
    }
    const uint32_t first = lo;
    lo = i;
    hi = m;  // first index with prefix > pre
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if ((akey[mid] >> sh) <= pre) lo = mid + 1u; else hi = mid;
    }
    atomicAdd(&out[k], (unsigned long long)(lo - first));
  }
  atomicAdd(&out[4], 1ull);
}

// cells = runs of the cell prefix inside sampled nodes: counted and built by one fused scan (swz_scan.h)
struct CellHeadF {
  MdArgs a;
  __device__ uint32_t operator()(uint32_t i) const { return md_is_head(a, i) ? 1u : 0u; }
};
struct CellBuildG {
  MdArgs a;
  __device__ void operator()(uint32_t i, uint32_t c, uint32_t head) const {
    if (!head) return;
    a.cells[c].rec = make_uint4(i, 0u, i, 0u);
    a.crel[c] = (uint32_t)((a.akey[i] >> a.cell_shift) & (a.cells_per_node - 1ull));
    a.csnode[c] = a.snode_of[a.nid[i]];
    a.cells[c].cst = make_uint4(i, NONE32, 0u, 0u);  // the frontier as the round bookkeeping sees it: a cell that never ran has not moved
  }
};

__global__ __launch_bounds__(256) void md_cell_end_kernel(MdArgs a, uint32_t ncells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t s = a.cells[c].rec.x;
  const uint32_t node_end = a.nstart[a.nid[s] + 1];
  const uint32_t next = (c + 1 < ncells) ? a.cells[c + 1].rec.x : a.m;
  a.cells[c].rec.y = next < node_end ? next : node_end;
  a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + a.crel[c]] = c;
}

// earlier adjacent cells of every cell, sorted latest first (a cell's blocker scans and accepted-point
// pulls walk this list; it never changes during the sweep).  Half a wavefront per cell: lane k looks up
// adjacent cell k, its rank is the number of found cells with a larger code.
// (Grid-stride over the cells: 32 lanes per cell would exceed the 2^32 work-items one dispatch can have from
// 134 M cells on -- the excess workgroups silently never run; found at 213 M cells, tools/debug_fullsize.py.)
__global__ __launch_bounds__(256) void md_nbr_build_kernel(MdArgs a, uint32_t ncells) {
 for (uint64_t cbase = (uint64_t)blockIdx.x * 8u; cbase < ncells; cbase += (uint64_t)gridDim.x * 8u) {
  const uint32_t c = (uint32_t)cbase + threadIdx.x / 32u;
  const uint32_t k = threadIdx.x & 31u;
  uint32_t nrel = 0, nb = NONE32;
  bool have = false;
  if (c < ncells && k < 27u && k != 13u) {
    const uint32_t rel = a.crel[c];
    // codes of the adjacent cells by arithmetic on the dilated coordinates (every third bit)
    const uint32_t all = (uint32_t)(a.cells_per_node - 1ull);
    const uint32_t mz = all & 0x09249249u, my = mz << 1, mx = mz << 2;
    const uint32_t v[3] = {rel & mx, rel & my, rel & mz};
    const uint32_t mk[3] = {mx, my, mz};
    const uint32_t d[3] = {k % 3u, (k / 3u) % 3u, k / 9u};  // 0: minus one, 1: same, 2: plus one
    bool inside = true;
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
    if (inside && nrel < rel) {
      nb = a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + nrel];
      have = nb != NONE32;
    }
  }
  const uint32_t half = lane_id() & 32u;
  const uint32_t found = (uint32_t)(__ballot(have) >> half);
  uint32_t rank = 0;
  for (uint32_t j = 0; j < 27u; ++j) {
    const uint32_t other = (uint32_t)__shfl((int)nrel, (int)(half + j), WAVE);
    rank += (((found >> j) & 1u) && other > nrel) ? 1u : 0u;
  }
  if (have) {
    a.nbr_id[(size_t)c * 27 + rank] = nb;
    a.nbr_slot[(size_t)c * 32 + rank] = (uint8_t)k;
  }
  if (k == 31u && c < ncells) a.nbr_slot[(size_t)c * 32 + 31] = (uint8_t)__popc(found);
 }
}

// next cell (rank) of a blocker scan: latest first when the level is throughput bound, earliest first when
// it is latency bound (measured: 1 B points, root 229 vs 281 ms, level 0 476 vs 265 ms)
__device__ __forceinline__ int md_next_rank(uint32_t m, uint32_t latest_first) {
  return latest_first ? __ffs((int)m) - 1 : 31 - __clz((int)m);
}

__device__ __forceinline__ double bcast_f64(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
// A kernel-argument scalar as a value the optimiser cannot see through.  md_sweep_cell is inlined into a loop over the
// cells of a wavefront; vector-register copies of its double constants are loop invariant, get hoisted in front of the
// loop and -- there are more such values than registers -- are spilled there: 15 dwords per lane and wavefront, 126 GB
// of scratch writes per step at 1 B points (rocprofv3 WRITE_SIZE of md_sweep_kernel; the loop runs once or twice per
// wavefront).  Taken through this, the copies are made where they are used.
__device__ __forceinline__ double opaque(double v) {
  asm volatile("" : "+s"(v));
  return v;
}

// Wavefront scan / maximum with DPP moves instead of ds_bpermute shuffles: a shuffle needs its source-lane address in a
// vector register, and the six addresses of a scan are one more set of loop-invariant values that got hoisted in front
// of the cell loop and spilled (see opaque()).
template <typename Op>
__device__ __forceinline__ uint32_t md_wave_scan(uint32_t v, Op op, uint32_t identity) {
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
  v = op(v, (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
  return v;
}
struct MdAdd {
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; }
};
struct MdMax {
  __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, src);
}

struct MdLds {
  double ex[MD_EXT_CAP], ey[MD_EXT_CAP], ez[MD_EXT_CAP];
  double fx[MD_FRESH_CAP], fy[MD_FRESH_CAP], fz[MD_FRESH_CAP];
  uint8_t owner[MD_EXT_CAP];
};

// Fills the LDS window [base, base + MD_EXT_CAP) of the flattened list of accepted points of the
// neighbourhood (lane k < 27 owns the n_cnt entries of adjacent cell k starting at list offset off).
__device__ __forceinline__ uint32_t md_fill_window(const MdArgs& a, MdLds& lds, uint32_t base, uint32_t T,
                                                   uint32_t maxcnt, uint32_t n_cnt, uint32_t n_start, uint32_t off, uint32_t nb) {
  const uint32_t l = lane_id();
  const uint32_t wn = (T - base) < (uint32_t)MD_EXT_CAP ? (T - base) : (uint32_t)MD_EXT_CAP;
  // which adjacent cell owns list entry ti: lane k marks its entries, then every lane fetches whole
  // entries independently (all loads of the window are in flight together)
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = 0; j < maxcnt; ++j) {
    if (j < n_cnt) {
      const uint32_t ti = off + j;
      if (ti >= base && ti < base + MD_EXT_CAP) lds.owner[ti - base] = (uint8_t)l;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t ti = l; ti < ((wn + WAVE - 1) / WAVE) * WAVE; ti += WAVE) {
    const uint32_t k = ti < wn ? lds.owner[ti] : 0u;
    const uint32_t ks = __shfl(n_start, (int)k, WAVE), ko = __shfl(off, (int)k, WAVE), kb = __shfl(nb, (int)k, WAVE);
    if (ti < wn) {
      const uint32_t j = base + ti - ko;  // the cell's j-th accepted point: in its record, or in the overflow array
      const double* src = j < MD_INLINE ? &a.cells[kb].acc[j][0] : a.acc_xyz + (size_t)(ks + j) * 3;
      lds.ex[ti] = src[0];
      lds.ey[ti] = src[1];
      lds.ez[ti] = src[2];
    }
  }
  __builtin_amdgcn_wave_barrier();
  return wn;
}

// true when (x, y, z) is closer than the spacing to one of n points in LDS.  Two entries per step with independent
// chains: as a plain loop every iteration waits for its own LDS reads and its own chain of dependent double operations
// (~130 cycles per entry; measured at 1 B points, level 1: 77 entries per chunk of 64 points, 9.8 us per chunk;
// four per step cost more in spilled registers than it gained).
__device__ __forceinline__ bool md_near_any(const double* ex, const double* ey, const double* ez, uint32_t n, double x, double y,
                                            double z, double t) {
  bool hit = false;
  uint32_t i = 0;
  for (; i + 2u <= n; i += 2u) {
    const double x0 = ex[i], y0 = ey[i], z0 = ez[i];
    const double x1 = ex[i + 1], y1 = ey[i + 1], z1 = ez[i + 1];
    const double d0 = sq_dist(x, y, z, x0, y0, z0), d1 = sq_dist(x, y, z, x1, y1, z1);
    hit |= (d0 < t) | (d1 < t);
  }
  if (i < n) hit |= sq_dist(x, y, z, ex[i], ey[i], ez[i]) < t;
  return hit;
}

// squared slab distance from a point with slab coordinates (sx,sy,sz) to the adjacent cell in slot k
__device__ __forceinline__ bool md_culled(const MdArgs& a, int k, int sx, int sy, int sz) {
  const int smax = (1 << a.sub_levels) - 1;
  const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
  const double gx = (double)(dx < 0 ? sx : (dx > 0 ? smax - sx : 0));
  const double gy = (double)(dy < 0 ? sy : (dy > 0 ? smax - sy : 0));
  const double gz = (double)(dz < 0 ? sz : (dz > 0 ? smax - sz : 0));
  return gx * gx * a.usq[0] + gy * gy * a.usq[1] + gz * gz * a.usq[2] >= a.cull_sq;
}

// First LIVE point of [qs, qe) closer than the spacing to (bx, by, bz), or NONE32.  A point that is closer
// than the spacing to one of the wn accepted points in the LDS window is dead: it will be rejected whenever
// its cell gets to it (accepted sets only grow, and an accepted neighbour of an undecided point is always
// the earlier of the two), so it cannot keep anybody waiting.  The scan is a chain of dependent load
// latencies, so long ranges go four chunks (256 points) per step with all loads in flight.
__device__ __forceinline__ uint64_t md_live_hits(const MdArgs& a, const MdLds& lds, uint32_t wn, bool hit, double x,
                                                 double y, double z) {
  uint64_t hb = __ballot(hit);
  if (hb && wn) {
    if (md_near_any(lds.ex, lds.ey, lds.ez, wn, x, y, z, a.sq_spacing)) hit = false;
    hb = __ballot(hit);
  }
  return hb;
}

// (U: chunks of 64 points requested together; the small-cell variant of the kernel uses 1 and saves the registers)
template <int U>
__device__ __forceinline__ uint32_t md_first_hit(const MdArgs& a, const MdLds& lds, uint32_t wn, uint32_t qs, uint32_t qe,
                                                 double bx, double by, double bz) {
  const uint32_t l = lane_id();
  const double t = opaque(a.sq_spacing);
  uint32_t q0 = qs;
  while (qe - q0 > (uint32_t)WAVE) {  // qe > q0 always
    double x[U], y[U], z[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t q = min(q0 + (uint32_t)u * WAVE + l, qe - 1u);
      x[u] = a.X[q];
      y[u] = a.Y[q];
      z[u] = a.Z[q];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t q = q0 + (uint32_t)u * WAVE + l;
      const uint64_t hb = md_live_hits(a, lds, wn, q < qe && sq_dist(bx, by, bz, x[u], y[u], z[u]) < t, x[u], y[u], z[u]);
      if (hb) return q0 + (uint32_t)u * WAVE + (uint32_t)__ffsll((unsigned long long)hb) - 1u;
    }
    if (qe - q0 <= (uint32_t)U * WAVE) return NONE32;
    q0 += (uint32_t)U * WAVE;
  }
  const uint32_t q = min(q0 + l, qe - 1u);
  const double x = a.X[q], y = a.Y[q], z = a.Z[q];
  const uint64_t hb = md_live_hits(a, lds, wn, q0 + l < qe && sq_dist(bx, by, bz, x, y, z) < t, x, y, z);
  return hb ? q0 + (uint32_t)__ffsll((unsigned long long)hb) - 1u : NONE32;
}

// One wavefront advances the frontier of one active cell as far as it can.
// U = 4: cells of hundreds of points and more (blocker scans and the fast-forward take four chunks per memory round
// trip); U = 1: levels of small cells, which never need either and run better with the registers it saves.
// BATCH: very sparse levels (almost every candidate is accepted) test all surviving lanes of a chunk in one pass over
// the neighbours' undecided points.
template <int U, bool BATCH, bool EARLY = false>
__device__ void md_sweep_cell(const MdArgs& a, uint32_t slot, uint32_t c, MdLds& lds) {
  const uint32_t l = lane_id();
#ifdef SWZ_MD_STATS
  const uint64_t dbg_t0 = wall_clock64();
#endif
  const uint4 me = a.cells[c].rec;
  const uint32_t s0 = me.x, e = me.y;
  const double t = opaque(a.sq_spacing);
  const uint32_t P = me.z, CNT = me.w;

  // Large cells (U > 1) request the next chunk before the current one is worked on (most chunks of a large cell only
  // find every point rejected).  EARLY requests the chunk at the frontier here, so that it travels with the neighbour
  // records instead of costing a round trip of its own -- measured twice at 1 B points, slower both times (levels
  // 0 / 1: 108 / 135 ms against 103 / 131), so it is off.
  double nx = 0, ny = 0, nz = 0;
  uint64_t nkey = 0;
  uint32_t nat = NONE32;  // the chunk (nx, ny, nz, nkey) belongs to
  auto request = [&](uint32_t at) {
    nat = at;
    if (at + l < e) {
      nx = a.X[at + l];
      ny = a.Y[at + l];
      nz = a.Z[at + l];
      nkey = a.akey[at + l];
    }
  };
  if (EARLY) request(P);

  // lane r < nnb: the r-th earlier adjacent cell, latest (largest Morton code) first -- decisions arrive
  // roughly in Morton order, so the blocker found first tends to be decided last and one sleep covers the
  // others; lane 27: this cell (its committed accepted points join the rejection list)
  // (the table rows are requested together with the cell record: lane 31 of the slot row holds the count)
  const uint32_t raw_nb = a.nbr_id[(size_t)c * 27 + (l < 27u ? l : 26u)];
  const uint32_t raw_slot = a.nbr_slot[(size_t)c * 32 + (l & 31u)];
  const uint32_t nnb = bcast_u32(raw_slot, 31);
  uint32_t nb = NONE32, n_cnt = 0, n_start = 0, n_pos = 0, n_end = 0, slot_of_rank = 13;
  bool earlier = false;
  if (l < nnb) {
    nb = raw_nb;
    slot_of_rank = raw_slot;
    earlier = true;
    const uint4 o = a.cells[nb].rec;
    n_start = o.x;
    n_end = o.y;
    n_pos = o.z;
    n_cnt = o.w;
  } else if (l == 27) {
    nb = c;
    n_cnt = CNT;
    n_start = s0;
  }
  const uint32_t emask_r = (uint32_t)__ballot(earlier && n_pos < n_end);  // cells that may hold undecided points
  const uint32_t incl = md_wave_scan(n_cnt, MdAdd{}, 0u);
  const uint32_t off = incl - n_cnt;
  const uint32_t T = bcast_u32(incl, WAVE - 1);  // accepted points of the neighbourhood (incl. own committed)
  const uint32_t maxcnt = bcast_u32(md_wave_scan(n_cnt, MdMax{}, 0u), WAVE - 1);
  const uint4 cst = a.cells[c].cst;
  const bool resume = cst.y == P;
  const uint32_t r_packed = resume ? cst.z : 0u;
  const uint32_t r_group = r_packed >> 8;    // rank (scan position) of that cell: earlier ranks were scanned clean
  const uint32_t r_q = resume ? cst.w : 0u;

  uint32_t wn0 = 0;
  if (T > 0) wn0 = md_fill_window(a, lds, 0, T, maxcnt, n_cnt, n_start, off, nb);

  // the whole list of accepted points is resident in LDS: blocker scans can tell dead points
  const uint32_t live_wn = (T <= (uint32_t)MD_EXT_CAP && !(a.ablate & 8u)) ? wn0 : 0u;
  uint32_t fresh = 0;
#ifdef SWZ_MD_STATS
  uint32_t dbg_chunk = 0, dbg_ranks = 0;
  uint64_t dbg_tscan = 0, dbg_tchunk = 0;
  const uint64_t dbg_t1 = wall_clock64();  // prologue done (records, window)
#endif
  uint32_t cur = P;
  uint32_t out_pos = e, out_status = ST_FINISHED;
  uint32_t b_slot = 0, b_q = 0, b_cell = 0;
  bool stop = false;

  // Very large cells (dense blobs: thousands of points per cell): skip stretches in which every point is already
  // rejected by the committed accepted points, four chunks per memory round trip.
  if (U > 1 && live_wn && e - cur > a.ff_min && !(a.ablate & 16u)) {
    while (e - cur > 4u * WAVE) {
      double x[4], y[4], z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t q = cur + (uint32_t)u * WAVE + l;
        x[u] = a.X[q];
        y[u] = a.Y[q];
        z[u] = a.Z[q];
      }
      bool alive_l = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        alive_l |= !md_near_any(lds.ex, lds.ey, lds.ez, live_wn, x[u], y[u], z[u], t);
      }
      if (__ballot(alive_l)) break;
      cur += 4u * WAVE;
    }
  }

  while (cur < e && !stop) {
#ifdef SWZ_MD_STATS
    const uint64_t dbg_tc = wall_clock64();
#endif
    const uint32_t p = cur + l;
    const bool valid = p < e;
    double px = 0, py = 0, pz = 0;
    uint64_t key = 0;
    if (nat != cur) request(cur);
    px = nx;
    py = ny;
    pz = nz;
    key = nkey;
    if (U > 1 && cur + WAVE < e) request(cur + WAVE);
#ifdef SWZ_MD_STATS
    ++dbg_chunk;
#endif
    bool rej = !valid;
    // (R) against the committed accepted points of the neighbourhood, window by window
    for (uint32_t base = 0; base < T && !(a.ablate & 2u); base += MD_EXT_CAP) {
      const uint32_t wn = (base == 0 && T <= (uint32_t)MD_EXT_CAP && cur == P)
                            ? wn0
                            : ((T <= (uint32_t)MD_EXT_CAP) ? wn0 : md_fill_window(a, lds, base, T, maxcnt, n_cnt, n_start, off, nb));
      rej |= md_near_any(lds.ex, lds.ey, lds.ez, wn, px, py, pz, t);
      if (!__ballot(!rej)) break;
    }
    // ... and against the points accepted earlier in this activation
    rej |= md_near_any(lds.fx, lds.fy, lds.fz, fresh, px, py, pz, t);

    uint64_t alive = __ballot(!rej);
#ifdef SWZ_MD_STATS
    dbg_tchunk += wall_clock64() - dbg_tc;
#endif
    // Adjacent cells that can hold a point closer than the spacing to this lane's point: squared slab gaps per axis
    // and direction, summed per slot (all lanes in parallel) -- only for chunks that have a survivor and somebody it
    // could wait for (most chunks of a large cell find every point rejected)
    uint32_t needmask = 0, needrank = 0;  // needrank: the same in scan order, restricted to cells that may hold undecided points
    if (U == 1 || (alive && emask_r)) {  // (small cells nearly always have both: no branch there)
      int sx = 0, sy = 0, sz = 0;
      if (valid) {
        // slab coordinates inside the cell (culling of blocker scans)
        // (at most 4 sub levels = 12 bits: the 32-bit form of the bit trick)
        const uint32_t sub = (uint32_t)(key >> (a.cell_shift - 3u * a.sub_levels)) & ((1u << (3u * a.sub_levels)) - 1u);
        sx = (int)contract_bits_by_3_u32(sub >> 2);
        sy = (int)contract_bits_by_3_u32(sub >> 1);
        sz = (int)contract_bits_by_3_u32(sub);
      }
      const int smax = (1 << a.sub_levels) - 1;
      const double lx = (double)sx, hx = (double)(smax - sx), ly = (double)sy, hy = (double)(smax - sy),
                   lz = (double)sz, hz = (double)(smax - sz);
      const double ux = opaque(a.usq[0]), uy = opaque(a.usq[1]), uz = opaque(a.usq[2]), cull = opaque(a.cull_sq);
      const double gx[3] = {lx * lx * ux, 0.0, hx * hx * ux};
      const double gy[3] = {ly * ly * uy, 0.0, hy * hy * uy};
      const double gz[3] = {lz * lz * uz, 0.0, hz * hz * uz};
#pragma unroll
      for (int k = 0; k < 27; ++k)
        if (gx[k % 3] + gy[(k / 3) % 3] + gz[k / 9] < cull) needmask |= 1u << k;
      for (uint32_t r = 0; r < nnb; ++r)
        if ((emask_r >> r) & 1u) needrank |= ((needmask >> bcast_u32(slot_of_rank, (int)r)) & 1u) << r;
    }
    // (A) for many survivors at once: one pass over the possibly-undecided points of the earlier
    // adjacent cells, every surviving lane testing its own point against the broadcast one
    bool pre = false;          // blocker flags of this chunk were precomputed for all lanes
    bool blk = false;          // lane: an undecided earlier point is closer than the spacing
    uint32_t blk_slot_l = 0, blk_q_l = 0;
    const int first_alive = alive ? __ffsll((unsigned long long)alive) - 1 : 0;
    bool still = false;  // the stalled candidate of the last activation is still blocked by the same point
    if (resume && cur == P && (alive & 1ull) && ((emask_r >> r_group) & 1u) &&
        bcast_u32(n_pos, (int)r_group) <= r_q) {
      still = true;
      blk = (l == 0);
      blk_slot_l = r_group;
      blk_q_l = r_q;
      pre = true;
    }
    if (BATCH && !still && __popcll(alive) > 8) {
      pre = true;
      uint32_t mm = emask_r;
      bool first_blocked = false;
      while (mm && !first_blocked) {
        const int k = __ffs((int)mm) - 1;  // rank
        mm &= mm - 1;
        const bool need = !rej && !blk && ((needrank >> k) & 1u);
        if (!__ballot(need)) continue;
        const uint32_t qs = bcast_u32(n_pos, k), qe = bcast_u32(n_end, k);
        for (uint32_t q0 = qs; q0 < qe; q0 += WAVE) {
          double qx = 0, qy = 0, qz = 0;
          if (q0 + l < qe) {
            qx = a.X[q0 + l];
            qy = a.Y[q0 + l];
            qz = a.Z[q0 + l];
          }
          const int nq = (int)((qe - q0) < (uint32_t)WAVE ? (qe - q0) : WAVE);
          for (int j = 0; j < nq; ++j) {
            const double bx = bcast_f64(qx, j), by = bcast_f64(qy, j), bz = bcast_f64(qz, j);
            if (need && !blk && sq_dist(px, py, pz, bx, by, bz) < t) {
              blk = true;
              blk_slot_l = (uint32_t)k;
              blk_q_l = q0 + (uint32_t)j;
            }
          }
          first_blocked = __builtin_amdgcn_readlane((int)blk, first_alive) != 0;
          if (first_blocked || !__ballot(need && !blk)) break;
        }
      }
    }

    // surviving lanes in order: accept or stall
    while (alive) {
      const int j = __ffsll((unsigned long long)alive) - 1;
      const uint32_t cand = cur + (uint32_t)j;
      const double bx = bcast_f64(px, j), by = bcast_f64(py, j), bz = bcast_f64(pz, j);
      bool blocked = false;
      if (pre) {
        blocked = __builtin_amdgcn_readlane((int)blk, j) != 0;
        if (blocked) {
          const uint32_t hr = bcast_u32(blk_slot_l, j);  // rank of the blocking cell
          b_q = bcast_u32(blk_q_l, j);
          b_cell = bcast_u32(nb, (int)hr);
          b_slot = bcast_u32(slot_of_rank, (int)hr) | (hr << 8);
        }
      } else {
        // few survivors: scan the earlier adjacent cells for this candidate, 64 points at a time
        uint32_t nm = bcast_u32(needrank, j);
        if (a.ablate & 1u) nm = 0;
        while (nm && !blocked) {
          const int r = md_next_rank(nm, a.latest_first);
          nm &= ~(1u << r);
          const uint32_t qs = bcast_u32(n_pos, r);
          const uint32_t qe = bcast_u32(n_end, r);
#ifdef SWZ_MD_STATS
          const uint64_t dbg_ts = wall_clock64();
#endif
          const uint32_t hq = md_first_hit<U>(a, lds, live_wn, qs, qe, bx, by, bz);
#ifdef SWZ_MD_STATS
          dbg_tscan += wall_clock64() - dbg_ts;
          ++dbg_ranks;
#endif
          if (hq != NONE32) {
            blocked = true;
            b_slot = bcast_u32(slot_of_rank, r) | ((uint32_t)r << 8);
            b_q = a.patient ? qe - 1u : hq;
            b_cell = bcast_u32(nb, r);
          }
        }
      }
      if (a.ablate & 4u) blocked = false;
      if (blocked) {
        out_pos = cand;
        out_status = ST_STALLED;
        stop = true;
        break;
      }
      // accepted
      if ((int)l == j) {
        a.taken[cand] = 1;
        const uint32_t ai = CNT + fresh;
        double* dst = ai < MD_INLINE ? &a.cells[c].acc[ai][0] : a.acc_xyz + (size_t)(s0 + ai) * 3;
        dst[0] = px;
        dst[1] = py;
        dst[2] = pz;
        lds.fx[fresh] = px;
        lds.fy[fresh] = py;
        lds.fz[fresh] = pz;
      }
      ++fresh;
      if ((int)l > j && !rej && sq_dist(px, py, pz, bx, by, bz) < t) rej = true;
      alive = __ballot(!rej && (int)l > j);
      __builtin_amdgcn_wave_barrier();
      if (fresh == (uint32_t)MD_FRESH_CAP) {  // LDS list full: publish and continue in the next round
        const uint32_t next = alive ? cur + (uint32_t)__ffsll((unsigned long long)alive) - 1u
                                    : ((cur + WAVE) < e ? cur + WAVE : e);
        if (next < e) {
          out_pos = next;
          out_status = ST_YIELD;
          stop = true;
        }
        break;
      }
    }
    if (!stop) cur += WAVE;
  }
  if (l == 0) {
#ifdef SWZ_MD_STATS
    if ((c & 63u) == 0u) {  // a sample of the cells: the atomics themselves disturb the timing
      const uint32_t dt = (uint32_t)(wall_clock64() - dbg_t0);  // 100 MHz
      atomicAdd(&a.counters[CTR_DBG_TIME], dt);
      atomicMax(&a.counters[CTR_DBG_TMAX], dt);
      atomicAdd(&a.counters[CTR_DBG_HIST + 0], 1u);
      atomicAdd(&a.counters[CTR_DBG_HIST + 1], (uint32_t)(dbg_t1 - dbg_t0));
      atomicAdd(&a.counters[CTR_DBG_HIST + 2], (uint32_t)dbg_tchunk);
      atomicAdd(&a.counters[CTR_DBG_HIST + 3], (uint32_t)dbg_tscan);
      atomicAdd(&a.counters[CTR_DBG_HIST + 4], dbg_chunk);
      atomicAdd(&a.counters[CTR_DBG_HIST + 5], dbg_ranks);
    }
#endif
    const uint32_t moved = (out_pos > P || out_status == ST_FINISHED) ? 1u : 0u;
    a.result[(size_t)slot * 2] = make_uint4(c, out_pos, CNT + fresh, out_status | (moved << 8));
    if (out_status == ST_STALLED) {
      a.result[(size_t)slot * 2 + 1] = make_uint4(b_cell, b_q, b_slot & 0xFFu, 0u);
      a.cells[c].cst = make_uint4(out_pos, out_pos, b_slot, b_q);
    } else {
      a.cells[c].cst = make_uint4(out_pos, NONE32, 0u, 0u);
    }
  }
}

#ifndef SWZ_MD_MIN_WAVES
#define SWZ_MD_MIN_WAVES 5
#endif
template <int U, bool BATCH>
__global__ __launch_bounds__(MD_THREADS, U == 1 ? SWZ_MD_MIN_WAVES : 4) void md_sweep_kernel(MdArgs a, uint32_t round) {
  __shared__ MdLds lds[MD_WAVES];
  const uint32_t w = threadIdx.x / WAVE;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[CTR_Q0 + (round + 2) % 3] = 0;
  const uint32_t nq = a.counters[CTR_Q0 + round % 3];
  const uint32_t* qin = a.queue[round & 1];
  // one loop for both mappings (the body is large: a second inlined copy doubles the kernel)
  uint32_t first = blockIdx.x * MD_WAVES + w, end = nq, step = gridDim.x * MD_WAVES;
  if (a.xcd_chunks && (gridDim.x & 7u) == 0) {
    // workgroups go round-robin over the 8 XCDs: give every XCD one contiguous eighth of the queue so that
    // neighbouring cells (which read each other's records and accepted points) share an L2
    const uint32_t seg = (nq + 7u) / 8u, x = blockIdx.x & 7u;
    first = x * seg + (blockIdx.x >> 3) * MD_WAVES + w;
    end = min(nq, (x + 1u) * seg);
    step = (gridDim.x >> 3) * MD_WAVES;
  }
  for (uint32_t i = first; i < end; i += step) md_sweep_cell<U, BATCH>(a, i, qin[i], lds[w]);
}

// append `value` of every lane with want == true to the queue: one atomic per wavefront
__device__ __forceinline__ void md_wave_push(bool want, uint32_t value, uint32_t* qout, uint32_t* cout) {
  const uint64_t m = __ballot(want);
  if (!m) return;
  const int leader = __ffsll((unsigned long long)m) - 1;
  uint32_t base = 0;
  if ((int)lane_id() == leader) base = atomicAdd(cout, (uint32_t)__popcll(m));
  base = __shfl(base, leader, WAVE);
  if (want) qout[base + (uint32_t)__popcll(m & lanemask_lt())] = value;
}

// After the sweep of a round, for every cell that took part (one lane per cell):
//  * publish its new frontier and number of accepted points (the record the next round's activations read);
//  * if the frontier moved, wake the adjacent cells that sleep on it and whose blocking point it has passed;
//  * if the cell itself ended the round stalled, put it to sleep on the blocking cell -- or straight back into the
//    queue when that cell's frontier has already passed the blocking point; yielded cells are re-queued.
// One launch does all three: a cell that goes to sleep decides from the blocker's PENDING frontier (npos, final since
// the sweep ended), so it does not depend on the blocker's record having been published in this launch; and a sleeper
// is an entry in the blocker's 27-slot array written by the sleeper alone and cleared by the blocker alone, so the
// blocker's lane may see the entry now or in its next round, either is right (the wake test is the same data the
// sleeper decided on).  (Until round 2 this was two launches, commit and requeue, with linked wait lists: 18 us per
// round at 1 B points, now ~13.)
// (wave0: index of this wavefront's first lane among all participating threads, stride: their number)
__device__ __forceinline__ void md_commit_requeue_range(const MdArgs& a, uint32_t round, uint32_t wave0, uint32_t stride) {
  const uint32_t nq = a.counters[CTR_Q0 + round % 3];
  uint32_t* cout = &a.counters[CTR_Q0 + (round + 1) % 3];
  uint32_t* qout = a.queue[(round + 1) & 1];
  for (uint32_t i0 = wave0; i0 < nq; i0 += stride) {  // wave-uniform
    const uint32_t i = i0 + lane_id();
    const bool valid = i < nq;
    // first round trip: what the activation in this slot of the queue ended with (written by the sweep)
    uint4 r0 = make_uint4(0, 0, 0, ST_FINISHED), r1 = make_uint4(0, 0, 0, 0);
    if (valid) {
      r0 = a.result[(size_t)i * 2];
      r1 = a.result[(size_t)i * 2 + 1];  // (only meaningful for a stalled cell; requested anyway: no second round trip)
    }
    const uint32_t c = r0.x, np = r0.y, st = r0.w & 0xFFu;
    const bool fin = valid && st == ST_FINISHED;
    const bool moved = valid && ((r0.w >> 8) & 1u);
    const bool stalled = valid && st == ST_STALLED;
    const uint32_t b = r1.x, bq = r1.y, bslot = r1.z;
    // second round trip: the entries of the cells sleeping on this one (with the points they wait for), the blocker's
    // pending frontier
    uint4 sl[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) sl[k] = make_uint4(NONE32, 0u, NONE32, 0u);
    if (moved) {
      const uint4* row = reinterpret_cast<const uint4*>(a.sleeper + (size_t)c * 28);
#pragma unroll
      for (int k = 0; k < 14; ++k) sl[k] = row[k];
    }
    const uint32_t bpos = stalled ? a.cells[b].cst.x : 0u;
    if (valid) {
      a.cells[c].rec.z = np;
      a.cells[c].rec.w = r0.z;
    }
    const uint64_t fm = __ballot(fin);
    if (fm && lane_id() == 0) atomicAdd(&a.counters[CTR_DONE_CELLS], (uint32_t)__popcll(fm));
    uint32_t w[27];
    uint32_t wake = 0;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const uint4 v = sl[k / 2];
      w[k] = (k & 1) ? v.z : v.x;
      const uint32_t wq = (k & 1) ? v.w : v.y;
      if (w[k] != NONE32 && (fin || wq < np)) wake |= 1u << k;
    }
    // the cell's own fate
    const bool push = (valid && st == ST_YIELD) || (stalled && bpos > bq);
    if (stalled && !push) a.sleeper[(size_t)b * 28 + (26u - bslot)] = make_uint2(c, bq);  // the direction from b to c
    // one queue reservation for the wavefront: the cells themselves, then the sleepers they wake
    const uint32_t cnt = (push ? 1u : 0u) + (uint32_t)__popc(wake);
    const uint32_t incl = wave_incl_sum(cnt);
    const uint32_t total = bcast_u32(incl, WAVE - 1);
    if (total) {
      uint32_t base = 0;
      if (lane_id() == 0) base = atomicAdd(cout, total);
      uint32_t off = bcast_u32(base, 0) + incl - cnt;
      if (push) qout[off++] = c;
#pragma unroll
      for (int k = 0; k < 27; ++k)
        if ((wake >> k) & 1u) {
          qout[off++] = w[k];
          a.sleeper[(size_t)c * 28 + k].x = NONE32;
        }
    }
  }
}
__global__ __launch_bounds__(256) void md_commit_requeue_kernel(MdArgs a, uint32_t round) {
  md_commit_requeue_range(a, round, blockIdx.x * 256 + (threadIdx.x & ~63u), gridDim.x * 256);
}

// ---- the rounds of one level inside ONE launch ---------------------------------------------------------------------
// A round of three launches costs about 44 us even when only a few cells are active, and a dense level runs well
// over a thousand dependent rounds.  Here one workgroup per CU stays resident and the three steps of a round are
// separated by grid barriers instead: every workgroup releases its stores at agent scope (buffer_wbl2), arrives on one
// monotonic counter, polls it relaxed, and acquires (buffer_inv) -- cdna_hip_programming.md Guideline 16 in its counter
// form; the state words are zeroed by the host before every launch and every spin is bounded: a workgroup that is not
// resident -- another context's kernels or a staged copy kernel on the device are enough -- would otherwise hang the
// others.  When a spin runs out the launch gives up and the CALL FAILS with SWZ_ERR_INTERNAL (the steps of a round are
// not restartable, so there is no falling back to plain launches): SWZ_MD_PERSISTENT=1 is an experiment switch that
// never changes a result but may fail a call; it is off by default and slower than the plain rounds.
struct MdBarrier {
  uint32_t arrived;   // monotonic over the launch
  uint32_t timeout;   // set when a spin ran out
  uint32_t rounds;    // rounds completed in this launch
  uint32_t pad;
};
constexpr int MDP_THREADS = 1024;
constexpr int MDP_WAVES = MDP_THREADS / WAVE;

__device__ __forceinline__ bool md_grid_barrier(MdBarrier* bar, uint32_t nblocks, uint32_t& epoch, uint32_t* lds_flag) {
  __syncthreads();  // every wave's stores are issued ...
  ++epoch;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... and written back before the arrival is visible
    __hip_atomic_fetch_add(&bar->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t target = epoch * nblocks;
    uint32_t ok = 1;
    for (uint32_t spins = 0;; ++spins) {
      if (__hip_atomic_load(&bar->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
      if (spins > (1u << 22) || __hip_atomic_load(&bar->timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(&bar->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}

__global__ __launch_bounds__(MDP_THREADS, 4) void md_persistent_kernel(MdArgs a, uint32_t ncells, uint32_t round0,
                                                                       uint32_t max_rounds, MdBarrier* bar) {
  __shared__ MdLds lds[MDP_WAVES];
  __shared__ uint32_t s_flag, s_done;
  const uint32_t w = threadIdx.x / WAVE;
  const uint32_t nblocks = gridDim.x;
  const uint32_t wave0 = blockIdx.x * MDP_THREADS + (threadIdx.x & ~63u);
  const uint32_t stride = nblocks * MDP_THREADS;
  uint32_t epoch = 0;
  for (uint32_t round = round0; round - round0 < max_rounds; ++round) {
    // sweep
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[CTR_Q0 + (round + 2) % 3] = 0;
    {
      const uint32_t nq = a.counters[CTR_Q0 + round % 3];
      const uint32_t* qin = a.queue[round & 1];
      for (uint32_t i = blockIdx.x * MDP_WAVES + w; i < nq; i += nblocks * MDP_WAVES) md_sweep_cell<1, true>(a, i, qin[i], lds[w]);
    }
    if (!md_grid_barrier(bar, nblocks, epoch, &s_flag)) return;
    md_commit_requeue_range(a, round, wave0, stride);
    if (!md_grid_barrier(bar, nblocks, epoch, &s_flag)) return;
    if (threadIdx.x == 0) {
      s_done = __hip_atomic_load(&a.counters[CTR_DONE_CELLS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0) bar->rounds = round - round0 + 1u;
    }
    __syncthreads();
    if (s_done >= ncells) return;
  }
}

__global__ __launch_bounds__(256) void md_gather_active_kernel(const uint32_t* __restrict__ aidx, uint32_t m,
                                                               const double* __restrict__ X,
                                                               const double* __restrict__ Y,
                                                               const double* __restrict__ Z, double* __restrict__ ax,
                                                               double* __restrict__ ay, double* __restrict__ az) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t s = aidx[i];
  ax[i] = X[s];
  ay[i] = Y[s];
  az[i] = Z[s];
}

__global__ __launch_bounds__(256) void md_fill_queue_kernel(MdArgs a, uint32_t n, uint32_t* q, uint32_t* counter) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  md_wave_push(i < n && a.csnode[i] % a.groups == a.group, i, q, counter);
}

// Patient levels start lazily: only the cells without an earlier adjacent cell are queued, every other cell
// sleeps until its latest earlier neighbour is finished (sleeping on a blocker that is not the real one is
// always safe, the cell looks again when it wakes up).  This replaces a first round in which every cell
// would scan its neighbourhood just to find out that it has to wait.
__global__ __launch_bounds__(256) void md_lazy_start_kernel(MdArgs a, uint32_t ncells, uint32_t* q, uint32_t* counter) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  bool push = false;
  if (c < ncells && a.csnode[c] % a.groups == a.group) {
    if (a.nbr_slot[(size_t)c * 32 + 31] == 0) {
      push = true;
    } else {
      const uint32_t b = a.nbr_id[(size_t)c * 27];
      const uint4 o = a.cells[b].rec;
      const uint32_t bq = o.x + (uint32_t)((float)(o.y - 1u - o.x) * a.lazy_frac);
      a.sleeper[(size_t)b * 28 + (26u - a.nbr_slot[(size_t)c * 32])] = make_uint2(c, bq);
    }
  }
  md_wave_push(push, c, q, counter);
}

int min_distance_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                       const LevelBuffers& lb, uint32_t nnodes, uint32_t sample_nodes, uint32_t sample_points,
                       uint32_t* rounds_out) {
  const uint32_t m = as.m;
  c->next_scratch_epoch();  // what the level before asked for ("md_*", "sp_*") may go if memory runs out

  // occupied cells at every candidate cell level (one pass over the keys)
  uint32_t occupied[12] = {0};
  {
    uint32_t* d_hist = nullptr;
    SWZ_TRY(c->get("md_hist", (size_t)16, &d_hist));
    SWZ_HIP(c, hipMemsetAsync(d_hist, 0, 64, c->stream));
    const uint32_t nsh = plan.node_shift == 63u ? 63u : plan.node_shift;
    const uint32_t tiles = div_up(m, 256);
    uint32_t skip = std::max(1u, m >> 23);  // about 8 M points are looked at
    if (const char* e = c->opt("SWZ_MD_HIST_SKIP")) skip = std::max(1, atoi(e));
    const uint32_t sampled_tiles = div_up(tiles, skip);
    hipLaunchKernelGGL(md_cell_hist_kernel, dim3(std::min<uint32_t>(sampled_tiles, 4096u)), dim3(256), 0, c->stream, as.akey,
                       lb.nid, lb.nmode, m, nsh, (uint32_t)plan.cell_levels_geo, skip, d_hist);
    SWZ_LAUNCH_CHECK(c);
    SWZ_STAGE(c, "md cell hist");
    uint32_t h[16];
    SWZ_HIP(c, hipMemcpyAsync(h, d_hist, 64, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    const double scale = skip == 1 ? 1.0 : (double)m / (double)std::min<uint64_t>(m, (uint64_t)sampled_tiles * 256u);
    double run = 0;
    for (int b = 0; b < 12; ++b) {
      run += h[b];
      occupied[b] = (uint32_t)std::min<double>(run * scale, (double)m);
    }
  }
  // cell size: as fine as the spacing allows, but coarse enough that an OCCUPIED cell holds >= 8 points on
  // average (clustered data leaves most of a node empty: the average over the node's volume would make the cells
  // of a dense sheet or blob far too large) and that the dense [node][cell] lookup table stays affordable
  int cl = plan.cell_levels_geo;
  const double avg = (double)sample_points / (double)sample_nodes;
  double per_cell = 8.0;
  if (const char* e = c->opt("SWZ_MD_DENSITY")) per_cell = atof(e);
  // ... but only while the TYPICAL point would not end up in an oversized cell (points-weighted mean population
  // after the step <= 160): with mixed densities (a dense blob in a sparse background) the average over the cells
  // says little, and cells that are too large for the dense part cost far more (long serial activations) than
  // cells that are too small for the sparse part (more, cheap activations).  Measured on 100 M clustered points:
  // 11.4 s with the volume average, 0.66 s with this rule; uniform data choose the same cells as before.
  double pop[4] = {0, 0, 0, 0};  // at cell level cl_geo, cl_geo - 1, - 2, - 3
  {
    unsigned long long* d_pop = nullptr;
    SWZ_TRY(c->get("md_pop", (size_t)8, &d_pop));
    SWZ_HIP(c, hipMemsetAsync(d_pop, 0, 64, c->stream));
    const uint32_t nsh = plan.node_shift == 63u ? 63u : plan.node_shift;
    hipLaunchKernelGGL(md_cell_pop_kernel, dim3(MD_POP_SAMPLES / 256), dim3(256), 0, c->stream, as.akey, lb.nid, lb.nmode, m, nsh,
                       (uint32_t)plan.cell_levels_geo, d_pop);
    SWZ_LAUNCH_CHECK(c);
    SWZ_STAGE(c, "md cell pop");
    unsigned long long h[5];
    SWZ_HIP(c, hipMemcpyAsync(h, d_pop, 40, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 4; ++k) pop[k] = h[4] ? (double)h[k] / (double)h[4] : 1e30;
  }
  double max_pop = 160.0;
  if (const char* e = c->opt("SWZ_MD_MAX_POP")) max_pop = atof(e);
  while (cl > 0 && plan.cell_levels_geo - cl < 3 && (double)sample_points / (double)std::max(1u, occupied[cl]) < per_cell &&
         pop[plan.cell_levels_geo - cl + 1] <= max_pop)
    --cl;
  // the dense [node][cell] map: at most 2^31 entries (8.6 GB; it is memset once per level, a few ms)
  while (cl > 0 && (double)sample_nodes * std::pow(8.0, cl) > 2147483648.0) --cl;
  if (const char* e = c->opt("SWZ_MD_COARSEN")) {
    const double thr = c->opt("SWZ_MD_COARSEN_MIN") ? atof(c->opt("SWZ_MD_COARSEN_MIN")) : 32.0;
    if (avg / std::pow(8.0, cl) >= thr) cl = std::max(0, cl - atoi(e));
  }
  const uint64_t cells_per_node = 1ull << (3 * cl);

  MdArgs a{};
  a.akey = as.akey;
  a.aidx = as.aidx;
  a.m = m;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.nstart = lb.nstart;
  a.X = sp.X;
  a.Y = sp.Y;
  a.Z = sp.Z;
  a.taken = lb.taken;
  a.counters = lb.counters;
  a.cell_levels = (uint32_t)cl;
  a.cells_per_node = cells_per_node;
  a.cell_shift = (plan.node_shift == 63u ? 63u : plan.node_shift) - 3u * (uint32_t)cl;
  a.sq_spacing = plan.sq_spacing;
  a.all_sampled = sample_nodes == nnodes ? 1u : 0u;
  {
    // absolute octree level of a cell is level + cl; up to 4 further levels of the key give the slabs
    const int cell_abs = plan.level + cl;
    const int sub = std::max(0, std::min(4, 20 - cell_abs));
    a.sub_levels = (uint32_t)sub;
    const double ext[3] = {plan.root.maxx - plan.root.minx, plan.root.maxy - plan.root.miny,
                           plan.root.maxz - plan.root.minz};
    for (int ax = 0; ax < 3; ++ax) {
      const double u = std::ldexp(ext[ax], -(plan.level + 1 + cl + sub));
      a.usq[ax] = u * u;
    }
    a.cull_sq = plan.sq_spacing * (1.0 + 0x1.0p-18);
    a.ff_min = c->opt("SWZ_MD_FF_MIN") ? (uint32_t)atoi(c->opt("SWZ_MD_FF_MIN")) : 1024u;
    a.xcd_chunks = c->opt("SWZ_MD_XCD") ? (uint32_t)atoi(c->opt("SWZ_MD_XCD")) & 1u : 0u;
    a.ablate = c->opt("SWZ_MD_ABLATE") ? (uint32_t)atoi(c->opt("SWZ_MD_ABLATE")) : 0u;
    // expected points per spacing-sized cell; far below one almost every candidate is accepted
    a.batch_blockers = (avg / std::pow(8.0, plan.cell_levels_geo) < 0.25) ? 1u : 0u;
  }

  uint32_t* snode = nullptr;
  SWZ_TRY(c->get("md_snode", (size_t)nnodes, &snode));
  a.snode_of = snode;
  hipLaunchKernelGGL(md_node_flag_kernel, dim3(div_up(nnodes, 256)), dim3(256), 0, c->stream, lb.nmode, nnodes, snode);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, snode, snode, nnodes, nullptr, "mdn"));
  if (c->md_shard_root && plan.level == -1) {
    // The root of a batch sharded over the GPUs of one process (swz_group): every shard sweeps the cells of its own
    // octants, on keys, with the same cells everywhere (the finest ones: what a shard sees of the cloud must not decide).
    bool used = false;
    SWZ_TRY(min_distance_keys_level(c, plan, as, sp, lb, nnodes, sample_nodes, sample_points, snode, plan.cell_levels_geo, pop[0], rounds_out, &used,
                                    static_cast<const MdShardRoot*>(c->md_shard_root)));
    if (!used) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE root of a sharded batch: the joint sweep needs a level that can be decided on keys");
    return SWZ_OK;
  }
  {
    // sparse levels (about one point per spacing-sized cell or fewer): one thread per point
    bool used = false;
    SWZ_TRY(min_distance_sparse_level(c, plan, as, sp, lb, snode, sample_nodes == nnodes, nnodes, sample_nodes, sample_points, occupied, rounds_out, &used));
    if (used) return SWZ_OK;
  }
  {
    // dense levels whose spacing spans enough key cells: the frontier sweep on key coordinates (swz_mdkeys.hip)
    bool used = false;
    SWZ_TRY(min_distance_keys_level(c, plan, as, sp, lb, nnodes, sample_nodes, sample_points, snode, cl,
                                    pop[std::min(3, plan.cell_levels_geo - cl)], rounds_out, &used));
    if (used) return SWZ_OK;
  }
  if (!sp.X) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE: this level needs the positions in Morton order and they were not gathered");
  if (as.aidx) {  // below the root the survivors are a subsequence: bring their positions into active order
    double *ax = nullptr, *ay = nullptr, *az = nullptr;
    // the two big per-point buffers are shared with the sparse path (never live at the same time): "md_pos" = x[], y[],
    // z[] here (24 B per point), {x,y,z,key} records there (32 B), "md_acc" the same sizes.  Each path asks for what it
    // uses: at 1 B clustered points the 2 x 8 GB between the two decide whether a level with 226 M cells fits
    SWZ_TRY(c->get("md_pos", (size_t)m * 3, &ax));
    ay = ax + m;
    az = ay + m;
    hipLaunchKernelGGL(md_gather_active_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, as.aidx, m, sp.X, sp.Y,
                       sp.Z, ax, ay, az);
    SWZ_LAUNCH_CHECK(c);
    a.X = ax;
    a.Y = ay;
    a.Z = az;
  }
  ProfScope ps(c, "sample_min_distance", (uint64_t)sample_points * 33ull, 1);

  // cells = runs of the cell prefix inside sampled nodes: count them, size the per-cell arrays, build them
  uint32_t* d_cell_sums = nullptr;
  SWZ_TRY(fused_scan_sums(c, CellHeadF{a}, m, lb.counters + CTR_NUM_CELLS, "mdc", &d_cell_sums));
  uint32_t ncells = 0;
  SWZ_HIP(c, hipMemcpyAsync(&ncells, lb.counters + CTR_NUM_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (ncells == 0) return SWZ_OK;

  uint32_t* cellbuf = nullptr;  // 2 per-cell u32 arrays + the packed records
  SWZ_TRY(c->get("md_cells", (size_t)ncells * 2, &cellbuf));
  uint32_t** fields[] = {&a.crel, &a.csnode};
  for (size_t f = 0; f < 2; ++f) *fields[f] = cellbuf + f * (size_t)ncells;
  SWZ_TRY(c->get("md_sleeper", (size_t)ncells * 28, &a.sleeper));  // 27 directions, rows of 14 x 16 bytes
  SWZ_HIP(c, memset_large(a.sleeper, 0xFF, (size_t)ncells * 28 * sizeof(uint2), c->stream));
  SWZ_TRY(c->get("md_result", (size_t)ncells * 2, &a.result));  // a round's queue never holds more than all cells
  SWZ_TRY(c->get("md_cell128", (size_t)ncells, &a.cells));
  SWZ_TRY(c->get("md_nbr_id", (size_t)ncells * 27, &a.nbr_id));
  SWZ_TRY(c->get("md_nbr_slot", (size_t)ncells * 32, &a.nbr_slot));
  SWZ_TRY(c->get("md_acc", (size_t)m * 3, &a.acc_xyz));  // 3 doubles per point (see md_pos above)
  SWZ_TRY(c->get("md_queue0", (size_t)ncells, &a.queue[0]));
  SWZ_TRY(c->get("md_queue1", (size_t)ncells, &a.queue[1]));
  const uint64_t grid_entries = (uint64_t)sample_nodes * cells_per_node;
  SWZ_TRY(c->get("md_gridmap", (size_t)grid_entries, &a.gridmap));
  SWZ_HIP(c, memset_large(a.gridmap, 0xFF, (size_t)grid_entries * 4, c->stream));

  SWZ_TRY(fused_scan_apply(c, CellHeadF{a}, CellBuildG{a}, m, d_cell_sums));
  const uint32_t cb = div_up(ncells, 256);
  hipLaunchKernelGGL(md_cell_end_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  SWZ_STAGE(c, "md cells");
  hipLaunchKernelGGL(md_nbr_build_kernel,
                     dim3(std::min<uint32_t>(div_up(ncells, 8), c->opt("SWZ_MD_NBR_GRID") ? (uint32_t)atoi(c->opt("SWZ_MD_NBR_GRID")) : 1u << 20)),
                     dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  SWZ_STAGE(c, "md neighbour tables");
  // With many small cells a level is bound by activation throughput: start lazily and let a stalled cell sleep
  // until the whole blocking cell is finished (fewer, later wake-ups; measured at 1 B points, level 1:
  // 302 -> 173 ms).  With few large cells it is bound by the latency of a round times the number of rounds:
  // wake up as early as possible.  The latest-first scan order (from before blocker scans could tell dead points)
  // no longer pays and stays off; it remains selectable for the scheduling tests.  (A cheap re-check of the stalled
  // candidate before the full activation, from the same time, was removed in round 2: it cost the kernel registers.)
  const bool many_small = ncells >= (4u << 20) && (double)sample_points / (double)ncells <= 128.0;
  a.latest_first = 0;
  a.patient = many_small ? 1u : 0u;
  if (const char* e = c->opt("SWZ_MD_PATIENT")) a.patient = (uint32_t)atoi(e);
  if (const char* e = c->opt("SWZ_MD_LATEST_FIRST")) a.latest_first = (uint32_t)atoi(e);
  // every level starts lazily (no first round in which all cells scan their neighbourhood only to learn that
  // they must wait); levels bound by the number of rounds wake a cell as soon as its latest earlier neighbour has
  // decided its first point (1 B points, root: 147 -> 117 ms), throughput-bound ones once half of it is decided
  // -- unless the typical point sits in a very large cell (dense blobs, N x denser roots of sharded batches),
  // where an early wake-up only adds expensive activations
  const double typical = pop[std::min(3, plan.cell_levels_geo - cl)];
  bool lazy = many_small || typical <= 1024.0;
  if (const char* e = c->opt("SWZ_MD_LAZY")) lazy = atoi(e) != 0;
  a.lazy_frac = c->opt("SWZ_MD_LAZY_FRAC") ? (float)atof(c->opt("SWZ_MD_LAZY_FRAC")) : (many_small ? 0.5f : 0.0f);
  // two builds of the sweep: for cells of hundreds of points and more (four chunks per memory round trip in blocker
  // scans and the fast-forward, 4 wavefronts per SIMD) and for levels of small cells (neither, 5 per SIMD)
  bool big_cells = typical > 128.0;
  if (const char* e = c->opt("SWZ_MD_BIG")) big_cells = atoi(e) != 0;
  // Node groups.  The nodes of a level are sampled independently, so their cells can be dealt to G sets that run their
  // rounds on G streams: while one set's second launch publishes (a few dependent round trips on a few thousand
  // cells) or its queue is short (the ramps at the start and the end of a level), the other sets' sweeps use the GPU.
  uint32_t groups = 1;
  if (sample_nodes >= 2 && !big_cells) groups = 2;
  if (const char* e = c->opt("SWZ_MD_GROUPS")) groups = (uint32_t)std::max(1, std::min(8, atoi(e)));
  groups = std::min(groups, sample_nodes);
  if (c->opt("SWZ_MD_PERSISTENT") && atoi(c->opt("SWZ_MD_PERSISTENT")) != 0) groups = 1;  // one launch runs all cells
  std::vector<MdArgs> ga(groups, a);
  std::vector<hipStream_t> gs(groups, c->stream);
  for (uint32_t g = 0; g < groups; ++g) {
    ga[g].group = g;
    ga[g].groups = groups;
    if (g > 0) {
      const std::string tag = std::to_string(g);
      SWZ_TRY(c->get(("md_queue0_g" + tag).c_str(), (size_t)ncells, &ga[g].queue[0]));
      SWZ_TRY(c->get(("md_queue1_g" + tag).c_str(), (size_t)ncells, &ga[g].queue[1]));
      SWZ_TRY(c->get(("md_result_g" + tag).c_str(), (size_t)ncells * 2, &ga[g].result));
      SWZ_TRY(c->get(("md_counters_g" + tag).c_str(), (size_t)CTR_COUNT, &ga[g].counters));
      SWZ_HIP(c, hipMemsetAsync(ga[g].counters, 0, CTR_COUNT * sizeof(uint32_t), c->stream));
      while (c->aux_streams.size() < g) {
        hipStream_t st = nullptr;
        SWZ_HIP(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c->aux_streams.push_back(st);
      }
      gs[g] = c->aux_streams[g - 1];
    }
    SWZ_HIP(c, hipMemsetAsync(ga[g].counters + CTR_Q0, 0, 4, c->stream));
    if (lazy)
      hipLaunchKernelGGL(md_lazy_start_kernel, dim3(cb), dim3(256), 0, c->stream, ga[g], ncells, ga[g].queue[0], ga[g].counters + CTR_Q0);
    else
      hipLaunchKernelGGL(md_fill_queue_kernel, dim3(cb), dim3(256), 0, c->stream, ga[g], ncells, ga[g].queue[0], ga[g].counters + CTR_Q0);
    SWZ_LAUNCH_CHECK(c);
  }
  a = ga[0];

  // rounds; the host only looks at the done counter every `batch` rounds
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[swz] MIN_DISTANCE level %d starts: %u pts in %u nodes, cell levels %d of %d, %u cells (occupied at the finest: %u), "
                    "points-weighted mean cell population at the finest level and coarser %.0f / %.0f / %.0f / %.0f, lazy %d patient %u\n",
            plan.level, sample_points, sample_nodes, cl, plan.cell_levels_geo, ncells, occupied[plan.cell_levels_geo], pop[0], pop[1],
            pop[2], pop[3], (int)lazy, a.patient);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (dbg) {
    ev0 = c->take_event();
    ev1 = c->take_event();
    (void)hipEventRecord(ev0, c->stream);
  }
  // grid-stride kernels: whole multiples of the resident workgroups (256 CUs x 5 per CU) avoid a ragged tail
  uint32_t sweep_cap = 2560u * (256u / MD_THREADS), commit_cap = 256u;
  if (const char* e = c->opt("SWZ_MD_GRID")) {
    sweep_cap = (uint32_t)atoi(e);
    commit_cap = std::max(1u, sweep_cap / 2u);
    sweep_cap *= 256u / MD_THREADS;
  }
  // (the sets of node groups share the GPU: each gets its part of the grid)
  const uint32_t sweep_grid = std::min<uint32_t>(std::max(8u, sweep_cap / groups), std::max<uint32_t>(1u, div_up(ncells, MD_WAVES)));
  const uint32_t commit_grid = std::min<uint32_t>(std::max(1u, commit_cap / groups), std::max<uint32_t>(1u, div_up(ncells, 256)));
  uint32_t round = 0, done = 0;
  // SWZ_MD_PERSISTENT=1: one resident workgroup per CU runs the rounds inside one launch (md_persistent_kernel).
  // Measured at 1 B points (round 2): root 169 ms against 117 ms with three launches per round (128 against 89 us
  // per round), level 0 199 against 129 ms -- a round is bound by its ~15 dependent memory round trips, not by the
  // launch boundaries (1.5-2 us each), and an agent-scope release (L2 write-back) per workgroup and barrier costs
  // more than they do.  Off by default; kept selectable for the scheduling tests.
  bool persistent = c->opt("SWZ_MD_PERSISTENT") && atoi(c->opt("SWZ_MD_PERSISTENT")) != 0;
  if (persistent) {
    int dev = 0, cus = 0, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, md_persistent_kernel, MDP_THREADS, 0) != hipSuccess || per_cu < 1 || cus < 1)
      persistent = false;
    MdBarrier* bar = nullptr;
    if (persistent) SWZ_TRY(c->get("md_barrier", sizeof(MdBarrier), reinterpret_cast<void**>(&bar)));
    const uint32_t per_launch = c->opt("SWZ_MD_ROUNDS_PER_LAUNCH") ? (uint32_t)atoi(c->opt("SWZ_MD_ROUNDS_PER_LAUNCH")) : 4096u;
    while (persistent && done < ncells) {
      SWZ_HIP(c, hipMemsetAsync(bar, 0, sizeof(MdBarrier), c->stream));
      hipLaunchKernelGGL(md_persistent_kernel, dim3((uint32_t)cus), dim3(MDP_THREADS), 0, c->stream, a, ncells, round,
                         per_launch, bar);
      SWZ_LAUNCH_CHECK(c);
      MdBarrier hb{};
      SWZ_HIP(c, hipMemcpyAsync(&hb, bar, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipMemcpyAsync(&done, lb.counters + CTR_DONE_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      round += hb.rounds;
      if (hb.timeout) {  // a round may have been left half done: its steps are idempotent per cell only as a whole
        return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE: grid barrier timed out (a workgroup of the persistent launch was not resident)");
      }
      if (round > 4ull * m + 1024) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE frontier sweep did not terminate");
    }
  }
  // rounds queued between two looks at the level's progress (the streams drain for the look: 30-50 us): 32 at first,
  // 128 once a level has shown that it takes hundreds of rounds (a level that is done keeps running the queued rounds,
  // empty, 15 us each); SWZ_MD_BATCH fixes the number
  uint32_t batch = 32, batches_done = 0;
  const bool fixed_batch = c->opt("SWZ_MD_BATCH") != nullptr;
  if (fixed_batch) batch = std::max(1u, (uint32_t)atoi(c->opt("SWZ_MD_BATCH")));
  // a level that does not finish is reported, not waited for: points that change while they are being tiled
  // (keys and positions no longer agree) can make single cells arbitrarily expensive
  const auto wall0 = std::chrono::steady_clock::now();
  double wall_limit = 900.0;
  if (const char* e = c->opt("SWZ_MD_TIME_LIMIT")) wall_limit = atof(e);
  uint64_t max_rounds = 4ull * m + 1024;
  if (const char* e = c->opt("SWZ_MD_ROUND_LIMIT")) max_rounds = (uint64_t)atoll(e);
  // the side streams start after everything queued so far and are joined again when the level is done
  hipEvent_t fork = nullptr;
  if (groups > 1) {
    fork = c->take_event();
    SWZ_HIP(c, hipEventRecord(fork, c->stream));
    for (uint32_t g = 1; g < groups; ++g) SWZ_HIP(c, hipStreamWaitEvent(gs[g], fork, 0));
  }
  std::vector<uint32_t> gdone(groups, 0);
  while (done < ncells) {
    for (uint32_t b = 0; b < batch; ++b, ++round) {
      for (uint32_t g = 0; g < groups; ++g) {
        if (a.batch_blockers)
          hipLaunchKernelGGL((md_sweep_kernel<1, true>), dim3(sweep_grid), dim3(MD_THREADS), 0, gs[g], ga[g], round);
        else if (big_cells)
          hipLaunchKernelGGL((md_sweep_kernel<4, false>), dim3(sweep_grid), dim3(MD_THREADS), 0, gs[g], ga[g], round);
        else
          hipLaunchKernelGGL((md_sweep_kernel<1, false>), dim3(sweep_grid), dim3(MD_THREADS), 0, gs[g], ga[g], round);
        hipLaunchKernelGGL(md_commit_requeue_kernel, dim3(commit_grid), dim3(256), 0, gs[g], ga[g], round);
      }
    }
    SWZ_LAUNCH_CHECK(c);
    if (!fixed_batch && ++batches_done % 4u == 0u && batch < 128u) batch *= 2u;
    for (uint32_t g = 0; g < groups; ++g)
      SWZ_HIP(c, hipMemcpyAsync(&gdone[g], ga[g].counters + CTR_DONE_CELLS, 4, hipMemcpyDeviceToHost, gs[g]));
    done = 0;
    for (uint32_t g = 0; g < groups; ++g) {
      SWZ_HIP(c, hipStreamSynchronize(gs[g]));
      done += gdone[g];
    }
    if (dbg && c->opt("SWZ_MD_TIMELINE")) {
      float t = 0.f;
      hipEvent_t e = c->take_event();
      (void)hipEventRecord(e, c->stream);
      (void)hipEventSynchronize(e);
      (void)hipEventElapsedTime(&t, ev0, e);
      c->event_pool.push_back(e);
      uint32_t qn[3] = {0, 0, 0};
      (void)hipMemcpy(qn, lb.counters + CTR_Q0, 12, hipMemcpyDeviceToHost);
      fprintf(stderr, " r%u:%.1fms:%.3f%%:q%u", round, t, 100.0 * done / ncells, qn[round % 3]);
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > wall_limit) {
      char msg[200];
      snprintf(msg, sizeof(msg), "MIN_DISTANCE frontier sweep exceeded %.0f s: level %d, %u of %u cells done after %u rounds "
               "(were the points modified while they were being tiled?)", wall_limit, plan.level, done, ncells, round);
      return c->fail(SWZ_ERR_INTERNAL, msg);
    }
    if (round > max_rounds) {
      char msg[160];
      snprintf(msg, sizeof(msg), "MIN_DISTANCE frontier sweep did not terminate: level %d, %u of %u cells done after %u rounds",
               plan.level, done, ncells, round);
      return c->fail(SWZ_ERR_INTERNAL, msg);
    }
  }
  if (fork) c->event_pool.push_back(fork);
  if (rounds_out) *rounds_out += round;
  if (dbg) {
    float ms = 0.f;
    (void)hipEventRecord(ev1, c->stream);
    (void)hipEventSynchronize(ev1);
    (void)hipEventElapsedTime(&ms, ev0, ev1);
    c->event_pool.push_back(ev0);
    c->event_pool.push_back(ev1);
    fprintf(stderr, "[swz] MIN_DISTANCE level %d sweep: %.2f ms\n", plan.level, ms);
    uint32_t h[CTR_COUNT];
    SWZ_HIP(c, hipMemcpy(h, lb.counters, sizeof(h), hipMemcpyDeviceToHost));
    fprintf(stderr, "[swz] MIN_DISTANCE level %d: %u pts in %u nodes, cell_levels %d, %u cells, %u rounds\n", plan.level, sample_points,
            sample_nodes, cl, ncells, round);
#ifdef SWZ_MD_STATS
    {
      const double ns = std::max(1u, h[CTR_DBG_HIST]);
      fprintf(stderr, "[swz]   sampled %u activations: mean %.2f us (max %.1f): prologue %.2f, chunks %.2f (%.2f chunks), scans %.2f (%.2f ranks)\n",
              h[CTR_DBG_HIST], h[CTR_DBG_TIME] / ns / 100.0, h[CTR_DBG_TMAX] / 100.0, h[CTR_DBG_HIST + 1] / ns / 100.0,
              h[CTR_DBG_HIST + 2] / ns / 100.0, h[CTR_DBG_HIST + 4] / ns, h[CTR_DBG_HIST + 3] / ns / 100.0, h[CTR_DBG_HIST + 5] / ns);
    }
#endif
  }
  return SWZ_OK;
}

}  // namespace swz
