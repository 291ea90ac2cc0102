// swz_mdprop.hip -- MIN_DISTANCE in "property" mode (SWZ_FLAG_MIN_DISTANCE_PROPERTY).
//
// The reference's PoissonDiskSampling (core/tiling/Sampling.h:421-471 + SparseGrid.cpp:116-146) is the greedy
// maximal independent set in Morton order; computing exactly that set needs thousands of dependent rounds per level
// (swz_mindist.hip).  What the sampler is FOR -- and what the reference's author checks, test/TestTiler.cpp:361-421 --
// is the property: inside a sampled node no two taken points are closer than the node's spacing, and (greedy sets
// being maximal) every point left out is closer than the spacing to a taken one.  This file computes a set with
// exactly that property, with the reference's arithmetic for the compare (squared double distance < float-squared
// spacing widened to double, GridCell.cpp:43-58), deterministically, in EIGHT phases per level:
//
//   * every sampled node is cut into octree cells at least one spacing wide (runs of the sorted keys), so only
//     points of the same or of adjacent cells can conflict;
//   * the colour of a cell is the parity of its coordinates = its last octant digit; cells of one colour are never
//     adjacent, so they decide independently: phase k lets every cell of colour k run the greedy rule over its own
//     points in Morton order against the points taken so far in its 26 neighbours (colours < k) and in itself.
//
// The result is the greedy set for the priority (colour, Morton order) instead of Morton order alone.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

constexpr uint32_t PM_NONE = 0xFFFFFFFFu;
constexpr int PM_THREADS = 256;
constexpr int PM_WAVES = PM_THREADS / WAVE;
constexpr int PM_WIN = 128;    // taken points of the neighbourhood held in LDS at a time
constexpr int PM_FRESH = 64;  // taken points of the cell itself held in LDS (more spill to memory reads)
constexpr uint32_t PM_BIG = 4096;   // cells with more points: neighbour tests in parallel first (pm_big_reject_kernel)
constexpr uint32_t PM_UNIT = 256;   // points per unit of that pass (one wavefront, four chunks)

struct PmArgs {
  const uint64_t* akey;
  uint32_t m;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* nstart;
  const double* X;  // positions in ACTIVE order
  const double* Y;
  const double* Z;
  uint8_t* taken;
  uint2* cell;        // {first point, end}
  uint32_t* ccnt;     // taken points of the cell so far
  uint32_t* crel;     // cell code inside its node
  uint32_t* csnode;   // index of its node among the sampled nodes
  uint32_t* nbr;      // [cell][27]: adjacent cell per direction slot (13 = itself), PM_NONE when absent
  uint32_t* gridmap;  // [sampled node][cell code] -> cell
  double* acc;        // taken positions of cell c at slots [cell.x, cell.x + ccnt), 3 doubles each
  const uint32_t* snode_of;
  uint32_t* ticket;   // work distribution of the running phase
  uint8_t* rej;       // big cells: 1 = struck out by a taken point of an adjacent cell (pm_big_reject_kernel)
  uint32_t cell_shift;
  uint64_t cells_per_node;
  double sq_spacing;
};

__device__ __forceinline__ bool pm_is_head(const PmArgs& a, uint32_t i) {
  if (a.nmode[a.nid[i]] != MODE_SAMPLE) return false;
  return i == 0 || ((a.akey[i] >> a.cell_shift) != (a.akey[i - 1] >> a.cell_shift));
}
struct PmHeadF {
  PmArgs a;
  __device__ uint32_t operator()(uint32_t i) const { return pm_is_head(a, i) ? 1u : 0u; }
};
struct PmCellG {
  PmArgs a;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t head) const {
    if (!head) return;
    a.cell[excl].x = i;
    a.ccnt[excl] = 0;
    a.crel[excl] = (uint32_t)((a.akey[i] >> a.cell_shift) & (a.cells_per_node - 1ull));
    a.csnode[excl] = a.snode_of[a.nid[i]];
  }
};

__global__ __launch_bounds__(256) void pm_cell_end_kernel(PmArgs a, uint32_t ncells) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t s = a.cell[c].x;
  const uint32_t node_end = a.nstart[a.nid[s] + 1];
  const uint32_t next = (c + 1 < ncells) ? a.cell[c + 1].x : a.m;
  a.cell[c].y = next < node_end ? next : node_end;
  a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + a.crel[c]] = c;
}

// all 26 adjacent cells of every cell (codes by arithmetic on the dilated coordinates), 32 lanes per cell
__global__ __launch_bounds__(256) void pm_nbr_kernel(PmArgs a, uint32_t ncells) {
  for (uint64_t cbase = (uint64_t)blockIdx.x * 8u; cbase < ncells; cbase += (uint64_t)gridDim.x * 8u) {
    const uint32_t c = (uint32_t)cbase + threadIdx.x / 32u;
    const uint32_t k = threadIdx.x & 31u;
    if (c >= ncells || k >= 27u) continue;
    uint32_t out = PM_NONE;
    if (k == 13u) {
      out = c;
    } else {
      const uint32_t rel = a.crel[c];
      const uint32_t all = (uint32_t)(a.cells_per_node - 1ull);
      const uint32_t mz = all & 0x09249249u, my = mz << 1, mx = mz << 2;
      const uint32_t v[3] = {rel & mx, rel & my, rel & mz};
      const uint32_t mk[3] = {mx, my, mz};
      const uint32_t d[3] = {k % 3u, (k / 3u) % 3u, k / 9u};  // 0: minus one, 1: same, 2: plus one
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
      if (inside) out = a.gridmap[(uint64_t)a.csnode[c] * a.cells_per_node + nrel];
    }
    a.nbr[(size_t)c * 27 + k] = out;
  }
}

struct PmLds {
  double wx[PM_WIN], wy[PM_WIN], wz[PM_WIN];
  double fx[PM_FRESH], fy[PM_FRESH], fz[PM_FRESH];
  uint8_t owner[PM_WIN];
};

// window [base, base + PM_WIN) of the flattened list of the neighbourhood's taken points into LDS: lane k < 27 owns
// the n_cnt entries of adjacent cell k, which start at list offset off
__device__ __forceinline__ uint32_t pm_fill_window(const PmArgs& a, PmLds& lds, uint32_t base, uint32_t T, uint32_t maxcnt,
                                                   uint32_t n_cnt, uint32_t n_start, uint32_t off) {
  const uint32_t l = lane_id();
  const uint32_t wn = (T - base) < (uint32_t)PM_WIN ? (T - base) : (uint32_t)PM_WIN;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t j = 0; j < maxcnt; ++j) {
    if (j < n_cnt) {
      const uint32_t ti = off + j;
      if (ti >= base && ti < base + PM_WIN) lds.owner[ti - base] = (uint8_t)l;
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t ti = l; ti < ((wn + WAVE - 1) / WAVE) * WAVE; ti += WAVE) {
    const uint32_t k = ti < wn ? lds.owner[ti] : 0u;
    const uint32_t ks = __shfl(n_start, (int)k, WAVE), ko = __shfl(off, (int)k, WAVE);
    if (ti < wn) {
      const double* src = a.acc + (size_t)(ks + (base + ti - ko)) * 3;
      lds.wx[ti] = src[0];
      lds.wy[ti] = src[1];
      lds.wz[ti] = src[2];
    }
  }
  __builtin_amdgcn_wave_barrier();
  return wn;
}

struct PmHood {  // the taken points of the 26 adjacent cells as one flattened list; lane k < 27 owns cell k's part
  uint32_t n_cnt, n_start, off, T, maxcnt;
};
__device__ __forceinline__ PmHood pm_hood(const PmArgs& a, uint32_t c) {
  const uint32_t l = lane_id();
  PmHood h{0, 0, 0, 0, 0};
  if (l < 27u && l != 13u) {
    const uint32_t nb = a.nbr[(size_t)c * 27 + l];
    if (nb != PM_NONE) {
      h.n_cnt = a.ccnt[nb];
      h.n_start = a.cell[nb].x;
    }
  }
  const uint32_t incl = wave_incl_sum(h.n_cnt);
  h.off = incl - h.n_cnt;
  h.T = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
  h.maxcnt = h.n_cnt;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(h.maxcnt, d, WAVE);
    h.maxcnt = o > h.maxcnt ? o : h.maxcnt;
  }
  return h;
}

// Big cells (dense blobs: tens of thousands of points in one cell): testing every point against the taken points of
// the neighbourhood is the bulk of the work and does not depend on the order inside the cell, so it runs first, 256
// points per wavefront, and leaves a flag per point; the cell's own wavefront then only walks the unflagged points.
__global__ __launch_bounds__(PM_THREADS) void pm_big_reject_kernel(PmArgs a, const uint2* __restrict__ units, uint32_t nunits) {
  __shared__ PmLds lds_all[PM_WAVES];
  const uint32_t w = threadIdx.x / WAVE, l = lane_id();
  PmLds& lds = lds_all[w];
  const double t = a.sq_spacing;
  for (uint32_t u = blockIdx.x * PM_WAVES + w; u < nunits; u += gridDim.x * PM_WAVES) {
    const uint2 unit = units[u];  // {cell, first point}
    const uint32_t c = unit.x;
    const uint32_t e = a.cell[c].y;
    const uint32_t end = unit.y + PM_UNIT < e ? unit.y + PM_UNIT : e;
    const PmHood h = pm_hood(a, c);
    if (h.T == 0) continue;
    bool rej[PM_UNIT / WAVE];
    double px[PM_UNIT / WAVE], py[PM_UNIT / WAVE], pz[PM_UNIT / WAVE];
#pragma unroll
    for (int k = 0; k < (int)(PM_UNIT / WAVE); ++k) {
      const uint32_t p = unit.y + (uint32_t)k * WAVE + l;
      rej[k] = false;
      px[k] = py[k] = pz[k] = 0.0;
      if (p < end) {
        px[k] = a.X[p];
        py[k] = a.Y[p];
        pz[k] = a.Z[p];
      }
    }
    for (uint32_t base = 0; base < h.T; base += PM_WIN) {
      const uint32_t wn = pm_fill_window(a, lds, base, h.T, h.maxcnt, h.n_cnt, h.n_start, h.off);
      for (uint32_t ti = 0; ti < wn; ++ti) {
        const double qx = lds.wx[ti], qy = lds.wy[ti], qz = lds.wz[ti];
#pragma unroll
        for (int k = 0; k < (int)(PM_UNIT / WAVE); ++k)
          if (sq_dist(px[k], py[k], pz[k], qx, qy, qz) < t) rej[k] = true;
      }
    }
#pragma unroll
    for (int k = 0; k < (int)(PM_UNIT / WAVE); ++k) {
      const uint32_t p = unit.y + (uint32_t)k * WAVE + l;
      if (p < end && rej[k]) a.rej[p] = 1;
    }
  }
}
// units of the big cells: first the counts per colour, then the units themselves behind per-colour cursors
__global__ __launch_bounds__(256) void pm_big_count_kernel(PmArgs a, uint32_t ncells, uint32_t* __restrict__ counts) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint2 r = a.cell[c];
  if (r.y - r.x > PM_BIG) atomicAdd(&counts[a.crel[c] & 7u], (r.y - r.x + PM_UNIT - 1u) / PM_UNIT);
}
__global__ __launch_bounds__(256) void pm_big_units_kernel(PmArgs a, uint32_t ncells, uint32_t* __restrict__ cursors,
                                                           uint2* __restrict__ units) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint2 r = a.cell[c];
  if (r.y - r.x <= PM_BIG) return;
  const uint32_t nu = (r.y - r.x + PM_UNIT - 1u) / PM_UNIT;
  const uint32_t at = atomicAdd(&cursors[a.crel[c] & 7u], nu);
  for (uint32_t j = 0; j < nu; ++j) units[at + j] = make_uint2(c, r.x + j * PM_UNIT);
}

// What a wavefront requests for the NEXT cell of its ticket while it works on the current one: the cell record and the
// row of adjacent cells first, then (they depend on those) the adjacent cells' counts and starts and the first 64
// points -- a cell is a chain of five dependent memory round trips otherwise, for often fewer than 64 points.
struct PmPre {
  uint2 me;
  uint32_t nb;              // lane k < 27, k != 13: adjacent cell in direction slot k
  uint32_t n_cnt, n_start;  // ... its taken points
  double px, py, pz;        // the lane's point of the first chunk
};
__device__ __forceinline__ void pm_prefetch1(const PmArgs& a, uint32_t c, PmPre& q) {
  const uint32_t l = lane_id();
  q.me = a.cell[c];
  q.nb = (l < 27u && l != 13u) ? a.nbr[(size_t)c * 27 + l] : PM_NONE;
}
__device__ __forceinline__ void pm_prefetch2(const PmArgs& a, PmPre& q) {
  q.n_cnt = 0;
  q.n_start = 0;
  if (q.nb != PM_NONE) {
    q.n_cnt = a.ccnt[q.nb];
    q.n_start = a.cell[q.nb].x;
  }
  const uint32_t p = q.me.x + lane_id();
  q.px = q.py = q.pz = 0.0;
  if (p < q.me.y) {
    q.px = a.X[p];
    q.py = a.Y[p];
    q.pz = a.Z[p];
  }
}

// one wavefront decides all points of one cell
__device__ void pm_cell(const PmArgs& a, uint32_t c, const PmPre& pre, PmLds& lds) {
  const uint32_t l = lane_id();
  const uint2 me = pre.me;
  const uint32_t s0 = me.x, e = me.y;
  const double t = a.sq_spacing;
  const bool big = e - s0 > PM_BIG;  // the neighbourhood has been dealt with (a.rej)
  PmHood h{0, 0, 0, 0, 0};
  if (!big) {
    h.n_cnt = pre.n_cnt;
    h.n_start = pre.n_start;
    const uint32_t incl = wave_incl_sum(h.n_cnt);
    h.off = incl - h.n_cnt;
    h.T = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    h.maxcnt = h.n_cnt;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t o = __shfl_xor(h.maxcnt, d, WAVE);
      h.maxcnt = o > h.maxcnt ? o : h.maxcnt;
    }
  }
  const uint32_t n_cnt = h.n_cnt, n_start = h.n_start, off = h.off, T = h.T, maxcnt = h.maxcnt;
  uint32_t wn0 = 0;
  if (T > 0) wn0 = pm_fill_window(a, lds, 0, T, maxcnt, n_cnt, n_start, off);
  uint32_t fresh = 0;
  for (uint32_t cur = s0; cur < e; cur += WAVE) {
    const uint32_t p = cur + l;
    const bool valid = p < e;
    double px = pre.px, py = pre.py, pz = pre.pz;
    if (cur != s0) {
      px = py = pz = 0.0;
      if (valid) {
        px = a.X[p];
        py = a.Y[p];
        pz = a.Z[p];
      }
    }
    bool rej = !valid || (big && a.rej[p] != 0);
    if (big && !__ballot(!rej)) continue;  // a stretch struck out completely
    // against the taken points of the adjacent cells, window by window
    for (uint32_t base = 0; base < T; base += PM_WIN) {
      const uint32_t wn = (T <= (uint32_t)PM_WIN) ? wn0 : pm_fill_window(a, lds, base, T, maxcnt, n_cnt, n_start, off);
      for (uint32_t ti = 0; ti < wn; ++ti)
        if (sq_dist(px, py, pz, lds.wx[ti], lds.wy[ti], lds.wz[ti]) < t) rej = true;
      if (!__ballot(!rej)) break;
    }
    // against the points this cell has taken already
    const uint32_t in_lds = fresh < (uint32_t)PM_FRESH ? fresh : (uint32_t)PM_FRESH;
    for (uint32_t ti = 0; ti < in_lds; ++ti)
      if (sq_dist(px, py, pz, lds.fx[ti], lds.fy[ti], lds.fz[ti]) < t) rej = true;
    for (uint32_t ti = PM_FRESH; ti < fresh; ++ti) {  // rare: more than PM_FRESH taken in one cell
      const double* q = a.acc + (size_t)(s0 + ti) * 3;  // stored by this wavefront: read past the L1
      const double qx = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double qy = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double qz = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sq_dist(px, py, pz, qx, qy, qz) < t) rej = true;
    }
    // the survivors of the chunk in Morton order
    uint64_t alive = __ballot(!rej);
    while (alive) {
      const int j = __ffsll((unsigned long long)alive) - 1;
      const int lo = __double2loint(px), hi = __double2hiint(px);
      const double bx = __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
      const int lo2 = __double2loint(py), hi2 = __double2hiint(py);
      const double by = __hiloint2double(__builtin_amdgcn_readlane(hi2, j), __builtin_amdgcn_readlane(lo2, j));
      const int lo3 = __double2loint(pz), hi3 = __double2hiint(pz);
      const double bz = __hiloint2double(__builtin_amdgcn_readlane(hi3, j), __builtin_amdgcn_readlane(lo3, j));
      if ((int)l == j) {
        a.taken[cur + (uint32_t)j] = 1;
        double* dst = a.acc + (size_t)(s0 + fresh) * 3;
        dst[0] = px;
        dst[1] = py;
        dst[2] = pz;
        if (fresh < (uint32_t)PM_FRESH) {
          lds.fx[fresh] = px;
          lds.fy[fresh] = py;
          lds.fz[fresh] = pz;
        }
      }
      ++fresh;
      if ((int)l > j && !rej && sq_dist(px, py, pz, bx, by, bz) < t) rej = true;
      alive = __ballot(!rej && (int)l > j);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (l == 0) a.ccnt[c] = fresh;
}

// phase `colour`: wavefronts pull cells of this colour from its list, sixteen per ticket (one atomic word serves
// only ~90 tickets per microsecond)
__global__ __launch_bounds__(PM_THREADS) void pm_phase_kernel(PmArgs a, const uint32_t* __restrict__ list, uint32_t count) {
  __shared__ PmLds lds[PM_WAVES];
  const uint32_t w = threadIdx.x / WAVE;
  for (;;) {
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(a.ticket, 16u);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (base >= count) return;
    const uint32_t end = base + 16u < count ? base + 16u : count;
    PmPre nxt;
    uint32_t cn = list[base];
    pm_prefetch1(a, cn, nxt);
    pm_prefetch2(a, nxt);
    for (uint32_t i = base; i < end; ++i) {
      const PmPre cur = nxt;
      const uint32_t c = cn;
      if (i + 1 < end) {
        cn = list[i + 1];
        pm_prefetch1(a, cn, nxt);
        pm_prefetch2(a, nxt);
      }
      pm_cell(a, c, cur, lds[w]);
    }
  }
}

// cells by colour (any order inside a colour: its cells are independent of each other)
__global__ __launch_bounds__(256) void pm_colour_lists_kernel(PmArgs a, uint32_t ncells, uint32_t* __restrict__ lists,
                                                              uint32_t* __restrict__ counts) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  const uint32_t colour = c < ncells ? (a.crel[c] & 7u) : 8u;
  for (uint32_t k = 0; k < 8u; ++k) {
    const uint64_t mk = __ballot(colour == k);
    if (!mk) continue;
    const int leader = __ffsll((unsigned long long)mk) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(&counts[k], (uint32_t)__popcll(mk));
    base = __shfl(base, leader, WAVE);
    if (colour == k) lists[(size_t)k * ncells + base + (uint32_t)__popcll(mk & lanemask_lt())] = c;
  }
}

__global__ __launch_bounds__(256) void pm_clear_taken_kernel(const uint32_t* __restrict__ nid, const uint8_t* __restrict__ nmode,
                                                             uint32_t m, uint8_t* __restrict__ taken) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < m && nmode[nid[i]] == MODE_SAMPLE) taken[i] = 0;
}
__global__ __launch_bounds__(256) void pm_snode_flag_kernel(const uint8_t* __restrict__ nmode, uint32_t nnodes,
                                                            uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < nnodes) out[j] = nmode[j] == MODE_SAMPLE ? 1u : 0u;
}
__global__ __launch_bounds__(256) void pm_gather_active_kernel(const uint32_t* __restrict__ aidx, uint32_t m,
                                                               const double* __restrict__ X, const double* __restrict__ Y,
                                                               const double* __restrict__ Z, double* __restrict__ ax,
                                                               double* __restrict__ ay, double* __restrict__ az) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t s = aidx[i];
  ax[i] = X[s];
  ay[i] = Y[s];
  az[i] = Z[s];
}
// occupied cells per candidate cell level (same counting as the exact path's md_cell_hist_kernel)
__global__ __launch_bounds__(256) void pm_cell_hist_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ nid,
                                                           const uint8_t* __restrict__ nmode, uint32_t m, uint32_t node_shift,
                                                           uint32_t cl_geo, uint32_t skip, uint32_t* __restrict__ hist) {
  __shared__ uint32_t lh[16];
  if (threadIdx.x < 16) lh[threadIdx.x] = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint64_t i0 = (uint64_t)blockIdx.x * 256u * skip; i0 < m; i0 += (uint64_t)gridDim.x * 256u * skip) {  // every skip-th tile
    const uint32_t i = (uint32_t)i0 + threadIdx.x;
    uint32_t bin = 0xFFu;
    if (i < m && nmode[nid[i]] == MODE_SAMPLE) {
      if (i == 0 || nid[i - 1] != nid[i]) {
        bin = 0;
      } else if (cl_geo) {
        const uint64_t diff = ((akey[i] ^ akey[i - 1]) >> (node_shift - 3u * cl_geo)) & ((1ull << (3u * cl_geo)) - 1ull);
        if (diff) bin = cl_geo - (uint32_t)(63 - __clzll((unsigned long long)diff)) / 3u;
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

// points-weighted mean cell population at cell level cl_geo (out[0]) and one level coarser (out[1]); out[2] = samples:
// the population of the cell of every 65536-th point, by binary search for the cell's run in the sorted keys
__global__ __launch_bounds__(256) void pm_cell_pop_kernel(const uint64_t* __restrict__ akey, const uint32_t* __restrict__ nid,
                                                          const uint8_t* __restrict__ nmode, uint32_t m, uint32_t node_shift,
                                                          uint32_t cl_geo, unsigned long long* __restrict__ out) {
  const uint32_t tsample = blockIdx.x * 256 + threadIdx.x;
  if (tsample >= 65536u) return;
  const uint32_t i = (uint32_t)(((uint64_t)tsample * m) / 65536u);
  if (i >= m || nmode[nid[i]] != MODE_SAMPLE) return;
  const uint64_t key = akey[i];
  for (uint32_t k = 0; k <= 1u && k <= cl_geo; ++k) {
    const uint32_t sh = node_shift - 3u * (cl_geo - k);
    const uint64_t pre = key >> sh;
    uint32_t lo = 0, hi = i;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if ((akey[mid] >> sh) < pre) lo = mid + 1u; else hi = mid;
    }
    const uint32_t first = lo;
    lo = i;
    hi = m;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2u;
      if ((akey[mid] >> sh) <= pre) lo = mid + 1u; else hi = mid;
    }
    atomicAdd(&out[k], (unsigned long long)(lo - first));
  }
  atomicAdd(&out[2], 1ull);
}

int min_distance_property_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                                const LevelBuffers& lb, uint32_t nnodes, uint32_t sample_nodes, uint32_t sample_points,
                                uint32_t* phases_out) {
  const uint32_t m = as.m;
  const uint32_t nsh = plan.node_shift;
  const auto wall0 = std::chrono::steady_clock::now();
  c->next_scratch_epoch();  // what the level before asked for ("md_*", "sp_*", "pm_*") may go if memory runs out
  uint32_t occupied[12] = {0};
  bool sampled_hist = false;
  auto count_cells = [&](bool exact) -> int {
    uint32_t* d_hist = nullptr;
    SWZ_TRY(c->get("md_hist", (size_t)16, &d_hist));
    SWZ_HIP(c, hipMemsetAsync(d_hist, 0, 64, c->stream));
    // (about 8 M points are looked at on large levels: the counts only steer the choice of path and cell size -- the
    // coloured phases below size buffers by them and count again, exactly)
    const uint32_t skip = exact ? 1u : std::max(1u, m >> 23);
    sampled_hist = skip > 1u;
    const uint32_t tiles = div_up(m, 256), sampled_tiles = div_up(tiles, skip);
    hipLaunchKernelGGL(pm_cell_hist_kernel, dim3(std::min<uint32_t>(sampled_tiles, 4096u)), dim3(256), 0, c->stream,
                       as.akey, lb.nid, lb.nmode, m, nsh, (uint32_t)plan.cell_levels_geo, skip, d_hist);
    SWZ_LAUNCH_CHECK(c);
    uint32_t h[16];
    SWZ_HIP(c, hipMemcpyAsync(h, d_hist, 64, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    const double scale = skip == 1 ? 1.0 : (double)m / (double)std::min<uint64_t>(m, (uint64_t)sampled_tiles * 256u);
    double run = 0;
    for (int b = 0; b < 12; ++b) {
      run += h[b];
      occupied[b] = (uint32_t)std::min<double>(run * scale, (double)m);
    }
    return SWZ_OK;
  };
  SWZ_TRY(count_cells(false));
  // Cell size: coarser cells mean fewer, better filled cells, but every point is tested against the taken points of
  // 27 cells: keep the expected number of taken points per cell small (<= 8).  A cell of side r spacings holds at
  // most about 0.75 r^3 points that are pairwise a spacing apart (and never more than it has points).
  const double node_ext = (plan.root.maxx - plan.root.minx) / std::pow(2.0, plan.level + 1);
  const double r0 = node_ext / std::pow(2.0, plan.cell_levels_geo) / plan.spacing_node;  // finest cells, in spacings
  // One wavefront per cell.  Coarser cells are better filled, but a cell of side r spacings can hold about
  // 0.75 r^3 taken points and every point is tested against those of 27 cells: go one level coarser only when the
  // finest cells are poorly filled and the coarser ones still hold few taken points WHATEVER their population (real
  // data is clustered: an average says nothing about the dense parts).
  int cl = plan.cell_levels_geo;
  if (cl > 0 && (double)sample_points / (double)std::max(1u, occupied[cl]) < 24.0 && 0.75 * 8.0 * r0 * r0 * r0 <= 48.0) {
    // ... and only while the TYPICAL point would not sit in an oversized cell afterwards (points-weighted mean
    // population: a dense blob in a sparse background keeps the plain average low)
    unsigned long long* d_pop = nullptr;
    SWZ_TRY(c->get("md_pop", (size_t)8, &d_pop));
    SWZ_HIP(c, hipMemsetAsync(d_pop, 0, 64, c->stream));
    hipLaunchKernelGGL(pm_cell_pop_kernel, dim3(256), dim3(256), 0, c->stream, as.akey, lb.nid, lb.nmode, m, nsh,
                       (uint32_t)plan.cell_levels_geo, d_pop);
    SWZ_LAUNCH_CHECK(c);
    unsigned long long hp[3];
    SWZ_HIP(c, hipMemcpyAsync(hp, d_pop, 24, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (hp[2] && (double)hp[1] / (double)hp[2] <= 1024.0) --cl;
  }
  while (cl > 0 && (double)sample_nodes * std::pow(8.0, cl) > 2147483648.0) --cl;
  const double pts_per_cell = (double)sample_points / (double)std::max(1u, occupied[cl]);
  const uint64_t cells_per_node = 1ull << (3 * cl);

  PmArgs a{};
  a.akey = as.akey;
  a.m = m;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.nstart = lb.nstart;
  a.X = sp.X;
  a.Y = sp.Y;
  a.Z = sp.Z;
  a.taken = lb.taken;
  a.cell_shift = nsh - 3u * (uint32_t)cl;
  a.cells_per_node = cells_per_node;
  a.sq_spacing = plan.sq_spacing;

  uint32_t* snode = nullptr;
  SWZ_TRY(c->get("md_snode", (size_t)nnodes, &snode));
  a.snode_of = snode;
  hipLaunchKernelGGL(pm_snode_flag_kernel, dim3(div_up(nnodes, 256)), dim3(256), 0, c->stream, lb.nmode, nnodes, snode);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, snode, snode, nnodes, nullptr, "mdn"));
  {
    // Sparse levels (about one point per spacing-sized cell or fewer): the exact Morton-order greedy needs only a few
    // dependent rounds there and one thread per point beats one wavefront per (nearly empty) cell -- the exact set
    // has the property a fortiori (swz_mdsparse.hip).
    bool used = false;
    SWZ_TRY(min_distance_sparse_level(c, plan, as, sp, lb, snode, sample_nodes == nnodes, nnodes, sample_nodes, sample_points, occupied, phases_out, &used));
    if (used) return SWZ_OK;
    // it may have given up half way (locally dense data): its decisions are those of ANOTHER priority order.  (A level it
    // declined at the door -- two or more points per occupied cell, the same test as in swz_mdsparse.hip -- is untouched.)
    int scl = plan.cell_levels_geo;
    while (scl > 0 && (double)sample_nodes * std::pow(8.0, scl) > 2147483648.0) --scl;
    double limit = 2.0;
    if (const char* e = c->opt("SWZ_MD_SPARSE_LIMIT")) limit = atof(e);
    if ((double)sample_points / (double)std::max(1u, occupied[scl]) < limit) {
      hipLaunchKernelGGL(pm_clear_taken_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, lb.nid, lb.nmode, m, lb.taken);
      SWZ_LAUNCH_CHECK(c);
    }
  }
  if (key_metric(c, plan, sp).ok) {
    // On key coordinates (cubic bounds, as the Tiler's are): a maximal independent set grown in data-parallel rounds, no
    // positions in Morton order, no dependent phases (swz_mdrounds.hip).  Levels it does not take get the exact set
    // below -- it has the property a fortiori and decides on the keys as well.
    bool used = false;
    SWZ_TRY(min_distance_rounds_level(c, plan, as, sp, lb, nnodes, sample_nodes, sample_points, snode, phases_out, &used));
    if (used) return SWZ_OK;
    if (!sp.X) return min_distance_level(c, plan, as, sp, lb, nnodes, sample_nodes, sample_points, phases_out);
  }
  if (sampled_hist) SWZ_TRY(count_cells(true));  // the coloured phases size their per-cell arrays by these counts
  if (as.aidx) {  // below the root the survivors are a subsequence: positions into active order
    double* ax = nullptr;
    SWZ_TRY(c->get("md_pos", (size_t)m * 4, &ax));
    hipLaunchKernelGGL(pm_gather_active_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, as.aidx, m, sp.X, sp.Y,
                       sp.Z, ax, ax + m, ax + 2 * (size_t)m);
    SWZ_LAUNCH_CHECK(c);
    a.X = ax;
    a.Y = ax + m;
    a.Z = ax + 2 * (size_t)m;
  }
  ProfScope ps(c, "sample_min_distance_property", (uint64_t)sample_points * 33ull, 1);

  // cells = runs of the cell prefix inside sampled nodes; the scan that numbers them writes their records
  const uint32_t max_cells = std::max(1u, std::min(sample_points, occupied[cl] ? occupied[cl] : sample_points));
  uint32_t* cellbuf = nullptr;
  SWZ_TRY(c->get("md_cells", (size_t)max_cells * 11, &cellbuf));
  a.ccnt = cellbuf;
  a.crel = cellbuf + (size_t)max_cells;
  a.csnode = cellbuf + 2 * (size_t)max_cells;
  a.ticket = lb.counters + CTR_Q0;
  uint4* cell4 = nullptr;
  SWZ_TRY(c->get("md_cell4", (size_t)max_cells, &cell4));
  a.cell = reinterpret_cast<uint2*>(cell4);
  SWZ_TRY(c->get("md_nbr_id", (size_t)max_cells * 27, &a.nbr));
  SWZ_TRY(c->get("md_acc", (size_t)m * 4, &a.acc));
  const uint64_t grid_entries = (uint64_t)sample_nodes * cells_per_node;
  SWZ_TRY(c->get("md_gridmap", (size_t)grid_entries, &a.gridmap));
  SWZ_HIP(c, memset_large(a.gridmap, 0xFF, (size_t)grid_entries * 4, c->stream));
  SWZ_TRY(fused_scan(c, PmHeadF{a}, PmCellG{a}, m, lb.counters + CTR_NUM_CELLS, "mdc"));
  uint32_t ncells = 0;
  SWZ_HIP(c, hipMemcpyAsync(&ncells, lb.counters + CTR_NUM_CELLS, 4, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  if (ncells == 0) return SWZ_OK;
  if (ncells > max_cells) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE property mode: more cells than counted");
  const uint32_t cb = div_up(ncells, 256);
  hipLaunchKernelGGL(pm_cell_end_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  hipLaunchKernelGGL(pm_nbr_kernel, dim3(std::min<uint32_t>(div_up(ncells, 8), 1u << 20)), dim3(256), 0, c->stream, a, ncells);
  SWZ_LAUNCH_CHECK(c);
  uint32_t *lists = nullptr, *counts = nullptr;
  SWZ_TRY(c->get("pm_lists", (size_t)ncells * 8, &lists));
  SWZ_TRY(c->get("pm_counts", (size_t)8, &counts));
  SWZ_HIP(c, hipMemsetAsync(counts, 0, 32, c->stream));
  hipLaunchKernelGGL(pm_colour_lists_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells, lists, counts);
  SWZ_LAUNCH_CHECK(c);
  uint32_t h[8];
  SWZ_HIP(c, hipMemcpyAsync(h, counts, 32, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  // big cells: their units of 256 points per colour
  uint32_t ucount[8], uoff[9] = {0};
  uint2* d_units = nullptr;
  {
    SWZ_HIP(c, hipMemsetAsync(counts, 0, 32, c->stream));
    hipLaunchKernelGGL(pm_big_count_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells, counts);
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipMemcpyAsync(ucount, counts, 32, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 8; ++k) uoff[k + 1] = uoff[k] + ucount[k];
    if (uoff[8]) {
      SWZ_TRY(c->get("pm_units", (size_t)uoff[8], &d_units));
      SWZ_TRY(c->get("pm_rej", (size_t)m, &a.rej));
      SWZ_HIP(c, hipMemsetAsync(a.rej, 0, (size_t)m, c->stream));
      SWZ_HIP(c, hipMemcpyAsync(counts, uoff, 32, hipMemcpyHostToDevice, c->stream));  // cursors start at the offsets
      hipLaunchKernelGGL(pm_big_units_kernel, dim3(cb), dim3(256), 0, c->stream, a, ncells, counts, d_units);
      SWZ_LAUNCH_CHECK(c);
    }
  }
  for (uint32_t colour = 0; colour < 8; ++colour) {
    if (!h[colour]) continue;
    if (ucount[colour]) {
      hipLaunchKernelGGL(pm_big_reject_kernel, dim3(std::min<uint32_t>(256u * 4u, div_up(ucount[colour], PM_WAVES))),
                         dim3(PM_THREADS), 0, c->stream, a, d_units + uoff[colour], ucount[colour]);
      SWZ_LAUNCH_CHECK(c);
    }
    const uint32_t grid = std::min<uint32_t>(256u * 8u, std::max(1u, div_up(h[colour], 16u * PM_WAVES)));
    SWZ_HIP(c, hipMemsetAsync(a.ticket, 0, 4, c->stream));
    hipLaunchKernelGGL(pm_phase_kernel, dim3(grid), dim3(PM_THREADS), 0, c->stream, a, lists + (size_t)colour * ncells, h[colour]);
    SWZ_LAUNCH_CHECK(c);
  }
  if (phases_out) *phases_out += 8;
  if (c->opt("SWZ_DEBUG")) {
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "[swz] MIN_DISTANCE property level %d: %u pts in %u nodes, cell levels %d of %d, %u cells (%.1f pts each), "
            "%.1f ms\n", plan.level, sample_points, sample_nodes, cl, plan.cell_levels_geo, ncells, pts_per_cell,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count());
  }
  return SWZ_OK;
}

}  // namespace swz
