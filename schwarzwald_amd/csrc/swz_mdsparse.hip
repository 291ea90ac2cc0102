// swz_mdsparse.hip -- MIN_DISTANCE for SPARSE levels: one thread per point.
//
// Since round 6 the levels that can be decided on keys go to swz_mdblock.hip first (blocks of 8^3 cells staged in LDS,
// decisions in the same launch: 107 -> 86 ms for levels 2 + 3 of the 1 B run, a tenth of the memory traffic); this file
// is what runs when that path declines -- positions instead of keys, a block that does not fit its LDS capacity,
// SWZ_SP_BLOCK=0 -- and the entry point that picks between the two.
//
// Same result as the frontier sweep of swz_mindist.hip (the lexicographically-first maximal
// independent set in Morton order, PoissonDiskSampling / SparseGrid::add, core/tiling/Sampling.h:421-471,
// core/datastructures/SparseGrid.cpp:116-146), computed the other way round.  When a spacing-sized cell
// holds about one point, nearly every point is accepted and a wavefront per cell idles 63 lanes, so:
//   1. every point looks up its 27 adjacent cells once (dense [node][cell] table of run starts) and
//      records its EARLIER neighbours closer than the spacing (exact compare, usually 0-3 of them).
//      The search is a chain of dependent loads (table -> points), so it is laid out for latency: the
//      table holds {first, end} of every cell's run, the active points are gathered into 16-byte
//      records, and the 27 cells are taken nine at a time with all nine table lookups, then the j-th
//      record of all nine runs, in flight together.  The kernel is bound by the L1's outstanding misses
//      (rocprofv3: TCP_PENDING_STALL 60 % of the cycles, 2.5 L1 misses per point, HBM at 0.6 TB/s), and most of the
//      lines it touches are records: a record holds the position as three floats relative to the node's corner, which
//      decides all but ~1 in 1000 compares (|d2 - s2| beyond the float error bound, sp_filter_eps); the rest are
//      repeated in double on the original positions, so the result is the exact one;
//      (Round 2 also built the search per BLOCK of 8 x 8 x 8 cells out of LDS -- table entries and records of the
//      block and its one-cell halo copied in once: at these densities, 0.4-0.8 points per cell, the 1000-cell region
//      costs more than the ~400 points it serves: level 2 of the 1 B run 219 ms against 162 ms.  Dropped.)
//   2. fixpoint rounds over the still undecided points: rejected when a recorded neighbour is accepted,
//      accepted when all of them are rejected, otherwise wait.  States only move undecided -> final, so a
//      stale read is conservative; the number of rounds is the dependency depth (tens).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "swz_level.h"

namespace swz {

constexpr uint32_t SP_NONE = 0xFFFFFFFFu;
#ifndef SWZ_SP_BATCH
#define SWZ_SP_BATCH 4
#endif
// adjacent cells searched together (their table entries, then their j-th records, in flight at once).  Four: the kernel
// then needs 62 registers and runs 8 wavefronts per SIMD; nine (all a point usually has) need 91 -> 5 per SIMD and are
// 25 % slower at 1 B points (level 2: 110 against 88 ms), 5-6 cells 72 registers and 92 ms.
constexpr int SP_BATCH = SWZ_SP_BATCH;
#ifndef SWZ_SP_TABLE4
#define SWZ_SP_TABLE4 1
#endif
#ifndef SWZ_SP_DEFER
#define SWZ_SP_DEFER 1
#endif
constexpr int SP_K = 8;  // recorded neighbours per point (32 B); more -> the point re-searches every round
enum : uint8_t { SP_U = 0, SP_A = 1, SP_R = 2 };

struct SpArgs {
  const uint64_t* akey;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* snode_of;
  uint32_t all_sampled;      // every node of the level is sampled (the usual case): no look at nmode, snode_of is the identity
  const float4* rec;   // active order: position relative to the node's min corner, rounded to float; w unused --
                       // or, when the level is decided on keys (xyz != null), the point's integer key coordinates
  const uint32_t* aidx;      // exact positions of active point i: X[aidx ? aidx[i] : i]
  const double* X;
  const double* Y;
  const double* Z;
  const double* xyz;         // on keys: exact position of active point i = xyz[3 * ids[i]]
  const uint32_t* ids;
  float f_lo, f_hi;          // float squared distance < f_lo: closer than the spacing for sure, >= f_hi: farther for sure
  uint32_t m;
  uint32_t cell_shift;      // key >> cell_shift = node prefix + cell code
  uint32_t cell_levels;
  uint64_t cells_per_node;
  double sq_spacing;
  uint32_t sub_levels;      // key levels below the cell level that give a point's slab inside its cell
  float usq_f[3];           // squared slab width per axis, rounded down
  float cull_f;             // squared spacing with a margin: adjacent cells farther than this are skipped
  uint2* table;             // [sample node][cell code] -> {first, end} active index of the cell's run
  uint32_t* table4;         // on keys (SWZ_SP_TABLE4): [sample node][cell code] -> first only, 4 bytes per cell; the run's
                            // length travels in the w field of its FIRST record, which the search loads anyway
  uint32_t* nbr;            // [point][SP_K]
  uint8_t* ncount;          // recorded neighbours, SP_K + 1 = overflow
  uint8_t* state;
  uint8_t* taken;
};

// states are polled while other wavefronts publish them: agent-scope relaxed atomics (served by L2)
__device__ __forceinline__ uint8_t sp_load(const uint8_t* st, uint32_t i) {
  return __hip_atomic_load(st + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sp_store(uint8_t* st, uint32_t i, uint8_t v) {
  __hip_atomic_store(st + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one evaluation of point p from its recorded neighbours: 0 = still waiting, SP_A, SP_R
// (all states are requested before the first is looked at: one memory round trip per evaluation, not one per neighbour --
// as a plain loop the eight in-kernel evaluations cost 55 of level 2's 148 ms at 1 B points)
__device__ __forceinline__ uint8_t sp_eval(const uint8_t* st, const uint32_t* mine, uint32_t cnt) {
  uint32_t q[SP_K];
  uint8_t s[SP_K];
  const uint4 lo = *reinterpret_cast<const uint4*>(mine), hi = *reinterpret_cast<const uint4*>(mine + 4);
  q[0] = lo.x, q[1] = lo.y, q[2] = lo.z, q[3] = lo.w, q[4] = hi.x, q[5] = hi.y, q[6] = hi.z, q[7] = hi.w;
#pragma unroll
  for (int j = 0; j < SP_K; ++j) s[j] = (uint32_t)j < cnt ? sp_load(st, q[j]) : (uint8_t)SP_R;
  bool rej = false, wait = false;
#pragma unroll
  for (int j = 0; j < SP_K; ++j) {
    rej |= s[j] == SP_A;
    wait |= s[j] == SP_U;
  }
  return rej ? SP_R : (wait ? SP_U : SP_A);
}

__device__ __forceinline__ bool sp_sampled(const SpArgs& a, uint32_t i) { return a.all_sampled || a.nmode[a.nid[i]] == MODE_SAMPLE; }
__device__ __forceinline__ uint32_t sp_snode(const SpArgs& a, uint32_t i) {
  const uint32_t node = a.nid[i];
  return a.all_sampled ? node : a.snode_of[node];
}

__global__ __launch_bounds__(256) void sp_table_kernel(SpArgs a) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m || !sp_sampled(a, i)) return;
  const uint64_t pre = a.akey[i] >> a.cell_shift;
  const bool head = i == 0 || (a.akey[i - 1] >> a.cell_shift) != pre;
  const bool tail = i + 1 == a.m || (a.akey[i + 1] >> a.cell_shift) != pre;
  if (!head && !tail) return;
  const uint64_t code = pre & (a.cells_per_node - 1ull);
  uint2* e = a.table + ((uint64_t)sp_snode(a, i) * a.cells_per_node + code);
  if (head) e->x = i;
  if (tail) e->y = i + 1u;
}

__global__ __launch_bounds__(256) void sp_gather_kernel(const uint32_t* __restrict__ aidx, const uint64_t* __restrict__ akey,
                                                        uint32_t m, const double* __restrict__ X, const double* __restrict__ Y,
                                                        const double* __restrict__ Z, Box root, int node_depth,
                                                        float4* __restrict__ rec) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const uint32_t s = aidx ? aidx[i] : i;
  const Box nb = bounds_from_key(akey[i], root, node_depth);  // the point's node (level + 1 octants of its key)
  rec[i] = make_float4((float)(X[s] - nb.minx), (float)(Y[s] - nb.miny), (float)(Z[s] - nb.minz), 0.f);
}

// Key coordinates as records (swz_level.h, KeyMetric): the differences of integers below 2^21 are exact in float, and
// the band [f_lo, f_hi) around the spacing holds every pair the quantisation cannot decide.
// Both in one pass over the keys (round 5: the two kernels above read every key once each, and a memset wrote the state
// bytes: 15 ms per 1 B-point step for the two sparse levels): record, state byte, and the table entries of the run's ends.
__global__ __launch_bounds__(256) void sp_prepare_keys_kernel(SpArgs a, float4* __restrict__ rec) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m) return;
  const uint64_t k = a.akey[i];
  uint32_t x, y, z;
  key_coords_u32(k, x, y, z);
  a.state[i] = SP_U;
  const uint64_t pre = k >> a.cell_shift;
  const bool sampled = sp_sampled(a, i);
  const bool head = sampled && (i == 0 || (a.akey[i - 1] >> a.cell_shift) != pre);
  if (a.table4) {
    // the head of a run walks to its end (runs are short on the levels this path takes) and leaves the length in its record
    uint32_t len = 0;
    if (head) {
      len = 1;
      while (i + len < a.m && (a.akey[i + len] >> a.cell_shift) == pre) ++len;
      a.table4[(uint64_t)sp_snode(a, i) * a.cells_per_node + (pre & (a.cells_per_node - 1ull))] = i;
    }
    rec[i] = make_float4((float)x, (float)y, (float)z, __uint_as_float(len));
    return;
  }
  rec[i] = make_float4((float)x, (float)y, (float)z, 0.f);
  if (!sampled) return;
  const bool tail = i + 1 == a.m || (a.akey[i + 1] >> a.cell_shift) != pre;
  if (!head && !tail) return;
  const uint64_t code = pre & (a.cells_per_node - 1ull);
  uint2* e = a.table + ((uint64_t)sp_snode(a, i) * a.cells_per_node + code);
  if (head) e->x = i;
  if (tail) e->y = i + 1u;
}

// the exact compare of the reference on the original positions (GridCell.cpp:52)
__device__ __forceinline__ bool sp_exact_near(const SpArgs& a, uint32_t p, uint32_t q) {
  if (a.xyz) {
    const double* u = a.xyz + (size_t)a.ids[p] * 3;
    const double* v = a.xyz + (size_t)a.ids[q] * 3;
    return sq_dist(u[0], u[1], u[2], v[0], v[1], v[2]) < a.sq_spacing;
  }
  const uint32_t sp = a.aidx ? a.aidx[p] : p, sq = a.aidx ? a.aidx[q] : q;
  return sq_dist(a.X[sp], a.Y[sp], a.Z[sp], a.X[sq], a.Y[sq], a.Z[sq]) < a.sq_spacing;
}

// Visits every EARLIER point closer than the spacing to point p.  f(q) returns false to stop.
// DEFER: no exact compare inside the search loops at all -- the first TWO in-band pairs of a point are put aside and
// evaluated after the search, a third one (one point in ten million) raises *over and the caller treats the point like one
// with too many neighbours to record (the rounds search again, with DEFER = false).  The search kernel is bound by the
// latency of its dependent loads at the eight wavefronts per SIMD that 64 registers allow, and it is touchy about what
// sits in its loops: composing the position index there (one more dependent load in this rare path) cost level 2 of the
// 1 B run 9 ms (82 -> 91), measured in round 5.
template <bool DEFER, bool T4, typename F>
__device__ __forceinline__ void sp_visit_earlier(const SpArgs& a, uint32_t p, F f, bool* over = nullptr) {
  const float4 me = a.rec[p];
  const uint64_t mykey = a.akey[p];
  const uint64_t pre = mykey >> a.cell_shift;
  const uint32_t code = (uint32_t)(pre & (a.cells_per_node - 1ull));
  const uint64_t base = (uint64_t)sp_snode(a, p) * a.cells_per_node;
  // neighbour cell codes by arithmetic on the dilated coordinates (code = x bits | y bits | z bits, every
  // third bit): minus one = (v - 1) & mask, plus one = ((v | ~mask) + 1) & mask; out of the node when the
  // coordinate is already 0 / all ones
  const uint32_t all = (uint32_t)(a.cells_per_node - 1ull);
  const uint32_t mz = all & 0x09249249u, my = mz << 1, mx = mz << 2;
  const uint32_t vx = code & mx, vy = code & my, vz = code & mz;
  // index 0: minus one, 1: same, 2: plus one; SP_NONE marks "outside"
  const uint32_t dx[3] = {vx ? ((vx - 1u) & mx) : SP_NONE, vx, vx != mx ? (((vx | ~mx) + 1u) & mx) : SP_NONE};
  const uint32_t dy[3] = {vy ? ((vy - 1u) & my) : SP_NONE, vy, vy != my ? (((vy | ~my) + 1u) & my) : SP_NONE};
  const uint32_t dz[3] = {vz ? ((vz - 1u) & mz) : SP_NONE, vz, vz != mz ? (((vz | ~mz) + 1u) & mz) : SP_NONE};
  // Adjacent cells this point cannot reach are skipped (about 40 % of them): squared gap between the point's
  // slab inside its cell and the neighbour, per axis and direction, from the key bits below the cell level
  // (conservative: a slab never overstates the gap).
  uint32_t reach = 0;  // bit k: adjacent cell k (x fastest) can hold a point closer than the spacing
  {
    // (at most 4 sub levels: 12 bits -- the 32-bit form of the bit trick, a third of the instructions of the 64-bit one)
    const uint32_t sub = (uint32_t)(mykey >> (a.cell_shift - 3u * a.sub_levels)) & ((1u << (3u * a.sub_levels)) - 1u);
    const int smax = (1 << a.sub_levels) - 1;
    const int sx = (int)contract_bits_by_3_u32(sub >> 2), sy = (int)contract_bits_by_3_u32(sub >> 1), sz = (int)contract_bits_by_3_u32(sub);
    // (in float: the kernel is bound by its vector ALU work and doubles cost twice; usq_f is rounded down and cull_f
    // carries a 1e-5 margin over the 2^-18 one, far more than the three roundings of the sum can add)
    const float lx = (float)sx, hx = (float)(smax - sx), ly = (float)sy, hy = (float)(smax - sy), lz = (float)sz,
                hz = (float)(smax - sz);
    const float gx[3] = {lx * lx * a.usq_f[0], 0.f, hx * hx * a.usq_f[0]};
    const float gy[3] = {ly * ly * a.usq_f[1], 0.f, hy * hy * a.usq_f[1]};
    const float gz[3] = {lz * lz * a.usq_f[2], 0.f, hz * hz * a.usq_f[2]};
#pragma unroll
    for (int k = 0; k < 27; ++k)
      if (gx[k % 3] + gy[(k / 3) % 3] + gz[k / 9] < a.cull_f) reach |= 1u << k;
  }
  // the cells this point has to look at: inside the node, not later in Morton order, within reach
  uint32_t need = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const uint32_t X = dx[k % 3], Y = dy[(k / 3) % 3], Z = dz[k / 9];
    if (X != SP_NONE && Y != SP_NONE && Z != SP_NONE && (X | Y | Z) <= code && ((reach >> k) & 1u)) need |= 1u << k;
  }
  // Up to nine of them at a time (usually all: about eight remain): their table entries, then the j-th record
  // of every run, are requested together, so the search costs 1 + (longest run) memory round trips.
  // Pairs inside the band of the float compare need the exact positions: a chain of dependent scattered loads that the
  // whole wavefront waits for.  The first one of a point is put aside and evaluated after the search, all lanes together
  // (a second one is rare and evaluated on the spot).
  uint32_t pend0 = SP_NONE, pend1 = SP_NONE;
  const uint2* __restrict__ tab = a.table + base;
  const uint32_t* __restrict__ tab4 = a.table4 + base;
  while (need) {
    uint32_t q[SP_BATCH], qe[SP_BATCH];
#pragma unroll
    for (int i = 0; i < SP_BATCH; ++i) {
      q[i] = 0u;
      qe[i] = 0u;
      if (need) {
        const uint32_t k = (uint32_t)__ffs((int)need) - 1u;
        need &= need - 1u;
        const uint32_t kx = k % 3u, ky = (k / 3u) % 3u, kz = k / 9u;
        const uint32_t ncode = (kx == 0u ? dx[0] : (kx == 1u ? dx[1] : dx[2])) | (ky == 0u ? dy[0] : (ky == 1u ? dy[1] : dy[2])) |
                               (kz == 0u ? dz[0] : (kz == 1u ? dz[1] : dz[2]));
        if (T4) {
          const uint32_t e = tab4[ncode];
          if (e < p) {  // (empty cells are NONE; own cell: earlier points only)
            q[i] = e;
            qe[i] = e + 1u;  // (the run's end comes with its first record, below)
          }
        } else {
          const uint2 e = tab[ncode];
          if (e.x != SP_NONE) {  // empty cells are {NONE, NONE}
            q[i] = e.x;
            qe[i] = min(e.y, p);  // own cell: earlier points only
          }
        }
      }
    }
    // One step over the batch's runs with all their loads in flight (runs are short: half of them end here) ...
    for (int step = 0; step < 1; ++step) {
      float rx[SP_BATCH], ry[SP_BATCH], rz[SP_BATCH];
      bool any = false;
#pragma unroll
      for (int i = 0; i < SP_BATCH; ++i) {
        if (q[i] < qe[i]) {
          const float4 r = a.rec[q[i]];
          rx[i] = r.x;
          ry[i] = r.y;
          rz[i] = r.z;
          // (only a run's FIRST record carries the length, and only the first step reads a first record: with the four-byte
          // table this loop must stay at one step -- a later one would cut every run after two records)
          if (T4 && step == 0) qe[i] = min(q[i] + __float_as_uint(r.w), p);
          any = true;
        }
      }
      if (!any) break;
#pragma unroll
      for (int i = 0; i < SP_BATCH; ++i) {
        if (q[i] < qe[i]) {
          const float dx = me.x - rx[i], dy = me.y - ry[i], dz = me.z - rz[i];
          const float d2 = dx * dx + dy * dy + dz * dz;
          if (d2 < a.f_hi) {
            if (d2 < a.f_lo) {
              if (!f(q[i])) return;
            } else if (pend0 == SP_NONE) {
              pend0 = q[i];
            } else if (DEFER) {
              if (pend1 == SP_NONE) pend1 = q[i]; else *over = true;
            } else if (sp_exact_near(a, p, q[i])) {
              if (!f(q[i])) return;
            }
          }
          ++q[i];
        }
      }
    }
    // ... then what is left of the longer runs, one record per step: the kernel is bound by its vector ALU instructions
    // (rocprofv3: 1950 per wavefront, 80 % of the cycles), and a step of the loop above costs all four runs' worth of
    // them for every wavefront that has a single lane with a single long run (measured at 1 B points, level 2: the
    // all-runs loop until everything is read 88 ms, three steps of it before the walk 88, two 85, one 80, none 82)
    uint32_t cq = 0, ce = 0;
    for (;;) {
      if (cq >= ce) {
        bool found = false;
#pragma unroll
        for (int i = 0; i < SP_BATCH; ++i)
          if (!found && q[i] < qe[i]) {
            cq = q[i];
            ce = qe[i];
            q[i] = qe[i];
            found = true;
          }
        if (!found) break;
      }
      const float4 r = a.rec[cq];
      const float dx = me.x - r.x, dy = me.y - r.y, dz = me.z - r.z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < a.f_hi) {
        if (d2 < a.f_lo) {
          if (!f(cq)) return;
        } else if (pend0 == SP_NONE) {
          pend0 = cq;
        } else if (DEFER) {
          if (pend1 == SP_NONE) pend1 = cq; else *over = true;
        } else if (sp_exact_near(a, p, cq)) {
          if (!f(cq)) return;
        }
      }
      ++cq;
    }
  }
  if (pend0 != SP_NONE && sp_exact_near(a, p, pend0)) {
    if (!f(pend0)) return;
  }
  if (DEFER && pend1 != SP_NONE && sp_exact_near(a, p, pend1)) {
    if (!f(pend1)) return;
  }
}

// phase 1: record the earlier neighbours; points without any are accepted right away
constexpr int SP_NB_THREADS = 256;
template <bool T4>
__global__ __launch_bounds__(SP_NB_THREADS, 8) void sp_neighbours_kernel(SpArgs a, uint32_t* __restrict__ overflow, uint32_t xcd) {
  // workgroups go round-robin over the 8 XCDs: XCD x takes the x-th contiguous eighth of the points, so the
  // neighbourhoods a workgroup reads were mostly fetched into the same L2 by the workgroups just before it
  const uint32_t blk = xcd ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const uint32_t p = blk * SP_NB_THREADS + threadIdx.x;
  if (p >= a.m) return;
  uint32_t cnt = 0;
  if (sp_sampled(a, p)) {
    uint32_t* mine = a.nbr + (size_t)p * SP_K;
    bool over = false;
    sp_visit_earlier<SWZ_SP_DEFER != 0, T4>(a, p, [&](uint32_t q) {
      if (cnt < (uint32_t)SP_K) mine[cnt] = q;
      ++cnt;
      return cnt <= (uint32_t)SP_K;  // one past the capacity marks the overflow, then stop
    }, &over);
    if (over) cnt = (uint32_t)SP_K + 1u;  // (more in-band pairs than are put aside: the rounds search this point again, exactly)
    if (cnt > (uint32_t)SP_K) atomicAdd(overflow, 1u);
    if (cnt == 0) {
      sp_store(a.state, p, SP_A);
      a.taken[p] = 1;
    }
  }
  a.ncount[p] = (uint8_t)cnt;  // 0 also for the points of nodes that are not sampled: nothing left to decide
}

// phase 1b: the decisions, in one light pass in Morton order.  A point depends on earlier points only, and workgroups
// are dispatched in the order of their index, so most of what a point waits for was decided by a workgroup that started
// before its own: a few polls of the neighbours' states (agent-scope atomics) settle almost every point; what is
// still open after max_polls goes to the list of the rounds below (so nothing here can wait forever).
// (Until round 2 the search kernel polled eight times itself: the polls kept its wavefronts -- with all their
// registers -- resident for 50 of level 2's 148 ms at 1 B points.  A strictly ordered variant -- stretches of points
// handed out by an atomic ticket, polling until decided -- was no faster: 1.3 M tickets on one word and a chain of
// five dependent round trips per stretch.)
__global__ __launch_bounds__(256) void sp_resolve_kernel(SpArgs a, uint32_t max_polls, uint32_t* __restrict__ ulist,
                                                         uint32_t* __restrict__ ucount) {
  const uint64_t pp = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t p = (uint32_t)pp;
  const uint32_t wave_first = p - lane_id();
  const uint32_t cnt = pp < a.m ? a.ncount[p] : 0u;
  // this lane's state as the other lanes of the wavefront see it (a point without recorded neighbours was accepted by
  // the search kernel; points of nodes that are not sampled are nobody's neighbour)
  uint32_t mystate = cnt != 0u ? (uint32_t)SP_U : (uint32_t)SP_A;
  const bool recorded = cnt <= (uint32_t)SP_K;
  uint4 lo = make_uint4(0, 0, 0, 0), hi = lo;
  if (cnt != 0u && recorded) {
    const uint4* mine = reinterpret_cast<const uint4*>(a.nbr + (size_t)p * SP_K);
    lo = mine[0];
    if (cnt > 4u) hi = mine[1];
  }
  const uint32_t q[SP_K] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  // neighbours in this wavefront are read from the lanes' registers, never from memory; of the others only those
  // that were still undecided at the last look are polled again
  uint32_t inwave = 0, open = 0;  // bit j: neighbour j is a lane of this wavefront / is outside and not known final
  bool ext_acc = false;           // an outside neighbour is accepted
  if (recorded) {
#pragma unroll
    for (int j = 0; j < SP_K; ++j)
      if ((uint32_t)j < cnt) {
        if (q[j] >= wave_first) inwave |= 1u << j; else open |= 1u << j;   // neighbours are earlier points: q < p
      }
  }
  // (a point with more neighbours than can be recorded stays undecided here and goes to the rounds, which search
  // again: with that search inlined this kernel needs 150 registers instead of 40 and runs at a third of the occupancy)
  for (uint32_t polls = 0; polls < max_polls; ++polls) {
    if (!__ballot(mystate == (uint32_t)SP_U && recorded)) break;
    if (mystate == (uint32_t)SP_U && recorded && open) {
      uint8_t st[SP_K];
#pragma unroll
      for (int j = 0; j < SP_K; ++j) st[j] = ((open >> j) & 1u) ? sp_load(a.state, q[j]) : (uint8_t)SP_R;
#pragma unroll
      for (int j = 0; j < SP_K; ++j)
        if ((open >> j) & 1u) {
          ext_acc |= st[j] == SP_A;
          if (st[j] != SP_U) open &= ~(1u << j);
        }
    }
    // the wavefront's own chains, without memory: every lane shows its state, every undecided lane looks
    for (;;) {
      bool in_acc = false, in_wait = false;
#pragma unroll
      for (int j = 0; j < SP_K; ++j) {
        const uint32_t sj = (uint32_t)__shfl((int)mystate, (int)((q[j] - wave_first) & 63u), WAVE);
        if ((inwave >> j) & 1u) {
          in_acc |= sj == (uint32_t)SP_A;
          in_wait |= sj == (uint32_t)SP_U;
        }
      }
      bool changed = false;
      if (mystate == (uint32_t)SP_U && recorded) {
        const bool rej = ext_acc || in_acc;
        if (rej || (!in_wait && !open)) {
          mystate = rej ? SP_R : SP_A;
          sp_store(a.state, p, (uint8_t)mystate);
          if (!rej) a.taken[p] = 1;
          changed = true;
        }
      }
      if (!__ballot(changed)) break;
    }
  }
  // leftovers: one atomic per wavefront
  const bool undecided = mystate == (uint32_t)SP_U;
  const uint64_t bm = __ballot(undecided);
  if (bm) {
    const int leader = __ffsll((unsigned long long)bm) - 1;
    uint32_t off = 0;
    if ((int)lane_id() == leader) off = atomicAdd(ucount, (uint32_t)__popcll(bm));
    off = __shfl(off, leader, WAVE);
    if (undecided) ulist[off + (uint32_t)__popcll(bm & lanemask_lt())] = p;
  }
}

// phase 2: one fixpoint round over the undecided points
template <bool T4>
__global__ __launch_bounds__(256) void sp_round_kernel(SpArgs a, const uint32_t* __restrict__ uin, const uint32_t* __restrict__ nin,
                                                       uint32_t* __restrict__ uout, uint32_t* __restrict__ nout) {
  const uint32_t n = *nin;
  for (uint32_t i0 = blockIdx.x * 256 + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * 256) {  // wave-uniform
    const uint32_t i = i0 + lane_id();
    bool again = false;
    uint32_t p = 0;
    if (i < n) {
      p = uin[i];
      const uint32_t cnt = a.ncount[p];
      uint8_t r = SP_U;
      if (cnt <= (uint32_t)SP_K) {
        const uint32_t* mine = a.nbr + (size_t)p * SP_K;
        for (int it = 0; it < 4 && r == SP_U; ++it) r = sp_eval(a.state, mine, cnt);
      } else {  // too many neighbours to record: search again
        bool rej = false, wait = false;
        sp_visit_earlier<false, T4>(a, p, [&](uint32_t q) {
          const uint8_t s = sp_load(a.state, q);
          rej |= s == SP_A;
          wait |= s == SP_U;
          return !rej;
        });
        r = rej ? SP_R : (wait ? SP_U : SP_A);
      }
      if (r == SP_U) {
        again = true;
      } else {
        sp_store(a.state, p, r);
        if (r == SP_A) a.taken[p] = 1;
      }
    }
    const uint64_t bm = __ballot(again);
    if (bm) {
      const int leader = __ffsll((unsigned long long)bm) - 1;
      uint32_t off = 0;
      if ((int)lane_id() == leader) off = atomicAdd(nout, (uint32_t)__popcll(bm));
      off = __shfl(off, leader, WAVE);
      if (again) uout[off + (uint32_t)__popcll(bm & lanemask_lt())] = p;
    }
  }
}

__global__ void sp_zero_kernel(uint32_t* p) { *p = 0; }

// positions already in active order (X/Y/Z); snode_of already scanned.  Returns SWZ_OK and sets *used
// to false when the level does not qualify (the caller then runs the frontier sweep).
int min_distance_sparse_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp,
                              const LevelBuffers& lb, const uint32_t* snode_of, bool all_sampled, uint32_t num_nodes,
                              uint32_t sample_nodes, uint32_t sample_points, const uint32_t occupied[12], uint32_t* rounds_out,
                              bool* used) {
  *used = false;
  int cl = plan.cell_levels_geo;
  // the table must stay addressable and affordable: at most 2^31 entries
  while (cl > 0 && (double)sample_nodes * std::pow(8.0, cl) > 2147483648.0) --cl;
  // points per OCCUPIED cell: clustered data fills a small part of a node's volume
  const double per_cell = (double)sample_points / (double)std::max(1u, occupied[cl]);
  double limit = 2.0;  // per occupied cell; a uniform level with 1.5 points per cell of volume has 1.93
  if (const char* e = c->opt("SWZ_MD_SPARSE_LIMIT")) limit = atof(e);
  if (!(per_cell < limit)) return SWZ_OK;
  // the root of a sharded batch with ghosts in front, decided on keys: the sweep looks up two position arrays, this path one
  if (sp.ghosts && plan.level == -1 && !sp.X) return SWZ_OK;
  const uint32_t m = as.m;
  const KeyMetric km = key_metric(c, plan, sp);
  {
    // round 6: blocks of cells out of LDS, decisions in the same launch (swz_mdblock.hip); levels it cannot take --
    // no key metric, a block that does not fit its LDS capacity -- go on below as before
    bool done = false;
    SWZ_TRY(min_distance_block_level(c, plan, as, sp, lb, snode_of, all_sampled, num_nodes, sample_nodes, sample_points, occupied, km,
                                     &done));
    if (done) {
      if (rounds_out) *rounds_out += 1;
      *used = true;
      return SWZ_OK;
    }
  }

  SpArgs a{};
  a.akey = as.akey;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.snode_of = snode_of;
  a.all_sampled = all_sampled ? 1u : 0u;
  a.m = m;
  a.cell_levels = (uint32_t)cl;
  a.cells_per_node = 1ull << (3 * cl);
  a.cell_shift = (plan.node_shift == 63u ? 63u : plan.node_shift) - 3u * (uint32_t)cl;
  a.sq_spacing = plan.sq_spacing;
  {
    const int sub = std::max(0, std::min(4, 20 - (plan.level + cl)));
    a.sub_levels = (uint32_t)sub;
    const double ext[3] = {plan.root.maxx - plan.root.minx, plan.root.maxy - plan.root.miny, plan.root.maxz - plan.root.minz};
    for (int ax = 0; ax < 3; ++ax) {
      const double u = std::ldexp(ext[ax], -(plan.level + 1 + cl + sub));
      a.usq_f[ax] = std::nextafterf((float)(u * u), 0.f);
      if ((double)a.usq_f[ax] > u * u) a.usq_f[ax] = std::nextafterf(a.usq_f[ax], 0.f);
    }
    a.cull_f = std::nextafterf((float)(plan.sq_spacing * (1.0 + 0x1.0p-18) * (1.0 + 1e-5)), INFINITY);
  }
  a.taken = lb.taken;
  const uint64_t entries = (uint64_t)sample_nodes * a.cells_per_node;
  SWZ_TRY(c->get("sp_table", (size_t)entries, &a.table));
  float4* rec = nullptr;
  SWZ_TRY(c->get("md_pos", (size_t)m * 4, reinterpret_cast<double**>(&rec)));  // shared with the sweep (half of it used)
  a.rec = rec;
  a.aidx = as.aidx;
  a.X = sp.X;
  a.Y = sp.Y;
  a.Z = sp.Z;
  if (!km.ok && !sp.X) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE: this level needs the positions in Morton order and they were not gathered");
  if (km.ok) {
    // (A sharded batch: the ghosts in front of the sorted order have their own position array and ids below sp.ghosts,
    // local point p has id ghosts + p, see key_point_ids.  Ghosts are accepted again at the root and never reach a level
    // below it, and the root level of a batch with ghosts is left to the sweep -- above --, so only local ids occur here.)
    a.xyz = sp.xyz - (size_t)sp.ghosts * 3;
    SWZ_TRY(key_point_ids(c, as, sp, &a.ids));
    a.f_lo = km.f_lo;
    a.f_hi = km.f_hi;
    if (const char* e = c->opt("SWZ_SP_FILTER_EPS"))  // tests: 1e30 sends every compare within reach to the exact path
      if (atof(e) >= 0.5) {
        a.f_lo = 0.f;
        a.f_hi = INFINITY;
      }
  } else {
    // Float filter.  A record coordinate is fl(x - corner) with |x - corner| <= E (the node's extent; twice that is
    // assumed): error <= 2^-24 * 2E; the float difference of two of them adds 2^-24 * 2E: |dxf - dx| <= delta =
    // 2^-22 * E.  |sum dxf^2 - d^2| <= 2 sqrt(3) d delta + 3 delta^2, plus 3 roundings of the sum (2^-22 relative, generous).
    // Around d = s: relative to s^2 at most 2 sqrt(3) r + 3 r^2 + 2^-22 with r = delta / s.  Four times that is used.
    const double ext = std::max({plan.root.maxx - plan.root.minx, plan.root.maxy - plan.root.miny, plan.root.maxz - plan.root.minz});
    const double E = std::ldexp(ext, -(plan.level + 1));
    const double r = std::ldexp(E, -22) / std::sqrt(plan.sq_spacing);
    double eps = 4.0 * (2.0 * std::sqrt(3.0) * r * 1.001 + 3.0 * r * r + std::ldexp(1.0, -22));
    if (const char* e = c->opt("SWZ_SP_FILTER_EPS")) eps = atof(e);  // tests: 1e30 sends every compare to the exact path
    if (!(eps < 0.5)) {  // the filter decides nothing: everything that is not far beyond reach is compared exactly
      a.f_lo = 0.f;
      a.f_hi = INFINITY;
    } else {
      a.f_lo = std::nextafterf((float)(plan.sq_spacing * (1.0 - eps)), 0.f);
      a.f_hi = std::nextafterf((float)(plan.sq_spacing * (1.0 + eps)), INFINITY);
    }
  }
  static_assert(SP_K * sizeof(uint32_t) == 4 * sizeof(double), "md_acc holds 32 bytes per point");
  SWZ_TRY(c->get("md_acc", (size_t)m * 4, reinterpret_cast<double**>(&a.nbr)));  // shared with the sweep
  SWZ_TRY(c->get("sp_ncount", (size_t)m, &a.ncount));
  SWZ_TRY(c->get("sp_state", (size_t)m, &a.state));
  uint32_t *u0 = nullptr, *u1 = nullptr, *cnt = nullptr;
  SWZ_TRY(c->get("sp_ulist0", (size_t)m, &u0));
  SWZ_TRY(c->get("sp_ulist1", (size_t)m, &u1));
  SWZ_TRY(c->get("sp_counts", (size_t)4, &cnt));
  ProfScope ps(c, "sample_min_distance", (uint64_t)sample_points * 33ull, 1);
  const bool t4 = km.ok && SWZ_SP_TABLE4 != 0;
  if (t4) {
    a.table4 = reinterpret_cast<uint32_t*>(a.table);
    a.table = nullptr;
  }
  SWZ_HIP(c, memset_large(t4 ? (void*)a.table4 : (void*)a.table, 0xFF, (size_t)entries * (t4 ? sizeof(uint32_t) : sizeof(uint2)), c->stream));
  SWZ_HIP(c, hipMemsetAsync(cnt, 0, 16, c->stream));
  const uint32_t nb = div_up(m, 256);
  if (km.ok) {
    hipLaunchKernelGGL(sp_prepare_keys_kernel, dim3(nb), dim3(256), 0, c->stream, a, rec);
    SWZ_LAUNCH_CHECK(c);
  } else {
    SWZ_HIP(c, memset_large(a.state, SP_U, (size_t)m, c->stream));
    hipLaunchKernelGGL(sp_gather_kernel, dim3(nb), dim3(256), 0, c->stream, as.aidx, as.akey, m, sp.X, sp.Y, sp.Z, plan.root,
                       plan.level + 1, rec);
    SWZ_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(sp_table_kernel, dim3(nb), dim3(256), 0, c->stream, a);
    SWZ_LAUNCH_CHECK(c);
  }
  const uint32_t xcd = c->opt("SWZ_MD_XCD") ? ((uint32_t)atoi(c->opt("SWZ_MD_XCD")) >> 1) & 1u : 1u;
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  if (dbg) {
    ev0 = c->take_event();
    ev1 = c->take_event();
    ev2 = c->take_event();
    (void)hipEventRecord(ev0, c->stream);
  }
  const uint32_t nbn = div_up(m, SP_NB_THREADS);
  if (t4)
    hipLaunchKernelGGL(sp_neighbours_kernel<true>, dim3(xcd ? div_up(nbn, 8) * 8 : nbn), dim3(SP_NB_THREADS), 0, c->stream, a, cnt + 2, xcd);
  else
    hipLaunchKernelGGL(sp_neighbours_kernel<false>, dim3(xcd ? div_up(nbn, 8) * 8 : nbn), dim3(SP_NB_THREADS), 0, c->stream, a, cnt + 2, xcd);
  SWZ_LAUNCH_CHECK(c);
  {
    const uint32_t max_polls = c->opt("SWZ_SP_POLLS") ? (uint32_t)atoi(c->opt("SWZ_SP_POLLS")) : 16u;
    hipLaunchKernelGGL(sp_resolve_kernel, dim3(nb), dim3(256), 0, c->stream, a, max_polls, u0, cnt);
    SWZ_LAUNCH_CHECK(c);
  }
  if (dbg) (void)hipEventRecord(ev1, c->stream);
  uint32_t* uin = u0;
  uint32_t* uout = u1;
  uint32_t cur = 0;  // index of the counter of uin
  uint32_t rounds = 0;
  for (;;) {
    uint32_t h[3] = {0, 0, 0};
    SWZ_HIP(c, hipMemcpyAsync(h, cnt, 12, hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    const uint32_t left = h[cur];
    if (left == 0) break;
    // Locally dense data inside an on-average sparse level (clusters): many points with more neighbours than
    // can be recorded, or long dependency chains.  Both are what the frontier sweep is good at: give up here
    // (every decision taken so far is exact and will simply be taken again).
    if (h[2] > std::max<uint32_t>(1024u, sample_points / 1024u) || rounds >= 512) {
      if (c->opt("SWZ_DEBUG"))
        fprintf(stderr, "[swz] MIN_DISTANCE level %d sparse path abandoned: %u overflow points, %u rounds, %u undecided\n",
                plan.level, h[2], rounds, left);
      return SWZ_OK;
    }
    // a few rounds per host look; an empty list makes the remaining launches no-ops
    for (int r = 0; r < 4; ++r, ++rounds) {
      hipLaunchKernelGGL(sp_zero_kernel, dim3(1), dim3(1), 0, c->stream, cnt + (cur ^ 1));
      if (t4)
        hipLaunchKernelGGL(sp_round_kernel<true>, dim3(std::min<uint32_t>(4096u, std::max(1u, div_up(left, 256)))), dim3(256), 0,
                           c->stream, a, uin, cnt + cur, uout, cnt + (cur ^ 1));
      else
        hipLaunchKernelGGL(sp_round_kernel<false>, dim3(std::min<uint32_t>(4096u, std::max(1u, div_up(left, 256)))), dim3(256), 0,
                           c->stream, a, uin, cnt + cur, uout, cnt + (cur ^ 1));
      SWZ_LAUNCH_CHECK(c);
      std::swap(uin, uout);
      cur ^= 1;
    }
  }
  if (rounds_out) *rounds_out += rounds;
  if (dbg) {
    float t1 = 0.f, t2 = 0.f;
    (void)hipEventRecord(ev2, c->stream);
    (void)hipEventSynchronize(ev2);
    (void)hipEventElapsedTime(&t1, ev0, ev1);
    (void)hipEventElapsedTime(&t2, ev1, ev2);
    c->event_pool.insert(c->event_pool.end(), {ev0, ev1, ev2});
    fprintf(stderr, "[swz] MIN_DISTANCE level %d sparse path: %u pts, cell_levels %d (%.2f pts/cell), %u rounds, %.1f + %.1f ms\n",
            plan.level, sample_points, cl, per_cell, rounds, t1, t2);
  }
  *used = true;
  return SWZ_OK;
}

}  // namespace swz
