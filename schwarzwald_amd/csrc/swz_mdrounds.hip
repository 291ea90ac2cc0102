// swz_mdrounds.hip -- MIN_DISTANCE in "property" mode (SWZ_FLAG_MIN_DISTANCE_PROPERTY) on KEY COORDINATES, without any
// dependency chain: a maximal independent set grown in a handful of data-parallel rounds.
//
// The reference's PoissonDiskSampling (core/tiling/Sampling.h:421-471 + SparseGrid.cpp:116-146) is the greedy set in
// Morton order; what the sampler is FOR -- and what its author checks, test/TestTiler.cpp:361-421 -- is the property:
// inside a sampled node no two taken points closer than the node's spacing, and every point left out closer than the
// spacing to a taken one, with the reference's compare (squared double distance < float-squared spacing,
// GridCell.cpp:43-58).  Any maximal independent set of the "closer than the spacing" graph has it.  This file builds
// one like this (Luby's scheme on a grid; deterministic):
//
//   * every sampled node is cut into octree cells at least one spacing wide: only points of the same or of adjacent
//     cells can conflict.  All points start alive.
//   * a round: (1) every cell's first alive point is its CANDIDATE; (2) a candidate WINS when no candidate of an adjacent
//     cell is closer than the spacing and has the better priority (a hash of the cell number: no spatial order, hence no
//     chains) -- two conflicting candidates never both win, and the best candidate anywhere always does; (3) every alive
//     point that is closer than the spacing to a winner of its own or an adjacent cell dies, the winners are taken.
//     Candidates are alive, i.e. at least the spacing away from everything taken before, so the taken set stays
//     independent; rounds go on until nothing is alive, so it ends maximal.
//   * the compares are taken on the key coordinates (the Morton key IS the position, quantised to 2^-21 of the cubic
//     bounds; swz_mdkeys.hip: key_metric) and repeated with the reference's arithmetic on the original positions for the
//     pairs inside the quantisation band.  Nothing is gathered or sorted; a round is three streaming kernels (one thread
//     per cell twice, one per alive point), and the number of alive points falls by about half per round.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

constexpr uint32_t PR_NONE = 0xFFFFFFFFu;
constexpr uint32_t PR_WL_ENTRIES = 28;  // winners around a cell written side by side for the first kill passes (pr_kill_wlist_kernel): 27 + one of padding
enum : uint8_t { PR_ALIVE = 0, PR_DEAD = 1 };
enum { PRC_ALIVE = CTR_DBG_HIST, PRC_LIST = CTR_DBG_HIST + 1, PRC_BAND = CTR_DBG_HIST + 2, PRC_WON = CTR_DBG_HIST + 3 };

struct PrArgs {
  const uint64_t* akey;
  uint32_t m;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* snode_of;
  uint32_t all_sampled;
  // exact position of active point i: xyz[3 * perm[aidx ? aidx[i] : i]] -- looked up for the pairs inside the band only (a
  // few in ten thousand), so the composed index is not materialised (2.7 ms per level and 1 B points, round 5)
  const uint32_t* aidx;
  const uint32_t* perm;
  const double* xyz;
  uint8_t* taken;
  uint8_t* state;        // [m] PR_*
  uint32_t* counters;
  uint32_t* cand[2];     // [cell] first alive point of the cell (this round's candidates / the next round's)
  uint64_t* candq;       // [cell] the candidate's key coordinates x | y << 21 | z << 42, bit 63 set (0: none)
  uint64_t* wonq;        // [cell] this round's winner (same packing), 0: none
  uint32_t* woni;        // [cell] ... its active index
  uint32_t* wmask;       // [cell] bit k: the adjacent cell in direction k holds a winner of this round
  uint32_t* list[2];     // alive points (compacted once few are left)
  // blocks of cells for pr_kill_block_kernel (null: not used): first point of every block's run of the sorted keys and the index
  // behind its last one, written by pr_init_kernel where the key above the block's bits changes
  uint32_t* bstart;
  uint32_t* bend;
  uint32_t block_bits;   // 3 x log2 of the block's edge in cells
  uint64_t ncells;       // sampled nodes x cells per node
  uint32_t cell_shift;
  uint64_t cells_per_node;
  float f_lo, f_hi;
  double sq_spacing;
};

// Key coordinates of a key with 32-bit arithmetic (the 64-bit bit trick costs twice as much and the kill pass of the first
// round runs it for every point of the level): bit 3j+2 of the key is bit j of x, 3j+1 of y, 3j of z (MortonIndex.h:62-79).
// The low word holds x bits 0-9, y bits 0-10, z bits 0-10; the high word (bit 32 on) the rest.
__device__ __forceinline__ void pr_coords_u(uint64_t key, uint32_t& x, uint32_t& y, uint32_t& z) {
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  x = contract_bits_by_3_u32(lo >> 2) | (contract_bits_by_3_u32(hi) << 10);
  y = contract_bits_by_3_u32(lo >> 1) | (contract_bits_by_3_u32(hi >> 2) << 11);
  z = contract_bits_by_3_u32(lo) | (contract_bits_by_3_u32(hi >> 1) << 11);
}
__device__ __forceinline__ uint64_t pr_pack(uint64_t key) {
  uint32_t x, y, z;
  pr_coords_u(key, x, y, z);
  return (uint64_t)x | ((uint64_t)y << 21) | ((uint64_t)z << 42);
}
__device__ __forceinline__ void pr_unpack(uint64_t v, float& x, float& y, float& z) {
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  x = (float)(lo & 0x1FFFFFu);
  y = (float)(((lo >> 21) | (hi << 11)) & 0x1FFFFFu);
  z = (float)((hi >> 10) & 0x1FFFFFu);
}
__device__ __forceinline__ float pr_d2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
// the reference's compare on the exact positions (GridCell.cpp:52) for active points i and j
__device__ __forceinline__ bool pr_exact_near(const PrArgs& a, uint32_t i, uint32_t j) {
  const double* p = a.xyz + (size_t)a.perm[a.aidx ? a.aidx[i] : i] * 3;
  const double* q = a.xyz + (size_t)a.perm[a.aidx ? a.aidx[j] : j] * 3;
  return sq_dist(p[0], p[1], p[2], q[0], q[1], q[2]) < a.sq_spacing;
}
__device__ __forceinline__ bool pr_sampled(const PrArgs& a, uint32_t i) { return a.all_sampled || a.nmode[a.nid[i]] == MODE_SAMPLE; }
__device__ __forceinline__ uint64_t pr_cell_of(const PrArgs& a, uint32_t i) {
  const uint32_t sn = a.all_sampled ? a.nid[i] : a.snode_of[a.nid[i]];
  return (uint64_t)sn * a.cells_per_node + ((a.akey[i] >> a.cell_shift) & (a.cells_per_node - 1ull));
}
// priority of a cell's candidate: smaller wins.  An odd multiplier permutes the 32-bit numbers, so no two cells tie.
__device__ __forceinline__ uint32_t pr_prio(uint64_t cell) { return (uint32_t)cell * 0x9E3779B1u; }

// The 27 cells around code (inside the node) by arithmetic on the dilated coordinates: dx/dy/dz[0..2] = minus one, same,
// plus one per axis; NONE when outside the node.
struct PrNbr {
  uint32_t dx[3], dy[3], dz[3];
};
__device__ __forceinline__ PrNbr pr_nbr(const PrArgs& a, uint32_t code) {
  const uint32_t all = (uint32_t)(a.cells_per_node - 1ull);
  const uint32_t mz = all & 0x09249249u, my = mz << 1, mx = mz << 2;
  const uint32_t vx = code & mx, vy = code & my, vz = code & mz;
  PrNbr n;
  n.dx[0] = vx ? ((vx - 1u) & mx) : PR_NONE; n.dx[1] = vx; n.dx[2] = vx != mx ? (((vx | ~mx) + 1u) & mx) : PR_NONE;
  n.dy[0] = vy ? ((vy - 1u) & my) : PR_NONE; n.dy[1] = vy; n.dy[2] = vy != my ? (((vy | ~my) + 1u) & my) : PR_NONE;
  n.dz[0] = vz ? ((vz - 1u) & mz) : PR_NONE; n.dz[1] = vz; n.dz[2] = vz != mz ? (((vz | ~mz) + 1u) & mz) : PR_NONE;
  return n;
}

// all points alive; the first point of every cell is its first candidate (cells = runs of the sorted keys)
__global__ __launch_bounds__(256) void pr_init_kernel(PrArgs a) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m) return;
  if (!pr_sampled(a, i)) {
    a.state[i] = PR_DEAD;
    return;
  }
  a.state[i] = PR_ALIVE;
  // (the key above the cell's bits names the node as well: a point of another node differs there, no node ids needed)
  const bool head = i == 0 || ((a.akey[i] >> a.cell_shift) != (a.akey[i - 1] >> a.cell_shift));
  uint64_t cell = 0;
  if (head) {
    cell = pr_cell_of(a, i);
    a.cand[0][cell] = i;
  }
  if (a.bstart) {
    const uint32_t bsh = a.cell_shift + a.block_bits;
    if (head && (i == 0 || (a.akey[i] >> bsh) != (a.akey[i - 1] >> bsh))) a.bstart[cell >> a.block_bits] = i;
    // the block's run ends where the key above the block's bits changes next (whoever follows: a point of a node that is not
    // sampled is dead from the start and only skipped)
    if (i + 1u == a.m || (a.akey[i + 1u] >> bsh) != (a.akey[i] >> bsh)) a.bend[pr_cell_of(a, i) >> a.block_bits] = i + 1u;
  }
}

// (1) the candidates' coordinates beside their index; the next round's candidate slots emptied
__global__ __launch_bounds__(256) void pr_candidates_kernel(PrArgs a, uint32_t cur) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= a.ncells) return;
  const uint32_t i = a.cand[cur][c];
  a.candq[c] = i != PR_NONE ? (pr_pack(a.akey[i]) | (1ull << 63)) : 0ull;
  a.cand[cur ^ 1u][c] = PR_NONE;
}

// (2) a candidate wins unless a conflicting candidate of an adjacent cell has the better priority
__device__ __forceinline__ uint64_t pr_winner_of(const PrArgs& a, uint32_t cur, uint64_t c);
__global__ __launch_bounds__(256) void pr_winners_kernel(PrArgs a, uint32_t cur) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= a.ncells) return;
  a.wonq[c] = pr_winner_of(a, cur, c);
}

// (2b) per cell that still has alive points: which of the 27 cells around it hold a winner of this round -- 27 lookups per
// cell instead of 27 per alive point in (3)
template <bool LIST>
__global__ __launch_bounds__(256) void pr_mask_kernel(PrArgs a, float4* __restrict__ wlist, uint8_t* __restrict__ wcount) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= a.ncells) return;
  uint32_t mask = 0, cnt = 0;
  if (a.candq[c]) {  // (a cell without a candidate has no alive point: nobody reads its mask)
    const uint64_t base = c & ~(a.cells_per_node - 1ull);
    const PrNbr n = pr_nbr(a, (uint32_t)(c & (a.cells_per_node - 1ull)));
    float4* e = LIST ? wlist + c * PR_WL_ENTRIES : nullptr;
    // the cell itself first, then the cells across a face, an edge, a corner: the closer winners kill most of the points
    constexpr int ORDER[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26};
#pragma unroll
    for (int kk = 0; kk < 27; ++kk) {
      const int k = LIST ? ORDER[kk] : kk;
      const uint32_t X = n.dx[k % 3], Y = n.dy[(k / 3) % 3], Z = n.dz[k / 9];
      if (X == PR_NONE || Y == PR_NONE || Z == PR_NONE) continue;
      const uint64_t nc = base + (X | Y | Z);
      const uint64_t q = a.wonq[nc];
      if (!q) continue;
      mask |= 1u << k;
      if (LIST) {
        float qx, qy, qz;
        pr_unpack(q, qx, qy, qz);
        e[cnt] = make_float4(qx, qy, qz, __uint_as_float(a.woni[nc]));
        ++cnt;
      }
    }
  }
  a.wmask[c] = mask;
  if (LIST) wcount[c] = (uint8_t)cnt;
}

// (3) alive points closer than the spacing to a winner around them die; the first survivor of every cell is the next
// candidate.  use_list: the alive points come from a list (and the survivors go to the next one).
__global__ __launch_bounds__(256) void pr_kill_kernel(PrArgs a, uint32_t cur, uint32_t use_list, uint32_t nlist) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t limit = use_list ? nlist : a.m;
  bool alive = false;
  uint32_t i = 0;
  if (t < limit) {
    i = use_list ? a.list[cur][t] : t;
    alive = a.state[i] == PR_ALIVE;
  }
  uint64_t cell = 0;
  if (alive) {
    cell = pr_cell_of(a, i);
    const uint64_t base = cell & ~(a.cells_per_node - 1ull);
    const PrNbr n = pr_nbr(a, (uint32_t)(cell & (a.cells_per_node - 1ull)));
    float x, y, z;
    {
      uint32_t ux, uy, uz;
      pr_coords_u(a.akey[i], ux, uy, uz);
      x = (float)ux;
      y = (float)uy;
      z = (float)uz;
    }
    bool dead = false;
    uint32_t nband = 0;
    // (Measured and dropped: the winners around a cell copied side by side into a per-cell list of 8 or 16 entries, four
    // points per thread with the cell's list unpacked once per run, branch-free tests of all entries: every variant cost
    // more than this loop -- the pass is bound by its ~300 vector instructions per point, not by the lookups.)
    uint32_t mask = a.wmask[cell];
    while (mask && !dead) {
      const uint32_t k = (uint32_t)__ffs((int)mask) - 1u;
      mask &= mask - 1u;
      const uint32_t kx = k % 3u, ky = (k / 3u) % 3u, kz = k / 9u;
      const uint32_t X = kx == 0u ? n.dx[0] : (kx == 1u ? n.dx[1] : n.dx[2]);
      const uint32_t Y = ky == 0u ? n.dy[0] : (ky == 1u ? n.dy[1] : n.dy[2]);
      const uint32_t Z = kz == 0u ? n.dz[0] : (kz == 1u ? n.dz[1] : n.dz[2]);
      const uint64_t nc = base + (X | Y | Z);
      const uint64_t q = a.wonq[nc];
      float qx, qy, qz;
      pr_unpack(q, qx, qy, qz);
      const float d2 = pr_d2(x, y, z, qx, qy, qz);
      if (d2 < a.f_lo) dead = true;
      else if (d2 < a.f_hi) {
        ++nband;
        dead = pr_exact_near(a, i, a.woni[nc]);
      }
    }
    if (nband) atomicAdd(&a.counters[PRC_BAND], nband);
    if (dead) {
      a.state[i] = PR_DEAD;
      alive = false;
    }
  }
  // survivors: the next round's candidate of their cell is the smallest index (one atomic per run of a cell in the
  // wavefront: a cell's points are consecutive -- the lists keep the order of the points)
  const uint64_t am = __ballot(alive);
  if (!am) return;
  {
    // (every lane takes part in the shuffles; the lane they read is an alive one)
    const uint64_t before = am & lanemask_lt();
    const int prev = before ? 63 - __clzll((unsigned long long)before) : 0;
    const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)cell, prev, WAVE), phi = (uint32_t)__shfl((int)(uint32_t)(cell >> 32), prev, WAVE);
    const bool head = alive && (before == 0ull || (((uint64_t)phi << 32) | plo) != cell);
    if (head) atomicMin(&a.cand[cur ^ 1u][cell], i);
  }
}

// (2c) + (3'): the first rounds of a level, while every point is still looked at (no list of alive points yet).  The kill
// pass above is bound by the instructions it issues (rocprofv3 and the ISA agree: ~830 per wavefront -- 160 to find the
// adjacent cells of the point's cell, then ~60 per winner around it for as many winners as the wavefront's unluckiest lane
// has to try -- 21 ms per 1 B points whatever the cell size), and nearly all of that is per-CELL work done per point.  So
// the cell does it: next to the mask, pr_mask_kernel<true> writes the winners around the cell side by side as ready-made
// {x, y, z, index} entries (own cell first), and a point loads them four at a time and spends ten instructions per winner.
// (Measured with SQ_INSTS_VALU per dispatch: 913 vector instructions per wavefront in the first round at the root for the
// loop over the mask.)  448 bytes per cell: levels of dozens of points per cell only.
constexpr uint32_t PR_WL = PR_WL_ENTRIES;  // (every cell around may hold a winner -- at the root nearly all do in the first round: the first points of
                                           // adjacent cells, their candidates, sit a cell apart -- so a record has room for all 27)

__global__ __launch_bounds__(256) void pr_kill_wlist_kernel(PrArgs a, uint32_t cur, const float4* __restrict__ wlist, const uint8_t* __restrict__ wcount) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  bool alive = i < a.m && a.state[i] == PR_ALIVE;
  uint64_t cell = 0;
  if (alive) {
    cell = pr_cell_of(a, i);
    float x, y, z;
    {
      uint32_t ux, uy, uz;
      pr_coords_u(a.akey[i], ux, uy, uz);
      x = (float)ux;
      y = (float)uy;
      z = (float)uz;
    }
    const uint32_t cnt = wcount[cell];
    bool dead = false;
    uint32_t nband = 0;
    {
      const float4* __restrict__ e = wlist + cell * PR_WL;
      for (uint32_t j0 = 0; j0 < PR_WL; j0 += 4u) {
        if (j0 >= cnt || dead) break;
        const float4 e0 = e[j0], e1 = e[j0 + 1u], e2 = e[j0 + 2u], e3 = e[j0 + 3u];  // (allocated for every cell; entries >= cnt are not looked at)
        const float4 en[4] = {e0, e1, e2, e3};
#pragma unroll
        for (uint32_t t = 0; t < 4u; ++t) {
          if (j0 + t < cnt && !dead) {
            const float d2 = pr_d2(x, y, z, en[t].x, en[t].y, en[t].z);
            if (d2 < a.f_lo) dead = true;
            else if (d2 < a.f_hi) {
              ++nband;
              dead = pr_exact_near(a, i, __float_as_uint(en[t].w));
            }
          }
        }
      }
    }
    if (nband) atomicAdd(&a.counters[PRC_BAND], nband);
    if (dead) {
      a.state[i] = PR_DEAD;
      alive = false;
    }
  }
  const uint64_t am = __ballot(alive);
  if (!am) return;
  {
    const uint64_t before = am & lanemask_lt();
    const int prev = before ? 63 - __clzll((unsigned long long)before) : 0;
    const uint32_t plo = (uint32_t)__shfl((int)(uint32_t)cell, prev, WAVE), phi = (uint32_t)__shfl((int)(uint32_t)(cell >> 32), prev, WAVE);
    const bool head = alive && (before == 0ull || (((uint64_t)phi << 32) | plo) != cell);
    if (head) atomicMin(&a.cand[cur ^ 1u][cell], i);
  }
}

// (3''): the kill pass of the rounds that still look at every point, by BLOCKS of cells (round 6).  Both passes above spend their
// instructions on finding the cells around a point's cell in a Morton-coded table (160 of ~830 per wavefront) and on fetching and
// unpacking one winner at a time from memory (~60 each), or need 448 bytes of records per cell.  Here a workgroup takes a block
// of B x B x B cells (B = 8, 4 or 2: at least some thousand blocks per level) -- one run of the sorted keys --, stages the winners of the
// (B + 2)^3 cells of and around it in LDS as ready-made {x, y, z, index} entries (a cell without a winner: coordinates that are far
// from everything), and a point finds the 27 entries around its cell by adding constants to one LDS index: a dozen instructions
// per winner, no table of masks, no records.  Same winners, same kills: the result is the one of the passes above.
struct PrBlockArgs {
  const uint32_t* eoff;    // [blocks + 1] chunks beyond the first of the blocks before this one ([blocks]: of all)
  uint32_t blocks;
  uint32_t bl;             // log2 of B
  uint32_t cl;             // cell levels of the grid
};
__device__ __forceinline__ uint32_t pr_dilate3(uint32_t v) {  // bit j -> bit 3j (v < 1024)
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
// A block of more than PR_CHUNK points is shared: its first PR_CHUNK points go to the workgroup of the block, every further
// chunk to a workgroup of its own behind them (surface-like data: a blob of 30 M points sits in a handful of blocks).
constexpr uint32_t PR_CHUNK = 32768;
__global__ __launch_bounds__(256) void pr_block_extra_kernel(const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ bend, uint32_t blocks,
                                                             uint32_t* __restrict__ extra) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b > blocks) return;
  extra[b] = (b < blocks && bstart[b] != PR_NONE) ? (bend[b] - bstart[b] - 1u) / PR_CHUNK : 0u;
}
// Two phases per point: the winners of its own cell and of the six cells across a face first -- most points of a dense level die
// there --, and only the survivors, collected in a queue in LDS until they fill the workgroup, go on to the other twenty (a
// wavefront executes every test one of its lanes still needs: with 64 consecutive points that meant all 27 for nearly every
// wavefront).
constexpr uint32_t PR_BQ = 512;       // queue entries (a workgroup appends at most 256 between two drains of 256)
#ifndef SWZ_PR_NEAR
#define SWZ_PR_NEAR 7
#endif
constexpr int PR_NEAR = SWZ_PR_NEAR;  // tests of the first phase
__global__ __launch_bounds__(256) void pr_kill_block_kernel(PrArgs a, PrBlockArgs g, uint32_t cur) {
  extern __shared__ float4 pr_lw[];  // (B + 2)^3 winners
  __shared__ uint32_t range[3];
  __shared__ uint32_t qn[1];          // entries appended so far (the queue is a ring; every thread counts the drained ones)
  __shared__ uint32_t qi[PR_BQ], qc[PR_BQ];
  __shared__ int qr[PR_BQ];
  __shared__ float qx[PR_BQ], qy[PR_BQ], qz[PR_BQ];
  const uint32_t tid = threadIdx.x;
  const uint32_t B = 1u << g.bl, R = B + 2u, R2 = R * R, R3 = R2 * R;
  if (tid == 0) {
    // workgroups [0, blocks): the first chunk of their block; behind them one per further chunk, found by search
    uint32_t b = blockIdx.x, k = 0;
    if (b >= g.blocks) {
      const uint32_t e = b - g.blocks;
      b = PR_NONE;
      if (e < g.eoff[g.blocks]) {
        uint32_t lo = 0, hi = g.blocks;  // the last block with eoff <= e (its own extra chunks follow it)
        while (hi - lo > 1u) {
          const uint32_t mid = lo + (hi - lo) / 2u;
          if (g.eoff[mid] <= e) lo = mid; else hi = mid;
        }
        b = lo;
        k = e - g.eoff[lo] + 1u;
      }
    }
    uint32_t lo = PR_NONE, hi = 0;
    if (b != PR_NONE && a.bstart[b] != PR_NONE) {
      lo = a.bstart[b] + k * PR_CHUNK;
      hi = min(a.bend[b], lo + PR_CHUNK);
    }
    range[0] = lo;
    range[1] = hi;
    range[2] = b;
    qn[0] = 0u;
  }
  __syncthreads();
  if (range[0] == PR_NONE) return;  // an empty block (most of them, on surface-like data) costs two loads
  const uint32_t b = range[2];
  // the block inside its node: cell coordinates of its origin (key bit 3j + 2 is bit j of x, 3j + 1 of y, 3j of z)
  const uint32_t per_node_bits = 3u * (g.cl - g.bl);
  const uint32_t bcode = b & ((1u << per_node_bits) - 1u);
  const uint64_t base = ((uint64_t)(b >> per_node_bits)) << (3u * g.cl);  // first cell of the node
  const uint32_t ox = contract_bits_by_3_u32(bcode >> 2) << g.bl, oy = contract_bits_by_3_u32(bcode >> 1) << g.bl, oz = contract_bits_by_3_u32(bcode) << g.bl;
  const uint32_t cmask = (1u << g.cl) - 1u;
  for (uint32_t r = tid; r < R3; r += 256u) {
    const uint32_t ix = r % R, iy = (r / R) % R, iz = r / R2;
    const uint32_t cx = ox + ix - 1u, cy = oy + iy - 1u, cz = oz + iz - 1u;  // (wraps below zero: > cmask)
    float4 e = make_float4(1e30f, 1e30f, 1e30f, __uint_as_float(PR_NONE));
    if (cx <= cmask && cy <= cmask && cz <= cmask) {
      const uint64_t c = base + ((pr_dilate3(cx) << 2) | (pr_dilate3(cy) << 1) | pr_dilate3(cz));
      const uint64_t q = a.wonq[c];
      if (q) {
        float qx_, qy_, qz_;
        pr_unpack(q, qx_, qy_, qz_);
        e = make_float4(qx_, qy_, qz_, __uint_as_float(a.woni[c]));
      }
    }
    pr_lw[r] = e;
  }
  __syncthreads();
  const uint32_t lo = range[0], hi = range[1];
  const uint32_t cb = a.cell_shift / 3u;
  const int iR = (int)R, iR2 = (int)R2;
  // the cell itself first, then the cells across a face, an edge, a corner: the closer winners kill most of the points
  constexpr int ORDER[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26};
  uint32_t nband = 0;
  // second phase for the queue's entries [from, from + 256) (fewer at the very end)
  auto drain = [&](uint32_t from, uint32_t count) {
    if (tid < count) {
      const uint32_t s = (from + tid) % PR_BQ;
      const uint32_t i = qi[s];
      const int r0 = qr[s];
      const float x = qx[s], y = qy[s], z = qz[s];
      bool dead = false;
#pragma unroll
      for (int kk = PR_NEAR; kk < 27; ++kk) {
        const int k = ORDER[kk];
        if (!dead) {
          const float4 e = pr_lw[r0 + (k % 3 - 1) + iR * ((k / 3) % 3 - 1) + iR2 * (k / 9 - 1)];
          const float d2 = pr_d2(x, y, z, e.x, e.y, e.z);
          if (d2 < a.f_lo) dead = true;
          else if (d2 < a.f_hi) {
            ++nband;
            dead = pr_exact_near(a, i, __float_as_uint(e.w));
          }
        }
      }
      if (dead) a.state[i] = PR_DEAD;
      else atomicMin(&a.cand[cur ^ 1u][base + qc[s]], i);  // a survivor: the smallest index of its cell is the next round's candidate
    }
  };
  uint32_t drained = 0;  // (every thread keeps the same count)
  for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
    const uint32_t i = i0 + tid;
    bool alive = i < hi && a.state[i] == PR_ALIVE;
    uint32_t code = 0;
    int r0 = 0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (alive) {
      const uint64_t key = a.akey[i];
      uint32_t ux, uy, uz;
      pr_coords_u(key, ux, uy, uz);
      x = (float)ux;
      y = (float)uy;
      z = (float)uz;
      code = (uint32_t)((key >> a.cell_shift) & (a.cells_per_node - 1ull));
      r0 = (int)((((ux >> cb) & cmask) - ox + 1u) + R * (((uy >> cb) & cmask) - oy + 1u) + R2 * (((uz >> cb) & cmask) - oz + 1u));
      bool dead = false;
#pragma unroll
      for (int kk = 0; kk < PR_NEAR; ++kk) {
        const int k = ORDER[kk];
        if (!dead) {
          const float4 e = pr_lw[r0 + (k % 3 - 1) + iR * ((k / 3) % 3 - 1) + iR2 * (k / 9 - 1)];
          const float d2 = pr_d2(x, y, z, e.x, e.y, e.z);
          if (d2 < a.f_lo) dead = true;
          else if (d2 < a.f_hi) {
            ++nband;
            dead = pr_exact_near(a, i, __float_as_uint(e.w));
          }
        }
      }
      if (dead) {
        a.state[i] = PR_DEAD;
        alive = false;
      }
    }
    // the survivors of the first phase queue up (one LDS atomic per wavefront)
    const uint64_t am = __ballot(alive);
    if (am) {
      uint32_t at = 0;
      const int leader = __ffsll((unsigned long long)am) - 1;
      if ((int)(tid & 63u) == leader) at = atomicAdd(&qn[0], (uint32_t)__popcll(am));
      at = (uint32_t)__shfl((int)at, leader, WAVE);
      if (alive) {
        const uint32_t s = (at + (uint32_t)__popcll(am & lanemask_lt())) % PR_BQ;
        qi[s] = i;
        qc[s] = code;
        qr[s] = r0;
        qx[s] = x;
        qy[s] = y;
        qz[s] = z;
      }
    }
    __syncthreads();
    if (qn[0] - drained >= 256u) {  // (uniform: everybody reads the same counter between two barriers)
      drain(drained, 256u);
      drained += 256u;
    }
    __syncthreads();
  }
  const uint32_t rest = qn[0] - drained;  // < 256
  if (rest) drain(drained, rest);
  if (nband) atomicAdd(&a.counters[PRC_BAND], nband);
}

// ---- the per-cell steps over a LIST of alive points.  Once the alive points are listed (in order: a cell's points are
// consecutive in the list) the cells that still matter are the cells of the list's entries -- a tenth of the grid after one
// round, a hundredth after two -- and the first entry of every cell (its head in the list) does the cell's work; the
// kernels over the whole cell grid cost 4.5 ms per round at level 1 of the 1 B run (134 M cells), seven rounds per level.
__device__ __forceinline__ bool pr_list_head(const PrArgs& a, const uint32_t* __restrict__ list, uint32_t t, uint32_t& i, uint64_t& cell) {
  i = list[t];
  cell = pr_cell_of(a, i);
  return t == 0u || pr_cell_of(a, list[t - 1u]) != cell;
}
// (1) over the list of the round BEFORE (the cells that had a candidate then): the new candidate's coordinates, or none --
// a cell whose last alive point has died must not keep its old candidate, the cells around it would lose against a dead
// point for ever --, the slot of the round after emptied, last round's winner forgotten (its kills are done)
__global__ __launch_bounds__(256) void pr_candidates_list_kernel(PrArgs a, uint32_t cur, const uint32_t* __restrict__ prev_list, uint32_t nprev) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nprev) return;
  uint32_t i;
  uint64_t c;
  if (!pr_list_head(a, prev_list, t, i, c)) return;
  const uint32_t ci = a.cand[cur][c];
  a.candq[c] = ci != PR_NONE ? (pr_pack(a.akey[ci]) | (1ull << 63)) : 0ull;
  a.cand[cur ^ 1u][c] = PR_NONE;
  a.wonq[c] = 0ull;
}
__device__ __forceinline__ uint64_t pr_winner_of(const PrArgs& a, uint32_t cur, uint64_t c) {
  const uint64_t mine = a.candq[c];
  if (!mine) return 0ull;
  const uint32_t i = a.cand[cur][c];
  float x, y, z;
  pr_unpack(mine, x, y, z);
  const uint64_t base = c & ~(a.cells_per_node - 1ull);
  const PrNbr n = pr_nbr(a, (uint32_t)(c & (a.cells_per_node - 1ull)));
  const uint32_t my_prio = pr_prio(c);
  bool lose = false;
  uint32_t nband = 0;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k == 13) continue;
    const uint32_t X = n.dx[k % 3], Y = n.dy[(k / 3) % 3], Z = n.dz[k / 9];
    if (X == PR_NONE || Y == PR_NONE || Z == PR_NONE || lose) continue;
    const uint64_t nc = base + (X | Y | Z);
    const uint64_t q = a.candq[nc];
    if (!q || pr_prio(nc) > my_prio) continue;
    float qx, qy, qz;
    pr_unpack(q, qx, qy, qz);
    const float d2 = pr_d2(x, y, z, qx, qy, qz);
    if (d2 < a.f_lo) lose = true;
    else if (d2 < a.f_hi) {
      ++nband;
      lose = pr_exact_near(a, i, a.cand[cur][nc]);
    }
  }
  if (nband) atomicAdd(&a.counters[PRC_BAND], nband);
  if (lose) return 0ull;
  a.woni[c] = i;
  a.taken[i] = 1;
  a.state[i] = PR_DEAD;
  return mine;
}
__global__ __launch_bounds__(256) void pr_winners_list_kernel(PrArgs a, uint32_t cur, const uint32_t* __restrict__ list, uint32_t nlist) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlist) return;
  uint32_t i;
  uint64_t c;
  if (!pr_list_head(a, list, t, i, c)) return;
  a.wonq[c] = pr_winner_of(a, cur, c);
}
__global__ __launch_bounds__(256) void pr_mask_list_kernel(PrArgs a, const uint32_t* __restrict__ list, uint32_t nlist) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nlist) return;
  uint32_t i;
  uint64_t c;
  if (!pr_list_head(a, list, t, i, c)) return;
  uint32_t mask = 0;
  const uint64_t base = c & ~(a.cells_per_node - 1ull);
  const PrNbr n = pr_nbr(a, (uint32_t)(c & (a.cells_per_node - 1ull)));
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const uint32_t X = n.dx[k % 3], Y = n.dy[(k / 3) % 3], Z = n.dz[k / 9];
    if (X == PR_NONE || Y == PR_NONE || Z == PR_NONE) continue;
    if (a.wonq[base + (X | Y | Z)]) mask |= 1u << k;
  }
  a.wmask[c] = mask;
}

// The alive points after a round, counted and -- once they are few -- listed in their order by a scan (no atomics: a
// single counter word under every wavefront of a billion-point level was what the first version of this file spent its
// time on, and appends in arrival order made the result depend on the scheduling).
struct PrAliveF {  // state byte == PR_ALIVE (0): the tile sums come from 16-byte loads (swz_scan.h, FsCountsZeroBytes)
  const uint8_t* taken;
  __device__ uint32_t operator()(uint32_t i) const { return taken[i] == PR_ALIVE ? 1u : 0u; }
};
template <>
struct FsCountsZeroBytes<PrAliveF> {
  static constexpr bool value = true;
};
struct PrListG {  // point i is alive: the next list's entry
  uint32_t* out;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t alive) const {
    if (alive) out[excl] = i;
  }
};
struct PrAliveOfListF {  // the same over a list of points
  const uint8_t* state;
  const uint32_t* list;
  __device__ uint32_t operator()(uint32_t t) const { return state[list[t]] == PR_ALIVE ? 1u : 0u; }
};
struct PrListOfListG {
  const uint32_t* list;
  uint32_t* out;
  __device__ void operator()(uint32_t t, uint32_t excl, uint32_t alive) const {
    if (alive) out[excl] = list[t];
  }
};

__global__ __launch_bounds__(256) void pr_snode_flag_kernel(const uint8_t* __restrict__ nmode, uint32_t nnodes, uint32_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < nnodes) out[j] = nmode[j] == MODE_SAMPLE ? 1u : 0u;
}

// One level.  *used = false when the level cannot be decided on keys or its cell grid would be too large.
int min_distance_rounds_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                              uint32_t nnodes, uint32_t sample_nodes, uint32_t sample_points, const uint32_t* snode_of, uint32_t* rounds_out,
                              bool* used) {
  *used = false;
  const KeyMetric km = key_metric(c, plan, sp);
  if (!km.ok || sp.ghosts) return SWZ_OK;
  if (const char* e = c->opt("SWZ_MD_ROUNDS"))
    if (atoi(e) == 0) return SWZ_OK;
  const uint32_t m = as.m;
  const uint32_t nsh = plan.node_shift == 63u ? 63u : plan.node_shift;
  // the finest cells the spacing allows give the most candidates per round; on sparse levels the dense cell grid would
  // dwarf the points, so cells are doubled while they hold fewer than four points on average
  int cl = plan.cell_levels_geo;
  if (cl < 0 || 3u * (uint32_t)cl > nsh || cl > 10) return SWZ_OK;
  double min_pop = 0.0;  // (measured: coarser cells mean one candidate per LARGER cell and round -- 57 rounds instead of 7)
  if (const char* e = c->opt("SWZ_MD_ROUNDS_MIN_POP")) min_pop = atof(e);
  while (cl > 1 && (double)sample_points / ((double)sample_nodes * std::pow(8.0, cl)) < min_pop) --cl;
  PrArgs a{};
  a.cells_per_node = 1ull << (3 * cl);
  a.ncells = (uint64_t)sample_nodes * a.cells_per_node;
  if (a.ncells > 2147483648ull) return SWZ_OK;
  a.akey = as.akey;
  a.m = m;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.snode_of = snode_of;
  a.all_sampled = sample_nodes == nnodes ? 1u : 0u;
  a.xyz = sp.xyz;
  a.taken = lb.taken;
  a.counters = lb.counters;
  a.cell_shift = nsh - 3u * (uint32_t)cl;
  a.f_lo = km.f_lo;
  a.f_hi = km.f_hi;
  a.sq_spacing = plan.sq_spacing;
  // a cell must be at least one spacing wide on the key grid, band included (cell_levels_geo guarantees it geometrically)
  if (std::ldexp(1.0, (int)(a.cell_shift / 3u)) < km.T + 4.0) return SWZ_OK;
  a.aidx = as.aidx;  // (no ghosts here: checked above)
  a.perm = sp.perm;

  ProfScope ps(c, "sample_min_distance_property", (uint64_t)sample_points * 33ull, 1);
  const auto wall0 = std::chrono::steady_clock::now();
  SWZ_TRY(c->get("md_pr_state", (size_t)m, &a.state));
  SWZ_TRY(c->get("md_pr_cand0", (size_t)a.ncells, &a.cand[0]));
  SWZ_TRY(c->get("md_pr_cand1", (size_t)a.ncells, &a.cand[1]));
  SWZ_TRY(c->get("md_pr_candq", (size_t)a.ncells, &a.candq));
  SWZ_TRY(c->get("md_pr_wonq", (size_t)a.ncells, &a.wonq));
  SWZ_TRY(c->get("md_pr_woni", (size_t)a.ncells, &a.woni));
  SWZ_TRY(c->get("md_pr_wmask", (size_t)a.ncells, &a.wmask));
  SWZ_HIP(c, memset_large(a.cand[0], 0xFF, (size_t)a.ncells * 4, c->stream));
  SWZ_HIP(c, hipMemsetAsync(lb.counters + CTR_DBG_HIST, 0, 8 * sizeof(uint32_t), c->stream));
  const uint32_t cblocks = div_up(a.ncells, 256);
  uint32_t cur = 0, rounds = 0, alive = sample_points, nlist = 0;
  bool use_list = false;
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  const bool lists = !(c->opt("SWZ_MD_ROUNDS_LIST") && atoi(c->opt("SWZ_MD_ROUNDS_LIST")) == 0);
  std::string trace;
  uint32_t* d_total = lb.counters + PRC_ALIVE;
  // the first kill passes (no list yet) read the winners around their cell from per-cell records (pr_kill_wlist_kernel) when
  // the records are affordable (448 bytes per cell: levels of dozens of points per cell)
  double wl_min_pop = 24.0;
  if (const char* e = c->opt("SWZ_MD_ROUNDS_WLIST_MIN_POP")) wl_min_pop = atof(e);
  bool wl = (double)sample_points / (double)a.ncells >= wl_min_pop;
  // ... or, round 6, by blocks of cells with the winners around them in LDS (pr_kill_block_kernel): no records, no masks
  PrBlockArgs g{};
  // (on every level: also on grids that are mostly empty -- the thin levels of surface-like data, 0.1-0.7 points per cell -- once an
  // empty block costs two loads and a block of millions of points, the blob of that cloud, is shared by many workgroups:
  // 100 M clustered points 60.5 ms against 70.1 with the mask loop; the first version, whole blocks only, took 100-112)
  double blk_min_pop = 0.0;
  if (const char* e = c->opt("SWZ_MD_ROUNDS_BLOCK_MIN_POP")) blk_min_pop = atof(e);
  bool blk = cl >= 1 && !(c->opt("SWZ_MD_ROUNDS_BLOCK") && atoi(c->opt("SWZ_MD_ROUNDS_BLOCK")) == 0) &&
             (double)sample_points / (double)a.ncells >= blk_min_pop;
  if (blk) {
    g.cl = (uint32_t)cl;
    g.bl = (uint32_t)std::min(cl, 3);
    while (g.bl > 1u && ((uint64_t)sample_nodes << (3u * (g.cl - g.bl))) < 8192ull) --g.bl;  // (the root: 32 768 blocks of 2^3 cells)
    const uint64_t nblocks = (uint64_t)sample_nodes << (3u * (g.cl - g.bl));
    if (nblocks >= (1ull << 31)) blk = false;
    g.blocks = (uint32_t)nblocks;
  }
  if (blk) wl = false;
  float4* wlist = nullptr;
  uint8_t* wcount = nullptr;
  if (wl) {
    // (an accelerator only: when the records do not fit -- 448 bytes per cell -- the loop over the mask does the same job)
    if (c->get("md_pr_wlist", (size_t)a.ncells * PR_WL, &wlist) != SWZ_OK || c->get("md_pr_wcount", (size_t)a.ncells, &wcount) != SWZ_OK) {
      (void)hipGetLastError();
      wl = false;
      wlist = nullptr;
      wcount = nullptr;
    }
  }
  if (blk) {
    SWZ_TRY(c->get("md_pr_bstart", (size_t)g.blocks, &a.bstart));
    SWZ_TRY(c->get("md_pr_bend", (size_t)g.blocks, &a.bend));
    a.block_bits = 3u * g.bl;
    SWZ_HIP(c, memset_large(a.bstart, 0xFF, (size_t)g.blocks * 4, c->stream));
    SWZ_HIP(c, memset_large(a.bend, 0, (size_t)g.blocks * 4, c->stream));
  }
  hipLaunchKernelGGL(pr_init_kernel, dim3(div_up(m, 256)), dim3(256), 0, c->stream, a);
  SWZ_LAUNCH_CHECK(c);
  uint32_t kill_grid = 0;
  if (blk) {
    uint32_t* eoff = nullptr;
    SWZ_TRY(c->get("md_pr_eoff", (size_t)g.blocks + 1, &eoff));
    hipLaunchKernelGGL(pr_block_extra_kernel, dim3(div_up(g.blocks + 1u, 256)), dim3(256), 0, c->stream, a.bstart, a.bend, g.blocks, eoff);
    SWZ_LAUNCH_CHECK(c);
    SWZ_TRY(scan_exclusive_u32(c, eoff, eoff, g.blocks + 1u, nullptr, "mdpe"));
    g.eoff = eoff;
    kill_grid = g.blocks + m / PR_CHUNK + 1u;  // (as many further chunks as there can be: who finds none leaves at once)
  }
  const bool cell_lists = !(c->opt("SWZ_MD_ROUNDS_CELL_LISTS") && atoi(c->opt("SWZ_MD_ROUNDS_CELL_LISTS")) == 0);
  bool prev_list = false;   // the round before ran over a list (list[cur ^ 1], nprev entries: still intact)
  uint32_t nprev = 0;
  for (;;) {
    ++rounds;
    // (a list of more entries than a quarter of the cells: the kernels over the cell grid are the cheaper ones -- a list entry
    // costs its head test, two keys and two node ids, before its cell's work)
    const bool by_list = use_list && cell_lists && (uint64_t)nlist * 4u < a.ncells;
    if (by_list) {
      if (prev_list) {
        hipLaunchKernelGGL(pr_candidates_list_kernel, dim3(div_up(std::max(nprev, 1u), 256)), dim3(256), 0, c->stream, a, cur, a.list[cur ^ 1u], nprev);
      } else {
        // The first round over the list.  Invariant of the list rounds: wonq holds winners of THIS round only --
        // pr_winners_list_kernel writes it for the cells at a list head, pr_candidates_list_kernel clears the cells of the
        // list before.  The rounds over the whole grid leave last round's winners in EVERY cell (they are taken points whose
        // victims are dead already: harmless for the result, but pr_mask_list_kernel would keep pointing the kill pass at them
        // for the rest of the level), so the switch clears them once.
        SWZ_HIP(c, memset_large(a.wonq, 0, (size_t)a.ncells * sizeof(uint64_t), c->stream));
        hipLaunchKernelGGL(pr_candidates_kernel, dim3(cblocks), dim3(256), 0, c->stream, a, cur);
      }
      hipLaunchKernelGGL(pr_winners_list_kernel, dim3(div_up(std::max(nlist, 1u), 256)), dim3(256), 0, c->stream, a, cur, a.list[cur], nlist);
      hipLaunchKernelGGL(pr_mask_list_kernel, dim3(div_up(std::max(nlist, 1u), 256)), dim3(256), 0, c->stream, a, a.list[cur], nlist);
    } else {
      hipLaunchKernelGGL(pr_candidates_kernel, dim3(cblocks), dim3(256), 0, c->stream, a, cur);
      hipLaunchKernelGGL(pr_winners_kernel, dim3(cblocks), dim3(256), 0, c->stream, a, cur);
      if (blk && !use_list) {
        // (the kill pass below reads the winners themselves)
      } else if (wl && !use_list) {
        hipLaunchKernelGGL(pr_mask_kernel<true>, dim3(cblocks), dim3(256), 0, c->stream, a, wlist, wcount);
      } else {
        hipLaunchKernelGGL(pr_mask_kernel<false>, dim3(cblocks), dim3(256), 0, c->stream, a, (float4*)nullptr, (uint8_t*)nullptr);
      }
    }
    const uint32_t threads = use_list ? nlist : m;
    if (threads) {
      if (!use_list && blk) {
        const uint32_t R = (1u << g.bl) + 2u;
        hipLaunchKernelGGL(pr_kill_block_kernel, dim3(kill_grid), dim3(256), (size_t)R * R * R * sizeof(float4), c->stream, a, g, cur);
      } else if (!use_list && wl)
        hipLaunchKernelGGL(pr_kill_wlist_kernel, dim3(div_up(threads, 256)), dim3(256), 0, c->stream, a, cur, wlist, wcount);
      else
        hipLaunchKernelGGL(pr_kill_kernel, dim3(div_up(threads, 256)), dim3(256), 0, c->stream, a, cur, use_list ? 1u : 0u, nlist);
    }
    SWZ_LAUNCH_CHECK(c);
    prev_list = by_list;
    nprev = nlist;
    // who is alive now: a count while they are many, a list (in order) once they are few -- decided on the count this round
    // leaves, not the one it started with: the first round of a level kills nine points in ten, and the second one must
    // not look at all of them again
    uint32_t now = 0;
    bool make_list = false;
    if (use_list) {
      make_list = true;
      SWZ_TRY(fused_scan(c, PrAliveOfListF{a.state, a.list[cur]}, PrListOfListG{a.list[cur], a.list[cur ^ 1u]}, nlist, d_total, "mdpr"));
      SWZ_HIP(c, hipMemcpyAsync(&now, d_total, 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
    } else {
      uint32_t* d_partial = nullptr;
      SWZ_TRY(fused_scan_sums(c, PrAliveF{a.state}, m, d_total, "mdpr", &d_partial));
      SWZ_HIP(c, hipMemcpyAsync(&now, d_total, 4, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      make_list = lists && now < m / 4u && now > 0u;
      if (make_list) {
        SWZ_TRY(c->get("md_pr_list0", (size_t)now + 64, &a.list[0]));
        SWZ_TRY(c->get("md_pr_list1", (size_t)now + 64, &a.list[1]));
        SWZ_TRY(fused_scan_apply(c, PrAliveF{a.state}, PrListG{a.list[cur ^ 1u]}, m, d_partial));
      }
    }
    if (dbg && rounds <= 24) trace += " " + std::to_string(now);
    if (now >= alive && rounds > 1)
      return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE property rounds: a round without progress at level " + std::to_string(plan.level));
    alive = now;
    if (make_list) {
      use_list = true;
      nlist = now;
    }
    cur ^= 1u;
    if (alive == 0) break;
    if (rounds > 4096) return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE property rounds did not terminate");
  }
  *used = true;
  if (rounds_out) *rounds_out += rounds;
  if (dbg) {
    uint32_t hc[CTR_COUNT];
    SWZ_HIP(c, hipMemcpy(hc, lb.counters, sizeof(hc), hipMemcpyDeviceToHost));
    fprintf(stderr, "[swz] MIN_DISTANCE property level %d in rounds: %u pts in %u nodes, %llu cells (cell levels %d, %.1f pts each), spacing %.1f key cells, "
                    "%u rounds, %u exact compares, %.1f ms; alive after each round:%s\n",
            plan.level, sample_points, sample_nodes, (unsigned long long)a.ncells, cl, (double)sample_points / (double)a.ncells, km.T, rounds, hc[PRC_BAND],
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count(), trace.c_str());
  }
  SWZ_HIP(c, hipMemsetAsync(lb.counters + CTR_DBG_HIST, 0, 8 * sizeof(uint32_t), c->stream));
  return SWZ_OK;
}

}  // namespace swz
