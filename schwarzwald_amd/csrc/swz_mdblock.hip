// swz_mdblock.hip -- MIN_DISTANCE on SPARSE levels, round 6: one workgroup per BLOCK of 8 x 8 x 8 cells, everything
// out of LDS, blocks taken in Morton order, decisions published as they are made.
//
// Same result as swz_mdsparse.hip and the sweeps: the lexicographically-first maximal independent set in Morton order
// (PoissonDiskSampling::sample_points, core/tiling/Sampling.h:421-471; SparseGrid::add, core/datastructures/
// SparseGrid.cpp:116-146; GridCell::isDistant, GridCell.cpp:43-58) -- accept(p) <=> no accepted earlier point of p's
// node is closer than the spacing.
//
// Why another design.  The thread-per-point search of swz_mdsparse.hip pays ~17 dependent memory round trips per
// wavefront (cell table -> first record -> further records, four cells at a time), writes a 32-byte neighbour slot per
// point (36.9 GB per 1 B-point step) and needs a second pass that polls one state byte per neighbour: 105-110 ms per
// step, 4.1x its algorithmic bytes, bound by the latency of those chains.  Here:
//   * the sorted keys ARE the records: a block of 8^3 cells is one run of the level's sorted keys, found through a table
//     of {first, end} per GRANULE (2 x 2 x 2 cells, 8 bytes per granule: 0.5 instead of 4.3 GB at level 2 of the 1 B run);
//     its one-cell halo is covered by the <= 152 granules around it, of which only the EARLIER ones (smaller Morton code
//     than the block: about half) can hold earlier neighbours.  Three dependent round trips per BLOCK (ticket, granule
//     entries, keys + states), each with all loads of the workgroup in flight;
//   * the search runs on LDS: a 10^3 cell index over the staged points, ~14 cells x ~1 point per own point; the earlier
//     neighbours that are still undecided go into an LDS list (<= 8 two-byte entries per point), nothing per neighbour is
//     ever written to memory;
//   * decisions are taken in the same launch: a point without undecided earlier neighbours is final at once; the others
//     iterate over LDS states.  What a block needs from other blocks are the states of halo points, i.e. of EARLIER
//     blocks of the same node.  Blocks are handed out by tickets in Morton order (per node; nodes interleaved so that the
//     resident workgroups spread over many nodes), so every block a workgroup waits for has been started before it:
//     waiting cannot deadlock.  States are two bits per point in one array, published with agent-scope atomic ORs as soon
//     as they are known and polled with relaxed agent-scope loads (MI355X_MICROARCH.md: "8-B agent atomics both sides").
//   * nothing here can wait forever: a workgroup that waits longer than the time-out raises the abort word, every
//     workgroup leaves at its next look, and the host returns an error (never a restart of the process).
// A block or halo that does not fit the LDS capacity chosen for the level (locally dense data) raises the same word with
// another code; the host then runs the level through swz_mdsparse.hip / the sweep as before -- every decision taken so
// far is exact and is simply taken again.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "swz_level.h"
#include "swz_scan.h"

namespace swz {

#ifndef SB_MINW
#define SB_MINW 6   // wavefronts per SIMD the kernel is compiled for (80 registers; measured against 5: level 2 of the 1 B run 69 -> 64 ms)
#endif
constexpr int SB_THREADS = 256;
constexpr int SB_K = 4;               // undecided earlier neighbours recorded per point (0.73 expected at level 2 of the 1 B run); more: the point searches again
constexpr int SB_PEND = 63;            // in-band pairs a wavefront puts aside per block (more: compared on the spot)
constexpr int SB_PMAX = 12;           // staged points per thread: a block and its halo hold at most SB_THREADS * SB_PMAX points
constexpr uint32_t SB_NC = 64;        // ticket counters (one per 128-byte line): nodes sn with sn % SB_NC == k draw from counter k
constexpr uint32_t SB_CTR_STRIDE = 32;
constexpr uint32_t SB_NONE = 0xFFFFFFFFu;
enum : uint32_t { SB_U = 0, SB_A = 1, SB_R = 2 };
enum : uint32_t { SB_ABORT_NONE = 0, SB_ABORT_CAPACITY = 1, SB_ABORT_TIMEOUT = 2 };
// words behind the ticket counters
enum : uint32_t { SBW_ABORT = 0, SBW_MAX_OWN = 1, SBW_MAX_HALO = 2, SBW_ITER = 3, SBW_BLOCKS = 4, SBW_WAITS = 5, SBW_RESEARCH = 6, SBW_STEPS = 7,
                  SBW_T0 = 8 /* -DSWZ_SB_STATS: ticks (10 ns) of thread 0 per phase: ticket, granules, stage, index, search, decide, tail */, SBW_COUNT = 16 };
#ifdef SWZ_SB_STATS
#define SB_T(i) do { if (tid == 0) { const uint64_t now_ = wall_clock64(); tacc[i] += (uint32_t)(now_ - tlast); tlast = now_; } } while (0)
#else
#define SB_T(i) do { } while (0)
#endif
// per-workgroup LDS words
enum : uint32_t { SBM_SN = 0, SBM_B = 1, SBM_FIRST = 2, SBM_NOWN = 3, SBM_ABORT = 4, SBM_COUNT = 8 };

struct SbArgs {
  const uint64_t* akey;
  uint32_t m;
  const uint2* gtab;       // [sampled node][granule code] -> {first, end} active index of the granule's run; end == 0: empty
  uint32_t* st2;           // two bits per active point: SB_U / SB_A / SB_R
  uint8_t* taken;
  // exact position of active point i: xyz[3 * perm[aidx ? aidx[i] : i]] (pairs inside the quantisation band only)
  const uint32_t* aidx;
  const uint32_t* perm;
  const double* xyz;
  float f_lo, f_hi;
  uint32_t i_lo, i_hi;     // the same thresholds for squared distances evaluated in integers (exact): floor(f_lo), ceil(f_hi)
  double sq_spacing;
  uint32_t cell_bits;      // key coordinate >> cell_bits = absolute cell coordinate
  uint32_t cl;             // cell levels below the node: 2^cl cells per axis and node
  uint32_t ns;             // sampled nodes
  uint32_t* ctr;           // SB_NC ticket counters (stride SB_CTR_STRIDE words), then SBW_* words
  uint32_t own_cap, halo_cap;
  uint64_t timeout_ticks;  // wall_clock64 ticks (100 MHz) a wavefront may wait without any progress
  uint32_t dbg;            // SWZ_SP_BLOCK_DBG: timing experiments that BREAK the result (1: no candidate loop, 2: no reach test, 4: no order test, 8: undecided halo points count as rejected)
};

__device__ __forceinline__ uint32_t sb_expand3(uint32_t v) {  // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ uint32_t sb_load_word(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// LDS states are read while other wavefronts of the workgroup write them, without a barrier in between
__device__ __forceinline__ uint32_t sb_lds_state(const uint8_t* st, uint32_t q) {
  return __hip_atomic_load(st + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sb_lds_set(uint8_t* st, uint32_t q, uint32_t v) {
  __hip_atomic_store(st + q, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A staged point in eight bytes, two formats.
//  * narrow (cells of at most 4096 key cells: every level below 0 at spacing = diagonal / 250): the coordinates relative to
//    the corner of the 10^3-cell region around the block, 16 bits each (the region spans at most 40 960 key cells), and the
//    point's region cell (4 bits per axis) -- squared distances are exact 32-bit integers, three subtractions and three
//    24-bit multiply-adds per pair;
//  * wide: the three 21-bit key coordinates; distances in float like every other path on keys (exact differences).
__device__ __forceinline__ uint64_t sb_pack(uint32_t x, uint32_t y, uint32_t z) { return (uint64_t)x | ((uint64_t)y << 21) | ((uint64_t)z << 42); }
__device__ __forceinline__ void sb_unpack(uint64_t p, uint32_t& x, uint32_t& y, uint32_t& z) {
  const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32);
  x = lo & 0x1FFFFFu;
  y = ((lo >> 21) | (hi << 11)) & 0x1FFFFFu;
  z = hi >> 10;
}
constexpr uint32_t SB_CELL_NONE = 0xFFFu;  // (ix | iy << 4 | iz << 8 with a coordinate of 15: never a region cell)

// the reference's compare on the original positions (GridCell.cpp:52)
__device__ __forceinline__ bool sb_exact_near(const SbArgs& a, uint32_t gp, uint32_t gq) {
  const uint32_t sp = a.aidx ? a.aidx[gp] : gp, sq = a.aidx ? a.aidx[gq] : gq;
  const double* u = a.xyz + (size_t)a.perm[sp] * 3;
  const double* v = a.xyz + (size_t)a.perm[sq] * 3;
  return sq_dist(u[0], u[1], u[2], v[0], v[1], v[2]) < a.sq_spacing;
}

struct SbLds {
  uint64_t* pts;    // [own_cap + halo_cap] packed key coordinates; the block's own points first, in Morton order
  uint32_t* cse;    // [1024] region cell -> first | end << 16 (LDS indices); valid where the cell's occupancy bit is set
  uint32_t* occ;    // [128] (iz, iy) -> bit ix: the region cell holds staged points
  uint32_t* soff;   // [224] halo slot (region granule 0..215) -> LDS offset of its run; [216] = total
  uint32_t* sfirst; // [224] ... -> active index of its first point
  uint32_t* pub;    // own state bits to publish, by word of st2
  uint32_t* misc;   // SBM_*
  uint32_t* scan;   // [8] block scan scratch
  uint16_t* nbr;    // [own_cap][SB_K]; while the points are staged: rc[own_cap + halo_cap], the region cell of every point
  uint8_t* st;      // [own_cap + halo_cap]
  uint32_t* cnt4;   // [own_cap / 4] undecided earlier neighbours of own point j in byte j % 4 (counted with 32-bit LDS atomics)
  uint32_t* pend;   // [4][SB_PEND + 1] per wavefront: pairs inside the quantisation band, j << 16 | q; [SB_PEND]: how many
};
__host__ __device__ inline size_t sb_lds_bytes(uint32_t own_cap, uint32_t halo_cap) {
  const size_t tot = (size_t)own_cap + halo_cap;
  size_t b = tot * 8;                                            // pts
  b += 1024 * 4 + 128 * 4 + 224 * 4 + 224 * 4;                   // cse, occ, soff, sfirst
  b += (((size_t)own_cap / 16 + 5) & ~(size_t)1) * 4 + SBM_COUNT * 4 + 8 * 4;   // pub (even: what follows stays 8-byte aligned), misc, scan
  b += std::max((size_t)own_cap * SB_K, tot) * 2;                // nbr / rc
  b += ((tot + 3) & ~(size_t)3) + own_cap + 4 * (SB_PEND + 1) * 4; // st, cnt4, pend
  return (b + 15) & ~(size_t)15;
}
__device__ __forceinline__ SbLds sb_carve(unsigned char* smem, uint32_t own_cap, uint32_t halo_cap) {
  const uint32_t tot = own_cap + halo_cap;
  SbLds l;
  l.pts = reinterpret_cast<uint64_t*>(smem);
  l.cse = reinterpret_cast<uint32_t*>(l.pts + tot);
  l.occ = l.cse + 1024;
  l.soff = l.occ + 128;
  l.sfirst = l.soff + 224;
  l.pub = l.sfirst + 224;
  l.misc = l.pub + ((own_cap / 16 + 5u) & ~1u);
  l.scan = l.misc + SBM_COUNT;
  l.nbr = reinterpret_cast<uint16_t*>(l.scan + 8);
  l.st = reinterpret_cast<uint8_t*>(l.nbr + max(own_cap * (uint32_t)SB_K, tot));
  l.cnt4 = reinterpret_cast<uint32_t*>(l.st + ((tot + 3u) & ~3u));
  l.pend = l.cnt4 + own_cap / 4u;
  return l;
}

// what a wavefront needs to know about the block it works on
struct SbBlock {
  uint32_t first;   // active index of the block's first point
  uint32_t n_own;
  uint32_t bx8, by8, bz8;  // the block's origin in cells of its node
};
// region cell of a point as ix | iy << 4 | iz << 8 (0..9 each), SB_CELL_NONE outside the 10^3 cells around the block
__device__ __forceinline__ uint32_t sb_region_cell(const SbArgs& a, const SbBlock& k, uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t cmask = (1u << a.cl) - 1u;
  const uint32_t ix = ((x >> a.cell_bits) & cmask) - k.bx8 + 1u, iy = ((y >> a.cell_bits) & cmask) - k.by8 + 1u, iz = ((z >> a.cell_bits) & cmask) - k.bz8 + 1u;
  return (ix < 10u && iy < 10u && iz < 10u) ? (ix | (iy << 4) | (iz << 8)) : SB_CELL_NONE;
}
__device__ __forceinline__ uint32_t sb_cell_index(uint32_t c) { return (c & 15u) + 10u * ((c >> 4) & 15u) + 100u * (c >> 8); }
template <bool WIDE>
__device__ __forceinline__ uint64_t sb_make_point(const SbArgs& a, const SbBlock& k, uint32_t x, uint32_t y, uint32_t z, uint32_t* cell) {
  const uint32_t c = sb_region_cell(a, k, x, y, z);
  *cell = c;
  if (WIDE) return sb_pack(x, y, z);
  // relative to the region's corner (one cell below the block's origin, inside the node); points outside the region keep
  // whatever the low 16 bits say -- nobody ever looks at them
  const uint32_t nmask = (1u << (a.cell_bits + a.cl)) - 1u;
  const uint32_t xr = (x & nmask) - ((k.bx8 - 1u) << a.cell_bits), yr = (y & nmask) - ((k.by8 - 1u) << a.cell_bits),
                 zr = (z & nmask) - ((k.bz8 - 1u) << a.cell_bits);
  return (uint64_t)((xr & 0xFFFFu) | (yr << 16)) | ((uint64_t)((zr & 0xFFFFu) | (c << 16)) << 32);
}
// the halo run that holds staged point j >= n_own: soff ascends, [216] = total (branch-free: eight steps for every lane)
__device__ __forceinline__ uint32_t sb_halo_slot(const SbLds& l, uint32_t j) {
  uint32_t lo = 0;
#pragma unroll
  for (uint32_t step = 128u; step; step >>= 1)
    if (lo + step <= 216u && l.soff[lo + step] <= j) lo += step;
  return lo;
}
// active index of staged point q (a halo point's is looked up: only polls and compares on the original positions want it)
__device__ __forceinline__ uint32_t sb_active_index(const SbLds& l, const SbBlock& k, uint32_t q) {
  if (q < k.n_own) return k.first + q;
  const uint32_t slot = sb_halo_slot(l, q);
  return l.sfirst[slot] + (q - l.soff[slot]);
}

// Which of the 27 cells around own cell (ix, iy, iz) -- 1..8 each -- can hold EARLIER points.  Bit 9 * zi + 3 * yi + xi,
// xi = 0 / 1 / 2 for the cell at x - 1 / x / x + 1.  Cells outside the block were staged only when their granule precedes
// the block in Morton order, so every one of them counts.  Inside the block the order of two adjacent cells is decided by
// the axis whose coordinate changes at the highest bit: a step of -1 flips the bits up to the lowest set one, a step of
// +1 up to the lowest clear one; in the interleaved code bit h of x sits at 3h + 2, of y at 3h + 1, of z at 3h.  The cell
// is earlier when the highest flipped bit belongs to an axis that steps down.
__device__ __forceinline__ uint32_t sb_earlier_mask(uint32_t ix, uint32_t iy, uint32_t iz) {
  const uint32_t lx = ix - 1u, ly = iy - 1u, lz = iz - 1u;
  const int mx = 3 * (int)__builtin_ctz(lx | 8u) + 2, my = 3 * (int)__builtin_ctz(ly | 8u) + 1, mz = 3 * (int)__builtin_ctz(lz | 8u);
  const int px = 3 * (int)__builtin_ctz(~lx) + 2, py = 3 * (int)__builtin_ctz(~ly) + 1, pz = 3 * (int)__builtin_ctz(~lz);
  uint32_t m = 0x361Bu;  // no axis steps up: bits 0, 1, 3, 4, 9, 10, 12, 13
  // two axes, one up and one down
  m |= (mx > py ? 1u : 0u) << 15;  // x-, y+
  m |= (my > px ? 1u : 0u) << 11;  // x+, y-
  m |= (mx > pz ? 1u : 0u) << 21;  // x-, z+
  m |= (mz > px ? 1u : 0u) << 5;   // x+, z-
  m |= (my > pz ? 1u : 0u) << 19;  // y-, z+
  m |= (mz > py ? 1u : 0u) << 7;   // y+, z-
  // three axes
  m |= (max(mx, my) > pz ? 1u : 0u) << 18;  // x-, y-, z+
  m |= (max(mx, mz) > py ? 1u : 0u) << 6;   // x-, y+, z-
  m |= (max(my, mz) > px ? 1u : 0u) << 2;   // x+, y-, z-
  m |= (mx > max(py, pz) ? 1u : 0u) << 24;  // x-, y+, z+
  m |= (my > max(px, pz) ? 1u : 0u) << 20;  // x+, y-, z+
  m |= (mz > max(px, py) ? 1u : 0u) << 8;   // x+, y+, z-
  // the halo
  const uint32_t hx = (ix == 1u ? 1u : 0u) | (ix == 8u ? 4u : 0u);
  const uint32_t hy = (iy == 1u ? 7u : 0u) | (iy == 8u ? 7u << 6 : 0u);
  const uint32_t hz = (iz == 1u ? 0x1FFu : 0u) | (iz == 8u ? 0x1FFu << 18 : 0u);
  return m | hx * 0x1249249u | hy * 0x40201u | hz;
}

// An own point as the search loops want it: coordinates, the corner of its 3 x 3 x 3 cells in the cell index, and which
// of those cells hold points that can matter -- occupied (nine rows of occupancy bits), not later in Morton order, and,
// in the narrow format, within reach: the squared gap between the point and the cell, exact in integers (the region's
// corner is a cell corner, so the low bits of a relative coordinate are the offset inside the cell).
struct SbOwn {
  uint32_t ux, uy, uz;
  float fx, fy, fz;
  uint32_t corner;
  uint32_t mask;
};
template <bool WIDE>
__device__ __forceinline__ SbOwn sb_own_point(const SbArgs& a, const SbLds& l, const SbBlock& k, uint32_t j) {
  SbOwn p;
  const uint64_t me = l.pts[j];
  uint32_t cell;
  if (WIDE) {
    sb_unpack(me, p.ux, p.uy, p.uz);
    cell = sb_region_cell(a, k, p.ux, p.uy, p.uz);
  } else {
    const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
    p.ux = lo & 0xFFFFu;
    p.uy = lo >> 16;
    p.uz = hi & 0xFFFFu;
    cell = hi >> 16;
  }
  p.fx = (float)p.ux;
  p.fy = (float)p.uy;
  p.fz = (float)p.uz;
  const uint32_t ix = cell & 15u, iy = (cell >> 4) & 15u, iz = cell >> 8;  // 1..8: an own point
  uint32_t mask = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t row = l.occ[(iz + (uint32_t)(i / 3) - 1u) * 10u + (iy + (uint32_t)(i % 3) - 1u)];
    mask |= ((row >> (ix - 1u)) & 7u) << (3 * i);
  }
  if (!(a.dbg & 4u)) mask &= sb_earlier_mask(ix, iy, iz);
  if (!WIDE && !(a.dbg & 2u)) {
    const uint32_t cs = 1u << a.cell_bits;
    const int ox = (int)(p.ux & (cs - 1u)), oy = (int)(p.uy & (cs - 1u)), oz = (int)(p.uz & (cs - 1u));
    const uint32_t gx[3] = {(uint32_t)__mul24(ox, ox), 0u, (uint32_t)__mul24((int)cs - ox, (int)cs - ox)};
    const uint32_t gy[3] = {(uint32_t)__mul24(oy, oy), 0u, (uint32_t)__mul24((int)cs - oy, (int)cs - oy)};
    const uint32_t gz[3] = {(uint32_t)__mul24(oz, oz), 0u, (uint32_t)__mul24((int)cs - oz, (int)cs - oz)};
    uint32_t reach = 0;
#pragma unroll 1
    for (int zi = 0; zi < 3; ++zi) {  // (rolled: the 27 sums at once cost the kernel 30 registers and a wavefront per SIMD)
      const uint32_t gzz = zi == 0 ? gz[0] : (zi == 1 ? 0u : gz[2]);
      uint32_t r9 = 0;
#pragma unroll
      for (int b = 0; b < 9; ++b)
        if (gx[b % 3] + gy[b / 3] + gzz < a.i_hi) r9 |= 1u << b;
      reach |= r9 << (9 * zi);
    }
    mask &= reach;
  }
  if (a.dbg & 1u) mask = 0;
  p.mask = mask;
  p.corner = (ix - 1u) + 10u * (iy - 1u) + 100u * (iz - 1u);
  return p;
}
// squared distance of own point p to staged point word o: closer than the spacing for sure / possibly
template <bool WIDE>
__device__ __forceinline__ void sb_compare(const SbArgs& a, const SbOwn& p, uint64_t o, bool& sure, bool& maybe) {
  if (WIDE) {
    uint32_t vx, vy, vz;
    sb_unpack(o, vx, vy, vz);
    const float ddx = p.fx - (float)vx, ddy = p.fy - (float)vy, ddz = p.fz - (float)vz;
    const float d2 = ddx * ddx + ddy * ddy + ddz * ddz;
    sure = d2 < a.f_lo;
    maybe = d2 < a.f_hi;
  } else {
    const uint32_t lo = (uint32_t)o, hi = (uint32_t)(o >> 32);
    const int ddx = (int)p.ux - (int)(lo & 0xFFFFu), ddy = (int)p.uy - (int)(lo >> 16), ddz = (int)p.uz - (int)(hi & 0xFFFFu);
    const uint32_t d2 = (uint32_t)(__mul24(ddx, ddx) + __mul24(ddy, ddy) + __mul24(ddz, ddz));  // < 3 * 2^26: adjacent cells
    sure = d2 < a.i_lo;
    maybe = d2 < a.i_hi;
  }
}
__device__ __forceinline__ uint32_t sb_cell_offset(uint32_t b) {  // bit of the 27-cell mask -> offset in the cell index
  const uint32_t dz = (b * 57u) >> 9, rem = b - 9u * dz, dy = (rem * 11u) >> 5, dx = rem - 3u * dy;
  return dx + 10u * dy + 100u * dz;
}

// Visits every EARLIER staged point closer than the spacing to own point j.  f(q) returns false to stop.  (The general
// form, for the rare point that has to search again; the search proper is sb_search below.)
template <bool WIDE, typename F>
__device__ __forceinline__ void sb_visit(const SbArgs& a, const SbLds& l, const SbBlock& k, uint32_t j, F f) {
  const SbOwn p = sb_own_point<WIDE>(a, l, k, j);
  uint32_t mask = p.mask;
  uint32_t q = 0, e = 0;
  for (;;) {
    if (q >= e) {
      if (!mask) break;
      const uint32_t b = (uint32_t)__ffs((int)mask) - 1u;
      mask &= mask - 1u;
      const uint32_t e2 = l.cse[p.corner + sb_cell_offset(b)];
      q = e2 & 0xFFFFu;
      e = e2 >> 16;
      if (q < k.n_own) e = min(e, j);  // a cell of the block itself: earlier points only (own points are staged in Morton order)
      if (q >= e) continue;
    }
    bool sure, maybe;
    sb_compare<WIDE>(a, p, l.pts[q], sure, maybe);
    if (maybe) {
      if (sure || sb_exact_near(a, k.first + j, sb_active_index(l, k, q))) {
        if (!f(q)) return;
      }
    }
    ++q;
  }
}

// The search of own point j (lanes without a point: active = false): its earlier neighbours go to mine[] (the first
// SB_K), *cnt counts them -- whatever their state: the states are looked at after the loop.  One loop over all candidates
// of the point -- a lane either takes its next cell or tests its next candidate, so the wavefront runs as long as its
// busiest lane has candidates, not 27 x the fullest cell -- written without divergent branches: every lane executes every
// step on clamped indices and the outcome is selected.  (With branches the loop spent a third of its instructions on the
// execution mask: 2 300 scalar instructions per wavefront and block, and the scalar unit -- one per CU -- was the kernel's
// bound.)  A wavefront issues an instruction every four cycles at best and every LDS read it waits for costs ~100 more, so
// a step is kept short and has ONE read on its critical path: the candidate; the entry of the lane's next cell is requested
// a step ahead and the neighbours' states are not read here at all.
// Pairs inside the quantisation band go to the wavefront's pending list.
template <bool WIDE>
__device__ __forceinline__ void sb_search(const SbArgs& a, const SbLds& l, const SbBlock& k, uint32_t j, bool active, uint16_t* mine,
                                          uint32_t* pend, uint32_t* cnt_out, uint32_t* steps) {
  const SbOwn p = sb_own_point<WIDE>(a, l, k, active ? j : 0u);
  uint32_t mask = active ? p.mask : 0u;
  uint32_t q = 0, e = 0, cnt = 0;
  uint32_t ne2 = l.cse[mask ? p.corner + sb_cell_offset((uint32_t)__ffs((int)mask) - 1u) : 0u];  // the first cell's entry
  while (__ballot(active)) {
    ++*steps;
    // lanes whose cell is used up take their next one (its entry is here already) and ask for the one after it
    const bool need = active && q >= e;
    const bool fetch = need && mask != 0u;
    const uint32_t s2 = ne2 & 0xFFFFu, t2 = ne2 >> 16;
    const uint32_t t3 = s2 < k.n_own ? min(t2, j) : t2;  // a cell of the block itself: earlier points only
    mask = fetch ? (mask & (mask - 1u)) : mask;
    q = fetch ? s2 : q;
    e = fetch ? t3 : e;
    active = active && !(need && !fetch);
    if (__ballot(fetch)) {
      const uint32_t nxt = l.cse[(fetch && mask) ? p.corner + sb_cell_offset((uint32_t)__ffs((int)mask) - 1u) : 0u];
      ne2 = fetch ? nxt : ne2;
    }
    // lanes that have a candidate test it
    const bool valid = active && q < e;
    bool sure, maybe;
    sb_compare<WIDE>(a, p, l.pts[valid ? q : 0u], sure, maybe);
    maybe = maybe && valid;
    if (__ballot(maybe)) {
      const bool band = maybe && !sure;
      bool near = maybe && sure;
      if (__ballot(band)) {
        if (band) {
          const uint32_t slot = atomicAdd(&pend[SB_PEND], 1u);
          if (slot < (uint32_t)SB_PEND) pend[slot] = (j << 16) | q;
          else near = sb_exact_near(a, k.first + j, sb_active_index(l, k, q));
        }
      }
      if (near && cnt < (uint32_t)SB_K) mine[cnt] = (uint16_t)q;
      cnt += near ? 1u : 0u;
    }
    q += valid ? 1u : 0u;
  }
  *cnt_out = cnt;
}

// state of staged point q as far as anybody knows: halo points still undecided are looked up again
__device__ __forceinline__ uint32_t sb_state_of(const SbArgs& a, const SbLds& l, const SbBlock& k, uint32_t q, bool poll) {
  uint32_t s = sb_lds_state(l.st, q);
  if (poll && s == SB_U && q >= k.n_own) {
    const uint32_t gi = sb_active_index(l, k, q);
    s = (sb_load_word(a.st2 + (gi >> 4)) >> ((gi & 15u) * 2u)) & 3u;
    if (s != SB_U) sb_lds_set(l.st, q, s);
  }
  return s;
}

template <bool WIDE>
__global__ __launch_bounds__(SB_THREADS, SB_MINW) void sb_block_kernel(SbArgs a) {
  extern __shared__ __align__(16) unsigned char sb_smem[];
  const SbLds l = sb_carve(sb_smem, a.own_cap, a.halo_cap);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t nb_per_node = 1u << (3u * (a.cl - 3u));
  const uint32_t gmax = 1u << (a.cl - 1u);
  const uint64_t gran_per_node = 1ull << (3u * (a.cl - 1u));
  uint32_t* words = a.ctr + SB_NC * SB_CTR_STRIDE;
  // counters are tried from this one on (the XCD of the workgroup first: the blocks of a node then mostly run on one
  // XCD, whose L2 holds the keys the blocks before it have read)
  uint32_t kc = (blockIdx.x & 7u) + 8u * ((blockIdx.x >> 3) % (SB_NC / 8u));
  uint32_t tried = 0;
  uint32_t my_blocks = 0, my_iters = 0, my_waits = 0, my_research = 0, my_steps = 0;
  uint16_t* rc = l.nbr;  // (alias: region cells while staging, neighbour lists afterwards)
#ifdef SWZ_SB_STATS
  uint32_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tlast = wall_clock64();
#endif
  if (tid == 0) l.misc[SBM_ABORT] = 0;

  for (;;) {
    // ---- the next block in Morton order of some node
    if (tid == 0) {
      uint32_t sn = SB_NONE, b = 0;
      if (l.misc[SBM_ABORT] == 0 && sb_load_word(words + SBW_ABORT) == SB_ABORT_NONE) {
        while (tried < SB_NC) {
          const uint32_t nodes_k = a.ns > kc ? (a.ns - kc + SB_NC - 1u) / SB_NC : 0u;
          if (nodes_k) {
            const uint32_t t = __hip_atomic_fetch_add(a.ctr + kc * SB_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint64_t)t < (uint64_t)nodes_k * nb_per_node) {
              sn = (t % nodes_k) * SB_NC + kc;
              b = t / nodes_k;
              break;
            }
          }
          kc = (kc + 1u) % SB_NC;
          ++tried;
        }
      }
      l.misc[SBM_SN] = sn;
      l.misc[SBM_B] = b;
      l.misc[SBM_FIRST] = SB_NONE;
      l.misc[SBM_NOWN] = 0;
    }
    if (tid < 128u) l.occ[tid] = 0u;
    __syncthreads();
    SB_T(0);
    const uint32_t sn = l.misc[SBM_SN], b = l.misc[SBM_B];
    if (sn == SB_NONE) break;

    // ---- the granules of the block (4 x 4 x 4) and around it (6 x 6 x 6): their runs
    const uint32_t bx = contract_bits_by_3_u32(b >> 2), by = contract_bits_by_3_u32(b >> 1), bz = contract_bits_by_3_u32(b);
    const uint2* gt = a.gtab + (uint64_t)sn * gran_per_node;
    uint32_t own_cnt = 0, halo_cnt = 0, run_first = 0;
    if (tid < 216u) {
      const uint32_t rx = tid % 6u, ry = (tid / 6u) % 6u, rz = tid / 36u;
      const int gx = (int)(bx * 4u + rx) - 1, gy = (int)(by * 4u + ry) - 1, gz = (int)(bz * 4u + rz) - 1;
      const bool inner = rx - 1u < 4u && ry - 1u < 4u && rz - 1u < 4u;
      if (gx >= 0 && gy >= 0 && gz >= 0 && gx < (int)gmax && gy < (int)gmax && gz < (int)gmax) {
        const uint32_t code = (sb_expand3((uint32_t)gx) << 2) | (sb_expand3((uint32_t)gy) << 1) | sb_expand3((uint32_t)gz);
        if (inner || code < (b << 6)) {  // around the block: only granules EARLIER in Morton order can hold earlier points
          const uint2 e = gt[code];
          if (e.y) {
            run_first = e.x;
            if (inner) own_cnt = e.y - e.x; else halo_cnt = e.y - e.x;
          }
        }
      }
      if (own_cnt) {
        atomicMin(&l.misc[SBM_FIRST], run_first);
        atomicAdd(&l.misc[SBM_NOWN], own_cnt);
      }
      l.sfirst[tid] = run_first;
    }
    uint32_t n_halo = 0;
    const uint32_t halo_off = block_excl_sum<SB_THREADS>(min(halo_cnt, 0xFFFFFu), l.scan, n_halo);
    SbBlock k;
    k.n_own = l.misc[SBM_NOWN];
    k.first = l.misc[SBM_FIRST];
    k.bx8 = bx * 8u;
    k.by8 = by * 8u;
    k.bz8 = bz * 8u;
    const uint32_t n_own = k.n_own;
    if (n_own == 0) {  // (uniform: every thread has the same totals)
      __syncthreads();
      continue;
    }
    const uint32_t total = n_own + n_halo;
    if (n_own > a.own_cap || n_halo > a.halo_cap) {
      if (tid == 0) {
        __hip_atomic_fetch_max(words + SBW_MAX_OWN, n_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(words + SBW_MAX_HALO, n_halo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(words + SBW_ABORT, (uint32_t)SB_ABORT_CAPACITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
    if (tid < 216u) l.soff[tid] = n_own + halo_off;
    if (tid == 216u) l.soff[216] = total;
    const uint32_t w0 = k.first >> 4;                                  // first word of st2 the block's points touch
    const uint32_t nw = ((k.first + n_own - 1u) >> 4) - w0 + 1u;       // ... and how many
    for (uint32_t w = tid; w < nw; w += SB_THREADS) l.pub[w] = 0u;
    __syncthreads();
    SB_T(1);

    // ---- stage the points: key coordinates, region cell, state.  Three loads of a thread in flight together (all six:
    // 30 registers more, a wavefront per SIMD less).
#pragma unroll 1
    for (uint32_t half = 0; half < (uint32_t)SB_PMAX; half += 3u) {
      if (half * SB_THREADS >= total) break;
      uint32_t gi[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t j = tid + (half + (uint32_t)p) * SB_THREADS;
        gi[p] = SB_NONE;
        if (j < n_own) {
          gi[p] = k.first + j;  // the block is one run of the sorted keys
        } else if (j < total) {
          // the halo run that holds staged point j: soff ascends, [216] = total (branch-free: eight steps for every lane)
          const uint32_t lo = sb_halo_slot(l, j);
          gi[p] = l.sfirst[lo] + (j - l.soff[lo]);
        }
      }
      uint64_t key[3];
      uint32_t sw[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t j = tid + (half + (uint32_t)p) * SB_THREADS;
        key[p] = 0;
        sw[p] = 0;
        if (gi[p] != SB_NONE) {
          key[p] = a.akey[gi[p]];
          if (j >= n_own) sw[p] = sb_load_word(a.st2 + (gi[p] >> 4));
        }
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint32_t j = tid + (half + (uint32_t)p) * SB_THREADS;
        if (gi[p] == SB_NONE) continue;
        uint32_t x, y, z, cell;
        key_coords_u32(key[p], x, y, z);
        l.pts[j] = sb_make_point<WIDE>(a, k, x, y, z, &cell);
        rc[j] = (uint16_t)cell;
        l.st[j] = (uint8_t)((sw[p] >> ((gi[p] & 15u) * 2u)) & 3u);
        if ((a.dbg & 8u) && j >= n_own && l.st[j] == SB_U) l.st[j] = SB_R;  // (timing experiment: nobody waits for another block)
      }
    }
    __syncthreads();
    SB_T(2);
    // ---- the cell index of the region: a cell's points are one run of the staged points
    for (uint32_t j = tid; j < total; j += SB_THREADS) {
      const uint32_t r = rc[j];
      if (r == SB_CELL_NONE) continue;
      uint16_t* e = reinterpret_cast<uint16_t*>(l.cse + sb_cell_index(r));
      if (j == 0 || rc[j - 1] != r) {
        e[0] = (uint16_t)j;
        atomicOr(&l.occ[(r >> 8) * 10u + ((r >> 4) & 15u)], 1u << (r & 15u));
      }
      if (j + 1 == total || rc[j + 1] != r) e[1] = (uint16_t)(j + 1);
    }
    __syncthreads();
    SB_T(3);

    // ---- from here on every wavefront works on its own share of the block's points -- whole words of the state array --
    // without a barrier: search, decisions, publication.  Other wavefronts' decisions are read from the LDS states.
    const uint32_t wa = wave * nw / 4u, wb = (wave + 1u) * nw / 4u;
    const uint32_t ja = max(k.first, (w0 + wa) << 4) - k.first;
    const uint32_t jb = wa < wb ? min(k.first + n_own, (w0 + wb) << 4) - k.first : ja;
    // search: the undecided earlier neighbours of every own point.  Pairs inside the quantisation band are put aside: their
    // compare on the original positions is a chain of dependent scattered loads that would stall the whole wavefront inside
    // the loop (half of the wavefronts meet one); afterwards the lanes take one pair each, all loads in flight together.
    uint32_t* pend = l.pend + wave * (uint32_t)(SB_PEND + 1);
    uint8_t* cnt1 = reinterpret_cast<uint8_t*>(l.cnt4);
    if (lane == 0) pend[SB_PEND] = 0u;
    for (uint32_t base = ja; base < jb; base += 64u) {
      const uint32_t j = base + lane;
      const bool have = j < jb;
      uint32_t cnt = 0;
      sb_search<WIDE>(a, l, k, j, have, l.nbr + (size_t)(have ? j : 0u) * SB_K, pend, &cnt, &my_steps);
      if (have) cnt1[j] = (uint8_t)min(cnt, 200u);
    }
    {
      const uint32_t np = min(pend[SB_PEND], (uint32_t)SB_PEND);
      for (uint32_t i = lane; i < np; i += 64u) {
        const uint32_t j = pend[i] >> 16, q = pend[i] & 0xFFFFu;
        if (!sb_exact_near(a, k.first + j, sb_active_index(l, k, q))) continue;
        // one more neighbour of j: a byte of a counter word, bumped atomically
        const uint32_t slot = (atomicAdd(&l.cnt4[j >> 2], 1u << (8u * (j & 3u))) >> (8u * (j & 3u))) & 0xFFu;
        if (slot < (uint32_t)SB_K) l.nbr[(size_t)j * SB_K + slot] = (uint16_t)q;
      }
    }
    // The neighbours' states as far as they are known now: an accepted one rejects the point, rejected ones are dropped
    // from the list, a point that keeps none is taken.  (A list that could not hold all neighbours stays as it is: the
    // decisions below search again once its entries are all rejected.)
    for (uint32_t base = ja; base < jb; base += 64u) {
      const uint32_t j = base + lane;
      if (j >= jb) continue;
      const uint32_t cnt = cnt1[j];
      uint32_t dec = SB_U;
      if (cnt == 0u) {
        dec = SB_A;
      } else {
        static_assert(SB_K == 4, "four two-byte entries = one eight-byte load");
        uint64_t* slot = reinterpret_cast<uint64_t*>(l.nbr + (size_t)j * SB_K);
        const uint64_t n4 = *slot;
        const uint32_t c4 = min(cnt, (uint32_t)SB_K);
        uint32_t st[SB_K];
#pragma unroll
        for (int i = 0; i < SB_K; ++i) st[i] = (uint32_t)i < c4 ? sb_lds_state(l.st, (uint32_t)(n4 >> (16 * i)) & 0xFFFFu) : (uint32_t)SB_R;
        uint64_t kept = 0;
        uint32_t nk = 0;
        bool any_a = false;
#pragma unroll
        for (int i = 0; i < SB_K; ++i) {
          any_a |= st[i] == SB_A;
          if (st[i] == SB_U) {
            kept |= ((n4 >> (16 * i)) & 0xFFFFull) << (16u * nk);
            ++nk;
          }
        }
        if (any_a) {
          dec = SB_R;
        } else if (cnt <= (uint32_t)SB_K) {
          if (nk == 0u) {
            dec = SB_A;
          } else if (nk != cnt) {
            *slot = kept;
            cnt1[j] = (uint8_t)nk;
          }
        }
      }
      if (dec != SB_U) {
        sb_lds_set(l.st, j, dec);
        const uint32_t gi = k.first + j;
        atomicOr(&l.pub[(gi >> 4) - w0], dec << ((gi & 15u) * 2u));
      }
    }
    SB_T(4);
    // decisions: passes over the wavefront's undecided points until every one is final.  A pass reads the LDS states (other
    // wavefronts' decisions included); when it decides nothing the next one also looks up the halo points that were
    // undecided when they were staged; what is new is published after every pass that decided something.
    uint64_t t_idle = 0;
    uint32_t idle = 0;
    bool gave_up = false, poll = false, to_publish = true;
    for (;;) {
      bool remaining = false, progress = false, halo_wait = false;
      for (uint32_t base = ja; base < jb; base += 64u) {
        const uint32_t j = base + lane;
        const bool have = j < jb;
        uint32_t cur = have ? sb_lds_state(l.st, j) : (uint32_t)SB_R;  // this lane's state as the other lanes see it
        const bool und = have && cur == SB_U;
        if (!__ballot(und)) continue;
        // the recorded neighbours: those among the 64 points of this round are read from the lanes' registers (shuffles,
        // below), the others -- earlier rounds, other wavefronts, the halo -- from the LDS states, once per pass
        static_assert(SB_K == 4, "four two-byte entries = one eight-byte load");
        const uint32_t cnt = und ? cnt1[j] : 0u;
        const uint64_t n4 = und ? *reinterpret_cast<const uint64_t*>(l.nbr + (size_t)j * SB_K) : 0ull;
        const uint32_t c4 = min(cnt, (uint32_t)SB_K);
        uint32_t q[SB_K], st[SB_K];
        uint32_t inround = 0;
#pragma unroll
        for (int i = 0; i < SB_K; ++i) {
          q[i] = (uint32_t)(n4 >> (16 * i)) & 0xFFFFu;
          const bool used = (uint32_t)i < c4;
          // (points of THIS round only: the halo is staged right behind the own points, and a halo index below base + 64, read
          // from a lane that holds no point, once looked rejected -- 17 wrong decisions in 400 000, found with smaller blocks)
          const bool here = used && q[i] >= base && q[i] < min(base + 64u, jb);
          inround |= here ? 1u << i : 0u;
          st[i] = used && !here ? sb_lds_state(l.st, q[i]) : (uint32_t)SB_R;
        }
        bool ext_a = false, ext_u = false;
#pragma unroll
        for (int i = 0; i < SB_K; ++i) {
          uint32_t v = st[i];
          if (poll && v == SB_U && q[i] >= k.n_own) v = sb_state_of(a, l, k, q[i], true);
          ext_a |= v == SB_A;
          ext_u |= v == SB_U;
          halo_wait |= v == SB_U && q[i] >= k.n_own;
        }
        // the chains inside the round, without memory: every lane shows its state, every undecided lane looks
        const bool listed = und && cnt <= (uint32_t)SB_K;
        bool in_u_last = false;
        for (;;) {
          bool in_a = false, in_u = false;
#pragma unroll
          for (int i = 0; i < SB_K; ++i) {
            const uint32_t sv = (uint32_t)__shfl((int)cur, (int)((q[i] - base) & 63u), WAVE);
            if ((inround >> i) & 1u) {
              in_a |= sv == SB_A;
              in_u |= sv == SB_U;
            }
          }
          in_u_last = in_u;
          bool changed = false;
          if (und && cur == SB_U) {
            if (ext_a || in_a) {
              cur = SB_R;
              changed = true;
            } else if (listed && !ext_u && !in_u) {
              cur = SB_A;
              changed = true;
            }
          }
          if (!__ballot(changed)) break;
        }
        if (und && cur == SB_U && cnt > (uint32_t)SB_K && !ext_u && !in_u_last) {
          // more neighbours than were recorded, and the recorded ones have all been rejected: search again (rare)
          bool any_a = false, any_u = false;
          ++my_research;
          sb_visit<WIDE>(a, l, k, j, [&](uint32_t qq) {
            const uint32_t sv = sb_state_of(a, l, k, qq, true);
            any_a |= sv == SB_A;
            any_u |= sv == SB_U;
            return !any_a;
          });
          if (any_a) cur = SB_R; else if (!any_u) cur = SB_A;
        }
        if (und && cur != SB_U) {
          sb_lds_set(l.st, j, cur);
          const uint32_t gi = k.first + j;
          atomicOr(&l.pub[(gi >> 4) - w0], cur << ((gi & 15u) * 2u));
          progress = true;
        } else if (und) {
          remaining = true;
        }
      }
      const bool any_progress = __ballot(progress) != 0, any_remaining = __ballot(remaining) != 0;
      if (lane == 0) ++my_iters;
      to_publish |= any_progress;
      if (any_progress && any_remaining && !poll) continue;  // the chains inside the wavefront's own points first
      if (to_publish) {
        for (uint32_t w = wa + lane; w < wb; w += 64u) {
          const uint32_t v = atomicExch(&l.pub[w], 0u);
          if (v) __hip_atomic_fetch_or(a.st2 + w0 + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        to_publish = false;
      }
      if (!any_remaining) break;
      if (any_progress) {
        idle = 0;
        t_idle = 0;
        poll = false;
        continue;
      }
      // nothing decided in this pass: the wavefront waits for other wavefronts of its workgroup (their decisions show up in
      // the LDS states) or for earlier blocks (the next pass asks the state array: a global round trip)
      poll = __ballot(halo_wait) != 0;
      ++idle;
      if (lane == 0) ++my_waits;
      if (idle > 1u) __builtin_amdgcn_s_sleep(4);
      if ((idle & 15u) == 0u) {
        uint32_t stop = 0;
        if (lane == 0) {
          const uint64_t now = wall_clock64();
          if (t_idle == 0) t_idle = now;
          if (sb_load_word(words + SBW_ABORT) != SB_ABORT_NONE || __hip_atomic_load(&l.misc[SBM_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            stop = 1;
          } else if (now - t_idle > a.timeout_ticks) {
            __hip_atomic_fetch_max(words + SBW_ABORT, (uint32_t)SB_ABORT_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stop = 1;
          }
        }
        if (__shfl((int)stop, 0, WAVE)) {
          gave_up = true;
          break;
        }
      }
    }
    SB_T(5);
    if (gave_up) {
      if (lane == 0) __hip_atomic_store(&l.misc[SBM_ABORT], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      // the level's output: one byte per taken point (the caller has cleared the array)
      for (uint32_t base = ja; base < jb; base += 64u) {
        const uint32_t j = base + lane;
        if (j < jb && sb_lds_state(l.st, j) == SB_A) a.taken[k.first + j] = 1;
      }
    }
    if (tid == 0) ++my_blocks;
    __syncthreads();
    SB_T(6);
  }
#ifdef SWZ_SB_STATS
  if (tid == 0)
    for (int i = 0; i < 7; ++i) atomicAdd(words + SBW_T0 + i, tacc[i] >> 4);
#endif
  if (lane == 0) {
    if (my_blocks) atomicAdd(words + SBW_BLOCKS, my_blocks);
    if (my_iters) atomicAdd(words + SBW_ITER, my_iters);
    if (my_waits) atomicAdd(words + SBW_WAITS, my_waits);
    if (my_steps) atomicAdd(words + SBW_STEPS, my_steps);
  }
  if (my_research) atomicAdd(words + SBW_RESEARCH, my_research);
}

// ----------------------------------------------------------------------------- the granule table
struct SbTabArgs {
  const uint64_t* akey;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* snode_of;
  const uint32_t* sn_direct;  // the sampled node of every point (a compacted subset: nid / nmode / snode_of are not looked at)
  uint32_t all_sampled;
  uint32_t m;
  uint32_t gran_shift;       // key >> gran_shift = node prefix + granule code
  uint64_t gran_per_node;
  uint2* gtab;
};
__device__ __forceinline__ uint64_t sb_granule_of(const SbTabArgs& t, uint32_t i) {  // ~0: the point's node is not sampled
  uint32_t sn;
  if (t.sn_direct) {
    sn = t.sn_direct[i];
  } else {
    const uint32_t node = t.nid[i];
    if (!t.all_sampled && t.nmode[node] != MODE_SAMPLE) return ~0ull;
    sn = t.all_sampled ? node : t.snode_of[node];
  }
  return (uint64_t)sn * t.gran_per_node + ((t.akey[i] >> t.gran_shift) & (t.gran_per_node - 1ull));
}
__global__ __launch_bounds__(256) void sb_table_kernel(SbTabArgs t) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= t.m) return;
  const uint64_t g = sb_granule_of(t, i);
  if (g == ~0ull) return;
  const bool head = i == 0 || sb_granule_of(t, i - 1) != g;
  const bool tail = i + 1 == t.m || sb_granule_of(t, i + 1) != g;
  if (head) t.gtab[g].x = i;
  if (tail) t.gtab[g].y = i + 1u;
}

// What one run of the block kernel works on: the level's active points, or the subset of them that new points can change.
struct SbInput {
  const uint64_t* akey = nullptr;
  uint32_t m = 0;
  const uint32_t* aidx = nullptr;       // sorted position -> working index (null: identity)
  const uint32_t* sn_direct = nullptr;  // see SbTabArgs
  uint8_t* taken = nullptr;             // m bytes, cleared
  uint32_t points = 0;                  // of sampled nodes
  const uint32_t* occupied = nullptr;   // [12]: occupied cells per cell level
  double spread = 1.0;                  // how much wider than Poisson a block's population scatters around the mean
  const char* what = "";
};

// *done = false: the level does not qualify or a block did not fit -- the caller goes on with another path.
static int sb_run(swz_ctx* c, const LevelPlan& plan, const SortedPoints& sp, const LevelBuffers& lb, const uint32_t* snode_of,
                  bool all_sampled, uint32_t sample_nodes, const KeyMetric& km, const SbInput& in, bool* done) {
  *done = false;
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  // cells: as fine as the spacing allows while a block of 8^3 of them still holds a workgroup's worth of points
  int cl = plan.cell_levels_geo;
  double min_block = 128.0;
  if (const char* e = c->opt("SWZ_SP_BLOCK_MIN")) min_block = atof(e);
  auto per_block = [&](int l) { return (double)in.points / (double)std::max(1u, in.occupied[std::max(0, l - 3)]); };
  while (cl > 3 && per_block(cl) < min_block) --cl;
  bool cl_forced = false, caps_forced = false;
  if (const char* e = c->opt("SWZ_SP_BLOCK_CL")) {
    cl = std::min(plan.cell_levels_geo, atoi(e));
    cl_forced = true;
  }
  if (cl < 3 || cl > 10) return SWZ_OK;
  const uint32_t limit = (uint32_t)(SB_THREADS * SB_PMAX);  // what a workgroup stages at once
  // LDS capacity from the expected population (an estimate: a block that does not fit stops the launch, see below).
  // Uniform data: a block's population is Poisson -- 416 +- 20 at level 2 of the 1 B run --, its halo's, counted in whole
  // granules, about 1.2 times that.
  auto estimate = [&](int l, uint32_t* own, uint32_t* halo) {
    const double expect = per_block(l), dev = std::sqrt(expect * in.spread);
    *own = (uint32_t)std::max(128.0, 32.0 * std::ceil((expect + (in.spread > 2.0 ? 6.0 : 8.0) * dev + 32.0) / 32.0));
    *halo = (uint32_t)(32.0 * std::ceil((1.25 * expect + (in.spread > 2.0 ? 7.0 : 10.0) * std::sqrt(1.25) * dev + 64.0) / 32.0));
  };
  // ... and what the estimate leaves of the LDS a workgroup gets anyway (six per CU by registers) is handed out too: the
  // batches of a tiler that arrive as tiles have blocks half inside their tile, which pull the mean below what a full
  // block holds (a level of 14.5 M points with 336 per block: capacities 544 / 704 for blocks of up to 550 / 746)
  auto fill_free_lds = [&](uint32_t* own, uint32_t* halo) {
    const size_t free_lds = 26u * 1024u;  // (six of them, each rounded up to the allocation granule, fit the CU's 160 KB)
    const uint32_t own_max = *own + *own / 2u, halo_max = *halo + *halo / 2u;
    for (bool grew = true; grew;) {
      grew = false;
      if (*halo + 32u <= halo_max && *own + *halo + 32u <= limit && sb_lds_bytes(*own, *halo + 32u) <= free_lds) *halo += 32u, grew = true;
      if (*own + 32u <= own_max && *own + 32u < limit / 2u && *own + *halo + 32u <= limit && sb_lds_bytes(*own + 32u, *halo) <= free_lds)
        *own += 32u, grew = true;
    }
  };
  uint32_t own_cap = 0, halo_cap = 0;
  estimate(cl, &own_cap, &halo_cap);
  fill_free_lds(&own_cap, &halo_cap);
  if (const char* e = c->opt("SWZ_SP_BLOCK_CAP_SCALE")) {  // tests: an estimate that is too small, so that launches are repeated
    own_cap = std::max(32u, (uint32_t)(own_cap * atof(e)) / 32u * 32u);
    halo_cap = std::max(32u, (uint32_t)(halo_cap * atof(e)) / 32u * 32u);
  }
  if (const char* e = c->opt("SWZ_SP_BLOCK_OWN")) {
    own_cap = (uint32_t)std::max(64, atoi(e)) / 16u * 16u;
    caps_forced = true;
  }
  if (const char* e = c->opt("SWZ_SP_BLOCK_HALO")) {
    halo_cap = (uint32_t)std::max(64, atoi(e)) / 16u * 16u;
    caps_forced = true;
  }
  const uint32_t node_shift = plan.node_shift == 63u ? 63u : plan.node_shift;
  SbTabArgs t{};
  t.akey = in.akey;
  t.nid = lb.nid;
  t.nmode = lb.nmode;
  t.snode_of = snode_of;
  t.sn_direct = in.sn_direct;
  t.all_sampled = all_sampled ? 1u : 0u;
  t.m = in.m;
  SbArgs a{};
  a.akey = in.akey;
  a.m = in.m;
  SWZ_TRY(c->get("sb_state", (size_t)in.m / 16 + 2, &a.st2));
  a.taken = in.taken;
  a.aidx = in.aidx;
  a.perm = sp.perm;
  a.xyz = sp.xyz;
  a.f_lo = km.f_lo;
  a.f_hi = km.f_hi;
  if (const char* e = c->opt("SWZ_SP_FILTER_EPS"))  // tests: 1e30 sends every compare within reach to the exact path
    if (atof(e) >= 0.5) {
      a.f_lo = 0.f;
      a.f_hi = INFINITY;
    }
  a.i_lo = a.f_lo >= 4294967040.f ? 0xFFFFFFFFu : (uint32_t)std::floor(a.f_lo);
  a.i_hi = a.f_hi >= 4294967040.f ? 0xFFFFFFFFu : (uint32_t)std::ceil(a.f_hi);
  a.sq_spacing = plan.sq_spacing;
  a.ns = sample_nodes;
  SWZ_TRY(c->get("sb_counters", (size_t)(SB_NC * SB_CTR_STRIDE + SBW_COUNT), &a.ctr));
  double timeout_s = 10.0;
  if (const char* e = c->opt("SWZ_SP_BLOCK_TIMEOUT_MS")) timeout_s = atof(e) * 1e-3;
  a.timeout_ticks = (uint64_t)(timeout_s * 1e8);
  if (const char* e = c->opt("SWZ_SP_BLOCK_DBG")) a.dbg = (uint32_t)atoi(e);
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);

  // A block or a halo beyond the capacity stops the launch (every decision taken so far is exact and simply taken again).
  // The launch is then repeated with what the stopped blocks needed plus a quarter or, when that is more than a workgroup
  // stages, with cells half the size; after four launches the level goes to the caller's other paths.
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (own_cap + halo_cap > limit) {
      if (own_cap >= limit / 2u) return SWZ_OK;
      halo_cap = limit - own_cap;
    }
    const size_t lds = sb_lds_bytes(own_cap, halo_cap);
    if (lds > 160u * 1024u) return SWZ_OK;
    const uint64_t gran_per_node = 1ull << (3 * (cl - 1));
    const uint64_t entries = (uint64_t)sample_nodes * gran_per_node;
    if (entries > (1ull << 30)) return SWZ_OK;  // 8 GB of table
    t.gran_shift = node_shift - 3u * (uint32_t)(cl - 1);
    t.gran_per_node = gran_per_node;
    SWZ_TRY(c->get("sb_gtab", (size_t)entries, &t.gtab));
    a.gtab = t.gtab;
    a.cell_bits = (node_shift - 3u * (uint32_t)cl) / 3u;
    a.cl = (uint32_t)cl;
    a.own_cap = own_cap;
    a.halo_cap = halo_cap;

    ProfScope ps(c, "sample_min_distance", (uint64_t)in.points * 33ull, 1);
    SWZ_HIP(c, memset_large(t.gtab, 0, (size_t)entries * sizeof(uint2), c->stream));
    SWZ_HIP(c, hipMemsetAsync(a.st2, 0, ((size_t)in.m / 16 + 2) * sizeof(uint32_t), c->stream));
    SWZ_HIP(c, hipMemsetAsync(a.ctr, 0, (size_t)(SB_NC * SB_CTR_STRIDE + SBW_COUNT) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(sb_table_kernel, dim3(div_up(in.m, 256)), dim3(256), 0, c->stream, t);
    SWZ_LAUNCH_CHECK(c);
    // (every time: the attribute belongs to the function on the CURRENT device, and a process may drive several)
    bool wide = a.cell_bits > 12u;  // a region of ten cells must fit sixteen bits
    if (const char* e = c->opt("SWZ_SP_BLOCK_WIDE")) wide = wide || atoi(e) != 0;
    SWZ_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(wide ? sb_block_kernel<true> : sb_block_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / lds));
    if (const char* e = c->opt("SWZ_SP_BLOCK_PER_CU")) per_cu = (uint32_t)std::max(1, atoi(e));
    const uint64_t blocks_total = (uint64_t)sample_nodes << (3 * (cl - 3));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)cus * per_cu, std::max<uint64_t>(1, blocks_total));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (dbg) {
      ev0 = c->take_event();
      ev1 = c->take_event();
      (void)hipEventRecord(ev0, c->stream);
    }
    if (wide)
      hipLaunchKernelGGL(sb_block_kernel<true>, dim3(grid), dim3(SB_THREADS), lds, c->stream, a);
    else
      hipLaunchKernelGGL(sb_block_kernel<false>, dim3(grid), dim3(SB_THREADS), lds, c->stream, a);
    SWZ_LAUNCH_CHECK(c);
    if (dbg) (void)hipEventRecord(ev1, c->stream);
    uint32_t h[SBW_COUNT] = {0};
    SWZ_HIP(c, hipMemcpyAsync(h, a.ctr + SB_NC * SB_CTR_STRIDE, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
    if (dbg) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev0, ev1);
      c->event_pool.insert(c->event_pool.end(), {ev0, ev1});
      fprintf(stderr,
              "[swz] MIN_DISTANCE level %d block path%s: %u pts in %u nodes, cell_levels %d (geo %d), %.0f pts/block, caps %u / %u, LDS %zu B, "
              "grid %u, %u blocks, %u search steps, %u passes, %u idle passes, %u searched again, abort %u (max %u / %u), %.2f ms\n",
              plan.level, in.what, in.points, sample_nodes, cl, plan.cell_levels_geo, per_block(cl), own_cap, halo_cap, lds, grid, h[SBW_BLOCKS],
              h[SBW_STEPS], h[SBW_ITER], h[SBW_WAITS], h[SBW_RESEARCH], h[SBW_ABORT], h[SBW_MAX_OWN], h[SBW_MAX_HALO], ms);
    }
#ifdef SWZ_SB_STATS
    if (dbg && h[SBW_BLOCKS])
      fprintf(stderr, "[swz]   thread 0, us per block: ticket %.2f, granules %.2f, stage %.2f, index %.2f, search %.2f, decide %.2f, tail %.2f\n",
              h[SBW_T0] * 0.16 / h[SBW_BLOCKS], h[SBW_T0 + 1] * 0.16 / h[SBW_BLOCKS], h[SBW_T0 + 2] * 0.16 / h[SBW_BLOCKS], h[SBW_T0 + 3] * 0.16 / h[SBW_BLOCKS],
              h[SBW_T0 + 4] * 0.16 / h[SBW_BLOCKS], h[SBW_T0 + 5] * 0.16 / h[SBW_BLOCKS], h[SBW_T0 + 6] * 0.16 / h[SBW_BLOCKS]);
#endif
    if (h[SBW_ABORT] == SB_ABORT_TIMEOUT)
      return c->fail(SWZ_ERR_INTERNAL, "MIN_DISTANCE block path: a workgroup waited longer than the time-out for an earlier block");
    if (h[SBW_ABORT] == SB_ABORT_NONE) {
      *done = true;
      return SWZ_OK;
    }
    if (caps_forced) return SWZ_OK;  // (tests)
    const uint32_t need_own = std::max(own_cap, (h[SBW_MAX_OWN] + h[SBW_MAX_OWN] / 4u + 63u) / 32u * 32u);
    const uint32_t need_halo = std::max(halo_cap, (h[SBW_MAX_HALO] + h[SBW_MAX_HALO] / 4u + 95u) / 32u * 32u);
    if (need_own < limit / 2u && need_own + need_halo <= limit) {
      own_cap = need_own;
      halo_cap = need_halo;
    } else if (cl < plan.cell_levels_geo && !cl_forced && per_block(cl + 1) >= 0.5 * min_block) {
      // cells half the size, capacities from the estimate again (what was seen says nothing about an eighth of the volume).
      // Only while a block still holds half a workgroup's worth of points: a level that is dense in a few places and thin
      // everywhere else (the blob of the surface-like test cloud: 323 points per block on average, 4 238 in the largest)
      // would end up with blocks of 57 points and capacities for 500 -- 21 ms for a level the thread-per-point path takes
      // in 5.
      ++cl;
      estimate(cl, &own_cap, &halo_cap);
      fill_free_lds(&own_cap, &halo_cap);
    } else {
      return SWZ_OK;
    }
  }
  return SWZ_OK;  // the caller's other paths take the level
}

// ----------------------------------------------------------------------------- multi-batch tiling: only what new points can change
// A batch of a multi-batch tiling (swz_tiler.hip) merges its points with the files earlier batches left in the nodes it
// touches and samples the union (tile_node, TilingAlgorithms.cpp:421-442).  The entries of a file that holds more than
// max_points points are what THIS sampler accepted at THIS spacing in an earlier batch -- a node takes everything only
// while it holds no file and at most max_points points (TilingAlgorithms.cpp:272-275, Sampling.h:201-208) --, so any two
// of them are at least one spacing apart: they never reject one another.  The Morton-order greedy over the union therefore
// decides an old point o exactly as the greedy over  S = {new points} + {old points within one spacing of a new point}
// does (o's only possible rejectors are new points; a new point's are new points and old points of S), and old points
// outside S are accepted.  S is found on the level's finest cells (one spacing wide or more): a new point marks its cell,
// an old point belongs to S when one of the 27 cells around it is marked.  With 10 M new points per batch on 390 M points of
// files (level 2 of the 1 B run in 100 batches) S is a quarter of the level, on level 3 a thirtieth.
// "New" also covers the entries of files with at most max_points points of nodes without a child one level down: such a file
// may be everything the node has ever seen (a take-all).  A node that has a child has handed points down, so it has been sampled,
// and every visit since has sampled it again: its file is an output of the sampler whatever its size.
struct SbiArgs {
  const uint64_t* akey;
  const uint32_t* aidx;
  const uint32_t* nid;
  const uint8_t* nmode;
  const uint32_t* snode_of;
  const uint8_t* indep;  // per node: its file's entries are pairwise a spacing apart
  uint32_t all_sampled;
  uint32_t m;
  uint32_t old_lo, old_n;
  uint32_t cell_shift;   // key >> cell_shift: node prefix + cell code
  uint32_t cell_mask;    // 8^cl - 1
  uint32_t cl;
  uint32_t* bits;        // one per (sampled node, cell)
  uint8_t* sel;          // per point: 1 = belongs to S
  uint8_t* taken;
};
__global__ __launch_bounds__(256) void sbi_indep_kernel(const uint32_t* __restrict__ nstart, const uint8_t* __restrict__ nmode,
                                                        uint32_t nnodes, const uint64_t* __restrict__ akey, uint32_t nsh,
                                                        const uint64_t* __restrict__ ckey, uint32_t nc, uint64_t max_points,
                                                        const uint64_t* __restrict__ child_nkey, uint32_t child_nn,
                                                        uint8_t* __restrict__ indep) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= nnodes) return;
  uint8_t r = 0;
  if (nmode[j] == MODE_SAMPLE) {
    const uint64_t prefix = akey[nstart[j]] >> nsh;
    uint32_t lo = 0, hi = nc;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if ((ckey[mid] >> nsh) < prefix) lo = mid + 1; else hi = mid;
    }
    const uint32_t first = lo;
    hi = nc;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if ((ckey[mid] >> nsh) <= prefix) lo = mid + 1; else hi = mid;
    }
    const uint32_t cached = lo - first;
    r = (uint64_t)cached > max_points ? 1 : 0;
    if (!r && cached && child_nn) {
      // a smaller file: the node's own output all the same once it has handed points down (a take-all keeps everything)
      const uint64_t from = prefix << nsh;
      lo = 0;
      hi = child_nn;
      while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (child_nkey[mid] < from) lo = mid + 1; else hi = mid;
      }
      r = lo < child_nn && (child_nkey[lo] >> nsh) == prefix ? 1 : 0;
    }
  }
  indep[j] = r;
}
__device__ __forceinline__ bool sbi_sampled(const SbiArgs& a, uint32_t i, uint32_t* node) {
  *node = a.nid[i];
  return a.all_sampled || a.nmode[*node] == MODE_SAMPLE;
}
__device__ __forceinline__ bool sbi_is_old(const SbiArgs& a, uint32_t i, uint32_t node) {
  return a.aidx[i] - a.old_lo < a.old_n && a.indep[node];
}
__device__ __forceinline__ uint64_t sbi_cell_base(const SbiArgs& a, uint32_t node) {
  return (uint64_t)(a.all_sampled ? node : a.snode_of[node]) << (3u * a.cl);
}
// One bit per cell that holds a new point, set by the batch's own points (their keys as they were before the merge with the
// files: a hundredth of the level).  The node of a key: the last one whose first point's prefix is not above it.
// (Marking the 27 cells around every new point instead and testing one bit per old point was the first version: 270 M
// scattered device-scope atomics per level took 30-54 ms.)
template <bool CLEAR>  // (true: the words that were marked go back to zero, so that the array is never cleared as a whole)
__global__ __launch_bounds__(256) void sbi_mark_kernel(SbiArgs a, const uint64_t* __restrict__ new_key, uint32_t new_m,
                                                       const uint32_t* __restrict__ nstart, uint32_t nnodes, uint32_t nsh) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= new_m) return;
  const uint64_t key = new_key[i], prefix = key >> nsh;
  uint32_t lo = 0, hi = nnodes;  // first node with a prefix above this one
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2u;
    if ((a.akey[nstart[mid]] >> nsh) <= prefix) lo = mid + 1u; else hi = mid;
  }
  if (lo == 0) return;
  const uint32_t node = lo - 1u;
  // (a node whose file may hold close pairs is sampled as a whole: nobody looks at its marks)
  if ((a.akey[nstart[node]] >> nsh) != prefix || !(a.all_sampled || a.nmode[node] == MODE_SAMPLE) || !a.indep[node]) return;
  const uint64_t bit = sbi_cell_base(a, node) + ((uint32_t)(key >> a.cell_shift) & a.cell_mask);
  const uint32_t b = 1u << (uint32_t)(bit & 31ull);
  uint32_t* w = a.bits + (bit >> 5);
  if (CLEAR) {
    *w = 0u;
  } else if (!(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & b)) {
    atomicOr(w, b);
  }
}
// Is there a new point in the 27 cells around this one?  The marks of an aligned group of 2 x 2 x 2 cells are one byte (the
// cell code is a Morton code: bit 0 x, bit 1 y, bit 2 z), the 27 cells lie in two groups per axis -- the cell's own, whole,
// and the one before it (its upper half) or behind it (its lower half): eight byte loads instead of 27 bit tests.  The
// groups' coordinates stay interleaved: +-1 on one axis is an add / subtract on that axis' bits alone.
__device__ __forceinline__ bool sbi_near_new(const SbiArgs& a, uint64_t base, uint32_t code) {
  const uint32_t gmask = a.cell_mask >> 3;
  const uint32_t MX = 0x09249249u & gmask, MY = 0x12492492u & gmask, MZ = 0x24924924u & gmask;
  const uint32_t g = code >> 3, x = g & MX, y = g & MY, z = g & MZ;
  const uint8_t* marks = reinterpret_cast<const uint8_t*>(a.bits) + (base >> 3);
  // per axis: the own group with every cell, the other one with the half that touches (nothing beyond the node's faces)
  uint32_t gx[2] = {x, x}, gy[2] = {y, y}, gz[2] = {z, z};
  uint32_t mx[2] = {0xFFu, 0u}, my[2] = {0xFFu, 0u}, mz[2] = {0xFFu, 0u};
  if (code & 1u) { if (x != MX) gx[1] = ((x | ~MX) + 1u) & MX, mx[1] = 0x55u; } else { if (x) gx[1] = (x - 1u) & MX, mx[1] = 0xAAu; }
  if (code & 2u) { if (y != MY) gy[1] = ((y | ~MY) + 1u) & MY, my[1] = 0x33u; } else { if (y) gy[1] = (y - 1u) & MY, my[1] = 0xCCu; }
  if (code & 4u) { if (z != MZ) gz[1] = ((z | ~MZ) + 1u) & MZ, mz[1] = 0x0Fu; } else { if (z) gz[1] = (z - 1u) & MZ, mz[1] = 0xF0u; }
  uint32_t any = 0;
#pragma unroll
  for (int iz = 0; iz < 2; ++iz)
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
      for (int ix = 0; ix < 2; ++ix) any |= (uint32_t)marks[gx[ix] | gy[iy] | gz[iz]] & mx[ix] & my[iy] & mz[iz];
  return any != 0;
}
__global__ __launch_bounds__(256) void sbi_select_kernel(SbiArgs a) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.m) return;
  uint32_t node;
  uint8_t s = 0;
  if (sbi_sampled(a, i, &node)) {
    s = 1;
    if (sbi_is_old(a, i, node)) {
      s = sbi_near_new(a, sbi_cell_base(a, node), (uint32_t)(a.akey[i] >> a.cell_shift) & a.cell_mask) ? 1 : 0;
      if (!s) a.taken[i] = 1;  // no new point within a spacing: nothing can reject it
    }
  }
  a.sel[i] = s;
}
struct SbiSelF {
  const uint8_t* sel;
  __device__ uint32_t operator()(uint32_t i) const { return sel[i]; }
};
struct SbiPackG {
  const uint64_t* akey;
  const uint32_t* aidx;
  const uint32_t* nid;
  const uint32_t* snode_of;
  uint32_t all_sampled;
  uint64_t* skey;
  uint32_t* saidx;
  uint32_t* ssn;
  uint32_t* sidx;
  __device__ void operator()(uint32_t i, uint32_t excl, uint32_t v) const {
    if (!v) return;
    skey[excl] = akey[i];
    saidx[excl] = aidx[i];
    const uint32_t node = nid[i];
    ssn[excl] = all_sampled ? node : snode_of[node];
    sidx[excl] = i;
  }
};
// occupied cells of the subset per cell level (as md_cell_hist_kernel counts them for the whole level): hist[0] counts the
// firsts of the nodes, hist[b] the points whose first digit below the node prefix that differs from their predecessor's is b
__global__ __launch_bounds__(256) void sbi_hist_kernel(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ ssn, uint32_t n,
                                                       uint32_t node_shift, uint32_t cl_geo, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t mine = 0;  // lane b accumulates the wavefront's count of bin b
  for (uint64_t j0 = (uint64_t)blockIdx.x * 256u; j0 < n; j0 += (uint64_t)gridDim.x * 256u) {
    const uint32_t j = (uint32_t)j0 + threadIdx.x;
    uint32_t bin = 0xFFu;
    if (j < n) {
      if (j == 0 || ssn[j - 1] != ssn[j]) {
        bin = 0;
      } else if (cl_geo) {
        const uint64_t diff = ((skey[j] ^ skey[j - 1]) >> (node_shift - 3u * cl_geo)) & ((1ull << (3u * cl_geo)) - 1ull);
        if (diff) bin = cl_geo - (uint32_t)(63 - __clzll((unsigned long long)diff)) / 3u;  // 1 .. cl_geo
      }
    }
    for (uint32_t b = 0; b <= cl_geo; ++b) {
      const uint32_t cnt = (uint32_t)__popcll(__ballot(bin == b));
      if ((threadIdx.x & 63u) == b) mine += cnt;
    }
  }
  if ((threadIdx.x & 63u) <= cl_geo && mine) atomicAdd(&h[threadIdx.x & 63u], mine);
  __syncthreads();
  if (threadIdx.x < 12 && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void sbi_scatter_kernel(const uint8_t* __restrict__ staken, const uint32_t* __restrict__ sidx, uint32_t n,
                                                          uint8_t* __restrict__ taken) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < n && staken[j]) taken[sidx[j]] = 1;
}

// *done = false: not worth it (or a block of the subset did not fit): the whole level is sampled as usual.  Old points that
// no new point can reach are marked taken either way -- they are in the exact result.
static int sb_incremental(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                          const uint32_t* snode_of, bool all_sampled, uint32_t num_nodes, uint32_t sample_nodes, uint32_t sample_points,
                          const KeyMetric& km, bool* done) {
  *done = false;
  const bool dbg = c->opt("SWZ_DEBUG") != nullptr;
  int cl = std::min(plan.cell_levels_geo, 10);
  while (cl > 0 && ((uint64_t)sample_nodes << (3 * cl)) > (1ull << 34)) --cl;  // 2 GB of marks at most; coarser cells mark more
  if (cl < 1) return SWZ_OK;
  const uint64_t nbits = (uint64_t)sample_nodes << (3 * cl);
  const size_t words = (size_t)((nbits + 31ull) / 32ull);
  const uint32_t node_shift = plan.node_shift == 63u ? 63u : plan.node_shift;
  SbiArgs a{};
  a.akey = as.akey;
  a.aidx = as.aidx;
  a.nid = lb.nid;
  a.nmode = lb.nmode;
  a.snode_of = snode_of;
  a.all_sampled = all_sampled ? 1u : 0u;
  a.m = as.m;
  a.old_lo = as.old_lo;
  a.old_n = as.old_hi - as.old_lo;
  a.cell_shift = node_shift - 3u * (uint32_t)cl;
  a.cell_mask = (uint32_t)((1ull << (3 * cl)) - 1ull);
  a.cl = (uint32_t)cl;
  a.taken = lb.taken;
  double worth = 0.35;  // of the level's points: above it the subset's own passes and its clumpier blocks cost what they save
  if (const char* e = c->opt("SWZ_SP_INCREMENTAL_MAX")) worth = atof(e);
  if ((double)as.new_m > worth * (double)sample_points) return SWZ_OK;  // (the new points alone are more than that)
  uint8_t* indep = nullptr;
  SWZ_TRY(c->get("sbi_indep", (size_t)num_nodes, &indep));
  SWZ_TRY(c->get("sbi_bits", words, &a.bits));
  SWZ_TRY(c->get("sbi_sel", (size_t)as.m, &a.sel));
  a.indep = indep;
  uint32_t* d_cnt = nullptr;  // [0]: |S|, [1..12]: the subset's cell histogram
  SWZ_TRY(c->get("sbi_counts", (size_t)16, &d_cnt));
  uint32_t total = 0;
  uint32_t* d_partial = nullptr;
  {
    ProfScope ps(c, "sample_min_distance", (uint64_t)as.m * 30ull, 1);
    hipLaunchKernelGGL(sbi_indep_kernel, dim3(div_up(num_nodes, 256)), dim3(256), 0, c->stream, lb.nstart, lb.nmode, num_nodes, as.akey,
                       node_shift, as.ckey, as.nc, plan.max_points, as.child_nkey, as.child_nn, indep);
    SWZ_LAUNCH_CHECK(c);
    // The marks (up to 2 GB) are cleared as a whole only when the array is new or a call before this one did not get to take
    // its marks back: afterwards the batch's points clear the words they set (a memset of 1 GB per level and batch is 0.35 ms).
    if (c->sbi_clean_ptr != a.bits || c->sbi_clean_bytes < words * sizeof(uint32_t)) {
      SWZ_HIP(c, memset_large(a.bits, 0, words * sizeof(uint32_t), c->stream));
      c->sbi_clean_bytes = words * sizeof(uint32_t);
    }
    c->sbi_clean_ptr = nullptr;  // (until the marks have been taken back)
    SWZ_HIP(c, hipMemsetAsync(d_cnt, 0, 16 * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(sbi_mark_kernel<false>, dim3(div_up(as.new_m, 256)), dim3(256), 0, c->stream, a, as.new_key, as.new_m, lb.nstart,
                       num_nodes, node_shift);
    SWZ_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(sbi_select_kernel, dim3(div_up(as.m, 256)), dim3(256), 0, c->stream, a);
    SWZ_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(sbi_mark_kernel<true>, dim3(div_up(as.new_m, 256)), dim3(256), 0, c->stream, a, as.new_key, as.new_m, lb.nstart,
                       num_nodes, node_shift);
    SWZ_LAUNCH_CHECK(c);
    c->sbi_clean_ptr = a.bits;
    SWZ_TRY(fused_scan_sums(c, SbiSelF{a.sel}, as.m, d_cnt, "sbi", &d_partial));
    SWZ_HIP(c, hipMemcpyAsync(&total, d_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (dbg)
    fprintf(stderr, "[swz] MIN_DISTANCE level %d: %u of %u points are files of earlier batches, %u of %u can change (marks at cell level %d)\n",
            plan.level, as.nc, as.m, total, sample_points, cl);
  if (total == 0) {
    *done = true;
    return SWZ_OK;
  }
  if ((double)total > worth * (double)sample_points) return SWZ_OK;
  uint64_t* skey = nullptr;
  uint32_t *saidx = nullptr, *ssn = nullptr, *sidx = nullptr;
  uint8_t* staken = nullptr;
  SWZ_TRY(c->get("sbi_key", (size_t)total, &skey));
  SWZ_TRY(c->get("sbi_aidx", (size_t)total, &saidx));
  SWZ_TRY(c->get("sbi_sn", (size_t)total, &ssn));
  SWZ_TRY(c->get("sbi_idx", (size_t)total, &sidx));
  SWZ_TRY(c->get("sbi_taken", (size_t)total, &staken));
  uint32_t hist[12] = {0};
  {
    ProfScope ps(c, "sample_min_distance", (uint64_t)as.m * 17ull + (uint64_t)total * 33ull, 1);
    SWZ_TRY(fused_scan_apply(c, SbiSelF{a.sel}, SbiPackG{as.akey, as.aidx, lb.nid, snode_of, a.all_sampled, skey, saidx, ssn, sidx}, as.m,
                             d_partial));
    SWZ_HIP(c, hipMemsetAsync(staken, 0, (size_t)total, c->stream));
    hipLaunchKernelGGL(sbi_hist_kernel, dim3(std::min<uint32_t>(div_up(total, 256), 2048u)), dim3(256), 0, c->stream, skey, ssn, total, node_shift,
                       (uint32_t)std::min(plan.cell_levels_geo, 11), d_cnt + 1);
    SWZ_LAUNCH_CHECK(c);
    SWZ_HIP(c, hipMemcpyAsync(hist, d_cnt + 1, sizeof(hist), hipMemcpyDeviceToHost, c->stream));
    SWZ_HIP(c, hipStreamSynchronize(c->stream));
  }
  uint32_t occ[12];
  for (int l = 0, s = 0; l < 12; ++l) occ[l] = (uint32_t)(s += (int)hist[l]);
  SbInput in;
  in.akey = skey;
  in.m = total;
  in.aidx = saidx;
  in.sn_direct = ssn;
  in.taken = staken;
  in.points = total;
  in.occupied = occ;
  // (what a new point brings into a block it brings at once: itself and the old points around it -- ten of them when the batch
  // is thin on top of full files, none when it is a tile of its own beside the files of its neighbours)
  in.spread = 1.0 + 9.0 * std::min(1.0, std::max(0.0, ((double)total - (double)as.new_m) / (double)total));
  if (const char* e = c->opt("SWZ_SP_INCREMENTAL_SPREAD")) in.spread = std::max(1.0, atof(e));
  in.what = " (what the new points can change)";
  bool ok = false;
  SWZ_TRY(sb_run(c, plan, sp, lb, snode_of, all_sampled, sample_nodes, km, in, &ok));
  if (!ok) return SWZ_OK;
  hipLaunchKernelGGL(sbi_scatter_kernel, dim3(div_up(total, 256)), dim3(256), 0, c->stream, staken, sidx, total, lb.taken);
  SWZ_LAUNCH_CHECK(c);
  *done = true;
  return SWZ_OK;
}

// *done = false: the level does not qualify or a block did not fit -- the caller goes on with the thread-per-point path.
int min_distance_block_level(swz_ctx* c, const LevelPlan& plan, const ActiveSet& as, const SortedPoints& sp, const LevelBuffers& lb,
                             const uint32_t* snode_of, bool all_sampled, uint32_t num_nodes, uint32_t sample_nodes,
                             uint32_t sample_points, const uint32_t occupied[12], const KeyMetric& km, bool* done) {
  *done = false;
  if (const char* e = c->opt("SWZ_SP_BLOCK"))
    if (atoi(e) == 0) return SWZ_OK;
  if (!km.ok || !sp.xyz || !sp.perm || sp.ghosts) return SWZ_OK;
  if (plan.cell_levels_geo < 3) return SWZ_OK;
  // a batch on top of the files of earlier ones: when those are most of the level, only what the batch can change
  bool inc = as.ckey && as.aidx && as.old_hi > as.old_lo && as.new_key;
  double old_share = 0.5, file_share = 0.5;
  if (const char* e = c->opt("SWZ_SP_INCREMENTAL")) {  // 0: never; a share: whenever the files are that much of the level
    if (atof(e) <= 0.0) inc = false;
    else old_share = std::min(1.0, atof(e)), file_share = 0.0;
  }
  // (... and when the files are big enough for that: one of at most max_points points may come from a node that took everything)
  // (with the node table of the level below a small file can be told from a take-all: then any level of files qualifies)
  if (inc && (double)(as.old_hi - as.old_lo) >= old_share * (double)as.m &&
      (as.child_nn || (double)(as.old_hi - as.old_lo) >= file_share * (double)plan.max_points * (double)num_nodes)) {
    SWZ_TRY(sb_incremental(c, plan, as, sp, lb, snode_of, all_sampled, num_nodes, sample_nodes, sample_points, km, done));
    if (*done) return SWZ_OK;
  }
  SbInput in;
  in.akey = as.akey;
  in.m = as.m;
  in.aidx = as.aidx;
  in.taken = lb.taken;
  in.points = sample_points;
  in.occupied = occupied;
  return sb_run(c, plan, sp, lb, snode_of, all_sampled, sample_nodes, km, in, done);
}

}  // namespace swz
