// swz_device.h -- device-side helpers shared by the kernels (wave64 only: gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace swz {

constexpr int WAVE = 64;

__device__ __forceinline__ uint32_t lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive sum over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
  const uint32_t l = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t t = __shfl_up(v, d, WAVE);
    if (l >= (uint32_t)d) v += t;
  }
  return v;
}

// Exclusive prefix of v over the THREADS threads of the block (THREADS multiple of 64, <= 1024).
// lds must hold THREADS/64 words.  total = block sum (same in every thread).  Ends with a barrier
// so lds may be reused right after.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* lds, uint32_t& total) {
  constexpr int NW = THREADS / WAVE;
  const uint32_t incl = wave_incl_sum(v);
  const uint32_t w = threadIdx.x / WAVE;
  if (lane_id() == WAVE - 1) lds[w] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t s = lds[i];
    if ((uint32_t)i < w) base += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return base + incl - v;
}

// MortonIndex<21> helpers -- core/datastructures/MortonIndex.h:123-145
__device__ __host__ __forceinline__ uint32_t level_shift(int level) { return (uint32_t)(20 - level) * 3u; }

// expand_bits_by_3(uint64_t) -- core/util/stuff.h:207-221: spread 21 bits to every third bit
__device__ __host__ __forceinline__ uint64_t expand_bits_by_3(uint64_t v) {
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x00FF00000000FFFFull;
  v = (v | (v << 16)) & 0x00FF0000FF0000FFull;
  v = (v | (v << 8)) & 0xF00F00F00F00F00Full;
  v = (v | (v << 4)) & 0x30C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
// contract_bits_by_3 -- core/util/stuff.h:223-234
__device__ __host__ __forceinline__ uint64_t contract_bits_by_3(uint64_t v) {
  v &= 0x1249249249249249ull;
  v = (v | (v >> 2)) & 0x30C30C30C30C30C3ull;
  v = (v | (v >> 4)) & 0xF00F00F00F00F00Full;
  v = (v | (v >> 8)) & 0x00FF0000FF0000FFull;
  v = (v | (v >> 16)) & 0x00FF00000000FFFFull;
  v = (v | (v >> 32)) & 0x00000000FFFFFFFFull;
  return v;
}

// the same for values below 2^32 (the 64-bit masks truncated: every step only shifts right)
__device__ __host__ __forceinline__ uint32_t contract_bits_by_3_u32(uint32_t v) {
  v &= 0x49249249u;
  v = (v | (v >> 2)) & 0xC30C30C3u;
  v = (v | (v >> 4)) & 0x0F00F00Fu;
  v = (v | (v >> 8)) & 0xFF0000FFu;
  v = (v | (v >> 16)) & 0x0000FFFFu;
  return v;
}

// The three 21-bit coordinates of a 63-bit Morton key (bit 3j+2 of the key is bit j of x, 3j+1 of y, 3j of z,
// MortonIndex.h:62-79) with 32-bit arithmetic -- a third of the instructions of three 64-bit contract_bits_by_3.  The low
// word holds x bits 0-9, y bits 0-10, z bits 0-10; the high word (bit 32 on) the rest.
__device__ __host__ __forceinline__ void key_coords_u32(uint64_t key, uint32_t& x, uint32_t& y, uint32_t& z) {
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  x = contract_bits_by_3_u32(lo >> 2) | (contract_bits_by_3_u32(hi) << 10);
  y = contract_bits_by_3_u32(lo >> 1) | (contract_bits_by_3_u32(hi >> 2) << 11);
  z = contract_bits_by_3_u32(lo) | (contract_bits_by_3_u32(hi >> 1) << 11);
}

struct Box {  // AABB, core/math/AABB.h
  double minx, miny, minz, maxx, maxy, maxz;
};

// get_bounds_from_morton_index(key, root, depth) -- core/tiling/OctreeAlgorithms.h:104-116 with
// get_octant_bounds -- OctreeAlgorithms.cpp:3-18 iterated: min' = bit ? min + extent/2 : min,
// max' = min' + extent/2, per axis, never a closed form (bit-exactness checklist item 5).
__device__ __host__ __forceinline__ void box_descend(Box& b, uint32_t o) {
  const double ex = b.maxx - b.minx, ey = b.maxy - b.miny, ez = b.maxz - b.minz;
  const double nz = (o & 1u) ? (b.minz + ez / 2) : b.minz;
  const double ny = (o & 2u) ? (b.miny + ey / 2) : b.miny;
  const double nx = (o & 4u) ? (b.minx + ex / 2) : b.minx;
  b.minx = nx;
  b.miny = ny;
  b.minz = nz;
  b.maxx = nx + ex / 2;
  b.maxy = ny + ey / 2;
  b.maxz = nz + ez / 2;
}
__device__ __host__ __forceinline__ Box bounds_from_key(uint64_t key, const Box& root, int depth) {
  Box b = root;
  for (int level = 0; level < depth; ++level) box_descend(b, (uint32_t)(key >> level_shift(level)) & 7u);
  return b;
}
// The same for N keys at once: N independent chains of dependent f64 operations in one loop.
// b[j] holds the bounds of key[j]'s ancestor at depth `from` on entry (the root box for from = 0).
template <int N>
__device__ __forceinline__ void bounds_from_keys(const uint64_t (&key)[N], int from, int depth, Box (&b)[N]) {
  for (int level = from; level < depth; ++level) {
    const uint32_t sh = level_shift(level);
#pragma unroll
    for (int j = 0; j < N; ++j) box_descend(b[j], (uint32_t)(key[j] >> sh) & 7u);
  }
}

// Vector3::squaredDistanceTo -- core/math/Vector3.h:55-62: (dx*dx + dy*dy) + dz*dz, no FMA
// (the library is compiled with -ffp-contract=off).
__device__ __host__ __forceinline__ double sq_dist(double ax, double ay, double az, double bx, double by,
                                                   double bz) {
  const double dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

}  // namespace swz
