// swz_encode.hip -- Morton encode (K1), synthetic point generator, position gather.
#include "swz_device.h"
#include "swz_internal.h"

namespace swz {

struct EncodeArgs {
  double minx, miny, minz, maxx, maxy, maxz;
  double sx, sy, sz;  // 2^21 / extent per axis, computed on the host like the reference does per call
};

constexpr int ENC_THREADS = 256;

// index_point<21>(ClampToBounds) + calculate_morton_index<21> -- core/tiling/OctreeAlgorithms.h:145-175,
// :64-87.  One point per thread; the block's 256 points are 6 KiB of contiguous AoS doubles, staged
// through LDS so that global loads are 8-byte-per-lane coalesced instead of stride-24.
__global__ __launch_bounds__(ENC_THREADS) void encode_kernel(double* __restrict__ xyz, uint32_t n, EncodeArgs a,
                                                             uint64_t* __restrict__ keys) {
  __shared__ double s[ENC_THREADS * 3];
  const uint32_t tid = threadIdx.x;
  const uint64_t base = (uint64_t)blockIdx.x * ENC_THREADS;
  const uint64_t nd = (uint64_t)n * 3ull;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint64_t j = base * 3ull + (uint64_t)k * ENC_THREADS + tid;
    s[k * ENC_THREADS + tid] = (j < nd) ? xyz[j] : 0.0;
  }
  __syncthreads();
  const uint64_t i = base + tid;
  if (i >= n) return;
  double x = s[3 * tid], y = s[3 * tid + 1], z = s[3 * tid + 2];
  // AABB::isInside -- core/math/AABB.h:27-31
  const bool inside = (x >= a.minx && x <= a.maxx && y >= a.miny && y <= a.maxy && z >= a.minz && z <= a.maxz);
  if (!inside) {
    // std::min(max, std::max(min, p)) -- OctreeAlgorithms.h:167-169; written back like the reference
    x = (a.minx < x) ? x : a.minx;  // std::max(min, p): returns p only if min < p
    x = (x < a.maxx) ? x : a.maxx;  // std::min(max, v): returns v only if v < max
    y = (a.miny < y) ? y : a.miny;
    y = (y < a.maxy) ? y : a.maxy;
    z = (a.minz < z) ? z : a.minz;
    z = (z < a.maxz) ? z : a.maxz;
    xyz[3 * i] = x;
    xyz[3 * i + 1] = y;
    xyz[3 * i + 2] = z;
  }
  // (position - min) * (2^21 / extent), truncating cast, clamp to 2^21-1
  const double nx = (x - a.minx) * a.sx;
  const double ny = (y - a.miny) * a.sy;
  const double nz = (z - a.minz) * a.sz;
  const uint64_t lim = (1ull << 21) - 1ull;
  uint64_t bx = (uint64_t)nx, by = (uint64_t)ny, bz = (uint64_t)nz;
  bx = bx < lim ? bx : lim;
  by = by < lim ? by : lim;
  bz = bz < lim ? bz : lim;
  keys[i] = expand_bits_by_3(bz) | (expand_bits_by_3(by) << 1) | (expand_bits_by_3(bx) << 2);
}

int encode_device(swz_ctx* c, double* d_xyz, uint32_t n, const double bmin[3], const double bmax[3],
                  uint64_t* d_keys) {
  if (n == 0) return SWZ_OK;
  EncodeArgs a;
  a.minx = bmin[0]; a.miny = bmin[1]; a.minz = bmin[2];
  a.maxx = bmax[0]; a.maxy = bmax[1]; a.maxz = bmax[2];
  const double two21 = 2097152.0;  // std::pow(2, 21)
  a.sx = two21 / (bmax[0] - bmin[0]);
  a.sy = two21 / (bmax[1] - bmin[1]);
  a.sz = two21 / (bmax[2] - bmin[2]);
  ProfScope ps(c, "morton_encode", (uint64_t)n * 32ull);
  hipLaunchKernelGGL(encode_kernel, dim3(div_up(n, ENC_THREADS)), dim3(ENC_THREADS), 0, c->stream, d_xyz, n, a,
                     d_keys);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

// ---------------------------------------------------------------- synthetic generator
__global__ __launch_bounds__(256) void generate_kernel(uint64_t seed, uint64_t first, uint64_t n3,
                                                       double* __restrict__ out) {
  const uint64_t G = 0x9E3779B97F4A7C15ull;
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n3; j += (uint64_t)gridDim.x * 256) {
    uint64_t z = seed + (3ull * first + j + 1ull) * G;  // draw k = 3*first + j of the splitmix64 stream
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    out[j] = (double)(z >> 11) * 0x1.0p-53;
  }
}

int generate_uniform_device(swz_ctx* c, uint64_t seed, uint64_t first, uint64_t n, double* d_xyz) {
  if (n == 0) return SWZ_OK;
  const uint64_t n3 = n * 3ull;
  const uint32_t blocks = (uint32_t)((n3 + 255) / 256 < 65536 ? (n3 + 255) / 256 : 65536);
  hipLaunchKernelGGL(generate_kernel, dim3(blocks), dim3(256), 0, c->stream, seed, first, n3, d_xyz);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

// ---------------------------------------------------------------- gather to Morton order (SoA)
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ xyz, const uint32_t* __restrict__ perm,
                                                     uint32_t n, double* __restrict__ X, double* __restrict__ Y,
                                                     double* __restrict__ Z) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t p = perm[i];
  X[i] = xyz[3 * p];
  Y[i] = xyz[3 * p + 1];
  Z[i] = xyz[3 * p + 2];
}

int gather_positions(swz_ctx* c, const double* d_xyz, const uint32_t* d_perm, uint32_t n, double* d_x,
                     double* d_y, double* d_z) {
  if (n == 0) return SWZ_OK;
  ProfScope ps(c, "gather_positions", (uint64_t)n * 52ull);
  hipLaunchKernelGGL(gather_kernel, dim3(div_up(n, 256)), dim3(256), 0, c->stream, d_xyz, d_perm, n, d_x, d_y, d_z);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

}  // namespace swz
