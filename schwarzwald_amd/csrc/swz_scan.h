// swz_scan.h -- fused device-wide exclusive scan: the scanned value of element i is COMPUTED by a
// functor (no flag array is materialised) and the result is CONSUMED by a second functor (node ids,
// stream compaction, ...), so a "flag + scan + scatter" sequence costs two passes over the inputs
// instead of five.  Three phases: per-tile sums -> scan of the tile sums (swz_sort.hip) -> apply.
// (A one-pass form with decoupled look-back -- inputs read once -- was built and measured in round 2: no faster, the
// second read of the keys is not what bounds these kernels; it was dropped again.)
#pragma once
#include <string>

#include "swz_device.h"
#include "swz_internal.h"

namespace swz {

constexpr int FS_THREADS = 256;
constexpr int FS_IPT = 2;  // 2 x u64 = one 16-byte access per lane: loads and the compacted stores stay coalesced
constexpr int FS_TILE = FS_THREADS * FS_IPT;

template <typename F>
__global__ __launch_bounds__(FS_THREADS) void fscan_partial_kernel(F f, uint32_t n, uint32_t* __restrict__ partial) {
  __shared__ uint32_t lds[FS_THREADS / WAVE];
  const uint32_t base = blockIdx.x * FS_TILE + threadIdx.x * FS_IPT;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < FS_IPT; ++j)
    if (base + j < n) s += f(base + j);
  uint32_t total;
  block_excl_sum<FS_THREADS>(s, lds, total);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// A functor whose value is "byte i of an array is zero" can say so: specialise this with value = true and give the
// functor a `const uint8_t* taken` member.  Its tile sums then come from 16-byte loads, 32 lanes per tile, with no LDS
// and no barrier (one byte per thread and a block scan per 512 bytes cost four times the time of reading them).
template <typename F>
struct FsCountsZeroBytes {
  static constexpr bool value = false;
};
__device__ __forceinline__ uint32_t fs_zero_bytes(uint32_t w) {
  const uint32_t t = (w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return (uint32_t)__popc(~(t | w | 0x7F7F7F7Fu));
}
static __global__ __launch_bounds__(FS_THREADS) void fscan_partial_zero_bytes_kernel(const uint8_t* __restrict__ bytes, uint32_t n,
                                                                              uint32_t nb, uint32_t* __restrict__ partial) {
  static_assert(FS_TILE == 32 * 16, "32 lanes x 16 bytes per tile");
  const uint32_t tile = blockIdx.x * (FS_THREADS / 32) + threadIdx.x / 32, sub = threadIdx.x % 32;
  const uint64_t off = (uint64_t)tile * FS_TILE + sub * 16u;
  uint32_t cnt = 0;
  if (off + 16u <= (uint64_t)n) {
    const uint4 v = *reinterpret_cast<const uint4*>(bytes + off);
    cnt = fs_zero_bytes(v.x) + fs_zero_bytes(v.y) + fs_zero_bytes(v.z) + fs_zero_bytes(v.w);
  } else {
    for (uint32_t k = 0; k < 16u; ++k)
      if (off + k < (uint64_t)n) cnt += bytes[off + k] == 0 ? 1u : 0u;
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) cnt += __shfl_down(cnt, d, 32);
  if (sub == 0 && tile < nb) partial[tile] = cnt;
}

template <typename F, typename G>
__global__ __launch_bounds__(FS_THREADS) void fscan_apply_kernel(F f, G g, uint32_t n,
                                                                 const uint32_t* __restrict__ partial_scanned) {
  __shared__ uint32_t lds[FS_THREADS / WAVE];
  const uint32_t base = blockIdx.x * FS_TILE + threadIdx.x * FS_IPT;
  uint32_t v[FS_IPT];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < FS_IPT; ++j) {
    v[j] = (base + j < n) ? f(base + j) : 0u;
    s += v[j];
  }
  uint32_t total;
  uint32_t ex = block_excl_sum<FS_THREADS>(s, lds, total) + partial_scanned[blockIdx.x];
#pragma unroll
  for (int j = 0; j < FS_IPT; ++j) {
    if (base + j < n) g(base + j, ex, v[j]);
    ex += v[j];
  }
}

// f: uint32_t(uint32_t i)   g: void(uint32_t i, uint32_t exclusive_prefix, uint32_t value)
// In two steps for callers that need the total on the host before they can build g (buffers sized by it):
// fused_scan_sums leaves the scanned tile sums in *d_partial_out and the total in d_total, fused_scan_apply consumes them.
template <typename F>
int fused_scan_sums(swz_ctx* c, F f, uint32_t n, uint32_t* d_total, const char* tag, uint32_t** d_partial_out) {
  *d_partial_out = nullptr;
  if (n == 0) {
    if (d_total) SWZ_HIP(c, hipMemsetAsync(d_total, 0, sizeof(uint32_t), c->stream));
    return SWZ_OK;
  }
  const uint32_t nb = div_up(n, FS_TILE);
  uint32_t* d_partial = nullptr;
  const std::string name = std::string("fscan_partial_") + tag;
  SWZ_TRY(c->get(name.c_str(), (size_t)nb, &d_partial));
  if constexpr (FsCountsZeroBytes<F>::value)
    hipLaunchKernelGGL(fscan_partial_zero_bytes_kernel, dim3(div_up(nb, (uint32_t)(FS_THREADS / 32))), dim3(FS_THREADS), 0,
                       c->stream, f.taken, n, nb, d_partial);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(fscan_partial_kernel<F>), dim3(nb), dim3(FS_THREADS), 0, c->stream, f, n, d_partial);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, d_partial, d_partial, nb, d_total, tag));
  *d_partial_out = d_partial;
  return SWZ_OK;
}
template <typename F, typename G>
int fused_scan_apply(swz_ctx* c, F f, G g, uint32_t n, const uint32_t* d_partial) {
  if (n == 0) return SWZ_OK;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fscan_apply_kernel<F, G>), dim3(div_up(n, FS_TILE)), dim3(FS_THREADS), 0, c->stream, f, g, n,
                     d_partial);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}
template <typename F, typename G>
int fused_scan(swz_ctx* c, F f, G g, uint32_t n, uint32_t* d_total, const char* tag) {
  uint32_t* d_partial = nullptr;
  SWZ_TRY(fused_scan_sums(c, f, n, d_total, tag, &d_partial));
  return fused_scan_apply(c, f, g, n, d_partial);
}

}  // namespace swz
