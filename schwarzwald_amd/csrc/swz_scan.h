// swz_scan.h -- fused device-wide exclusive scan: the scanned value of element i is COMPUTED by a
// functor (no flag array is materialised) and the result is CONSUMED by a second functor (node ids,
// stream compaction, ...), so a "flag + scan + scatter" sequence costs two passes over the inputs
// instead of five.  Three phases: per-tile sums -> scan of the tile sums (swz_sort.hip) -> apply.
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

// ---- one-pass form: the inputs are read ONCE.  A tile (4096 elements, taken in ticket order) computes its values and
// their sum, publishes the sum, learns the sum of everything in front of it by decoupled look-back -- one wavefront
// looks at the 64 tiles in front at a time and adds their sums up to the nearest tile whose INCLUSIVE prefix is
// known -- publishes its own inclusive prefix and applies.  A status word carries flag and value together (one
// relaxed agent-scope 64-bit access), tiles only wait for tiles that already run.
constexpr int FO_THREADS = 256;
constexpr int FO_IPT = 16;
constexpr int FO_TILE = FO_THREADS * FO_IPT;
constexpr unsigned long long FO_LOCAL = 1ull << 32, FO_INCL = 2ull << 32;

// One tile per ticket.  (Handing out batches of consecutive tiles per ticket was tried: a workgroup then publishes its
// tiles one after the other, so the workgroup behind it waits for its LAST tile -- a serial chain, 100x slower.  Large
// tiles keep the number of tickets, which all hit one atomic word, small instead.)
constexpr int FO_BATCH = 1;

template <typename F, typename G>
__global__ __launch_bounds__(FO_THREADS) void fscan_onepass_kernel(F f, G g, uint32_t n, uint32_t ntiles,
                                                                   unsigned long long* __restrict__ status,
                                                                   uint32_t* __restrict__ ticket, uint32_t* __restrict__ d_total) {
  __shared__ uint32_t wave_tot[2][FO_THREADS / WAVE];
  __shared__ uint32_t s_tile, s_excl;
  const uint32_t tid = threadIdx.x, w = tid / WAVE, l = lane_id();
  if (tid == 0) s_tile = atomicAdd(ticket, (uint32_t)FO_BATCH);
  __syncthreads();
  const uint32_t tile0 = s_tile;
  uint32_t carry = 0;  // sum of everything in front of the current tile (known from the second tile of the batch on)
  for (uint32_t b = 0; b < (uint32_t)FO_BATCH; ++b) {
    const uint32_t tile = tile0 + b;
    if (tile >= ntiles) break;
    // wave w owns FO_IPT chunks of 64 consecutive elements
    const uint32_t wbase = tile * FO_TILE + w * (WAVE * FO_IPT);
    uint32_t v[FO_IPT], ex[FO_IPT];
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < FO_IPT; ++k) {
      const uint32_t i = wbase + k * WAVE + l;
      v[k] = i < n ? f(i) : 0u;
    }
#pragma unroll
    for (int k = 0; k < FO_IPT; ++k) {
      const uint32_t incl = wave_incl_sum(v[k]);
      ex[k] = run + incl - v[k];
      run += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    }
    if (l == 0) wave_tot[b & 1u][w] = run;
    __syncthreads();
    uint32_t wave_base = 0, tile_total = 0;
#pragma unroll
    for (int i = 0; i < FO_THREADS / WAVE; ++i) {
      const uint32_t t = wave_tot[b & 1u][i];
      if ((uint32_t)i < w) wave_base += t;
      tile_total += t;
    }
    if (b == 0 && tile > 0) {
      // the first tile of the batch: what lies in front of it comes from other workgroups
      if (w == 0) {
        if (l == 0) __hip_atomic_store(&status[tile], FO_LOCAL | tile_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        uint32_t hi = tile;  // tiles [.., hi) are still to be accounted for
        for (;;) {
          const bool have = l < hi;
          const uint32_t p = have ? hi - 1u - l : 0u;  // lane 0 looks at the nearest tile
          unsigned long long sv = FO_INCL;               // lanes beyond tile 0 stand for "prefix 0, inclusive"
          if (have) sv = __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t ready = __ballot((sv >> 32) != 0ull);
          const uint64_t incl = __ballot((sv >> 32) == 2ull);
          // usable: the lanes before the first not-ready one; stop at the first inclusive one among them
          const int first_unready = ~ready ? __ffsll((unsigned long long)~ready) - 1 : WAVE;
          const int first_incl = incl ? __ffsll((unsigned long long)incl) - 1 : WAVE;
          const int take = first_incl < first_unready ? first_incl + 1 : first_unready;  // lanes [0, take)
          uint32_t part = ((int)l < take && have) ? (uint32_t)sv : 0u;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, WAVE);
          excl += part;
          if (first_incl < first_unready) break;
          hi -= (uint32_t)take < hi ? (uint32_t)take : hi;
          if (hi == 0) break;
          if (take == 0) __builtin_amdgcn_s_sleep(1);
        }
        if (l == 0) s_excl = excl;
      }
      __syncthreads();
      carry = s_excl;
    }
    if (tid == 0) {
      __hip_atomic_store(&status[tile], FO_INCL | (unsigned long long)(carry + tile_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d_total && tile + 1u == ntiles) *d_total = carry + tile_total;
    }
    const uint32_t base = carry + wave_base;
#pragma unroll
    for (int k = 0; k < FO_IPT; ++k) {
      const uint32_t i = wbase + k * WAVE + l;
      if (i < n) g(i, base + ex[k], v[k]);
    }
    carry += tile_total;
  }
}

template <typename F, typename G>
int fused_scan_onepass(swz_ctx* c, F f, G g, uint32_t n, uint32_t* d_total) {
  const uint32_t nt = div_up(n, FO_TILE);
  unsigned long long* d_status = nullptr;
  SWZ_TRY(c->get("fscan_status", (size_t)nt + 1, &d_status));
  // the ticket lives behind the status words: one memset clears both
  SWZ_HIP(c, hipMemsetAsync(d_status, 0, ((size_t)nt + 1) * sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fscan_onepass_kernel<F, G>), dim3(div_up(nt, FO_BATCH)), dim3(FO_THREADS), 0, c->stream, f, g,
                     n, nt, d_status, reinterpret_cast<uint32_t*>(d_status + nt), d_total);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

// f: uint32_t(uint32_t i)   g: void(uint32_t i, uint32_t exclusive_prefix, uint32_t value)
template <typename F, typename G>
int fused_scan(swz_ctx* c, F f, G g, uint32_t n, uint32_t* d_total, const char* tag) {
  if (n == 0) {
    if (d_total) SWZ_HIP(c, hipMemsetAsync(d_total, 0, sizeof(uint32_t), c->stream));
    return SWZ_OK;
  }
  {
    const char* o = c->opt("SWZ_SCAN_ONEPASS");
    if (!(o && atoi(o) == 0)) return fused_scan_onepass(c, f, g, n, d_total);
  }
  const uint32_t nb = div_up(n, FS_TILE);
  uint32_t* d_partial = nullptr;
  const std::string name = std::string("fscan_partial_") + tag;
  SWZ_TRY(c->get(name.c_str(), (size_t)nb, &d_partial));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fscan_partial_kernel<F>), dim3(nb), dim3(FS_THREADS), 0, c->stream, f, n, d_partial);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, d_partial, d_partial, nb, d_total, tag));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(fscan_apply_kernel<F, G>), dim3(nb), dim3(FS_THREADS), 0, c->stream, f, g, n,
                     d_partial);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

}  // namespace swz
