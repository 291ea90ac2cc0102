// swz_sort.hip -- device-wide exclusive scan and the stable LSD radix sort by Morton key (K2).
//
// Replaces Range::sort = std::sort of IndexedPoint64 (util/containers/Range.h:62-66) called at
// core/tiling/TilingAlgorithms.cpp:602 / :1292.  Keys are 63-bit, payload is the 32-bit point index;
// 8 passes of 8-bit digits (9-bit digits / 7 passes were measured: the shorter digit runs of a
// 4096-key tile store less contiguously and the saved pass is lost again).  Per pass: per-tile digit histogram -> one device-wide scan over the
// [digit][tile] table -> scatter.  The scatter ranks keys with wave64 ballots (match-any on the
// digit), prefix-sums per-wave digit counts in LDS, exchanges the tile through LDS so that global
// stores of one digit run are contiguous, and is stable (ties keep their input order), which makes
// the final order (key, original index).
#include <algorithm>

#include "swz_device.h"
#include "swz_internal.h"

namespace swz {

// ------------------------------------------------------------------------------------- scan
constexpr int SC_THREADS = 256;
constexpr int SC_IPT = 8;
constexpr int SC_TILE = SC_THREADS * SC_IPT;

__global__ __launch_bounds__(SC_THREADS) void scan_partial_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                  uint32_t* __restrict__ partial) {
  __shared__ uint32_t lds[SC_THREADS / WAVE];
  const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_IPT;
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_IPT; ++j)
    if (base + j < n) s += in[base + j];
  uint32_t total;
  block_excl_sum<SC_THREADS>(s, lds, total);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// single block: exclusive scan of a short array (n <= a few thousand tiles), sequential over chunks
__global__ __launch_bounds__(1024) void scan_small_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          uint32_t n, uint32_t* __restrict__ total_out) {
  __shared__ uint32_t lds[1024 / WAVE];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < n; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = (i < n) ? in[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_sum<1024>(v, lds, total);
    if (i < n) out[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(const uint32_t* __restrict__ in,
                                                                uint32_t* __restrict__ out, uint64_t n,
                                                                const uint32_t* __restrict__ partial_scanned) {
  __shared__ uint32_t lds[SC_THREADS / WAVE];
  const uint64_t base = (uint64_t)blockIdx.x * SC_TILE + (uint64_t)threadIdx.x * SC_IPT;
  uint32_t v[SC_IPT];
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < SC_IPT; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0u;
    s += v[j];
  }
  uint32_t total;
  uint32_t ex = block_excl_sum<SC_THREADS>(s, lds, total) + partial_scanned[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SC_IPT; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

static int scan_rec(swz_ctx* c, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint32_t* d_total,
                    const std::string& tag, int depth) {
  if (n <= 4096) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, c->stream, d_in, d_out, (uint32_t)n, d_total);
    SWZ_LAUNCH_CHECK(c);
    return SWZ_OK;
  }
  const uint32_t nb = div_up(n, SC_TILE);
  uint32_t* d_partial = nullptr;
  const std::string name = "scan_partial_" + tag + "_" + std::to_string(depth);
  SWZ_TRY(c->get(name.c_str(), (size_t)nb, &d_partial));
  hipLaunchKernelGGL(scan_partial_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, d_in, n, d_partial);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_rec(c, d_partial, d_partial, nb, d_total, tag, depth + 1));
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, d_in, d_out, n, d_partial);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

int scan_exclusive_u32(swz_ctx* c, const uint32_t* d_in, uint32_t* d_out, uint64_t n, uint32_t* d_total,
                       const char* tag) {
  if (n == 0) {
    if (d_total) SWZ_HIP(c, hipMemsetAsync(d_total, 0, sizeof(uint32_t), c->stream));
    return SWZ_OK;
  }
  return scan_rec(c, d_in, d_out, n, d_total, tag, 0);
}

// ------------------------------------------------------------------------------------- radix sort
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_KPT = 16;                      // keys per thread
constexpr int RS_TILE = RS_THREADS * RS_KPT;    // 4096 keys per workgroup
constexpr int RS_WAVE_SPAN = WAVE * RS_KPT;     // 1024 consecutive keys per wave
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;         // 256 bins
constexpr int RADIX_PASSES = 8;                // 8 * 8 = 64 >= 63 key bits
constexpr int DPT = RADIX / RS_THREADS;        // digits owned by one thread

// per-tile digit histogram -> hist[digit * ntiles + tile]
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                                int shift, uint32_t* __restrict__ hist,
                                                                uint32_t ntiles) {
  __shared__ uint32_t h[RS_WAVES][RADIX];
  const uint32_t tid = threadIdx.x, w = tid / WAVE;
#pragma unroll
  for (int i = 0; i < RS_WAVES; ++i)
#pragma unroll
    for (int d = 0; d < DPT; ++d) h[i][tid * DPT + d] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int k = 0; k < RS_KPT; ++k) {
    const uint64_t i = base + (uint64_t)k * RS_THREADS + tid;
    if (i < n) atomicAdd(&h[w][(uint32_t)(keys[i] >> shift) & (RADIX - 1)], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < DPT; ++d) {
    const uint32_t dg = tid * DPT + d;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) s += h[i][dg];
    hist[(uint64_t)dg * ntiles + blockIdx.x] = s;
  }
}

// wave64 match-any on a digit among the lanes set in `valid`
__device__ __forceinline__ uint64_t match_digit(uint32_t d, uint64_t valid) {
  uint64_t peers = valid;
#pragma unroll
  for (int b = 0; b < RADIX_BITS; ++b) {
    const bool bit = (d >> b) & 1u;
    const uint64_t m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint64_t* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, uint32_t n,
                                                                   int shift, const uint32_t* __restrict__ offs,
                                                                   uint32_t ntiles) {
  __shared__ uint32_t whist[RS_WAVES][RADIX];  // per-wave digit counts, then exclusive wave bases
  __shared__ uint32_t dstart[RADIX];           // start of each digit run inside the LDS-sorted tile
  __shared__ uint32_t gbase[RADIX];            // global offset of this tile's run of each digit
  __shared__ uint32_t scan_lds[RS_WAVES];
  __shared__ uint64_t xkeys[RS_TILE];
  __shared__ uint32_t xvals[RS_TILE];

  const uint32_t tid = threadIdx.x, w = tid / WAVE, l = lane_id();
  const uint64_t tile_base = (uint64_t)blockIdx.x * RS_TILE;
  const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)RS_TILE ? (n - tile_base) : RS_TILE);

#pragma unroll
  for (int d = 0; d < DPT; ++d) {
    const uint32_t dg = tid * DPT + d;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) whist[i][dg] = 0;
    gbase[dg] = offs[(uint64_t)dg * ntiles + blockIdx.x];
  }

  // wave-blocked arrangement: wave w owns keys [w*1024, (w+1)*1024) of the tile, item k of lane l is
  // element w*1024 + k*64 + l, so the order inside the tile is (wave, item, lane)
  uint64_t key[RS_KPT];
  uint32_t val[RS_KPT];
  uint32_t rank[RS_KPT];
#pragma unroll
  for (int k = 0; k < RS_KPT; ++k) {
    const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
    if (e < tile_n) {
      key[k] = keys_in[tile_base + e];
      val[k] = vals_in ? vals_in[tile_base + e] : (uint32_t)(tile_base + e);
    } else {
      key[k] = ~0ull;
      val[k] = 0;
    }
  }
  __syncthreads();

  // rank of every key among the equal-digit keys of its own wave, in (item, lane) order.  The
  // per-wave counters are read and written by different lanes in successive items: volatile keeps
  // every access a real, in-order LDS operation.
  volatile uint32_t* wh = whist[w];
#pragma unroll
  for (int k = 0; k < RS_KPT; ++k) {
    const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
    const bool ok = e < tile_n;
    const uint64_t valid = __ballot(ok);
    const uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
    const uint64_t peers = match_digit(d, valid);
    uint32_t pre = 0;
    if (ok) {
      const uint32_t leader = (uint32_t)__ffsll((unsigned long long)peers) - 1u;
      if (l == leader) {
        pre = wh[d];
        wh[d] = pre + (uint32_t)__popcll(peers);
      }
      pre = __shfl(pre, leader, WAVE);
      rank[k] = pre + (uint32_t)__popcll(peers & lanemask_lt());
    } else {
      rank[k] = 0;
    }
  }
  __syncthreads();

  // thread t owns digits t*DPT ..: exclusive prefix over the waves, tile totals, then exclusive scan over digits
  uint32_t cnt[DPT];
  uint32_t s = 0;
#pragma unroll
  for (int d = 0; d < DPT; ++d) {
    const uint32_t dg = tid * DPT + d;
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) {
      const uint32_t t = whist[i][dg];
      whist[i][dg] = acc;
      acc += t;
    }
    cnt[d] = acc;
    s += acc;
  }
  uint32_t total;
  uint32_t ds = block_excl_sum<RS_THREADS>(s, scan_lds, total);
#pragma unroll
  for (int d = 0; d < DPT; ++d) {
    dstart[tid * DPT + d] = ds;
    ds += cnt[d];
  }
  __syncthreads();

  // exchange through LDS: position inside the digit-sorted tile
#pragma unroll
  for (int k = 0; k < RS_KPT; ++k) {
    const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
    if (e < tile_n) {
      const uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
      const uint32_t p = dstart[d] + whist[w][d] + rank[k];
      xkeys[p] = key[k];
      xvals[p] = val[k];
    }
  }
  __syncthreads();

  // contiguous stores per digit run
#pragma unroll
  for (int k = 0; k < RS_KPT; ++k) {
    const uint32_t j = k * RS_THREADS + tid;
    if (j < tile_n) {
      const uint64_t kk = xkeys[j];
      const uint32_t d = (uint32_t)(kk >> shift) & (RADIX - 1);
      const uint32_t dst = gbase[d] + (j - dstart[d]);
      keys_out[dst] = kk;
      vals_out[dst] = xvals[j];
    }
  }
}


// ------------------------------------------------------------------------------------- one-sweep passes
// The three-kernel pass above reads the keys twice (tile histograms, then the scatter) and scans a [digit][tile]
// table in between.  For n < 2^30 the passes run "one sweep" instead: ONE kernel computes the eight digit histograms
// of the whole input up front (the multiset of keys is the same in every pass), and each pass is a single scatter
// kernel in which a tile learns where its digit runs start by decoupled look-back over the tiles in front of it:
// thread d of tile t publishes the tile's count of digit d (flag LOCAL), walks back over earlier tiles adding their
// counts until it meets an INCLUSIVE prefix, then publishes its own inclusive prefix.  Tiles take their index from
// an atomic ticket, so every tile only ever waits for tiles that are already running.  Status words carry flag and
// value together (one relaxed agent-scope 32-bit access, no ordering needed): 2 flag bits + 30 value bits.
constexpr int OS_BATCH = 1;  // tiles per ticket: batches serialise the workgroups (each waits for the LAST tile of the one in front)
__device__ __forceinline__ uint32_t div_up_dev(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
constexpr uint32_t OS_FLAG_LOCAL = 1u << 30, OS_FLAG_INCL = 2u << 30, OS_VALUE_MASK = (1u << 30) - 1u;

// (first: the lowest digit anybody will ask for -- a sort that runs passes over the top four digits only spends half the
// LDS atomics, which are what this kernel takes its time for: 2.5 ms per 1 B keys with eight digits)
__global__ __launch_bounds__(RS_THREADS) void radix_ghist_kernel(const uint64_t* __restrict__ keys, uint32_t n,
                                                                 uint32_t* __restrict__ ghist /*[8][256]*/, int first) {
  __shared__ uint32_t h[RADIX_PASSES][RADIX];
  for (uint32_t i = threadIdx.x; i < RADIX_PASSES * RADIX; i += RS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  for (uint64_t base = (uint64_t)blockIdx.x * RS_TILE; base < n; base += (uint64_t)gridDim.x * RS_TILE) {
#pragma unroll 4
    for (int k = 0; k < RS_KPT; ++k) {
      const uint64_t i = base + (uint64_t)k * RS_THREADS + threadIdx.x;
      if (i < n) {
        const uint64_t key = keys[i];
#pragma unroll
        for (int p = 0; p < RADIX_PASSES; ++p)
          if (p >= first) atomicAdd(&h[p][(uint32_t)(key >> (p * RADIX_BITS)) & (RADIX - 1)], 1u);
      }
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < RADIX_PASSES * RADIX; i += RS_THREADS) {
    const uint32_t v = (&h[0][0])[i];
    if (v) atomicAdd(&ghist[i], v);
  }
}
// exclusive scan of each pass's 256-bin histogram (one block per pass)
__global__ __launch_bounds__(RADIX) void radix_gscan_kernel(uint32_t* __restrict__ ghist) {
  __shared__ uint32_t lds[RADIX / WAVE];
  uint32_t* hrow = ghist + blockIdx.x * RADIX;
  const uint32_t v = hrow[threadIdx.x];
  uint32_t total;
  const uint32_t ex = block_excl_sum<RADIX>(v, lds, total);
  hrow[threadIdx.x] = ex;
}

__global__ __launch_bounds__(RS_THREADS) void radix_onesweep_kernel(const uint64_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint64_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out, uint32_t n, int shift,
                                                                    const uint32_t* __restrict__ gstart /*[256]*/,
                                                                    uint32_t* __restrict__ status /*[ntiles][256]*/,
                                                                    uint32_t* __restrict__ ticket) {
  // 52 KB of LDS: three workgroups per CU (the per-wave counters fit 16 bits: a wave holds 1024 keys)
  __shared__ uint16_t whist[RS_WAVES][RADIX];
  __shared__ uint32_t doff[RADIX];  // start of the digit's run inside the tile, later: global start minus that
  __shared__ uint32_t scan_lds[RS_WAVES];
  __shared__ uint64_t xkeys[RS_TILE];
  __shared__ uint32_t xvals[RS_TILE];
  __shared__ uint32_t s_tile;

  const uint32_t tid = threadIdx.x, w = tid / WAVE, l = lane_id();
  if (tid == 0) s_tile = atomicAdd(ticket, (uint32_t)OS_BATCH);
  __syncthreads();
  const uint32_t tile0 = s_tile;
  const uint32_t ntiles = div_up_dev(n, RS_TILE);
  uint32_t carry = 0;  // digit tid: keys with this digit in all tiles in front of the current one
  for (uint32_t b = 0; b < (uint32_t)OS_BATCH; ++b) {
    const uint32_t tile = tile0 + b;
    if (tile >= ntiles) break;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) whist[i][tid] = 0;
    __syncthreads();
    const uint64_t tile_base = (uint64_t)tile * RS_TILE;
    const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)RS_TILE ? (n - tile_base) : RS_TILE);

    uint64_t key[RS_KPT];
    uint32_t val[RS_KPT];
    uint32_t rank[RS_KPT];
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
      const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
      if (e < tile_n) {
        key[k] = keys_in[tile_base + e];
        val[k] = vals_in ? vals_in[tile_base + e] : (uint32_t)(tile_base + e);
      } else {
        key[k] = ~0ull;
        val[k] = 0;
      }
    }
    volatile uint16_t* wh = whist[w];
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
      const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
      const bool ok = e < tile_n;
      const uint64_t valid = __ballot(ok);
      const uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
      const uint64_t peers = match_digit(d, valid);
      uint32_t pre = 0;
      if (ok) {
        const uint32_t leader = (uint32_t)__ffsll((unsigned long long)peers) - 1u;
        if (l == leader) {
          pre = wh[d];
          wh[d] = (uint16_t)(pre + (uint32_t)__popcll(peers));
        }
        pre = __shfl(pre, leader, WAVE);
        rank[k] = pre + (uint32_t)__popcll(peers & lanemask_lt());
      } else {
        rank[k] = 0;
      }
    }
    __syncthreads();

    // thread tid owns digit tid: exclusive prefix over the waves, the tile's count of the digit
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) {
      const uint32_t t = whist[i][tid];
      whist[i][tid] = (uint16_t)cnt;
      cnt += t;
    }
    uint32_t* my_status = status + (size_t)tile * RADIX + tid;
    const bool look_back = b == 0 && tile > 0;  // the tiles in front belong to other workgroups
    uint32_t look = 0;
    if (look_back) {
      // publish the tile's own count right away: later tiles can walk over it while this one is still busy; and take
      // a first look at the tile in front: by the time the exchange below is done its answer has arrived
      __hip_atomic_store(my_status, OS_FLAG_LOCAL | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      look = __hip_atomic_load(status + (size_t)(tile - 1) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __hip_atomic_store(my_status, OS_FLAG_INCL | (carry + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t total;
    const uint32_t ds = block_excl_sum<RS_THREADS>(cnt, scan_lds, total);
    doff[tid] = ds;
    __syncthreads();

    // exchange through LDS: position inside the digit-sorted tile
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
      const uint32_t e = w * RS_WAVE_SPAN + k * WAVE + l;
      if (e < tile_n) {
        const uint32_t d = (uint32_t)(key[k] >> shift) & (RADIX - 1);
        const uint32_t p = doff[d] + whist[w][d] + rank[k];
        xkeys[p] = key[k];
        xvals[p] = val[k];
      }
    }

    if (look_back) {  // decoupled look-back for digit tid
      uint32_t p = tile - 1;
      uint32_t v = look;
      for (;;) {
        while ((v >> 30) == 0u) {
          __builtin_amdgcn_s_sleep(1);
          v = __hip_atomic_load(status + (size_t)p * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        carry += v & OS_VALUE_MASK;
        if ((v & OS_FLAG_INCL) || p == 0) break;
        --p;
        v = __hip_atomic_load(status + (size_t)p * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __hip_atomic_store(my_status, OS_FLAG_INCL | (carry + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // every thread has read doff for the exchange
    doff[tid] = gstart[tid] + carry - ds;
    __syncthreads();

    // contiguous stores per digit run
#pragma unroll
    for (int k = 0; k < RS_KPT; ++k) {
      const uint32_t j = k * RS_THREADS + tid;
      if (j < tile_n) {
        const uint64_t kk = xkeys[j];
        const uint32_t d = (uint32_t)(kk >> shift) & (RADIX - 1);
        const uint32_t dst = doff[d] + j;
        keys_out[dst] = kk;
        vals_out[dst] = xvals[j];
      }
    }
    carry += cnt;
    __syncthreads();  // the tile's LDS image is free again
  }
}

// ---- hybrid: LSD passes over the TOP digits only, then one pass that orders the runs of equal top bits -----------------
// Eight LSD passes move every (key, value) pair eight times.  After K passes over the top K digits the pairs are sorted by
// the top 8K bits and, inside a run of equal top bits, still in input order; what is left is a stable sort of every run
// by the whole key.  For Morton keys of a big cloud those runs are short (1 B uniform points: 0.23 points per 2^-32 of the
// volume, longest run ~10), so one more pass in which every element finds its run, counts the run's elements that have
// to precede it and writes itself to run start + that rank finishes the sort: 4 + 1 moves instead of 8.  K is chosen
// from a sorted sample of the keys (no tie in the top 32 bits -> 4 passes, none in the top 48 -> 6, else all 8); a run
// longer than a thread wants to walk goes to a workgroup (rank sort out of LDS), one longer than that makes the whole
// sort fall back to eight passes -- the choice of K only ever costs time.
constexpr uint32_t FIX_SHORT = 16;    // longest run its own elements rank themselves in
constexpr uint32_t FIX_LONG = 4096;   // longest run a workgroup ranks out of LDS

__global__ __launch_bounds__(256) void radix_sample_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t samples,
                                                           uint64_t* __restrict__ out) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < samples) out[j] = keys[(uint64_t)j * n / samples];
}
// ties[0] / ties[1]: adjacent pairs of the sorted sample that agree in the top 32 / 48 bits of the 63-bit key
__global__ __launch_bounds__(256) void radix_sample_ties_kernel(const uint64_t* __restrict__ sorted, uint32_t samples,
                                                                uint32_t* __restrict__ ties) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j == 0 || j >= samples) return;
  const uint64_t a = sorted[j - 1], b = sorted[j];
  if ((a >> 32) == (b >> 32)) atomicAdd(&ties[0], 1u);
  if ((a >> 16) == (b >> 16)) atomicAdd(&ties[1], 1u);
}

// Every element of a run of at most short_max equal prefixes writes itself to its place; the first element of a longer
// run reports the run.  in: sorted by key >> shift, stable; out: the final order.  A workgroup stages its 1024 keys and
// a halo of short_max + 1 on either side in LDS: the walks along the run are chains of dependent reads.
constexpr int FIX_TILE = 1024, FIX_HALO = FIX_SHORT + 1;
__global__ __launch_bounds__(256) void radix_fix_short_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                              uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                              int shift, uint32_t short_max, uint32_t* __restrict__ long_runs,
                                                              uint32_t* __restrict__ counters /*[0] long runs*/) {
  __shared__ uint64_t lk[FIX_TILE + 2 * FIX_HALO];
  const uint64_t base = (uint64_t)blockIdx.x * FIX_TILE;  // lk[j] = kin[base - FIX_HALO + j]
  for (uint32_t j = threadIdx.x; j < FIX_TILE + 2 * FIX_HALO; j += 256u) {
    const uint64_t g = base + j;
    if (g >= FIX_HALO && g - FIX_HALO < n) lk[j] = kin[g - FIX_HALO];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < FIX_TILE / 256; ++u) {
    const uint32_t t = (uint32_t)u * 256u + threadIdx.x;  // coalesced reads of the values, neighbouring writes
    const uint64_t gi = base + t;
    if (gi >= n) continue;
    const uint32_t i = (uint32_t)gi;
    const uint32_t c = t + FIX_HALO;  // own slot
    const uint64_t key = lk[c];
    const uint64_t pre = key >> shift;
    // the run [s, e) around i, as far as it matters (never farther than the halo reaches)
    uint32_t s = i, e = i + 1u, rank = 0;
    while (s > 0 && i - s <= short_max) {
      const uint64_t o = lk[c - (i - s) - 1u];
      if ((o >> shift) != pre) break;
      --s;
      rank += o <= key ? 1u : 0u;  // an earlier element precedes unless its key is larger
    }
    while (e < n && e - s <= short_max && e - i <= short_max) {
      const uint64_t o = lk[c + (e - i)];
      if ((o >> shift) != pre) break;
      ++e;
      rank += o < key ? 1u : 0u;   // a later element precedes only with a smaller key
    }
    const bool more_right = e < n && e - i > short_max && (lk[c + (e - i)] >> shift) == pre;  // the walk ran out of halo
    if (e - s > short_max || more_right) {  // a long run: its first element reports it
      if (s == i && (i == 0 || (lk[c - 1u] >> shift) != pre)) long_runs[atomicAdd(&counters[0], 1u)] = i;
      continue;
    }
    kout[s + rank] = key;
    vout[s + rank] = vin[i];
  }
}

// one workgroup per long run: rank sort out of LDS; a run longer than long_max raises counters[1]
__global__ __launch_bounds__(256) void radix_fix_long_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                             uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n,
                                                             int shift, uint32_t long_max, const uint32_t* __restrict__ long_runs,
                                                             uint32_t* __restrict__ counters) {
  __shared__ uint64_t lk[FIX_LONG];
  const uint32_t nruns = counters[0];
  for (uint32_t r = blockIdx.x; r < nruns; r += gridDim.x) {
    const uint32_t s = long_runs[r];
    const uint64_t pre = kin[s] >> shift;
    __syncthreads();
    // length: every thread probes a stride of positions until the prefix changes (runs are contiguous)
    uint32_t len = 0;
    for (uint32_t base = 0;; base += 256u) {
      const uint32_t j = s + base + threadIdx.x;
      const bool in = j < n && base + threadIdx.x <= long_max && (kin[j] >> shift) == pre;
      if (in) lk[(base + threadIdx.x) < FIX_LONG ? (base + threadIdx.x) : 0u] = kin[j];
      const uint32_t cnt = __syncthreads_count(in ? 1 : 0);
      len += cnt;
      if (cnt < 256u) break;
    }
    if (len > long_max) {
      if (threadIdx.x == 0) atomicAdd(&counters[1], 1u);
      continue;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < len; j += 256u) {
      const uint64_t key = lk[j];
      uint32_t rank = 0;
      for (uint32_t m = 0; m < len; ++m) {
        const uint64_t o = lk[m];
        rank += (o < key || (o == key && m < j)) ? 1u : 0u;
      }
      kout[s + rank] = key;
      vout[s + rank] = vin[s + j];
    }
  }
}

__global__ __launch_bounds__(256) void radix_copy_pairs_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                               uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, uint32_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    kout[i] = kin[i];
    vout[i] = vin[i];
  }
}

// The sorted pairs always end in the SECOND buffer pair: the hybrid moves the data an odd number of times (4 or 6 passes
// and the run pass); the plain eight passes end in the first pair and are copied over (small inputs, 2^30 points and
// more, keys with long runs of equal top bits).
bool radix_result_in_second() { return true; }

static int radix_lsd_passes(swz_ctx* c, uint64_t*& kin, uint32_t*& vin, uint64_t*& kout, uint32_t*& vout, uint32_t n, int first_pass,
                            int last_pass, bool first_synthesises_values, uint32_t* d_ghist, uint32_t* d_status, uint32_t* d_ticket) {
  const uint32_t ntiles = div_up(n, RS_TILE);
  SWZ_HIP(c, hipMemsetAsync(d_ticket, 0, sizeof(uint32_t) * RADIX_PASSES, c->stream));
  for (int pass = first_pass; pass <= last_pass; ++pass) {
    ProfScope ps(c, "radix_scatter", (uint64_t)n * 24ull);
    SWZ_HIP(c, hipMemsetAsync(d_status, 0, (size_t)ntiles * RADIX * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(radix_onesweep_kernel, dim3(div_up(ntiles, OS_BATCH)), dim3(RS_THREADS), 0, c->stream, kin,
                       (pass == first_pass && first_synthesises_values) ? (const uint32_t*)nullptr : vin, kout, vout, n,
                       pass * RADIX_BITS, d_ghist + pass * RADIX, d_status, d_ticket + pass);
    SWZ_LAUNCH_CHECK(c);
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  return SWZ_OK;
}

static int radix_copy_pairs(swz_ctx* c, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout, uint32_t n) {
  ProfScope ps(c, "radix_copy", (uint64_t)n * 24ull);
  hipLaunchKernelGGL(radix_copy_pairs_kernel, dim3(std::min<uint32_t>(div_up(n, 256), 16384u)), dim3(256), 0, c->stream, kin, vin,
                     kout, vout, n);
  SWZ_LAUNCH_CHECK(c);
  return SWZ_OK;
}

int radix_sort_pairs(swz_ctx* c, uint64_t* d_keys_in, uint32_t* d_vals_tmp, uint64_t* d_keys_out,
                     uint32_t* d_vals_out, uint32_t n, bool vals_identity) {
  if (n == 0) return SWZ_OK;
  const uint32_t ntiles = div_up(n, RS_TILE);
  // the data ping-pongs first -> second -> first ...
  uint64_t* kin = d_keys_in;
  uint32_t* vin = d_vals_tmp;
  uint64_t* kout = d_keys_out;
  uint32_t* vout = d_vals_out;
  const char* os_opt = c->opt("SWZ_SORT_ONESWEEP");
  if (n < (1u << 30) && !(os_opt && atoi(os_opt) == 0)) {
    static_assert(RADIX == RS_THREADS, "one thread per digit");
    uint32_t *d_ghist = nullptr, *d_status = nullptr, *d_ticket = nullptr;
    SWZ_TRY(c->get("radix_ghist", (size_t)RADIX_PASSES * RADIX, &d_ghist));
    SWZ_TRY(c->get("radix_status", (size_t)ntiles * RADIX, &d_status));
    SWZ_TRY(c->get("radix_ticket", (size_t)RADIX_PASSES, &d_ticket));
    // the digit histograms of the whole input, for the digits from `first` up
    auto global_histograms = [&](int first) -> int {
      ProfScope ps(c, "radix_hist", (uint64_t)n * 8ull);
      SWZ_HIP(c, hipMemsetAsync(d_ghist, 0, sizeof(uint32_t) * RADIX_PASSES * RADIX, c->stream));
      hipLaunchKernelGGL(radix_ghist_kernel, dim3(std::min<uint32_t>(ntiles, 256u * 8u)), dim3(RS_THREADS), 0, c->stream, kin, n, d_ghist, first);
      hipLaunchKernelGGL(radix_gscan_kernel, dim3(RADIX_PASSES), dim3(RADIX), 0, c->stream, d_ghist);
      SWZ_LAUNCH_CHECK(c);
      return SWZ_OK;
    };
    // how many top digits: from a sorted sample of the keys
    int top = RADIX_PASSES;
    uint32_t min_n = 1u << 24, short_max = FIX_SHORT, long_max = FIX_LONG;
    if (const char* e = c->opt("SWZ_SORT_HYBRID_MIN_N")) min_n = (uint32_t)atoll(e);
    if (const char* e = c->opt("SWZ_SORT_FIX_SHORT")) short_max = std::max(1u, std::min<uint32_t>(FIX_SHORT, (uint32_t)atoi(e)));
    if (const char* e = c->opt("SWZ_SORT_FIX_LONG")) long_max = std::max(short_max, std::min<uint32_t>(FIX_LONG, (uint32_t)atoi(e)));
    uint32_t* d_fix = nullptr;  // [0] long runs, [1] runs too long, [2..3] sample ties
    SWZ_TRY(c->get("radix_fix_counters", (size_t)4, &d_fix));
    if (n >= min_n) {
      const uint32_t samples = std::min<uint32_t>(n, 32768u);
      uint64_t *d_sk = nullptr, *d_sk2 = nullptr;
      uint32_t *d_sv = nullptr, *d_sv2 = nullptr;
      SWZ_TRY(c->get("radix_sample_k", (size_t)samples, &d_sk));
      SWZ_TRY(c->get("radix_sample_k2", (size_t)samples, &d_sk2));
      SWZ_TRY(c->get("radix_sample_v", (size_t)samples, &d_sv));
      SWZ_TRY(c->get("radix_sample_v2", (size_t)samples, &d_sv2));
      uint32_t* d_sghist = nullptr;
      SWZ_TRY(c->get("radix_sample_ghist", (size_t)RADIX_PASSES * RADIX, &d_sghist));
      hipLaunchKernelGGL(radix_sample_kernel, dim3(div_up(samples, 256)), dim3(256), 0, c->stream, kin, n, samples, d_sk);
      SWZ_HIP(c, hipMemsetAsync(d_sghist, 0, sizeof(uint32_t) * RADIX_PASSES * RADIX, c->stream));
      hipLaunchKernelGGL(radix_ghist_kernel, dim3(div_up(samples, RS_TILE)), dim3(RS_THREADS), 0, c->stream, d_sk, samples, d_sghist, 0);
      hipLaunchKernelGGL(radix_gscan_kernel, dim3(RADIX_PASSES), dim3(RADIX), 0, c->stream, d_sghist);
      SWZ_LAUNCH_CHECK(c);
      uint64_t *a = d_sk, *b = d_sk2;
      uint32_t *va = d_sv, *vb = d_sv2;
      SWZ_TRY(radix_lsd_passes(c, a, va, b, vb, samples, 0, RADIX_PASSES - 1, true, d_sghist, d_status, d_ticket));
      SWZ_HIP(c, hipMemsetAsync(d_fix, 0, 16, c->stream));
      hipLaunchKernelGGL(radix_sample_ties_kernel, dim3(div_up(samples, 256)), dim3(256), 0, c->stream, a, samples, d_fix + 2);
      SWZ_LAUNCH_CHECK(c);
      uint32_t h[4];
      SWZ_HIP(c, hipMemcpyAsync(h, d_fix, 16, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      // (two tied pairs are tolerated: 32768 samples of 1 B uniform points tie in the top 32 bits with probability 0.12)
      top = h[2] <= 2u ? 4 : (h[3] <= 2u ? 6 : RADIX_PASSES);
    }
    if (const char* e = c->opt("SWZ_SORT_HYBRID_TOP")) top = std::max(2, std::min(RADIX_PASSES, atoi(e) & ~1));
    if (c->opt("SWZ_DEBUG") && n >= min_n) fprintf(stderr, "[swz] sort: %u keys, passes over the top %d digits\n", n, top);
    if (top < RADIX_PASSES) {
      const int first_pass = RADIX_PASSES - top;
      SWZ_TRY(global_histograms(first_pass));
      SWZ_TRY(radix_lsd_passes(c, kin, vin, kout, vout, n, first_pass, RADIX_PASSES - 1, vals_identity, d_ghist, d_status, d_ticket));
      // an even number of passes: the data is back in the first pair; the run pass writes the second
      uint32_t* d_long = nullptr;
      SWZ_TRY(c->get("radix_fix_long", (size_t)div_up(n, short_max + 1u) + 1u, &d_long));
      {
        ProfScope ps(c, "radix_runs", (uint64_t)n * 24ull);
        SWZ_HIP(c, hipMemsetAsync(d_fix, 0, 8, c->stream));
        hipLaunchKernelGGL(radix_fix_short_kernel, dim3(div_up(n, FIX_TILE)), dim3(256), 0, c->stream, kin, vin, kout, vout, n,
                           first_pass * RADIX_BITS, short_max, d_long, d_fix);
        hipLaunchKernelGGL(radix_fix_long_kernel, dim3(1024), dim3(256), 0, c->stream, kin, vin, kout, vout, n,
                           first_pass * RADIX_BITS, long_max, d_long, d_fix);
        SWZ_LAUNCH_CHECK(c);
      }
      uint32_t h[2];
      SWZ_HIP(c, hipMemcpyAsync(h, d_fix, 8, hipMemcpyDeviceToHost, c->stream));
      SWZ_HIP(c, hipStreamSynchronize(c->stream));
      if (h[1] == 0) return SWZ_OK;
      // runs of equal top bits too long to rank: all eight passes over what the top passes left in the first pair
      // (equal keys are still in input order there, so the result is the same stable order)
      if (c->opt("SWZ_DEBUG")) fprintf(stderr, "[swz] sort: %u runs of equal top %d bits longer than %u, falling back to eight passes\n", h[1], 8 * top - 1, long_max);
      SWZ_TRY(global_histograms(0));  // (of what the top passes left in the first pair: the same multiset of keys)
      SWZ_TRY(radix_lsd_passes(c, kin, vin, kout, vout, n, 0, RADIX_PASSES - 1, false, d_ghist, d_status, d_ticket));
      return radix_copy_pairs(c, kin, vin, kout, vout, n);
    }
    SWZ_TRY(global_histograms(0));
    SWZ_TRY(radix_lsd_passes(c, kin, vin, kout, vout, n, 0, RADIX_PASSES - 1, vals_identity, d_ghist, d_status, d_ticket));
    return radix_copy_pairs(c, kin, vin, kout, vout, n);
  }
  uint32_t* d_hist = nullptr;
  SWZ_TRY(c->get("radix_hist", (size_t)ntiles * RADIX, &d_hist));
  for (int pass = 0; pass < RADIX_PASSES; ++pass) {
    const int shift = pass * RADIX_BITS;
    {
      ProfScope ps(c, "radix_hist", (uint64_t)n * 8ull);
      hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(RS_THREADS), 0, c->stream, kin, n, shift, d_hist,
                         ntiles);
      SWZ_LAUNCH_CHECK(c);
    }
    {
      ProfScope ps(c, "radix_scan", (uint64_t)ntiles * RADIX * 8ull);
      SWZ_TRY(scan_exclusive_u32(c, d_hist, d_hist, (uint64_t)ntiles * RADIX, nullptr, "radix"));
    }
    {
      ProfScope ps(c, "radix_scatter", (uint64_t)n * 24ull);
      hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(RS_THREADS), 0, c->stream, kin,
                         (pass == 0 && vals_identity) ? (const uint32_t*)nullptr : vin, kout, vout, n, shift,
                         d_hist, ntiles);
      SWZ_LAUNCH_CHECK(c);
    }
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  return radix_copy_pairs(c, kin, vin, kout, vout, n);
}

__global__ __launch_bounds__(RADIX) void digit_starts_kernel(const uint32_t* __restrict__ offs, uint32_t ntiles,
                                                             uint32_t* __restrict__ out) {
  out[threadIdx.x] = offs[(uint64_t)threadIdx.x * ntiles];
}

int partition_by_top_byte(swz_ctx* c, const uint64_t* d_keys, uint32_t n, uint32_t* d_perm_out, uint32_t starts[256]) {
  static_assert(RADIX == 256, "one radix digit is one byte");
  for (int d = 0; d < RADIX; ++d) starts[d] = 0;
  if (n == 0) return SWZ_OK;
  const uint32_t ntiles = div_up(n, RS_TILE);
  uint32_t *d_hist = nullptr, *d_starts = nullptr;
  uint64_t* d_keys_tmp = nullptr;
  SWZ_TRY(c->get("radix_hist", (size_t)ntiles * RADIX, &d_hist));
  SWZ_TRY(c->get("part_starts", (size_t)RADIX, &d_starts));
  SWZ_TRY(c->get("part_keys", (size_t)n, &d_keys_tmp));
  const int shift = 64 - RADIX_BITS;
  hipLaunchKernelGGL(radix_hist_kernel, dim3(ntiles), dim3(RS_THREADS), 0, c->stream, d_keys, n, shift, d_hist, ntiles);
  SWZ_LAUNCH_CHECK(c);
  SWZ_TRY(scan_exclusive_u32(c, d_hist, d_hist, (uint64_t)ntiles * RADIX, nullptr, "radix"));
  hipLaunchKernelGGL(digit_starts_kernel, dim3(1), dim3(RADIX), 0, c->stream, d_hist, ntiles, d_starts);
  SWZ_LAUNCH_CHECK(c);
  hipLaunchKernelGGL(radix_scatter_kernel, dim3(ntiles), dim3(RS_THREADS), 0, c->stream, d_keys,
                     (const uint32_t*)nullptr, d_keys_tmp, d_perm_out, n, shift, d_hist, ntiles);
  SWZ_LAUNCH_CHECK(c);
  SWZ_HIP(c, hipMemcpyAsync(starts, d_starts, sizeof(uint32_t) * RADIX, hipMemcpyDeviceToHost, c->stream));
  SWZ_HIP(c, hipStreamSynchronize(c->stream));
  return SWZ_OK;
}

// the octant (key bits 60..62) is the top digit's bits below the unused bit 63
int partition_top_digit(swz_ctx* c, const uint64_t* d_keys, uint32_t n, uint32_t* d_perm_out, uint64_t octants[8]) {
  for (int o = 0; o < 8; ++o) octants[o] = 0;
  if (n == 0) return SWZ_OK;
  uint32_t starts[RADIX];
  SWZ_TRY(partition_by_top_byte(c, d_keys, n, d_perm_out, starts));
  for (int d = 0; d < RADIX; ++d)
    octants[(d >> (RADIX_BITS - 4)) & 7] += (uint64_t)((d == RADIX - 1 ? n : starts[d + 1]) - starts[d]);
  return SWZ_OK;
}

}  // namespace swz
