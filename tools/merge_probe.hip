// tools/merge_probe.hip -- a stand-alone copy of tl_merge_rank_kernel (schwarzwald_amd/csrc/swz_tiler.hip) against std::merge:
// what found the wrong ranks of its first four-elements-per-thread version.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIPT=4 tools/merge_probe.hip -o /tmp/merge_probe && /tmp/merge_probe
// -DBRACKET_ON_THREAD_64 restores the bracket searches on threads 0 and 64 (the second one alone in its wavefront: hipcc 7.2
// scalarises it and, with IPT > 1, leaves the shift count of the later loops undefined for wavefronts 2 and 3): 36 of 40 cases wrong.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
constexpr uint32_t TL_MERGE_LDS = 2048;
#ifndef IPT
#define IPT 4
#endif
constexpr uint32_t TL_MERGE_IPT = IPT;
constexpr uint32_t TL_MERGE_TILE = 256 * TL_MERGE_IPT;
template <bool UPPER>
__device__ __forceinline__ uint32_t tl_rank(const uint64_t* __restrict__ k, uint32_t lo, uint32_t hi, uint64_t ks, uint32_t sh) {
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    const uint64_t v = k[mid] >> sh;
    if (UPPER ? (v <= ks) : (v < ks)) lo = mid + 1; else hi = mid;
  }
  return lo;
}
template <bool UPPER>
__global__ __launch_bounds__(256) void tl_merge_rank_kernel(const uint64_t* __restrict__ ka, const uint32_t* __restrict__ va, uint32_t na,
                                                            const uint64_t* __restrict__ kb, uint32_t nb, uint32_t sh, uint32_t base,
                                                            uint64_t* __restrict__ ok, uint32_t* __restrict__ ov) {
  __shared__ uint32_t s_lo, s_hi;
  __shared__ uint64_t sk[TL_MERGE_LDS];
  const uint32_t tid = threadIdx.x;
  const uint32_t i0 = blockIdx.x * TL_MERGE_TILE;
  const uint32_t last = (na - i0) > TL_MERGE_TILE ? i0 + TL_MERGE_TILE - 1u : na - 1u;
#ifdef BRACKET_ON_THREAD_64
  if (tid == 0) s_lo = tl_rank<UPPER>(kb, 0u, nb, ka[i0] >> sh, sh);
  if (tid == 64) s_hi = tl_rank<UPPER>(kb, 0u, nb, ka[last] >> sh, sh);
#else
  if (tid < 2) {
    const uint32_t r = tl_rank<UPPER>(kb, 0u, nb, ka[tid ? last : i0] >> sh, sh);
    if (tid) s_hi = r; else s_lo = r;
  }
#endif
  __syncthreads();
  const uint32_t lo = s_lo, hi = s_hi;
  const bool in_lds = hi - lo <= TL_MERGE_LDS;
  if (in_lds)
    for (uint32_t j = tid; j < hi - lo; j += 256u) sk[j] = kb[lo + j] >> sh;
  __syncthreads();
  for (uint32_t q = 0; q < TL_MERGE_IPT; ++q) {
    const uint32_t i = i0 + q * 256u + tid;
    if (i >= na) break;
    const uint64_t k = ka[i];
    const uint64_t ks = k >> sh;
    uint32_t r;
    if (in_lds) r = lo + tl_rank<UPPER>(sk, 0u, hi - lo, ks, 0u);
    else r = tl_rank<UPPER>(kb, lo, hi, ks, sh);
    ok[i + r] = k;
    ov[i + r] = va ? va[i] : base + i;
  }
}
int main() {
  std::mt19937_64 rng(5);
  int bad_cases = 0;
  for (int tc = 0; tc < 40; ++tc) {
    const uint32_t na = 1 + rng() % 200000, nb = rng() % 200000;
    const uint32_t sh = (tc % 4 == 3) ? 30 : 0;
    const uint64_t range = (tc % 3 == 0) ? 1000 : (1ull << 40);
    std::vector<uint64_t> a(na), b(nb);
    for (auto& x : a) x = rng() % range;
    for (auto& x : b) x = rng() % range;
    if (sh) {  // sorted by prefix only
      std::stable_sort(a.begin(), a.end(), [&](uint64_t x, uint64_t y) { return (x >> sh) < (y >> sh); });
      std::stable_sort(b.begin(), b.end(), [&](uint64_t x, uint64_t y) { return (x >> sh) < (y >> sh); });
    } else {
      std::sort(a.begin(), a.end());
      std::sort(b.begin(), b.end());
    }
    std::vector<uint64_t> want(na + nb);
    std::merge(a.begin(), a.end(), b.begin(), b.end(), want.begin(), [&](uint64_t x, uint64_t y) { return (x >> sh) < (y >> sh); });
    uint64_t *da, *db, *dok;
    uint32_t* dov;
    (void)hipMalloc(&da, na * 8 + 8); (void)hipMalloc(&db, nb * 8 + 8); (void)hipMalloc(&dok, (na + nb) * 8); (void)hipMalloc(&dov, (na + nb) * 4);
    (void)hipMemcpy(da, a.data(), na * 8, hipMemcpyHostToDevice);
    if (nb) (void)hipMemcpy(db, b.data(), nb * 8, hipMemcpyHostToDevice);
    (void)hipMemset(dok, 0xFF, (na + nb) * 8);
    hipLaunchKernelGGL(tl_merge_rank_kernel<false>, dim3((na + TL_MERGE_TILE - 1) / TL_MERGE_TILE), dim3(256), 0, 0, da, (const uint32_t*)nullptr, na, db, nb, sh, 0u, dok, dov);
    if (nb) hipLaunchKernelGGL(tl_merge_rank_kernel<true>, dim3((nb + TL_MERGE_TILE - 1) / TL_MERGE_TILE), dim3(256), 0, 0, db, (const uint32_t*)nullptr, nb, da, na, sh, na, dok, dov);
    std::vector<uint64_t> got(na + nb);
    (void)hipMemcpy(got.data(), dok, (na + nb) * 8, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < got.size(); ++i)
      if (got[i] != want[i]) { if (!bad) first = i; ++bad; }
    printf("case %d na %u nb %u sh %u range %llu: %zu wrong (first %zu)\n", tc, na, nb, sh, (unsigned long long)range, bad, first);
    bad_cases += bad != 0;
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dok); (void)hipFree(dov);
  }
  printf("%d bad cases (IPT %u)\n", bad_cases, TL_MERGE_IPT);
  return 0;
}
