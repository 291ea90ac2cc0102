// How long does device memory take to get?  hipMalloc of one big block, of many small ones, hipMallocAsync from the
// default pool, and the virtual-memory API (reserve + create + map).  usage: malloc_probe [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;
  const size_t bytes = gib << 30;
  CK(hipSetDevice(0));
  void* p = nullptr;
  CK(hipMalloc(&p, 1 << 20));  // runtime up
  CK(hipFree(p));
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    CK(hipMalloc(&p, bytes));
    double t1 = now();
    CK(hipMemset(p, 1, bytes));
    CK(hipDeviceSynchronize());
    double t2 = now();
    CK(hipFree(p));
    double t3 = now();
    std::printf("hipMalloc %zu GiB: %.1f ms (%.2f ms/GiB), first touch (memset) %.1f ms, hipFree %.1f ms\n", gib, t1 - t0, (t1 - t0) / gib, t2 - t1, t3 - t2);
  }
  {
    std::vector<void*> ps(gib * 4);
    double t0 = now();
    for (auto& q : ps) CK(hipMalloc(&q, size_t(256) << 20));
    double t1 = now();
    for (auto& q : ps) CK(hipFree(q));
    double t2 = now();
    std::printf("%zu x hipMalloc 256 MiB: %.1f ms (%.2f ms/GiB), frees %.1f ms\n", ps.size(), t1 - t0, (t1 - t0) / gib, t2 - t1);
  }
  {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    double t0 = now();
    CK(hipMallocAsync(&p, bytes, s));
    CK(hipStreamSynchronize(s));
    double t1 = now();
    CK(hipFreeAsync(p, s));
    CK(hipStreamSynchronize(s));
    double t2 = now();
    CK(hipMallocAsync(&p, bytes, s));
    CK(hipStreamSynchronize(s));
    double t3 = now();
    CK(hipFreeAsync(p, s));
    CK(hipStreamSynchronize(s));
    std::printf("hipMallocAsync %zu GiB: %.1f ms, free %.1f ms, again %.1f ms\n", gib, t1 - t0, t2 - t1, t3 - t2);
  }
  {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    double t0 = now();
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, bytes, gran, nullptr, 0));
    double t1 = now();
    const size_t chunk = size_t(1) << 30;
    std::vector<hipMemGenericAllocationHandle_t> hs(gib);
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t i = 0; i < gib; ++i) {
      CK(hipMemCreate(&hs[i], chunk, &prop, 0));
      CK(hipMemMap((char*)va + i * chunk, chunk, 0, hs[i], 0));
      CK(hipMemSetAccess((char*)va + i * chunk, chunk, &acc, 1));
    }
    double t2 = now();
    CK(hipMemset(va, 1, bytes));
    CK(hipDeviceSynchronize());
    double t3 = now();
    std::printf("VMM: granularity %zu KiB, reserve %.2f ms, create+map+access %zu x 1 GiB: %.1f ms (%.2f ms/GiB), first touch %.1f ms\n", gran >> 10, t1 - t0, gib, t2 - t1,
                (t2 - t1) / gib, t3 - t2);
    for (size_t i = 0; i < gib; ++i) {
      CK(hipMemUnmap((char*)va + i * chunk, chunk));
      CK(hipMemRelease(hs[i]));
    }
    CK(hipMemAddressFree(va, bytes));
  }
  return 0;
}
