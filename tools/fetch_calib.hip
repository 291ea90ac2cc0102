// Calibrates rocprofv3's FETCH_SIZE on gfx950 for the two access shapes of this library: wide coalesced loads and
// scattered 8-byte loads (one 64-bit word per lane, every lane in another cache line) -- what the MIN_DISTANCE kernels
// do.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o c -- /tmp/fetch_calib
// Each kernel reads a known number of useful bytes (printed); FETCH_SIZE (KiB) per kernel comes from the counter file.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ void coalesced_read(const uint64_t* __restrict__ a, uint64_t n, uint64_t* __restrict__ out) {
  uint64_t s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 0x123456789ull) out[0] = s;
}
// lane i reads word (hash(i) mod lines) * words_per_line: every load of a wavefront in another line, no line twice
__global__ void scattered_read(const uint64_t* __restrict__ a, uint64_t lines, uint32_t words_per_line, uint64_t loads,
                               uint64_t* __restrict__ out) {
  uint64_t s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t line = (i * 0x9E3779B97F4A7C15ull >> 20) % lines;  // odd multiplier: a permutation-like walk
    s += a[line * words_per_line];
  }
  if (s == 0x123456789ull) out[0] = s;
}

// the same with non-temporal loads (do they fetch less than a 128-byte line?)
__global__ void scattered_read_nt(const uint64_t* __restrict__ a, uint64_t lines, uint32_t words_per_line, uint64_t loads,
                                  uint64_t* __restrict__ out) {
  uint64_t s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t line = (i * 0x9E3779B97F4A7C15ull >> 20) % lines;
    s += __builtin_nontemporal_load(&a[line * words_per_line]);
  }
  if (s == 0x123456789ull) out[0] = s;
}
// ... and with agent-scope atomic loads (sc1: what the state polls of the sparse levels use)
__global__ void scattered_read_sc1(const uint64_t* __restrict__ a, uint64_t lines, uint32_t words_per_line, uint64_t loads,
                                   uint64_t* __restrict__ out) {
  uint64_t s = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < loads; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t line = (i * 0x9E3779B97F4A7C15ull >> 20) % lines;
    s += __hip_atomic_load(&a[line * words_per_line], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (s == 0x123456789ull) out[0] = s;
}

int main() {
  const uint64_t bytes = 16ull << 30;
  uint64_t *a = nullptr, *out = nullptr;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  (void)hipMemset(a, 1, bytes);
  (void)hipDeviceSynchronize();
  const uint64_t n = bytes / 8;
  hipLaunchKernelGGL(coalesced_read, dim3(8192), dim3(256), 0, 0, a, n, out);
  (void)hipDeviceSynchronize();
  printf("coalesced_read: %llu useful bytes\n", (unsigned long long)bytes);
  const uint64_t loads = 1ull << 28;  // 268 M scattered 8-byte loads
  for (uint32_t line_bytes : {64u, 128u, 256u}) {
    hipLaunchKernelGGL(scattered_read, dim3(8192), dim3(256), 0, 0, a, bytes / line_bytes, line_bytes / 8, loads, out);
    (void)hipDeviceSynchronize();
    printf("scattered_read (one word per %u-byte block): %llu loads, %llu useful bytes\n", line_bytes, (unsigned long long)loads,
           (unsigned long long)(loads * 8));
  }
  hipLaunchKernelGGL(scattered_read_nt, dim3(8192), dim3(256), 0, 0, a, bytes / 128, 16u, loads, out);
  (void)hipDeviceSynchronize();
  printf("scattered_read_nt: %llu loads\n", (unsigned long long)loads);
  hipLaunchKernelGGL(scattered_read_sc1, dim3(8192), dim3(256), 0, 0, a, bytes / 128, 16u, loads / 16, out);
  (void)hipDeviceSynchronize();
  printf("scattered_read_sc1: %llu loads\n", (unsigned long long)(loads / 16));
  return 0;
}
