// Micro-benchmark: throughput of device-scope atomics on a small counter array (tile histogram / slot
// allocation pattern of an atomic counting sort).  hipcc --offload-arch=gfx950 -O3 atomics_bench.hip -o atomics_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void count_noret(const uint32_t* __restrict__ key, uint32_t n, uint32_t* cnt) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&cnt[key[i]], 1u);
}
__global__ void place_ret(const uint32_t* __restrict__ key, uint32_t n, uint32_t* cnt, uint32_t* __restrict__ slot) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot[i] = atomicAdd(&cnt[key[i]], 1u);
}
__global__ void place_scatter(const uint32_t* __restrict__ key, uint32_t n, uint32_t* cnt, const uint32_t* __restrict__ base,
                              uint64_t* __restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t k = key[i]; uint32_t s = atomicAdd(&cnt[k], 1u); out[base[k] + s] = ((uint64_t)k << 32) | i; }
}

int main() {
  const uint32_t n = 1000000, T = 2500;
  std::vector<uint32_t> h(n);
  std::mt19937 rng(1);
  // Gaussian-major emission: each "Gaussian" touches a small rect of tiles (row-major), centres random
  uint32_t e = 0;
  while (e < n) {
    int cx = rng() % 50, cy = rng() % 50, w = 1 + rng() % 4, hh = 1 + rng() % 4;
    for (int y = cy; y < std::min(50, cy + hh) && e < n; ++y)
      for (int x = cx; x < std::min(50, cx + w) && e < n; ++x) h[e++] = y * 50 + x;
  }
  std::vector<uint32_t> hc(T, 0), hb(T, 0);
  for (auto k : h) hc[k]++;
  for (uint32_t t = 1; t < T; ++t) hb[t] = hb[t - 1] + hc[t - 1];
  uint32_t *key, *cnt, *slot, *base; uint64_t* out;
  CK(hipMalloc(&key, n * 4)); CK(hipMalloc(&cnt, T * 4)); CK(hipMalloc(&slot, n * 4)); CK(hipMalloc(&base, T * 4)); CK(hipMalloc(&out, n * 8));
  CK(hipMemcpy(key, h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(base, hb.data(), T * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 10; ++rep) {
      CK(hipMemset(cnt, 0, T * 4));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(count_noret, dim3((n + 255) / 256), dim3(256), 0, 0, key, n, cnt);
      if (mode == 1) hipLaunchKernelGGL(place_ret, dim3((n + 255) / 256), dim3(256), 0, 0, key, n, cnt, slot);
      if (mode == 2) hipLaunchKernelGGL(place_scatter, dim3((n + 255) / 256), dim3(256), 0, 0, key, n, cnt, base, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("mode %d (%s): %.1f us for %u atomics over %u counters\n", mode,
           mode == 0 ? "count, no return" : mode == 1 ? "slot, with return" : "slot + scattered 8 B store", best * 1000.f, n, T);
  }
  return 0;
}
