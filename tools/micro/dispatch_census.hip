// dispatch_census.hip -- where do the workgroups of a persistent launch land?  (one-view regime of the blend kernels:
// 1130 busy tiles for 1536 resident workgroups, so the kernel's time is set by WHICH workgroups share a CU.)
//   hipcc --offload-arch=gfx950 -O3 dispatch_census.hip -o dispatch_census && ./dispatch_census
// Every workgroup records its XCC id and HW_ID (SE / SH / CU), spins ~30 us so that all of them are resident together, and the
// host prints (a) workgroups per CU: min / max / histogram, (b) the CU of consecutive blockIdx values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDS_BYTES>
__global__ __launch_bounds__(256) void census(uint32_t* out, int spin) {
  __shared__ unsigned char pad[LDS_BYTES];
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    pad[0] = (unsigned char)hw;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (pad[threadIdx.x % LDS_BYTES] == 255 && spin < 0) out[0] = 1;
}

template <int LDS_BYTES>
static int run(int grid, const char* what) {
  uint32_t* d; CK(hipMalloc(&d, 8 * grid));
  hipLaunchKernelGGL(census<LDS_BYTES>, dim3(grid), dim3(256), 0, 0, d, 70000);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> h(2 * grid); CK(hipMemcpy(h.data(), d, 8 * grid, hipMemcpyDeviceToHost));
  std::map<uint32_t, int> per_cu; std::vector<uint32_t> cu_of(grid);
  for (int b = 0; b < grid; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const uint32_t id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cu_of[b] = id; per_cu[id]++;
  }
  std::map<int, int> hist; int mn = 1 << 30, mx = 0;
  for (auto& kv : per_cu) { hist[kv.second]++; mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
  printf("%s: grid %d, LDS %d B -> %zu distinct CUs, workgroups per CU min %d max %d; histogram:", what, grid, LDS_BYTES, per_cu.size(), mn, mx);
  for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
  printf("\n  first 48 blocks -> CU id (xcc.se.sh.cu):");
  for (int b = 0; b < 48 && b < grid; ++b) printf(" %x", cu_of[b]);
  printf("\n  blocks 0, 8, 16, ... (same XCD) -> CU:");
  for (int b = 0; b < 8 * 40 && b < grid; b += 8) printf(" %x", cu_of[b] & 0xfff);
  // how many of the FIRST 256 blocks (the longest tiles) share a CU
  std::map<uint32_t, int> first; int worst = 0;
  for (int b = 0; b < 256 && b < grid; ++b) worst = std::max(worst, ++first[cu_of[b]]);
  printf("\n  the first 256 blocks sit on %zu CUs, at most %d on one CU\n", first.size(), worst);
  CK(hipFree(d));
  return 0;
}

int main() {
  if (run<25000>(1536, "render_fwd-like (6 per CU)")) return 1;
  if (run<25000>(1130, "render_fwd-like, grid = busy tiles")) return 1;
  if (run<40000>(1024, "render_bwd-like (4 per CU)")) return 1;
  if (run<40000>(512, "render_bwd-like, half grid")) return 1;
  if (run<1024>(2048, "small-LDS kernel (8 per CU)")) return 1;
  if (run<1024>(1024, "small-LDS kernel, grid 1024")) return 1;
  return 0;
}
