// traffic_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the blend
// kernels (the blanket x2 on FETCH_SIZE is only calibrated for wide streaming reads).
//
//   hipcc --offload-arch=gfx950 -O3 traffic_calib.hip -o traffic_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./traffic_calib      (and a second pass with WRITE_SIZE)
//
// Every kernel moves a KNOWN number of bytes from/to buffers far larger than L2 + Infinity Cache (1 GiB), each byte once:
//   calib_stream_read16   : 16 B per lane, coalesced                        (the guide's calibrated case: FETCH_SIZE = 1/2)
//   calib_stream_read4    :  4 B per lane, coalesced
//   calib_gather64        : one 64-byte aligned record per lane at a random index, read as 4 x 16 B (render_fwd / render_bwd
//                           staging: `rec[4g + 0..3]`); every record read exactly once (random permutation)
//   calib_gather64_l2     : the same gather over a 2 MiB table (L2-resident: fabric traffic should be ~0)
//   calib_stream_write16  : 16 B per lane, coalesced
//   calib_scatter36       : one 36-byte record per lane at a random slot of a packed 36-byte-stride array, as 3 x 12 B stores
//                           (render_bwd's gsr_store_partial); every slot written exactly once
//   calib_scatter36_sorted: the same stores with consecutive lanes writing consecutive slots (what staging through LDS would give)
// tools/prof_calib.sh divides the counters by these known byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void calib_stream_read16(const float4* __restrict__ in, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) sink[0] = acc;
}
__global__ void calib_stream_read4(const float* __restrict__ in, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 123.456f) sink[0] = acc;
}
__global__ void calib_gather64(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, size_t n, float* sink) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t g = idx[i];
    const float4 a = rec[4 * (size_t)g], b = rec[4 * (size_t)g + 1], c = rec[4 * (size_t)g + 2], d = rec[4 * (size_t)g + 3];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 123.456f) sink[0] = acc;
}
struct F3 { float x, y, z; };
__global__ void calib_stream_write16(float4* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void calib_scatter36(float* __restrict__ out, const uint32_t* __restrict__ idx, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    F3* p = reinterpret_cast<F3*>(out + 9 * (size_t)idx[i]);
    p[0] = F3{1.f, 2.f, 3.f}; p[1] = F3{4.f, 5.f, 6.f}; p[2] = F3{7.f, 8.f, (float)i};
  }
}

int main() {
  const size_t BYTES = 1ull << 30;                 // 1 GiB per buffer
  const size_t NREC = BYTES / 64;                  // 16 Mi records of 64 B
  const size_t NSLOT = 24u << 20;                  // 24 Mi slots of 36 B = 864 MiB
  float4* big; float* sink; uint32_t *perm, *perm36, *ident36, *perm_small; float* out36;
  CK(hipMalloc(&big, BYTES)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&perm, NREC * 4)); CK(hipMalloc(&perm36, NSLOT * 4));
  CK(hipMalloc(&ident36, NSLOT * 4)); CK(hipMalloc(&out36, NSLOT * 36)); CK(hipMalloc(&perm_small, NREC * 4));
  CK(hipMemset(big, 0, BYTES));
  {
    std::mt19937_64 rng(7);
    std::vector<uint32_t> h(NREC); std::iota(h.begin(), h.end(), 0u); std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(perm, h.data(), NREC * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v &= (32768u - 1u);        // 32 Ki records = 2 MiB table
    CK(hipMemcpy(perm_small, h.data(), NREC * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> s(NSLOT); std::iota(s.begin(), s.end(), 0u);
    CK(hipMemcpy(ident36, s.data(), NSLOT * 4, hipMemcpyHostToDevice));
    std::shuffle(s.begin(), s.end(), rng);
    CK(hipMemcpy(perm36, s.data(), NSLOT * 4, hipMemcpyHostToDevice));
  }
  const int grid = 256 * 8;
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(calib_stream_read16, dim3(grid), dim3(256), 0, 0, big, BYTES / 16, sink);
  hipLaunchKernelGGL(calib_stream_read4, dim3(grid), dim3(256), 0, 0, (const float*)big, BYTES / 4, sink);
  hipLaunchKernelGGL(calib_gather64, dim3(grid), dim3(256), 0, 0, big, perm, NREC, sink);
  hipLaunchKernelGGL(calib_gather64, dim3(grid), dim3(256), 0, 0, big, perm_small, NREC, sink);   // L2-resident table (second launch of the name)
  hipLaunchKernelGGL(calib_stream_write16, dim3(grid), dim3(256), 0, 0, big, BYTES / 16);
  hipLaunchKernelGGL(calib_scatter36, dim3(grid), dim3(256), 0, 0, out36, perm36, NSLOT);
  hipLaunchKernelGGL(calib_scatter36, dim3(grid), dim3(256), 0, 0, out36, ident36, NSLOT);        // sorted slots (second launch of the name)
  CK(hipDeviceSynchronize());
  // known bytes, in launch order (index bytes listed separately: they are coalesced 4-byte reads)
  printf("KNOWN calib_stream_read16 0 read %zu write 0\n", BYTES);
  printf("KNOWN calib_stream_read4 0 read %zu write 0\n", BYTES);
  printf("KNOWN calib_gather64 0 read %zu write 0 index %zu\n", NREC * 64, NREC * 4);
  printf("KNOWN calib_gather64 1 read %zu write 0 index %zu  (2 MiB table: L2 hits)\n", (size_t)0, NREC * 4);
  printf("KNOWN calib_stream_write16 0 read 0 write %zu\n", BYTES);
  printf("KNOWN calib_scatter36 0 read 0 write %zu index %zu\n", NSLOT * 36, NSLOT * 4);
  printf("KNOWN calib_scatter36 1 read 0 write %zu index %zu  (sorted slots)\n", NSLOT * 36, NSLOT * 4);
  return 0;
}
