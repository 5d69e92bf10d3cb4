// valu_peak.hip -- what is the VALU issue peak of one gfx950 SIMD?  (settles the "1 op / 4 cycles" vs "1 op / 2 cycles"
// question of VERDICT r01 #4: the blend kernels' slot model in bench.py uses the number measured here.)
//
//   hipcc --offload-arch=gfx950 -O3 valu_peak.hip -o valu_peak && ./valu_peak [json-out]
//
// Each wave runs ITER trips of a straight-line block of 64 instructions of one kind on CH independent register chains
// (inline asm, so the instruction mix is exactly what is written).  256-thread workgroups (one wave per SIMD), k
// workgroups per CU for k = 1, 2, 4, 8  ->  k waves per SIMD.  Every wave brackets its loop with s_memtime
// (shader cycles); the kernel's span is also timed with HIP events (-> effective clock).  Reported:
//   wave-instructions per cycle per SIMD = k * 64 * ITER / (cycles of the slowest wave)
//   and the chip-wide lane-instruction rate from the event time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(S) S S S S S S S S
// 8 independent chains, 8 instructions each per block repetition -> 64 instructions per block
#define FMA8                                   \
  "v_fma_f32 %0, %0, %8, %9\n\t"               \
  "v_fma_f32 %1, %1, %8, %9\n\t"               \
  "v_fma_f32 %2, %2, %8, %9\n\t"               \
  "v_fma_f32 %3, %3, %8, %9\n\t"               \
  "v_fma_f32 %4, %4, %8, %9\n\t"               \
  "v_fma_f32 %5, %5, %8, %9\n\t"               \
  "v_fma_f32 %6, %6, %8, %9\n\t"               \
  "v_fma_f32 %7, %7, %8, %9\n\t"
#define ADD8                                   \
  "v_add_f32 %0, %0, %8\n\t"                   \
  "v_add_f32 %1, %1, %8\n\t"                   \
  "v_add_f32 %2, %2, %8\n\t"                   \
  "v_add_f32 %3, %3, %8\n\t"                   \
  "v_add_f32 %4, %4, %8\n\t"                   \
  "v_add_f32 %5, %5, %8\n\t"                   \
  "v_add_f32 %6, %6, %8\n\t"                   \
  "v_add_f32 %7, %7, %8\n\t"
#define EXP8                                   \
  "v_exp_f32 %0, %0\n\t"                       \
  "v_exp_f32 %1, %1\n\t"                       \
  "v_exp_f32 %2, %2\n\t"                       \
  "v_exp_f32 %3, %3\n\t"                       \
  "v_exp_f32 %4, %4\n\t"                       \
  "v_exp_f32 %5, %5\n\t"                       \
  "v_exp_f32 %6, %6\n\t"                       \
  "v_exp_f32 %7, %7\n\t"
#define DPP8                                                                             \
  "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CND8                                   \
  "v_cndmask_b32 %0, %0, %8, vcc\n\t"          \
  "v_cndmask_b32 %1, %1, %8, vcc\n\t"          \
  "v_cndmask_b32 %2, %2, %8, vcc\n\t"          \
  "v_cndmask_b32 %3, %3, %8, vcc\n\t"          \
  "v_cndmask_b32 %4, %4, %8, vcc\n\t"          \
  "v_cndmask_b32 %5, %5, %8, vcc\n\t"          \
  "v_cndmask_b32 %6, %6, %8, vcc\n\t"          \
  "v_cndmask_b32 %7, %7, %8, vcc\n\t"
// the blend backward's flavour: fma / mul / cndmask / dpp-add / exp in about its proportions (per 8: 3 fma, 2 mul, 1 cndmask, 1 dpp, 1 of {exp | rcp})
#define MIX8                                   \
  "v_fma_f32 %0, %0, %8, %9\n\t"               \
  "v_mul_f32 %1, %1, %8\n\t"                   \
  "v_fma_f32 %2, %2, %8, %9\n\t"               \
  "v_cndmask_b32 %3, %3, %8, vcc\n\t"          \
  "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_mul_f32 %5, %5, %8\n\t"                   \
  "v_fma_f32 %6, %6, %8, %9\n\t"               \
  "v_exp_f32 %7, %7\n\t"
// packed fp32: 8 chains of float2 -> 8 instructions, 16 lane-flops x 2
#define PK8                                    \
  "v_pk_fma_f32 %0, %0, %8, %9\n\t"            \
  "v_pk_fma_f32 %1, %1, %8, %9\n\t"            \
  "v_pk_fma_f32 %2, %2, %8, %9\n\t"            \
  "v_pk_fma_f32 %3, %3, %8, %9\n\t"            \
  "v_pk_fma_f32 %4, %4, %8, %9\n\t"            \
  "v_pk_fma_f32 %5, %5, %8, %9\n\t"            \
  "v_pk_fma_f32 %6, %6, %8, %9\n\t"            \
  "v_pk_fma_f32 %7, %7, %8, %9\n\t"
// ONE dependent chain (latency): 64 dependent fmas
#define DEP8 REP8("v_fma_f32 %0, %0, %8, %9\n\t")

template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(int iters, float seed, unsigned long long* cycles, float* sink) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.001f;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
  const float2v pm = {m, m}, pc = {c, c};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();   // s_memtime
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) asm volatile(REP8(FMA8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    if (KIND == 1) asm volatile(REP8(ADD8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c));
    if (KIND == 2) asm volatile(REP8(EXP8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    if (KIND == 3) asm volatile("s_nop 1\n\t" REP8(DPP8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    if (KIND == 4) asm volatile(REP8(CND8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
    if (KIND == 5) asm volatile("s_nop 1\n\t" REP8(MIX8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc");
    if (KIND == 6) asm volatile(REP8(PK8) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));
    if (KIND == 7) asm volatile(REP8(DEP8) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (s == 123.456f) sink[0] = s;
}

struct Kind { const char* name; int lanes_per_instr; };
static const Kind kinds[8] = {{"v_fma_f32 x8 chains", 1}, {"v_add_f32 x8 chains", 1}, {"v_exp_f32 x8 chains", 1}, {"v_add_f32_dpp quad_perm x8 chains", 1},
                              {"v_cndmask_b32 x8 chains", 1}, {"blend-bwd mix (3 fma, 2 mul, cndmask, dpp-add, exp)", 1},
                              {"v_pk_fma_f32 x8 chains (2 fp32 per lane per op)", 2}, {"v_fma_f32 one dependent chain", 1}};

template <int KIND>
static int run_kind(int cus, FILE* js, bool& first) {
  const int iters = 4000;
  unsigned long long* cyc; float* sink;
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * 4 * cus * 8)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int k : {1, 2, 4, 8}) {
    const int grid = cus * k;
    float best_ms = 1e9; unsigned long long worst = 0, med = 0;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(valu_kernel<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, cyc, sink);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best_ms) {
        best_ms = ms;
        std::vector<unsigned long long> h(4 * grid);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end()); worst = h.back(); med = h[h.size() / 2];
      }
    }
    const double instr_per_wave = 64.0 * iters;
    const double ipc_simd = k * instr_per_wave / (double)med;          // wave-instructions per shader cycle per SIMD (median wave)
    const double chip_lane_rate = (double)grid * 4 * instr_per_wave * 64.0 * kinds[KIND].lanes_per_instr / (best_ms * 1e-3);
    const double clock_ghz = (double)worst / (best_ms * 1e-3) / 1e9;   // slowest wave's cycles over the kernel's span
    printf("%-52s waves/SIMD %d: %.3f wave-instr/cycle/SIMD (median wave %llu cyc, slowest %llu), %.2f T lane-ops/s chip, span %.3f ms, ~%.2f GHz\n",
           kinds[KIND].name, k, ipc_simd, med, worst, chip_lane_rate / 1e12, best_ms, clock_ghz);
    if (js) {
      fprintf(js, "%s\n  {\"kind\": \"%s\", \"waves_per_simd\": %d, \"wave_instr_per_cycle_per_simd\": %.4f, \"chip_T_lane_ops_per_s\": %.3f, "
                  "\"span_ms\": %.4f, \"effective_clock_GHz\": %.3f}", first ? "" : ",", kinds[KIND].name, k, ipc_simd, chip_lane_rate / 1e12, best_ms, clock_ghz);
      first = false;
    }
  }
  CK(hipFree(cyc)); CK(hipFree(sink));
  return 0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  bool first = true;
  if (js) fprintf(js, "{\"device\": \"%s\", \"cus\": %d, \"source\": \"tools/micro/valu_peak.hip\", \"rows\": [", prop.name, cus);
  if (run_kind<0>(cus, js, first)) return 1;
  if (run_kind<1>(cus, js, first)) return 1;
  if (run_kind<2>(cus, js, first)) return 1;
  if (run_kind<3>(cus, js, first)) return 1;
  if (run_kind<4>(cus, js, first)) return 1;
  if (run_kind<5>(cus, js, first)) return 1;
  if (run_kind<6>(cus, js, first)) return 1;
  if (run_kind<7>(cus, js, first)) return 1;
  if (js) { fprintf(js, "\n]}\n"); fclose(js); }
  return 0;
}
