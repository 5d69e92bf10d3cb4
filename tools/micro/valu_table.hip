// valu_table.hip -- issue cost of the instructions the blend loops are made of, on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 valu_table.hip -o valu_table && ./valu_table [out.json]
//
// valu_peak.hip showed: SIMD peak = one wave-64 VALU op per ~2.2 cycles (needs >= 2 waves per SIMD; ONE wave issues a VALU
// op every ~4.4 cycles), v_exp_f32 quarter rate, DPP adds half rate -- and v_cndmask_b32 at ~1/10 rate, which this
// table re-measures in several encodings next to the alternatives (masked DPP, permlane swaps, arithmetic selects).
//
// Method: 64-instruction straight-line blocks (8 chains), ITER trips, 256-thread workgroups, 8 per CU (full occupancy: 8 waves
// per SIMD) and 1 per CU.  Cost = SIMD-cycles per wave-instruction = 1024 SIMDs x span x f / (waves x instructions), with f the
// effective clock measured in the same launch (slowest wave's s_memtime delta / span).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(S) S S S S S S S S
#define CH8(OP, TAIL)           \
  OP " %0, %0" TAIL "\n\t"      \
  OP " %1, %1" TAIL "\n\t"      \
  OP " %2, %2" TAIL "\n\t"      \
  OP " %3, %3" TAIL "\n\t"      \
  OP " %4, %4" TAIL "\n\t"      \
  OP " %5, %5" TAIL "\n\t"      \
  OP " %6, %6" TAIL "\n\t"      \
  OP " %7, %7" TAIL "\n\t"
// unary: OP dst, src
#define UN8(OP, TAIL)           \
  OP " %0, %0" TAIL "\n\t" OP " %1, %1" TAIL "\n\t" OP " %2, %2" TAIL "\n\t" OP " %3, %3" TAIL "\n\t" \
  OP " %4, %4" TAIL "\n\t" OP " %5, %5" TAIL "\n\t" OP " %6, %6" TAIL "\n\t" OP " %7, %7" TAIL "\n\t"

#define DPPX(OP, CTRL)                                  \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\t" OP " %2, %2, %2 " CTRL "\n\t" OP " %3, %3, %3 " CTRL "\n\t" \
  OP " %4, %4, %4 " CTRL "\n\t" OP " %5, %5, %5 " CTRL "\n\t" OP " %6, %6, %6 " CTRL "\n\t" OP " %7, %7, %7 " CTRL "\n\t"
#define MOVDPP(CTRL)                                    \
  "v_mov_b32_dpp %0, %1 " CTRL "\n\t" "v_mov_b32_dpp %1, %2 " CTRL "\n\t" "v_mov_b32_dpp %2, %3 " CTRL "\n\t" "v_mov_b32_dpp %3, %4 " CTRL "\n\t" \
  "v_mov_b32_dpp %4, %5 " CTRL "\n\t" "v_mov_b32_dpp %5, %6 " CTRL "\n\t" "v_mov_b32_dpp %6, %7 " CTRL "\n\t" "v_mov_b32_dpp %7, %0 " CTRL "\n\t"
#define C3(OP) OP " %0, %8, %9\n\t" OP " %1, %8, %9\n\t" OP " %2, %8, %9\n\t" OP " %3, %8, %9\n\t" OP " %4, %8, %9\n\t" OP " %5, %8, %9\n\t" OP " %6, %8, %9\n\t" OP " %7, %8, %9\n\t"
#define SWAP4(OP) OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t" OP " %0, %1\n\t" OP " %2, %3\n\t" OP " %4, %5\n\t" OP " %6, %7\n\t"

#define NKINDS 30
static const char* kind_name[NKINDS] = {
  "v_mov_b32 dst, const", "v_add_f32", "v_mul_f32", "v_fma_f32 (3 distinct VGPR sources)", "v_fma_f32 a*a+c (2 distinct VGPRs)",
  "v_fmac_f32 (VOP2)", "v_max_f32", "v_and_b32", "v_add_u32", "v_cndmask_b32 e32 vcc (dst = src0)",
  "v_cndmask_b32_e64 SGPR-pair mask (dst = src0)", "v_cndmask_b32_e64 SGPR-pair mask, const sources", "v_cmp_gt_f32 vcc ; v_cndmask vcc (per instruction)",
  "v_cmp_gt_f32 vcc", "v_cmp_gt_f32_e64 SGPR pair dst", "v_add_f32_dpp quad_perm:[1,0,3,2]", "v_add_f32_dpp row_shr:4 bank_mask:0xa",
  "v_add_f32_dpp row_ror:8", "v_add_f32_dpp row_bcast:15 row_mask:0xa", "v_mov_b32_dpp quad_perm", "v_permlane32_swap (per instruction)",
  "v_permlane16_swap (per instruction)", "v_exp_f32", "v_rcp_f32", "v_mul_f32 ; v_fma_f32 arithmetic select (per instruction)",
  "v_readfirstlane_b32", "v_mbcnt_lo_u32_b32", "v_min_f32 ; v_cmp_ge_f32 (blend test pattern, per instruction)", "v_pk_fma_f32", "v_pk_mul_f32"};

template <int KIND>
__global__ __launch_bounds__(256) void k_table(int iters, float seed, unsigned long long mask, unsigned long long* cycles, float* sink) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.001f;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
  const float2v pm = {m, m}, pc = {c, c};
  int sg = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#define BODY(STR, ...) asm volatile(REP8(STR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(mask) : __VA_ARGS__)
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) BODY("v_mov_b32 %0, %8\n\tv_mov_b32 %1, %8\n\tv_mov_b32 %2, %8\n\tv_mov_b32 %3, %8\n\tv_mov_b32 %4, %8\n\tv_mov_b32 %5, %8\n\tv_mov_b32 %6, %8\n\tv_mov_b32 %7, %8\n\t", "memory");
    if (KIND == 1) BODY(CH8("v_add_f32", ", %8"), "memory");
    if (KIND == 2) BODY(CH8("v_mul_f32", ", %8"), "memory");
    if (KIND == 3) BODY(CH8("v_fma_f32", ", %8, %9"), "memory");
    if (KIND == 4) BODY("v_fma_f32 %0, %0, %0, %8\n\tv_fma_f32 %1, %1, %1, %8\n\tv_fma_f32 %2, %2, %2, %8\n\tv_fma_f32 %3, %3, %3, %8\n\tv_fma_f32 %4, %4, %4, %8\n\tv_fma_f32 %5, %5, %5, %8\n\tv_fma_f32 %6, %6, %6, %8\n\tv_fma_f32 %7, %7, %7, %8\n\t", "memory");
    if (KIND == 5) BODY(C3("v_fmac_f32"), "memory");
    if (KIND == 6) BODY(CH8("v_max_f32", ", %8"), "memory");
    if (KIND == 7) BODY(CH8("v_and_b32", ", %8"), "memory");
    if (KIND == 8) BODY(CH8("v_add_u32", ", %8"), "memory");
    if (KIND == 9) BODY(CH8("v_cndmask_b32", ", %8, vcc"), "memory", "vcc");
    if (KIND == 10) BODY(CH8("v_cndmask_b32_e64", ", %8, %10"), "memory");
    if (KIND == 11) BODY("v_cndmask_b32_e64 %0, %8, %9, %10\n\tv_cndmask_b32_e64 %1, %8, %9, %10\n\tv_cndmask_b32_e64 %2, %8, %9, %10\n\tv_cndmask_b32_e64 %3, %8, %9, %10\n\tv_cndmask_b32_e64 %4, %8, %9, %10\n\tv_cndmask_b32_e64 %5, %8, %9, %10\n\tv_cndmask_b32_e64 %6, %8, %9, %10\n\tv_cndmask_b32_e64 %7, %8, %9, %10\n\t", "memory");
    if (KIND == 12) BODY("v_cmp_gt_f32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %9, vcc\n\tv_cmp_gt_f32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cmp_gt_f32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cmp_gt_f32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t", "memory", "vcc");
    if (KIND == 13) BODY("v_cmp_gt_f32 vcc, %0, %8\n\tv_cmp_gt_f32 vcc, %1, %8\n\tv_cmp_gt_f32 vcc, %2, %8\n\tv_cmp_gt_f32 vcc, %3, %8\n\tv_cmp_gt_f32 vcc, %4, %8\n\tv_cmp_gt_f32 vcc, %5, %8\n\tv_cmp_gt_f32 vcc, %6, %8\n\tv_cmp_gt_f32 vcc, %7, %8\n\t", "memory", "vcc");
    if (KIND == 14) BODY("v_cmp_gt_f32_e64 s[20:21], %0, %8\n\tv_cmp_gt_f32_e64 s[22:23], %1, %8\n\tv_cmp_gt_f32_e64 s[24:25], %2, %8\n\tv_cmp_gt_f32_e64 s[26:27], %3, %8\n\tv_cmp_gt_f32_e64 s[20:21], %4, %8\n\tv_cmp_gt_f32_e64 s[22:23], %5, %8\n\tv_cmp_gt_f32_e64 s[24:25], %6, %8\n\tv_cmp_gt_f32_e64 s[26:27], %7, %8\n\t", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    if (KIND == 15) BODY(DPPX("v_add_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory");
    if (KIND == 16) BODY(DPPX("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xa"), "memory");
    if (KIND == 17) BODY(DPPX("v_add_f32_dpp", "row_ror:8 row_mask:0xf bank_mask:0xf"), "memory");
    if (KIND == 18) BODY(DPPX("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf"), "memory");
    if (KIND == 19) BODY(MOVDPP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory");
    if (KIND == 20) BODY(SWAP4("v_permlane32_swap_b32"), "memory");
    if (KIND == 21) BODY(SWAP4("v_permlane16_swap_b32"), "memory");
    if (KIND == 22) BODY(UN8("v_exp_f32", ""), "memory");
    if (KIND == 23) BODY(UN8("v_rcp_f32", ""), "memory");
    if (KIND == 24) BODY("v_mul_f32 %0, %8, %0\n\tv_fma_f32 %0, %9, %1, %0\n\tv_mul_f32 %2, %8, %2\n\tv_fma_f32 %2, %9, %3, %2\n\tv_mul_f32 %4, %8, %4\n\tv_fma_f32 %4, %9, %5, %4\n\tv_mul_f32 %6, %8, %6\n\tv_fma_f32 %6, %9, %7, %6\n\t", "memory");
    if (KIND == 25) asm volatile(REP8("v_readfirstlane_b32 %8, %0\n\tv_readfirstlane_b32 %8, %1\n\tv_readfirstlane_b32 %8, %2\n\tv_readfirstlane_b32 %8, %3\n\tv_readfirstlane_b32 %8, %4\n\tv_readfirstlane_b32 %8, %5\n\tv_readfirstlane_b32 %8, %6\n\tv_readfirstlane_b32 %8, %7\n\t")
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(sg) : : "memory");
    if (KIND == 26) BODY("v_mbcnt_lo_u32_b32 %0, -1, %0\n\tv_mbcnt_lo_u32_b32 %1, -1, %1\n\tv_mbcnt_lo_u32_b32 %2, -1, %2\n\tv_mbcnt_lo_u32_b32 %3, -1, %3\n\tv_mbcnt_lo_u32_b32 %4, -1, %4\n\tv_mbcnt_lo_u32_b32 %5, -1, %5\n\tv_mbcnt_lo_u32_b32 %6, -1, %6\n\tv_mbcnt_lo_u32_b32 %7, -1, %7\n\t", "memory");
    if (KIND == 27) BODY("v_min_f32 %0, %8, %0\n\tv_cmp_ge_f32 vcc, %0, %9\n\tv_min_f32 %1, %8, %1\n\tv_cmp_ge_f32 vcc, %1, %9\n\tv_min_f32 %2, %8, %2\n\tv_cmp_ge_f32 vcc, %2, %9\n\tv_min_f32 %3, %8, %3\n\tv_cmp_ge_f32 vcc, %3, %9\n\t", "memory", "vcc");
    if (KIND == 28) asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\tv_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t")
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc) : "memory");
    if (KIND == 29) asm volatile(REP8("v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\tv_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_mul_f32 %6, %6, %8\n\tv_pk_mul_f32 %7, %7, %8\n\t")
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc) : "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)sg;
  if (s == 123.456f) sink[0] = s;
}

template <int KIND>
static int run_kind(int cus, FILE* js, bool& first) {
  const int iters = 1500;
  unsigned long long* cyc; float* sink;
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * 4 * cus * 8)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  double cost[2] = {0, 0}, ghz[2] = {0, 0}, wavecyc[2] = {0, 0};
  int idx = 0;
  for (int k : {1, 8}) {
    const int grid = cus * k;
    float best_ms = 1e9; unsigned long long worst = 0, med = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_table<KIND>, dim3(grid), dim3(256), 0, 0, iters, 1.0f, 0x5555555555555555ull, cyc, sink);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best_ms) {
        best_ms = ms;
        std::vector<unsigned long long> h(4 * grid);
        CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end()); worst = h.back(); med = h[h.size() / 2];
      }
    }
    const double instr = 64.0 * iters;
    const double f = (double)worst / (best_ms * 1e-3);                 // effective clock
    cost[idx] = cus * 4.0 * (best_ms * 1e-3) * f / (grid * 4.0 * instr);   // SIMD-cycles per wave-instruction (chip average)
    ghz[idx] = f / 1e9; wavecyc[idx] = (double)med / instr;
    ++idx;
  }
  printf("%-62s  1 wave/SIMD: %5.2f cyc/instr (wave view %5.2f)   8 waves/SIMD: %5.2f SIMD-cyc/instr  [%.2f / %.2f GHz]\n", kind_name[KIND], cost[0], wavecyc[0], cost[1], ghz[0], ghz[1]);
  if (js) {
    fprintf(js, "%s\n  {\"instr\": \"%s\", \"cycles_per_instr_one_wave_per_simd\": %.3f, \"simd_cycles_per_instr_8_waves_per_simd\": %.3f}", first ? "" : ",", kind_name[KIND], cost[0], cost[1]);
    first = false;
  }
  CK(hipFree(cyc)); CK(hipFree(sink));
  return 0;
}

template <int K> struct Runner { static int go(int cus, FILE* js, bool& first) { if (run_kind<K>(cus, js, first)) return 1; return Runner<K + 1>::go(cus, js, first); } };
template <> struct Runner<NKINDS> { static int go(int, FILE*, bool&) { return 0; } };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
  bool first = true;
  if (js) fprintf(js, "{\"source\": \"tools/micro/valu_table.hip\", \"cus\": %d, \"rows\": [", cus);
  if (Runner<0>::go(cus, js, first)) return 1;
  if (js) { fprintf(js, "\n]}\n"); fclose(js); }
  return 0;
}
