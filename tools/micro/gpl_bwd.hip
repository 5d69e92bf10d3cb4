// gpl_bwd.hip -- PROTOTYPE of a Gaussian-per-lane blend backward: lane = list entry, the tile's 256 pixels stream
// through the wave lane to lane (one `wave_shl:1` DPP move per state register and step), every lane keeps its entry's nine gradient sums
// in registers and stores ONE record at the end -- no 24-issue wave reduction, no per-visit LDS round trip, no ballot branch.
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize gpl_bwd.hip -o gpl_bwd && ./gpl_bwd [tiles] [entries_per_tile]
//
// Shape of the prototype (what a product kernel of this design would also have to do):
//   * a wave owns one BUCKET of 64 consecutive entries of one tile's depth-sorted list.  Back-to-front as the reference: a pixel's state
//     {T behind the bucket, colour-behind . dL, dL/dcolour (3), -T_final bg . dL, pixel index | last contributor} enters at lane 63 (the
//     bucket's deepest entry) and leaves at lane 0.  Buckets of a tile are independent here because the state each bucket starts from
//     comes from a CHECKPOINT array (per bucket and pixel: what a forward pass would have to write every 64 list positions, 32 bytes
//     per pixel and bucket -- 53 MB per 800 x 800 view at the benchmark's list lengths, written once and read once);
//   * the 256-pixel checkpoint of the bucket is copied to the wave's LDS block (8 KiB) up front; lane 63 reads one pixel's state per step;
//   * per step and lane: the same arithmetic as render_bwd's contributing visit (alpha re-evaluation, T / (1 - alpha), the colour-dot
//     recurrence, dL/dalpha, the nine partial products) -- predicated, not branched: a lane cannot skip, its neighbours do not;
//   * 256 + 63 steps per bucket (pipeline fill and drain), ALL 256 x 64 pixel-entry pairs evaluated: there is no quad culling, no
//     alpha-box, no trimming behind the deepest contributor.
// The program checks the kernel's nine sums against a plain per-pixel replay on the host (fp64) and prints time, pairs evaluated and the
// rate; compare with render_bwd on the same pair count (profiles/r04_gaussian_per_lane.txt).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct PixState { float T, acc, dL0, dL1, dL2, ntfbg; uint32_t pix_last; float pad; };   // 32 bytes: T = transmittance BEHIND the bucket
struct Entry { float mx, my, A, B, C, o, r, g, b, pad0, pad1, pad2; };                   // conic pre-scaled by -log2(e)/2, -log2(e), -log2(e)/2

#define ALPHA_MIN (1.0f / 255.0f)
#define ALPHA_MAX 0.99f

// lane i <- lane i + 1; lane 63 keeps `incoming` (the DPP source is invalid there and bound_ctrl is off: the old value stays)
__device__ __forceinline__ float shl1(float incoming, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, incoming), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t shl1u(uint32_t incoming, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)incoming, (int)v, 0x130, 0xf, 0xf, false);
}

__global__ __launch_bounds__(256) void gpl_bwd_kernel(const Entry* __restrict__ entries, const PixState* __restrict__ chk, float* __restrict__ out,
                                                      int n_buckets, int tile_x0, int tile_y0) {
  __shared__ float4 s_chk[4][256 * 2];                      // a wave's 256 pixel states
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bucket = blockIdx.x * 4 + wv;
  if (bucket >= n_buckets) return;
  {  // checkpoint -> LDS (coalesced: 8 KiB per wave)
    const float4* src = reinterpret_cast<const float4*>(chk + (size_t)bucket * 256);
#pragma unroll
    for (int k = 0; k < 8; ++k) s_chk[wv][k * 64 + lane] = src[k * 64 + lane];
  }
  const Entry e = entries[(size_t)bucket * 64 + lane];      // lane 63 = the bucket's deepest entry
  const int pos = lane;                                     // list position inside the bucket (a product kernel adds the bucket's base)
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0, s8 = 0;
  // state registers; "empty" = pix_last with the valid bit (bit 31) clear
  float T = 0.f, acc = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f, ntf = 0.f, lastc = 0.f, lasta = 0.f;
  uint32_t pl = 0u;
  const float4* __restrict__ my = s_chk[wv];
  float4 n0 = my[0], n1 = my[1];                            // (every lane reads: the broadcast costs what one lane's read costs)
  for (int step = 0; step < 256 + 63; ++step) {
    // ---- shift the pipeline by one lane; the new pixel enters at lane 63
    const bool fresh = step < 256;
    const float iT = fresh ? n0.x : 0.f, iacc = fresh ? n0.y : 0.f, id0 = fresh ? n0.z : 0.f, id1 = fresh ? n0.w : 0.f;
    const float id2 = fresh ? n1.x : 0.f, intf = fresh ? n1.y : 0.f;
    const uint32_t ipl = fresh ? (__float_as_uint(n1.z) | 0x80000000u) : 0u;
    T = shl1(iT, T); acc = shl1(iacc, acc); d0 = shl1(id0, d0); d1 = shl1(id1, d1); d2 = shl1(id2, d2); ntf = shl1(intf, ntf);
    lastc = shl1(0.f, lastc); lasta = shl1(0.f, lasta);      // the pending (alpha, colour . dL) of the entry behind travel with the pixel
    pl = shl1u(ipl, pl);
    if (step + 1 < 256) { n0 = my[2 * (step + 1)]; n1 = my[2 * (step + 1) + 1]; }
    // ---- this lane's entry against the pixel it now holds
    const float pxf = (float)(tile_x0 + (int)(pl & 15u)), pyf = (float)(tile_y0 + (int)((pl >> 4) & 15u));
    const int last = (int)((pl >> 8) & 0x7fffffu);
    const bool valid = (pl >> 31) != 0u;
    const float dx = e.mx - pxf, dy = e.my - pyf;
    const float power = __builtin_fmaf(__builtin_fmaf(e.B, dy, e.A * dx), dx, (e.C * dy) * dy);
    const float G0 = __builtin_amdgcn_exp2f(power);
    const bool hit = valid && pos < last && power <= 0.0f && e.o * G0 >= ALPHA_MIN;
    const float G = hit ? G0 : 0.0f;
    const float alpha = fminf(ALPHA_MAX, e.o * G);
    const float rcp = __builtin_amdgcn_rcpf(1.0f - alpha);
    T = T * rcp;
    acc = __builtin_fmaf(lasta, lastc - acc, acc);
    const float cdot = __builtin_fmaf(e.b, d2, __builtin_fmaf(e.g, d1, e.r * d0));
    lastc = cdot;                                            // (a lane that does not take the entry has alpha = 0: the pending pair it leaves
    const float dL_dalpha = __builtin_fmaf(cdot - acc, T, ntf * rcp);   //  behind changes nothing -- render_bwd's predicated body, unchanged)
    lasta = alpha;
    const float w = alpha * T;
    const float v5 = G * dL_dalpha, t = e.o * v5, tx = t * dx, ty = t * dy;
    s0 += tx; s1 += ty; s2 = __builtin_fmaf(tx, dx, s2); s3 = __builtin_fmaf(tx, dy, s3); s4 = __builtin_fmaf(ty, dy, s4); s5 += v5;
    s6 = __builtin_fmaf(w, d0, s6); s7 = __builtin_fmaf(w, d1, s7); s8 = __builtin_fmaf(w, d2, s8);
  }
  float* o = out + ((size_t)bucket * 64 + lane) * 9;
  o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = s5; o[6] = s6; o[7] = s7; o[8] = s8;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 1115;          // one 800 x 800 view of the benchmark: 1115 busy tiles ...
  const int per_tile = argc > 2 ? atoi(argv[2]) : 372;        // ... 414 543 entries: 372 per busy tile
  const int bpt = (per_tile + 63) / 64, n_buckets = tiles * bpt;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  std::vector<Entry> ent((size_t)n_buckets * 64);
  std::vector<PixState> chk((size_t)n_buckets * 256);
  const float L2E = 1.4426950408889634f;
  for (auto& e : ent) {
    const float sx = 1.5f + 4.f * U(rng), sy = 1.5f + 4.f * U(rng), rho = 0.6f * (2.f * U(rng) - 1.f);
    const float a = sx * sx, c = sy * sy, b = rho * sx * sy, det = a * c - b * b;
    e.mx = -6.f + 28.f * U(rng); e.my = -6.f + 28.f * U(rng);
    e.A = -0.5f * L2E * (c / det); e.B = -L2E * (-b / det); e.C = -0.5f * L2E * (a / det);
    e.o = 0.1f + 0.85f * U(rng); e.r = U(rng); e.g = U(rng); e.b = U(rng); e.pad0 = e.pad1 = e.pad2 = 0.f;
  }
  for (size_t i = 0; i < chk.size(); ++i) {
    PixState& p = chk[i];
    const uint32_t pix = (uint32_t)(i & 255u);
    p.T = 0.05f + 0.9f * U(rng); p.acc = U(rng) - 0.5f; p.dL0 = 2.f * U(rng) - 1.f; p.dL1 = 2.f * U(rng) - 1.f; p.dL2 = 2.f * U(rng) - 1.f;
    p.ntfbg = 0.1f * (U(rng) - 0.5f);
    const uint32_t last = (uint32_t)(64.f * U(rng) * 1.3f);    // some pixels stopped inside the bucket
    p.pix_last = pix | ((last > 64u ? 64u : last) << 8); p.pad = 0.f;
  }
  Entry* d_ent; PixState* d_chk; float* d_out;
  CK(hipMalloc(&d_ent, ent.size() * sizeof(Entry))); CK(hipMalloc(&d_chk, chk.size() * sizeof(PixState)));
  CK(hipMalloc(&d_out, (size_t)n_buckets * 64 * 9 * 4));
  CK(hipMemcpy(d_ent, ent.data(), ent.size() * sizeof(Entry), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_chk, chk.data(), chk.size() * sizeof(PixState), hipMemcpyHostToDevice));
  const int blocks = (n_buckets + 3) / 4;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gpl_bwd_kernel, dim3(blocks), dim3(256), 0, 0, d_ent, d_chk, d_out, n_buckets, 0, 0);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gpl_bwd_kernel, dim3(blocks), dim3(256), 0, 0, d_ent, d_chk, d_out, n_buckets, 0, 0);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  const double us = 1e3 * ms / reps;
  std::vector<float> out((size_t)n_buckets * 64 * 9);
  CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  // ---- host check on a few buckets: per pixel, replay the bucket back to front (fp64)
  double worst = 0, scale = 0;
  for (int bk = 0; bk < n_buckets; bk += (n_buckets / 7 > 0 ? n_buckets / 7 : 1)) {
    std::vector<double> ref(64 * 9, 0.0);
    for (int p = 0; p < 256; ++p) {
      const PixState& ps = chk[(size_t)bk * 256 + p];
      double T = ps.T, acc = ps.acc, lastc = 0, lasta = 0;
      const int last = (int)((ps.pix_last >> 8) & 0x7fffff);
      const double px = (double)(ps.pix_last & 15u), py = (double)((ps.pix_last >> 4) & 15u);
      for (int k = 63; k >= 0; --k) {
        const Entry& e = ent[(size_t)bk * 64 + k];
        const double dx = e.mx - px, dy = e.my - py;
        const double power = (e.A * dx + e.B * dy) * dx + e.C * dy * dy;
        const double G0 = std::exp2(power);
        const bool hit = k < last && power <= 0.0 && (double)e.o * G0 >= ALPHA_MIN;
        if (!hit) { if (lasta != 0) { acc = lasta * lastc + (1 - lasta) * acc; lasta = 0; } continue; }
        const double alpha = std::min((double)ALPHA_MAX, (double)e.o * G0);
        T = T / (1 - alpha);
        acc = lasta * lastc + (1 - lasta) * acc;
        const double cdot = e.r * ps.dL0 + e.g * ps.dL1 + e.b * ps.dL2;
        lastc = cdot; lasta = alpha;
        const double dLa = (cdot - acc) * T + ps.ntfbg / (1 - alpha);
        const double v5 = G0 * dLa, t = e.o * v5, tx = t * dx, ty = t * dy, w = alpha * T;
        double* r = &ref[k * 9];
        r[0] += tx; r[1] += ty; r[2] += tx * dx; r[3] += tx * dy; r[4] += ty * dy; r[5] += v5; r[6] += w * ps.dL0; r[7] += w * ps.dL1; r[8] += w * ps.dL2;
      }
    }
    for (int i = 0; i < 64 * 9; ++i) {
      worst = std::max(worst, std::fabs(ref[i] - (double)out[(size_t)bk * 64 * 9 + i]));
      scale = std::max(scale, std::fabs(ref[i]));
    }
  }
  const double pairs = (double)n_buckets * 64.0 * 256.0;
  printf("gpl_bwd: %d tiles x %d entries (%d buckets of 64), %.0f pixel-entry pairs evaluated (all of them: no culling)\n", tiles, per_tile, n_buckets, pairs);
  printf("  kernel %.1f us per launch = %.2f G pairs/s; checkpoint bytes read %.1f MB; host check max |diff| %.3e of scale %.3e (rel %.2e)\n",
         us, pairs / us * 1e-3, chk.size() * 32.0 / 1e6, worst, scale, worst / (scale + 1e-30));
  return worst <= 2e-4 * scale ? 0 : 2;
}
