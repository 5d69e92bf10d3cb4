// gsr_render.hip -- per-tile alpha-compositing with depth, forward and backward, for gfx950
// (replaces the reference extension's two renderCUDA kernels).
//
// One 256-thread workgroup (4 waves of 64) per 16x16 tile, one lane per pixel.  A wave owns a 16x4
// strip, so its colour/depth stores are 64-byte runs.
//
// Forward: the tile's depth-sorted list is staged through LDS 256 entries at a time (one entry gathered
// per lane: 16 + 16 + 8 B from the record arrays), then every lane walks the staged batch with
// broadcast ds_read_b128/b64 (all lanes read the same address: conflict-free).  Early termination is
// per WAVE (`__ballot(!done) == 0` skips the rest of the batch for that wave) and per workgroup
// (`__syncthreads_count`).
//
// Backward: lanes replay their pixel back-to-front.  For every list entry the nine partial gradients
// are summed over the wave's 64 pixels with a DPP reduction (no LDS traffic), lane 63 parks the wave
// total in LDS, and after the batch one lane per entry adds the four wave totals and writes ONE 48-byte
// record to the entry's Gaussian-major slot.  No global atomics: gradients are deterministic, and the
// per-Gaussian reduce in gsr_preprocess_bwd.hip reads contiguous records.
#include "gsr_common.h"

namespace {

#define FWD_BATCH 256

__global__ __launch_bounds__(GSR_BLOCK) void render_fwd_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ recA, const float4* __restrict__ recB, const float2* __restrict__ recC,
    const float* __restrict__ bg, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
    float* __restrict__ out_color, float* __restrict__ out_depth) {
  __shared__ float4 sA[FWD_BATCH];
  __shared__ float4 sB[FWD_BATCH];
  __shared__ float2 sC[FWD_BATCH];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  const int px = (tile % gx) * GSR_TILE + (tid & 15);
  const int py = (tile / gx) * GSR_TILE + (tid >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 rg = ranges[tile];
  const int n = (int)(rg.y - rg.x);

  float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  for (int base = 0; base < n; base += FWD_BATCH) {
    if (__syncthreads_count(done) == GSR_BLOCK) break;
    const int idx = base + tid;
    if (idx < n) {
      const uint32_t g = point_list[rg.x + idx];
      sA[tid] = recA[g];
      sB[tid] = recB[g];
      sC[tid] = recC[g];
    }
    __syncthreads();
    const int m = min(FWD_BATCH, n - base);
    if (__ballot(!done) != 0ull) {
      for (int j = 0; j < m; ++j) {
        if (__ballot(!done) == 0ull) break;
        const float4 a = sA[j];
        const float4 b = sB[j];
        const float2 c = sC[j];
        const float dx = a.x - pxf, dy = a.y - pyf;
        const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
        const float alpha = fminf(GSR_ALPHA_MAX, b.y * gsr_exp(power));
        const bool hit = !done && power <= 0.0f && alpha >= GSR_ALPHA_MIN;
        const float test_T = T * (1.0f - alpha);
        if (hit) {
          if (test_T < GSR_T_EPS) {
            done = true;
          } else {
            const float w = alpha * T;
            C0 += b.z * w; C1 += b.w * w; C2 += c.x * w; Dp += c.y * w;
            T = test_T;
            last = (uint32_t)(base + j + 1);
          }
        }
      }
    }
  }
  if (inside) {
    const int pix = py * W + px;
    const size_t N = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[N + pix] = C1 + T * bg[1];
    out_color[2 * N + pix] = C2 + T * bg[2];
    out_depth[pix] = Dp;
  }
}

// ------------------------------------------------------------------------------------------ backward
#define BWD_BATCH 128

__global__ __launch_bounds__(GSR_BLOCK) void render_bwd_kernel(
    int W, int H, int gx, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const float4* __restrict__ recA, const float4* __restrict__ recB, const float2* __restrict__ recC,
    const float* __restrict__ bg, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ dL_dcolor, const uint2* __restrict__ rect, const uint32_t* __restrict__ offsets,
    float4* __restrict__ partials) {
  __shared__ float4 sA[BWD_BATCH];       // mx, my, A, B
  __shared__ float4 sB[BWD_BATCH];       // C, opacity, r, g
  __shared__ float sBlue[BWD_BATCH];     // b
  __shared__ uint32_t sG[BWD_BATCH];     // gaussian id
  __shared__ float4 sRed[4][BWD_BATCH][GSR_PARTIAL_F4];  // per-wave totals
  __shared__ uint64_t sActive[4][BWD_BATCH / 64];        // which (wave, entry) totals are valid
  __shared__ int sMaxLast;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tile = blockIdx.x;
  const int tx = tile % gx, ty = tile / gx;
  const int px = tx * GSR_TILE + (tid & 15);
  const int py = ty * GSR_TILE + (tid >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const uint2 rg = ranges[tile];
  const int n = (int)(rg.y - rg.x);
  if (n == 0) return;
  const size_t N = (size_t)H * W;
  const int pix = py * W + px;

  const float T_final = inside ? final_T[pix] : 0.f;
  const int last = inside ? (int)n_contrib[pix] : 0;
  float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
  if (inside) { dL0 = dL_dcolor[pix]; dL1 = dL_dcolor[N + pix]; dL2 = dL_dcolor[2 * N + pix]; }
  const float bg_dot = bg[0] * dL0 + bg[1] * dL1 + bg[2] * dL2;
  float T = T_final;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;

  if (tid == 0) sMaxLast = 0;
  __syncthreads();
  atomicMax(&sMaxLast, last);
  __syncthreads();
  const int max_last = sMaxLast;  // entries [max_last, n) are used by no pixel of this tile

  // entries nobody reached still own a slot in the Gaussian-major partial buffer: zero them
  for (int k = max_last + tid; k < n; k += GSR_BLOCK) {
    const uint32_t g = point_list[rg.x + k];
    const uint2 r = rect[g];
    const uint32_t minx = r.x & 0xffffu, miny = r.x >> 16, maxx = r.y & 0xffffu;
    const uint32_t e = offsets[g] + ((uint32_t)ty - miny) * (maxx - minx) + ((uint32_t)tx - minx);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    partials[(size_t)e * GSR_PARTIAL_F4 + 0] = z;
    partials[(size_t)e * GSR_PARTIAL_F4 + 1] = z;
    partials[(size_t)e * GSR_PARTIAL_F4 + 2] = z;
  }

  for (int base = 0; base < max_last; base += BWD_BATCH) {
    // batch entry j (0 = deepest still unprocessed) is list position k = max_last - 1 - (base + j)
    const int m = min(BWD_BATCH, max_last - base);
    if (tid < m) {
      const int k = max_last - 1 - (base + tid);
      const uint32_t g = point_list[rg.x + k];
      sG[tid] = g;
      sA[tid] = recA[g];
      sB[tid] = recB[g];
      sBlue[tid] = recC[g].x;
    }
    __syncthreads();
    uint64_t active_lo = 0ull, active_hi = 0ull;
    for (int j = 0; j < m; ++j) {
      const int k = max_last - 1 - (base + j);
      const float4 a = sA[j];
      const float4 b = sB[j];
      const float blue = sBlue[j];
      const float dx = a.x - pxf, dy = a.y - pyf;
      const float power = -0.5f * (a.z * dx * dx + b.x * dy * dy) - a.w * dx * dy;
      const float G = gsr_exp(power);
      const float alpha = fminf(GSR_ALPHA_MAX, b.y * G);
      const bool hit = (k < last) && power <= 0.0f && alpha >= GSR_ALPHA_MIN;
      if (__ballot(hit) == 0ull) continue;  // wave-uniform: nothing to add for this entry
      float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f, v5 = 0.f, v6 = 0.f, v7 = 0.f, v8 = 0.f;
      if (hit) {
        T = T / (1.0f - alpha);
        const float w = alpha * T;
        acc0 = last_alpha * lc0 + (1.0f - last_alpha) * acc0;
        acc1 = last_alpha * lc1 + (1.0f - last_alpha) * acc1;
        acc2 = last_alpha * lc2 + (1.0f - last_alpha) * acc2;
        lc0 = b.z; lc1 = b.w; lc2 = blue;
        float dL_dalpha = (b.z - acc0) * dL0 + (b.w - acc1) * dL1 + (blue - acc2) * dL2;
        dL_dalpha *= T;
        last_alpha = alpha;
        dL_dalpha += (-T_final / (1.0f - alpha)) * bg_dot;
        const float dL_dG = b.y * dL_dalpha;  // min(0.99, .) is straight-through
        const float gdx = G * dx, gdy = G * dy;
        v0 = dL_dG * (-gdx * a.z - gdy * a.w);
        v1 = dL_dG * (-gdy * b.x - gdx * a.w);
        v2 = -0.5f * gdx * dx * dL_dG;
        v3 = -gdx * dy * dL_dG;
        v4 = -0.5f * gdy * dy * dL_dG;
        v5 = G * dL_dalpha;
        v6 = w * dL0; v7 = w * dL1; v8 = w * dL2;
      }
      v0 = gsr_wave_sum_to_lane63(v0); v1 = gsr_wave_sum_to_lane63(v1); v2 = gsr_wave_sum_to_lane63(v2);
      v3 = gsr_wave_sum_to_lane63(v3); v4 = gsr_wave_sum_to_lane63(v4); v5 = gsr_wave_sum_to_lane63(v5);
      v6 = gsr_wave_sum_to_lane63(v6); v7 = gsr_wave_sum_to_lane63(v7); v8 = gsr_wave_sum_to_lane63(v8);
      if (lane == 63) {
        sRed[wv][j][0] = make_float4(v0, v1, v2, v3);
        sRed[wv][j][1] = make_float4(v4, v5, v6, v7);
        sRed[wv][j][2] = make_float4(v8, 0.f, 0.f, 0.f);
      }
      if (j < 64) active_lo |= (1ull << j); else active_hi |= (1ull << (j - 64));
    }
    if (lane == 0) { sActive[wv][0] = active_lo; sActive[wv][1] = active_hi; }
    __syncthreads();
    if (tid < m) {
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if ((sActive[w][tid >> 6] >> (tid & 63)) & 1ull) {
          const float4 q0 = sRed[w][tid][0], q1 = sRed[w][tid][1], q2 = sRed[w][tid][2];
          r0.x += q0.x; r0.y += q0.y; r0.z += q0.z; r0.w += q0.w;
          r1.x += q1.x; r1.y += q1.y; r1.z += q1.z; r1.w += q1.w;
          r2.x += q2.x;
        }
      }
      const uint32_t g = sG[tid];
      const uint2 r = rect[g];
      const uint32_t minx = r.x & 0xffffu, miny = r.x >> 16, maxx = r.y & 0xffffu;
      const uint32_t e = offsets[g] + ((uint32_t)ty - miny) * (maxx - minx) + ((uint32_t)tx - minx);
      partials[(size_t)e * GSR_PARTIAL_F4 + 0] = r0;
      partials[(size_t)e * GSR_PARTIAL_F4 + 1] = r1;
      partials[(size_t)e * GSR_PARTIAL_F4 + 2] = r2;
    }
    __syncthreads();
  }
}

}  // namespace

int gsr_launch_render_fwd(const GsrCam& cam, const GeomState& g, const BinningState& bs, const ImageState& im,
                          float* out_color, float* out_depth, hipStream_t st) {
  if (cam.T <= 0) return 0;
  { GSR_PROF("render_fwd", st);
  hipLaunchKernelGGL(render_fwd_kernel, dim3(cam.T), dim3(GSR_BLOCK), 0, st, cam.W, cam.H, cam.gx, im.ranges,
                     bs.point_list, g.recA, g.recB, g.recC, cam.bg, im.final_T, im.n_contrib, out_color, out_depth); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_render_bwd(const GsrCam& cam, uint32_t D, const GeomState& g, const BinningState& bs,
                          const ImageState& im, const float* dL_dcolor, float4* partials, hipStream_t st) {
  if (cam.T <= 0 || D == 0) return 0;
  { GSR_PROF("render_bwd", st);
  hipLaunchKernelGGL(render_bwd_kernel, dim3(cam.T), dim3(GSR_BLOCK), 0, st, cam.W, cam.H, cam.gx, im.ranges,
                     bs.point_list, g.recA, g.recB, g.recC, cam.bg, im.final_T, im.n_contrib, dL_dcolor, g.rect,
                     g.offsets, partials); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
