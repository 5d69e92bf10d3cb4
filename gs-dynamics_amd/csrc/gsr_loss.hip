// gsr_loss.hip -- fused image terms of the tracking step for gfx950:
//     loss_i = w_l1 * mean|pred_i - target_i| + w_ssim * (1 - mean SSIM(pred_i, target_i)),   total = sum_i weight_i loss_i
// with pred_i = exp(cam_m[row_i]) * render_i + cam_c[row_i] for the colour renders (the per-camera affine of
// /root/reference/src/tracking/train_utils.py:181-183) and pred_i = render_i for the segmentation renders -- exactly the
// terms the reference evaluates twice per iteration with PyTorch ops (train_utils.py:185,195; SSIM = five zero-padded
// depthwise 11x11 Gaussian convolutions, /root/reference/src/tracking/external.py:101-135).  Here: one forward and one
// backward kernel for ALL images of a step, reading the rasterizer's output batch in place and writing the gradient batch
// the rasterizer's backward consumes, plus two single-workgroup finishing kernels (losses / camera-affine gradients).
//
// A 256-thread workgroup owns a 32 x 54 output tile of one channel of one image.  The 42 x 64 input patch (halo 5, zeros
// outside the image, as conv2d's zero padding) goes to LDS; the 11-tap Gaussian is applied separably with register blocking:
//   horizontal: 64 patch rows x 4 segments of 8 outputs = 256 items; an item reads its 18 inputs once (5 ds_read_b128 per
//               map) and produces 8 outputs per moment, written back over the patch (all reads are in registers by then);
//   vertical:   32 columns x 8 segments of 7 rows = 256 items; 17 reads per moment for 7 outputs.
// ~130 VALU lane-operations and ~25 LDS dwords per pixel, against ~200 and ~90 for one output per thread.
// Forward also stores, per pixel, the three partials of the SSIM map w.r.t. the blurred moments that depend on pred:
// f_A (d/d blur(x)), f_C (d/d blur(xx)), f_E (d/d blur(xy)); because the window is symmetric,
//     d loss / d x(q) = [blur(f_A) + 2 x blur(f_C) + y blur(f_E)](q) * (-w_ssim / N) + w_l1 sign(x - y) / N,
// which the backward kernel evaluates with the same machinery.  Block partial sums go to arrays that the finishing kernels
// (or the caller) add up in a fixed order: no atomics, deterministic.
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_loss::<kernel>, not "(anonymous namespace)"
namespace gsr_loss {

#define TW 32               // output tile width
#define TH 54               // output tile height
#define HALO 5
#define PW (TW + 2 * HALO)  // 42 patch columns
#define PH (TH + 2 * HALO)  // 64 patch rows
#define PS 44               // patch row stride in floats (16-byte aligned rows, >= 3 * 8 + 20)
#define VSEG 7              // output rows per vertical item (8 segments cover 56 >= TH)

struct Win { float g[11]; };

struct LossTab {                 // one entry per image, passed by value
  int n_images, channels;
  int cam_row[GSR_LOSS_MAX_IMAGES];
  int grad_idx[GSR_LOSS_MAX_IMAGES];
  float weight[GSR_LOSS_MAX_IMAGES];
  const float* target[GSR_LOSS_MAX_IMAGES];
  const float* tmom[GSR_LOSS_MAX_IMAGES];      // blur(y), blur(y*y) of the target ([2, channels, H, W]) or nullptr
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// 8 outputs of the 11-tap filter over 18 inputs
#define GSR_FIR8(out, in, g)                                  \
  _Pragma("unroll") for (int o_ = 0; o_ < 8; ++o_) {          \
    float s_ = 0.f;                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < 11; ++k_) s_ = fmaf((g)[k_], (in)[o_ + k_], s_); \
    (out)[o_] = s_;                                           \
  }

// base[byte_off / 4] with a wave-uniform base and a 32-bit per-lane byte offset: the global_load saddr + voffset form (no 64-bit VALU
// address arithmetic per access; a channel plane stays below 4 GiB by far)
__device__ __forceinline__ float ld_off(const float* __restrict__ base, uint32_t byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st_off(float* __restrict__ base, uint32_t byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

__device__ __forceinline__ void load20(const float* __restrict__ row, float* __restrict__ v) {
  const float4* __restrict__ p = reinterpret_cast<const float4*>(row);
#pragma unroll
  for (int i = 0; i < 5; ++i) { const float4 q = p[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
}

__device__ __forceinline__ void store8(float* __restrict__ dst, const float* __restrict__ v) {
  float4* __restrict__ p = reinterpret_cast<float4*>(dst);
  p[0] = make_float4(v[0], v[1], v[2], v[3]);
  p[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// One pixel of the SSIM map and its three partials w.r.t. the blurred moments that depend on pred.  Contraction is OFF in here: which
// product-sum pairs the compiler would fuse depends on where B and D come from -- registers (MODE 0) or loads (MODE 1) -- and MODE 1 must
// reproduce MODE 0 bit for bit (the first forward of a target runs MODE 0, the later ones MODE 1).
__device__ __forceinline__ void ssim_point(float A, float B, float Cc, float D, float E, float& ssim, float& vA, float& vC, float& vE) {
#pragma clang fp contract(off)
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  const float AB = A * B, AA = A * A, BB = B * B;
  const float num1 = 2.0f * AB + c1, num2 = 2.0f * (E - AB) + c2;
  const float den1 = (AA + BB) + c1, den2 = ((Cc - AA) + (D - BB)) + c2;
  const float i1 = __builtin_amdgcn_rcpf(den1), i2 = __builtin_amdgcn_rcpf(den2), inv = i1 * i2;   // 1 ulp: den >= c1, c2
  ssim = (num1 * num2) * inv;
  vA = ((2.0f * B) * (num2 - num1)) * inv - ((ssim * 2.0f) * A) * (i1 - i2);
  vC = -(ssim * i2);
  vE = (2.0f * num1) * inv;
}

// Which tile is this workgroup's?  The launch is ONE-dimensional with 8 * ceil(N / 8) workgroups for the N = tiles x planes of the call,
// and workgroup w takes slot (w % 8) * ceil(N / 8) + w / 8 of the (plane, tile row, tile column) order: the dispatcher places block b on
// XCD b % 8 (observed, MI355X_MICROARCH.md: a speed matter only), so every XCD works through a CONTIGUOUS eighth of the tiles and the
// halos that neighbouring tiles share (a patch is 1.56x its tile) are read from that XCD's L2 instead of crossing the fabric again --
// with blockIdx.x = tile column, neighbours sat on different XCDs and both kernels ran at the fabric's rate (round 5: FETCH_SIZE x 2 +
// WRITE_SIZE = 659 / 724 MiB per launch for 236 + 176 / 294 + 59 MiB of compulsory traffic; profiles/r05_loss_kernels.txt).
struct LossTile { int tx, ty, ch, bidx; };
__device__ __forceinline__ bool loss_tile(int W, int H, int planes, LossTile& t) {
  const uint32_t gx = (uint32_t)(W + TW - 1) / TW, gy = (uint32_t)(H + TH - 1) / TH, T = gx * gy, N = T * (uint32_t)planes;
  const uint32_t per = (N + 7u) / 8u, w = blockIdx.x, slot = (w & 7u) * per + (w >> 3);
  if (slot >= N) return false;
  const uint32_t ch = slot / T, r = slot - ch * T, ty = r / gx;
  t.ch = (int)ch; t.ty = (int)ty; t.tx = (int)(r - ty * gx); t.bidx = (int)slot;     // slot = (ch * gy + ty) * gx + tx: the finishing kernels' order
  return true;
}

// MODE 0: all five blurred moments in the kernel.  MODE 1: blur(y) and blur(y*y) of the (fixed) target come from `tmom`
// (tab.tmom[img]: [2, channels, H, W], written once per target by MODE 2) -- 3 instead of 5 FIRs per pass, 25 instead of 41 KB of
// LDS.  MODE 2: only those two maps of the target are computed and stored (x is not read).  The arithmetic of a moment is the same
// instruction sequence in every mode, so MODE 1 reproduces MODE 0 bit for bit (tested).
template <int MODE>
__global__ __launch_bounds__(256) void image_loss_fwd_kernel(Win win, LossTab tab, int H, int W, const float* __restrict__ x_img,
                                                             const float* __restrict__ cam_m, const float* __restrict__ cam_c,
                                                             float* __restrict__ fA, float* __restrict__ fC,
                                                             float* __restrict__ fE, float* __restrict__ block_l1,
                                                             float* __restrict__ block_ssim, int planes) {
  constexpr int NM = MODE == 0 ? 5 : (MODE == 1 ? 3 : 2);            // blurred moments formed here
  constexpr int MR = PH + 2;     // rows of a blurred-moment plane: the vertical items of the last segment read up to row 65 -- two rows that
                                 // nobody writes and whose outputs (tile rows >= TH) nobody keeps: no clamp in the read loop
  constexpr int SM = NM * MR * TW > 2 * PH * PS ? NM * MR * TW : 2 * PH * PS;
  __shared__ __attribute__((aligned(16))) float smem[SM];   // patch x|y (2 * 64 * 44), then the NM blurred moments
  __shared__ float red[2][4];
  float* __restrict__ sx = smem;
  float* __restrict__ sy = smem + PH * PS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the patch rows are scalar work)
  LossTile lt;
  if (!loss_tile(W, H, planes, lt)) return;
  const int tx0 = lt.tx * TW, ty0 = lt.ty * TH, ch = lt.ch;
  const int img = ch / tab.channels, c = ch - img * tab.channels;
  const size_t HW = (size_t)H * W;
  const float* __restrict__ yc = tab.target[img] + (size_t)c * HW;
  const float* __restrict__ xc = MODE == 2 ? yc : x_img + (size_t)ch * HW;
  float a = 1.0f, b = 0.0f;
  const int row = MODE == 2 ? -1 : tab.cam_row[img];
  if (row >= 0) { a = expf(cam_m[row * tab.channels + c]); b = cam_c[row * tab.channels + c]; }

  {  // patch load: all 32 global loads of a lane are issued before the first LDS store (addresses clamped, values masked).  A wave loads
     // whole patch rows: the row pointers are scalar, a lane contributes its (clamped) column as a 32-bit byte offset
    const int gx = tx0 + lane - HALO;
    const uint32_t cxb = 4u * (uint32_t)min(max(gx, 0), W - 1);
    const bool okx = gx >= 0 && gx < W && lane < PW;
    float xv[PH / 4], yv[PH / 4];
#pragma unroll
    for (int i = 0; i < PH / 4; ++i) {
      const int gy = ty0 + wave + 4 * i - HALO, cy = min(max(gy, 0), H - 1);
      const size_t ro = (size_t)cy * W;
      xv[i] = MODE == 2 ? 0.f : ld_off(xc + ro, cxb);
      yv[i] = ld_off(yc + ro, cxb);
    }
    if (lane < PW) {
      float* __restrict__ px = sx + wave * PS + lane;
      float* __restrict__ py = sy + wave * PS + lane;
#pragma unroll
      for (int i = 0; i < PH / 4; ++i) {
        const int gy = ty0 + wave + 4 * i - HALO;
        const bool ok = okx && gy >= 0 && gy < H;
        px[4 * i * PS] = ok ? fmaf(a, xv[i], b) : 0.f;
        py[4 * i * PS] = ok ? yv[i] : 0.f;
      }
    }
  }
  // MODE 1: this thread's 2 x 7 target moments (needed by the SSIM map at the very end) are requested NOW: they travel while the filters run
  // (they used to be loaded behind the vertical pass: a second memory latency on every workgroup's critical path -- 103 -> ~90 us, round 5)
  const int vc = tid & 31, vs = tid >> 5, r0 = vs * VSEG;
  const int gxo = tx0 + vc;
  const uint32_t pb0 = 4u * ((uint32_t)(ty0 + r0) * (uint32_t)W + (uint32_t)gxo), rowb = 4u * (uint32_t)W;
  float* __restrict__ tm = const_cast<float*>(tab.tmom[img]);      // MODE 1: read, MODE 2: written
  float* __restrict__ tmB = tm + (size_t)c * HW;                    // plane bases are scalar; a pixel is a 32-bit byte offset
  float* __restrict__ tmD = tm + (size_t)(tab.channels + c) * HW;
  float tB[VSEG], tD[VSEG];
  if (MODE == 1) {
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      const bool in = r0 + o < TH && gxo < W && ty0 + r0 + o < H;
      const uint32_t pb = in ? pb0 + (uint32_t)o * rowb : 0u;
      tB[o] = ld_off(tmB, pb); tD[o] = ld_off(tmD, pb);
    }
  }
  __syncthreads();
  // horizontal pass
  const int hr = tid >> 2, hs = tid & 3;
  float xr[20], yr[20];
  load20(sx + hr * PS + hs * 8, xr);
  load20(sy + hr * PS + hs * 8, yr);
  float l1 = 0.f;
  if (MODE != 2 && hr >= HALO && hr < HALO + TH) {
#pragma unroll
    for (int o = 0; o < 8; ++o) l1 += fabsf(xr[o + HALO] - yr[o + HALO]);   // pixels outside the image are 0 - 0
  }
  __syncthreads();                      // every row is in registers: the moments may overwrite the patch
  {
    float out[8], prod[18];
    float* __restrict__ dst = smem + hr * TW + hs * 8;
    int plane = 0;
    if (MODE != 2) { GSR_FIR8(out, xr, win.g); store8(dst + plane * MR * TW, out); ++plane; }
    if (MODE != 1) { GSR_FIR8(out, yr, win.g); store8(dst + plane * MR * TW, out); ++plane; }
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < 18; ++j) prod[j] = xr[j] * xr[j];
      GSR_FIR8(out, prod, win.g); store8(dst + plane * MR * TW, out); ++plane;
    }
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < 18; ++j) prod[j] = yr[j] * yr[j];
      GSR_FIR8(out, prod, win.g); store8(dst + plane * MR * TW, out); ++plane;
    }
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < 18; ++j) prod[j] = xr[j] * yr[j];
      GSR_FIR8(out, prod, win.g); store8(dst + plane * MR * TW, out); ++plane;
    }
  }
  __syncthreads();
  // vertical pass + SSIM map
  float mom[NM][VSEG];
  const float* __restrict__ vbase = smem + r0 * TW + vc;      // one address; plane and row offsets are immediates
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    float v[VSEG + 10];
#pragma unroll
    for (int i = 0; i < VSEG + 10; ++i) v[i] = vbase[m * MR * TW + i * TW];
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) s = fmaf(win.g[k], v[o + k], s);
      mom[m][o] = s;
    }
  }
  float ssim_sum = 0.f;
  const int gx = gxo;
  float* __restrict__ fAc = fA + (size_t)ch * HW;
  float* __restrict__ fCc = fC + (size_t)ch * HW;
  float* __restrict__ fEc = fE + (size_t)ch * HW;
#pragma unroll
  for (int o = 0; o < VSEG; ++o) {
    const int gy = ty0 + r0 + o;
    if (r0 + o < TH && gx < W && gy < H) {
      const uint32_t pb = pb0 + (uint32_t)o * rowb;
      if (MODE == 2) {
        st_off(tmB, pb, mom[0][o]);
        st_off(tmD, pb, mom[1][o]);
        continue;
      }
      float A, B, Cc, D, E;
      if (MODE == 0) { A = mom[0][o]; B = mom[1][o]; Cc = mom[2][o]; D = mom[3][o]; E = mom[4][o]; }
      else { A = mom[0][o]; Cc = mom[1][o]; E = mom[2][o]; B = tB[o]; D = tD[o]; }
      float ssim, vA, vC, vE;
      ssim_point(A, B, Cc, D, E, ssim, vA, vC, vE);
      st_off(fAc, pb, vA);
      st_off(fCc, pb, vC);
      st_off(fEc, pb, vE);
      ssim_sum += ssim;
    }
  }
  if (MODE == 2) return;
  ssim_sum = wave_sum(ssim_sum);
  l1 = wave_sum(l1);
  if (lane == 0) { red[0][wave] = ssim_sum; red[1][wave] = l1; }
  __syncthreads();
  if (tid == 0) {
    const int bidx = lt.bidx;
    block_ssim[bidx] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    block_l1[bidx] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void image_loss_bwd_kernel(Win win, LossTab tab, int H, int W, const float* __restrict__ x_img,
                                                             const float* __restrict__ cam_m, const float* __restrict__ cam_c,
                                                             const float* __restrict__ fA, const float* __restrict__ fC,
                                                             const float* __restrict__ fE, const float* __restrict__ grad,
                                                             float invN, float w_l1, float w_ssim, float* __restrict__ dx,
                                                             float* __restrict__ block_dm, float* __restrict__ block_dc, int planes) {
  __shared__ __attribute__((aligned(16))) float smem[3 * PH * PS];   // three patches, then the 3 blurred maps (3 * 64 * 32)
  __shared__ float red[2][4];
  constexpr int MR = PH + 2;     // rows of a blurred plane (see image_loss_fwd_kernel)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  LossTile lt;
  if (!loss_tile(W, H, planes, lt)) return;
  const int tx0 = lt.tx * TW, ty0 = lt.ty * TH, ch = lt.ch;
  const int img = ch / tab.channels, c = ch - img * tab.channels;
  const size_t HW = (size_t)H * W, coff = (size_t)ch * HW;
  float a = 1.0f, b = 0.0f;
  const int row = tab.cam_row[img];
  if (row >= 0) { a = expf(cam_m[row * tab.channels + c]); b = cam_c[row * tab.channels + c]; }
  const float g = grad[tab.grad_idx[img]] * tab.weight[img] * invN;

  {
    const int gx = tx0 + lane - HALO;
    const uint32_t cxb = 4u * (uint32_t)min(max(gx, 0), W - 1);       // scalar row pointers + a 32-bit column offset per lane
    const bool okx = gx >= 0 && gx < W && lane < PW;
    float* __restrict__ pl = smem + wave * PS + lane;
#pragma unroll
    for (int half = 0; half < 2; ++half) {      // 2 x 24 loads in flight per lane
      float v0[PH / 8], v1[PH / 8], v2[PH / 8];
#pragma unroll
      for (int i = 0; i < PH / 8; ++i) {
        const int gy = ty0 + wave + 4 * (i + half * (PH / 8)) - HALO, cy = min(max(gy, 0), H - 1);
        const size_t ro = coff + (size_t)cy * W;
        v0[i] = ld_off(fA + ro, cxb); v1[i] = ld_off(fC + ro, cxb); v2[i] = ld_off(fE + ro, cxb);
      }
      if (lane < PW) {
#pragma unroll
        for (int i = 0; i < PH / 8; ++i) {
          const int r4 = 4 * (i + half * (PH / 8)), gy = ty0 + wave + r4 - HALO;
          const bool ok = okx && gy >= 0 && gy < H;
          pl[r4 * PS] = ok ? v0[i] : 0.f;
          pl[PH * PS + r4 * PS] = ok ? v1[i] : 0.f;
          pl[2 * PH * PS + r4 * PS] = ok ? v2[i] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  const int hr = tid >> 2, hs = tid & 3;
  float r0v[20], r1v[20], r2v[20];
  load20(smem + hr * PS + hs * 8, r0v);
  load20(smem + PH * PS + hr * PS + hs * 8, r1v);
  load20(smem + 2 * PH * PS + hr * PS + hs * 8, r2v);
  __syncthreads();
  {
    float out[8];
    float* __restrict__ dst = smem + hr * TW + hs * 8;
    GSR_FIR8(out, r0v, win.g); store8(dst, out);
    GSR_FIR8(out, r1v, win.g); store8(dst + MR * TW, out);
    GSR_FIR8(out, r2v, win.g); store8(dst + 2 * MR * TW, out);
  }
  __syncthreads();
  const int vc = tid & 31, vs = tid >> 5, r0 = vs * VSEG;
  float bl[3][VSEG];
  const float* __restrict__ vbase = smem + r0 * TW + vc;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float v[VSEG + 10];
#pragma unroll
    for (int i = 0; i < VSEG + 10; ++i) v[i] = vbase[m * MR * TW + i * TW];
#pragma unroll
    for (int o = 0; o < VSEG; ++o) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) s = fmaf(win.g[k], v[o + k], s);
      bl[m][o] = s;
    }
  }
  const float* __restrict__ yc = tab.target[img] + (size_t)c * HW;
  const float* __restrict__ xcp = x_img + coff;
  float* __restrict__ dxc = dx + coff;
  const int gx = tx0 + vc;
  const uint32_t pb0 = 4u * ((uint32_t)(ty0 + r0) * (uint32_t)W + (uint32_t)gx), rowb = 4u * (uint32_t)W;
  float sum_dm = 0.f, sum_dc = 0.f;
#pragma unroll
  for (int o = 0; o < VSEG; ++o) {
    const int gy = ty0 + r0 + o;
    if (r0 + o < TH && gx < W && gy < H) {
      const uint32_t pb = pb0 + (uint32_t)o * rowb;
      const float xraw = ld_off(xcp, pb), yv = ld_off(yc, pb);
      const float xv = fmaf(a, xraw, b);
      const float dssim = bl[0][o] + 2.0f * xv * bl[1][o] + yv * bl[2][o];   // d(sum of SSIM map)/d pred
      const float dl1 = xv > yv ? 1.0f : (xv < yv ? -1.0f : 0.0f);           // torch: sign(x - y), 0 at ties
      const float dpred = g * (w_l1 * dl1 - w_ssim * dssim);
      const float dr = a * dpred;
      st_off(dxc, pb, dr);
      sum_dm += dr * xraw;       // d/d cam_m: pred = exp(m) render + c
      sum_dc += dpred;
    }
  }
  if (block_dm) {
    sum_dm = wave_sum(sum_dm);
    sum_dc = wave_sum(sum_dc);
    if (lane == 0) { red[0][wave] = sum_dm; red[1][wave] = sum_dc; }
    __syncthreads();
    if (tid == 0) {
      const int bidx = lt.bidx;
      block_dm[bidx] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      block_dc[bidx] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
  }
}

// strided sum of n floats by one wave, four loads in flight per lane, fixed order
__device__ __forceinline__ float wave_strided_sum(const float* __restrict__ p, int n, int lane) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int j = lane;
  for (; j + 192 < n; j += 256) { s0 += p[j]; s1 += p[j + 64]; s2 += p[j + 128]; s3 += p[j + 192]; }
  for (; j < n; j += 64) s0 += p[j];
  return wave_sum((s0 + s1) + (s2 + s3));
}

// one wave per image: loss_i from the block partials; then the weighted total.  Fixed summation order.
__global__ __launch_bounds__(1024) void loss_finish_fwd_kernel(LossTab tab, int per_image, const float* __restrict__ block_l1,
                                                               const float* __restrict__ block_ssim, float invN, float w_l1,
                                                               float w_ssim, float* __restrict__ losses) {
  __shared__ float li[GSR_LOSS_MAX_IMAGES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < tab.n_images; i += 16) {
    const float s1 = wave_strided_sum(block_l1 + (size_t)i * per_image, per_image, lane);
    const float s2 = wave_strided_sum(block_ssim + (size_t)i * per_image, per_image, lane);
    if (lane == 0) {
      const float l = w_l1 * (s1 * invN) + w_ssim * (1.0f - s2 * invN);
      li[i] = l;
      losses[i] = l;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < tab.n_images; ++i) t += tab.weight[i] * li[i];
    losses[tab.n_images] = t;
  }
}

// d_cam_m / d_cam_c [n_cams, channels]: one wave per (image, channel) sum, then a serial scatter in image order (duplicates add up)
__global__ __launch_bounds__(1024) void loss_finish_bwd_kernel(LossTab tab, int per_channel, const float* __restrict__ block_dm,
                                                               const float* __restrict__ block_dc, int n_cams,
                                                               float* __restrict__ d_cam_m, float* __restrict__ d_cam_c) {
  __shared__ float sm[GSR_LOSS_MAX_IMAGES * 4], sc[GSR_LOSS_MAX_IMAGES * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = tab.channels, pairs = tab.n_images * nch;
  for (int i = threadIdx.x; i < n_cams * nch; i += 1024) { d_cam_m[i] = 0.f; d_cam_c[i] = 0.f; }
  for (int p = wave; p < pairs; p += 16) {
    float s1 = 0.f, s2 = 0.f;
    if (tab.cam_row[p / nch] >= 0) {
      s1 = wave_strided_sum(block_dm + (size_t)p * per_channel, per_channel, lane);
      s2 = wave_strided_sum(block_dc + (size_t)p * per_channel, per_channel, lane);
    }
    if (lane == 0) { sm[p] = s1; sc[p] = s2; }
  }
  __syncthreads();
  if (threadIdx.x < nch) {
    const int c = threadIdx.x;
    for (int i = 0; i < tab.n_images; ++i) {
      const int row = tab.cam_row[i];
      if (row >= 0 && row < n_cams) { d_cam_m[row * nch + c] += sm[i * nch + c]; d_cam_c[row * nch + c] += sc[i * nch + c]; }
    }
  }
}

Win make_win(const float* w11) {
  Win w;
  for (int i = 0; i < 11; ++i) w.g[i] = w11[i];
  return w;
}

LossTab make_tab(const gsr_loss_views* v) {
  LossTab t;
  t.n_images = v->n_images;
  t.channels = v->channels;
  for (int i = 0; i < GSR_LOSS_MAX_IMAGES; ++i) {
    const bool on = i < v->n_images;
    t.cam_row[i] = on ? v->cam_row[i] : -1;
    t.grad_idx[i] = 0;
    t.weight[i] = on ? v->weight[i] : 0.f;
    t.target[i] = on ? v->target[i] : nullptr;
    t.tmom[i] = on ? v->target_moments[i] : nullptr;
  }
  return t;
}

inline dim3 loss_grid(int C, int H, int W) {      // one-dimensional, a multiple of 8 (see loss_tile)
  const unsigned N = (unsigned)(((W + TW - 1) / TW) * ((H + TH - 1) / TH)) * (unsigned)C;
  return dim3(8u * ((N + 7u) / 8u));
}

}  // namespace gsr_loss
using namespace gsr_loss;

int gsr_loss_blocks_per_channel(int H, int W) { return ((W + TW - 1) / TW) * ((H + TH - 1) / TH); }

// plain batch of C channels (C / cpi images): the original entry points, chunked through the table kernels
int gsr_launch_image_loss_fwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, float* fA,
                              float* fC, float* fE, float* block_l1, float* block_ssim, hipStream_t st) {
  const Win w = make_win(win11_host);
  const size_t HW = (size_t)H * W;
  const int nb = gsr_loss_blocks_per_channel(H, W);
  for (int c0 = 0; c0 < C; c0 += GSR_LOSS_MAX_IMAGES) {
    const int n = C - c0 < GSR_LOSS_MAX_IMAGES ? C - c0 : GSR_LOSS_MAX_IMAGES;
    LossTab t;
    t.n_images = n; t.channels = 1;
    for (int i = 0; i < GSR_LOSS_MAX_IMAGES; ++i) {
      t.cam_row[i] = -1; t.grad_idx[i] = 0; t.weight[i] = 1.f;
      t.target[i] = i < n ? y + (size_t)(c0 + i) * HW : nullptr;
      t.tmom[i] = nullptr;
    }
    { GSR_PROF("image_loss_fwd", st);
      hipLaunchKernelGGL(image_loss_fwd_kernel<0>, loss_grid(n, H, W), dim3(256), 0, st, w, t, H, W, x + (size_t)c0 * HW,
                         (const float*)nullptr, (const float*)nullptr, fA + (size_t)c0 * HW, fC + (size_t)c0 * HW,
                         fE + (size_t)c0 * HW, block_l1 + (size_t)c0 * nb, block_ssim + (size_t)c0 * nb, n); }
  }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_image_loss_bwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, const float* fA,
                              const float* fC, const float* fE, const float* grad_loss, int cpi, float w_l1, float w_ssim, float* dx,
                              hipStream_t st) {
  const Win w = make_win(win11_host);
  const size_t HW = (size_t)H * W;
  const float invN = 1.0f / ((float)cpi * (float)H * (float)W);   // a batch of C / cpi images of cpi channels each
  for (int c0 = 0; c0 < C; c0 += GSR_LOSS_MAX_IMAGES) {
    const int n = C - c0 < GSR_LOSS_MAX_IMAGES ? C - c0 : GSR_LOSS_MAX_IMAGES;
    LossTab t;
    t.n_images = n; t.channels = 1;
    for (int i = 0; i < GSR_LOSS_MAX_IMAGES; ++i) {
      t.cam_row[i] = -1; t.grad_idx[i] = i < n ? (c0 + i) / cpi : 0; t.weight[i] = 1.f;
      t.target[i] = i < n ? y + (size_t)(c0 + i) * HW : nullptr;
      t.tmom[i] = nullptr;
    }
    { GSR_PROF("image_loss_bwd", st);
      hipLaunchKernelGGL(image_loss_bwd_kernel, loss_grid(n, H, W), dim3(256), 0, st, w, t, H, W, x + (size_t)c0 * HW,
                         (const float*)nullptr, (const float*)nullptr, fA + (size_t)c0 * HW, fC + (size_t)c0 * HW,
                         fE + (size_t)c0 * HW, grad_loss, invN, w_l1, w_ssim, dx + (size_t)c0 * HW, (float*)nullptr, (float*)nullptr, n); }
  }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_views_loss_fwd(const float* win11_host, const gsr_loss_views* v, int H, int W, const float* renders,
                              const float* cam_m, const float* cam_c, float w_l1, float w_ssim, float* fA, float* fC, float* fE,
                              float* partials, float* losses, hipStream_t st) {
  const Win w = make_win(win11_host);
  const LossTab t = make_tab(v);
  const int C = v->n_images * v->channels, nb = gsr_loss_blocks_per_channel(H, W);
  float* block_l1 = partials;
  float* block_ssim = partials + (size_t)C * nb;
  bool cached = true;      // every target brought its two blurred maps along: the 3-moment build
  for (int i = 0; i < v->n_images; ++i) cached = cached && v->target_moments[i] != nullptr;
  { GSR_PROF("image_loss_fwd", st);
    auto k = cached ? image_loss_fwd_kernel<1> : image_loss_fwd_kernel<0>;
    hipLaunchKernelGGL(k, loss_grid(C, H, W), dim3(256), 0, st, w, t, H, W, renders, cam_m, cam_c, fA, fC, fE, block_l1, block_ssim, C); }
  const float invN = 1.0f / ((float)v->channels * (float)H * (float)W);
  { GSR_PROF("loss_finish_fwd", st);
    hipLaunchKernelGGL(loss_finish_fwd_kernel, dim3(1), dim3(1024), 0, st, t, v->channels * nb, (const float*)block_l1,
                       (const float*)block_ssim, invN, w_l1, w_ssim, losses); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_views_loss_bwd(const float* win11_host, const gsr_loss_views* v, int H, int W, const float* renders,
                              const float* cam_m, const float* cam_c, int n_cams, const float* fA, const float* fC,
                              const float* fE, const float* grad_total, float w_l1, float w_ssim, float* d_renders,
                              float* partials, float* d_cam_m, float* d_cam_c, hipStream_t st) {
  const Win w = make_win(win11_host);
  const LossTab t = make_tab(v);
  const int C = v->n_images * v->channels, nb = gsr_loss_blocks_per_channel(H, W);
  const bool cams = d_cam_m && d_cam_c && n_cams > 0;
  float* block_dm = cams ? partials : nullptr;
  float* block_dc = cams ? partials + (size_t)C * nb : nullptr;
  const float invN = 1.0f / ((float)v->channels * (float)H * (float)W);
  { GSR_PROF("image_loss_bwd", st);
    hipLaunchKernelGGL(image_loss_bwd_kernel, loss_grid(C, H, W), dim3(256), 0, st, w, t, H, W, renders, cam_m, cam_c, fA, fC, fE,
                       grad_total, invN, w_l1, w_ssim, d_renders, block_dm, block_dc, C); }
  if (cams) {
    GSR_PROF("loss_finish_bwd", st);
    hipLaunchKernelGGL(loss_finish_bwd_kernel, dim3(1), dim3(1024), 0, st, t, nb, (const float*)block_dm, (const float*)block_dc,
                       n_cams, d_cam_m, d_cam_c);
  }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_target_moments(const float* win11_host, int channels, int H, int W, const float* target, float* moments, hipStream_t st) {
  const Win w = make_win(win11_host);
  LossTab t;
  t.n_images = 1; t.channels = channels;
  for (int i = 0; i < GSR_LOSS_MAX_IMAGES; ++i) { t.cam_row[i] = -1; t.grad_idx[i] = 0; t.weight[i] = 0.f; t.target[i] = nullptr; t.tmom[i] = nullptr; }
  t.target[0] = target; t.tmom[0] = moments;
  { GSR_PROF("target_moments", st);
    hipLaunchKernelGGL(image_loss_fwd_kernel<2>, loss_grid(channels, H, W), dim3(256), 0, st, w, t, H, W, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, channels); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
