// gsr_loss.hip -- fused image loss of the tracking step for gfx950:
//     loss = 0.8 * mean|x - y| + 0.2 * (1 - mean SSIM(x, y))
// exactly the term the reference evaluates twice per iteration with PyTorch ops
// (/root/reference/src/tracking/train_utils.py:185,195; SSIM = five zero-padded depthwise 11x11 Gaussian
// convolutions, /root/reference/src/tracking/external.py:101-135).  Here: one forward and one backward kernel.
//
// A 256-thread workgroup owns a 16x16 output tile of one channel.  The 26x26 input patch (halo 5, zeros
// outside the image, as conv2d's zero padding) of x and y goes to LDS; the 11-tap Gaussian is applied
// separably (horizontal pass into LDS for the five moments x, y, xx, yy, xy; vertical pass in registers).
// Forward also stores, per pixel, the three partials of the SSIM map w.r.t. the blurred moments that depend
// on x: f_A (d/d blur(x)), f_C (d/d blur(xx)), f_E (d/d blur(xy)); because the window is symmetric,
//     d loss / d x(q) = [blur(f_A) + 2 x blur(f_C) + y blur(f_E)](q) * (-0.2 / N) + 0.8 sign(x - y) / N,
// which the backward kernel evaluates with the same separable machinery.  Block partial sums are written
// to an array and summed by the caller: no atomics, deterministic.
#include "gsr_common.h"

namespace {

#define LT 16          // output tile edge
#define HALO 5
#define PT (LT + 2 * HALO)  // 26

__device__ __forceinline__ float load_px(const float* __restrict__ img, int H, int W, int y, int x) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : 0.0f;
}

struct Win { float g[11]; };

__global__ __launch_bounds__(256) void image_loss_fwd_kernel(Win win, int C, int H, int W, const float* __restrict__ x_img,
                                                             const float* __restrict__ y_img,
                                                             float* __restrict__ fA, float* __restrict__ fC,
                                                             float* __restrict__ fE, float* __restrict__ block_l1,
                                                             float* __restrict__ block_ssim) {
  __shared__ float sx[PT][PT + 1];
  __shared__ float sy[PT][PT + 1];
  __shared__ float hb[5][PT][LT + 1];   // horizontally blurred moments
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT, ch = blockIdx.z;
  const float* __restrict__ xc = x_img + (size_t)ch * H * W;
  const float* __restrict__ yc = y_img + (size_t)ch * H * W;
  for (int i = tid; i < PT * PT; i += 256) {
    const int py = i / PT, px = i % PT;
    sx[py][px] = load_px(xc, H, W, ty0 + py - HALO, tx0 + px - HALO);
    sy[py][px] = load_px(yc, H, W, ty0 + py - HALO, tx0 + px - HALO);
  }
  __syncthreads();
  for (int i = tid; i < PT * LT; i += 256) {   // horizontal pass: 26 rows x 16 columns
    const int py = i / LT, ox = i % LT;
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f, e = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float xv = sx[py][ox + k], yv = sy[py][ox + k], w = win.g[k];
      a += w * xv; b += w * yv; c += w * xv * xv; d += w * yv * yv; e += w * xv * yv;
    }
    hb[0][py][ox] = a; hb[1][py][ox] = b; hb[2][py][ox] = c; hb[3][py][ox] = d; hb[4][py][ox] = e;
  }
  __syncthreads();
  float A = 0.f, B = 0.f, Cc = 0.f, D = 0.f, E = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = win.g[k];
    A += w * hb[0][ly + k][lx]; B += w * hb[1][ly + k][lx]; Cc += w * hb[2][ly + k][lx];
    D += w * hb[3][ly + k][lx]; E += w * hb[4][ly + k][lx];
  }
  const int gx = tx0 + lx, gy = ty0 + ly;
  const bool inside = gx < W && gy < H;
  float ssim = 0.f, l1 = 0.f;
  if (inside) {
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const float num1 = 2.0f * A * B + c1, num2 = 2.0f * (E - A * B) + c2;
    const float den1 = A * A + B * B + c1, den2 = (Cc - A * A) + (D - B * B) + c2;
    const float inv = 1.0f / (den1 * den2);
    ssim = num1 * num2 * inv;
    const size_t o = (size_t)ch * H * W + (size_t)gy * W + gx;
    fA[o] = 2.0f * B * (num2 - num1) * inv - ssim * 2.0f * A * (1.0f / den1 - 1.0f / den2);
    fC[o] = -ssim / den2;
    fE[o] = 2.0f * num1 * inv;
    l1 = fabsf(sx[ly + HALO][lx + HALO] - sy[ly + HALO][lx + HALO]);
  }
  // block sums (wave shuffle + LDS), one partial per block
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { ssim += __shfl_xor(ssim, m, 64); l1 += __shfl_xor(l1, m, 64); }
  if ((tid & 63) == 0) { red[0][tid >> 6] = ssim; red[1][tid >> 6] = l1; }
  __syncthreads();
  if (tid == 0) {
    const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    block_ssim[b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    block_l1[b] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ __launch_bounds__(256) void image_loss_bwd_kernel(Win win, int C, int H, int W, const float* __restrict__ x_img,
                                                             const float* __restrict__ y_img,
                                                             const float* __restrict__ fA, const float* __restrict__ fC,
                                                             const float* __restrict__ fE,
                                                             const float* __restrict__ grad_loss, int cpi, float w_l1, float w_ssim,
                                                             float* __restrict__ dx) {
  __shared__ float s0[PT][PT + 1];
  __shared__ float s1[PT][PT + 1];
  __shared__ float s2[PT][PT + 1];
  __shared__ float hb[3][PT][LT + 1];
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const int tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT, ch = blockIdx.z;
  const size_t coff = (size_t)ch * H * W;
  for (int i = tid; i < PT * PT; i += 256) {
    const int py = i / PT, px = i % PT;
    const int yy = ty0 + py - HALO, xx = tx0 + px - HALO;
    s0[py][px] = load_px(fA + coff, H, W, yy, xx);
    s1[py][px] = load_px(fC + coff, H, W, yy, xx);
    s2[py][px] = load_px(fE + coff, H, W, yy, xx);
  }
  __syncthreads();
  for (int i = tid; i < PT * LT; i += 256) {
    const int py = i / LT, ox = i % LT;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = win.g[k];
      a += w * s0[py][ox + k]; b += w * s1[py][ox + k]; c += w * s2[py][ox + k];
    }
    hb[0][py][ox] = a; hb[1][py][ox] = b; hb[2][py][ox] = c;
  }
  __syncthreads();
  float bA = 0.f, bC = 0.f, bE = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = win.g[k];
    bA += w * hb[0][ly + k][lx]; bC += w * hb[1][ly + k][lx]; bE += w * hb[2][ly + k][lx];
  }
  const int gx = tx0 + lx, gy = ty0 + ly;
  if (gx < W && gy < H) {
    const size_t o = coff + (size_t)gy * W + gx;
    const float xv = x_img[o], yv = y_img[o];
    const float invN = 1.0f / ((float)cpi * (float)H * (float)W);   // a batch of C / cpi images of cpi channels each
    const float dssim = bA + 2.0f * xv * bC + yv * bE;            // d(sum of SSIM map)/dx
    const float dl1 = xv > yv ? 1.0f : (xv < yv ? -1.0f : 0.0f);  // torch: sign(x - y), 0 at ties
    dx[o] = grad_loss[ch / cpi] * invN * (w_l1 * dl1 - w_ssim * dssim);
  }
}

}  // namespace

int gsr_launch_image_loss_fwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, float* fA,
                              float* fC, float* fE, float* block_l1, float* block_ssim, hipStream_t st) {
  Win w;
  for (int i = 0; i < 11; ++i) w.g[i] = win11_host[i];
  const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
  { GSR_PROF("image_loss_fwd", st);
    hipLaunchKernelGGL(image_loss_fwd_kernel, grid, dim3(256), 0, st, w, C, H, W, x, y, fA, fC, fE, block_l1, block_ssim); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_image_loss_bwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, const float* fA,
                              const float* fC, const float* fE, const float* grad_loss, int cpi, float w_l1, float w_ssim, float* dx,
                              hipStream_t st) {
  Win w;
  for (int i = 0; i < 11; ++i) w.g[i] = win11_host[i];
  const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
  { GSR_PROF("image_loss_bwd", st);
    hipLaunchKernelGGL(image_loss_bwd_kernel, grid, dim3(256), 0, st, w, C, H, W, x, y, fA, fC, fE, grad_loss, cpi, w_l1, w_ssim,
                       dx); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
