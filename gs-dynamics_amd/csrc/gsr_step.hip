// gsr_step.hip -- the small per-Gaussian pieces of the tracking step around the rasterizer (caller side of the path,
// SURVEY.md section 8a row A9), each a handful of PyTorch element-wise / indexing / reduction kernels in the reference and
// host-launch-bound there (~5 us of GPU work behind ~15 us of dispatch each):
//   * activations  (/root/reference/src/tracking/helpers.py:36-45, params2rendervar):
//         rotations = normalize(unnorm_rotations), opacities = sigmoid(logit_opacities), scales = exp(log_scales)
//     one forward and one backward kernel instead of ~6 + ~12;
//   * the view-independent terms of the t > 0 loss (/root/reference/src/tracking/train_utils.py:198-241): the three neighbour
//     terms (gsr_rigidity.hip) plus floor = mean(clamp(y_fg, min=0)) and bg = mean|p_bg - p_bg0|_1 + mean|q_bg - q_bg0|_1, their
//     weighted sum formed on the device -> two kernels + one finishing kernel forward, three kernels backward, where the
//     reference runs ~40 + ~60.
// No float atomics anywhere: block partials + fixed-order finishing sums.
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_step::<kernel>, not "(anonymous namespace)"
namespace gsr_step {

#define ST_BLOCK 256

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__global__ __launch_bounds__(ST_BLOCK) void activate_fwd_kernel(int P, const float* __restrict__ unnorm, const float* __restrict__ logit,
                                                                const float* __restrict__ logs, float* __restrict__ rot,
                                                                float* __restrict__ op, float* __restrict__ sc) {
  const int i = blockIdx.x * ST_BLOCK + threadIdx.x;
  if (i >= P) return;
  reinterpret_cast<float4*>(rot)[i] = gsr_act_rotation(reinterpret_cast<const float4*>(unnorm)[i]);
  op[i] = gsr_act_opacity(logit[i]);
  sc[3 * (size_t)i] = gsr_act_scale(logs[3 * (size_t)i]);
  sc[3 * (size_t)i + 1] = gsr_act_scale(logs[3 * (size_t)i + 1]);
  sc[3 * (size_t)i + 2] = gsr_act_scale(logs[3 * (size_t)i + 2]);
}

// any of the incoming gradients may be NULL (that output was not used): its parameter gradient is zero
__global__ __launch_bounds__(ST_BLOCK) void activate_bwd_kernel(int P, const float* __restrict__ unnorm, const float* __restrict__ op,
                                                                const float* __restrict__ sc, const float* __restrict__ d_rot,
                                                                const float* __restrict__ d_op, const float* __restrict__ d_sc,
                                                                float* __restrict__ d_unnorm, float* __restrict__ d_logit,
                                                                float* __restrict__ d_logs) {
  const int i = blockIdx.x * ST_BLOCK + threadIdx.x;
  if (i >= P) return;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (d_rot) g = gsr_act_rotation_bwd(reinterpret_cast<const float4*>(unnorm)[i], reinterpret_cast<const float4*>(d_rot)[i]);
  reinterpret_cast<float4*>(d_unnorm)[i] = g;
  const float o = op[i];
  d_logit[i] = d_op ? d_op[i] * o * (1.0f - o) : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) d_logs[3 * (size_t)i + c] = d_sc ? d_sc[3 * (size_t)i + c] * sc[3 * (size_t)i + c] : 0.f;
}

// floor / background terms: item < n_fg -> foreground point (floor), else background point (two L1 terms)
__global__ __launch_bounds__(ST_BLOCK) void point_terms_fwd_kernel(int nfg, int nbg, const float* __restrict__ means3D,
                                                                   const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
                                                                   const int64_t* __restrict__ bg_idx,
                                                                   const float* __restrict__ init_pts, const float* __restrict__ init_rot,
                                                                   float* __restrict__ partial /*[3][blocks]*/) {
  __shared__ float red[3][ST_BLOCK / 64];
  const int t = blockIdx.x * ST_BLOCK + threadIdx.x;
  float l_floor = 0.f, l_pts = 0.f, l_rot = 0.f;
  if (t < nfg) {
    l_floor = fmaxf(means3D[3 * (size_t)fg_idx[t] + 1], 0.f);
  } else if (t < nfg + nbg) {
    const int b = t - nfg;
    const size_t gb = (size_t)bg_idx[b];
#pragma unroll
    for (int c = 0; c < 3; ++c) l_pts += fabsf(means3D[3 * gb + c] - init_pts[3 * (size_t)b + c]);
#pragma unroll
    for (int c = 0; c < 4; ++c) l_rot += fabsf(rot[4 * gb + c] - init_rot[4 * (size_t)b + c]);
  }
  l_floor = wave_sum(l_floor); l_pts = wave_sum(l_pts); l_rot = wave_sum(l_rot);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = l_floor; red[1][wv] = l_pts; red[2][wv] = l_rot; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < ST_BLOCK / 64; ++w) s += red[threadIdx.x][w];
    partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

// runs after the neighbour-term gather (which WROTE the foreground rows): adds the floor gradient there, writes the background rows
__global__ __launch_bounds__(ST_BLOCK) void point_terms_bwd_kernel(int nfg, int nbg, const float* __restrict__ means3D,
                                                                   const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
                                                                   const int64_t* __restrict__ bg_idx,
                                                                   const float* __restrict__ init_pts, const float* __restrict__ init_rot,
                                                                   const float* __restrict__ grad_total, float s_floor, float s_bg,
                                                                   float* __restrict__ d_means3D, float* __restrict__ d_rot, int accumulate) {
  const int t = blockIdx.x * ST_BLOCK + threadIdx.x;
  const float g = grad_total[0];
  if (t < nfg) {
    const size_t gi = (size_t)fg_idx[t];
    if (means3D[3 * gi + 1] >= 0.f) d_means3D[3 * gi + 1] += g * s_floor;      // torch.clamp passes the gradient at the bound
  } else if (t < nfg + nbg) {
    const int b = t - nfg;
    const size_t gb = (size_t)bg_idx[b];
    const float gs = g * s_bg;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = means3D[3 * gb + c] - init_pts[3 * (size_t)b + c];
      const float gv = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
      d_means3D[3 * gb + c] = accumulate ? d_means3D[3 * gb + c] + gv : gv;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = rot[4 * gb + c] - init_rot[4 * (size_t)b + c];
      const float gv = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
      d_rot[4 * gb + c] = accumulate ? d_rot[4 * gb + c] + gv : gv;
    }
  }
}

__device__ __forceinline__ float wave_strided_sum(const float* __restrict__ p, int n, int lane) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int j = lane;
  for (; j + 192 < n; j += 256) { s0 += p[j]; s1 += p[j + 64]; s2 += p[j + 128]; s3 += p[j + 192]; }
  for (; j < n; j += 64) s0 += p[j];
  return wave_sum((s0 + s1) + (s2 + s3));
}

struct TermScales { float inv_edges, inv_fg, inv_bg, w[5]; };

// terms[0..4] = rigid, rot, iso, floor, bg;  terms[5] = sum_k w_k terms[k].  One wave per partial array.
__global__ __launch_bounds__(384) void shared_terms_finish_kernel(TermScales sc, int nbe, const float* __restrict__ edge_partial,
                                                                  int nbp, const float* __restrict__ point_partial,
                                                                  float* __restrict__ terms) {
  __shared__ float s[6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float v = wave < 3 ? wave_strided_sum(edge_partial + (size_t)wave * nbe, nbe, lane)
                           : wave_strided_sum(point_partial + (size_t)(wave - 3) * nbp, nbp, lane);
  if (lane == 0) s[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t0 = s[0] * sc.inv_edges, t1 = s[1] * sc.inv_edges, t2 = s[2] * sc.inv_edges;
    const float t3 = s[3] * sc.inv_fg, t4 = (s[4] + s[5]) * sc.inv_bg;
    terms[0] = t0; terms[1] = t1; terms[2] = t2; terms[3] = t3; terms[4] = t4;
    terms[5] = sc.w[0] * t0 + sc.w[1] * t1 + sc.w[2] * t2 + sc.w[3] * t3 + sc.w[4] * t4;
  }
}

// Adam update of up to GSR_ADAM_MAX_TENSORS parameter tensors in ONE launch (the tracking loop's optimiser has one group per
// parameter, /root/reference/src/tracking/train_utils.py:152-164: eight launches-plus-dispatch per iteration in torch, ~0.2 ms of
// host time against ~0.4 ms for the whole loss step).  Same arithmetic, in the same order, as torch.optim.Adam's default path:
//   m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
struct AdamTab { gsr_adam_tensor t[GSR_ADAM_MAX_TENSORS]; int first_block[GSR_ADAM_MAX_TENSORS + 1]; int n; };

#define ADAM_PER_BLOCK (ST_BLOCK * 4)
__global__ __launch_bounds__(ST_BLOCK) void adam_step_kernel(AdamTab tab) {
  int k = 0;
  while (k + 1 < tab.n && (int)blockIdx.x >= tab.first_block[k + 1]) ++k;     // uniform: scalar
  const gsr_adam_tensor& t = tab.t[k];
  const int64_t base = (int64_t)((int)blockIdx.x - tab.first_block[k]) * ADAM_PER_BLOCK;
  const float one_m_b1 = t.one_minus_beta1, one_m_b2 = t.one_minus_beta2;
  const float step_size = t.lr / t.bias_correction1;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = base + u * ST_BLOCK + threadIdx.x;
    if (i < t.n) {
      const float g = t.grad[i];
      float m = t.exp_avg[i], v = t.exp_avg_sq[i];
      m = m + one_m_b1 * (g - m);
      v = v * t.beta2 + one_m_b2 * (g * g);
      t.exp_avg[i] = m;
      t.exp_avg_sq[i] = v;
      const float denom = sqrtf(v) / t.bias_correction2_sqrt + t.eps;
      t.param[i] = t.param[i] - step_size * (m / denom);
    }
  }
}

// max_2D_radius[i] = max(max_2D_radius[i], max_v radii[v][i]);  seen[i] = any_v radii[v][i] > 0   over the rows v = 0, step, 2 step, ...
// of a [V,P] int32 array (the colour renders of a step): the bookkeeping of /root/reference/src/tracking/train_utils.py:243-245.
__global__ __launch_bounds__(ST_BLOCK) void radius_bookkeeping_kernel(int V, int step, int P, const int32_t* __restrict__ radii,
                                                                      float* __restrict__ max_2d, uint8_t* __restrict__ seen) {
  const int i = blockIdx.x * ST_BLOCK + threadIdx.x;
  if (i >= P) return;
  int m = 0;
  for (int v = 0; v < V; v += step) m = max(m, radii[(size_t)v * P + i]);
  max_2d[i] = fmaxf(max_2d[i], (float)m);
  seen[i] = m > 0 ? 1 : 0;
}

inline int blocks_for(int n) { return (n + ST_BLOCK - 1) / ST_BLOCK; }

}  // namespace gsr_step
using namespace gsr_step;

int gsr_launch_activate_fwd(int P, const float* unnorm, const float* logit, const float* logs, float* rot, float* op, float* sc,
                            hipStream_t st) {
  if (P <= 0) return 0;
  { GSR_PROF("activate_fwd", st);
    hipLaunchKernelGGL(activate_fwd_kernel, dim3(blocks_for(P)), dim3(ST_BLOCK), 0, st, P, unnorm, logit, logs, rot, op, sc); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_activate_bwd(int P, const float* unnorm, const float* op, const float* sc, const float* d_rot, const float* d_op,
                            const float* d_sc, float* d_unnorm, float* d_logit, float* d_logs, hipStream_t st) {
  if (P <= 0) return 0;
  { GSR_PROF("activate_bwd", st);
    hipLaunchKernelGGL(activate_bwd_kernel, dim3(blocks_for(P)), dim3(ST_BLOCK), 0, st, P, unnorm, op, sc, d_rot, d_op, d_sc, d_unnorm,
                       d_logit, d_logs); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_shared_terms_point_blocks(int nfg, int nbg) { return blocks_for(nfg + nbg > 0 ? nfg + nbg : 1); }

int gsr_launch_shared_terms_fwd(int nfg, int K, int nbg, const float* means3D, const float* rot, const int64_t* fg_idx,
                                const int64_t* bg_idx, const int64_t* nbr, const float* nw, const float* nd, const float* prev_inv,
                                const float* prev_off, const float* init_pts, const float* init_rot, const float* w5,
                                float* partials, float* terms, hipStream_t st) {
  const int nbe = gsr_rigidity_fwd_blocks(nfg), nbp = gsr_shared_terms_point_blocks(nfg, nbg);
  float* frames = partials;                                  // 16 nfg floats first (the buffer's base alignment carries over)
  float* edge_partial = partials + 16 * (size_t)(nfg > 0 ? nfg : 0);
  float* point_partial = edge_partial + 3 * (size_t)nbe;
  if (int e = gsr_launch_rigidity_fwd(nfg, K, means3D, rot, fg_idx, nbr, nw, nd, prev_inv, prev_off, frames, edge_partial, st)) return e;
  { GSR_PROF("point_terms_fwd", st);
    hipLaunchKernelGGL(point_terms_fwd_kernel, dim3(nbp), dim3(ST_BLOCK), 0, st, nfg, nbg, means3D, rot, fg_idx, bg_idx, init_pts,
                       init_rot, point_partial); }
  TermScales sc;
  sc.inv_edges = nfg > 0 && K > 0 ? 1.0f / ((float)nfg * (float)K) : 0.f;
  sc.inv_fg = nfg > 0 ? 1.0f / (float)nfg : 0.f;
  sc.inv_bg = nbg > 0 ? 1.0f / (float)nbg : 0.f;
  for (int k = 0; k < 5; ++k) sc.w[k] = w5[k];
  { GSR_PROF("shared_terms_finish", st);
    hipLaunchKernelGGL(shared_terms_finish_kernel, dim3(1), dim3(384), 0, st, sc, nbe, (const float*)edge_partial, nbp,
                       (const float*)point_partial, terms); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_shared_terms_bwd(int P, int nfg, int K, int nbg, const float* means3D, const float* rot, const int64_t* fg_idx,
                                const int64_t* bg_idx, const int64_t* nbr, const float* nw, const float* nd, const float* prev_inv,
                                const float* prev_off, const float* init_pts, const float* init_rot, const float* w5,
                                const float* grad_total, const int32_t* rev_ptr, const int32_t* rev_edge, float* scratch,
                                float* d_means3D, float* d_rot, int flags, hipStream_t st) {
  const int accumulate = flags & GSR_SHARED_ACCUMULATE, frames_valid = (flags & GSR_SHARED_FRAMES_VALID) ? 1 : 0;
  if (!accumulate) {
    GSR_HIP_CHECK(hipMemsetAsync(d_means3D, 0, sizeof(float) * 3 * (size_t)P, st));
    GSR_HIP_CHECK(hipMemsetAsync(d_rot, 0, sizeof(float) * 4 * (size_t)P, st));
  }
  const float inv_edges = nfg > 0 && K > 0 ? 1.0f / ((float)nfg * (float)K) : 0.f;
  float* frames = scratch;                                   // 16 nfg floats, then 8 (nfg + nfg K) of the neighbour terms
  float* self7 = scratch + 16 * (size_t)(nfg > 0 ? nfg : 0);
  float* edge7 = self7 + 8 * (size_t)nfg;
  if (int e = gsr_launch_rigidity_bwd(nfg, K, means3D, rot, fg_idx, nbr, nw, nd, prev_inv, prev_off, grad_total, 0, w5[0] * inv_edges,
                                      w5[1] * inv_edges, w5[2] * inv_edges, rev_ptr, rev_edge, frames, frames_valid, self7, edge7, d_means3D,
                                      d_rot, accumulate, st))
    return e;
  { GSR_PROF("point_terms_bwd", st);
    hipLaunchKernelGGL(point_terms_bwd_kernel, dim3(gsr_shared_terms_point_blocks(nfg, nbg)), dim3(ST_BLOCK), 0, st, nfg, nbg, means3D,
                       rot, fg_idx, bg_idx, init_pts, init_rot, grad_total, nfg > 0 ? w5[3] / (float)nfg : 0.f,
                       nbg > 0 ? w5[4] / (float)nbg : 0.f, d_means3D, d_rot, accumulate); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_adam_step(int n_tensors, const gsr_adam_tensor* t, hipStream_t st) {
  AdamTab tab;
  tab.n = 0;
  int blocks = 0;
  for (int i = 0; i < n_tensors; ++i) {
    if (t[i].n <= 0) continue;
    tab.t[tab.n] = t[i];
    tab.first_block[tab.n] = blocks;
    blocks += (int)((t[i].n + ADAM_PER_BLOCK - 1) / ADAM_PER_BLOCK);
    ++tab.n;
  }
  tab.first_block[tab.n] = blocks;
  if (blocks == 0) return 0;
  { GSR_PROF("adam_step", st);
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(ST_BLOCK), 0, st, tab); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_radius_bookkeeping(int V, int step, int P, const int32_t* radii, float* max_2d, uint8_t* seen, hipStream_t st) {
  if (P <= 0 || V <= 0) return 0;
  { GSR_PROF("radius_bookkeeping", st);
    hipLaunchKernelGGL(radius_bookkeeping_kernel, dim3(blocks_for(P)), dim3(ST_BLOCK), 0, st, V, step, P, radii, max_2d, seen); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
