// gsr_dynamics.hip -- the two data-parallel pieces of the rollout plumbing:
//   gsr_fps : farthest point sampling of a point cloud (role of dgl.geometry.farthest_point_sampler, which the
//             reference calls at /root/reference/src/render/dynamics_module.py:46,65 -- a third-party routine absent here);
//   gsr_lbs : moving every Gaussian with the bones -- inverse-distance weights, blended rigid transforms and blended
//             quaternions in ONE pass (the reference loops over bones in Python and materialises [P, n_bones, 3],
//             /root/reference/src/render/utils.py:207-239).
#include "gsr_common.h"

namespace {

// ---------------------------------------------------------------- farthest point sampling
// One workgroup of 1024 threads walks the cloud once per pick: min-distance update + arg-max (first maximum on ties).
// Distances are squared, evaluated as (dx*dx + dy*dy) + dz*dz without contraction so that a host restatement in the
// same order takes the same decisions.
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ pos, int N, int npoints, int start,
                                                   float* __restrict__ mind, long long* __restrict__ out) {
  __shared__ float s_val[16];
  __shared__ int s_idx[16];
  __shared__ int s_cur;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < N; i += 1024) mind[i] = __builtin_inff();
  if (tid == 0) s_cur = start;
  __syncthreads();
  for (int k = 0; k < npoints; ++k) {
    const int cur = s_cur;
    if (tid == 0) out[k] = cur;
    const float cx = pos[3 * cur], cy = pos[3 * cur + 1], cz = pos[3 * cur + 2];
    float best = -1.0f;
    int besti = 0x7fffffff;
    for (int i = tid; i < N; i += 1024) {
      const float dx = pos[3 * i] - cx, dy = pos[3 * i + 1] - cy, dz = pos[3 * i + 2] - cz;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      const float m = fminf(mind[i], d);
      mind[i] = m;
      if (m > best) { best = m; besti = i; }          // increasing i: keeps the first maximum of this thread
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(besti, off, 64);
      if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { s_val[wv] = best; s_idx[wv] = besti; }
    __syncthreads();
    if (tid == 0) {
      float b = s_val[0];
      int bi = s_idx[0];
      for (int w = 1; w < 16; ++w)
        if (s_val[w] > b || (s_val[w] == b && s_idx[w] < bi)) { b = s_val[w]; bi = s_idx[w]; }
      s_cur = bi;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- linear blend skinning of the Gaussians
#define LBS_CHUNK 256   // bones staged per LDS round
__global__ __launch_bounds__(GSR_BLOCK) void lbs_kernel(int P, int nb, const float* __restrict__ bones,
                                                        const float* __restrict__ R, const float* __restrict__ t,
                                                        const float* __restrict__ bq, const float* __restrict__ xyz,
                                                        const float* __restrict__ quat, float* __restrict__ out_xyz,
                                                        float* __restrict__ out_quat) {
  __shared__ float sB[LBS_CHUNK][3], sR[LBS_CHUNK][9], sT[LBS_CHUNK][3], sQ[LBS_CHUNK][4];
  const int p = blockIdx.x * GSR_BLOCK + threadIdx.x;
  const bool live = p < P;
  float x = 0.f, y = 0.f, z = 0.f;
  if (live) { x = xyz[3 * p]; y = xyz[3 * p + 1]; z = xyz[3 * p + 2]; }
  float ws = 0.f, ax = 0.f, ay = 0.f, az = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  for (int b0 = 0; b0 < nb; b0 += LBS_CHUNK) {
    const int n = min(LBS_CHUNK, nb - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < n * 3; i += GSR_BLOCK) { (&sB[0][0])[i] = bones[3 * b0 + i]; (&sT[0][0])[i] = t[3 * b0 + i]; }
    for (int i = threadIdx.x; i < n * 9; i += GSR_BLOCK) (&sR[0][0])[i] = R[9 * b0 + i];
    for (int i = threadIdx.x; i < n * 4; i += GSR_BLOCK) (&sQ[0][0])[i] = bq[4 * b0 + i];
    __syncthreads();
    for (int b = 0; b < n; ++b) {   // every lane reads the same LDS address: broadcast, conflict-free
      const float dx = x - sB[b][0], dy = y - sB[b][1], dz = z - sB[b][2];
      const float w = 1.0f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-4f);
      ws += w;
      ax += w * (sR[b][0] * dx + sR[b][1] * dy + sR[b][2] * dz + sT[b][0] + sB[b][0]);
      ay += w * (sR[b][3] * dx + sR[b][4] * dy + sR[b][5] * dz + sT[b][1] + sB[b][1]);
      az += w * (sR[b][6] * dx + sR[b][7] * dy + sR[b][8] * dz + sT[b][2] + sB[b][2]);
      q0 += w * sQ[b][0]; q1 += w * sQ[b][1]; q2 += w * sQ[b][2]; q3 += w * sQ[b][3];
    }
  }
  if (!live) return;
  const float inv = 1.0f / ws;
  out_xyz[3 * p] = ax * inv; out_xyz[3 * p + 1] = ay * inv; out_xyz[3 * p + 2] = az * inv;
  if (quat && out_quat) {
    const float qn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);   // F.normalize's eps
    const float a0 = q0 / qn, a1 = q1 / qn, a2 = q2 / qn, a3 = q3 / qn;
    const float b0 = quat[4 * p], b1 = quat[4 * p + 1], b2 = quat[4 * p + 2], b3 = quat[4 * p + 3];
    out_quat[4 * p + 0] = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3;
    out_quat[4 * p + 1] = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2;
    out_quat[4 * p + 2] = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1;
    out_quat[4 * p + 3] = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0;
  }
}

}  // namespace

int gsr_launch_fps(int N, const float* pos, int npoints, int start, float* mind, long long* out, hipStream_t st) {
  if (N <= 0 || npoints <= 0) return 0;
  { GSR_PROF("fps", st);
    hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), 0, st, pos, N, npoints, start, mind, out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_lbs(int P, int nb, const float* bones, const float* R, const float* t, const float* bq, const float* xyz,
                   const float* quat, float* out_xyz, float* out_quat, hipStream_t st) {
  if (P <= 0) return 0;
  { GSR_PROF("lbs", st);
    hipLaunchKernelGGL(lbs_kernel, dim3((P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, P, nb, bones, R, t, bq, xyz, quat,
                       out_xyz, out_quat); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
