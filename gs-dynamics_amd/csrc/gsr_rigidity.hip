// gsr_rigidity.hip -- the neighbour terms of the t > 0 tracking loss (rigid, rot, iso) fused into three kernels
// (caller side of the path, SURVEY.md section 8a row A9: /root/reference/src/tracking/train_utils.py:198-222).
// In PyTorch they are ~100 element-wise / indexing kernels over [N_fg, K, 3] tensors (3.9 ms per step at 70 k foreground
// points x 20 neighbours on MI355X); here every foreground point walks its K neighbours once.
//
// Per foreground point i (all arrays are indexed by foreground rank; fg_idx maps rank -> Gaussian):
//   q_i = rot[fg_idx[i]] * prev_inv_rot[i]                     (Hamilton product, w,x,y,z)
//   R_i = rotation matrix of q_i / |q_i|
//   for neighbour k: j = nbr[i,k], off = p_j - p_i
//     rigid += sqrt(|R_i^T off - prev_offset[i,k]|^2 w + 1e-20)
//     rot   += sqrt(|q_j - q_i|^2 w + 1e-20)
//     iso   += sqrt((sqrt(|off|^2 + 1e-20) - dist[i,k])^2 w + 1e-20)
// each divided by N_fg * K (torch .mean()).  Forward writes one partial sum per block and term (the caller adds them up:
// no atomics).  Backward: kernel 2 recomputes the edge terms, keeps what flows to point i itself and writes what flows to
// the neighbour j (d/dp_j: 3 floats, d/dq_j: 4 floats) into an edge-major buffer; kernel 3 gathers a point's incoming edges
// through a reverse adjacency (CSR, built once per timestep by the caller) and applies the chain rule to the rotation
// input -- deterministic, no float atomics.
#include "gsr_common.h"

namespace {

struct Quat { float w, x, y, z; };

__device__ __forceinline__ Quat qmul(const Quat a, const Quat b) {
  return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
              a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Quat load_q(const float* __restrict__ p, size_t i) { return Quat{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}; }

__device__ __forceinline__ void rotmat(const Quat u, float R[9]) {   // u is a unit quaternion
  const float r = u.w, x = u.x, y = u.y, z = u.z;
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - r * z); R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y); R[7] = 2.f * (y * z + r * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

#define RG_BLOCK 256
__global__ __launch_bounds__(RG_BLOCK) void rigidity_fwd_kernel(
    int nfg, int K, const float* __restrict__ means3D, const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
    const int64_t* __restrict__ nbr, const float* __restrict__ nw, const float* __restrict__ nd,
    const float* __restrict__ prev_inv, const float* __restrict__ prev_off, float* __restrict__ partial /*[3][blocks]*/) {
  __shared__ float red[3][RG_BLOCK / 64];
  const int i = blockIdx.x * RG_BLOCK + threadIdx.x;
  float l1 = 0.f, l2 = 0.f, l3 = 0.f;
  if (i < nfg) {
    const size_t gi = (size_t)fg_idx[i];
    const float px = means3D[3 * gi], py = means3D[3 * gi + 1], pz = means3D[3 * gi + 2];
    const Quat q = qmul(load_q(rot, gi), load_q(prev_inv, i));
    const float inv = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    float R[9];
    rotmat(Quat{q.w * inv, q.x * inv, q.y * inv, q.z * inv}, R);
    for (int k = 0; k < K; ++k) {
      const size_t e = (size_t)i * K + k;
      const int j = (int)nbr[e];
      const size_t gj = (size_t)fg_idx[j];
      const float ox = means3D[3 * gj] - px, oy = means3D[3 * gj + 1] - py, oz = means3D[3 * gj + 2] - pz;
      const float w = nw[e];
      const float dx = (ox * R[0] + oy * R[3] + oz * R[6]) - prev_off[3 * e];
      const float dy = (ox * R[1] + oy * R[4] + oz * R[7]) - prev_off[3 * e + 1];
      const float dz = (ox * R[2] + oy * R[5] + oz * R[8]) - prev_off[3 * e + 2];
      l1 += sqrtf((dx * dx + dy * dy + dz * dz) * w + 1e-20f);
      const Quat qj = qmul(load_q(rot, gj), load_q(prev_inv, j));
      const float ew = qj.w - q.w, ex = qj.x - q.x, ey = qj.y - q.y, ez = qj.z - q.z;
      l2 += sqrtf((ew * ew + ex * ex + ey * ey + ez * ez) * w + 1e-20f);
      const float t = sqrtf(ox * ox + oy * oy + oz * oz + 1e-20f) - nd[e];
      l3 += sqrtf(t * t * w + 1e-20f);
    }
  }
  l1 = gsr_wave_sum_shfl(l1); l2 = gsr_wave_sum_shfl(l2); l3 = gsr_wave_sum_shfl(l3);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = l1; red[1][wv] = l2; red[2][wv] = l3; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < RG_BLOCK / 64; ++w) s += red[threadIdx.x][w];
    partial[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s;
  }
}

// Kernel 2: edge terms again, now for the gradient.  g[3] = upstream gradients of the three means, already divided by
// N_fg * K.  self[i] = {d/dp_i (3), d/dq_i (4)} from point i's own edges; edge[e] = {d/dp_j (3), d/dq_j (4)}.
__global__ __launch_bounds__(RG_BLOCK) void rigidity_bwd_edges_kernel(
    int nfg, int K, const float* __restrict__ means3D, const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
    const int64_t* __restrict__ nbr, const float* __restrict__ nw, const float* __restrict__ nd,
    const float* __restrict__ prev_inv, const float* __restrict__ prev_off, const float* __restrict__ g,
    float* __restrict__ self7, float* __restrict__ edge7) {
  const int i = blockIdx.x * RG_BLOCK + threadIdx.x;
  if (i >= nfg) return;
  const float g1 = g[0], g2 = g[1], g3 = g[2];
  const size_t gi = (size_t)fg_idx[i];
  const float px = means3D[3 * gi], py = means3D[3 * gi + 1], pz = means3D[3 * gi + 2];
  const Quat q = qmul(load_q(rot, gi), load_q(prev_inv, i));
  const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  const float inv = 1.0f / sqrtf(n2);
  const Quat u{q.w * inv, q.x * inv, q.y * inv, q.z * inv};
  float R[9];
  rotmat(u, R);
  float G[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // d loss / d R_i
  float sp[3] = {0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const size_t e = (size_t)i * K + k;
    const int j = (int)nbr[e];
    const size_t gj = (size_t)fg_idx[j];
    const float ox = means3D[3 * gj] - px, oy = means3D[3 * gj + 1] - py, oz = means3D[3 * gj + 2] - pz;
    const float w = nw[e];
    // rigid
    const float dx = (ox * R[0] + oy * R[3] + oz * R[6]) - prev_off[3 * e];
    const float dy = (ox * R[1] + oy * R[4] + oz * R[7]) - prev_off[3 * e + 1];
    const float dz = (ox * R[2] + oy * R[5] + oz * R[8]) - prev_off[3 * e + 2];
    const float c1 = g1 * w / sqrtf((dx * dx + dy * dy + dz * dz) * w + 1e-20f);
    const float ax = c1 * dx, ay = c1 * dy, az = c1 * dz;              // d/d(R^T off)
    float fx = R[0] * ax + R[1] * ay + R[2] * az;                        // d/d off = R ga
    float fy = R[3] * ax + R[4] * ay + R[5] * az;
    float fz = R[6] * ax + R[7] * ay + R[8] * az;
    G[0] += ox * ax; G[1] += ox * ay; G[2] += ox * az;
    G[3] += oy * ax; G[4] += oy * ay; G[5] += oy * az;
    G[6] += oz * ax; G[7] += oz * ay; G[8] += oz * az;
    // iso
    const float mag = sqrtf(ox * ox + oy * oy + oz * oz + 1e-20f);
    const float t = mag - nd[e];
    const float c3 = g3 * w * t / sqrtf(t * t * w + 1e-20f) / mag;
    fx += c3 * ox; fy += c3 * oy; fz += c3 * oz;
    // rot
    const Quat qj = qmul(load_q(rot, gj), load_q(prev_inv, j));
    const float ew = qj.w - q.w, ex = qj.x - q.x, ey = qj.y - q.y, ez = qj.z - q.z;
    const float c2 = g2 * w / sqrtf((ew * ew + ex * ex + ey * ey + ez * ez) * w + 1e-20f);
    const float hw = c2 * ew, hx = c2 * ex, hy = c2 * ey, hz = c2 * ez;
    float* E = edge7 + 7 * e;
    E[0] = fx; E[1] = fy; E[2] = fz; E[3] = hw; E[4] = hx; E[5] = hy; E[6] = hz;
    sp[0] -= fx; sp[1] -= fy; sp[2] -= fz;
    sq[0] -= hw; sq[1] -= hx; sq[2] -= hy; sq[3] -= hz;
  }
  // d loss / d u from G (R as a function of the unit quaternion u), then through the normalisation q -> q / |q|
  const float r = u.w, x = u.x, y = u.y, z = u.z;
  const float dr = 2.f * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
  const float dxq = 2.f * (y * G[1] + z * G[2] + y * G[3] - 2.f * x * G[4] - r * G[5] + z * G[6] + r * G[7] - 2.f * x * G[8]);
  const float dyq = 2.f * (-2.f * y * G[0] + x * G[1] + r * G[2] + x * G[3] + z * G[5] - r * G[6] + z * G[7] - 2.f * y * G[8]);
  const float dzq = 2.f * (-2.f * z * G[0] - r * G[1] + x * G[2] + r * G[3] - 2.f * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
  const float dot = r * dr + x * dxq + y * dyq + z * dzq;
  sq[0] += (dr - r * dot) * inv; sq[1] += (dxq - x * dot) * inv; sq[2] += (dyq - y * dot) * inv; sq[3] += (dzq - z * dot) * inv;
  float* S = self7 + 7 * (size_t)i;
  S[0] = sp[0]; S[1] = sp[1]; S[2] = sp[2]; S[3] = sq[0]; S[4] = sq[1]; S[5] = sq[2]; S[6] = sq[3];
}

// Kernel 3: total gradient of point j = its own part + its incoming edges; d/dq_j -> d/d rot (q = rot * c is linear in rot).
// Writes the rows of the foreground Gaussians (the caller zero-fills the others).
__global__ __launch_bounds__(RG_BLOCK) void rigidity_bwd_gather_kernel(
    int nfg, const int64_t* __restrict__ fg_idx, const int32_t* __restrict__ rev_ptr, const int32_t* __restrict__ rev_edge,
    const float* __restrict__ prev_inv, const float* __restrict__ self7, const float* __restrict__ edge7,
    float* __restrict__ d_means3D, float* __restrict__ d_rot) {
  const int j = blockIdx.x * RG_BLOCK + threadIdx.x;
  if (j >= nfg) return;
  float a[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) a[c] = self7[7 * (size_t)j + c];
  for (int t = rev_ptr[j]; t < rev_ptr[j + 1]; ++t) {
    const float* E = edge7 + 7 * (size_t)rev_edge[t];
#pragma unroll
    for (int c = 0; c < 7; ++c) a[c] += E[c];
  }
  const size_t gj = (size_t)fg_idx[j];
  d_means3D[3 * gj] = a[0]; d_means3D[3 * gj + 1] = a[1]; d_means3D[3 * gj + 2] = a[2];
  const Quat c = load_q(prev_inv, j);
  const float gw = a[3], gx = a[4], gy = a[5], gz = a[6];
  d_rot[4 * gj + 0] = gw * c.w + gx * c.x + gy * c.y + gz * c.z;
  d_rot[4 * gj + 1] = -gw * c.x + gx * c.w - gy * c.z + gz * c.y;
  d_rot[4 * gj + 2] = -gw * c.y + gx * c.z + gy * c.w - gz * c.x;
  d_rot[4 * gj + 3] = -gw * c.z - gx * c.y + gy * c.x + gz * c.w;
}

}  // namespace

int gsr_launch_rigidity_fwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, float* partial,
                            hipStream_t st) {
  if (nfg <= 0) return 0;
  { GSR_PROF("rigidity_fwd", st);
    hipLaunchKernelGGL(rigidity_fwd_kernel, dim3((nfg + RG_BLOCK - 1) / RG_BLOCK), dim3(RG_BLOCK), 0, st, nfg, K, means3D, rot, fg_idx,
                       nbr, nw, nd, prev_inv, prev_off, partial); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_rigidity_bwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, const float* g,
                            const int32_t* rev_ptr, const int32_t* rev_edge, float* self7, float* edge7, float* d_means3D,
                            float* d_rot, hipStream_t st) {
  if (nfg <= 0) return 0;
  const dim3 grid((nfg + RG_BLOCK - 1) / RG_BLOCK), block(RG_BLOCK);
  { GSR_PROF("rigidity_bwd_edges", st);
    hipLaunchKernelGGL(rigidity_bwd_edges_kernel, grid, block, 0, st, nfg, K, means3D, rot, fg_idx, nbr, nw, nd, prev_inv, prev_off, g,
                       self7, edge7); }
  GSR_HIP_CHECK(hipGetLastError());
  { GSR_PROF("rigidity_bwd_gather", st);
    hipLaunchKernelGGL(rigidity_bwd_gather_kernel, grid, block, 0, st, nfg, fg_idx, rev_ptr, rev_edge, prev_inv, self7, edge7,
                       d_means3D, d_rot); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
