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
// each divided by N_fg * K (torch .mean()).  Forward: one thread per EDGE, one partial sum per block and term (added up by the
// caller or the finishing kernel: no atomics).  Backward: kernel 2 (a lane group per point, a lane per edge) recomputes the
// edge terms, reduces over the group what flows to point i itself and writes what flows to
// the neighbour j (d/dp_j: 3 floats, d/dq_j: 4 floats) into an edge-major buffer; kernel 3 gathers a point's incoming edges
// through a reverse adjacency (CSR, built once per timestep by the caller) and applies the chain rule to the rotation
// input -- deterministic, no float atomics.
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_rigidity::<kernel>, not "(anonymous namespace)"
namespace gsr_rigidity {

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
#define RG_PTS 32          // foreground points per forward block (their K edges are contiguous)

struct PointFrame { float px, py, pz; Quat q; float inv; float R[9]; };

__device__ __forceinline__ PointFrame point_frame(const float* __restrict__ means3D, const float* __restrict__ rot,
                                                  const int64_t* __restrict__ fg_idx, const float* __restrict__ prev_inv, int i) {
  PointFrame f;
  const size_t gi = (size_t)fg_idx[i];
  f.px = means3D[3 * gi]; f.py = means3D[3 * gi + 1]; f.pz = means3D[3 * gi + 2];
  f.q = qmul(load_q(rot, gi), load_q(prev_inv, i));
  f.inv = 1.0f / sqrtf(f.q.w * f.q.w + f.q.x * f.q.x + f.q.y * f.q.y + f.q.z * f.q.z);
  rotmat(Quat{f.q.w * f.inv, f.q.x * f.inv, f.q.y * f.inv, f.q.z * f.inv}, f.R);
  return f;
}

// Optional pre-pass: one 64-byte frame per foreground point -- {p, q.w | q.xyz, R0 | R1..R4 | R5..R8} with q = rot * prev_inv (not
// normalised) and R the rotation matrix of q / |q| -- so that an edge thread reads its own point's frame (shared by the K
// adjacent lanes) and half a line of its neighbour's (p, q) by foreground RANK: no fg_idx indirection (one dependent load less
// per edge), no quaternion products per edge.
__global__ __launch_bounds__(RG_BLOCK) void rigidity_frames_kernel(int nfg, const float* __restrict__ means3D, const float* __restrict__ rot,
                                                                   const int64_t* __restrict__ fg_idx, const float* __restrict__ prev_inv,
                                                                   float4* __restrict__ frames) {
  const int i = blockIdx.x * RG_BLOCK + threadIdx.x;
  if (i >= nfg) return;
  const PointFrame f = point_frame(means3D, rot, fg_idx, prev_inv, i);
  frames[4 * (size_t)i] = make_float4(f.px, f.py, f.pz, f.q.w);
  frames[4 * (size_t)i + 1] = make_float4(f.q.x, f.q.y, f.q.z, f.R[0]);
  frames[4 * (size_t)i + 2] = make_float4(f.R[1], f.R[2], f.R[3], f.R[4]);
  frames[4 * (size_t)i + 3] = make_float4(f.R[5], f.R[6], f.R[7], f.R[8]);
}

__device__ __forceinline__ PointFrame load_frame(const float4* __restrict__ frames, int i) {
  const float4 a = frames[4 * (size_t)i], b = frames[4 * (size_t)i + 1], c = frames[4 * (size_t)i + 2], d = frames[4 * (size_t)i + 3];
  PointFrame f;
  f.px = a.x; f.py = a.y; f.pz = a.z;
  f.q = Quat{a.w, b.x, b.y, b.z};
  f.inv = 1.0f / sqrtf(f.q.w * f.q.w + f.q.x * f.q.x + f.q.y * f.q.y + f.q.z * f.q.z);
  f.R[0] = b.w; f.R[1] = c.x; f.R[2] = c.y; f.R[3] = c.z; f.R[4] = c.w; f.R[5] = d.x; f.R[6] = d.y; f.R[7] = d.z; f.R[8] = d.w;
  return f;
}

// position and relative quaternion of neighbour j (foreground rank)
template <bool FRAMES>
__device__ __forceinline__ void load_neighbour(const float4* __restrict__ frames, const float* __restrict__ means3D,
                                               const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
                                               const float* __restrict__ prev_inv, int j, float& x, float& y, float& z, Quat& qj) {
  if (FRAMES) {
    const float4 a = frames[4 * (size_t)j], b = frames[4 * (size_t)j + 1];
    x = a.x; y = a.y; z = a.z;
    qj = Quat{a.w, b.x, b.y, b.z};
  } else {
    const size_t gj = (size_t)fg_idx[j];
    x = means3D[3 * gj]; y = means3D[3 * gj + 1]; z = means3D[3 * gj + 2];
    qj = qmul(load_q(rot, gj), load_q(prev_inv, j));
  }
}

// Forward: one thread per EDGE (~1.4 M independent threads hide the latency of the neighbour gathers, which one thread per
// point with a serial K loop cannot).  A block owns RG_PTS consecutive points, i.e. RG_PTS * K consecutive edges.
template <bool FRAMES>
__global__ __launch_bounds__(RG_BLOCK) void rigidity_fwd_kernel(
    int nfg, int K, const float* __restrict__ means3D, const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
    const int64_t* __restrict__ nbr, const float* __restrict__ nw, const float* __restrict__ nd,
    const float* __restrict__ prev_inv, const float* __restrict__ prev_off, const float4* __restrict__ frames,
    float* __restrict__ partial /*[3][blocks]*/) {
  __shared__ float red[3][RG_BLOCK / 64];
  const size_t e0 = (size_t)blockIdx.x * RG_PTS * K;
  const size_t e1 = min(e0 + (size_t)RG_PTS * K, (size_t)nfg * K);
  float l1 = 0.f, l2 = 0.f, l3 = 0.f;
  for (size_t e = e0 + threadIdx.x; e < e1; e += RG_BLOCK) {
    const int i = (int)(e / K);
    const PointFrame f = FRAMES ? load_frame(frames, i) : point_frame(means3D, rot, fg_idx, prev_inv, i);
    const float* R = f.R;
    const int j = (int)nbr[e];
    float nx, ny, nz;
    Quat qj;
    load_neighbour<FRAMES>(frames, means3D, rot, fg_idx, prev_inv, j, nx, ny, nz, qj);
    const float ox = nx - f.px, oy = ny - f.py, oz = nz - f.pz;
    const float w = nw[e];
    const float dx = (ox * R[0] + oy * R[3] + oz * R[6]) - prev_off[3 * e];
    const float dy = (ox * R[1] + oy * R[4] + oz * R[7]) - prev_off[3 * e + 1];
    const float dz = (ox * R[2] + oy * R[5] + oz * R[8]) - prev_off[3 * e + 2];
    l1 += sqrtf((dx * dx + dy * dy + dz * dz) * w + 1e-20f);
    const float ew = qj.w - f.q.w, ex = qj.x - f.q.x, ey = qj.y - f.q.y, ez = qj.z - f.q.z;
    l2 += sqrtf((ew * ew + ex * ex + ey * ey + ez * ez) * w + 1e-20f);
    const float t = sqrtf(ox * ox + oy * oy + oz * oz + 1e-20f) - nd[e];
    l3 += sqrtf(t * t * w + 1e-20f);
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

// Kernel 2: edge terms again, now for the gradient, GRP lanes per point (GRP = power of two >= K up to 64; longer lists are
// strided).  g1..g3 = upstream gradients of the three means, already divided by N_fg * K.
// self[i] = {d/dp_i (3), d/dq_i (4)} from point i's own edges; edge[e] = {d/dp_j (3), d/dq_j (4)}.
template <int GRP, bool FRAMES>
__global__ __launch_bounds__(RG_BLOCK) void rigidity_bwd_edges_kernel(
    int nfg, int K, const float* __restrict__ means3D, const float* __restrict__ rot, const int64_t* __restrict__ fg_idx,
    const int64_t* __restrict__ nbr, const float* __restrict__ nw, const float* __restrict__ nd,
    const float* __restrict__ prev_inv, const float* __restrict__ prev_off, const float* __restrict__ g, int gstride,
    float s1, float s2, float s3, const float4* __restrict__ frames, float* __restrict__ self7, float* __restrict__ edge7) {
  const int i = (blockIdx.x * RG_BLOCK + threadIdx.x) / GRP, lg = threadIdx.x & (GRP - 1);
  const bool live = i < nfg;                       // whole groups are live or dead; dead lanes still take part in the shuffles
  const float g1 = g[0] * s1, g2 = g[gstride] * s2, g3 = g[2 * gstride] * s3;
  PointFrame f{};
  if (live) f = FRAMES ? load_frame(frames, i) : point_frame(means3D, rot, fg_idx, prev_inv, i);
  const float* R = f.R;
  float acc[16];                                   // G[9] = d loss / d R_i, sp[3], sq[4]
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  if (live) {
    for (int k = lg; k < K; k += GRP) {
      const size_t e = (size_t)i * K + k;
      const int j = (int)nbr[e];
      float nx, ny, nz;
      Quat qj;
      load_neighbour<FRAMES>(frames, means3D, rot, fg_idx, prev_inv, j, nx, ny, nz, qj);
      const float ox = nx - f.px, oy = ny - f.py, oz = nz - f.pz;
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
      acc[0] += ox * ax; acc[1] += ox * ay; acc[2] += ox * az;
      acc[3] += oy * ax; acc[4] += oy * ay; acc[5] += oy * az;
      acc[6] += oz * ax; acc[7] += oz * ay; acc[8] += oz * az;
      // iso
      const float mag = sqrtf(ox * ox + oy * oy + oz * oz + 1e-20f);
      const float t = mag - nd[e];
      const float c3 = g3 * w * t / sqrtf(t * t * w + 1e-20f) / mag;
      fx += c3 * ox; fy += c3 * oy; fz += c3 * oz;
      // rot
      const float ew = qj.w - f.q.w, ex = qj.x - f.q.x, ey = qj.y - f.q.y, ez = qj.z - f.q.z;
      const float c2 = g2 * w / sqrtf((ew * ew + ex * ex + ey * ey + ez * ez) * w + 1e-20f);
      const float hw = c2 * ew, hx = c2 * ex, hy = c2 * ey, hz = c2 * ez;
      if (FRAMES) {   // fused path: 32-byte records, two aligned 16-byte stores (a record never straddles a 64-byte line)
        float4* E = reinterpret_cast<float4*>(edge7) + 2 * e;
        E[0] = make_float4(fx, fy, fz, hw);
        E[1] = make_float4(hx, hy, hz, 0.f);
      } else {
        float* E = edge7 + 7 * e;
        E[0] = fx; E[1] = fy; E[2] = fz; E[3] = hw; E[4] = hx; E[5] = hy; E[6] = hz;
      }
      acc[9] -= fx; acc[10] -= fy; acc[11] -= fz;
      acc[12] -= hw; acc[13] -= hx; acc[14] -= hy; acc[15] -= hz;
    }
  }
#pragma unroll
  for (int m = GRP / 2; m >= 1; m >>= 1) {
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] += __shfl_xor(acc[c], m, 64);
  }
  if (!live || lg != 0) return;
  // d loss / d u from G (R as a function of the unit quaternion u), then through the normalisation q -> q / |q|
  const float* G = acc;
  const float inv = f.inv;
  const float r = f.q.w * inv, x = f.q.x * inv, y = f.q.y * inv, z = f.q.z * inv;
  const float dr = 2.f * (-z * G[1] + y * G[2] + z * G[3] - x * G[5] - y * G[6] + x * G[7]);
  const float dxq = 2.f * (y * G[1] + z * G[2] + y * G[3] - 2.f * x * G[4] - r * G[5] + z * G[6] + r * G[7] - 2.f * x * G[8]);
  const float dyq = 2.f * (-2.f * y * G[0] + x * G[1] + r * G[2] + x * G[3] + z * G[5] - r * G[6] + z * G[7] - 2.f * y * G[8]);
  const float dzq = 2.f * (-2.f * z * G[0] - r * G[1] + x * G[2] + r * G[3] - 2.f * z * G[4] + y * G[5] + x * G[6] + y * G[7]);
  const float dot = r * dr + x * dxq + y * dyq + z * dzq;
  float* S = self7 + (FRAMES ? 8 : 7) * (size_t)i;
  S[0] = acc[9]; S[1] = acc[10]; S[2] = acc[11];
  S[3] = acc[12] + (dr - r * dot) * inv; S[4] = acc[13] + (dxq - x * dot) * inv;
  S[5] = acc[14] + (dyq - y * dot) * inv; S[6] = acc[15] + (dzq - z * dot) * inv;
}

// Kernel 3: total gradient of point j = its own part + its incoming edges (8 lanes per point stride over the reverse
// adjacency); d/dq_j -> d/d rot (q = rot * c is linear in rot).  Writes the rows of the foreground Gaussians (the caller
// zero-fills the others).  Fixed lane assignment and reduction order: deterministic.
#define RG_GATHER 8
template <int REC>   // floats per record: 7 (standalone entry point) or 8 (fused path, 16-byte aligned)
__global__ __launch_bounds__(RG_BLOCK) void rigidity_bwd_gather_kernel(
    int nfg, const int64_t* __restrict__ fg_idx, const int32_t* __restrict__ rev_ptr, const int32_t* __restrict__ rev_edge,
    const float* __restrict__ prev_inv, const float* __restrict__ self7, const float* __restrict__ edge7,
    float* __restrict__ d_means3D, float* __restrict__ d_rot, int accumulate) {
  const int j = (blockIdx.x * RG_BLOCK + threadIdx.x) / RG_GATHER, lg = threadIdx.x & (RG_GATHER - 1);
  const bool live = j < nfg;
  float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (live) {
    const int t1 = rev_ptr[j + 1];
    for (int t = rev_ptr[j] + lg; t < t1; t += RG_GATHER) {
      if (REC == 8) {
        const float4* E = reinterpret_cast<const float4*>(edge7) + 2 * (size_t)rev_edge[t];
        const float4 e0 = E[0], e1 = E[1];
        a[0] += e0.x; a[1] += e0.y; a[2] += e0.z; a[3] += e0.w; a[4] += e1.x; a[5] += e1.y; a[6] += e1.z;
      } else {
        const float* E = edge7 + 7 * (size_t)rev_edge[t];
#pragma unroll
        for (int c = 0; c < 7; ++c) a[c] += E[c];
      }
    }
  }
#pragma unroll
  for (int m = RG_GATHER / 2; m >= 1; m >>= 1) {
#pragma unroll
    for (int c = 0; c < 7; ++c) a[c] += __shfl_xor(a[c], m, 64);
  }
  if (!live || lg != 0) return;
#pragma unroll
  for (int c = 0; c < 7; ++c) a[c] += self7[REC * (size_t)j + c];
  const size_t gj = (size_t)fg_idx[j];
  const Quat c = load_q(prev_inv, j);
  const float gw = a[3], gx = a[4], gy = a[5], gz = a[6];
  float o[7] = {a[0], a[1], a[2], gw * c.w + gx * c.x + gy * c.y + gz * c.z, -gw * c.x + gx * c.w - gy * c.z + gz * c.y,
                -gw * c.y + gx * c.z + gy * c.w - gz * c.x, -gw * c.z - gx * c.y + gy * c.x + gz * c.w};
  if (accumulate) {   // on top of what the caller already holds there (the rasterizer's gradient)
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] += d_means3D[3 * gj + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[3 + k] += d_rot[4 * gj + k];
  }
  d_means3D[3 * gj] = o[0]; d_means3D[3 * gj + 1] = o[1]; d_means3D[3 * gj + 2] = o[2];
  d_rot[4 * gj + 0] = o[3]; d_rot[4 * gj + 1] = o[4]; d_rot[4 * gj + 2] = o[5]; d_rot[4 * gj + 3] = o[6];
}

}  // namespace gsr_rigidity
using namespace gsr_rigidity;

int gsr_rigidity_fwd_blocks(int nfg) { return nfg > 0 ? (nfg + RG_PTS - 1) / RG_PTS : 0; }

int gsr_launch_rigidity_fwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, float* frames,
                            float* partial, hipStream_t st) {
  if (nfg <= 0) return 0;
  float4* fr = reinterpret_cast<float4*>(frames);
  if (fr) {
    GSR_PROF("rigidity_frames", st);
    hipLaunchKernelGGL(rigidity_frames_kernel, dim3((nfg + RG_BLOCK - 1) / RG_BLOCK), dim3(RG_BLOCK), 0, st, nfg, means3D, rot, fg_idx,
                       prev_inv, fr);
  }
  { GSR_PROF("rigidity_fwd", st);
    if (fr) hipLaunchKernelGGL(rigidity_fwd_kernel<true>, dim3(gsr_rigidity_fwd_blocks(nfg)), dim3(RG_BLOCK), 0, st, nfg, K, means3D, rot,
                               fg_idx, nbr, nw, nd, prev_inv, prev_off, (const float4*)fr, partial);
    else hipLaunchKernelGGL(rigidity_fwd_kernel<false>, dim3(gsr_rigidity_fwd_blocks(nfg)), dim3(RG_BLOCK), 0, st, nfg, K, means3D, rot,
                            fg_idx, nbr, nw, nd, prev_inv, prev_off, (const float4*)nullptr, partial); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_rigidity_bwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, const float* g,
                            int gstride, float s1, float s2, float s3, const int32_t* rev_ptr, const int32_t* rev_edge,
                            float* frames, int frames_valid, float* self7, float* edge7, float* d_means3D, float* d_rot, int accumulate,
                            hipStream_t st) {
  if (nfg <= 0) return 0;
  float4* fr = reinterpret_cast<float4*>(frames);
  if (fr && !frames_valid) {
    GSR_PROF("rigidity_frames", st);
    hipLaunchKernelGGL(rigidity_frames_kernel, dim3((nfg + RG_BLOCK - 1) / RG_BLOCK), dim3(RG_BLOCK), 0, st, nfg, means3D, rot, fg_idx,
                       prev_inv, fr);
  }
  // a lane per edge (8 or 16 lanes per point striding over K = 20 edges measured 57 / 45 us against 42: gather latency, not issue, bounds it)
  const int grp = K <= 8 ? 8 : (K <= 16 ? 16 : (K <= 32 ? 32 : 64));
  const dim3 block(RG_BLOCK), grid(((size_t)nfg * grp + RG_BLOCK - 1) / RG_BLOCK);
  { GSR_PROF("rigidity_bwd_edges", st);
#define GSR_RG_LAUNCH(G_)                                                                                                              \
  do {                                                                                                                                 \
    if (fr) hipLaunchKernelGGL((rigidity_bwd_edges_kernel<G_, true>), grid, block, 0, st, nfg, K, means3D, rot, fg_idx, nbr, nw, nd,   \
                               prev_inv, prev_off, g, gstride, s1, s2, s3, (const float4*)fr, self7, edge7);                           \
    else hipLaunchKernelGGL((rigidity_bwd_edges_kernel<G_, false>), grid, block, 0, st, nfg, K, means3D, rot, fg_idx, nbr, nw, nd,     \
                            prev_inv, prev_off, g, gstride, s1, s2, s3, (const float4*)nullptr, self7, edge7);                         \
  } while (0)
    if (grp == 8) GSR_RG_LAUNCH(8); else if (grp == 16) GSR_RG_LAUNCH(16); else if (grp == 32) GSR_RG_LAUNCH(32); else GSR_RG_LAUNCH(64);
#undef GSR_RG_LAUNCH
  }
  GSR_HIP_CHECK(hipGetLastError());
  { GSR_PROF("rigidity_bwd_gather", st);
    const dim3 ggrid(((size_t)nfg * RG_GATHER + RG_BLOCK - 1) / RG_BLOCK);
    if (fr) hipLaunchKernelGGL(rigidity_bwd_gather_kernel<8>, ggrid, block, 0, st, nfg, fg_idx, rev_ptr, rev_edge, prev_inv, self7, edge7,
                               d_means3D, d_rot, accumulate);
    else hipLaunchKernelGGL(rigidity_bwd_gather_kernel<7>, ggrid, block, 0, st, nfg, fg_idx, rev_ptr, rev_edge, prev_inv, self7, edge7,
                            d_means3D, d_rot, accumulate); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
