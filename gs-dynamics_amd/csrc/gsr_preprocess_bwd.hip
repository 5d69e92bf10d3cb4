// gsr_preprocess_bwd.hip -- per-Gaussian backward for gfx950 (replaces the
// reference extension's computeCov2D-backward + preprocess-backward kernels, fused into one).
//
// One lane per Gaussian:
//   1. reduce its entry records (one 36-byte record per touched tile, contiguous, written by
//      render_bwd) in ascending tile order -> dL/d{mean2D_pix, conic(A,B,C), opacity, rgb};
//   2. conic -> cov2D -> (cov3D, view-space mean) -> (scale, rotation, mean3D);  the 3D covariance is
//      recomputed from scale/rotation rather than stored by the forward (saves 24 B/Gaussian each way);
//   3. 2D mean -> 3D mean through the 4x4 projection; SH backward when colours came from SH.
// Every output element is written (zeros for culled Gaussians): callers need no memset.
// HBM per Gaussian: reads 36*tiles + 12 + 12 + 16 + 8 + 4 + 4, writes 12+12+12+4+12+16+24 = 92 B.
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_preprocess_bwd::<kernel>, not "(anonymous namespace)"
namespace gsr_preprocess_bwd {

__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                             -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                             0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                             -0.5900435899266435f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// SH backward for one Gaussian: writes dL_dsh (all M coefficients; inactive ones get 0) and returns
// the gradient w.r.t. the mean through the view direction.
__device__ void sh_backward(int deg, int M, const float* __restrict__ sh, float3 p, const float* __restrict__ campos,
                            uint32_t clamped, float g0, float g1, float g2, float* __restrict__ dsh, float dmean[3]) {
  float dL[3] = {(clamped & 1u) ? 0.f : g0, (clamped & 2u) ? 0.f : g1, (clamped & 4u) ? 0.f : g2};
  const float ox = p.x - campos[0], oy = p.y - campos[1], oz = p.z - campos[2];
  const float len = sqrtf(ox * ox + oy * oy + oz * oz), inv = 1.0f / len;
  const float x = ox * inv, y = oy * inv, z = oz * inv;
  float basis[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) basis[k] = 0.f;
  float dRdx[3] = {0.f, 0.f, 0.f}, dRdy[3] = {0.f, 0.f, 0.f}, dRdz[3] = {0.f, 0.f, 0.f};
  basis[0] = SH_C0;
  if (deg > 0) {
    basis[1] = -SH_C1 * y; basis[2] = SH_C1 * z; basis[3] = -SH_C1 * x;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      dRdx[ch] = -SH_C1 * sh[3 * 3 + ch]; dRdy[ch] = -SH_C1 * sh[1 * 3 + ch]; dRdz[ch] = SH_C1 * sh[2 * 3 + ch];
    }
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      basis[4] = kC2[0] * xy; basis[5] = kC2[1] * yz; basis[6] = kC2[2] * (2.0f * zz - xx - yy);
      basis[7] = kC2[3] * xz; basis[8] = kC2[4] * (xx - yy);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[(k)*3 + ch]
        dRdx[ch] += kC2[0] * y * S(4) - 2.0f * kC2[2] * x * S(6) + kC2[3] * z * S(7) + 2.0f * kC2[4] * x * S(8);
        dRdy[ch] += kC2[0] * x * S(4) + kC2[1] * z * S(5) - 2.0f * kC2[2] * y * S(6) - 2.0f * kC2[4] * y * S(8);
        dRdz[ch] += kC2[1] * y * S(5) + 4.0f * kC2[2] * z * S(6) + kC2[3] * x * S(7);
#undef S
      }
      if (deg > 2) {
        basis[9] = kC3[0] * y * (3.0f * xx - yy); basis[10] = kC3[1] * xy * z;
        basis[11] = kC3[2] * y * (4.0f * zz - xx - yy); basis[12] = kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        basis[13] = kC3[4] * x * (4.0f * zz - xx - yy); basis[14] = kC3[5] * z * (xx - yy);
        basis[15] = kC3[6] * x * (xx - 3.0f * yy);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
#define S(k) sh[(k)*3 + ch]
          dRdx[ch] += kC3[0] * S(9) * 6.0f * xy + kC3[1] * S(10) * yz - kC3[2] * S(11) * 2.0f * xy -
                      kC3[3] * S(12) * 6.0f * xz + kC3[4] * S(13) * (-3.0f * xx + 4.0f * zz - yy) +
                      kC3[5] * S(14) * 2.0f * xz + kC3[6] * S(15) * 3.0f * (xx - yy);
          dRdy[ch] += kC3[0] * S(9) * 3.0f * (xx - yy) + kC3[1] * S(10) * xz +
                      kC3[2] * S(11) * (-3.0f * yy + 4.0f * zz - xx) - kC3[3] * S(12) * 6.0f * yz -
                      kC3[4] * S(13) * 2.0f * xy - kC3[5] * S(14) * 2.0f * yz - kC3[6] * S(15) * 6.0f * xy;
          dRdz[ch] += kC3[1] * S(10) * xy + kC3[2] * S(11) * 8.0f * yz +
                      kC3[3] * S(12) * 3.0f * (2.0f * zz - xx - yy) + kC3[4] * S(13) * 8.0f * xz +
                      kC3[5] * S(14) * (xx - yy);
#undef S
        }
      }
    }
  }
  const int ncoef = (deg + 1) * (deg + 1);
  for (int k = 0; k < M; ++k) {
    const float bk = k < ncoef ? basis[k < 16 ? k : 15] : 0.f;
    dsh[k * 3 + 0] = bk * dL[0]; dsh[k * 3 + 1] = bk * dL[1]; dsh[k * 3 + 2] = bk * dL[2];
  }
  const float ddx = dRdx[0] * dL[0] + dRdx[1] * dL[1] + dRdx[2] * dL[2];
  const float ddy = dRdy[0] * dL[0] + dRdy[1] * dL[1] + dRdy[2] * dL[2];
  const float ddz = dRdz[0] * dL[0] + dRdz[1] * dL[1] + dRdz[2] * dL[2];
  const float sum2 = ox * ox + oy * oy + oz * oz;
  const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
  dmean[0] = ((sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
  dmean[1] = (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
  dmean[2] = (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
}

// ---- shared pieces ---------------------------------------------------------------------------------
struct PartialSum { float gmx, gmy, gA, gB, gC, gop, dr, dg, db; };

// Sum of a Gaussian's entry records (one per touched tile, contiguous, ascending tile order).
// `col` = false: the blend backward wrote the six geometry sums only (24 of the 36 bytes; no colour gradient wanted).
__device__ __forceinline__ PartialSum reduce_partials(const float4* __restrict__ partials, uint32_t e0, uint32_t e1, bool col = true) {
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  float r2x = 0.f;
  // FOUR records per trip, all of their loads issued before the first is used: the walk is a chain of memory round trips (a lane's
  // records are contiguous, the lanes of a wave are ~150 bytes apart), and the trip count -- the longest rect of the wave -- was the
  // kernel's time.  The additions keep their order (record e0 first).
  constexpr int RU = 4;
  // `col` = false: the blend backward wrote the six geometry sums only (24 of the 36 bytes).  The walk reads whole records all the same
  // -- the same cache lines, and the three-load form measured FASTER than a two-load one (52.5 vs 58 us at 8 views) -- the colour sums
  // then hold whatever the scratch held and are zeroed below (garbage must not propagate).
  for (uint32_t e = e0; e < e1; e += RU) {
    float4 q0[RU], q1[RU];
    float q2x[RU];
#pragma unroll
    for (int k = 0; k < RU; ++k) if (e + k < e1) gsr_load_partial(partials, e + k, q0[k], q1[k], q2x[k]);
#pragma unroll
    for (int k = 0; k < RU; ++k)
      if (e + k < e1) {
        r0.x += q0[k].x; r0.y += q0[k].y; r0.z += q0[k].z; r0.w += q0[k].w;
        r1.x += q1[k].x; r1.y += q1[k].y; r1.z += q1[k].z; r1.w += q1[k].w;
        r2x += q2x[k];
      }
  }
  PartialSum ps;
  // the blend backward stores the conic partials without their constant factors (dA: -1/2, dB: -1, dC: -1/2)
  ps.gmx = r0.x; ps.gmy = r0.y; ps.gA = -0.5f * r0.z; ps.gB = -r0.w; ps.gC = -0.5f * r1.x; ps.gop = r1.y;
  // !col: those twelve bytes were never written (possibly NaN patterns): they must not reach any sum or LDS slot
  ps.dr = col ? r1.z : 0.f; ps.dg = col ? r1.w : 0.f; ps.db = col ? r2x : 0.f;
  return ps;
}

// The per-Gaussian chain (3D covariance, conic -> cov2D -> cov3D / view-space mean, cov3D -> scale / rotation) runs in DOUBLE precision
// (round 6).  It is a chain of differences of nearly equal products -- det = a c - b b, the inverse of the 2D covariance, the congruence
// with T, the quaternion terms -- and for a Gaussian that is elongated on screen (a nearly singular conic) fp32 loses 1e-4 ... 6e-3 of that
// Gaussian's gradient row THERE, in any operation order: the fp32 oracle's rows are as far from its fp64 build as this kernel's were.  Inputs
// (the record sums of the blend backward, the parameters) and outputs stay fp32; in between, fp64 costs this latency-bound kernel 24 -> 29 us
// at four views, +1 us at one (MI355X issues fp64 FMAs at the fp32 rate; the price is registers: 166 instead of 90 VGPRs), and puts the
// parity soak's tail cases 3 - 40 x closer to the fp64 oracle than the fp32 oracle is (profiles/r06_fp64_chain.txt): seed 9 case 895
// scales 1.03e-4 -> 1.2e-5, GSR_SOAK_BIG seed 6 case 156 worst row 6.2e-3 -> 7.4e-5.
typedef double real;
// Rotation matrix, scaled axes and 3D covariance of one Gaussian (view independent).
struct Cov3 { real R[3][3]; real s[3]; real q[4]; real c[6]; };
__device__ __forceinline__ void build_cov3(int i, float mod, const float* __restrict__ scales,
                                           const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
                                           Cov3& o) {
  if (cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.c[k] = cov3D_precomp[6 * i + k];
#pragma unroll
    for (int a = 0; a < 3; ++a) { o.s[a] = 0.f; for (int b = 0; b < 3; ++b) o.R[a][b] = 0.f; }
    o.q[0] = o.q[1] = o.q[2] = o.q[3] = 0.f;
    return;
  }
  const real r = rotations[4 * i], x = rotations[4 * i + 1], y = rotations[4 * i + 2], z = rotations[4 * i + 3];
  o.q[0] = r; o.q[1] = x; o.q[2] = y; o.q[3] = z;
  o.R[0][0] = 1.f - 2.f * (y * y + z * z); o.R[0][1] = 2.f * (x * y - r * z); o.R[0][2] = 2.f * (x * z + r * y);
  o.R[1][0] = 2.f * (x * y + r * z); o.R[1][1] = 1.f - 2.f * (x * x + z * z); o.R[1][2] = 2.f * (y * z - r * x);
  o.R[2][0] = 2.f * (x * z - r * y); o.R[2][1] = 2.f * (y * z + r * x); o.R[2][2] = 1.f - 2.f * (x * x + y * y);
  o.s[0] = (real)mod * scales[3 * i]; o.s[1] = (real)mod * scales[3 * i + 1]; o.s[2] = (real)mod * scales[3 * i + 2];
  real Mm[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) Mm[a][b] = o.R[a][b] * o.s[b];
  o.c[0] = Mm[0][0] * Mm[0][0] + Mm[0][1] * Mm[0][1] + Mm[0][2] * Mm[0][2];
  o.c[1] = Mm[0][0] * Mm[1][0] + Mm[0][1] * Mm[1][1] + Mm[0][2] * Mm[1][2];
  o.c[2] = Mm[0][0] * Mm[2][0] + Mm[0][1] * Mm[2][1] + Mm[0][2] * Mm[2][2];
  o.c[3] = Mm[1][0] * Mm[1][0] + Mm[1][1] * Mm[1][1] + Mm[1][2] * Mm[1][2];
  o.c[4] = Mm[1][0] * Mm[2][0] + Mm[1][1] * Mm[2][1] + Mm[1][2] * Mm[2][2];
  o.c[5] = Mm[2][0] * Mm[2][0] + Mm[2][1] * Mm[2][1] + Mm[2][2] * Mm[2][2];
}

// One view's chain: conic -> cov2D -> (cov3D, view-space mean) and 2D mean -> 3D mean.  ACCUMULATES into
// gcov[6] and gm3[3]; returns this view's dL/d(NDC mean) in gm2.
__device__ __forceinline__ void view_chain(const float* __restrict__ view, const float* __restrict__ proj, int W, int H,
                                           float tanfovx, float tanfovy, float3 p, const real c[6],
                                           const PartialSum& ps, float gcov[6], float gm3[3], float gm2[2],
                                           float sx_first = 0.f, float sy_first = 0.f, float* gm2_first = nullptr) {
  const real pvx = (real)view[0] * p.x + view[4] * p.y + view[8] * p.z + view[12];
  const real pvy = (real)view[1] * p.x + view[5] * p.y + view[9] * p.z + view[13];
  const real pvz = (real)view[2] * p.x + view[6] * p.y + view[10] * p.z + view[14];
  const real fx = (real)W / (2.0f * (real)tanfovx), fy = (real)H / (2.0f * (real)tanfovy);
  const real limx = 1.3f * (real)tanfovx, limy = 1.3f * (real)tanfovy;
  const real tz = pvz;
  const real txtz = pvx / tz, tytz = pvy / tz;
  const real xm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  const real ym = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  const real tx = (txtz < -limx ? -limx : (txtz > limx ? limx : txtz)) * tz, ty = (tytz < -limy ? -limy : (tytz > limy ? limy : tytz)) * tz;
  const real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz), J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
  const real T0[3] = {J00 * view[0] + J02 * view[2], J00 * view[4] + J02 * view[6], J00 * view[8] + J02 * view[10]};
  const real T1[3] = {J11 * view[1] + J12 * view[2], J11 * view[5] + J12 * view[6], J11 * view[9] + J12 * view[10]};
  const real S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
  real U0[3], U1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    U0[k] = S[k][0] * T0[0] + S[k][1] * T0[1] + S[k][2] * T0[2];
    U1[k] = S[k][0] * T1[0] + S[k][1] * T1[1] + S[k][2] * T1[2];
  }
  const real a = U0[0] * T0[0] + U0[1] * T0[1] + U0[2] * T0[2] + 0.3f;
  const real b = U0[0] * T1[0] + U0[1] * T1[1] + U0[2] * T1[2];
  const real cc = U1[0] * T1[0] + U1[1] * T1[1] + U1[2] * T1[2] + 0.3f;
  const real det = a * cc - b * b;
  const real d2inv = 1.0f / (det * det + 0.0000001f);
  const real gA = ps.gA, gB = ps.gB, gC = ps.gC;
  const real dL_da = d2inv * (-cc * cc * gA + b * cc * gB - b * b * gC);
  const real dL_dc = d2inv * (-b * b * gA + a * b * gB - a * a * gC);
  const real dL_db = d2inv * (2.0f * b * cc * gA - (det + 2.0f * b * b) * gB + 2.0f * a * b * gC);
  gcov[0] += (float)(T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
  gcov[3] += (float)(T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
  gcov[5] += (float)(T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
  gcov[1] += (float)(2.0f * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2.0f * T1[0] * T1[1] * dL_dc);
  gcov[2] += (float)(2.0f * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2.0f * T1[0] * T1[2] * dL_dc);
  gcov[4] += (float)(2.0f * T0[1] * T0[2] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2.0f * T1[1] * T1[2] * dL_dc);
  real dT0[3], dT1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    dT0[j] = 2.0f * U0[j] * dL_da + U1[j] * dL_db;
    dT1[j] = 2.0f * U1[j] * dL_dc + U0[j] * dL_db;
  }
  const real dJ00 = dT0[0] * view[0] + dT0[1] * view[4] + dT0[2] * view[8];
  const real dJ02 = dT0[0] * view[2] + dT0[1] * view[6] + dT0[2] * view[10];
  const real dJ11 = dT1[0] * view[1] + dT1[1] * view[5] + dT1[2] * view[9];
  const real dJ12 = dT1[0] * view[2] + dT1[1] * view[6] + dT1[2] * view[10];
  const real itz = 1.0f / tz, itz2 = itz * itz, itz3 = itz2 * itz;
  const real dtx = xm * -fx * itz2 * dJ02;
  const real dty = ym * -fy * itz2 * dJ12;
  const real dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + (2.0f * fx * tx) * itz3 * dJ02 + (2.0f * fy * ty) * itz3 * dJ12;
  // the blend backward hands over the raw sums of t*dx and t*dy (ps.gmx, ps.gmy); with the conic
  // (A, B, C) = (cc, -b, a) / det:  d/d mean2D.x = -(A sx + B sy),  d/d mean2D.y = -(C sy + B sx)
  const real det_inv = 1.0f / det;
  const real cA = cc * det_inv, cB = -b * det_inv, cC = a * det_inv;
  const real m2x = -(cA * ps.gmx + cB * ps.gmy) * 0.5f * (real)W;
  gm2[0] = (float)m2x;
  const real m2y = -(cC * ps.gmy + cB * ps.gmx) * 0.5f * (real)H;
  gm2[1] = (float)m2y;
  if (gm2_first) {   // fused pair: the share of the first view of the pair (ps.gmx / gmy are the sums over both)
    gm2_first[0] = (float)(-(cA * sx_first + cB * sy_first) * 0.5f * (real)W);
    gm2_first[1] = (float)(-(cC * sy_first + cB * sx_first) * 0.5f * (real)H);
  }
  const real hx = (real)proj[0] * p.x + proj[4] * p.y + proj[8] * p.z + proj[12];
  const real hy = (real)proj[1] * p.x + proj[5] * p.y + proj[9] * p.z + proj[13];
  const real hw = (real)proj[3] * p.x + proj[7] * p.y + proj[11] * p.z + proj[15];
  const real mw = 1.0f / (hw + 0.0000001f);
  const real mul1 = hx * mw * mw, mul2 = hy * mw * mw;
  gm3[0] += (float)(view[0] * dtx + view[1] * dty + view[2] * dtz + (proj[0] * mw - proj[3] * mul1) * m2x + (proj[1] * mw - proj[3] * mul2) * m2y);
  gm3[1] += (float)(view[4] * dtx + view[5] * dty + view[6] * dtz + (proj[4] * mw - proj[7] * mul1) * m2x + (proj[5] * mw - proj[7] * mul2) * m2y);
  gm3[2] += (float)(view[8] * dtx + view[9] * dty + view[10] * dtz + (proj[8] * mw - proj[11] * mul1) * m2x + (proj[9] * mw - proj[11] * mul2) * m2y);
}

// dL/dcov3D -> dL/dscale, dL/drotation (linear in gcov: in the multi-view kernel it runs once on the sum).
__device__ __forceinline__ void cov3_to_scale_rot(const Cov3& cv, float mod, const float gcov[6], float gs[3], float gq[4]) {
  const real dS[3][3] = {{(real)gcov[0], 0.5f * (real)gcov[1], 0.5f * (real)gcov[2]},
                          {0.5f * (real)gcov[1], (real)gcov[3], 0.5f * (real)gcov[4]},
                          {0.5f * (real)gcov[2], 0.5f * (real)gcov[4], (real)gcov[5]}};
  real G[3][3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const real dM0 = 2.0f * (dS[0][0] * cv.R[0][jj] + dS[0][1] * cv.R[1][jj] + dS[0][2] * cv.R[2][jj]) * cv.s[jj];
    const real dM1 = 2.0f * (dS[1][0] * cv.R[0][jj] + dS[1][1] * cv.R[1][jj] + dS[1][2] * cv.R[2][jj]) * cv.s[jj];
    const real dM2 = 2.0f * (dS[2][0] * cv.R[0][jj] + dS[2][1] * cv.R[1][jj] + dS[2][2] * cv.R[2][jj]) * cv.s[jj];
    gs[jj] = (float)((real)mod * (cv.R[0][jj] * dM0 + cv.R[1][jj] * dM1 + cv.R[2][jj] * dM2));
    G[0][jj] = dM0 * cv.s[jj]; G[1][jj] = dM1 * cv.s[jj]; G[2][jj] = dM2 * cv.s[jj];
  }
  const real r = cv.q[0], x = cv.q[1], y = cv.q[2], z = cv.q[3];
  gq[0] = (float)(2.0f * (-z * G[0][1] + y * G[0][2] + z * G[1][0] - x * G[1][2] - y * G[2][0] + x * G[2][1]));
  gq[1] = (float)(2.0f * (y * G[0][1] + z * G[0][2] + y * G[1][0] - 2.0f * x * G[1][1] - r * G[1][2] + z * G[2][0] + r * G[2][1] - 2.0f * x * G[2][2]));
  gq[2] = (float)(2.0f * (-2.0f * y * G[0][0] + x * G[0][1] + r * G[0][2] + x * G[1][0] + z * G[1][2] - r * G[2][0] + z * G[2][1] - 2.0f * y * G[2][2]));
  gq[3] = (float)(2.0f * (-2.0f * z * G[0][0] - r * G[0][1] + x * G[0][2] + r * G[1][0] - 2.0f * z * G[1][1] + y * G[1][2] + x * G[2][0] + y * G[2][1]));
}

// The tracking forward marks the Gaussians some pixel of the view blended (GeomState::used; trusted when the view's `tracked` word is
// set): the records of the others are all zeros -- 45 % of a benchmark view's entries -- and so is everything the chain rule would make
// of them (exact zeros: every term carries a factor from the records): skipped.
__device__ __forceinline__ bool gsr_view_used(const GsrBwdView& w, int i) {
  return !w.used || !w.tracked || *w.tracked == 0u || w.used[i] != 0;
}

// ---- single view ------------------------------------------------------------------------------------
template <bool USE_SH>
__global__ __launch_bounds__(GSR_BLOCK) void preprocess_bwd_kernel(
    int P, int W, int H, float tanfovx, float tanfovy, float mod, int sh_degree, int M,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos,
    const float* __restrict__ means3D, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ colors_precomp, const float* __restrict__ shs, const float* __restrict__ cov3D_precomp,
    const int32_t* __restrict__ radii, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ clamped,
    const float4* __restrict__ partials, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dmeans2D,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drot, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    const uint8_t* __restrict__ used, const uint32_t* __restrict__ tracked, const uint32_t* __restrict__ bwd_error) {
  const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (i >= P) return;
  float gm3[3] = {0.f, 0.f, 0.f}, gm2[2] = {0.f, 0.f}, gcol[3] = {0.f, 0.f, 0.f}, gop = 0.f;
  float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool alive = radii[i] > 0 && (!used || !tracked || *tracked == 0u || used[i] != 0);   // (see gsr_view_used)
  if (USE_SH && !alive && dL_dsh) {
    for (int k = 0; k < M * 3; ++k) dL_dsh[(size_t)i * M * 3 + k] = 0.f;
  }
  if (alive) {
    const PartialSum ps = reduce_partials(partials, offsets[i], offsets[i + 1], USE_SH || dL_dcolors != nullptr);
    gop = ps.gop;
    const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    if (USE_SH) {
      float dmean_sh[3] = {0.f, 0.f, 0.f};
      sh_backward(sh_degree, M, shs + (size_t)i * M * 3, p, campos, clamped[i], ps.dr, ps.dg, ps.db,
                  dL_dsh + (size_t)i * M * 3, dmean_sh);
      gm3[0] = dmean_sh[0]; gm3[1] = dmean_sh[1]; gm3[2] = dmean_sh[2];
    } else {
      gcol[0] = ps.dr; gcol[1] = ps.dg; gcol[2] = ps.db;
    }
    Cov3 cv;
    build_cov3(i, mod, scales, rotations, cov3D_precomp, cv);
    view_chain(view, proj, W, H, tanfovx, tanfovy, p, cv.c, ps, gcov, gm3, gm2);
    if (!cov3D_precomp) cov3_to_scale_rot(cv, mod, gcov, gs, gq);
  }
  if (bwd_error && *bwd_error != 0u) gm3[0] = gm3[1] = gm3[2] = __builtin_nanf("");   // the blend backward of this call aborted: loud, not garbage (GSR_QUEUE_BWD_ERROR)
  dL_dmeans3D[3 * i] = gm3[0]; dL_dmeans3D[3 * i + 1] = gm3[1]; dL_dmeans3D[3 * i + 2] = gm3[2];
  dL_dmeans2D[3 * i] = gm2[0]; dL_dmeans2D[3 * i + 1] = gm2[1]; dL_dmeans2D[3 * i + 2] = 0.f;
  if (dL_dcolors) { dL_dcolors[3 * i] = gcol[0]; dL_dcolors[3 * i + 1] = gcol[1]; dL_dcolors[3 * i + 2] = gcol[2]; }
  dL_dopacity[i] = gop;
  if (dL_dscales) { dL_dscales[3 * i] = gs[0]; dL_dscales[3 * i + 1] = gs[1]; dL_dscales[3 * i + 2] = gs[2]; }
  if (dL_drot) { dL_drot[4 * i] = gq[0]; dL_drot[4 * i + 1] = gq[1]; dL_drot[4 * i + 2] = gq[2]; dL_drot[4 * i + 3] = gq[3]; }
  if (dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gcov[k];
  }
}

// ---- all views of a step at once (precomputed colours) ----------------------------------------------
// One lane per Gaussian loops over the V views: per view it reduces that view's entry records and runs the
// view-dependent chain; colour / opacity / mean / cov3D gradients are summed in registers and the
// scale/rotation chain (linear in dL/dcov3D) runs once.  Replaces V kernels + the host-side sums over views.
__global__ __launch_bounds__(GSR_BLOCK) void preprocess_bwd_views_kernel(
    GsrBwdViews vw, int P, float mod, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drot, float* __restrict__ dL_dcov3D) {
  const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (i >= P) return;
  float gm3[3] = {0.f, 0.f, 0.f}, gcol[3] = {0.f, 0.f, 0.f}, gop = 0.f;
  float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float3 p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
  Cov3 cv;
  build_cov3(i, mod, scales, rotations, cov3D_precomp, cv);
  bool any = false;
  for (int v = 0; v < vw.V; ++v) {
    const GsrBwdView& w = vw.v[v];
    if (w.fused_alias) continue;      // its owner's records carry it (fused pair): the owner writes its dL_dmeans2D too
    float gm2[2] = {0.f, 0.f}, gm2a[2] = {0.f, 0.f};
    const bool pair = w.partner_dL_dmeans2D != nullptr;
    if (w.radii[i] > 0 && gsr_view_used(w, i)) {
      any = true;
      const PartialSum ps = reduce_partials(w.partials, min(w.offsets[i], w.cap), min(w.offsets[i + 1], w.cap),
                                            pair || w.dL_dcolors != nullptr || dL_dcolors != nullptr);
      gop += ps.gop;
      if (pair) {   // record layout of the pair backward: geometry sums of both views, then (sum t dx, sum t dy) of this view alone
        view_chain(w.view, w.proj, w.W, w.H, w.tanfovx, w.tanfovy, p, cv.c, ps, gcov, gm3, gm2, ps.dr, ps.dg, gm2a);
      } else {
        if (w.dL_dcolors) { w.dL_dcolors[3 * i] = ps.dr; w.dL_dcolors[3 * i + 1] = ps.dg; w.dL_dcolors[3 * i + 2] = ps.db; }
        else { gcol[0] += ps.dr; gcol[1] += ps.dg; gcol[2] += ps.db; }
        view_chain(w.view, w.proj, w.W, w.H, w.tanfovx, w.tanfovy, p, cv.c, ps, gcov, gm3, gm2);
      }
    } else if (w.dL_dcolors) {
      w.dL_dcolors[3 * i] = 0.f; w.dL_dcolors[3 * i + 1] = 0.f; w.dL_dcolors[3 * i + 2] = 0.f;
    }
    if (pair) {
      w.dL_dmeans2D[3 * i] = gm2a[0]; w.dL_dmeans2D[3 * i + 1] = gm2a[1]; w.dL_dmeans2D[3 * i + 2] = 0.f;
      float* m2b = w.partner_dL_dmeans2D;
      m2b[3 * i] = gm2[0] - gm2a[0]; m2b[3 * i + 1] = gm2[1] - gm2a[1]; m2b[3 * i + 2] = 0.f;
    } else {
      w.dL_dmeans2D[3 * i] = gm2[0]; w.dL_dmeans2D[3 * i + 1] = gm2[1]; w.dL_dmeans2D[3 * i + 2] = 0.f;
    }
  }
  if (any && !cov3D_precomp) cov3_to_scale_rot(cv, mod, gcov, gs, gq);
  if (vw.bwd_error && *vw.bwd_error != 0u) gm3[0] = gm3[1] = gm3[2] = __builtin_nanf("");   // the blend backward of this call aborted: loud, not garbage (GSR_QUEUE_BWD_ERROR)
  dL_dmeans3D[3 * i] = gm3[0]; dL_dmeans3D[3 * i + 1] = gm3[1]; dL_dmeans3D[3 * i + 2] = gm3[2];
  if (dL_dcolors) { dL_dcolors[3 * i] = gcol[0]; dL_dcolors[3 * i + 1] = gcol[1]; dL_dcolors[3 * i + 2] = gcol[2]; }
  if (vw.d_raw_rot) {   // raw-parameter mode: the chain through normalize / sigmoid / exp, here instead of in a launch of its own
    reinterpret_cast<float4*>(vw.d_raw_rot)[i] =
        gsr_act_rotation_bwd(reinterpret_cast<const float4*>(vw.raw_rot)[i], make_float4(gq[0], gq[1], gq[2], gq[3]));
    const float o = vw.act_op[i];
    vw.d_raw_op[i] = gop * o * (1.0f - o);
#pragma unroll
    for (int k = 0; k < 3; ++k) vw.d_raw_sc[3 * (size_t)i + k] = gs[k] * vw.act_sc[3 * (size_t)i + k];
  }
  if (dL_dopacity) dL_dopacity[i] = gop;
  if (dL_dscales) { dL_dscales[3 * i] = gs[0]; dL_dscales[3 * i + 1] = gs[1]; dL_dscales[3 * i + 2] = gs[2]; }
  if (dL_drot) { dL_drot[4 * i] = gq[0]; dL_drot[4 * i + 1] = gq[1]; dL_drot[4 * i + 2] = gq[2]; dL_drot[4 * i + 3] = gq[3]; }
  if (dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gcov[k];
  }
}

// ---- all views of a step at once, ONE WAVE PER VIEW (V >= 2) ------------------------------------------
// The loop above walks the views strictly load -> chain -> load at 1.5 waves per SIMD: 62 us for 8 views of 100 k Gaussians, 22 % of the
// HBM roofline.  Here a workgroup owns 64 Gaussians and has one wave per (non-alias) view: the view is wave-uniform (its matrices and
// pointers stay scalar loads), every wave reduces its view's records and runs that view's chain for the 64 Gaussians, parks its 13
// per-Gaussian sums in LDS ([view][value][lane]: conflict-free), and wave 0 adds the views up in view order -- the same order of
// additions as the loop, hence the same bits -- and finishes with the view-independent part (scale / rotation chain, activations).
// V x more waves in flight, no second pass over HBM.
#define PBW_VALUES 13      // gcov[6], gm3[3], gop, gcol[3]
// (No occupancy bound: with the chain in fp64 the kernel needs 166 VGPRs; bounded to the fp32 build's 80 it spilled and took 70 us at four views.)
template <int MAXW>      // waves per workgroup the instantiation is compiled for (= views it can take): its register budget follows
__global__ __launch_bounds__(64 * MAXW) void preprocess_bwd_views_waves_kernel(
    GsrBwdViews vw, int P, float mod, const float* __restrict__ means3D, const float* __restrict__ scales,
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, float* __restrict__ dL_dmeans3D,
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity, float* __restrict__ dL_dscales,
    float* __restrict__ dL_drot, float* __restrict__ dL_dcov3D) {
  extern __shared__ float s_part[];                  // [waves][PBW_VALUES + 1][64]  (+1: "this view saw the Gaussian")
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
  const int i = blockIdx.x * 64 + lane;
  const bool live = i < P;
  // wave wv's view: the wv-th view that is not a fused alias (uniform: scalar code)
  int v = -1;
  for (int u = 0, k = 0; u < vw.V; ++u)
    if (!vw.v[u].fused_alias) { if (k == wv) { v = u; break; } ++k; }
  const GsrBwdView& w = vw.v[v < 0 ? 0 : v];
  float3 p = make_float3(0.f, 0.f, 0.f);
  Cov3 cv;
  if (live) {
    p = make_float3(means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]);
    build_cov3(i, mod, scales, rotations, cov3D_precomp, cv);
  }
  float gm3[3] = {0.f, 0.f, 0.f}, gcol[3] = {0.f, 0.f, 0.f}, gop = 0.f, gcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float seen = 0.f;
  if (live && v >= 0) {
    float gm2[2] = {0.f, 0.f}, gm2a[2] = {0.f, 0.f};
    const bool pair = w.partner_dL_dmeans2D != nullptr;
    if (w.radii[i] > 0 && gsr_view_used(w, i)) {
      seen = 1.f;
      const PartialSum ps = reduce_partials(w.partials, min(w.offsets[i], w.cap), min(w.offsets[i + 1], w.cap),
                                            pair || w.dL_dcolors != nullptr || dL_dcolors != nullptr);
      gop = ps.gop;
      if (pair) {
        view_chain(w.view, w.proj, w.W, w.H, w.tanfovx, w.tanfovy, p, cv.c, ps, gcov, gm3, gm2, ps.dr, ps.dg, gm2a);
      } else {
        if (w.dL_dcolors) { w.dL_dcolors[3 * i] = ps.dr; w.dL_dcolors[3 * i + 1] = ps.dg; w.dL_dcolors[3 * i + 2] = ps.db; }
        else { gcol[0] = ps.dr; gcol[1] = ps.dg; gcol[2] = ps.db; }
        view_chain(w.view, w.proj, w.W, w.H, w.tanfovx, w.tanfovy, p, cv.c, ps, gcov, gm3, gm2);
      }
    } else if (w.dL_dcolors) {
      w.dL_dcolors[3 * i] = 0.f; w.dL_dcolors[3 * i + 1] = 0.f; w.dL_dcolors[3 * i + 2] = 0.f;
    }
    if (pair) {
      w.dL_dmeans2D[3 * i] = gm2a[0]; w.dL_dmeans2D[3 * i + 1] = gm2a[1]; w.dL_dmeans2D[3 * i + 2] = 0.f;
      float* m2b = w.partner_dL_dmeans2D;
      m2b[3 * i] = gm2[0] - gm2a[0]; m2b[3 * i + 1] = gm2[1] - gm2a[1]; m2b[3 * i + 2] = 0.f;
    } else {
      w.dL_dmeans2D[3 * i] = gm2[0]; w.dL_dmeans2D[3 * i + 1] = gm2[1]; w.dL_dmeans2D[3 * i + 2] = 0.f;
    }
  }
  float* mine = s_part + (size_t)wv * (PBW_VALUES + 1) * 64 + lane;
#pragma unroll
  for (int k = 0; k < 6; ++k) mine[k * 64] = gcov[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) { mine[(6 + k) * 64] = gm3[k]; mine[(10 + k) * 64] = gcol[k]; }
  mine[9 * 64] = gop;
  mine[PBW_VALUES * 64] = seen;
  __syncthreads();
  if (wv != 0 || !live) return;
  // view order: the loop kernel adds view 0's terms to zero-initialised sums first -- start from this wave's own values (view order
  // = wave order) and add the others in order
  bool any = seen != 0.f;
  for (int u = 1; u < nw; ++u) {
    const float* q = s_part + (size_t)u * (PBW_VALUES + 1) * 64 + lane;
#pragma unroll
    for (int k = 0; k < 6; ++k) gcov[k] += q[k * 64];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gm3[k] += q[(6 + k) * 64]; gcol[k] += q[(10 + k) * 64]; }
    gop += q[9 * 64];
    any = any || q[PBW_VALUES * 64] != 0.f;
  }
  float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
  if (any && !cov3D_precomp) cov3_to_scale_rot(cv, mod, gcov, gs, gq);
  if (vw.bwd_error && *vw.bwd_error != 0u) gm3[0] = gm3[1] = gm3[2] = __builtin_nanf("");   // the blend backward of this call aborted: loud, not garbage (GSR_QUEUE_BWD_ERROR)
  dL_dmeans3D[3 * i] = gm3[0]; dL_dmeans3D[3 * i + 1] = gm3[1]; dL_dmeans3D[3 * i + 2] = gm3[2];
  if (dL_dcolors) { dL_dcolors[3 * i] = gcol[0]; dL_dcolors[3 * i + 1] = gcol[1]; dL_dcolors[3 * i + 2] = gcol[2]; }
  if (vw.d_raw_rot) {
    reinterpret_cast<float4*>(vw.d_raw_rot)[i] =
        gsr_act_rotation_bwd(reinterpret_cast<const float4*>(vw.raw_rot)[i], make_float4(gq[0], gq[1], gq[2], gq[3]));
    const float o = vw.act_op[i];
    vw.d_raw_op[i] = gop * o * (1.0f - o);
#pragma unroll
    for (int k = 0; k < 3; ++k) vw.d_raw_sc[3 * (size_t)i + k] = gs[k] * vw.act_sc[3 * (size_t)i + k];
  }
  if (dL_dopacity) dL_dopacity[i] = gop;
  if (dL_dscales) { dL_dscales[3 * i] = gs[0]; dL_dscales[3 * i + 1] = gs[1]; dL_dscales[3 * i + 2] = gs[2]; }
  if (dL_drot) { dL_drot[4 * i] = gq[0]; dL_drot[4 * i + 1] = gq[1]; dL_drot[4 * i + 2] = gq[2]; dL_drot[4 * i + 3] = gq[3]; }
  if (dL_dcov3D) {
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * i + k] = gcov[k];
  }
}

}  // namespace gsr_preprocess_bwd
using namespace gsr_preprocess_bwd;

int gsr_launch_preprocess_bwd(const GsrCam& cam, int P, const float* means3D, const float* scales,
                              const float* rotations, const float* colors_precomp, const float* shs,
                              const float* cov3D_precomp, const int32_t* radii, const GeomState& g,
                              const float4* partials, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                              float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                              float* dL_dsh, const uint32_t* bwd_error, hipStream_t st) {
  if (P <= 0) return 0;
  const dim3 grid((P + GSR_BLOCK - 1) / GSR_BLOCK), block(GSR_BLOCK);
#define GSR_PBWD_ARGS                                                                                              \
  P, cam.W, cam.H, cam.tanfovx, cam.tanfovy, cam.scale_modifier, cam.sh_degree, cam.M, cam.view, cam.proj,       \
      cam.campos, means3D, scales, rotations, colors_precomp, shs, cov3D_precomp, radii, g.offsets, g.clamped,  \
      partials, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, dL_dsh, g.used, g.counters + 1, bwd_error
  if (shs) {
    if (!dL_dsh) { gsr_set_error("gsr_backward: shs given but dL_dsh is NULL"); return -2; }
    { GSR_PROF("preprocess_bwd", st);
  hipLaunchKernelGGL(preprocess_bwd_kernel<true>, grid, block, 0, st, GSR_PBWD_ARGS); }
  } else {
    { GSR_PROF("preprocess_bwd", st);
  hipLaunchKernelGGL(preprocess_bwd_kernel<false>, grid, block, 0, st, GSR_PBWD_ARGS); }
  }
#undef GSR_PBWD_ARGS
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_preprocess_bwd_views(const GsrBwdViews& vw, int P, float scale_modifier, const float* means3D,
                                    const float* scales, const float* rotations, const float* cov3D_precomp,
                                    float* dL_dmeans3D, float* dL_dcolors, float* dL_dopacity, float* dL_dscales,
                                    float* dL_drotations, float* dL_dcov3D, hipStream_t st) {
  if (P <= 0) return 0;
  int nact = 0;
  for (int v = 0; v < vw.V; ++v) nact += vw.v[v].fused_alias ? 0 : 1;
  if (nact >= 2) {      // one wave per view
    { GSR_PROF("preprocess_bwd_views", st);
#define PBW_LAUNCH(MAXW) hipLaunchKernelGGL(preprocess_bwd_views_waves_kernel<MAXW>, dim3((P + 63) / 64), dim3(64 * nact),              \
                           sizeof(float) * (size_t)nact * (PBW_VALUES + 1) * 64, st, vw, P, scale_modifier, means3D, scales, rotations,       \
                           cov3D_precomp, dL_dmeans3D, dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D)
      if (nact <= 4) PBW_LAUNCH(4); else if (nact <= 8) PBW_LAUNCH(8); else PBW_LAUNCH(GSR_MAX_BATCH);
#undef PBW_LAUNCH
    }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
  }
  { GSR_PROF("preprocess_bwd_views", st);
    hipLaunchKernelGGL(preprocess_bwd_views_kernel, dim3((P + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, vw, P,
                       scale_modifier, means3D, scales, rotations, cov3D_precomp, dL_dmeans3D, dL_dcolors, dL_dopacity,
                       dL_dscales, dL_drotations, dL_dcov3D); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
