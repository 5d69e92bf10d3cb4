// gsr_dynamics.hip -- the two data-parallel pieces of the rollout plumbing:
//   gsr_fps : farthest point sampling of a point cloud (role of dgl.geometry.farthest_point_sampler, which the
//             reference calls at /root/reference/src/render/dynamics_module.py:46,65 -- a third-party routine absent here);
//   gsr_lbs : moving every Gaussian with the bones -- inverse-distance weights, blended rigid transforms and blended
//             quaternions in ONE pass (the reference loops over bones in Python and materialises [P, n_bones, 3],
//             /root/reference/src/render/utils.py:207-239).
#include <stdlib.h>
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_dynamics::<kernel>, not "(anonymous namespace)"
namespace gsr_dynamics {

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

// ---------------------------------------------------------------- bones of a rollout step: sampling + thinning in one launch
// downsample_vertices of the rollout (/root/reference/src/render/dynamics_module.py:44-51): `npoints` farthest points of a SMALL cloud
// (N <= 1024: the 1000 tracked particles; one point and its running minimum per thread, in registers), then the radius thinning of
// /root/reference/src/data/utils.py:50-65 on the picked points (start at `thin_start`, keep adding the point farthest from the kept
// set until every point lies within `radius` of it).  Sampling: the arithmetic and the tie rule of fps_kernel.  Thinning: distances
// as torch.norm evaluates them on the host, sqrt(fma(dz, dz, fma(dy, dy, dx * dx))), first maximum on ties -- the host path's picks.
// One launch and ONE 4-byte read-back (the count) instead of a sampling launch, a device -> host copy of the points, ~100 numpy
// steps and a host -> device copy of the indices.
// Arg-max with "first maximum" over a workgroup of FT_THREADS threads whose thread t owns the CONSECUTIVE items 4 t .. 4 t + 3
// (so a lower lane holds lower indices).  (best, besti) = the thread's own candidate (best >= 0, or -1 with besti = INT_MAX when
// it has none).  Wave level: the maximum by DPP (values are non-negative: the zeros a masked DPP step reads are neutral), then the
// first lane holding it (ballot + ffs) -- no LDS crossbar trip; workgroup level: one LDS word pair per wave and ONE barrier (two
// buffers, alternated by the caller), every thread reduces the FT_THREADS / 64 candidates itself.
#define FT_THREADS 256
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float ft_dpp_max(float v) {
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ void block_argmax_first(float& best, int& besti, float* s_val, int* s_idx, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  float m = fmaxf(best, 0.0f);
  m = ft_dpp_max<0xB1>(m); m = ft_dpp_max<0x4E>(m); m = ft_dpp_max<0x141>(m); m = ft_dpp_max<0x140>(m);
  m = ft_dpp_max<0x142, 0xA>(m); m = ft_dpp_max<0x143, 0xC>(m);
  const float wmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
  const uint64_t who = __ballot(best == wmax && best >= 0.0f);
  float cv = -1.0f;
  int ci = 0x7fffffff;
  if (who) { cv = wmax; ci = __shfl(besti, __ffsll((long long)who) - 1, 64); }
  if (lane == 0) { s_val[wv] = cv; s_idx[wv] = ci; }
  __syncthreads();
  best = s_val[0]; besti = s_idx[0];
#pragma unroll
  for (int w = 1; w < FT_THREADS / 64; ++w) {
    const float v = s_val[w]; const int i = s_idx[w];
    if (v > best || (v == best && i < besti)) { best = v; besti = i; }
  }
}
__global__ __launch_bounds__(FT_THREADS) void fps_thin_small_kernel(const float* __restrict__ pos, int N, int npoints, int start, float radius,
                                                                  int thin_start, long long* __restrict__ out_idx,
                                                                  long long* __restrict__ thin_idx, int* __restrict__ thin_count) {
  __shared__ float s_val[2][FT_THREADS / 64];
  __shared__ int s_idx[2][FT_THREADS / 64];
  __shared__ float sp[3 * 1024];          // the picked points, in pick order
  __shared__ float sa[3 * 1024];          // the whole cloud (a pick's coordinates come from LDS, not from a global round trip per pick)
  const int tid = threadIdx.x;
  float px[4], py[4], pz[4], mind[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * tid + q;
    const bool have = i < N;
    px[q] = have ? pos[3 * i] : 0.f; py[q] = have ? pos[3 * i + 1] : 0.f; pz[q] = have ? pos[3 * i + 2] : 0.f;
    sa[3 * i] = px[q]; sa[3 * i + 1] = py[q]; sa[3 * i + 2] = pz[q];
    mind[q] = __builtin_inff();
  }
  __syncthreads();
  int cur = start;
  for (int k = 0; k < npoints; ++k) {
    const float cx = sa[3 * cur], cy = sa[3 * cur + 1], cz = sa[3 * cur + 2];
    if (tid == 0) { out_idx[k] = cur; sp[3 * k] = cx; sp[3 * k + 1] = cy; sp[3 * k + 2] = cz; }
    float best = -1.0f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * tid + q;
      if (i < N) {
        const float dx = px[q] - cx, dy = py[q] - cy, dz = pz[q] - cz;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        mind[q] = fminf(mind[q], d);
        if (mind[q] > best) { best = mind[q]; besti = i; }      // ascending i: the thread's first maximum
      }
    }
    block_argmax_first(best, besti, s_val[k & 1], s_idx[k & 1], tid);     // (two LDS buffers: one barrier per pick)
    cur = besti;
  }
  __syncthreads();                         // sp[] complete
  // ---- thinning over the npoints picks: thread t owns picks 4 t .. 4 t + 3
  float qx[4], qy[4], qz[4], dist[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 4 * tid + q;
    const bool mine = i < npoints;
    qx[q] = mine ? sp[3 * i] : 0.f; qy[q] = mine ? sp[3 * i + 1] : 0.f; qz[q] = mine ? sp[3 * i + 2] : 0.f;
    dist[q] = __builtin_inff();
  }
  int kept = 0, nxt = thin_start;
  for (int it = 0; it < npoints; ++it) {
    if (tid == 0) thin_idx[kept] = nxt;
    ++kept;
    const float nx = sp[3 * nxt], ny = sp[3 * nxt + 1], nz = sp[3 * nxt + 2];
    float best = -1.0f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * tid + q;
      if (i < npoints) {
        const float dx = qx[q] - nx, dy = qy[q] - ny, dz = qz[q] - nz;
        dist[q] = fminf(dist[q], __fsqrt_rn(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)))));
        if (dist[q] > best) { best = dist[q]; besti = i; }
      }
    }
    block_argmax_first(best, besti, s_val[it & 1], s_idx[it & 1], tid);
    if (!(best > radius)) break;           // uniform: every thread reduced the same candidates
    nxt = besti;
  }
  if (tid == 0) *thin_count = kept;
  for (int i = kept + tid; i < npoints; i += FT_THREADS) thin_idx[i] = thin_start;   // padding: a valid position (fixed-shape consumers gather with it)
}

// The same in ONE wave (N <= 1024, npoints <= 128: the rollout's 100-of-1000): lane l owns the CONSECUTIVE points 16 l .. 16 l + 15 in
// registers; a pick is the wave maximum (DPP), the first lane holding it (ballot + ffs) and that lane's first maximum (readlane) -- no
// barrier, no LDS exchange, one LDS read for the pick's coordinates.  Four waves with a barrier per pick cost 0.65 us per pick, the
// per-lane arithmetic was a quarter of it: 131 -> ~70 us per rollout step for the 100 picks + ~85 thinning rounds.  Same arithmetic,
// same tie rule (first maximum), bit-identical picks (test_bones_sampling_and_thinning_in_one_launch).
// (Round 5, measured and not kept: the pick loop on packed f32 -- v_pk_add / v_pk_mul, two points per issue -- with the minima compared as
// integers and the winner's slot found after the wave maximum: 208 -> 121 VALU + 22 -> 79 SALU instructions per pick, and 79.0 -> 82.4 us per
// call, same box, three alternating rounds: a lone wave's pick is bound by its dependent chain -- LDS read of the pick, the DPP maximum, the
// readlanes -- not by how many independent VALU operations sit between them.)
__global__ __launch_bounds__(64) void fps_thin_wave_kernel(const float* __restrict__ pos, int N, int npoints, int start, float radius,
                                                           int thin_start, long long* __restrict__ out_idx,
                                                           long long* __restrict__ thin_idx, int* __restrict__ thin_count) {
  __shared__ float sp[3 * 128];           // the picked points, in pick order
  __shared__ float sa[3 * 1024];          // the whole cloud (a pick's coordinates)
  const int lane = threadIdx.x;
  float px[16], py[16], pz[16], mind[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = 16 * lane + q;
    const bool have = i < N;
    px[q] = have ? pos[3 * i] : 0.f; py[q] = have ? pos[3 * i + 1] : 0.f; pz[q] = have ? pos[3 * i + 2] : 0.f;
    sa[3 * i] = px[q]; sa[3 * i + 1] = py[q]; sa[3 * i + 2] = pz[q];
    mind[q] = have ? __builtin_inff() : -1.0f;      // a slot beyond N never beats `best` (>= -1): the pick loop needs no bounds test
  }
  __syncthreads();
  int cur = start;
  for (int k = 0; k < npoints; ++k) {
    const float cx = sa[3 * cur], cy = sa[3 * cur + 1], cz = sa[3 * cur + 2];
    if (lane == 0) { out_idx[k] = cur; sp[3 * k] = cx; sp[3 * k + 1] = cy; sp[3 * k + 2] = cz; }
    float best = -1.0f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float dx = px[q] - cx, dy = py[q] - cy, dz = pz[q] - cz;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      mind[q] = fminf(mind[q], d);
      if (mind[q] > best) { best = mind[q]; besti = 16 * lane + q; }      // ascending index: the lane's first maximum
    }
    float m = fmaxf(best, 0.0f);
    m = ft_dpp_max<0xB1>(m); m = ft_dpp_max<0x4E>(m); m = ft_dpp_max<0x141>(m); m = ft_dpp_max<0x140>(m);
    m = ft_dpp_max<0x142, 0xA>(m); m = ft_dpp_max<0x143, 0xC>(m);
    const float wmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
    const uint64_t who = __ballot(best == wmax && best >= 0.0f);
    cur = who ? __builtin_amdgcn_readlane(besti, __ffsll((long long)who) - 1) : 0x7fffffff;
    if (!who) break;                        // (N == 0: nothing to pick; the launcher does not get here)
  }
  __syncthreads();                          // sp[] complete (one wave: orders the LDS writes of lane 0 before the reads below)
  float qx[2], qy[2], qz[2], dist[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = 2 * lane + q;
    const bool mine = i < npoints;
    qx[q] = mine ? sp[3 * i] : 0.f; qy[q] = mine ? sp[3 * i + 1] : 0.f; qz[q] = mine ? sp[3 * i + 2] : 0.f;
    dist[q] = mine ? __builtin_inff() : -1.0f;       // (as above)
  }
  int kept = 0, nxt = thin_start;
  for (int it = 0; it < npoints; ++it) {
    if (lane == 0) thin_idx[kept] = nxt;
    ++kept;
    const float nx = sp[3 * nxt], ny = sp[3 * nxt + 1], nz = sp[3 * nxt + 2];
    float best = -1.0f;
    int besti = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float dx = qx[q] - nx, dy = qy[q] - ny, dz = qz[q] - nz;
      dist[q] = fminf(dist[q], __fsqrt_rn(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)))));
      if (dist[q] > best) { best = dist[q]; besti = 2 * lane + q; }
    }
    float m = fmaxf(best, 0.0f);
    m = ft_dpp_max<0xB1>(m); m = ft_dpp_max<0x4E>(m); m = ft_dpp_max<0x141>(m); m = ft_dpp_max<0x140>(m);
    m = ft_dpp_max<0x142, 0xA>(m); m = ft_dpp_max<0x143, 0xC>(m);
    const float wmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
    const uint64_t who = __ballot(best == wmax && best >= 0.0f);
    if (!who || !(wmax > radius)) break;              // wave-uniform
    nxt = __builtin_amdgcn_readlane(besti, __ffsll((long long)who) - 1);
  }
  if (lane == 0) *thin_count = kept;
  for (int i = kept + lane; i < npoints; i += 64) thin_idx[i] = thin_start;   // padding: a valid position (fixed-shape consumers gather with it)
}

// ---------------------------------------------------------------- relations of a rollout step, fixed shapes
// construct_edges (gsdyn.dynamics; /root/reference/src/data/dataset.py:88-147) for the rollout's graph -- object particles 0 .. n_obj_cap - 1
// of which the first *n_valid are real, ONE tool particle at index n_obj_cap -- as one launch with padded outputs: receiver / sender
// lists of e_cap entries (row-major order of the adjacency matrix, as torch's nonzero gives them; unused entries = `dummy`) and the
// count.  A pair is related when both ends are real, not both tools, closer than thr (squared distance (dx*dx + dy*dy) + dz*dz < thr2,
// rounded as the torch expression) and -- among objects -- the sender is one of the receiver's topk nearest objects (itself included;
// equal distances: the lower index first).  One workgroup, one receiver per thread.
// Round 4: one WAVE per receiver (16 waves, receivers w, w + 16, ...), lane l holds the senders l and l + 64 (N <= 128).  The k nearest
// objects of a receiver are found as k successive wave minima of the 64-bit keys (distance bits << 32 | sender) above the previous one
// -- ties go to the lower index by construction -- so the k-th key is a threshold and membership one compare; the relations of a row
// are two ballots.  (Round 3's form -- one THREAD per receiver walking all senders through a 16-deep insertion list, three times --
// issued ~30 k instructions from two waves: 96 us of the 0.8 ms rollout step; this one 6 us.)
#define CE_THREADS 1024
#define CE_MAXK 16
// wave minimum of 64-bit keys, the same value in every lane: six DPP steps (both halves moved with the same control; lanes a step does
// not reach read the identity ~0) leave it in lane 63, two readlanes broadcast it -- the butterfly of 64-bit __shfl_xor it replaces was
// twelve ds_bpermute round trips per minimum, five minima per receiver.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned long long ce_dpp_min(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  return o < v ? o : v;
}
__device__ __forceinline__ unsigned long long ce_wave_min(unsigned long long v) {
  v = ce_dpp_min<0xB1>(v); v = ce_dpp_min<0x4E>(v); v = ce_dpp_min<0x141>(v); v = ce_dpp_min<0x140>(v);
  v = ce_dpp_min<0x142, 0xA>(v); v = ce_dpp_min<0x143, 0xC>(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__global__ __launch_bounds__(CE_THREADS) void construct_edges_kernel(const float* __restrict__ pos, int n_obj_cap, const int* __restrict__ n_valid_p,
                                                                   float thr2, int topk, long long dummy, int e_cap,
                                                                   long long* __restrict__ recv, long long* __restrict__ send,
                                                                   int* __restrict__ count, long long* __restrict__ rel, int rel_n,
                                                                   long long* __restrict__ row_start) {
  __shared__ float sp[3 * 128];
  __shared__ unsigned long long s_mask[128][2];      // row i: which senders it is related to
  __shared__ int s_base[128];
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = n_obj_cap + 1, n_valid = min(*n_valid_p, n_obj_cap);
  if (tid < N) { sp[3 * tid] = pos[3 * tid]; sp[3 * tid + 1] = pos[3 * tid + 1]; sp[3 * tid + 2] = pos[3 * tid + 2]; }
  __syncthreads();
  const int k = min(min(topk, CE_MAXK), n_valid);
  for (int i = wv; i < N; i += CE_THREADS / 64) {
    const bool i_tool = i == n_obj_cap, i_obj = i < n_valid;
    const float px = sp[3 * i], py = sp[3 * i + 1], pz = sp[3 * i + 2];
    unsigned long long key[2];
    bool near_[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lane + 64 * h;
      float d = __builtin_inff();
      if (j < N) {
        const float dx = px - sp[3 * j], dy = py - sp[3 * j + 1], dz = pz - sp[3 * j + 2];
        d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      }
      near_[h] = d < thr2;
      key[h] = j < n_valid ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j) : ~0ull;   // (d >= 0: its bits order like its value)
    }
    // the receiver's k-th nearest object as a key: k minima, each above the one before
    unsigned long long kth = 0ull;
    bool first = true;
    if (i_obj)
      for (int q = 0; q < k; ++q) {
        const unsigned long long c0 = (first || key[0] > kth) ? key[0] : ~0ull, c1 = (first || key[1] > kth) ? key[1] : ~0ull;
        kth = ce_wave_min(c0 < c1 ? c0 : c1);
        first = false;
      }
    unsigned long long m[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = lane + 64 * h;
      const bool j_tool = j == n_obj_cap, j_obj = j < n_valid;
      bool rel = (i_obj || i_tool) && (j_tool || j_obj) && !(i_tool && j_tool) && near_[h];
      if (i_obj && j_obj) rel = rel && key[h] <= kth;
      m[h] = __ballot(rel);
    }
    if (lane == 0) { s_mask[i][0] = m[0]; s_mask[i][1] = m[1]; }
  }
  __syncthreads();
  if (wv == 0) {        // exclusive scan of the row counts (N <= 128: two rows per lane)
    const int c0 = lane < N ? __popcll(s_mask[lane][0]) + __popcll(s_mask[lane][1]) : 0;
    const int c1 = lane + 64 < N ? __popcll(s_mask[lane + 64][0]) + __popcll(s_mask[lane + 64][1]) : 0;
    int inc0 = c0, inc1 = c1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o0 = __shfl_up(inc0, d, 64), o1 = __shfl_up(inc1, d, 64);
      if (lane >= d) { inc0 += o0; inc1 += o1; }
    }
    const int tot0 = __shfl(inc0, 63, 64);
    s_base[lane] = inc0 - c0;
    s_base[lane + 64] = tot0 + inc1 - c1;
    if (lane == 63) s_total = tot0 + inc1;
  }
  __syncthreads();
  const int total = s_total;
  for (int i = wv; i < N; i += CE_THREADS / 64) {       // row-major order of the adjacency matrix, as torch's nonzero gives it
    const unsigned long long m0 = s_mask[i][0], m1 = s_mask[i][1];
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int b = s_base[i];
    if ((m0 >> lane) & 1ull) { const int e = b + __popcll(m0 & lt); if (e < e_cap) { recv[e] = i; send[e] = lane; } }
    if ((m1 >> lane) & 1ull) { const int e = b + __popcll(m0) + __popcll(m1 & lt); if (e < e_cap) { recv[e] = i; send[e] = lane + 64; } }
  }
  if (tid == 0) *count = min(total, e_cap);
  for (int e = total + tid; e < e_cap; e += CE_THREADS) { recv[e] = dummy; send[e] = dummy; }
  // the same relations as a dense rel_n x rel_n 0 / 1 matrix (what gsr_fit_bones reads): one launch less than scattering the lists into zeros
  if (rel)
    for (int o = tid; o < rel_n * rel_n; o += CE_THREADS) {
      const int i = o / rel_n, j = o - i * rel_n;
      rel[o] = (i < N && j < N) ? (long long)((s_mask[i][j >> 6] >> (j & 63)) & 1ull) : 0ll;
    }
  // row_start[i] = the number of list entries whose receiver is below i, i = 0 .. rel_n (what torch.searchsorted(receivers, arange(rel_n + 1))
  // gives on the padded list: the padding's receiver is `dummy`) -- the segment bounds of gsr_gnn_aggregate, two launches less per step
  if (row_start)
    for (int i = tid; i <= rel_n; i += CE_THREADS)
      row_start[i] = i < N ? (long long)min(s_base[i], e_cap) : ((long long)i <= dummy ? (long long)min(total, e_cap) : (long long)e_cap);
}

// ---------------------------------------------------------------- farthest point sampling, several workgroups
// The single-workgroup kernel above streams the whole cloud from memory once per pick (20 us per pick at 100 k points).  Here every
// workgroup keeps a slice of the cloud AND its running minimum distances in LDS (2048 points = 32 KB), so a pick costs one pass
// over LDS plus ONE device-wide exchange without atomics: each workgroup publishes its best candidate as one 64-bit word (key =
// distance bits << 32 | ~index: the maximum key is the largest distance and, on ties, the smallest index -- the same "first
// maximum" as the single-workgroup kernel) with a write-through store into ITS slot of the pick's row, then wave 0 of every
// workgroup sweeps the row with L1-bypassing loads until all slots are filled (a key is never 0, so the data is its own flag)
// and reduces it.  One row PER PICK (zeroed by the launcher): nothing is ever reset or reused.  All workgroups must be resident
// together: at most 256 of them (one per CU), i.e. N <= 524 288; larger clouds take the single-workgroup kernel.
#define FPS_SLICE 2048
__global__ __launch_bounds__(GSR_BLOCK) void fps_multi_kernel(const float* __restrict__ pos, int N, int npoints, int start,
                                                             unsigned long long* __restrict__ cand, long long* __restrict__ out) {
  __shared__ float sx[FPS_SLICE], sy[FPS_SLICE], sz[FPS_SLICE], sm[FPS_SLICE];
  __shared__ unsigned long long s_key[GSR_BLOCK / GSR_WAVE];
  __shared__ int s_cur;
  __shared__ float s_c[3];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int base = (int)blockIdx.x * FPS_SLICE, n = min(FPS_SLICE, N - base);
  for (int i = tid; i < n; i += GSR_BLOCK) {
    sx[i] = pos[3 * (base + i)]; sy[i] = pos[3 * (base + i) + 1]; sz[i] = pos[3 * (base + i) + 2];
    sm[i] = __builtin_inff();
  }
  int cur = start;
  if (tid < 3) s_c[tid] = pos[3 * cur + tid];            // three lanes, one cache line
  __syncthreads();
  // Two barriers per pick (round 5; three before): the coordinates of the next pick are fetched by wave 0 as soon as it knows the winner
  // and published together with its index.
  for (int k = 0; k < npoints; ++k) {
    if (blockIdx.x == 0 && tid == 0) out[k] = cur;
    const float cx = s_c[0], cy = s_c[1], cz = s_c[2];   // (read by everybody before the barrier below; rewritten by wave 0 behind it)
    float best = -1.0f;
    int besti = 0x7fffffff;
    for (int i = tid; i < n; i += GSR_BLOCK) {
      const float dx = sx[i] - cx, dy = sy[i] - cy, dz = sz[i] - cz;
      const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      const float m = fminf(sm[i], d);
      sm[i] = m;
      if (m > best) { best = m; besti = base + i; }     // increasing i: keeps the first maximum of this thread
    }
    // distances are >= 0, so their bit patterns order like the values; ~index makes the smaller index the larger key
    unsigned long long key = best < 0.0f ? 0ull : (((unsigned long long)__float_as_uint(best) << 32) | (uint32_t)(~(uint32_t)besti));
    key = ~ce_wave_min(~key);        // wave maximum on DPP (six steps + two readlanes; the 64-bit shuffle butterfly was twelve ds_bpermute trips)
    if (lane == 0) s_key[wv] = key;
    __syncthreads();
    if (wv == 0) {
      unsigned long long* row = cand + (size_t)k * gridDim.x;
      if (lane == 0) {
        unsigned long long kk = s_key[0];
        for (int w = 1; w < GSR_BLOCK / GSR_WAVE; ++w) kk = s_key[w] > kk ? s_key[w] : kk;
        __hip_atomic_store(&row[blockIdx.x], kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // sweep the whole row at once -- up to 256 slots, four per lane, all four loads in flight together -- until every slot is filled:
      // one memory round trip per poll instead of one per 64 slots (round 5, with the two-barrier loop: 4.9 -> 4.2 ms for 1000 picks of 500 k points)
      unsigned long long v[4];
      bool missing;
      do {
        missing = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int slot = q * 64 + lane;
          v[q] = slot < (int)gridDim.x ? __hip_atomic_load(&row[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) missing |= v[q] == 0ull;
      } while (__ballot(missing) != 0ull);
      unsigned long long win = 0ull;
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q * 64 + lane < (int)gridDim.x) win = v[q] > win ? v[q] : win;
      win = ~ce_wave_min(~win);
      const int nxt = (int)(~(uint32_t)win);
      if (lane < 3) s_c[lane] = pos[3 * nxt + lane];
      if (lane == 0) s_cur = nxt;
    }
    __syncthreads();
    cur = s_cur;
  }
}

// ---------------------------------------------------------------- linear blend skinning of the Gaussians
#define LBS_CHUNK 256   // bones staged per LDS round
// Round 4: two Gaussians per thread (every broadcast LDS read of a bone serves both) and the inverse distance from v_rsq_f32 (<= 1 ulp;
// the reference's cdist + division are not bit-defined either) instead of a correctly rounded sqrt and a correctly rounded division,
// which were 20 of the ~45 instructions per (Gaussian, bone) pair: 65 -> 45 us for 500 k Gaussians x 85 bones.
#define LBS_PER_THREAD 2
__global__ __launch_bounds__(GSR_BLOCK) void lbs_kernel(int P, int nb, const float* __restrict__ bones,
                                                        const float* __restrict__ R, const float* __restrict__ t,
                                                        const float* __restrict__ bq, const float* xyz,
                                                        const float* quat, float* out_xyz,
                                                        float* out_quat, const int* __restrict__ nb_valid) {   // out_* may BE xyz / quat (a thread reads its Gaussians, then writes them)
  __shared__ float sB[LBS_CHUNK][3], sR[LBS_CHUNK][9], sT[LBS_CHUNK][3], sQ[LBS_CHUNK][4];
  if (nb_valid) nb = min(nb, *nb_valid);      // fixed-shape callers: only the first *nb_valid bones are real
  const int p0 = blockIdx.x * (GSR_BLOCK * LBS_PER_THREAD) + threadIdx.x;      // Gaussians p0 and p0 + GSR_BLOCK: coalesced either way
  float x[LBS_PER_THREAD], y[LBS_PER_THREAD], z[LBS_PER_THREAD];
  float ws[LBS_PER_THREAD], ax[LBS_PER_THREAD], ay[LBS_PER_THREAD], az[LBS_PER_THREAD], q0[LBS_PER_THREAD], q1[LBS_PER_THREAD], q2[LBS_PER_THREAD], q3[LBS_PER_THREAD];
#pragma unroll
  for (int u = 0; u < LBS_PER_THREAD; ++u) {
    const int p = p0 + u * GSR_BLOCK;
    const bool live = p < P;
    x[u] = live ? xyz[3 * p] : 0.f; y[u] = live ? xyz[3 * p + 1] : 0.f; z[u] = live ? xyz[3 * p + 2] : 0.f;
    ws[u] = ax[u] = ay[u] = az[u] = q0[u] = q1[u] = q2[u] = q3[u] = 0.f;
  }
  for (int b0 = 0; b0 < nb; b0 += LBS_CHUNK) {
    const int n = min(LBS_CHUNK, nb - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < n * 3; i += GSR_BLOCK) { (&sB[0][0])[i] = bones[3 * b0 + i]; (&sT[0][0])[i] = t[3 * b0 + i]; }
    for (int i = threadIdx.x; i < n * 9; i += GSR_BLOCK) (&sR[0][0])[i] = R[9 * b0 + i];
    for (int i = threadIdx.x; i < n * 4; i += GSR_BLOCK) (&sQ[0][0])[i] = bq[4 * b0 + i];
    __syncthreads();
    for (int b = 0; b < n; ++b) {   // every lane reads the same LDS address: broadcast, conflict-free
      const float bx = sB[b][0], by = sB[b][1], bz = sB[b][2];
      const float r0 = sR[b][0], r1 = sR[b][1], r2 = sR[b][2], r3 = sR[b][3], r4 = sR[b][4], r5 = sR[b][5], r6 = sR[b][6], r7 = sR[b][7], r8 = sR[b][8];
      const float tx = sT[b][0], ty = sT[b][1], tz = sT[b][2];
      const float c0 = sQ[b][0], c1 = sQ[b][1], c2 = sQ[b][2], c3 = sQ[b][3];
#pragma unroll
      for (int u = 0; u < LBS_PER_THREAD; ++u) {
        const float dx = x[u] - bx, dy = y[u] - by, dz = z[u] - bz;
        const float w = fminf(__builtin_amdgcn_rsqf(dx * dx + dy * dy + dz * dz), 1e4f);      // = 1 / max(distance, 1e-4)
        ws[u] += w;
        ax[u] += w * (r0 * dx + r1 * dy + r2 * dz + tx + bx);
        ay[u] += w * (r3 * dx + r4 * dy + r5 * dz + ty + by);
        az[u] += w * (r6 * dx + r7 * dy + r8 * dz + tz + bz);
        q0[u] += w * c0; q1[u] += w * c1; q2[u] += w * c2; q3[u] += w * c3;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < LBS_PER_THREAD; ++u) {
    const int p = p0 + u * GSR_BLOCK;
    if (p >= P) continue;
    const float inv = 1.0f / ws[u];
    out_xyz[3 * p] = ax[u] * inv; out_xyz[3 * p + 1] = ay[u] * inv; out_xyz[3 * p + 2] = az[u] * inv;
    if (quat && out_quat) {
      const float qn = fmaxf(sqrtf(q0[u] * q0[u] + q1[u] * q1[u] + q2[u] * q2[u] + q3[u] * q3[u]), 1e-12f);   // F.normalize's eps
      const float a0 = q0[u] / qn, a1 = q1[u] / qn, a2 = q2[u] / qn, a3 = q3[u] / qn;
      const float b0 = quat[4 * p], b1 = quat[4 * p + 1], b2 = quat[4 * p + 2], b3 = quat[4 * p + 3];
      out_quat[4 * p + 0] = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3;
      out_quat[4 * p + 1] = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2;
      out_quat[4 * p + 2] = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1;
      out_quat[4 * p + 3] = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0;
    }
  }
}

// ---------------------------------------------------------------- per-bone rotation fit (batched 3x3 problems)
// One thread per bone: singular value decomposition of the 3x3 moment matrix F by one-sided Jacobi rotations in fp64 (accurate
// small singular values, so the rank test of the reference -- S > S.max * 3 eps_fp32, /root/reference/src/render/utils.py:160-170
// -- takes the same decisions as LAPACK's), then the reference's decision tree (utils.py:147-205):
//   no related bone, or F = 0            -> identity                                                   code 0
//   rank 1                               -> the rotation taking the x axis onto U[:, 0], with LAPACK's sign of that vector (see below:
//                                           the reference's answer hangs on its SVD backend's convention)  code 3; a bone whose F has a
//                                           vanishing first column is left to the caller's LAPACK           code 1
//   rank 3 with det F < 0                -> identity (the reference indexes S[3, 3], fails, and falls back)   code 0
//   rank 2, or rank 3 with det F > 0     -> u1 v1^T + u2 v2^T + (u1 x u2)(v1 x v2)^T: for a full-rank F with positive determinant
//                                           this IS U V^T; for rank 2 it is the reference's U S V^T after its det = -1 repair
//                                           (both null vectors completed right-handed), independent of sign conventions   code 2
// One bone: Fb = its 3x3 moment matrix (row-major fp32), n = its neighbour count -> Rb (row-major fp32); returns the code.
__device__ __forceinline__ int fit_one_rotation(const float* Fb, float n, float* Rb) {
  double A[3][3], V[3][3];
  bool zero = true;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { A[i][j] = (double)Fb[3 * i + j]; V[i][j] = i == j ? 1.0 : 0.0; zero = zero && A[i][j] == 0.0; }
  auto identity = [&](int c) { for (int i = 0; i < 9; ++i) Rb[i] = (i % 4 == 0) ? 1.0f : 0.0f; return c; };
  if (n <= 0.0f || zero) return identity(0);
  for (int sweep = 0; sweep < 40; ++sweep) {            // A V0 = U S: rotate column pairs until they are orthogonal
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
        for (int i = 0; i < 3; ++i) { al += A[i][p] * A[i][p]; be += A[i][q] * A[i][q]; ga += A[i][p] * A[i][q]; }
        if (ga == 0.0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        off = fmax(off, fabs(ga) / sqrt(al * be));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sg[3];
  for (int j = 0; j < 3; ++j) sg[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  int o0 = 0, o1 = 1, o2 = 2;                          // descending order of the singular values
  if (sg[o0] < sg[o1]) { const int t = o0; o0 = o1; o1 = t; }
  if (sg[o1] < sg[o2]) { const int t = o1; o1 = o2; o2 = t; }
  if (sg[o0] < sg[o1]) { const int t = o0; o0 = o1; o1 = t; }
  const double thr = sg[o0] * 3.0 * 1.1920928955078125e-07;
  const int rank = (sg[o0] > thr) + (sg[o1] > thr) + (sg[o2] > thr);
  if (rank == 0) return identity(0);
  if (rank == 1) {
    // The reference turns the x axis onto U[:, 0] of its SVD (/root/reference/src/render/utils.py:166-181): the answer hangs on the SIGN the
    // backend gives that vector.  The host path uses LAPACK, whose bidiagonalisation starts with a Householder reflection of F's first
    // column c0 onto -sign(F00) |c0| e1 and never negates a column of U afterwards (negative singular values flip rows of V^T):
    // U[:, 0] = -sign(F00) c0 / |c0| up to rounding, i.e. the dominant left singular vector with a non-positive first component (or
    // along -c0 when F00 = 0).  Checked against torch.linalg.svd on 2 300 random rank-1 matrices (0 mismatches).  A vanishing first
    // column has no such rule: those bones stay flagged (code 1) for the host.
    const double c0[3] = {(double)Fb[0], (double)Fb[3], (double)Fb[6]};
    const double c0n = sqrt(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    if (!(c0n > 1e-6 * sg[o0])) return identity(1);
    double ax[3] = {A[0][o0] / sg[o0], A[1][o0] / sg[o0], A[2][o0] / sg[o0]};
    const double along = ax[0] * c0[0] + ax[1] * c0[1] + ax[2] * c0[2];              // u1 is +- c0 / |c0|
    const double want = Fb[0] < 0.0f ? 1.0 : -1.0;                                 // -sign(F00), sign(0) = +
    if (along * want < 0.0) { ax[0] = -ax[0]; ax[1] = -ax[1]; ax[2] = -ax[2]; }
    double pp[3] = {0.0, ax[2], -ax[1]};                                           // axis x (1, 0, 0)
    const double pn = sqrt(pp[1] * pp[1] + pp[2] * pp[2]);
    if (pn < 1e-6) return identity(3);
    pp[1] /= pn; pp[2] /= pn;
    const double t3[3] = {0.0, -pp[2], pp[1]};                                     // x cross perp
    const double a3[3] = {ax[1] * pp[2] - ax[2] * pp[1], ax[2] * pp[0] - ax[0] * pp[2], ax[0] * pp[1] - ax[1] * pp[0]};   // axis cross perp
    const double xx[3] = {1.0, 0.0, 0.0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rb[3 * i + j] = (float)(ax[i] * xx[j] + pp[i] * pp[j] + a3[i] * t3[j]);
    return 3;
  }
  double u1[3], u2[3], v1[3], v2[3];
  for (int i = 0; i < 3; ++i) { u1[i] = A[i][o0] / sg[o0]; u2[i] = A[i][o1] / sg[o1]; v1[i] = V[i][o0]; v2[i] = V[i][o1]; }
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
  if (rank == 3) {   // sign of det F = handedness of (u1, u2, u_o2) times handedness of (v1, v2, v_o2)
    double du = 0.0, dv = 0.0;
    for (int i = 0; i < 3; ++i) { du += u3[i] * A[i][o2]; dv += v3[i] * V[i][o2]; }
    if (du * dv < 0.0) return identity(0);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rb[3 * i + j] = (float)(u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j]);
  return 2;
}
__global__ __launch_bounds__(64) void fit_rotations_kernel(int nb, const float* __restrict__ F, const float* __restrict__ n_adj,
                                                           float* __restrict__ R, int* __restrict__ code) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nb) return;
  float Rb[9];
  code[b] = fit_one_rotation(F + 9 * b, n_adj[b], Rb);
  for (int i = 0; i < 9; ++i) R[9 * b + i] = Rb[i];
}

// Rotation matrix -> (w, x, y, z) with the comparisons and the arithmetic of /root/reference/src/render/utils.py:71-111 (branch on the
// trace, then on the largest diagonal element; the result of a branch is NOT a unit quaternion), then normalised as
// torch.nn.functional.normalize does (eps 1e-12).  Every operation is a separately rounded fp32 one, as in the torch expression the
// host path evaluates (gsdyn.dynamics.mat2quat): no contraction into FMAs.
__device__ __forceinline__ void mat2quat_unit(const float* m, float* q) {
  auto M = [&](int i, int j) { return m[3 * i + j]; };
  const float tr = fmaxf(__fadd_rn(__fadd_rn(M(0, 0), M(1, 1)), M(2, 2)), -1.0f);
  float w, x, y, z;
  if (tr > -1.0f) {
    const float s0 = sqrtf(__fadd_rn(tr, 1.0f)), h = __fdiv_rn(0.5f, s0);
    w = __fmul_rn(0.5f, s0); x = __fmul_rn(__fsub_rn(M(2, 1), M(1, 2)), h); y = __fmul_rn(__fsub_rn(M(0, 2), M(2, 0)), h); z = __fmul_rn(__fsub_rn(M(1, 0), M(0, 1)), h);
  } else if (M(0, 0) >= M(1, 1) && M(0, 0) >= M(2, 2)) {
    const float s1 = sqrtf(fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(1.0f, M(0, 0)), M(1, 1)), M(2, 2)), 1e-30f)), h = __fdiv_rn(0.5f, s1);
    w = __fmul_rn(__fsub_rn(M(2, 1), M(1, 2)), h); x = __fmul_rn(0.5f, h); y = __fmul_rn(__fadd_rn(M(1, 0), M(0, 1)), h); z = __fmul_rn(__fadd_rn(M(2, 0), M(0, 2)), h);
  } else if (M(1, 1) >= M(2, 2) && M(1, 1) > M(0, 0)) {
    const float s2 = sqrtf(fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(1.0f, M(1, 1)), M(0, 0)), M(2, 2)), 1e-30f)), h = __fdiv_rn(0.5f, s2);
    w = __fmul_rn(__fsub_rn(M(0, 2), M(2, 0)), h); x = __fmul_rn(__fadd_rn(M(2, 1), M(1, 2)), h); y = __fmul_rn(0.5f, h); z = __fmul_rn(__fadd_rn(M(0, 1), M(1, 0)), h);
  } else {
    const float s3 = sqrtf(fmaxf(__fsub_rn(__fsub_rn(__fadd_rn(1.0f, M(2, 2)), M(0, 0)), M(1, 1)), 1e-30f)), h = __fdiv_rn(0.5f, s3);
    w = __fmul_rn(__fsub_rn(M(1, 0), M(0, 1)), h); x = __fmul_rn(__fadd_rn(M(0, 2), M(2, 0)), h); y = __fmul_rn(__fadd_rn(M(1, 2), M(2, 1)), h); z = __fmul_rn(0.5f, h);
  }
  const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(w, w), __fmul_rn(x, x)), __fmul_rn(y, y)), __fmul_rn(z, z)));
  const float d = fmaxf(nrm, 1e-12f);
  q[0] = __fdiv_rn(w, d); q[1] = __fdiv_rn(x, d); q[2] = __fdiv_rn(y, d); q[3] = __fdiv_rn(z, d);
}

// The whole of fit_bone_rotations + mat2quat for the rollout step (gsdyn.dynamics.interpolate_motions) in ONE launch: per bone the
// moment matrix F_b = sum_j rel[b][j] != 0 (new_j - new_b)(old_j - old_b)^T (fp32, ascending j: deterministic, unlike the atomics of
// torch's index_add), the fit above, and the bone's unit quaternion.  ~45 torch launches and two host synchronisations less per
// rollout step (0.94 -> ~0.05 ms at 100 bones).  rel: [nb] rows of `rel_stride` int64 each.
__global__ __launch_bounds__(64) void fit_bones_kernel(int nb, const float* __restrict__ bones, const float* __restrict__ motions,
                                                       const long long* __restrict__ rel, long long rel_stride,
                                                       float* __restrict__ R, float* __restrict__ quat, int* __restrict__ code) {
  // Round 4: the wave first turns its 64 rows of the relation matrix into bit masks with COALESCED loads (lane = column, a ballot per
  // row half) and keeps the bones' positions in LDS; a thread then walks only its bone's set bits, in ascending order -- the same
  // sums in the same order as the row walk it replaces (100 dependent 8-byte loads per thread at a row stride: 32 us for 100 bones).
  __shared__ unsigned long long s_rel[64][2];
  __shared__ float s_bone[128 * 3], s_mot[128 * 3];
  const bool masks = nb <= 128;
  if (masks) {
    const int lane = threadIdx.x, b0 = blockIdx.x * 64;
    for (int i = lane; i < nb * 3; i += 64) { s_bone[i] = bones[i]; s_mot[i] = motions[i]; }
    const int c0 = min(lane, nb - 1), c1 = min(lane + 64, nb - 1);       // (clamped: every load is in range, the ballots mask the rest)
    for (int r0 = 0; r0 < 64; r0 += 8) {        // eight rows' loads in flight, then their ballots
      long long v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const size_t row = (size_t)min(b0 + r0 + u, nb - 1) * rel_stride;
        v0[u] = rel[row + c0]; v1[u] = rel[row + c1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool in = b0 + r0 + u < nb;
        const unsigned long long m0 = __ballot(in && lane < nb && v0[u] != 0), m1 = __ballot(in && lane + 64 < nb && v1[u] != 0);
        if (lane == 0) { s_rel[r0 + u][0] = m0; s_rel[r0 + u][1] = m1; }
      }
    }
    __syncthreads();
  }
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= nb) return;
  const float bx = bones[3 * b], by = bones[3 * b + 1], bz = bones[3 * b + 2];
  const float nx = __fadd_rn(bx, motions[3 * b]), ny = __fadd_rn(by, motions[3 * b + 1]), nz = __fadd_rn(bz, motions[3 * b + 2]);
  float F[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int n = 0;
  if (masks) {
    for (int h = 0; h < 2; ++h) {
      unsigned long long m = s_rel[threadIdx.x][h];
      while (m) {
        const int j = 64 * h + __ffsll((long long)m) - 1;
        m &= m - 1ull;
        ++n;
        const float jx = s_bone[3 * j], jy = s_bone[3 * j + 1], jz = s_bone[3 * j + 2];
        const float ox = __fsub_rn(jx, bx), oy = __fsub_rn(jy, by), oz = __fsub_rn(jz, bz);
        const float wx = __fsub_rn(__fadd_rn(jx, s_mot[3 * j]), nx), wy = __fsub_rn(__fadd_rn(jy, s_mot[3 * j + 1]), ny),
                    wz = __fsub_rn(__fadd_rn(jz, s_mot[3 * j + 2]), nz);
        F[0] = __fadd_rn(F[0], __fmul_rn(wx, ox)); F[1] = __fadd_rn(F[1], __fmul_rn(wx, oy)); F[2] = __fadd_rn(F[2], __fmul_rn(wx, oz));
        F[3] = __fadd_rn(F[3], __fmul_rn(wy, ox)); F[4] = __fadd_rn(F[4], __fmul_rn(wy, oy)); F[5] = __fadd_rn(F[5], __fmul_rn(wy, oz));
        F[6] = __fadd_rn(F[6], __fmul_rn(wz, ox)); F[7] = __fadd_rn(F[7], __fmul_rn(wz, oy)); F[8] = __fadd_rn(F[8], __fmul_rn(wz, oz));
      }
    }
  } else
  for (int j = 0; j < nb; ++j) {
    if (rel[(size_t)b * rel_stride + j] == 0) continue;
    ++n;
    const float ox = __fsub_rn(bones[3 * j], bx), oy = __fsub_rn(bones[3 * j + 1], by), oz = __fsub_rn(bones[3 * j + 2], bz);
    const float wx = __fsub_rn(__fadd_rn(bones[3 * j], motions[3 * j]), nx), wy = __fsub_rn(__fadd_rn(bones[3 * j + 1], motions[3 * j + 1]), ny),
                wz = __fsub_rn(__fadd_rn(bones[3 * j + 2], motions[3 * j + 2]), nz);
    F[0] = __fadd_rn(F[0], __fmul_rn(wx, ox)); F[1] = __fadd_rn(F[1], __fmul_rn(wx, oy)); F[2] = __fadd_rn(F[2], __fmul_rn(wx, oz));
    F[3] = __fadd_rn(F[3], __fmul_rn(wy, ox)); F[4] = __fadd_rn(F[4], __fmul_rn(wy, oy)); F[5] = __fadd_rn(F[5], __fmul_rn(wy, oz));
    F[6] = __fadd_rn(F[6], __fmul_rn(wz, ox)); F[7] = __fadd_rn(F[7], __fmul_rn(wz, oy)); F[8] = __fadd_rn(F[8], __fmul_rn(wz, oz));
  }
  float Rb[9], q[4];
  code[b] = fit_one_rotation(F, (float)n, Rb);
  mat2quat_unit(Rb, q);
  for (int i = 0; i < 9; ++i) R[9 * b + i] = Rb[i];
  for (int i = 0; i < 4; ++i) quat[4 * b + i] = q[i];
}

}  // namespace gsr_dynamics
using namespace gsr_dynamics;

static int fps_workgroups(int N) { return (N + FPS_SLICE - 1) / FPS_SLICE; }
size_t gsr_fps_scratch_size(int N, int npoints) {   // N floats of running minima (single-workgroup path) + one row of candidate words per pick
  const int wgs = fps_workgroups(N);
  const size_t rows = (wgs > 1 && wgs <= 256) ? (size_t)(npoints > 0 ? npoints : 1) * wgs * 8 : 0;
  return gsr_align((size_t)(N > 0 ? N : 1) * 4) + gsr_align(rows);
}

int gsr_launch_fps(int N, const float* pos, int npoints, int start, float* mind, long long* out, hipStream_t st) {
  if (N <= 0 || npoints <= 0) return 0;
  static const bool single = [] { const char* e = getenv("GSR_FPS_SINGLE_WG"); return e && *e && atoi(e) != 0; }();
  const int wgs = fps_workgroups(N);
  // fps_multi_kernel's workgroups wait for each other's candidates: every one of them must be RESIDENT, or the launch never ends.
  // Bound the grid by what the device can hold at once (occupancy of this kernel x compute units, queried once per process);
  // anything larger -- or a device shared with other streams, GSR_FPS_SINGLE_WG=1 -- takes the single-workgroup kernel.
  static const int resident = [] {
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_multi_kernel, GSR_BLOCK, 0) != hipSuccess) return 0;
    return (per_cu > 0 ? 1 : 0) * prop.multiProcessorCount;   // ONE workgroup per CU counted: others may hold the rest
  }();
  if (!single && wgs > 1 && wgs <= 256 && wgs <= resident) {
    unsigned long long* cand = (unsigned long long*)((char*)mind + gsr_align((size_t)N * 4));
    GSR_HIP_CHECK(hipMemsetAsync(cand, 0, (size_t)npoints * wgs * 8, st));
    { GSR_PROF("fps", st);
      hipLaunchKernelGGL(fps_multi_kernel, dim3(wgs), dim3(GSR_BLOCK), 0, st, pos, N, npoints, start, cand, out); }
    GSR_HIP_CHECK(hipGetLastError());
    return 0;
  }
  { GSR_PROF("fps", st);
    hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(1024), 0, st, pos, N, npoints, start, mind, out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_fps_thin(int N, const float* pos, int npoints, int start, float radius, int thin_start, long long* out_idx, long long* thin_idx,
                        int* thin_count, hipStream_t st) {
  { GSR_PROF("fps_thin", st);
    if (N <= 1024 && npoints <= 128)       // one wave, 16 points per lane in registers (round 4); larger inputs: the four-wave form
      hipLaunchKernelGGL(fps_thin_wave_kernel, dim3(1), dim3(64), 0, st, pos, N, npoints, start, radius, thin_start, out_idx, thin_idx, thin_count);
    else
      hipLaunchKernelGGL(fps_thin_small_kernel, dim3(1), dim3(FT_THREADS), 0, st, pos, N, npoints, start, radius, thin_start, out_idx, thin_idx, thin_count); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_lbs(int P, int nb, const float* bones, const float* R, const float* t, const float* bq, const float* xyz,
                   const float* quat, float* out_xyz, float* out_quat, hipStream_t st, const int* nb_valid) {
  if (P <= 0) return 0;
  { GSR_PROF("lbs", st);
    hipLaunchKernelGGL(lbs_kernel, dim3((P + GSR_BLOCK * LBS_PER_THREAD - 1) / (GSR_BLOCK * LBS_PER_THREAD)), dim3(GSR_BLOCK), 0, st, P, nb, bones, R, t, bq, xyz, quat,
                       out_xyz, out_quat, nb_valid); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- the torch glue of a graphed rollout step, as two launches
// A step replayed from a hipGraph is a CHAIN of ~55 nodes and lasts about 4.5 us per node whatever the node does (round 5: 433 us per step
// of which the kernels with real work -- sampling, skinning, relations, rotation fit -- are 160): what counts is the node count.
// rollout_head_kernel: everything between the bone sampling and the network that torch spelled as gathers, transposes and concatenations
// (/root/reference/src/render/dynamics_module.py:104-131 builds the same tensors with torch.cat per step) -- one thread per padded row r:
//   state row  = the n_his positions of bone r (hist[:, idx1[thin[r]]]), of the tool (row nb: eef_hist), zeros behind;
//   action row = eef_next - eef_hist[-1] for the tool, zeros elsewhere;
//   p_in  [n_cap, A + (S ? 3 n_his : 0) + 3] = (attributes, [state], action)      -- the particle encoder's input
//   nodes [n_cap, A + 1 + 3 n_his]           = (attributes, instance, state)      -- what gsr_gnn_rel_inputs gathers from
//   bones_last [nb, 3], states_last [nb + 1, 3] = the last frame's positions (for gsr_fit_bones / gsr_construct_edges).
// Nine launches (2 gathers, 5 concatenations, a transpose copy, a subtraction) become one.
__global__ __launch_bounds__(GSR_BLOCK) void rollout_head_kernel(int n_track, int n_his, int nb, int n_cap, int A, int with_state,
                                                                const float* __restrict__ hist, const long long* __restrict__ idx1,
                                                                const long long* __restrict__ thin, const float* __restrict__ eef_hist,
                                                                const float* __restrict__ eef_next, const float* __restrict__ attrs,
                                                                const float* __restrict__ inst, float* __restrict__ bones_last,
                                                                float* __restrict__ states_last, float* __restrict__ state_t,
                                                                float* __restrict__ act, float* __restrict__ p_in, float* __restrict__ nodes) {
  const int r = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (r >= n_cap) return;
  const int S3 = 3 * n_his, Dp = A + (with_state ? S3 : 0) + 3, Dn = A + 1 + S3;
  const long long p = r < nb ? idx1[thin[r]] : 0;
  float a3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) a3[c] = r == nb ? eef_next[c] - eef_hist[3 * (n_his - 1) + c] : 0.f;
  for (int k = 0; k < A; ++k) { const float v = attrs[(size_t)r * A + k]; p_in[(size_t)r * Dp + k] = v; nodes[(size_t)r * Dn + k] = v; }
  nodes[(size_t)r * Dn + A] = inst[r];
  for (int h = 0; h < n_his; ++h)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = r < nb ? hist[((size_t)h * n_track + p) * 3 + c] : (r == nb ? eef_hist[3 * h + c] : 0.f);
      state_t[(size_t)r * S3 + 3 * h + c] = v;
      nodes[(size_t)r * Dn + A + 1 + 3 * h + c] = v;
      if (with_state) p_in[(size_t)r * Dp + A + 3 * h + c] = v;
      if (h == n_his - 1) {
        if (r < nb) bones_last[3 * r + c] = v;
        if (r <= nb) states_last[3 * r + c] = v;
      }
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) { act[3 * r + c] = a3[c]; p_in[(size_t)r * Dp + Dp - 3 + c] = a3[c]; }
}
// rollout_motion_kernel: behind the network's last product -- predicted position = last position + clamp(predicted motion) (model.py:240-244),
// the bones' motion = predicted - last (dynamics_module.py:133), written straight into the step's skinning packet (gsdyn.dynamics.pack_skin:
// head, bones, [rotations], motions, [quaternions], predicted; gsr_fit_bones writes the two bracketed blocks in place): a clamp, an add, a
// subtraction, a conversion and a concatenation become one launch.  The arithmetic and its order are torch's (NaN stays NaN).
__global__ __launch_bounds__(GSR_BLOCK) void rollout_motion_kernel(int nb, int n_his, float clampv, const float* __restrict__ state_t,
                                                                  const float* __restrict__ pred_motion, const int* __restrict__ cnt,
                                                                  float* __restrict__ packet) {
  const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (i == 0) { packet[0] = (float)*cnt; packet[1] = 1.0f; }
  if (i >= 3 * nb) return;
  const int r = i / 3, c = i - 3 * r;
  const float s = state_t[(size_t)r * 3 * n_his + 3 * (n_his - 1) + c], m = pred_motion[3 * r + c];
  const float cl = m < -clampv ? -clampv : (m > clampv ? clampv : m);
  const float pr = s + cl;
  packet[2 + i] = s;
  packet[2 + 12 * nb + i] = pr - s;
  packet[2 + 19 * nb + i] = pr;
}
int gsr_launch_rollout_head(int n_track, int n_his, int nb, int n_cap, int A, int with_state, const float* hist, const long long* idx1,
                            const long long* thin, const float* eef_hist, const float* eef_next, const float* attrs, const float* inst,
                            float* bones_last, float* states_last, float* state_t, float* act, float* p_in, float* nodes, hipStream_t st) {
  { GSR_PROF("rollout_head", st);
    hipLaunchKernelGGL(rollout_head_kernel, dim3((n_cap + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, n_track, n_his, nb, n_cap, A, with_state, hist, idx1,
                       thin, eef_hist, eef_next, attrs, inst, bones_last, states_last, state_t, act, p_in, nodes); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
int gsr_launch_rollout_motion(int nb, int n_his, float clampv, const float* state_t, const float* pred_motion, const int* cnt, float* packet,
                              hipStream_t st) {
  { GSR_PROF("rollout_motion", st);
    hipLaunchKernelGGL(rollout_motion_kernel, dim3((3 * nb + GSR_BLOCK - 1) / GSR_BLOCK > 0 ? (3 * nb + GSR_BLOCK - 1) / GSR_BLOCK : 1), dim3(GSR_BLOCK), 0, st, nb,
                       n_his, clampv, state_t, pred_motion, cnt, packet); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- bookkeeping at the end of a graphed rollout step
// After the skinning: the tracked particles' new positions gathered out of the cloud, the history windows shifted by one frame
// (dynamics_module.py:150-165 of the reference does this with torch.cat per step), the predicted bones masked to the valid ones, the
// count of bones the device could not resolve -- one launch instead of ~16 tiny ones.
__global__ __launch_bounds__(GSR_BLOCK) void rollout_tail_kernel(int n_track, int n_his, int nb, const float* __restrict__ all_pos,
                                                                const long long* __restrict__ track, float* __restrict__ pos_track,
                                                                float* __restrict__ hist, float* __restrict__ eef_hist,
                                                                const float* __restrict__ eef_next, const float* __restrict__ pred_in,
                                                                const int* __restrict__ cnt, const int* __restrict__ code,
                                                                float* __restrict__ pred_out, int* __restrict__ n_valid_out,
                                                                long long* __restrict__ bad) {
  const int t = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (t < n_track) {
    const long long g = track[t];
    const float x = all_pos[3 * g], y = all_pos[3 * g + 1], z = all_pos[3 * g + 2];
    pos_track[3 * t] = x; pos_track[3 * t + 1] = y; pos_track[3 * t + 2] = z;
    for (int h = 0; h + 1 < n_his; ++h)
      for (int c = 0; c < 3; ++c) hist[((size_t)h * n_track + t) * 3 + c] = hist[((size_t)(h + 1) * n_track + t) * 3 + c];
    float* last = hist + ((size_t)(n_his - 1) * n_track + t) * 3;
    last[0] = x; last[1] = y; last[2] = z;
  }
  if (blockIdx.x == 0) {
    const int n = min(*cnt, nb);
    if (threadIdx.x < 3) {
      for (int h = 0; h + 1 < n_his; ++h) eef_hist[3 * h + threadIdx.x] = eef_hist[3 * (h + 1) + threadIdx.x];
      eef_hist[3 * (n_his - 1) + threadIdx.x] = eef_next[threadIdx.x];
    }
    int wrong = 0;
    for (int b = threadIdx.x; b < nb; b += GSR_BLOCK) {
      const bool valid = b < n;
      for (int c = 0; c < 3; ++c) pred_out[3 * b + c] = valid ? pred_in[3 * b + c] : 0.f;     // the reference leaves the unused bone rows at zero
      wrong += (valid && code[b] == 1) ? 1 : 0;
    }
    __shared__ int s_wrong;
    if (threadIdx.x == 0) { s_wrong = 0; *n_valid_out = *cnt; }
    __syncthreads();
    if (wrong) atomicAdd(&s_wrong, wrong);
    __syncthreads();
    if (threadIdx.x == 0 && s_wrong) *bad += s_wrong;
  }
}
int gsr_launch_rollout_tail(int n_track, int n_his, int nb, const float* all_pos, const long long* track, float* pos_track, float* hist,
                            float* eef_hist, const float* eef_next, const float* pred_in, const int* cnt, const int* code, float* pred_out,
                            int* n_valid_out, long long* bad, hipStream_t st) {
  { GSR_PROF("rollout_tail", st);
    hipLaunchKernelGGL(rollout_tail_kernel, dim3((n_track + GSR_BLOCK - 1) / GSR_BLOCK > 0 ? (n_track + GSR_BLOCK - 1) / GSR_BLOCK : 1), dim3(GSR_BLOCK), 0, st,
                       n_track, n_his, nb, all_pos, track, pos_track, hist, eef_hist, eef_next, pred_in, cnt, code, pred_out, n_valid_out, bad); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_construct_edges(const float* pos, int n_obj_cap, const int* n_valid, float thr2, int topk, long long dummy, int e_cap,
                               long long* recv, long long* send, int* count, long long* rel, int rel_n, hipStream_t st, long long* row_start) {
  { GSR_PROF("construct_edges", st);
    hipLaunchKernelGGL(construct_edges_kernel, dim3(1), dim3(CE_THREADS), 0, st, pos, n_obj_cap, n_valid, thr2, topk, dummy, e_cap, recv, send, count, rel, rel_n,
                       row_start); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_fit_bones(int nb, const float* bones, const float* motions, const long long* rel, long long rel_stride, float* R, float* quat,
                         int* code, hipStream_t st) {
  if (nb <= 0) return 0;
  { GSR_PROF("fit_bones", st);
    hipLaunchKernelGGL(fit_bones_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, nb, bones, motions, rel, rel_stride, R, quat, code); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_fit_rotations(int nb, const float* F, const float* n_adj, float* R, int* code, hipStream_t st) {
  if (nb <= 0) return 0;
  { GSR_PROF("fit_rotations", st);
    hipLaunchKernelGGL(fit_rotations_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, nb, F, n_adj, R, code); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
