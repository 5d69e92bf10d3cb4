// gsr_gnn.hip -- the propagation network of the particle dynamics (DynamicsPredictor: /root/reference/src/gnn/model.py:70-246, run per rollout
// step by /root/reference/src/gnn/dynamics_module.py:53-172) for ONE graph in ONE launch.
//
// Why: a rollout step at BASELINE configs[4] evaluates the network on ~100 nodes and ~600 relations at width 512.  Through the GEMM
// library that is 17 matrix products of 5 - 16 us each plus ~40 gathers / concatenations / index_adds / activations of ~4.7 us each
// (profiles/r04_rollout_trace.txt): 290 of the step's 820 us, all of it launch-to-launch latency on tiny dependent kernels.  Here a
// persistent grid of workgroups walks the layers with a device-wide barrier between them; the matrix products run on the f32 MFMA
// (v_mfma_f32_16x16x4_f32: exact f32, bit-for-bit an fmaf chain in k order), operands straight from L2 (the activations of a layer
// are at most 1.5 MB, a weight matrix 1 MB), epilogues (bias, ReLU, residual) in the accumulator registers.
//
// Algebra (exact in real arithmetic, a different summation order in f32 -- within the 1e-6 the GEMM library's own blocking moves):
// the relation propagator's input is cat(relation_encode, effect[receiver], effect[sender]) @ W^T with W = [W1 | W2 | W3]:
//   relation_encode @ W1^T + b          does not change over the propagation steps: computed once            (E x H x H)
//   (effect @ W2^T)[receiver], (effect @ W3^T)[sender]      products on the N nodes, gathered per relation      (2 x N x H x H per step)
// instead of an E x 3H x H product per step (3 x 1.5 GFLOP -> 0.17 + 6 x 0.07); likewise the particle propagator's
// cat(particle_encode, agg) @ Wp^T = particle_encode @ Wp1^T + b (once) + agg @ Wp2^T.  The relations arrive sorted by receiver
// (row-major order of the adjacency matrix), so the aggregation is a segmented sum in list order: deterministic, no atomics.
#include "gsr_common.h"

namespace gsr_gnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef GNN_THREADS
#define GNN_THREADS 512
#endif
#define GNN_WAVES (GNN_THREADS / 64)

typedef GsrGnnArgs GnnArgs;

// Device-wide barrier of a persistent grid (every workgroup resident: the launcher starts at most one per CU).
// GNN_COHERENT = 1: everything a layer hands to the next one is STORED write-through and LOADED past the non-coherent cache levels (agent-
// scope relaxed atomics: the sc1 forms of global_store / global_load), so the barrier needs no cache maintenance -- only that a
// workgroup's stores have been acknowledged before it arrives.  GNN_COHERENT = 0: plain accesses and agent-scope release / acquire
// fences (buffer_wbl2 / buffer_inv) at every barrier: measured 50 us per barrier (the L2 write-back), 760 us for the 15 of a call.
#ifndef GNN_COHERENT
#define GNN_COHERENT 1
#endif
// GNN_CACHED_LOADS = 1 (with GNN_COHERENT): the stores stay write-through (no dirty lines: nothing to write back at a barrier), but the
// loads are plain, cacheable ones behind an acquire fence (buffer_inv) after every barrier: a layer's X rows are read by 32 - 56 tiles,
// and as sc1 loads every one of those reads went to memory (2.3 TB/s across the chip: 56 us per relation-side layer).
#ifndef GNN_CACHED_LOADS
#define GNN_CACHED_LOADS 0      // measured: the invalidate costs 7 us per barrier and drops the cached WEIGHTS too: 438 us per call against 379
#endif
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
#if GNN_COHERENT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
  if (threadIdx.x == 0) {
#if !GNN_COHERENT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded: a grid that is not resident together -- it always is: at most one workgroup per CU -- or words left over from an aborted
    //  launch must not hang the device; ~0.5 s, then the error word is set and the results are garbage)
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 23)) { __hip_atomic_store(ctr + 2, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
#if !GNN_COHERENT || GNN_CACHED_LOADS
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // buffer_inv: drop what this CU's L1 and this XCD's L2 hold of the buffers the last stage rewrote
#endif
}
// accessors of the inter-layer buffers
__device__ __forceinline__ void st_x(float* p, float v) {
#if GNN_COHERENT
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *p = v;
#endif
}
__device__ __forceinline__ float ld_x(const float* p) {
#if GNN_COHERENT && !GNN_CACHED_LOADS
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}
__device__ __forceinline__ float4 ld_x4(const float* p) {      // 16-byte aligned
#if GNN_COHERENT && !GNN_CACHED_LOADS
  const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ int ld_i(const int* p) {
#if GNN_COHERENT && !GNN_CACHED_LOADS
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}

// A layer = up to two products that do not depend on each other (the particle side and the relation side of an encoder stage, ...):
// their 16 x 16 output tiles form one list.  A tile is computed by FOUR waves, a quarter of K each (split-K): the operands of a layer
// were written by other CUs a barrier ago, so every load is a trip to memory (~2 us), and one wave walking K = 512 in batches of what
// its registers hold pays that trip four times in a row -- measured 14 - 21 us per tile against 1.7 us of MFMA issue.  With a quarter
// per wave all of a tile's loads are in flight at once, and the loads of the workgroup's NEXT tile are issued before the MFMAs of this
// one (two register sets; sched_barrier: left alone, the scheduler sinks every load to its use to save registers).  The four partial
// tiles meet in LDS and are summed in a fixed order ((q0 + q1) + (q2 + q3)) by the first wave of the four, which runs the epilogue.
// Lane (r = lane & 15, q = lane >> 4) loads X[i0 + r][k0 + 4 q .. + 3] and W[j0 + r][k0 + 4 q .. + 3] per chunk of 16 k: MFMA t of the
// chunk contracts k = k0 + 4 q + t over q; two accumulators (even / odd t) keep the 40-cycle dependent latency of the 16x16x4 form off
// its 32-cycle issue.
enum { EPI_BIAS_RELU = 0, EPI_BIAS = 1, EPI_NONE = 2, EPI_ADD2_RELU = 3 };
struct Prod { const float* X; int ldx; const float* W; int ldw; int K; float* out; int ldo; const float* bias; const float* add0; const float* add1; int ldadd; int M; int Nc; };
#define GNN_KSPLIT 4
#define GNN_UNITS (GNN_WAVES / GNN_KSPLIT)        // tiles a workgroup has in flight
struct GnnLds { float4 part[2][GNN_UNITS][GNN_KSPLIT][64]; };

struct TileRef { const float* xp; const float* wp; int nch; int i0, j0; const Prod* p; };    // nch: chunks of 16 k this wave contracts (0: no tile)

template <int EPI>
__device__ __attribute__((noinline)) void layer(const Prod& p0, const Prod& p1, GnnLds& L) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int unit = wv / GNN_KSPLIT, kq = wv % GNN_KSPLIT;
  const int t0 = (p0.M >> 4) * (p0.Nc >> 4), t1 = p1.X ? (p1.M >> 4) * (p1.Nc >> 4) : 0, total = t0 + t1;
  const int per_round = gridDim.x * GNN_UNITS, rounds = (total + per_round - 1) / per_round;
  auto ref = [&](int round) {
    TileRef t;
    const int u = round * per_round + blockIdx.x * GNN_UNITS + unit;
    t.nch = 0; t.xp = p0.X + 4 * q; t.wp = p0.W + 4 * q; t.i0 = 0; t.j0 = 0; t.p = &p0;
    if (round < rounds && u < total) {
      const Prod& p = u < t0 ? p0 : p1;
      const int v = u < t0 ? u : u - t0, ct = p.Nc >> 4;
      const int kq_len = (((p.K + GNN_KSPLIT - 1) / GNN_KSPLIT) + 15) & ~15, k0 = kq * kq_len, k1 = min(p.K, k0 + kq_len);
      t.p = &p; t.i0 = 16 * (v / ct); t.j0 = 16 * (v % ct);
      t.nch = k1 > k0 ? (k1 - k0) >> 4 : 0;         // <= 8: K <= 512 (checked by the host side)
      if (t.nch > 0) {
        t.xp = p.X + (size_t)(t.i0 + r) * p.ldx + k0 + 4 * q;
        t.wp = p.W + (size_t)(t.j0 + r) * p.ldw + k0 + 4 * q;
      }
    }
    return t;
  };
  float4 xa[8], wa[8], xb[8], wb[8];
  // (a wave without a tile, or whose quarter of K is empty or short, still loads -- from the start of the matrices -- and skips the MFMAs:
  //  unconditional loads keep the two register sets free of merges with their previous contents)
#define GNN_LOAD(x_, w_, t_) _Pragma("unroll") for (int u = 0; u < 8; ++u) { const int uu = u < (t_).nch ? u : 0; x_[u] = ld_x4((t_).xp + 16 * uu); w_[u] = *reinterpret_cast<const float4*>((t_).wp + 16 * uu); }
#define GNN_MFMA(x_, w_, t_) _Pragma("unroll") for (int u = 0; u < 8; ++u) if (u < (t_).nch) {         \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x_[u].x, w_[u].x, acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x_[u].y, w_[u].y, acc1, 0, 0, 0);                    \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x_[u].z, w_[u].z, acc0, 0, 0, 0);                    \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x_[u].w, w_[u].w, acc1, 0, 0, 0); }
#define GNN_REST(t_)
#define GNN_FINISH(t_, round_)                                                                                   \
  {                                                                                                             \
    const int par = (round_) & 1;                                                                               \
    const f32x4 sum_ = acc0 + acc1;                                                                             \
    L.part[par][unit][kq][lane] = make_float4(sum_[0], sum_[1], sum_[2], sum_[3]);                              \
    __syncthreads();                                                                                            \
    if (kq == 0 && (t_).nch > 0) {                                                                              \
      const float4 s0 = L.part[par][unit][0][lane], s1 = L.part[par][unit][1][lane], s2 = L.part[par][unit][2][lane], s3 = L.part[par][unit][3][lane]; \
      const float v4[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)}; \
      const Prod& p = *(t_).p;                                                                                  \
      /* C / D layout of the 16 x 16 forms: col = lane & 15, row = 4 (lane >> 4) + reg */                       \
      const int col = (t_).j0 + r;                                                                              \
      const float bj = (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS) ? p.bias[col] : 0.f;                           \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                           \
        const int row = (t_).i0 + 4 * q + g;                                                                    \
        float v = v4[g];                                                                                        \
        if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bj, 0.f);                                                       \
        else if (EPI == EPI_BIAS) v = v + bj;                                                                   \
        else if (EPI == EPI_ADD2_RELU) v = fmaxf((v + ld_x(p.add0 + (size_t)row * p.ldadd + col)) + ld_x(p.add1 + (size_t)row * p.ldadd + col), 0.f); \
        st_x(p.out + (size_t)row * p.ldo + col, v);                                                             \
      }                                                                                                         \
    }                                                                                                           \
  }
  TileRef ta = ref(0), tb = ta;
  GNN_LOAD(xa, wa, ta)
  for (int round = 0; round < rounds; round += 2) {
    tb = ref(round + 1);
    GNN_LOAD(xb, wb, tb)
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      GNN_MFMA(xa, wa, ta)
      GNN_REST(ta)
      GNN_FINISH(ta, round)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (round + 1 < rounds) {
      ta = ref(round + 2);
      GNN_LOAD(xa, wa, ta)
      __builtin_amdgcn_sched_barrier(0);
      {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        GNN_MFMA(xb, wb, tb)
        GNN_REST(tb)
        GNN_FINISH(tb, round + 1)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef GNN_LOAD
#undef GNN_MFMA
#undef GNN_REST
#undef GNN_FINISH
  __syncthreads();       // the last round's partials are read before the next layer's first round writes the other half ... and this one again
}

__global__ __launch_bounds__(GNN_THREADS) void gnn_propagate_kernel(const GnnArgs a) {
  const int N = a.N, E = a.E, H = a.H;
  const int gtid = blockIdx.x * GNN_THREADS + threadIdx.x, gthreads = gridDim.x * GNN_THREADS;
  unsigned phase = 0;
  const unsigned G = gridDim.x;
  // (block 0 leaves a 100 MHz time stamp per stage in the workspace: tools/gnn_stage_times.py)
#define GNN_BARRIER() { grid_barrier(a.sync, ++phase * G); if (blockIdx.x == 0 && threadIdx.x == 0) a.stamps[phase] = wall_clock64(); }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.stamps[0] = wall_clock64();
  __shared__ GnnLds L;
  const Prod none = {nullptr, 0, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0};

  // ---- stage 1: first encoder layers (K = a handful of inputs: plain dot products, the relation inputs formed on the fly) and the
  // receivers' segments of the relation list
  {
    const int Dp = a.Dp, A = a.A, Gd = a.G, S = a.S, F = A + Gd + S, Dr = 2 * A + 1 + S;
    for (int o = gtid; o < N * H; o += gthreads) {
      const int i = o / H, j = o - i * H;
      const float* __restrict__ x = a.p_in + (size_t)i * Dp;
      const float* __restrict__ w = a.pe_w0 + (size_t)j * Dp;
      float s = 0.f;
      for (int k = 0; k < Dp; ++k) s = __builtin_fmaf(x[k], w[k], s);
      st_x(a.xp0 + o, fmaxf(s + a.pe_b0[j], 0.f));
    }
    for (int o = gtid; o < E * H; o += gthreads) {
      const int e = o / H, j = o - e * H;
      const float* __restrict__ xr = a.nodes + (size_t)a.recv[e] * F;
      const float* __restrict__ xs = a.nodes + (size_t)a.send[e] * F;
      const float* __restrict__ w = a.re_w0 + (size_t)j * Dr;
      float s = 0.f;
      for (int k = 0; k < A; ++k) s = __builtin_fmaf(xr[k], w[k], s);
      for (int k = 0; k < A; ++k) s = __builtin_fmaf(xs[k], w[A + k], s);
      float gd = 0.f;
      for (int k = 0; k < Gd; ++k) gd += fabsf(xr[A + k] - xs[A + k]);
      s = __builtin_fmaf(gd, w[2 * A], s);
      for (int k = 0; k < S; ++k) s = __builtin_fmaf(xr[A + Gd + k] - xs[A + Gd + k], w[2 * A + 1 + k], s);
      st_x(a.xr0 + o, fmaxf(s + a.re_b0[j], 0.f));
    }
    if (blockIdx.x == 0) {       // row_start[i] = first relation whose receiver is >= i (the list is ascending in the receiver)
      for (int e = threadIdx.x; e <= E; e += GNN_THREADS) {     // relation e opens the segments of the rows in (receiver[e - 1], receiver[e]]
        const long long prev = e > 0 ? a.recv[e - 1] : -1ll, cur = e < E ? a.recv[e] : (long long)N;
        if (prev > cur) __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (long long row = prev + 1; row <= cur && row <= (long long)N; ++row)
          __hip_atomic_store(a.row_start + row, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  GNN_BARRIER();
  // ---- encoder layers 2 and 3
  layer<EPI_BIAS_RELU>(Prod{a.xp0, H, a.pe_w1, H, H, a.xp1, H, a.pe_b1, nullptr, nullptr, 0, N, H},
                       Prod{a.xr0, H, a.re_w1, H, H, a.xr1, H, a.re_b1, nullptr, nullptr, 0, E, H}, L);
  GNN_BARRIER();
  layer<EPI_BIAS_RELU>(Prod{a.xp1, H, a.pe_w2, H, H, a.pe, H, a.pe_b2, nullptr, nullptr, 0, N, H},
                       Prod{a.xr1, H, a.re_w2, H, H, a.xr0, H, a.re_b2, nullptr, nullptr, 0, E, H}, L);
  GNN_BARRIER();
  // ---- the step-invariant parts of the two propagators
  layer<EPI_BIAS>(Prod{a.pe, H, a.pp_w, 2 * H, H, a.pewp, H, a.pp_b, nullptr, nullptr, 0, N, H},
                  Prod{a.xr0, H, a.rp_w, 3 * H, H, a.rew1, H, a.rp_b, nullptr, nullptr, 0, E, H}, L);
  GNN_BARRIER();
  const float* eff = a.pe;
  for (int s = 0; s < a.pstep; ++s) {
    // effect @ W2^T | effect @ W3^T
    layer<EPI_NONE>(Prod{eff, H, a.rp_w + H, 3 * H, H, a.a23, 2 * H, nullptr, nullptr, nullptr, 0, N, H},
                    Prod{eff, H, a.rp_w + 2 * H, 3 * H, H, a.a23 + H, 2 * H, nullptr, nullptr, nullptr, 0, N, H}, L);
    GNN_BARRIER();
    // agg[i] = sum over the relations e of receiver i, in list order, of relu(rew1[e] + a2[i] + a3[sender_e])
    for (int o = gtid; o < N * (H >> 2); o += gthreads) {
      const int i = o / (H >> 2), k4 = (o - i * (H >> 2)) << 2;
      const float4 a2 = ld_x4(a.a23 + (size_t)i * 2 * H + k4);
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool dummy = i == N - 1;                 // the padding's dummy row collects every dummy relation: nobody reads its effect
      const int e1 = dummy ? 0 : ld_i(a.row_start + i + 1);
      for (int e = dummy ? 0 : ld_i(a.row_start + i); e < e1; ++e) {
        const float4 r1 = ld_x4(a.rew1 + (size_t)e * H + k4);
        const float4 a3 = ld_x4(a.a23 + (size_t)a.send[e] * 2 * H + H + k4);
        sum.x += fmaxf((r1.x + a2.x) + a3.x, 0.f); sum.y += fmaxf((r1.y + a2.y) + a3.y, 0.f);
        sum.z += fmaxf((r1.z + a2.z) + a3.z, 0.f); sum.w += fmaxf((r1.w + a2.w) + a3.w, 0.f);
      }
      float* dst = a.agg + (size_t)i * H + k4;
      st_x(dst, sum.x); st_x(dst + 1, sum.y); st_x(dst + 2, sum.z); st_x(dst + 3, sum.w);
    }
    GNN_BARRIER();
    // effect' = relu(particle_encode @ Wp1^T + b + agg @ Wp2^T + effect)
    float* nxt = (s & 1) ? a.eff1 : a.eff0;
    layer<EPI_ADD2_RELU>(Prod{a.agg, H, a.pp_w + H, 2 * H, H, nxt, H, nullptr, a.pewp, eff, H, N, H}, none, L);
    GNN_BARRIER();
    eff = nxt;
  }
  // ---- head
  layer<EPI_BIAS_RELU>(Prod{eff, H, a.h_w0, H, H, a.xp0, H, a.h_b0, nullptr, nullptr, 0, N, H}, none, L);
  GNN_BARRIER();
  layer<EPI_BIAS_RELU>(Prod{a.xp0, H, a.h_w1, H, H, a.xp1, H, a.h_b1, nullptr, nullptr, 0, N, H}, none, L);
  GNN_BARRIER();
  {  // three outputs per node: one wave per (node, output), k ascending within a lane, lanes summed by a butterfly
    const int wave = blockIdx.x * GNN_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * GNN_WAVES, lane = threadIdx.x & 63;
    for (int o = wave; o < 3 * N; o += nwaves) {
      const int i = o / 3, c = o - 3 * i;
      const float* __restrict__ x = a.xp1 + (size_t)i * H;
      const float* __restrict__ w = a.h_w2 + (size_t)c * H;
      float s = 0.f;
      for (int k = lane; k < H; k += 64) s = __builtin_fmaf(ld_x(x + k), w[k], s);
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
      if (lane == 0) {
        const float mot = s + a.h_b2[c];
        a.out_mot[o] = mot;
        a.out_pos[o] = a.last_pos[(size_t)i * a.last_stride + c] + fminf(fmaxf(mot, -a.clamp), a.clamp);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) a.stamps[phase + 1] = wall_clock64();
  // the last workgroup out resets the barrier words for the next launch (every workgroup has left the last barrier by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
      __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace gsr_gnn

// ---------------------------------------------------------------- pieces for the GEMM-library path (the default)
// The same algebra with the products left to the library: what remains between them is (a) the relation inputs and (b) the aggregation
//   agg[i] = sum over the relations e of receiver i, in list order, of relu(rew1[e] + a2[i] + a3[sender_e])
// -- (b) replaces, per propagation step, two row gathers, a concatenation, an E x 3H x H product, a zero fill and an index_add.
namespace gsr_gnn {
__global__ __launch_bounds__(256) void aggregate_kernel(int N, int n_sum, int H, const float* __restrict__ rew1, const float* __restrict__ a23,
                                                        const long long* __restrict__ send, const long long* __restrict__ row_start,
                                                        float* __restrict__ agg) {
  const int o = blockIdx.x * 256 + threadIdx.x, q = H >> 2;
  if (o >= N * q) return;
  const int i = o / q, k4 = (o - i * q) << 2;
  const float4 a2 = *reinterpret_cast<const float4*>(a23 + (size_t)i * 2 * H + k4);
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int e1 = i < n_sum ? (int)row_start[i + 1] : 0;      // rows >= n_sum (the padding's dummy row, which collects every dummy relation) get zeros
  for (int e = i < n_sum ? (int)row_start[i] : 0; e < e1; ++e) {
    const float4 r1 = *reinterpret_cast<const float4*>(rew1 + (size_t)e * H + k4);
    const float4 a3 = *reinterpret_cast<const float4*>(a23 + (size_t)send[e] * 2 * H + H + k4);
    sum.x += fmaxf((r1.x + a2.x) + a3.x, 0.f); sum.y += fmaxf((r1.y + a2.y) + a3.y, 0.f);
    sum.z += fmaxf((r1.z + a2.z) + a3.z, 0.f); sum.w += fmaxf((r1.w + a2.w) + a3.w, 0.f);
  }
  *reinterpret_cast<float4*>(agg + (size_t)i * H + k4) = sum;
}
// rel_inputs[e] = (attrs[recv], attrs[send], sum |instance[recv] - instance[send]|, state[recv] - state[send]); nodes = [attrs | instance | state]
__global__ __launch_bounds__(256) void rel_inputs_kernel(int E, int A, int G, int S, const float* __restrict__ nodes,
                                                         const long long* __restrict__ recv, const long long* __restrict__ send,
                                                         float* __restrict__ out) {
  const int Dr = 2 * A + 1 + S, F = A + G + S, o = blockIdx.x * 256 + threadIdx.x;
  if (o >= E * Dr) return;
  const int e = o / Dr, c = o - e * Dr;
  const float* __restrict__ xr = nodes + (size_t)recv[e] * F;
  const float* __restrict__ xs = nodes + (size_t)send[e] * F;
  float v;
  if (c < A) v = xr[c];
  else if (c < 2 * A) v = xs[c - A];
  else if (c == 2 * A) { v = 0.f; for (int k = 0; k < G; ++k) v += fabsf(xr[A + k] - xs[A + k]); }
  else v = xr[A + G + (c - 2 * A - 1)] - xs[A + G + (c - 2 * A - 1)];
  out[o] = v;
}
}  // namespace gsr_gnn

int gsr_launch_gnn_aggregate(int N, int n_sum, int H, const float* rew1, const float* a23, const long long* send, const long long* row_start, float* agg,
                             hipStream_t st) {
  { GSR_PROF("gnn_aggregate", st);
    hipLaunchKernelGGL(gsr_gnn::aggregate_kernel, dim3((N * (H >> 2) + 255) / 256), dim3(256), 0, st, N, n_sum, H, rew1, a23, send, row_start, agg); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
int gsr_launch_gnn_rel_inputs(int E, int A, int G, int S, const float* nodes, const long long* recv, const long long* send, float* out,
                              hipStream_t st) {
  { GSR_PROF("gnn_rel_inputs", st);
    hipLaunchKernelGGL(gsr_gnn::rel_inputs_kernel, dim3((E * (2 * A + 1 + S) + 255) / 256), dim3(256), 0, st, E, A, G, S, nodes, recv, send, out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_gnn_propagate(const GsrGnnArgs& a, int workgroups, hipStream_t st) {
  { GSR_PROF("gnn_propagate", st);
    hipLaunchKernelGGL(gsr_gnn::gnn_propagate_kernel, dim3(workgroups), dim3(GNN_THREADS), 0, st, a); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
