// gsr_render.hip -- per-tile alpha-compositing with depth, forward and backward, for gfx950
// (replaces the reference extension's two renderCUDA kernels).
//
// Tile work: a 256-thread workgroup (4 waves of 64) renders one 16x16 tile, one lane per pixel; a wave
// owns an 8x8 pixel quad ("strip" below = that wave's pixel region; 8x8 regions intersect 11 % fewer list
// entries than 16x4 rows on the benchmark scene: measured -4 % forward, -2 % backward kernel time).
//
// Staging + strip culling: the tile's depth-sorted list is staged through LDS 128 entries at a time (one
// entry gathered per lane: 16 + 16 + 8 + 8 B from the record arrays).  Each staged entry carries the
// conservative pixel box of its alpha >= 1/255 ellipse (preprocess); the staging lanes classify it against
// the tile's four strips and the workgroup builds FOUR per-strip compacted lists in LDS with wave-64
// __ballot + popcount prefixes (stable, so blend order is preserved).  A wave then walks only the entries
// that can reach its strip -- on the benchmark scene about half of the (strip, entry) pairs vanish, and
// entries that touch no strip at all (30 % of the reference's 3-sigma-rect duplicates) cost nothing.
// Results are unchanged: a skipped pair would have failed the alpha test.
//
// Forward: every lane walks its wave's list with broadcast ds_read_b128 (all lanes read the same address:
// conflict-free).  Early termination is per WAVE (`__ballot(!done) == 0`) and per workgroup.
//
// Backward: lanes replay their pixel back-to-front.  For every list entry the nine partial gradients are
// summed over the wave's 64 pixels with a packed DPP / swizzle reduction (one value per lane at the end), nine lanes
// park the wave totals in LDS, and after the batch one lane per entry adds the four wave totals and writes ONE 36-byte record to
// the entry's Gaussian-major slot.  No global atomics on the data path: gradients are deterministic, and
// the per-Gaussian reduce in gsr_preprocess_bwd.hip reads contiguous records.
//
// Scheduling: tile list lengths are wildly uneven (on the benchmark scene 54 % of the tiles are empty and
// the rest hold ~500 entries, up to 1130).  Instead of one workgroup per tile in blockIdx order, a fixed
// grid of PERSISTENT workgroups pops tickets (one device-scope atomicAdd per tile) from a queue of tiles
// sorted longest-first (tile_order_kernel in gsr_binning.hip): greedy longest-processing-time scheduling,
// so every CU stays occupied until the queue drains and the tail consists of the cheapest tiles.
#include <stdlib.h>
#include "gsr_common.h"

// Optional phase timing of the forward tile loop (debug build only: make TIMING=1).  Wave 0 lane 0 of every
// workgroup accumulates s_memtime deltas per phase and adds them to g_fwd_timing at kernel end.
#ifdef GSR_TILE_TIMING
__device__ unsigned long long g_fwd_timing[16];
__device__ unsigned long long g_bwd_timing[16];
#define GSR_TFLUSH_B() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 8; ++_i) atomicAdd(&g_bwd_timing[_i], _t_acc[_i]); atomicAdd(&g_bwd_timing[15], 1ull); } } while (0)
#define GSR_T0() unsigned long long _t_prev = __builtin_readcyclecounter(), _t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define GSR_TP(i) do { const unsigned long long _t = __builtin_readcyclecounter(); _t_acc[i] += _t - _t_prev; _t_prev = _t; } while (0)
#define GSR_TFLUSH() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 8; ++_i) atomicAdd(&g_fwd_timing[_i], _t_acc[_i]); atomicAdd(&g_fwd_timing[15], 1ull); } } while (0)
#else
#define GSR_T0() do {} while (0)
#define GSR_TP(i) do {} while (0)
#define GSR_TFLUSH() do {} while (0)
#define GSR_TFLUSH_B() do {} while (0)
#endif

// kernels live in a NAMED namespace: profilers and traces show gsr_render::<kernel>, not "(anonymous namespace)"
namespace gsr_render {

#define GSR_QW 8   // pixel region of one wave inside the 16x16 tile: 8x8 quad (2 x 2 quads per tile)
#define GSR_QH 8
#define GSR_LOG2E 1.44269502162933349609375f
// The exponent of a visit.  The staged conic carries the EXACT factors of the quadratic form only -- (-A/2, -B, -C/2): powers of two, no
// rounding -- and the exponent is converted to v_exp_f32's base-2 units per visit (one multiply).  Round 5 staged (-A/2, -B, -C/2) * log2(e):
// one rounding of each coefficient per Gaussian, the same for every pixel -- a 6e-8 perturbation that a nearly singular conic (elongated
// Gaussians larger than their image) amplified to 1e-4 .. 1e-3 of that Gaussian's gradient row (profiles/r05_conic_prescale_precision.txt;
// the parity soak's seed 77 case 671 and GSR_SOAK_BIG seed 6 case 23 missed the norm-wise bar by it; round 6, profiles/r06_conic_forms.txt:
// this form puts case 671's `scales` 1.9e-5 from fp64 where the pre-scaled one sat at 1.17e-4, for +1.0 % of the step; the oracle's
// operation order -- every product rounded -- is no closer to fp64 than this one and costs three issues more).
#define GSR_STAGE_A(A) (-0.5f * (A))
#define GSR_STAGE_B(B) (-(B))
__device__ __forceinline__ float gsr_power(float hA, float hB, float hC, float dx, float dy) {
  return __builtin_fmaf(__builtin_fmaf(hB, dy, hA * dx), dx, (hC * dy) * dy);
}
#define GSR_EXP_OF_POWER(p) __builtin_amdgcn_exp2f(GSR_LOG2E * (p))
#ifndef FWD_BATCH
#define FWD_BATCH 128
#define FWD_UNROLL 8     // entries per unrolled block of the forward blend loop
#ifndef FWD_UNROLL_PAIR
#define FWD_UNROLL_PAIR 3   // the pair build: six colour accumulators; 3 -> 90 VGPRs (five workgroups per CU need <= 96), 4 -> 98
#endif
#endif
#ifndef BWD_BATCH
#define BWD_BATCH 128
#endif
#ifndef FWD_WAVES_PER_EU
#define FWD_WAVES_PER_EU 1
#endif
#ifndef BWD_SMALL_BB
// Batch of the long-queue backward build; its LDS decides the workgroups per CU.  Rounds 2-3: 96 entries (30 KB) at five per CU.  With the
// exact per-quad lists of round 4 the build needs 80 VGPRs, so six fit: 80 entries (25 KB) at six per CU -- render_bwd 397 -> 389.5 us at
// eight views, equal at four (profiles/r04_exact_lists.txt).
#define BWD_SMALL_BB 80
#define BWD_SMALL_WAVES 6
#endif
#ifndef FWD_TRACK_WAVES
#define FWD_TRACK_WAVES 6    // the tracking build is bounded to the occupancy the plain one reaches by itself (75 VGPRs)
#endif
#ifndef BWD_WAVES_PER_EU
#define BWD_WAVES_PER_EU 1
#endif

__device__ __forceinline__ int sext16(uint32_t v) { return (int)(short)(v & 0xffffu); }

// Selects driven by an explicit 64-bit lane mask in a scalar register pair (what v_cndmask takes): the tracking forward keeps its predicates
// as masks -- ballots of fresh compares fold into the compares' SGPR results, the combinations are scalar ANDs / XORs, and "did ANY pixel
// blend this entry" is one scalar compare on the mask the selects use anyway.  (Written in C++ with bools, __ballot(blend) made the
// compiler rebuild the predicate in a VGPR: +2 VALU per entry and 40 VGPRs of pressure.)
__device__ __forceinline__ float gsr_sel(uint64_t m, float a, float b) {          // m ? a : b, per lane
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ float gsr_sel_or_zero(uint64_t m, float a) {           // m ? a : 0
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ float gsr_sel_neg_abs(uint64_t m, float a) {           // m ? -|a| : a
  float r;
  asm("v_cndmask_b32_e64 %0, %1, -|%1|, %2" : "=v"(r) : "v"(a), "s"(m));
  return r;
}
// The any-pixel-blended bits of a walk shift into a scalar register, acc = 2 * acc + (the mask has a lane set); the first entry of a group
// of 32 ends up in the top bit.
// blend = hit ^ stop (stop implies hit) AND the shift-in of "blend has a lane set" in one go: a scalar logical operation leaves
// SCC = (result != 0), so the s_cmp of gsr_shift_in_any is the xor the masks need anyway
__device__ __forceinline__ uint64_t gsr_xor_shift_in_any(uint64_t hit, uint64_t stop, uint32_t& acc) {
  uint64_t blend;
  asm("s_xor_b64 %0, %2, %3\n\ts_addc_u32 %1, %1, %1" : "=s"(blend), "+s"(acc) : "s"(hit), "s"(stop) : "scc");
  return blend;
}
__device__ __forceinline__ uint32_t gsr_sel_u(uint64_t m, uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}

// Which of the tile's four per-wave pixel regions (8x8 quads) can this Gaussian reach with alpha >= 1/255?  (bit w = wave w)
// Level 1: the integer pixel box from preprocess.  Level 2, for strips that pass: the exact minimum of the
// quadratic form over the strip rectangle (0 when the mean is inside, else the least edge minimum -- the
// form is convex) against 2 ln(255 o) + 0.04.  Both are conservative: a pair they drop fails the alpha
// test at every pixel of the strip, so results are unchanged.
__device__ __forceinline__ uint32_t strip_mask(uint2 box, float4 a, float conicC, float opacity, int tx0, int ty0) {
  const int xmin = sext16(box.x), xmax = sext16(box.x >> 16), ymin = sext16(box.y), ymax = sext16(box.y >> 16);
  if (xmax < tx0 || xmin > tx0 + 15 || xmin > xmax || ymax < ty0 || ymin > ty0 + 15) return 0u;
  const float mx = a.x, my = a.y, A = a.z, B = a.w, C = conicC;
  const float tau2 = 2.0f * (__logf(255.0f * opacity) + 0.02f);
  const float invA = __builtin_amdgcn_rcpf(A), invC = __builtin_amdgcn_rcpf(C);
  uint32_t m = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {   // wave w owns the 8x8 pixel quad (w & 1, w >> 1) of the tile
    const int x0 = tx0 + GSR_QW * (w & 1), y0 = ty0 + GSR_QH * (w >> 1);
    if (xmax < x0 || xmin > x0 + GSR_QW - 1 || ymax < y0 || ymin > y0 + GSR_QH - 1) continue;
    const float dxlo = (float)x0 - mx, dxhi = (float)(x0 + GSR_QW - 1) - mx;
    const float dylo = (float)y0 - my, dyhi = (float)(y0 + GSR_QH - 1) - my;
    bool keep = dxlo <= 0.0f && dxhi >= 0.0f && dylo <= 0.0f && dyhi >= 0.0f;
    if (!keep) keep = gsr_qmin_facing_edges(A, B, C, invA, invC, dxlo, dxhi, dylo, dyhi) <= tau2;   // the two edges that face the mean
    if (keep) m |= 1u << w;
  }
  return m;
}

// Per-strip compacted entry lists of one staged batch (stable: list order is preserved).
// PAIR: the tile pass also blends a partner view that shares this view's camera and differs only in its colours (the
// segmentation render next to the colour render of get_loss, the mask render next to the colour render of predict.py):
// geometry, alpha and transmittance are evaluated once, three more colour accumulators ride along (3 of ~25 VALU per
// pixel-entry pair instead of a second pass).
template <bool PAIR>
struct FwdLdsT {
  float4 sA[4][FWD_BATCH + 1];   // mean2D.x, mean2D.y, conic A, conic B   (+1: the loop prefetches entry j+1)
  float4 sB[4][FWD_BATCH + 1];   // conic C, opacity, r, g
  float4 sC[4][FWD_BATCH + 1];   // b, depth, bits(1-based list position), partner r
  float2 sD[4][PAIR ? FWD_BATCH + 1 : 1];   // partner g, b
  uint32_t cnt[4][4];        // [staging wave][strip]
  uint32_t cmask[4][4];      // TRACK: per wave and group of 32 entries of its compacted list, bit 31 - k = "some pixel of the quad blended entry k of the group"
  uint32_t cpos[2][FWD_BATCH];  // TRACK: per staged entry, its position in each quad's compacted list (8 bits per quad, 0xff = not in that list);
  //                               by batch parity: waves 2 and 3 turn the previous batch's into bytes while waves 0 and 1 stage the next one
};

// the partner's side of a fused pair (all nullptr / unused when !PAIR)
struct FwdPartner {
  const float4* rec; const float* bg; float* final_T; uint32_t* n_contrib; float* out_color; float* out_depth;
  const float* colors;   // != nullptr: the partner's colours come from its [P,3] colour array (it has no records: forward-only calls)
  uint8_t* used;         // TRACK: the partner's per-Gaussian used flags (same lists, same blend decisions: set together with the owner's)
};
__device__ __forceinline__ float3 fwd_partner_colour(const FwdPartner& pt, uint32_t g) {
  if (pt.colors) return make_float3(pt.colors[3 * (size_t)g], pt.colors[3 * (size_t)g + 1], pt.colors[3 * (size_t)g + 2]);
  const float4 q1 = pt.rec[GSR_REC_F4 * g + 1];
  return make_float3(q1.z, q1.w, pt.rec[GSR_REC_F4 * g + 2].x);
}

// TRACK (round 4): the forward records, per list entry and quad, whether ANY pixel of the quad blended the entry -- one byte per entry
// (bit w = quad w) in `contrib`, next to the tile lists.  The backward's hit test of a pixel (list position below its last contributor,
// power <= 0, opacity * G >= 1/255, evaluated on the same LDS-staged values in the same order) is exactly the forward's `blend`, so these
// bytes ARE the backward's per-quad lists: it stages an entry for a quad only when that quad used it -- 46 % of its quad visits evaluated
// 64 pixels to find none (conservative rectangle tests, pixels that had terminated) -- and needs no rectangle test of its own.
// Cost here: one scalar compare-select-or per visit and 128 byte stores per batch.  Forward-only calls (GSR_FORWARD_ONLY) run TRACK = false.
// The contribution bytes of a batch are written by its staging threads once every wave's masks are in LDS, i.e. behind the next
// barrier: the next batch's loop-top barrier, and for the last batch of a tile a barrier of the CALLER (fwd_tile returns the batch's
// first list position, or -1; the persistent kernel stores behind the barrier of its ticket pop: no barrier is added per tile).
template <bool PAIR>
__device__ __forceinline__ void fwd_store_contrib(const FwdLdsT<PAIR>& L, uint8_t* __restrict__ contrib, uint8_t* __restrict__ used,
                                                  uint8_t* __restrict__ used_partner, uint32_t list0, int n, int pend_base, uint32_t pend_g) {
  static_assert(GSR_BLOCK == 2 * FWD_BATCH, "the upper half of the workgroup writes the bytes of the batch the lower half staged");
  const int e = (int)threadIdx.x - FWD_BATCH, pidx = pend_base + e;   // waves 2 and 3: they stage nothing, so the bytes cost the staging waves no time
  if (pend_base >= 0 && e >= 0 && pidx < n && contrib) {   // (contrib == nullptr: a view without lists -- it has no busy tile to get here with)
    const uint32_t pc = L.cpos[(pend_base / FWD_BATCH) & 1][e];
    uint32_t byte = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t pw = (pc >> (8 * w)) & 0xffu;
      if (pw != 0xffu) byte |= ((L.cmask[w][pw >> 5] >> (31u - (pw & 31u))) & 1u) << w;
    }
    contrib[list0 + pidx] = (uint8_t)byte;
    if (byte && used) {            // the Gaussian is blended somewhere in this view (plain byte stores of the same value: no atomics needed)
      used[pend_g] = 1;
      if (PAIR && used_partner) used_partner[pend_g] = 1;
    }
  }
}

template <bool PAIR, bool TRACK = false>
__device__ __forceinline__ int fwd_tile(
    const int tile, const uint2 rg, FwdLdsT<PAIR>& L, int W, int H, int gx,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, const float* __restrict__ bg,
    float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, float* __restrict__ out_color,
    float* __restrict__ out_depth, const FwdPartner pt, uint8_t* __restrict__ contrib, uint8_t* __restrict__ used, uint32_t& pend_g,
    const uint32_t* __restrict__ cut_in = nullptr, uint32_t* __restrict__ cut_out = nullptr, uint32_t* __restrict__ redo = nullptr,
    const float cut_margin = 1.0f) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx0 = (tile % gx) * GSR_TILE, ty0 = (tile / gx) * GSR_TILE;
  const int px = tx0 + GSR_QW * (wv & 1) + (lane & 7), py = ty0 + GSR_QH * (wv >> 1) + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const int n = (int)(rg.y - rg.x);
  pend_g = 0u;                // TRACK, waves 2 - 3: the Gaussian of the entry whose byte this thread writes next (loaded a batch ahead of its use)

  float T = inside ? 1.0f : -1.0f;      // the sign of T is the stopped flag (see GSR_FWD_ENTRY)
#define done (!(T > 0.0f))
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
  float C3 = 0.f, C4 = 0.f, C5 = 0.f;   // PAIR: the partner's colour
  uint32_t last = 0;
  GSR_T0();

  // Software-pipelined staging: the gathers of batch b+1 are issued before batch b is walked, so their
  // latency (point_list -> record arrays, two dependent trips to L2/HBM) hides behind the blend loop.
  float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
  float2 nc = make_float2(0.f, 0.f);
  float3 np = make_float3(0.f, 0.f, 0.f);   // PAIR: partner colour
  uint2 nbox = make_uint2(1u, 1u);
  // The gather is two dependent trips (list entry -> record).  The list entry of batch b+2 is fetched while batch b is walked,
  // so the record loads of batch b+1 issue without waiting for an index.
  uint32_t g_ahead = 0;
  if (tid < FWD_BATCH && tid < n) {
    const uint32_t g = point_list[rg.x + tid];
    if (tid + FWD_BATCH < n) g_ahead = point_list[rg.x + tid + FWD_BATCH];
    { const float4 t2 = rec[GSR_REC_F4 * g + 2]; na = rec[GSR_REC_F4 * g]; nb = rec[GSR_REC_F4 * g + 1]; nc = make_float2(t2.x, t2.y);
      nbox = make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)); }
    if (PAIR) np = fwd_partner_colour(pt, g);
  }
  GSR_TP(0);
  int stop_base = -1;             // >= 0: every pixel had finished when batch `stop_base` came up (the walk ended before the list did)
  int pend_base = -1;             // TRACK (wave-uniform): first list position of the batch whose contribution bytes are due -- they are written
  //                                 by the staging threads once the waves' masks are in LDS (the positions of their entries in the four
  //                                 compacted lists wait in L.cpos: no per-thread register lives across the blend loop for this)
  for (int base = 0; base < n; base += FWD_BATCH) {
    const bool all_done = __syncthreads_count(done) == GSR_BLOCK;  // also fences the previous batch's LDS reads (and publishes its cmask)
    if (TRACK) {
      fwd_store_contrib<PAIR>(L, contrib, used, pt.used, rg.x, n, pend_base, pend_g);
      pend_base = -1;
      if (!all_done && used && tid >= FWD_BATCH && base + tid - FWD_BATCH < n) pend_g = point_list[rg.x + base + tid - FWD_BATCH];
    }
    if (all_done) { stop_base = base; break; }
    GSR_TP(1);
    // ---- stage: threads 0..127 each classify the entry they prefetched against the four strips
    const float4 a = na, b = nb;
    const int idx = base + tid;
    const float4 c = make_float4(nc.x, nc.y, __uint_as_float((uint32_t)(idx + 1)), np.x);
    const float2 d = make_float2(np.y, np.z);
    uint32_t mask = 0;
    if (tid < FWD_BATCH && idx < n) mask = strip_mask(nbox, a, b.x, b.y, tx0, ty0);
    {
      const int nidx = idx + FWD_BATCH;
      if (tid < FWD_BATCH && nidx < n) {
        const uint32_t g = g_ahead;
        if (nidx + FWD_BATCH < n) g_ahead = point_list[rg.x + nidx + FWD_BATCH];
        { const float4 t2 = rec[GSR_REC_F4 * g + 2]; na = rec[GSR_REC_F4 * g]; nb = rec[GSR_REC_F4 * g + 1]; nc = make_float2(t2.x, t2.y);
      nbox = make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)); }
        if (PAIR) np = fwd_partner_colour(pt, g);
      }
    }
    uint64_t bal[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bal[w] = __ballot((mask >> w) & 1u);
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) L.cnt[wv][w] = (uint32_t)__popcll(bal[w]);
    }
    GSR_TP(2);
    __syncthreads();
    GSR_TP(3);
    uint32_t cpos = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if ((mask >> w) & 1u) {
        uint32_t pos = (uint32_t)__popcll(bal[w] & gsr_lanemask_lt());
        for (int v = 0; v < wv; ++v) pos += L.cnt[v][w];
        // the conic staged with its exact factors (-A/2, -B, -C/2): see gsr_power
        L.sA[w][pos] = make_float4(a.x, a.y, GSR_STAGE_A(a.z), GSR_STAGE_B(a.w));
        L.sB[w][pos] = make_float4(GSR_STAGE_A(b.x), b.y, b.z, b.w);
        L.sC[w][pos] = c;
        if (PAIR) L.sD[w][pos] = d;
        if (TRACK) cpos = (cpos & ~(0xffu << (8 * w))) | (pos << (8 * w));
      }
    }
    if (TRACK) { if (tid < FWD_BATCH) L.cpos[(base / FWD_BATCH) & 1][tid] = cpos; pend_base = base; }
    // readfirstlane makes the trip count a scalar
    const int m = __builtin_amdgcn_readfirstlane((int)(L.cnt[0][wv] + L.cnt[1][wv] + L.cnt[2][wv] + L.cnt[3][wv]));
    __syncthreads();
    GSR_TP(4);
    // ---- blend: wave wv walks only the entries that can reach its strip.  Straight-line body: the next
    // entry's record is fetched from LDS while this one is evaluated, and the blend itself is predicated
    // (w = 0 when the pair does not contribute) instead of branched, so consecutive iterations overlap.
    if (TRACK && lane < 4) L.cmask[wv][lane] = 0u;   // groups the walk does not reach (same wave, in order: its later writes win)
    if (__ballot(!done) != 0ull) {
      const float4* __restrict__ wA = L.sA[wv];
      const float4* __restrict__ wB = L.sB[wv];
      const float4* __restrict__ wC = L.sC[wv];
      const float2* __restrict__ wD = L.sD[wv];
      // One list entry: evaluate, then blend predicated.  (Skipping the blend arithmetic of a visit no pixel of the quad uses -- a
      // wave-uniform branch on __ballot(hit) -- was measured 9 % SLOWER: straight-line code lets the compiler overlap the visits.)
      /* Round 4 ("predicates in VALU registers"; measured: render_fwd 45.4 -> 43.5 us at one view, 103.8 -> 102.3 at four, 192.7 -> 189.9 at eight): the pixel's stopped flag lives in the SIGN of T (T > 0: live,
         T = -|T at the stop|: stopped) instead of in an SGPR mask that every entry ANDs into its hit mask and ORs its stop mask into --
         a stopped pixel has test_T < 0 < T_EPS, so `stop` holds for it by itself and `blend` is false: three SALU mask operations
         per entry fewer, one v_cndmask more. */                                                                           \
#define GSR_FWD_ENTRY(ea, eb, ec, ed, UBIT)                                                             \
      {                                                                                             \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                               \
        const float power = gsr_power(ea.z, ea.w, eb.x, dx, dy);                                    \
        const float alpha = fminf(GSR_ALPHA_MAX, eb.y * GSR_EXP_OF_POWER(power));                   \
        const float test_T = T * (1.0f - alpha);                                                    \
        if constexpr (TRACK) {                                                                      \
          /* the same predicates as lane masks (see gsr_sel): hit = power <= 0 && alpha >= 1/255, stop = hit && test_T < eps, blend = hit ^ stop */ \
          const uint64_t mh = __ballot(power <= 0.0f) & __ballot(alpha >= GSR_ALPHA_MIN);           \
          const uint64_t ms = mh & __ballot(test_T < GSR_T_EPS);                                    \
          const uint64_t mb = gsr_xor_shift_in_any(mh, ms, acc);  /* blend = hit ^ stop, + "some pixel of the quad blended this entry" */ \
          const float w = gsr_sel_or_zero(mb, alpha * T);                                           \
          C0 = __builtin_fmaf(eb.z, w, C0); C1 = __builtin_fmaf(eb.w, w, C1);                       \
          C2 = __builtin_fmaf(ec.x, w, C2); Dp = __builtin_fmaf(ec.y, w, Dp);                       \
          if (PAIR) { C3 = __builtin_fmaf(ec.w, w, C3); C4 = __builtin_fmaf(ed.x, w, C4); C5 = __builtin_fmaf(ed.y, w, C5); } \
          T = gsr_sel_neg_abs(ms, gsr_sel(mb, test_T, T));                                          \
          last = gsr_sel_u(mb, __float_as_uint(ec.z), last);                                        \
        } else {                                                                                    \
        const bool hit = power <= 0.0f && alpha >= GSR_ALPHA_MIN;                                   \
        const bool stop = hit && test_T < GSR_T_EPS;                                                \
        const bool blend = hit != stop;                                                             \
        const float w = blend ? alpha * T : 0.0f;                                                   \
        C0 = __builtin_fmaf(eb.z, w, C0); C1 = __builtin_fmaf(eb.w, w, C1);                         \
        C2 = __builtin_fmaf(ec.x, w, C2); Dp = __builtin_fmaf(ec.y, w, Dp);                         \
        if (PAIR) { C3 = __builtin_fmaf(ec.w, w, C3); C4 = __builtin_fmaf(ed.x, w, C4); C5 = __builtin_fmaf(ed.y, w, C5); } \
        T = blend ? test_T : T;                                                                     \
        T = stop ? -__builtin_fabsf(T) : T;                                                         \
        last = blend ? __float_as_uint(ec.z) : last;                                                \
        }                                                                                           \
      }
      // Blocks of FWD_UNROLL entries, fully unrolled: the LDS reads are immediate offsets off one running pointer, the compiler
      // places them ahead of their uses without register rotation, and the all-done check runs once per block.  (Round 1's form --
      // two entries per trip on ping-pong registers with the check folded into the trip -- compiled to five register copies, two
      // address computations and four scalar branches per trip: render_fwd 222 -> 201 us at 8 views, 58 -> 49 us at one view.  The
      // backward's visits branch on __ballot(hit), the loads cannot move across that, and there the hand-rotated prefetch is 3 %
      // faster than blocks.)
#ifndef FWD_UNROLL_TRACK
#define FWD_UNROLL_TRACK 8      // the tracking build: blocks of eight like the plain one (76 VGPRs, no spills, with the shift-register form of the bits)
#endif
#ifndef FWD_UNROLL_PAIR_TRACK
#define FWD_UNROLL_PAIR_TRACK 2
#endif
      constexpr int UN = PAIR ? (TRACK ? FWD_UNROLL_PAIR_TRACK : FWD_UNROLL_PAIR) : (TRACK ? FWD_UNROLL_TRACK : FWD_UNROLL);
      if constexpr (TRACK) {
        // The any-pixel-blended bit of each entry shifts into a scalar register (gsr_shift_in_any).  The list is walked as groups of
        // 32 entries, one register each; a group that ends early (list end, every pixel finished) is shifted up so that entry k of
        // a group always sits at bit 31 - k.  (Round 4's first form -- static bit positions ORed per block and placed with a 64-bit
        // shift and a three-way branch on the word -- cost a compare-select-or per entry plus four branches and six register copies
        // per block.)
        static_assert(32 % UN == 0, "a block of the tracking walk must not straddle two groups");
        bool live = true;
#pragma unroll 1
        for (int g = 0; g < 4 && live; ++g) {
          const int mm = min(m - 32 * g, 32);
          if (mm <= 0) break;
          const float4* __restrict__ hA = wA + 32 * g;
          const float4* __restrict__ hB = wB + 32 * g;
          const float4* __restrict__ hC = wC + 32 * g;
          const float2* __restrict__ hD = wD + 32 * g;
          uint32_t acc = 0u;
          int j = 0;
          for (; j + UN <= mm; j += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              const float4 ea = hA[j + u], eb = hB[j + u], ec = hC[j + u];
              const float2 ed = PAIR ? hD[j + u] : make_float2(0.f, 0.f);
              GSR_FWD_ENTRY(ea, eb, ec, ed, u)
            }
            if (__ballot(!done) == 0ull) { j += UN; live = false; break; }
          }
          if (live)
            for (; j < mm; ++j) {
              const float4 ea = hA[j], eb = hB[j], ec = hC[j];
              const float2 ed = PAIR ? hD[j] : make_float2(0.f, 0.f);
              GSR_FWD_ENTRY(ea, eb, ec, ed, 0)
            }
          acc <<= (32 - j) & 31;                   // j = entries walked in this group (1..32)
          if (lane == 0) L.cmask[wv][g] = acc;     // read by the staging threads behind the next barrier
        }
      } else {
        uint32_t acc = 0u;        // (unused: the entry body names it only under TRACK)
        int j = 0;
        for (; j + UN <= m; j += UN) {
#pragma unroll
          for (int u = 0; u < UN; ++u) {
            const float4 ea = wA[j + u], eb = wB[j + u], ec = wC[j + u];
            const float2 ed = PAIR ? wD[j + u] : make_float2(0.f, 0.f);
            GSR_FWD_ENTRY(ea, eb, ec, ed, u)
          }
          if (__ballot(!done) == 0ull) { j = m; break; }
        }
        for (; j < m; ++j) {
          const float4 ea = wA[j], eb = wB[j], ec = wC[j];
          const float2 ed = PAIR ? wD[j] : make_float2(0.f, 0.f);
          GSR_FWD_ENTRY(ea, eb, ec, ed, 0)
        }
        (void)acc;
      }
#undef GSR_FWD_ENTRY
    }
    GSR_TP(5);
  }
  GSR_TP(1);
  // Speculative depth cuts (gsr_arm_depth_cuts; forward-only calls of a frame sequence): this tile's proposal for the NEXT frame, and the
  // verdict on the cut THIS frame was binned with.  The walk ended before the list did: every pixel is finished whatever lies deeper -- the
  // result is exact under any cut, and the deepest entry already prefetched (up to 128 positions past the last one walked) bounds what the
  // next frame needs, times the caller's margin.  The list ran out: with pixels still alive, entries a cut removed might have reached them -- if this
  // tile was cut, the frame must be redone without cuts (the caller's flag); either way the tile proposes no cut.
  if (cut_out) {
    if (stop_base >= 0) {
      if (tid == min(FWD_BATCH - 1, n - 1 - stop_base)) cut_out[tile] = __float_as_uint(nc.y * cut_margin);
    } else {
      const bool alive = __syncthreads_count(!done) != 0;
      if (tid == 0) {
        cut_out[tile] = 0x7f800000u;
        if (alive && cut_in && cut_in[tile] != 0x7f800000u && redo) atomicAdd(redo, 1u);   // (rare; the count of such tiles is a diagnostic, any non-zero value means redo)
      }
    }
  }
#undef done
  T = __builtin_fabsf(T);
  if (inside) {
    const int pix = py * W + px;
    const size_t N = (size_t)H * W;
    final_T[pix] = T;
    n_contrib[pix] = last;
    out_color[pix] = C0 + T * bg[0];
    out_color[N + pix] = C1 + T * bg[1];
    out_color[2 * N + pix] = C2 + T * bg[2];
    out_depth[pix] = Dp;
    if (PAIR) {   // same geometry: same transmittance, contributor count and depth
      pt.final_T[pix] = T;
      pt.n_contrib[pix] = last;
      pt.out_color[pix] = C3 + T * pt.bg[0];
      pt.out_color[N + pix] = C4 + T * pt.bg[1];
      pt.out_color[2 * N + pix] = C5 + T * pt.bg[2];
      pt.out_depth[pix] = Dp;
    }
  }
  GSR_TP(6);
  GSR_TFLUSH();
  return pend_base;     // TRACK: the last batch's contribution bytes are still due (fwd_store_contrib, behind a barrier of the caller)
}

// ------------------------------------------------------------------------------------------ backward
// PAIR (fused pair, see FwdLdsT): both views' dL/dcolour drive ONE replay of the list.  Only used when no colour gradient is
// wanted (colours are frozen while tracking): the record then carries
//   { sum t dx, sum t dy, sum tx dx, sum tx dy, sum ty dy, sum G dL/dalpha }  of both views together  (the Gaussian's geometry
//   gradients only ever need the sum over views)  and  { sum t_A dx, sum t_A dy }  of this view alone, from which preprocess_bwd
//   forms the two per-view screen-space gradients -- eight sums instead of 2 x 9, one alpha evaluation instead of two.
// The pair build stages 96 entries per batch instead of 128 and keeps 8 sums per entry: with the partner colours its LDS
// footprint then still allows four workgroups per CU (3 -> 4 measured +4 % on the get_loss step).
// Plain passes come in two batch sizes: 128 entries (39.7 KB of LDS, four workgroups per CU) for launches with few busy tiles --
// there a tile's critical path sets the kernel's time, and it grows with the number of batches -- and 96 entries (29.9 KB, five per
// CU) when the queue is long and latency hiding is what counts: -4 % kernel time at 4 views x 2500 tiles, +3 % at one view.
// The nine per-entry wave totals of a visit: the packed reduction finishes in the wave, nine lanes store.  (Round 3 measured the alternative --
// stop after the in-row levels and let one ds_add_f32 add the four rows -- slower: profiles/r03_rejected_lds_accumulate.txt.)
#ifdef GSR_ABL_NOREDUCE   // ablation build (tools/r05_ablation.sh): the nine partials are added up per lane (8 adds keep them alive) instead of reduced over the wave (24 issues)
#define GSR_BWD_PARK9(a0, a1, a2, a3, a4, a5, a6, a7, a8)                                            \
        { const float z = (((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7))) + a8;                  \
          if (lane >= 48 && lane <= 56) L.sRed[wv][j][lane - 48] = z; }
#else
#define GSR_BWD_PARK9(a0, a1, a2, a3, a4, a5, a6, a7, a8)                                            \
        { const float z = gsr_wave_sum9_packed<ROWS_PERM>(a0, a1, a2, a3, a4, a5, a6, a7, a8);       \
          if (lane >= 48 && lane <= 56) L.sRed[wv][j][lane - 48] = z; /* one ds_write_b32 */ }
#endif
#define GSR_BWD_BB(PAIR) ((PAIR) ? 96 : BWD_BATCH)
template <bool PAIR, int NBB = GSR_BWD_BB(PAIR)>
struct BwdLdsT {
  static constexpr int BB = NBB;
  float4 sA[4][BB + 1];                 // mx, my, A, B            (per-strip compacted; +1: prefetch)
  float4 sB[4][BB + 1];                 // C, opacity, r, g
  float2 sC[4][BB + 1];                 // b, bits(batch index j)
  float4 sD[4][PAIR ? BB + 1 : 1];      // partner r, g, b, -
  float sRed[4][BB][9];                        // per-wave totals of the 9 partials, by batch index (36 B stride: odd
                                               // word count, so both the 9-lane write and the per-entry read are conflict-free)
  uint32_t sSlot[BB];                          // by batch index: the entry's record slot in the Gaussian-major scratch
  uint8_t sQuads[BB];                          // by batch index: which quads staged the entry (= whose totals the combine adds up)
  uint32_t cnt[4][4];                          // [staging wave][strip]
  int sQuadLast[4];                            // per wave: the deepest list position any of its 64 pixels used
};

struct BwdPartner { const float4* rec; const float* bg; const float* dL_dcolor; };

// COL = false (plain passes only): the caller wants no colour gradient -- six sums per entry instead of nine (15 instead of 24
// VALU issues of reduction, four multiplies less) and 24-byte records.
// The no-colour reduction is the six-value one (gsr_wave_sum6_packed; its build must stay within the launch's register budget: round 2's
// "six values are slower than nine with zeros" was a 97th VGPR, see render_bwd_persistent's launch bounds).
#define GSR_NOCOL_REDUCE { const float z = gsr_wave_sum6_packed(tx, ty, tx * dx, tx * dy, ty * dy, v5); \
                           if (red6 >= 0) L.sRed[wv][j][red6] = z; }
#define GSR_NOCOL_STORE6(p, e, r0, a, b) gsr_store_partial6(p, e, r0, a, b)   // (36-byte stores instead: measured the same)
// BASE: the ticket's base wave priority (one-view launches: the longest tickets run at base 1 -- the longest lists of a short queue get a
// larger share of their CU and finish with the pack instead of draining alone; see gsr_launch_render_bwd)
template <bool PAIR, int NBB = GSR_BWD_BB(PAIR), bool COL = true, int BASE = 0>
__device__ __forceinline__ void bwd_tile(
    const int tile, const uint2 rg, BwdLdsT<PAIR, NBB>& L, int W, int H, int gx,
    const uint32_t* __restrict__ point_list, const float4* __restrict__ rec, const float* __restrict__ bg,
    const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
    float4* __restrict__ partials, const uint8_t* __restrict__ contrib, const uint8_t* __restrict__ used, const BwdPartner pt) {
  constexpr int BB = NBB;
  constexpr bool ROWS_PERM = !PAIR && NBB == BWD_BATCH;   // the short-queue build: see gsr_rows_sum
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx = tile % gx, ty = tile / gx;
  const int tx0 = tx * GSR_TILE, ty0 = ty * GSR_TILE;
  const int px = tx0 + GSR_QW * (wv & 1) + (lane & 7), py = ty0 + GSR_QH * (wv >> 1) + (lane >> 3);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const int n = (int)(rg.y - rg.x);
  if (n == 0) return;  // uniform: nothing to differentiate in an empty tile
  if (BASE) __builtin_amdgcn_s_setprio(BASE);
  const size_t N = (size_t)H * W;
  const int pix = py * W + px;

  const float T_final = inside ? final_T[pix] : 0.f;
  const int last = inside ? (int)n_contrib[pix] : 0;
  float dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
  if (inside) { dL0 = dL_dcolor[pix]; dL1 = dL_dcolor[N + pix]; dL2 = dL_dcolor[2 * N + pix]; }
  const float nTfbg = -T_final * (bg[0] * dL0 + bg[1] * dL1 + bg[2] * dL2);
  float T = T_final;
  // Colour accumulated behind the current entry, only ever used dotted with this pixel's dL/dcolour: kept as that dot
  // product (acc_dot), with the previous entry's colour . dL (last_cdot) pending -- 6 VALU per entry instead of 12.
  float acc_dot = 0.f, last_cdot = 0.f, last_alpha = 0.f;
  // PAIR: the partner's dL/dcolour and its own dot-product state (the two views' dL/dalpha are needed separately for the
  // per-view screen-space gradients, and together for everything else)
  float dL3 = 0.f, dL4 = 0.f, dL5 = 0.f, nTfbg2 = 0.f, acc_dot2 = 0.f, last_cdot2 = 0.f;
  if (PAIR) {
    if (inside) { dL3 = pt.dL_dcolor[pix]; dL4 = pt.dL_dcolor[N + pix]; dL5 = pt.dL_dcolor[2 * N + pix]; }
    nTfbg2 = -T_final * (pt.bg[0] * dL3 + pt.bg[1] * dL4 + pt.bg[2] * dL5);
  }

  GSR_T0();
  {  // per-quad and per-tile maxima of `last`: a wave reduction, then four words in LDS
    int wl = last;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) wl = max(wl, __shfl_xor(wl, m, 64));
    if (lane == 0) L.sQuadLast[wv] = wl;
  }
  __syncthreads();
  const int ql0 = L.sQuadLast[0], ql1 = L.sQuadLast[1], ql2 = L.sQuadLast[2], ql3 = L.sQuadLast[3];
  const int max_last = max(max(ql0, ql1), max(ql2, ql3));  // entries [max_last, n) are used by no pixel of this tile
  // The list entries of the zero-fill (entries nobody reached still own a record in the Gaussian-major scratch) and of the
  // first two batches are requested together, then their records: two dependent memory trips instead of four.
  const int kz = max_last + tid;
  uint32_t gz = 0;
  if (kz < n) gz = point_list[rg.x + kz];
  uint32_t ng = 0, ng_ahead = 0;                     // ng_ahead: list entry of batch b+2 (see fwd_tile)
  uint32_t nquads = 0;                               // the entry's contribution byte (which quads blended it in the forward)
  if (tid < BB && tid < max_last) {
    ng = point_list[rg.x + (max_last - 1 - tid)];
    nquads = contrib ? contrib[rg.x + (max_last - 1 - tid)] : 0xfu;
    if (tid + BB < max_last) ng_ahead = point_list[rg.x + (max_last - 1 - (tid + BB))];
  }
  const int red6 = lane >= 48 ? gsr_sum6_slot(lane) : -1;   // !COL (GSR_NOCOL_SUM6): where this lane's total of the six-value reduction goes
  (void)red6;

  // Software-pipelined staging (see fwd_tile): batch b+1 is fetched while batch b is processed.
  float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
  float nblue = 0.f;
  float4 nslot = na;   // record word 3: rect bits, offsets[g]
  float4 np = na;      // PAIR: partner colour
  uint2 nbox = make_uint2(1u, 1u);
  float4 slz = na;
  // (round 4) a Gaussian no pixel of the view blended has no record anybody reads (the per-Gaussian backward skips it: GeomState::used):
  // no zeros for it -- 97 % of the unreached entries
  bool fill0 = kz < n;
  if (fill0 && used) fill0 = used[gz] != 0;
  if (fill0) slz = rec[GSR_REC_F4 * gz + 3];   // rect bits, offsets[g]
  if (tid < BB && tid < max_last) {
    { const float4 t2 = rec[GSR_REC_F4 * ng + 2]; na = rec[GSR_REC_F4 * ng]; nb = rec[GSR_REC_F4 * ng + 1]; nblue = t2.x;
      nslot = rec[GSR_REC_F4 * ng + 3];
      nbox = make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)); }
    if (PAIR) { const float4 q1 = pt.rec[GSR_REC_F4 * ng + 1]; np = make_float4(q1.z, q1.w, pt.rec[GSR_REC_F4 * ng + 2].x, 0.f); }
  }
  // zero-fill of the unreached entries: the first 256 rode along with the loads above, the rest (rare) in a plain loop
  for (int k = kz; k < n; k += GSR_BLOCK) {
    float4 sl = slz;
    if (k != kz) {
      const uint32_t g2 = point_list[rg.x + k];
      if (used && used[g2] == 0) continue;
      sl = rec[GSR_REC_F4 * g2 + 3];
    } else if (!fill0) continue;
    const uint32_t rx = __float_as_uint(sl.x), ry = __float_as_uint(sl.y);
    const uint32_t minx = rx & 0xffffu, miny = rx >> 16, maxx = ry & 0xffffu, maxy = ry >> 16;
    const uint32_t e = __float_as_uint(sl.z) + gsr_tile_rank(__float_as_uint(sl.w), (maxx - minx) * (maxy - miny),
                                                            ((uint32_t)ty - miny) * (maxx - minx) + ((uint32_t)tx - minx));
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PAIR || COL) gsr_store_partial(partials, e, z, z, 0.f);
    else GSR_NOCOL_STORE6(partials, e, z, 0.f, 0.f);
  }
  GSR_TP(0);
  for (int base = 0; base < max_last; base += BB) {
    // batch entry j (0 = deepest still unprocessed) is list position pos = max_last - 1 - (base + j)
    const int m_all = min(BB, max_last - base);
    const float4 a = na, b = nb, d = np;
    const float2 c = make_float2(nblue, __uint_as_float((uint32_t)tid));
    uint32_t mask = 0;
    if (tid < m_all) {
      {  // slot of this (Gaussian, tile) pair: offsets[g] + row-major rank of the tile inside the Gaussian's rect
        const uint32_t rx = __float_as_uint(nslot.x), ry = __float_as_uint(nslot.y);
        const uint32_t minx = rx & 0xffffu, miny = rx >> 16, maxx = ry & 0xffffu, maxy = ry >> 16;
        L.sSlot[tid] = __float_as_uint(nslot.z) + gsr_tile_rank(__float_as_uint(nslot.w), (maxx - minx) * (maxy - miny),
                                                                ((uint32_t)ty - miny) * (maxx - minx) + ((uint32_t)tx - minx));
      }
      mask = nquads;          // exactly the quads with a pixel that blended this entry (the forward's contribution byte; a forward that did
      L.sQuads[tid] = (uint8_t)mask;   // not track hands 0xf: every quad then replays the entry and the per-pixel hit test keeps the result exact)
    }
    {
      const int nj = base + BB + tid;
      if (tid < BB && nj < max_last) {
        ng = ng_ahead;
        nquads = contrib ? contrib[rg.x + (max_last - 1 - nj)] : 0xfu;
        if (nj + BB < max_last) ng_ahead = point_list[rg.x + (max_last - 1 - (nj + BB))];
        { const float4 t2 = rec[GSR_REC_F4 * ng + 2]; na = rec[GSR_REC_F4 * ng]; nb = rec[GSR_REC_F4 * ng + 1]; nblue = t2.x;
      nslot = rec[GSR_REC_F4 * ng + 3];
      nbox = make_uint2(__float_as_uint(t2.z), __float_as_uint(t2.w)); }
        if (PAIR) { const float4 q1 = pt.rec[GSR_REC_F4 * ng + 1]; np = make_float4(q1.z, q1.w, pt.rec[GSR_REC_F4 * ng + 2].x, 0.f); }
      }
    }
    uint64_t bal[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) bal[w] = __ballot((mask >> w) & 1u);
    if (lane == 0) {
#pragma unroll
      for (int w = 0; w < 4; ++w) L.cnt[wv][w] = (uint32_t)__popcll(bal[w]);
    }
    GSR_TP(1);
    __syncthreads();
    GSR_TP(2);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if ((mask >> w) & 1u) {
        uint32_t p = (uint32_t)__popcll(bal[w] & gsr_lanemask_lt());
        for (int v = 0; v < wv; ++v) p += L.cnt[v][w];
        L.sA[w][p] = make_float4(a.x, a.y, GSR_STAGE_A(a.z), GSR_STAGE_B(a.w));   // (-A/2, -B): exact factors, see gsr_power
        L.sB[w][p] = make_float4(GSR_STAGE_A(b.x), b.y, b.z, b.w);               // -C/2
        L.sC[w][p] = c;
        if (PAIR) L.sD[w][p] = d;
      }
    }
#ifdef GSR_ABL_NOVISIT   // ablation build (tools/r05_ablation.sh): the replay loop walks nothing -- what is left is staging, barriers, combine and stores
    const int m = 0;
#else
    const int m = __builtin_amdgcn_readfirstlane((int)(L.cnt[0][wv] + L.cnt[1][wv]));
#endif
    __syncthreads();
    GSR_TP(3);
    const float4* __restrict__ wA = L.sA[wv];
    const float4* __restrict__ wB = L.sB[wv];
    const float2* __restrict__ wC = L.sC[wv];
    const float4* __restrict__ wD = L.sD[wv];
    // One list entry: re-evaluate alpha; when some pixel of the quad used the entry, back out T, form the nine partials,
    // reduce them over the wave and park the totals.
#ifndef GSR_PRIO_VISIT
#define GSR_PRIO_VISIT 1      /* wave priority (s_setprio) of the replay loop; 0 = leave it alone.  Measured, 8 views, one box, 3 rounds each:
                                 0: 441.7 us, 1 / 2 / 3: 433.6 / 431.9 / 431.9 us per launch; raising the staging phase as well or instead: 437.8 / 448 */
#endif
#define GSR_BWD_ENTRY(ea, eb, ec, ed)                                                                           \
    {                                                                                                         \
      const int j = __builtin_amdgcn_readfirstlane((int)__float_as_uint(ec.y)); /* batch index, wave-uniform */ \
      const int pos = max_last - 1 - (base + j);                                                              \
      const float blue = ec.x;                                                                                \
      const float dx = ea.x - pxf, dy = ea.y - pyf;                                                           \
      const float power = gsr_power(ea.z, ea.w, eb.x, dx, dy);                                                \
      const float G0 = GSR_EXP_OF_POWER(power);                                                               \
      const bool hit = (pos < last) && power <= 0.0f && eb.y * G0 >= GSR_ALPHA_MIN; /* = min(0.99, .) >= 1/255 */ \
      { /* every staged visit contributes (exact lists) */                                                    \
        /* No exec-masked region: a lane that does not use the entry runs the same arithmetic with G = 0.  Then  \
           alpha = 0, 1/(1-alpha) = 1, T and the accumulated colour are unchanged (an alpha = 0 entry only flushes \
           the pending (last_alpha, last colour) pair, which the next real entry would have done with the same    \
           operands), and all nine partials come out as exact zeros because each carries a factor G or alpha.    \
           Saves the 9 zero-moves, the state copies and the exec save/restore of the masked form (VALU-bound). */ \
        const float G = hit ? G0 : 0.0f;                                                                      \
        const float alpha = fminf(GSR_ALPHA_MAX, eb.y * G);                                                   \
        const float rcp = __builtin_amdgcn_rcpf(1.0f - alpha);                                                \
        T = T * rcp;                                                                                          \
        acc_dot = __builtin_fmaf(last_alpha, last_cdot - acc_dot, acc_dot);                                   \
        const float cdot = __builtin_fmaf(blue, dL2, __builtin_fmaf(eb.w, dL1, eb.z * dL0));                  \
        last_cdot = cdot;                                                                                     \
        float dL_dalpha = cdot - acc_dot;                                                                     \
        dL_dalpha = __builtin_fmaf(dL_dalpha, T, nTfbg * rcp);                                                \
        if (PAIR) {                                                                                           \
          /* the partner's dL/dalpha from its own colour recurrence; t_A (this view alone) and t (both) */     \
          acc_dot2 = __builtin_fmaf(last_alpha, last_cdot2 - acc_dot2, acc_dot2);                             \
          const float cdot2 = __builtin_fmaf(ed.z, dL5, __builtin_fmaf(ed.y, dL4, ed.x * dL3));               \
          last_cdot2 = cdot2;                                                                                 \
          last_alpha = alpha;                                                                                 \
          const float dL_dalpha2 = __builtin_fmaf(cdot2 - acc_dot2, T, nTfbg2 * rcp);                         \
          const float u1 = G * dL_dalpha;                                                                     \
          const float v5 = __builtin_fmaf(G, dL_dalpha2, u1);                                                 \
          const float tA = eb.y * u1, t = eb.y * v5;                                                          \
          const float tx = t * dx, ty = t * dy;                                                               \
          const float z = gsr_wave_sum8_packed(tx, ty, tx * dx, tx * dy, ty * dy, v5, tA * dx, tA * dy);      \
          if (lane >= 48 && lane <= 55) L.sRed[wv][j][lane - 48] = z; /* one ds_write_b32 */                  \
        } else if (!COL) {                                                                                    \
          last_alpha = alpha;                                                                                 \
          const float v5 = G * dL_dalpha;                                                                     \
          const float t = eb.y * v5;                                                                          \
          const float tx = t * dx, ty = t * dy;                                                               \
          GSR_NOCOL_REDUCE                                                                                    \
        } else {                                                                                              \
        last_alpha = alpha;                                                                                   \
        const float w = alpha * T;                                                                            \
        /* t = dL/dG * G with dL/dG = opacity * dL/dalpha (min(0.99, .) is straight-through).  The conic partials are     \
           stored WITHOUT their constant factors (-1/2, -1, -1/2: preprocess_bwd applies them to the per-Gaussian sums). */ \
        const float v5 = G * dL_dalpha;                                                                       \
        const float t = eb.y * v5;                                                                            \
        const float tx = t * dx, ty = t * dy;                                                                 \
        /* mean2D: d/dmx = -(A tx + B ty), d/dmy = -(C ty + B tx) is linear in (tx, ty) with per-Gaussian constants, so    \
           the raw sums of tx and ty travel and preprocess_bwd applies the conic once per Gaussian */                    \
        const float v0 = tx, v1 = ty;                                                                         \
        const float v2 = tx * dx, v3 = tx * dy, v4 = ty * dy;                                                 \
        const float v6 = w * dL0, v7 = w * dL1, v8 = w * dL2;                                                 \
        GSR_BWD_PARK9(v0, v1, v2, v3, v4, v5, v6, v7, v8)                                                     \
        }                                                                                                     \
      }                                                                                                       \
    }
    // two entries per trip on ping-pong registers: the record of entry j+1 (j+2) is fetched from LDS while entry j (j+1) is
    // evaluated, no copies to rotate the prefetch (unrolled blocks as in fwd_tile measured 3 % slower here)
    if (GSR_PRIO_VISIT) __builtin_amdgcn_s_setprio(GSR_PRIO_VISIT + BASE);
    float4 ea = wA[0], eb = wB[0];
    float2 ec = wC[0];
    float4 ed = wD[0];
    int jj = 0;
    for (; jj + 1 < m; jj += 2) {
      const float4 xa = wA[jj + 1], xb = wB[jj + 1];
      const float2 xc = wC[jj + 1];
      const float4 xd = PAIR ? wD[jj + 1] : ed;
      GSR_BWD_ENTRY(ea, eb, ec, ed)
      ea = wA[jj + 2]; eb = wB[jj + 2]; ec = wC[jj + 2];
      if (PAIR) ed = wD[jj + 2];
      GSR_BWD_ENTRY(xa, xb, xc, xd)
    }
    if (jj < m) GSR_BWD_ENTRY(ea, eb, ec, ed)
#undef GSR_BWD_ENTRY
    if (GSR_PRIO_VISIT) __builtin_amdgcn_s_setprio(BASE);
    GSR_TP(4);
    __syncthreads();
    GSR_TP(5);
    if (tid < m_all) {
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
      const uint32_t quads = (uint32_t)L.sQuads[tid];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if ((quads >> w) & 1u) {
          const float* q = L.sRed[w][tid];
          r0.x += q[0]; r0.y += q[1]; r0.z += q[2]; r0.w += q[3];
          r1.x += q[4]; r1.y += q[5];
          if (PAIR || COL) { r1.z += q[6]; r1.w += q[7]; }
          if (!PAIR && COL) r2.x += q[8];
        }
      }
      const uint32_t e = L.sSlot[tid];
      if (PAIR || COL) gsr_store_partial(partials, e, r0, r1, r2.x);
      else GSR_NOCOL_STORE6(partials, e, r0, r1.x, r1.y);
    }
    GSR_TP(6);
    __syncthreads();
    GSR_TP(7);
  }
  if (BASE) __builtin_amdgcn_s_setprio(0);
  GSR_TFLUSH_B();
}

// ------------------------------------------------------------------------------------------ kernels
// One launch serves every view of the call: a ticket of the combined LPT order is {tile, list start, list end,
// view}; the view's pointers come from the kernarg table (uniform index: scalar loads).
#define GSR_FWD_PASS(vw) tab.W, tab.H, tab.gx, (vw).point_list, (vw).rec, (vw).bg, (vw).final_T, (vw).n_contrib, (vw).out_color, (vw).out_depth
// the per-Gaussian used flags are trusted by the backward only when the view's `tracked` word says a tracking forward wrote them
#define GSR_FWD_MARK_TRACKED() if (TRACK) { if (threadIdx.x < (unsigned)tab.V && tab.v[threadIdx.x].tracked) *tab.v[threadIdx.x].tracked = 1u; }
#define GSR_BWD_PASS(vw) \
  tab.W, tab.H, tab.gx, (vw).point_list, (vw).rec, (vw).bg, (vw).final_T, (vw).n_contrib, (vw).dL_dcolor, (vw).partials,                 \
  /* a forward that did not track (GSR_FORWARD_ONLY -- against the contract, but cheap to survive) left neither contribution bytes nor   \
     used flags: every quad then stages every entry below the tile's deepest contributor (the per-pixel hit test keeps the result exact) */ \
  (!(vw).tracked || *(vw).tracked != 0u) ? (vw).contrib : nullptr,                                                                        \
  ((vw).used && (vw).tracked && *(vw).tracked != 0u) ? (vw).used : nullptr

__device__ __forceinline__ FwdPartner fwd_partner(const GsrRenderView& p) {
  return FwdPartner{p.rec, p.bg, p.final_T, p.n_contrib, p.out_color, p.out_depth, p.colors, p.used};
}

// PAIRS: the call holds fused pairs (GsrRenderView::partner): tickets of such views blend both; the other tickets take the
// plain path.  A call without pairs runs the PAIRS = false build (smaller LDS footprint: one more workgroup per CU).
static_assert(sizeof(FwdLdsT<true>) >= sizeof(FwdLdsT<false>), "the pair build's LDS must hold the plain layout too");
// Persistent forward.  Empty tiles never enter the queue: the workgroups first paint their background
// (grid-stride over the tail of the order array), then pop busy tiles longest-first.
// Launch bounds = the residency the launch asks for (six workgroups per CU, five for the pair build): without them the allocator took
// 112 - 121 VGPRs for the tracking build (four waves per SIMD) where the round-3 code happened to land on 75 / 94.
template <bool PAIRS, bool TRACK = false>
__global__ __launch_bounds__(GSR_BLOCK, TRACK ? (PAIRS ? 5 : FWD_TRACK_WAVES) : FWD_WAVES_PER_EU) void render_fwd_persistent(GsrRenderViews tab) {
  __shared__ FwdLdsT<PAIRS> L;
  __shared__ uint32_t s_ticket;
  const uint4* __restrict__ tile_order = tab.order;
  uint32_t* __restrict__ queue = tab.queue;
  const uint32_t n_busy = queue[4], n_all = queue[7];   // n_all < V * T when fused pairs took their partners' busy tiles along
  // Tickets: the first one is implicit (blockIdx.x, no atomic: no thundering herd at kernel start); later ones
  // are gridDim.x + atomicAdd(head), popped after the tile.  (Popping the next ticket one tile ahead -- even from
  // wave 3, which stages nothing, so the returning atomic blocks no gather wait -- was measured 30 % slower: a
  // workgroup then commits to its next tile while it is still busy, and the tail of the launch loses its balance.)
  {  // background of the empty tiles
    const int W = tab.W, H = tab.H, gx = tab.gx;
    const size_t N = (size_t)H * W;
    for (uint32_t i = n_busy + blockIdx.x; i < n_all; i += gridDim.x) {
      const uint4 ord = tile_order[i];
      const GsrRenderView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];  // uniform: scalar loads
      const int tile = (int)ord.x;
      const int px = (tile % gx) * GSR_TILE + (threadIdx.x & 15), py = (tile / gx) * GSR_TILE + (threadIdx.x >> 4);
      if (!TRACK && vw.cut_out && threadIdx.x == 0) {   // an empty tile proposes no cut; one that was CUT empty cannot vouch for its background
        vw.cut_out[tile] = 0x7f800000u;
        if (vw.cut_in && vw.cut_in[tile] != 0x7f800000u && vw.redo) atomicAdd(vw.redo, 1u);
      }
      if (px < W && py < H) {
        const int pix = py * W + px;
        vw.final_T[pix] = 1.0f; vw.n_contrib[pix] = 0u;
        vw.out_color[pix] = vw.bg[0]; vw.out_color[N + pix] = vw.bg[1]; vw.out_color[2 * N + pix] = vw.bg[2];
        vw.out_depth[pix] = 0.0f;
        if (PAIRS && vw.partner >= 0) {   // the partner's tile is empty too (same lists): it has no order entry of its own
          const GsrRenderView& pw = tab.v[vw.partner];
          pw.final_T[pix] = 1.0f; pw.n_contrib[pix] = 0u;
          pw.out_color[pix] = pw.bg[0]; pw.out_color[N + pix] = pw.bg[1]; pw.out_color[2 * N + pix] = pw.bg[2];
          pw.out_depth[pix] = 0.0f;
        }
      }
    }
  }
  if (blockIdx.x == 0) GSR_FWD_MARK_TRACKED()
  uint32_t ticket = blockIdx.x;
  while (ticket < n_busy) {
    const uint4 ord = tile_order[ticket];
    const GsrRenderView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];  // uniform: scalar loads
    int pend;
    uint32_t pend_g;
    // (Round 6: the longest 1/16 ... 1/2 of the tickets at wave priority 2 changed nothing at one, two or four views -- 49.5 / 66.8 / 109.5 us
    //  either way, profiles/r06_v1_experiments.txt: a tile's walk is a dependent chain, not starved of issue slots.)
    if (PAIRS && vw.partner >= 0)
      pend = fwd_tile<PAIRS, TRACK>((int)ord.x, make_uint2(ord.y, ord.z), L, GSR_FWD_PASS(vw), fwd_partner(tab.v[vw.partner]), vw.contrib, vw.used, pend_g);
    else
      pend = fwd_tile<false, TRACK>((int)ord.x, make_uint2(ord.y, ord.z), reinterpret_cast<FwdLdsT<false>&>(L), GSR_FWD_PASS(vw), FwdPartner{}, vw.contrib, vw.used, pend_g,
                                    TRACK ? nullptr : vw.cut_in, TRACK ? nullptr : vw.cut_out, TRACK ? nullptr : vw.redo, vw.cut_margin);
#ifdef GSR_TILE_TIMING
    const unsigned long long tq0 = __builtin_readcyclecounter();
#endif
    if (threadIdx.x == 0) s_ticket = gridDim.x + atomicAdd(&queue[0], 1u);
    __syncthreads();
    ticket = s_ticket;
    if (TRACK) {     // the tile's last contribution bytes: every wave's masks are in LDS behind the barrier above, and the one below keeps the next tile's staging off them
      if (PAIRS && vw.partner >= 0) fwd_store_contrib<PAIRS>(L, vw.contrib, vw.used, tab.v[vw.partner].used, ord.y, (int)(ord.z - ord.y), pend, pend_g);
      else fwd_store_contrib<false>(reinterpret_cast<FwdLdsT<false>&>(L), vw.contrib, vw.used, nullptr, ord.y, (int)(ord.z - ord.y), pend, pend_g);
    }
    __syncthreads();  // every wave has its copy before thread 0 overwrites the slot
#ifdef GSR_TILE_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_fwd_timing[7], __builtin_readcyclecounter() - tq0);
#endif
  }
}

__device__ __forceinline__ BwdPartner bwd_partner(const GsrRenderView& p) { return BwdPartner{p.rec, p.bg, p.dL_dcolor}; }

// LDS of a backward workgroup: the pair build also runs plain tickets (views without a partner), whose layout is the larger one
template <bool PAIRS, int NBB = BWD_BATCH>   // NBB: batch size of the plain tickets
struct BwdLdsAny {
  static constexpr size_t BYTES = PAIRS && sizeof(BwdLdsT<true>) > sizeof(BwdLdsT<false, NBB>) ? sizeof(BwdLdsT<true>) : sizeof(BwdLdsT<false, NBB>);
  alignas(16) unsigned char raw[BYTES];
};

#ifdef GSR_TRACE_TICKETS
// Debug build (tools/ticket_trace.py): per backward ticket {start, end (s_memrealtime, 100 MHz), workgroup, list length}
#define GSR_TRACE_MAX 65536
__device__ uint4 g_ticket_trace[GSR_TRACE_MAX];
#endif
// The 96-entry build is launched five workgroups per CU: it must stay within 96 VGPRs (five waves per SIMD).  The six-value reduction
// of the no-colour-gradient build came out at 97 -- one register over, four waves per SIMD, the fifth workgroup of every CU waiting for
// a slot: that was round 2's "six-value reduction is 17 % slower than nine values with zeros" (the nine-value build happened to need 96).
template <bool PAIRS, int NBB = BWD_BATCH, bool COL = true>
__global__ __launch_bounds__(GSR_BLOCK, (!PAIRS && NBB == BWD_SMALL_BB) ? BWD_SMALL_WAVES : BWD_WAVES_PER_EU) void render_bwd_persistent(GsrRenderViews tab) {
  __shared__ BwdLdsAny<PAIRS, NBB> L;
  __shared__ uint32_t s_ticket;
  const uint4* __restrict__ tile_order = tab.order;
  uint32_t* __restrict__ queue = tab.queue;
  const uint32_t n_busy = queue[4];  // empty tiles have nothing to differentiate: they never enter the queue
  // First ticket implicit (blockIdx.x); later ones popped after the tile.  (No ticket prefetch here: every wave
  // of bwd_tile issues global loads right at tile start, and they would queue behind the in-flight atomic.)
  uint32_t ticket = blockIdx.x;
  while (ticket < n_busy) {
#ifdef GSR_TRACE_TICKETS
    const unsigned long long tr0 = wall_clock64();
#endif
    const uint4 ord = tile_order[ticket];
    const GsrRenderView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];  // uniform: scalar loads
    if (PAIRS && vw.partner >= 0)
      bwd_tile<PAIRS>((int)ord.x, make_uint2(ord.y, ord.z), reinterpret_cast<BwdLdsT<PAIRS>&>(L), GSR_BWD_PASS(vw), bwd_partner(tab.v[vw.partner]));
    else if (ticket * 256u < n_busy * (uint32_t)tab.prio_frac256)
      bwd_tile<false, NBB, COL, 1>((int)ord.x, make_uint2(ord.y, ord.z), reinterpret_cast<BwdLdsT<false, NBB>&>(L), GSR_BWD_PASS(vw), BwdPartner{});
    else
      bwd_tile<false, NBB, COL>((int)ord.x, make_uint2(ord.y, ord.z), reinterpret_cast<BwdLdsT<false, NBB>&>(L), GSR_BWD_PASS(vw), BwdPartner{});
#ifdef GSR_TRACE_TICKETS
    if (threadIdx.x == 0 && ticket < GSR_TRACE_MAX)
      g_ticket_trace[ticket] = make_uint4((uint32_t)tr0, (uint32_t)wall_clock64(), blockIdx.x, ord.z - ord.y);
#endif
    if (threadIdx.x == 0) s_ticket = gridDim.x + atomicAdd(&queue[1], 1u);
    __syncthreads();  // also: the tile's LDS (incl. sQuadLast) is dead before the next tile reuses it
    ticket = s_ticket;
    __syncthreads();
  }
  // The last workgroup to leave re-arms the queue for a possible second backward over the same state
  // (retain_graph): no workgroup can still be popping once all of them have checked out.  Saves a memset launch.
  if (threadIdx.x == 0 && atomicAdd(&queue[5], 1u) == gridDim.x - 1u) {
    atomicExch(&queue[1], 0u);
    atomicExch(&queue[5], 0u);
  }
}


// ------------------------------------------------------------------------------------------ backward, producer / consumer form (round 5)
// What round 5 measured first (profiles/r05_ablation.txt, r05_visit_peak.txt; 4 views): render_bwd 213 us; with the replay loop walking
// NOTHING 94 us; the visit body alone (LDS-resident entries, no staging, no barriers) peaks at 98 - 104 ns per visit per SIMD with 6 - 8
// waves, i.e. ~140 us for this launch's 1.39 M visits.  The staging / barrier / combine skeleton of bwd_tile and the four-barriers-per-batch
// lockstep of unequal quad lists cost the difference.
//
// Here the roles are split between the five waves of a 320-thread workgroup:
//   waves 0 - 3  CONSUMERS: wave q replays quad q of every tile the workgroup works on; it only ever waits for a published batch, and
//                           loads the next tile's per-pixel values while it still replays the current one.  The consumer that arrives LAST
//                           at the end of a batch (an LDS counter) adds the per-quad totals of the batch's entries and stores their records.
//   wave 4       STAGER:    pops the tickets, walks a tile's list back to front in chunks of 64 positions (the next chunk's list words are
//                           in flight while this one's records are gathered), keeps the entries some quad blended (the forward's
//                           contribution bytes: the useful-entry compaction of item 1a), gathers their records ONCE per tile, appends them
//                           + four per-quad offset lists to a ring of batches in LDS, and zero-fills the records of unreached entries of
//                           used Gaussians while the next ticket's atomic is in flight
// Hand-offs are words in LDS polled with s_sleep: no workgroup barrier anywhere.  A consumer runs up to PC_R batches ahead of the slowest
// quad, across tile boundaries, so quad imbalance averages over the ring instead of being paid per batch.  24.9 KB of LDS and 5 waves per
// workgroup: SIX workgroups per CU = 24 replaying waves.  Per visit the arithmetic and its order are bwd_tile's, and so is the order in
// which the combine adds the quads: the records are bit-identical to the barrier form's (tools/r05_pc_check.sh).
#ifndef PC_BB
#define PC_BB 40          // entries per batch
#endif
#ifndef PC_R
#define PC_R 3            // batches in the ring
#endif
#define PC_WAVES 5
#define PC_THREADS (64 * PC_WAVES)
#define PC_ENT_BYTES 48
#define PC_FLAG_FIRST 1u
#define PC_FLAG_EXIT 2u
#ifndef PC_CONS_PRIO
#define PC_CONS_PRIO 1
#endif
#ifndef PC_POP_EARLY
#define PC_POP_EARLY 0
#endif
#ifndef PC_POLL_SLEEP
#define PC_POLL_SLEEP 2   // x 64 clocks between two polls of a hand-off word
#endif
static_assert(PC_BB >= 32 && PC_BB <= 64, "a half chunk (32 entries) must fit one batch; the combine is one lane per entry");
struct alignas(16) PcSlot {
  float4 ent[PC_BB][3];              // {mx, my, A', B'} {C', opacity, r, g} {b, bits(list position), bits(byte offset of the entry's totals: 36 j), 0}
  float red[4][PC_BB][9];            // per quad and batch index: the nine wave totals
  uint32_t slot[PC_BB];              // record slot in the Gaussian-major scratch
  uint16_t idx[4][PC_BB + 2];        // per quad: byte offsets (48 j) of its entries in blend-replay order; two zero sentinels behind the list
  uint8_t quads[PC_BB];              // which quads hold totals for the entry
};
struct alignas(16) PcLds {
  PcSlot s[PC_R];
  uint32_t cnt[PC_R][4];             // batch header: entries per quad list,
  uint32_t n[PC_R];                  //   entries in the batch,
  uint32_t flags[PC_R];              //   PC_FLAG_*,
  uint32_t tile[PC_R];               //   tile id and
  uint32_t view[PC_R];               //   view index of the batch's tile
  // ---- control words (zeroed at kernel start: prod_seq .. ctl_end)
  uint32_t prod_seq;                 // batches published (= number of the open batch)
  uint32_t arrive[PC_R];             // consumers through the batch in this slot: the fourth one combines it
  uint32_t retired[PC_R];            // number + 1 of the last batch retired from this slot (its records are on their way: the slot is free)
  uint32_t abort;                    // a wait timed out (protocol bug): everybody leaves
  uint32_t ctl_end;
};

__device__ uint32_t g_pc_error;     // != 0: some wait of the producer / consumer backward ran out of patience (see pc_wait_gt); read by gsr_debug_pc_error
__device__ __forceinline__ uint32_t pc_peek(const uint32_t* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// Until *p > v (an LDS word written by another wave of the workgroup).  A poll costs the SIMD a handful of issue slots: the wave sleeps
// PC_POLL_SLEEP x 64 clocks between polls, at priority 0, so that waiting waves stay out of the replaying waves' way.  BOUNDED: a wait that
// outlasts ~2^19 polls (tens of milliseconds; a batch takes microseconds) can only be a protocol bug -- it raises the workgroup's abort
// word, which ends every other wait at once, and the device error word; the kernel then drains instead of hanging the GPU.  -> false: aborted
template <int PRIO_AFTER>
__device__ __forceinline__ bool pc_wait_gt(const uint32_t* p, uint32_t v, uint32_t* abort_word) {
  if (pc_peek(p) > v) { asm volatile("" ::: "memory"); return true; }
  if (PRIO_AFTER) __builtin_amdgcn_s_setprio(0);
  uint32_t spins = 0;
  bool ok = true;
  while (pc_peek(p) <= v) {
    __builtin_amdgcn_s_sleep(PC_POLL_SLEEP);
    if ((++spins & 255u) == 0u) {
      if (pc_peek(abort_word) != 0u) { ok = false; break; }
      if (spins >= (1u << 19)) {
        if (gsr_lane() == 0) { __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); atomicExch(&g_pc_error, 1u); }
        ok = false; break;
      }
    }
  }
  if (PRIO_AFTER) __builtin_amdgcn_s_setprio(PRIO_AFTER);
  asm volatile("" ::: "memory");
  return ok;
}
// LDS operations of one wave are performed in order: once this wave's earlier DS operations are done (lgkmcnt 0), a flag store is seen
// by the other waves after them
__device__ __forceinline__ void pc_publish(uint32_t* p, uint32_t v) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (gsr_lane() == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#ifdef GSR_TILE_TIMING
__device__ unsigned long long g_pc_start[2048];
// phase clocks of the producer / consumer backward (debug build, wall_clock64: the constant 100 MHz counter -- sums in units of 10 ns):
// g_bwd_timing[0..3] consumer 0 {waiting for a batch, tile head, visits, combining}, [4..8] stager {waiting for a ring slot, tile start chain,
// chunk loads + gathers, appending, ticket + zero fill}.  Accumulated in registers, flushed once per wave at its exit (an atomic per
// phase clock would queue in front of every later load).
#define PC_T0() unsigned long long _pt = wall_clock64(), _pa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PC_TP(i) do { const unsigned long long _n = wall_clock64(); _pa[i] += _n - _pt; _pt = _n; } while (0)
#define PC_TFLUSH() do { if (gsr_lane() == 0) for (int _i = 0; _i < 9; ++_i) if (_pa[_i]) atomicAdd(&g_bwd_timing[_i], _pa[_i]); } while (0)
#else
#define PC_T0() do {} while (0)
#define PC_TP(i) do {} while (0)
#define PC_TFLUSH() do {} while (0)
#endif

__device__ __forceinline__ uint32_t pc_slot_of(const float4 w3, int tx, int ty) {
  const uint32_t rx = __float_as_uint(w3.x), ry = __float_as_uint(w3.y);
  const uint32_t minx = rx & 0xffffu, miny = rx >> 16, maxx = ry & 0xffffu, maxy = ry >> 16;
  return __float_as_uint(w3.z) + gsr_tile_rank(__float_as_uint(w3.w), (maxx - minx) * (maxy - miny),
                                               ((uint32_t)ty - miny) * (maxx - minx) + ((uint32_t)tx - minx));
}
template <bool COL>
__device__ __forceinline__ void pc_zero_record(float4* __restrict__ partials, uint32_t e) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (COL) gsr_store_partial(partials, e, z, z, 0.f);
  else gsr_store_partial6(partials, e, z, 0.f, 0.f);
}
// what the backward may trust of a view's forward (see GSR_BWD_PASS)
__device__ __forceinline__ const uint8_t* pc_contrib(const GsrRenderView& vw) { return (!vw.tracked || *vw.tracked != 0u) ? vw.contrib : nullptr; }
__device__ __forceinline__ const uint8_t* pc_used(const GsrRenderView& vw) { return (vw.used && vw.tracked && *vw.tracked != 0u) ? vw.used : nullptr; }

template <bool COL>
__device__ __forceinline__ void pc_stager(PcLds& L, const GsrRenderViews& tab) {
  const int lane = gsr_lane();
  const uint4* __restrict__ tile_order = tab.order;
  uint32_t* __restrict__ queue = tab.queue;
  const uint32_t n_busy = queue[4];
  const int W = tab.W, H = tab.H, gx = tab.gx;
  uint32_t seq = 0;                          // number of the open batch (= batches published)
  uint32_t ticket = blockIdx.x;              // the first ticket is implicit (see render_bwd_persistent)
  __builtin_amdgcn_s_setprio(3);             // light and on everybody's critical path (its waits drop to 0)
  PC_T0();
  uint4 ord = make_uint4(0u, 0u, 0u, 0u);
  if (ticket < n_busy) ord = tile_order[ticket];
  while (ticket < n_busy) {
    const int vi = __builtin_amdgcn_readfirstlane((int)ord.w);
    const GsrRenderView& vw = tab.v[vi];
    const int tile = __builtin_amdgcn_readfirstlane((int)ord.x);
    const uint32_t list0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)ord.y);
    const int n = __builtin_amdgcn_readfirstlane((int)(ord.z - ord.y));
    const int tx = tile % gx, ty = tile / gx;
    const uint8_t* __restrict__ contrib = pc_contrib(vw);
    const uint8_t* __restrict__ used = pc_used(vw);
    const uint32_t* __restrict__ point_list = vw.point_list;
    const float4* __restrict__ rec = vw.rec;
    float4* __restrict__ partials = vw.partials;
    // deepest list position any pixel of the tile used (entries behind it have no contribution byte worth reading)
    int max_last;
    {
      int ml = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = lane + 64 * u, px = tx * GSR_TILE + (p & 15), py = ty * GSR_TILE + (p >> 4);
        if (px < W && py < H) ml = max(ml, (int)vw.n_contrib[py * W + px]);
      }
#pragma unroll
      for (int msk = 32; msk >= 1; msk >>= 1) ml = max(ml, __shfl_xor(ml, msk, 64));
      max_last = __builtin_amdgcn_readfirstlane(ml);
    }
    // (Popping the NEXT ticket here, so that the atomic and the order entry it leads to travel while this tile is staged, was measured
    // 15 % SLOWER (render_bwd 208 -> 241 us, 4 views): a workgroup then commits to a tile while it is still busy with another, and the
    // launch's tail loses its balance -- what round 3 saw in the forward.)
    uint32_t next = 0;
#if PC_POP_EARLY
    if (lane == 0) next = gridDim.x + atomicAdd(&queue[1], 1u);
#endif
    PC_TP(5);
    int fill = 0;                            // entries in the open batch
    uint32_t cqa[4] = {0u, 0u, 0u, 0u};      // its per-quad list lengths
    bool first = true, aborted = false;
    auto publish_open = [&](int nent) {      // the open batch (number seq) holds nent entries: header, sentinels, sequence counter
      PcSlot& S = L.s[seq % PC_R];
      const int sl = (int)(seq % PC_R);
      if (lane < 4) {
        const uint32_t cc = lane == 0 ? cqa[0] : lane == 1 ? cqa[1] : lane == 2 ? cqa[2] : cqa[3];
        S.idx[lane][cc] = 0; S.idx[lane][cc + 1] = 0;
        L.cnt[sl][lane] = cc;
      }
      if (lane == 0) { L.n[sl] = (uint32_t)nent; L.flags[sl] = first ? PC_FLAG_FIRST : 0u; L.tile[sl] = (uint32_t)tile; L.view[sl] = (uint32_t)vi; }
      pc_publish(&L.prod_seq, seq + 1u);
      ++seq; first = false;
    };
    // ---- the walked part, back to front, 64 list positions per chunk; the entries some quad blended are appended to the batch stream
    const int C = (max_last + 63) >> 6;
    uint32_t ng = 0, nbyte = 0;              // list word and contribution byte of the NEXT chunk's position (in flight during this chunk's gathers)
    if (C > 0) {
      const int pos0 = max_last - 1 - lane;
      if (pos0 >= 0) { ng = point_list[list0 + pos0]; nbyte = contrib ? (uint32_t)contrib[list0 + pos0] : 0xfu; }
    }
    for (int c = 0; c < C && !aborted; ++c) {
      const int pos = max_last - 1 - 64 * c - lane;
      const bool valid = pos >= 0;
      const uint32_t g = ng, byte = valid ? nbyte : 0u;
      {
        const int npos = pos - 64;
        if (c + 1 < C && npos >= 0) { ng = point_list[list0 + npos]; nbyte = contrib ? (uint32_t)contrib[list0 + npos] : 0xfu; }
      }
      const bool useful = byte != 0u;
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
      bool zf = valid && !useful;            // walked, but blended by no quad: a zero record when the view uses the Gaussian anywhere
      if (valid) r3 = rec[GSR_REC_F4 * g + 3];
      if (zf && used) zf = used[g] != 0;
      if (useful) { r0 = rec[GSR_REC_F4 * g]; r1 = rec[GSR_REC_F4 * g + 1]; r2 = rec[GSR_REC_F4 * g + 2]; }
      const uint32_t e = pc_slot_of(r3, tx, ty);
      if (zf) pc_zero_record<COL>(partials, e);
#ifdef GSR_TILE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      PC_TP(6);
      // appended in two halves of 32 lanes: a half never spans more than two batches
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool u = useful && (lane >> 5) == h;
        const uint64_t bu = __ballot(u);
        const int nu = __popcll(bu);
        if (nu == 0) continue;
        const int tgt = fill + (int)__popcll(bu & gsr_lanemask_lt());
        const bool straddle = fill + nu > PC_BB;
        // the ring slots this half writes must have been retired
        if (seq >= PC_R && !pc_wait_gt<3>(&L.retired[seq % PC_R], seq - PC_R, &L.abort)) { aborted = true; break; }
        if (straddle && seq + 1u >= PC_R && !pc_wait_gt<3>(&L.retired[(seq + 1u) % PC_R], seq + 1u - PC_R, &L.abort)) { aborted = true; break; }
        PC_TP(4);
        const bool inA = tgt < PC_BB;
        const int j = inA ? tgt : tgt - PC_BB;
        PcSlot& SA = L.s[seq % PC_R];
        PcSlot& SB = L.s[(seq + 1u) % PC_R];
        if (u) {
          PcSlot& S = inA ? SA : SB;
          S.ent[j][0] = make_float4(r0.x, r0.y, GSR_STAGE_A(r0.z), GSR_STAGE_B(r0.w));   // (-A/2, -B): exact factors, see gsr_power
          S.ent[j][1] = make_float4(GSR_STAGE_A(r1.x), r1.y, r1.z, r1.w);               // -C/2
          S.ent[j][2] = make_float4(r2.x, __uint_as_float((uint32_t)pos), __uint_as_float((uint32_t)(36 * j)), 0.f);
          S.slot[j] = e;
          S.quads[j] = (uint8_t)byte;
        }
        uint32_t cqb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const bool has = u && ((byte >> w) & 1u);
          const uint64_t balA = __ballot(has && inA), balB = __ballot(has && !inA);
          if (has) {
            if (inA) SA.idx[w][cqa[w] + (uint32_t)__popcll(balA & gsr_lanemask_lt())] = (uint16_t)(PC_ENT_BYTES * j);
            else SB.idx[w][cqb[w] + (uint32_t)__popcll(balB & gsr_lanemask_lt())] = (uint16_t)(PC_ENT_BYTES * j);
          }
          cqa[w] += (uint32_t)__popcll(balA);
          cqb[w] += (uint32_t)__popcll(balB);
        }
        if (fill + nu >= PC_BB) {
          publish_open(PC_BB);
          fill = fill + nu - PC_BB;
#pragma unroll
          for (int w = 0; w < 4; ++w) cqa[w] = cqb[w];
        } else {
          fill += nu;
        }
      }
      PC_TP(7);
    }
    if (aborted) return;
    if (fill > 0) {
      if (seq >= PC_R && !pc_wait_gt<3>(&L.retired[seq % PC_R], seq - PC_R, &L.abort)) return;   // (free already: the entries are in it)
      publish_open(fill);
    }
    PC_TP(7);
    // the next tile's order entry is requested before this tile's zero fill: zeros for the entries nobody reached (they still own a
    // record in the Gaussian-major scratch; only those of Gaussians the view uses somewhere are ever read)
#if PC_POP_EARLY
    ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    if (ticket < n_busy) ord = tile_order[ticket];
#else
    if (lane == 0) next = gridDim.x + atomicAdd(&queue[1], 1u);      // the next ticket: its latency hides behind the zero fill
#endif
    for (int base = max_last; base < n; base += 256) {
      uint32_t gz[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int kk = base + lane + 64 * u; on[u] = kk < n; gz[u] = on[u] ? point_list[list0 + kk] : 0u; }
      if (used) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (on[u]) on[u] = used[gz[u]] != 0;
      }
      float4 w3[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (on[u]) w3[u] = rec[GSR_REC_F4 * gz[u] + 3];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (on[u]) pc_zero_record<COL>(partials, pc_slot_of(w3[u], tx, ty));
    }
#if !PC_POP_EARLY
    ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
    if (ticket < n_busy) ord = tile_order[ticket];
#endif
    PC_TP(8);
  }
  // exit marker for the consumers
  if (seq >= PC_R && !pc_wait_gt<3>(&L.retired[seq % PC_R], seq - PC_R, &L.abort)) return;
  if (lane == 0) L.flags[seq % PC_R] = PC_FLAG_EXIT;
  pc_publish(&L.prod_seq, seq + 1u);
  PC_TFLUSH();
  // The last workgroup to leave re-arms the queue for a possible second backward over the same state (see render_bwd_persistent)
  if (lane == 0 && atomicAdd(&queue[5], 1u) == gridDim.x - 1u) {
    atomicExch(&queue[1], 0u);
    atomicExch(&queue[5], 0u);
  }
}

template <bool COL>
__device__ __forceinline__ void pc_consumer(PcLds& L, const GsrRenderViews& tab, const int wv) {
  const int lane = gsr_lane();
  const int W = tab.W, H = tab.H, gx = tab.gx;
  const size_t N = (size_t)H * W;
  float pxf = 0.f, pyf = 0.f, T = 0.f, nTfbg = 0.f, dL0 = 0.f, dL1 = 0.f, dL2 = 0.f;
  float acc_dot = 0.f, last_cdot = 0.f, last_alpha = 0.f;
  int last = 0;
  // the NEXT tile's per-pixel values, requested while the current tile is still replayed (have_next = sequence number of its first batch + 1)
  float nT = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
  uint32_t nlast = 0u, have_next = 0u;
  const int red6 = lane >= 48 ? gsr_sum6_slot(lane) : -1;
  (void)red6;
  auto pixel_loads = [&](int sl, float& t_, uint32_t& l_, float& a_, float& b_, float& c_) {
    const int tile = (int)pc_peek(&L.tile[sl]);
    const GsrRenderView& vw = tab.v[pc_peek(&L.view[sl])];
    const int px = (tile % gx) * GSR_TILE + GSR_QW * (wv & 1) + (lane & 7), py = (tile / gx) * GSR_TILE + GSR_QH * (wv >> 1) + (lane >> 3);
    t_ = 0.f; l_ = 0u; a_ = 0.f; b_ = 0.f; c_ = 0.f;
    if (px < W && py < H) {
      const int pix = py * W + px;
      t_ = vw.final_T[pix]; l_ = vw.n_contrib[pix];
      a_ = vw.dL_dcolor[pix]; b_ = vw.dL_dcolor[N + pix]; c_ = vw.dL_dcolor[2 * N + pix];
    }
  };
  __builtin_amdgcn_s_setprio(PC_CONS_PRIO);  // (its waits drop to 0)
  PC_T0();
  for (uint32_t seq = 0;; ++seq) {
    if (!pc_wait_gt<PC_CONS_PRIO>(&L.prod_seq, seq, &L.abort)) return;
    if (wv == 0) PC_TP(0);
    const int sl = (int)(seq % PC_R);
    const uint32_t flags = pc_peek(&L.flags[sl]);
    if (flags & PC_FLAG_EXIT) break;
    if (flags & PC_FLAG_FIRST) {             // a new tile: this quad's pixels
      const int tile = (int)pc_peek(&L.tile[sl]);
      const GsrRenderView& vw = tab.v[pc_peek(&L.view[sl])];
      if (have_next != seq + 1u) pixel_loads(sl, nT, nlast, n0, n1, n2);
      have_next = 0u;
      pxf = (float)((tile % gx) * GSR_TILE + GSR_QW * (wv & 1) + (lane & 7));
      pyf = (float)((tile / gx) * GSR_TILE + GSR_QH * (wv >> 1) + (lane >> 3));
      const float T_final = nT;
      last = (int)nlast; dL0 = n0; dL1 = n1; dL2 = n2;
      nTfbg = -T_final * (vw.bg[0] * dL0 + vw.bg[1] * dL1 + vw.bg[2] * dL2);
      T = T_final;
      acc_dot = 0.f; last_cdot = 0.f; last_alpha = 0.f;
    }
    // the batch behind this one already opens the next tile: ask for its pixels now
    if (have_next == 0u && pc_peek(&L.prod_seq) > seq + 1u) {
      const int sn = (int)((seq + 1u) % PC_R);
      if (pc_peek(&L.flags[sn]) == PC_FLAG_FIRST) { pixel_loads(sn, nT, nlast, n0, n1, n2); have_next = seq + 2u; }
    }
#ifdef GSR_TILE_TIMING
    if (wv == 0) { if (flags & PC_FLAG_FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PC_TP(1); }
#endif
    const int m = (int)pc_peek(&L.cnt[sl][wv]);
    PcSlot& S = L.s[sl];
    if (m > 0) {
      const char* __restrict__ ebase = reinterpret_cast<const char*>(&S.ent[0][0]);
      char* __restrict__ rbase = reinterpret_cast<char*>(&S.red[wv][0][0]) + 4 * (lane >= 48 ? (COL ? lane - 48 : (red6 >= 0 ? red6 : 0)) : 0);
      const uint16_t* __restrict__ ip = S.idx[wv];
#define PC_LOAD(o, ea, eb, ec) { const float4* e_ = reinterpret_cast<const float4*>(ebase + (o)); ea = e_[0]; eb = e_[1]; ec = e_[2]; }
#define PC_VISIT(ea, eb, ec)                                                                                    \
      {                                                                                                         \
        const int pos = (int)__float_as_uint(ec.y);                                                             \
        const float blue = ec.x;                                                                                \
        const float dx = ea.x - pxf, dy = ea.y - pyf;                                                           \
        const float power = gsr_power(ea.z, ea.w, eb.x, dx, dy);                                                \
        const float G0 = GSR_EXP_OF_POWER(power);                                                               \
        const bool hit = (pos < last) && power <= 0.0f && eb.y * G0 >= GSR_ALPHA_MIN;                           \
        const float G = hit ? G0 : 0.0f;                                                                        \
        const float alpha = fminf(GSR_ALPHA_MAX, eb.y * G);                                                     \
        const float rcp = __builtin_amdgcn_rcpf(1.0f - alpha);                                                  \
        T = T * rcp;                                                                                            \
        acc_dot = __builtin_fmaf(last_alpha, last_cdot - acc_dot, acc_dot);                                     \
        const float cdot = __builtin_fmaf(blue, dL2, __builtin_fmaf(eb.w, dL1, eb.z * dL0));                    \
        last_cdot = cdot;                                                                                       \
        float dL_dalpha = cdot - acc_dot;                                                                       \
        dL_dalpha = __builtin_fmaf(dL_dalpha, T, nTfbg * rcp);                                                  \
        last_alpha = alpha;                                                                                     \
        const float v5 = G * dL_dalpha;                                                                         \
        const float t = eb.y * v5;                                                                              \
        const float tx_ = t * dx, ty_ = t * dy;                                                                 \
        float* rp_ = reinterpret_cast<float*>(rbase + __float_as_uint(ec.z));                                   \
        if (COL) {                                                                                              \
          const float w = alpha * T;                                                                            \
          const float z = gsr_wave_sum9_packed<false>(tx_, ty_, tx_ * dx, tx_ * dy, ty_ * dy, v5, w * dL0, w * dL1, w * dL2); \
          if (lane >= 48 && lane <= 56) *rp_ = z;                                                               \
        } else {                                                                                                \
          const float z = gsr_wave_sum6_packed(tx_, ty_, tx_ * dx, tx_ * dy, ty_ * dy, v5);                      \
          if (red6 >= 0) *rp_ = z;                                                                              \
        }                                                                                                       \
      }
      uint32_t o0 = ip[0], o1 = ip[1];
      float4 ea, eb, ec, xa, xb, xc;
      PC_LOAD(o0, ea, eb, ec)
      int p = 0;
      for (; p + 1 < m; p += 2) {
        PC_LOAD(o1, xa, xb, xc)
        o0 = ip[p + 2]; o1 = ip[p + 3];
        PC_VISIT(ea, eb, ec)
        PC_LOAD(o0, ea, eb, ec)
        PC_VISIT(xa, xb, xc)
      }
      if (p < m) PC_VISIT(ea, eb, ec)
#undef PC_VISIT
#undef PC_LOAD
    }
    if (wv == 0) PC_TP(2);
    // through with the batch.  The consumer that arrives last adds up the quads' totals of every entry and stores its record
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // this wave's totals are in LDS before it counts itself in
    uint32_t arrived = 0;
    if (lane == 0) arrived = atomicAdd(&L.arrive[sl], 1u);
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    if (arrived == 3u) {
      const int n = (int)pc_peek(&L.n[sl]);
      float4* __restrict__ partials = tab.v[pc_peek(&L.view[sl])].partials;
      if (lane == 0) L.arrive[sl] = 0u;
      if (lane < n) {
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        float r2x = 0.f;
        const uint32_t quads = (uint32_t)S.quads[lane];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if ((quads >> w) & 1u) {
            const float* q = S.red[w][lane];
            r0.x += q[0]; r0.y += q[1]; r0.z += q[2]; r0.w += q[3];
            r1.x += q[4]; r1.y += q[5];
            if (COL) { r1.z += q[6]; r1.w += q[7]; r2x += q[8]; }
          }
        }
        const uint32_t e = S.slot[lane];
        if (COL) gsr_store_partial(partials, e, r0, r1, r2x);
        else gsr_store_partial6(partials, e, r0, r1.x, r1.y);
      }
      pc_publish(&L.retired[sl], seq + 1u);    // the slot's LDS reads are done (the record stores may still be in flight: they hold their data)
      if (wv == 0) PC_TP(3);
    }
  }
  if (wv == 0) PC_TFLUSH();
}

#ifndef PC_WAVES_PER_EU
#define PC_WAVES_PER_EU 8
#endif
#define PC_WG_PER_CU 5        // what the hardware admits (hipOccupancyMaxActiveBlocksPerMultiprocessor says 6: the sixth never becomes resident, and its
                             // implicit first tickets would run at the very end -- tools/r05_pc_phase_timing.py counts the late starters)
#ifndef PC_NUM_SGPR
#define PC_NUM_SGPR 80       // eight waves per SIMD need <= 80 SGPRs per wave (MI355X_MICROARCH: residency by .sgpr_count)
#endif
template <bool COL>
__global__ __launch_bounds__(PC_THREADS, PC_WAVES_PER_EU) __attribute__((amdgpu_num_sgpr(PC_NUM_SGPR))) void render_bwd_pc(GsrRenderViews tab) {
  __shared__ PcLds L;
#ifdef GSR_TILE_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 2048) g_pc_start[blockIdx.x] = wall_clock64();     // residency census: do all workgroups start together?
#endif
  for (uint32_t* w = &L.prod_seq + threadIdx.x; w < &L.ctl_end; w += PC_THREADS) *w = 0u;
  __syncthreads();
  const int wv = (int)(threadIdx.x >> 6);
  if (wv < 4) pc_consumer<COL>(L, tab, wv);
  else pc_stager<COL>(L, tab);
  // a wait timed out: this launch's gradients are not to be trusted.  The CALL's error word (device memory, zeroed by its forward) makes the
  // per-Gaussian backward queued behind this kernel write NaN for dL/dmeans3D -- loud in the stream, whoever consumes the gradients -- and
  // the device's pinned word tells the host at its next backward launch on this device (gsr_launch_render_bwd)
  if (gsr_lane() == 0 && pc_peek(&L.abort) != 0u) {
    atomicExch(&tab.queue[GSR_QUEUE_BWD_ERROR], 1u);
    if (tab.pc_error_out) __hip_atomic_store(tab.pc_error_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace gsr_render
using namespace gsr_render;

// Debug: copy out and clear the forward phase timers (zeros in a normal build).
int gsr_debug_fwd_timing(unsigned long long* out16) {
#ifdef GSR_TILE_TIMING
  GSR_HIP_CHECK(hipDeviceSynchronize());
  const char* which = getenv("GSR_TIMING_KERNEL");
  unsigned long long z[16] = {0};
  if (which && which[0] == 'b') {
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bwd_timing), sizeof(unsigned long long) * 16));
    GSR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_timing), z, sizeof z));
  } else {
    GSR_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fwd_timing), sizeof(unsigned long long) * 16));
    GSR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_timing), z, sizeof z));
  }
  return 0;
#else
  for (int i = 0; i < 16; ++i) out16[i] = 0;
  return 1;
#endif
}

#ifdef GSR_TILE_TIMING
extern "C" int gsr_debug_pc_starts(unsigned long long* out2048) {   // timing builds: when each workgroup of the last render_bwd_pc launch started (100 MHz clock)
  GSR_HIP_CHECK(hipDeviceSynchronize());
  GSR_HIP_CHECK(hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_pc_start), sizeof(unsigned long long) * 2048));
  return 0;
}
#endif
extern "C" int gsr_debug_pc_occupancy() {   // workgroups of render_bwd_pc per CU according to the runtime (the hardware may admit one fewer: MI355X_MICROARCH)
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_bwd_pc<true>, PC_THREADS, 0) != hipSuccess) return -1;
  return nb;
}
extern "C" int gsr_debug_pc_error(uint32_t* out) {   // 0: no wait of render_bwd_pc ever timed out (tests read it; not part of include/gsr.h)
  GSR_HIP_CHECK(hipDeviceSynchronize());
  GSR_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pc_error), sizeof(uint32_t)));
  return 0;
}

#ifdef GSR_TRACE_TICKETS
extern "C" int gsr_debug_ticket_trace(uint32_t* out4, int n) {   // debug builds only: not part of include/gsr.h
  GSR_HIP_CHECK(hipDeviceSynchronize());
  GSR_HIP_CHECK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_ticket_trace), sizeof(uint4) * (size_t)(n < GSR_TRACE_MAX ? n : GSR_TRACE_MAX)));
  return 0;
}
#endif

static uint32_t* pc_error_word() {   // pinned, device-visible, ONE PER DEVICE (the current one); nullptr when it could not be allocated
  static uint32_t* words = [] {      //   (then a timeout still shows as NaN gradients of its call and in g_pc_error)
    uint32_t* w = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&w), 64 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess)
      return static_cast<uint32_t*>(nullptr);
    for (int i = 0; i < 64; ++i) w[i] = 0u;
    return w;
  }();
  int dev = 0;
  if (!words || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return words + dev;
}
extern "C" int gsr_debug_pc_inject_error() {   // tests: pretend a wait of the last render_bwd_pc launch timed out (not part of include/gsr.h)
  uint32_t* w = pc_error_word();
  if (!w) return 1;
  __atomic_store_n(w, 1u, __ATOMIC_RELAXED);
  return 0;
}

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

static bool has_pairs(const GsrRenderViews& tab) {
  for (int v = 0; v < tab.V; ++v)
    if (tab.v[v].partner >= 0) return true;
  return false;
}

// Launch parameters: workgroups per CU of the two blend kernels and the queue length from which the backward takes its small-batch build.
// The defaults are the optimum of the round-5 sweep (tools/autotune.py -> profiles/r05_autotune.json: V = 1 .. 8 views x three densities);
// the switches exist for that sweep.
int gsr_launch_render_fwd(const GsrRenderViews& tab, hipStream_t st) {
  if (tab.T <= 0 || tab.V <= 0) return 0;
  static const int wg_per_cu = env_int("GSR_FWD_WG_PER_CU", 6);
  const bool pairs = has_pairs(tab);
  const bool track = tab.track != 0;                // record the per-quad contribution bytes for a backward (not in forward-only calls)
  { GSR_PROF("render_fwd", st);
    const int tiles = tab.T * tab.V;
    const int per_cu = pairs ? (wg_per_cu < 5 ? wg_per_cu : 5) : wg_per_cu;   // the pair build's LDS fits 5 workgroups per CU
    const int grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    if (pairs && track) hipLaunchKernelGGL((render_fwd_persistent<true, true>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else if (pairs) hipLaunchKernelGGL((render_fwd_persistent<true, false>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else if (track) hipLaunchKernelGGL((render_fwd_persistent<false, true>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else hipLaunchKernelGGL((render_fwd_persistent<false, false>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
  }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_render_bwd(const GsrRenderViews& tab_in, hipStream_t st) {
  if (tab_in.T <= 0 || tab_in.V <= 0) return 0;
  GsrRenderViews tab = tab_in;
  // Base wave priority of a ticket (round 4).  One view per launch -- the per-GPU share of a view-sharded step -- is ONE round of tickets
  // on the resident workgroups followed by a drain in which they finish one by one: the longest 10/16 of the tickets (the order is
  // longest-first: rank = ticket) run at base priority 1, so that a CU's long lists get a larger share of it and end with the pack:
  // render_bwd 90.5 -> 87 us, the one-view step 213 -> 208.5 us (profiles/r04_small_blend_experiments.txt); nothing at 2+ views (off there).
  // Round 5, 1/256 steps, 3 alternating rounds on one box (profiles/r05_prio_frac.txt): with 3+ views the shortest quarter of the tickets
  // left at base priority 0 (192/256 raised) takes render_bwd 211.0 -> 207.9 us at four views and 379.1 -> 376.9 at eight; small raised
  // fractions (8..64/256: only the longest lists) and a second level for the longest did nothing -- the first round of tickets is ALL long.
  tab.prio_frac256 = tab.V <= 1 ? 160 : (tab.V >= 3 ? 192 : 0);
  // render_bwd_pc's waits are bounded; one that ran out marks its call's error word -- that call's dL/dmeans3D comes out as NaN -- and this
  // device's pinned host word.  Nobody syncs here, so the host's report comes with the NEXT backward launch on the device (the word is
  // cleared: the caller decides whether to go on); the NaNs do not wait for it.
  uint32_t* const pc_error = pc_error_word();
  if (pc_error && __atomic_load_n(pc_error, __ATOMIC_RELAXED) != 0u) {
    __atomic_store_n(pc_error, 0u, __ATOMIC_RELAXED);
    gsr_set_error("render_bwd_pc: a wait between the staging wave and the replaying waves timed out in an EARLIER backward launch; "
                  "the gradients of that launch are invalid (its dL/dmeans3D is NaN; set GSR_BWD_PC=0 to use the plain backward and report this)");
    return -5;
  }
  tab.pc_error_out = pc_error;
  static const int wg_per_cu = env_int("GSR_BWD_WG_PER_CU", 4);
  const bool pairs = has_pairs(tab);
  { GSR_PROF("render_bwd", st);
    const int tiles = tab.T * tab.V;
    // long queue (>= 2 tiles per resident workgroup slot, counting the empty ones): the small-batch build, more workgroups per CU
    // (round 4, with the exact lists: 80 entries at six per CU; two views -- 5000 tiles -- gain too: render_bwd 132 -> 127 us; one view loses: 81 -> 87)
    static const int small_batch_from = env_int("GSR_BWD_SMALL_BATCH_TILES", 4000);
    // ... but not for sparse scenes (mean list below ~100 entries: one batch per tile either way, and fewer workgroups per CU contend less --
    // the demo's 9 k Gaussians at 640x480: 72.8 -> 64.7 us at four views, 113.6 -> 103.5 at eight; profiles/r05_autotune.json)
    const bool small = !pairs && tiles >= small_batch_from && tab.avg_list >= 96u;
    // a sparse scene with a short queue (the demo: 9 k Gaussians, 1200 tiles per 640x480 view, one or two views): three workgroups per CU
    // (24.7 vs 27.2 us at one view, 42.3 vs 45.9 at two, in both sweeps)
    const bool few_sparse = !pairs && tab.avg_list < 96u && tiles < 4000;
    const int per_cu = pairs ? 4 : (small ? wg_per_cu + (BWD_SMALL_WAVES - 4) : (few_sparse && wg_per_cu > 3 ? 3 : wg_per_cu));
    const int grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    const bool col = !tab.no_colour_grad;
    // The producer / consumer form (round 5) for calls without fused pairs: GSR_BWD_PC = 1 / 0 forces it on / off; by default it takes the
    // DENSE scenes (mean list of 600 entries and more), where the sweep measured it 4 - 6 % ahead of the barrier form at every view count
    // (500 k Gaussians at 1080p: 241 vs 256 us at one view ... 1413 vs 1470 at eight); at BASELINE's density the two tie, on one sparse view
    // the barrier form leads (profiles/r05_autotune.json)
    static const int pc_mode = env_int("GSR_BWD_PC", -1);
    const bool use_pc = pc_mode >= 0 ? pc_mode != 0 : (tab.avg_list >= 600u && tab.avg_list < (1u << 20));
    if (!pairs && use_pc) {
      const int pgrid = tiles < 256 * PC_WG_PER_CU ? tiles : 256 * PC_WG_PER_CU;
      if (col) hipLaunchKernelGGL(render_bwd_pc<true>, dim3(pgrid), dim3(PC_THREADS), 0, st, tab);
      else hipLaunchKernelGGL(render_bwd_pc<false>, dim3(pgrid), dim3(PC_THREADS), 0, st, tab);
    }
    else if (pairs) hipLaunchKernelGGL(render_bwd_persistent<true>, dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else if (small && col) hipLaunchKernelGGL((render_bwd_persistent<false, BWD_SMALL_BB>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else if (small) hipLaunchKernelGGL((render_bwd_persistent<false, BWD_SMALL_BB, false>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else if (col) hipLaunchKernelGGL(render_bwd_persistent<false>, dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
    else hipLaunchKernelGGL((render_bwd_persistent<false, BWD_BATCH, false>), dim3(grid), dim3(GSR_BLOCK), 0, st, tab);
  }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
