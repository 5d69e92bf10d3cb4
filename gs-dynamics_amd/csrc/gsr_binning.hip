// gsr_binning.hip -- duplicates ("duplicateWithKeys"), sort and tile ranges for gfx950
// (replaces the reference extension's scan / duplicateWithKeys / radix sort /
// identifyTileRanges stage).
//
// The reference semantics: every (Gaussian, touched tile) pair becomes an entry keyed
// (tile_id << 32 | depth_bits) and the entries are sorted with a STABLE LSD radix sort, so each tile's
// list is ordered by fp32 depth bits with ties in ascending Gaussian index.  The same order is produced
// here with a hybrid MSD/LSD radix sort laid out for CDNA4:
//   1. emit_entries     entries written in Gaussian-major order, fully coalesced (one entry per lane,
//                       owner Gaussian found by an 8-step binary search in LDS), so emission is
//                       load-balanced no matter how many tiles a single Gaussian covers; a workgroup owns the
//                       entries of one radix block and also writes that block's first digit histogram;
//   2. radix passes     stable LSD passes over the TILE-ID bits only (ceil(log2 T) bits, <= 8 per pass:
//                       2 passes at 800x800 instead of the 6 a full 44-bit key sort needs); ranks inside a
//                       wave come from wave-64 __ballot peer masks, histograms live in LDS, and no global
//                       atomics are used anywhere, so the result is deterministic;
//   3. tile ranges      found by the LAST scatter pass itself (atomicMin / atomicMax of the run ends per block);
//   4. tile_sort        one workgroup per tile sorts its segment by depth: a stable LSD radix sort on the 32
//                       depth bits inside LDS (segments arrive in Gaussian-id order, so stability gives the
//                       reference tie rule); segments of 2049..4096 entries use a compare-exchange network
//                       on the unique 64-bit key (depth_bits << 32 | gaussian_id) in LDS, larger ones the
//                       same network in global memory.
// HBM traffic: emit 12 B/entry written; each radix pass 4 B read (hist) + 12 B read + 12 B written
// (scatter) per entry; tile_sort 8 B read + 4 B written per entry.
// Launches per call (all views): emit(+hist), scatter, hist, scatter(+ranges), tile_order, tile_sort = 6 at 800x800.
#include "gsr_common.h"

// kernels live in a NAMED namespace: profilers and traces show gsr_binning::<kernel>, not "(anonymous namespace)"
namespace gsr_binning {

// ------------------------------------------------------------------ exclusive scan, single workgroup
// out[i] = sum_{j<i} in[j], out[n] = total.  1024 threads x 8 items per round, carry between rounds.
// Used for per-Gaussian offsets (n = P) and radix block histograms (n = bins * blocks).
#define SCAN_THREADS 1024
#define SCAN_ITEMS 8
__global__ __launch_bounds__(SCAN_THREADS) void scan_exclusive_kernel(const uint32_t* __restrict__ in,
                                                                      uint32_t* __restrict__ out, uint32_t n,
                                                                      uint32_t* __restrict__ total_out) {
  __shared__ uint32_t wave_tot[SCAN_THREADS / GSR_WAVE];
  __shared__ uint32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  const uint32_t per_round = SCAN_THREADS * SCAN_ITEMS;
  for (uint32_t base = 0; base < n; base += per_round) {
    uint32_t v[SCAN_ITEMS];
    uint32_t first = base + (uint32_t)tid * SCAN_ITEMS;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      uint32_t idx = first + k;
      v[k] = idx < n ? in[idx] : 0u;
      sum += v[k];
    }
    // inclusive scan of `sum` across the wave
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < wv; ++w) wave_off += wave_tot[w];
    uint32_t carry = carry_s;
    uint32_t run = carry + wave_off + inc - sum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      uint32_t idx = first + k;
      if (idx < n) out[idx] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == SCAN_THREADS - 1) carry_s = run;  // last thread holds carry + round total
    __syncthreads();
  }
  if (tid == 0) {
    uint32_t tot = carry_s;
    out[n] = tot;
    if (total_out) *total_out = tot;
  }
}

// Entry count and radix block count of a view: from the table (the host knows D), or -- capacity mode, no host round trip --
// from the device word emit_entries wrote, clamped to the capacity the buffers were sized for (an overflowing view renders a
// truncated list without touching memory it does not own; the host learns the true count later and repeats the call).
__device__ __forceinline__ uint32_t bin_entries(const GsrBinView& vw) { return vw.D_dev ? min(vw.D_dev[0], vw.D) : vw.D; }
__device__ __forceinline__ uint32_t bin_blocks(const GsrBinView& vw, uint32_t D) {
  return vw.D_dev ? (D == 0 ? 1u : (D + GSR_RADIX_EPB - 1) / GSR_RADIX_EPB) : vw.nblocks;
}

// ------------------------------------------------------------------ emit (duplicateWithKeys) + first digit histogram
// ENTRY-parallel: workgroup rb owns the entries [2048 rb, 2048 rb + 2048) of its view -- exactly one block of the first radix
// pass -- so it also produces that block's digit histogram (a plain row store: no histogram launch, no atomics on global
// memory, nothing to zero).  It finds the Gaussians behind its entries from the per-256-Gaussian entry counts preprocess left
// (their prefix is rebuilt in LDS, <= 2048 blocks; beyond that a scan kernel prepared block_offsets), scans the tiles_touched
// of those Gaussians (1024 at a time -- normally all of them in one trip) into LDS offsets, and writes one entry per lane,
// coalesced; the owner of an entry is found by a 10-step binary search in LDS, so emission stays balanced however many tiles
// one Gaussian covers.
// offsets[g] (what render_bwd / preprocess_bwd use to address the Gaussian-major partial records) is written by the
// workgroup whose entry range holds it; offsets[P] = D by workgroup 0.
#define EMIT_MAX_GBLOCKS GSR_HOST_SCAN_MAX_BLOCKS
#define EMIT_G (4 * GSR_BLOCK)     // Gaussians per trip of the emission loop
__global__ __launch_bounds__(GSR_BLOCK) void emit_entries_kernel(int P, GsrBinViews tab, int shift, int bits) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  const int gx = tab.gx, T = tab.T;
  float4* __restrict__ rec = const_cast<float4*>(vw.rec);
  const uint2* __restrict__ rect = vw.rect;
  const uint32_t* __restrict__ tiles_touched = vw.tiles_touched;
  const uint32_t* __restrict__ block_sums = vw.block_sums;
  const uint32_t* __restrict__ block_offsets = vw.block_offsets;
  uint32_t* __restrict__ offsets = vw.offsets;
  uint32_t* __restrict__ tkey = vw.tkey[0];
  uint64_t* __restrict__ dg = vw.dg[0];
  uint2* __restrict__ ranges = vw.ranges;
  __shared__ uint32_t sS[EMIT_MAX_GBLOCKS + 1];   // exclusive prefix of the per-block entry counts (when no scanned array exists)
  __shared__ uint32_t soff[EMIT_G + 1];
  __shared__ uint32_t swave[GSR_BLOCK / GSR_WAVE];
  __shared__ uint32_t hist[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nblk = (P + GSR_BLOCK - 1) / GSR_BLOCK;
  // housekeeping for the last radix scatter further down the stream, which finds the tile ranges with atomicMin / atomicMax
  // (saves a memset launch): every tile starts out empty = (0xffffffff, 0)
  for (int t = (int)blockIdx.x * GSR_BLOCK + tid; t < T; t += (int)gridDim.x * GSR_BLOCK) ranges[t] = make_uint2(0xffffffffu, 0u);
  hist[tid] = 0;
  if (!block_offsets) {   // prefix of the block counts, 8 per thread (nblk <= 2048)
    uint32_t v[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int j = tid * 8 + k; v[k] = j < nblk ? block_sums[j] : 0u; sum += v[k]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) swave[wv] = inc;
    __syncthreads();
    uint32_t run = inc - sum;
    for (int w = 0; w < wv; ++w) run += swave[w];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int j = tid * 8 + k; if (j <= nblk) sS[j] = run; run += v[k]; }
    if (tid == GSR_BLOCK - 1) sS[8 * GSR_BLOCK] = run;   // nblk == 2048 (P in 524033 .. 524288): the total sits one past the last thread's slots
    __syncthreads();
  }
  auto S = [&](int j) -> uint32_t { return block_offsets ? block_offsets[j] : sS[j]; };
  const uint32_t D_total = S(nblk);
  if (blockIdx.x == 0 && tid == 0) offsets[P] = D_total;
  const uint32_t E0 = blockIdx.x * GSR_RADIX_EPB, Elim = E0 + GSR_RADIX_EPB;
  if (E0 > D_total) return;                                   // nothing to own (uniform)
  uint32_t E1 = min(D_total, Elim);
  if (vw.D_dev) E1 = min(E1, vw.D);                           // capacity mode: never past the buffers
  // Gaussian blocks to walk: jb0 = first block whose entries (or trailing zero-entry Gaussians) reach E0, jb1 = last block that
  // starts below Elim.  Uniform binary searches.
  int lo = 0, hi = nblk;                                      // jb0 = min j with S(j + 1) >= E0
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (S(mid + 1) >= E0) hi = mid; else lo = mid + 1; }
  const int jb0 = lo;
  lo = 0; hi = nblk;                                          // first j with S(j) >= Elim
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (S(mid) >= Elim) hi = mid; else lo = mid + 1; }
  const int jb1 = lo - 1;
  const uint32_t mask = (1u << bits) - 1u;
  // The Gaussians of blocks jb0 .. jb1, EMIT_G at a time (4 consecutive ones per thread; one trip unless most of them are
  // culled): local scan of tiles_touched -> offsets in LDS, then one entry per lane.
  const int g_lo = jb0 * GSR_BLOCK, g_hi = min(P, (min(jb1, nblk - 1) + 1) * GSR_BLOCK);
  uint32_t chunk_base = S(jb0);                               // entries before Gaussian g_lo
  for (int g0 = g_lo; g0 < g_hi; g0 += EMIT_G) {
    uint32_t mine[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int g = g0 + tid * 4 + q; mine[q] = g < g_hi ? tiles_touched[g] : 0u; sum += mine[q]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    __syncthreads();                                          // previous trip's soff / swave readers are done
    if (lane == 63) swave[wv] = inc;
    __syncthreads();
    uint32_t run = chunk_base + inc - sum;
    for (int w = 0; w < wv; ++w) run += swave[w];
    const uint32_t chunk_total = swave[0] + swave[1] + swave[2] + swave[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int g = g0 + tid * 4 + q;
      soff[tid * 4 + q] = run;
      if (g < g_hi && run >= E0 && run < Elim) {
        offsets[g] = run;
        reinterpret_cast<uint32_t*>(rec + GSR_REC_F4 * (size_t)g + 3)[2] = run;   // the blend backward reads it from the record
      }
      run += mine[q];
    }
    if (tid == GSR_BLOCK - 1) soff[EMIT_G] = run;
    __syncthreads();
    chunk_base += chunk_total;
    if (vw.shares_lists) continue;   // the lists come from the view with the same camera: only offsets were needed here
    const uint32_t begin = max(soff[0], E0), end = min(soff[EMIT_G], E1);
    for (uint32_t e = begin + tid; e < end; e += GSR_BLOCK) {
      int l2 = 0, h2 = EMIT_G;  // invariant: soff[l2] <= e < soff[h2]
#pragma unroll
      for (int it = 0; it < 10; ++it) {
        const int mid = (l2 + h2) >> 1;
        if (soff[mid] <= e) l2 = mid; else h2 = mid;
      }
      const int g = g0 + l2;
      const uint32_t k = e - soff[l2];
      const uint2 r = rect[g];
      const uint32_t minx = r.x & 0xffffu, miny = r.x >> 16, maxx = r.y & 0xffffu, maxy = r.y >> 16;
      const uint32_t w = maxx - minx;
      uint32_t b = k;                                    // index of the entry's tile inside the rect (row-major)
      if (w * (maxy - miny) <= 32u) {                    // small rect: the k-th set bit of the Gaussian's tile mask
        uint32_t m = __float_as_uint(rec[GSR_REC_F4 * g + 3].w);
        for (uint32_t t = 0; t < k; ++t) m &= m - 1u;
        b = (uint32_t)__ffs((int)m) - 1u;
      }
      const uint32_t ty = miny + b / w, tx = minx + b % w;
      const uint32_t key = ty * (uint32_t)gx + tx;
      tkey[e] = key;
      dg[e] = ((uint64_t)__float_as_uint(rec[GSR_REC_F4 * g + 2].y) << 32) | (uint32_t)g;
      atomicAdd(&hist[(key >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  if (!vw.shares_lists && E0 < E1 && tid < (1 << bits)) vw.block_hist[blockIdx.x * (uint32_t)(1 << bits) + tid] = hist[tid];
}

// ------------------------------------------------------------------ stable radix pass on tile-id digits
__device__ __forceinline__ uint64_t match_peers(uint32_t digit, bool valid, int bits) {
  uint64_t peers = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (digit >> b) & 1u;
    const uint64_t m = __ballot(valid && bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// Per-block digit histogram, block-major: block_hist[block * nbins + bin] (a coalesced row per block).
__global__ __launch_bounds__(GSR_BLOCK) void radix_hist_kernel(GsrBinViews tab, int cur, int shift, int bits) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  if (vw.D == 0) return;
  const uint32_t D = bin_entries(vw);
  if (D == 0 || blockIdx.x >= bin_blocks(vw, D)) return;   // grid.x is the maximum over the views (or their capacities)
  const uint32_t* __restrict__ tkey = vw.tkey[cur];
  uint32_t* __restrict__ block_hist = vw.block_hist;
  __shared__ uint32_t hist[256];
  const int tid = threadIdx.x;
  hist[tid] = 0;
  __syncthreads();
  const uint32_t mask = (1u << bits) - 1u;
  const uint32_t start = blockIdx.x * GSR_RADIX_EPB;
  const uint32_t stop = min(D, start + GSR_RADIX_EPB);
  uint32_t k[GSR_RADIX_ITEMS];
#pragma unroll
  for (int u = 0; u < GSR_RADIX_ITEMS; ++u) { const uint32_t i = start + (uint32_t)u * GSR_BLOCK + tid; k[u] = i < stop ? tkey[i] : 0u; }
#pragma unroll
  for (int u = 0; u < GSR_RADIX_ITEMS; ++u)
    if (start + (uint32_t)u * GSR_BLOCK + tid < stop) atomicAdd(&hist[(k[u] >> shift) & mask], 1u);
  __syncthreads();
  if (tid < (1 << bits)) block_hist[blockIdx.x * (uint32_t)(1 << bits) + tid] = hist[tid];
}

// Large sorts (many radix blocks): exclusive prefix of every histogram column over the blocks, in place, plus the
// bin totals behind the matrix.  One workgroup per (bin, view).
__global__ __launch_bounds__(1024) void radix_colscan_kernel(GsrBinViews tab, int bits) {
  __shared__ uint32_t wave_tot[16];
  __shared__ uint32_t carry_s;
  const GsrBinView& vw = tab.v[blockIdx.y];
  if (vw.D == 0) return;
  const uint32_t Dv = bin_entries(vw);
  if (Dv == 0) return;
  const uint32_t nb = 1u << bits, bin = blockIdx.x, nblocks = bin_blocks(vw, Dv);
  uint32_t* __restrict__ hist = vw.block_hist;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < nblocks; base += 1024) {
    const uint32_t i = base + (uint32_t)tid;
    const uint32_t v = i < nblocks ? hist[(size_t)i * nb + bin] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wv; ++w) off += wave_tot[w];
    if (i < nblocks) hist[(size_t)i * nb + bin] = off + inc - v;
    __syncthreads();
    if (tid == 1023) carry_s = off + inc;
    __syncthreads();
  }
  if (tid == 0) hist[(size_t)nblocks * nb + bin] = carry_s;
}

// Stable scatter.  No separate scan launch (below GSR_COLSCAN_MIN_BLOCKS radix blocks): every block derives its own bases from the histogram matrix
// (nblocks x nbins, L2-resident): base(bin) = sum of all counts of lower bins + counts of this bin in
// earlier blocks.  Then wave w of the block owns the w-th quarter of the block's chunk and walks it in
// order, 64 keys per step; rank inside a step = popcount of lower-lane peers with the same digit.
// The LAST pass also finds the tile ranges (no tile_ranges launch): in the block's digit-ordered LDS image the entries of a tile
// are contiguous, so the first / last entry of every tile inside the block lowers / raises that tile's range ends with
// atomicMin / atomicMax (order-independent, hence deterministic; emit_entries initialised every tile to (0xffffffff, 0)).  The
// tile keys themselves have no reader behind the last pass and are not written any more.
__global__ __launch_bounds__(GSR_BLOCK) void radix_scatter_kernel(GsrBinViews tab, int cur, int shift, int bits, int prescanned, int last_pass) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  if (vw.D == 0) return;
  const uint32_t D = bin_entries(vw), nblocks = bin_blocks(vw, D);
  if (D == 0 || blockIdx.x >= nblocks) return;
  const uint32_t* __restrict__ tkey_in = vw.tkey[cur];
  const uint64_t* __restrict__ dg_in = vw.dg[cur];
  uint32_t* __restrict__ tkey_out = vw.tkey[cur ^ 1];
  uint64_t* __restrict__ dg_out = vw.dg[cur ^ 1];
  const uint32_t* __restrict__ block_hist = vw.block_hist;
  __shared__ uint32_t wcount[4][256];
  __shared__ uint32_t s_tot[4][256];   // per-wave partial: total count of each bin over all blocks
  __shared__ uint32_t s_pre[4][256];   // per-wave partial: count of each bin in blocks before this one
  __shared__ uint32_t s_binbase[256];
  __shared__ uint32_t s_local[256];
  __shared__ uint32_t s_delta[256];
  __shared__ uint32_t s_key[GSR_RADIX_EPB];
  __shared__ uint64_t s_pay[GSR_RADIX_EPB];
  volatile uint32_t(*wbase)[256] = wcount;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t mask = (1u << bits) - 1u;
  const int nb = 1 << bits;
  for (int i = tid; i < 4 * 256; i += GSR_BLOCK) (&wcount[0][0])[i] = 0;
  // this wave's keys and payloads: all loads are issued here, ahead of the histogram-matrix sums that hide them
  const uint32_t chunk0 = blockIdx.x * GSR_RADIX_EPB;
  constexpr uint32_t per_wave = GSR_RADIX_EPB / 4;
  constexpr int STEPS = per_wave / 64;
  const uint32_t wstart = chunk0 + wv * per_wave;
  const uint32_t wstop = min(D, wstart + per_wave);
  uint32_t key[STEPS];
  uint64_t pay[STEPS];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const uint32_t i = wstart + (uint32_t)k * 64u + lane;
    key[k] = 0; pay[k] = 0;
    if (i < wstop) { key[k] = tkey_in[i]; pay[k] = dg_in[i]; }
  }
  // column sums of the histogram matrix: wave w takes blocks w, w+4, ...; lane l takes bins l, l+64, ...
  // (prescanned: radix_colscan_kernel already turned the columns into exclusive prefixes + bin totals -- the
  // per-block summing is O(nblocks^2) overall and only pays while the matrix is small)
  for (int bin = lane; bin < nb; bin += 64) {
    uint32_t tot = 0, pre = 0;
    uint32_t b = wv;
    if (prescanned) {
      if (wv == 0) { tot = block_hist[(size_t)nblocks * nb + bin]; pre = block_hist[(size_t)blockIdx.x * nb + bin]; }
      b = nblocks;
    }
    for (; b + 28 < nblocks; b += 32) {  // 8 independent loads in flight per lane
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = block_hist[(b + 4 * u) * (uint32_t)nb + bin];
#pragma unroll
      for (int u = 0; u < 8; ++u) { tot += v[u]; pre += (b + 4 * u < blockIdx.x) ? v[u] : 0u; }
    }
    for (; b < nblocks; b += 4) {
      const uint32_t v = block_hist[b * (uint32_t)nb + bin];
      tot += v;
      pre += (b < blockIdx.x) ? v : 0u;
    }
    s_tot[wv][bin] = tot;
    s_pre[wv][bin] = pre;
  }
  __syncthreads();
  if (wv == 0) {  // exclusive scan over the bin totals: lane l owns bins 4l .. 4l+3
    uint32_t t4[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bin = lane * 4 + q;
      t4[q] = bin < nb ? (s_tot[0][bin] + s_tot[1][bin] + s_tot[2][bin] + s_tot[3][bin]) : 0u;
      sum += t4[q];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    uint32_t run = inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bin = lane * 4 + q;
      if (bin < nb) s_binbase[bin] = run;
      run += t4[q];
    }
  }
  // A: per-wave digit histogram
#pragma unroll
  for (int k = 0; k < STEPS; ++k)
    if (wstart + (uint32_t)k * 64u + lane < wstop) atomicAdd(&wcount[wv][(key[k] >> shift) & mask], 1u);
  __syncthreads();
  // B: block-local layout.  The chunk is first reordered by digit inside LDS (slot = local exclusive prefix of the
  // digit + keys of that digit in earlier waves + rank inside the step), then written out: consecutive LDS slots of
  // one digit go to consecutive global addresses, so the stores are runs (~32 keys at 64 bins) instead of one
  // scattered element per lane.  (Pays once the pass is bandwidth-bound, i.e. with all views in one launch.)
  if (wv == 0) {  // exclusive scan of this block's bin counts: lane l owns bins 4l .. 4l+3
    uint32_t t4[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bin = lane * 4 + q;
      t4[q] = bin < nb ? (wcount[0][bin] + wcount[1][bin] + wcount[2][bin] + wcount[3][bin]) : 0u;
      sum += t4[q];
    }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    uint32_t run = inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int bin = lane * 4 + q;
      if (bin < nb) s_local[bin] = run;
      run += t4[q];
    }
  }
  __syncthreads();
  if (tid < nb) {
    const uint32_t gbase = s_binbase[tid] + s_pre[0][tid] + s_pre[1][tid] + s_pre[2][tid] + s_pre[3][tid];
    uint32_t lbase = s_local[tid];
    s_delta[tid] = gbase - lbase;   // global position = local slot + delta(digit)   (mod 2^32)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = wcount[w][tid];
      wcount[w][tid] = lbase;
      lbase += c;
    }
  }
  __syncthreads();
  // C: ordered walk into LDS
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const bool valid = wstart + (uint32_t)k * 64u + lane < wstop;
    const uint32_t digit = (key[k] >> shift) & mask;
    const uint64_t peers = match_peers(digit, valid, bits);
    const uint32_t rank = (uint32_t)__popcll(peers & gsr_lanemask_lt());
    uint32_t pos = 0;
    if (valid) pos = wbase[wv][digit] + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) wbase[wv][digit] = pos + (uint32_t)__popcll(peers);  // group leader advances
    __builtin_amdgcn_wave_barrier();
    if (valid) { s_key[pos] = key[k]; s_pay[pos] = pay[k]; }
  }
  __syncthreads();
  // D: write-out in runs
  const uint32_t nvalid = min(D, chunk0 + GSR_RADIX_EPB) - chunk0;
#pragma unroll
  for (int j = 0; j < GSR_RADIX_ITEMS; ++j) {
    const uint32_t l = (uint32_t)j * GSR_BLOCK + tid;
    if (l < nvalid) {
      const uint32_t kk = s_key[l];
      const uint32_t pos = l + s_delta[(kk >> shift) & mask];
      dg_out[pos] = s_pay[l];
      if (last_pass) {
        if (l == 0 || s_key[l - 1] != kk) atomicMin(&vw.ranges[kk].x, pos);
        if (l + 1 == nvalid || s_key[l + 1] != kk) atomicMax(&vw.ranges[kk].y, pos + 1u);
      } else {
        tkey_out[pos] = kk;
      }
    }
  }
}

// ------------------------------------------------------------------ tile ranges
// (tile_ranges_kernel: after the LPT-order builder below.  Merging the two into one launch -- last finished
// workgroup builds the order -- was measured slower: the agent-scope release each workgroup needs writes back its
// XCD's L2, 15 us against 6.4 + 6.3 us for two launches.)

// ------------------------------------------------------------------ LPT work queue
// Entries {tile, list start, list end, view} for ALL tiles of ALL views of the call.
// Tiles bucketed by list length (8 entries per bucket, 256 buckets), longest first.  The blend kernels'
// persistent workgroups pop tickets from this order, so heavy tiles start first and the tail of the
// kernel is made of the cheapest tiles (greedy longest-processing-time scheduling).  Order inside a bucket
// is arbitrary: per-tile results do not depend on it.
#define GSR_COLSCAN_MIN_BLOCKS 768u
#define ORD_CHUNK 12
// tile_sort's wave path (one wave per tile, entries in registers, no barriers) takes lists up to tab.wave_cap entries; longer ones
// get a workgroup each.  512 for ordinary scenes -- a lone wave sorting 800 entries is slower than a workgroup, and with only a few
// such tiles it is the launch's tail (800^2 / 100 k, one view: tile_sort 17 -> 31 us with 1024) -- and 1024 when the lists are long
// on average (1080p / 500 k: 735 entries per tile, tile_sort 589 -> 547 us per frame).
// n > 512 <=> (n + 7) >> 3 >= 65 <=> bucket <= 190: the long tickets end where bucket 191 starts (the fourth bucket of lane 47);
// n > 1024 <=> bucket <= 126: lane 31.
#define ORD_WAVES 16
#define ORD_MAX_WG 32
// Several workgroups (one 1024-thread workgroup took 14.8 us for 10 000 tiles, most of it contention on a few
// dozen hot LDS counters).  Every workgroup COUNTS all tiles of the call (the ranges are a few hundred KB, L2-resident) into
// per-wave private histograms -- split into "tiles of work items before mine" and "the rest" -- so it knows the global bucket
// starts and how many tiles of each bucket the workgroups before it will place; it then PLACES only its own work items.  No
// inter-workgroup communication, no global atomics; the order inside a bucket is arbitrary as before.
struct TileOrderLds {
  uint32_t h_before[ORD_WAVES][256];   // per wave: tiles per bucket in work items before this workgroup's
  uint32_t h_rest[ORD_WAVES][256];     // per wave: tiles per bucket in this workgroup's work items and later ones
  uint32_t start[256];                 // placement cursor of this workgroup per bucket
  uint32_t empty_before_s, n_busy_s, n_long_s, n_empty_s;
};
// `bid`: the workgroup's index among those building the order (blockIdx.x of tile_order_kernel; 0 when another kernel builds the
// whole order)
__device__ __forceinline__ void tile_order_body(const GsrBinViews& tab, int items_per_wg, int bid, TileOrderLds& L) {
  auto& h_before = L.h_before; auto& h_rest = L.h_rest; auto& start = L.start;
  uint32_t& empty_before_s = L.empty_before_s; uint32_t& n_busy_s = L.n_busy_s; uint32_t& n_long_s = L.n_long_s; uint32_t& n_empty_s = L.n_empty_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int T = tab.T;   // all tiles of all views, one order
  uint4* __restrict__ tile_order = tab.order;
  uint32_t* __restrict__ queue = tab.queue;
  for (int i = tid; i < ORD_WAVES * 256; i += 1024) { (&h_before[0][0])[i] = 0; (&h_rest[0][0])[i] = 0; }
  if (tid == 0) { empty_before_s = 0; n_empty_s = 0; }
  if (bid == 0) {
    if (tid < 9) queue[tid] = 0;     // (word 8: the error word of the call's backward, see GsrRenderViews::queue)
    // capacity mode: the counts for the host.  counts_out may be PINNED HOST memory (the caller then needs no copy on the stream --
    // a 4 us blit plus a 6 us bubble between the forward and the backward): a system-scope store, visible once this kernel has ended
    if (tab.counts_out && tid < tab.V)
      __hip_atomic_store(&tab.counts_out[tid], tab.v[tid].offsets[tab.P], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  // Work items = (view, 1024-tile slice) pairs, ORD_CHUNK of them at a time: all loads of a chunk are issued before
  // any is used (one memory latency per chunk instead of one per item); the view index is uniform, so the table
  // lookup stays a scalar load.
  const int n_it = (T + 1023) / 1024, n_items = tab.V * n_it;
  const int k_lo = bid * items_per_wg, k_hi = min(n_items, k_lo + items_per_wg);
  uint32_t empties_before = 0, empties_all = 0;    // wave-uniform counts (ballots)
  for (int k0 = 0; k0 < n_items; k0 += ORD_CHUNK) {
    uint2 r[ORD_CHUNK];
    bool live[ORD_CHUNK];
#pragma unroll
    for (int u = 0; u < ORD_CHUNK; ++u) {
      const int k = k0 + u, t = (k % n_it) * 1024 + tid;
      r[u] = make_uint2(0u, 0u);
      // a fused alias view has no entry of its own in the order: its busy tiles ride on its owner's tickets and its empty
      // tiles are painted together with the owner's (same ranges)
      live[u] = k < n_items && t < T && !tab.v[k / n_it].fused_alias;
      if (live[u]) r[u] = tab.v[k / n_it].ranges[t];
    }
#pragma unroll
    for (int u = 0; u < ORD_CHUNK; ++u) {
      const int k = k0 + u;
      const uint32_t len = r[u].y > r[u].x ? r[u].y - r[u].x : 0u;     // empty tiles arrive as (0xffffffff, 0) or (0, 0)
      const uint32_t bucket = 255u - min((len + 7u) >> 3, 255u);
      if (bucket != 255u) atomicAdd(k < k_lo ? &h_before[wv][bucket] : &h_rest[wv][bucket], 1u);
      const uint32_t ne = (uint32_t)__popcll(__ballot(live[u] && bucket == 255u));   // empty tiles: counted per wave, no hot word
      empties_all += ne;
      if (k < k_lo) empties_before += ne;
    }
  }
  if (lane == 0) { atomicAdd(&empty_before_s, empties_before); atomicAdd(&n_empty_s, empties_all); }
  __syncthreads();
  uint32_t tot = 0, bef = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < ORD_WAVES; ++w) { bef += h_before[w][tid]; tot += h_rest[w][tid]; }
    tot += bef;
  }
  __syncthreads();
  if (tid < 256) { h_rest[0][tid] = tot; h_before[0][tid] = bef; }
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the 256 bucket totals: lane l owns buckets 4l .. 4l+3
    uint32_t c4[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { c4[q] = h_rest[0][lane * 4 + q]; sum += c4[q]; }
    uint32_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    uint32_t run = inc - sum;
    uint32_t st4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { st4[q] = run; run += c4[q]; }
    if (lane == 63) n_busy_s = run;   // buckets 0..254 only: bucket 255 (empty tiles) is never counted in the histograms
    // lists of more than 1016 entries (buckets 0 .. 127): the order's first tickets.  In an ordinary scene (wave_cap 512) they are the 4096-entry
    // block's; the count also goes to the host's hint word (a launch heuristic for the NEXT call: see gsr_launch_binning)
    if (bid == 0 && lane == 32) {
      queue[2] = st4[0];
      if (tab.vlong_out) __hip_atomic_store(tab.vlong_out, st4[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // tiles longer than 512 entries (tile_sort's workgroup path): (n + 7) >> 3 >= 65, i.e. buckets 0 .. 190
    if (tab.wave_cap == 2048) {      // lists of more than 2032 entries share bucket 0: they are the workgroup tickets; queue[3] = where the lists
      if (lane == 0) n_long_s = st4[0] + c4[0];                                   // of at most 1024 entries start (bucket 127)
      if (bid == 0 && lane == 31) queue[3] = st4[0] + c4[0] + c4[1] + c4[2];
    } else if (lane == (tab.wave_cap == 1024 ? 31 : 47)) n_long_s = st4[0] + c4[0] + c4[1] + c4[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) start[lane * 4 + q] = st4[q] + h_before[0][lane * 4 + q];
  }
  __syncthreads();
  const uint32_t n_busy = n_busy_s;
  if (tid == 0) start[255] = n_busy + empty_before_s;   // the empty tiles go behind the busy ones
  __syncthreads();
  for (int k0 = k_lo; k0 < k_hi; k0 += ORD_CHUNK) {
    uint2 r[ORD_CHUNK];
#pragma unroll
    for (int u = 0; u < ORD_CHUNK; ++u) {
      const int k = k0 + u, t = (k % n_it) * 1024 + tid;
      r[u] = make_uint2(0u, 0u);
      if (k < k_hi && t < T && !tab.v[k / n_it].fused_alias) r[u] = tab.v[k / n_it].ranges[t];
    }
#pragma unroll
    for (int u = 0; u < ORD_CHUNK; ++u) {
      const int k = k0 + u, t = (k % n_it) * 1024 + tid;
      if (k < k_hi && t < T && !tab.v[k / n_it].fused_alias) {
        if (r[u].y <= r[u].x) {        // empty: (0, 0) from here on, for every later reader of the ranges
          r[u] = make_uint2(0u, 0u);
          if (!tab.v[k / n_it].shares_lists) tab.v[k / n_it].ranges[t] = r[u];
        }
        const uint32_t bucket = 255u - min((r[u].y - r[u].x + 7u) >> 3, 255u);
        // empty tiles go behind the busy ones in any order: a wave-aggregated slot (one LDS atomic per wave).
        // (Aggregating the busy buckets too -- 8 ballots per item -- was measured slower than the plain atomics.)
        uint32_t slot;
        if (bucket != 255u) slot = atomicAdd(&start[bucket], 1u);
        else {
          const uint64_t m = __ballot(1);
          const int leader = __ffsll((long long)m) - 1;
          uint32_t base = 0;
          if (lane == leader) base = atomicAdd(&start[255], (uint32_t)__popcll(m));
          base = __shfl(base, leader, 64);
          slot = base + (uint32_t)__popcll(m & gsr_lanemask_lt());
        }
        tile_order[slot] = make_uint4((uint32_t)t, r[u].x, r[u].y, (uint32_t)(k / n_it));
      }
    }
  }
  // the empty tiles (bucket 255) are sorted last and never enter the queues; queue[7] = entries of the order array
  if (bid == 0 && tid == 0) { queue[4] = n_busy; queue[6] = n_long_s; queue[7] = n_busy + n_empty_s; }
}

__global__ __launch_bounds__(1024) void tile_order_kernel(GsrBinViews tab, int items_per_wg) {
  __shared__ TileOrderLds L;
  tile_order_body(tab, items_per_wg, (int)blockIdx.x, L);
}

// ================================================================== tile-row binning (multi-view calls, T <= GSR_BIN_MAX_T)
// A single-pass counting sort on the tile id, without global atomics and without moving the entries more than once:
//   1. bin_count   a workgroup of 1024 threads owns GSR_BIN_G consecutive Gaussians of a view: it walks their tile sets (rect + mask)
//                  and counts entries per tile in LDS (one counter per tile of the image), then stores its counters as ONE ROW of the
//                  view's (workgroups x tiles) matrix -- plain coalesced stores.  The same kernel scans tiles_touched into
//                  offsets[] (the Gaussian-major slots of the backward's partial records).
//   2. bin_scan    per view: column prefix over the workgroups (in place), tile totals, exclusive scan over the tiles -> ranges.
//                  (More than BIN_DIRECT_ROWS workgroups per view: bin_colprefix first, one thread per tile column.)  Calls with
//                  few tiles (V * ceil(T / 1024) <= 4: one 800 x 800 view) run bin_scan_order instead: ONE workgroup scans and also
//                  builds the longest-first tile order -- no tile_order launch.
//   3. bin_emit    the workgroups walk their Gaussians again: an entry of tile t goes to ranges[t].x + prefix_row[t] + (LDS atomic
//                  on the workgroup's cursor of t) -- straight into its tile's segment, 8 bytes {gaussian id, depth bits}.
//   4. tile_order, tile_sort as before.  Entries ARRIVE in a tile's segment in no particular order (LDS atomics); every path of
//      tile_sort orders them by the unique 64-bit key (depth bits << 32 | gaussian id) -- the radix path by repairing runs of equal
//      depth afterwards -- so the lists are the same bits as with the stable radix passes (the reference's order), run after run.
// Against the radix path (emit, histogram, two scatter passes): 3 - 4 launches instead of 6 in front of the tile sort, entries written
// once (8 B) instead of 12 + 12 + 8 B, nothing that depends on the entry count in any grid size (capacity mode needs no special case).
// (BIN_THREADS, BinGauss, bin_tile_of, bin_for_tiles: gsr_common.h -- shared with the counting form of preprocess_fwd)
__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(int P, GsrBinViews tab) {
  extern __shared__ uint32_t s_cnt[];                 // [T] tile counters
  __shared__ uint32_t s_red[BIN_THREADS / 64];
  __shared__ uint32_t s_wave[BIN_THREADS / 64];
  __shared__ uint2 s_big[BIN_BIG_MAX];                // rects of the Gaussians that are walked by a whole wave (area > BIN_BIG_AREA)
  __shared__ uint32_t s_bigdepth[BIN_BIG_MAX];        // ... and their depth bits (depth cuts)
  __shared__ uint32_t s_nbig;
  const GsrBinView& vw = tab.v[blockIdx.y];
  const int T = tab.T, Ts = gsr_bin_stride(T), gx = tab.gx, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g0 = (int)blockIdx.x * GSR_BIN_G;
  const bool lists = !vw.shares_lists;
  // Speculative depth cuts (forward-only calls, gsr_arm_depth_cuts): a pair deeper than its tile's cut is not counted here and not emitted by
  // bin_emit -- the same test in both, so the lists stay consistent; render_fwd reports every cut tile whose list then ran out.
  const uint32_t* __restrict__ cut = vw.depth_cut;
  const bool cut_l = cut && tab.cut_lds;     // the cuts next to the counters: the test is one LDS read per pair (from global memory it cost the walk 50 %)
  const uint32_t* s_cut = s_cnt + T;
  const uint32_t* s_cc = s_cnt + 2 * T;      // ... and behind them the deepest cut of every 4 x 4 block of tiles: a Gaussian behind the coarse cuts of all
  const int gxc = (gx + 3) >> 2;             // blocks its rect touches has no pair to count -- its walk is skipped whole (measured: bin_emit 129 -> 122 us per
                                             // configs[4] frame, bin_count unchanged: with dilated cuts few 4 x 4 blocks are without a tile that has no cut)
  if (cut_l)
    for (int t = tid; t < T; t += BIN_THREADS) s_cnt[T + t] = cut[t];      // (published by the barrier in front of the tile counts)
  auto cut_at = [&](uint32_t t) { return cut_l ? s_cut[t] : cut[t]; };
  auto behind_all = [&](uint2 r, uint32_t dbits) {     // true: every tile of rect r has a cut in front of depth `dbits`
    const uint32_t x0 = (r.x & 0xffffu) >> 2, y0 = (r.x >> 16) >> 2, x1 = ((r.y & 0xffffu) - 1u) >> 2, y1 = ((r.y >> 16) - 1u) >> 2;
    uint32_t m = 0u;
    for (uint32_t y = y0; y <= y1; ++y)
      for (uint32_t x = x0; x <= x1; ++x) m = max(m, s_cc[y * gxc + x]);
    return dbits > m;
  };
  // forward-only calls: offsets[] and the offset word of the records have no reader (only the backward addresses record slots); a
  // view that shares its lists has nothing else to do here.  (The word is ONE scattered 4-byte store per Gaussian and view into the
  // 64-byte records: 57 of this kernel's 134 us at 500 k Gaussians x 8 views.)
  const bool want_offsets = !tab.forward_only;
  if (!lists && !want_offsets) {
    if (blockIdx.x == 0 && tid == 0) vw.offsets[P] = 0;
    return;
  }
  // every global load of the kernel is issued here, ahead of the first wait (the launch is a chain of memory round trips otherwise)
  uint32_t tt[BIN_PER_THREAD];
  uint2 rc[BIN_PER_THREAD], ek[BIN_PER_THREAD];
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) {
    const int g = g0 + tid * BIN_PER_THREAD + q;
    tt[q] = 0; rc[q] = make_uint2(0u, 0u); ek[q] = rc[q];
    if (g < P) { tt[q] = vw.tiles_touched[g]; if (lists) { rc[q] = vw.rect[g]; ek[q] = vw.ekey[g]; } }
  }
  // entries before this workgroup's first Gaussian: from the scanned array, or summed here from the per-256-Gaussian counts
  const int jb = g0 / GSR_BLOCK;
  uint32_t part = 0;
  if (vw.block_offsets) { if (tid == 0) part = vw.block_offsets[jb]; }
  else for (int j = tid; j < jb; j += BIN_THREADS) part += vw.block_sums[j];
  if (lists) for (int t = tid; t < T; t += BIN_THREADS) s_cnt[t] = 0;
  if (tid == 0) s_nbig = 0;
  uint32_t sum = 0;
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) sum += tt[q];
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
  if (lane == 63) s_wave[wv] = inc;
  if (lane == 0) s_red[wv] = part;
  __syncthreads();
  uint32_t run = inc - sum;
#pragma unroll
  for (int w = 0; w < BIN_THREADS / 64; ++w) { run += s_red[w]; if (w < wv) run += s_wave[w]; }
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) {
    const int g = g0 + tid * BIN_PER_THREAD + q;
    if (g < P && want_offsets) {
      vw.offsets[g] = run;
      reinterpret_cast<uint32_t*>(vw.rec_w + GSR_REC_F4 * (size_t)g + 3)[2] = run;   // the blend backward reads it from the record
    }
    run += tt[q];
  }
  if (g0 + GSR_BIN_G >= P && tid == BIN_THREADS - 1) vw.offsets[P] = run;            // the last workgroup knows the view's entry count
  if (!lists) return;
  if (cut_l) {           // the coarse cuts (uniform branch: barriers inside are safe)
    const int gyc = (T / gx + 3) >> 2;
    for (int c = tid; c < gxc * gyc; c += BIN_THREADS) {
      const int cx = c % gxc, cy = c / gxc;
      uint32_t m = 0u;
      for (int y = 4 * cy; y < min(4 * cy + 4, T / gx); ++y)
        for (int x = 4 * cx; x < min(4 * cx + 4, gx); ++x) m = max(m, s_cut[y * gx + x]);
      s_cnt[2 * T + c] = m;
    }
    __syncthreads();
  }
  // ---- tile counts
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) {
    if (!tt[q]) continue;
    if (cut_l && behind_all(rc[q], ek[q].x)) continue;
    const BinGauss b = bin_gauss(rc[q], ek[q].y);
    if (b.area > BIN_BIG_AREA) {                      // a large rect (taken whole): parked for a whole wave (a lane walking hundreds
      const uint32_t slot = atomicAdd(&s_nbig, 1u);   // of tiles alone would hold its wave up); a full list: the lane does walk it
      if (slot < BIN_BIG_MAX) { s_big[slot] = rc[q]; s_bigdepth[slot] = ek[q].x; continue; }
    }
    const uint32_t dbits = ek[q].x;
    if (cut) bin_for_tiles(b, gx, [&](uint32_t t) { if (dbits <= cut_at(t)) atomicAdd(&s_cnt[t], 1u); });
    else bin_for_tiles(b, gx, [&](uint32_t t) { atomicAdd(&s_cnt[t], 1u); });
  }
  __syncthreads();
  const uint32_t nbig = min(s_nbig, (uint32_t)BIN_BIG_MAX);
  for (uint32_t i = wv; i < nbig; i += BIN_THREADS / 64) {   // one wave per parked Gaussian, a lane per tile
    const BinGauss b = bin_gauss(s_big[i], 0u);
    const uint32_t dbits = s_bigdepth[i];
    for (uint32_t k = lane; k < b.area; k += 64) {
      const uint32_t t = bin_tile_of(b, k, gx);
      if (!cut || dbits <= cut_at(t)) atomicAdd(&s_cnt[t], 1u);
    }
  }
  if (nbig) __syncthreads();
  uint32_t* __restrict__ row = vw.tile_rows + (size_t)blockIdx.x * Ts;
  for (int t = tid; t < Ts; t += BIN_THREADS) row[t] = t < T ? s_cnt[t] : 0u;      // (the padding columns of the row stay zero)
  if (blockIdx.x == 0 && tid < 64) vw.block_hist[tid] = 0u;                        // the chained scan's words (bin_colprefix_scan_kernel)
}

// More than BIN_DIRECT_ROWS workgroups per view: one thread per tile column turns the counts into exclusive prefixes over the
// workgroups (in place) and leaves the column total in row `rows`.
__global__ __launch_bounds__(GSR_BLOCK) void bin_colprefix_kernel(GsrBinViews tab) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  if (vw.shares_lists) return;
  const int Ts = gsr_bin_stride(tab.T), rows = tab.rows, t = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (t >= Ts) return;
  uint32_t* __restrict__ m = vw.tile_rows;
  uint32_t run = 0;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = m[(size_t)(r + u) * Ts + t];
#pragma unroll
    for (int u = 0; u < 8; ++u) { m[(size_t)(r + u) * Ts + t] = run; run += v[u]; }
  }
  for (; r < rows; ++r) { const uint32_t v = m[(size_t)r * Ts + t]; m[(size_t)r * Ts + t] = run; run += v; }
  m[(size_t)rows * Ts + t] = run;
}

// The same + the exclusive scan of the tile totals -> ranges, i.e. what bin_scan did in a launch of its own (6.5 us of a step for a
// scan of T numbers): a chained scan over the view's few workgroups (<= 40).  A workgroup publishes its total as ONE word (bit 31 = valid)
// with an agent-scope relaxed store and sums its predecessors' words as they appear -- the data is its own flag, so no fence and no
// cache maintenance is involved (round 3's "last workgroup does the scan" needed both and was slower).  The words live in the view's
// radix histogram block (unused on this path), zeroed by bin_count's first workgroup.  All workgroups of the launch are resident
// together (<= 640 of 256 threads) and a workgroup only ever waits for lower-numbered ones.
__global__ __launch_bounds__(GSR_BLOCK) void bin_colprefix_scan_kernel(GsrBinViews tab) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  if (vw.shares_lists) return;
  __shared__ uint32_t s_w[GSR_BLOCK / 64];
  __shared__ uint32_t s_base;
  const int T = tab.T, Ts = gsr_bin_stride(T), rows = tab.rows, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int t = blockIdx.x * GSR_BLOCK + tid;
  uint32_t* __restrict__ m = vw.tile_rows;
  uint32_t run = 0;
  if (t < Ts) {
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = m[(size_t)(r + u) * Ts + t];
#pragma unroll
      for (int u = 0; u < 8; ++u) { m[(size_t)(r + u) * Ts + t] = run; run += v[u]; }
    }
    for (; r < rows; ++r) { const uint32_t v = m[(size_t)r * Ts + t]; m[(size_t)r * Ts + t] = run; run += v; }
    m[(size_t)rows * Ts + t] = run;
  }
  // inclusive scan of the 256 column totals
  uint32_t inc = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < GSR_BLOCK / 64; ++w) { if (w < wv) before += s_w[w]; total += s_w[w]; }
  uint32_t* __restrict__ words = vw.block_hist;
  if (tid == 0) __hip_atomic_store(&words[blockIdx.x], 0x80000000u | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (wv == 0) {            // predecessors' totals (at most 40 workgroups per view: one lane each)
    uint32_t mine = 0;
    if (lane < (int)blockIdx.x) {
      uint32_t w;
      do { w = __hip_atomic_load(&words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(w & 0x80000000u));
      mine = w & 0x7fffffffu;
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) mine += __shfl_xor(mine, k, 64);
    if (lane == 0) s_base = mine;
  }
  __syncthreads();
  const uint32_t start = s_base + before + inc - run, cap = vw.D;
  if (t < T) vw.ranges[t] = make_uint2(min(start, cap), min(start + run, cap));
}

// One view: (column prefix over the workgroups, unless bin_colprefix ran) + exclusive scan of the tile totals -> ranges.  A thread
// owns FOUR consecutive tiles (16-byte loads and stores on the rows).  Capacity mode clamps the ranges to the capacity the entry
// buffers were sized for.  Called by all BIN_THREADS threads of a workgroup.
__device__ __forceinline__ void bin_scan_view(const GsrBinViews& tab, const GsrBinView& vw, int prefixed, uint32_t* s_wave, uint32_t* s_carry) {
  const int T = tab.T, Ts = gsr_bin_stride(T), rows = tab.rows, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* __restrict__ m = vw.tile_rows;
  const uint32_t cap = vw.D;                          // entries the buffers hold (the exact count outside capacity mode)
  if (tid == 0) *s_carry = 0;
  __syncthreads();
  for (int q0 = 0; q0 < Ts / 4; q0 += BIN_THREADS) {
    const int q = q0 + tid;                           // tiles 4q .. 4q + 3
    const bool live = q < Ts / 4;
    uint4 tot = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
      if (prefixed) tot = *reinterpret_cast<const uint4*>(m + (size_t)rows * Ts + 4 * q);
      else {
        for (int r0 = 0; r0 < rows; r0 += BIN_ROW_CHUNK) {
          uint4 v[BIN_ROW_CHUNK];
#pragma unroll
          for (int r = 0; r < BIN_ROW_CHUNK; ++r)
            v[r] = r0 + r < rows ? *reinterpret_cast<const uint4*>(m + (size_t)(r0 + r) * Ts + 4 * q) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int r = 0; r < BIN_ROW_CHUNK; ++r)
            if (r0 + r < rows) {
              *reinterpret_cast<uint4*>(m + (size_t)(r0 + r) * Ts + 4 * q) = tot;
              tot.x += v[r].x; tot.y += v[r].y; tot.z += v[r].z; tot.w += v[r].w;
            }
        }
      }
    }
    const uint32_t mine = tot.x + tot.y + tot.z + tot.w;
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t start = *s_carry + inc - mine;
    for (int w = 0; w < wv; ++w) start += s_wave[w];
    if (live) {
      const uint32_t c[4] = {tot.x, tot.y, tot.z, tot.w};
      uint32_t st_ = start;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (4 * q + k < T) vw.ranges[4 * q + k] = make_uint2(min(st_, cap), min(st_ + c[k], cap));
        st_ += c[k];
      }
    }
    __syncthreads();
    if (tid == BIN_THREADS - 1) *s_carry = start + mine;
    __syncthreads();
  }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_scan_kernel(GsrBinViews tab, int prefixed) {   // grid: V
  __shared__ uint32_t s_wave[BIN_THREADS / 64];
  __shared__ uint32_t s_carry;
  const GsrBinView& vw = tab.v[blockIdx.x];
  if (vw.shares_lists) return;
  bin_scan_view(tab, vw, prefixed, s_wave, &s_carry);
}

// `order_wgs` > 0: the launch carries the tile-order builder as well -- workgroups tab.rows .. tab.rows + order_wgs - 1 of grid row 0
// run tile_order_body (they need the ranges only, like the emitting workgroups: the two run side by side on different CUs instead of
// one launch after the other; 20 us of an 8-view step, 12 us of a one-view step).
__global__ __launch_bounds__(BIN_THREADS) void bin_emit_kernel(int P, GsrBinViews tab, int order_wgs, int items_per_wg) {
  extern __shared__ uint32_t s_cur[];                 // [T] next free slot of every tile for this workgroup
  if ((int)blockIdx.x >= tab.rows) {
    __shared__ TileOrderLds L;
    if (blockIdx.y == 0) tile_order_body(tab, items_per_wg, (int)blockIdx.x - tab.rows, L);
    return;
  }
  (void)order_wgs;
  __shared__ uint2 s_big[BIN_BIG_MAX];                // see bin_count_kernel
  __shared__ uint64_t s_bigkey[BIN_BIG_MAX];
  __shared__ uint32_t s_nbig;
  const GsrBinView& vw = tab.v[blockIdx.y];
  const int T = tab.T, Ts = gsr_bin_stride(T), gx = tab.gx, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g0 = (int)blockIdx.x * GSR_BIN_G;
  if (vw.shares_lists) return;
  const uint32_t cap = vw.D;
  const uint32_t* __restrict__ row = vw.tile_rows + (size_t)blockIdx.x * Ts;
  const uint32_t* __restrict__ cut = vw.depth_cut;    // (see bin_count_kernel: the same test, pair by pair)
  const bool cut_l = cut && tab.cut_lds;
  const uint32_t* s_cut = s_cur + T;
  const uint32_t* s_cc = s_cur + 2 * T;      // (see bin_count_kernel: the same coarse test, Gaussian by Gaussian)
  const int gxc = (gx + 3) >> 2;
  if (cut_l)
    for (int t = tid; t < T; t += BIN_THREADS) s_cur[T + t] = cut[t];      // (published by the barrier behind the cursors' set-up)
  auto cut_at = [&](uint32_t t) { return cut_l ? s_cut[t] : cut[t]; };
  auto behind_all = [&](uint2 r, uint32_t dbits) {
    const uint32_t x0 = (r.x & 0xffffu) >> 2, y0 = (r.x >> 16) >> 2, x1 = ((r.y & 0xffffu) - 1u) >> 2, y1 = ((r.y >> 16) - 1u) >> 2;
    uint32_t m = 0u;
    for (uint32_t y = y0; y <= y1; ++y)
      for (uint32_t x = x0; x <= x1; ++x) m = max(m, s_cc[y * gxc + x]);
    return dbits > m;
  };
  uint64_t* __restrict__ dg = vw.dg[0];
  uint2 rc[BIN_PER_THREAD], ek[BIN_PER_THREAD];
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) {          // all loads up front; a culled Gaussian has an empty rect
    const int g = g0 + tid * BIN_PER_THREAD + q;
    rc[q] = make_uint2(0u, 0u); ek[q] = rc[q];
    if (g < P) { rc[q] = vw.rect[g]; ek[q] = vw.ekey[g]; }
  }
  if (tid == 0) s_nbig = 0;
  for (int t = tid; t < T; t += BIN_THREADS) s_cur[t] = vw.ranges[t].x + row[t];
  __syncthreads();
  if (cut_l) {
    const int gyc = (T / gx + 3) >> 2;
    for (int c = tid; c < gxc * gyc; c += BIN_THREADS) {
      const int cx = c % gxc, cy = c / gxc;
      uint32_t m = 0u;
      for (int y = 4 * cy; y < min(4 * cy + 4, T / gx); ++y)
        for (int x = 4 * cx; x < min(4 * cx + 4, gx); ++x) m = max(m, s_cut[y * gx + x]);
      s_cur[2 * T + c] = m;
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < BIN_PER_THREAD; ++q) {
    const BinGauss b = bin_gauss(rc[q], ek[q].y);
    if (b.area == 0u) continue;
    if (cut_l && behind_all(rc[q], ek[q].x)) continue;
    const uint64_t key = ((uint64_t)ek[q].x << 32) | (uint32_t)(g0 + tid * BIN_PER_THREAD + q);
    if (b.area > BIN_BIG_AREA) {
      const uint32_t slot = atomicAdd(&s_nbig, 1u);
      if (slot < BIN_BIG_MAX) { s_big[slot] = rc[q]; s_bigkey[slot] = key; continue; }
    }
    const uint32_t dbits = ek[q].x;
    bin_for_tiles(b, gx, [&](uint32_t t) {
      if (cut && dbits > cut_at(t)) return;
      const uint32_t slot = atomicAdd(&s_cur[t], 1u);
      if (slot < cap) dg[slot] = key;
    });
  }
  __syncthreads();
  const uint32_t nbig = min(s_nbig, (uint32_t)BIN_BIG_MAX);
  for (uint32_t i = wv; i < nbig; i += BIN_THREADS / 64) {
    const BinGauss b = bin_gauss(s_big[i], 0u);
    const uint64_t key = s_bigkey[i];
    for (uint32_t k = lane; k < b.area; k += 64) {
      const uint32_t t = bin_tile_of(b, k, gx);
      if (cut && (uint32_t)(key >> 32) > cut_at(t)) continue;
      const uint32_t slot = atomicAdd(&s_cur[t], 1u);
      if (slot < cap) dg[slot] = key;
    }
  }
}

__global__ __launch_bounds__(GSR_BLOCK) void tile_ranges_kernel(GsrBinViews tab, int cur) {
  const GsrBinView& vw = tab.v[blockIdx.y];
  const uint32_t* __restrict__ tkey = vw.tkey[cur];
  uint2* __restrict__ ranges = vw.ranges;
  const uint32_t D = bin_entries(vw);
  uint32_t i = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (i >= D) return;
  uint32_t t = tkey[i];
  if (i == 0) ranges[t].x = 0;
  else {
    uint32_t pt = tkey[i - 1];
    if (pt != t) { ranges[pt].y = i; ranges[t].x = i; }
  }
  if (i == D - 1) ranges[t].y = D;
}


// ------------------------------------------------------------------ per-tile depth sort
// Normalised bitonic network (every compare-exchange puts the minimum at the lower index), so virtual
// +inf padding above n never moves and pairs touching it are skipped.
template <typename PTR>
__device__ __forceinline__ void tile_sort_network(PTR a, uint32_t n, int tid, const uint32_t NT = GSR_BLOCK) {
  uint32_t lg = 1;  // log2 of the padded size
  while ((1u << lg) < n) ++lg;
  const uint32_t npairs = (1u << lg) >> 1;
  for (uint32_t lk = 1; lk <= lg; ++lk) {  // merge size k = 2^lk; all index maths are shifts and masks
    const uint32_t k = 1u << lk, hk = k >> 1;
    for (uint32_t p = tid; p < npairs; p += NT) {  // mirror step
      const uint32_t base = (p >> (lk - 1)) << lk, off = p & (hk - 1);
      const uint32_t l = base + off, r = base + (k - 1 - off);
      if (r < n) {
        uint64_t x = a[l], y = a[r];
        if (y < x) { a[l] = y; a[r] = x; }
      }
    }
    __syncthreads();
    for (int lj = (int)lk - 2; lj >= 0; --lj) {
      const uint32_t j = 1u << lj;
      for (uint32_t p = tid; p < npairs; p += NT) {
        const uint32_t l = ((p >> lj) << (lj + 1)) + (p & (j - 1)), r = l + j;
        if (r < n) {
          uint64_t x = a[l], y = a[r];
          if (y < x) { a[l] = y; a[r] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// ---- wave-level sort for short lists (n <= 512: the typical tile).  One WAVE sorts one tile: the entries sit in
// registers, NR per lane (element e = r * 64 + lane), and run through the same normalised bitonic network on the
// unique 64-bit key (depth bits << 32 | gaussian id) as tile_sort_network -- but the partner of a compare-exchange
// is fetched through the DPP / swizzle / bpermute crossbar, so there are no LDS arrays and no workgroup barriers,
// and the four waves of a workgroup work on four different tiles.
template <int M>
__device__ __forceinline__ uint32_t gsr_lane_xor(uint32_t v) {   // value of lane (lane ^ M), M in 1..63
  if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  else if constexpr (M == 3) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xf, 0xf, true);   // quad_perm [3,2,1,0]
  else if constexpr (M == 7) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);  // row_half_mirror
  else if constexpr (M == 15) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); // row_mirror
  else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
  // (xor 4 as two bank-masked DPP moves instead of ds_swizzle: measured the same, 242 vs 244 us per configs[4] frame -- not kept)
  else if constexpr (M < 32) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);          // bit mode: xor M
  else return (uint32_t)__shfl_xor((int)v, M, 64);
}
// One compare-exchange step of the network: partner of element e is e ^ MASK; the lower index keeps the minimum.
// Element layout (round 4): e = lane * NR + r -- the LOW bits of the network index select the register, the high bits the lane -- so the
// steps with a stride below NR exchange between two registers of one lane (one 64-bit compare per PAIR, four selects) and only the
// strides >= NR go through the DPP / swizzle crossbar (two moves + compare + two selects per register).  Rounds 1 - 3 had e = r * 64 + lane:
// strides 1 .. 32 crossed lanes, i.e. 51 of the 66 steps of a 2048-key sort (39 of 45 at 512 keys); now 21 of 66 (21 of 45) do:
// ~25 % fewer VALU issues per sorted list.  The network sorts whatever order the keys arrive in, so they are still LOADED coalesced
// (register r <- entry r * 64 + lane); only the sorted ids leave lane by lane (a lane stores its NR consecutive ranks).
template <int V> struct gsr_log2 { static constexpr int value = 1 + gsr_log2<(V >> 1)>::value; };
template <> struct gsr_log2<1> { static constexpr int value = 0; };
template <int MASK, int NR>
__device__ __forceinline__ void wave_sort_step(uint64_t (&x)[NR], int lane) {
  constexpr int LR = gsr_log2<NR>::value;
  constexpr int R = MASK & (NR - 1), L = MASK >> LR;       // register xor, lane xor
  constexpr int HB = MASK >= 1024 ? 1024 : MASK >= 512 ? 512 : MASK >= 256 ? 256 : MASK >= 128 ? 128 : MASK >= 64 ? 64 : MASK >= 32 ? 32 : MASK >= 16 ? 16 : MASK >= 8 ? 8 : MASK >= 4 ? 4 : MASK >= 2 ? 2 : 1;
  if constexpr (L == 0) {
    // partner = another register of this lane: decide once per pair (keys are distinct; padding entries tie and need no swap)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if ((r & HB) == 0) {                       // r = the lower index of the pair (HB = the top bit of MASK: it is set in r ^ R)
        const uint64_t a = x[r], b = x[r ^ R];
        const bool sw = b < a;
        x[r] = sw ? b : a;
        x[r ^ R] = sw ? a : b;
      }
    }
  } else {
    uint64_t y[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const uint64_t src = x[r ^ R];
      y[r] = ((uint64_t)gsr_lane_xor<L>((uint32_t)(src >> 32)) << 32) | gsr_lane_xor<L>((uint32_t)src);
    }
    // The element with the lower index keeps the minimum: take the partner's key when it is smaller (lower) or larger (upper).  Keys
    // are distinct -- only padding entries (~0) tie, and swapping equals changes nothing -- so "larger" is "not smaller" and one 64-bit
    // compare XOR the upper-half predicate decides.  HB >= NR here (the top bit of MASK is a lane bit): upper = a lane predicate.
    const bool upper = (lane & (HB >> LR)) != 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool take = (y[r] < x[r]) != upper;
      x[r] = take ? y[r] : x[r];
    }
  }
}
template <int LK, int NR>
__device__ __forceinline__ void wave_sort_stage(uint64_t (&x)[NR], int lane) {   // merge size k = 2^LK
  static_assert(LK <= 11, "the half-cleaner steps below stop at stride 512");
  wave_sort_step<(1 << LK) - 1, NR>(x, lane);
  if constexpr (LK >= 11) wave_sort_step<512, NR>(x, lane);
  if constexpr (LK >= 10) wave_sort_step<256, NR>(x, lane);
  if constexpr (LK >= 9) wave_sort_step<128, NR>(x, lane);
  if constexpr (LK >= 8) wave_sort_step<64, NR>(x, lane);
  if constexpr (LK >= 7) wave_sort_step<32, NR>(x, lane);
  if constexpr (LK >= 6) wave_sort_step<16, NR>(x, lane);
  if constexpr (LK >= 5) wave_sort_step<8, NR>(x, lane);
  if constexpr (LK >= 4) wave_sort_step<4, NR>(x, lane);
  if constexpr (LK >= 3) wave_sort_step<2, NR>(x, lane);
  if constexpr (LK >= 2) wave_sort_step<1, NR>(x, lane);
}
template <int NR>
__device__ __forceinline__ void wave_sort_tile(const uint64_t* __restrict__ seg, uint32_t* __restrict__ out, uint32_t n, int lane) {
  uint64_t x[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) { const uint32_t e = (uint32_t)(r * 64 + lane); x[r] = e < n ? seg[e] : ~0ull; }   // coalesced; any order will do
  wave_sort_stage<1, NR>(x, lane); wave_sort_stage<2, NR>(x, lane); wave_sort_stage<3, NR>(x, lane);
  wave_sort_stage<4, NR>(x, lane); wave_sort_stage<5, NR>(x, lane); wave_sort_stage<6, NR>(x, lane);
  if constexpr (NR >= 2) wave_sort_stage<7, NR>(x, lane);
  if constexpr (NR >= 4) wave_sort_stage<8, NR>(x, lane);
  if constexpr (NR >= 8) wave_sort_stage<9, NR>(x, lane);
  if constexpr (NR >= 16) wave_sort_stage<10, NR>(x, lane);
  if constexpr (NR >= 32) wave_sort_stage<11, NR>(x, lane);
  // rank e = lane * NR + r sits in register r of lane `lane`: a lane stores its NR consecutive ids (16 bytes at a time where aligned)
  const uint32_t e0 = (uint32_t)lane * NR;
  if constexpr (NR >= 4) {
    if ((((uintptr_t)out) & 15u) == 0 && e0 + NR <= n) {
#pragma unroll
      for (int r = 0; r < NR; r += 4)
        *reinterpret_cast<uint4*>(out + e0 + r) = make_uint4((uint32_t)x[r], (uint32_t)x[r + 1], (uint32_t)x[r + 2], (uint32_t)x[r + 3]);
      return;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) if (e0 + r < n) out[e0 + r] = (uint32_t)x[r];
}

// Stable LSD radix sort of one tile's entries inside LDS (n <= RCAP).  The segment arrives in
// ascending Gaussian-id order (the global tile-digit passes are stable and emission was id-major), so a
// STABLE sort on the 32-bit depth key alone yields exactly the reference order (depth, ties by id).
// 8-bit digits; a pass whose digit is identical for every key of the tile is skipped (the top exponent
// byte almost always).  Per pass: per-wave histograms (LDS atomics) -> 256-bin scan -> ordered walk with
// ranks from wave-64 __ballot peer masks.  ~5 barriers per pass instead of one per compare-exchange stage.
// Two builds of the kernel: RCAP = 2048 (36 KiB of LDS, 4 workgroups per CU) for ordinary scenes, RCAP = 4096
// (68 KiB, 2 per CU) when the average list is long (dense scenes: the network path above RCAP is n log^2 n).
template <int RCAP, int NW = 4>                          // NW: waves of the workgroup
struct TileSortLds {
  union {
    struct { uint32_t key[2][RCAP]; uint32_t val[2][RCAP]; } r;   // radix path, n <= RCAP
    uint64_t net[2 * RCAP];                                       // network path, n <= 2 RCAP
  };
  uint32_t whist[NW][256];
  uint32_t wtot[4];
  uint32_t diff;
};

template <int RCAP, int NW>
__device__ __forceinline__ int tile_radix_sort(TileSortLds<RCAP, NW>& L, uint32_t n, int tid) {
  constexpr int NT = 64 * NW;
  const int lane = tid & 63, wv = tid >> 6;
  const uint32_t q = ((n + (uint32_t)NT - 1u) / (uint32_t)NT) << 6;   // per-wave share, a multiple of 64
  const uint32_t wstart = min(n, (uint32_t)wv * q), wstop = min(n, wstart + q);
  const uint32_t diff = L.diff;
  int cur = 0;
  for (int shift = 0; shift < 32; shift += 8) {
    if (((diff >> shift) & 0xffu) == 0u) continue;      // uniform: this digit is the same for all keys
    const uint32_t* __restrict__ kin = L.r.key[cur];
    const uint32_t* __restrict__ vin = L.r.val[cur];
    uint32_t* __restrict__ kout = L.r.key[cur ^ 1];
    uint32_t* __restrict__ vout = L.r.val[cur ^ 1];
    for (int i = tid; i < NW * 256; i += NT) (&L.whist[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = wstart + lane; i < wstop; i += 64) atomicAdd(&L.whist[wv][(kin[i] >> shift) & 0xffu], 1u);
    __syncthreads();
    {  // bin tid (the first 256 threads): total over waves, exclusive scan over the 256 bins, then per-wave bases
      uint32_t c[NW], tot = 0;
      if (tid < 256) {
#pragma unroll
        for (int w = 0; w < NW; ++w) { c[w] = L.whist[w][tid]; tot += c[w]; }
      }
      uint32_t inc = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      if (tid < 256 && lane == 63) L.wtot[wv] = inc;
      __syncthreads();
      if (tid < 256) {
        uint32_t base = inc - tot;
        for (int w = 0; w < wv; ++w) base += L.wtot[w];
#pragma unroll
        for (int w = 0; w < NW; ++w) { L.whist[w][tid] = base; base += c[w]; }
      }
    }
    __syncthreads();
    volatile uint32_t* wbase = L.whist[wv];
    for (uint32_t i0 = wstart; i0 < wstop; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool valid = i < wstop;
      uint32_t key = 0, val = 0;
      if (valid) { key = kin[i]; val = vin[i]; }
      const uint32_t digit = (key >> shift) & 0xffu;
      const uint64_t peers = match_peers(digit, valid, 8);
      const uint32_t rank = (uint32_t)__popcll(peers & gsr_lanemask_lt());
      uint32_t pos = 0;
      if (valid) pos = wbase[digit] + rank;
      __builtin_amdgcn_wave_barrier();
      if (valid && rank == 0) wbase[digit] = pos + (uint32_t)__popcll(peers);
      __builtin_amdgcn_wave_barrier();
      if (valid) { kout[pos] = key; vout[pos] = val; }
    }
    __syncthreads();
    cur ^= 1;
  }
  return cur;
}

// MODE 0: both kinds of ticket in one launch (blocks below n_long: a workgroup per long list, the others: a wave per short list).
// The workgroup paths' LDS block then bounds the occupancy of the wave-sorted lists too: 2 workgroups per CU with the 68 KiB of the
// RCAP = 4096 build.  So that build launches twice: MODE 2 = the wave tickets alone in a kernel WITHOUT any LDS (6 waves per SIMD, the
// register limit), MODE 1 = the long tickets alone.
template <int RCAP, int NW>
__device__ __forceinline__ void tile_sort_long_ticket(const GsrBinViews& tab, int cur, uint32_t ticket);

// NW: waves per workgroup (4; the MODE 1 launch of the RCAP = 4096 build runs 16: a long list is sorted by 1024 threads -- the LDS
// block allows two such workgroups per CU either way, with 4 waves each that is 2 waves per SIMD working through 5 barriers per pass)
#ifndef TS_W32_WAVES
#define TS_W32_WAVES 5     // 32 keys per lane: 96 VGPRs (8 bytes of scratch) -> five waves per SIMD: tile_sort 252.5 -> 246 us per configs[4] frame (4: 97 VGPRs; 6: 268 us)
#endif
#ifndef TS_MIN_WAVES
#define TS_MIN_WAVES 7   // wave-sorted lists: 7 waves per SIMD (72 VGPRs) measured 47.0 -> 44.9 us at 8 views; 8 (64 VGPRs, spills): 47.4
#endif
// MODE 3 (dense scenes, wave_cap = 2048): the lists of 1025 .. 2032 entries -- 83 % of the entries of a configs[4] frame -- each by ONE wave
// with 32 keys per lane in registers (tickets [n_long, queue[3])), a launch of its own (its register count would cost the shorter lists
// their occupancy); MODE 2 then starts at queue[3].
template <int RCAP, int MODE = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, (RCAP <= 1024 || MODE == 2) ? TS_MIN_WAVES : (MODE == 3 ? TS_W32_WAVES : 1)) void tile_sort_kernel(GsrBinViews tab, int cur) {
  const int tid = threadIdx.x;
  if constexpr (MODE == 3) {
    const uint32_t n_long3 = tab.queue[6], n_mid = tab.queue[3];
    const uint32_t ticket = n_long3 + blockIdx.x * 4u + (uint32_t)(tid >> 6);
    if (ticket >= n_mid) return;
    const uint4 ord = tab.order[ticket];
    const GsrBinView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];
    if (vw.shares_lists) return;
    const uint32_t n = __builtin_amdgcn_readfirstlane(ord.z - ord.y);
    wave_sort_tile<32>(vw.dg[cur] + ord.y, vw.point_list + ord.y, n, tid & 63);
    return;
  }
  // The order array leads with the longest lists.  Its first n_long tickets (n > TS_WAVE_CAP) take a whole workgroup
  // each; behind them every WAVE takes one ticket (register-resident wave sort, no barriers).
  const uint32_t n_busy = tab.queue[4], n_long = tab.queue[6];
  if (MODE == 2 || (MODE == 0 && blockIdx.x >= n_long)) {
    const uint32_t first = (MODE == 2 && tab.wave_cap == 2048) ? tab.queue[3] : n_long;      // (wave_cap 2048: MODE 3 took [n_long, queue[3]))
    const uint32_t ticket = first + (MODE == 2 ? blockIdx.x : blockIdx.x - n_long) * 4u + (uint32_t)(tid >> 6);
    if (ticket >= n_busy) return;
    const uint4 ord = tab.order[ticket];
    const GsrBinView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];
    if (vw.shares_lists) return;   // sorted as part of the view that owns the lists
    const uint32_t n = __builtin_amdgcn_readfirstlane(ord.z - ord.y);
    const uint64_t* seg = vw.dg[cur] + ord.y;
    uint32_t* out = vw.point_list + ord.y;
    if (n <= 64) wave_sort_tile<1>(seg, out, n, tid & 63);
    else if (n <= 128) wave_sort_tile<2>(seg, out, n, tid & 63);
    else if (n <= 256) wave_sort_tile<4>(seg, out, n, tid & 63);
    else if (n <= 512) wave_sort_tile<8>(seg, out, n, tid & 63);
    else wave_sort_tile<16>(seg, out, n, tid & 63);
    return;
  }
  if constexpr (MODE == 1) {
    // the wave tickets have their own launch; this one STRIDES over the long tickets with a grid that fits the chip once (round 4: it
    // used to be one workgroup per tile of the call -- 32 640 workgroups of 36 KiB of LDS for a configs[4] frame, four resident per
    // CU, nearly all of them returning at once: 110 - 600 us of dispatch for a handful of lists)
    // (ordinary scenes, wave_cap 512: only the lists of more than 1016 entries -- queue[2] of them -- come here; the one-launch build takes the rest)
    const uint32_t n_mine = tab.wave_cap == 512 ? tab.queue[2] : n_long;
    for (uint32_t ticket = blockIdx.x; ticket < n_mine; ticket += gridDim.x) {
      tile_sort_long_ticket<RCAP, NW>(tab, cur, ticket);
      __syncthreads();                           // the LDS block is the next ticket's
    }
  } else if constexpr (MODE != 2) {
    if (blockIdx.x >= n_long) return;
    // An ordinary scene's few long lists (> 1016 entries: close-ups, a cluster behind one tile) would take this block's compare-exchange
    // network (LDS up to 2048 entries: ~30 us; global memory above: ~160 us for one list); they belong to the strided launch of the
    // 4096-entry block that follows (round 5: the build is chosen per ticket, not per scene) -- when the launcher issued it
    if (RCAP <= 1024 && tab.vlong_launch && blockIdx.x < tab.queue[2]) return;
    tile_sort_long_ticket<RCAP, NW>(tab, cur, blockIdx.x);
  }
}

template <int RCAP, int NW>
__device__ __forceinline__ void tile_sort_long_ticket(const GsrBinViews& tab, int cur, uint32_t ticket) {
  __shared__ TileSortLds<RCAP, NW> L;
  constexpr uint32_t NT = 64 * NW;
  const int tid = threadIdx.x;
  const uint4 ord = tab.order[ticket];         // longest lists are dispatched first
  const GsrBinView& vw = tab.v[__builtin_amdgcn_readfirstlane(ord.w)];
  if (vw.shares_lists) return;
  uint64_t* __restrict__ dg = vw.dg[cur];
  uint32_t* __restrict__ point_list = vw.point_list;
  const uint2 rg = make_uint2(ord.y, ord.z);
  const uint32_t n = rg.y - rg.x;
  if (n == 0) return;
  uint64_t* seg = dg + rg.x;
  if (n == 1) {
    if (tid == 0) point_list[rg.x] = (uint32_t)seg[0];
  } else if (n <= (uint32_t)RCAP) {
    if (tid == 0) L.diff = 0;
    __syncthreads();
    const uint32_t k0 = (uint32_t)(seg[0] >> 32);
    uint32_t d = 0;
    for (uint32_t i = tid; i < n; i += NT) {
      const uint64_t e = seg[i];
      const uint32_t k = (uint32_t)(e >> 32);
      L.r.key[0][i] = k; L.r.val[0][i] = (uint32_t)e;
      d |= k ^ k0;
    }
    if (d) atomicOr(&L.diff, d);
    __syncthreads();
    const int cur = tile_radix_sort(L, n, tid);
    // Runs of EQUAL depth: the reference order breaks such ties by ascending Gaussian id.  With the stable radix passes over
    // entries that arrive in id order nothing is left to do; entries placed by the tile-row binning arrive in no particular order,
    // so every run is put in id order here -- by the thread that holds its first element (runs are rare and short; a long one, e.g.
    // a whole tile of coplanar Gaussians, sends the tile to the 64-bit network below instead).
    {
      uint32_t* __restrict__ kk = L.r.key[cur];
      uint32_t* __restrict__ vv = L.r.val[cur];
      bool long_run = false;
      for (uint32_t i = tid; i + 1 < n; i += NT) {
        if (kk[i] == kk[i + 1] && (i == 0 || kk[i - 1] != kk[i])) {
          uint32_t j = i + 1;
          while (j + 1 < n && kk[j + 1] == kk[i] && j - i < 32u) ++j;
          if (j + 1 < n && kk[j + 1] == kk[i]) { long_run = true; continue; }
          for (uint32_t a = i + 1; a <= j; ++a) {      // insertion sort of vv[i .. j]
            const uint32_t x = vv[a];
            uint32_t b = a;
            while (b > i && vv[b - 1] > x) { vv[b] = vv[b - 1]; --b; }
            vv[b] = x;
          }
        }
      }
      if (__syncthreads_or(long_run ? 1 : 0)) {        // rare: rebuild the 64-bit keys and sort them with the network
        uint64_t e[(RCAP + NT - 1) / NT];
#pragma unroll
        for (int u = 0; u < (RCAP + NT - 1) / NT; ++u) {
          const uint32_t i = (uint32_t)u * NT + tid;
          e[u] = i < n ? (((uint64_t)kk[i] << 32) | vv[i]) : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (RCAP + NT - 1) / NT; ++u) {
          const uint32_t i = (uint32_t)u * NT + tid;
          if (i < n) L.net[i] = e[u];
        }
        __syncthreads();
        tile_sort_network(L.net, n, tid, NT);
        for (uint32_t i = tid; i < n; i += NT) point_list[rg.x + i] = (uint32_t)L.net[i];
        return;
      }
    }
    for (uint32_t i = tid; i < n; i += NT) point_list[rg.x + i] = L.r.val[cur][i];
  } else if (n <= 2u * (uint32_t)RCAP) {
    for (uint32_t i = tid; i < n; i += NT) L.net[i] = seg[i];
    __syncthreads();
    tile_sort_network(L.net, n, tid, NT);
    for (uint32_t i = tid; i < n; i += NT) point_list[rg.x + i] = (uint32_t)L.net[i];
  } else {
    __syncthreads();
    tile_sort_network((volatile uint64_t*)seg, n, tid, NT);
    for (uint32_t i = tid; i < n; i += NT) point_list[rg.x + i] = (uint32_t)seg[i];
  }
}

// A view rendered from ANOTHER call's tile lists (gsr_forward_render_shared): its own image state gets the owner's ranges, tile
// order and queue counts (the queue heads start at zero).
__global__ __launch_bounds__(GSR_BLOCK) void copy_tile_state_kernel(int T, int n_order, const uint2* __restrict__ src_ranges,
                                                                    const uint4* __restrict__ src_order, const uint32_t* __restrict__ src_queue,
                                                                    uint2* __restrict__ ranges, uint4* __restrict__ order, uint32_t* __restrict__ queue) {
  const int i = blockIdx.x * GSR_BLOCK + threadIdx.x;
  if (i < T) ranges[i] = src_ranges[i];
  if (i < n_order) order[i] = src_order[i];
  if (i < 9) queue[i] = (i == 4 || i == 6 || i == 7) ? src_queue[i] : 0u;
}

}  // namespace gsr_binning
using namespace gsr_binning;

// offsets[] of a view whose lists are another call's (emit_entries in its offsets-only form), then the owner's tile state
int gsr_launch_shared_lists(const GsrBinViews& tab_in, int P, uint32_t D, const uint2* owner_ranges, const uint4* owner_order,
                            const uint32_t* owner_queue, uint2* ranges, uint4* order, uint32_t* queue, hipStream_t st) {
  GsrBinViews tab = tab_in;
  tab.v[0].shares_lists = 1;
  { GSR_PROF("emit_entries", st);
    hipLaunchKernelGGL(emit_entries_kernel, dim3(D / GSR_RADIX_EPB + 1u, 1), dim3(GSR_BLOCK), 0, st, P, tab, 0, 1); }
  GSR_HIP_CHECK(hipGetLastError());
  { GSR_PROF("copy_tile_state", st);
    hipLaunchKernelGGL(copy_tile_state_kernel, dim3((tab.T + GSR_BLOCK - 1) / GSR_BLOCK), dim3(GSR_BLOCK), 0, st, tab.T, tab.T, owner_ranges,
                       owner_order, owner_queue, ranges, order, queue); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

int gsr_launch_scan_exclusive(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_out, hipStream_t st) {
  { GSR_PROF("scan", st);
  hipLaunchKernelGGL(scan_exclusive_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, in, out, n, total_out); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}

static int ceil_log2_u32(uint32_t n) {
  int b = 0;
  while (((uint64_t)1 << b) < n) ++b;
  return b;
}

static size_t bin_lds_limit() {
  static const size_t v = [] {
    int dev = 0, a = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&a, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && a > 0) return (size_t)a;
    return (size_t)(64 << 10);
  }();
  return v;
}
static size_t bin_lds_static() {     // the larger static LDS block of the two walks (bin_emit carries the tile-order builder)
  static const size_t v = [] {
    hipFuncAttributes a{}, b{};
    size_t m = 0;
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(bin_emit_kernel)) == hipSuccess) m = a.sharedSizeBytes;
    if (hipFuncGetAttributes(&b, reinterpret_cast<const void*>(bin_count_kernel)) == hipSuccess && b.sharedSizeBytes > m) m = b.sharedSizeBytes;
    return m ? m : (size_t)(52 << 10);
  }();
  return v;
}
// static + dynamic LDS of the two walks against what a workgroup may have on THIS device (160 KiB on gfx950; a 64 KiB part would
// fail the launch for T above ~3800): the radix path is the fallback, as for tile grids above GSR_BIN_MAX_T
bool gsr_rows_path_ok(int T) {
  const char* e = getenv("GSR_RADIX_BINNING");        // (read per call: the tests pin the radix path inside one process)
  const bool radix_only = e && *e && atoi(e) != 0;
  return !radix_only && T > 0 && T <= GSR_BIN_MAX_T && bin_lds_static() + sizeof(uint32_t) * (size_t)T <= bin_lds_limit();
}
int gsr_launch_binning(const GsrBinViews& tab_in, int P, hipStream_t st) {
  if (tab_in.V <= 0 || tab_in.T <= 0) return 0;
  GsrBinViews tab = tab_in;
  uint32_t maxD = 0, maxblk = 0;
  for (int v = 0; v < tab.V; ++v) { maxD = tab.v[v].D > maxD ? tab.v[v].D : maxD; maxblk = tab.v[v].nblocks > maxblk ? tab.v[v].nblocks : maxblk; }
  const char* force = getenv("GSR_TILE_SORT_RCAP");   // tests: "2048" / "4096" pin the build
  const bool big = force ? (force[0] == '4') : (maxD / (uint32_t)tab.T > 600u);   // long lists on average
  tab.wave_cap = big ? 2048 : 512;
  {   // long lists in an ordinary scene: see the tile_sort launches below
    static uint32_t* hint = [] { uint32_t* p = nullptr; if (hipHostMalloc((void**)&p, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return (uint32_t*)nullptr; *p = 1u; return p; }();
    tab.vlong_out = hint;
    tab.vlong_launch = (!hint || __atomic_load_n(hint, __ATOMIC_RELAXED) != 0u) ? 1 : 0;
  }
  int cur = 0;
  bool order_done = false;
  bool any_cut = false;
  for (int v = 0; v < tab.V; ++v) any_cut = any_cut || tab.v[v].depth_cut != nullptr;
  const size_t coarse = (size_t)((tab.gx + 3) / 4) * (size_t)((tab.T / tab.gx + 3) / 4);       // the deepest cut per 4 x 4 block of tiles
  tab.cut_lds = (any_cut && bin_lds_static() + sizeof(uint32_t) * (2 * (size_t)tab.T + coarse) <= bin_lds_limit()) ? 1 : 0;
  const size_t lds = sizeof(uint32_t) * (tab.cut_lds ? 2 * (size_t)tab.T + coarse : (size_t)tab.T);
  const bool rows_path = tab.rows > 0 && gsr_rows_path_ok(tab.T) && maxD > 0 && P > 0;
  if (rows_path) {             // tile-row binning: count -> (column prefix) -> scan -> emit, each ONE launch for all views
    { GSR_PROF("bin_count", st);
      hipLaunchKernelGGL(bin_count_kernel, dim3(tab.rows, tab.V), dim3(BIN_THREADS), lds, st, P, tab);
    }
    GSR_HIP_CHECK(hipGetLastError());
    const int prefixed = tab.rows > BIN_DIRECT_ROWS ? 1 : 0;
    // the column prefix also scans the tile totals into the ranges (one launch less: bin_colprefix + bin_scan 13.9 -> 8.9 us per step)
    const int col_wgs = (gsr_bin_stride(tab.T) + GSR_BLOCK - 1) / GSR_BLOCK;
    const bool chained = prefixed && col_wgs <= 64;
    if (chained) {
      GSR_PROF("bin_colprefix", st);
      hipLaunchKernelGGL(bin_colprefix_scan_kernel, dim3(col_wgs, tab.V), dim3(GSR_BLOCK), 0, st, tab);
    } else if (prefixed) {
      GSR_PROF("bin_colprefix", st);
      hipLaunchKernelGGL(bin_colprefix_kernel, dim3(col_wgs, tab.V), dim3(GSR_BLOCK), 0, st, tab);
    }
    GSR_HIP_CHECK(hipGetLastError());
    const int n_items = tab.V * ((tab.T + 1023) / 1024);
    // the tile order is built INSIDE the emit launch by extra workgroups (a launch of its own behind the emit, or scan + order in one
    // workgroup for small calls, were measured no better)
    if (!chained) {
      GSR_PROF("bin_scan", st);
      hipLaunchKernelGGL(bin_scan_kernel, dim3(tab.V), dim3(BIN_THREADS), 0, st, tab, prefixed);
    }
    GSR_HIP_CHECK(hipGetLastError());
    int order_wgs = 0, per_wg = 1;
    {
      order_wgs = n_items < ORD_MAX_WG ? n_items : ORD_MAX_WG;
      if (order_wgs < 1) order_wgs = 1;
      per_wg = (n_items + order_wgs - 1) / order_wgs;
      if (per_wg < 1) per_wg = 1;
      order_wgs = (n_items + per_wg - 1) / per_wg;
      if (order_wgs < 1) order_wgs = 1;
      order_done = true;
    }
    { GSR_PROF("bin_emit", st);
      hipLaunchKernelGGL(bin_emit_kernel, dim3(tab.rows + order_wgs, tab.V), dim3(BIN_THREADS), lds, st, P, tab, order_wgs, per_wg); }
    GSR_HIP_CHECK(hipGetLastError());
  } else if (maxD == 0 || P <= 0) {   // nothing visible in any view: every tile is empty
    for (int v = 0; v < tab.V; ++v) GSR_HIP_CHECK(hipMemsetAsync(tab.v[v].ranges, 0, sizeof(uint2) * (size_t)tab.T, st));
  } else {
    const int tbits = ceil_log2_u32((uint32_t)tab.T);
    const int npass = (tbits + 7) / 8;
    const int bpp = npass ? (tbits + npass - 1) / npass : 0;
    { GSR_PROF("emit_entries", st);      // + the digit histogram of the first radix pass
    const int bits0 = npass ? (tbits < bpp ? tbits : bpp) : 1;
    hipLaunchKernelGGL(emit_entries_kernel, dim3(maxD / GSR_RADIX_EPB + 1u, tab.V), dim3(GSR_BLOCK), 0, st, P, tab, 0, bits0); }
    GSR_HIP_CHECK(hipGetLastError());
    for (int pass = 0; pass < npass; ++pass) {
      const int shift = pass * bpp;
      const int bits = (tbits - shift) < bpp ? (tbits - shift) : bpp;
      if (pass > 0) {
        GSR_PROF("radix_hist", st);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(maxblk, tab.V), dim3(GSR_BLOCK), 0, st, tab, cur, shift, bits);
      }
      GSR_HIP_CHECK(hipGetLastError());
      const int prescanned = maxblk >= GSR_COLSCAN_MIN_BLOCKS ? 1 : 0;
      if (prescanned) {
        GSR_PROF("radix_colscan", st);
        hipLaunchKernelGGL(radix_colscan_kernel, dim3(1u << bits, tab.V), dim3(1024), 0, st, tab, bits);
      }
      GSR_HIP_CHECK(hipGetLastError());
      { GSR_PROF("radix_scatter", st);
      hipLaunchKernelGGL(radix_scatter_kernel, dim3(maxblk, tab.V), dim3(GSR_BLOCK), 0, st, tab, cur, shift, bits, prescanned,
                         pass == npass - 1 ? 1 : 0); }
      GSR_HIP_CHECK(hipGetLastError());
      cur ^= 1;
    }
    if (npass == 0) {   // a single tile: no radix pass ran, the run boundaries come from the emitted keys directly
      GSR_PROF("tile_ranges", st);
      hipLaunchKernelGGL(tile_ranges_kernel, dim3((maxD + GSR_BLOCK - 1) / GSR_BLOCK, tab.V), dim3(GSR_BLOCK), 0, st, tab, cur);
    }
    GSR_HIP_CHECK(hipGetLastError());
  }
  if (!order_done)
    if (int rc = gsr_launch_tile_order(tab, st)) return rc;
  if (maxD > 0 && P > 0) {
    { GSR_PROF("tile_sort", st);
    if (big) {  // long lists on average: the big-LDS build for the long tickets, the wave tickets in a launch of their own
      hipLaunchKernelGGL((tile_sort_kernel<2048, 2>), dim3((tab.V * tab.T + 3) / 4), dim3(GSR_BLOCK), 0, st, tab, cur);
      if (tab.wave_cap == 2048) hipLaunchKernelGGL((tile_sort_kernel<2048, 3>), dim3((tab.V * tab.T + 3) / 4), dim3(GSR_BLOCK), 0, st, tab, cur);
      // Long tickets (lists above 2032 entries).  Round 4: (i) the launch STRIDES over them with a grid that fits the chip once -- it used to
      // be one workgroup per tile of the call (32 640 workgroups of 36 - 68 KiB of LDS for a configs[4] frame, nearly all returning at
      // once), and that dispatch, not the sorting, was what made the 4096-entry block look slow in round 3 (507 vs 309 us); (ii) the
      // 4096-entry block is the default: radix sort in LDS up to 4096 entries, 64-bit network in LDS up to 8192.  A deforming configs[4]
      // episode has thousands of lists of 2033 .. 7000 entries per frame; with the 2048-entry block those ran the LDS network (<= 4096)
      // or the network in GLOBAL memory (above): tile_sort 1040 us per frame of the episode, now ~450; its renders 1.67 -> 1.14 ms per
      // frame.  (One wave per list with 64 keys per lane -- 192 VGPRs, 116 us per list -- was
      // measured too: slower than either.)
      const int g1 = tab.V * tab.T < 512 ? tab.V * tab.T : 512;        // two workgroups of the 68 KiB block per CU, eight waves each
      hipLaunchKernelGGL((tile_sort_kernel<4096, 1, 8>), dim3(g1), dim3(512), 0, st, tab, cur);     // (4 / 16 waves: render of the configs[4] episode 1.12 / 1.11 ms per frame against 1.00)
    } else
      {
        // ordinary scenes: the 1024-entry LDS block (20 KiB: eight workgroups per CU -- the wave-sorted lists are the bulk of the work
        // and want the occupancy); GSR_TILE_SORT_RCAP=2048 keeps the 36 KiB build
        const bool small_lds = !(force && force[0] == '2');
        if (small_lds) {
          hipLaunchKernelGGL(tile_sort_kernel<1024>, dim3(tab.V * tab.T), dim3(GSR_BLOCK), 0, st, tab, cur);
          // ... and the scene's few lists of more than 1016 entries on the 4096-entry block: LDS radix sort up to 4096 entries, LDS network up
          // to 8192 (was: the LDS network up to 2048, the network in GLOBAL memory above).  16 waves per workgroup: a lone list's latency
          // counts here, not throughput.  Normally there is no such list and the launch would cost ~2.7 us for nothing, so it is issued only
          // when the PREVIOUS call reported one (tab.vlong_launch; a call that meets its first long list sorts it the old way, once)
          if (tab.vlong_launch) hipLaunchKernelGGL((tile_sort_kernel<4096, 1, 16>), dim3(64), dim3(1024), 0, st, tab, cur);
        } else hipLaunchKernelGGL(tile_sort_kernel<2048>, dim3(tab.V * tab.T), dim3(GSR_BLOCK), 0, st, tab, cur);
      } }
    GSR_HIP_CHECK(hipGetLastError());
  }
  return 0;
}

int gsr_launch_tile_order(const GsrBinViews& tab, hipStream_t st) {
  { GSR_PROF("tile_order", st);
    const int n_items = tab.V * ((tab.T + 1023) / 1024);            // (view, 1024-tile slice) work items
    int wgs = n_items < ORD_MAX_WG ? n_items : ORD_MAX_WG;
    if (wgs < 1) wgs = 1;
    const int per_wg = (n_items + wgs - 1) / wgs;
    wgs = per_wg > 0 ? (n_items + per_wg - 1) / per_wg : 1;
    hipLaunchKernelGGL(tile_order_kernel, dim3(wgs < 1 ? 1 : wgs), dim3(1024), 0, st, tab, per_wg > 0 ? per_wg : 1); }
  GSR_HIP_CHECK(hipGetLastError());
  return 0;
}
