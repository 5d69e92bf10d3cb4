// gsr_common.h -- shared declarations of the gfx950 Gaussian rasterizer kernels.
// Wave = 64 lanes everywhere; workgroups are 256 threads (4 waves) unless stated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gsr.h"

#define GSR_WAVE 64
#define GSR_BLOCK 256
#define GSR_NEAR_Z 0.2f
#define GSR_ALPHA_MIN (1.0f / 255.0f)
#define GSR_ALPHA_MAX 0.99f
#define GSR_T_EPS 0.0001f

// ---------------------------------------------------------------- state layouts (HBM)
// Geometry state: per-Gaussian records written by preprocess, read (gathered) by the blend kernels.
// ONE 64-byte, 64-byte-aligned AoS record per Gaussian (4 x float4): a blend-kernel gather touches exactly one
// cache line (a 48-byte record straddled two lines half of the time), and the backward finds everything it needs
// to address the Gaussian's partial-gradient slots in the same line (no separate rect / offsets gathers):
//   rec[4g+0] {mean2D.x, mean2D.y, conicA, conicB}
//   rec[4g+1] {conicC, opacity, r, g}
//   rec[4g+2] {b, depth, bits(alpha-box x: xmin | xmax<<16), bits(alpha-box y: ymin | ymax<<16)}
//   rec[4g+3] {bits(rect.x: minx | miny<<16), bits(rect.y: maxx | maxy<<16), bits(offsets[g]) (written by emit),
//              bits(tile mask: which tiles of a rect of <= 32 tiles are in the Gaussian's lists)}
// The alpha-box is the int16 pixel box outside which alpha < 1/255.
#define GSR_REC_F4 4
// Tile-row binning (gsr_binning.hip): a workgroup of 1024 threads owns GSR_BIN_G consecutive Gaussians of a view.
#ifndef GSR_BIN_G
#define GSR_BIN_G 2048              // Gaussians per counting / emitting workgroup.  Measured (100 k Gaussians, step us at 1 / 2 / 4 / 8 views): 4096: 228 / 318 /
                                    // 488 / 850, 2048: 220 / 306 / 479 / 850, 1024: 219 / 308 / 479; configs[4] frame: the same; 8192: worse everywhere
#endif
#define GSR_BIN_MAX_T 10240         // tile counters of a view live in LDS: 4 T dynamic bytes (40 KiB at the limit) next to the kernels' static
                                    // arrays -- ~50 KiB in bin_emit_kernel since the tile-order builder moved into it (TileOrderLds 33.8 KiB, s_big /
                                    // s_bigkey 8 KiB each): ~83 KiB at 1080p, ~90 KiB at the limit, i.e. gfx950's 160 KiB at one workgroup per CU.
                                    // gsr_launch_binning checks static + dynamic bytes against the device's limit and takes the radix path
                                    // otherwise (as larger tile grids do)
struct GeomState {
  float4* rec;            // [4P]
  uint2* rect;            // {minx | miny<<16, maxx | maxy<<16} in tiles
  uint32_t* tiles_touched;
  uint32_t* offsets;      // [P+1] exclusive prefix of tiles_touched (written by emit_entries)
  uint32_t* block_sums;   // [ceil(P/256)] per-preprocess-block totals of tiles_touched
  uint32_t* block_offsets;// [ceil(P/256)+1] their exclusive scan
  uint32_t* clamped;      // [P] bit ch set when SH colour channel was clamped at 0
  uint32_t* counters;     // [0] = num_rendered
  uint2* ekey;            // [P] {depth bits, tile mask}: what the tile-row binning reads per Gaussian next to rect[] (coalesced 8 + 8 bytes
                          //     instead of two 16-byte gathers from the 64-byte record)
  uint2* block_hash;      // [ceil(P/256)] per preprocess block: .x = 1 when a Gaussian of the block differs from the geometry state the
                          //     forward was asked to compare itself with (single-view entry points: a second render with the same
                          //     geometry reuses the first one's lists)
  uint32_t* tile_rows;    // [(ceil(P / GSR_BIN_G) + 1) x GSR_BIN_MAX_T] the (workgroups x tiles) matrix of the tile-row binning for the
                          //     SINGLE-VIEW entry points (round 4; multi-view calls keep theirs in the batch state)
  uint8_t* used;          // [P] 1: some pixel of this view blended the Gaussian (set by the tracking forward, cleared by preprocess; valid when
                          //     counters[1] != 0): the per-Gaussian backward skips the records of the others -- all zeros: 45 % of the entries
};
struct ImageState {
  float* final_T;         // [H*W]
  uint32_t* n_contrib;    // [H*W]
  uint2* ranges;          // [T]
  uint4* tile_order;      // [T] {tile id, list start, list end, 0} sorted by list length, longest first (LPT queue)
  uint32_t* queue;        // [16] work-queue heads: [0] render_fwd, [1] render_bwd; [4] = number of non-empty tiles; [8] = error word of the call's
                          //   backward (zeroed with the heads by every forward; render_bwd_pc sets it when one of its bounded waits ran out, and the
                          //   per-Gaussian backward behind it then hands out NaN instead of garbage: GSR_QUEUE_BWD_ERROR)
};
// Binning state: tile-key / (depth,gid) entries, double-buffered for the radix passes.
struct BinningState {
  uint32_t* tkey[2];      // [D] tile id of each duplicate
  uint64_t* dg[2];        // [D] depth_bits << 32 | gaussian id
  uint32_t* point_list;   // [D] final per-tile depth-sorted Gaussian ids
  uint32_t* block_hist;   // [nblocks * 256] radix block histograms (block-major: hist[block][bin])
  uint8_t* contrib;       // [D] per list entry: bit w = some pixel of quad w of the entry's tile blended it (written by the forward blend,
                          //     read by the backward: its per-quad lists are exactly these bits)
};

static inline __host__ __device__ size_t gsr_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

#define GSR_RADIX_ITEMS 8                      // keys per thread per radix block
#define GSR_RADIX_EPB (GSR_BLOCK * GSR_RADIX_ITEMS)  // keys per radix block

static inline uint32_t gsr_radix_blocks(uint32_t D) { return D == 0 ? 1u : (D + GSR_RADIX_EPB - 1) / GSR_RADIX_EPB; }

static inline size_t gsr_carve_geom(void* base, int32_t P, GeomState* g) {
  size_t off = 0, Pn = (size_t)(P > 0 ? P : 1);
  char* b = (char*)base;
  auto take = [&](size_t bytes) { char* p = b ? b + off : nullptr; off += gsr_align(bytes); return p; };
  g->rec = (float4*)take(Pn * 16 * GSR_REC_F4);
  g->rect = (uint2*)take(Pn * 8);
  g->tiles_touched = (uint32_t*)take(Pn * 4);
  g->offsets = (uint32_t*)take((Pn + 1) * 4);
  const size_t nblk = (Pn + GSR_BLOCK - 1) / GSR_BLOCK;
  g->block_sums = (uint32_t*)take(nblk * 4);
  g->block_offsets = (uint32_t*)take((nblk + 1) * 4);
  g->clamped = (uint32_t*)take(Pn * 4);
  g->counters = (uint32_t*)take(64);
  g->ekey = (uint2*)take(Pn * 8);
  g->block_hash = (uint2*)take(nblk * 8);
  g->tile_rows = (uint32_t*)take(((Pn + GSR_BIN_G - 1) / GSR_BIN_G + 1) * (size_t)GSR_BIN_MAX_T * 4);
  g->used = (uint8_t*)take(Pn);
  return off;
}
static inline size_t gsr_carve_image(void* base, int32_t H, int32_t W, ImageState* im) {
  size_t off = 0, N = (size_t)H * W;
  size_t T = (size_t)((H + GSR_TILE - 1) / GSR_TILE) * ((W + GSR_TILE - 1) / GSR_TILE);
  char* b = (char*)base;
  auto take = [&](size_t bytes) { char* p = b ? b + off : nullptr; off += gsr_align(bytes); return p; };
  im->final_T = (float*)take((N ? N : 1) * 4);
  im->n_contrib = (uint32_t*)take((N ? N : 1) * 4);
  im->ranges = (uint2*)take((T ? T : 1) * 8);
  im->tile_order = (uint4*)take((T ? T : 1) * 16);
  im->queue = (uint32_t*)take(64);
  return off;
}
static inline size_t gsr_carve_binning(void* base, uint32_t D, BinningState* bs) {
  size_t off = 0, Dn = D ? D : 1;
  size_t nb = gsr_radix_blocks(D);
  char* b = (char*)base;
  auto take = [&](size_t bytes) { char* p = b ? b + off : nullptr; off += gsr_align(bytes); return p; };
  bs->tkey[0] = (uint32_t*)take(Dn * 4);
  bs->tkey[1] = (uint32_t*)take(Dn * 4);
  bs->dg[0] = (uint64_t*)take(Dn * 8);
  bs->dg[1] = (uint64_t*)take(Dn * 8);
  bs->point_list = (uint32_t*)take(Dn * 4);
  bs->block_hist = (uint32_t*)take(256 * (nb + 1) * 4);   // + one row: bin totals of the column-scanned form
  bs->contrib = (uint8_t*)take(Dn);
  return off;
}

// Backward scratch: one 36-byte record per list entry, Gaussian-major (entry e = offsets[g] + k,
// k = row-major rank of the tile inside the Gaussian's rect): nine floats
//   {d mean2D.x, d mean2D.y, dA, dB, dC, d opacity, dr, dg, db}, moved as three 12-byte (dwordx3) accesses.
#define GSR_PARTIAL_FLOATS 9
// Up to this many preprocess blocks (P <= 512 Ki Gaussians) the per-block entry counts are summed on the host
// (one small pinned copy that replaces the count read-back) and emit blocks add up their own base; above it a
// scan kernel prepares block_offsets as before.
#define GSR_HOST_SCAN_MAX_BLOCKS 2048
static inline bool gsr_host_block_scan(int P) { return (P + GSR_BLOCK - 1) / GSR_BLOCK <= GSR_HOST_SCAN_MAX_BLOCKS; }

// ---------------------------------------------------------------- propagation network pieces (gsr_gnn.hip)
int gsr_launch_gnn_aggregate(int N, int n_sum, int H, const float* rew1, const float* a23, const long long* send, const long long* row_start, float* agg, hipStream_t st,
                             const float* res_a = nullptr, const float* res_b = nullptr, float* res_out = nullptr);
int gsr_launch_gnn_rel_inputs(int E, int A, int G, int S, const float* nodes, const long long* recv, const long long* send, float* out, hipStream_t st);

// ---------------------------------------------------------------- error plumbing (gsr_api.hip)
void gsr_set_error(const char* fmt, ...);
#define GSR_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      gsr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (int)_e ? (int)_e : -1;                                                          \
    }                                                                                         \
  } while (0)

// ---------------------------------------------------------------- per-kernel HIP-event timing (gsr_api.hip)
// When profiling is enabled (gsr_profile_begin) every launch site brackets its kernel with two events
// recorded on the launch stream; gsr_profile_end turns them into per-kernel totals.  Off by default:
// no events are created or recorded in normal operation.
struct GsrRange {          // roctx range (no-op when the marker library is absent or GSR_ROCTX=0), see gsr_api.hip
  explicit GsrRange(const char* name);
  ~GsrRange();
  bool on;
};
struct GsrProfScope {
  GsrProfScope(const char* name, hipStream_t st);
  ~GsrProfScope();
  int slot;
  hipStream_t st;
  GsrRange range;        // every launch site is a range named like its kernel
};
#define GSR_PROF_CAT2(a, b) a##b
#define GSR_PROF_CAT(a, b) GSR_PROF_CAT2(a, b)
#define GSR_PROF(name, st) GsrProfScope GSR_PROF_CAT(_gsr_prof_, __LINE__)(name, st)

// ---------------------------------------------------------------- launchers (one per .hip file)
struct GsrCam {  // host copy of the scalar settings; matrices stay on the device
  int H, W, gx, gy, T;
  float tanfovx, tanfovy, scale_modifier;
  int sh_degree, M;
  const float *bg, *view, *proj, *campos;
};

// Every stage is ONE launch for all V views of a call (V = 1 for the single-view entry points): a kernel finds
// its view in blockIdx.y -- or, for the per-tile kernels, in the 4th word of the work-queue ticket -- and takes
// that view's pointers from a table passed by value as a kernel argument (kernarg segment: scalar loads, no
// extra copy).  One launch per stage instead of one per view keeps the host off the critical path (a HIP launch
// costs ~3 us of host time; a 4-view step used to issue ~75 of them) and lets all views share one LPT tile queue.
struct GsrPreView {            // preprocess
  const float *view, *proj, *campos;
  const float* colors;   // this view's precomputed colours [P,3] (nullptr: the colours shared by all views, or SH)
  float tanfovx, tanfovy;
  float4* rec; uint2* rect; uint32_t* tiles_touched; uint32_t* clamped; int32_t* radii; uint32_t* block_sums;
  uint2* ekey;
  uint2* block_hash;     // != nullptr (compare mode): per preprocess block, .x = 1 when a Gaussian of the block differs from the compared state
  const float4* cmp_rec; const uint2* cmp_rect; const uint2* cmp_ekey; const uint32_t* cmp_tiles;   // the compared (earlier) geometry state, or nullptr
  uint8_t* used; uint32_t* tracked;   // GeomState::used, &GeomState::counters[1]: both cleared here, set by the tracking forward
  int skip;              // 1: nothing to preprocess for this view (forward-only fused alias: its owner's tile pass reads its colours
                         //    straight from its colour array, nobody reads a record of its own)
};
struct GsrPreViews {
  int V;
  // raw-parameter mode (include/gsr.h: gsr_raw_params): activations applied inline, activated values written by the view-0 blocks
  const float *raw_rot, *raw_op, *raw_sc;
  float *rot_out, *op_out, *sc_out;
  GsrPreView v[GSR_MAX_BATCH];
};
struct GsrBinView {            // emit .. tile_sort
  const float4* rec; const uint2* rect; const uint32_t* tiles_touched;
  const uint32_t* block_sums; const uint32_t* block_offsets;   // block_offsets == nullptr: emit adds up block_sums itself
  uint32_t* offsets;
  uint32_t* tkey[2]; uint64_t* dg[2]; uint32_t* point_list; uint32_t* block_hist;
  uint2* ranges;
  uint32_t D, nblocks;     // entries and radix blocks of the view -- or, with D_dev set, the CAPACITY the buffers were sized for
  const uint32_t* D_dev;   // != nullptr: the entry count lives on the device (offsets[P], written by emit_entries): no host round trip
  uint32_t shares_lists;   // 1: same camera as an earlier view of the call -- its tile lists are that view's (no binning of its own)
  int32_t owner;           // index of the view whose lists this one uses (itself unless shares_lists)
  uint32_t fused_alias;    // 1: additionally blended INSIDE its owner's tile pass (GsrRenderView::partner): no tickets for its busy tiles
  const uint2* ekey;       // tile-row binning: {depth bits, tile mask} per Gaussian
  float4* rec_w;           // the record array again, writable (the offset word of a record is filled in by the binning stage)
  uint32_t* tile_rows;     // tile-row binning: [rows + 1][T] per-workgroup tile counts (-> exclusive prefixes over the workgroups), row `rows` = totals
  const uint32_t* depth_cut;   // speculative depth cuts (gsr_arm_depth_cuts; forward-only calls): [T] depth bits per tile, a (Gaussian, tile) pair with a
                               //   larger depth is neither counted nor emitted; nullptr = none
};
struct GsrBinViews {
  int V, T, gx; uint4* order; uint32_t* queue;
  uint32_t* counts_out; int P;   // capacity mode: tile_order also copies every view's entry count (offsets_v[P]) to counts_out[v]
  int wave_cap;                  // tile_sort: lists up to this length (512 / 1024 / 2048) are sorted by one wave each (set by gsr_launch_binning)
  int rows;                      // tile-row binning: workgroups per view of the count / emit kernels (0: the radix path)
  uint32_t* vlong_out;           // pinned host word: the tile order stores the call's number of lists above 1016 entries here (or nullptr)
  int vlong_launch;              // 1: a launch of the 4096-entry block follows the ordinary tile_sort launch and takes those lists
  int forward_only;              // GSR_FORWARD_ONLY: no backward will read these states (the record-slot offsets are not produced)
  int cut_lds;                   // 1: the count / emit walks were given LDS for a copy of their view's depth cuts (behind the tile counters / cursors)
  GsrBinView v[GSR_MAX_BATCH];
};
struct GsrRenderView {         // blend forward / backward
  const uint32_t* point_list; const float4* rec; const float* bg;
  float* final_T; uint32_t* n_contrib; float* out_color; float* out_depth;
  const float* dL_dcolor; const uint2* rect; const uint32_t* offsets; float4* partials;
  const uint2* ranges;
  const float* colors;   // != nullptr (fused alias of a forward-only call): this view's colours [P,3]; it has no records of its own
  uint8_t* contrib;      // the lists' per-entry quad-contribution bytes (BinningState::contrib of the view that owns the lists)
  uint8_t* used; uint32_t* tracked;   // this view's GeomState::used / &counters[1] (forward: written; see GeomState)
  int partner;       // >= 0: index of a view with the same camera whose colours are blended in this view's tile pass (6 channels)
  int fused_alias;   // 1: this view is some view's partner (it owns no tickets)
  float cut_margin;                                            // ... the factor on the proposed depths (gsr_arm_depth_cuts' margin)
  const uint32_t* cut_in; uint32_t* cut_out; uint32_t* redo;   // speculative depth cuts (see gsr_arm_depth_cuts): the cuts this call binned with, the
                                                               //   cuts it proposes for the next frame, the word it sets when a cut tile ran out of list
};
struct GsrRenderViews {
  int V, W, H, gx, T; const uint4* order; uint32_t* queue;
  int no_colour_grad;   // backward: the caller wants no dL/dcolour (records carry their six geometry sums only)
  uint32_t* pc_error_out;  // render_bwd_pc: pinned host word, set to 1 when one of its bounded waits ran out (nullptr: not reported)
  int prio_frac256;     // backward: the longest prio_frac256 / 256 of the busy tickets run at base wave priority 1 (0 = off)
  int track;            // forward: 1 = record the contribution bytes (a backward may follow); 0 = forward-only call
  uint32_t avg_list;    // backward: mean entries per tile over the call's views (from the entry counts / capacities the caller holds): a launch
                        //   heuristic only (sparse scenes keep the 128-entry build whatever their queue length: profiles/r05_autotune.json)
  GsrRenderView v[GSR_MAX_BATCH];
};

// Batch state (V > 1): the structures shared by the views of one call.
// Tile-row binning (gsr_binning.hip): a workgroup of 1024 threads owns GSR_BIN_G consecutive Gaussians of a view.
static inline int gsr_bin_rows(int P) { return ((P > 0 ? P : 1) + GSR_BIN_G - 1) / GSR_BIN_G; }
static inline __host__ __device__ int gsr_bin_stride(int T) { return (T + 3) & ~3; }   // row stride of the matrix: rows stay 16-byte aligned
struct BatchState {
  uint32_t* sums;    // [V][ceil(P/256)] per-preprocess-block entry counts of every view (one D2H copy)
  uint4* order;      // [V*T] {tile, list start, list end, view}: all tiles of the call, longest list first
  uint32_t* queue;   // [16] work-queue heads, see ImageState::queue
  uint32_t* tile_rows;   // [V][rows + 1][T] tile-row binning matrix (nullptr when T > GSR_BIN_MAX_T)
};
static inline size_t gsr_carve_batch(void* base, int V, int32_t P, int32_t H, int32_t W, BatchState* b) {
  size_t off = 0;
  const size_t nblk = ((size_t)(P > 0 ? P : 1) + GSR_BLOCK - 1) / GSR_BLOCK;
  const size_t T = (size_t)((H + GSR_TILE - 1) / GSR_TILE) * ((W + GSR_TILE - 1) / GSR_TILE);
  char* p0 = (char*)base;
  auto take = [&](size_t bytes) { char* p = p0 ? p0 + off : nullptr; off += gsr_align(bytes); return p; };
  b->sums = (uint32_t*)take((size_t)V * nblk * 4);
  b->order = (uint4*)take((size_t)V * (T ? T : 1) * 16);
  b->queue = (uint32_t*)take(64);
  b->tile_rows = nullptr;
  if (T <= GSR_BIN_MAX_T) b->tile_rows = (uint32_t*)take((size_t)V * (gsr_bin_rows(P) + 1) * (size_t)gsr_bin_stride((int)(T ? T : 1)) * 4);
  return off;
}

int gsr_launch_preprocess(const GsrPreViews& tab, const GsrCam& cam, int P, const float* means3D, const float* scales,
                          const float* rotations, const float* opacities, const float* colors_precomp,
                          const float* shs, const float* cov3D_precomp, hipStream_t st);
// Will a call of this tile grid take the tile-row binning?  Decided from (T, device, environment) alone.
bool gsr_rows_path_ok(int T);
int gsr_launch_scan_exclusive(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_out, hipStream_t st);
int gsr_launch_binning(const GsrBinViews& tab, int P, hipStream_t st);
int gsr_launch_tile_order(const GsrBinViews& tab, hipStream_t st);
int gsr_launch_shared_lists(const GsrBinViews& tab, int P, uint32_t D, const uint2* owner_ranges, const uint4* owner_order,
                            const uint32_t* owner_queue, uint2* ranges, uint4* order, uint32_t* queue, hipStream_t st);
int gsr_launch_gather_counts(const GsrBinViews& tab, int P, uint32_t* counts_dev, hipStream_t st);   // counts_dev[v] = offsets_v[P]   // uses only V, T, order, queue, v[].ranges, v[].fused_alias
int gsr_launch_render_fwd(const GsrRenderViews& tab, hipStream_t st);
int gsr_launch_render_bwd(const GsrRenderViews& tab, hipStream_t st);
int gsr_launch_preprocess_bwd(const GsrCam& cam, int P, const float* means3D, const float* scales,
                              const float* rotations, const float* colors_precomp, const float* shs,
                              const float* cov3D_precomp, const int32_t* radii, const GeomState& g,
                              const float4* partials, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors,
                              float* dL_dopacity, float* dL_dscales, float* dL_drotations, float* dL_dcov3D,
                              float* dL_dsh, const uint32_t* bwd_error, hipStream_t st);
// Per-view pointers of the multi-view preprocess backward (passed by value as a kernel argument).
struct GsrBwdView {
  const float *view, *proj;
  const int32_t* radii;
  const uint8_t* used; const uint32_t* tracked;   // GeomState::used, valid when *tracked != 0 (nullptr: not available)
  const uint32_t* offsets;
  const float4* partials;
  float* dL_dmeans2D;
  float* dL_dcolors;      // per-view colour gradient [P,3] (views with their own colours), else nullptr: summed
  float* partner_dL_dmeans2D;   // != nullptr: fused pair backward -- the records carry both views (layout in gsr_render.hip), the
                                // partner's screen-space gradient goes here
  int fused_alias;        // 1: handled by its owner (see partner_dL_dmeans2D): nothing to do for this view
  uint32_t cap;           // entries the record buffer holds: reads are clamped to it, so a backward over a capacity-mode forward
                          // that overflowed (its results are discarded by the caller) never reads past the buffer
  int W, H;
  float tanfovx, tanfovy;
};
#define GSR_QUEUE_BWD_ERROR 8
struct GsrBwdViews {
  int V;
  const uint32_t* bwd_error;   // the call's queue word GSR_QUEUE_BWD_ERROR (or nullptr): != 0 -> dL_dmeans3D is written as NaN
  // raw-parameter mode: chain through the activations applied at the end of the per-Gaussian kernel (all nullptr otherwise)
  const float *raw_rot, *act_op, *act_sc;
  float *d_raw_rot, *d_raw_op, *d_raw_sc;
  GsrBwdView v[GSR_MAX_BATCH];
};
int gsr_launch_preprocess_bwd_views(const GsrBwdViews& vw, int P, float scale_modifier, const float* means3D,
                                    const float* scales, const float* rotations, const float* cov3D_precomp,
                                    float* dL_dmeans3D, float* dL_dcolors, float* dL_dopacity, float* dL_dscales,
                                    float* dL_drotations, float* dL_dcov3D, hipStream_t st);
int gsr_launch_image_loss_fwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, float* fA,
                              float* fC, float* fE, float* block_l1, float* block_ssim, hipStream_t st);
int gsr_launch_image_loss_bwd(const float* win11_host, int C, int H, int W, const float* x, const float* y, const float* fA,
                              const float* fC, const float* fE, const float* grad_loss, int cpi, float w_l1, float w_ssim, float* dx,
                              hipStream_t st);
int gsr_loss_blocks_per_channel(int H, int W);
int gsr_launch_target_moments(const float* win11_host, int channels, int H, int W, const float* target, float* moments, hipStream_t st);
int gsr_launch_views_loss_fwd(const float* win11_host, const gsr_loss_views* v, int H, int W, const float* renders,
                              const float* cam_m, const float* cam_c, float w_l1, float w_ssim, float* fA, float* fC, float* fE,
                              float* partials, float* losses, hipStream_t st);
int gsr_launch_views_loss_bwd(const float* win11_host, const gsr_loss_views* v, int H, int W, const float* renders,
                              const float* cam_m, const float* cam_c, int n_cams, const float* fA, const float* fC,
                              const float* fE, const float* grad_total, float w_l1, float w_ssim, float* d_renders,
                              float* partials, float* d_cam_m, float* d_cam_c, hipStream_t st);
int gsr_launch_rigidity_fwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, float* frames,
                            float* partial, hipStream_t st);   // frames: 16 nfg floats (16-byte aligned) or nullptr
int gsr_launch_rigidity_bwd(int nfg, int K, const float* means3D, const float* rot, const int64_t* fg_idx, const int64_t* nbr,
                            const float* nw, const float* nd, const float* prev_inv, const float* prev_off, const float* g,
                            int gstride, float s1, float s2, float s3, const int32_t* rev_ptr, const int32_t* rev_edge,
                            float* frames, int frames_valid, float* self7, float* edge7, float* d_means3D, float* d_rot, int accumulate,
                            hipStream_t st);   // frames != nullptr: 8-float records in self7 / edge7, else 7
int gsr_rigidity_fwd_blocks(int nfg);
int gsr_launch_activate_fwd(int P, const float* unnorm, const float* logit, const float* logs, float* rot, float* op, float* sc,
                            hipStream_t st);
int gsr_launch_activate_bwd(int P, const float* unnorm, const float* op, const float* sc, const float* d_rot, const float* d_op,
                            const float* d_sc, float* d_unnorm, float* d_logit, float* d_logs, hipStream_t st);
int gsr_shared_terms_point_blocks(int nfg, int nbg);
int gsr_launch_radius_bookkeeping(int V, int step, int P, const int32_t* radii, float* max_2d, uint8_t* seen, hipStream_t st);
int gsr_launch_adam_step(int n_tensors, const gsr_adam_tensor* t, hipStream_t st);
int gsr_launch_shared_terms_fwd(int nfg, int K, int nbg, const float* means3D, const float* rot, const int64_t* fg_idx,
                                const int64_t* bg_idx, const int64_t* nbr, const float* nw, const float* nd, const float* prev_inv,
                                const float* prev_off, const float* init_pts, const float* init_rot, const float* w5,
                                float* partials, float* terms, hipStream_t st);
int gsr_launch_shared_terms_bwd(int P, int nfg, int K, int nbg, const float* means3D, const float* rot, const int64_t* fg_idx,
                                const int64_t* bg_idx, const int64_t* nbr, const float* nw, const float* nd, const float* prev_inv,
                                const float* prev_off, const float* init_pts, const float* init_rot, const float* w5,
                                const float* grad_total, const int32_t* rev_ptr, const int32_t* rev_edge, float* scratch,
                                float* d_means3D, float* d_rot, int flags, hipStream_t st);
int gsr_launch_fps(int N, const float* pos, int npoints, int start, float* mind, long long* out, hipStream_t st);
size_t gsr_fps_scratch_size(int N, int npoints);
int gsr_launch_fit_rotations(int nb, const float* F, const float* n_adj, float* R, int* code, hipStream_t st);
int gsr_launch_fit_bones(int nb, const float* bones, const float* motions, const long long* rel, long long rel_stride, float* R, float* quat,
                         int* code, hipStream_t st);
int gsr_launch_fps_thin(int N, const float* pos, int npoints, int start, float radius, int thin_start, long long* out_idx, long long* thin_idx,
                        int* thin_count, hipStream_t st);
int gsr_launch_lbs(int P, int nb, const float* bones, const float* R, const float* t, const float* bq, const float* xyz,
                   const float* quat, float* out_xyz, float* out_quat, hipStream_t st, const int* nb_valid = nullptr);
int gsr_launch_construct_edges(const float* pos, int n_obj_cap, const int* n_valid, float thr2, int topk, long long dummy, int e_cap,
                               long long* recv, long long* send, int* count, long long* rel, int rel_n, hipStream_t st, long long* row_start = nullptr);
int gsr_launch_rollout_head(int n_track, int n_his, int nb, int n_cap, int A, int with_state, const float* hist, const long long* idx1,
                            const long long* thin, const float* eef_hist, const float* eef_next, const float* attrs, const float* inst,
                            float* bones_last, float* states_last, float* state_t, float* act, float* p_in, float* nodes, hipStream_t st);
int gsr_launch_rollout_motion(int nb, int n_his, float clampv, const float* state_t, const float* pred_motion, const int* cnt, float* packet,
                              hipStream_t st);
int gsr_launch_rollout_tail(int n_track, int n_his, int nb, const float* all_pos, const long long* track, float* pos_track, float* hist,
                            float* eef_hist, const float* eef_next, const float* pred_in, const int* cnt, const int* code, float* pred_out,
                            int* n_valid_out, long long* bad, hipStream_t st);
int gsr_launch_mark_visible(const float* view, int P, const float* means3D, uint8_t* present, hipStream_t st);
int gsr_run_selftest(hipStream_t st);
int gsr_debug_fwd_timing(unsigned long long* out16);

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__
__device__ __forceinline__ int gsr_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t gsr_lanemask_lt() { return (1ull << gsr_lane()) - 1ull; }

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float gsr_dpp_add(float v) {
  // v + (v read through the DPP crossbar); lanes whose source is masked read 0.
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
// Sum over the 64 lanes of a wave; the total is valid in lane 63 (all of row 3).
// quad_perm[1,0,3,2] -> quad_perm[2,3,0,1] -> row_half_mirror -> row_mirror (each row of 16 now
// holds its row sum) -> row_bcast:15 into rows 1,3 -> row_bcast:31 into rows 2,3.
__device__ __forceinline__ float gsr_wave_sum_to_lane63(float v) {
  v = gsr_dpp_add<0xB1>(v);
  v = gsr_dpp_add<0x4E>(v);
  v = gsr_dpp_add<0x141>(v);
  v = gsr_dpp_add<0x140>(v);
  v = gsr_dpp_add<0x142, 0xA>(v);
  v = gsr_dpp_add<0x143, 0xC>(v);
  return v;
}
// Nine wave sums at once (the backward's per-entry gradient partials), totals valid in lane 63.
// Written as ONE stage-major block of 54 fused v_add_f32_dpp: hipcc otherwise SLP-packs the adds into
// v_pk_add_f32, which blocks the mov_dpp+add fusion (2.5 instructions per step instead of 1), and cannot
// fuse the row_bcast steps at all.  Stage-major order keeps every DPP read 9 instructions behind the
// VALU write of its source (the ISA wants >= 2 wait states); the leading s_nop covers the first stage.
// Rows not enabled by row_mask keep their value because dst == src1.
#define GSR_DPP9(CTRL)                              \
  "v_add_f32_dpp %0, %0, %0 " CTRL "\n\t"           \
  "v_add_f32_dpp %1, %1, %1 " CTRL "\n\t"           \
  "v_add_f32_dpp %2, %2, %2 " CTRL "\n\t"           \
  "v_add_f32_dpp %3, %3, %3 " CTRL "\n\t"           \
  "v_add_f32_dpp %4, %4, %4 " CTRL "\n\t"           \
  "v_add_f32_dpp %5, %5, %5 " CTRL "\n\t"           \
  "v_add_f32_dpp %6, %6, %6 " CTRL "\n\t"           \
  "v_add_f32_dpp %7, %7, %7 " CTRL "\n\t"           \
  "v_add_f32_dpp %8, %8, %8 " CTRL "\n\t"
__device__ __forceinline__ void gsr_wave_sum9_to_lane63(float& v0, float& v1, float& v2, float& v3, float& v4,
                                                        float& v5, float& v6, float& v7, float& v8) {
  asm volatile(
      "s_nop 1\n\t"
      GSR_DPP9("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
      GSR_DPP9("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1")
      GSR_DPP9("row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
      GSR_DPP9("row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
      GSR_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf")
      GSR_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 0"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8));
}

// Packed ("transposition") form of the nine wave sums: 33 instructions instead of 54.  At butterfly level
// xor-1 the nine values are folded pairwise by lane parity (each lane keeps one value of a pair, sends the
// other to its partner: 2 selects + 1 fused DPP add per pair), so the register count goes 9 -> 5 -> 3 -> 2 -> 1
// across the xor-1 / xor-2 / xor-4 / xor-8 levels; the two row levels then run on ONE register (33 instructions).
// Result z (in every row): lane with (lane & 15) = i < 8 holds the wave total of v_i, lanes with bit 3 set the total of v8.
template <int CTRL>
__device__ __forceinline__ float gsr_dpp_get(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float gsr_swz_xor4(float x) {  // ds_swizzle bit mode: src lane = lane ^ 4
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x101F));
}
// The two row levels (xor 16, xor 32) of a packed reduction: every lane adds its partners in the other rows.
// PERM = false: ds_swizzle + ds_bpermute -- two LDS-crossbar round trips, 2 VALU issues.  PERM = true: v_permlane16/32_swap on
// two copies -- no LDS, 6 VALU issues.  Measured (same box): with five workgroups per CU and a long ticket queue the kernel is
// short of issue slots and the crossbar form wins (render_bwd 452 vs 459 us, 8 views); with four per CU and about one tile per
// workgroup (one or two views) the serial chain of a tile counts and the swap form wins (103 -> 96.5 us).
template <bool PERM>
__device__ __forceinline__ float gsr_rows_sum(float z) {
  if (PERM) {
    float a = z, b = z;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: rows {0,0,2,2} of z, b: rows {1,1,3,3}
    z = a + b;
    a = z; b = z;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a: low half twice, b: high half twice
    return a + b;
  }
  z += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(z), 0x401F));
  z += __shfl_xor(z, 32, 64);
  return z;
}
template <bool PERM = false>
__device__ __forceinline__ float gsr_wave_sum9_packed(float v0, float v1, float v2, float v3, float v4, float v5,
                                                      float v6, float v7, float v8) {
  const int lane = gsr_lane();
  const bool b0 = lane & 1, b1 = lane & 2;
  // A level of the packed butterfly keeps "x_self + x_partner" of value a in the lanes whose selector bit is clear and of value
  // b in the others.  Where the selector is a DPP write-mask group -- lane bits 2 and 3 are the four-lane "banks" of a 16-lane
  // row -- the select folds into the add: one DPP add per half with bank_mask, two issues per pair instead of two v_cndmask +
  // one DPP add.  So the WIDE level (9 -> 5, four pairs) runs on bit 2 (row_shl:4 into banks 0/2, row_shr:4 into banks 1/3) and
  // the last in-row level on bit 3 (row_ror:8); bits 0 and 1 (no write mask inside a quad) take the two narrow levels with
  // selects.  24 VALU issues instead of 34, the same final layout.  The masked adds are inline asm (the compiler has no pattern
  // for them and its hazard recogniser does not look inside): the leading s_nop covers a VALU write of a source just before
  // the block (2 wait states) or an EXEC write (5); inside, no source was written by the two preceding instructions.
  float r01, r23, r45, r67;   // bit 2 clear: totals over xor 4 of v0 / v1 / v2 / v3, set: of v4 / v5 / v6 / v7
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(r01), "=&v"(r23), "=&v"(r45), "=&v"(r67)
      : "v"(v0), "v"(v4), "v"(v1), "v"(v5), "v"(v2), "v"(v6), "v"(v3), "v"(v7));
  const float r8 = v8 + gsr_swz_xor4(v8);
  // xor 1 (quad_perm [1,0,3,2]): 5 -> 3   (bit 0 picks v1 / v5 over v0 / v4, v3 / v7 over v2 / v6)
  const float q03 = (b0 ? r23 : r01) + gsr_dpp_get<0xB1>(b0 ? r01 : r23);
  const float q47 = (b0 ? r67 : r45) + gsr_dpp_get<0xB1>(b0 ? r45 : r67);
  const float q8 = r8 + gsr_dpp_get<0xB1>(r8);
  // xor 2 (quad_perm [2,3,0,1]): 3 -> 2   (lane & 7 now indexes v0..v7)
  const float p07 = (b1 ? q47 : q03) + gsr_dpp_get<0x4E>(b1 ? q03 : q47);
  const float p8 = q8 + gsr_dpp_get<0x4E>(q8);
  // xor 8 (row_ror:8): 2 -> 1   (lanes with bit 3 clear: v_(lane & 7); bit 3 set: v8)
  float z;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
      : "=&v"(z) : "v"(p07), "v"(p8));
  // rows: the packed layout differs per lane, so the row levels must be lane-wise exchanges (row_bcast would
  // broadcast a single lane), see gsr_rows_sum.  Every lane ends up with the total of "its" value: lane & 15 in 0..7 ->
  // v_(lane & 7), lane & 8 set -> v8.
  return gsr_rows_sum<PERM>(z);
}


// Eight values (fused pair backward without colour gradients): 8 -> 4 -> 2 -> 1 registers, 20 VALU issues (see above).
// Result z (in every lane): lane with (lane & 7) = i holds the wave total of v_i.
__device__ __forceinline__ float gsr_wave_sum8_packed(float v0, float v1, float v2, float v3, float v4, float v5,
                                                      float v6, float v7) {
  const int lane = gsr_lane();
  const bool b0 = lane & 1, b1 = lane & 2;
  float r01, r23, r45, r67;
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %1, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %3, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(r01), "=&v"(r23), "=&v"(r45), "=&v"(r67)
      : "v"(v0), "v"(v4), "v"(v1), "v"(v5), "v"(v2), "v"(v6), "v"(v3), "v"(v7));
  const float q03 = (b0 ? r23 : r01) + gsr_dpp_get<0xB1>(b0 ? r01 : r23);
  const float q47 = (b0 ? r67 : r45) + gsr_dpp_get<0xB1>(b0 ? r45 : r67);
  float z = (b1 ? q47 : q03) + gsr_dpp_get<0x4E>(b1 ? q03 : q47);
  z += gsr_dpp_get<0x128>(z);                                                        // xor 8 (row_ror:8)
  return gsr_rows_sum<false>(z);                                                     // xor 16, xor 32
}

// Six values (plain backward when no colour gradient is wanted -- rgb_colors is frozen throughout the reference's training,
// /root/reference/src/tracking/train_utils.py:133,155): 6 -> 3 -> 2 -> 1 registers, 15 VALU issues.
// Result z: of the lanes with bit 1 clear, an even lane holds v_(2 * bit 3 + bit 2), an odd lane with bit 3 clear v_(4 + bit 2)
// (gsr_sum6_slot gives a lane's value index, -1 for the lanes that hold a duplicate).
__device__ __forceinline__ int gsr_sum6_slot(int lane) {
  if (lane & 2) return -1;
  if (!(lane & 1)) return 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
  return (lane & 8) ? -1 : 4 + ((lane >> 2) & 1);
}
__device__ __forceinline__ float gsr_wave_sum6_packed(float v0, float v1, float v2, float v3, float v4, float v5) {
  const bool b0 = gsr_lane() & 1;
  float r01, r23, r45, q03, q45;
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"      // bit 2 (xor 4): 6 -> 3
      "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %2, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %2, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(r01), "=&v"(r23), "=&v"(r45)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5));
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"      // bit 3 (xor 8): 3 -> 2
      "v_add_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"      // (inside the block: r45 comes from the asm above)
      "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc"
      : "=&v"(q03), "=&v"(q45) : "v"(r01), "v"(r23), "v"(r45));
  float z = (b0 ? q45 : q03) + gsr_dpp_get<0xB1>(b0 ? q03 : q45);               // bit 0 (xor 1): 2 -> 1
  z += gsr_dpp_get<0x4E>(z);                                                     // xor 2
  return gsr_rows_sum<false>(z);                                                 // xor 16, xor 32
}

// ---- parameter activations of the tracking step (/root/reference/src/tracking/helpers.py:36-45): rotation = normalize(unnorm),
// opacity = sigmoid(logit), scale = exp(log_scale).  One definition for the stand-alone kernels (gsr_step.hip) and for the
// preprocess kernels that apply them inline (raw-parameter mode); the sums are spelled with fmaf so that files compiled with and
// without contraction produce the same bits.
__device__ __forceinline__ float gsr_quat_norm(float4 q) { return sqrtf(fmaf(q.w, q.w, fmaf(q.z, q.z, fmaf(q.y, q.y, q.x * q.x)))); }
__device__ __forceinline__ float4 gsr_act_rotation(float4 q) {
  const float d = fmaxf(gsr_quat_norm(q), 1e-12f);   // torch.nn.functional.normalize: x / max(|x|, eps)
  return make_float4(q.x / d, q.y / d, q.z / d, q.w / d);
}
__device__ __forceinline__ float gsr_act_opacity(float logit) { return 1.0f / (1.0f + expf(-logit)); }
__device__ __forceinline__ float gsr_act_scale(float logs) { return expf(logs); }
__device__ __forceinline__ float4 gsr_act_rotation_bwd(float4 q, float4 dr) {
  const float n = gsr_quat_norm(q);
  if (n > 1e-12f) {
    const float inv = 1.0f / n;
    const float rx = q.x * inv, ry = q.y * inv, rz = q.z * inv, rw = q.w * inv;
    const float dot = fmaf(rw, dr.w, fmaf(rz, dr.z, fmaf(ry, dr.y, rx * dr.x)));
    return make_float4(fmaf(-rx, dot, dr.x) * inv, fmaf(-ry, dot, dr.y) * inv, fmaf(-rz, dot, dr.z) * inv, fmaf(-rw, dot, dr.w) * inv);   // (spelled out: see above)
  }
  return make_float4(dr.x * 1e12f, dr.y * 1e12f, dr.z * 1e12f, dr.w * 1e12f);   // clamped denominator: a constant
}

// exp(x) for x <= 0: v_exp_f32 on x * log2(e) -- two VALU issues.  Relative error ~ |x| * 6e-8 + 1 ulp (|x| <= 5.6 wherever
// alpha >= 1/255).  A compensated form (exact product error folded back in, ~1-2 ulp, 7 issues) was used until the parity
// margins were measured for both: gradient errors against the oracle are 1e-7 .. 4e-6 of the tensor maximum with either
// (other roundings dominate; tolerance 1e-4), while the blend kernels, which sit at the VALU issue limit, run 3-4 % faster
// with this one.  Forward and backward use the same function, so they take the same alpha >= 1/255 decisions.
__device__ __forceinline__ float gsr_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f); }
struct GsrF3 { float x, y, z; };   // 12 bytes, 4-byte aligned: one global_load/store_dwordx3
__device__ __forceinline__ void gsr_store_partial(float4* base, size_t e, float4 r0, float4 r1, float r2x) {
  GsrF3* p = reinterpret_cast<GsrF3*>(reinterpret_cast<float*>(base) + e * GSR_PARTIAL_FLOATS);
  p[0] = GsrF3{r0.x, r0.y, r0.z}; p[1] = GsrF3{r0.w, r1.x, r1.y}; p[2] = GsrF3{r1.z, r1.w, r2x};
}
// the geometry part alone (no colour gradient wanted): the record's first 24 bytes
__device__ __forceinline__ void gsr_store_partial6(float4* base, size_t e, float4 r0, float r1x, float r1y) {
  GsrF3* p = reinterpret_cast<GsrF3*>(reinterpret_cast<float*>(base) + e * GSR_PARTIAL_FLOATS);
  p[0] = GsrF3{r0.x, r0.y, r0.z}; p[1] = GsrF3{r0.w, r1x, r1y};
}
__device__ __forceinline__ void gsr_load_partial6(const float4* base, size_t e, float4& r0, float& r1x, float& r1y) {
  const GsrF3* p = reinterpret_cast<const GsrF3*>(reinterpret_cast<const float*>(base) + e * GSR_PARTIAL_FLOATS);
  const GsrF3 a = p[0], b = p[1];
  r0 = make_float4(a.x, a.y, a.z, b.x); r1x = b.y; r1y = b.z;
}
__device__ __forceinline__ void gsr_load_partial(const float4* base, size_t e, float4& r0, float4& r1, float& r2x) {
  const GsrF3* p = reinterpret_cast<const GsrF3*>(reinterpret_cast<const float*>(base) + e * GSR_PARTIAL_FLOATS);
  const GsrF3 a = p[0], b = p[1], c = p[2];
  r0 = make_float4(a.x, a.y, a.z, b.x); r1 = make_float4(b.y, b.z, c.x, c.y); r2x = c.z;
}
// q(d) = A dx^2 + 2 B dx dy + C dy^2 restricted to a vertical (dx fixed) or horizontal (dy fixed) edge,
// minimised over the edge's extent (used for the exact Gaussian-vs-rectangle culling tests).
__device__ __forceinline__ float qmin_on_vertical_edge(float A, float B, float C, float invC, float dx, float dylo, float dyhi) {
  const float dy = fminf(fmaxf(-B * dx * invC, dylo), dyhi);
  return A * dx * dx + (2.0f * B * dx + C * dy) * dy;
}
__device__ __forceinline__ float qmin_on_horizontal_edge(float A, float B, float C, float invA, float dy, float dxlo, float dxhi) {
  const float dx = fminf(fmaxf(-B * dy * invA, dxlo), dxhi);
  return C * dy * dy + (2.0f * B * dy + A * dx) * dx;
}
// Minimum of the (convex) form over the rectangle [dxlo, dxhi] x [dylo, dyhi] (offsets from the mean) when the mean is NOT inside it.
// Only the edges that FACE the mean can hold it (round 4; four edges were evaluated before): take any point p of the rectangle -- the
// segment from the mean to p enters the rectangle through the nearer vertical edge (when the mean lies outside the x-range) or the
// nearer horizontal edge (outside the y-range), at a point p' of that edge, and the form grows along the segment away from the mean,
// so q(p') <= q(p).  One or two edge minimisations instead of four: 70 -> ~42 VALU per tested rectangle (the per-tile loop of
// preprocess_fwd is half of that kernel; the per-quad tests of the blend kernels' staging lanes).
__device__ __forceinline__ float gsr_qmin_facing_edges(float A, float B, float C, float invA, float invC, float dxlo, float dxhi, float dylo, float dyhi) {
  const bool in_x = dxlo <= 0.0f && dxhi >= 0.0f, in_y = dylo <= 0.0f && dyhi >= 0.0f;
  const float dxe = dxlo > 0.0f ? dxlo : dxhi, dye = dylo > 0.0f ? dylo : dyhi;   // the nearer edge of each pair
  const float qv = qmin_on_vertical_edge(A, B, C, invC, dxe, dylo, dyhi);
  const float qh = qmin_on_horizontal_edge(A, B, C, invA, dye, dxlo, dxhi);
  const float inf = __builtin_inff();
  return fminf(in_x ? inf : qv, in_y ? inf : qh);
}
// Can alpha reach 1/255 anywhere on the pixel rectangle [x0, x1] x [y0, y1]?  Exact minimum of the (convex) quadratic form
// over the rectangle -- 0 when the mean is inside, else the least minimum over the facing edges -- against tau2 = 2 ln(255 o) + 0.04:
// conservative.
__device__ __forceinline__ bool gsr_rect_reachable(float mx, float my, float A, float B, float C, float invA, float invC, float tau2,
                                                   int x0, int y0, int x1, int y1) {
  const float dxlo = (float)x0 - mx, dxhi = (float)x1 - mx, dylo = (float)y0 - my, dyhi = (float)y1 - my;
  if (dxlo <= 0.0f && dxhi >= 0.0f && dylo <= 0.0f && dyhi >= 0.0f) return true;
  return gsr_qmin_facing_edges(A, B, C, invA, invC, dxlo, dxhi, dylo, dyhi) <= tau2;
}
// Tile set of a Gaussian: its (tight) tile rect, and -- for rects of at most 32 tiles -- a bit per tile of the rect
// (row-major) telling whether the tile is in the set.  Larger rects are taken whole.
__device__ __forceinline__ uint32_t gsr_tile_rank(uint32_t mask, uint32_t area, uint32_t r) {   // rank of rect tile r in the set
  return area <= 32u ? (uint32_t)__popc(mask & ((1u << r) - 1u)) : r;
}
// ---- tile-row binning helpers (gsr_binning.hip; the counting form of preprocess_fwd)
#define BIN_THREADS 1024
#define BIN_PER_THREAD (GSR_BIN_G / BIN_THREADS)
#ifndef BIN_DIRECT_ROWS
#define BIN_DIRECT_ROWS 32
#endif
#ifndef BIN_ROW_CHUNK
#define BIN_ROW_CHUNK 16
#endif
#define BIN_BIG_AREA 64        // rects with more tiles are walked by a whole wave, not by their Gaussian's lane
#define BIN_BIG_MAX 1024       // such Gaussians parked per workgroup (LDS); beyond it their lanes walk them after all

struct BinGauss { uint32_t minx, miny, w, area, mask; float rw; };
__device__ __forceinline__ BinGauss bin_gauss(uint2 r, uint32_t mask) {
  BinGauss b;
  b.minx = r.x & 0xffffu; b.miny = r.x >> 16;
  const uint32_t maxx = r.y & 0xffffu, maxy = r.y >> 16;
  b.w = maxx - b.minx; b.area = b.w * (maxy - b.miny); b.mask = mask;
  b.rw = __builtin_amdgcn_rcpf((float)b.w);
  return b;
}
// Tile id of the k-th tile (row-major) of a Gaussian's rect.  k / w without an integer division (~40 VALU issues each on this
// part, and the walk below is the whole cost of the count / emit kernels): (k + 0.5) * (1 / w) in fp32, off by less than 1e-3 for
// every k the kernels see (k < 2^14, so k + 0.5 is exact), while the true quotient's fractional part stays >= 0.5 / w away from an integer.
__device__ __forceinline__ uint32_t bin_tile_of(const BinGauss& b, uint32_t k, int gx) {
  const uint32_t ky = (uint32_t)(((float)k + 0.5f) * b.rw);
  return (b.miny + ky) * (uint32_t)gx + b.minx + (k - ky * b.w);
}
// f(tile id) for every tile of a Gaussian's set (small rects: the set bits of its mask, row-major; larger rects: all of it)
// (Round 4: every lane starting its walk at another tile of its set -- rotated by a lane-dependent amount, so that the LDS atomics of a
// step spread over more counters -- measured no faster: bin_count 66 -> 69 us, bin_emit 113 -> 115 us per configs[4] frame in Morton order;
// same-address conflicts are not what these kernels wait for.  Not kept.)
template <typename F>
__device__ __forceinline__ void bin_for_tiles(const BinGauss& b, int gx, F f) {
  if (b.area <= 32u) {
    uint32_t m = b.mask;
    while (m) {
      const uint32_t k = (uint32_t)__ffs((int)m) - 1u;
      m &= m - 1u;
      f(bin_tile_of(b, k, gx));
    }
  } else {
    for (uint32_t k = 0; k < b.area; ++k) f(bin_tile_of(b, k, gx));
  }
}

// Reference (slow, LDS-crossbar) version used by the self-test.
__device__ __forceinline__ float gsr_wave_sum_shfl(float v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
#endif
