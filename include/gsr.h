/*
 * gsr.h -- C-ABI of libgsr_hip.so: the MI355X (gfx950) differentiable 3D-Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of robo-alex/gs-dynamics: the rasterizer behind
 * `diff_gaussian_rasterization.GaussianRasterizer / GaussianRasterizationSettings`.  In the reference
 * that package is an un-vendored third-party CUDA extension (/root/reference/README.md:28-32) whose
 * native binding exposes three functions to its own Python wrapper -- `rasterize_gaussians`,
 * `rasterize_gaussians_backward`, `mark_visible` -- none of which is visible to any reference caller.
 * The reference-visible contract is the Python API used at
 *     /root/reference/src/tracking/helpers.py:20-32      (settings record, 11 fields)
 *     /root/reference/src/tracking/train_utils.py:174-192 (forward + autograd backward, twice per step)
 *     /root/reference/src/render/renderer.py:18-23       (forward only, no_grad)
 * The entry points below are what a ctypes / pybind binding for that package binds instead
 * (see INTEGRATION.md); plain pointers and sizes, no torch types.
 *
 * Conventions
 *   - every `const float*` / `void*` data pointer is a DEVICE pointer unless the name ends in _host;
 *   - all float tensors are fp32, row-major, contiguous; radii is int32; images are CHW;
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); all work is enqueued on it;
 *   - functions return 0 on success, non-zero on error (message via gsr_last_error());
 *   - no hidden allocations on the data path: the caller owns every buffer (sizes from gsr_*_bytes);
 *     the only persistent state is the stream/event pool of the *_batch entry points.
 */
#ifndef GSR_H_
#define GSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 124 /* 0.1.24 -- the history of the ABI is in CHANGELOG.md ("ABI history") */
#define GSR_TILE 16     /* tiles are 16x16 pixels, as in the reference extension */

/* Mirror of GaussianRasterizationSettings (/root/reference/src/tracking/helpers.py:20-32).
 * bg / viewmatrix / projmatrix / campos stay device tensors exactly as the reference passes them
 * (viewmatrix = w2c transposed, [1,4,4] or [4,4]: 16 contiguous floats). */
typedef struct gsr_settings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float scale_modifier;
  int32_t sh_degree;   /* active SH degree (0..3) */
  int32_t sh_coeffs;   /* M: coefficients per Gaussian in `shs` ([P,M,3]); 0 when colors_precomp is used */
  int32_t prefiltered; /* accepted for API parity; reference call sites always pass False */
  const float* bg;         /* [3]  device */
  const float* viewmatrix; /* [16] device */
  const float* projmatrix; /* [16] device */
  const float* campos;     /* [3]  device */
} gsr_settings;

/* ---- buffer sizes (bytes).  The three opaque state buffers play the role of the reference
 * extension's geomBuffer / binningBuffer / imgBuffer, but are sized by the caller up front. */
size_t gsr_geom_bytes(int32_t P);
/* The image state BEGINS with final_T[image_height * image_width] (float32): the per-pixel transmittance after the last blended
 * entry, 1 where nothing was blended -- upstream's accum_alpha.  It is the one part of a state a caller may read (after any
 * forward on that state): 1 - final_T is the accumulated alpha, i.e. every channel of a render with colours = 1 on a black
 * background, which is how gsdyn.render produces predict.py's mask render (/root/reference/src/predict.py:119-121) without a
 * second blend pass.  The rest of the three states is opaque. */
size_t gsr_image_bytes(int32_t image_height, int32_t image_width);
size_t gsr_binning_bytes(uint32_t num_rendered, int32_t image_height, int32_t image_width);
size_t gsr_backward_scratch_bytes(int32_t P, uint32_t num_rendered);

/* ---- forward, stage 1  (replaces the first half of `rasterize_gaussians`: preprocess + offsets scan)
 * Per Gaussian: frustum cull, projection, 3D->2D covariance, conic, radius, tile rect, colour
 * (colors_precomp or SH->RGB).  Writes radii[P] and the geometry state, and returns the number of
 * (Gaussian,tile) duplicates in *num_rendered_host (this call synchronises `stream` once to do so).
 * Exactly one of {colors_precomp, shs} and exactly one of {scales+rotations, cov3D_precomp} non-NULL. */
int gsr_forward_preprocess(const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                           const float* rotations, const float* opacities, const float* colors_precomp,
                           const float* shs, const float* cov3D_precomp, void* geom_state, int32_t* radii,
                           uint32_t* num_rendered_host, void* stream);

/* ---- forward, stage 2  (replaces the second half of `rasterize_gaussians`: duplicateWithKeys,
 * sort, identifyTileRanges, render).  out_color[3,H,W], out_depth[1,H,W].  `binning_state` must hold
 * gsr_binning_bytes(num_rendered,...) bytes, `image_state` gsr_image_bytes(...). */
int gsr_forward_render(const gsr_settings* s, int32_t P, uint32_t num_rendered, const void* geom_state,
                       void* binning_state, void* image_state, float* out_color, float* out_depth,
                       void* stream);

/* ---- tile-list reuse between two single-view forwards (the reference renders every camera twice with the same geometry: colours,
 * then segmentation colours, /root/reference/src/tracking/train_utils.py:178,192; colours, then an all-ones mask,
 * /root/reference/src/predict.py:115-123 -- and hands the second call FRESH copies of the geometry tensors, so tensor identity cannot
 * key a cache).  gsr_forward_preprocess_same is gsr_forward_preprocess that also COMPARES its outputs, on the device and bit for bit, with
 * the geometry state of an earlier forward of the same P and image size (prev_geom_state; it must outlive the call): per Gaussian the entry
 * count, tile rect, tile mask, depth bits and -- because the forward leaves the backward's per-quad contribution bytes in the binning
 * state, next to the lists, so a sharer rewrites the owner's bytes and they must be the same bytes -- 2D mean, conic and opacity; colours
 * may differ.  *same_host = 1: everything equal (then num_rendered is equal too) -- the second forward may call gsr_forward_render_shared
 * with the first call's binning and image states instead of gsr_forward_render: no duplicates are emitted or sorted, its own image state
 * receives a copy of the owner's ranges and tile order, and gsr_backward takes (own geom, OWNER's binning, own image) as usual.
 * 0: something differs, or no comparison was made (prev_geom_state NULL, P > 512 Ki).  The verdict rides in the copy that brings the entry
 * count back: no extra synchronisation; the comparison reads 84 bytes per Gaussian (~3 us at 100 k).  (ABI <= 119 compared a 64-bit
 * fingerprint: equal "up to a 2^-64 coincidence"; the bar for integer work is bit-exact.)
 * Results are bit-identical to gsr_forward_render. */
/* gsr_forward_render_ex / gsr_forward_render_shared_ex (ABI 119): the same with `flags` -- GSR_FORWARD_ONLY (defined below): the caller
 * will not run gsr_backward on these states, so the blend skips recording what only a backward reads (the per-entry contribution bytes
 * and the per-Gaussian used flags: 7 % of a forward).  The torch layer passes it when no input of the call requires a gradient. */
int gsr_forward_render_ex(const gsr_settings* s, int32_t P, uint32_t num_rendered, const void* geom_state, void* binning_state,
                          void* image_state, float* out_color, float* out_depth, uint32_t flags, void* stream);
int gsr_forward_render_shared_ex(const gsr_settings* s, int32_t P, uint32_t num_rendered, void* geom_state, void* owner_binning_state,
                                 const void* owner_image_state, void* image_state, float* out_color, float* out_depth, uint32_t flags,
                                 void* stream);
int gsr_forward_preprocess_same(const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                const float* rotations, const float* opacities, const float* colors_precomp,
                                const float* shs, const float* cov3D_precomp, void* geom_state, int32_t* radii,
                                uint32_t* num_rendered_host, const void* prev_geom_state, int32_t* same_host, void* stream);
int gsr_forward_render_shared(const gsr_settings* s, int32_t P, uint32_t num_rendered, void* geom_state,
                              void* owner_binning_state, const void* owner_image_state, void* image_state, float* out_color,
                              float* out_depth, void* stream);

/* ---- the single-view forward WITHOUT a host wait inside (ABI 121; upstream reads num_rendered back in the middle of its forward,
 * rasterizer_impl.cu -- reached from /root/reference/src/tracking/train_utils.py:178 and src/render/renderer.py:22 -- and the GPU idles
 * while the host sizes the binning buffers and launches the rest: ~20 us of a 98 us forward at 50 k Gaussians / 800 x 800).
 * Both stages in one call: binning_state holds gsr_binning_bytes of `capacity_entries` (the caller's estimate: the previous call's
 * count with some slack), every kernel clamps to it, and the true count arrives in *count_pinned -- PINNED host memory that the caller
 * pre-set to -1 -- by a system-scope store of the tile-order kernel: wait with gsr_wait_counts (no stream synchronisation) once all the
 * call's launches are queued.  count <= capacity_entries: the outputs and states are valid, and `capacity_entries` is the num_rendered
 * that gsr_backward, gsr_forward_render_shared and gsr_backward_scratch_bytes must be given for these states (it fixed their layout).
 * count > capacity_entries: nothing of the call may be used -- repeat with gsr_forward_preprocess + gsr_forward_render.
 * Earlier still (P <= 512 Ki): block_words_pinned -- pinned host memory, 8-BYTE ALIGNED, 2 * ceil(P / 256) words, the ODD words pre-set to 0xffffffff by the
 * caller.  Every preprocess block stores {differs, its entry count} there with one 8-byte store; gsr_wait_block_counts polls the words and
 * returns their sum as soon as the preprocess kernel's blocks are through -- typically before the host has finished queueing the rest of
 * the call -- so the host never waits and the GPU never idles.  count_pinned may then be NULL.  With prev_geom_state as well the blocks
 * COMPARE as in gsr_forward_preprocess_same: *any_differs = 0 after the wait means every Gaussian equals the compared state.
 * flags: GSR_FORWARD_ONLY. */
int gsr_forward_capacity(const gsr_settings* s, int32_t P, const float* means3D, const float* scales, const float* rotations,
                         const float* opacities, const float* colors_precomp, const float* shs, const float* cov3D_precomp,
                         void* geom_state, int32_t* radii, void* binning_state, uint32_t capacity_entries, void* image_state,
                         float* out_color, float* out_depth, const void* prev_geom_state, uint32_t* block_words_pinned,
                         int32_t* count_pinned, uint32_t flags, void* stream);
/* -> the sum of the blocks' entry counts (>= 0) once all nblk odd words differ from 0xffffffff; -1 after timeout_us.  Polls as gsr_wait_counts. */
int64_t gsr_wait_block_counts(const volatile uint32_t* words, int32_t nblk, int64_t spin_us, int64_t timeout_us, int32_t* any_differs);

/* ---- backward  (replaces `rasterize_gaussians_backward`).
 * dL_dcolor[3,H,W] in; gradients out (every output is fully written, no pre-zeroing needed):
 *   dL_dmeans3D[P,3] dL_dmeans2D[P,3] (x,y = dL/d(NDC), z = 0) dL_dcolors[P,3] dL_dopacity[P]
 *   dL_dscales[P,3] dL_drotations[P,4] dL_dcov3D[P,6] dL_dsh[P,M,3]
 * Pointers for unused outputs (dL_dsh without shs; dL_dcolors with shs) may be NULL.
 * dL_dcolors == NULL with colors_precomp means "no colour gradient wanted" (rgb_colors is frozen throughout the reference's
 * training, /root/reference/src/tracking/train_utils.py:133,155): the blend backward then keeps six sums per list entry instead
 * of nine.  Every other gradient is the same up to the rounding of a different reduction tree.
 * Incoming gradients for radii and depth do not exist in this ABI: they are ignored by contract
 * (no reference call site differentiates them, /root/reference/src/tracking/train_utils.py:178,192). */
int gsr_backward(const gsr_settings* s, int32_t P, uint32_t num_rendered, const float* means3D,
                 const float* scales, const float* rotations, const float* colors_precomp, const float* shs,
                 const float* cov3D_precomp, const int32_t* radii, const void* geom_state,
                 const void* binning_state, const void* image_state, const float* dL_dcolor, void* scratch,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dsh, void* stream);

/* ---- multi-view batch (new design, no counterpart in the reference: its training loop renders one view per
 * optimiser step, /root/reference/src/tracking/train_gs.py:25-39).  The V views of a sharded step share the
 * Gaussian inputs and the image size.  Every stage of a batch call is ONE kernel launch covering all views (the
 * kernels take per-view pointer tables as arguments; all tiles of all views share one longest-first work queue),
 * enqueued on `stream`; stage 1 synchronises ONCE for all V duplicate counts.
 * Array arguments have V entries (host arrays of device pointers).  `batch_state` is one more caller-allocated
 * device buffer of gsr_batch_state_bytes(V, P, H, W) bytes holding what the views share (per-block entry counts,
 * the combined tile order, queue heads); keep it from the forward to the backward like the other states.
 * Per-view colours (row N1 of SURVEY.md section 8f: the colour and the segmentation render of get_loss share the
 * geometry, /root/reference/src/tracking/train_utils.py:174-192): pass `colors_views` ([V] device pointers to [P,3],
 * with colors_precomp = shs = NULL) and every "view" blends its own colour array; the backward then writes one colour
 * gradient per view into `dL_dcolors_views` instead of the sum into `dL_dcolors`.
 * `geometry_of` ([V] or NULL): geometry_of[v] = u <= v declares that view v has the SAME camera as the earlier view u
 * (geometry_of[u] = u) and differs only in its colours.  View v then uses u's tile lists: nothing is emitted, sorted or
 * ranged for it (binning_states[v] may be NULL), and its entry count equals u's. */
#define GSR_MAX_BATCH 16
size_t gsr_batch_state_bytes(int32_t V, int32_t P, int32_t image_height, int32_t image_width);
int gsr_forward_preprocess_batch(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                 const float* rotations, const float* opacities, const float* colors_precomp,
                                 const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                                 void* const* geom_states,
                                 int32_t* const* radii, void* batch_state, uint32_t* num_rendered_host, void* stream);
/* ABI 123 -- speculative depth cuts for FORWARD-ONLY frame sequences (predict.py renders the same cameras frame after frame, and in a dense
 * scene ~90 % of the (Gaussian, tile) pairs lie behind the depth at which every pixel of their tile has saturated: they are binned, sorted
 * and never blended).  gsr_arm_depth_cuts arms the NEXT gsr_forward_batch / gsr_forward_render_batch of this host thread (flags must carry
 * GSR_FORWARD_ONLY; V views without shared lists; tile grids the tile-row binning serves):
 *   cut_in[v]   [T] uint32 per tile, device: depth bits (a positive float's bits order like the float); a pair whose depth is LARGER than its
 *               tile's cut is neither counted nor emitted.  0x7f800000 (+inf) = no cut for the tile.  cut_in == NULL or cut_in[v] == NULL: no cuts.
 *   cut_out[v]  [T] written by the blend: the tile's proposal for the next frame -- the depth of the deepest entry it had prefetched when the
 *               last pixel finished (up to 128 list positions past the last one walked) x margin (>= 1), or +inf when its list ran out first.
 *   redo_flags  [V] uint32, device, zeroed by the caller: redo_flags[v] counts the tiles of view v that WERE cut and ran out of list with a
 *               pixel still alive (or were cut empty); non-zero: entries the cut removed might have reached that pixel, the images of view v are not
 *               to be trusted and the caller renders the frame again without cuts.  A flag that stays 0 PROVES the frame exact: every cut
 *               tile finished all of its pixels inside the kept prefix of its list, which is a prefix of the full list.
 * The arming is one-shot (V = 0 disarms) and costs nothing when unused.  Lists, n_contrib and final_T of a validated frame equal the uncut frame's on the
 * kept prefix; the states of a cut call must not be handed to a backward (forward-only calls never are). */
int gsr_arm_depth_cuts(int32_t V, const uint32_t* const* cut_in, uint32_t* const* cut_out, uint32_t* redo_flags, float margin);
/* flags of the batch forward: GSR_FORWARD_ONLY = the caller will NOT run gsr_backward_batch on the states of this call (the
 * no-grad renders of /root/reference/src/render/renderer.py:18-23, /root/reference/src/predict.py:115-123): the forward then skips
 * what only the backward reads (the per-Gaussian record-slot offsets: one scattered store per Gaussian and view), and a view that is
 * blended inside its owner's tile pass (geometry_of + colors_views: the mask render next to the colour render) is not preprocessed at
 * all -- the tile pass reads its colours from colors_views[v]; its radii[v] are then NOT written (they equal its owner's).
 * gsr_forward_render_batch must be given the same colors_views / flags as the gsr_forward_batch call it completes. */
#define GSR_FORWARD_ONLY 1
int gsr_forward_render_batch(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered,
                             void* const* geom_states, void* const* binning_states, void* const* image_states,
                             void* batch_state, const int32_t* geometry_of, const float* const* colors_views, float* const* out_color,
                             float* const* out_depth, int32_t flags, void* stream);
/* Both forward stages in ONE call: preprocess all views, synchronise once for the duplicate counts, and -- when
 * every view's binning state fits the buffer the caller provided (binning_bytes[v] >= gsr_binning_bytes(D_v)) --
 * launch the render stage straight away, with no host round trip through the caller in between (that round trip
 * is ~60 us of GPU idle time per step from Python).  Callers size the buffers from the previous step's counts.
 * Returns 0: rendered; 1: some buffer was too small (or NULL) -- nothing was rendered, num_rendered_host is
 * filled, allocate exact sizes and call gsr_forward_render_batch; < 0: error. */
int gsr_forward_batch(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                      const float* rotations, const float* opacities, const float* colors_precomp,
                      const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                      void* const* geom_states, int32_t* const* radii,
                      void* const* binning_states, const size_t* binning_bytes, void* const* image_states,
                      void* batch_state, const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                      uint32_t* num_rendered_host, int32_t flags, void* stream);
/* Backward of all V views (precomputed colours only; with SH use gsr_backward per view): ONE blend-backward launch
 * over the combined tile queue, then ONE per-Gaussian kernel that loops over the views and writes the
 * gradients SUMMED over views.  Only dL_dmeans2D stays per view ([V] pointers to [P,3]). */
/* Capacity mode of gsr_forward_batch: NO host synchronisation.  The caller sizes every binning buffer for `capacity_entries[v]`
 * list entries (gsr_binning_bytes(capacity_entries[v], H, W); a view with geometry_of[v] != v repeats its owner's capacity) and
 * keeps passing those capacities as `num_rendered` to gsr_backward_batch (they fix the buffer layouts; scratch:
 * gsr_backward_scratch_bytes(P, capacity)).  The true counts are written to counts_dev[V] (device) at the end of the call;
 * a view whose count exceeds its capacity was rendered from a TRUNCATED list: the caller must read counts_dev before using
 * anything of that call and repeat it with enough room (gsdyn/step.py: loss_and_grads_views does, its images never leave
 * the library).  The kernels read the counts on the device (the word emit_entries leaves behind the offsets).
 * counts_dev may be device memory or device-mapped PINNED HOST memory (hipHostMalloc): the counts are written with system-scope
 * stores by the tile-order kernel, so a host that waits for any later event of the stream reads them without a copy. */
int gsr_forward_batch_capacity(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                               const float* rotations, const float* opacities, const float* colors_precomp,
                               const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                               void* const* geom_states, int32_t* const* radii, void* const* binning_states,
                               const uint32_t* capacity_entries, void* const* image_states, void* batch_state,
                               const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                               uint32_t* counts_dev, void* stream);
/* ---- raw-parameter mode (the tracking step, /root/reference/src/tracking/helpers.py:36-45): the caller holds unnormalised
 * rotations, logit opacities and log scales; their activations (normalize / sigmoid / exp) and the chain back through them are
 * applied INSIDE the per-Gaussian kernels of the forward and the backward instead of in two launches of their own.
 * Forward: the activated values are also written to rotations_out / opacities_out / scales_out -- the SAME buffers must be passed
 * as the call's `rotations` / `opacities` / `scales` arguments (and again to the backward).  Backward: the parameter gradients go
 * to d_unnorm_rotations / d_logit_opacities / d_log_scales; dL_drotations / dL_dopacity / dL_dscales may then be NULL.
 * Values are bit-identical to gsr_activate_forward / gsr_activate_backward. */
typedef struct gsr_raw_params {
  const float* unnorm_rotations;   /* [P,4] */
  const float* logit_opacities;    /* [P]   */
  const float* log_scales;         /* [P,3] */
  float* rotations_out;            /* forward: [P,4] */
  float* opacities_out;            /* forward: [P]   */
  float* scales_out;               /* forward: [P,3] */
  float* d_unnorm_rotations;       /* backward: [P,4] */
  float* d_logit_opacities;        /* backward: [P]   */
  float* d_log_scales;             /* backward: [P,3] */
} gsr_raw_params;
/* gsr_forward_batch_capacity / gsr_backward_batch with the activations fused (raw == NULL: exactly those functions). */
int gsr_forward_batch_capacity_raw(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                   const float* rotations, const float* opacities, const float* colors_precomp,
                                   const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                                   void* const* geom_states, int32_t* const* radii, void* const* binning_states,
                                   const uint32_t* capacity_entries, void* const* image_states, void* batch_state,
                                   const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                                   uint32_t* counts_dev, const gsr_raw_params* raw, void* stream);
int gsr_backward_batch_raw(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered, const float* means3D,
                           const float* scales, const float* rotations, const float* colors_precomp,
                           const float* cov3D_precomp, const int32_t* const* radii, void* const* geom_states,
                           void* const* binning_states, void* const* image_states, void* batch_state,
                           const int32_t* geometry_of, const float* const* dL_dcolor, void* const* scratch, float* dL_dmeans3D,
                           float* const* dL_dmeans2D, float* dL_dcolors, float* const* dL_dcolors_views, float* dL_dopacity,
                           float* dL_dscales, float* dL_drotations, float* dL_dcov3D, const gsr_raw_params* raw, void* stream);
int gsr_backward_batch(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered, const float* means3D,
                       const float* scales, const float* rotations, const float* colors_precomp,
                       const float* cov3D_precomp, const int32_t* const* radii, void* const* geom_states,
                       void* const* binning_states, void* const* image_states, void* batch_state,
                       const int32_t* geometry_of, const float* const* dL_dcolor, void* const* scratch, float* dL_dmeans3D,
                       float* const* dL_dmeans2D, float* dL_dcolors, float* const* dL_dcolors_views, float* dL_dopacity,
                       float* dL_dscales, float* dL_drotations, float* dL_dcov3D, void* stream);

/* ---- neighbour terms of the t > 0 tracking loss, fused (caller side of the path):
 *   rigid, rot, iso of /root/reference/src/tracking/train_utils.py:198-222 as three means over (foreground point, neighbour).
 * All per-point arrays are indexed by foreground rank; fg_idx[n_fg] (int64) maps rank -> Gaussian; neighbor_* are [n_fg,K];
 * rotations are the NORMALISED quaternions (w,x,y,z) of all P Gaussians.
 * forward: block_partials[3][gsr_rigidity_blocks(n_fg)] = per-block sums of the three terms (caller: sum / (n_fg K)).
 * backward: grad3[3] (device) = upstream gradients of the three means ALREADY divided by n_fg K; rev_ptr[n_fg+1] / rev_edge[n_fg K]
 *   (int32) = reverse adjacency (edges sorted by neighbour); scratch = 7 (n_fg + n_fg K) floats; writes the foreground rows of
 *   d_means3D[P,3] and d_rotations[P,4] (the caller zero-fills the rest).  No float atomics: deterministic. */
int32_t gsr_rigidity_blocks(int32_t n_fg);
int gsr_rigidity_forward(int32_t n_fg, int32_t K, const float* means3D, const float* rotations, const int64_t* fg_idx,
                         const int64_t* neighbor_indices, const float* neighbor_weight, const float* neighbor_dist,
                         const float* prev_inv_rot_fg, const float* prev_offset, float* block_partials, void* stream);
int gsr_rigidity_backward(int32_t n_fg, int32_t K, const float* means3D, const float* rotations, const int64_t* fg_idx,
                          const int64_t* neighbor_indices, const float* neighbor_weight, const float* neighbor_dist,
                          const float* prev_inv_rot_fg, const float* prev_offset, const float* grad3, const int32_t* rev_ptr,
                          const int32_t* rev_edge, float* scratch, float* d_means3D, float* d_rotations, void* stream);

/* ---- the whole view-independent part of the t > 0 loss in one call (train_utils.py:198-241): the three neighbour terms above plus
 *   floor = mean(clamp(means3D[fg].y, min=0))                                                     (train_utils.py:225)
 *   bg    = mean_b sum_c |means3D[bg] - init_bg_pts| + mean_b sum_c |rotations[bg] - init_bg_rot| (train_utils.py:227-229)
 * and their weighted sum.  bg_idx[n_bg] (int64) lists the background Gaussians; init_bg_* are indexed by background rank.
 * forward: terms6 (device) = rigid, rot, iso, floor, bg, sum_k weights5_host[k] * term_k;  partials = gsr_shared_terms_partials floats, 16-byte aligned.
 * backward: d_means3D[P,3] / d_rotations[P,4] = grad_total[0] (device) * d terms6[5] / d input -- fully written, or, with
 *   GSR_SHARED_ACCUMULATE in `flags`, ADDED to what the buffers hold (rows in neither index list are left alone);
 *   scratch = gsr_shared_terms_scratch(n_fg, K) floats, 16-byte aligned; its first 16 n_fg floats are the per-point frames the
 *   forward left at the start of `partials`: pass that buffer (if large enough) with GSR_SHARED_FRAMES_VALID to skip their
 *   recomputation.  rev_ptr / rev_edge as for gsr_rigidity_backward. */
#define GSR_SHARED_ACCUMULATE 1
#define GSR_SHARED_FRAMES_VALID 2
int32_t gsr_shared_terms_scratch(int32_t n_fg, int32_t K);
int32_t gsr_shared_terms_partials(int32_t n_fg, int32_t n_bg);
int gsr_shared_terms_forward(int32_t n_fg, int32_t K, int32_t n_bg, const float* means3D, const float* rotations, const int64_t* fg_idx,
                             const int64_t* bg_idx, const int64_t* neighbor_indices, const float* neighbor_weight,
                             const float* neighbor_dist, const float* prev_inv_rot_fg, const float* prev_offset,
                             const float* init_bg_pts, const float* init_bg_rot, const float* weights5_host, float* partials,
                             float* terms6, void* stream);
int gsr_shared_terms_backward(int32_t P, int32_t n_fg, int32_t K, int32_t n_bg, const float* means3D, const float* rotations,
                              const int64_t* fg_idx, const int64_t* bg_idx, const int64_t* neighbor_indices,
                              const float* neighbor_weight, const float* neighbor_dist, const float* prev_inv_rot_fg,
                              const float* prev_offset, const float* init_bg_pts, const float* init_bg_rot,
                              const float* weights5_host, const float* grad_total, const int32_t* rev_ptr, const int32_t* rev_edge,
                              float* scratch, float* d_means3D, float* d_rotations, int32_t flags, void* stream);

/* ---- activations of the raw parameters (replaces the torch ops of params2rendervar, /root/reference/src/tracking/helpers.py:36-45):
 *   rotations[P,4] = unnorm_rotations / max(|unnorm_rotations|, 1e-12), opacities[P,1] = sigmoid(logit_opacities), scales[P,3] = exp(log_scales).
 * backward: any of d_rotations / d_opacities / d_scales may be NULL (treated as zero); the three outputs are always written. */
int gsr_activate_forward(int32_t P, const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                         float* rotations, float* opacities, float* scales, void* stream);
int gsr_activate_backward(int32_t P, const float* unnorm_rotations, const float* opacities, const float* scales,
                          const float* d_rotations, const float* d_opacities, const float* d_scales, float* d_unnorm_rotations,
                          float* d_logit_opacities, float* d_log_scales, void* stream);

/* ---- densification bookkeeping of a step (/root/reference/src/tracking/train_utils.py:243-245), for rows 0, view_step, 2 view_step, ...
 * of radii[V,P] (view_step = 2: the colour renders of a colour + segmentation batch):
 *   max_2D_radius[i] = max(max_2D_radius[i], max_v radii[v][i]);   seen[i] = any_v radii[v][i] > 0   (one byte per Gaussian) */
int gsr_radius_bookkeeping(int32_t V, int32_t view_step, int32_t P, const int32_t* radii, float* max_2D_radius, uint8_t* seen,
                           void* stream);

/* ---- optimiser step of the tracking loop: torch.optim.Adam's default update (no weight decay, no amsgrad, not maximising) for up
 * to GSR_ADAM_MAX_TENSORS parameter tensors in ONE launch.  The reference builds Adam with one parameter group per tensor
 * (/root/reference/src/tracking/train_utils.py:152-164: per-group lr, eps 1e-15).  bias_correction1 = 1 - beta1^step and
 * bias_correction2_sqrt = sqrt(1 - beta2^step) are formed by the caller from its step counter.  All pointers: device, fp32, n elements. */
#define GSR_ADAM_MAX_TENSORS 16
typedef struct gsr_adam_tensor {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
  int64_t n;
  float lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt;
  float one_minus_beta1, one_minus_beta2;   /* formed in double by the caller, as torch does (1 - float(0.999) is off by 1e-5) */
} gsr_adam_tensor;
int gsr_adam_step(int32_t n_tensors, const gsr_adam_tensor* tensors, void* stream);

/* ---- rollout plumbing (callers: gsdyn/dynamics.py)
 * gsr_fps: farthest point sampling of pos[N,3] -> out_idx[npoints] (int64), first pick start_idx, every further pick the
 *   point with the largest squared distance to the picked set (first maximum on ties).  Stands in for
 *   dgl.geometry.farthest_point_sampler (/root/reference/src/render/dynamics_module.py:46,65).  scratch: gsr_fps_scratch_bytes(N,
 *   npoints) bytes (256-byte aligned).  Clouds of 2049 .. 524 288 points are sampled by up to 256 co-resident workgroups that keep
 *   their slice in LDS and meet once per pick (same picks as the single-workgroup path, bit for bit).
 * gsr_lbs: moves P Gaussians with n_bones bones (/root/reference/src/render/utils.py:207-239): weights 1/max(|x - bone|, 1e-4)
 *   normalised over the bones; out_xyz = sum_b w_b (R_b (x - bone_b) + t_b + bone_b); out_quat = normalise(sum_b w_b q_b) * quat
 *   (quaternions w,x,y,z; rotations row-major 3x3; quat / out_quat may be NULL). */
size_t gsr_fps_scratch_bytes(int32_t N, int32_t npoints);
/* gsr_fit_rotations: one rotation per bone from its 3x3 moment matrix F_i = sum_j (new_j - new_i)(old_j - old_i)^T (row-major, fp32)
 *   and the number of related bones, with the decision tree of /root/reference/src/render/utils.py:147-205 (batched on the device:
 *   fp64 one-sided Jacobi SVD per bone).  code[i]: 0 = identity by rule (no related bone, F = 0, or full rank with det F < 0),
 *   2 = Kabsch rotation written, 3 = rank-1 bone: the x axis turned onto the dominant left singular vector, signed as LAPACK signs
 *   U[:, 0] (the reference's answer there depends on its SVD backend's convention; LAPACK's is -sign(F00) c0 / |c0| for F's first
 *   column c0), 1 = rank-1 bone whose F has a vanishing first column, LEFT AS IDENTITY for the caller's own SVD backend. */
int gsr_fit_rotations(int32_t n_bones, const float* moments, const float* n_related, float* rotations, int32_t* code, void* stream);
/* gsr_fps_thin: the bones of a rollout step in one launch (downsample_vertices, /root/reference/src/render/dynamics_module.py:44-51):
 * out_idx[npoints] = farthest point sampling of pos[N,3] as gsr_fps (N <= 1024, first pick start_idx); then the radius thinning of
 * /root/reference/src/data/utils.py:50-65 over those picks: thin_idx[0 .. *thin_count) = positions in out_idx of the kept points
 * (first thin_start_idx, then the point farthest from the kept set while that distance exceeds `radius`; distances as torch.norm
 * evaluates them in fp32, first maximum on ties).  thin_count: device int32. */
int gsr_fps_thin(int32_t N, const float* pos, int32_t npoints, int32_t start_idx, float radius, int32_t thin_start_idx, int64_t* out_idx,
                 int64_t* thin_idx, int32_t* thin_count, void* stream);
/* Fixed-shape forms for a rollout step that is replayed from a hipGraph (no host round trip for the bone count or the relation count):
 * gsr_fps_thin pads thin_idx[*thin_count .. npoints) with thin_start_idx; gsr_construct_edges = the relations of the rollout's graph
 * (gsdyn.dynamics.construct_edges, /root/reference/src/data/dataset.py:88-147: object particles 0 .. n_obj_cap - 1 of which the first
 * *n_valid are real, one tool particle at index n_obj_cap; related = both real, not both tools, squared distance < thresh_sq, and among
 * objects the sender one of the receiver's topk nearest, itself included) as index lists of e_cap entries in the adjacency matrix's
 * row-major order, padded with dummy_index, *count = the real ones; gsr_lbs_valid = gsr_lbs over the first *n_valid bones. */
int gsr_construct_edges(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                        int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, void* stream);
/* gsr_construct_edges_dense (ABI 119) = gsr_construct_edges that also writes the relations as a dense relations_n x relations_n 0 / 1
 * int64 matrix (what gsr_fit_bones reads; rows / columns beyond the tool are zero).  gsr_rollout_step_tail (ABI 119): the bookkeeping
 * that ends a graphed rollout step, one launch -- pos_track[t] = all_pos[track[t]]; hist [n_his, n_track, 3] and eef_hist [n_his, 3] shifted
 * by one frame with the new positions / eef_next appended (/root/reference/src/gnn/dynamics_module.py:150-165 does this with torch.cat);
 * pred_out [n_bones, 3] = pred_in rows below *n_valid, zeros above; *n_valid_out = *n_valid; *bad += bones below *n_valid with code 1.
 * gsr_lbs / gsr_lbs_valid: out_xyz / out_quat may be xyz / quat themselves (in place). */
int gsr_construct_edges_dense(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                              int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, int64_t* relations, int32_t relations_n,
                              void* stream);
int gsr_rollout_step_tail(int32_t n_track, int32_t n_his, int32_t n_bones, const float* all_pos, const int64_t* track, float* pos_track, float* hist,
                          float* eef_hist, const float* eef_next, const float* pred_in, const int32_t* n_valid, const int32_t* code, float* pred_out,
                          int32_t* n_valid_out, int64_t* bad, void* stream);
/* ABI 122 -- the glue of a rollout step (/root/reference/src/render/dynamics_module.py:104-133 spells it with torch.cat / index / clamp per
 * step) as launches of this library; a step replayed from a hipGraph lasts ~4.5 us per node whatever the node does.
 * gsr_construct_edges_rows = gsr_construct_edges_dense + row_start [relations_n + 1]: row_start[i] = the number of list entries whose
 *   receiver is below i (torch.searchsorted(receivers, arange(relations_n + 1)) on the padded list; n_obj_cap < dummy_index < relations_n).
 * gsr_rollout_step_head: from the tracked particles' history hist [n_his, n_track, 3], the step's bone picks sample_idx[thin_idx[r]]
 *   (gsr_fps_thin's two outputs, n_bones of each), the tool's history eef_hist [n_his, 3] and target eef_next [3], and the constant
 *   per-row attributes attrs [n_rows, attr_dim] / instance [n_rows] -- one thread per padded row r < n_rows (bones 0 .. n_bones - 1, the
 *   tool at n_bones, zero rows behind):  state_rows [n_rows, 3 n_his] (a row = its n_his positions), action_rows [n_rows, 3] (the tool's
 *   eef_next - eef_hist[-1], zeros elsewhere), particle_inputs [n_rows, attr_dim + (with_state ? 3 n_his : 0) + 3] = (attributes, [state],
 *   action), rel_nodes [n_rows, attr_dim + 1 + 3 n_his] = (attributes, instance, state), bones_last [n_bones, 3] and states_last
 *   [n_bones + 1, 3] = the last frame's positions.
 * gsr_rollout_step_motion: predicted = last position + clamp(pred_motion, +-motion_clamp), motion = predicted - last, for the n_bones bone
 *   rows, written into the step's skinning packet (gsdyn.dynamics.pack_skin: [0] = *n_valid as a float, [1] = 1, then blocks of n_bones
 *   rows: bones 3, rotations 9, motions 3, quaternions 4, predicted 3 floats per row) -- the rotations and quaternions blocks are
 *   gsr_fit_bones' outputs, which may point into the packet. */
int gsr_construct_edges_rows(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                             int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, int64_t* relations, int32_t relations_n,
                             int64_t* row_start, void* stream);
int gsr_rollout_step_head(int32_t n_track, int32_t n_his, int32_t n_bones, int32_t n_rows, int32_t attr_dim, int32_t with_state, const float* hist,
                          const int64_t* sample_idx, const int64_t* thin_idx, const float* eef_hist, const float* eef_next, const float* attrs,
                          const float* instance, float* bones_last, float* states_last, float* state_rows, float* action_rows, float* particle_inputs,
                          float* rel_nodes, void* stream);
int gsr_rollout_step_motion(int32_t n_bones, int32_t n_his, float motion_clamp, const float* state_rows, const float* pred_motion, const int32_t* n_valid,
                            float* skin_packet, void* stream);
int gsr_lbs_valid(int32_t P, int32_t n_bones, const int32_t* n_valid, const float* bones, const float* rotations, const float* translations,
                  const float* bone_quats, const float* xyz, const float* quat, float* out_xyz, float* out_quat, void* stream);
/* gsr_fit_bones: the moment matrices, gsr_fit_rotations and the bones' unit quaternions in one launch -- what interpolate_motions
 * (/root/reference/src/render/utils.py:138-243) needs per bone: F_b = sum over the bones j with relations[b][j] != 0 of
 * (new_j - new_b)(old_j - old_b)^T with old = bones, new = bones + motions (fp32, ascending j), rotations[b] as gsr_fit_rotations
 * (same codes), quats[b] = normalize(mat2quat(rotations[b])) with the reference's branches (utils.py:71-111), (w, x, y, z).
 * relations: n_bones rows of `relations_row_stride` int64 (>= n_bones; a [:n, :n] view of a larger square matrix needs no copy).
 * A bone with code 1 has the identity and its quaternion: the caller replaces both (as for gsr_fit_rotations). */
int gsr_fit_bones(int32_t n_bones, const float* bones, const float* motions, const int64_t* relations, int64_t relations_row_stride,
                  float* rotations, float* quats, int32_t* code, void* stream);
/* The propagation network of the particle dynamics (DynamicsPredictor.forward, /root/reference/src/gnn/model.py:70-246) runs its matrix
 * products through the GEMM library; the two steps between them are kernels of this library.  rel_nodes [n_rows, attr_dim + group_dim +
 * state_cols] = per node (attributes, instance columns, state history); receivers / senders [n_rel] int64, ASCENDING in the receiver.
 * (ABI 119 also exported the whole network as one persistent launch, gsr_gnn_propagate: measured 2x slower, removed in ABI 120.) */
/* gsr_gnn_rel_inputs: out [n_rel, 2 attr_dim + 1 + state_cols] = the relation encoder's input rows, formed from rel_nodes as above;
 * gsr_gnn_aggregate:  agg[i] = sum over the relations e in [row_start[i], row_start[i + 1]) -- receivers ascending -- in list order of
 *   relu(rel_part[e] + node_parts[i][0 : width] + node_parts[senders[e]][width : 2 width]),  rel_part [n_rel, width] = relation_encode
 *   @ W1^T + b, node_parts [n_rows, 2 width] = effect @ [W2 | W3]^T with W = [W1 | W2 | W3] the relation propagator's weight: what
 *   relation_propagator(cat(relation_encode, effect[recv], effect[send])) followed by the index_add onto the receivers computes.
 *   Rows >= n_sum_rows get zeros (a padded graph's dummy last row collects every dummy relation: hundreds, walked by one thread each). */
int gsr_gnn_rel_inputs(int32_t n_rel, int32_t attr_dim, int32_t group_dim, int32_t state_cols, const float* rel_nodes, const int64_t* receivers,
                       const int64_t* senders, float* out, void* stream);
int gsr_gnn_aggregate(int32_t n_rows, int32_t n_sum_rows, int32_t width, const float* rel_part, const float* node_parts, const int64_t* senders,
                      const int64_t* row_start, float* agg, void* stream);
/* gsr_gnn_aggregate_res (ABI 122) = gsr_gnn_aggregate that also writes res_out = res_a + res_b ([n_rows, width] each): the particle
 * propagator's addend of the step (particle_encode @ Wp1^T + b, plus the effect as the residual) -- one launch less per propagation step. */
int gsr_gnn_aggregate_res(int32_t n_rows, int32_t n_sum_rows, int32_t width, const float* rel_part, const float* node_parts, const int64_t* senders,
                          const int64_t* row_start, float* agg, const float* res_a, const float* res_b, float* res_out, void* stream);
int gsr_fps(int32_t N, const float* pos, int32_t npoints, int32_t start_idx, float* scratch, int64_t* out_idx, void* stream);
int gsr_lbs(int32_t P, int32_t n_bones, const float* bones, const float* rotations, const float* translations,
            const float* bone_quats, const float* xyz, const float* quat, float* out_xyz, float* out_quat, void* stream);

/* ---- fused image loss of the tracking step (caller side of the path):
 *   loss = w_l1 * mean|pred - target| + w_ssim * (1 - mean SSIM(pred, target)),  SSIM with the reference's 11x11
 *   Gaussian window (sigma 1.5, zero padding): /root/reference/src/tracking/external.py:101-135, used at
 *   /root/reference/src/tracking/train_utils.py:185,195 with w_l1 = 0.8, w_ssim = 0.2.
 * forward: writes one partial sum of |.| and of the SSIM map per block (gsr_image_loss_blocks of them; the caller
 * adds them up) and the three per-pixel partials fA/fC/fE ([C,H,W] each) the backward needs.
 * backward: d_pred[C,H,W] = grad_loss[c / channels_per_image] * d loss / d pred.  A batch of N images is passed as C = N *
 * channels_per_image channels (the means then run over one image each; block partial sums are channel-major, so the caller
 * adds them up per image); grad_loss has N entries.  `window11_host` = the 11 normalised 1-D weights (host). */
int32_t gsr_image_loss_blocks(int32_t C, int32_t H, int32_t W);
int gsr_image_loss_forward(const float* window11_host, int32_t C, int32_t H, int32_t W, const float* pred, const float* target,
                           float* fA, float* fC, float* fE, float* block_l1, float* block_ssim, void* stream);
int gsr_image_loss_backward(const float* window11_host, int32_t C, int32_t H, int32_t W, const float* pred,
                            const float* target, const float* fA, const float* fC, const float* fE, const float* grad_loss,
                            int32_t channels_per_image, float w_l1, float w_ssim, float* d_pred, void* stream);

/* ---- image terms of ALL renders of a tracking step in one call (SURVEY.md section 8f rows N1 + N2; caller: gsdyn/step.py
 * get_loss_views).  `renders` is the rasterizer's output batch [n_images, channels, H, W] read in place; image i is compared with
 * target[i] after the per-camera affine of /root/reference/src/tracking/train_utils.py:181-183,
 *     pred_i = exp(cam_m[cam_row[i]]) * render_i + cam_c[cam_row[i]]        (per channel; cam_row[i] < 0: pred_i = render_i),
 * and  losses[i] = w_l1 mean|pred_i - target_i| + w_ssim (1 - mean SSIM(pred_i, target_i)),  losses[n_images] = sum_i weight[i] losses[i]
 * (train_utils.py:185,195 and the weighted sum at :235-241).
 * forward: fA/fC/fE [n_images, channels, H, W] keep the per-pixel SSIM partials; partials = 2 * gsr_views_loss_blocks floats.
 * backward: d_renders = grad_total[0] * d losses[n_images] / d renders, laid out like `renders` (what gsr_backward_batch takes as
 *   dL_dcolor); d_cam_m / d_cam_c [n_cams, channels] (zero-filled, rows hit by several images add up in image order; both may be
 *   NULL when no image has a camera row).  No float atomics: deterministic. */
#define GSR_LOSS_MAX_IMAGES 32
typedef struct gsr_loss_views {
  int32_t n_images;                            /* <= GSR_LOSS_MAX_IMAGES */
  int32_t channels;                            /* per image, <= 4 */
  int32_t cam_row[GSR_LOSS_MAX_IMAGES];
  float weight[GSR_LOSS_MAX_IMAGES];
  const float* target[GSR_LOSS_MAX_IMAGES];    /* DEVICE [channels,H,W] each */
  const float* target_moments[GSR_LOSS_MAX_IMAGES];  /* optional (all or none): gsr_target_moments of target[i], [2,channels,H,W]:
                                                        a target that stays fixed over many steps has its two blurred maps computed
                                                        once; the forward then runs 3 instead of 5 window passes, same bits */
} gsr_loss_views;
int32_t gsr_views_loss_blocks(int32_t n_images, int32_t channels, int32_t H, int32_t W);
/* moments[2,channels,H,W] = window blur of target and of target^2 (zero padding), for gsr_loss_views::target_moments */
int gsr_target_moments(const float* window11_host, int32_t channels, int32_t H, int32_t W, const float* target, float* moments,
                       void* stream);
int gsr_views_loss_forward(const float* window11_host, const gsr_loss_views* views, int32_t H, int32_t W, const float* renders,
                           const float* cam_m, const float* cam_c, float w_l1, float w_ssim, float* fA, float* fC, float* fE,
                           float* partials, float* losses, void* stream);
int gsr_views_loss_backward(const float* window11_host, const gsr_loss_views* views, int32_t H, int32_t W, const float* renders,
                            const float* cam_m, const float* cam_c, int32_t n_cams, const float* fA, const float* fC,
                            const float* fE, const float* grad_total, float w_l1, float w_ssim, float* d_renders, float* partials,
                            float* d_cam_m, float* d_cam_c, void* stream);

/* ---- mark_visible  (replaces `mark_visible`; GaussianRasterizer.markVisible).  present[P] = view z > 0.2 */
int gsr_mark_visible(const float* viewmatrix, int32_t P, const float* means3D, uint8_t* present, void* stream);

/* ---- introspection for tests / benches (copies of internal state, device -> caller's DEVICE buffers) */
typedef struct gsr_debug_views {
  const float* rec;           /* [P,16] mean2D.xy, conic A,B | conic C, opacity, r, g | b, depth, alpha-box bits x2 |
                                 tile-rect bits x2, offsets[g] bits, 0   (one 64-byte line per Gaussian) */
  const uint32_t* rect;       /* [P,2] minx|miny<<16, maxx|maxy<<16 */
  const uint32_t* tiles_touched; /* [P] */
  const uint32_t* offsets;    /* [P+1] exclusive prefix of tiles_touched */
  const uint32_t* point_list; /* [D] sorted Gaussian ids   (binning state) */
  const uint32_t* ranges;     /* [T,2]                      (image state)   */
  const float* final_T;       /* [H*W] */
  const uint32_t* n_contrib;  /* [H*W] */
} gsr_debug_views;
int gsr_debug_get_views(int32_t P, uint32_t num_rendered, int32_t image_height, int32_t image_width,
                        const void* geom_state, const void* binning_state, const void* image_state,
                        gsr_debug_views* out);

/* Debug builds only (make TIMING=1): accumulated s_memtime cycles per phase of the forward tile loop
 * (16 words; returns 1 and zeros in a normal build). */
int gsr_debug_phase_timing(uint64_t* out16);

/* Device self-test of the wave-64 building blocks (DPP reduction, ballot match, scan, sorts).
 * Returns 0 when all pass, else a bitmask of failed checks. */
int gsr_selftest(void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * gsr_profile_begin() arms it; every kernel launched by gsr_* calls after that is bracketed by two events;
 * gsr_profile_end() waits for them and returns one (name, total ms, launches) row per kernel name.
 * Not thread-safe; off by default (no events exist in normal operation). */
typedef struct gsr_kernel_time {
  char name[32];
  double total_ms;
  int64_t launches;
} gsr_kernel_time;
int gsr_profile_begin(void);
int gsr_profile_end(gsr_kernel_time* out, int32_t max_entries, int32_t* n_out);

/* Host side of the capacity-mode forward (gsr_forward_batch_capacity*): the tile-order kernel stores every view's entry count with
 * system-scope stores into the caller's PINNED host array `counts` (pre-set to -1 by the caller).  gsr_wait_counts spins on that
 * array from C -- no device call, no stream synchronisation, nothing of the caller's runtime (a Python caller's interpreter lock is
 * released for the duration by ctypes) -- until all `n` values are >= 0 and returns their maximum; -1 after `timeout_us`
 * microseconds.  Between polls it yields the core (`sched_yield`) once `spin_us` microseconds have passed, so that N ranks of a
 * node waiting at the same time do not each hold a core at 100 %.  Replaces the reference-side pattern of a blocking
 * num_rendered read inside the forward (upstream rasterizer_impl.cu; called from /root/reference/src/tracking/train_utils.py:178). */
int64_t gsr_wait_counts(const volatile int32_t* counts, int32_t n, int64_t spin_us, int64_t timeout_us);

const char* gsr_last_error(void);
int gsr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H_ */
