// gsr_api.hip -- extern "C" entry points of libgsr_hip.so (declared in include/gsr.h) and the
// device self-test.  Host-side orchestration only: buffer carving, kernel sequencing, the one
// device->host read of num_rendered.  No allocation, no global state beyond a thread-local error string.
#include <atomic>
#include <chrono>
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "gsr_common.h"

static thread_local char g_err[512] = "";

// ---- profiling state (process-wide; benches are single-threaded per process)
namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace
// ---- roctx ranges (tracing): every exported compute entry point and every kernel launch site opens a range
// named after itself, so a rocprofv3 --marker-trace timeline reads gsr_forward_batch > preprocess_fwd, emit_entries, ...  The marker
// library is looked up at first use (librocprofiler-sdk-roctx, else libroctx64) and is optional: without it the ranges are no-ops.
// GSR_ROCTX=0 switches them off.
#include <dlfcn.h>
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
bool roctx_ready() {
  static const bool ok = [] {
    const char* e = getenv("GSR_ROCTX");
    if (e && *e && atoi(e) == 0) return false;
    for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
      if (!h) continue;
      g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
      g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
      if (g_roctx_push && g_roctx_pop) return true;
    }
    return false;
  }();
  return ok;
}
}  // namespace
GsrRange::GsrRange(const char* name) : on(roctx_ready()) { if (on) g_roctx_push(name); }
GsrRange::~GsrRange() { if (on) g_roctx_pop(); }

GsrProfScope::GsrProfScope(const char* name, hipStream_t s) : slot(-1), st(s), range(name) {
  if (!g_prof_on) return;
  ProfRec r; r.name = name;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
GsrProfScope::~GsrProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].b, st);
}

void gsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

static int make_cam(const gsr_settings* s, GsrCam* c) {
  if (!s) { gsr_set_error("gsr: settings is NULL"); return -2; }
  if (s->image_height <= 0 || s->image_width <= 0) { gsr_set_error("gsr: image size must be positive"); return -2; }
  if (!s->bg || !s->viewmatrix || !s->projmatrix || !s->campos) {
    gsr_set_error("gsr: bg/viewmatrix/projmatrix/campos must be device pointers");
    return -2;
  }
  c->H = s->image_height; c->W = s->image_width;
  c->gx = (c->W + GSR_TILE - 1) / GSR_TILE; c->gy = (c->H + GSR_TILE - 1) / GSR_TILE;
  c->T = c->gx * c->gy;
  if (c->gx > 65535 || c->gy > 65535) { gsr_set_error("gsr: image too large (tile grid exceeds 65535)"); return -2; }
  c->tanfovx = s->tanfovx; c->tanfovy = s->tanfovy; c->scale_modifier = s->scale_modifier;
  c->sh_degree = s->sh_degree; c->M = s->sh_coeffs;
  c->bg = s->bg; c->view = s->viewmatrix; c->proj = s->projmatrix; c->campos = s->campos;
  return 0;
}

extern "C" {

int gsr_version(void) { return GSR_VERSION; }
const char* gsr_last_error(void) { return g_err; }

size_t gsr_geom_bytes(int32_t P) { GeomState g; return gsr_carve_geom(nullptr, P, &g); }
size_t gsr_image_bytes(int32_t H, int32_t W) { ImageState im; return gsr_carve_image(nullptr, H, W, &im); }
size_t gsr_binning_bytes(uint32_t D, int32_t, int32_t) { BinningState b; return gsr_carve_binning(nullptr, D, &b); }
size_t gsr_backward_scratch_bytes(int32_t, uint32_t D) { return gsr_align((size_t)(D ? D : 1) * GSR_PARTIAL_FLOATS * 4); }

// ---------------------------------------------------------------------------------------- table builders
}  // extern "C"
namespace {
void fill_pre_view(GsrPreView& o, const GsrCam& cam, const GeomState& g, int32_t* radii, uint32_t* block_sums,
                   const float* colors) {
  o.colors = colors;
  o.view = cam.view; o.proj = cam.proj; o.campos = cam.campos; o.tanfovx = cam.tanfovx; o.tanfovy = cam.tanfovy;
  o.rec = g.rec; o.rect = g.rect; o.tiles_touched = g.tiles_touched; o.clamped = g.clamped; o.radii = radii;
  o.block_sums = block_sums; o.ekey = g.ekey; o.block_hash = nullptr; o.cmp_rec = nullptr; o.cmp_rect = nullptr; o.cmp_ekey = nullptr; o.cmp_tiles = nullptr; o.skip = 0;
  o.used = g.used; o.tracked = g.counters + 1;
}

// Speculative depth cuts armed by gsr_arm_depth_cuts for the NEXT forward-only binning + blend of this host thread (one-shot).
struct ArmedCuts { int V = 0; const uint32_t* in[GSR_MAX_BATCH]; uint32_t* out[GSR_MAX_BATCH]; uint32_t* redo = nullptr; float margin = 1.0f; };
thread_local ArmedCuts t_cuts;
void fill_bin_view(GsrBinView& o, int P, uint32_t D, const GeomState& g, const BinningState& bs, const ImageState& im,
                   const uint32_t* block_sums) {
  o.shares_lists = 0; o.fused_alias = 0; o.D_dev = nullptr; o.owner = 0;
  o.ekey = g.ekey; o.rec_w = g.rec; o.tile_rows = nullptr; o.depth_cut = nullptr;
  o.rec = g.rec; o.rect = g.rect; o.tiles_touched = g.tiles_touched; o.block_sums = block_sums;
  o.block_offsets = gsr_host_block_scan(P) ? nullptr : g.block_offsets;
  o.offsets = g.offsets;
  o.tkey[0] = bs.tkey[0]; o.tkey[1] = bs.tkey[1]; o.dg[0] = bs.dg[0]; o.dg[1] = bs.dg[1];
  o.point_list = bs.point_list; o.block_hist = bs.block_hist; o.ranges = im.ranges;
  o.D = D; o.nblocks = D ? gsr_radix_blocks(D) : 0u;
}
void fill_render_view(GsrRenderView& o, const GsrCam& cam, const GeomState& g, const BinningState& bs, const ImageState& im,
                      float* out_color, float* out_depth, const float* dL_dcolor, float4* partials) {
  o.point_list = bs.point_list; o.rec = g.rec; o.bg = cam.bg; o.final_T = im.final_T; o.n_contrib = im.n_contrib;
  o.out_color = out_color; o.out_depth = out_depth; o.dL_dcolor = dL_dcolor; o.rect = g.rect; o.offsets = g.offsets;
  o.partials = partials; o.ranges = im.ranges; o.partner = -1; o.fused_alias = 0; o.colors = nullptr; o.contrib = bs.contrib;
  o.used = g.used; o.tracked = g.counters + 1;
  o.cut_in = nullptr; o.cut_out = nullptr; o.redo = nullptr; o.cut_margin = 1.0f;
}

// Fused pairs: the FIRST alias of a view (same camera, other colours) is blended inside its owner's tile pass instead of
// getting tile tickets of its own; further aliases of the same owner keep the plain shared-list path.
void pair_up(int V, const int32_t* geometry_of, const uint32_t* num_rendered, int partner[GSR_MAX_BATCH], int fused[GSR_MAX_BATCH]) {
  static const bool off = [] { const char* e = getenv("GSR_NO_PAIR_FUSION"); return e && *e && atoi(e) != 0; }();
  for (int v = 0; v < V; ++v) { partner[v] = -1; fused[v] = 0; }
  if (!geometry_of || off) return;
  for (int v = 0; v < V; ++v) {
    const int u = geometry_of[v];
    if (u != v && partner[u] < 0 && num_rendered[u] > 0) { partner[u] = v; fused[v] = 1; }
  }
}
// Forward-only calls: the views pair_up will fuse into their owner's tile pass (when the owner renders anything at all) need no
// preprocess and no records -- decided before the entry counts exist.
void skippable_aliases(int V, const int32_t* geometry_of, const float* const* colors_views, int flags, int skip[GSR_MAX_BATCH]) {
  static const bool off = [] { const char* e = getenv("GSR_NO_PAIR_FUSION"); return e && *e && atoi(e) != 0; }();
  int taken[GSR_MAX_BATCH];
  for (int v = 0; v < V; ++v) { skip[v] = 0; taken[v] = 0; }
  if (!geometry_of || !colors_views || off || !(flags & GSR_FORWARD_ONLY)) return;
  for (int v = 0; v < V; ++v) {
    const int u = geometry_of[v];
    if (u != v && !taken[u]) { taken[u] = 1; skip[v] = 1; }
  }
}
void render_header(GsrRenderViews& t, int V, const GsrCam& cam, const uint4* order, uint32_t* queue) {
  t.V = V; t.W = cam.W; t.H = cam.H; t.gx = cam.gx; t.T = cam.T; t.order = order; t.queue = queue; t.no_colour_grad = 0; t.prio_frac256 = 0; t.pc_error_out = nullptr; t.track = 1; t.avg_list = 1u << 20;
}

// Pinned host words for the per-block {differs, entry count} pairs of up to GSR_MAX_BATCH views (per host thread; lives for the process).
uint32_t* pinned_sums() {
  static thread_local uint32_t* p = nullptr;
  if (!p && hipHostMalloc((void**)&p, sizeof(uint32_t) * 2 * GSR_MAX_BATCH * GSR_HOST_SCAN_MAX_BLOCKS, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
    p = nullptr;
  return p;
}

int check_inputs(const char* who, const float* means3D, const float* opacities, const float* colors_precomp, const float* shs,
                 const float* scales, const float* rotations, const float* cov3D_precomp, const GsrCam& cam) {
  if (!means3D || !opacities) { gsr_set_error("%s: NULL argument", who); return -2; }
  if ((colors_precomp == nullptr) == (shs == nullptr)) {
    gsr_set_error("%s: provide exactly one of colors_precomp / shs", who);
    return -2;
  }
  if (((scales == nullptr) || (rotations == nullptr)) == (cov3D_precomp == nullptr)) {
    gsr_set_error("%s: provide exactly one of scales+rotations / cov3D_precomp", who);
    return -2;
  }
  if (shs && (cam.M < (cam.sh_degree + 1) * (cam.sh_degree + 1) || cam.sh_degree > 3 || cam.sh_degree < 0 || cam.M > 16)) {
    gsr_set_error("%s: sh_degree %d needs (deg+1)^2 <= sh_coeffs (%d) <= 16", who, cam.sh_degree, cam.M);
    return -2;
  }
  return 0;
}

// Stage 1 of V views: ONE preprocess launch, the entry counts of all views back in one pinned copy (P <= 512 Ki;
// above that a scan launch + a 4-byte copy per view), ONE stream synchronisation.
// `sums`: device array [V][nblk] (the views' block_sums; for V = 1 the view's own GeomState::block_sums).
int stage1(int V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales, const float* rotations,
           const float* opacities, const float* colors_precomp, const float* const* colors_views, const float* shs,
           const float* cov3D_precomp, void* const* geom_states, int32_t* const* radii, uint32_t* sums,
           uint32_t* num_rendered_host, hipStream_t st, const gsr_raw_params* raw = nullptr, int32_t* same_host = nullptr,
           const int* skip = nullptr, const int32_t* geometry_of = nullptr, const void* prev_geom = nullptr,
           uint32_t* differs_pinned = nullptr) {   // differs_pinned: capacity mode's compare -- the per-block verdict words go straight to pinned host memory
  if (colors_views) {   // every view brings its own colours: they stand in for the shared array in the checks below
    if (shs || colors_precomp) { gsr_set_error("gsr forward: per-view colours exclude colors_precomp / shs"); return -2; }
    for (int v = 0; v < V; ++v) if (!colors_views[v]) { gsr_set_error("gsr forward: NULL per-view colour pointer"); return -2; }
    colors_precomp = colors_views[0];
  }
  GsrPreViews tab;
  tab.V = V;
  tab.raw_rot = tab.raw_op = tab.raw_sc = nullptr;
  tab.rot_out = tab.op_out = tab.sc_out = nullptr;
  if (raw) {
    if (!raw->unnorm_rotations || !raw->logit_opacities || !raw->log_scales || !raw->rotations_out || !raw->opacities_out || !raw->scales_out ||
        raw->rotations_out != rotations || raw->opacities_out != opacities || raw->scales_out != scales || cov3D_precomp) {
      gsr_set_error("gsr forward (raw parameters): NULL pointer, cov3D_precomp given, or rotations / opacities / scales are not the *_out buffers");
      return -2;
    }
    tab.raw_rot = raw->unnorm_rotations; tab.raw_op = raw->logit_opacities; tab.raw_sc = raw->log_scales;
    tab.rot_out = raw->rotations_out; tab.op_out = raw->opacities_out; tab.sc_out = raw->scales_out;
  }
  GsrCam cam0;
  const uint32_t nblk = (uint32_t)((P + GSR_BLOCK - 1) / GSR_BLOCK);
  for (int v = 0; v < V; ++v) {
    GsrCam cam;
    if (int rc = make_cam(&s[v], &cam)) return rc;
    if (v == 0) {
      cam0 = cam;
      if (int rc = check_inputs("gsr forward", means3D, opacities, colors_precomp, shs, scales, rotations, cov3D_precomp, cam)) return rc;
    } else if (cam.W != cam0.W || cam.H != cam0.H || cam.sh_degree != cam0.sh_degree || cam.M != cam0.M ||
               cam.scale_modifier != cam0.scale_modifier) {
      gsr_set_error("gsr batch: all views must share image size, sh_degree and scale_modifier");
      return -2;
    }
    if (!geom_states[v] || !radii[v]) { gsr_set_error("gsr forward: NULL geom_state / radii"); return -2; }
    GeomState g;
    gsr_carve_geom(geom_states[v], P, &g);
    fill_pre_view(tab.v[v], cam, g, radii[v], sums + (size_t)v * nblk, colors_views ? colors_views[v] : nullptr);
    // Per-block words {differs, entry count} straight to pinned host memory: the caller's (capacity mode, V = 1) or this thread's own
    // (count-first mode, P <= 512 Ki: the host adds them up as soon as the preprocess blocks are through -- no copy on the stream, no
    // stream synchronisation; round 5: ~10 us per forward less than hipMemcpyAsync + hipStreamSynchronize)
    uint32_t* own_words = (num_rendered_host && gsr_host_block_scan(P) && !(skip && skip[v])) ? pinned_sums() : nullptr;
    if (differs_pinned && V == 1 && gsr_host_block_scan(P)) tab.v[v].block_hash = reinterpret_cast<uint2*>(differs_pinned);
    else if (own_words) {
      uint32_t* w = own_words + (size_t)v * 2 * GSR_HOST_SCAN_MAX_BLOCKS;
      for (uint32_t b = 0; b < nblk; ++b) w[2 * b + 1] = 0xffffffffu;
      tab.v[v].block_hash = reinterpret_cast<uint2*>(w);
    }
    if (V == 1 && tab.v[v].block_hash && (differs_pinned || same_host) && prev_geom) {
      {      // compare mode: this forward against an earlier one's geometry state
        GeomState pg;
        gsr_carve_geom(const_cast<void*>(prev_geom), P, &pg);
        tab.v[v].cmp_rec = pg.rec; tab.v[v].cmp_rect = pg.rect; tab.v[v].cmp_ekey = pg.ekey; tab.v[v].cmp_tiles = pg.tiles_touched;
      }
    }
    if (skip && skip[v]) tab.v[v].skip = 1;
  }
  if (int rc = gsr_launch_preprocess(tab, cam0, P, means3D, scales, rotations, opacities, colors_precomp, shs, cov3D_precomp, st))
    return rc;
  if (!num_rendered_host) {   // capacity mode: the counts stay on the device (emit_entries adds the block sums up itself)
    if (!gsr_host_block_scan(P))
      for (int v = 0; v < V; ++v) {
        GeomState g;
        gsr_carve_geom(geom_states[v], P, &g);
        if (int rc = gsr_launch_scan_exclusive(sums + (size_t)v * nblk, g.block_offsets, nblk, g.counters, st)) return rc;
      }
    return 0;
  }
  uint32_t* host = pinned_sums();
  if (!host) { gsr_set_error("gsr forward: pinned host allocation failed"); return -1; }
  if (same_host) *same_host = 0;
  if (gsr_host_block_scan(P)) {
    for (int v = 0; v < V; ++v) {
      uint64_t tot = 0;
      if (skip && skip[v]) {
        tot = num_rendered_host[geometry_of[v]];     // not preprocessed: its owner's lists are its lists
      } else {
        const volatile uint32_t* w = host + (size_t)v * 2 * GSR_HOST_SCAN_MAX_BLOCKS;
        int32_t any = 1;
        int64_t got = gsr_wait_block_counts(w, (int32_t)nblk, 2000, 30ll * 1000 * 1000, &any);
        if (got < 0) {      // thirty seconds without the stores: let the runtime say what happened to the stream
          GSR_HIP_CHECK(hipStreamSynchronize(st));
          got = gsr_wait_block_counts(w, (int32_t)nblk, 0, 1000, &any);
          if (got < 0) { gsr_set_error("gsr forward: the preprocess kernel's per-block counts never reached the host"); return -1; }
        }
        tot = (uint64_t)got;
        if (v == 0 && same_host && tab.v[0].cmp_rec) *same_host = any ? 0 : 1;
      }
      if (tot > 0xffffffffull) { gsr_set_error("gsr forward: %llu tile entries overflow 32 bits", (unsigned long long)tot); return -3; }
      num_rendered_host[v] = (uint32_t)tot;
    }
  } else {
    for (int v = 0; v < V; ++v) {
      GeomState g;
      gsr_carve_geom(geom_states[v], P, &g);
      if (int rc = gsr_launch_scan_exclusive(sums + (size_t)v * nblk, g.block_offsets, nblk, g.counters, st)) return rc;
      GSR_HIP_CHECK(hipMemcpyAsync(host + v, g.counters, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    }
    GSR_HIP_CHECK(hipStreamSynchronize(st));
    for (int v = 0; v < V; ++v) num_rendered_host[v] = (skip && skip[v]) ? host[geometry_of[v]] : host[v];
  }
  return 0;
}

// Stage 2 of V views: binning chain + blend, one launch per kernel for all views.
// `geometry_of` (may be NULL): geometry_of[v] = u <= v means view v has the same camera as view u and differs only in its
// colours (the colour and the segmentation render of get_loss): v then uses u's tile lists --
// no entries emitted, no global sort passes, no tile sort for v.
int check_geometry_of(int V, const int32_t* geometry_of) {
  if (!geometry_of) return 0;
  for (int v = 0; v < V; ++v) {
    const int u = geometry_of[v];
    if (u < 0 || u > v || geometry_of[u] != u) { gsr_set_error("gsr batch: geometry_of[%d] = %d is not an earlier primary view", v, u); return -2; }
  }
  return 0;
}
int stage2(int V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered, void* const* geom_states,
           void* const* binning_states, void* const* image_states, float* const* out_color, float* const* out_depth,
           const uint32_t* sums, uint4* order, uint32_t* queue, const int32_t* geometry_of, hipStream_t st,
           uint32_t* counts_dev = nullptr, uint32_t* tile_rows = nullptr, int flags = 0, const float* const* colors_views = nullptr) {   // tile_rows: the batch state's matrix (tile-row binning), or nullptr   // counts_dev != nullptr: capacity mode -- num_rendered[] are capacities, the counts go there
  if (int rc = check_geometry_of(V, geometry_of)) return rc;
  GsrBinViews bt;
  bt.vlong_out = nullptr; bt.vlong_launch = 0;      // (set by gsr_launch_binning: the long-list hint of tile_sort)
  bt.cut_lds = 0;
  GsrRenderViews rt;
  int partner[GSR_MAX_BATCH], fused[GSR_MAX_BATCH], skip[GSR_MAX_BATCH];
  pair_up(V, geometry_of, num_rendered, partner, fused);
  skippable_aliases(V, geometry_of, colors_views, flags, skip);
  const uint32_t nblk = (uint32_t)(((P > 0 ? P : 1) + GSR_BLOCK - 1) / GSR_BLOCK);
  for (int v = 0; v < V; ++v) {
    GsrCam cam;
    if (int rc = make_cam(&s[v], &cam)) return rc;
    if (!image_states[v] || !out_color[v] || !out_depth[v]) { gsr_set_error("gsr forward render: NULL argument"); return -2; }
    const int owner = geometry_of ? geometry_of[v] : v;   // the view whose tile lists this one uses
    if (num_rendered[v] > 0 && (!geom_states[v] || !binning_states[owner])) { gsr_set_error("gsr forward render: NULL state"); return -2; }
    if (owner != v && num_rendered[owner] != num_rendered[v]) { gsr_set_error("gsr batch: view %d shares view %d's camera but not its entry count", v, owner); return -2; }
    GeomState g; ImageState im, im_owner; BinningState bs;
    gsr_carve_geom(geom_states[v], P, &g);
    gsr_carve_image(image_states[v], cam.H, cam.W, &im);
    gsr_carve_image(image_states[owner], cam.H, cam.W, &im_owner);
    gsr_carve_binning(binning_states[owner], num_rendered[owner], &bs);
    if (v == 0) {
      bt.V = V; bt.T = cam.T; bt.gx = cam.gx; bt.counts_out = counts_dev; bt.P = P;
      bt.rows = tile_rows ? gsr_bin_rows(P) : 0;
      bt.forward_only = (flags & GSR_FORWARD_ONLY) ? 1 : 0;
      bt.order = order ? order : im.tile_order;
      bt.queue = queue ? queue : im.queue;
      render_header(rt, V, cam, bt.order, bt.queue);
    } else if (cam.W != rt.W || cam.H != rt.H) {
      gsr_set_error("gsr batch: all views must share the image size");
      return -2;
    }
    fill_bin_view(bt.v[v], P, num_rendered[v], g, bs, im, sums ? sums + (size_t)v * nblk : g.block_sums);
    if (tile_rows) bt.v[v].tile_rows = tile_rows + (size_t)v * (gsr_bin_rows(P) + 1) * (size_t)gsr_bin_stride(cam.T);
    bt.v[v].owner = owner;
    fill_render_view(rt.v[v], cam, g, bs, im, out_color[v], out_depth[v], nullptr, nullptr);
    if (owner != v) {   // lists, ranges and sort belong to the owner; this view only gets its offsets from emit
      bt.v[v].shares_lists = 1; bt.v[v].D = 0; bt.v[v].nblocks = 0; bt.v[v].ranges = im_owner.ranges;
      rt.v[v].ranges = im_owner.ranges;
    }
    bt.v[v].fused_alias = (uint32_t)fused[v];
    rt.v[v].partner = partner[v]; rt.v[v].fused_alias = fused[v];
    if (skip[v]) {      // no records of its own: only ever reached through its owner's tile pass (or painted as background)
      if (!fused[v] && num_rendered[v] > 0) { gsr_set_error("gsr batch: view %d was not preprocessed but is not fused", v); return -2; }
      rt.v[v].colors = colors_views[v];
    }
    if (counts_dev && owner == v) bt.v[v].D_dev = g.offsets + P;
  }
  if (t_cuts.V) {       // one-shot, whatever happens below
    const ArmedCuts ac = t_cuts;
    t_cuts.V = 0;
    if (ac.V != V || !(flags & GSR_FORWARD_ONLY) || !tile_rows) {
      gsr_set_error("gsr_arm_depth_cuts: the armed cuts need a forward-only call of %d views on the tile-row binning path", ac.V);
      return -2;
    }
    for (int v = 0; v < V; ++v) {
      if (bt.v[v].shares_lists || rt.v[v].partner >= 0 || rt.v[v].fused_alias) { gsr_set_error("gsr_arm_depth_cuts: views that share tile lists cannot be cut"); return -2; }
      bt.v[v].depth_cut = ac.in[v];
      rt.v[v].cut_in = ac.in[v]; rt.v[v].cut_out = ac.out[v]; rt.v[v].redo = ac.redo ? ac.redo + v : nullptr; rt.v[v].cut_margin = ac.margin;
    }
  }
  if (int rc = gsr_launch_binning(bt, P, st)) return rc;
  rt.track = (flags & GSR_FORWARD_ONLY) ? 0 : 1;
  return gsr_launch_render_fwd(rt, st);
}
}  // namespace
extern "C" {

int gsr_forward_preprocess(const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                           const float* rotations, const float* opacities, const float* colors_precomp,
                           const float* shs, const float* cov3D_precomp, void* geom_state, int32_t* radii,
                           uint32_t* num_rendered_host, void* stream) {
  return gsr_forward_preprocess_same(s, P, means3D, scales, rotations, opacities, colors_precomp, shs, cov3D_precomp, geom_state, radii,
                                     num_rendered_host, nullptr, nullptr, stream);
}

int gsr_forward_preprocess_same(const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                const float* rotations, const float* opacities, const float* colors_precomp,
                                const float* shs, const float* cov3D_precomp, void* geom_state, int32_t* radii,
                                uint32_t* num_rendered_host, const void* prev_geom_state, int32_t* same_host, void* stream) {
  GsrRange _range("gsr_forward_preprocess");
  if (same_host) *same_host = 0;
  GsrCam cam;
  if (int rc = make_cam(s, &cam)) return rc;
  if (num_rendered_host) *num_rendered_host = 0;
  if (P <= 0) return 0;
  if (!geom_state || !radii) { gsr_set_error("gsr_forward_preprocess: NULL argument"); return -2; }
  if (prev_geom_state == geom_state) { gsr_set_error("gsr_forward_preprocess_same: a forward cannot be compared with the state it writes"); return -2; }
  GeomState g;
  gsr_carve_geom(geom_state, P, &g);
  uint32_t D = 0;
  if (int rc = stage1(1, s, P, means3D, scales, rotations, opacities, colors_precomp, nullptr, shs, cov3D_precomp, &geom_state,
                      &radii, g.block_sums, &D, (hipStream_t)stream, nullptr, same_host, nullptr, nullptr, prev_geom_state))
    return rc;
  if (num_rendered_host) *num_rendered_host = D;
  return 0;
}

int gsr_forward_render(const gsr_settings* s, int32_t P, uint32_t num_rendered, const void* geom_state,
                       void* binning_state, void* image_state, float* out_color, float* out_depth, void* stream) {
  return gsr_forward_render_ex(s, P, num_rendered, geom_state, binning_state, image_state, out_color, out_depth, 0u, stream);
}
int gsr_forward_render_ex(const gsr_settings* s, int32_t P, uint32_t num_rendered, const void* geom_state, void* binning_state,
                          void* image_state, float* out_color, float* out_depth, uint32_t flags, void* stream) {
  GsrRange _range("gsr_forward_render");
  if (!s) { gsr_set_error("gsr: settings is NULL"); return -2; }
  void* geom = const_cast<void*>(geom_state);
  // round 4: the single-view entry points bin with the tile-row counting sort too (the matrix lives in the geometry state): one
  // binning = bin_count + bin_colprefix + bin_scan + bin_emit (with the tile order) ~ 37 us at 100 k / 800^2 instead of emit_entries +
  // radix_hist + 2 x radix_scatter + tile_order ~ 85 us.  Larger tile grids than GSR_BIN_MAX_T and GSR_RADIX_BINNING=1 keep the radix path.
  GeomState g;
  gsr_carve_geom(geom, P, &g);
  GsrCam cam;
  if (int rc = make_cam(s, &cam)) return rc;
  return stage2(1, s, P, &num_rendered, &geom, &binning_state, &image_state, &out_color, &out_depth, nullptr, nullptr,
                nullptr, nullptr, (hipStream_t)stream, nullptr, (geom && P > 0 && gsr_rows_path_ok(cam.T)) ? g.tile_rows : nullptr,
                (int)(flags & GSR_FORWARD_ONLY));
}

int gsr_forward_capacity(const gsr_settings* s, int32_t P, const float* means3D, const float* scales, const float* rotations,
                         const float* opacities, const float* colors_precomp, const float* shs, const float* cov3D_precomp,
                         void* geom_state, int32_t* radii, void* binning_state, uint32_t capacity_entries, void* image_state,
                         float* out_color, float* out_depth, const void* prev_geom_state, uint32_t* block_words_pinned,
                         int32_t* count_pinned, uint32_t flags, void* stream) {
  GsrRange _range("gsr_forward_capacity");
  GsrCam cam;
  if (int rc = make_cam(s, &cam)) return rc;
  if (P <= 0) { gsr_set_error("gsr_forward_capacity: P must be positive (use gsr_forward_preprocess / gsr_forward_render)"); return -2; }
  if (!geom_state || !radii || !binning_state || !image_state || !out_color || !out_depth || capacity_entries == 0) {
    gsr_set_error("gsr_forward_capacity: NULL argument or no capacity");
    return -2;
  }
  if (block_words_pinned && (reinterpret_cast<uintptr_t>(block_words_pinned) & 7u)) {
    gsr_set_error("gsr_forward_capacity: block_words_pinned must be 8-byte aligned (every block stores its word pair with one 8-byte store)");
    return -2;
  }
  const bool words = block_words_pinned && gsr_host_block_scan(P);
  if (!count_pinned && !words) { gsr_set_error("gsr_forward_capacity: count_pinned is needed (no block_words_pinned, or P > 512 Ki)"); return -2; }
  if (prev_geom_state == geom_state) { gsr_set_error("gsr_forward_capacity: a forward cannot be compared with the state it writes"); return -2; }
  GeomState g;
  gsr_carve_geom(geom_state, P, &g);
  if (int rc = stage1(1, s, P, means3D, scales, rotations, opacities, colors_precomp, nullptr, shs, cov3D_precomp, &geom_state, &radii,
                      g.block_sums, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr,
                      words ? prev_geom_state : nullptr, words ? block_words_pinned : nullptr))
    return rc;
  // (the tile-order kernel stores the count too: to the caller's pinned word, or -- nobody waits for it -- to a spare word of the geometry state)
  uint32_t* late_count = count_pinned ? reinterpret_cast<uint32_t*>(count_pinned) : g.counters + 8;
  return stage2(1, s, P, &capacity_entries, &geom_state, &binning_state, &image_state, &out_color, &out_depth, nullptr, nullptr, nullptr,
                nullptr, (hipStream_t)stream, late_count, gsr_rows_path_ok(cam.T) ? g.tile_rows : nullptr, (int)(flags & GSR_FORWARD_ONLY));
}

int gsr_forward_render_shared(const gsr_settings* s, int32_t P, uint32_t num_rendered, void* geom_state,
                              void* owner_binning_state, const void* owner_image_state, void* image_state, float* out_color,
                              float* out_depth, void* stream) {
  return gsr_forward_render_shared_ex(s, P, num_rendered, geom_state, owner_binning_state, owner_image_state, image_state, out_color, out_depth,
                                      0u, stream);
}
int gsr_forward_render_shared_ex(const gsr_settings* s, int32_t P, uint32_t num_rendered, void* geom_state, void* owner_binning_state,
                                 const void* owner_image_state, void* image_state, float* out_color, float* out_depth, uint32_t flags,
                                 void* stream) {
  GsrRange _range("gsr_forward_render_shared");
  GsrCam cam;
  if (int rc = make_cam(s, &cam)) return rc;
  if (P <= 0 || num_rendered == 0 || !geom_state || !owner_binning_state || !owner_image_state || !image_state || !out_color || !out_depth) {
    gsr_set_error("gsr_forward_render_shared: NULL argument, or nothing rendered (use gsr_forward_render)");
    return -2;
  }
  hipStream_t st = (hipStream_t)stream;
  GeomState g; ImageState im, im_owner; BinningState bs;
  gsr_carve_geom(geom_state, P, &g);
  gsr_carve_image(image_state, cam.H, cam.W, &im);
  gsr_carve_image(const_cast<void*>(owner_image_state), cam.H, cam.W, &im_owner);
  gsr_carve_binning(const_cast<void*>(owner_binning_state), num_rendered, &bs);
  GsrBinViews bt;
  bt.vlong_out = nullptr; bt.vlong_launch = 0;      // (set by gsr_launch_binning: the long-list hint of tile_sort)
  bt.V = 1; bt.T = cam.T; bt.gx = cam.gx; bt.counts_out = nullptr; bt.P = P; bt.rows = 0; bt.forward_only = 0; bt.wave_cap = 512; bt.cut_lds = 0;
  bt.order = im.tile_order; bt.queue = im.queue;
  fill_bin_view(bt.v[0], P, num_rendered, g, bs, im, g.block_sums);
  if (int rc = gsr_launch_shared_lists(bt, P, num_rendered, im_owner.ranges, im_owner.tile_order, im_owner.queue, im.ranges, im.tile_order,
                                       im.queue, st))
    return rc;
  GsrRenderViews rt;
  render_header(rt, 1, cam, im.tile_order, im.queue);
  rt.track = (flags & GSR_FORWARD_ONLY) ? 0 : 1;
  fill_render_view(rt.v[0], cam, g, bs, im, out_color, out_depth, nullptr, nullptr);
  return gsr_launch_render_fwd(rt, st);
}

int gsr_backward(const gsr_settings* s, int32_t P, uint32_t num_rendered, const float* means3D,
                 const float* scales, const float* rotations, const float* colors_precomp, const float* shs,
                 const float* cov3D_precomp, const int32_t* radii, const void* geom_state,
                 const void* binning_state, const void* image_state, const float* dL_dcolor, void* scratch,
                 float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dcolors, float* dL_dopacity,
                 float* dL_dscales, float* dL_drotations, float* dL_dcov3D, float* dL_dsh, void* stream) {
  GsrRange _range("gsr_backward");
  GsrCam cam;
  if (int rc = make_cam(s, &cam)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (P <= 0) return 0;
  if (!means3D || !radii || !geom_state || !image_state || !dL_dcolor || !dL_dmeans3D || !dL_dmeans2D || !dL_dopacity) {
    gsr_set_error("gsr_backward: NULL argument");
    return -2;
  }
  if (num_rendered > 0 && (!binning_state || !scratch)) { gsr_set_error("gsr_backward: NULL binning/scratch"); return -2; }
  GeomState g; ImageState im; BinningState bs;
  gsr_carve_geom(const_cast<void*>(geom_state), P, &g);
  gsr_carve_image(const_cast<void*>(image_state), cam.H, cam.W, &im);
  gsr_carve_binning(const_cast<void*>(binning_state), num_rendered, &bs);
  float4* partials = (float4*)scratch;
  if (num_rendered > 0) {
    GsrRenderViews rt;
    render_header(rt, 1, cam, im.tile_order, im.queue);
    rt.no_colour_grad = (!shs && !dL_dcolors) ? 1 : 0;   // precomputed colours and no gradient wanted for them
    rt.avg_list = num_rendered / (uint32_t)(cam.T > 0 ? cam.T : 1);
    fill_render_view(rt.v[0], cam, g, bs, im, nullptr, nullptr, dL_dcolor, partials);
    if (int rc = gsr_launch_render_bwd(rt, st)) return rc;
  }
  return gsr_launch_preprocess_bwd(cam, P, means3D, scales, rotations, colors_precomp, shs, cov3D_precomp, radii, g,
                                   partials, dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dscales,
                                   dL_drotations, dL_dcov3D, dL_dsh, num_rendered > 0 ? im.queue + GSR_QUEUE_BWD_ERROR : nullptr, st);
}

// ------------------------------------------------------------------------------------------ multi-view batch
// The views of one optimisation step share the Gaussians and are independent until the gradients are summed.
// Every stage of a batch call is ONE launch covering all views (tables above) on the caller's stream: no internal
// streams, no events, ~13 HIP calls per 4-view step.  `batch_state` (gsr_batch_state_bytes) holds what the views
// share: the per-block entry counts of all views (one D2H copy), the combined LPT tile order and its queue heads;
// the caller keeps it from the forward to the backward like the other state buffers.
size_t gsr_batch_state_bytes(int32_t V, int32_t P, int32_t H, int32_t W) { BatchState b; return gsr_carve_batch(nullptr, V, P, H, W, &b); }

static int check_batch(const char* who, int32_t V, const gsr_settings* s, const void* batch_state) {
  if (V <= 0 || V > GSR_MAX_BATCH) { gsr_set_error("%s: V must be in 1..%d", who, GSR_MAX_BATCH); return -2; }
  if (!s || !batch_state) { gsr_set_error("%s: NULL settings / batch_state", who); return -2; }
  return 0;
}

int gsr_arm_depth_cuts(int32_t V, const uint32_t* const* cut_in, uint32_t* const* cut_out, uint32_t* redo_flags, float margin) {
  if (V == 0) { t_cuts.V = 0; return 0; }      // disarm (a caller whose forward failed before it got to the binning)
  if (V < 1 || V > GSR_MAX_BATCH || !cut_out || !redo_flags || !(margin >= 1.0f) || !(margin < 1e6f)) { gsr_set_error("gsr_arm_depth_cuts: bad argument (margin >= 1)"); return -2; }
  t_cuts.V = V; t_cuts.redo = redo_flags; t_cuts.margin = margin;
  for (int v = 0; v < V; ++v) {
    if (!cut_out[v]) { t_cuts.V = 0; gsr_set_error("gsr_arm_depth_cuts: cut_out[%d] is NULL", v); return -2; }
    t_cuts.in[v] = cut_in ? cut_in[v] : nullptr;
    t_cuts.out[v] = cut_out[v];
  }
  return 0;
}
int gsr_forward_preprocess_batch(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                 const float* rotations, const float* opacities, const float* colors_precomp,
                                 const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                                 void* const* geom_states,
                                 int32_t* const* radii, void* batch_state, uint32_t* num_rendered_host, void* stream) {
  GsrRange _range("gsr_forward_preprocess_batch");
  if (int rc = check_batch("gsr_forward_preprocess_batch", V, s, batch_state)) return rc;
  if (!geom_states || !radii || !num_rendered_host) { gsr_set_error("gsr_forward_preprocess_batch: NULL argument"); return -2; }
  for (int v = 0; v < V; ++v) num_rendered_host[v] = 0;
  if (P <= 0) return 0;
  BatchState b;
  gsr_carve_batch(batch_state, V, P, s[0].image_height, s[0].image_width, &b);
  // (no geometry_of at this entry point: every view counts its rows -- a view that turns out to share lists leaves its rows unused)
  return stage1(V, s, P, means3D, scales, rotations, opacities, colors_precomp, colors_views, shs, cov3D_precomp, geom_states,
                radii, b.sums, num_rendered_host, (hipStream_t)stream);
}

int gsr_forward_render_batch(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered,
                             void* const* geom_states, void* const* binning_states, void* const* image_states,
                             void* batch_state, const int32_t* geometry_of, const float* const* colors_views, float* const* out_color,
                             float* const* out_depth, int32_t flags, void* stream) {
  GsrRange _range("gsr_forward_render_batch");
  if (int rc = check_batch("gsr_forward_render_batch", V, s, batch_state)) return rc;
  if (!num_rendered || !geom_states || !binning_states || !image_states || !out_color || !out_depth) {
    gsr_set_error("gsr_forward_render_batch: NULL argument");
    return -2;
  }
  BatchState b;
  gsr_carve_batch(batch_state, V, P, s[0].image_height, s[0].image_width, &b);
  return stage2(V, s, P, num_rendered, geom_states, binning_states, image_states, out_color, out_depth, b.sums, b.order,
                b.queue, geometry_of, (hipStream_t)stream, nullptr, b.tile_rows, flags, colors_views);
}

int gsr_forward_batch(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                      const float* rotations, const float* opacities, const float* colors_precomp,
                      const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                      void* const* geom_states, int32_t* const* radii,
                      void* const* binning_states, const size_t* binning_bytes, void* const* image_states,
                      void* batch_state, const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                      uint32_t* num_rendered_host, int32_t flags, void* stream) {
  GsrRange _range("gsr_forward_batch");
  if (int rc = check_batch("gsr_forward_batch", V, s, batch_state)) return rc;
  if (int rc = check_geometry_of(V, geometry_of)) return rc;
  if (!geom_states || !radii || !num_rendered_host || !image_states || !out_color || !out_depth) {
    gsr_set_error("gsr_forward_batch: NULL argument");
    return -2;
  }
  for (int v = 0; v < V; ++v) num_rendered_host[v] = 0;
  if (P <= 0) return 1;  // nothing to preprocess: the caller takes the render-stage call (it paints the background)
  BatchState b;
  gsr_carve_batch(batch_state, V, P, s[0].image_height, s[0].image_width, &b);
  int skip[GSR_MAX_BATCH];
  skippable_aliases(V, geometry_of, colors_views, flags, skip);
  if (int rc = stage1(V, s, P, means3D, scales, rotations, opacities, colors_precomp, colors_views, shs, cov3D_precomp,
                      geom_states, radii, b.sums, num_rendered_host, (hipStream_t)stream, nullptr, nullptr, skip, geometry_of))
    return rc;
  bool fits = binning_states != nullptr && binning_bytes != nullptr;
  for (int v = 0; fits && v < V; ++v) {
    if (geometry_of && geometry_of[v] != v) continue;   // uses its owner's binning state
    fits = num_rendered_host[v] == 0 || (binning_states[v] && binning_bytes[v] >= gsr_binning_bytes(num_rendered_host[v], 0, 0));
  }
  if (!fits) return 1;
  return stage2(V, s, P, num_rendered_host, geom_states, binning_states, image_states, out_color, out_depth, b.sums, b.order,
                b.queue, geometry_of, (hipStream_t)stream, nullptr, b.tile_rows, flags, colors_views);
}

int gsr_forward_batch_capacity(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                               const float* rotations, const float* opacities, const float* colors_precomp,
                               const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                               void* const* geom_states, int32_t* const* radii, void* const* binning_states,
                               const uint32_t* capacity_entries, void* const* image_states, void* batch_state,
                               const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                               uint32_t* counts_dev, void* stream) {
  return gsr_forward_batch_capacity_raw(V, s, P, means3D, scales, rotations, opacities, colors_precomp, colors_views, shs, cov3D_precomp,
                                        geom_states, radii, binning_states, capacity_entries, image_states, batch_state, geometry_of,
                                        out_color, out_depth, counts_dev, nullptr, stream);
}

int gsr_forward_batch_capacity_raw(int32_t V, const gsr_settings* s, int32_t P, const float* means3D, const float* scales,
                                   const float* rotations, const float* opacities, const float* colors_precomp,
                                   const float* const* colors_views, const float* shs, const float* cov3D_precomp,
                                   void* const* geom_states, int32_t* const* radii, void* const* binning_states,
                                   const uint32_t* capacity_entries, void* const* image_states, void* batch_state,
                                   const int32_t* geometry_of, float* const* out_color, float* const* out_depth,
                                   uint32_t* counts_dev, const gsr_raw_params* raw, void* stream) {
  GsrRange _range("gsr_forward_batch_capacity");
  if (int rc = check_batch("gsr_forward_batch_capacity", V, s, batch_state)) return rc;
  if (int rc = check_geometry_of(V, geometry_of)) return rc;
  if (!geom_states || !radii || !binning_states || !capacity_entries || !image_states || !out_color || !out_depth || !counts_dev) {
    gsr_set_error("gsr_forward_batch_capacity: NULL argument");
    return -2;
  }
  if (P <= 0) { gsr_set_error("gsr_forward_batch_capacity: P must be positive (use gsr_forward_batch)"); return -2; }
  for (int v = 0; v < V; ++v) {
    const bool owner = !geometry_of || geometry_of[v] == v;
    if (capacity_entries[v] == 0 || (owner && !binning_states[v])) { gsr_set_error("gsr_forward_batch_capacity: view %d has no capacity / binning state", v); return -2; }
    if (!owner && capacity_entries[v] != capacity_entries[geometry_of[v]]) { gsr_set_error("gsr_forward_batch_capacity: view %d must have its owner's capacity", v); return -2; }
  }
  BatchState b;
  gsr_carve_batch(batch_state, V, P, s[0].image_height, s[0].image_width, &b);
  if (int rc = stage1(V, s, P, means3D, scales, rotations, opacities, colors_precomp, colors_views, shs, cov3D_precomp,
                      geom_states, radii, b.sums, nullptr, (hipStream_t)stream, raw, nullptr, nullptr, geometry_of))
    return rc;
  return stage2(V, s, P, capacity_entries, geom_states, binning_states, image_states, out_color, out_depth, b.sums, b.order,
                b.queue, geometry_of, (hipStream_t)stream, counts_dev, b.tile_rows);
}

int gsr_backward_batch(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered, const float* means3D,
                       const float* scales, const float* rotations, const float* colors_precomp,
                       const float* cov3D_precomp, const int32_t* const* radii, void* const* geom_states,
                       void* const* binning_states, void* const* image_states, void* batch_state,
                       const int32_t* geometry_of, const float* const* dL_dcolor, void* const* scratch, float* dL_dmeans3D,
                       float* const* dL_dmeans2D,
                       float* dL_dcolors, float* const* dL_dcolors_views, float* dL_dopacity, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D, void* stream) {
  return gsr_backward_batch_raw(V, s, P, num_rendered, means3D, scales, rotations, colors_precomp, cov3D_precomp, radii, geom_states,
                                binning_states, image_states, batch_state, geometry_of, dL_dcolor, scratch, dL_dmeans3D, dL_dmeans2D, dL_dcolors,
                                dL_dcolors_views, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, nullptr, stream);
}

int gsr_backward_batch_raw(int32_t V, const gsr_settings* s, int32_t P, const uint32_t* num_rendered, const float* means3D,
                       const float* scales, const float* rotations, const float* colors_precomp,
                       const float* cov3D_precomp, const int32_t* const* radii, void* const* geom_states,
                       void* const* binning_states, void* const* image_states, void* batch_state,
                       const int32_t* geometry_of, const float* const* dL_dcolor, void* const* scratch, float* dL_dmeans3D,
                       float* const* dL_dmeans2D,
                       float* dL_dcolors, float* const* dL_dcolors_views, float* dL_dopacity, float* dL_dscales,
                       float* dL_drotations, float* dL_dcov3D, const gsr_raw_params* raw, void* stream) {
  GsrRange _range("gsr_backward_batch");
  if (int rc = check_batch("gsr_backward_batch", V, s, batch_state)) return rc;
  if (int rc = check_geometry_of(V, geometry_of)) return rc;
  if (!num_rendered || !radii || !geom_states || !binning_states || !image_states || !dL_dcolor || !scratch ||
      !dL_dmeans3D || !dL_dmeans2D || (!dL_dopacity && !raw) || !means3D) {
    gsr_set_error("gsr_backward_batch: NULL argument");
    return -2;
  }
  if (raw && (!raw->unnorm_rotations || !raw->d_unnorm_rotations || !raw->d_logit_opacities || !raw->d_log_scales || !scales || !rotations ||
              !raw->opacities_out || cov3D_precomp)) {
    gsr_set_error("gsr_backward_batch (raw parameters): NULL pointer or cov3D_precomp given");
    return -2;
  }
  if (P <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  BatchState b;
  gsr_carve_batch(batch_state, V, P, s[0].image_height, s[0].image_width, &b);
  GsrBwdViews vw;
  vw.V = V;
  vw.bwd_error = nullptr;
  vw.raw_rot = raw ? raw->unnorm_rotations : nullptr; vw.act_op = raw ? raw->opacities_out : nullptr; vw.act_sc = raw ? scales : nullptr;
  vw.d_raw_rot = raw ? raw->d_unnorm_rotations : nullptr; vw.d_raw_op = raw ? raw->d_logit_opacities : nullptr;
  vw.d_raw_sc = raw ? raw->d_log_scales : nullptr;
  GsrRenderViews rt;
  GsrBinViews bt;          // only for a tile_order rebuild (ranges + flags)
  bt.vlong_out = nullptr; bt.vlong_launch = 0;      // (set by gsr_launch_binning: the long-list hint of tile_sort)
  bt.cut_lds = 0;
  bool any = false;
  // Pairs fused by the forward (pair_up) stay fused in the backward when no colour gradient is wanted (the pair pass carries
  // none); otherwise every view takes its own pass over an LPT order rebuilt WITH the partners' tickets, and the fused order is
  // put back afterwards (batch_state always holds the forward's order between calls: two tile_order launches on the rare path,
  // none on the common one).
  int partner[GSR_MAX_BATCH], fused[GSR_MAX_BATCH];
  pair_up(V, geometry_of, num_rendered, partner, fused);
  bool pairs_fwd = false;
  for (int v = 0; v < V; ++v) pairs_fwd = pairs_fwd || fused[v];
  const bool fuse_bwd = pairs_fwd && !dL_dcolors && !dL_dcolors_views;
  if (!fuse_bwd)
    for (int v = 0; v < V; ++v) { partner[v] = -1; fused[v] = 0; }
  for (int v = 0; v < V; ++v) {
    GsrCam cam;
    if (int rc = make_cam(&s[v], &cam)) return rc;
    const int owner = geometry_of ? geometry_of[v] : v;
    GeomState g; ImageState im, im_owner; BinningState bs;
    gsr_carve_geom(geom_states[v], P, &g);
    gsr_carve_image(image_states[v], cam.H, cam.W, &im);
    gsr_carve_image(image_states[owner], cam.H, cam.W, &im_owner);
    gsr_carve_binning(binning_states[owner], num_rendered[owner], &bs);
    if (num_rendered[v] > 0 && (!binning_states[owner] || !scratch[v])) { gsr_set_error("gsr_backward_batch: NULL binning/scratch"); return -2; }
    if (v == 0) { render_header(rt, V, cam, b.order, b.queue); rt.no_colour_grad = (!dL_dcolors && !dL_dcolors_views) ? 1 : 0; }
    fill_render_view(rt.v[v], cam, g, bs, im, nullptr, nullptr, dL_dcolor[v], (float4*)scratch[v]);
    rt.v[v].ranges = im_owner.ranges;
    rt.v[v].partner = partner[v]; rt.v[v].fused_alias = fused[v];
    if (v == 0) { bt.V = V; bt.T = cam.T; bt.gx = cam.gx; bt.order = b.order; bt.queue = b.queue; bt.counts_out = nullptr; bt.P = P; bt.wave_cap = 512; bt.rows = 0; bt.forward_only = 0; }
    bt.v[v].ranges = im_owner.ranges; bt.v[v].fused_alias = (uint32_t)fused[v]; bt.v[v].shares_lists = owner != v;
    any = any || num_rendered[v] > 0;
    GsrBwdView& w = vw.v[v];
    w.view = cam.view; w.proj = cam.proj; w.radii = radii[v]; w.offsets = g.offsets;
    w.used = g.used; w.tracked = g.counters + 1;
    w.partials = (const float4*)scratch[v]; w.dL_dmeans2D = dL_dmeans2D[v];
    w.dL_dcolors = dL_dcolors_views ? dL_dcolors_views[v] : nullptr;
    w.partner_dL_dmeans2D = partner[v] >= 0 ? dL_dmeans2D[partner[v]] : nullptr;
    w.fused_alias = fused[v];
    w.cap = num_rendered[v];
    w.W = cam.W; w.H = cam.H; w.tanfovx = cam.tanfovx; w.tanfovy = cam.tanfovy;
  }
  if (any) {
    { uint64_t tot = 0; for (int v = 0; v < V; ++v) tot += num_rendered[v]; rt.avg_list = (uint32_t)(tot / ((uint64_t)V * (uint64_t)(rt.T > 0 ? rt.T : 1))); }
    const bool rebuild = pairs_fwd && !fuse_bwd;
    if (rebuild)
      if (int rc = gsr_launch_tile_order(bt, st)) return rc;
    if (int rc = gsr_launch_render_bwd(rt, st)) return rc;
    vw.bwd_error = rt.queue + GSR_QUEUE_BWD_ERROR;
    if (rebuild) {
      pair_up(V, geometry_of, num_rendered, partner, fused);
      for (int v = 0; v < V; ++v) bt.v[v].fused_alias = (uint32_t)fused[v];
      if (int rc = gsr_launch_tile_order(bt, st)) return rc;
    }
  }
  (void)colors_precomp;
  return gsr_launch_preprocess_bwd_views(vw, P, s[0].scale_modifier, means3D, scales, rotations, cov3D_precomp, dL_dmeans3D,
                                         dL_dcolors, dL_dopacity, dL_dscales, dL_drotations, dL_dcov3D, st);
}

int32_t gsr_rigidity_blocks(int32_t n_fg) { return gsr_rigidity_fwd_blocks(n_fg); }

int gsr_rigidity_forward(int32_t n_fg, int32_t K, const float* means3D, const float* rotations, const int64_t* fg_idx,
                         const int64_t* neighbor_indices, const float* neighbor_weight, const float* neighbor_dist,
                         const float* prev_inv_rot_fg, const float* prev_offset, float* block_partials, void* stream) {
  GsrRange _range("gsr_rigidity_forward");
  if (n_fg < 0 || K <= 0 || (n_fg > 0 && (!means3D || !rotations || !fg_idx || !neighbor_indices || !neighbor_weight || !neighbor_dist ||
                                          !prev_inv_rot_fg || !prev_offset || !block_partials))) {
    gsr_set_error("gsr_rigidity_forward: bad argument");
    return -2;
  }
  return gsr_launch_rigidity_fwd(n_fg, K, means3D, rotations, fg_idx, neighbor_indices, neighbor_weight, neighbor_dist, prev_inv_rot_fg,
                                 prev_offset, nullptr, block_partials, (hipStream_t)stream);
}

int gsr_rigidity_backward(int32_t n_fg, int32_t K, const float* means3D, const float* rotations, const int64_t* fg_idx,
                          const int64_t* neighbor_indices, const float* neighbor_weight, const float* neighbor_dist,
                          const float* prev_inv_rot_fg, const float* prev_offset, const float* grad3, const int32_t* rev_ptr,
                          const int32_t* rev_edge, float* scratch, float* d_means3D, float* d_rotations, void* stream) {
  GsrRange _range("gsr_rigidity_backward");
  if (n_fg < 0 || K <= 0 || (n_fg > 0 && (!means3D || !rotations || !fg_idx || !neighbor_indices || !neighbor_weight || !neighbor_dist ||
                                          !prev_inv_rot_fg || !prev_offset || !grad3 || !rev_ptr || !rev_edge || !scratch || !d_means3D ||
                                          !d_rotations))) {
    gsr_set_error("gsr_rigidity_backward: bad argument");
    return -2;
  }
  float* self7 = scratch;
  float* edge7 = scratch + (size_t)7 * n_fg;
  return gsr_launch_rigidity_bwd(n_fg, K, means3D, rotations, fg_idx, neighbor_indices, neighbor_weight, neighbor_dist, prev_inv_rot_fg,
                                 prev_offset, grad3, 1, 1.0f, 1.0f, 1.0f, rev_ptr, rev_edge, nullptr, 0, self7, edge7, d_means3D, d_rotations, 0, (hipStream_t)stream);
}

int gsr_activate_forward(int32_t P, const float* unnorm_rotations, const float* logit_opacities, const float* log_scales,
                         float* rotations, float* opacities, float* scales, void* stream) {
  GsrRange _range("gsr_activate_forward");
  if (P < 0 || (P > 0 && (!unnorm_rotations || !logit_opacities || !log_scales || !rotations || !opacities || !scales))) {
    gsr_set_error("gsr_activate_forward: bad argument");
    return -2;
  }
  return gsr_launch_activate_fwd(P, unnorm_rotations, logit_opacities, log_scales, rotations, opacities, scales, (hipStream_t)stream);
}

int gsr_activate_backward(int32_t P, const float* unnorm_rotations, const float* opacities, const float* scales,
                          const float* d_rotations, const float* d_opacities, const float* d_scales, float* d_unnorm_rotations,
                          float* d_logit_opacities, float* d_log_scales, void* stream) {
  GsrRange _range("gsr_activate_backward");
  if (P < 0 || (P > 0 && (!unnorm_rotations || !opacities || !scales || !d_unnorm_rotations || !d_logit_opacities || !d_log_scales))) {
    gsr_set_error("gsr_activate_backward: bad argument");
    return -2;
  }
  return gsr_launch_activate_bwd(P, unnorm_rotations, opacities, scales, d_rotations, d_opacities, d_scales, d_unnorm_rotations,
                                 d_logit_opacities, d_log_scales, (hipStream_t)stream);
}

int32_t gsr_shared_terms_scratch(int32_t n_fg, int32_t K) {
  const int64_t n = n_fg > 0 ? n_fg : 0;
  return (int32_t)(16 * n + 8 * (n + n * (K > 0 ? K : 0)));
}

int32_t gsr_shared_terms_partials(int32_t n_fg, int32_t n_bg) {
  return 16 * (n_fg > 0 ? n_fg : 0) + 3 * (gsr_rigidity_fwd_blocks(n_fg) + gsr_shared_terms_point_blocks(n_fg, n_bg));
}

static int shared_terms_check(const char* who, int32_t n_fg, int32_t K, int32_t n_bg, const void* const* ptrs, int n) {
  if (n_fg < 0 || n_bg < 0 || K <= 0) { gsr_set_error("%s: bad sizes", who); return -2; }
  for (int i = 0; i < n; ++i)
    if (!ptrs[i]) { gsr_set_error("%s: NULL argument (#%d)", who, i); return -2; }
  return 0;
}

int gsr_shared_terms_forward(int32_t n_fg, int32_t K, int32_t n_bg, const float* means3D, const float* rotations, const int64_t* fg_idx,
                             const int64_t* bg_idx, const int64_t* neighbor_indices, const float* neighbor_weight,
                             const float* neighbor_dist, const float* prev_inv_rot_fg, const float* prev_offset,
                             const float* init_bg_pts, const float* init_bg_rot, const float* weights5_host, float* partials,
                             float* terms6, void* stream) {
  GsrRange _range("gsr_shared_terms_forward");
  const void* ptrs[] = {means3D, rotations, fg_idx, bg_idx, neighbor_indices, neighbor_weight, neighbor_dist, prev_inv_rot_fg,
                        prev_offset, init_bg_pts, init_bg_rot, weights5_host, partials, terms6};
  if (int e = shared_terms_check("gsr_shared_terms_forward", n_fg, K, n_bg, ptrs, 14)) return e;
  return gsr_launch_shared_terms_fwd(n_fg, K, n_bg, means3D, rotations, fg_idx, bg_idx, neighbor_indices, neighbor_weight,
                                     neighbor_dist, prev_inv_rot_fg, prev_offset, init_bg_pts, init_bg_rot, weights5_host, partials,
                                     terms6, (hipStream_t)stream);
}

int gsr_shared_terms_backward(int32_t P, int32_t n_fg, int32_t K, int32_t n_bg, const float* means3D, const float* rotations,
                              const int64_t* fg_idx, const int64_t* bg_idx, const int64_t* neighbor_indices,
                              const float* neighbor_weight, const float* neighbor_dist, const float* prev_inv_rot_fg,
                              const float* prev_offset, const float* init_bg_pts, const float* init_bg_rot,
                              const float* weights5_host, const float* grad_total, const int32_t* rev_ptr, const int32_t* rev_edge,
                              float* scratch, float* d_means3D, float* d_rotations, int32_t flags, void* stream) {
  GsrRange _range("gsr_shared_terms_backward");
  const void* ptrs[] = {means3D, rotations, fg_idx, bg_idx, neighbor_indices, neighbor_weight, neighbor_dist, prev_inv_rot_fg,
                        prev_offset, init_bg_pts, init_bg_rot, weights5_host, grad_total, rev_ptr, rev_edge, scratch, d_means3D,
                        d_rotations};
  if (int e = shared_terms_check("gsr_shared_terms_backward", n_fg, K, n_bg, ptrs, 18)) return e;
  if (P < n_fg + n_bg) { gsr_set_error("gsr_shared_terms_backward: P < n_fg + n_bg"); return -2; }
  return gsr_launch_shared_terms_bwd(P, n_fg, K, n_bg, means3D, rotations, fg_idx, bg_idx, neighbor_indices, neighbor_weight,
                                     neighbor_dist, prev_inv_rot_fg, prev_offset, init_bg_pts, init_bg_rot, weights5_host, grad_total,
                                     rev_ptr, rev_edge, scratch, d_means3D, d_rotations, flags, (hipStream_t)stream);
}

int gsr_radius_bookkeeping(int32_t V, int32_t view_step, int32_t P, const int32_t* radii, float* max_2D_radius, uint8_t* seen,
                           void* stream) {
  GsrRange _range("gsr_radius_bookkeeping");
  if (V < 0 || view_step <= 0 || P < 0 || (V > 0 && P > 0 && (!radii || !max_2D_radius || !seen))) {
    gsr_set_error("gsr_radius_bookkeeping: bad argument");
    return -2;
  }
  return gsr_launch_radius_bookkeeping(V, view_step, P, radii, max_2D_radius, seen, (hipStream_t)stream);
}

int gsr_adam_step(int32_t n_tensors, const gsr_adam_tensor* tensors, void* stream) {
  GsrRange _range("gsr_adam_step");
  if (n_tensors < 0 || n_tensors > GSR_ADAM_MAX_TENSORS || (n_tensors > 0 && !tensors)) {
    gsr_set_error("gsr_adam_step: 0..%d tensors per call", GSR_ADAM_MAX_TENSORS);
    return -2;
  }
  for (int i = 0; i < n_tensors; ++i) {
    const gsr_adam_tensor& t = tensors[i];
    if (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)) { gsr_set_error("gsr_adam_step: NULL pointer in tensor %d", i); return -2; }
    if (t.n > 0 && !(t.bias_correction1 > 0.f && t.bias_correction2_sqrt > 0.f)) { gsr_set_error("gsr_adam_step: bias corrections of tensor %d must be positive (step >= 1)", i); return -2; }
  }
  return gsr_launch_adam_step(n_tensors, tensors, (hipStream_t)stream);
}

size_t gsr_fps_scratch_bytes(int32_t N, int32_t npoints) { return gsr_fps_scratch_size(N, npoints); }

int gsr_fit_rotations(int32_t n_bones, const float* moments, const float* n_related, float* rotations, int32_t* code, void* stream) {
  GsrRange _range("gsr_fit_rotations");
  if (n_bones < 0 || (n_bones > 0 && (!moments || !n_related || !rotations || !code))) { gsr_set_error("gsr_fit_rotations: bad argument"); return -2; }
  return gsr_launch_fit_rotations(n_bones, moments, n_related, rotations, (int*)code, (hipStream_t)stream);
}

int gsr_fps_thin(int32_t N, const float* pos, int32_t npoints, int32_t start_idx, float radius, int32_t thin_start_idx, int64_t* out_idx,
                 int64_t* thin_idx, int32_t* thin_count, void* stream) {
  GsrRange _range("gsr_fps_thin");
  if (N <= 0 || N > 1024 || npoints <= 0 || npoints > N || !pos || !out_idx || !thin_idx || !thin_count || start_idx < 0 || start_idx >= N ||
      thin_start_idx < 0 || thin_start_idx >= npoints) {
    gsr_set_error("gsr_fps_thin: bad argument (1 <= npoints <= N <= 1024, start indices in range)");
    return -2;
  }
  return gsr_launch_fps_thin(N, pos, npoints, start_idx, radius, thin_start_idx, (long long*)out_idx, (long long*)thin_idx, (int*)thin_count,
                             (hipStream_t)stream);
}

int gsr_lbs_valid(int32_t P, int32_t n_bones, const int32_t* n_valid, const float* bones, const float* rotations, const float* translations,
                  const float* bone_quats, const float* xyz, const float* quat, float* out_xyz, float* out_quat, void* stream) {
  GsrRange _range("gsr_lbs");
  if (P < 0 || n_bones <= 0 || !n_valid || !bones || !rotations || !translations || !bone_quats || (P > 0 && (!xyz || !out_xyz))) {
    gsr_set_error("gsr_lbs_valid: bad argument");
    return -2;
  }
  return gsr_launch_lbs(P, n_bones, bones, rotations, translations, bone_quats, xyz, quat, out_xyz, out_quat, (hipStream_t)stream, (const int*)n_valid);
}

int gsr_construct_edges(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                        int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, void* stream) {
  GsrRange _range("gsr_construct_edges");
  if (!positions || !n_valid || !receivers || !senders || !count || n_obj_cap < 1 || n_obj_cap > 127 || topk < 1 || topk > 16 || e_cap < 1) {
    gsr_set_error("gsr_construct_edges: bad argument (1 <= n_obj_cap <= 127, 1 <= topk <= 16)");
    return -2;
  }
  return gsr_launch_construct_edges(positions, n_obj_cap, (const int*)n_valid, thresh_sq, topk, (long long)dummy_index, e_cap, (long long*)receivers,
                                    (long long*)senders, (int*)count, nullptr, 0, (hipStream_t)stream);
}
int gsr_construct_edges_dense(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                              int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, int64_t* relations, int32_t relations_n,
                              void* stream) {
  GsrRange _range("gsr_construct_edges");
  if (!positions || !n_valid || !receivers || !senders || !count || !relations || n_obj_cap < 1 || n_obj_cap > 127 || topk < 1 || topk > 16 || e_cap < 1 ||
      relations_n < n_obj_cap + 1) {
    gsr_set_error("gsr_construct_edges_dense: bad argument (1 <= n_obj_cap <= 127, 1 <= topk <= 16, relations_n > n_obj_cap)");
    return -2;
  }
  return gsr_launch_construct_edges(positions, n_obj_cap, (const int*)n_valid, thresh_sq, topk, (long long)dummy_index, e_cap, (long long*)receivers,
                                    (long long*)senders, (int*)count, (long long*)relations, relations_n, (hipStream_t)stream);
}
int gsr_construct_edges_rows(const float* positions, int32_t n_obj_cap, const int32_t* n_valid, float thresh_sq, int32_t topk, int64_t dummy_index,
                             int32_t e_cap, int64_t* receivers, int64_t* senders, int32_t* count, int64_t* relations, int32_t relations_n,
                             int64_t* row_start, void* stream) {
  GsrRange _range("gsr_construct_edges");
  if (!positions || !n_valid || !receivers || !senders || !count || !relations || !row_start || n_obj_cap < 1 || n_obj_cap > 127 || topk < 1 || topk > 16 ||
      e_cap < 1 || relations_n < n_obj_cap + 1 || dummy_index < n_obj_cap + 1 || dummy_index >= relations_n) {
    gsr_set_error("gsr_construct_edges_rows: bad argument (1 <= n_obj_cap <= 127, 1 <= topk <= 16, n_obj_cap < dummy_index < relations_n)");
    return -2;
  }
  return gsr_launch_construct_edges(positions, n_obj_cap, (const int*)n_valid, thresh_sq, topk, (long long)dummy_index, e_cap, (long long*)receivers,
                                    (long long*)senders, (int*)count, (long long*)relations, relations_n, (hipStream_t)stream, (long long*)row_start);
}
int gsr_rollout_step_head(int32_t n_track, int32_t n_his, int32_t n_bones, int32_t n_rows, int32_t attr_dim, int32_t with_state, const float* hist,
                          const int64_t* sample_idx, const int64_t* thin_idx, const float* eef_hist, const float* eef_next, const float* attrs,
                          const float* instance, float* bones_last, float* states_last, float* state_rows, float* action_rows, float* particle_inputs,
                          float* rel_nodes, void* stream) {
  GsrRange _range("gsr_rollout_step_head");
  if (n_track < 1 || n_his < 1 || n_bones < 1 || n_rows < n_bones + 1 || attr_dim < 0 || !hist || !sample_idx || !thin_idx || !eef_hist || !eef_next ||
      (attr_dim > 0 && !attrs) || !instance || !bones_last || !states_last || !state_rows || !action_rows || !particle_inputs || !rel_nodes) {
    gsr_set_error("gsr_rollout_step_head: bad argument (n_rows > n_bones)");
    return -2;
  }
  return gsr_launch_rollout_head(n_track, n_his, n_bones, n_rows, attr_dim, with_state ? 1 : 0, hist, (const long long*)sample_idx, (const long long*)thin_idx,
                                 eef_hist, eef_next, attrs, instance, bones_last, states_last, state_rows, action_rows, particle_inputs, rel_nodes,
                                 (hipStream_t)stream);
}
int gsr_rollout_step_motion(int32_t n_bones, int32_t n_his, float motion_clamp, const float* state_rows, const float* pred_motion, const int32_t* n_valid,
                            float* skin_packet, void* stream) {
  GsrRange _range("gsr_rollout_step_motion");
  if (n_bones < 1 || n_his < 1 || !(motion_clamp >= 0.0f) || !state_rows || !pred_motion || !n_valid || !skin_packet) {
    gsr_set_error("gsr_rollout_step_motion: bad argument");
    return -2;
  }
  return gsr_launch_rollout_motion(n_bones, n_his, motion_clamp, state_rows, pred_motion, (const int*)n_valid, skin_packet, (hipStream_t)stream);
}
int gsr_rollout_step_tail(int32_t n_track, int32_t n_his, int32_t n_bones, const float* all_pos, const int64_t* track, float* pos_track, float* hist,
                          float* eef_hist, const float* eef_next, const float* pred_in, const int32_t* n_valid, const int32_t* code, float* pred_out,
                          int32_t* n_valid_out, int64_t* bad, void* stream) {
  GsrRange _range("gsr_rollout_step_tail");
  if (n_track < 0 || n_his < 1 || n_bones < 0 || !all_pos || !track || !pos_track || !hist || !eef_hist || !eef_next || !pred_in || !n_valid || !code ||
      !pred_out || !n_valid_out || !bad) {
    gsr_set_error("gsr_rollout_step_tail: bad argument");
    return -2;
  }
  return gsr_launch_rollout_tail(n_track, n_his, n_bones, all_pos, (const long long*)track, pos_track, hist, eef_hist, eef_next, pred_in,
                                 (const int*)n_valid, (const int*)code, pred_out, (int*)n_valid_out, (long long*)bad, (hipStream_t)stream);
}

int gsr_fit_bones(int32_t n_bones, const float* bones, const float* motions, const int64_t* relations, int64_t relations_row_stride,
                  float* rotations, float* quats, int32_t* code, void* stream) {
  GsrRange _range("gsr_fit_bones");
  if (n_bones < 0 || (n_bones > 0 && (!bones || !motions || !relations || !rotations || !quats || !code)) || relations_row_stride < n_bones) {
    gsr_set_error("gsr_fit_bones: bad argument");
    return -2;
  }
  return gsr_launch_fit_bones(n_bones, bones, motions, (const long long*)relations, (long long)relations_row_stride, rotations, quats, (int*)code,
                              (hipStream_t)stream);
}

int gsr_gnn_aggregate(int32_t n_rows, int32_t n_sum_rows, int32_t width, const float* rel_part, const float* node_parts, const int64_t* senders,
                      const int64_t* row_start, float* agg, void* stream) {
  GsrRange _range("gsr_gnn_aggregate");
  if (n_rows <= 0 || n_sum_rows < 0 || n_sum_rows > n_rows || width <= 0 || (width & 3) || !rel_part || !node_parts || !senders || !row_start || !agg) {
    gsr_set_error("gsr_gnn_aggregate: bad argument (width a multiple of 4)");
    return -2;
  }
  return gsr_launch_gnn_aggregate(n_rows, n_sum_rows, width, rel_part, node_parts, (const long long*)senders, (const long long*)row_start, agg, (hipStream_t)stream);
}
int gsr_gnn_aggregate_res(int32_t n_rows, int32_t n_sum_rows, int32_t width, const float* rel_part, const float* node_parts, const int64_t* senders,
                          const int64_t* row_start, float* agg, const float* res_a, const float* res_b, float* res_out, void* stream) {
  GsrRange _range("gsr_gnn_aggregate");
  if (n_rows <= 0 || n_sum_rows < 0 || n_sum_rows > n_rows || width <= 0 || (width & 3) || !rel_part || !node_parts || !senders || !row_start || !agg ||
      !res_a || !res_b || !res_out) {
    gsr_set_error("gsr_gnn_aggregate_res: bad argument (width a multiple of 4)");
    return -2;
  }
  return gsr_launch_gnn_aggregate(n_rows, n_sum_rows, width, rel_part, node_parts, (const long long*)senders, (const long long*)row_start, agg, (hipStream_t)stream,
                                  res_a, res_b, res_out);
}
int gsr_gnn_rel_inputs(int32_t n_rel, int32_t attr_dim, int32_t group_dim, int32_t state_cols, const float* rel_nodes, const int64_t* receivers,
                       const int64_t* senders, float* out, void* stream) {
  GsrRange _range("gsr_gnn_rel_inputs");
  if (n_rel <= 0 || attr_dim < 0 || group_dim < 0 || state_cols < 0 || !rel_nodes || !receivers || !senders || !out) {
    gsr_set_error("gsr_gnn_rel_inputs: bad argument");
    return -2;
  }
  return gsr_launch_gnn_rel_inputs(n_rel, attr_dim, group_dim, state_cols, rel_nodes, (const long long*)receivers, (const long long*)senders, out,
                                   (hipStream_t)stream);
}

int gsr_fps(int32_t N, const float* pos, int32_t npoints, int32_t start_idx, float* scratch, int64_t* out_idx, void* stream) {
  GsrRange _range("gsr_fps");
  if (N < 0 || npoints < 0 || (N > 0 && npoints > 0 && (!pos || !scratch || !out_idx))) { gsr_set_error("gsr_fps: bad argument"); return -2; }
  if (npoints > N || (N > 0 && (start_idx < 0 || start_idx >= N))) { gsr_set_error("gsr_fps: npoints / start_idx out of range"); return -2; }
  return gsr_launch_fps(N, pos, npoints, start_idx, scratch, (long long*)out_idx, (hipStream_t)stream);
}

int gsr_lbs(int32_t P, int32_t n_bones, const float* bones, const float* rotations, const float* translations,
            const float* bone_quats, const float* xyz, const float* quat, float* out_xyz, float* out_quat, void* stream) {
  GsrRange _range("gsr_lbs");
  if (P < 0 || n_bones <= 0 || !bones || !rotations || !translations || !bone_quats || (P > 0 && (!xyz || !out_xyz))) {
    gsr_set_error("gsr_lbs: bad argument");
    return -2;
  }
  return gsr_launch_lbs(P, n_bones, bones, rotations, translations, bone_quats, xyz, quat, out_xyz, out_quat, (hipStream_t)stream);
}

int32_t gsr_image_loss_blocks(int32_t C, int32_t H, int32_t W) { return C * gsr_loss_blocks_per_channel(H, W); }

int gsr_image_loss_forward(const float* window11_host, int32_t C, int32_t H, int32_t W, const float* pred, const float* target,
                           float* fA, float* fC, float* fE, float* block_l1, float* block_ssim, void* stream) {
  GsrRange _range("gsr_image_loss_forward");
  if (!window11_host || !pred || !target || !fA || !fC || !fE || !block_l1 || !block_ssim || C <= 0 || H <= 0 || W <= 0) {
    gsr_set_error("gsr_image_loss_forward: bad argument");
    return -2;
  }
  return gsr_launch_image_loss_fwd(window11_host, C, H, W, pred, target, fA, fC, fE, block_l1, block_ssim, (hipStream_t)stream);
}

int gsr_image_loss_backward(const float* window11_host, int32_t C, int32_t H, int32_t W, const float* pred,
                            const float* target, const float* fA, const float* fC, const float* fE, const float* grad_loss,
                            int32_t channels_per_image, float w_l1, float w_ssim, float* d_pred, void* stream) {
  GsrRange _range("gsr_image_loss_backward");
  if (!window11_host || !pred || !target || !fA || !fC || !fE || !grad_loss || !d_pred || C <= 0 || H <= 0 || W <= 0) {
    gsr_set_error("gsr_image_loss_backward: bad argument");
    return -2;
  }
  if (channels_per_image <= 0 || C % channels_per_image != 0) { gsr_set_error("gsr_image_loss_backward: C must be a multiple of channels_per_image"); return -2; }
  return gsr_launch_image_loss_bwd(window11_host, C, H, W, pred, target, fA, fC, fE, grad_loss, channels_per_image, w_l1, w_ssim, d_pred,
                                   (hipStream_t)stream);
}

int32_t gsr_views_loss_blocks(int32_t n_images, int32_t channels, int32_t H, int32_t W) {
  return n_images * channels * gsr_loss_blocks_per_channel(H, W);
}

static int views_loss_check(const char* who, const gsr_loss_views* v, int32_t H, int32_t W, const float* cam_m, const float* cam_c) {
  if (!v || v->n_images <= 0 || v->n_images > GSR_LOSS_MAX_IMAGES || v->channels <= 0 || v->channels > 4 || H <= 0 || W <= 0) {
    gsr_set_error("%s: bad view table (1..%d images of 1..4 channels)", who, GSR_LOSS_MAX_IMAGES);
    return -2;
  }
  for (int i = 0; i < v->n_images; ++i) {
    if (!v->target[i]) { gsr_set_error("%s: target[%d] is NULL", who, i); return -2; }
    if (v->cam_row[i] >= 0 && (!cam_m || !cam_c)) { gsr_set_error("%s: image %d has a camera row but cam_m / cam_c is NULL", who, i); return -2; }
  }
  return 0;
}

int gsr_target_moments(const float* window11_host, int32_t channels, int32_t H, int32_t W, const float* target, float* moments,
                       void* stream) {
  GsrRange _range("gsr_target_moments");
  if (!window11_host || !target || !moments || channels <= 0 || channels > 4 || H <= 0 || W <= 0) { gsr_set_error("gsr_target_moments: bad argument"); return -2; }
  return gsr_launch_target_moments(window11_host, channels, H, W, target, moments, (hipStream_t)stream);
}

int gsr_views_loss_forward(const float* window11_host, const gsr_loss_views* views, int32_t H, int32_t W, const float* renders,
                           const float* cam_m, const float* cam_c, float w_l1, float w_ssim, float* fA, float* fC, float* fE,
                           float* partials, float* losses, void* stream) {
  GsrRange _range("gsr_views_loss_forward");
  if (int e = views_loss_check("gsr_views_loss_forward", views, H, W, cam_m, cam_c)) return e;
  if (!window11_host || !renders || !fA || !fC || !fE || !partials || !losses) { gsr_set_error("gsr_views_loss_forward: NULL argument"); return -2; }
  return gsr_launch_views_loss_fwd(window11_host, views, H, W, renders, cam_m, cam_c, w_l1, w_ssim, fA, fC, fE, partials, losses,
                                   (hipStream_t)stream);
}

int gsr_views_loss_backward(const float* window11_host, const gsr_loss_views* views, int32_t H, int32_t W, const float* renders,
                            const float* cam_m, const float* cam_c, int32_t n_cams, const float* fA, const float* fC,
                            const float* fE, const float* grad_total, float w_l1, float w_ssim, float* d_renders, float* partials,
                            float* d_cam_m, float* d_cam_c, void* stream) {
  GsrRange _range("gsr_views_loss_backward");
  if (int e = views_loss_check("gsr_views_loss_backward", views, H, W, cam_m, cam_c)) return e;
  if (!window11_host || !renders || !fA || !fC || !fE || !grad_total || !d_renders || !partials) { gsr_set_error("gsr_views_loss_backward: NULL argument"); return -2; }
  for (int i = 0; i < views->n_images; ++i)
    if (views->cam_row[i] >= n_cams && d_cam_m) { gsr_set_error("gsr_views_loss_backward: cam_row[%d] = %d >= n_cams = %d", i, views->cam_row[i], n_cams); return -2; }
  return gsr_launch_views_loss_bwd(window11_host, views, H, W, renders, cam_m, cam_c, n_cams, fA, fC, fE, grad_total, w_l1, w_ssim,
                                   d_renders, partials, d_cam_m, d_cam_c, (hipStream_t)stream);
}

int gsr_mark_visible(const float* viewmatrix, int32_t P, const float* means3D, uint8_t* present, void* stream) {
  GsrRange _range("gsr_mark_visible");
  if (P <= 0) return 0;
  if (!viewmatrix || !means3D || !present) { gsr_set_error("gsr_mark_visible: NULL argument"); return -2; }
  return gsr_launch_mark_visible(viewmatrix, P, means3D, present, (hipStream_t)stream);
}

int gsr_debug_get_views(int32_t P, uint32_t num_rendered, int32_t H, int32_t W, const void* geom_state,
                        const void* binning_state, const void* image_state, gsr_debug_views* out) {
  if (!out) return -2;
  memset(out, 0, sizeof *out);
  GeomState g; ImageState im; BinningState bs;
  if (geom_state) {
    gsr_carve_geom(const_cast<void*>(geom_state), P, &g);
    out->rec = (const float*)g.rec;
    out->rect = (const uint32_t*)g.rect; out->tiles_touched = g.tiles_touched; out->offsets = g.offsets;
  }
  if (binning_state) {
    gsr_carve_binning(const_cast<void*>(binning_state), num_rendered, &bs);
    out->point_list = bs.point_list;
  }
  if (image_state) {
    gsr_carve_image(const_cast<void*>(image_state), H, W, &im);
    out->ranges = (const uint32_t*)im.ranges; out->final_T = im.final_T; out->n_contrib = im.n_contrib;
  }
  return 0;
}

// Tests: mark the backward error word of a single-view image state as render_bwd_pc does when one of its waits times out (the per-Gaussian
// backward of that state then writes NaN for dL/dmeans3D).  Not part of include/gsr.h, like the other gsr_debug_pc_* exports.
extern "C" int gsr_debug_pc_mark_call_error(int32_t H, int32_t W, void* image_state, void* stream) {
  if (!image_state) return -2;
  ImageState im;
  gsr_carve_image(image_state, H, W, &im);
  const uint32_t one = 1u;
  GSR_HIP_CHECK(hipMemcpyAsync(im.queue + GSR_QUEUE_BWD_ERROR, &one, sizeof one, hipMemcpyHostToDevice, (hipStream_t)stream));
  GSR_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int gsr_selftest(void* stream) { return gsr_run_selftest((hipStream_t)stream); }

int64_t gsr_wait_counts(const volatile int32_t* counts, int32_t n, int64_t spin_us, int64_t timeout_us) {
  if (!counts || n <= 0) return 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t it = 0;; ++it) {
    int32_t lo = 0x7fffffff, hi = -1;
    for (int32_t i = 0; i < n; ++i) { const int32_t c = counts[i]; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
    if (lo >= 0) { std::atomic_thread_fence(std::memory_order_acquire); return (int64_t)hi; }
    if ((it & 63) == 63) {
      const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      if (us > timeout_us) return -1;
      if (us > spin_us) sched_yield();
    } else {
      __builtin_ia32_pause();
    }
  }
}

int64_t gsr_wait_block_counts(const volatile uint32_t* words, int32_t nblk, int64_t spin_us, int64_t timeout_us, int32_t* any_differs) {
  if (!words || nblk <= 0) return 0;
  const auto t0 = std::chrono::steady_clock::now();
  int32_t b = 0;
  uint64_t total = 0;
  uint32_t differs = 0;
  for (uint64_t it = 0;; ++it) {
    while (b < nblk) {                       // blocks finish roughly in order: resume where the last poll stopped
      const uint32_t c = words[2 * b + 1];
      if (c == 0xffffffffu) break;
      std::atomic_thread_fence(std::memory_order_acquire);
      differs |= words[2 * b];              // (both words come from one 8-byte store)
      total += c;
      ++b;
    }
    if (b == nblk) {
      if (any_differs) *any_differs = differs ? 1 : 0;
      return (int64_t)total;
    }
    if ((it & 63) == 63) {
      const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
      if (us > timeout_us) return -1;
      if (us > spin_us) sched_yield();
    } else {
      __builtin_ia32_pause();
    }
  }
}

int gsr_debug_phase_timing(uint64_t* out16) { return gsr_debug_fwd_timing((unsigned long long*)out16); }

int gsr_profile_begin(void) {
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on = true;
  return 0;
}

int gsr_profile_end(gsr_kernel_time* out, int32_t max_entries, int32_t* n_out) {
  g_prof_on = false;
  int n = 0;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      int k = 0;
      for (; k < n; ++k) if (strcmp(out[k].name, r.name) == 0) break;
      if (k == n) {
        if (n >= max_entries) continue;
        memset(&out[n], 0, sizeof out[n]);
        strncpy(out[n].name, r.name, sizeof(out[n].name) - 1);
        ++n;
      }
      out[k].total_ms += ms;
      out[k].launches += 1;
    }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  g_prof.clear();
  if (n_out) *n_out = n;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------ self-test
namespace {
__global__ void st_wave_sum_kernel(const float* in, float* out_dpp, float* out_ref) {
  float v = in[threadIdx.x];
  float a = gsr_wave_sum_to_lane63(v);
  float b = gsr_wave_sum_shfl(v);
  // the nine-at-once asm form must agree too: feed it v, 2v, ..., 9v
  float q0 = v, q1 = 2.f * v, q2 = 3.f * v, q3 = 4.f * v, q4 = 5.f * v, q5 = 6.f * v, q6 = 7.f * v, q7 = 8.f * v, q8 = 9.f * v;
  gsr_wave_sum9_to_lane63(q0, q1, q2, q3, q4, q5, q6, q7, q8);
  // packed form: lane with (lane & 15) = i < 8 holds the total of value i, lanes with bit 3 set the total of value 8
  const float z = gsr_wave_sum9_packed(v, 2.f * v, 3.f * v, 4.f * v, 5.f * v, 6.f * v, 7.f * v, 8.f * v, 9.f * v);
  const int l = threadIdx.x & 63;
  const float zexp = ((l & 8) == 0) ? (float)((l & 7) + 1) * b : 9.f * b;
  bool okz = __ballot(z != zexp) == 0ull;
  const float zp = gsr_wave_sum9_packed<true>(v, 2.f * v, 3.f * v, 4.f * v, 5.f * v, 6.f * v, 7.f * v, 8.f * v, 9.f * v);   // row levels by lane swaps
  okz = okz && __ballot(zp != zexp) == 0ull;
  // eight-value form of the fused pair backward: lane & 7 = i holds the total of value i
  const float z8 = gsr_wave_sum8_packed(v, 2.f * v, 3.f * v, 4.f * v, 5.f * v, 6.f * v, 7.f * v, 8.f * v);
  okz = okz && __ballot(z8 != (float)((l & 7) + 1) * b) == 0ull;
  // six-value form (no colour gradient): gsr_sum6_slot names the value a lane holds
  const float z6 = gsr_wave_sum6_packed(v, 2.f * v, 3.f * v, 4.f * v, 5.f * v, 6.f * v);
  const int s6 = gsr_sum6_slot(l);
  okz = okz && __ballot(s6 >= 0 && z6 != (float)(s6 + 1) * b) == 0ull;
  {  // every value 0..5 has a lane in the last row (the lanes that write the totals to LDS)
    uint32_t have = 0;
    for (int q = 48; q < 64; ++q) if (gsr_sum6_slot(q) >= 0) have |= 1u << gsr_sum6_slot(q);
    okz = okz && have == 0x3fu;
  }
  // the blend backward runs unused lanes with alpha = 0 and relies on 1 / (1 - 0) being exactly 1 (v_rcp_f32)
  const float alpha0 = fminf(0.99f, 0.7f * (v * 0.0f));
  okz = okz && __ballot(__builtin_amdgcn_rcpf(1.0f - alpha0) != 1.0f) == 0ull;
  if ((threadIdx.x & 63) == 63) {
    const bool ok9 = okz && q0 == b && q1 == 2.f * b && q2 == 3.f * b && q3 == 4.f * b && q4 == 5.f * b && q5 == 6.f * b &&
                     q6 == 7.f * b && q7 == 8.f * b && q8 == 9.f * b;
    out_dpp[threadIdx.x >> 6] = ok9 ? a : -1e30f;
    out_ref[threadIdx.x >> 6] = b;
  }
}
}  // namespace

int gsr_run_selftest(hipStream_t st) {
  int fail = 0;
  // 1. DPP wave reduction vs shuffle reduction vs host sum
  {
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)((i * 37) % 101) - 50.0f;  // small integers: sums are exact
    float *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    GSR_HIP_CHECK(hipMalloc(&d_in, n * 4)); GSR_HIP_CHECK(hipMalloc(&d_a, 16)); GSR_HIP_CHECK(hipMalloc(&d_b, 16));
    GSR_HIP_CHECK(hipMemcpyAsync(d_in, h.data(), n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(st_wave_sum_kernel, dim3(1), dim3(n), 0, st, d_in, d_a, d_b);
    float a[4], b[4];
    GSR_HIP_CHECK(hipMemcpyAsync(a, d_a, 16, hipMemcpyDeviceToHost, st));
    GSR_HIP_CHECK(hipMemcpyAsync(b, d_b, 16, hipMemcpyDeviceToHost, st));
    GSR_HIP_CHECK(hipStreamSynchronize(st));
    for (int w = 0; w < 4; ++w) {
      float ref = 0.f;
      for (int l = 0; l < 64; ++l) ref += h[w * 64 + l];
      if (a[w] != ref) fail |= 1;
      if (b[w] != ref) fail |= 2;
    }
    (void)hipFree(d_in); (void)hipFree(d_a); (void)hipFree(d_b);
  }
  // 2. exclusive scan
  {
    const uint32_t n = 20011;
    std::vector<uint32_t> h(n), ref(n + 1), got(n + 1);
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n; ++i) { h[i] = (i * 2654435761u) >> 28; ref[i] = acc; acc += h[i]; }
    ref[n] = acc;
    uint32_t *d_in = nullptr, *d_out = nullptr;
    GSR_HIP_CHECK(hipMalloc(&d_in, n * 4)); GSR_HIP_CHECK(hipMalloc(&d_out, (n + 1) * 4));
    GSR_HIP_CHECK(hipMemcpyAsync(d_in, h.data(), n * 4, hipMemcpyHostToDevice, st));
    if (int rc = gsr_launch_scan_exclusive(d_in, d_out, n, nullptr, st)) return rc;
    GSR_HIP_CHECK(hipMemcpyAsync(got.data(), d_out, (n + 1) * 4, hipMemcpyDeviceToHost, st));
    GSR_HIP_CHECK(hipStreamSynchronize(st));
    if (memcmp(ref.data(), got.data(), (n + 1) * 4) != 0) fail |= 4;
    (void)hipFree(d_in); (void)hipFree(d_out);
  }
  return fail;
}
