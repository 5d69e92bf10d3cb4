// gsr_torch.cpp -- the thin torch C++ layer over the C-ABI of libgsr_hip.so (include/gsr.h): the three entry points the upstream
// extension's Python wrapper binds -- rasterize_gaussians, rasterize_gaussians_backward, mark_visible (upstream is
// installed per /root/reference/README.md:28-32) -- with upstream's argument order and result tuples.  This layer owns ALLOCATION
// (torch's caching allocator) and the current-stream lookup; every byte of compute stays behind the C-ABI.  One native call per
// forward and one per backward: the unchanged-caller path (`GaussianRasterizer(raster_settings=cam)(**rendervar)` per render,
// /root/reference/src/tracking/train_utils.py:178,192) no longer pays ~20 ctypes calls and their Python glue.
//
// Built in-tree by __graft_entry__.build() (torch.utils.cpp_extension, host compiler only: there is no device code here) as
// gs-dynamics_amd/diff_gaussian_rasterization/_C.so, linked against ../csrc/libgsr_hip.so through an $ORIGIN rpath.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <tuple>

#include "../../include/gsr.h"

namespace {

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", gsr_last_error());
}

const float* fptr(const torch::Tensor& t) { return t.numel() ? t.data_ptr<float>() : nullptr; }

torch::Tensor f32c(const torch::Tensor& t, const c10::Device& dev) {   // contiguous fp32 on the render device (no copy when already so)
  if (t.numel() == 0) return t;
  if (t.device() == dev && t.scalar_type() == torch::kFloat32 && t.is_contiguous()) return t;
  return t.to(dev, torch::kFloat32).contiguous();
}

// element counts of the optional per-Gaussian inputs (an empty tensor = "not given"): a wrongly sized tensor would otherwise be read
// out of bounds on the device
void check_counts(const char* who, int64_t P, const torch::Tensor& colors, const torch::Tensor& opacity, const torch::Tensor& scales,
                  const torch::Tensor& rotations, const torch::Tensor& cov3D, const torch::Tensor& sh) {
  auto want = [&](const torch::Tensor& t, int64_t n, const char* name) {
    TORCH_CHECK(t.numel() == 0 || t.numel() == n, who, ": ", name, " must hold ", n, " floats for ", P, " Gaussians, got ", t.numel());
  };
  want(colors, 3 * P, "colors_precomp"); want(scales, 3 * P, "scales"); want(rotations, 4 * P, "rotations"); want(cov3D, 6 * P, "cov3D_precomp");
  TORCH_CHECK(opacity.numel() == P, who, ": opacities must hold ", P, " floats, got ", opacity.numel());
  TORCH_CHECK(sh.numel() == 0 || (sh.dim() == 3 && sh.size(0) == P && sh.size(2) == 3), who, ": shs must be [P, M, 3]");
}

struct Settings {
  gsr_settings s;
  torch::Tensor bg, view, proj, campos;   // keep-alives of the converted settings tensors
};

Settings make_settings(const torch::Tensor& bg, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const torch::Tensor& campos,
                       double tanfovx, double tanfovy, int64_t H, int64_t W, double scale_modifier, int64_t degree, int64_t M, bool prefiltered,
                       const c10::Device& dev) {
  Settings o;
  o.bg = f32c(bg, dev); o.view = f32c(viewmatrix, dev); o.proj = f32c(projmatrix, dev); o.campos = f32c(campos, dev);
  TORCH_CHECK(o.bg.numel() >= 3 && o.view.numel() >= 16 && o.proj.numel() >= 16 && o.campos.numel() >= 3,
              "raster settings: bg / campos must hold 3 floats, viewmatrix / projmatrix 16");
  o.s.image_height = (int32_t)H; o.s.image_width = (int32_t)W;
  o.s.tanfovx = (float)tanfovx; o.s.tanfovy = (float)tanfovy; o.s.scale_modifier = (float)scale_modifier;
  o.s.sh_degree = (int32_t)degree; o.s.sh_coeffs = (int32_t)M; o.s.prefiltered = prefiltered ? 1 : 0;
  o.s.bg = o.bg.data_ptr<float>(); o.s.viewmatrix = o.view.data_ptr<float>(); o.s.projmatrix = o.proj.data_ptr<float>();
  o.s.campos = o.campos.data_ptr<float>();
  return o;
}

// Tile-list reuse (include/gsr.h: gsr_forward_preprocess_same / gsr_forward_render_shared).  The reference renders every camera twice with
// the same geometry and other colours -- get_loss: colours then segmentation colours (/root/reference/src/tracking/train_utils.py:178,192),
// predict.py: colours then an all-ones mask (:115-123) -- through two separate GaussianRasterizer calls whose geometry tensors are fresh
// copies.  The layer remembers the LAST forward's states (geometry + binning + image tensors); the next forward of the same (P, H, W,
// stream) has its preprocess COMPARE itself with that geometry state on the device, bit for bit (entry counts, tile rects and masks, depth
// bits, 2D means, conics, opacities: everything the lists and the blend decisions depend on), and when nothing differs it blends from those
// lists instead of emitting and sorting its own: bit-identical outputs, ~55 us of GPU time less per second render at 100k / 800^2, and no
// "up to a hash coincidence" (rounds 3 - 4 compared a 64-bit fingerprint).  One entry per host thread; the tensors it holds (~30 MB at
// that size) are released when another geometry replaces them.  GSR_NO_LIST_REUSE=1 / set_list_reuse(false) switch it off.
struct ListCache {
  bool valid = false;
  int dev = -1;
  int64_t P = 0, H = 0, W = 0;
  uint32_t D = 0;           // entries of the cached forward
  uint32_t layout = 0;      // the num_rendered its binning state was laid out for (= D, or the capacity of a capacity-mode forward)
  void* stream = nullptr;   // the stream the lists were produced on: a forward on ANOTHER stream is not ordered behind their writes -- it bins its own
  torch::Tensor geom, binning, image;
};

// Capacity mode (include/gsr.h: gsr_forward_capacity; round 5).  Upstream's forward -- and this layer until then -- reads
// the entry count back between its two stages: the GPU idles while the host sizes the binning buffer and queues the remaining launches
// (~20 us of a 98 us forward at 50 k Gaussians / 800^2).  The node's forward instead sizes the buffer from the previous call of the same
// (P, H, W) plus 25 %, queues BOTH stages, and only then looks at the count (pinned words the preprocess blocks store: by then
// they have arrived -- the host does not wait and the GPU does not idle).  A count above the capacity -- the scene grew by more than a quarter between two calls -- repeats the forward
// through the exact path before anything is returned.  The comparison with the previous forward's geometry (tile-list reuse) rides along:
// its per-block verdict words go to pinned memory too, so the layer KNOWS after every call whether it was a twin of its predecessor, and
// sends a call through the comparing (synchronising, list-sharing) path when the call two before it was a twin -- the reference's
// colour / seg and colour / mask alternation, or a static scene; mispredictions cost time (one readback, or one redundant binning), never results.
//
// EVERYTHING the layer remembers between two calls lives in ONE LayerState object (round 6; SURVEY.md section 8b: "no global state").  The
// Python module owns the objects -- diff_gaussian_rasterization.layer_state(device): one per device, created on first use, inspectable
// (stats()), resettable (reset()) -- and hands the device's to every GaussianRasterizer call.  Upstream's three entry points take it as an
// optional last argument: WITHOUT one they are pure functions of their arguments, as upstream's are (exact path, nothing remembered).
constexpr int64_t kCompareMaxP = 2048 * 256;     // the library compares up to this many Gaussians (GSR_HOST_SCAN_MAX_BLOCKS blocks)
struct LayerState {
  ListCache lists;                                              // the last forward's states (tile-list reuse)
  std::map<std::tuple<int64_t, int64_t, int64_t>, uint32_t> cap;   // remembered entry capacity per (P, H, W)
  int32_t* pinned = nullptr;   // COHERENT pinned host words: [0] the tile-order kernel's count (P > 512 Ki), [2 ...] the preprocess blocks' {differs, entry count} pairs
  bool twin[2] = {false, false};   // was the last forward / the one before it a twin of its predecessor
  bool capacity = true;        // switch: capacity-mode forwards
  bool reuse = [] { const char* e = getenv("GSR_NO_LIST_REUSE"); return !(e && *e && atoi(e) != 0); }();   // switch: tile-list reuse
  int64_t reuse_hits = 0, capacity_calls = 0, capacity_overflows = 0, twins_seen_late = 0;
  LayerState() = default;
  LayerState(const LayerState&) = delete;
  LayerState& operator=(const LayerState&) = delete;
  ~LayerState() { if (pinned) (void)hipHostFree(pinned); }
  void note_twin(bool t) { twin[1] = twin[0]; twin[0] = t; }
  void reset() { lists = ListCache(); cap.clear(); twin[0] = twin[1] = false; }     // forget everything learned; switches and counters stay
};
thread_local LayerState* t_call_state = nullptr;   // the state of the GaussianRasterizer call in flight on this thread (set around RasterizeFn::apply)

// One forward (both stages) for the upstream-shaped binding and for the autograd node below.
struct Forward {
  int64_t D = 0;           // entries
  int64_t layout = 0;      // the num_rendered the binning state was laid out for: what the backward must be given
  torch::Tensor color, depth, radii, geom, binning, image;
};
Forward rasterize_forward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                          const torch::Tensor& scales, const torch::Tensor& rotations, double scale_modifier, const torch::Tensor& cov3D_precomp,
                          const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height,
                          int64_t image_width, const torch::Tensor& sh, int64_t degree, const torch::Tensor& campos, bool prefiltered,
                          bool will_backward,       // false = no input requires a gradient -- the blend records nothing for a backward
                          bool allow_capacity,      // false = the caller hands num_rendered on (upstream's tuple): layout must equal the count
                          LayerState* S) {          // what the layer may remember and reuse; nullptr = nothing (a pure function, as upstream's)
  TORCH_CHECK(means3D.dim() == 2 && means3D.size(1) == 3, "means3D must have dimensions (num_points, 3)");
  TORCH_CHECK(means3D.is_cuda(), "diff_gaussian_rasterization (MI355X build) runs on a HIP device only; there is no CPU fallback");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuard guard(dev.index());
  const int64_t P = means3D.size(0), H = image_height, W = image_width;
  const int64_t M = sh.numel() ? sh.size(1) : 0;
  auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  auto u8 = torch::TensorOptions().dtype(torch::kUInt8).device(dev);
  if (P == 0) {   // zero-filled outputs without launching anything
    Forward o;
    o.color = torch::zeros({3, H, W}, f32); o.depth = torch::zeros({1, H, W}, f32); o.radii = torch::zeros({0}, f32.dtype(torch::kInt32));
    o.geom = torch::empty({0}, u8); o.binning = torch::empty({0}, u8); o.image = torch::empty({0}, u8);
    return o;
  }
  check_counts("rasterize_gaussians", P, colors, opacity, scales, rotations, cov3D_precomp, sh);
  const torch::Tensor m3 = f32c(means3D, dev), col = f32c(colors, dev), op = f32c(opacity, dev), sc = f32c(scales, dev),
                      rot = f32c(rotations, dev), cov = f32c(cov3D_precomp, dev), shs = f32c(sh, dev);
  Settings st = make_settings(background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W, scale_modifier, degree, M, prefiltered, dev);
  torch::Tensor color = torch::empty({3, H, W}, f32), depth = torch::empty({1, H, W}, f32);
  torch::Tensor radii = torch::empty({P}, f32.dtype(torch::kInt32));
  torch::Tensor geom = torch::empty({(int64_t)gsr_geom_bytes((int32_t)P)}, u8);
  torch::Tensor image = torch::empty({(int64_t)gsr_image_bytes((int32_t)H, (int32_t)W)}, u8);
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  {   // either path needs the entry count on the host before it returns (the count-first path synchronises, the capacity path polls pinned
      // words): inside a stream capture neither ever arrives -- say so at once (gsdyn.step.GraphedRenderStep is the capturable step)
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &capturing) == hipSuccess)
      TORCH_CHECK(capturing == hipStreamCaptureStatusNone, "GaussianRasterizer cannot be captured into a graph: its forward needs the entry count "
                  "on the host (as upstream's does); use gsdyn.step.GraphedRenderStep / render_step_views for a capturable step");
  }
  uint32_t D = 0;
  int32_t same = 0;
  ListCache none;
  ListCache& lc = S ? S->lists : none;
  const bool reuse = S && S->reuse;
  const bool candidate = reuse && lc.valid && lc.dev == dev.index() && lc.P == P && lc.H == H && lc.W == W && lc.stream == stream &&
                         (allow_capacity || lc.layout == lc.D);
  const uint32_t fwd_flags = will_backward ? 0u : (uint32_t)GSR_FORWARD_ONLY;
  auto finish = [&](uint32_t count, uint32_t layout, const torch::Tensor& binning) {
    Forward o;
    o.D = count; o.layout = layout; o.color = color; o.depth = depth; o.radii = radii; o.geom = geom; o.binning = binning; o.image = image;
    return o;
  };
  auto remember = [&](uint32_t count, uint32_t layout, const torch::Tensor& binning) {
    if (!S) return;
    if (reuse && count > 0) {
      lc.valid = true; lc.dev = dev.index(); lc.P = P; lc.H = H; lc.W = W; lc.D = count; lc.layout = layout; lc.stream = stream;
      lc.geom = geom; lc.binning = binning; lc.image = image;
    } else {
      lc = ListCache();
    }
  };
  const auto key = std::make_tuple(P, H, W);
  if (S && allow_capacity && S->capacity && S->cap.count(key) && !(candidate && S->twin[1])) {
    LayerState& cs = *S;
    const auto known = cs.cap.find(key);
    // ---- both stages queued, then the count (see CapacityState)
    const uint32_t cap = known->second;
    const int64_t nblk = (P + 255) / 256;
    const bool compare = candidate && P <= kCompareMaxP;
    if (!cs.pinned)      // (explicitly coherent: the stores must be visible to the polling host while kernels of the stream are still running)
      C10_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&cs.pinned), sizeof(int32_t) * (2 + 2 * (kCompareMaxP / 256)), hipHostMallocMapped | hipHostMallocCoherent));
    int32_t* pin = cs.pinned;
    // P <= 512 Ki: every preprocess block stores {differs, its entry count} to pinned words -- the count is known when that kernel is through;
    // above: the tile-order kernel's count word
    const bool words = P <= kCompareMaxP;
    // layout [count, pad, {differs, count} x nblk]: the pairs are 8-byte aligned (one 8-byte store each); BOTH words of a pair are reset --
    // the count to the sentinel, the verdict to "differs" (the conservative value should a pair ever be seen half written)
    if (words) for (int64_t b = 0; b < nblk; ++b) { pin[2 + 2 * b] = 1; pin[3 + 2 * b] = -1; }
    else pin[0] = -1;
    torch::Tensor binning = torch::empty({(int64_t)gsr_binning_bytes(cap, (int32_t)H, (int32_t)W)}, u8);
    check(gsr_forward_capacity(&st.s, (int32_t)P, fptr(m3), fptr(sc), fptr(rot), fptr(op), fptr(col), fptr(shs), fptr(cov), geom.data_ptr(),
                               radii.data_ptr<int32_t>(), binning.data_ptr(), cap, image.data_ptr(), color.data_ptr<float>(),
                               depth.data_ptr<float>(), compare ? lc.geom.data_ptr() : nullptr,
                               words ? reinterpret_cast<uint32_t*>(pin + 2) : nullptr, words ? nullptr : pin, fwd_flags, stream), "gsr_forward_capacity");
    ++cs.capacity_calls;
    int32_t any = 1;
    int64_t count = words ? gsr_wait_block_counts(reinterpret_cast<const volatile uint32_t*>(pin + 2), (int32_t)nblk, 500, 20 * 1000 * 1000, &any)
                          : gsr_wait_counts(pin, 1, 500, 20 * 1000 * 1000);
    if (count < 0) {      // twenty seconds without the stores: let the runtime tell what happened to the stream
      C10_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
      count = words ? gsr_wait_block_counts(reinterpret_cast<const volatile uint32_t*>(pin + 2), (int32_t)nblk, 0, 1000, &any) : (int64_t)pin[0];
      TORCH_CHECK(count >= 0, "gsr_forward_capacity: the entry count never arrived");
    }
    if ((uint64_t)count <= cap) {
      const bool twin = compare && !any && lc.D == (uint32_t)count;
      if (twin) ++cs.twins_seen_late;       // its lists were built a second time: the predictor sends the next one of the pattern the sharing way
      cs.note_twin(twin);
      cs.cap[key] = (uint32_t)std::min<uint64_t>(0xffffffffull, (uint64_t)count + (uint64_t)count / 4 + 1024);
      remember((uint32_t)count, cap, binning);
      return finish((uint32_t)count, cap, binning);
    }
    ++cs.capacity_overflows;      // the scene outgrew the estimate: everything again, with the count known (the outputs above are overwritten)
  }
  check(gsr_forward_preprocess_same(&st.s, (int32_t)P, fptr(m3), fptr(sc), fptr(rot), fptr(op), fptr(col), fptr(shs), fptr(cov), geom.data_ptr(),
                                    radii.data_ptr<int32_t>(), &D, candidate ? lc.geom.data_ptr() : nullptr, candidate ? &same : nullptr, stream),
        "gsr_forward_preprocess");
  if (S) S->cap[key] = (uint32_t)std::min<uint64_t>(0xffffffffull, (uint64_t)D + (uint64_t)D / 4 + 1024);
  if (candidate && same && D > 0 && lc.D == D) {
    // same geometry, same camera as the previous forward (compared on the device, bit for bit): its lists are this render's lists
    check(gsr_forward_render_shared_ex(&st.s, (int32_t)P, lc.layout, geom.data_ptr(), lc.binning.data_ptr(), lc.image.data_ptr(), image.data_ptr(),
                                       color.data_ptr<float>(), depth.data_ptr<float>(), fwd_flags, stream), "gsr_forward_render_shared");
    ++S->reuse_hits;
    S->note_twin(true);
    return finish(D, lc.layout, lc.binning);
  }
  if (S) S->note_twin(false);
  torch::Tensor binning = torch::empty({(int64_t)gsr_binning_bytes(D, (int32_t)H, (int32_t)W)}, u8);
  check(gsr_forward_render_ex(&st.s, (int32_t)P, D, geom.data_ptr(), binning.data_ptr(), image.data_ptr(), color.data_ptr<float>(),
                              depth.data_ptr<float>(), fwd_flags, stream), "gsr_forward_render");
  remember(D, D, binning);
  return finish(D, D, binning);
}

// upstream: RasterizeGaussiansCUDA(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
//           projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered)
//           -> (num_rendered, out_color, out_depth, radii, geomBuffer, binningBuffer, imgBuffer)        [the w-depth fork's tuple]
// (exact mode: the num_rendered it returns is what rasterize_gaussians_backward is handed as R)
std::tuple<int64_t, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& colors, const torch::Tensor& opacity,
                    const torch::Tensor& scales, const torch::Tensor& rotations, double scale_modifier, const torch::Tensor& cov3D_precomp,
                    const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, double tan_fovx, double tan_fovy, int64_t image_height,
                    int64_t image_width, const torch::Tensor& sh, int64_t degree, const torch::Tensor& campos, bool prefiltered,
                    bool will_backward,       // extension over upstream (default true)
                    const std::shared_ptr<LayerState>& state) {   // extension (default none): tile-list reuse across calls (exact mode either way)
  Forward o = rasterize_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
                                tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, will_backward, false, state.get());
  return std::make_tuple(o.D, o.color, o.depth, o.radii, o.geom, o.binning, o.image);
}

// upstream: RasterizeGaussiansBackwardCUDA(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
//           projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer)
//           -> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)
// (image height / width travel in upstream's dL_dout_color shape: here too)
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
rasterize_gaussians_backward(const torch::Tensor& background, const torch::Tensor& means3D, const torch::Tensor& radii, const torch::Tensor& colors,
                             const torch::Tensor& scales, const torch::Tensor& rotations, double scale_modifier, const torch::Tensor& cov3D_precomp,
                             const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, double tan_fovx, double tan_fovy,
                             const torch::Tensor& dL_dout_color, const torch::Tensor& sh, int64_t degree, const torch::Tensor& campos,
                             const torch::Tensor& geomBuffer, int64_t R, const torch::Tensor& binningBuffer, const torch::Tensor& imageBuffer,
                             bool want_color_grad) {   // extension over upstream (default true): false = colours_precomp needs no gradient
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuard guard(dev.index());
  const int64_t P = means3D.size(0);
  const int64_t M = sh.numel() ? sh.size(1) : 0;
  TORCH_CHECK(dL_dout_color.dim() == 3 && dL_dout_color.size(0) == 3, "dL_dout_color must be [3, H, W]");
  const int64_t H = dL_dout_color.size(1), W = dL_dout_color.size(2);
  auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(dev);
  torch::Tensor d_means3D = torch::empty({P, 3}, f32), d_means2D = torch::empty({P, 3}, f32), d_opacity = torch::empty({P, 1}, f32);
  torch::Tensor d_colors = (M || !want_color_grad) ? torch::empty({0}, f32) : torch::empty({P, 3}, f32);
  torch::Tensor d_cov = torch::empty({P, 6}, f32);
  torch::Tensor d_sh = M ? torch::empty({P, M, 3}, f32) : torch::empty({0}, f32);
  const bool has_sr = scales.numel() > 0;
  torch::Tensor d_scales = has_sr ? torch::empty({P, 3}, f32) : torch::empty({0}, f32);
  torch::Tensor d_rot = has_sr ? torch::empty({P, 4}, f32) : torch::empty({0}, f32);
  if (P == 0) return std::make_tuple(d_means2D, d_colors, d_opacity, d_means3D, d_cov, d_sh, d_scales, d_rot);
  {
    auto want = [&](const torch::Tensor& t, int64_t n, const char* name) {
      TORCH_CHECK(t.numel() == 0 || t.numel() == n, "rasterize_gaussians_backward: ", name, " must hold ", n, " elements, got ", t.numel());
    };
    want(colors, 3 * P, "colors_precomp"); want(scales, 3 * P, "scales"); want(rotations, 4 * P, "rotations"); want(cov3D_precomp, 6 * P, "cov3D_precomp");
    TORCH_CHECK(radii.numel() == P && radii.scalar_type() == torch::kInt32, "rasterize_gaussians_backward: radii must be int32 [P]");
    TORCH_CHECK(sh.numel() == 0 || (sh.dim() == 3 && sh.size(0) == P && sh.size(2) == 3), "rasterize_gaussians_backward: shs must be [P, M, 3]");
    TORCH_CHECK((size_t)geomBuffer.numel() >= gsr_geom_bytes((int32_t)P) && (size_t)imageBuffer.numel() >= gsr_image_bytes((int32_t)H, (int32_t)W) &&
                (R == 0 || (size_t)binningBuffer.numel() >= gsr_binning_bytes((uint32_t)R, (int32_t)H, (int32_t)W)),
                "rasterize_gaussians_backward: a state buffer is smaller than this (P, H, W, num_rendered) needs");
  }
  const torch::Tensor m3 = f32c(means3D, dev), col = f32c(colors, dev), sc = f32c(scales, dev), rot = f32c(rotations, dev),
                      cov = f32c(cov3D_precomp, dev), shs = f32c(sh, dev), g = f32c(dL_dout_color, dev);
  Settings st = make_settings(background, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, H, W, scale_modifier, degree, M, false, dev);
  torch::Tensor scratch = torch::empty({(int64_t)gsr_backward_scratch_bytes((int32_t)P, (uint32_t)R)},
                                       torch::TensorOptions().dtype(torch::kUInt8).device(dev));
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  auto optr = [](torch::Tensor& t) -> float* { return t.numel() ? t.data_ptr<float>() : nullptr; };
  check(gsr_backward(&st.s, (int32_t)P, (uint32_t)R, fptr(m3), fptr(sc), fptr(rot), fptr(col), fptr(shs), fptr(cov), radii.data_ptr<int32_t>(),
                     geomBuffer.data_ptr(), R ? binningBuffer.data_ptr() : nullptr, imageBuffer.data_ptr(), g.data_ptr<float>(),
                     R ? scratch.data_ptr() : nullptr, d_means3D.data_ptr<float>(), d_means2D.data_ptr<float>(), optr(d_colors),
                     d_opacity.data_ptr<float>(), optr(d_scales), optr(d_rot), d_cov.data_ptr<float>(), optr(d_sh), stream),
        "gsr_backward");
  return std::make_tuple(d_means2D, d_colors, d_opacity, d_means3D, d_cov, d_sh, d_scales, d_rot);
}

// ---- the autograd node of one GaussianRasterizer call, in C++ (round 5).  The Python wrapper used to be a
// torch.autograd.Function whose forward / backward each crossed into this layer once and spent ~0.12 ms per camera in the interpreter
// (profiles/r04_dropin_host_profile.txt); here the whole call is ONE crossing: GaussianRasterizer.forward -> _C.rasterize -> this node.
// Inputs in the order of upstream's Python _RasterizeGaussians.apply (means3D, means2D, sh, colors_precomp, opacities, scales,
// rotations, cov3Ds_precomp) followed by the settings record's fields; outputs (color, radii, depth).  means2D is the gradient holder
// the reference reads back (`rendervar['means2D'].grad`, /root/reference/src/tracking/external.py:139-140): its values are ignored.
struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
  static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor means3D, torch::Tensor means2D,
                                                torch::Tensor sh, torch::Tensor colors, torch::Tensor opacities, torch::Tensor scales,
                                                torch::Tensor rotations, torch::Tensor cov3D, torch::Tensor bg, torch::Tensor viewmatrix,
                                                torch::Tensor projmatrix, torch::Tensor campos, double tanfovx, double tanfovy, int64_t H,
                                                int64_t W, double scale_modifier, int64_t degree, bool prefiltered, bool will_backward) {
    (void)means2D;
    ctx->set_materialize_grads(false);        // grad_depth is ignored: do not let autograd fill a zero image for it
    Forward r = rasterize_forward(bg, means3D, colors, opacities, scales, rotations, scale_modifier, cov3D, viewmatrix, projmatrix, tanfovx,
                                  tanfovy, H, W, sh, degree, campos, prefiltered, will_backward, true, t_call_state);
    torch::Tensor color = r.color, depth = r.depth, radii = r.radii;
    ctx->saved_data["R"] = r.layout;      // what the states were laid out for (the count, or a capacity-mode forward's capacity)
    ctx->saved_data["tanfovx"] = tanfovx; ctx->saved_data["tanfovy"] = tanfovy; ctx->saved_data["scale_modifier"] = scale_modifier;
    ctx->saved_data["degree"] = degree; ctx->saved_data["H"] = H; ctx->saved_data["W"] = W;
    ctx->saved_data["has_sh"] = sh.numel() > 0; ctx->saved_data["has_col"] = colors.numel() > 0;
    ctx->saved_data["has_sc"] = scales.numel() > 0; ctx->saved_data["has_cov"] = cov3D.numel() > 0;
    ctx->saved_data["empty"] = means3D.size(0) == 0;
    ctx->save_for_backward({means3D, radii, colors, sh, scales, rotations, cov3D, r.geom, r.binning, r.image, bg, viewmatrix, projmatrix, campos});
    ctx->mark_non_differentiable({radii});
    return {color, radii, depth};
  }
  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    torch::autograd::variable_list out(20);       // one slot per forward argument; undefined = no gradient
    if (ctx->saved_data["empty"].toBool()) return out;
    const auto sv = ctx->get_saved_variables();
    const torch::Tensor &m3 = sv[0], &radii = sv[1], &col = sv[2], &sh = sv[3], &sc = sv[4], &rot = sv[5], &cov = sv[6], &geom = sv[7],
                        &binning = sv[8], &image = sv[9], &bg = sv[10], &view = sv[11], &proj = sv[12], &campos = sv[13];
    const int64_t H = ctx->saved_data["H"].toInt(), W = ctx->saved_data["W"].toInt();
    torch::Tensor g = grads[0];                   // grads[1] (radii), grads[2] (depth): accepted, ignored -- no reference call site differentiates them
    if (!g.defined()) g = torch::zeros({3, H, W}, m3.options().dtype(torch::kFloat32));
    const bool has_sh = ctx->saved_data["has_sh"].toBool(), has_col = ctx->saved_data["has_col"].toBool(),
               has_sc = ctx->saved_data["has_sc"].toBool(), has_cov = ctx->saved_data["has_cov"].toBool();
    const bool want_col = has_col && ctx->needs_input_grad(3);      // frozen colours (the reference's training): the six-sum backward
    auto r = rasterize_gaussians_backward(bg, m3, radii, col, sc, rot, ctx->saved_data["scale_modifier"].toDouble(), cov, view, proj,
                                          ctx->saved_data["tanfovx"].toDouble(), ctx->saved_data["tanfovy"].toDouble(), g, sh,
                                          ctx->saved_data["degree"].toInt(), campos, geom, ctx->saved_data["R"].toInt(), binning, image, want_col);
    out[0] = std::get<3>(r);                      // means3D
    out[1] = std::get<0>(r);                      // means2D (x, y in NDC units, z = 0)
    if (has_sh) out[2] = std::get<5>(r);
    if (want_col) out[3] = std::get<1>(r);
    out[4] = std::get<2>(r);                      // opacities
    if (has_sc) { out[5] = std::get<6>(r); out[6] = std::get<7>(r); }
    if (has_cov) out[7] = std::get<4>(r);
    return out;
  }
};

// GaussianRasterizer.forward in one call.  Whether a backward can follow is decided HERE, before the node is built: inside a node's
// forward the grad mode is always off and needs_input_grad reports requires_grad flags even under torch.no_grad() --
// evaluation renders of trainable parameters under no_grad take the untracked forward (GSR_FORWARD_ONLY).
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor>
rasterize(const std::shared_ptr<LayerState>& state,      // the calling module's state for the device (may be none: nothing remembered)
          const torch::Tensor& means3D, const torch::Tensor& means2D, const torch::Tensor& sh, const torch::Tensor& colors,
          const torch::Tensor& opacities, const torch::Tensor& scales, const torch::Tensor& rotations, const torch::Tensor& cov3D,
          const torch::Tensor& bg, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix, const torch::Tensor& campos,
          double tanfovx, double tanfovy, int64_t H, int64_t W, double scale_modifier, int64_t degree, bool prefiltered) {
  bool will_backward = false;
  if (at::GradMode::is_enabled())
    for (const torch::Tensor* t : {&means3D, &means2D, &sh, &colors, &opacities, &scales, &rotations, &cov3D})
      will_backward = will_backward || (t->defined() && t->requires_grad());
  struct Scope {      // the node's forward runs inside apply(), on this thread: it finds the state here
    LayerState* prev;
    explicit Scope(LayerState* s) : prev(t_call_state) { t_call_state = s; }
    ~Scope() { t_call_state = prev; }
  } scope(state.get());
  auto o = RasterizeFn::apply(means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, bg, viewmatrix, projmatrix, campos, tanfovx,
                              tanfovy, H, W, scale_modifier, degree, prefiltered, will_backward);
  return std::make_tuple(o[0], o[1], o[2]);
}

// upstream: markVisible(means3D, viewmatrix, projmatrix) -> bool[P]
torch::Tensor mark_visible(const torch::Tensor& means3D, const torch::Tensor& viewmatrix, const torch::Tensor& projmatrix) {
  (void)projmatrix;   // the test is the near-plane cull in view space, as upstream's
  TORCH_CHECK(means3D.is_cuda(), "mark_visible: HIP device tensors only");
  const c10::Device dev = means3D.device();
  c10::hip::HIPGuard guard(dev.index());
  const int64_t P = means3D.size(0);
  torch::Tensor present = torch::zeros({P}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
  if (P == 0) return present.to(torch::kBool);
  const torch::Tensor m3 = f32c(means3D, dev), vm = f32c(viewmatrix, dev);
  void* stream = (void*)c10::hip::getCurrentHIPStream(dev.index()).stream();
  check(gsr_mark_visible(vm.data_ptr<float>(), (int32_t)P, m3.data_ptr<float>(), present.data_ptr<uint8_t>(), stream), "gsr_mark_visible");
  return present.to(torch::kBool);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  namespace py = pybind11;
  m.doc() = "MI355X rasterizer: torch layer over libgsr_hip.so (rasterize_gaussians / rasterize_gaussians_backward / mark_visible)";
  py::class_<LayerState, std::shared_ptr<LayerState>>(m, "LayerState",
      "What the layer remembers between calls on one device: the last forward's tile lists, entry capacities per (P, H, W), the twin predictor")
      .def(py::init<>())
      .def("reset", &LayerState::reset, "forget the cached lists, the capacities and the predictor (switches and counters stay)")
      .def("drop_list_cache", [](LayerState& s) { s.lists = ListCache(); })
      .def_property("list_reuse", [](const LayerState& s) { return s.reuse; }, [](LayerState& s, bool on) { s.reuse = on; s.lists = ListCache(); })
      .def_property("capacity_mode", [](const LayerState& s) { return s.capacity; },
                    [](LayerState& s, bool on) { s.capacity = on; s.twin[0] = s.twin[1] = false; })
      .def("stats", [](const LayerState& s) {
        py::dict d;
        d["list_reuse_hits"] = s.reuse_hits; d["capacity_calls"] = s.capacity_calls; d["capacity_overflows"] = s.capacity_overflows;
        d["twins_seen_late"] = s.twins_seen_late; d["cached_entries"] = s.lists.valid ? (int64_t)s.lists.D : (int64_t)0;
        d["capacities"] = (int64_t)s.cap.size(); d["twin_history"] = py::make_tuple(s.twin[0], s.twin[1]);
        return d;
      });
  // upstream's 18 arguments, then the two extensions (will_backward, state)
  m.def("rasterize_gaussians", &rasterize_gaussians, py::arg("background"), py::arg("means3D"), py::arg("colors"), py::arg("opacity"),
        py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"),
        py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("image_height"), py::arg("image_width"), py::arg("sh"), py::arg("degree"),
        py::arg("campos"), py::arg("prefiltered"), py::arg("will_backward") = true, py::arg("state") = std::shared_ptr<LayerState>());
  m.def("rasterize_gaussians_backward", &rasterize_gaussians_backward, py::arg("background"), py::arg("means3D"), py::arg("radii"),
        py::arg("colors"), py::arg("scales"), py::arg("rotations"), py::arg("scale_modifier"), py::arg("cov3D_precomp"),
        py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("tan_fovx"), py::arg("tan_fovy"), py::arg("dL_dout_color"), py::arg("sh"),
        py::arg("degree"), py::arg("campos"), py::arg("geomBuffer"), py::arg("R"), py::arg("binningBuffer"), py::arg("imageBuffer"),
        py::arg("want_color_grad") = true);
  m.def("rasterize", &rasterize);      // one GaussianRasterizer call: forward + the autograd node (C++); first argument: the LayerState or None
  m.def("mark_visible", &mark_visible);
  m.def("abi_version", []() { return (int)GSR_VERSION; });   // the header this layer was COMPILED against (compare with the library's gsr_version())
}
