"""ctypes binding of ``libgsr_hip.so`` (C-ABI in ``include/gsr.h``) -- the ONLY compute backend.

There is no CPU path and no fallback: if the shared library is missing, or the tensors are not on
a HIP device, the calls below raise.  PyTorch is used for device memory (caching allocator), the
current stream handle, and nothing else.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("GSR_HIP_LIB", os.path.join(_CSRC, "libgsr_hip.so"))

_lib = None


class GsrSettings(C.Structure):
    """struct gsr_settings (include/gsr.h)."""
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32), ("prefiltered", C.c_int32),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
    ]


class GsrDebugViews(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rec", "rect", "tiles_touched", "offsets",
                                          "point_list", "ranges", "final_T", "n_contrib")]


LOSS_MAX_IMAGES = 32


class GsrLossViews(C.Structure):
    _fields_ = [("n_images", C.c_int32), ("channels", C.c_int32), ("cam_row", C.c_int32 * LOSS_MAX_IMAGES),
                ("weight", C.c_float * LOSS_MAX_IMAGES), ("target", C.c_void_p * LOSS_MAX_IMAGES),
                ("target_moments", C.c_void_p * LOSS_MAX_IMAGES)]


ADAM_MAX_TENSORS = 16


class GsrAdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias_correction1", C.c_float), ("bias_correction2_sqrt", C.c_float),
                ("one_minus_beta1", C.c_float), ("one_minus_beta2", C.c_float)]


class GsrKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("total_ms", C.c_double), ("launches", C.c_int64)]


class GsrRawParams(C.Structure):
    """gsr_raw_params of include/gsr.h: the tracking step's parameters before their activations (fused into the per-Gaussian kernels)."""
    _fields_ = [(n, C.c_void_p) for n in ("unnorm_rotations", "logit_opacities", "log_scales", "rotations_out", "opacities_out", "scales_out",
                                          "d_unnorm_rotations", "d_logit_opacities", "d_log_scales")]


EXPORTS = ("gsr_version", "gsr_last_error", "gsr_geom_bytes", "gsr_image_bytes", "gsr_binning_bytes",
           "gsr_backward_scratch_bytes", "gsr_forward_preprocess", "gsr_forward_render", "gsr_backward",
           "gsr_forward_preprocess_same", "gsr_forward_render_shared", "gsr_forward_render_ex", "gsr_forward_render_shared_ex", "gsr_forward_capacity", "gsr_wait_block_counts",
           "gsr_mark_visible", "gsr_debug_get_views", "gsr_selftest", "gsr_profile_begin", "gsr_profile_end",
           "gsr_batch_state_bytes", "gsr_forward_preprocess_batch", "gsr_forward_render_batch", "gsr_forward_batch",
           "gsr_forward_batch_capacity", "gsr_forward_batch_capacity_raw",
           "gsr_backward_batch", "gsr_backward_batch_raw", "gsr_debug_phase_timing",
           "gsr_image_loss_blocks", "gsr_image_loss_forward", "gsr_image_loss_backward", "gsr_fps", "gsr_fps_scratch_bytes", "gsr_fit_rotations", "gsr_fit_bones", "gsr_fps_thin", "gsr_construct_edges", "gsr_lbs_valid", "gsr_lbs",
           "gsr_rigidity_blocks", "gsr_rigidity_forward", "gsr_rigidity_backward",
           "gsr_views_loss_blocks", "gsr_views_loss_forward", "gsr_views_loss_backward", "gsr_target_moments",
           "gsr_shared_terms_partials", "gsr_shared_terms_scratch", "gsr_shared_terms_forward", "gsr_shared_terms_backward",
           "gsr_activate_forward", "gsr_activate_backward", "gsr_adam_step", "gsr_radius_bookkeeping", "gsr_wait_counts",
           "gsr_gnn_aggregate", "gsr_gnn_rel_inputs", "gsr_construct_edges_dense", "gsr_rollout_step_tail",
           "gsr_construct_edges_rows", "gsr_rollout_step_head", "gsr_rollout_step_motion", "gsr_gnn_aggregate_res", "gsr_arm_depth_cuts")


def load_library():
    """dlopen libgsr_hip.so and declare the prototypes.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"diff_gaussian_rasterization: HIP extension not found at {LIB_PATH}. "
            "Build it with `make -C gs-dynamics_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    lib.gsr_version.restype = C.c_int
    lib.gsr_last_error.restype = C.c_char_p
    lib.gsr_geom_bytes.restype = sz; lib.gsr_geom_bytes.argtypes = [i32]
    lib.gsr_image_bytes.restype = sz; lib.gsr_image_bytes.argtypes = [i32, i32]
    lib.gsr_binning_bytes.restype = sz; lib.gsr_binning_bytes.argtypes = [u32, i32, i32]
    lib.gsr_backward_scratch_bytes.restype = sz; lib.gsr_backward_scratch_bytes.argtypes = [i32, u32]
    lib.gsr_forward_preprocess.restype = C.c_int
    lib.gsr_forward_preprocess.argtypes = [C.POINTER(GsrSettings), i32] + [vp] * 7 + [vp, vp, C.POINTER(u32), vp]
    lib.gsr_forward_render.restype = C.c_int
    lib.gsr_forward_render.argtypes = [C.POINTER(GsrSettings), i32, u32, vp, vp, vp, vp, vp, vp]
    lib.gsr_backward.restype = C.c_int
    lib.gsr_backward.argtypes = [C.POINTER(GsrSettings), i32, u32] + [vp] * 20 + [vp]
    PS = C.POINTER(GsrSettings)
    PV = C.POINTER(C.c_void_p)
    lib.gsr_batch_state_bytes.restype = sz; lib.gsr_batch_state_bytes.argtypes = [i32, i32, i32, i32]
    lib.gsr_forward_preprocess_batch.restype = C.c_int
    lib.gsr_forward_preprocess_batch.argtypes = [i32, PS, i32] + [vp] * 5 + [PV, vp, vp] + [PV, PV, vp, C.POINTER(u32), vp]
    lib.gsr_forward_render_batch.restype = C.c_int
    lib.gsr_forward_render_batch.argtypes = [i32, PS, i32, C.POINTER(u32), PV, PV, PV, vp, C.POINTER(i32), PV, PV, PV, i32, vp]
    lib.gsr_forward_batch.restype = C.c_int
    lib.gsr_forward_batch.argtypes = ([i32, PS, i32] + [vp] * 5 + [PV, vp, vp] + [PV, PV, PV, C.POINTER(sz), PV, vp,
                                      C.POINTER(i32), PV, PV, C.POINTER(u32), i32, vp])
    lib.gsr_backward_batch.restype = C.c_int
    lib.gsr_backward_batch.argtypes = ([i32, PS, i32, C.POINTER(u32)] + [vp] * 5 + [PV] * 4 + [vp, C.POINTER(i32), PV, PV]
                                       + [vp, PV, vp, PV, vp, vp, vp, vp, vp])
    lib.gsr_forward_batch_capacity.restype = C.c_int
    lib.gsr_forward_batch_capacity.argtypes = ([i32, PS, i32] + [vp] * 5 + [PV, vp, vp] + [PV, PV, PV, C.POINTER(u32), PV, vp,
                                               C.POINTER(i32), PV, PV, vp, vp])
    lib.gsr_forward_batch_capacity_raw.restype = C.c_int
    lib.gsr_forward_batch_capacity_raw.argtypes = lib.gsr_forward_batch_capacity.argtypes[:-1] + [C.POINTER(GsrRawParams), vp]
    lib.gsr_backward_batch_raw.restype = C.c_int
    lib.gsr_backward_batch_raw.argtypes = lib.gsr_backward_batch.argtypes[:-1] + [C.POINTER(GsrRawParams), vp]
    lib.gsr_image_loss_blocks.restype = i32
    lib.gsr_image_loss_blocks.argtypes = [i32, i32, i32]
    lib.gsr_image_loss_forward.restype = C.c_int
    lib.gsr_image_loss_forward.argtypes = [C.POINTER(C.c_float), i32, i32, i32] + [vp] * 7 + [vp]
    lib.gsr_image_loss_backward.restype = C.c_int
    lib.gsr_image_loss_backward.argtypes = [C.POINTER(C.c_float), i32, i32, i32] + [vp] * 6 + [i32, C.c_float, C.c_float, vp, vp]
    lib.gsr_target_moments.restype = C.c_int
    lib.gsr_target_moments.argtypes = [C.POINTER(C.c_float), i32, i32, i32, vp, vp, vp]
    lib.gsr_views_loss_blocks.restype = i32
    lib.gsr_views_loss_blocks.argtypes = [i32] * 4
    lib.gsr_views_loss_forward.restype = C.c_int
    lib.gsr_views_loss_forward.argtypes = ([C.POINTER(C.c_float), C.POINTER(GsrLossViews), i32, i32, vp, vp, vp, C.c_float, C.c_float]
                                           + [vp] * 6)
    lib.gsr_views_loss_backward.restype = C.c_int
    lib.gsr_views_loss_backward.argtypes = ([C.POINTER(C.c_float), C.POINTER(GsrLossViews), i32, i32, vp, vp, vp, i32, vp, vp, vp, vp,
                                             C.c_float, C.c_float] + [vp] * 5)
    lib.gsr_shared_terms_partials.restype = i32
    lib.gsr_shared_terms_partials.argtypes = [i32, i32]
    lib.gsr_shared_terms_scratch.restype = i32
    lib.gsr_shared_terms_scratch.argtypes = [i32, i32]
    lib.gsr_shared_terms_forward.restype = C.c_int
    lib.gsr_shared_terms_forward.argtypes = [i32, i32, i32] + [vp] * 11 + [C.POINTER(C.c_float), vp, vp, vp]
    lib.gsr_shared_terms_backward.restype = C.c_int
    lib.gsr_shared_terms_backward.argtypes = [i32, i32, i32, i32] + [vp] * 11 + [C.POINTER(C.c_float)] + [vp] * 6 + [i32, vp]
    lib.gsr_activate_forward.restype = C.c_int
    lib.gsr_activate_forward.argtypes = [i32] + [vp] * 7
    lib.gsr_activate_backward.restype = C.c_int
    lib.gsr_activate_backward.argtypes = [i32] + [vp] * 10
    lib.gsr_wait_counts.restype = C.c_int64
    lib.gsr_wait_counts.argtypes = [vp, i32, C.c_int64, C.c_int64]
    lib.gsr_radius_bookkeeping.restype = C.c_int
    lib.gsr_radius_bookkeeping.argtypes = [i32, i32, i32, vp, vp, vp, vp]
    lib.gsr_adam_step.restype = C.c_int
    lib.gsr_adam_step.argtypes = [i32, C.POINTER(GsrAdamTensor), vp]
    lib.gsr_rigidity_blocks.restype = i32
    lib.gsr_rigidity_blocks.argtypes = [i32]
    lib.gsr_rigidity_forward.restype = C.c_int
    lib.gsr_rigidity_forward.argtypes = [i32, i32] + [vp] * 9 + [vp]
    lib.gsr_rigidity_backward.restype = C.c_int
    lib.gsr_rigidity_backward.argtypes = [i32, i32] + [vp] * 14 + [vp]
    lib.gsr_fps.restype = C.c_int
    lib.gsr_fps.argtypes = [i32, vp, i32, i32, vp, vp, vp]
    lib.gsr_fps_scratch_bytes.restype = sz
    lib.gsr_fps_scratch_bytes.argtypes = [i32, i32]
    lib.gsr_fit_rotations.restype = C.c_int
    lib.gsr_fit_rotations.argtypes = [i32, vp, vp, vp, vp, vp]
    lib.gsr_construct_edges_dense.restype = C.c_int
    lib.gsr_construct_edges_dense.argtypes = [vp, i32, vp, C.c_float, i32, C.c_int64, i32, vp, vp, vp, vp, i32, vp]
    lib.gsr_rollout_step_tail.restype = C.c_int
    lib.gsr_rollout_step_tail.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_construct_edges.restype = C.c_int
    lib.gsr_construct_edges.argtypes = [vp, i32, vp, C.c_float, i32, C.c_int64, i32, vp, vp, vp, vp]
    lib.gsr_lbs_valid.restype = C.c_int
    lib.gsr_lbs_valid.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_fps_thin.restype = C.c_int
    lib.gsr_fps_thin.argtypes = [i32, vp, i32, i32, C.c_float, i32, vp, vp, vp, vp]
    lib.gsr_gnn_aggregate.restype = C.c_int
    lib.gsr_gnn_aggregate.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.gsr_gnn_rel_inputs.restype = C.c_int
    lib.gsr_gnn_rel_inputs.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.gsr_gnn_aggregate_res.restype = C.c_int
    lib.gsr_gnn_aggregate_res.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsr_construct_edges_rows.restype = C.c_int
    lib.gsr_construct_edges_rows.argtypes = [vp, i32, vp, C.c_float, i32, C.c_int64, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.gsr_rollout_step_head.restype = C.c_int
    lib.gsr_rollout_step_head.argtypes = [i32] * 6 + [vp] * 14
    lib.gsr_arm_depth_cuts.restype = C.c_int
    lib.gsr_arm_depth_cuts.argtypes = [i32, vp, vp, vp, C.c_float]
    lib.gsr_rollout_step_motion.restype = C.c_int
    lib.gsr_rollout_step_motion.argtypes = [i32, i32, C.c_float, vp, vp, vp, vp, vp]
    lib.gsr_fit_bones.restype = C.c_int
    lib.gsr_fit_bones.argtypes = [i32, vp, vp, vp, C.c_int64, vp, vp, vp, vp]
    lib.gsr_lbs.restype = C.c_int
    lib.gsr_lbs.argtypes = [i32, i32] + [vp] * 8 + [vp]
    lib.gsr_mark_visible.restype = C.c_int
    lib.gsr_mark_visible.argtypes = [vp, i32, vp, vp, vp]
    lib.gsr_debug_get_views.restype = C.c_int
    lib.gsr_debug_get_views.argtypes = [i32, u32, i32, i32, vp, vp, vp, C.POINTER(GsrDebugViews)]
    lib.gsr_selftest.restype = C.c_int
    lib.gsr_selftest.argtypes = [vp]
    lib.gsr_profile_begin.restype = C.c_int
    lib.gsr_profile_end.restype = C.c_int
    lib.gsr_profile_end.argtypes = [C.POINTER(GsrKernelTime), i32, C.POINTER(i32)]
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().gsr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_pinned_stream = None    # (device index, handle) while a caller holds the stream fixed over several library calls


def _stream(dev: torch.device):
    p = _pinned_stream
    if p is not None and p[0] == dev.index:
        return p[1]
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class hold_stream:
    """``with hold_stream(dev):`` -- look the current stream up once for a sequence of library calls on it (the lookup costs
    ~5 us per call through torch; the direct tracking step makes ten calls).  Do not change streams inside the block."""

    def __init__(self, dev: torch.device):
        self.dev = dev

    def __enter__(self):
        global _pinned_stream
        self.prev = _pinned_stream
        _pinned_stream = (self.dev.index, C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream))
        return self

    def __exit__(self, *exc):
        global _pinned_stream
        _pinned_stream = self.prev
        return False


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def _on(dev: torch.device):
    """``torch.cuda.device(dev)`` only when dev is not already the current device (the context manager costs ~8 us)."""
    return _NULL if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


_settings_cache = {}   # id(tensor) -> (weakref, version, contiguous fp32 copy on the render device)


def _dev_f32(t: torch.Tensor, dev: torch.device, n: int, name: str) -> torch.Tensor:
    """Settings tensors: n contiguous fp32 values on the render device (viewmatrix may be [1,4,4])."""
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.float32)
    if not (t.is_contiguous() and t.dtype == torch.float32 and t.device == dev):
        # The reference's setup_camera hands over transposed / strided views (helpers.py:14,18): convert once per
        # tensor object and version instead of launching a copy kernel per camera per call.
        hit = _settings_cache.get(id(t))
        if hit is not None and hit[0]() is t and hit[1] == t._version:
            t = hit[2]
        else:
            conv = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            if len(_settings_cache) > 1024:
                _settings_cache.clear()
            _settings_cache[id(t)] = (weakref.ref(t), t._version, conv)
            t = conv
    if t.numel() < n:
        raise ValueError(f"raster_settings.{name} must hold {n} floats, got shape {tuple(t.shape)}")
    return t


class RasterState:
    """What forward hands to backward (the role of the reference extension's three opaque buffers)."""
    __slots__ = ("settings", "keep", "P", "num_rendered", "geom", "binning", "image", "H", "W", "pre", "batch", "geometry_of",
                 "pending", "act", "raw_fused", "forward_only")

    def __init__(self):
        self.forward_only = False  # batch forward with forward_only=True: the states cannot be differentiated
        self.act = None          # batch forward with raw=...: (rotations, opacities, scales) after their activations (view 0's state)
        self.raw_fused = None    # (unnorm_rotations,) when the activations ran inside the forward: the backward applies their chain too


def _make_settings(rs, dev, sh_coeffs: int):
    keep = (_dev_f32(rs.bg, dev, 3, "bg"), _dev_f32(rs.viewmatrix, dev, 16, "viewmatrix"),
            _dev_f32(rs.projmatrix, dev, 16, "projmatrix"), _dev_f32(rs.campos, dev, 3, "campos"))
    s = GsrSettings()
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree, s.sh_coeffs, s.prefiltered = int(rs.sh_degree), int(sh_coeffs), int(bool(rs.prefiltered))
    s.bg, s.viewmatrix, s.projmatrix, s.campos = (k.data_ptr() for k in keep)
    return s, keep


def _require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError(
            "diff_gaussian_rasterization (MI355X build) runs on a HIP device only; got a tensor on "
            f"'{t.device}'. There is no CPU fallback.")


def rasterize_forward(rs, means3D, opacities, colors_precomp, shs, scales, rotations, cov3D_precomp
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, RasterState]:
    """K1..K6.  Returns (color[3,H,W], radii[P] int32, depth[1,H,W], state)."""
    lib = load_library()
    _require_device(means3D)
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(rs.image_height), int(rs.image_width)
    M = 0 if shs is None else int(shs.shape[1])
    with _on(dev):
        s, keep = _make_settings(rs, dev, M)
        u8 = dict(dtype=torch.uint8, device=dev)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)       # preprocess writes every element
        geom = torch.empty((lib.gsr_geom_bytes(P),), **u8)
        image = torch.empty((lib.gsr_image_bytes(H, W),), **u8)
        D = C.c_uint32(0)
        st = _stream(dev)
        _check(lib.gsr_forward_preprocess(C.byref(s), P, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(opacities),
                                          _ptr(colors_precomp), _ptr(shs), _ptr(cov3D_precomp), _ptr(geom),
                                          _ptr(radii), C.byref(D), st), "gsr_forward_preprocess")
        binning = torch.empty((lib.gsr_binning_bytes(D.value, H, W),), **u8)
        _check(lib.gsr_forward_render(C.byref(s), P, D.value, _ptr(geom), _ptr(binning), _ptr(image), _ptr(color),
                                      _ptr(depth), st), "gsr_forward_render")
    state = RasterState()
    state.settings, state.keep, state.P, state.num_rendered = s, keep, P, int(D.value)
    state.geom, state.binning, state.image, state.H, state.W = geom, binning, image, H, W
    state.pre = state.batch = state.geometry_of = state.pending = None
    return color, radii, depth, state


def rasterize_backward(state: RasterState, grad_color, means3D, radii, colors_precomp, shs, scales, rotations,
                       cov3D_precomp, want_color_grad: bool = True):
    """K7..K9.  Returns (dmeans3D, dmeans2D, dcolors, dopacity[P,1], dscales, drotations, dcov3D, dsh).
    ``want_color_grad=False`` (precomputed colours that need no gradient): dcolors is None and the blend backward keeps six sums
    per list entry instead of nine."""
    lib = load_library()
    dev = means3D.device
    P, D = state.P, state.num_rendered
    M = 0 if shs is None else int(shs.shape[1])
    f32 = dict(dtype=torch.float32, device=dev)
    with _on(dev):
        g = grad_color.to(**f32).contiguous()
        d_means3D = torch.empty((P, 3), **f32)
        d_means2D = torch.empty((P, 3), **f32)
        d_colors = torch.empty((P, 3), **f32) if (shs is None and want_color_grad) else None
        d_opacity = torch.empty((P, 1), **f32)
        d_scales = torch.empty((P, 3), **f32) if cov3D_precomp is None else None
        d_rot = torch.empty((P, 4), **f32) if cov3D_precomp is None else None
        d_cov = torch.empty((P, 6), **f32)
        d_sh = torch.empty((P, M, 3), **f32) if shs is not None else None
        scratch = torch.empty((lib.gsr_backward_scratch_bytes(P, D),), dtype=torch.uint8, device=dev)
        _check(lib.gsr_backward(C.byref(state.settings), P, D, _ptr(means3D), _ptr(scales), _ptr(rotations),
                                _ptr(colors_precomp), _ptr(shs), _ptr(cov3D_precomp), _ptr(radii), _ptr(state.geom),
                                _ptr(state.binning), _ptr(state.image), _ptr(g), _ptr(scratch), _ptr(d_means3D),
                                _ptr(d_means2D), _ptr(d_colors), _ptr(d_opacity), _ptr(d_scales), _ptr(d_rot),
                                _ptr(d_cov), _ptr(d_sh), _stream(dev)), "gsr_backward")
    return d_means3D, d_means2D, d_colors, d_opacity, d_scales, d_rot, d_cov, d_sh


MAX_BATCH = 16           # GSR_MAX_BATCH of include/gsr.h: views per library call
_counts_slots = {}       # (device, V) -> list of [pinned int32[V], event, in_flight] of the capacity-mode forward (a small ring:
                         # a slot is taken by one call and handed back by forward_counts_ok, so two calls never share counts)
_COUNTS_RING = 8
_COUNTS_EVENT = os.environ.get("GSR_COUNTS_EVENT") == "1"   # A/B: wait on an event recorded behind the forward instead of polling the pinned slot
_entries_capacity = {}   # (device, P, H, W) -> list entries per view the capacity-mode forward sizes its buffers for
_ENTRIES_SLACK = 1.5
_binning_capacity = {}   # (device, P, H, W) -> bytes to pre-allocate per view for the binning state
_scratch_capacity = {}   # same key -> bytes per view of backward scratch
_BINNING_SLACK = 1.25


def _ptr_array(tensors):
    """Host array of device pointers (void* const*) for the *_batch entry points."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _alloc_backward(dev, V, P, scratch_bytes, with_scale_rot, per_view_col=False, grad_out=None):
    """Output + scratch tensors of the batched backward.  The forward allocates them ahead of its stage-1 wait, where
    the host has slack; between that wait and the backward launch the host is the critical path of a step.
    ``grad_out``: optional dict of caller-owned fp32 tensors the backward writes INSTEAD of fresh ones -- keys ``d_means3D`` [P,3],
    ``d_colors`` [P,3], ``d_opacity`` [P,1], ``d_scales`` [P,3], ``d_rot`` [P,4] (e.g. slices of an all-reduce bucket: no copy later)."""
    f32 = dict(dtype=torch.float32, device=dev)

    def out(name, shape):
        t = None if grad_out is None else grad_out.get(name)
        if t is None:
            return torch.empty(shape, **f32)
        if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
            raise ValueError(f"grad_out[{name!r}] must be a contiguous float32 tensor of shape {tuple(shape)} on {dev}")
        return t
    return dict(
        d_means3D=out("d_means3D", (P, 3)), d_means2D=torch.empty((V, P, 3), **f32),
        d_colors=torch.empty((V, P, 3), **f32) if per_view_col else out("d_colors", (P, 3)),
        d_opacity=out("d_opacity", (P, 1)), d_scales=out("d_scales", (P, 3)) if with_scale_rot else None,
        d_rot=out("d_rot", (P, 4)) if with_scale_rot else None, d_cov=torch.empty((P, 6), **f32),
        scratch=[torch.empty((b,), dtype=torch.uint8, device=dev) for b in scratch_bytes])


FORWARD_ONLY = 1        # GSR_FORWARD_ONLY of include/gsr.h


def rasterize_forward_batch(settings_list, means3D, opacities, colors_precomp, shs, scales, rotations, cov3D_precomp,
                            prepare_backward: bool = False, no_host_sync: bool = False, raw=None, forward_only: bool = False,
                            grad_out=None, depth_cuts=None):
    """All views of a step in one call: one launch per stage for all views, one host sync for all duplicate counts.  Returns (color[V,3,H,W], radii[V,P] int32, depth[V,1,H,W], states[V]).
    ``depth_cuts = (cut_in, cut_out, redo[, margin = 1.01])`` (forward_only calls; include/gsr.h: gsr_arm_depth_cuts): per view a [T] int32 tensor of depth
    bits to bin with (or None), a [T] int32 tensor that receives the next frame's proposal, and one zeroed [V] int32 tensor of redo flags.
    ``raw = (unnorm_rotations, logit_opacities, log_scales)`` (then ``opacities`` / ``scales`` / ``rotations`` are None): the
    activations are applied inside the preprocess kernel when the call runs in capacity mode (``states[0].raw_fused``), by
    ``activate_forward`` otherwise; either way ``states[0].act = (rotations, opacities, scales)`` holds the activated tensors."""
    lib = load_library()
    _require_device(means3D)
    act = None
    if depth_cuts is not None and (no_host_sync or raw is not None or shs is not None):
        raise ValueError("rasterize_forward_batch(depth_cuts=...): the count-first forward-only call with precomputed colours only")
    if raw is not None:
        if cov3D_precomp is not None or shs is not None:
            raise ValueError("rasterize_forward_batch(raw=...): precomputed colours and scales / rotations only")
        key_ = (means3D.device.index, int(means3D.shape[0]), int(settings_list[0].image_height), int(settings_list[0].image_width))
        if not (no_host_sync and int(means3D.shape[0]) > 0 and _entries_capacity.get(key_, 0)):
            act = activate_forward(*raw)               # no capacity yet (first call / overflow repeat): the stand-alone kernel
            rotations, opacities, scales = act
            raw = None
    dev = means3D.device
    V = len(settings_list)
    P = int(means3D.shape[0])
    H, W = int(settings_list[0].image_height), int(settings_list[0].image_width)
    if any(int(rs.image_height) != H or int(rs.image_width) != W for rs in settings_list):
        raise ValueError("rasterize_forward_batch: all views must share the image size")
    M = 0 if shs is None else int(shs.shape[1])
    if shs is not None:
        # SH colours depend on the camera position, and the multi-view backward kernel covers precomputed colours only: every
        # view takes the single-view entry points, so that each state owns the tile order / queue / binning that
        # gsr_backward reads (the batch state of a multi-view call keeps them in one shared table instead).
        outs = [rasterize_forward(rs, means3D, opacities, None, shs, scales, rotations, cov3D_precomp) for rs in settings_list]
        return (torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.stack([o[2] for o in outs]),
                [o[3] for o in outs])
    with _on(dev):
        sarr = (GsrSettings * V)()
        keeps = []
        for v, rs in enumerate(settings_list):
            s, keep = _make_settings(rs, dev, M)
            sarr[v] = s
            keeps.append(keep)
        u8 = dict(dtype=torch.uint8, device=dev)
        color = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((V, P), dtype=torch.int32, device=dev)     # preprocess writes every element
        geoms = [torch.empty((lib.gsr_geom_bytes(P),), **u8) for _ in range(V)]
        images = [torch.empty((lib.gsr_image_bytes(H, W),), **u8) for _ in range(V)]
        batch = torch.empty((lib.gsr_batch_state_bytes(V, P, H, W),), **u8)
        Ds = (C.c_uint32 * V)()
        st = _stream(dev)
        # Binning buffers sized from the previous call with the same shape (+25 %): the library then goes from stage 1
        # to stage 2 without returning here in between.  First call / grown scene: it returns 1 and stage 2 is
        # launched from here with exact sizes.
        key = (dev.index, P, H, W)
        cap = _binning_capacity.get(key, 0)
        # views with the same camera (the colour and the segmentation render of get_loss) share one set of tile lists
        first, geo = {}, []
        for v, rs in enumerate(settings_list):
            k = (id(rs.viewmatrix), id(rs.projmatrix), id(rs.campos), float(rs.tanfovx), float(rs.tanfovy),
                 float(rs.scale_modifier), bool(rs.prefiltered))
            geo.append(first.setdefault(k, v))
        geometry_of = (C.c_int32 * V)(*geo) if any(g != v for v, g in enumerate(geo)) else None
        owner = [geometry_of is None or geo[v] == v for v in range(V)]
        cap_e = _entries_capacity.get(key, 0) if (no_host_sync and shs is None and P > 0) else 0
        if cap_e:
            # Capacity mode (gsr_forward_batch_capacity): no host wait inside the forward.  Buffers are sized for cap_e entries per
            # view; the true counts come back through a pinned copy that `forward_counts_ok` inspects later.  The caller must not
            # let anything of this call escape before that check (gsdyn.step.loss_and_grads_views repeats the call on overflow).
            bbytes = int(lib.gsr_binning_bytes(cap_e, H, W))
            binnings = [torch.empty((bbytes,), **u8) if owner[v] else None for v in range(V)]
            pre = None
            if prepare_backward:
                pre = _alloc_backward(dev, V, P, [int(lib.gsr_backward_scratch_bytes(P, cap_e))] * V, cov3D_precomp is None,
                                      colors_precomp is not None and colors_precomp.dim() == 3, grad_out)
            color_v, depth_v = [color[v] for v in range(V)], [depth[v] for v in range(V)]
            per_view_col = colors_precomp is not None and colors_precomp.dim() == 3
            if per_view_col and colors_precomp.shape[0] != V:
                raise ValueError("rasterize_forward_batch: per-view colours must be [V,P,3]")
            col_views = _ptr_array([colors_precomp[v] for v in range(V)]) if per_view_col else None
            capv = (C.c_uint32 * V)(*([cap_e] * V))
            rawp = None
            if raw is not None:     # activations inside the preprocess kernel: its view-0 blocks write the activated values here
                f32_ = dict(dtype=torch.float32, device=dev)
                rotations, opacities, scales = torch.empty((P, 4), **f32_), torch.empty((P, 1), **f32_), torch.empty((P, 3), **f32_)
                act = (rotations, opacities, scales)
                un_, lo_, ls_ = (t.contiguous() for t in raw)
                rawp = GsrRawParams(_ptr(un_), _ptr(lo_), _ptr(ls_), _ptr(rotations), _ptr(opacities), _ptr(scales), None, None, None)
            # the entry counts land in a pinned host slot straight from the tile-order kernel (device-mapped memory): no copy on the stream
            ring = _counts_slots.setdefault((dev.index, V), [])
            slot = next((sl for sl in ring if not sl[2]), None)
            if slot is None and len(ring) >= _COUNTS_RING:
                # Every slot is held by a call whose counts were never looked at.  A caller that raised between its forward and
                # forward_counts_ok() (or dropped the states) leaves such a slot behind: take over the oldest one whose forward has
                # completed -- its generation changes, so a late forward_counts_ok() of the abandoned call raises instead of reading
                # another call's counts.
                slot = next((sl for sl in ring if (sl[1].query() if _COUNTS_EVENT else int(sl[0].min()) >= 0)), None)
                if slot is None:
                    raise RuntimeError(f"rasterize_forward_batch(no_host_sync=True): {_COUNTS_RING} capacity-mode forwards are in flight "
                                       "without forward_counts_ok(); check each call's counts before issuing more")
            if slot is None:      # pinned staging + event, allocated once per ring entry (hipHostMalloc costs tens of microseconds)
                slot = [torch.empty((V,), dtype=torch.int32, pin_memory=True), torch.cuda.Event(), False, 0]
                ring.append(slot)
            slot[2] = True
            slot[3] += 1          # generation
            counts_host, ev = slot[0], slot[1]
            counts_dev = counts_host
            counts_host.fill_(-1)     # sentinel (no view has 2^32 - 1 entries): forward_counts_ok polls for the kernel's stores
            _check(lib.gsr_forward_batch_capacity_raw(V, sarr, P, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(opacities),
                                                      _ptr(None if per_view_col else colors_precomp), col_views, None, _ptr(cov3D_precomp),
                                                      _ptr_array(geoms), _ptr_array([radii[v] for v in range(V)]), _ptr_array(binnings),
                                                      capv, _ptr_array(images), _ptr(batch), geometry_of, _ptr_array(color_v),
                                                      _ptr_array(depth_v), _ptr(counts_dev), C.byref(rawp) if rawp is not None else None, st),
                   "gsr_forward_batch_capacity")
            if _COUNTS_EVENT:
                ev.record(torch.cuda.current_stream(dev))
            states = []
            for v in range(V):
                state = RasterState()
                state.settings, state.keep, state.P, state.num_rendered = sarr[v], keeps[v], P, int(cap_e)   # capacities fix the layouts
                state.geom, state.binning, state.image, state.H, state.W = geoms[v], binnings[v], images[v], H, W
                state.pre = pre if v == 0 else None
                state.batch = batch if v == 0 else None
                state.geometry_of = geometry_of if v == 0 else None
                state.pending = (ev, counts_host, counts_dev, int(cap_e), key, slot, slot[3]) if v == 0 else None
                state.act = act if v == 0 else None
                state.raw_fused = (un_, ) if (v == 0 and rawp is not None) else None
                states.append(state)
            return color, radii, depth, states
        binnings = [torch.empty((cap,), **u8) if (cap and owner[v]) else None for v in range(V)]
        caps = (C.c_size_t * V)(*([cap] * V))
        pre = None
        if prepare_backward and cap and shs is None:   # scratch sized like the binning buffers: from the last call
            pre = _alloc_backward(dev, V, P, [_scratch_capacity.get(key, 0)] * V, cov3D_precomp is None,
                                  colors_precomp is not None and colors_precomp.dim() == 3, grad_out)
        color_v, depth_v = [color[v] for v in range(V)], [depth[v] for v in range(V)]
        per_view_col = colors_precomp is not None and colors_precomp.dim() == 3   # [V,P,3]: every view its own colours
        if per_view_col and colors_precomp.shape[0] != V:
            raise ValueError("rasterize_forward_batch: per-view colours must be [V,P,3]")
        col_shared = None if per_view_col else colors_precomp
        col_views = _ptr_array([colors_precomp[v] for v in range(V)]) if per_view_col else None
        if depth_cuts is not None:
            if not forward_only or geometry_of is not None:
                raise ValueError("rasterize_forward_batch(depth_cuts=...): forward_only calls of views with their own tile lists only")
            cin, cout, redo = depth_cuts[:3]
            margin = float(depth_cuts[3]) if len(depth_cuts) > 3 else 1.01
            T_ = ((H + 15) // 16) * ((W + 15) // 16)
            for t_ in list(cout) + [c for c in (cin or []) if c is not None]:
                if not (t_.is_contiguous() and t_.dtype == torch.int32 and t_.numel() == T_ and t_.device == dev):
                    raise ValueError("rasterize_forward_batch(depth_cuts=...): contiguous int32 tensors of one word per tile, please")
            if not (redo.dtype == torch.int32 and redo.numel() == V and redo.is_contiguous() and redo.device == dev):
                raise ValueError("rasterize_forward_batch(depth_cuts=...): redo = a contiguous int32 tensor of V words")
            _check(lib.gsr_arm_depth_cuts(V, _ptr_array(list(cin)) if cin is not None else None, _ptr_array(list(cout)), _ptr(redo), margin), "gsr_arm_depth_cuts")
        try:
            rc = lib.gsr_forward_batch(V, sarr, P, _ptr(means3D), _ptr(scales), _ptr(rotations), _ptr(opacities),
                                       _ptr(col_shared), col_views, _ptr(shs), _ptr(cov3D_precomp), _ptr_array(geoms),
                                       _ptr_array([radii[v] for v in range(V)]), _ptr_array(binnings), caps,
                                       _ptr_array(images), _ptr(batch), geometry_of, _ptr_array(color_v), _ptr_array(depth_v), Ds,
                                       FORWARD_ONLY if forward_only else 0, st)
            if rc not in (0, 1):
                _check(rc, "gsr_forward_batch")
            need = max(lib.gsr_binning_bytes(Ds[v], H, W) for v in range(V))
            if rc == 1:
                binnings = [torch.empty((lib.gsr_binning_bytes(Ds[v], H, W),), **u8) if owner[v] else None for v in range(V)]
                _check(lib.gsr_forward_render_batch(V, sarr, P, Ds, _ptr_array(geoms), _ptr_array(binnings), _ptr_array(images),
                                                    _ptr(batch), geometry_of, col_views, _ptr_array(color_v), _ptr_array(depth_v),
                                                    FORWARD_ONLY if forward_only else 0, st),
                       "gsr_forward_render_batch")
        finally:
            if depth_cuts is not None:
                lib.gsr_arm_depth_cuts(0, None, None, None, 1.0)      # (consumed by the binning above; disarmed here if the call failed before it)
        if forward_only and geometry_of is not None:      # a fused alias is not preprocessed in this mode: its radii are its owner's
            if V % 2 == 0 and all(geo[v] == v - (v & 1) for v in range(V)):      # (colour, mask) pairs: one strided copy
                radii[1::2].copy_(radii[0::2])
            else:
                for v in range(V):
                    if geo[v] != v:
                        radii[v].copy_(radii[geo[v]])
        _entries_capacity[key] = max(int(max(Ds[v] for v in range(V)) * _ENTRIES_SLACK), 1024)
        if rc == 1 or need * 2 < cap:
            _binning_capacity[key] = int(need * _BINNING_SLACK)
            _scratch_capacity[key] = int(max(lib.gsr_backward_scratch_bytes(P, Ds[v]) for v in range(V)) * _BINNING_SLACK)
        if pre is not None and any(lib.gsr_backward_scratch_bytes(P, Ds[v]) > pre["scratch"][v].numel() for v in range(V)):
            pre = None
    states = []
    for v in range(V):
        state = RasterState()
        state.settings, state.keep, state.P, state.num_rendered = sarr[v], keeps[v], P, int(Ds[v])
        state.geom, state.binning, state.image, state.H, state.W = geoms[v], binnings[v], images[v], H, W
        state.pre = pre if v == 0 else None
        state.batch = batch if v == 0 else None
        state.geometry_of = geometry_of if v == 0 else None
        state.pending = None
        state.act = act if v == 0 else None
        state.forward_only = bool(forward_only)
        states.append(state)
    return color, radii, depth, states


def detach_counts_slot(states) -> None:
    """Take the pinned counts slot of a capacity-mode forward OUT of the ring for good: a captured graph (gsdyn.step.GraphedRenderStep)
    rewrites its slot on every replay and never hands it back through forward_counts_ok, so the ring's take-over rule (oldest slot whose
    counts have arrived) must never give it to an eager call -- that call would read the graph's counts as its own."""
    pending = states[0].pending
    if pending is None:
        return
    slot = pending[5]
    for ring in _counts_slots.values():
        if any(sl is slot for sl in ring):
            ring[:] = [sl for sl in ring if sl is not slot]


def forward_counts_ok(states) -> bool:
    """After a capacity-mode forward: wait for its entry counts (long since on the host when the caller did anything in
    between), remember them for the next call's capacity, and tell whether every view fitted.  True for a synchronous forward."""
    pending = states[0].pending
    if pending is None:
        return True
    ev, counts_host, _counts_dev, cap_e, key, slot, gen = pending
    if slot[3] != gen:
        raise RuntimeError("forward_counts_ok: this capacity-mode forward was abandoned (its counts slot serves a later call)")
    if _COUNTS_EVENT:
        ev.synchronize()
    else:
        # The tile-order kernel writes the counts with system-scope stores into this pinned slot: poll for them instead of recording
        # an event behind the forward (the event's signal packet costs ~6 us of idle GPU between render_fwd and render_bwd).
        # The wait itself is C (gsr_wait_counts: reads the pinned words, pauses, yields after 200 us; ctypes drops the interpreter lock):
        # round 3 spun here with ~330 Tensor.min() calls per step through the torch dispatcher -- a core per rank at 100 %.
        lib = load_library()
        top = int(lib.gsr_wait_counts(counts_host.data_ptr(), int(counts_host.numel()), 200, 100_000))
        if top < 0:                  # 0.1 s: something is wrong (or the stores are not visible before the kernel ends): fall back
            torch.cuda.current_stream(torch.device("cuda", key[0])).synchronize()
            if int(counts_host.min()) < 0:
                raise RuntimeError("forward_counts_ok: the forward never wrote its entry counts")
    top = int(counts_host.max())
    slot[2] = False          # the ring entry may serve the next call
    _entries_capacity[key] = max(int(top * _ENTRIES_SLACK), 1024)
    states[0].pending = None
    return top <= cap_e


def rasterize_backward_batch(states, grad_color, means3D, radii, colors_precomp, shs, scales, rotations, cov3D_precomp,
                             want_color_grad: bool = True, grad_out=None):
    """Backward of all views.  Returns gradients already SUMMED over views (dmeans3D[P,3], dcolors, dopacity[P,1],
    dscales, drotations, dcov3D, dsh) plus the per-view means2D gradients [V,P,3]."""
    lib = load_library()
    dev = means3D.device
    V = len(states)
    P = states[0].P
    if states[0].forward_only:
        raise RuntimeError("rasterize_backward_batch: these states come from a forward_only forward (no record-slot offsets were produced)")
    f32 = dict(dtype=torch.float32, device=dev)
    if shs is not None:  # SH colours: per-view backward + sum (the fused multi-view kernel covers precomputed colours)
        outs = [rasterize_backward(states[v], grad_color[v], means3D, radii[v], None, shs, scales, rotations, cov3D_precomp)
                for v in range(V)]
        sm = lambda k: None if outs[0][k] is None else torch.stack([o[k] for o in outs]).sum(0)  # noqa: E731
        return sm(0), torch.stack([o[1] for o in outs]), None, sm(3), sm(4), sm(5), sm(6), sm(7)
    with _on(dev):
        g = grad_color.to(**f32).contiguous()
        sarr = (GsrSettings * V)()
        Ds = (C.c_uint32 * V)()
        for v, stt in enumerate(states):
            sarr[v] = stt.settings
            Ds[v] = stt.num_rendered
        per_view_col = colors_precomp is not None and colors_precomp.dim() == 3
        pre, states[0].pre = states[0].pre, None   # one use only: autograd may keep the returned tensors as .grad
        if pre is None:
            pre = _alloc_backward(dev, V, P, [lib.gsr_backward_scratch_bytes(P, stt.num_rendered) for stt in states],
                                  cov3D_precomp is None, per_view_col, grad_out)
        d_means3D, d_means2D, d_colors, d_opacity = pre["d_means3D"], pre["d_means2D"], pre["d_colors"], pre["d_opacity"]
        d_scales, d_rot, d_cov, scratch = pre["d_scales"], pre["d_rot"], pre["d_cov"], pre["scratch"]

        def per_view(t):
            return _ptr_array([t[v] for v in range(V)])
        rawp, fused = None, states[0].raw_fused
        if fused is not None:    # the chain through the activations runs inside the per-Gaussian kernel: d_rot / d_opacity / d_scales
            #                      come back as the gradients of the UNACTIVATED parameters (same shapes)
            rawp = GsrRawParams(_ptr(fused[0]), None, None, _ptr(rotations), _ptr(states[0].act[1]), _ptr(scales),
                                _ptr(d_rot), _ptr(d_opacity), _ptr(d_scales))
        _check(lib.gsr_backward_batch_raw(V, sarr, P, Ds, _ptr(means3D), _ptr(scales), _ptr(rotations),
                                          _ptr(None if per_view_col else colors_precomp),
                                          _ptr(cov3D_precomp), per_view(radii), _ptr_array([stt.geom for stt in states]),
                                          _ptr_array([stt.binning for stt in states]), _ptr_array([stt.image for stt in states]),
                                          _ptr(states[0].batch), states[0].geometry_of, per_view(g), _ptr_array(scratch), _ptr(d_means3D),
                                          per_view(d_means2D),
                                          _ptr(None if (per_view_col or not want_color_grad) else d_colors),
                                          per_view(d_colors) if (per_view_col and want_color_grad) else None,
                                          _ptr(None if fused is not None else d_opacity), _ptr(None if fused is not None else d_scales),
                                          _ptr(None if fused is not None else d_rot), _ptr(d_cov),
                                          C.byref(rawp) if rawp is not None else None, _stream(dev)),
               "gsr_backward_batch")
    # without a colour gradient, views that share a camera stay fused in the backward (one replay of the tile lists for both)
    return d_means3D, d_means2D, (d_colors if want_color_grad else None), d_opacity, d_scales, d_rot, d_cov, None


def rigidity_forward(means3D, rotations, fg_idx, nbr, nw, nd, prev_inv, prev_off):
    """Sums of the rigid / rot / iso terms over all (foreground point, neighbour) pairs: tensor [3] (gsr_rigidity_forward)."""
    lib = load_library()
    _require_device(means3D)
    dev = means3D.device
    nfg, K = int(nbr.shape[0]), int(nbr.shape[1])
    with _on(dev):
        nb = int(lib.gsr_rigidity_blocks(nfg))
        part = torch.empty((3, max(nb, 1)), dtype=torch.float32, device=dev)
        if nb == 0:
            part.zero_()
        _check(lib.gsr_rigidity_forward(nfg, K, _ptr(means3D), _ptr(rotations), _ptr(fg_idx), _ptr(nbr), _ptr(nw), _ptr(nd),
                                        _ptr(prev_inv), _ptr(prev_off), _ptr(part), _stream(dev)), "gsr_rigidity_forward")
    return part.sum(1)


def rigidity_backward(means3D, rotations, fg_idx, nbr, nw, nd, prev_inv, prev_off, grad3, rev_ptr, rev_edge):
    """grad3 [3]: upstream gradients of the three MEANS divided by n_fg * K.  Returns (d_means3D [P,3], d_rotations [P,4])."""
    lib = load_library()
    dev = means3D.device
    nfg, K = int(nbr.shape[0]), int(nbr.shape[1])
    with _on(dev):
        scratch = torch.empty((7 * (nfg + nfg * K),), dtype=torch.float32, device=dev)
        d_m = torch.zeros_like(means3D)
        d_r = torch.zeros_like(rotations)
        _check(lib.gsr_rigidity_backward(nfg, K, _ptr(means3D), _ptr(rotations), _ptr(fg_idx), _ptr(nbr), _ptr(nw), _ptr(nd),
                                         _ptr(prev_inv), _ptr(prev_off), _ptr(grad3), _ptr(rev_ptr), _ptr(rev_edge), _ptr(scratch),
                                         _ptr(d_m), _ptr(d_r), _stream(dev)), "gsr_rigidity_backward")
    return d_m, d_r


def _shared_args(v):
    return (_ptr(v["fg_idx"]), _ptr(v["bg_idx"]), _ptr(v["neighbor_indices"]), _ptr(v["neighbor_weight"]), _ptr(v["neighbor_dist"]),
            _ptr(v["prev_inv_rot_fg"]), _ptr(v["prev_offset"]), _ptr(v["init_bg_pts"]), _ptr(v["init_bg_rot"]))


def shared_terms_forward(means3D, rotations, v, weights5):
    """View-independent terms of the t > 0 loss (gsr_shared_terms_forward).  ``v``: the contiguous tensors of
    gsdyn.step.make_rigidity_variables.  Returns (terms[6] = rigid, rot, iso, floor, bg, weighted sum; work buffer to hand to
    ``shared_terms_backward``: it starts with the per-point frames the backward needs again)."""
    lib = load_library()
    _require_device(means3D)
    dev = means3D.device
    nfg, K = (int(d) for d in v["neighbor_indices"].shape)
    nbg = int(v["bg_idx"].shape[0])
    w5 = (C.c_float * 5)(*[float(x) for x in weights5])
    with _on(dev):
        n = max(int(lib.gsr_shared_terms_partials(nfg, nbg)), int(lib.gsr_shared_terms_scratch(nfg, K)), 1)
        work = torch.empty((n,), dtype=torch.float32, device=dev)
        terms = torch.empty((6,), dtype=torch.float32, device=dev)
        _check(lib.gsr_shared_terms_forward(nfg, K, nbg, _ptr(means3D), _ptr(rotations), *_shared_args(v), w5, _ptr(work), _ptr(terms),
                                            _stream(dev)), "gsr_shared_terms_forward")
    return terms, work


def shared_terms_backward(means3D, rotations, v, weights5, grad_total, accumulate_into=None, work=None):
    """``accumulate_into`` = (d_means3D, d_rotations): the gradients are ADDED to these tensors (and they are returned).
    ``work``: the buffer ``shared_terms_forward`` returned for the same inputs (saves recomputing the per-point frames)."""
    lib = load_library()
    dev = means3D.device
    nfg, K = (int(d) for d in v["neighbor_indices"].shape)
    nbg = int(v["bg_idx"].shape[0])
    w5 = (C.c_float * 5)(*[float(x) for x in weights5])
    with _on(dev):
        g = grad_total.to(dtype=torch.float32, device=dev).reshape(1)
        need = max(int(lib.gsr_shared_terms_scratch(nfg, K)), 1)
        flags = 1 if accumulate_into is not None else 0
        if work is not None and work.numel() >= need:
            scratch, flags = work, flags | 2
        else:
            scratch = torch.empty((need,), dtype=torch.float32, device=dev)
        d_m, d_r = accumulate_into if accumulate_into is not None else (torch.empty_like(means3D), torch.empty_like(rotations))
        _check(lib.gsr_shared_terms_backward(int(means3D.shape[0]), nfg, K, nbg, _ptr(means3D), _ptr(rotations), *_shared_args(v), w5,
                                             _ptr(g), _ptr(v["rev_ptr"]), _ptr(v["rev_edge"]), _ptr(scratch), _ptr(d_m), _ptr(d_r),
                                             flags, _stream(dev)), "gsr_shared_terms_backward")
    return d_m, d_r


def activate_forward(unnorm_rotations, logit_opacities, log_scales):
    lib = load_library()
    _require_device(unnorm_rotations)
    dev = unnorm_rotations.device
    P = int(unnorm_rotations.shape[0])
    with _on(dev):
        rot, op, sc = torch.empty_like(unnorm_rotations), torch.empty_like(logit_opacities), torch.empty_like(log_scales)
        _check(lib.gsr_activate_forward(P, _ptr(unnorm_rotations), _ptr(logit_opacities), _ptr(log_scales), _ptr(rot), _ptr(op), _ptr(sc),
                                        _stream(dev)), "gsr_activate_forward")
    return rot, op, sc


def activate_backward(unnorm_rotations, opacities, scales, d_rot, d_op, d_sc, out=None):
    """``out`` = (d_unnorm_rotations, d_logit_opacities, d_log_scales) caller-owned contiguous fp32 tensors to write instead of fresh
    ones (entries may be None)."""
    lib = load_library()
    dev = unnorm_rotations.device
    P = int(unnorm_rotations.shape[0])
    with _on(dev):
        o = out if out is not None else (None, None, None)
        pick = lambda t, like: t if (t is not None and t.is_contiguous() and t.dtype == torch.float32 and t.shape == like.shape) else torch.empty_like(like)  # noqa: E731
        d_u, d_l, d_s = pick(o[0], unnorm_rotations), pick(o[1], opacities), pick(o[2], scales)
        c = lambda t: None if t is None else t.contiguous()   # noqa: E731
        d_rot, d_op, d_sc = c(d_rot), c(d_op), c(d_sc)
        _check(lib.gsr_activate_backward(P, _ptr(unnorm_rotations), _ptr(opacities), _ptr(scales), _ptr(d_rot), _ptr(d_op), _ptr(d_sc),
                                         _ptr(d_u), _ptr(d_l), _ptr(d_s), _stream(dev)), "gsr_activate_backward")
    return d_u, d_l, d_s


def radius_bookkeeping(radii, view_step, max_2D_radius):
    """In-place max_2D_radius update and the ``seen`` mask (bool [P]) from the rows 0, view_step, ... of radii[V,P] (int32)."""
    lib = load_library()
    dev = radii.device
    V, P = (int(d) for d in radii.shape)
    with _on(dev):
        seen = torch.empty((P,), dtype=torch.bool, device=dev)
        _check(lib.gsr_radius_bookkeeping(V, int(view_step), P, _ptr(radii), _ptr(max_2D_radius), _ptr(seen), _stream(dev)),
               "gsr_radius_bookkeeping")
    return seen


def adam_step(entries):
    """One launch for the Adam update of several tensors.  ``entries``: (param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
    step) with contiguous fp32 HIP tensors and ``step`` >= 1 (the count AFTER this update, as torch keeps it)."""
    lib = load_library()
    if not entries:
        return
    dev = entries[0][0].device
    with _on(dev):
        st = _stream(dev)
        for lo in range(0, len(entries), ADAM_MAX_TENSORS):
            chunk = entries[lo:lo + ADAM_MAX_TENSORS]
            arr = (GsrAdamTensor * len(chunk))()
            for a_, (p, g, m, v, lr, b1, b2, eps, step) in zip(arr, chunk):
                a_.param, a_.grad, a_.exp_avg, a_.exp_avg_sq, a_.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                a_.lr, a_.beta1, a_.beta2, a_.eps = lr, b1, b2, eps
                a_.bias_correction1 = 1.0 - b1 ** step
                a_.bias_correction2_sqrt = (1.0 - b2 ** step) ** 0.5
                a_.one_minus_beta1, a_.one_minus_beta2 = 1.0 - b1, 1.0 - b2
            _check(lib.gsr_adam_step(len(chunk), arr, st), "gsr_adam_step")


def farthest_point_sampling(pos: torch.Tensor, npoints: int, start_idx: int = 0) -> torch.Tensor:
    """pos [N,3] fp32 on a HIP device -> int64 indices [npoints] (gsr_fps)."""
    lib = load_library()
    _require_device(pos)
    N = int(pos.shape[0])
    with torch.cuda.device(pos.device):
        scratch = torch.empty((int(lib.gsr_fps_scratch_bytes(N, int(npoints))),), dtype=torch.uint8, device=pos.device)
        out = torch.empty((npoints,), dtype=torch.int64, device=pos.device)
        _check(lib.gsr_fps(N, _ptr(pos), int(npoints), int(start_idx), _ptr(scratch), _ptr(out), _stream(pos.device)), "gsr_fps")
    return out


def fit_rotations(moments: torch.Tensor, n_related: torch.Tensor):
    """moments [nb,3,3] fp32, n_related [nb] (any dtype) on a HIP device -> (rotations [nb,3,3], code [nb] int32), see gsr_fit_rotations."""
    lib = load_library()
    _require_device(moments)
    dev = moments.device
    nb = int(moments.shape[0])
    with _on(dev):
        F = moments.to(torch.float32).contiguous()
        n = n_related.to(torch.float32).contiguous()
        R = torch.empty((nb, 3, 3), dtype=torch.float32, device=dev)
        code = torch.empty((nb,), dtype=torch.int32, device=dev)
        _check(lib.gsr_fit_rotations(nb, _ptr(F), _ptr(n), _ptr(R), _ptr(code), _stream(dev)), "gsr_fit_rotations")
    return R, code


def fps_thin_padded(pos: torch.Tensor, npoints: int, radius: float, start_idx: int = 0, thin_start_idx: int = 0):
    """gsr_fps_thin without the read-back: (fps indices [npoints], kept positions PADDED to [npoints] with thin_start_idx, count [1] int32
    on the device) -- for callers with fixed shapes (a rollout step replayed from a graph)."""
    lib = load_library()
    _require_device(pos)
    dev = pos.device
    N = int(pos.shape[0])
    if not (0 < N <= 1024 and 0 < npoints <= N):
        raise ValueError("fps_thin_padded: 1 <= npoints <= N <= 1024")
    with _on(dev):
        p = pos.to(torch.float32).contiguous()
        out = torch.empty((npoints,), dtype=torch.int64, device=dev)
        thin = torch.empty((npoints,), dtype=torch.int64, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        _check(lib.gsr_fps_thin(N, _ptr(p), int(npoints), int(start_idx), float(radius), int(thin_start_idx), _ptr(out), _ptr(thin), _ptr(cnt),
                                _stream(dev)), "gsr_fps_thin")
    return out, thin, cnt


def construct_edges_padded(pos: torch.Tensor, n_valid: torch.Tensor, thresh: float, topk: int, e_cap: int, dummy: int, dense_n: int = 0,
                           row_start: bool = False):
    """gsr_construct_edges: pos [n_obj_cap + 1, 3] (the tool last), n_valid [1] int32 on the device -> (receivers [e_cap], senders [e_cap]
    int64 padded with ``dummy``, count [1] int32); dense_n > 0: also the relations as a dense [dense_n, dense_n] int64 0 / 1 matrix
    (gsr_construct_edges_dense) as a fourth result; row_start (with dense_n): also the segment bounds [dense_n + 1] of the receiver-sorted
    list (gsr_construct_edges_rows: what searchsorted(receivers, arange(dense_n + 1)) gives) as a fifth."""
    import numpy as np
    lib = load_library()
    _require_device(pos)
    dev = pos.device
    with _on(dev):
        p = pos.to(torch.float32).contiguous()
        recv = torch.empty((e_cap,), dtype=torch.int64, device=dev)
        send = torch.empty((e_cap,), dtype=torch.int64, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        thr2 = float(np.float32(float(thresh) * float(thresh)))          # the scalar a float32 tensor is compared with
        if dense_n and row_start:
            rel = torch.empty((int(dense_n), int(dense_n)), dtype=torch.int64, device=dev)
            rows = torch.empty((int(dense_n) + 1,), dtype=torch.int64, device=dev)
            _check(lib.gsr_construct_edges_rows(_ptr(p), int(p.shape[0]) - 1, _ptr(n_valid), thr2, int(topk), int(dummy), int(e_cap), _ptr(recv), _ptr(send),
                                                _ptr(cnt), _ptr(rel), int(dense_n), _ptr(rows), _stream(dev)), "gsr_construct_edges_rows")
            return recv, send, cnt, rel, rows
        if dense_n:
            rel = torch.empty((int(dense_n), int(dense_n)), dtype=torch.int64, device=dev)
            _check(lib.gsr_construct_edges_dense(_ptr(p), int(p.shape[0]) - 1, _ptr(n_valid), thr2, int(topk), int(dummy), int(e_cap), _ptr(recv), _ptr(send),
                                                 _ptr(cnt), _ptr(rel), int(dense_n), _stream(dev)), "gsr_construct_edges_dense")
            return recv, send, cnt, rel
        _check(lib.gsr_construct_edges(_ptr(p), int(p.shape[0]) - 1, _ptr(n_valid), thr2, int(topk), int(dummy), int(e_cap), _ptr(recv), _ptr(send),
                                       _ptr(cnt), _stream(dev)), "gsr_construct_edges")
    return recv, send, cnt


def rollout_step_head(hist, sample_idx, thin_idx, eef_hist, eef_next, attrs, instance, with_state: bool):
    """gsr_rollout_step_head (include/gsr.h): the inputs of a graphed rollout step's network from the tracked particles' history and the
    step's bone picks, one launch.  hist [n_his, n_track, 3], sample_idx / thin_idx [n_bones] int64, eef_hist [n_his, 1, 3], eef_next [1, 3],
    attrs [n_rows, A], instance [n_rows, 1]; returns (bones_last [nb,3], states_last [nb+1,3], state_rows [n_rows, 3 n_his], action_rows
    [n_rows,3], particle_inputs, rel_nodes)."""
    lib = load_library()
    _require_device(hist)
    dev = hist.device
    n_his, n_track, nb, n_rows, A = int(hist.shape[0]), int(hist.shape[1]), int(sample_idx.shape[0]), int(attrs.shape[0]), int(attrs.shape[1])
    for t in (hist, eef_hist, eef_next, attrs, instance):
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.device == dev):
            raise ValueError("rollout_step_head: contiguous float32 tensors on one device, please")
    with _on(dev):
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)  # noqa: E731
        bones, states, st, act = e(nb, 3), e(nb + 1, 3), e(n_rows, 3 * n_his), e(n_rows, 3)
        p_in, nodes = e(n_rows, A + (3 * n_his if with_state else 0) + 3), e(n_rows, A + 1 + 3 * n_his)
        _check(lib.gsr_rollout_step_head(n_track, n_his, nb, n_rows, A, 1 if with_state else 0, _ptr(hist), _ptr(sample_idx), _ptr(thin_idx), _ptr(eef_hist),
                                         _ptr(eef_next), _ptr(attrs), _ptr(instance), _ptr(bones), _ptr(states), _ptr(st), _ptr(act), _ptr(p_in), _ptr(nodes),
                                         _stream(dev)), "gsr_rollout_step_head")
    return bones, states, st, act, p_in, nodes


def rollout_step_motion(state_rows, pred_motion, n_valid, packet, n_bones: int, n_his: int, motion_clamp: float):
    """gsr_rollout_step_motion (include/gsr.h): predicted bone positions and motions of a step into its skinning packet (head, bones,
    motions, predicted blocks); state_rows [n_rows, 3 n_his], pred_motion [n_rows, 3] contiguous float32, n_valid [1] int32."""
    lib = load_library()
    dev = packet.device
    if not (pred_motion.is_contiguous() and state_rows.is_contiguous() and packet.is_contiguous() and int(packet.numel()) >= 2 + 22 * int(n_bones)):
        raise ValueError("rollout_step_motion: contiguous tensors and a packet of 2 + 22 n_bones floats, please")
    with _on(dev):
        _check(lib.gsr_rollout_step_motion(int(n_bones), int(n_his), float(motion_clamp), _ptr(state_rows), _ptr(pred_motion), _ptr(n_valid), _ptr(packet),
                                           _stream(dev)), "gsr_rollout_step_motion")


def rollout_step_tail(all_pos, track, pos_track, hist, eef_hist, eef_next, pred_in, n_valid, code, pred_out, n_valid_out, bad):
    """gsr_rollout_step_tail: the in-place bookkeeping that ends a graphed rollout step (see include/gsr.h); every tensor on the device, contiguous."""
    lib = load_library()
    dev = all_pos.device
    with _on(dev):
        _check(lib.gsr_rollout_step_tail(int(track.shape[0]), int(hist.shape[0]), int(pred_out.shape[0]), _ptr(all_pos), _ptr(track), _ptr(pos_track),
                                         _ptr(hist), _ptr(eef_hist), _ptr(eef_next), _ptr(pred_in), _ptr(n_valid), _ptr(code), _ptr(pred_out),
                                         _ptr(n_valid_out), _ptr(bad), _stream(dev)), "gsr_rollout_step_tail")


def fps_thin(pos: torch.Tensor, npoints: int, radius: float, start_idx: int = 0, thin_start_idx: int = 0):
    """gsr_fps_thin: pos [N,3] (N <= 1024) on a HIP device -> (fps indices [npoints] int64, kept positions in that list [M] int64) -- one
    launch, one 4-byte read-back for M."""
    lib = load_library()
    _require_device(pos)
    dev = pos.device
    N = int(pos.shape[0])
    npoints = min(int(npoints), N)
    with _on(dev):
        p = pos.to(torch.float32).contiguous()
        out = torch.empty((npoints,), dtype=torch.int64, device=dev)
        thin = torch.empty((npoints,), dtype=torch.int64, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        _check(lib.gsr_fps_thin(N, _ptr(p), npoints, int(start_idx), float(radius), int(thin_start_idx), _ptr(out), _ptr(thin), _ptr(cnt),
                                _stream(dev)), "gsr_fps_thin")
        m = int(cnt.item())
    return out, thin[:m]


def gnn_rel_inputs(rel_nodes: torch.Tensor, receivers: torch.Tensor, senders: torch.Tensor, attr_dim: int, group_dim: int) -> torch.Tensor:
    """gsr_gnn_rel_inputs: rel_nodes [N, attr + group + state] -> the relation encoder's input rows [E, 2 attr + 1 + state] in one launch."""
    lib = load_library()
    _require_device(rel_nodes)
    dev = rel_nodes.device
    E, S = int(receivers.shape[0]), int(rel_nodes.shape[1]) - attr_dim - group_dim
    with _on(dev):
        out = torch.empty((E, 2 * attr_dim + 1 + S), dtype=torch.float32, device=dev)
        if E:
            _check(lib.gsr_gnn_rel_inputs(E, int(attr_dim), int(group_dim), S, _ptr(rel_nodes), _ptr(receivers), _ptr(senders), _ptr(out), _stream(dev)),
                   "gsr_gnn_rel_inputs")
    return out


def gnn_aggregate(rel_part: torch.Tensor, node_parts: torch.Tensor, senders: torch.Tensor, row_start: torch.Tensor, n_sum_rows: int = None, res=None):
    """gsr_gnn_aggregate: rel_part [E, H], node_parts [N, 2 H], senders [E], row_start [N + 1] (int64; receivers ascending) -> agg [N, H];
    rows >= n_sum_rows (default N) get zeros.  res = (a, b), two contiguous float32 [N, H]: also returns a + b, written by the same launch
    (gsr_gnn_aggregate_res) -> (agg, a + b)."""
    lib = load_library()
    _require_device(rel_part)
    dev = rel_part.device
    N, H = int(node_parts.shape[0]), int(rel_part.shape[1])
    with _on(dev):
        agg = torch.empty((N, H), dtype=torch.float32, device=dev)
        if res is not None:
            ra, rb = res
            if not all(t.is_contiguous() and t.dtype == torch.float32 and tuple(t.shape) == (N, H) for t in (ra, rb)):
                raise ValueError("gnn_aggregate(res=...): two contiguous float32 [N, H] tensors, please")
            out = torch.empty((N, H), dtype=torch.float32, device=dev)
            _check(lib.gsr_gnn_aggregate_res(N, N if n_sum_rows is None else int(n_sum_rows), H, _ptr(rel_part), _ptr(node_parts), _ptr(senders), _ptr(row_start),
                                             _ptr(agg), _ptr(ra), _ptr(rb), _ptr(out), _stream(dev)), "gsr_gnn_aggregate_res")
            return agg, out
        _check(lib.gsr_gnn_aggregate(N, N if n_sum_rows is None else int(n_sum_rows), H, _ptr(rel_part), _ptr(node_parts), _ptr(senders), _ptr(row_start), _ptr(agg), _stream(dev)), "gsr_gnn_aggregate")
    return agg


def fit_bones(bones: torch.Tensor, motions: torch.Tensor, relations: torch.Tensor, out=None):
    """gsr_fit_bones: bones, motions [nb,3], relations [nb,nb] (int64 0/1; any row stride, unit column stride) on a HIP device ->
    (rotations [nb,3,3], unit quaternions [nb,4], code [nb] int32).  out = (rotations, quaternions): contiguous float32 tensors of 9 nb and
    4 nb elements that receive them (two blocks of a step's skinning packet: no concatenation afterwards)."""
    lib = load_library()
    _require_device(bones)
    dev = bones.device
    nb = int(bones.shape[0])
    with _on(dev):
        b = bones.to(torch.float32).contiguous()
        m = motions.to(device=dev, dtype=torch.float32).contiguous()
        rel = relations if (relations.dtype == torch.int64 and relations.dim() == 2 and (nb == 0 or relations.stride(1) == 1)) else relations.to(torch.int64).contiguous()
        if tuple(rel.shape) != (nb, nb):
            raise ValueError("fit_bones: relations must be [n_bones, n_bones]")
        if out is not None:
            R, q = out[0].view(nb, 3, 3), out[1].view(nb, 4)
            if not (R.is_contiguous() and q.is_contiguous() and R.dtype == torch.float32 and q.dtype == torch.float32 and R.device == dev and q.device == dev):
                raise ValueError("fit_bones(out=...): contiguous float32 tensors on the bones' device, please")
        else:
            R = torch.empty((nb, 3, 3), dtype=torch.float32, device=dev)
            q = torch.empty((nb, 4), dtype=torch.float32, device=dev)
        code = torch.empty((nb,), dtype=torch.int32, device=dev)
        _check(lib.gsr_fit_bones(nb, _ptr(b), _ptr(m), _ptr(rel), int(rel.stride(0)) if nb else 0, _ptr(R), _ptr(q), _ptr(code), _stream(dev)), "gsr_fit_bones")
    return R, q, code


def linear_blend_skinning(bones, rotations, translations, bone_quats, xyz, quat, n_valid=None, in_place=False, out=None):
    """gsr_lbs: returns (xyz_new [P,3], quat_new [P,4] or None, None) -- the [P, n_bones] weight matrix of the reference is
    never materialised.  in_place: xyz / quat (float32, contiguous) are overwritten and returned.  out = (xyz_out, quat_out):
    contiguous float32 tensors of the inputs' shapes that receive the result (a frame's slot of the episode arrays: no copy)."""
    lib = load_library()
    _require_device(xyz)
    dev = xyz.device
    P, nb = int(xyz.shape[0]), int(bones.shape[0])
    f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
    bones, rotations, translations, bone_quats = f(bones), f(rotations).reshape(nb, 9), f(translations), f(bone_quats)
    with _on(dev):
        if in_place:
            if not (xyz.is_contiguous() and xyz.dtype == torch.float32 and (quat is None or (quat.is_contiguous() and quat.dtype == torch.float32))):
                raise ValueError("linear_blend_skinning(in_place=True): contiguous float32 tensors, please")
            out_xyz, out_q = xyz, quat
        elif out is not None:
            out_xyz, out_q = out[0], (out[1] if quat is not None else None)
            for o, ref in ((out_xyz, xyz), (out_q, quat)):
                if o is not None and not (o.is_contiguous() and o.dtype == torch.float32 and o.device == dev and tuple(o.shape) == tuple(ref.shape)):
                    raise ValueError("linear_blend_skinning(out=...): contiguous float32 tensors of the inputs' shapes on their device, please")
        else:
            out_xyz = torch.empty((P, 3), dtype=torch.float32, device=dev)
            out_q = torch.empty((P, 4), dtype=torch.float32, device=dev) if quat is not None else None
        if n_valid is not None:      # only the first n_valid[0] bones (device int32) are real: fixed-shape callers
            _check(lib.gsr_lbs_valid(P, nb, _ptr(n_valid), _ptr(bones), _ptr(rotations), _ptr(translations), _ptr(bone_quats), _ptr(xyz), _ptr(quat),
                                     _ptr(out_xyz), _ptr(out_q), _stream(dev)), "gsr_lbs_valid")
        else:
            _check(lib.gsr_lbs(P, nb, _ptr(bones), _ptr(rotations), _ptr(translations), _ptr(bone_quats), _ptr(xyz), _ptr(quat),
                               _ptr(out_xyz), _ptr(out_q), _stream(dev)), "gsr_lbs")
    return out_xyz, out_q, None


def mark_visible(positions, viewmatrix) -> torch.Tensor:
    lib = load_library()
    _require_device(positions)
    dev = positions.device
    P = int(positions.shape[0])
    with _on(dev):
        pos = positions.to(dtype=torch.float32).contiguous()
        vm = _dev_f32(viewmatrix, dev, 16, "viewmatrix")
        present = torch.zeros((P,), dtype=torch.uint8, device=dev)
        _check(lib.gsr_mark_visible(_ptr(vm), P, _ptr(pos), _ptr(present), _stream(dev)), "gsr_mark_visible")
    return present.bool()


def final_transmittance(state: RasterState) -> torch.Tensor:
    """[H, W] float32 VIEW of the forward's per-pixel final transmittance T_final = prod (1 - alpha_i) over the blended entries
    (1 where nothing was blended): the first H * W floats of the image state (include/gsr.h: layout of ``image_state``).
    1 - T_final is the accumulated alpha, i.e. any channel of a render with colours = 1 on a black background.  Valid until the
    state is released or reused."""
    n = state.H * state.W
    return state.image[:4 * n].view(torch.float32).reshape(state.H, state.W)


def debug_views(state: RasterState):
    """Tensors copied out of the opaque state buffers (tests / benches only)."""
    lib = load_library()
    v = GsrDebugViews()
    _check(lib.gsr_debug_get_views(state.P, state.num_rendered, state.H, state.W, _ptr(state.geom),
                                   _ptr(state.binning), _ptr(state.image), C.byref(v)), "gsr_debug_get_views")
    P, D, H, W = state.P, state.num_rendered, state.H, state.W
    T = ((H + 15) // 16) * ((W + 15) // 16)

    def view(buf: torch.Tensor, addr, dtype, shape):
        n = 1
        for d in shape:
            n *= d
        if n == 0:
            return torch.empty(shape, dtype=dtype, device=buf.device)
        off = addr - buf.data_ptr()
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return buf[off:off + nbytes].view(dtype).reshape(shape).clone()

    i32 = torch.int32  # torch has no uint32 arithmetic; values here are < 2^31
    return dict(
        rec=view(state.geom, v.rec, torch.float32, (P, 16)), rect=view(state.geom, v.rect, i32, (P, 2)),
        tiles_touched=view(state.geom, v.tiles_touched, i32, (P,)), offsets=view(state.geom, v.offsets, i32, (P + 1,)),
        point_list=view(state.binning, v.point_list, i32, (D,)), ranges=view(state.image, v.ranges, i32, (T, 2)),
        final_T=view(state.image, v.final_T, torch.float32, (H, W)),
        n_contrib=view(state.image, v.n_contrib, i32, (H, W)))


def selftest(device=None) -> int:
    lib = load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with _on(dev):
        return int(lib.gsr_selftest(_stream(dev)))


def profile_begin():
    """Arm per-kernel HIP-event timing inside libgsr_hip.so (bench.py's roofline leg)."""
    _check(load_library().gsr_profile_begin(), "gsr_profile_begin")


def profile_end():
    """Returns {kernel name: (total_ms, launches)} for every kernel launched since profile_begin()."""
    lib = load_library()
    arr = (GsrKernelTime * 64)()
    n = C.c_int32(0)
    _check(lib.gsr_profile_end(arr, 64, C.byref(n)), "gsr_profile_end")
    return {arr[i].name.decode(): (float(arr[i].total_ms), int(arr[i].launches)) for i in range(n.value)}


def image_loss_forward(window11, pred, target):
    """Fused 0.8 L1 + 0.2 (1 - SSIM) building blocks for one image [C,H,W] or a batch [N,C,H,W]:
    returns (l1_sum, ssim_sum, fA, fC, fE); the sums are scalars, or [N] tensors for a batch (one reduction for all)."""
    lib = load_library()
    _require_device(pred)
    dev = pred.device
    batched = pred.dim() == 4
    N = int(pred.shape[0]) if batched else 1
    Cc, H, W = (int(d) for d in pred.shape[-3:])
    win = (C.c_float * 11)(*[float(v) for v in window11])
    with _on(dev):
        nb = int(lib.gsr_image_loss_blocks(N * Cc, H, W))
        f32 = dict(dtype=torch.float32, device=dev)
        fA, fC, fE = (torch.empty(tuple(pred.shape), **f32) for _ in range(3))
        part = torch.empty((2, nb), **f32)
        _check(lib.gsr_image_loss_forward(win, N * Cc, H, W, _ptr(pred), _ptr(target), _ptr(fA), _ptr(fC), _ptr(fE), _ptr(part[0]),
                                          _ptr(part[1]), _stream(dev)), "gsr_image_loss_forward")
    sums = part.view(2, N, nb // N).sum(2)          # block partials are channel-major: contiguous per image
    return (sums[0], sums[1], fA, fC, fE) if batched else (sums[0, 0], sums[1, 0], fA, fC, fE)


_win_cache = {}


def _window(window11):
    key = tuple(window11)
    w = _win_cache.get(key)
    if w is None:
        w = _win_cache[key] = (C.c_float * 11)(*[float(v) for v in window11])
    return w


_target_moments = {}     # id(target as the caller passed it) -> [weakref, version, converted fp32 image | None, moments | None, bytes]
_TARGET_CACHE_BYTES = 1 << 30   # converted targets + moments kept alive at most (cleared when exceeded)
_TARGET_CACHE_ENTRIES = 256     # and at most this many records (a loader that hands over a fresh target every step)


def _target_entry(t):
    """Cache record of a target image, keyed on the tensor object the CALLER holds (id + version, weakref-checked): the
    reference's loader hands over ``permute(2, 0, 1) / 255`` views, whose contiguous copy would otherwise be a fresh temporary
    (and a fresh address) every step.  Returns [ref, version, converted image or None, moments or None, bytes].  A target that
    already is contiguous fp32 is NOT referenced from here (slot 2 stays None: the record must not keep its own key alive)."""
    e = _target_moments.get(id(t))
    if e is not None and e[0]() is t and e[1] == t._version:
        return e
    direct = t.is_contiguous() and t.dtype == torch.float32
    conv = None if direct else t.contiguous().float()
    nbytes = 0 if direct else conv.numel() * 4
    for k in [k for k, x in _target_moments.items() if x[0]() is None]:   # dead targets: drop their images
        del _target_moments[k]
    if len(_target_moments) >= _TARGET_CACHE_ENTRIES or sum(x[4] for x in _target_moments.values()) + nbytes > _TARGET_CACHE_BYTES:
        _target_moments.clear()
    e = _target_moments[id(t)] = [weakref.ref(t), t._version, conv, None, nbytes]
    return e


def _moments_of(win, t):
    """Window moments blur(y), blur(y*y) of a target, computed the SECOND time the same target (object and version) is seen
    (a target that changes every step would pay for a kernel it never profits from).  Returns (image, moments or None)."""
    known = id(t) in _target_moments and _target_moments[id(t)][0]() is t and _target_moments[id(t)][1] == t._version
    e = _target_entry(t)
    img = t if e[2] is None else e[2]
    if known and e[3] is None:
        lib = load_library()
        Cc, H, W = (int(d) for d in img.shape)
        m = torch.empty((2, Cc, H, W), dtype=torch.float32, device=img.device)
        _check(lib.gsr_target_moments(win, Cc, H, W, _ptr(img), _ptr(m), _stream(img.device)), "gsr_target_moments")
        e[3] = m
        e[4] += m.numel() * 4
    return img, e[3]


def _loss_table(targets, cam_rows, weights, channels, moments=None):
    tab = GsrLossViews()
    tab.n_images, tab.channels = len(targets), channels
    for i, (t, r, w) in enumerate(zip(targets, cam_rows, weights)):
        tab.cam_row[i], tab.weight[i], tab.target[i] = int(r), float(w), t.data_ptr()
        tab.target_moments[i] = moments[i].data_ptr() if moments is not None else None
    return tab


def views_loss_forward(window11, renders, targets, cam_rows, weights, cam_m, cam_c, w_l1, w_ssim):
    """Image terms of all renders of a step (gsr_views_loss_forward): ``renders`` [n,C,H,W] (the rasterizer's output batch),
    ``targets`` n tensors [C,H,W], ``cam_rows`` n ints (row of cam_m / cam_c, or -1), ``weights`` n floats.
    Returns (losses[n+1] with the weighted total last, state for the backward)."""
    lib = load_library()
    _require_device(renders)
    dev = renders.device
    n, Cc, H, W = (int(d) for d in renders.shape)
    if n > LOSS_MAX_IMAGES or len(targets) != n or len(cam_rows) != n or len(weights) != n:
        raise RuntimeError(f"views_loss_forward: 1..{LOSS_MAX_IMAGES} images with one target / camera row / weight each")
    if not renders.is_contiguous() or renders.dtype != torch.float32:
        raise RuntimeError("views_loss_forward: renders must be a contiguous float32 batch")
    for t in targets:
        if tuple(t.shape) != (Cc, H, W) or t.device != dev:
            raise RuntimeError("views_loss_forward: every target must be [C,H,W] on the renders' device")
    n_rows = 0 if cam_m is None else int(cam_m.shape[0])
    for r in cam_rows:     # the kernel indexes cam_m / cam_c with these rows: an id past the table would read foreign memory
        if int(r) >= n_rows or (int(r) >= 0 and (cam_m is None or cam_c is None)):
            raise RuntimeError(f"views_loss_forward: camera row {int(r)} outside cam_m / cam_c with {n_rows} rows")
    if cam_m is not None and (cam_c is None or tuple(cam_c.shape) != tuple(cam_m.shape)):
        raise RuntimeError("views_loss_forward: cam_m and cam_c must have the same shape")
    win = _window(window11)
    with _on(dev):
        pairs = [_moments_of(win, t) for t in targets]
    targets = [p[0] for p in pairs]
    moms = [p[1] for p in pairs]
    moms = moms if all(m is not None for m in moms) else None
    tab = _loss_table(targets, cam_rows, weights, Cc, moms)
    with _on(dev):
        nb = int(lib.gsr_views_loss_blocks(n, Cc, H, W))
        f32 = dict(dtype=torch.float32, device=dev)
        maps = torch.empty((3,) + tuple(renders.shape), **f32)
        part = torch.empty((2, nb), **f32)
        losses = torch.empty((n + 1,), **f32)
        _check(lib.gsr_views_loss_forward(win, C.byref(tab), H, W, _ptr(renders), _ptr(cam_m), _ptr(cam_c), float(w_l1), float(w_ssim),
                                          _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]), _ptr(part), _ptr(losses), _stream(dev)),
               "gsr_views_loss_forward")
    return losses, (win, tab, (targets, moms), maps, part)


def views_loss_backward(state, renders, cam_m, cam_c, grad_total, w_l1, w_ssim, want_cam_grads=True):
    """Returns (d_renders like renders, d_cam_m, d_cam_c like cam_m / None)."""
    lib = load_library()
    win, tab, _targets, maps, part = state
    dev = renders.device
    _n, _Cc, H, W = (int(d) for d in renders.shape)
    with _on(dev):
        g = grad_total.to(dtype=torch.float32, device=dev).reshape(1)
        d_renders = torch.empty_like(renders)
        d_m = d_c = None
        n_cams = 0
        if want_cam_grads and cam_m is not None:
            n_cams = int(cam_m.shape[0])
            d_m, d_c = torch.empty_like(cam_m), torch.empty_like(cam_c)
        _check(lib.gsr_views_loss_backward(win, C.byref(tab), H, W, _ptr(renders), _ptr(cam_m), _ptr(cam_c), n_cams, _ptr(maps[0]),
                                           _ptr(maps[1]), _ptr(maps[2]), _ptr(g), float(w_l1), float(w_ssim), _ptr(d_renders), _ptr(part),
                                           _ptr(d_m), _ptr(d_c), _stream(dev)), "gsr_views_loss_backward")
    return d_renders, d_m, d_c


def image_loss_backward(window11, pred, target, fA, fC, fE, grad_loss, w_l1, w_ssim):
    """``grad_loss``: scalar for one image, [N] for a batch [N,C,H,W]."""
    lib = load_library()
    dev = pred.device
    N = int(pred.shape[0]) if pred.dim() == 4 else 1
    Cc, H, W = (int(d) for d in pred.shape[-3:])
    win = (C.c_float * 11)(*[float(v) for v in window11])
    with _on(dev):
        g = grad_loss.to(dtype=torch.float32, device=dev).reshape(N).contiguous()
        d_pred = torch.empty_like(pred)
        _check(lib.gsr_image_loss_backward(win, N * Cc, H, W, _ptr(pred), _ptr(target), _ptr(fA), _ptr(fC), _ptr(fE), _ptr(g), Cc,
                                           float(w_l1), float(w_ssim), _ptr(d_pred), _stream(dev)), "gsr_image_loss_backward")
    return d_pred
