"""``diff_gaussian_rasterization`` -- MI355X-native drop-in for the rasterizer package gs-dynamics imports.

Same import surface the reference uses:

    from diff_gaussian_rasterization import GaussianRasterizer                          # render/renderer.py:3
    from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera     # tracking/helpers.py:5
    im, radius, depth = GaussianRasterizer(raster_settings=cam)(**rendervar)            # tracking/train_utils.py:178

(paths relative to /root/reference/src).  The compute path is ``libgsr_hip.so`` (hand-written gfx950
kernels behind the C-ABI of ``include/gsr.h``); this module is the thin host mirror: argument checks,
allocation through torch's caching allocator, and the autograd glue.  There is no CPU implementation.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import nn

import os

from . import _hip

# The torch C++ layer (csrc/gsr_torch.cpp -> _C.so, built by __graft_entry__.build()): upstream's three entry points, one native call
# per forward / backward.  Absent (not built) or switched off (GSR_NO_TORCH_EXT=1, or GSR_HIP_LIB pointing at another library
# build, which _C is not linked against): the ctypes binding in _hip.py does the same work call by call.
_C = None
if os.environ.get("GSR_NO_TORCH_EXT") != "1" and "GSR_HIP_LIB" not in os.environ:
    try:
        import importlib
        _C = importlib.import_module(__name__ + "._C")
    except ImportError:
        _C = None
_CTYPES_FORWARD, _CTYPES_BACKWARD = _hip.rasterize_forward, _hip.rasterize_backward
_PY_NODE = os.environ.get("GSR_PY_AUTOGRAD") == "1"   # A/B: the autograd node as a Python torch.autograd.Function (rounds 1 - 4) instead of _C.rasterize


_LAYER_STATES = {}     # device index -> _C.LayerState: what the torch C++ layer remembers between GaussianRasterizer calls on that device


def layer_state(device=None):
    """The torch C++ layer's state for ``device`` (default: the current HIP device), created on first use: the last forward's tile lists
    (the reference renders every camera twice with the same geometry), the entry capacities per (P, H, W) of the capacity-mode forward and
    the twin predictor.  Owned HERE, not by the extension (no global state in the library or its torch layer):
    ``layer_state().stats()`` inspects it, ``.reset()`` forgets everything learned, ``.list_reuse`` / ``.capacity_mode`` are the switches,
    ``reset_layer_states()`` drops every device's.  None when the C++ layer is not loaded."""
    if _C is None:
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = torch.cuda.current_device() if dev.index is None else dev.index
    st = _LAYER_STATES.get(idx)
    if st is None:
        st = _LAYER_STATES[idx] = _C.LayerState()
    return st


def reset_layer_states():
    _LAYER_STATES.clear()


def _native():
    """The torch C++ layer, unless a test double / spy has replaced the ctypes entry points (then those must be the ones called)."""
    if _C is not None and _hip.rasterize_forward is _CTYPES_FORWARD and _hip.rasterize_backward is _CTYPES_BACKWARD:
        return _C
    return None


__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_views", "layer_state",
           "reset_layer_states"]


class GaussianRasterizationSettings(NamedTuple):
    """Per-view camera record; the 11 fields (names and order) the reference constructs at
    /root/reference/src/tracking/helpers.py:20-32."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool


def _prep(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """None for absent/empty inputs, else a contiguous fp32 tensor (no copy when already so)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _RasterizeGaussians(torch.autograd.Function):
    """forward -> (color, radii, depth); backward -> grads for the 8 tensor inputs, None for settings."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        ctx.set_materialize_grads(False)  # grad_depth is ignored: do not let autograd fill a zero image for it
        m3 = _prep(means3D)
        if m3 is None:
            if means3D is not None and means3D.dim() == 2 and means3D.shape[1] == 3:
                # P = 0: zero-filled outputs without launching anything
                dev = means3D.device
                H, W = int(raster_settings.image_height), int(raster_settings.image_width)
                ctx.empty = True
                return (torch.zeros((3, H, W), device=dev), torch.zeros((0,), dtype=torch.int32, device=dev),
                        torch.zeros((1, H, W), device=dev))
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        if m3.dim() != 2 or m3.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        ctx.empty = False
        native = _native() if m3.is_cuda else None
        if native is not None:
            rs = raster_settings
            e = m3.new_empty(0)
            t = lambda x: e if (x is None or x.numel() == 0) else x   # noqa: E731
            col_, sh_, sc_, rot_, cov_ = t(colors_precomp), t(sh), t(scales), t(rotations), t(cov3Ds_precomp)
            D, color, depth, radii, geom, binning, image = native.rasterize_gaussians(
                rs.bg, m3, col_, opacities, sc_, rot_, float(rs.scale_modifier), cov_, rs.viewmatrix, rs.projmatrix, float(rs.tanfovx),
                float(rs.tanfovy), int(rs.image_height), int(rs.image_width), sh_, int(rs.sh_degree), rs.campos, bool(rs.prefiltered),
                bool(any(ctx.needs_input_grad)), layer_state(m3.device))       # (only reached through the Python node: test doubles; under torch.no_grad() this still says True -- the C++ node decides before it is built)
            ctx.native, ctx.rs, ctx.num_rendered = native, rs, int(D)
            ctx.has = (sh_.numel() > 0, col_.numel() > 0, sc_.numel() > 0, cov_.numel() > 0)
            ctx.save_for_backward(m3, radii, col_, sh_, sc_, rot_, cov_, geom, binning, image)
            ctx.mark_non_differentiable(radii)
            return color, radii, depth
        ctx.native = None
        sh_, col_, op_ = _prep(sh), _prep(colors_precomp), _prep(opacities)
        sc_, rot_, cov_ = _prep(scales), _prep(rotations), _prep(cov3Ds_precomp)
        color, radii, depth, state = _hip.rasterize_forward(raster_settings, m3, op_, col_, sh_, sc_, rot_, cov_)
        ctx.state = state
        ctx.has = (sh_ is not None, col_ is not None, sc_ is not None, cov_ is not None)
        empty = m3.new_empty(0)
        ctx.save_for_backward(m3, radii, col_ if col_ is not None else empty, sh_ if sh_ is not None else empty,
                              sc_ if sc_ is not None else empty, rot_ if rot_ is not None else empty,
                              cov_ if cov_ is not None else empty)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth):  # grad_radii / grad_depth: accepted, ignored
        if ctx.empty:
            return (None,) * 9
        if ctx.native is not None:
            m3, radii, col_, sh_, sc_, rot_, cov_, geom, binning, image = ctx.saved_tensors
            has_sh, has_col, has_sc, has_cov = ctx.has
            rs = ctx.rs
            if grad_color is None:
                grad_color = torch.zeros((3, int(rs.image_height), int(rs.image_width)), device=m3.device)
            d2, dc, do, d3, dcov, dsh, ds, dr = ctx.native.rasterize_gaussians_backward(
                rs.bg, m3, radii, col_, sc_, rot_, float(rs.scale_modifier), cov_, rs.viewmatrix, rs.projmatrix, float(rs.tanfovx),
                float(rs.tanfovy), grad_color, sh_, int(rs.sh_degree), rs.campos, geom, ctx.num_rendered, binning, image,
                bool(has_col and ctx.needs_input_grad[3]))   # frozen colours (the reference's training): the six-sum backward
            return (d3, d2, dsh if has_sh else None, dc if (has_col and ctx.needs_input_grad[3]) else None, do, ds if has_sc else None, dr if has_sc else None,
                    dcov if has_cov else None, None)
        m3, radii, col_, sh_, sc_, rot_, cov_ = ctx.saved_tensors
        has_sh, has_col, has_sc, has_cov = ctx.has
        if grad_color is None:
            grad_color = torch.zeros((3, ctx.state.H, ctx.state.W), device=m3.device)
        d_means3D, d_means2D, d_colors, d_opacity, d_scales, d_rot, d_cov, d_sh = _hip.rasterize_backward(
            ctx.state, grad_color, m3, radii, col_ if has_col else None, sh_ if has_sh else None,
            sc_ if has_sc else None, rot_ if has_sc else None, cov_ if has_cov else None,
            want_color_grad=bool(has_col and ctx.needs_input_grad[3]))
        return (d_means3D, d_means2D, d_sh if has_sh else None, d_colors if has_col else None, d_opacity,
                d_scales if has_sc else None, d_rot if has_sc else None, d_cov if has_cov else None, None)


class _RasterizeGaussiansViews(torch.autograd.Function):
    """V views of the same Gaussians in one call (extension of the reference API, which renders one view
    per call): forward -> (color[V,3,H,W], radii[V,P], depth[V,1,H,W]); backward sums the per-view input
    gradients.  ``means2D`` is a [V,P,3] holder so each view's screen-space gradient stays separate
    (densification accumulates their norms per view, /root/reference/src/tracking/external.py:138-142)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings_list):
        ctx.set_materialize_grads(False)
        m3 = _prep(means3D)
        if m3 is None or m3.dim() != 2 or m3.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        sh_, col_, op_ = _prep(sh), _prep(colors_precomp), _prep(opacities)
        sc_, rot_, cov_ = _prep(scales), _prep(rotations), _prep(cov3Ds_precomp)
        wants_grad = any(ctx.needs_input_grad)
        color, radii, depth, states = _hip.rasterize_forward_batch(list(settings_list), m3, op_, col_, sh_, sc_, rot_, cov_,
                                                                   prepare_backward=wants_grad, **({} if wants_grad else {"forward_only": True}))
        ctx.states = states
        ctx.has = (sh_ is not None, col_ is not None, sc_ is not None, cov_ is not None)
        empty = m3.new_empty(0)
        ctx.save_for_backward(m3, radii, col_ if col_ is not None else empty, sh_ if sh_ is not None else empty,
                              sc_ if sc_ is not None else empty, rot_ if rot_ is not None else empty,
                              cov_ if cov_ is not None else empty)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth):
        m3, radii, col_, sh_, sc_, rot_, cov_ = ctx.saved_tensors
        has_sh, has_col, has_sc, has_cov = ctx.has
        V = len(ctx.states)
        if grad_color is None:
            grad_color = torch.zeros((V, 3, ctx.states[0].H, ctx.states[0].W), device=m3.device)
        d3, d2, dc, do, ds, dr, dcov, dsh = _hip.rasterize_backward_batch(
            ctx.states, grad_color, m3, radii, col_ if has_col else None, sh_ if has_sh else None,
            sc_ if has_sc else None, rot_ if has_sc else None, cov_ if has_cov else None,
            want_color_grad=bool(has_col and ctx.needs_input_grad[3]))
        # gradients arrive already summed over views (means2D stays per view); the state stays on ctx so that a
        # second backward (retain_graph=True) works, and is released with the graph
        return (d3, d2, dsh if has_sh else None, dc if has_col else None, do, ds if has_sc else None,
                dr if has_sc else None, dcov if has_cov else None, None)


def rasterize_gaussians_views(settings_list, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                              rotations=None, cov3D_precomp=None):
    """Render ``len(settings_list)`` views of one set of Gaussians.  ``means2D``: [V,P,3] gradient holder."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    empty = torch.Tensor([])
    settings_list = tuple(settings_list)
    V = len(settings_list)
    if V == 0:
        raise ValueError("rasterize_gaussians_views: no views")
    per_view_col = colors_precomp is not None and colors_precomp.dim() == 3

    def call(lo, hi):
        whole = lo == 0 and hi == V          # no slice nodes in the graph (their backward is a zero-fill + copy each)
        return _RasterizeGaussiansViews.apply(
            means3D, means2D if whole else means2D[lo:hi], empty if shs is None else shs,
            empty if colors_precomp is None else (colors_precomp[lo:hi] if (per_view_col and not whole) else colors_precomp), opacities,
            empty if scales is None else scales, empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp, settings_list[lo:hi])
    if V <= _hip.MAX_BATCH:
        return call(0, V)
    # more views than one library call takes: several calls, outputs concatenated (autograd sums the shared inputs)
    parts = [call(lo, min(V, lo + _hip.MAX_BATCH)) for lo in range(0, V, _hip.MAX_BATCH)]
    return tuple(torch.cat([p[k] for p in parts]) for k in range(3))


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    native = _native() if (means3D is not None and means3D.is_cuda) else None
    if native is not None and hasattr(native, "rasterize") and not _PY_NODE:
        # one crossing into the torch C++ layer: forward and the autograd node live there (csrc/gsr_torch.cpp: RasterizeFn)
        rs = raster_settings
        return native.rasterize(layer_state(means3D.device), means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg, rs.viewmatrix,
                                rs.projmatrix, rs.campos, float(rs.tanfovx), float(rs.tanfovy), int(rs.image_height), int(rs.image_width),
                                float(rs.scale_modifier), int(rs.sh_degree), bool(rs.prefiltered))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    """Constructed per call by the reference (``Renderer(raster_settings=cam)(**rendervar)``,
    /root/reference/src/tracking/train_utils.py:178): construction does no work."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            native = _native() if positions.is_cuda else None
            if native is not None:
                return native.mark_visible(positions, self.raster_settings.viewmatrix, self.raster_settings.projmatrix)
            return _hip.mark_visible(positions, self.raster_settings.viewmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)
