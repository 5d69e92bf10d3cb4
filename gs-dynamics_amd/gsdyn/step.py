"""The per-iteration tracking step: activation, two renders, losses -- the caller side of the hot path.

Mirrors, call for call, /root/reference/src/tracking/train_utils.py:167-246 (``get_loss``) and
/root/reference/src/tracking/helpers.py:36-45 (``params2rendervar``), with the shipped defects of the
reference's driver worked around as SURVEY.md Appendix C records (weights passed explicitly, views
sampled uniformly with replacement).  Every render goes through ``GaussianRasterizer``.
"""
from __future__ import annotations

from dataclasses import dataclass

import os

import torch

from diff_gaussian_rasterization import GaussianRasterizer as Renderer

from .losses import (activate, build_rotation, calc_psnr, image_loss, l1_loss_v2, quat_mult, reverse_adjacency, shared_terms,
                     views_image_loss, weighted_l2_loss_v1, weighted_l2_loss_v2)


@dataclass
class LossWeights:
    """Defaults of /root/reference/src/tracking/train_gs.py:54-61."""
    soft_col_cons: float = 0.01
    im: float = 50.0
    seg: float = 200.0
    rigid: float = 200.0
    bg: float = 200.0
    iso: float = 1000.0
    rot: float = 4.0
    floor: float = 2.0


def params2rendervar(params, colors_key: str = "rgb_colors"):
    return {
        "means3D": params["means3D"],
        "colors_precomp": params[colors_key],
        "rotations": torch.nn.functional.normalize(params["unnorm_rotations"]),
        "opacities": torch.sigmoid(params["logit_opacities"]),
        "scales": torch.exp(params["log_scales"]),
        "means2D": torch.zeros_like(params["means3D"], requires_grad=True) + 0,
    }


def initialize_optimizer(params, scene_radius: float):
    """Adam, one group per parameter, lrs of /root/reference/src/tracking/train_utils.py:152-164."""
    lrs = {"means3D": 0.00016 * scene_radius, "rgb_colors": 0.0, "seg_colors": 0.0, "unnorm_rotations": 0.001,
           "logit_opacities": 0.05, "log_scales": 0.001, "cam_m": 1e-4, "cam_c": 1e-4}
    groups = [{"params": [v], "name": k, "lr": lrs[k]} for k, v in params.items()]
    if all(v.is_cuda for v in params.values()):      # same update, all groups in one kernel launch (gsdyn/optim.py)
        from .optim import FusedAdam
        return FusedAdam(groups, lr=0.0, eps=1e-15)
    return torch.optim.Adam(groups, lr=0.0, eps=1e-15)


def _image_term(pred, target):
    return image_loss(pred, target, 0.8, 0.2)


def get_loss(params, curr_data, variables, is_initial_timestep: bool, w: LossWeights):
    """Returns (loss, variables).  ``curr_data``: dict(cam=settings, im=[3,H,W], seg=[3,H,W], id=int)."""
    losses = {}
    rendervar = params2rendervar(params)
    rendervar["means2D"].retain_grad()
    im, radius, _ = Renderer(raster_settings=curr_data["cam"])(**rendervar)
    cid = curr_data["id"]
    im = torch.exp(params["cam_m"][cid])[:, None, None] * im + params["cam_c"][cid][:, None, None]
    losses["im"] = _image_term(im, curr_data["im"])
    variables["means2D"] = rendervar["means2D"]  # densification reads the colour render's gradient only

    segrendervar = params2rendervar(params, colors_key="seg_colors")
    seg, _, _ = Renderer(raster_settings=curr_data["cam"])(**segrendervar)
    losses["seg"] = _image_term(seg, curr_data["seg"])

    weights = {"im": w.im, "seg": w.seg, "rigid": w.rigid, "iso": w.iso, "rot": w.rot, "floor": w.floor, "bg": w.bg,
               "soft_col_cons": w.soft_col_cons}
    loss = sum(weights[k] * v for k, v in losses.items())
    if not is_initial_timestep:
        # R^T applied to every neighbour offset: the reference writes this as a batched 3x3 @ 3x1 matmul
        # (train_utils.py:207), which on ROCm dispatches ~1.4 M tiny GEMMs (18 + 12 + 11 ms fwd+bwd at 70 k foreground
        # points x 20 neighbours); _shared_terms uses the fused kernels on a HIP device, a broadcast multiply + sum otherwise.
        shared, _ = _shared_terms(params, rendervar, variables, weights)
        loss = loss + shared
    # Same values as the reference's `max_2D_radius[seen] = max(radius[seen], max_2D_radius[seen])`
    # (train_utils.py:243-245) without boolean-mask indexing, which costs a device->host sync per call.
    seen = radius > 0
    m2r = variables["max_2D_radius"]
    variables["max_2D_radius"] = torch.where(seen, torch.maximum(radius.to(m2r.dtype), m2r), m2r)
    variables["seen"] = seen
    return loss, variables


_SHARED_NAMES = ("rigid", "rot", "iso", "floor", "bg")


def _shared_terms(params, rendervar, variables, weights, scale: float = 1.0):
    """View-independent terms of the t > 0 loss (/root/reference/src/tracking/train_utils.py:198-232).
    Returns (scale * sum_k weights[k] * term_k, the five terms rigid / rot / iso / floor / bg as a detached [5] tensor)."""
    if rendervar["means3D"].is_cuda and all(k in variables for k in ("fg_idx", "bg_idx", "rev_ptr", "rev_edge")):
        # all of it in 3 + 3 fused kernels (gsr_rigidity.hip, gsr_step.hip) instead of ~100 + ~100 torch kernels
        return shared_terms(rendervar["means3D"], rendervar["rotations"], variables, [scale * weights[k] for k in _SHARED_NAMES])
    losses = {}
    if "fg_idx" in variables:     # index tensors prepared once per timestep: boolean-mask indexing costs a host sync per call
        fg_idx, bg_idx = variables["fg_idx"], variables["bg_idx"]
        fg_pts = rendervar["means3D"].index_select(0, fg_idx)
        pick_fg = lambda t: t.index_select(0, fg_idx)   # noqa: E731
        pick_bg = lambda t: t.index_select(0, bg_idx)   # noqa: E731
    else:
        is_fg = (params["seg_colors"][:, 0] > 0.5).detach()
        fg_pts = rendervar["means3D"][is_fg]
        pick_fg = lambda t: t[is_fg]                    # noqa: E731
        pick_bg = lambda t: t[~is_fg]                   # noqa: E731
    fg_rot = pick_fg(rendervar["rotations"])
    rel_rot = quat_mult(fg_rot, variables["prev_inv_rot_fg"])
    rot = build_rotation(rel_rot)
    nbr = variables["neighbor_indices"]
    curr_offset = fg_pts[nbr] - fg_pts[:, None]
    offset_prev_frame = (curr_offset[:, :, :, None] * rot[:, None, :, :]).sum(2)   # see get_loss
    nw = variables["neighbor_weight"]
    losses["rigid"] = weighted_l2_loss_v2(offset_prev_frame, variables["prev_offset"], nw)
    losses["rot"] = weighted_l2_loss_v2(rel_rot[nbr], rel_rot[:, None], nw)
    offset_mag = torch.sqrt((curr_offset ** 2).sum(-1) + 1e-20)
    losses["iso"] = weighted_l2_loss_v1(offset_mag, variables["neighbor_dist"], nw)
    losses["floor"] = torch.clamp(fg_pts[:, 1], min=0).mean()
    bg_pts = pick_bg(rendervar["means3D"])
    bg_rot = pick_bg(rendervar["rotations"])
    losses["bg"] = l1_loss_v2(bg_pts, variables["init_bg_pts"]) + l1_loss_v2(bg_rot, variables["init_bg_rot"])
    total = scale * sum(weights[k] * losses[k] for k in _SHARED_NAMES)
    return total, torch.stack([losses[k].detach() for k in _SHARED_NAMES])


def params2rendervar_fused(params, colors_key: str = "rgb_colors"):
    """``params2rendervar`` with the three activations as one fused kernel each way (same values; gsr_step.hip)."""
    rot, op, sc = activate(params["unnorm_rotations"], params["logit_opacities"], params["log_scales"])
    return {"means3D": params["means3D"], "colors_precomp": params[colors_key], "rotations": rot, "opacities": op, "scales": sc}


def _view_colours(params, variables, V: int, frozen: bool):
    """[rgb, seg] x V as one [2V,P,3] array; rebuilt only when a colour tensor changed or needs a gradient."""
    rgb, seg = params["rgb_colors"], params["seg_colors"]
    if not frozen and (rgb.requires_grad or seg.requires_grad):
        return torch.stack([rgb, seg]).repeat(V, 1, 1)
    rgb, seg = rgb.detach(), seg.detach()
    key = (rgb.data_ptr(), rgb._version, seg.data_ptr(), seg._version, V, tuple(rgb.shape))
    hit = variables.get("_view_colours")
    if hit is None or hit[0] != key:
        hit = (key, torch.stack([rgb, seg]).repeat(V, 1, 1))
        variables["_view_colours"] = hit
    return hit[1]


def _radius_bookkeeping(variables, rad):
    """``max_2D_radius`` / ``seen`` of /root/reference/src/tracking/train_utils.py:243-245 for the colour renders ``rad`` [V,P].
    Their only reader is the density control, which runs in the first timestep only (train_gs.py:31-37), so the direct step
    (``loss_and_grads_views``) skips these five small kernels at t > 0; ``get_loss`` / ``get_loss_views`` keep the reference's behaviour."""
    m2r = variables["max_2D_radius"]
    variables["max_2D_radius"] = torch.maximum(m2r, rad.max(0).values.to(m2r.dtype))
    variables["seen"] = (rad > 0).any(0)


def get_loss_views(params, datas, variables, is_initial_timestep: bool, w: LossWeights, frozen_colours: bool = False):
    """``get_loss`` for several cameras at once, equal to the SUM of the per-camera ``get_loss`` values, with ONE
    rasterizer call: the colour and the segmentation render of every camera (2 V renders) share the Gaussians'
    geometry and differ only in their colour array, so they go through ``rasterize_gaussians_views`` as 2 V views
    with per-view colours -- one kernel launch per stage for all of them.  Even the
    reference's own pattern (one camera per iteration, /root/reference/src/tracking/train_gs.py:25-39) becomes a 2-view batch.
    ``frozen_colours``: no gradient for rgb_colors / seg_colors.  The reference leaves ``seg_colors.requires_grad`` on but
    gives both colour groups lr 0 (/root/reference/src/tracking/train_utils.py:133,152-164), i.e. computes that gradient and
    never applies it; without it the backward of a colour + segmentation pair stays ONE replay of the tile lists
    (fused pair, gsr_render.hip) -- callers that own the optimiser (``ViewShardedStep``) switch it on when both lrs are 0.
    Returns (loss, variables, aux) with aux = dict(means2D=[2V,P,3] gradient holder (rows 0, 2, ... = colour renders),
    radii=[V,P] of the colour renders)."""
    from diff_gaussian_rasterization import rasterize_gaussians_views
    V = len(datas)
    rendervar = params2rendervar_fused(params)
    P = rendervar["means3D"].shape[0]
    cams = [d["cam"] for d in datas for _ in (0, 1)]
    colours = _view_colours(params, variables, V, frozen_colours)                                               # [2V,P,3]
    m2 = torch.zeros((2 * V, P, 3), device=rendervar["means3D"].device, requires_grad=True)
    ims, radii, _ = rasterize_gaussians_views(cams, rendervar["means3D"], m2, rendervar["opacities"], colors_precomp=colours,
                                              scales=rendervar["scales"], rotations=rendervar["rotations"])
    # both image terms of all cameras, camera affine included, as ONE fused loss evaluation on the render batch
    ids = [int(d["id"]) for d in datas]
    targets = [t for d in datas for t in (d["im"], d["seg"])]
    rows = [r for i in ids for r in (i, -1)]
    total, _ = views_image_loss(ims, targets, rows, [w.im, w.seg] * V, params["cam_m"], params["cam_c"], 0.8, 0.2)
    if not is_initial_timestep:
        weights = {"rigid": w.rigid, "iso": w.iso, "rot": w.rot, "floor": w.floor, "bg": w.bg}
        shared, _ = _shared_terms(params, rendervar, variables, weights, scale=float(V))   # every per-camera get_loss adds them once
        total = total + shared
    rad = radii[0::2]                                                             # colour renders
    _radius_bookkeeping(variables, rad)
    variables["means2D"] = m2
    return total, variables, dict(means2D=m2, radii=rad)


_ONES = {}


def _acc_grad(p, g):
    if g is None:
        return
    if p.grad is None:
        p.grad = g
    else:
        p.grad.add_(g)


@torch.no_grad()
def loss_and_grads_views(params, datas, variables, is_initial_timestep: bool, w: LossWeights, _no_host_sync: bool = True, grad_out=None):
    """``get_loss_views(..., frozen_colours=True)`` followed by ``loss.backward()``, as a straight sequence of library calls
    (activations, rasterizer, image terms, shared terms and their backward passes in reverse) without the autograd engine:
    same kernels, same values, about a third of the host time -- which is what bounds the reference's own loop shape, one camera
    per iteration (/root/reference/src/tracking/train_gs.py:25-39: ~0.3 ms of GPU work per iteration).  The gradients are ADDED
    to the ``.grad`` of means3D / unnorm_rotations / logit_opacities / log_scales / cam_m / cam_c (colours are frozen: lr 0 in
    the tracking schedule).  Needs a HIP device, at most 8 cameras per call, and at t > 0 the tensors of ``make_rigidity_variables``.
    ``grad_out`` (``GradBucket.views()``): the gradients of means3D / unnorm_rotations / logit_opacities / log_scales are written into
    these caller-owned tensors (the all-reduce bucket's slices) when the parameter has no ``.grad`` yet -- nothing to pack afterwards.
    Returns (loss, variables, aux) with aux = dict(means2D_grad=[2V,P,3] (rows 0, 2, ... = colour renders), radii=[V,P])."""
    from diff_gaussian_rasterization import _hip
    from .losses import _SHARED_KEYS, _window_1d
    go = {} if grad_out is None else {k: grad_out[k] for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales")
                                       if k in grad_out and params[k].grad is None and params[k].requires_grad}
    V = len(datas)
    m3 = params["means3D"]
    dev = m3.device
    if not m3.is_cuda or 2 * V > _hip.MAX_BATCH:
        raise RuntimeError("loss_and_grads_views: needs a HIP device and at most %d cameras per call" % (_hip.MAX_BATCH // 2))
    with _hip.hold_stream(dev):        # one stream lookup for the ten library calls below
        rot, op, sc = _hip.activate_forward(params["unnorm_rotations"], params["logit_opacities"], params["log_scales"])
        shared = terms = work = w5 = None
        if not is_initial_timestep:   # enqueued FIRST: the GPU works through it while the host prepares the rasterizer call
            shared = {k: (variables[k] if variables[k].is_contiguous() else variables[k].contiguous()) for k in _SHARED_KEYS}
            w5 = [float(V) * x for x in (w.rigid, w.rot, w.iso, w.floor, w.bg)]     # every per-camera get_loss adds them once
            terms, work = _hip.shared_terms_forward(m3, rot, shared, w5)
        cams = [d["cam"] for d in datas for _ in (0, 1)]
        colours = _view_colours(params, variables, V, True)
        # no host wait inside the forward when the capacity of the previous step is known: the entry counts are checked below, before
        # anything is differentiated (the images never leave this function, so a forward that overflowed its buffers is simply redone)
        ims, radii, _depth, states = _hip.rasterize_forward_batch(cams, m3, op, colours, None, sc, rot, None, prepare_backward=True,
                                                                  no_host_sync=_no_host_sync)
        ids = [int(d["id"]) for d in datas]
        targets = [t for d in datas for t in (d["im"], d["seg"])]
        rows = [r for i in ids for r in (i, -1)]
        win = _window_1d()
        cam_m, cam_c = params["cam_m"], params["cam_c"]
        losses, lstate = _hip.views_loss_forward(win, ims, targets, rows, [w.im, w.seg] * V, cam_m, cam_c, 0.8, 0.2)
        total = losses[-1]
        one = _ONES.get(dev)
        if one is None:
            one = _ONES[dev] = torch.ones((1,), dtype=torch.float32, device=dev)
        if shared is not None:
            total = total + terms[5]
        # ---- backward, in reverse
        d_ims, d_cm, d_cc = _hip.views_loss_backward(lstate, ims, cam_m, cam_c, one, 0.8, 0.2)
        # the image terms do not depend on the list sizes; the rasterizer's backward does (its scratch is sized by the capacity):
        # look at the counts now -- the GPU still has the image-term kernels queued, so the host wait hides behind them
        if not _hip.forward_counts_ok(states):     # more entries than the buffers were sized for (the scene grew by > 50 % in one step)
            return loss_and_grads_views(params, datas, variables, is_initial_timestep, w, _no_host_sync=False, grad_out=grad_out)
        if "means3D" in go and states[0].pre is not None:      # (the pre-allocated outputs of the forward are fresh tensors: keep all but d_means3D)
            states[0].pre["d_means3D"] = go["means3D"]
        d3, d2, _dc, d_op, d_sc, d_rot, _dcov, _dsh = _hip.rasterize_backward_batch(states, d_ims, m3, radii, colours, None, sc, rot, None,
                                                                                  want_color_grad=False,
                                                                                  grad_out={"d_means3D": go["means3D"]} if "means3D" in go else None)
        if shared is not None:
            _hip.shared_terms_backward(m3, rot, shared, w5, one, accumulate_into=(d3, d_rot), work=work)
        d_un, d_lo, d_ls = _hip.activate_backward(params["unnorm_rotations"], op, sc, d_rot, d_op, d_sc,
                                                  out=(go.get("unnorm_rotations"), go.get("logit_opacities"), go.get("log_scales")))
        for k, g in (("means3D", d3), ("unnorm_rotations", d_un), ("logit_opacities", d_lo), ("log_scales", d_ls), ("cam_m", d_cm),
                     ("cam_c", d_cc)):
            if params[k].requires_grad:
                _acc_grad(params[k], g)
    rad = radii[0::2]
    if is_initial_timestep:
        m2r = variables["max_2D_radius"]
        if m2r.dtype == torch.float32 and m2r.is_contiguous() and radii.is_contiguous():
            variables["seen"] = _hip.radius_bookkeeping(radii, 2, m2r)      # one kernel; max_2D_radius updated in place
        else:
            _radius_bookkeeping(variables, rad)
    return total, variables, dict(means2D_grad=d2, radii=rad)


@torch.no_grad()
def render_step_views(params, cams, dL, colours_key: str = "rgb_colors", want_colour_grad: bool = True, _no_host_sync: bool = True,
                      _defer_counts: bool = False, grad_out=None):
    """Forward + backward of the colour render of every camera in ``cams`` with the upstream gradient ``dL`` [V,3,H,W], as a
    straight sequence of library calls -- fused activations, ONE multi-view rasterizer forward (capacity mode: no host wait),
    ONE multi-view backward, fused activation backward -- without the autograd engine.  This is the rasterizer share of one
    ``train_gs.py`` iteration (/root/reference/src/tracking/train_gs.py:25-39) for this rank's views, the step `bench.py` times.
    Returns (images [V,3,H,W], grads) with grads = dict of the PARAMETER gradients summed over the views (means3D,
    unnorm_rotations, logit_opacities, log_scales and, if wanted, the colour array) plus ``means2D`` [V,P,3] and ``radii`` [V,P].
    ``grad_out``: dict parameter name -> caller-owned tensor shaped like the parameter (``gsdyn.dp.GradBucket.views()``): the per-Gaussian
    backward kernel writes those parameter gradients there -- straight into the all-reduce bucket, no packing copy afterwards -- and the
    returned dict holds these very tensors.  (Only with the activations fused into the rasterizer's kernels -- the steady state; the first
    call of a shape and an overflow repeat return fresh tensors, which the bucket then packs as before.)"""
    from diff_gaussian_rasterization import _hip
    m3 = params["means3D"]
    dev = m3.device
    V = len(cams)
    if not m3.is_cuda or V > _hip.MAX_BATCH:
        raise RuntimeError("render_step_views: needs a HIP device and at most %d views per call" % _hip.MAX_BATCH)
    colours = params[colours_key].detach()
    with _hip.hold_stream(dev):
        # the activations (normalize / sigmoid / exp) ride inside the preprocess kernel once the call runs in capacity mode, and
        # their chain inside the per-Gaussian backward kernel (gsr_raw_params): two launches per step fewer
        raw = (params["unnorm_rotations"].detach(), params["logit_opacities"].detach(), params["log_scales"].detach())
        if os.environ.get("GSR_NO_FUSED_ACTIVATIONS") == "1":     # A/B switch: the two stand-alone activation launches
            rot, op, sc = _hip.activate_forward(*raw)
            ims, radii, _depth, states = _hip.rasterize_forward_batch(list(cams), m3, op, colours, None, sc, rot, None,
                                                                      prepare_backward=True, no_host_sync=_no_host_sync)
        else:
            go = None
            if grad_out is not None:     # fused mode: d_rot / d_opacity / d_scales come back as the gradients of the RAW parameters
                go = {a: grad_out.get(b) for a, b in (("d_means3D", "means3D"), ("d_rot", "unnorm_rotations"), ("d_opacity", "logit_opacities"),
                                                        ("d_scales", "log_scales"), ("d_colors", colours_key if want_colour_grad else None))}
            ims, radii, _depth, states = _hip.rasterize_forward_batch(list(cams), m3, None, colours, None, None, None, None,
                                                                      prepare_backward=True, no_host_sync=_no_host_sync, raw=raw, grad_out=go)
            rot, op, sc = states[0].act
            if go is not None and states[0].raw_fused is None and states[0].pre is not None:
                states[0].pre = None     # activations NOT fused this time: the backward's outputs are gradients of the ACTIVATED tensors
        # The backward is queued right behind the forward, BEFORE the forward's entry counts are known on the host: on lists
        # that overflowed their capacity it is still memory-safe (emit never writes past the capacity, record reads are clamped
        # to it), its results are then simply dropped and the step is repeated synchronously.
        d3, d2, dc, d_op, d_sc, d_rot, _dcov, _dsh = _hip.rasterize_backward_batch(states, dL, m3, radii, colours, None, sc, rot, None,
                                                                                  want_color_grad=want_colour_grad)
        if states[0].raw_fused is not None:
            d_un, d_lo, d_ls = d_rot, d_op, d_sc      # already the gradients of the unactivated parameters
        else:
            d_un, d_lo, d_ls = _hip.activate_backward(params["unnorm_rotations"], op, sc, d_rot, d_op, d_sc)
        if _defer_counts:      # stream capture (GraphedRenderStep): no host wait in here, the caller checks the counts after a replay
            pass
        elif not _hip.forward_counts_ok(states):      # the scene outgrew the remembered capacity (> 50 % more entries in one step)
            return render_step_views(params, cams, dL, colours_key, want_colour_grad, _no_host_sync=False, grad_out=grad_out)
    grads = {"means3D": d3, "unnorm_rotations": d_un, "logit_opacities": d_lo, "log_scales": d_ls, "means2D": d2, "radii": radii}
    if _defer_counts:
        grads["_states"] = states
    if want_colour_grad:
        grads[colours_key] = dc
    return ims, grads


class GraphedRenderStep:
    """``render_step_views`` captured ONCE into a hipGraph and replayed: the ~12 kernel launches of a step become one graph launch
    (no per-launch host work, back-to-back dispatch on the GPU).  The parameter tensors, the camera records and ``dL`` are read in
    place on every replay (Adam updates the parameters in place; write a new upstream gradient with ``dL.copy_``); images and
    gradients come back in the same buffers each time.  The forward runs in capacity mode: after a replay ``ok()`` tells whether
    every view's entry count fitted the captured buffers -- if not (the scene grew by more than 50 % since the capture) the
    results of that replay are invalid and the step must be re-captured (``GraphedRenderStep(...)`` again)."""

    def __init__(self, params, cams, dL, colours_key: str = "rgb_colors", want_colour_grad: bool = True, warmup: int = 2):
        from diff_gaussian_rasterization import _hip
        self._hip = _hip
        self.args = (params, list(cams), dL, colours_key, want_colour_grad)
        dev = params["means3D"].device
        for _ in range(max(warmup, 1)):       # establishes the capacity, the pinned counts slot and the allocator's pools
            render_step_views(*self.args)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            with torch.cuda.graph(self.graph, stream=side):
                self.images, self.grads = render_step_views(*self.args, _defer_counts=True)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._states = self.grads.pop("_states")
        pend = self._states[0].pending
        if pend is None:
            raise RuntimeError("GraphedRenderStep: the captured forward did not run in capacity mode")
        self._counts_host, self._cap = pend[1], pend[3]
        _hip.detach_counts_slot(self._states)     # the replays own this pinned slot from now on: it leaves the ring of the eager calls

    def replay(self):
        self.graph.replay()
        return self.images, self.grads

    def ok(self) -> bool:
        """After a replay has completed (synchronise first): did every view fit the captured capacity?"""
        return int(self._counts_host.max()) <= self._cap


@torch.no_grad()
def report_psnr(params, data):
    """The extra forward render of /root/reference/src/tracking/train_utils.py:377-384."""
    im, _, _ = Renderer(raster_settings=data["cam"])(**params2rendervar(params))
    cid = data["id"]
    im = torch.exp(params["cam_m"][cid])[:, None, None] * im + params["cam_c"][cid][:, None, None]
    return calc_psnr(im, data["im"]).mean()


def make_rigidity_variables(params, num_knn: int = 20, device=None):
    """Neighbour tensors for the t>0 loss terms (/root/reference/src/tracking/train_utils.py:354-374).
    The reference builds the kNN with Open3D on the CPU; here a dense torch.cdist top-k (fg points only)."""
    with torch.no_grad():
        is_fg = params["seg_colors"][:, 0] > 0.5
        fg = params["means3D"][is_fg]
        rot = torch.nn.functional.normalize(params["unnorm_rotations"])
        n = fg.shape[0]
        k = min(num_knn, max(n - 1, 1))
        idx_chunks, d_chunks = [], []
        for s in range(0, n, 4096):
            d = torch.cdist(fg[s:s + 4096], fg)
            dk, ik = torch.topk(d, k + 1, dim=1, largest=False)
            idx_chunks.append(ik[:, 1:]); d_chunks.append(dk[:, 1:] ** 2)
        nbr = torch.cat(idx_chunks) if idx_chunks else torch.zeros((0, k), dtype=torch.long, device=fg.device)
        sq = torch.cat(d_chunks) if d_chunks else torch.zeros((0, k), device=fg.device)
        inv = rot[is_fg].clone()
        inv[:, 1:] = -inv[:, 1:]
        rev_ptr, rev_edge = reverse_adjacency(nbr.long())
        return dict(fg_idx=is_fg.nonzero().squeeze(1).contiguous(), bg_idx=(~is_fg).nonzero().squeeze(1).contiguous(),
                    rev_ptr=rev_ptr, rev_edge=rev_edge,
                    neighbor_indices=nbr.long().contiguous(), neighbor_weight=torch.exp(-2000 * sq).contiguous(),
                    neighbor_dist=torch.sqrt(sq).contiguous(), init_bg_pts=params["means3D"][~is_fg].detach().clone(),
                    init_bg_rot=rot[~is_fg].detach().clone(), prev_inv_rot_fg=inv.detach(),
                    prev_offset=(fg[nbr] - fg[:, None]).detach())
