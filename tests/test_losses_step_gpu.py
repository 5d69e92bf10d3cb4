"""Rows N2, A9, A10: fused SSIM + L1, rigidity terms, activations, the direct tracking step, Adam, densification, end-to-end fits.
(split out of the former tests/test_hip_gpu.py; shared machinery: tests/hipcheck.py, fixtures: tests/conftest.py)"""
import os

import numpy as np
import pytest
import torch

from hipcheck import *  # noqa: F401,F403
from hipcheck import _check_against_oracle, _check_lists, _margin, _pin_tile_sort_build, _row_check, _run_hip, _settings  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W", [(64, 48), (37, 53), (800, 800)])
def test_fused_image_loss_matches_torch_formula(dev, H, W):
    """Fused 0.8 L1 + 0.2 (1 - SSIM) kernels vs the torch restatement of the reference formula (value and gradient)."""
    from gsdyn import losses as L
    rng = np.random.default_rng(H * 1000 + W)
    x = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev, requires_grad=True)
    y = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32), device=dev)
    xd = x.double()                # fp64 reference: no timing-dependent MIOpen solver choice in the comparison
    ref = 0.8 * L.l1_loss_v1(xd, y.double()) + 0.2 * (1.0 - L.calc_ssim(xd, y.double()))
    (gref,) = torch.autograd.grad(ref * 3.0, x)
    x2 = x.detach().clone().requires_grad_(True)
    got = L.image_loss(x2, y)
    (ggot,) = torch.autograd.grad(got * 3.0, x2)
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert (ggot - gref).abs().max().item() <= 1e-4 * gref.abs().max().item()


def test_fused_image_loss_batch_equals_per_image(dev):
    """A batch [N,3,H,W] through one kernel pair gives the per-image losses and gradients of N separate calls."""
    from gsdyn import losses as L
    rng = np.random.default_rng(11)
    N, H, W = 5, 72, 100
    x = torch.tensor(rng.uniform(0, 1, (N, 3, H, W)).astype(np.float32), device=dev)
    y = torch.tensor(rng.uniform(0, 1, (N, 3, H, W)).astype(np.float32), device=dev)
    wts = torch.tensor([1.0, -2.0, 0.5, 3.0, 0.25], device=dev)
    xb = x.clone().requires_grad_(True)
    lb = L.image_loss(xb, y)
    assert lb.shape == (N,)
    (lb * wts).sum().backward()
    for i in range(N):
        xi = x[i].clone().requires_grad_(True)
        li = L.image_loss(xi, y[i])
        (li * wts[i]).backward()
        assert torch.equal(li, lb[i]) or abs(li.item() - lb[i].item()) <= 1e-6 * abs(li.item())
        assert (xi.grad - xb.grad[i]).abs().max().item() <= 1e-6 * xi.grad.abs().max().item()


@pytest.mark.parametrize("H,W", [(61, 45), (120, 200)])
def test_views_loss_matches_torch_formula(dev, H, W):
    """gsr_views_loss_*: the image terms of all renders of a step (camera affine included, one camera used twice) vs the
    torch restatement of /root/reference/src/tracking/train_utils.py:181-195 -- total, per-image losses, and the gradients to
    the render batch, cam_m and cam_c."""
    from gsdyn import losses as L
    rng = np.random.default_rng(H + W)
    n, ncam = 6, 5
    mk = lambda *sh: torch.tensor(rng.uniform(0, 1, sh).astype(np.float32), device=dev)   # noqa: E731
    renders = mk(n, 3, H, W)
    targets = [mk(3, H, W) for _ in range(n)]
    rows = [3, -1, 0, -1, 3, -1]
    weights = [50.0, 200.0, 50.0, 200.0, 50.0, 200.0]
    cam_m = (mk(ncam, 3) * 0.4 - 0.2).requires_grad_(True)
    cam_c = (mk(ncam, 3) * 0.2 - 0.1).requires_grad_(True)

    def torch_total(r, m, c):
        per = []
        for i in range(n):
            pred = r[i] if rows[i] < 0 else torch.exp(m[rows[i]])[:, None, None] * r[i] + c[rows[i]][:, None, None]
            pd, td = pred.double(), targets[i].double()          # fp64: see test_fused_image_loss_matches_torch_formula
            per.append((0.8 * L.l1_loss_v1(pd, td) + 0.2 * (1.0 - L.calc_ssim(pd, td))).float())
        return sum(w * l for w, l in zip(weights, per)), torch.stack(per)

    r1 = renders.clone().requires_grad_(True)
    ref, ref_per = torch_total(r1, cam_m, cam_c)
    g_ref = torch.autograd.grad(ref * 0.7, (r1, cam_m, cam_c))
    r2 = renders.clone().requires_grad_(True)
    got, got_per = L.views_image_loss(r2, targets, rows, weights, cam_m, cam_c)
    g_got = torch.autograd.grad(got * 0.7, (r2, cam_m, cam_c))
    assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert (got_per - ref_per.detach()).abs().max().item() <= 1e-5
    for a_, b_ in zip(g_got, g_ref):
        assert a_.shape == b_.shape
        assert (a_ - b_).abs().max().item() <= 1e-4 * b_.abs().max().item()
    assert float(g_got[1][1].abs().max()) == 0.0 and float(g_got[1][3].abs().max()) > 0.0    # unused / doubly used camera rows
    # deterministic: a second evaluation gives the same bits
    r3 = renders.clone().requires_grad_(True)
    got2, _ = L.views_image_loss(r3, targets, rows, weights, cam_m, cam_c)
    g2 = torch.autograd.grad(got2 * 0.7, (r3, cam_m, cam_c))
    assert torch.equal(got, got2) and all(torch.equal(x_, y_) for x_, y_ in zip(g_got, g2))


@pytest.mark.parametrize("initial", [True, False])
def test_direct_step_equals_autograd_step(dev, initial):
    """``loss_and_grads_views`` (the library calls back to back, no autograd graph) == ``get_loss_views(frozen_colours=True)`` +
    ``backward()``: loss, every parameter gradient, the per-view screen-space gradients and the bookkeeping tensors."""
    from gsdyn import LossWeights, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    from gsdyn.step import make_rigidity_variables
    P, W, H = 4000, 160, 120
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.08)
    with torch.no_grad():
        params["cam_m"].add_(0.05 * torch.randn_like(params["cam_m"]))
        params["means3D"].add_(0.002 * torch.randn_like(params["means3D"]))
    cams = synth_ring_cameras(3, W, H, device=dev)
    im_gt, seg_gt = synth_targets(W, H, device=dev)
    w = LossWeights()
    views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in (1, 0, 1)]
    rig = make_rigidity_variables(params, num_knn=8)

    def fresh_variables():
        v = init_variables(P, dev)
        v.update(rig)
        return v
    for p_ in params.values():
        p_.grad = None
    loss_a, var_a, aux_a = get_loss_views(params, views, fresh_variables(), initial, w, frozen_colours=True)
    loss_a.backward()
    ga = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
    m2a = aux_a["means2D"].grad.clone()
    for p_ in params.values():
        p_.grad = None
    loss_b, var_b, aux_b = loss_and_grads_views(params, views, fresh_variables(), initial, w)
    gb = {k: v.grad for k, v in params.items() if v.grad is not None}
    assert abs(float(loss_a.detach()) - float(loss_b)) <= 1e-6 * abs(float(loss_b))
    assert set(ga) == set(gb) and "cam_m" in ga and "seg_colors" not in ga
    for k in ga:
        assert (ga[k] - gb[k]).abs().max().item() <= 1e-6 * ga[k].abs().max().item() + 1e-20, k
    assert (m2a - aux_b["means2D_grad"]).abs().max().item() <= 1e-6 * m2a.abs().max().item() + 1e-20
    if initial:   # the direct step keeps the densification bookkeeping for the first timestep only (its only reader)
        assert torch.equal(var_a["max_2D_radius"], var_b["max_2D_radius"]) and torch.equal(var_a["seen"], var_b["seen"])
    # a second call accumulates
    loss_and_grads_views(params, views, fresh_variables(), initial, w)
    assert (params["means3D"].grad - 2 * ga["means3D"]).abs().max().item() <= 1e-5 * ga["means3D"].abs().max().item()


def test_full_size_direct_step_against_literal_torch_step(dev):
    """BASELINE-size end-to-end check of the most fused path against the most literal one, t > 0, 2 cameras at 800 x 800, 100 k
    Gaussians: ``loss_and_grads_views`` (fused activations, pair passes, fused image terms with the camera affine, fused shared
    terms, no autograd) vs one ``GaussianRasterizer`` call per render, the camera affine / 0.8 L1 + 0.2 (1 - SSIM) / rigid / rot /
    iso / floor / bg terms as the reference's torch formulas, and autograd."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import LossWeights, loss_and_grads_views, params2rendervar, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn import losses as L
    from gsdyn.dp import init_variables
    from gsdyn.step import _SHARED_NAMES, _shared_terms, make_rigidity_variables
    torch.manual_seed(1234)
    P, W, H = 100_000, 800, 800
    params = synth_scene_params(P, device=dev)
    with torch.no_grad():
        params["cam_m"].add_(0.05 * torch.randn_like(params["cam_m"]))
        params["cam_c"].add_(0.02 * torch.randn_like(params["cam_c"]))
    cams = synth_ring_cameras(4, W, H, device=dev)
    im_gt, seg_gt = synth_targets(W, H, device=dev)
    w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)
    views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in (0, 2)]
    rig = make_rigidity_variables(params, num_knn=20)
    with torch.no_grad():
        params["means3D"].add_(0.003 * torch.randn_like(params["means3D"]))        # move away from the rest pose
        params["unnorm_rotations"].add_(0.02 * torch.randn_like(params["unnorm_rotations"]))

    for p_ in params.values():
        p_.grad = None
    v1 = init_variables(P, dev)
    v1.update(rig)
    loss_f, _, aux = loss_and_grads_views(params, views, v1, False, w)
    g_f = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}

    for p_ in params.values():
        p_.grad = None
    torch_vars = {k: v for k, v in rig.items() if k not in ("rev_ptr", "rev_edge")}     # no reverse adjacency -> torch formulas
    weights = dict(rigid=w.rigid, rot=w.rot, iso=w.iso, floor=w.floor, bg=w.bg)
    total = 0.0
    for d in views:
        rv = params2rendervar(params)
        im, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**rv)
        im = torch.exp(params["cam_m"][d["id"]])[:, None, None] * im + params["cam_c"][d["id"]][:, None, None]
        # image terms in fp64: no MIOpen solver choice (it is timing-dependent, and some fp32 solvers are not accurate to 1e-4)
        imd, segt = im.double(), d["im"].double()
        l_im = (0.8 * L.l1_loss_v1(imd, segt) + 0.2 * (1.0 - L.calc_ssim(imd, segt))).float()
        sv = params2rendervar(params, colors_key="seg_colors")
        seg, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**sv)
        segd, segg = seg.double(), d["seg"].double()
        l_seg = (0.8 * L.l1_loss_v1(segd, segg) + 0.2 * (1.0 - L.calc_ssim(segd, segg))).float()
        shared, _ = _shared_terms(params, rv, torch_vars, weights)        # fp32 torch ops; their fp64 evaluation agrees to 1.5e-5
        loss = w.im * l_im + w.seg * l_seg + shared                       # (tools/micro/shared_terms_precision.py)
        loss.backward()
        total += float(loss.detach())
    assert abs(float(loss_f) - total) <= 2e-5 * abs(total), (float(loss_f), total)
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "cam_m", "cam_c"):
        want, got = params[k].grad, g_f[k]
        assert _margin(f"direct_step/{k}", (got - want).abs().max().item(), want.abs().max().item()) <= TOL, k
    assert len(_SHARED_NAMES) == 5 and aux["means2D_grad"].shape == (4, P, 3)


def test_fused_adam_equals_torch_adam(dev):
    """gsdyn.optim.FusedAdam (all parameter groups in one gsr_adam_step launch) against torch.optim.Adam with the reference's
    group layout (per-group lr, eps 1e-15, one group with lr 0, one parameter without a gradient): parameters and both moment
    buffers after several steps; state layout interchangeable (state_dict round trip into torch's Adam)."""
    from gsdyn.optim import FusedAdam
    g = torch.Generator(device="cpu").manual_seed(12)
    shapes = dict(means3D=(5003, 3), rot=(5003, 4), op=(5003, 1), frozen=(5003, 3), cam=(50, 3), nograd=(77, 3))
    lrs = dict(means3D=6.4e-4, rot=1e-3, op=0.05, frozen=0.0, cam=1e-4, nograd=0.01)
    init = {k: torch.randn(sh, generator=g).to(dev) for k, sh in shapes.items()}
    grads = [{k: torch.randn(sh, generator=g).to(dev) * (10.0 ** (i - 2)) for k, sh in shapes.items()} for i in range(5)]

    def run(cls):
        ps = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
        opt = cls([{"params": [v], "name": k, "lr": lrs[k]} for k, v in ps.items()], lr=0.0, eps=1e-15)
        for gr in grads:
            for k, v in ps.items():
                v.grad = None if k == "nograd" else gr[k].clone()
            opt.step()
        return ps, opt
    pa, oa = run(torch.optim.Adam)
    pb, ob = run(FusedAdam)
    for k in shapes:
        assert (pa[k] - pb[k]).abs().max().item() <= 2e-6 * max(1.0, pa[k].abs().max().item()), k
        if k != "nograd":
            sa, sb = oa.state[pa[k]], ob.state[pb[k]]
            assert float(sa["step"]) == float(sb["step"]) == 5.0
            for name in ("exp_avg", "exp_avg_sq"):
                assert (sa[name] - sb[name]).abs().max().item() <= 2e-6 * sa[name].abs().max().item() + 1e-30, (k, name)
    assert torch.equal(pb["nograd"], init["nograd"]) and len(ob.state[pb["nograd"]]) == 0
    assert torch.equal(pb["frozen"], init["frozen"])             # lr 0: moments move, the parameter does not
    oa.load_state_dict(ob.state_dict())                            # same state layout


def test_direct_step_without_host_sync_and_its_overflow_path(dev):
    """The direct step renders in capacity mode (gsr_forward_batch_capacity: buffers sized from the previous call, entry counts
    read on the device, no host wait in the forward).  (1) Steady state: same loss and gradients as the synchronous autograd
    step, bit for bit in the rasterizer's integers (radii) and within rounding in the floats.  (2) The scene grows by more than
    the slack between two calls: the forward overflows its buffers, the step notices before differentiating and repeats itself
    synchronously -- still the right answer."""
    from diff_gaussian_rasterization import _hip
    from gsdyn import LossWeights, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    P, W, H = 5000, 176, 144                       # a shape no other test uses: its capacity cache starts empty
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.04)
    cams = synth_ring_cameras(2, W, H, device=dev)
    im_gt, seg_gt = synth_targets(W, H, device=dev)
    w = LossWeights()
    views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in (0, 1)]
    key = (dev.index, P, H, W)
    _hip._entries_capacity.pop(key, None)

    def both():
        for p_ in params.values():
            p_.grad = None
        la, _, aux_a = get_loss_views(params, views, init_variables(P, dev), True, w, frozen_colours=True)
        la.backward()
        ga = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
        for p_ in params.values():
            p_.grad = None
        lb, _, aux_b = loss_and_grads_views(params, views, init_variables(P, dev), True, w)
        gb = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
        assert abs(float(la.detach()) - float(lb)) <= 1e-6 * abs(float(lb))
        assert torch.equal(aux_a["radii"], aux_b["radii"])
        for k in ga:
            assert (ga[k] - gb[k]).abs().max().item() <= 1e-6 * ga[k].abs().max().item() + 1e-20, k
    both()                                          # first call: no capacity yet -> synchronous; leaves one behind
    cap1 = _hip._entries_capacity[key]
    calls = []
    orig = _hip.rasterize_forward_batch

    def spy(*a_, **k_):
        out = orig(*a_, **k_)
        calls.append((bool(k_.get("no_host_sync")), out[3][0].pending is not None))
        return out
    _hip.rasterize_forward_batch = spy
    try:
        both()                                      # steady state: the direct step runs in capacity mode
        assert (True, True) in calls
        with torch.no_grad():
            params["log_scales"].add_(0.9)          # every Gaussian 2.5x larger: far more list entries than the capacity
        calls.clear()
        both()
        assert (True, True) in calls and (False, False) in calls     # overflowed, then repeated synchronously
    finally:
        _hip.rasterize_forward_batch = orig
    assert _hip._entries_capacity[key] > cap1


def test_views_loss_with_cached_target_moments_is_bit_identical(dev):
    """A target seen for the second time has blur(y), blur(y*y) computed once (gsr_target_moments) and the forward runs its 3-moment
    build from then on: same bits as the 5-moment build, for the losses and for every gradient; an in-place change of a target
    invalidates its entry."""
    from diff_gaussian_rasterization import _hip
    from gsdyn import losses as L
    rng = np.random.default_rng(21)
    n, H, W = 4, 131, 97
    mk = lambda *sh: torch.tensor(rng.uniform(0, 1, sh).astype(np.float32), device=dev)   # noqa: E731
    renders, targets = mk(n, 3, H, W), [mk(3, H, W) for _ in range(n)]
    rows, weights = [1, -1, 0, -1], [50.0, 200.0, 50.0, 200.0]
    cam_m, cam_c = (mk(3, 3) * 0.2).requires_grad_(True), (mk(3, 3) * 0.1).requires_grad_(True)
    _hip._target_moments.clear()
    seen = []
    orig = _hip._loss_table

    def spy(*a_, **k_):
        seen.append(a_[4] is not None if len(a_) > 4 else k_.get("moments") is not None)
        return orig(*a_, **k_)
    _hip._loss_table = spy
    try:
        outs = []
        for it in range(3):
            r = renders.clone().requires_grad_(True)
            total, per = L.views_image_loss(r, targets, rows, weights, cam_m, cam_c)
            g = torch.autograd.grad(total, (r, cam_m, cam_c))
            outs.append((total.detach().clone(), per.clone(), [x.clone() for x in g]))
        assert seen == [False, True, True]
        for it in (1, 2):
            assert torch.equal(outs[0][0], outs[it][0]) and torch.equal(outs[0][1], outs[it][1])
            assert all(torch.equal(x, y) for x, y in zip(outs[0][2], outs[it][2]))
        targets[2].mul_(0.5)                        # new version of one target: its cached maps no longer apply
        seen.clear()
        r = renders.clone().requires_grad_(True)
        t1, _ = L.views_image_loss(r, targets, rows, weights, cam_m, cam_c)
        t2, _ = L.views_image_loss(r, targets, rows, weights, cam_m, cam_c)
        assert seen == [False, True] and torch.equal(t1, t2) and not torch.equal(t1, outs[0][0])
    finally:
        _hip._loss_table = orig


def test_views_loss_more_images_than_one_library_call(dev):
    """40 images (> GSR_LOSS_MAX_IMAGES = 32): the Python entry point splits the call; total and gradients equal the per-image sums."""
    from gsdyn import losses as L
    rng = np.random.default_rng(3)
    n, H, W = 40, 33, 47
    mk = lambda *sh: torch.tensor(rng.uniform(0, 1, sh).astype(np.float32), device=dev)   # noqa: E731
    renders, targets = mk(n, 3, H, W).requires_grad_(True), [mk(3, H, W) for _ in range(n)]
    rows = [i % 3 if i % 2 == 0 else -1 for i in range(n)]
    weights = [1.0 + 0.1 * i for i in range(n)]
    cam_m, cam_c = (mk(3, 3) * 0.2).requires_grad_(True), (mk(3, 3) * 0.1).requires_grad_(True)
    total, per = L.views_image_loss(renders, targets, rows, weights, cam_m, cam_c)
    g = torch.autograd.grad(total, (renders, cam_m, cam_c))
    r2 = renders.detach().clone().requires_grad_(True)
    ref = 0.0
    for i in range(n):
        pred = r2[i] if rows[i] < 0 else torch.exp(cam_m[rows[i]])[:, None, None] * r2[i] + cam_c[rows[i]][:, None, None]
        ref = ref + weights[i] * L.image_loss(pred, targets[i])
    g_ref = torch.autograd.grad(ref, (r2, cam_m, cam_c))
    assert per.shape == (n,) and abs(total.item() - ref.item()) <= 1e-5 * abs(ref.item())
    for a_, b_ in zip(g, g_ref):
        assert (a_ - b_).abs().max().item() <= 1e-5 * b_.abs().max().item()


def test_get_loss_views_equals_sum_of_get_loss(dev):
    """The fused multi-camera step (one rasterizer call + one loss call) against the per-camera ``get_loss`` sum: value and
    every parameter gradient, cam_m / cam_c included (a camera sampled twice)."""
    from gsdyn import LossWeights, get_loss, get_loss_views, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    P, W, H = 3000, 160, 120
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.08)
    with torch.no_grad():
        params["cam_m"].add_(0.05 * torch.randn_like(params["cam_m"]))
        params["cam_c"].add_(0.02 * torch.randn_like(params["cam_c"]))
    cams = synth_ring_cameras(4, W, H, device=dev)
    im_gt, seg_gt = synth_targets(W, H, device=dev)
    w = LossWeights()
    ids = [2, 0, 2]
    views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in ids]
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar
    from gsdyn import losses as L
    for p_ in params.values():
        p_.grad = None
    total, total32 = 0.0, 0.0
    for d in views:       # the literal per-camera step (train_utils.py:174-195) with the image terms evaluated in fp64 on the renders
        im, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**params2rendervar(params))
        im = torch.exp(params["cam_m"][d["id"]])[:, None, None] * im + params["cam_c"][d["id"]][:, None, None]
        seg, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**params2rendervar(params, colors_key="seg_colors"))
        l_im = 0.8 * L.l1_loss_v1(im.double(), d["im"].double()) + 0.2 * (1.0 - L.calc_ssim(im.double(), d["im"].double()))
        l_seg = 0.8 * L.l1_loss_v1(seg.double(), d["seg"].double()) + 0.2 * (1.0 - L.calc_ssim(seg.double(), d["seg"].double()))
        loss = w.im * l_im + w.seg * l_seg
        loss.backward()
        total += float(loss.detach())
    ref = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
    for p_ in params.values():
        p_.grad = None
    for d in views:       # ... and gsdyn.get_loss (fp32 torch loss ops) agrees on the value
        loss, _ = get_loss(params, d, init_variables(P, dev), True, w)
        total32 += float(loss.detach())
    assert abs(total32 - total) <= 2e-5 * abs(total)
    for p_ in params.values():
        p_.grad = None
    loss, _, _ = get_loss_views(params, views, init_variables(P, dev), True, w)
    loss.backward()
    assert abs(float(loss.detach()) - total) <= 2e-5 * abs(total)
    assert "cam_m" in ref and "cam_c" in ref and float(ref["cam_m"].abs().max()) > 0
    for k, g in ref.items():
        got = params[k].grad
        assert got is not None, k
        assert _margin(f"get_loss_views/{k}", (got - g).abs().max().item(), g.abs().max().item() + 1e-30) <= TOL, k


def test_fused_image_loss_matches_reference_golden(dev, golden_dir):
    """Against vectors captured from the imported reference (calc_ssim value and input gradient)."""
    from gsdyn import losses as L
    ref = np.load(os.path.join(golden_dir, "reference_host.npz"))
    x = torch.tensor(ref["ssim_im1"], device=dev, requires_grad=True)
    y = torch.tensor(ref["ssim_im2"], device=dev)
    got = L.image_loss(x, y, 0.0, 1.0)           # = 1 - SSIM
    np.testing.assert_allclose(1.0 - got.item(), float(ref["ssim"]), rtol=2e-5)
    got.backward()
    gx = -x.grad.cpu().numpy()
    assert _margin("ssim_golden/grad", np.abs(gx - ref["ssim_grad"]).max(), np.abs(ref["ssim_grad"]).max()) <= TOL
    comb = L.image_loss(x.detach(), y)
    np.testing.assert_allclose(comb.item(), float(ref["im_term"]), rtol=2e-5)


def test_end_to_end_fit(dev):
    """End-to-end sanity: targets are rendered from ground-truth Gaussians, the parameters
    are perturbed, and the corrected loop (gsdyn.train.train_timestep, 4 views per optimiser step) must pull
    the render back towards the targets: PSNR on view 0 improves by > 3 dB in 150 steps."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import (LossWeights, initialize_optimizer, params2rendervar, synth_ring_cameras, synth_scene_params,
                       train_timestep)
    from gsdyn.dp import init_variables
    from gsdyn.step import report_psnr
    P, W, H, V = 4000, 200, 152, 4
    gt = synth_scene_params(P, seed=3, device=dev, scale_lo=0.02, scale_hi=0.08)
    cams = synth_ring_cameras(V, W, H, device=dev)
    views = []
    with torch.no_grad():
        for i, cam in enumerate(cams):
            im, _, _ = GaussianRasterizer(raster_settings=cam)(**params2rendervar(gt))
            seg, _, _ = GaussianRasterizer(raster_settings=cam)(**params2rendervar(gt, colors_key="seg_colors"))
            views.append(dict(cam=cam, im=im, seg=seg, id=i))
    params = synth_scene_params(P, seed=3, device=dev, scale_lo=0.02, scale_hi=0.08)
    g = torch.Generator(device="cpu").manual_seed(0)
    with torch.no_grad():
        params["means3D"].add_(0.03 * torch.randn(P, 3, generator=g).to(dev))
        params["logit_opacities"].add_(0.8 * torch.randn(P, 1, generator=g).to(dev))
        params["log_scales"].add_(0.25 * torch.randn(P, 3, generator=g).to(dev))
    opt = initialize_optimizer(params, scene_radius=4.0)
    for grp in opt.param_groups:   # a short test: larger steps than the 10 000-iteration schedule of the reference
        grp["lr"] *= 5.0
    variables = init_variables(P, dev)
    psnr0 = float(report_psnr(params, views[0]))
    train_timestep(params, variables, opt, views, iters=150, is_initial_timestep=True, weights=LossWeights(),
                   views_per_step=4, seed=0)
    psnr1 = float(report_psnr(params, views[0]))
    print(f"end-to-end fit: PSNR {psnr0:.2f} -> {psnr1:.2f} dB")
    assert psnr1 > psnr0 + 3.0
    assert variables["denom"].sum() > 0 and variables["means2D_gradient_accum"].sum() > 0


def test_fused_rigidity_terms_match_torch_autograd(dev):
    """rigid / rot / iso through gsr_rigidity.hip against the torch formulas evaluated in fp64 (values and the gradients
    w.r.t. means3D and the normalised rotations), on a scene with foreground and background Gaussians."""
    from gsdyn import synth_scene_params
    from gsdyn.losses import build_rotation, quat_mult, rigidity_terms, weighted_l2_loss_v1, weighted_l2_loss_v2
    from gsdyn.step import make_rigidity_variables
    P = 6000
    params = synth_scene_params(P, device=dev)
    variables = make_rigidity_variables(params, num_knn=20)
    g = torch.Generator(device="cpu").manual_seed(2)
    means = (params["means3D"].detach() + 0.01 * torch.randn(P, 3, generator=g).to(dev))
    rots = torch.nn.functional.normalize(params["unnorm_rotations"].detach() + 0.05 * torch.randn(P, 4, generator=g).to(dev))
    m1, r1 = means.clone().requires_grad_(True), rots.clone().requires_grad_(True)
    a, b, c = rigidity_terms(m1, r1, variables)
    wts = (200.0, 4.0, 1000.0)
    (wts[0] * a + wts[1] * b + wts[2] * c).backward()
    # fp64 torch reference of the same formulas
    is_fg = params["seg_colors"][:, 0] > 0.5
    m2, r2 = means.double().clone().requires_grad_(True), rots.double().clone().requires_grad_(True)
    fg_pts, fg_rot = m2[is_fg], r2[is_fg]
    rel = quat_mult(fg_rot, variables["prev_inv_rot_fg"].double())
    R = build_rotation(rel)
    nbr = variables["neighbor_indices"]
    off = fg_pts[nbr] - fg_pts[:, None]
    offp = (off[:, :, :, None] * R[:, None, :, :]).sum(2)
    nw = variables["neighbor_weight"].double()
    ra = weighted_l2_loss_v2(offp, variables["prev_offset"].double(), nw)
    rb = weighted_l2_loss_v2(rel[nbr], rel[:, None], nw)
    rc = weighted_l2_loss_v1(torch.sqrt((off ** 2).sum(-1) + 1e-20), variables["neighbor_dist"].double(), nw)
    (wts[0] * ra + wts[1] * rb + wts[2] * rc).backward()
    for got, want in ((a, ra), (b, rb), (c, rc)):
        assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item()) + 1e-9
    for got, want, name in ((m1.grad, m2.grad, "means3D"), (r1.grad, r2.grad, "rotations")):
        err = (got.double() - want).abs().max().item()
        assert _margin(f"rigidity/{name}", err, want.abs().max().item()) <= TOL, (name, err, want.abs().max().item())
    assert torch.all(m1.grad[~is_fg] == 0) and torch.all(r1.grad[~is_fg] == 0)


def test_fused_shared_terms_match_torch(dev):
    """All five view-independent t > 0 terms and their weighted sum in the fused kernels (gsr_shared_terms_*) against the torch
    path of ``_shared_terms`` evaluated in fp64: value and the gradients w.r.t. means3D and the normalised rotations."""
    from gsdyn import synth_scene_params
    from gsdyn.step import _SHARED_NAMES, _shared_terms, make_rigidity_variables
    P = 7000
    params = synth_scene_params(P, device=dev)
    variables = make_rigidity_variables(params, num_knn=20)
    g = torch.Generator(device="cpu").manual_seed(5)
    means = params["means3D"].detach() + 0.01 * torch.randn(P, 3, generator=g).to(dev)
    with torch.no_grad():
        means[::7, 1] = means[::7, 1].abs() + 0.01          # some foreground points above the floor
    rots = torch.nn.functional.normalize(params["unnorm_rotations"].detach() + 0.05 * torch.randn(P, 4, generator=g).to(dev))
    weights = dict(rigid=200.0, rot=4.0, iso=1000.0, floor=2.0, bg=200.0)
    m1, r1 = means.clone().requires_grad_(True), rots.clone().requires_grad_(True)
    total, each = _shared_terms(params, dict(means3D=m1, rotations=r1), variables, weights, scale=3.0)
    (total * 0.5).backward()
    v64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in variables.items()
           if k not in ("rev_ptr", "rev_edge")}
    m2, r2 = means.double().clone().requires_grad_(True), rots.double().clone().requires_grad_(True)
    ref_total, ref_each = _shared_terms(params, dict(means3D=m2, rotations=r2), v64, weights, scale=3.0)
    (ref_total * 0.5).backward()
    assert abs(total.item() - ref_total.item()) <= 2e-5 * abs(ref_total.item())
    for i, k in enumerate(_SHARED_NAMES):
        assert abs(each[i].item() - ref_each[i].item()) <= 2e-5 * abs(ref_each[i].item()) + 1e-9, k
    assert float(ref_each[3]) > 0
    for got, want, name in ((m1.grad, m2.grad, "means3D"), (r1.grad, r2.grad, "rotations")):
        err = (got.double() - want).abs().max().item()
        assert _margin(f"shared_terms/{name}", err, want.abs().max().item()) <= TOL, (name, err, want.abs().max().item())
    # deterministic
    m3, r3 = means.clone().requires_grad_(True), rots.clone().requires_grad_(True)
    total3, _ = _shared_terms(params, dict(means3D=m3, rotations=r3), variables, weights, scale=3.0)
    (total3 * 0.5).backward()
    assert torch.equal(total, total3) and torch.equal(m1.grad, m3.grad) and torch.equal(r1.grad, r3.grad)


def test_fused_activations_match_torch(dev):
    """normalize / sigmoid / exp in one kernel each way (gsr_activate_*) vs the torch ops of params2rendervar, incl. a zero quaternion
    and an unused output (its incoming gradient is None)."""
    from gsdyn.losses import activate
    g = torch.Generator(device="cpu").manual_seed(9)
    P = 5003
    u = torch.randn(P, 4, generator=g).to(dev)
    u[5] = 0.0
    lo, ls = torch.randn(P, 1, generator=g).to(dev) * 3, torch.randn(P, 3, generator=g).to(dev)
    wr, wo, ws = torch.randn(P, 4, generator=g).to(dev), torch.randn(P, 1, generator=g).to(dev), torch.randn(P, 3, generator=g).to(dev)
    a = [t.clone().requires_grad_(True) for t in (u, lo, ls)]
    b = [t.clone().requires_grad_(True) for t in (u, lo, ls)]
    rot, op, sc = activate(*a)
    rot_t, op_t, sc_t = torch.nn.functional.normalize(b[0]), torch.sigmoid(b[1]), torch.exp(b[2])
    for x_, y_ in ((rot, rot_t), (op, op_t), (sc, sc_t)):
        assert (x_ - y_).abs().max().item() <= 2e-6 * max(1.0, y_.abs().max().item())
    ((rot * wr).sum() + (op * wo).sum() + (sc * ws).sum()).backward()
    ((rot_t * wr).sum() + (op_t * wo).sum() + (sc_t * ws).sum()).backward()
    for x_, y_ in zip(a, b):
        ok = torch.ones(P, dtype=torch.bool, device=dev)
        ok[5] = False                                         # the zero quaternion: 1e12-scaled gradient, compared relatively below
        assert (x_.grad[ok] - y_.grad[ok]).abs().max().item() <= 1e-5 * max(1.0, y_.grad[ok].abs().max().item())
    assert torch.allclose(a[0].grad[5], b[0].grad[5], rtol=1e-5)
    c = [t.clone().requires_grad_(True) for t in (u, lo, ls)]
    _, op_c, _ = activate(*c)
    (op_c * wo).sum().backward()
    assert float(c[0].grad.abs().max()) == 0.0 and float(c[2].grad.abs().max()) == 0.0
    assert (c[1].grad - b[1].grad).abs().max().item() <= 1e-6


def test_density_control_on_device(dev):
    """A first-timestep step at a density iteration on the GPU: clone / split / prune between backward and the optimiser
    step, then the grown cloud keeps training through the batched get_loss path."""
    from gsdyn import LossWeights, initialize_optimizer, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import ViewShardedStep, init_variables
    P, W, H = 5000, 160, 120
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.15)
    cams = synth_ring_cameras(3, W, H, device=dev)
    views = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=3 + i, device=dev)
        views.append(dict(cam=cam, im=im, seg=seg, id=i))
    opt = initialize_optimizer(params, scene_radius=4.0)
    variables = init_variables(P, dev)
    variables["scene_radius"] = 4.0
    step = ViewShardedStep(params, opt, LossWeights(), density_control=dict(remove_thresh=0.005, remove_thresh_5k=0.25,
                                                                           scale_scene_radius=0.01))
    step(views, variables, is_initial_timestep=True, iteration=10)
    assert params["means3D"].shape[0] == P and float(variables["denom"].sum()) > 0
    variables["means2D_gradient_accum"] += 1.0
    step(views, variables, is_initial_timestep=True, iteration=600)
    n = params["means3D"].shape[0]
    assert n > P and variables["denom"].shape[0] == n
    total, variables = step(views, variables, is_initial_timestep=True, iteration=601)
    assert torch.isfinite(total) and params["means3D"].shape[0] == n


def test_end_to_end_fit_on_the_demo_assets(dev, golden_dir):
    """SURVEY.md section 4's end-to-end sanity on the reference's own demo scene (tests/golden/demo_scene.npz = assets/demo at a
    quarter of the resolution): what demo.py:124-159 -> GSTrainer.update_state_no_env -> GSTrainer.train does
    (/root/reference/src/real_world/gs/trainer.py:76-126, train_utils.py:53-100) -- Gaussians initialised from pcd.ply (scale from
    the 3 nearest neighbours, opacity 0.5, identity rotations), the four masked camera images as targets, colour + segmentation
    render per camera, 0.8 L1 + 0.2 (1 - SSIM), Adam with the reference's learning rates, one camera per iteration.  The fit must
    raise the PSNR of every camera and end above a floor."""
    from gsdyn import LossWeights, Rt_to_w2c, initialize_optimizer, loss_and_grads_views, params2rendervar, setup_camera
    from gsdyn.dp import init_variables
    from diff_gaussian_rasterization import GaussianRasterizer
    z = np.load(os.path.join(golden_dir, "demo_scene.npz"))
    pts = torch.tensor(z["xyz"], device=dev)
    P = pts.shape[0]
    H, W = z["imgs"].shape[1:3]
    d2 = torch.cdist(pts, pts)
    mean3 = torch.topk(d2, 4, dim=1, largest=False)[0][:, 1:].pow(2).mean(1).clamp(min=1e-7)       # the 3 nearest neighbours
    mk = lambda t, g=True: torch.nn.Parameter(t.float().contiguous().to(dev), requires_grad=g)     # noqa: E731
    params = {"means3D": mk(pts), "rgb_colors": mk(torch.tensor(z["rgb"]).float() / 255.0),
              "seg_colors": mk(torch.tensor([1.0, 0.0, 0.0]).repeat(P, 1)), "unnorm_rotations": mk(torch.tensor([1.0, 0, 0, 0]).repeat(P, 1)),
              "logit_opacities": mk(torch.zeros(P, 1)), "log_scales": mk(torch.log(torch.sqrt(mean3))[:, None].repeat(1, 3)),
              "cam_m": mk(torch.zeros(4, 3)), "cam_c": mk(torch.zeros(4, 3))}
    w2cs = [Rt_to_w2c(R, t) for R, t in zip(z["R_list"], z["t_list"])]
    centres = np.stack([np.linalg.inv(m)[:3, 3] for m in w2cs])
    scene_radius = 1.1 * np.max(np.linalg.norm(centres - centres.mean(0)[None], axis=-1))
    data = []
    for c in range(4):
        mask = torch.tensor(z["masks"][c], device=dev).float() / 255.0
        im = (torch.tensor(z["imgs"][c], device=dev).float() / 255.0 * mask[..., None]).permute(2, 0, 1).contiguous()
        seg = torch.stack([mask, torch.zeros_like(mask), 1 - mask]).contiguous()
        data.append(dict(cam=setup_camera(W, H, z["intr_list"][c], w2cs[c], near=0.01, far=100.0, device=dev), im=im, seg=seg, id=c))

    def psnr(c):
        with torch.no_grad():
            im, _, _ = GaussianRasterizer(raster_settings=data[c]["cam"])(**params2rendervar(params))
            return float(-10.0 * torch.log10(((im.clamp(0, 1) - data[c]["im"]) ** 2).mean()))
    before = [psnr(c) for c in range(4)]
    assert all(np.isfinite(before)) and min(before) > 5.0, before      # the cloud projects into every image (cameras are right)
    opt = initialize_optimizer(params, float(scene_radius))
    with torch.no_grad():      # the demo trains the colours too (real_world/gs/train_utils.py:83 leaves requires_grad on)
        for gparam in opt.param_groups:
            if gparam["name"] == "rgb_colors":
                gparam["lr"] = 0.0025
    variables = init_variables(P, dev)
    w = LossWeights(im=1.0, seg=3.0)
    rng = np.random.default_rng(0)
    first = last = None
    from gsdyn import get_loss_views
    for it in range(400):
        d = data[int(rng.integers(4))]
        loss, variables, _ = get_loss_views(params, [d], variables, True, w)      # colour gradients wanted: the autograd path
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if it < 8:
            first = float(loss.detach()) if first is None else max(first, float(loss.detach()))
        last = float(loss.detach())
    after = [psnr(c) for c in range(4)]
    print("demo fit: PSNR before", [round(x, 2) for x in before], "after", [round(x, 2) for x in after], "loss", first, "->", last)
    assert last < first
    assert all(a > b + 1.0 for a, b in zip(after, before)), (before, after)
    assert min(after) > 30.0, after      # measured: 22 dB before, 37 .. 44 dB after 400 iterations


def test_fused_activations_in_the_direct_step(dev):
    """render_step_views: the first call has no capacity yet and runs the stand-alone activation kernels; later calls apply the
    activations inside preprocess_fwd and their chain inside preprocess_bwd_views (gsr_raw_params).  One definition of the
    arithmetic (gsr_common.h), so images and every parameter gradient are bit-identical between the two."""
    from diff_gaussian_rasterization import _hip
    from gsdyn import synth_ring_cameras, synth_scene_params
    from gsdyn.step import render_step_views
    P, W, H, V = 25000, 336, 256, 3
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(21).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    key = (dev.index, P, H, W)
    _hip._entries_capacity.pop(key, None)
    seen = []
    orig = _hip.rasterize_backward_batch

    def spy(states, *a, **k):
        seen.append(states[0].raw_fused is not None)
        return orig(states, *a, **k)
    _hip.rasterize_backward_batch = spy
    try:
        im0, g0 = render_step_views(params, cams, dL)      # no capacity known: stand-alone activations
        im1, g1 = render_step_views(params, cams, dL)      # capacity mode: fused
        im2, g2 = render_step_views(params, cams, dL, want_colour_grad=False)
    finally:
        _hip.rasterize_backward_batch = orig
    torch.cuda.synchronize()
    assert seen[0] is False and seen[-2] is True and seen[-1] is True, seen
    assert torch.equal(im0, im1) and torch.equal(im0, im2)
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors", "means2D"):
        assert torch.equal(g0[k], g1[k]), k
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales"):     # frozen colours: another reduction tree
        assert (g0[k] - g2[k]).abs().max().item() <= 4e-6 * g0[k].abs().max().item(), k


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "12"))))
def test_random_shapes_of_the_fused_image_loss(dev, seed):
    """Seeded random image shapes (down to one pixel row, ragged against the kernels' 16 x 16 tiles and the 11-tap window, smooth and noisy
    content) through the fused 0.8 L1 + 0.2 (1 - SSIM) kernels against the fp64 torch statement of the reference formula: value, gradient."""
    from gsdyn import losses as L
    rng = np.random.default_rng(2100 + seed)
    H, W = int(rng.choice([1, 5, 10, 11, 12, 16, 17, int(rng.integers(18, 500))])), int(rng.choice([1, 7, 11, 16, 31, 33, int(rng.integers(18, 700))]))
    smooth = bool(rng.integers(0, 2))
    base = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    if smooth:
        yy, xx = np.mgrid[0:H, 0:W]
        base = (0.5 + 0.5 * np.sin(0.07 * xx + 0.05 * yy)[None] * rng.uniform(0.2, 1.0, (3, 1, 1))).astype(np.float32)
    x = torch.tensor(np.clip(base + 0.05 * rng.normal(size=base.shape), 0, 1).astype(np.float32), device=dev, requires_grad=True)
    y = torch.tensor(base, device=dev)
    xd = x.double()
    ref = 0.8 * L.l1_loss_v1(xd, y.double()) + 0.2 * (1.0 - L.calc_ssim(xd, y.double()))
    (gref,) = torch.autograd.grad(ref * 3.0, x)
    x2 = x.detach().clone().requires_grad_(True)
    got = L.image_loss(x2, y)
    (ggot,) = torch.autograd.grad(got * 3.0, x2)
    assert abs(got.item() - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-7, (seed, H, W, smooth, got.item(), ref.item())
    assert (ggot - gref).abs().max().item() <= 2e-4 * gref.abs().max().item(), (seed, H, W, smooth, (ggot - gref).abs().max().item(), gref.abs().max().item())


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "6"))))
def test_random_sequences_of_the_direct_step(dev, seed):
    """The direct get_loss step keeps capacities from call to call and renders without a host wait.  Seeded random sequences -- the Gaussians
    grow, shrink and move between calls, by little or by a lot (overflowing the previous call's buffers), one or two cameras -- against the
    synchronous autograd step, call by call: radii bit for bit, loss and gradients within rounding."""
    from gsdyn import LossWeights, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    rng = np.random.default_rng(2900 + seed)
    P, W, H = int(rng.choice([700, 4000, 12000])), int(rng.integers(40, 280)), int(rng.integers(40, 220))
    params = synth_scene_params(P, seed=seed, device=dev, scale_lo=0.01, scale_hi=0.04)
    cams = synth_ring_cameras(2, W, H, device=dev)
    im_gt, seg_gt = synth_targets(W, H, device=dev)
    w = LossWeights()
    g = torch.Generator(device=dev).manual_seed(seed)
    for step in range(10):
        views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in range(int(rng.integers(1, 3)))]
        with torch.no_grad():
            params["log_scales"].add_(float(rng.choice([-0.8, -0.1, 0.0, 0.1, 0.9])))
            params["log_scales"].clamp_(-7.0, -1.0)
            params["means3D"].add_(float(rng.choice([0.0, 0.01, 0.2])) * torch.randn(P, 3, device=dev, generator=g))
        for p_ in params.values():
            p_.grad = None
        la, _, aux_a = get_loss_views(params, views, init_variables(P, dev), True, w, frozen_colours=True)
        la.backward()
        ga = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
        for p_ in params.values():
            p_.grad = None
        lb, _, aux_b = loss_and_grads_views(params, views, init_variables(P, dev), True, w)
        gb = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
        tag = (seed, step, P, W, H, len(views))
        assert abs(float(la.detach()) - float(lb)) <= 2e-6 * abs(float(lb)), tag
        assert torch.equal(aux_a["radii"], aux_b["radii"]), tag
        assert sorted(ga) == sorted(gb), tag
        for k in ga:
            assert (ga[k] - gb[k]).abs().max().item() <= 2e-6 * ga[k].abs().max().item() + 1e-20, tag + (k,)
