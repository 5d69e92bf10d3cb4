"""GPU box: error margins of tests/test_losses_step_gpu.py::test_full_size_direct_step_against_literal_torch_step over several seeds."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import GaussianRasterizer
from gsdyn import LossWeights, loss_and_grads_views, params2rendervar, synth_ring_cameras, synth_scene_params, synth_targets
from gsdyn import losses as L
from gsdyn.dp import init_variables
from gsdyn.step import _shared_terms, make_rigidity_variables
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
for seed in range(2):
    torch.manual_seed(seed)
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
        params["means3D"].add_(0.003 * torch.randn_like(params["means3D"]))
        params["unnorm_rotations"].add_(0.02 * torch.randn_like(params["unnorm_rotations"]))
    for p_ in params.values():
        p_.grad = None
    v1 = init_variables(P, dev); v1.update(rig)
    loss_f, _, aux = loss_and_grads_views(params, views, v1, False, w)
    g_f = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
    for p_ in params.values():
        p_.grad = None
    torch_vars = {k: v for k, v in rig.items() if k not in ("rev_ptr", "rev_edge")}
    weights = dict(rigid=w.rigid, rot=w.rot, iso=w.iso, floor=w.floor, bg=w.bg)
    total = 0.0
    for d in views:
        rv = params2rendervar(params)
        im, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**rv)
        im = torch.exp(params["cam_m"][d["id"]])[:, None, None] * im + params["cam_c"][d["id"]][:, None, None]
        l_im = 0.8 * L.l1_loss_v1(im, d["im"]) + 0.2 * (1.0 - L.calc_ssim(im, d["im"]))
        sv = params2rendervar(params, colors_key="seg_colors")
        seg, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**sv)
        l_seg = 0.8 * L.l1_loss_v1(seg, d["seg"]) + 0.2 * (1.0 - L.calc_ssim(seg, d["seg"]))
        shared, _ = _shared_terms(params, rv, torch_vars, weights)
        loss = w.im * l_im + w.seg * l_seg + shared
        loss.backward(); total += float(loss.detach())
    errs = {k: ((g_f[k] - params[k].grad).abs().max() / params[k].grad.abs().max()).item() for k in g_f}
    print(seed, "loss rel", abs(float(loss_f) - total) / abs(total), {k: f"{v:.1e}" for k, v in errs.items()})
    # the shared terms alone, in fp32 and in fp64, through their own graph
    g_all = {k: params[k].grad.clone() for k in g_f}
    for tag, cast in (("fp32", lambda t: t), ("fp64", lambda t: t.double())):
        pp = {k: cast(v.detach()).requires_grad_(True) for k, v in params.items()}
        tv = {k: (cast(v) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in torch_vars.items()}
        sh, _ = _shared_terms(pp, params2rendervar(pp), tv, weights)
        (2.0 * sh).backward()
        for k in ("means3D", "unnorm_rotations"):
            print("   shared-only", tag, k, "max|g|", pp[k].grad.abs().max().item(), "share of total max", (pp[k].grad.abs().max() / g_all[k].abs().max()).item())
    # raster + image part alone (no shared terms in the graph)
    for p_ in params.values():
        p_.grad = None
    for d in views:
        rv = params2rendervar(params)
        im, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**rv)
        im = torch.exp(params["cam_m"][d["id"]])[:, None, None] * im + params["cam_c"][d["id"]][:, None, None]
        l_im = 0.8 * L.l1_loss_v1(im, d["im"]) + 0.2 * (1.0 - L.calc_ssim(im, d["im"]))
        sv = params2rendervar(params, colors_key="seg_colors")
        seg, _, _ = GaussianRasterizer(raster_settings=d["cam"])(**sv)
        l_seg = 0.8 * L.l1_loss_v1(seg, d["seg"]) + 0.2 * (1.0 - L.calc_ssim(seg, d["seg"]))
        (w.im * l_im + w.seg * l_seg).backward()
    for k in ("means3D", "unnorm_rotations"):
        resid = g_all[k] - params[k].grad          # = what the shared terms contributed inside the joint graph
        print("   joint-minus-raster vs shared-only fp64", k, ((resid.double() - pp[k].grad).abs().max() / g_all[k].abs().max()).item())
