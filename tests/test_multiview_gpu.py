"""Multi-view entry points (one launch per stage for V views), fused pairs (row N1), per-view colours.
(split out of the former tests/test_hip_gpu.py; shared machinery: tests/hipcheck.py, fixtures: tests/conftest.py)"""
import os

import numpy as np
import pytest
import torch

from hipcheck import *  # noqa: F401,F403
from hipcheck import _check_against_oracle, _check_lists, _margin, _pin_tile_sort_build, _row_check, _run_hip, _settings  # noqa: F401

pytestmark = pytest.mark.gpu


def test_committed_multi_view_goldens(dev, golden_dir):
    """raster_cases_views.npz through ``rasterize_gaussians_views`` (one library call per case): a 3-camera case with shared
    colours, and a case where each camera is rendered with two colour sets (the colour + segmentation pattern of get_loss:
    views of one camera share their tile lists and are blended in one tile pass).  Per-view images / radii / depth / means2D
    gradients / colour gradients and the view-summed gradients of the other inputs against the oracle's."""
    from diff_gaussian_rasterization import rasterize_gaussians_views
    z = np.load(os.path.join(golden_dir, "raster_cases_views.npz"))
    t = lambda a, **k: torch.tensor(np.asarray(a, np.float32), device=dev, **k)  # noqa: E731
    for n in [str(x) for x in z["names"]]:
        vc, vcol = z[f"{n}/view_cam"], z[f"{n}/view_colour"]
        V = len(vc)
        by_cam = {}
        settings = []
        for vi in range(V):          # views with the same ring index get the SAME settings tensors (that is how the library
            v = z[f"{n}/cam"][vi]    # recognises a shared camera), with their own background
            if vc[vi] not in by_cam:
                cam = OracleCamera(int(v[0]), int(v[1]), float(v[2]), float(v[3]), v[4:7].astype(np.float32), 1.0,
                                   v[7:23].astype(np.float32), v[23:39].astype(np.float32), 0, v[39:42].astype(np.float32))
                by_cam[vc[vi]] = _settings(cam, dev)
            settings.append(by_cam[vc[vi]])
        inp = {k: t(z[f"{n}/in_{k}"], requires_grad=True) for k in ("means3D", "scales", "rotations", "opacities")}
        cols = z[f"{n}/in_colours"]
        per_view_col = cols.shape[0] > 1
        colours = t(cols[vcol] if per_view_col else cols[0], requires_grad=True)
        P = inp["means3D"].shape[0]
        m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        im, radii, depth = rasterize_gaussians_views(settings, inp["means3D"], m2, inp["opacities"], colors_precomp=colours,
                                                     scales=inp["scales"], rotations=inp["rotations"])
        im.backward(gradient=t(z[f"{n}/dL_dcolor"]))
        torch.cuda.synchronize()
        ok = ~z[f"{n}/ambiguous"]
        for vi in range(V):
            assert np.array_equal(radii[vi].cpu().numpy(), z[f"{n}/radii"][vi]), (n, vi)
            assert mixed_err(im[vi].detach().cpu().numpy()[:, ok[vi]], z[f"{n}/color"][vi][:, ok[vi]]) < TOL, (n, vi)
            assert mixed_err(depth[vi].detach().cpu().numpy()[:, ok[vi]], z[f"{n}/depth"][vi][:, ok[vi]]) < TOL, (n, vi)
            assert rel_err(m2.grad[vi].cpu().numpy(), z[f"{n}/grad_means2D"][vi]) < TOL, (n, vi)
            _row_check(f"golden views {n} view {vi} grad means2D", m2.grad[vi].cpu().numpy(), z[f"{n}/grad_means2D"][vi])
        gc = colours.grad.cpu().numpy()
        want_c = z[f"{n}/grad_colours_per_view"]
        assert rel_err(gc, want_c if per_view_col else want_c.sum(0)) < TOL, n
        if per_view_col:
            for vi in range(V):
                _row_check(f"golden views {n} view {vi} grad colours", gc[vi], want_c[vi])
        else:
            _row_check(f"golden views {n} grad colours (view sum)", gc, want_c.sum(0))
        for k in ("means3D", "opacities", "scales", "rotations"):
            assert rel_err(inp[k].grad.cpu().numpy(), z[f"{n}/grad_sum_{k}"]) < TOL, (n, k)
            _row_check(f"golden views {n} grad {k} (view sum)", inp[k].grad.cpu().numpy(), z[f"{n}/grad_sum_{k}"])


def test_sh_colours_through_the_multi_view_call(dev):
    """``rasterize_gaussians_views(shs=...)``: forward and backward equal per-view ``GaussianRasterizer`` calls (which are
    oracle-checked above), including a camera that appears twice.  (the batched forward used to hand the
    single-view backward states whose tile order and queue lived in the shared batch state.)"""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    P, W, H = 600, 112, 80
    g = random_gaussians(P, seed=71, scale_lo=0.03, scale_hi=0.25, sh_M=16)
    cams = [ring_camera(W, H, v=i, sh_degree=2, bg=(0.2, 0.3, 0.1)) for i in (0, 1)]
    s0, s1 = _settings(cams[0], dev), _settings(cams[1], dev)
    settings = [s0, s1, s0]          # the third view repeats the first camera
    dL = torch.tensor(np.random.default_rng(9).uniform(-1, 1, (3, 3, H, W)).astype(np.float32), device=dev)
    names = ("means3D", "opacities", "shs", "scales", "rotations")

    def leaves():
        return {k: torch.tensor(g[k], device=dev, requires_grad=True) for k in names}
    a = leaves()
    ims, m2g = [], []
    for vi, rs in enumerate(settings):
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        im, radii, depth = GaussianRasterizer(raster_settings=rs)(means3D=a["means3D"], means2D=m2, opacities=a["opacities"],
                                                                 shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
        im.backward(gradient=dL[vi])
        ims.append(im.detach()); m2g.append(m2.grad)
    b = leaves()
    m2v = torch.zeros((3, P, 3), device=dev, requires_grad=True)
    imb, radb, depb = rasterize_gaussians_views(settings, b["means3D"], m2v, b["opacities"], shs=b["shs"], scales=b["scales"],
                                                rotations=b["rotations"])
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    assert torch.equal(imb.detach(), torch.stack(ims))
    assert torch.equal(m2v.grad, torch.stack(m2g))
    for k in names:
        ga, gb = a[k].grad, b[k].grad
        assert gb is not None and torch.isfinite(gb).all(), k
        assert (ga - gb).abs().max().item() <= 2e-6 * ga.abs().max().item(), k     # same per-view values, summed in another order


def test_batched_views_equal_per_view_calls(dev):
    """rasterize_gaussians_views (per-view chains on internal streams) == V separate GaussianRasterizer calls:
    identical images / radii / depth, and input gradients equal to the sum over views."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H, V = 20000, 320, 240, 4
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(3).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)

    def leaves():
        with torch.no_grad():
            rv = params2rendervar(params)
        return {k: v.detach().clone().requires_grad_(k != "colors_precomp" or True) for k, v in rv.items()}

    a = leaves()
    ims, rads, deps, m2g = [], [], [], []
    for v in range(V):
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        im, radii, depth = GaussianRasterizer(raster_settings=cams[v])(
            means3D=a["means3D"], means2D=m2, opacities=a["opacities"], colors_precomp=a["colors_precomp"],
            scales=a["scales"], rotations=a["rotations"])
        im.backward(gradient=dL[v])
        ims.append(im.detach()); rads.append(radii); deps.append(depth.detach()); m2g.append(m2.grad)
    b = leaves()
    m2v = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    imb, radb, depb = rasterize_gaussians_views(cams, b["means3D"], m2v, b["opacities"], colors_precomp=b["colors_precomp"],
                                                scales=b["scales"], rotations=b["rotations"])
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    assert torch.equal(imb.detach(), torch.stack(ims)) and torch.equal(radb, torch.stack(rads))
    assert torch.equal(depb.detach(), torch.stack(deps))
    # (two different kernels evaluate the same per-Gaussian chain: equal up to instruction contraction)
    assert (m2v.grad - torch.stack(m2g)).abs().max().item() <= 2e-6 * torch.stack(m2g).abs().max().item()
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations"):
        ga, gb = a[k].grad, b[k].grad
        scale = ga.abs().max().item()
        # same per-view records (one blend kernel); the single-view and the multi-view preprocess backward sum them in a
        # different order and contract differently: a few ulp of the largest element (measured 2.1e-6)
        assert (ga - gb).abs().max().item() <= 4e-6 * scale, k


def test_one_call_forward_paths_and_repeated_backward(dev):
    """gsr_forward_batch: first call has no pre-sized binning buffers (falls back to the two-stage calls), the second
    one runs both stages inside the library, a shrunken capacity falls back again -- all three bit-identical.
    Backward twice over the same state (retain_graph) re-arms the blend kernel's work queue by itself."""
    from diff_gaussian_rasterization import GaussianRasterizer, _hip, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H, V = 15000, 256, 192, 3
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(4).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    key = (dev.index, P, H, W)

    def run():
        leaves = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
        m2v = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        im, radii, depth = rasterize_gaussians_views(cams, leaves["means3D"], m2v, leaves["opacities"],
                                                     colors_precomp=leaves["colors_precomp"], scales=leaves["scales"],
                                                     rotations=leaves["rotations"])
        im.backward(gradient=dL, retain_graph=True)
        g1 = {k: v.grad.clone() for k, v in leaves.items() if v.grad is not None}
        assert len(g1) >= 5
        for v in leaves.values():
            v.grad = None
        im.backward(gradient=dL)
        g2 = {k: v.grad.clone() for k, v in leaves.items() if v.grad is not None}
        torch.cuda.synchronize()
        for k in g1:
            assert torch.equal(g1[k], g2[k]), f"second backward differs: {k}"
        return im.detach(), radii, depth.detach(), g1

    _hip._binning_capacity.pop(key, None)
    first = run()                                   # no capacity yet: two-stage fallback
    assert _hip._binning_capacity.get(key, 0) > 0
    second = run()                                  # both stages inside gsr_forward_batch
    _hip._binning_capacity[key] = 4096              # far too small: fallback again, capacity re-learnt
    third = run()
    assert _hip._binning_capacity[key] > 4096
    for other in (second, third):
        assert torch.equal(first[0], other[0]) and torch.equal(first[1], other[1]) and torch.equal(first[2], other[2])
        for k in first[3]:
            assert torch.equal(first[3][k], other[3][k]), k
    # single-view path: backward twice as well
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    im, _, _ = GaussianRasterizer(raster_settings=cams[0])(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                           colors_precomp=leaves["colors_precomp"], scales=leaves["scales"],
                                                           rotations=leaves["rotations"])
    im.backward(gradient=dL[0], retain_graph=True)
    ga = leaves["means3D"].grad.clone()
    leaves["means3D"].grad = None
    im.backward(gradient=dL[0])
    assert torch.equal(ga, leaves["means3D"].grad)


def test_strided_camera_tensors_are_converted_once_and_tracked(dev):
    """The reference's setup_camera passes transposed / column views; the wrapper caches their contiguous copies per
    tensor object + version, so an in-place camera update must still be picked up."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    g = random_gaussians(500, seed=5, scale_lo=0.03, scale_hi=0.2)
    cam = ring_camera(64, 48, v=0)
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)  # noqa: E731
    vm_store = t(cam.viewmatrix).reshape(4, 4).t().contiguous()      # holds the transpose; .t() view = the matrix
    pm_store = t(cam.projmatrix).reshape(4, 4).t().contiguous()
    def settings():
        return GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, t(cam.bg), 1.0,
                                             vm_store.t().unsqueeze(0), pm_store.t().unsqueeze(0), 0, t(cam.campos), False)
    rs = settings()
    assert not rs.viewmatrix.is_contiguous()
    x = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    def render(rs_):
        return GaussianRasterizer(raster_settings=rs_)(means3D=x["means3D"], means2D=torch.zeros_like(x["means3D"]),
                                                       opacities=x["opacities"], colors_precomp=x["colors_precomp"],
                                                       scales=x["scales"], rotations=x["rotations"])[0]
    a = render(rs)
    ref = _run_hip(cam, g, dev)[0]
    assert np.array_equal(a.cpu().numpy(), ref)
    assert torch.equal(render(rs), a)                                  # cached conversion
    cam2 = ring_camera(64, 48, v=1)
    vm_store.copy_(t(cam2.viewmatrix).reshape(4, 4).t()); pm_store.copy_(t(cam2.projmatrix).reshape(4, 4).t())
    rs2 = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, rs.bg, 1.0,
                                        rs.viewmatrix, rs.projmatrix, 0, t(cam2.campos), False)   # same view objects, new content
    b = render(rs2)
    assert np.array_equal(b.cpu().numpy(), _run_hip(cam2, g, dev)[0])


def test_per_view_colours_share_geometry(dev):
    """Row N1: the colour and the segmentation render of a camera as ONE 2-view call with per-view colours ==
    two separate GaussianRasterizer calls: identical images, colour gradients per view, geometry gradients summed."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H = 12000, 240, 176
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(2, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(9).uniform(-1, 1, (4, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    cols = torch.stack([rv["colors_precomp"], params["seg_colors"].detach()])             # [2,P,3]

    a = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    ca = cols.clone().requires_grad_(True)
    ims = []
    for v in range(4):          # views 0,1: camera 0 colour / seg; views 2,3: camera 1
        im, _, _ = GaussianRasterizer(raster_settings=cams[v // 2])(
            means3D=a["means3D"], means2D=torch.zeros((P, 3), device=dev, requires_grad=True), opacities=a["opacities"],
            colors_precomp=ca[v % 2], scales=a["scales"], rotations=a["rotations"])
        im.backward(gradient=dL[v])
        ims.append(im.detach())
    b = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    cb = cols.repeat(2, 1, 1).clone().requires_grad_(True)                                # [4,P,3]
    m2 = torch.zeros((4, P, 3), device=dev, requires_grad=True)
    from diff_gaussian_rasterization import _hip
    seen_states = {}
    orig = _hip.rasterize_forward_batch

    def spy(*a_, **k_):
        out = orig(*a_, **k_)
        seen_states["s"] = out[3]
        return out
    _hip.rasterize_forward_batch = spy
    try:
        imb, radb, _ = rasterize_gaussians_views([cams[0], cams[0], cams[1], cams[1]], b["means3D"], m2, b["opacities"],
                                                 colors_precomp=cb, scales=b["scales"], rotations=b["rotations"])
    finally:
        _hip.rasterize_forward_batch = orig
    st = seen_states["s"]   # views 1 and 3 have the cameras of views 0 and 2: they own no binning state (shared tile lists)
    assert list(st[0].geometry_of) == [0, 0, 2, 2] and st[1].binning is None and st[3].binning is None
    assert st[0].binning is not None and st[1].num_rendered == st[0].num_rendered
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    assert torch.equal(imb.detach(), torch.stack(ims))
    assert torch.equal(radb[0], radb[1]) and torch.equal(radb[2], radb[3])
    gcb = cb.grad
    assert torch.equal(gcb[0] + gcb[2], ca.grad[0]) or (gcb[0] + gcb[2] - ca.grad[0]).abs().max() <= 2e-6 * ca.grad[0].abs().max()
    assert (gcb[1] + gcb[3] - ca.grad[1]).abs().max() <= 2e-6 * ca.grad[1].abs().max()
    for k in ("means3D", "opacities", "scales", "rotations"):
        ga, gb = a[k].grad, b[k].grad
        assert (ga - gb).abs().max().item() <= 2e-6 * ga.abs().max().item(), k


def test_fused_pair_backward_with_frozen_colours(dev):
    """Views that share a camera are blended in ONE tile pass (6 channels); with frozen colours (tracking: lr 0) the backward
    stays fused too -- one replay of the lists driven by both views' dL/dcolour.  Against one GaussianRasterizer call per
    view: identical images, summed geometry gradients, and the per-view screen-space gradients (densification reads the colour
    render's alone).  Camera 0 is used three times (pair + a plain alias), camera 1 twice, camera 2 once."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H = 12000, 240, 176
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(3, W, H, device=dev)
    cam_of = [0, 0, 1, 1, 0, 2]
    V = len(cam_of)
    rng = np.random.default_rng(19)
    dL = torch.tensor(rng.uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    cols = torch.tensor(rng.uniform(0, 1, (V, P, 3)).astype(np.float32), device=dev)          # frozen: no gradient
    cols[0], cols[1] = rv["colors_precomp"], params["seg_colors"].detach()
    a = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    ims, m2s = [], []
    for v in range(V):
        holder = torch.zeros((P, 3), device=dev, requires_grad=True)
        im, _, _ = GaussianRasterizer(raster_settings=cams[cam_of[v]])(
            means3D=a["means3D"], means2D=holder, opacities=a["opacities"], colors_precomp=cols[v], scales=a["scales"],
            rotations=a["rotations"])
        im.backward(gradient=dL[v])
        ims.append(im.detach())
        m2s.append(holder.grad.clone())
    b = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    imb, radb, depb = rasterize_gaussians_views([cams[c] for c in cam_of], b["means3D"], m2, b["opacities"], colors_precomp=cols,
                                                scales=b["scales"], rotations=b["rotations"])
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    assert torch.equal(imb.detach(), torch.stack(ims))
    assert torch.equal(depb[0], depb[1]) and torch.equal(depb[0], depb[4]) and torch.equal(radb[2], radb[3])
    for k in ("means3D", "opacities", "scales", "rotations"):
        ga, gb = a[k].grad, b[k].grad
        assert (ga - gb).abs().max().item() <= 5e-6 * ga.abs().max().item(), k
    for v in range(V):
        want = m2s[v]
        assert (m2.grad[v] - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-12, v
    # a second backward over the same graph state gives the same bits (queue re-armed, order rebuilt)
    b2 = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    m2b = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    imb2, _, _ = rasterize_gaussians_views([cams[c] for c in cam_of], b2["means3D"], m2b, b2["opacities"], colors_precomp=cols,
                                           scales=b2["scales"], rotations=b2["rotations"])
    imb2.backward(gradient=dL, retain_graph=True)
    g1 = {k: v.grad.clone() for k, v in b2.items()}
    for v_ in b2.values():
        v_.grad = None
    imb2.backward(gradient=dL)
    for k in g1:
        assert torch.equal(g1[k], b2[k].grad) and torch.equal(g1[k], b[k].grad), k


_VARIANT_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(sys.argv[1], "gs-dynamics_amd"))
from diff_gaussian_rasterization import rasterize_gaussians_views
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
P, W, H = 6000, 203, 117                      # image size not a multiple of the tile
params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.08)
cams = synth_ring_cameras(2, W, H, device=dev)
cam_b = cams[0]._replace(bg=torch.tensor([0.2, 0.5, 0.9], device=dev))   # the partner has its own background
rng = np.random.default_rng(5)
with torch.no_grad():
    rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
cols = torch.tensor(rng.uniform(0, 1, (3, P, 3)).astype(np.float32), device=dev)
dL = torch.tensor(rng.uniform(-1, 1, (3, 3, H, W)).astype(np.float32), device=dev)
b = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
m2 = torch.zeros((3, P, 3), device=dev, requires_grad=True)
im, rad, dep = rasterize_gaussians_views([cams[0], cam_b, cams[1]], b["means3D"], m2, b["opacities"], colors_precomp=cols,
                                         scales=b["scales"], rotations=b["rotations"])
im.backward(gradient=dL)
torch.cuda.synchronize()
np.savez(sys.argv[2], im=im.detach().cpu().numpy(), dep=dep.detach().cpu().numpy(), m2=m2.grad.cpu().numpy(),
         **{"g_" + k: v.grad.cpu().numpy() for k, v in b.items()})
"""


def test_pair_fusion_against_per_view_passes(dev, tmp_path):
    """The fused pair pass against the same call with GSR_NO_PAIR_FUSION=1 (one pass per view); the partner view has a different
    background, the image size is not a multiple of 16.  Images are identical bit for bit; gradients: fused vs per-view passes within
    rounding.  (Round 5 removed the static-launch, counted-rows and no-used-flags arms together with their switches.)"""
    import subprocess
    import sys
    script = tmp_path / "variant.py"
    script.write_text(_VARIANT_SCRIPT)
    outs = {}
    for name, env in (("default", {}), ("nopair", {"GSR_NO_PAIR_FUSION": "1"})):
        e = dict(os.environ)
        e.update(env)
        out = tmp_path / (name + ".npz")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    d, n = outs["default"], outs["nopair"]
    assert np.array_equal(d["im"], n["im"]) and np.array_equal(d["dep"], n["dep"])
    assert float(np.abs(d["im"][1] - d["im"][0]).max()) > 0.1            # different colours and background
    for k in d.files:
        if k.startswith("g_") or k == "m2":
            assert np.abs(d[k] - n[k]).max() <= 2e-5 * np.abs(n[k]).max() + 1e-12, k


def test_more_views_than_one_library_call(dev):
    """18 views (> GSR_MAX_BATCH = 16): the Python entry point splits the call; results equal per-view calls."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H, V = 3000, 96, 64, 18
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.1)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(2).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items() if k != "means2D"}
    a = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    ims = []
    for v in range(V):
        im, _, _ = GaussianRasterizer(raster_settings=cams[v])(means2D=torch.zeros((P, 3), device=dev), **a)
        im.backward(gradient=dL[v])
        ims.append(im.detach())
    b = {k: v.clone().requires_grad_(True) for k, v in rv.items()}
    m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    imb, radb, depb = rasterize_gaussians_views(cams, b["means3D"], m2, b["opacities"], colors_precomp=b["colors_precomp"],
                                                scales=b["scales"], rotations=b["rotations"])
    imb.backward(gradient=dL)
    assert imb.shape == (V, 3, H, W) and radb.shape == (V, P) and depb.shape == (V, 1, H, W)
    assert torch.equal(imb.detach(), torch.stack(ims))
    assert m2.grad is not None and m2.grad.shape == (V, P, 3)
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations"):
        ga, gb = a[k].grad, b[k].grad
        assert (ga - gb).abs().max().item() <= 4e-6 * ga.abs().max().item(), k


def test_forward_only_renderer_colour_and_mask_in_one_call(dev):
    """Row A11: the predict.py pattern (colour render + all-ones mask render per camera) as one multi-view call equals the
    reference-shaped two calls per camera."""
    from gsdyn import params2rendervar, synth_scene_params
    from gsdyn.camera import look_at_w2c
    from gsdyn.render import Renderer
    P = 8000
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.05)
    with torch.no_grad():
        data = {k: v.detach() for k, v in params2rendervar(params).items()}
    r = Renderer(dev, w=320, h=180)
    k = np.array([[300.0, 0, 160], [0, 300.0, 90], [0, 0, 1]])
    cams = [(look_at_w2c(np.array([3.5 * np.cos(a), 0.6, 3.5 * np.sin(a)]), np.zeros(3)), k) for a in (0.3, 1.9)]
    ims, depths, masks = r.render_cameras_with_mask(cams, data, bg=(0.0, 0.0, 0.0))                                  # mask = 1 - final_T
    ims2, depths2, masks2 = r.render_cameras_with_mask(cams, data, bg=(0.0, 0.0, 0.0), mask_from_alpha=False)       # mask blended (fused pair)
    ones = dict(data)
    ones["colors_precomp"] = torch.ones_like(data["colors_precomp"])
    worst = 0.0
    for i, (w2c, kk) in enumerate(cams):
        im, depth = r.render(w2c, kk, data, bg=(0.0, 0.0, 0.0))
        mask, _ = r.render(w2c, kk, ones, bg=(0.0, 0.0, 0.0))
        assert torch.equal(ims[i], im) and torch.equal(depths[i], depth)
        assert torch.equal(ims2[i], im) and torch.equal(depths2[i], depth) and torch.equal(masks2[i], mask)
        # sum_i alpha_i T_i (the second render) against 1 - prod (1 - alpha_i) (the colour render's final transmittance): the same
        # number up to the fp32 rounding of the two evaluation orders
        assert masks[i].shape == mask.shape and float(mask.max()) > 0.5
        worst = max(worst, float((masks[i] - mask).abs().max()))
    _margin("mask_from_alpha_vs_second_render", worst, 2e-5)
    assert worst <= 2e-5
    grey = (0.25, 0.5, 0.75)                       # a background: mask_ch = (1 - T) + T bg_ch
    _, _, mg = r.render_cameras_with_mask(cams[:1], data, bg=grey)
    mref, _ = r.render(cams[0][0], k, ones, bg=grey)
    assert float((mg[0] - mref).abs().max()) <= 2e-5
    a, d, m = r.render_with_mask(cams[0][0], k, data)
    assert torch.equal(a, ims[0]) and torch.equal(m, masks[0]) and float(m.max()) <= 1.0 + 1e-5


def test_batch_with_a_view_that_sees_nothing(dev):
    """One camera of a multi-view call looks away from the scene (no entries at all for it): background image, zero depth,
    and the summed gradients equal those of the other views alone."""
    from diff_gaussian_rasterization import rasterize_gaussians_views
    from gsdyn import params2rendervar, setup_camera, synth_ring_cameras, synth_scene_params
    from gsdyn.camera import look_at_w2c
    P, W, H = 4000, 128, 96
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.08)
    cams = synth_ring_cameras(2, W, H, device=dev)
    k = np.array([[float(W), 0, W / 2], [0, float(W), H / 2], [0, 0, 1]])
    away = setup_camera(W, H, k, look_at_w2c(np.array([4.0, 0.8, 0.0]), np.array([8.0, 0.8, 0.0])), bg=(0.2, 0.4, 0.6), device=dev)
    dL = torch.tensor(np.random.default_rng(6).uniform(-1, 1, (3, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {kk: v.detach().clone() for kk, v in params2rendervar(params).items() if kk != "means2D"}

    def run(views, g):
        leaves = {kk: v.clone().requires_grad_(True) for kk, v in rv.items()}
        m2 = torch.zeros((len(views), P, 3), device=dev, requires_grad=True)
        im, radii, depth = rasterize_gaussians_views(views, leaves["means3D"], m2, leaves["opacities"], colors_precomp=leaves["colors_precomp"],
                                                     scales=leaves["scales"], rotations=leaves["rotations"])
        im.backward(gradient=g)
        return im.detach(), radii, depth.detach(), {kk: v.grad for kk, v in leaves.items()}, m2.grad
    im3, rad3, dep3, g3, m3 = run([cams[0], away, cams[1]], dL)
    im2, rad2, dep2, g2, m2_ = run([cams[0], cams[1]], dL[[0, 2]])
    assert int((rad3[1] > 0).sum()) == 0 and torch.all(dep3[1] == 0)
    assert torch.allclose(im3[1], torch.tensor([0.2, 0.4, 0.6], device=dev)[:, None, None].expand(3, H, W))
    assert torch.equal(im3[0], im2[0]) and torch.equal(im3[2], im2[1]) and torch.all(m3[1] == 0)
    for kk in g2:
        assert (g3[kk] - g2[kk]).abs().max().item() <= 1e-6 * g2[kk].abs().max().item(), kk


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "12"))))      # GSR_MV_SOAK=N: a longer hunt
def test_random_batches_equal_per_view_calls(dev, seed):
    """Seeded random scenes through the multi-view call against V separate drop-in calls: Gaussian count, ragged image sizes, view count,
    camera distance (inside the cloud included) and Gaussian size vary.  Images, radii and depth bit for bit; gradients = the sum over the
    views up to the summation order of the two per-Gaussian backward kernels."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    rng = np.random.default_rng(500 + seed)
    P = int(rng.choice([1, 60, 900, 6000, 30000]))
    W, H, V = int(rng.integers(17, 420)), int(rng.integers(17, 300)), int(rng.integers(1, 6))
    lo = float(rng.choice([0.004, 0.02, 0.08]))
    params = synth_scene_params(P, seed=seed, device=dev, scale_lo=lo, scale_hi=lo * float(rng.choice([1.5, 6.0])))
    cams = synth_ring_cameras(V, W, H, device=dev, radius=float(rng.choice([0.6, 2.5, 5.0])), height=float(rng.choice([-0.5, 0.8])))
    dL = torch.tensor(rng.uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)

    def leaves():
        with torch.no_grad():
            rv = params2rendervar(params)
        return {k: v.detach().clone().requires_grad_(True) for k, v in rv.items()}

    a = leaves()
    ims, rads, deps, m2g = [], [], [], []
    for v in range(V):
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        im, radii, depth = GaussianRasterizer(raster_settings=cams[v])(
            means3D=a["means3D"], means2D=m2, opacities=a["opacities"], colors_precomp=a["colors_precomp"], scales=a["scales"], rotations=a["rotations"])
        im.backward(gradient=dL[v])
        ims.append(im.detach()); rads.append(radii); deps.append(depth.detach()); m2g.append(m2.grad if m2.grad is not None else torch.zeros_like(m2))  # noqa: E702
    b = leaves()
    m2v = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    imb, radb, depb = rasterize_gaussians_views(cams, b["means3D"], m2v, b["opacities"], colors_precomp=b["colors_precomp"], scales=b["scales"],
                                                rotations=b["rotations"])
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    tag = (seed, P, W, H, V)
    assert torch.equal(imb.detach(), torch.stack(ims)) and torch.equal(radb, torch.stack(rads)) and torch.equal(depb.detach(), torch.stack(deps)), tag
    ref = torch.stack(m2g)
    assert (m2v.grad - ref).abs().max().item() <= 4e-6 * max(ref.abs().max().item(), 1e-30), tag
    for k in ("means3D", "opacities", "colors_precomp", "scales", "rotations"):
        ga = a[k].grad if a[k].grad is not None else torch.zeros_like(a[k])
        gb = b[k].grad if b[k].grad is not None else torch.zeros_like(b[k])
        assert (ga - gb).abs().max().item() <= 1e-5 * max(ga.abs().max().item(), 1e-30), (k,) + tag


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "12"))))
def test_random_fused_pairs_equal_separate_renders(dev, seed):
    """Row N1 on seeded random scenes: C cameras x (colour, segmentation) as ONE call with per-view colours (the second view of a camera is
    blended inside the first one's tile pass) against 2 C separate drop-in calls -- images bit for bit, colour gradients per colour set,
    geometry gradients summed (up to the summation order)."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    rng = np.random.default_rng(900 + seed)
    P = int(rng.choice([1, 60, 900, 6000, 30000]))
    W, H, C = int(rng.integers(17, 420)), int(rng.integers(17, 300)), int(rng.integers(1, 4))
    lo = float(rng.choice([0.004, 0.02, 0.08]))
    params = synth_scene_params(P, seed=seed, device=dev, scale_lo=lo, scale_hi=lo * float(rng.choice([1.5, 6.0])))
    cams = synth_ring_cameras(C, W, H, device=dev, radius=float(rng.choice([0.6, 2.5, 5.0])), height=float(rng.choice([-0.5, 0.8])))
    dL = torch.tensor(rng.uniform(-1, 1, (2 * C, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    cols = torch.stack([rv["colors_precomp"], params["seg_colors"].detach()])             # [2,P,3]
    zero = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)                # noqa: E731
    a = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    ca = cols.clone().requires_grad_(True)
    ims = []
    for v in range(2 * C):
        im, _, _ = GaussianRasterizer(raster_settings=cams[v // 2])(
            means3D=a["means3D"], means2D=torch.zeros((P, 3), device=dev, requires_grad=True), opacities=a["opacities"],
            colors_precomp=ca[v % 2], scales=a["scales"], rotations=a["rotations"])
        im.backward(gradient=dL[v])
        ims.append(im.detach())
    b = {k: v.clone().requires_grad_(True) for k, v in rv.items() if k not in ("colors_precomp", "means2D")}
    cb = cols.repeat(C, 1, 1).clone().requires_grad_(True)                                # [2C,P,3]
    m2 = torch.zeros((2 * C, P, 3), device=dev, requires_grad=True)
    imb, _, _ = rasterize_gaussians_views([cams[v // 2] for v in range(2 * C)], b["means3D"], m2, b["opacities"], colors_precomp=cb,
                                          scales=b["scales"], rotations=b["rotations"])
    imb.backward(gradient=dL)
    torch.cuda.synchronize()
    tag = (seed, P, W, H, C)
    assert torch.equal(imb.detach(), torch.stack(ims)), tag
    gcb, gca = zero(cb), zero(ca)
    for s in range(2):
        assert (gcb[s::2].sum(0) - gca[s]).abs().max().item() <= 1e-5 * max(gca[s].abs().max().item(), 1e-30), (s,) + tag
    for k in ("means3D", "opacities", "scales", "rotations"):
        ga, gb = zero(a[k]), zero(b[k])
        assert (ga - gb).abs().max().item() <= 1e-5 * max(ga.abs().max().item(), 1e-30), (k,) + tag


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "12"))))
def test_random_forward_only_frames_equal_the_reference_shaped_calls(dev, seed):
    """Row A11 on seeded random scenes (dense ones included: the forward-only call picks its sort / binning builds from the list lengths): the
    predict.py frame -- C cameras x (colour + mask) as one forward-only call -- against two reference-shaped renders per camera."""
    from gsdyn import params2rendervar, synth_scene_params
    from gsdyn.predict import ring_poses
    from gsdyn.render import Renderer
    rng = np.random.default_rng(1300 + seed)
    P = int(rng.choice([1, 300, 5000, 40000, 150000]))
    W, H, C = int(rng.integers(17, 700)), int(rng.integers(17, 420)), int(rng.integers(1, 5))
    lo = float(rng.choice([0.004, 0.02, 0.06]))
    params = synth_scene_params(P, seed=seed, device=dev, scale_lo=lo, scale_hi=lo * float(rng.choice([1.5, 4.0])))
    with torch.no_grad():
        data = {k: v.detach() for k, v in params2rendervar(params).items()}
        data["opacities"] = data["opacities"].clamp_min(float(rng.choice([0.0, 0.5])))
    r = Renderer(dev, w=W, h=H)
    cams = ring_poses(C, W, H)
    bg = tuple(float(x) for x in rng.choice([0.0, 0.3], 3))
    ims, depths, masks = r.render_cameras_with_mask(cams, data, bg=bg)
    ims2, depths2, masks2 = r.render_cameras_with_mask(cams, data, bg=bg, mask_from_alpha=False)
    ones = dict(data)
    ones["colors_precomp"] = torch.ones_like(data["colors_precomp"])
    tag = (seed, P, W, H, C)
    for i, (w2c, kk) in enumerate(cams):
        im, depth = r.render(w2c, kk, data, bg=bg)
        mask, _ = r.render(w2c, kk, ones, bg=bg)
        assert torch.equal(ims[i], im) and torch.equal(depths[i], depth), tag + (i,)
        assert torch.equal(ims2[i], im) and torch.equal(depths2[i], depth) and torch.equal(masks2[i], mask), tag + (i,)
        assert float((masks[i] - mask).abs().max()) <= 3e-5, tag + (i, float((masks[i] - mask).abs().max()))
