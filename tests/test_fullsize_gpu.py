"""BASELINE.json's sizes: configs[1] / [2] / [3] / [4] against the oracle and through size-independent properties; the function bench.py times.
(split out of the former tests/test_hip_gpu.py; shared machinery: tests/hipcheck.py, fixtures: tests/conftest.py)"""
import os

import numpy as np
import pytest
import torch

from hipcheck import *  # noqa: F401,F403
from hipcheck import _check_against_oracle, _check_lists, _margin, _pin_tile_sort_build, _row_check, _run_hip, _settings  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,backward", [(100_000, True), (50_000, False)])
def test_full_size_matches_oracle_one_view(dev, full_scene, P, backward):
    """BASELINE configs[2] (100k Gaussians, 800x800, forward + backward) and configs[1] (50k Gaussians, one 800x800 view,
    forward only, no autograd graph), view 0, against the (threaded) oracle."""
    params, cams, p2r = full_scene
    if P != 100_000:
        from gsdyn import synth_scene_params
        params = synth_scene_params(P, device=dev)
    with torch.no_grad():
        rv = {k: v.detach().cpu().numpy() for k, v in p2r(params).items()}
    cam = cams[0]
    ocam = OracleCamera(800, 800, cam.tanfovx, cam.tanfovy, cam.bg.cpu().numpy(), 1.0,
                        cam.viewmatrix.cpu().numpy().reshape(-1), cam.projmatrix.cpu().numpy().reshape(-1), 0,
                        cam.campos.cpu().numpy())
    g = dict(means3D=rv["means3D"], scales=rv["scales"], rotations=rv["rotations"], opacities=rv["opacities"],
             colors_precomp=rv["colors_precomp"])
    o2 = _check_against_oracle(ocam, g, dev, seed=11, nthreads=os.cpu_count() or 8, backward=backward)
    print("num_rendered", o2.num_rendered, "ambiguous px", int(o2.ambiguous.sum()))


def test_full_size_properties(dev, full_scene):
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    params, cams, p2r = full_scene
    W = H = 800
    dL = torch.tensor(np.random.default_rng(5).uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev)

    def run(cam, scale=1.0, colors=None, bg=None):
        rv = p2r(params)
        rv = {k: v.detach().requires_grad_(True) for k, v in rv.items()}
        if colors is not None:
            rv["colors_precomp"] = colors
        if bg is not None:
            cam = cam._replace(bg=torch.tensor(bg, device=dev, dtype=torch.float32))
        im, radii, depth = GaussianRasterizer(raster_settings=cam)(**rv)
        (im * (dL * scale)).sum().backward()
        return im.detach(), radii, depth.detach(), {k: v.grad for k, v in rv.items() if v.grad is not None}

    im1, rad1, dep1, g1 = run(cams[1])
    im2, rad2, dep2, g2 = run(cams[1])
    # determinism: no atomics anywhere -> bit-identical reruns
    assert torch.equal(im1, im2) and torch.equal(dep1, dep2) and torch.equal(rad1, rad2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    # linearity of the backward in the incoming gradient (exact for a power-of-two scale)
    _, _, _, g4 = run(cams[1], scale=4.0)
    for k in g1:
        assert torch.equal(g1[k] * 4.0, g4[k]), k
    # partition of unity: colours == 1 and background == 1  =>  every pixel renders 1
    ones = torch.ones_like(params["rgb_colors"])
    im_one, _, _, _ = run(cams[2], colors=ones, bg=(1.0, 1.0, 1.0))
    assert (im_one - 1.0).abs().max().item() < 2e-5
    # sortedness / partition of the tile lists
    rv = p2r(params)
    orig = _hip.rasterize_forward
    st = {}

    def spy(*a, **k):
        out = orig(*a, **k)
        st["s"] = out[3]
        return out
    _hip.rasterize_forward = spy
    try:
        with torch.no_grad():
            GaussianRasterizer(raster_settings=cams[3])(**rv)
    finally:
        _hip.rasterize_forward = orig
    v = _hip.debug_views(st["s"])
    ranges, pl, depth_g = v["ranges"].long(), v["point_list"].long(), v["rec"][:, 9]
    D = st["s"].num_rendered
    lens = ranges[:, 1] - ranges[:, 0]
    assert int(lens.sum()) == D and int(v["offsets"][-1]) == D
    nz = lens > 0
    starts = ranges[nz, 0].sort().values
    assert starts[0] == 0 and torch.equal(starts[1:], (ranges[nz, 1].sort().values)[:-1])
    tile_of = torch.repeat_interleave(torch.arange(ranges.shape[0], device=dev), lens.clamp(min=0))
    order = torch.argsort(ranges[:, 0].masked_fill(~nz, 2 ** 40), stable=True)
    tile_sorted = torch.repeat_interleave(order[: int(nz.sum())], lens[order[: int(nz.sum())]])
    d = depth_g[pl]
    same = tile_sorted[1:] == tile_sorted[:-1]
    assert torch.all((d[1:] >= d[:-1]) | ~same), "per-tile depth order violated"
    tie = same & (d[1:] == d[:-1])
    assert torch.all((pl[1:] > pl[:-1]) | ~tie), "depth ties must keep ascending Gaussian index"
    del tile_of


def test_config5_size_forward(dev):
    """BASELINE config 5 sizes: 500k Gaussians, 1920x1080 (T = 8160 tiles, 13 tile-id bits -> 7+6-bit passes),
    forward only, against the threaded oracle: exact radii / lists, colour and depth within tolerance."""
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    P, W, H = 500_000, 1920, 1080
    g = random_gaussians(P, seed=5, scale_lo=0.004, scale_hi=0.02, spread=1.2)
    cam = ring_camera(W, H, v=2, bg=(0.0, 0.0, 0.0))
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    rs = _settings(cam, dev)
    st = {}
    orig = _hip.rasterize_forward

    def spy(*a, **k):
        out = orig(*a, **k)
        st["s"] = out[3]
        return out
    _hip.rasterize_forward = spy
    try:
        with torch.no_grad():
            color, radii, depth = GaussianRasterizer(raster_settings=rs)(
                means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"],
                colors_precomp=t["colors_precomp"], scales=t["scales"], rotations=t["rotations"])
    finally:
        _hip.rasterize_forward = orig
    o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                     rotations=g["rotations"], nthreads=min(64, os.cpu_count() or 8))
    ok = ~o2.ambiguous
    v = _hip.debug_views(st["s"])
    assert np.array_equal(radii.cpu().numpy(), o2.radii)
    _check_lists(v, cam.image_height, cam.image_width, o2.point_list, o2.ranges, o2.n_contrib, ok, o2.means2D,
                 o2.conic_opacity, o2.tiles_touched, o2.offsets)
    assert mixed_err(color.cpu().numpy()[:, ok], o2.color[:, ok]) < TOL
    assert mixed_err(depth.cpu().numpy()[:, ok], o2.depth[:, ok]) < TOL
    print("config5: num_rendered", o2.num_rendered, "ambiguous px", int(o2.ambiguous.sum()))


def test_config5_frame_as_bench_times_it(dev):
    """What ``bench.py --config 5`` times, at its size and with its knobs: ``Renderer.render_cameras_with_mask`` = ONE forward-only
    multi-view call for predict.py's four cameras (/root/reference/src/predict.py:100-123) on 500k Gaussians at 1920x1080, handed
    over in Morton order (``spatial_order``, as ``collect_scene_data`` does per episode): tile-row binning at T = 8160, the dense-scene
    per-tile sorts (wave tickets with 32 keys per lane, long tickets on the 2048-entry block), GSR_FORWARD_ONLY, the mask from the
    colour render's final transmittance.  Per camera against oracle O2 on the same (permuted) arrays: radii and tile lists bit-exact,
    colour / depth / final_T within tolerance, mask = 1 - final_T of the oracle."""
    from diff_gaussian_rasterization import _hip
    from gsdyn.dynamics import spatial_order
    from gsdyn.predict import ring_poses
    from gsdyn.render import Renderer
    P, W, H, CAMS = 500_000, 1920, 1080, 4
    from gsdyn import params2rendervar, synth_scene_params
    with torch.no_grad():       # bench_config5's scene: SynthScene-v1 at 500k (D = 6.0 M entries per camera, lists up to ~2000)
        data_in = {k: v.detach() for k, v in params2rendervar(synth_scene_params(P, seed=0, device=dev)).items()}
    perm = spatial_order(data_in["means3D"])
    assert sorted(perm.cpu().tolist()) == list(range(P))
    data = {k: v[perm].contiguous() for k, v in data_in.items()}
    g = {k: data[k].cpu().numpy() for k in ("means3D", "colors_precomp", "rotations", "opacities", "scales")}
    rdr = Renderer(dev, w=W, h=H)
    poses = ring_poses(CAMS, W, H)
    got = {}
    orig = _hip.rasterize_forward_batch

    def spy(*a, **k):
        assert k.get("forward_only") is True and len(a[0]) == CAMS          # one plain view per camera, no-grad flags
        out = orig(*a, **k)
        got["radii"], got["states"], got["cams"] = out[1], out[3], a[0]
        return out
    _hip.rasterize_forward_batch = spy
    try:
        ims, depths, masks = rdr.render_cameras_with_mask(poses, data, bg=(0.0, 0.0, 0.0))
    finally:
        _hip.rasterize_forward_batch = orig
    torch.cuda.synchronize()
    assert len(ims) == CAMS and ims[0].shape == (3, H, W) and masks[0].shape == (3, H, W)
    longest = 0
    for i in range(CAMS):
        rs = got["cams"][i]
        cam = OracleCamera(H, W, float(rs.tanfovx), float(rs.tanfovy), rs.bg.cpu().numpy(), 1.0, rs.viewmatrix.cpu().numpy().reshape(-1),
                           rs.projmatrix.cpu().numpy().reshape(-1), 0, rs.campos.cpu().numpy())
        o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                         rotations=g["rotations"], nthreads=min(64, os.cpu_count() or 8))
        ok = ~o2.ambiguous
        assert ok.mean() > 0.98
        v = _hip.debug_views(got["states"][i])
        assert np.array_equal(got["radii"][i].cpu().numpy(), o2.radii), i
        _check_lists(v, H, W, o2.point_list, o2.ranges, o2.n_contrib, ok, o2.means2D, o2.conic_opacity, o2.tiles_touched, o2.offsets)
        rg = v["ranges"].cpu().numpy().astype(np.int64)
        longest = max(longest, int((rg[:, 1] - rg[:, 0]).max()))
        assert mixed_err(ims[i].cpu().numpy()[:, ok], o2.color[:, ok]) < TOL, i
        assert mixed_err(depths[i].cpu().numpy()[:, ok], o2.depth[:, ok]) < TOL, i
        assert mixed_err(v["final_T"].cpu().numpy()[ok], o2.final_T[ok]) < TOL, i
        m = masks[i].cpu().numpy()
        assert np.array_equal(m[0], m[1]) and np.array_equal(m[0], m[2])
        want = 1.0 - o2.final_T                                   # black background: every mask channel = sum_i alpha_i T_i = 1 - T_final
        _margin(f"cfg5 mask cam {i}", float(np.abs(m[0][ok] - want[ok]).max()), 1e-4)
        assert float(np.abs(m[0][ok] - want[ok]).max()) <= 1e-4 * max(1.0, float(np.abs(want).max())), i
    assert longest > 1024, f"the scene does not reach the dense-scene sort paths (longest list {longest})"
    print("config5 frame: longest tile list", longest)


def test_bench_step_against_oracle_and_autograd(dev):
    """The function bench.py TIMES -- gsdyn.step.render_step_views at the bench workload (8 views 800x800 of SynthScene-v1, 100k
    Gaussians, fused raw-parameter mode, capacity-mode forward), both colour-gradient modes -- directly against oracle O2:
    every view's image / radii, and the parameter gradients summed over the 8 views (O2 gives the gradients of the ACTIVATED
    parameters; their chain to the raw ones is torch autograd in fp64 on the CPU).  And against the autograd path
    (rasterize_gaussians_views + stand-alone activations) on all 8 views.  /root/reference/src/tracking/train_gs.py:25-39,
    train_utils.py:174-192 are the reference's form of this step."""
    from diff_gaussian_rasterization import rasterize_gaussians_views
    from gsdyn import synth_ring_cameras, synth_scene_params
    from gsdyn.step import params2rendervar_fused, render_step_views
    P, W, H, V = 100_000, 800, 800, 8
    params = synth_scene_params(P, seed=0, device=dev)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dLn = np.random.default_rng(1234).uniform(-1, 1, (V, 3, H, W)).astype(np.float32)     # bench.py's seed
    raw = {k: params[k].detach().cpu().double().requires_grad_(True) for k in ("unnorm_rotations", "logit_opacities", "log_scales")}
    act = dict(rotations=torch.nn.functional.normalize(raw["unnorm_rotations"]), opacities=torch.sigmoid(raw["logit_opacities"]),
               scales=torch.exp(raw["log_scales"]))
    # the oracle is fed the activated values the DEVICE computes (gsr_activate_forward: bit-identical to the fused form), so that
    # integer outputs (radii) stay comparable bit for bit; the fp64 graph above only carries the gradients back
    from diff_gaussian_rasterization import _hip
    rot_d, op_d, sc_d = _hip.activate_forward(params["unnorm_rotations"].detach(), params["logit_opacities"].detach(), params["log_scales"].detach())
    g_in = dict(means3D=params["means3D"].detach().cpu().numpy(), colors_precomp=params["rgb_colors"].detach().cpu().numpy(),
                rotations=rot_d.cpu().numpy(), opacities=op_d.cpu().numpy(), scales=sc_d.cpu().numpy())
    nthreads = os.cpu_count() or 8
    o_imgs, o_radii, o_m2, sums = [], [], [], None
    for v, cam in enumerate(cams):
        ocam = OracleCamera(H, W, cam.tanfovx, cam.tanfovy, cam.bg.cpu().numpy(), 1.0, cam.viewmatrix.cpu().numpy().reshape(-1),
                            cam.projmatrix.cpu().numpy().reshape(-1), 0, cam.campos.cpu().numpy())
        o2 = TiledOracle(ocam, g_in["means3D"], g_in["opacities"], colors_precomp=g_in["colors_precomp"], scales=g_in["scales"],
                         rotations=g_in["rotations"], nthreads=nthreads)
        amb = o2.ambiguous
        assert amb.mean() < 0.005
        dLn[v][:, amb] = 0.0
        gr = o2.backward(dLn[v])
        o_imgs.append((o2.color, amb)); o_radii.append(o2.radii)
        gr = {k: np.asarray(x, np.float64) for k, x in gr.items() if x is not None and k != "cov3D_precomp"}
        o_m2.append(gr.pop("means2D"))
        sums = gr if sums is None else {k: sums[k] + gr[k] for k in gr}
        del o2
    # chain of the summed activated-parameter gradients back to the raw parameters (fp64 autograd on the CPU)
    torch.autograd.backward([act["rotations"], act["opacities"], act["scales"]],
                            [torch.tensor(sums["rotations"]), torch.tensor(sums["opacities"]).reshape(P, 1), torch.tensor(sums["scales"])])
    want = {"means3D": sums["means3D"], "rgb_colors": sums["colors_precomp"], "unnorm_rotations": raw["unnorm_rotations"].grad.numpy(),
            "logit_opacities": raw["logit_opacities"].grad.numpy(), "log_scales": raw["log_scales"].grad.numpy()}
    dL = torch.tensor(dLn, device=dev)
    render_step_views(params, cams, dL)                       # first call: establishes the capacity (synchronous forward)
    for colour in (True, False):
        ims, g = render_step_views(params, cams, dL, want_colour_grad=colour)      # capacity mode + fused activations: what bench.py times
        torch.cuda.synchronize()
        ims_n = ims.cpu().numpy()
        for v in range(V):
            ok = ~o_imgs[v][1]
            assert mixed_err(ims_n[v][:, ok], o_imgs[v][0][:, ok]) < TOL, f"view {v} colour"
            assert np.array_equal(g["radii"][v].cpu().numpy(), o_radii[v]), f"view {v} radii"
            assert rel_err(g["means2D"][v].cpu().numpy()[:, :2], o_m2[v][:, :2]) < TOL, f"view {v} means2D gradient"
        for k, ref in want.items():
            if k == "rgb_colors" and not colour:
                assert k not in g
                continue
            got = g[k].cpu().numpy().reshape(ref.shape)
            assert rel_err(got, ref) < TOL, (colour, k, rel_err(got, ref))
            _row_check(f"bench step (8 x 800^2, 100k, colour grad {colour}) vs O2: {k}", got, ref)
    # the autograd path on all 8 views (stand-alone activation kernels, rasterize_gaussians_views): same sums
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")}
    rv = params2rendervar_fused(leaves)
    m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
    im_a, _, _ = rasterize_gaussians_views(cams, rv["means3D"], m2, rv["opacities"], colors_precomp=rv["colors_precomp"], scales=rv["scales"],
                                           rotations=rv["rotations"])
    im_a.backward(gradient=dL)
    ims, g = render_step_views(params, cams, dL)
    torch.cuda.synchronize()
    assert torch.equal(im_a.detach(), ims)
    for k in leaves:
        a, b = leaves[k].grad, g[k].reshape(leaves[k].shape)
        assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item(), k      # same kernels, same order: equal up to the fused chain's rounding
    assert torch.equal(m2.grad, g["means2D"])


@pytest.mark.parametrize("P,W,H,seed", [(5000, 256, 192, 4), (100_000, 800, 800, 11)])
def test_row_wise_error_against_the_fp64_oracle(dev, P, W, H, seed):
    """Whose error is the row-wise gap between the HIP path and oracle O2?  Both are fp32.  Against the fp64 build of the same oracle
    (same tile lists: it takes over the fp32 run's discrete decisions) the HIP gradients and the fp32 oracle's gradients are about
    equally far from the exact values, row by row: the worst rows of either are ~1e-4 of the row's own magnitude.  Asserted: the HIP
    path is no further from fp64 than 2x the fp32 oracle is (+ 2e-5), per tensor; logged to the row-margins file."""
    if P == 100_000:
        from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
        params = synth_scene_params(P, device=dev)
        cam_t = synth_ring_cameras(4, W, H, device=dev)[0]
        with torch.no_grad():
            rv = {k: v.detach().cpu().numpy() for k, v in params2rendervar(params).items()}
        cam = OracleCamera(H, W, cam_t.tanfovx, cam_t.tanfovy, cam_t.bg.cpu().numpy(), 1.0, cam_t.viewmatrix.cpu().numpy().reshape(-1),
                           cam_t.projmatrix.cpu().numpy().reshape(-1), 0, cam_t.campos.cpu().numpy())
        g = {k: rv[k] for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp")}
    else:
        g = random_gaussians(P, seed=seed, scale_lo=0.02, scale_hi=0.25)
        cam = ring_camera(W, H, v=seed, bg=(0.1, 0.3, 0.5))
    nt = os.cpu_count() or 8
    kw = dict(colors_precomp=g["colors_precomp"], scales=g["scales"], rotations=g["rotations"], nthreads=nt)
    o32 = TiledOracle(cam, g["means3D"], g["opacities"], **kw)
    o64 = TiledOracle(cam, g["means3D"], g["opacities"], f64=True, decisions_of=o32, **kw)
    assert np.array_equal(o32.radii, o64.radii) and np.array_equal(o32.point_list, o64.point_list)
    # pixels where a threshold decision (alpha >= 1/255, T >= 1e-4) may differ between the builds: flagged by either, or visibly
    # decided differently (T accumulates ~1e-5 of relative error over hundreds of factors in fp32: outside the fp32 run's own band)
    ok = ~(o32.ambiguous | o64.ambiguous | (o32.n_contrib != o64.n_contrib) | (np.abs(o32.color - o64.color).max(0) > 2e-5))
    assert ok.mean() > 0.995
    dL = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
    dL[:, ~ok] = 0.0
    g32, g64 = o32.backward(dL), o64.backward(dL)
    color, radii, depth, grads, _ = _run_hip(cam, g, dev, dL=dL)
    assert np.array_equal(radii, o32.radii)
    assert np.abs(color[:, ok] - o64.color[:, ok]).max() < 2e-5
    with open(_ROW_LOG, "a") as f:
        for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
            e_hip, r_hip = row_err(grads[k], g64[k])
            e_o2, r_o2 = row_err(g32[k], g64[k])
            f.write(f"vs fp64 oracle P={P} {W}x{H} grad {k}: HIP worst row {r_hip} err {e_hip:.3e} (norm-wise {rel_err(grads[k], g64[k]):.2e}); "
                    f"fp32 oracle worst row {r_o2} err {e_o2:.3e} (norm-wise {rel_err(g32[k], g64[k]):.2e})\n")
            assert rel_err(grads[k], g64[k]) < TOL, k
            assert e_hip <= 2.0 * e_o2 + 2e-5, (k, e_hip, e_o2)
            assert e_hip <= 2e-3, (k, e_hip)      # (fp32 vs fp64 includes decision flips the masks above do not catch: both fp32 evaluations share them)


def test_scene_well_past_the_benchmark_sizes(dev):
    """2 M Gaussians, two 1920x1080 views (~7.8 M tile entries per view; tools/r05_big_scene.py runs 3 M x 4 views and 8 M x 1): no 32-bit
    offset or capacity wraps anywhere -- finite outputs, bit-identical reruns, the batched forward equal to the single-view forward, the
    backward linear in the incoming gradient (exact for a power of two)."""
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, V, W, H = 2_000_000, 2, 1920, 1080
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.002, scale_hi=0.012)
    cams = synth_ring_cameras(4, W, H, device=dev)[:V]
    dL = torch.rand((V, 3, H, W), device=dev) - 0.5
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}

    def step(scale):
        ims, radii, _, states = _hip.rasterize_forward_batch(list(cams), rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"],
                                                             rv["rotations"], None, prepare_backward=True)
        g = _hip.rasterize_backward_batch(states, dL * scale, rv["means3D"], radii, rv["colors_precomp"], None, rv["scales"], rv["rotations"], None)
        torch.cuda.synchronize()
        return ims, radii, [x for x in g if x is not None and x.numel()], [int(s.num_rendered) for s in states]
    ims, radii, g1, D = step(1.0)
    ims2, radii2, g1b, _ = step(1.0)
    _, _, g2, _ = step(2.0)
    assert min(D) > 7_000_000, D
    assert torch.isfinite(ims).all() and all(torch.isfinite(x).all() for x in g1)
    assert torch.equal(ims, ims2) and torch.equal(radii, radii2) and all(torch.equal(a, b) for a, b in zip(g1, g1b))
    assert all(torch.equal(2.0 * a, b) for a, b in zip(g1, g2))
    with torch.no_grad():
        im0, rad0, _ = GaussianRasterizer(raster_settings=cams[0])(**rv)
    assert torch.equal(im0, ims[0]) and torch.equal(rad0, radii[0])
    assert int((radii > 0).sum()) > P       # most Gaussians are in view
