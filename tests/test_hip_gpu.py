"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path, called through the drop-in API and
hence through the C-ABI, against the CPU oracle on the same seeded inputs and the committed goldens.

Bars (DESIGN.md section 'Parity'):
  * integers -- radii, tile lists (point_list), tile ranges: bit-exact;
  * n_contrib: exact on pixels the oracle does not flag as threshold-ambiguous;
  * colour / depth / final_T: |a-b| <= 1e-4 * (|b| + max|b|*1e-4 ...) on non-ambiguous pixels (mixed_err);
  * gradients: max|a-b| <= 1e-4 * max|b| per tensor (rel_err; sums with cancellation).
"""
import os

import numpy as np
import pytest
import torch

from util import mixed_err, oracle_camera, random_gaussians, rel_err, ring_camera, row_err, row_err_quantiles
from oracle import OracleCamera, TiledOracle

pytestmark = pytest.mark.gpu

TOL = 1e-4
def _margin(tag, err, scale):
    """Relative error, printed with GSR_TEST_VERBOSE=1 (tools/test_margins.sh collects them from a GPU run)."""
    r = err / max(scale, 1e-300)
    if os.environ.get("GSR_TEST_VERBOSE"):
        print(f"margin {tag}: {r:.2e}")
    return r


# Row-wise (per-Gaussian) gradient bounds, next to the norm-wise one (util.row_err: |a - b| / (|b| + 1e-3 max|b|) per Gaussian):
#   * 99.9 % of the Gaussians within ROW_TOL = 1e-4 of their OWN gradient,
#   * every Gaussian within ROW_TOL_WORST = 2e-4 (round 4; 5e-4 before).  Both sides of these comparisons are fp32: a Gaussian whose
#     gradient is the small difference of large per-pixel terms (scales / rotations through the 3D covariance) is conditioned worse
#     than 1e-4 in ANY fp32 evaluation order, the oracle's included -- measured worst rows are 1e-5 .. 1.43e-4 in every comparison but
#     one (profiles/r03_pytest_gpu_row_margins.log): scales row 4946 of the 5 000-Gaussian 256x192 scene, 2.035e-4 against the fp32
#     oracle, where the fp32 oracle itself sits 1.1e-4 from its own fp64 build (test_row_wise_error_against_the_fp64_oracle); that
#     one case passes ROW_TOL_WORST_P5000 = 2.5e-4 explicitly.
# The worst row of every comparison goes to gpurun_out/row_margins.log (GSR_ROW_MARGINS_LOG overrides).
ROW_TOL = 1e-4
ROW_TOL_WORST = 2e-4
ROW_TOL_WORST_P5000 = 2.5e-4
_ROW_LOG = os.environ.get("GSR_ROW_MARGINS_LOG", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "row_margins.log"))


def _row_check(tag, a, b, tol=ROW_TOL, tol_worst=ROW_TOL_WORST):
    """Per-Gaussian bounds of one gradient tensor; logs the worst row and the quantiles of the row-wise error."""
    worst, row = row_err(a, b)
    q50, q999, q9999 = row_err_quantiles(a, b, qs=(0.5, 0.999, 0.9999))
    try:
        os.makedirs(os.path.dirname(_ROW_LOG), exist_ok=True)
        with open(_ROW_LOG, "a") as f:
            f.write(f"{tag}: worst row {row} err {worst:.3e}; median {q50:.2e} p99.9 {q999:.2e} p99.99 {q9999:.2e}; norm-wise {rel_err(a, b):.2e}\n")
    except OSError:
        pass
    assert q999 <= tol, f"{tag}: 0.1 % of the rows are off by more than {q999:.3e} of (|b| + 1e-3 max|b|)"
    assert worst <= tol_worst, f"{tag}: row {row} off by {worst:.3e} of (|b| + 1e-3 max|b|)"
    return worst


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _settings(cam: OracleCamera, dev, sh_degree=None):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)  # noqa: E731
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=t(cam.bg), scale_modifier=cam.scale_modifier, viewmatrix=t(cam.viewmatrix).reshape(1, 4, 4),
        projmatrix=t(cam.projmatrix).reshape(1, 4, 4), sh_degree=cam.sh_degree if sh_degree is None else sh_degree,
        campos=t(cam.campos), prefiltered=False)


def _run_hip(cam, g, dev, dL=None, want_state=False):
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    t = {k: torch.tensor(v, device=dev, requires_grad=dL is not None) for k, v in g.items()}
    means2D = torch.zeros((g["means3D"].shape[0], 3), device=dev, requires_grad=dL is not None)
    rs = _settings(cam, dev)
    state = {}
    if want_state:
        orig = _hip.rasterize_forward

        def spy(*a, **k):
            out = orig(*a, **k)
            state["s"] = out[3]
            return out
        _hip.rasterize_forward = spy
    try:
        color, radii, depth = GaussianRasterizer(raster_settings=rs)(
            means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t.get("shs"),
            colors_precomp=t.get("colors_precomp"), scales=t.get("scales"), rotations=t.get("rotations"),
            cov3D_precomp=t.get("cov3D_precomp"))
    finally:
        if want_state:
            _hip.rasterize_forward = orig
    views = _hip.debug_views(state["s"]) if want_state else None
    grads = None
    if dL is not None:
        (color * torch.tensor(dL, device=dev)).sum().backward()
        grads = {k: v.grad.detach().cpu().numpy() for k, v in t.items() if v.grad is not None}
        grads["means2D"] = means2D.grad.detach().cpu().numpy()
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), depth.detach().cpu().numpy(), grads, views


def _check_lists(views, H, W, o_point_list, o_ranges, o_n_contrib, ok, o_means2D=None, o_conic_opacity=None,
                 o_tiles_touched=None, o_offsets=None):
    """Tile lists / ranges / n_contrib of the HIP path against the oracle's.

    With GSR_REFERENCE_LISTS=1 the library keeps the reference's 3-sigma-rect duplicates and everything must be
    bit-identical.  By default it drops (Gaussian, tile) pairs that lie outside the Gaussian's alpha >= 1/255
    pixel box ('tight' rect, exposed as views['rect']).  Then the check is:
      * soundness -- every dropped pair really is invisible: the oracle's alpha, evaluated in fp64 at the point
        of every pixel of the tile, stays below 1/255 (when the oracle arrays are given);
      * the HIP lists equal the oracle's lists with exactly those pairs removed, order preserved;
      * n_contrib equals the oracle's index re-counted over the kept entries."""
    pl = views["point_list"].cpu().numpy().astype(np.uint32)
    rg = views["ranges"].cpu().numpy().astype(np.uint32)
    nc = views["n_contrib"].cpu().numpy().astype(np.uint32)
    if os.environ.get("GSR_REFERENCE_LISTS") == "1":
        assert np.array_equal(pl, o_point_list), "tile lists differ"
        assert np.array_equal(rg, o_ranges), "tile ranges differ"
        assert np.array_equal(nc[ok], o_n_contrib[ok])
        if o_tiles_touched is not None:
            assert np.array_equal(views["tiles_touched"].cpu().numpy().astype(np.uint32), o_tiles_touched)
            assert np.array_equal(views["offsets"].cpu().numpy().astype(np.uint32), o_offsets)
        return
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lens = (o_ranges[:, 1] - o_ranges[:, 0]).astype(np.int64)
    order = np.argsort(o_ranges[:, 0].astype(np.int64) + (lens == 0) * (1 << 40), kind="stable")
    tile_of = np.repeat(order, lens[order])                    # entry e -> tile id (ranges are contiguous)
    g_of = o_point_list.astype(np.int64)
    tx, ty = tile_of % gx, tile_of // gx
    rect = views["rect"].cpu().numpy().astype(np.uint32)
    x0, y0 = (rect[:, 0] & 0xffff).astype(np.int64), (rect[:, 0] >> 16).astype(np.int64)
    x1, y1 = (rect[:, 1] & 0xffff).astype(np.int64), (rect[:, 1] >> 16).astype(np.int64)
    keep = (tx >= x0[g_of]) & (tx < x1[g_of]) & (ty >= y0[g_of]) & (ty < y1[g_of])
    # rects of at most 32 tiles carry a bit per tile (row-major) in the record's last word: exact per-tile culling
    tmask = views["rec"][:, 15].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    wt, area = (x1 - x0), (x1 - x0) * (y1 - y0)
    r_in = np.where(keep, (ty - y0[g_of]) * wt[g_of] + (tx - x0[g_of]), 0)
    small = area[g_of] <= 32
    keep &= ~small | (((tmask[g_of] >> np.minimum(r_in, 31)) & 1) == 1)
    if o_means2D is not None:                                  # soundness of every dropped pair
        dropped = np.flatnonzero(~keep)
        for c0 in range(0, dropped.size, 1 << 15):
            d = dropped[c0:c0 + (1 << 15)]
            m = o_means2D[g_of[d]].astype(np.float64)
            co = o_conic_opacity[g_of[d]].astype(np.float64)
            px = tx[d, None] * 16 + np.arange(16)[None, :]
            py = ty[d, None] * 16 + np.arange(16)[None, :]
            dx = (m[:, 0, None] - px)[:, None, :]              # [n,1,16]
            dy = (m[:, 1, None] - py)[:, :, None]              # [n,16,1]
            power = -0.5 * (co[:, 0, None, None] * dx * dx + co[:, 2, None, None] * dy * dy) - co[:, 1, None, None] * dx * dy
            alpha = co[:, 3, None, None] * np.exp(np.minimum(power, 0.0))
            assert float(alpha.max()) < 1.0 / 255.0, "a dropped (Gaussian, tile) pair is visible"
    # kept entries, in the oracle's order, laid out tile by tile in tile-id order of first appearance
    exp_pl = o_point_list[keep]
    kept_per_tile = np.bincount(tile_of[keep], minlength=gx * gy)
    assert int(views["offsets"][-1]) == int(keep.sum())
    assert np.array_equal(rg[:, 1] - rg[:, 0], kept_per_tile.astype(np.uint32)), "tile list lengths differ"
    # HIP ranges are contiguous in increasing tile id (global sort key = tile id), like the oracle's
    nzt = np.flatnonzero(kept_per_tile)
    starts = np.concatenate([[0], np.cumsum(kept_per_tile[nzt])[:-1]])
    assert np.array_equal(rg[nzt, 0].astype(np.int64), starts), "tile ranges differ"
    # oracle entries are stored in increasing tile id too, so filtering preserves the layout
    assert np.array_equal(np.sort(order[:np.count_nonzero(lens)]), order[:np.count_nonzero(lens)])
    assert np.array_equal(pl, exp_pl), "tile lists differ"
    # n_contrib: oracle index n (1-based position of the last contributor in its tile list) -> position among kept
    ck = np.concatenate([[0], np.cumsum(keep)])
    ys, xs = np.nonzero(ok)
    t = (ys // 16) * gx + (xs // 16)
    n_o = o_n_contrib[ys, xs].astype(np.int64)
    base = o_ranges[t, 0].astype(np.int64)
    exp_n = ck[base + n_o] - ck[base]
    assert np.array_equal(nc[ys, xs].astype(np.int64), exp_n), "n_contrib differs"



AMBIGUOUS_PIXEL_BOUND = 4e-3     # one flipped alpha >= 1/255 decision moves a pixel by at most (1/255) * T * colour


def _check_against_oracle(cam, g, dev, seed=0, nthreads=4, check_lists=True, min_ok=0.995, backward=True, tol_worst=ROW_TOL_WORST):
    """Forward + backward of the HIP path vs oracle O2 on the same inputs (``backward=False``: forward only, under the
    caller's no-grad inputs -- the forward-only configs).

    Pixels where the oracle saw a threshold decision (alpha >= 1/255, T >= 1e-4) within 1e-5 relative of
    flipping are 'ambiguous': two correct fp32 implementations may legitimately decide differently there,
    and one flipped pair changes a pixel by up to ~4e-3 * colour.  They are excluded from the image
    comparison, and the upstream gradient is zeroed on them (for both sides) so they cannot leak into
    the per-Gaussian gradient comparison through the 1/(1-alpha) amplification.  At least 98 % of the pixels must take
    part in the tight comparison, and the excluded ones are still compared, at the bound a single flipped pair allows
    (4e-3 x the colour / depth scale), so they are not a blind spot of the image comparison."""
    assert min_ok >= 0.98, "a comparison that drops more than 2 % of the pixels proves little"
    H, W = cam.image_height, cam.image_width
    o2 = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g.get("colors_precomp"), shs=g.get("shs"),
                     scales=g.get("scales"), rotations=g.get("rotations"), cov3D_precomp=g.get("cov3D_precomp"),
                     nthreads=nthreads)
    ok = ~o2.ambiguous
    assert ok.mean() > min_ok, "too many threshold-ambiguous pixels for a meaningful comparison"
    dL = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
    dL[:, ~ok] = 0.0
    # The DEFAULT path of the drop-in module (the torch C++ layer, _C.so) is the one compared with the oracle below; the spied run
    # (the spy makes the module take its ctypes binding) supplies the internal lists and must reproduce the default path bit for bit.
    color, radii, depth, grads, _ = _run_hip(cam, g, dev, dL=dL if backward else None, want_state=False)
    color_s, radii_s, depth_s, grads_s, views = _run_hip(cam, g, dev, dL=dL if backward else None, want_state=True)
    assert np.array_equal(color, color_s) and np.array_equal(radii, radii_s) and np.array_equal(depth, depth_s), "torch C++ layer vs ctypes"
    if backward:
        for k in grads:
            assert np.array_equal(grads[k], grads_s[k]), f"torch C++ layer vs ctypes: grad {k}"
    assert np.array_equal(radii, o2.radii), "radii differ"
    if check_lists and o2.num_rendered == 0:      # nothing visible: no binning ran (offsets[] / point_list are never written or read)
        rg0 = views["ranges"].cpu().numpy()
        assert np.array_equal(rg0[:, 0], rg0[:, 1]), "an empty scene must have empty tile ranges"
        assert not views["n_contrib"].cpu().numpy().any()
    elif check_lists:
        _check_lists(views, H, W, o2.point_list, o2.ranges, o2.n_contrib, ok, o2.means2D, o2.conic_opacity,
                     o2.tiles_touched, o2.offsets)
        assert mixed_err(views["final_T"].cpu().numpy()[ok], o2.final_T[ok]) < TOL
    assert mixed_err(color[:, ok], o2.color[:, ok]) < TOL, "colour"
    assert mixed_err(depth[:, ok], o2.depth[:, ok]) < TOL, "depth"
    if (~ok).any():     # threshold-ambiguous pixels: within what one flipped decision can move them
        cs = max(1.0, float(np.abs(o2.color).max()))
        assert np.abs(color[:, ~ok] - o2.color[:, ~ok]).max() <= AMBIGUOUS_PIXEL_BOUND * cs, "colour on ambiguous pixels"
        assert np.abs(depth[:, ~ok] - o2.depth[:, ~ok]).max() <= AMBIGUOUS_PIXEL_BOUND * max(1.0, float(o2.depth.max())), "depth on ambiguous pixels"
    rg = views["ranges"].cpu().numpy().astype(np.int64)
    o2.hip_max_list = int((rg[:, 1] - rg[:, 0]).max()) if o2.num_rendered else 0        # longest per-tile list the HIP path sorted
    if not backward:
        return o2
    gr = o2.backward(dL)
    worst = {}
    for k, v in grads.items():
        e = rel_err(v, gr[k])
        worst[k] = e
        assert e < TOL, f"grad {k}: rel err {e:.3e}"
        _row_check(f"oracle P={g['means3D'].shape[0]} {W}x{H} seed {seed} grad {k}", v, gr[k], tol_worst=tol_worst)
    if os.environ.get("GSR_TEST_VERBOSE"):
        print("parity margins:", {k: f"{e:.2e}" for k, e in worst.items()}, "colour", f"{mixed_err(color[:, ok], o2.color[:, ok]):.2e}")
    return o2


def _pin_tile_sort_build(monkeypatch, rcap):
    """The builds of tile_sort (gsr_launch_binning): "1024" = the default for ordinary scenes (one launch, 20 KiB LDS block), "2048" = the
    36 KiB block, "4096" = the dense-scene form (wave tickets in a launch without LDS + long tickets with the 2048-entry block),
    "4096L" = the same with the long tickets on the 4096-entry block (the default of the dense-scene form since round 4)."""
    monkeypatch.setenv("GSR_TILE_SORT_RCAP", {"1024": "1", "2048": "2048", "4096": "4096", "4096L": "4096"}[rcap])
    if rcap == "4096":
        monkeypatch.setenv("GSR_LONG_SORT", "2048")       # the 2048-entry block for the long tickets (the default until round 4)
    else:
        monkeypatch.delenv("GSR_LONG_SORT", raising=False)


def test_device_selftest(dev):
    from diff_gaussian_rasterization import _hip
    assert _hip.selftest(dev) == 0


def test_committed_goldens(dev, golden_dir):
    z = np.load(os.path.join(golden_dir, "raster_cases.npz"))
    for n in [str(x) for x in z["names"]]:
        v = z[f"{n}/cam"]
        cam = OracleCamera(int(v[0]), int(v[1]), float(v[2]), float(v[3]), v[4:7].astype(np.float32), 1.0,
                           v[7:23].astype(np.float32), v[23:39].astype(np.float32), 0, v[39:42].astype(np.float32))
        g = {k: z[f"{n}/in_{k}"] for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp")}
        color, radii, depth, grads, _ = _run_hip(cam, g, dev, dL=z[f"{n}/dL_dcolor"], want_state=False)     # default path: _C.so
        color_s, radii_s, depth_s, grads_s, views = _run_hip(cam, g, dev, dL=z[f"{n}/dL_dcolor"], want_state=True)   # ctypes (spy): lists
        assert np.array_equal(color, color_s) and np.array_equal(radii, radii_s) and np.array_equal(depth, depth_s), n
        assert all(np.array_equal(grads[k], grads_s[k]) for k in grads), n
        ok = ~z[f"{n}/ambiguous"]
        assert np.array_equal(radii, z[f"{n}/radii"]), n
        _check_lists(views, cam.image_height, cam.image_width, z[f"{n}/point_list"], z[f"{n}/ranges"],
                     z[f"{n}/n_contrib"], ok)
        assert mixed_err(color[:, ok], z[f"{n}/color"][:, ok]) < TOL, n
        assert mixed_err(depth[:, ok], z[f"{n}/depth"][:, ok]) < TOL, n
        for k in ("means3D", "means2D", "colors_precomp", "opacities", "scales", "rotations"):
            assert rel_err(grads[k], z[f"{n}/grad_{k}"]) < TOL, (n, k)
            _row_check(f"golden {n} grad {k}", grads[k], z[f"{n}/grad_{k}"])


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


@pytest.mark.parametrize("P,W,H,seed", [(1, 16, 16, 1), (37, 33, 17, 2), (700, 130, 94, 3), (5000, 256, 192, 4)])
def test_random_scenes_vs_oracle(dev, P, W, H, seed):
    g = random_gaussians(P, seed=seed, scale_lo=0.02, scale_hi=0.25)
    _check_against_oracle(ring_camera(W, H, v=seed, bg=(0.1, 0.3, 0.5)), g, dev, seed=seed,
                          tol_worst=ROW_TOL_WORST_P5000 if P == 5000 else ROW_TOL_WORST)


def test_randomised_sweep_vs_oracle(dev):
    """Round 4: 36 seeded random (scene, camera, image size) combinations against the oracle in one go -- sizes that are not multiples of
    the tile, one- and few-Gaussian scenes, Gaussians far larger than a tile and far smaller than a pixel, cameras inside the cloud
    (near-plane culls, frustum clamp), opaque and nearly transparent scenes -- with the full check of `_check_against_oracle` (radii and
    lists bit-exact, images, all gradients norm-wise and row-wise)."""
    rng = np.random.default_rng(2024)
    done = 0
    for case in range(36):
        P = int(rng.choice([1, 2, 3, 17, 64, 257, 900, 2500]))
        W, H = int(rng.integers(9, 220)), int(rng.integers(9, 160))
        lo = float(rng.choice([0.002, 0.02, 0.1]))
        hi = lo * float(rng.choice([2.0, 10.0, 40.0]))
        g = random_gaussians(P, seed=1000 + case, scale_lo=lo, scale_hi=hi, spread=float(rng.choice([0.3, 1.0, 2.5])))
        shift = float(rng.choice([-3.0, 0.0, 2.5]))                     # opacity logits shifted: faint / as is / opaque scenes
        g["opacities"] = (1.0 / (1.0 + np.exp(-(np.log(g["opacities"] / (1.0 - g["opacities"])) + shift)))).astype(np.float32)
        cam = ring_camera(W, H, v=int(rng.integers(0, 7)), V=7, radius=float(rng.choice([0.6, 2.0, 4.0, 9.0])),
                          height=float(rng.choice([-0.5, 0.8, 3.0])), bg=tuple(float(x) for x in rng.uniform(0, 1, 3)))
        try:
            # Gaussians larger than the whole scene (scale up to 4 in a unit cloud) cover every pixel of every tile: their gradients are
            # sums over ~25 000 pixels of terms that cancel, and BOTH fp32 evaluations sit up to ~1e-3 from the fp64 oracle row-wise
            # (test_row_wise_error_against_the_fp64_oracle; measured here: 3.1e-4 between the two fp32 evaluations) -- the worst-row bound
            # is 1e-3 for those cases; the norm-wise 1e-4 and the 99.9 % row bound of 1e-4 hold for all
            _check_against_oracle(cam, g, dev, seed=case, min_ok=0.98, tol_worst=1e-3 if hi >= 1.0 else ROW_TOL_WORST_P5000)
            done += 1
        except AssertionError as e:
            if "too many threshold-ambiguous pixels" in str(e):     # (a scene of a few huge faint Gaussians: nothing to compare tightly)
                continue
            raise AssertionError(f"case {case}: P={P} {W}x{H} scales {lo}..{hi}: {e}") from e
    assert done >= 30, done


def test_reference_list_mode_is_bit_identical(dev, monkeypatch, golden_dir):
    """GSR_REFERENCE_LISTS=1 keeps the reference's 3-sigma-rect duplicates: tiles_touched / offsets / point_list /
    ranges / n_contrib are then bit-identical to the oracle's, and -- because the pairs the default mode drops
    fail the alpha test everywhere -- images and gradients of the two modes are bit-identical to each other."""
    g = random_gaussians(2500, seed=77, scale_lo=0.02, scale_hi=0.3)
    g["opacities"] = (g["opacities"] * 0.6).astype(np.float32)          # more faint Gaussians: more dropped pairs
    cam = ring_camera(176, 144, v=3, bg=(0.2, 0.1, 0.4))
    dL = np.random.default_rng(5).uniform(-1, 1, (3, 144, 176)).astype(np.float32)
    tight = _run_hip(cam, g, dev, dL=dL, want_state=True)
    monkeypatch.setenv("GSR_REFERENCE_LISTS", "1")
    _check_against_oracle(cam, g, dev, seed=3)                          # exact list comparison branch
    ref = _run_hip(cam, g, dev, dL=dL, want_state=True)
    assert int(ref[4]["offsets"][-1]) > int(tight[4]["offsets"][-1])   # the default mode really dropped pairs
    assert np.array_equal(tight[0], ref[0]) and np.array_equal(tight[1], ref[1]) and np.array_equal(tight[2], ref[2])
    for k in ref[3]:
        assert np.array_equal(tight[3][k], ref[3][k]), k
    test_committed_goldens(dev, golden_dir)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_colours_vs_oracle(dev, deg):
    g = random_gaussians(400, seed=10 + deg, scale_lo=0.03, scale_hi=0.3, sh_M=16)
    del g["colors_precomp"]
    _check_against_oracle(ring_camera(96, 80, v=deg, sh_degree=deg), g, dev, seed=deg)


def test_cov3d_precomp_vs_oracle(dev):
    g = random_gaussians(300, seed=21, scale_lo=0.03, scale_hi=0.3)
    cam = ring_camera(80, 64)
    probe = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"],
                        rotations=g["rotations"])
    g2 = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=g["colors_precomp"], cov3D_precomp=probe.cov3D)
    _check_against_oracle(cam, g2, dev, seed=5)


@pytest.mark.parametrize("rcap,P", [("1024", 6000), ("2048", 6000), ("4096", 9000), ("4096L", 9000)])
def test_huge_tile_lists_take_the_global_sort_path(dev, monkeypatch, rcap, P):
    """More than 2 x RCAP entries per tile: the per-tile sort leaves LDS and runs its network in global memory
    (RCAP = radix capacity of the tile_sort build, pinned here; the library picks it from the average list length)."""
    _pin_tile_sort_build(monkeypatch, rcap)
    # Gaussians far wider than the image (sigma 45-80 pixels over 32): alpha = 0.017 .. 0.02 at every pixel, nowhere near the
    # 1/255 threshold, so (almost) no pixel is threshold-ambiguous although thousands of entries cover each one
    g = random_gaussians(P, seed=33, scale_lo=5.0, scale_hi=9.0, spread=0.5)
    g["opacities"][:] = 0.02
    o2 = _check_against_oracle(ring_camera(32, 32), g, dev, seed=6, min_ok=0.98)
    assert o2.hip_max_list > 2 * int(rcap.rstrip("L"))


@pytest.mark.parametrize("rcap", ["1024", "2048", "4096", "4096L"])
@pytest.mark.parametrize("P", [50, 100, 200, 400, 1500, 3000])
def test_tile_sort_paths(dev, monkeypatch, P, rcap):
    """Per-tile list lengths that select each tile_sort path: <= 64 / 128 / 256 / 512 one wave in registers (1, 2, 4, 8
    keys per lane), <= RCAP LDS radix sort, <= 2 RCAP LDS network (beyond: test_huge_tile_lists...), for both builds of
    the kernel (RCAP 2048 / 4096)."""
    _pin_tile_sort_build(monkeypatch, rcap)
    g = random_gaussians(P, seed=40 + P, scale_lo=5.0, scale_hi=9.0, spread=0.5)   # wider than the image: see the test above
    g["opacities"][:] = 0.03
    o2 = _check_against_oracle(ring_camera(32, 32), g, dev, seed=8, min_ok=0.98)
    n = o2.hip_max_list
    lo, hi = {50: (1, 64), 100: (65, 128), 200: (129, 256), 400: (257, 512), 1500: (513, 2048), 3000: (2049, 4096)}[P]
    assert lo <= n <= hi, n


def test_many_gaussians_take_the_scan_kernel_path(dev):
    """More than 512 Ki Gaussians: per-block entry counts are scanned on the device (below that the host adds
    them up and emit blocks derive their own base)."""
    g = random_gaussians(540_000, seed=91, scale_lo=0.004, scale_hi=0.02, spread=1.2)
    _check_against_oracle(ring_camera(96, 64, v=1), g, dev, seed=9, nthreads=min(64, os.cpu_count() or 8))


@pytest.mark.parametrize("P,W,H,seed", [(700, 130, 94, 3), (5000, 256, 192, 4), (540_000, 96, 64, 91)])
def test_radix_binning_path_vs_oracle(dev, monkeypatch, P, W, H, seed):
    """Round 4: the single-view entry points bin with the tile-row counting sort as well; the radix path (emit_entries -> radix_hist ->
    radix_scatter x 2 -> tile_order) stays the fallback for tile grids above GSR_BIN_MAX_T and for devices whose LDS cannot hold a
    view's tile counters.  GSR_RADIX_BINNING=1 pins it: the same oracle check (lists bit-exact), incl. the device-side block scan
    above 512 Ki Gaussians."""
    monkeypatch.setenv("GSR_RADIX_BINNING", "1")
    big = P > 100_000
    g = random_gaussians(P, seed=seed, scale_lo=0.004 if big else 0.02, scale_hi=0.02 if big else 0.25, spread=1.2 if big else 1.0)
    _check_against_oracle(ring_camera(W, H, v=seed % 4, bg=(0.1, 0.3, 0.5)), g, dev, seed=seed, nthreads=min(64, os.cpu_count() or 8),
                          tol_worst=ROW_TOL_WORST_P5000 if P == 5000 else ROW_TOL_WORST)


@pytest.mark.parametrize("P", [524_033, 524_288])
def test_last_host_scanned_block_count(dev, monkeypatch, P):
    """P in 524 033 .. 524 288 = exactly 2048 preprocess blocks, the most emit_entries prefixes in LDS itself: the total sits in
    slot 2048, one past the 256 x 8 slots the threads fill (it was never written).  Both the synchronous forward and
    the capacity-mode forward (entry count read on the device) against the oracle."""
    from diff_gaussian_rasterization import _hip
    monkeypatch.setenv("GSR_RADIX_BINNING", "1")      # emit_entries is the radix path's kernel (the default is the tile-row binning now)
    g = random_gaussians(P, seed=92, scale_lo=0.004, scale_hi=0.02, spread=1.2)
    cam = ring_camera(96, 64, v=2)
    o2 = _check_against_oracle(cam, g, dev, seed=10, nthreads=min(64, os.cpu_count() or 8))
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    rs = _settings(cam, dev)
    key = (dev.index, P, cam.image_height, cam.image_width)
    _hip._entries_capacity.pop(key, None)
    for no_sync in (False, True):      # the second call runs in capacity mode (the first one left the capacity behind)
        im, radii, _d, states = _hip.rasterize_forward_batch([rs], t["means3D"], t["opacities"], t["colors_precomp"], None, t["scales"],
                                                             t["rotations"], None, no_host_sync=no_sync)
        assert (states[0].pending is not None) == no_sync
        assert _hip.forward_counts_ok(states)
        ok = ~o2.ambiguous
        assert mixed_err(im[0].cpu().numpy()[:, ok], o2.color[:, ok]) < TOL
        assert np.array_equal(radii[0].cpu().numpy(), o2.radii)


@pytest.mark.parametrize("P,W,H", [(300, 48, 32), (1500, 32, 32), (3000, 32, 32), (9000, 32, 32), (20000, 200, 120)])
def test_tile_row_binning_lists_vs_oracle(dev, P, W, H):
    """The multi-view entry points bin with the tile-row counting sort (gsr_binning.hip: bin_count / bin_scan / bin_emit): entries arrive
    in their tile's segment in no particular order and every tile_sort path must still produce the reference order -- depth, ties by
    ascending Gaussian id.  Lists, ranges, n_contrib and images of a 2-view call against the oracle, bit for bit, with many
    equal-depth ties (duplicated positions: pairs, and one run of 150 identical depths that exceeds the in-place repair),
    synchronous and capacity mode."""
    from diff_gaussian_rasterization import _hip
    wide = P <= 9000
    g = random_gaussians(P, seed=70 + P, scale_lo=5.0 if wide else 0.02, scale_hi=9.0 if wide else 0.2, spread=0.5 if wide else 1.0)
    if wide:
        g["opacities"][:] = 0.03
    half = P // 2
    g["means3D"][half:2 * half] = g["means3D"][:half]          # pairs of Gaussians at one position: equal depth bits
    g["means3D"][:min(150, P)] = g["means3D"][0]                # and a long run
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    cams = [ring_camera(W, H, v=1), ring_camera(W, H, v=3)]
    rss = [_settings(c, dev) for c in cams]
    o2s = [TiledOracle(c, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"], rotations=g["rotations"],
                       nthreads=8) for c in cams]
    _hip._entries_capacity.pop((dev.index, P, H, W), None)
    for no_sync in (False, True):
        im, radii, depth, states = _hip.rasterize_forward_batch(rss, t["means3D"], t["opacities"], t["colors_precomp"], None, t["scales"],
                                                                t["rotations"], None, no_host_sync=no_sync)
        assert (states[0].pending is not None) == no_sync and _hip.forward_counts_ok(states)
        torch.cuda.synchronize()
        for v, o2 in enumerate(o2s):
            views = _hip.debug_views(states[v])
            D = int(views["offsets"][-1])
            views["point_list"] = views["point_list"][:D]
            ok = ~o2.ambiguous
            assert np.array_equal(radii[v].cpu().numpy(), o2.radii)
            _check_lists(views, H, W, o2.point_list, o2.ranges, o2.n_contrib, ok, o2.means2D, o2.conic_opacity, o2.tiles_touched, o2.offsets)
            assert mixed_err(im[v].cpu().numpy()[:, ok], o2.color[:, ok]) < TOL
            assert mixed_err(depth[v].cpu().numpy()[:, ok], o2.depth[:, ok]) < TOL


def test_early_termination_dense_scene(dev):
    g = random_gaussians(3000, seed=34, scale_lo=0.1, scale_hi=0.5, spread=0.6)
    g["opacities"][:] = 0.95
    _check_against_oracle(ring_camera(120, 88, bg=(1, 1, 1)), g, dev, seed=7)


@pytest.mark.parametrize("case", ["sparse", "dense", "wide"])
def test_exact_lists_and_used_flags_equal_the_conservative_backward(dev, case):
    """Round 4: the tracking forward leaves per-entry contribution bytes (which of a tile's quads blended the entry) and per-Gaussian
    used flags; the blend backward builds its per-quad lists from the bytes and skips the zero fill of unmarked Gaussians, the
    per-Gaussian backward skips their records.  The geometry state's `tracked` word says whether to trust them: cleared (here: by
    hand, between forward and backward), every quad stages every entry below the tile's deepest contributor and every record is
    written and read -- the conservative evaluation.  Every gradient of the two must be EQUAL (the bytes and flags only remove visits
    and records whose contributions are exact zeros): this pins both against the per-pixel hit test itself."""
    from diff_gaussian_rasterization import _hip
    if case == "sparse":
        g, cam = random_gaussians(4000, seed=51), ring_camera(200, 136, bg=(0.1, 0.2, 0.3))
    elif case == "dense":
        g, cam = random_gaussians(3000, seed=52, scale_lo=0.1, scale_hi=0.5, spread=0.6), ring_camera(120, 88, bg=(1, 1, 1))
        g["opacities"][:] = 0.95
    else:
        g, cam = random_gaussians(300, seed=53, scale_lo=0.5, scale_hi=2.0), ring_camera(96, 64, bg=(0, 0, 0))
    rs = _settings(cam, dev)
    t = {k: torch.tensor(v, device=dev) for k, v in g.items()}
    P = t["means3D"].shape[0]
    dL = torch.tensor(np.random.default_rng(9).uniform(-1, 1, (3, cam.image_height, cam.image_width)).astype(np.float32), device=dev)
    outs = []
    for conservative in (False, True):
        color, radii, depth, st = _hip.rasterize_forward(rs, t["means3D"], t["opacities"], t["colors_precomp"], None, t["scales"], t["rotations"], None)
        # the geometry state's counters: behind rec (64 P), rect (8 P), tiles_touched (4 P), offsets (4 (P + 1)), block sums / offsets, clamped (4 P)
        al = lambda x: (x + 255) // 256 * 256      # noqa: E731
        nblk = (P + 255) // 256
        off = al(64 * P) + al(8 * P) + al(4 * P) + al(4 * (P + 1)) + al(4 * nblk) + al(4 * (nblk + 1)) + al(4 * P)
        words = st.geom[off:off + 8].view(torch.int32)
        assert int(words[1]) == 1, words.tolist()       # [1] = tracked ([0]: the entry count when the device scans the block sums)
        used_off = off + al(64) + al(8 * P) + al(8 * nblk) + al(((P + 2047) // 2048 + 1) * 10240 * 4)
        used = st.geom[used_off:used_off + P]
        assert 0 < int(used.sum()) <= int((radii > 0).sum()) and int(used.max()) == 1
        if conservative:
            words[1] = 0
        grads = _hip.rasterize_backward(st, dL, t["means3D"], radii, t["colors_precomp"], None, t["scales"], t["rotations"], None)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in grads if x is not None and x.numel()])
    assert len(outs[0]) == len(outs[1]) >= 5
    for a, b in zip(*outs):
        assert torch.equal(a, b), (case, float((a - b).abs().max()))
    assert float(outs[0][0].abs().max()) > 0


def test_edge_cases(dev):
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = ring_camera(40, 24, bg=(0.3, 0.6, 0.9))
    rs = _settings(cam, dev)
    # P = 0
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    color, radii, depth = GaussianRasterizer(raster_settings=rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1),
                                                                 colors_precomp=z(0, 3), scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, 24, 40) and radii.numel() == 0
    # everything culled (behind the camera): background only, zero gradients, zero depth
    g = random_gaussians(50, seed=1)
    g["means3D"] = (g["means3D"] * 0.1 + np.array([40.0, 8.0, 12.0], np.float32)).astype(np.float32)  # behind the ring camera
    color, radii, depth, grads, _ = _run_hip(cam, g, dev, dL=np.ones((3, 24, 40), np.float32))
    assert np.all(radii == 0) and np.all(depth == 0)
    np.testing.assert_allclose(color[:, 5, 7], [0.3, 0.6, 0.9], atol=1e-6)
    assert all(np.all(v == 0) for v in grads.values())
    # argument validation (same exceptions as the reference extension's Python wrapper)
    r = GaussianRasterizer(raster_settings=rs)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=z(3, 3), means2D=z(3, 3), opacities=z(3, 1), scales=z(3, 3), rotations=z(3, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=z(3, 3), means2D=z(3, 3), opacities=z(3, 1), colors_precomp=z(3, 3))


def test_mark_visible(dev):
    from diff_gaussian_rasterization import GaussianRasterizer
    from oracle.tiled import mark_visible
    cam = ring_camera(32, 32)
    pts = np.random.default_rng(0).uniform(-6, 6, (1000, 3)).astype(np.float32)
    got = GaussianRasterizer(raster_settings=_settings(cam, dev)).markVisible(torch.tensor(pts, device=dev))
    assert np.array_equal(got.cpu().numpy(), mark_visible(cam.viewmatrix, pts))


def test_reference_call_pattern_get_loss(dev):
    """The literal call sequence of /root/reference/src/tracking/train_utils.py:174-192, 243-245 and
    /root/reference/src/tracking/external.py:138-142 runs against the HIP backend."""
    from gsdyn import get_loss, LossWeights, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import init_variables
    W, H, P = 160, 128, 3000
    params = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, W, H, device=dev)[0]
    im, seg = synth_targets(W, H, device=dev)
    variables = init_variables(P, dev)
    loss, variables = get_loss(params, dict(cam=cam, im=im, seg=seg, id=0), variables, True, LossWeights())
    loss.backward()
    assert torch.isfinite(loss)
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "seg_colors", "cam_m", "cam_c"):
        assert params[k].grad is not None and torch.isfinite(params[k].grad).all(), k
    g2 = variables["means2D"].grad
    assert g2.shape == (P, 3) and torch.all(g2[:, 2] == 0)
    seen = variables["seen"]
    accum = torch.norm(g2[seen, :2], dim=-1)
    assert seen.any() and torch.isfinite(accum).all()


# ------------------------------------------------------------------ full-size, size-independent properties
@pytest.fixture(scope="module")
def full_scene(dev):
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H = 100_000, 800, 800
    params = synth_scene_params(P, device=dev)
    cams = synth_ring_cameras(4, W, H, device=dev)
    return params, cams, params2rendervar


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


def test_pair_fusion_and_static_launch_variants(dev, tmp_path):
    """The fused pair pass against the same call with GSR_NO_PAIR_FUSION=1 (one pass per view) and with GSR_RENDER_STATIC=1 (one
    workgroup per tile instead of the persistent LPT queue); the partner view has a different background, the image size is
    not a multiple of 16.  Images are identical bit for bit in all three; gradients: static == persistent bit for bit, fused
    vs per-view passes within rounding."""
    import subprocess
    import sys
    script = tmp_path / "variant.py"
    script.write_text(_VARIANT_SCRIPT)
    outs = {}
    for name, env in (("default", {}), ("nopair", {"GSR_NO_PAIR_FUSION": "1"}), ("static", {"GSR_RENDER_STATIC": "1"}),
                      ("counted", {"GSR_FUSED_COUNT": "1"}),       # the counting form of preprocess_fwd (round-4 A/B switch, off by default)
                      ("no_used", {"GSR_NO_USED_FLAGS": "1"})):    # every record written and read (the per-Gaussian used flags ignored)
        e = dict(os.environ)
        e.update(env)
        out = tmp_path / (name + ".npz")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        r = subprocess.run([sys.executable, str(script), root, str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    d, n, st = outs["default"], outs["nopair"], outs["static"]
    for k in d.files:
        assert np.array_equal(d[k], st[k]), ("static", k)
        assert np.array_equal(d[k], outs["counted"][k]), ("counted", k)   # same lists, same record slots: bit-identical
        assert np.array_equal(d[k], outs["no_used"][k]), ("no_used", k)   # the flags only remove records that are all zeros
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


# ------------------------------------------------------------------ fused image loss (row N2)
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


# ------------------------------------------------------------------ conventions and larger configurations
def test_frustum_clamp_and_offcentre_camera(dev):
    """Gaussians far outside the field of view exercise the 1.3 * tanfov clamp of the EWA Jacobian and its
    gradient mask (convention A-3); the camera has an off-centre principal point and fx != fy, like the
    reference's calibrated demo cameras, and scale_modifier != 1."""
    W, H, P = 160, 120, 1500
    g = random_gaussians(P, seed=77, scale_lo=0.05, scale_hi=0.6, spread=3.5)   # many centres outside the frustum
    w2c = np.eye(4); w2c[2, 3] = 4.0
    cam = oracle_camera(W, H, w2c, fx=190.0, fy=150.0, cx=71.3, cy=66.9, bg=(0.2, 0.1, 0.4))
    cam.scale_modifier = 1.3
    # make sure the case is actually exercised
    vm = np.asarray(cam.viewmatrix, np.float32).reshape(4, 4)
    pv = g["means3D"] @ vm[:3, :3] + vm[3, :3]
    vis = pv[:, 2] > 0.2
    clamped = vis & ((np.abs(pv[:, 0] / pv[:, 2]) > 1.3 * cam.tanfovx) | (np.abs(pv[:, 1] / pv[:, 2]) > 1.3 * cam.tanfovy))
    assert clamped.sum() > 50
    _check_against_oracle(cam, g, dev, seed=9, min_ok=0.98)


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


# ------------------------------------------------------------------ fused neighbour terms of the t > 0 loss
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


def test_torch_extension_path_equals_ctypes_path(dev):
    """The torch C++ layer (_C.so: upstream's rasterize_gaussians / rasterize_gaussians_backward / mark_visible over the C-ABI) and
    the ctypes binding drive the same kernels: images, radii, depth and every gradient are bit-identical; SH colours and
    cov3D_precomp inputs, P = 0 and markVisible go through it too."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer, _hip
    assert dgr._C is not None and dgr._native() is dgr._C, "the torch C++ layer must be built (and used) on a GPU box"
    cam = ring_camera(144, 104, v=1, bg=(0.2, 0.4, 0.1), sh_degree=1)
    rs = _settings(cam, dev)
    for variant in ("colors", "shs", "cov3d"):
        g = random_gaussians(900, seed=5, scale_lo=0.03, scale_hi=0.3, sh_M=4 if variant == "shs" else 0)
        if variant == "shs":
            del g["colors_precomp"]
        if variant == "cov3d":
            probe = TiledOracle(cam, g["means3D"], g["opacities"], colors_precomp=g["colors_precomp"], scales=g["scales"], rotations=g["rotations"])
            g = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=g["colors_precomp"], cov3D_precomp=probe.cov3D)
        dL = torch.tensor(np.random.default_rng(2).uniform(-1, 1, (3, 104, 144)).astype(np.float32), device=dev)
        outs = []
        for use_ext in (True, False):
            saved = dgr._C
            if not use_ext:
                dgr._C = None
            try:
                assert (dgr._native() is not None) == use_ext
                t = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in g.items()}
                m2 = torch.zeros((900, 3), device=dev, requires_grad=True)
                im, radii, depth = GaussianRasterizer(raster_settings=rs)(
                    means3D=t["means3D"], means2D=m2, opacities=t["opacities"], shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
                    scales=t.get("scales"), rotations=t.get("rotations"), cov3D_precomp=t.get("cov3D_precomp"))
                im.backward(gradient=dL)
                outs.append((im.detach(), radii, depth.detach(), m2.grad, {k: v.grad for k, v in t.items()}))
            finally:
                dgr._C = saved
        a, b = outs
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), variant
        for k in a[4]:
            assert (a[4][k] is None) == (b[4][k] is None) and (a[4][k] is None or torch.equal(a[4][k], b[4][k])), (variant, k)
    z = lambda *s: torch.zeros(*s, device=dev)  # noqa: E731
    color, radii, depth = GaussianRasterizer(raster_settings=rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), colors_precomp=z(0, 3),
                                                                 scales=z(0, 3), rotations=z(0, 4))
    assert color.shape == (3, 104, 144) and radii.numel() == 0 and float(color.abs().max()) == 0.0
    pts = torch.tensor(np.random.default_rng(0).uniform(-6, 6, (500, 3)).astype(np.float32), device=dev)
    assert torch.equal(GaussianRasterizer(raster_settings=rs).markVisible(pts), _hip.mark_visible(pts, rs.viewmatrix))


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


def test_frozen_colours_backward_equals_full_backward(dev):
    """colors_precomp.requires_grad == False (rgb_colors in the reference's training, /root/reference/src/tracking/train_utils.py:133):
    the blend backward keeps six sums per list entry instead of nine.  Every other gradient must equal the full backward's -- the
    geometry sums are the same products, only their reduction tree differs (rounding level) -- through the drop-in module (torch
    C++ layer), the ctypes path and the multi-view call."""
    from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    P, W, H, V = 30000, 400, 304, 3
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.06)
    cams = synth_ring_cameras(V, W, H, device=dev)
    dL = torch.tensor(np.random.default_rng(11).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    with torch.no_grad():
        rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
    names = ("means3D", "opacities", "scales", "rotations")

    def one_view(frozen, env=None):
        leaves = {k: rv[k].clone().requires_grad_(not (frozen and k == "colors_precomp")) for k in names + ("colors_precomp",)}
        m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
        im, _, _ = GaussianRasterizer(raster_settings=cams[0])(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"],
                                                              colors_precomp=leaves["colors_precomp"], scales=leaves["scales"],
                                                              rotations=leaves["rotations"])
        im.backward(gradient=dL[0])
        return leaves, m2

    def views(frozen):
        leaves = {k: rv[k].clone().requires_grad_(not (frozen and k == "colors_precomp")) for k in names + ("colors_precomp",)}
        m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
        im, _, _ = rasterize_gaussians_views(cams, leaves["means3D"], m2, leaves["opacities"], colors_precomp=leaves["colors_precomp"],
                                             scales=leaves["scales"], rotations=leaves["rotations"])
        im.backward(gradient=dL)
        return leaves, m2

    import diff_gaussian_rasterization as dgr
    for run in (one_view, views):
        (a, m2a), (b, m2b) = run(False), run(True)
        assert b["colors_precomp"].grad is None and a["colors_precomp"].grad is not None
        for k in names:
            scale = a[k].grad.abs().max().item()
            assert (a[k].grad - b[k].grad).abs().max().item() <= 4e-6 * scale, (run.__name__, k)
        assert (m2a.grad - m2b.grad).abs().max().item() <= 4e-6 * m2a.grad.abs().max().item(), run.__name__
    # the ctypes path of the single-view module (what runs when the torch C++ layer is absent)
    saved = dgr._C
    try:
        dgr._C = None
        (a, m2a), (b, m2b) = one_view(False), one_view(True)
    finally:
        dgr._C = saved
    assert b["colors_precomp"].grad is None
    for k in names:
        assert (a[k].grad - b[k].grad).abs().max().item() <= 4e-6 * a[k].grad.abs().max().item(), ("ctypes", k)


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


def test_unchanged_two_call_pattern_reuses_the_tile_lists(dev):
    """The reference's own call pattern -- two separate ``GaussianRasterizer`` calls per camera with the same geometry and other colours,
    the second one fed FRESH copies of the geometry tensors (/root/reference/src/tracking/train_utils.py:174-192: ``params2rendervar`` is
    evaluated twice; /root/reference/src/predict.py:115-123: ``copy.deepcopy``) -- through the unchanged drop-in API: the torch C++ layer
    recognises the second call by the fingerprint of its preprocess outputs and blends from the first call's tile lists.  Images and
    every gradient must equal the non-reusing path bit for bit; a changed Gaussian or another camera must NOT reuse."""
    import copy
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    C_ = dgr._C
    assert C_ is not None, "the torch C++ layer must be built on a GPU box"
    P, W, H = 40_000, 400, 304
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.05)
    cams = synth_ring_cameras(4, W, H, device=dev)
    rng = np.random.default_rng(3)
    g1, g2 = (torch.tensor(rng.uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev) for _ in range(2))
    keys = ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors", "seg_colors")

    def get_loss_pair(reuse, cam):
        C_.set_list_reuse(reuse)
        h0 = C_.list_reuse_hits()
        for k in keys:
            params[k].grad = None
        params["rgb_colors"].requires_grad_(True)
        rv = params2rendervar(params)
        rv["means2D"].retain_grad()
        im, radius, depth = GaussianRasterizer(raster_settings=cam)(**rv)
        seg_rv = params2rendervar(params, colors_key="seg_colors")          # fresh rotations / opacities / scales tensors
        seg_rv["means2D"].retain_grad()
        seg, radius2, _ = GaussianRasterizer(raster_settings=cam)(**seg_rv)
        ((im * g1).sum() + (seg * g2).sum()).backward()
        torch.cuda.synchronize()
        out = dict(im=im.detach(), seg=seg.detach(), depth=depth.detach(), radius=radius, radius2=radius2, m2=rv["means2D"].grad, m2s=seg_rv["means2D"].grad)
        out.update({k: params[k].grad.clone() for k in keys})
        return out, C_.list_reuse_hits() - h0

    try:
        a, hits_a = get_loss_pair(True, cams[0])
        b, hits_b = get_loss_pair(False, cams[0])
        assert hits_a == 1 and hits_b == 0, (hits_a, hits_b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        # another camera, then a moved Gaussian: new lists each time
        C_.set_list_reuse(True)
        with torch.no_grad():
            rv = {k: v.detach() for k, v in params2rendervar(params).items()}
            h0 = C_.list_reuse_hits()
            im0, _, _ = GaussianRasterizer(raster_settings=cams[1])(**rv)
            im1, _, _ = GaussianRasterizer(raster_settings=cams[2])(**rv)                    # other camera
            assert C_.list_reuse_hits() == h0
            moved = dict(rv)
            moved["means3D"] = rv["means3D"].clone()
            moved["means3D"][123, 0] += 0.05
            im2, _, _ = GaussianRasterizer(raster_settings=cams[2])(**moved)                 # same camera, one Gaussian moved
            assert C_.list_reuse_hits() == h0
            # predict.py's mask render: deep copy of the frame's data with colours = 1
            ones = copy.deepcopy(moved)
            ones["colors_precomp"] = torch.ones_like(moved["colors_precomp"])
            mask, _, _ = GaussianRasterizer(raster_settings=cams[2])(**ones)
            assert C_.list_reuse_hits() == h0 + 1
            C_.set_list_reuse(False)
            mask_ref, _, _ = GaussianRasterizer(raster_settings=cams[2])(**ones)
            im2_ref, _, _ = GaussianRasterizer(raster_settings=cams[2])(**moved)
        assert torch.equal(mask, mask_ref) and torch.equal(im2, im2_ref)
        assert float(mask.max()) <= 1.0 + 1e-5 and not torch.equal(im1, im2)

        # Same geometry, OTHER OPACITIES, both forwards before either backward: the tile lists would be the same, but the forward leaves
        # the backward's per-quad contribution bytes next to the lists (round 4) and those depend on the opacities -- the fingerprint
        # covers them, so the second call bins for itself and the first call's backward still finds its own bytes.
        def two_opacities(reuse):
            C_.set_list_reuse(reuse)
            h0 = C_.list_reuse_hits()
            leaves = {k: params[k].detach().clone().requires_grad_(True) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors")}
            thin = {k: (v - 1.5 if k == "logit_opacities" else v) for k, v in leaves.items()}
            im_a, _, _ = GaussianRasterizer(raster_settings=cams[3])(**params2rendervar(leaves))
            im_b, _, _ = GaussianRasterizer(raster_settings=cams[3])(**params2rendervar(thin))
            (im_a * g1).sum().backward()
            ga = {k: v.grad.clone() for k, v in leaves.items()}
            (im_b * g2).sum().backward()
            torch.cuda.synchronize()
            return im_a.detach(), im_b.detach(), ga, {k: v.grad.clone() for k, v in leaves.items()}, C_.list_reuse_hits() - h0
        ra, rb = two_opacities(True), two_opacities(False)
        assert ra[4] == 0 and rb[4] == 0
        assert torch.equal(ra[0], rb[0]) and torch.equal(ra[1], rb[1]) and not torch.equal(ra[0], ra[1])
        for k in ra[2]:
            assert torch.equal(ra[2][k], rb[2][k]) and torch.equal(ra[3][k], rb[3][k]), k
    finally:
        C_.set_list_reuse(True)


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
