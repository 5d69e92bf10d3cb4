"""Row A5 (binning / per-tile sort paths) and the device self-test: tile lists bit-exact against the oracle on every sort build and binning path.
(split out of the former tests/test_hip_gpu.py; shared machinery: tests/hipcheck.py, fixtures: tests/conftest.py)"""
import os

import numpy as np
import pytest
import torch

from hipcheck import *  # noqa: F401,F403
from hipcheck import _check_against_oracle, _check_lists, _margin, _pin_tile_sort_build, _row_check, _run_hip, _settings  # noqa: F401

pytestmark = pytest.mark.gpu


def test_device_selftest(dev):
    from diff_gaussian_rasterization import _hip
    assert _hip.selftest(dev) == 0


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
    from test_forward_backward_gpu import test_committed_goldens
    test_committed_goldens(dev, golden_dir)


@pytest.mark.parametrize("rcap,P", [("1024", 6000), ("2048", 6000), ("4096", 9000)])
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


@pytest.mark.parametrize("rcap", ["1024", "2048", "4096"])
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


def test_one_very_long_list_in_an_ordinary_scene(dev):
    """an ordinary (sparse) 800x800 scene whose tile-sort build is chosen from the AVERAGE list length, plus ONE tile
    with ~3000 entries (a cluster of tiny Gaussians behind one tile corner: the reference's object close-ups).  Rounds 1 - 4 sorted such
    lists with the compare-exchange network (LDS up to 2048 entries: ~30 us; GLOBAL memory above: ~+160 us for one list); the build is now
    chosen per ticket -- lists of more than 1016 entries go to a launch of the 4096-entry LDS block (radix sort), issued when the previous
    call reported such lists.  Lists bit-exact against the oracle, and the cluster's sorts cost at most ~25 us on top."""
    from diff_gaussian_rasterization import _hip
    W = H = 800
    cam = ring_camera(W, H, bg=(0.1, 0.1, 0.1))
    base = random_gaussians(20_000, seed=71, scale_lo=0.005, scale_hi=0.03)
    clu = random_gaussians(3_000, seed=72, scale_lo=0.0015, scale_hi=0.003, spread=1.0)
    # the cluster: along the camera's viewing ray through the world origin (= the image centre, pixel 400 +- a few: inside one tile), at
    # 3000 distinct depths; lateral jitter of ~1 pixel
    th = 0.3
    eye = np.array([4.0 * np.cos(th), 0.8, 4.0 * np.sin(th)], np.float32)
    ray = -eye / np.linalg.norm(eye)
    rng = np.random.default_rng(73)
    t = rng.uniform(-0.8, 0.8, (3_000, 1)).astype(np.float32)
    clu["means3D"] = (t * ray[None] + rng.normal(0, 0.004, (3_000, 3))).astype(np.float32) + np.array([0.02, 0.02, 0.0], np.float32)
    clu["opacities"][:] = 0.02            # faint: nothing terminates early, every entry is walked
    both = {k: np.concatenate([base[k], clu[k]]) for k in base}
    o2 = _check_against_oracle(cam, both, dev, seed=5, nthreads=os.cpu_count() or 8)
    lens = o2.ranges[:, 1].astype(np.int64) - o2.ranges[:, 0].astype(np.int64)
    assert lens.max() > 2032, f"the cluster did not make a list of more than 2032 entries (longest {lens.max()})"
    assert np.median(lens[lens > 0]) < 200          # ... in an otherwise ordinary scene

    def sort_us(g):
        rs = _settings(cam, dev)
        tt = {k: torch.tensor(v, device=dev) for k, v in g.items()}
        f = lambda: _hip.rasterize_forward(rs, tt["means3D"], tt["opacities"], tt["colors_precomp"], None, tt["scales"], tt["rotations"], None)   # noqa: E731
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        _hip.profile_begin()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        prof = _hip.profile_end()
        return 1e3 * prof["tile_sort"][0] / prof["tile_sort"][1]
    t_base, t_both = sort_us(base), sort_us(both)
    print(f"tile_sort: {t_base:.1f} us without the cluster, {t_both:.1f} us with it (longest list {lens.max()})")
    assert t_both <= t_base + 25.0, (t_base, t_both)
