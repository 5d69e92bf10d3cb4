"""Row (b): the unchanged caller -- the reference's call pattern through GaussianRasterizer, the torch C++ layer against the ctypes binding, tile-list reuse.
(split out of the former tests/test_hip_gpu.py; shared machinery: tests/hipcheck.py, fixtures: tests/conftest.py)"""
import os

import numpy as np
import pytest
import torch

from hipcheck import *  # noqa: F401,F403
from hipcheck import _check_against_oracle, _check_lists, _margin, _pin_tile_sort_build, _row_check, _run_hip, _settings  # noqa: F401

pytestmark = pytest.mark.gpu

class _LayerSwitches:
    """The switches / counters of the torch C++ layer's per-device state (diff_gaussian_rasterization.layer_state) under the names these
    tests were written with (until round 6 they were module-level functions of _C acting on process globals)."""

    def __init__(self, dgr, dev):
        self.st = dgr.layer_state(dev)

    def set_list_reuse(self, on): self.st.list_reuse = bool(on)
    def set_capacity_mode(self, on): self.st.capacity_mode = bool(on)
    def list_reuse_hits(self): return self.st.stats()["list_reuse_hits"]
    def drop_list_cache(self): self.st.drop_list_cache()
    def forget_capacities(self): self.st.reset()
    def capacity_stats(self):
        d = self.st.stats()
        return d["capacity_calls"], d["capacity_overflows"], d["twins_seen_late"]



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


def test_unchanged_two_call_pattern_reuses_the_tile_lists(dev):
    """The reference's own call pattern -- two separate ``GaussianRasterizer`` calls per camera with the same geometry and other colours,
    the second one fed FRESH copies of the geometry tensors (/root/reference/src/tracking/train_utils.py:174-192: ``params2rendervar`` is
    evaluated twice; /root/reference/src/predict.py:115-123: ``copy.deepcopy``) -- through the unchanged drop-in API: the torch C++ layer
    recognises the second call by comparing its preprocess outputs with the first call's on the device, bit for bit, and blends from the first call's tile lists.  Images and
    every gradient must equal the non-reusing path bit for bit; a changed Gaussian or another camera must NOT reuse."""
    import copy
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    C_ = _LayerSwitches(dgr, dev)
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
        C_.set_capacity_mode(False)      # the count-first path: every second call of a pair compares (capacity mode PREDICTS which calls to compare:
        #                                  test_capacity_mode_learns_the_two_render_pattern)
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
        # the backward's per-quad contribution bytes next to the lists (round 4) and those depend on the opacities -- the comparison
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

        # An evaluation render under no_grad FIRST (forward-only: it leaves no contribution bytes next to its lists), then the training
        # render of the same geometry: it reuses those lists, writes the bytes itself, and its backward must equal the non-reusing path.
        def eval_then_train(reuse):
            C_.set_list_reuse(reuse)
            h0 = C_.list_reuse_hits()
            leaves = {k: params[k].detach().clone().requires_grad_(True) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors")}
            with torch.no_grad():
                im_e, _, _ = GaussianRasterizer(raster_settings=cams[1])(**params2rendervar(leaves))
            im_t, _, _ = GaussianRasterizer(raster_settings=cams[1])(**params2rendervar(leaves))
            (im_t * g1).sum().backward()
            torch.cuda.synchronize()
            return im_e, im_t.detach(), {k: v.grad.clone() for k, v in leaves.items()}, C_.list_reuse_hits() - h0
        ea, eb = eval_then_train(True), eval_then_train(False)
        assert ea[3] == 1 and eb[3] == 0, (ea[3], eb[3])
        assert torch.equal(ea[0], eb[0]) and torch.equal(ea[1], eb[1]) and torch.equal(ea[0], ea[1])
        for k in ea[2]:
            assert torch.equal(ea[2][k], eb[2][k]), k
    finally:
        C_.set_list_reuse(True)
        C_.set_capacity_mode(True)


@pytest.mark.parametrize("P,W,H", [(50_000, 1280, 720), (8_957, 640, 480)])
def test_reference_workload_shapes_vs_oracle(dev, P, W, H):
    """The reference's own sizes (VERDICT r04 'missing' 3): 1280x720 (80 x 45 tiles; /root/reference/src/tracking/utils/metadata.py:96-97,
    src/render/renderer.py:13-14) and the demo's 640x480 with the Gaussian count of assets/demo/gs_orig.splat -- one ring camera,
    forward + backward through the drop-in module (the C++ autograd node) against the oracle, lists bit-exact."""
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(P, device=dev)
    cam = synth_ring_cameras(4, W, H, device=dev)[1]
    with torch.no_grad():
        rv = {k: v.detach().cpu().numpy() for k, v in params2rendervar(params).items()}
    ocam = OracleCamera(H, W, cam.tanfovx, cam.tanfovy, cam.bg.cpu().numpy(), 1.0, cam.viewmatrix.cpu().numpy().reshape(-1),
                        cam.projmatrix.cpu().numpy().reshape(-1), 0, cam.campos.cpu().numpy())
    g = dict(means3D=rv["means3D"], scales=rv["scales"], rotations=rv["rotations"], opacities=rv["opacities"], colors_precomp=rv["colors_precomp"])
    o2 = _check_against_oracle(ocam, g, dev, seed=21, nthreads=os.cpu_count() or 8, backward=True)
    print("num_rendered", o2.num_rendered)


def test_no_grad_render_of_trainable_parameters_is_forward_only_and_equal(dev):
    """an evaluation render under torch.no_grad() with parameters that require gradients must take the untracked forward
    (decided before the autograd node is built) and give the same image as the tracked one; a render with gradients enabled still
    differentiates."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(3000, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, 160, 96, device=dev)[0]
    rv = params2rendervar(params)
    assert any(v.requires_grad for v in rv.values())
    im_t, rad_t, dep_t = GaussianRasterizer(raster_settings=cam)(**rv)
    with torch.no_grad():
        im_n, rad_n, dep_n = GaussianRasterizer(raster_settings=cam)(**params2rendervar(params))
    assert not im_n.requires_grad and im_t.requires_grad
    assert torch.equal(im_t.detach(), im_n) and torch.equal(rad_t, rad_n) and torch.equal(dep_t.detach(), dep_n)
    im_t.sum().backward()
    assert params["means3D"].grad is not None and torch.isfinite(params["means3D"].grad).all()


def test_upstream_made_splat_lands_on_its_images():
    """VERDICT r04 item 6b -- sanity, not parity: tests/golden/demo_splat.npz holds the reference's assets/demo/gs_orig.splat, Gaussians its
    trainer fitted with the REAL rasterizer to four masked camera images, and those images.  tools/splat_sanity.py undoes save_to_splat's
    rotation, fits the three dropped numbers of the mean, renders the four cameras through the HIP path and repeats with the principal
    point shifted by half a pixel and a pixel: the nominal pixel-centre / axis conventions must fit the images best, cover the masks, and
    reach the PSNR measured when the script was committed (profiles/r05_splat_sanity.txt: 25.9 dB over the foreground region, 41.5 dB
    over the image; u8 colours / quaternions and untrained colours cap it there)."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "splat_sanity.py")], capture_output=True, text=True, timeout=600, check=True).stdout
    assert "best convention: nominal" in out, out[-1500:]
    nominal = float(re.search(r"foreground region \[[^\]]*\] dB \(mean ([0-9.]+)\)", out).group(1))
    shifted = [float(m) for m in re.findall(r"principal point [+-]0\.5, \+0\.0 px.*mean ([0-9.]+)", out)]
    cover = [float(x) for x in re.search(r"coverage of the mask \[([^\]]*)\]", out).group(1).split()]
    assert nominal >= 24.5 and min(cover) >= 0.98, (nominal, cover)
    assert len(shifted) == 2 and max(shifted) <= nominal - 1.0, (nominal, shifted)     # half a pixel in x costs 2.5 - 4 dB


def test_capacity_mode_forward_is_the_exact_forward(dev):
    """ABI 121 / VERDICT r04 item 4: GaussianRasterizer's forward queues both stages with the binning buffer sized from the previous call of the
    shape and reads the entry count afterwards (gsr_forward_capacity).  Images, radii, depth and every gradient must equal the exact
    (count first) path bit for bit; a scene that outgrew the estimate is rendered again before anything is returned."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    C_ = _LayerSwitches(dgr, dev)
    P, W, H = 30_000, 400, 304
    cam = synth_ring_cameras(4, W, H, device=dev)[2]
    small = synth_scene_params(P, device=dev, scale_lo=0.005, scale_hi=0.02)
    big = synth_scene_params(P, device=dev, scale_lo=0.02, scale_hi=0.06)      # same P, several times the entries
    dL = torch.tensor(np.random.default_rng(9).uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev)

    def run(params):
        leaves = {k: params[k].detach().clone().requires_grad_(True) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors")}
        im, rad, dep = GaussianRasterizer(raster_settings=cam)(**params2rendervar(leaves))
        (im * dL).sum().backward()
        torch.cuda.synchronize()
        return [im.detach(), rad, dep.detach()] + [leaves[k].grad for k in sorted(leaves)]
    try:
        C_.set_list_reuse(False)
        C_.set_capacity_mode(False)
        ref_small, ref_big = run(small), run(big)
        C_.set_capacity_mode(True)
        C_.forget_capacities()
        c0 = C_.capacity_stats()
        first = run(small)                         # no estimate yet: the exact path, which leaves one
        assert C_.capacity_stats()[0] == c0[0]
        second = run(small)                        # capacity mode
        c1 = C_.capacity_stats()
        assert c1[0] == c0[0] + 1 and c1[1] == c0[1]
        grown = run(big)                           # overflows the estimate: repeated through the exact path
        c2 = C_.capacity_stats()
        assert c2[0] == c1[0] + 1 and c2[1] == c1[1] + 1
        again = run(big)                           # fits now
        c3 = C_.capacity_stats()
        assert c3[0] == c2[0] + 1 and c3[1] == c2[1]
        shrunk = run(small)                        # a smaller scene in the larger buffers
        for got, ref, what in ((first, ref_small, "first"), (second, ref_small, "capacity"), (grown, ref_big, "overflow"), (again, ref_big, "again"),
                               (shrunk, ref_small, "shrunk")):
            for i, (a, b) in enumerate(zip(got, ref)):
                assert torch.equal(a, b), (what, i)
        with torch.no_grad():                      # forward-only calls take the same route
            rv = {k: v.detach() for k, v in params2rendervar(small).items()}
            im_n, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
        assert torch.equal(im_n, ref_small[0]) and C_.capacity_stats()[0] == c3[0] + 2
    finally:
        C_.set_list_reuse(True)
        C_.set_capacity_mode(True)


def test_capacity_mode_learns_the_two_render_pattern(dev):
    """With tile-list reuse on, a capacity-mode forward compares itself with its predecessor as well (the verdict words travel to pinned
    memory): the first colour / seg pair is noticed AFTER the seg render built its own lists; from the second pair on the seg render is
    predicted, goes the comparing way and shares the colour render's lists -- equal results throughout."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    C_ = _LayerSwitches(dgr, dev)
    P, W, H = 20_000, 320, 240
    cam = synth_ring_cameras(4, W, H, device=dev)[1]
    params = synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.04)
    g1, g2 = (torch.tensor(np.random.default_rng(s).uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev) for s in (1, 2))

    def pair(shift):
        leaves = {k: params[k].detach().clone().requires_grad_(True) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors", "seg_colors")}
        with torch.no_grad():
            leaves["means3D"] += shift            # another geometry every iteration, as training has it
        im, _, _ = GaussianRasterizer(raster_settings=cam)(**params2rendervar(leaves))
        seg, _, _ = GaussianRasterizer(raster_settings=cam)(**params2rendervar(leaves, colors_key="seg_colors"))
        ((im * g1).sum() + (seg * g2).sum()).backward()
        torch.cuda.synchronize()
        return [im.detach(), seg.detach()] + [leaves[k].grad for k in sorted(leaves)]
    try:
        C_.set_list_reuse(True)
        C_.set_capacity_mode(False)
        C_.drop_list_cache()
        ref = [pair(0.001 * i) for i in range(4)]
        C_.set_capacity_mode(True)
        C_.forget_capacities()
        C_.drop_list_cache()
        h0, s0 = C_.list_reuse_hits(), C_.capacity_stats()
        got = [pair(0.001 * i) for i in range(4)]
        h1, s1 = C_.list_reuse_hits(), C_.capacity_stats()
        for a, b in zip(got, ref):
            for i, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x, y), i
        # pair 0: colour exact (no estimate), seg capacity + noticed late; pairs 1..3: colour capacity, seg predicted -> shared lists
        assert s1[2] - s0[2] == 1, (s0, s1)
        assert h1 - h0 == 3, (h0, h1)
        assert s1[0] - s0[0] == 4 and s1[1] == s0[1], (s0, s1)
    finally:
        C_.set_list_reuse(True)
        C_.set_capacity_mode(True)


def test_capturing_the_drop_in_call_fails_at_once(dev):
    """The forward needs the entry count on the host (upstream reads it back too): inside a stream capture it never arrives.  The call
    must say so immediately instead of polling for it, and the stream must be usable afterwards."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    params = synth_scene_params(2000, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, 128, 96, device=dev)[0]
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
        ref, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
        ref2, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        import time
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    GaussianRasterizer(raster_settings=cam)(**rv)
        assert time.perf_counter() - t0 < 5.0
        torch.cuda.synchronize()
        again, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
    assert torch.equal(ref, again) and torch.equal(ref, ref2)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GSR_MV_SOAK", "6"))))
def test_random_call_sequences_through_the_stateful_layer(dev, seed):
    """The drop-in layer carries state from call to call (capacity estimates per shape, the previous forward's geometry for tile-list reuse, the
    two-render predictor).  Seeded random SEQUENCES of calls -- scenes of the same shape with very different entry counts, repeats (twins),
    two cameras, with and without gradients -- must return, call by call, exactly what a stateless layer returns."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    C_ = _LayerSwitches(dgr, dev)
    rng = np.random.default_rng(2500 + seed)
    P, W, H = int(rng.choice([2000, 15000])), int(rng.integers(60, 330)), int(rng.integers(60, 260))
    cams = synth_ring_cameras(3, W, H, device=dev)[:2]
    scenes = [synth_scene_params(P, seed=10 * seed + i, device=dev, scale_lo=lo, scale_hi=3 * lo) for i, lo in enumerate((0.004, 0.012, 0.03, 0.07))]
    dL = torch.tensor(rng.uniform(-1, 1, (3, H, W)).astype(np.float32), device=dev)
    keys = ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors")

    def run(si, ci, grad):
        if not grad:
            with torch.no_grad():
                rv = {k: v.detach() for k, v in params2rendervar(scenes[si]).items()}
                im, rad, dep = GaussianRasterizer(raster_settings=cams[ci])(**rv)
            torch.cuda.synchronize()
            return [im, rad, dep]
        leaves = {k: scenes[si][k].detach().clone().requires_grad_(True) for k in keys}
        im, rad, dep = GaussianRasterizer(raster_settings=cams[ci])(**params2rendervar(leaves))
        (im * dL).sum().backward()
        torch.cuda.synchronize()
        return [im.detach(), rad, dep.detach()] + [leaves[k].grad for k in sorted(leaves)]
    try:
        C_.set_list_reuse(False)
        C_.set_capacity_mode(False)
        ref = {(si, ci, gr): run(si, ci, gr) for si in range(len(scenes)) for ci in range(2) for gr in (False, True)}
        C_.set_list_reuse(True)
        C_.set_capacity_mode(True)
        C_.forget_capacities()
        prev = None
        for step in range(40):
            if prev is not None and rng.uniform() < 0.35:
                si, ci = prev                                # a twin of the previous call (the colour / mask pattern)
            else:
                si, ci = int(rng.integers(0, len(scenes))), int(rng.integers(0, 2))
            gr = bool(rng.integers(0, 2))
            got = run(si, ci, gr)
            for i, (a, b) in enumerate(zip(got, ref[(si, ci, gr)])):
                assert torch.equal(a, b), (seed, step, si, ci, gr, i)
            prev = (si, ci)
    finally:
        C_.set_list_reuse(True)
        C_.set_capacity_mode(True)


def test_layer_state_is_owned_by_the_module(dev):
    """no global state.  What the torch C++ layer remembers between calls (tile lists of the last forward, capacities,
    the twin predictor) lives in ONE object per device that the Python module owns: inspectable, resettable, droppable; upstream's entry
    points called WITHOUT a state are pure functions of their arguments."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
    assert dgr._C is not None
    dgr.reset_layer_states()
    st = dgr.layer_state(dev)
    assert st is dgr.layer_state(dev) and st.stats()["cached_entries"] == 0 and st.stats()["capacities"] == 0
    params = synth_scene_params(3000, device=dev, scale_lo=0.02, scale_hi=0.08)
    cam = synth_ring_cameras(4, 160, 128, device=dev)[0]
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
        a = GaussianRasterizer(raster_settings=cam)(**rv)
        s1 = st.stats()
        assert s1["cached_entries"] > 0 and s1["capacities"] == 1
        b = GaussianRasterizer(raster_settings=cam)(**rv)          # the same frame again: served from the remembered lists or re-binned -- same bits
        st.reset()
        assert st.stats()["cached_entries"] == 0 and st.stats()["capacities"] == 0
        c = GaussianRasterizer(raster_settings=cam)(**rv)
        e = rv["means3D"].new_empty(0)
        h0 = st.stats()
        up = dgr._C.rasterize_gaussians(cam.bg, rv["means3D"], rv["colors_precomp"], rv["opacities"], rv["scales"], rv["rotations"], 1.0, e,
                                        cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, 128, 160, e, 0, cam.campos, False)
        assert st.stats() == h0, "a call without a state must not touch any"
    for x, y in ((a, b), (a, c)):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    assert torch.equal(up[1], a[0]) and torch.equal(up[3], a[1]) and torch.equal(up[2], a[2])
    dgr.reset_layer_states()
    assert dgr.layer_state(dev) is not st
