"""Host-side callers of the path, pinned by vectors captured from the imported Python reference
(tests/golden/gen_reference_goldens.py -> reference_host.npz)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ref(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_host.npz"))


def test_setup_camera_matches_reference(ref):
    from gsdyn import setup_camera
    for row_in, row_out in zip(ref["cam_in"], ref["cam_out"]):
        w2c, k = row_in[:16].reshape(4, 4), row_in[16:25].reshape(3, 3)
        w, h, near, far = int(row_in[25]), int(row_in[26]), row_in[27], row_in[28]
        cam = setup_camera(w, h, k, w2c, near=near, far=far, device="cpu")
        got = np.concatenate([cam.viewmatrix.reshape(-1).numpy(), cam.projmatrix.reshape(-1).numpy(),
                              cam.campos.reshape(-1).numpy(), [cam.tanfovx, cam.tanfovy]])
        np.testing.assert_allclose(got, row_out, rtol=1e-6, atol=1e-6)
        assert cam.viewmatrix.shape == (1, 4, 4) and cam.sh_degree == 0 and cam.prefiltered is False


def test_setup_camera_known_values():
    """K=[[600,0,400],[0,600,400]], w2c = I with tz=3, 800x800, near=1."""
    from gsdyn import setup_camera
    w2c = np.eye(4); w2c[2, 3] = 3
    cam = setup_camera(800, 800, [[600, 0, 400], [0, 600, 400], [0, 0, 1]], w2c, near=1.0, far=100, device="cpu")
    pm = cam.projmatrix[0].numpy()
    np.testing.assert_allclose(pm[0], [1.5, 0, 0, 0], atol=1e-6)
    np.testing.assert_allclose(pm[2], [0, 0, 1.010101, 1], atol=1e-5)
    np.testing.assert_allclose(pm[3], [0, 0, 2.020202, 3], atol=1e-5)
    np.testing.assert_allclose(cam.campos.numpy(), [0, 0, -3], atol=1e-6)


def test_params2rendervar_matches_reference(ref):
    from gsdyn import params2rendervar
    params = {k: torch.tensor(ref["p2r_in_" + k]) for k in ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")}
    rv = params2rendervar(params)
    for k in ("rotations", "opacities", "scales", "means2D", "colors_precomp"):
        np.testing.assert_allclose(rv[k].detach().numpy(), ref["p2r_out_" + k], rtol=1e-6, atol=1e-7)
    assert rv["means2D"].requires_grad and not rv["means2D"].is_leaf


def test_scalar_losses_match_reference(ref):
    from gsdyn import losses as L
    a, b, w = (torch.tensor(ref[k]) for k in ("loss_a", "loss_b", "loss_w"))
    np.testing.assert_allclose(L.l1_loss_v1(a, b).numpy(), ref["l1_v1"], rtol=1e-6)
    np.testing.assert_allclose(L.l1_loss_v2(a, b).numpy(), ref["l1_v2"], rtol=1e-6)
    np.testing.assert_allclose(L.weighted_l2_loss_v1(a[..., 0], b[..., 0], w).numpy(), ref["wl2_v1"], rtol=1e-6)
    np.testing.assert_allclose(L.weighted_l2_loss_v2(a, b, w).numpy(), ref["wl2_v2"], rtol=1e-6)
    q1, q2 = torch.tensor(ref["q1"]), torch.tensor(ref["q2"])
    np.testing.assert_allclose(L.quat_mult(q1, q2).numpy(), ref["quat_mult"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(L.build_rotation(q1).numpy(), ref["build_rotation"], rtol=1e-5, atol=1e-6)


def test_ssim_psnr_match_reference(ref):
    from gsdyn import losses as L
    im1 = torch.tensor(ref["ssim_im1"], requires_grad=True)
    im2 = torch.tensor(ref["ssim_im2"])
    s = L.calc_ssim(im1, im2)
    s.backward()
    np.testing.assert_allclose(s.detach().numpy(), ref["ssim"], rtol=1e-5)
    np.testing.assert_allclose(im1.grad.numpy(), ref["ssim_grad"], rtol=1e-4, atol=1e-8)
    np.testing.assert_allclose(L.calc_psnr(im1.detach(), im2).numpy(), ref["psnr"], rtol=1e-6)
    comb = 0.8 * L.l1_loss_v1(im1.detach(), im2) + 0.2 * (1.0 - L.calc_ssim(im1.detach(), im2))
    np.testing.assert_allclose(comb.numpy(), ref["im_term"], rtol=1e-6)


def test_rigidity_block_matches_reference(ref, monkeypatch):
    """t>0 terms of get_loss (/root/reference/src/tracking/train_utils.py:198-232): same values and
    gradients as the reference's helpers produce on the same seeded tensors (renders stubbed out)."""
    import oracle_double
    oracle_double.install(monkeypatch)
    from gsdyn import LossWeights, get_loss, synth_ring_cameras
    n = ref["rig_fg_pts"].shape[0]
    P = n + 1  # n foreground Gaussians + one background Gaussian (the reference's bg term averages over them)
    far = torch.tensor([[50.0, 50.0, 50.0]])
    seg = torch.tensor([[1.0, 0, 0]]).repeat(P, 1)
    seg[-1] = torch.tensor([0.0, 0, 1.0])
    params = {
        "means3D": torch.nn.Parameter(torch.cat([torch.tensor(ref["rig_fg_pts"]), far])),
        "unnorm_rotations": torch.nn.Parameter(torch.cat([torch.tensor(ref["rig_fg_rot_un"]), torch.tensor([[1.0, 0, 0, 0]])])),
        "rgb_colors": torch.zeros(P, 3), "seg_colors": seg,
        "logit_opacities": torch.full((P, 1), -20.0), "log_scales": torch.full((P, 3), -5.0),
        "cam_m": torch.zeros(50, 3), "cam_c": torch.zeros(50, 3)}
    cam = synth_ring_cameras(4, 32, 32, device="cpu")[0]
    variables = {"max_2D_radius": torch.zeros(P), "prev_inv_rot_fg": torch.tensor(ref["rig_prev_inv"]),
                 "neighbor_indices": torch.tensor(ref["rig_nbr"]).long(), "prev_offset": torch.tensor(ref["rig_prev_offset"]),
                 "neighbor_weight": torch.tensor(ref["rig_nw"]), "neighbor_dist": torch.tensor(ref["rig_nd"]),
                 "init_bg_pts": far.clone(), "init_bg_rot": torch.tensor([[1.0, 0, 0, 0]])}
    data = dict(cam=cam, im=torch.zeros(3, 32, 32), seg=torch.zeros(3, 32, 32), id=0)
    w = LossWeights(im=0.0, seg=0.0, bg=0.0, soft_col_cons=0.0)  # isolate rigid/rot/iso/floor
    loss, _ = get_loss(params, data, variables, False, w)
    loss.backward()
    np.testing.assert_allclose(loss.item(), ref["rig_losses"][4], rtol=1e-5)
    np.testing.assert_allclose(params["means3D"].grad.numpy()[:n], ref["rig_grad_pts"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(params["unnorm_rotations"].grad.numpy()[:n], ref["rig_grad_rot"], rtol=1e-4, atol=1e-5)


def test_batched_get_loss_equals_sum_of_per_camera_get_loss(monkeypatch):
    """get_loss_views (one rasterizer call: colour + segmentation renders of all cameras as 2 V views with per-view
    colours) == sum of get_loss over the cameras: value, parameter gradients, densification bookkeeping."""
    import oracle_double
    oracle_double.install(monkeypatch)
    from gsdyn import LossWeights, get_loss, get_loss_views, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.step import make_rigidity_variables
    P, W, H, V = 60, 48, 32, 3
    cams = synth_ring_cameras(V, W, H, device="cpu")
    datas = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=3 + i, device="cpu")
        datas.append(dict(cam=cam, im=im, seg=seg, id=i))
    w = LossWeights()

    def fresh():
        params = synth_scene_params(P, device="cpu", scale_lo=0.05, scale_hi=0.25)
        params["seg_colors"].requires_grad_(True)
        variables = dict(make_rigidity_variables(params, num_knn=5), max_2D_radius=torch.zeros(P))
        with torch.no_grad():
            params["means3D"].add_(0.01 * torch.randn(P, 3, generator=torch.Generator().manual_seed(1)))
        return params, variables

    pa, va = fresh()
    total_a, g2_a, seen_a = 0.0, [], []
    for d in datas:
        loss, va = get_loss(pa, d, va, False, w)
        loss.backward()
        total_a += loss.item()
        g2_a.append(va["means2D"].grad.clone()); seen_a.append(va["seen"].clone())
    pb, vb = fresh()
    loss_b, vb, aux = get_loss_views(pb, datas, vb, False, w)
    loss_b.backward()
    np.testing.assert_allclose(loss_b.item(), total_a, rtol=1e-5)
    for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "seg_colors", "cam_m", "cam_c"):
        ga, gb = pa[k].grad, pb[k].grad
        assert gb is not None, k
        np.testing.assert_allclose(gb.numpy(), ga.numpy(), rtol=2e-4, atol=1e-6 * float(ga.abs().max()) + 1e-12, err_msg=k)
    for v in range(V):
        np.testing.assert_allclose(aux["means2D"].grad[2 * v].numpy(), g2_a[v].numpy(), rtol=1e-4, atol=1e-9)
        assert torch.equal(aux["radii"][v] > 0, seen_a[v])
    np.testing.assert_allclose(vb["max_2D_radius"].numpy(), va["max_2D_radius"].numpy())


def test_synth_scene_is_deterministic_and_shaped():
    from gsdyn import synth_ring_cameras, synth_scene_params, synth_targets
    a, b = synth_scene_params(100, device="cpu"), synth_scene_params(100, device="cpu")
    assert set(a) == {"means3D", "rgb_colors", "seg_colors", "unnorm_rotations", "logit_opacities", "log_scales", "cam_m", "cam_c"}
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert a["means3D"].shape == (100, 3) and a["cam_m"].shape == (50, 3) and not a["rgb_colors"].requires_grad
    cams = synth_ring_cameras(4, 80, 60, device="cpu")
    assert len(cams) == 4 and abs(cams[0].tanfovx - 0.5) < 1e-9 and abs(cams[0].tanfovy - 60 / 160) < 1e-9
    im, seg = synth_targets(80, 60, device="cpu")
    assert im.shape == (3, 60, 80) and seg.shape == (3, 60, 80) and torch.all(seg[0] + seg[2] == 1)


def test_fused_pieces_have_host_fallbacks_or_fail_loudly():
    """CPU tensors: FusedAdam steps through torch's own Adam, views_image_loss / activate evaluate the reference's torch formulas,
    and the direct step -- which has no meaning without the library -- refuses to run."""
    import numpy as np
    import pytest
    import torch
    from gsdyn import LossWeights, loss_and_grads_views, synth_scene_params
    from gsdyn import losses as L
    from gsdyn.optim import FusedAdam
    g = torch.Generator().manual_seed(3)
    a = torch.nn.Parameter(torch.randn(7, 3, generator=g))
    b = torch.nn.Parameter(a.detach().clone())
    oa, ob = torch.optim.Adam([{"params": [a], "lr": 0.01}], lr=0.0, eps=1e-15), FusedAdam([{"params": [b], "lr": 0.01}], lr=0.0, eps=1e-15)
    for _ in range(3):
        gr = torch.randn(7, 3, generator=g)
        a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
    assert torch.equal(a, b)
    r = torch.rand(2, 3, 20, 24, generator=g, requires_grad=True)
    t = [torch.rand(3, 20, 24, generator=g) for _ in range(2)]
    m, c = torch.zeros(4, 3, requires_grad=True), torch.zeros(4, 3, requires_grad=True)
    total, per = L.views_image_loss(r, t, [2, -1], [50.0, 200.0], m, c)
    want = 50.0 * L.image_loss(torch.exp(m[2])[:, None, None] * r[0] + c[2][:, None, None], t[0]) + 200.0 * L.image_loss(r[1], t[1])
    np.testing.assert_allclose(float(total.detach()), float(want.detach()), rtol=1e-6)
    assert per.shape == (2,)
    u = torch.randn(5, 4, generator=g)
    rot, op, sc = L.activate(u, torch.zeros(5, 1), torch.zeros(5, 3))
    assert torch.allclose(rot.norm(dim=1), torch.ones(5)) and torch.all(op == 0.5) and torch.all(sc == 1.0)
    params = synth_scene_params(10, device="cpu")
    with pytest.raises(RuntimeError, match="HIP device"):
        loss_and_grads_views(params, [], {}, True, LossWeights())


def test_rt_to_w2c_inverts_the_camera_pose(golden_dir):
    """``Rt_to_w2c`` (/root/reference/src/real_world/gs/trainer.py:15-18) on the demo cameras: [R t; 0 1] is the camera's pose, the
    result is its inverse -- so the camera centre recovered from the world-to-camera matrix is t itself."""
    from gsdyn import Rt_to_w2c
    z = np.load(os.path.join(golden_dir, "demo_scene.npz"))
    for R, t in zip(z["R_list"], z["t_list"]):
        w2c = Rt_to_w2c(R, t)
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = R, t
        np.testing.assert_allclose(w2c @ pose, np.eye(4), atol=1e-12)
        np.testing.assert_allclose(np.linalg.inv(w2c)[:3, 3], t, atol=1e-12)
