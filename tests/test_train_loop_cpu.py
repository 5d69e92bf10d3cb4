"""Plumbing of the corrected per-timestep loop (gsdyn.train) on CPU with the oracle-backed test double:
two timesteps run, parameters move, and the params.npz contract of the reference holds
(/root/reference/src/tracking/helpers.py:141-158, read back at /root/reference/src/render/dynamics_module.py:177-184)."""
import numpy as np
import torch


def test_two_timesteps_and_npz_contract(monkeypatch, tmp_path):
    import oracle_double
    oracle_double.install(monkeypatch)
    from gsdyn import initialize_optimizer, synth_ring_cameras, synth_scene_params, synth_targets, train
    P, W, H = 40, 32, 24
    params = synth_scene_params(P, device="cpu", scale_lo=0.05, scale_hi=0.3)
    cams = synth_ring_cameras(2, W, H, device="cpu")
    views = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=3 + i, device="cpu")
        views.append(dict(cam=cam, im=im, seg=seg, id=i))
    opt = initialize_optimizer(params, scene_radius=4.0)
    before = params["means3D"].detach().clone()
    out = str(tmp_path / "params.npz")
    params, variables, outputs = train(params, opt, [views, views, views], iters_first=3, iters_next=2, num_knn=4,
                                       views_per_step=2, out_path=out)
    assert not torch.equal(before, params["means3D"].detach())
    assert {"neighbor_indices", "neighbor_weight", "neighbor_dist", "prev_pts", "prev_rot", "prev_offset"} <= set(variables)
    z = np.load(out)
    assert z["means3D"].shape == (3, P, 3) and z["unnorm_rotations"].shape == (3, P, 4) and z["rgb_colors"].shape == (3, P, 3)
    assert z["logit_opacities"].shape == (P, 1) and z["log_scales"].shape == (P, 3) and z["seg_colors"].shape == (P, 3)
    frozen = {g["name"]: g["lr"] for g in opt.param_groups}
    assert frozen["logit_opacities"] == 0.0 and frozen["log_scales"] == 0.0 and frozen["means3D"] > 0.0
