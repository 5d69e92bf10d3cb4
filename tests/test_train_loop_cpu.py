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


def test_density_control_inside_the_step(monkeypatch):
    """A first-timestep step at a density iteration (600): the stepper accumulates the screen-space gradient statistics,
    runs densify between backward and the optimiser step, rebuilds its gradient bucket around the new parameters and the
    next step still works."""
    import oracle_double
    oracle_double.install(monkeypatch)
    from gsdyn import LossWeights, initialize_optimizer, synth_ring_cameras, synth_scene_params, synth_targets
    from gsdyn.dp import ViewShardedStep, init_variables
    P, W, H = 60, 32, 24
    params = synth_scene_params(P, device="cpu", scale_lo=0.05, scale_hi=0.3)
    cams = synth_ring_cameras(2, W, H, device="cpu")
    views = []
    for i, cam in enumerate(cams):
        im, seg = synth_targets(W, H, seed=3 + i, device="cpu")
        views.append(dict(cam=cam, im=im, seg=seg, id=i))
    opt = initialize_optimizer(params, scene_radius=4.0)
    variables = init_variables(P, "cpu")
    variables["scene_radius"] = 4.0
    step = ViewShardedStep(params, opt, LossWeights(), density_control=dict(remove_thresh=0.005, remove_thresh_5k=0.25,
                                                                           scale_scene_radius=0.01))
    step(views, variables, is_initial_timestep=True, iteration=10)          # builds Adam moments, accumulates statistics
    assert params["means3D"].shape[0] == P and float(variables["denom"].sum()) > 0
    variables["means2D_gradient_accum"] += 1.0                              # make every seen Gaussian a candidate
    step(views, variables, is_initial_timestep=True, iteration=600)
    n = params["means3D"].shape[0]
    assert n != P and all(params[k].shape[0] == n for k in ("rgb_colors", "seg_colors", "unnorm_rotations", "logit_opacities", "log_scales"))
    assert variables["denom"].shape[0] == n and float(variables["denom"].sum()) == 0.0
    assert all(p is params[k] for k, p in zip(step.bucket.names, step.bucket.params))
    total, variables = step(views, variables, is_initial_timestep=True, iteration=601)   # the grown cloud trains on
    assert torch.isfinite(total) and variables["denom"].shape[0] == n
