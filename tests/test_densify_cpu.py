"""Adaptive density control (gsdyn/densify.py) against golden vectors captured from the imported reference
(tests/golden/gen_densify_goldens.py)."""
import os

import numpy as np
import pytest
import torch

KEYS = ("means3D", "rgb_colors", "seg_colors", "unnorm_rotations", "logit_opacities", "log_scales", "cam_m", "cam_c")
LRS = {"means3D": 1e-3, "rgb_colors": 0.0, "seg_colors": 0.0, "unnorm_rotations": 1e-3, "logit_opacities": 0.05,
       "log_scales": 1e-3, "cam_m": 1e-4, "cam_c": 1e-4}


def _load(z, it):
    params = {k: torch.nn.Parameter(torch.tensor(z[f"i{it}_in_p_{k}"])) for k in KEYS}
    opt = torch.optim.Adam([{"params": [v], "name": k, "lr": LRS[k]} for k, v in params.items()], lr=0.0, eps=1e-15)
    for k, v in params.items():
        v.grad = torch.zeros_like(v)
    opt.step()                                   # creates the Adam state (lr * 0-gradient: parameters unchanged)
    opt.zero_grad(set_to_none=True)
    for k, v in params.items():
        v.data.copy_(torch.tensor(z[f"i{it}_in_p_{k}"]))
        opt.state[v]["exp_avg"] = torch.tensor(z[f"i{it}_in_m_{k}"])
        opt.state[v]["exp_avg_sq"] = torch.tensor(z[f"i{it}_in_v_{k}"])
    P = params["means3D"].shape[0]
    m2 = torch.zeros(P, 3, requires_grad=True)
    m2.grad = torch.tensor(z[f"i{it}_m2grad"])
    variables = {k: torch.tensor(z[f"i{it}_in_{k}"]) for k in ("means2D_gradient_accum", "denom", "max_2D_radius")}
    variables.update(scene_radius=2.0, seen=torch.tensor(z[f"i{it}_seen"]), means2D=m2)
    return params, variables, opt


@pytest.mark.parametrize("it", [600, 3000])
def test_densify_matches_reference(golden_dir, it):
    from gsdyn.densify import densify
    z = np.load(os.path.join(golden_dir, "densify_host.npz"))
    params, variables, opt = _load(z, it)
    torch.manual_seed(1234)                      # the split offsets are the one random draw
    params, variables, n = densify(params, variables, opt, it, 0.005, 0.25, 0.05)
    assert n == int(z[f"i{it}_n"][0]) and n != z[f"i{it}_in_p_means3D"].shape[0]
    for k in KEYS:
        np.testing.assert_allclose(params[k].detach().numpy(), z[f"i{it}_out_p_{k}"], rtol=1e-6, atol=1e-7, err_msg=k)
        st = opt.state[params[k]]
        np.testing.assert_allclose(st["exp_avg"].numpy(), z[f"i{it}_out_m_{k}"], rtol=1e-6, atol=1e-9, err_msg=k)
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), z[f"i{it}_out_v_{k}"], rtol=1e-6, atol=1e-12, err_msg=k)
        assert params[k].requires_grad and _group_param(opt, k) is params[k]
    for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
        np.testing.assert_allclose(variables[k].numpy(), z[f"i{it}_out_{k}"], err_msg=k)


def _group_param(opt, name):
    return next(g for g in opt.param_groups if g["name"] == name)["params"][0]


def test_densify_is_a_no_op_between_density_steps(golden_dir):
    from gsdyn.densify import densify
    z = np.load(os.path.join(golden_dir, "densify_host.npz"))
    params, variables, opt = _load(z, 600)
    before = {k: v.detach().clone() for k, v in params.items()}
    acc0 = variables["means2D_gradient_accum"].clone()
    params, variables, n = densify(params, variables, opt, 601, 0.005, 0.25, 0.05)
    assert n == before["means3D"].shape[0] and all(torch.equal(params[k], before[k]) for k in KEYS)
    seen = variables["seen"]
    assert torch.all(variables["means2D_gradient_accum"][seen] >= acc0[seen]) and torch.equal(variables["means2D_gradient_accum"][~seen], acc0[~seen])
    params, variables, n2 = densify(params, variables, opt, 7000, 0.005, 0.25, 0.05)   # past iteration 5000: nothing at all
    assert n2 == n
