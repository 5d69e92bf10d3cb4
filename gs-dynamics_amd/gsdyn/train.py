"""Corrected per-timestep optimisation loop of the tracking stage.

The reference's driver, /root/reference/src/tracking/train_gs.py:10-46, does not run as shipped (it imports a
``densify`` that ``train_utils`` does not define and calls ``train()`` with 9 positional arguments against a
15-parameter signature -- SURVEY.md Appendix C).  This module is the working equivalent of its loop body:

    for t in timesteps:
        (t > 0) initialize_per_timestep            # velocity extrapolation, train_utils.py:331-351
        for i in range(iters):                     # 10 000 at t = 0, 2 000 after (train_gs.py:25)
            batch = views sampled uniformly WITH replacement (the effective behaviour of get_batch,
                    train_utils.py:82-86: its 'todo' list is rebound locally, so it never empties)
            loss = get_loss(...); loss.backward(); optimizer.step(); optimizer.zero_grad()
        (t == 0) initialize_post_first_timestep    # neighbour tensors, lr freeze, train_utils.py:354-374
        params2cpu -> save_params                  # np.savez contract of helpers.py:141-158

With ``views_per_step > 1`` (or world_size > 1) one optimiser step sums the gradients of several views -- the
view-sharded data-parallel step of ``gsdyn.dp`` -- instead of the reference's one view per step.  Densification
(/root/reference/src/tracking/external.py:229-299) is ``gsdyn.densify`` (golden-tested against the imported
reference); pass ``density_control`` to switch it on for the first timestep.
"""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .dp import ViewShardedStep, init_variables
from .step import LossWeights, make_rigidity_variables, report_psnr


def initialize_per_timestep(params, variables, optimizer):
    """Constant-velocity initial guess for the new timestep (/root/reference/src/tracking/train_utils.py:331-351)."""
    with torch.no_grad():
        pts = params["means3D"]
        rot = torch.nn.functional.normalize(params["unnorm_rotations"])
        new_pts = pts + (pts - variables["prev_pts"])
        new_rot = torch.nn.functional.normalize(rot + (rot - variables["prev_rot"]))
        is_fg = params["seg_colors"][:, 0] > 0.5
        inv = rot[is_fg].clone()
        inv[:, 1:] = -inv[:, 1:]
        fg_pts = pts[is_fg]
        variables["prev_inv_rot_fg"] = inv.detach()
        variables["prev_offset"] = (fg_pts[variables["neighbor_indices"]] - fg_pts[:, None]).detach()
        variables["prev_col"] = params["rgb_colors"].detach().clone()
        variables["prev_pts"] = pts.detach().clone()
        variables["prev_rot"] = rot.detach().clone()
        # overwrite the parameter values in place and reset their Adam moments (update_params_and_optimizer,
        # /root/reference/src/tracking/external.py:145-157, replaces the tensors and zeroes exp_avg / exp_avg_sq)
        for name, value in (("means3D", new_pts), ("unnorm_rotations", new_rot)):
            p = params[name]
            p.data.copy_(value)
            st = optimizer.state.get(p, None)
            if st:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
    return params, variables


def initialize_post_first_timestep(params, variables, optimizer, num_knn: int = 20):
    """Neighbour tensors + frozen learning rates after t = 0 (/root/reference/src/tracking/train_utils.py:354-374).
    The reference builds the kNN with Open3D on the host; here a dense torch top-k over the foreground points."""
    variables.update(make_rigidity_variables(params, num_knn=num_knn))
    with torch.no_grad():
        variables["prev_pts"] = params["means3D"].detach().clone()
        variables["prev_rot"] = torch.nn.functional.normalize(params["unnorm_rotations"]).detach().clone()
    for group in optimizer.param_groups:
        if group["name"] in ("logit_opacities", "log_scales", "cam_m", "cam_c", "rgb_colors"):
            group["lr"] = 0.0
    return variables


def params2cpu(params, is_initial_timestep: bool) -> Dict[str, np.ndarray]:
    keys = params.keys() if is_initial_timestep else ("means3D", "rgb_colors", "unnorm_rotations")
    return {k: params[k].detach().cpu().contiguous().numpy() for k in keys}


def save_params(output_params: List[Dict[str, np.ndarray]], path: str) -> str:
    """``params.npz`` with per-timestep keys stacked to [T, P, .] and static keys kept [P, .] -- the contract
    /root/reference/src/render/dynamics_module.py:177-184 reads back."""
    to_save = {}
    later = output_params[1].keys() if len(output_params) > 1 else ()
    for k in output_params[0].keys():
        to_save[k] = np.stack([p[k] for p in output_params]) if k in later else output_params[0][k]
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    np.savez(path, **to_save)
    return path


def train_timestep(params, variables, optimizer, dataset: Sequence[dict], iters: int, is_initial_timestep: bool,
                   weights: LossWeights = LossWeights(), views_per_step: int = 1, seed: Optional[int] = None,
                   progress_every: int = 0, group=None, density_control: Optional[dict] = None):
    """One timestep of the loop.  ``dataset``: list of dict(cam, im, seg, id).  Returns the list of PSNR probes.
    ``density_control`` (first timestep only): dict(remove_thresh, remove_thresh_5k, scale_scene_radius) -- the reference's
    adaptive density control (gsdyn/densify.py); ``variables['scene_radius']`` must be set."""
    rng = random.Random(seed)
    stepper = ViewShardedStep(params, optimizer, weights, group=group,
                              density_control=density_control if is_initial_timestep else None)
    psnrs = []
    for i in range(iters):
        batch = [dataset[rng.randint(0, len(dataset) - 1)] for _ in range(views_per_step * stepper.world)]
        _, variables = stepper(batch, variables, is_initial_timestep=is_initial_timestep, iteration=i)
        if progress_every and i % progress_every == 0:
            psnrs.append(float(report_psnr(params, dataset[0])))
    return psnrs


def train(params, optimizer, timesteps: Sequence[Sequence[dict]], iters_first: int = 10000, iters_next: int = 2000,
          weights: LossWeights = LossWeights(), num_knn: int = 20, views_per_step: int = 1, out_path: Optional[str] = None,
          seed: Optional[int] = 0, density_control: Optional[dict] = None, scene_radius: Optional[float] = None):
    """The whole loop over timesteps (``timesteps[t]`` = that timestep's views).  Returns (params, variables, outputs).
    ``density_control`` + ``scene_radius`` switch on the adaptive density control of the first timestep."""
    P = params["means3D"].shape[0]
    variables = init_variables(P, params["means3D"].device)
    if scene_radius is not None:
        variables["scene_radius"] = float(scene_radius)
    outputs = []
    for t, dataset in enumerate(timesteps):
        first = t == 0
        if not first:
            params, variables = initialize_per_timestep(params, variables, optimizer)
        train_timestep(params, variables, optimizer, dataset, iters_first if first else iters_next, first, weights,
                       views_per_step=views_per_step, seed=None if seed is None else seed + t,
                       density_control=density_control if first else None)
        outputs.append(params2cpu(params, first))
        if first:
            variables = initialize_post_first_timestep(params, variables, optimizer, num_knn)
    if out_path:
        save_params(outputs, out_path)
    return params, variables, outputs
