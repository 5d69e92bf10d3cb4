"""Adaptive density control of the first timestep -- the consumer of the rasterizer's
``means2D`` gradients, ``radii`` and ``max_2D_radius``: /root/reference/src/tracking/external.py:138-299, called once per
iteration at t = 0 between ``loss.backward()`` and ``optimizer.step()`` (/root/reference/src/tracking/train_gs.py:31-37).

Same decisions, thresholds and random draws as the reference (one ``torch.normal`` call for the split offsets, so equal
seeds give equal clouds -- which is also what keeps data-parallel replicas identical), organised around two primitives:
``_append`` (grow every per-Gaussian parameter and its Adam moments) and ``_keep`` (select rows of them).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .losses import build_rotation

GRAD_THRESH = 0.0002          # screen-space gradient (dL/dNDC units) above which a Gaussian is cloned or split
_GLOBAL = ("cam_m", "cam_c")  # parameters that are not per Gaussian


def _group(optimizer, name):
    return next(g for g in optimizer.param_groups if g["name"] == name)


def _swap(params: Dict, optimizer, name: str, value: torch.Tensor, exp_avg=None, exp_avg_sq=None):
    """Install ``value`` as the parameter ``name`` (a fresh leaf), carrying over / replacing its Adam moments."""
    g = _group(optimizer, name)
    old = g["params"][0]
    state = optimizer.state.pop(old, None)
    new = torch.nn.Parameter(value.requires_grad_(True))
    g["params"][0] = new
    if state is not None:
        if exp_avg is not None:
            state["exp_avg"], state["exp_avg_sq"] = exp_avg, exp_avg_sq
        optimizer.state[new] = state
    params[name] = new


def _append(params: Dict, optimizer, rows: Dict[str, torch.Tensor]):
    for k, v in rows.items():
        old = _group(optimizer, k)["params"][0]
        st = optimizer.state.get(old, None)
        if st is not None:
            _swap(params, optimizer, k, torch.cat((old, v), 0), torch.cat((st["exp_avg"], torch.zeros_like(v)), 0),
                  torch.cat((st["exp_avg_sq"], torch.zeros_like(v)), 0))
        else:
            _swap(params, optimizer, k, torch.cat((old, v), 0))


def _keep(params: Dict, variables: Dict, optimizer, keep: torch.Tensor):
    for k in [k for k in params if k not in _GLOBAL]:
        old = _group(optimizer, k)["params"][0]
        st = optimizer.state.get(old, None)
        if st is not None:
            _swap(params, optimizer, k, old[keep], st["exp_avg"][keep], st["exp_avg_sq"][keep])
        else:
            _swap(params, optimizer, k, old[keep])
    for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
        variables[k] = variables[k][keep]


def accumulate_mean2d_gradient(variables: Dict) -> Dict:
    """Per seen Gaussian: += |d loss / d means2D (x, y)| and += 1 (/root/reference/src/tracking/external.py:138-142)."""
    seen = variables["seen"]
    variables["means2D_gradient_accum"][seen] += torch.norm(variables["means2D"].grad[seen, :2], dim=-1)
    variables["denom"][seen] += 1
    return variables


@torch.no_grad()
def densify(params: Dict, variables: Dict, optimizer, i: int, remove_thresh: float, remove_thresh_5k: float,
            scale_scene_radius: float, accumulate: bool = True) -> Tuple[Dict, Dict, int]:
    """One call per iteration of the first timestep.  ``accumulate=False`` when the caller (``ViewShardedStep``) has
    already folded this iteration's gradient norms into the accumulators."""
    if i <= 5000:
        if accumulate:
            variables = accumulate_mean2d_gradient(variables)
        if i >= 500 and i % 100 == 0:
            dev = params["means3D"].device
            limit = scale_scene_radius * variables["scene_radius"]
            grads = variables["means2D_gradient_accum"] / variables["denom"]
            grads[grads.isnan()] = 0.0
            per_g = [k for k in params if k not in _GLOBAL]
            # clone: high gradient, small
            clone = (grads >= GRAD_THRESH) & (torch.exp(params["log_scales"]).max(1).values <= limit)
            _append(params, optimizer, {k: params[k][clone] for k in per_g})
            # split: high gradient, large (clones carry gradient 0, so only original rows qualify)
            n_now = params["means3D"].shape[0]
            padded = torch.zeros(n_now, device=dev)
            padded[:grads.shape[0]] = grads
            split = (padded >= GRAD_THRESH) & (torch.exp(params["log_scales"]).max(1).values > limit)
            n = 2
            rows = {k: params[k][split].repeat(n, 1) for k in per_g}
            stds = torch.exp(params["log_scales"])[split].repeat(n, 1)
            offsets = torch.normal(mean=torch.zeros((stds.shape[0], 3), device=dev), std=stds)
            rots = build_rotation(params["unnorm_rotations"][split]).repeat(n, 1, 1)
            rows["means3D"] = rows["means3D"] + torch.bmm(rots, offsets.unsqueeze(-1)).squeeze(-1)
            rows["log_scales"] = torch.log(torch.exp(rows["log_scales"]) / (0.8 * n))
            _append(params, optimizer, rows)
            n_now = params["means3D"].shape[0]
            for k in ("means2D_gradient_accum", "denom", "max_2D_radius"):
                variables[k] = torch.zeros(n_now, device=dev)
            gone = torch.cat((split, torch.zeros(n * int(split.sum()), dtype=torch.bool, device=dev)))
            _keep(params, variables, optimizer, ~gone)
            # prune: nearly transparent, and (from iteration 3000) very large
            thr = remove_thresh_5k if i == 5000 else remove_thresh
            drop = (torch.sigmoid(params["logit_opacities"]) < thr).squeeze(-1)
            if i >= 3000:
                drop = drop | (torch.exp(params["log_scales"]).max(1).values > 0.1 * variables["scene_radius"])
            _keep(params, variables, optimizer, ~drop)
        if i > 0 and i % 3000 == 0:   # opacity reset
            v = torch.full_like(params["logit_opacities"], 0.01)
            v = torch.log(v / (1 - v))
            _swap(params, optimizer, "logit_opacities", v, torch.zeros_like(v), torch.zeros_like(v))
    return params, variables, int(params["means3D"].shape[0])
