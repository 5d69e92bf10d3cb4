"""SynthScene-v1: the synthetic benchmark scene of SURVEY.md section 8d / BASELINE.md section 3.

Draw order and distributions are part of the definition (``numpy.random.default_rng(0)``), so every
implementation (HIP path, oracles, CPU baseline) renders exactly the same Gaussians.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .camera import look_at_w2c, setup_camera


def synth_scene_params(P: int, seed: int = 0, device=None, scale_lo=0.005, scale_hi=0.03):
    """Parameter dict with the reference's 8 keys (/root/reference/src/tracking/train_utils.py:119-128)."""
    rng = np.random.default_rng(seed)
    means3D = rng.uniform(-1, 1, (P, 3)).astype(np.float32)
    log_scales = rng.uniform(math.log(scale_lo), math.log(scale_hi), (P, 3)).astype(np.float32)
    unnorm_rotations = rng.normal(0, 1, (P, 4)).astype(np.float32)
    logit_opacities = rng.uniform(-2, 4, (P, 1)).astype(np.float32)
    rgb_colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    seg = (rng.uniform(0, 1, P) < 0.7).astype(np.float32)
    seg_colors = np.stack((seg, np.zeros_like(seg), 1 - seg), -1)
    max_cams = 50
    params = dict(means3D=means3D, rgb_colors=rgb_colors, seg_colors=seg_colors, unnorm_rotations=unnorm_rotations,
                  logit_opacities=logit_opacities, log_scales=log_scales,
                  cam_m=np.zeros((max_cams, 3), np.float32), cam_c=np.zeros((max_cams, 3), np.float32))
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else torch.device(device)
    out = {k: torch.nn.Parameter(torch.tensor(v).to(dev).float().contiguous().requires_grad_(True))
           for k, v in params.items()}
    out["rgb_colors"].requires_grad = False
    return out


def synth_ring_cameras(V: int, W: int, H: int, device=None, radius=4.0, height=0.8, near=0.01, far=100.0,
                       first: int = 0, count=None):
    """Cameras v = first .. first+count-1 of a V-camera ring: centre (r cos t, h, r sin t), t = 2 pi v / V + 0.3."""
    cams = []
    count = V if count is None else count
    for v in range(first, first + count):
        th = 2 * math.pi * v / V + 0.3
        w2c = look_at_w2c((radius * math.cos(th), height, radius * math.sin(th)))
        k = [[float(W), 0.0, W / 2.0], [0.0, float(W), H / 2.0], [0.0, 0.0, 1.0]]
        cams.append(setup_camera(W, H, k, w2c, near=near, far=far, device=device))
    return cams


def synth_targets(W: int, H: int, seed: int = 1, device=None):
    """(im_gt ~ U(0,1)[3,H,W], seg_gt = centred disc of radius H/3 as (m, 0, 1-m))."""
    rng = np.random.default_rng(seed)
    im = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    m = (((xs - W / 2.0) ** 2 + (ys - H / 2.0) ** 2) <= (H / 3.0) ** 2).astype(np.float32)
    seg = np.stack((m, np.zeros_like(m), 1 - m))
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu") if device is None else torch.device(device)
    return torch.tensor(im).to(dev), torch.tensor(seg).to(dev)
