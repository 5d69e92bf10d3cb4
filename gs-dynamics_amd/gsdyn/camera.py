"""Camera -> ``GaussianRasterizationSettings`` adapter.

Mirrors ``setup_camera`` of the reference (/root/reference/src/tracking/helpers.py:10-33; duplicated at
/root/reference/src/render/renderer.py:25-50 and /root/reference/src/real_world/gs/helpers.py:10-33, the
latter two with a ``bg`` argument): same arithmetic in the same order, but device-agnostic (the reference
hard-codes ``.cuda()``), so the result can be pinned against vectors captured from the reference on CPU.
"""
from __future__ import annotations

import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings as Camera


def _default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def setup_camera(w, h, k, w2c, near=0.01, far=100, bg=(0, 0, 0), device=None) -> Camera:
    device = _default_device() if device is None else torch.device(device)
    fx, fy, cx, cy = k[0][0], k[1][1], k[0][2], k[1][2]
    w2c = torch.tensor(np.asarray(w2c)).to(device).float()
    cam_center = torch.inverse(w2c)[:3, 3]
    w2c = w2c.unsqueeze(0).transpose(1, 2)
    opengl_proj = torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                                [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]]).to(device).float().unsqueeze(0).transpose(1, 2)
    full_proj = w2c.bmm(opengl_proj)
    return Camera(
        image_height=h,
        image_width=w,
        tanfovx=w / (2 * fx),
        tanfovy=h / (2 * fy),
        bg=torch.tensor(list(bg), dtype=torch.float32, device=device),
        scale_modifier=1.0,
        viewmatrix=w2c.contiguous(),   # same values as the reference's transposed view; no per-call copy
        projmatrix=full_proj,
        sh_degree=0,
        campos=cam_center.contiguous(),
        prefiltered=False,
    )


def Rt_to_w2c(R, t):
    """4x4 world-to-camera from a camera POSE (rotation + translation of the camera in the world): the pose matrix [R t; 0 1] is
    assembled and INVERTED (/root/reference/src/real_world/gs/trainer.py:15-18)."""
    c2w = np.concatenate([np.concatenate([np.asarray(R, np.float64), np.asarray(t, np.float64).reshape(3, 1)], axis=1),
                          np.array([[0.0, 0.0, 0.0, 1.0]])], axis=0)
    return np.linalg.inv(c2w)


def look_at_w2c(center, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)) -> np.ndarray:
    """World-to-camera with +z forward, +x right, +y down-free 'up' convention of SynthScene-v1
    ('look-at origin, up +y, +z forward')."""
    c = np.asarray(center, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    r = np.cross(np.asarray(up, dtype=np.float64), f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    R = np.stack([r, u, f])
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ c
    return w2c
