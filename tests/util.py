"""Shared test helpers: seeded small scenes, camera construction for the oracles, comparison metrics."""
import math

import numpy as np

from oracle import OracleCamera


def look_at(center, target=(0, 0, 0), up=(0, 1.0, 0)):
    c = np.asarray(center, np.float64)
    f = np.asarray(target, np.float64) - c
    f /= np.linalg.norm(f)
    r = np.cross(np.asarray(up, np.float64), f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    R = np.stack([r, u, f])
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ c
    return w2c


def oracle_camera(W, H, w2c, fx=None, fy=None, cx=None, cy=None, near=0.01, far=100.0, bg=(0, 0, 0), sh_degree=0):
    """Same arithmetic as the reference's setup_camera (/root/reference/src/tracking/helpers.py:10-33), numpy fp32."""
    fx = float(W) if fx is None else fx
    fy = float(W) if fy is None else fy
    cx = W / 2.0 if cx is None else cx
    cy = H / 2.0 if cy is None else cy
    w2c32 = np.asarray(w2c, np.float32)
    proj = np.array([[2 * fx / W, 0, -(W - 2 * cx) / W, 0], [0, 2 * fy / H, -(H - 2 * cy) / H, 0],
                     [0, 0, far / (far - near), -(far * near) / (far - near)], [0, 0, 1, 0]], np.float32)
    vm = np.ascontiguousarray(w2c32.T)
    full = (vm @ proj.T).astype(np.float32)
    campos = np.linalg.inv(w2c32.astype(np.float64))[:3, 3].astype(np.float32)
    return OracleCamera(H, W, W / (2 * fx), H / (2 * fy), np.asarray(bg, np.float32), 1.0, vm, full, sh_degree, campos)


def ring_camera(W, H, v=0, V=4, radius=4.0, height=0.8, **kw):
    th = 2 * math.pi * v / V + 0.3
    return oracle_camera(W, H, look_at((radius * math.cos(th), height, radius * math.sin(th))), **kw)


def random_gaussians(P, seed=0, scale_lo=0.02, scale_hi=0.3, spread=1.0, sh_M=0):
    rng = np.random.default_rng(seed)
    means = rng.uniform(-spread, spread, (P, 3)).astype(np.float32)
    scales = np.exp(rng.uniform(np.log(scale_lo), np.log(scale_hi), (P, 3))).astype(np.float32)
    rot = rng.normal(size=(P, 4)).astype(np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    op = (1 / (1 + np.exp(-rng.uniform(-2, 4, (P, 1))))).astype(np.float32)
    col = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    out = dict(means3D=means, scales=scales, rotations=rot, opacities=op, colors_precomp=col)
    if sh_M:
        out["shs"] = (rng.normal(size=(P, sh_M, 3)) * 0.4).astype(np.float32)
    return out


def rel_err(a, b):
    """Norm-wise relative error: max|a-b| / max|b| (the metric behind every '<= 1e-4 rel' claim for
    tensors whose entries are sums with cancellation -- gradients)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def mixed_err(a, b, atol_frac=1e-4):
    """Element-wise |a-b| / (|b| + atol_frac * max|b|): max over elements."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) / (np.abs(b) + atol_frac * (np.abs(b).max() + 1e-30))).max())


def row_err(a, b, floor_frac=1e-3):
    """Row-wise (per-Gaussian) relative error: for every row i,  max_j |a_ij - b_ij| / (max_j |b_ij| + floor_frac * max|b|).
    Returns (worst value, its row).  Tighter than ``rel_err``: a Gaussian whose gradient is 1e-3 of the tensor maximum must
    still be right to the stated relative tolerance, not to 1e-4 of the LARGEST gradient in the tensor."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0, -1
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    num = np.abs(a2 - b2).max(1)
    den = np.abs(b2).max(1) + floor_frac * (np.abs(b2).max() + 1e-300)
    r = num / den
    i = int(np.argmax(r))
    return float(r[i]), i


def row_err_quantiles(a, b, floor_frac=1e-3, qs=(0.5, 0.99, 0.9999)):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    r = np.abs(a2 - b2).max(1) / (np.abs(b2).max(1) + floor_frac * (np.abs(b2).max() + 1e-300))
    return [float(np.quantile(r, q)) for q in qs]
