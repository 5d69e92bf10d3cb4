"""ORACLE O1: fp64, dense (every pixel x every Gaussian) PyTorch *autograd* restatement of the
differentiable 3D-Gaussian rasterizer with depth.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED: the reference's
implementation of this path is an absent third-party CUDA extension
(/root/reference/README.md:28-32); this file follows SURVEY.md Appendix A (A.1-A.5, conventions
A-1..A-9) and is anchored on the call sites /root/reference/src/tracking/train_utils.py:174-192.

No hand-derived backward exists here: gradients come from autograd through the forward maths,
with the four non-calculus conventions expressed through ``detach``:
  A-1  min(0.99, alpha) is straight-through,
  A-2  skipped pairs get zero gradient (boolean masks are constants),
  A-3  frustum clamp zeroes the gradient of the clamped tx/ty, tz sees the clamped values,
  A-4  depth output carries no gradient,
plus the 1e-7 added to det^2 in the conic backward (custom Function, A.5 [M]).
Sizes: meant for P <= ~1000 and images <= ~100x100 (memory is O(H*W*P) doubles).
"""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]
TILE = 16


class _ConicFromCov(torch.autograd.Function):
    """(a,b,c) -> (c,-b,a)/det, backward with 1/(det^2 + 1e-7) instead of 1/det^2 (A.5)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c, det)
        return c / det, -b / det, a / det

    @staticmethod
    def backward(ctx, gA, gB, gC):
        a, b, c, det = ctx.saved_tensors
        d2 = 1.0 / (det * det + 1e-7)
        da = d2 * (-c * c * gA + b * c * gB - b * b * gC)
        dc = d2 * (-b * b * gA + a * b * gB - a * a * gC)
        db = d2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
        return da, db, dc


def _sh_to_rgb(deg, shs, means3D, campos):
    d = means3D - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = SH_C0 * shs[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2 * zz - xx - yy) * shs[:, 6]
                 + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                     + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                     + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                     + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                     + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    r = r + 0.5
    return torch.clamp_min(r, 0.0)  # autograd of clamp_min zeroes the gradient where clamped


def dense_rasterize(H, W, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix, sh_degree, campos,
                    means3D, opacities, colors_precomp=None, scales=None, rotations=None, shs=None,
                    cov3D_precomp=None, order_dtype=torch.float32):
    """Returns (color[3,H,W], radii[P] int32, depth[1,H,W], means2D_pix[P,2] (graph leaf-like, see below)).

    All tensor inputs are promoted to float64.  ``means2D_pix`` is returned with ``retain_grad`` so
    callers can read d(loss)/d(pixel mean) and apply A-6's (0.5 W, 0.5 H) scaling themselves.
    Depth ordering uses the depth rounded to ``order_dtype`` (fp32, as the tiled pipelines sort on
    fp32 bit patterns), ties broken by ascending Gaussian index (A-9).
    """
    f64 = torch.float64
    V = viewmatrix.reshape(4, 4).to(f64)   # stored transposed: p_view = p_row @ V
    Pm = projmatrix.reshape(4, 4).to(f64)
    means3D = means3D.to(f64) if means3D.dtype != f64 else means3D
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=f64)
    p_hom4 = torch.cat([means3D, ones], 1)
    pv = (p_hom4 @ V)[:, :3]
    ph = p_hom4 @ Pm
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    visible = pv[:, 2] > 0.2

    if cov3D_precomp is not None:
        c6 = cov3D_precomp.to(f64)
        Sig = torch.stack([torch.stack([c6[:, 0], c6[:, 1], c6[:, 2]], 1),
                           torch.stack([c6[:, 1], c6[:, 3], c6[:, 4]], 1),
                           torch.stack([c6[:, 2], c6[:, 4], c6[:, 5]], 1)], 1)
    else:
        q = rotations.to(f64)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
            torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
            torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
        s = scales.to(f64) * scale_modifier
        M = R * s[:, None, :]
        Sig = M @ M.transpose(1, 2)

    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = pv[:, 2]
    tz_safe = torch.where(visible, tz, torch.ones_like(tz))
    txtz = pv[:, 0] / tz_safe
    tytz = pv[:, 1] / tz_safe
    xcl = (txtz < -limx) | (txtz > limx)
    ycl = (tytz < -limy) | (tytz > limy)
    # A-3: when the clamp is active the clamped value is a constant of the graph
    tx = torch.where(xcl, (txtz.clamp(-limx, limx) * tz_safe).detach(), pv[:, 0])
    ty = torch.where(ycl, (tytz.clamp(-limy, limy) * tz_safe).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe)], 1),
                     torch.stack([zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], 1)], 1)  # [P,2,3]
    Wrot = V[:3, :3].t()  # W_rot(r,c) = V[c, r]
    T = J @ Wrot[None]    # [P,2,3]
    cov2 = T @ Sig @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    okdet = det != 0
    a_s = torch.where(okdet, a, torch.ones_like(a))
    c_s = torch.where(okdet, c, torch.ones_like(c))
    b_s = torch.where(okdet, b, torch.zeros_like(b))
    cA, cB, cC = _ConicFromCov.apply(a_s, b_s, c_s)
    mid = 0.5 * (a + c)
    sq = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    lam = torch.maximum(mid + sq, mid - sq)
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    means2D_pix = torch.stack([px, py], 1)
    if means2D_pix.requires_grad:
        means2D_pix.retain_grad()
    px, py = means2D_pix[:, 0], means2D_pix[:, 1]  # the blend consumes the retained tensor
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    def _tile(v, g):
        return torch.clamp(torch.trunc(v.detach() / TILE), 0, g).to(torch.int64)
    minx, miny = _tile(px - radius, gx), _tile(py - radius, gy)
    maxx, maxy = _tile(px + radius + (TILE - 1), gx), _tile(py + radius + (TILE - 1), gy)
    alive = visible & okdet & ((maxx - minx) * (maxy - miny) > 0)
    radii = torch.where(alive, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None:
        rgb = colors_precomp.to(f64)
    else:
        rgb = _sh_to_rgb(sh_degree, shs.to(f64), means3D, campos.to(f64).reshape(3))

    # ---- ordering: fp32 depth bits, ties by index (A-9)
    dkey = pv[:, 2].detach().to(order_dtype).to(f64)
    order = torch.argsort(dkey, stable=True)
    # ---- dense blend
    ys, xs = torch.meshgrid(torch.arange(H, dtype=f64), torch.arange(W, dtype=f64), indexing="ij")
    pxf, pyf = xs.reshape(-1), ys.reshape(-1)                       # [N]
    ptx, pty = (pxf // TILE).to(torch.int64), (pyf // TILE).to(torch.int64)
    o = order
    gate = (alive[o][None] & (ptx[:, None] >= minx[o][None]) & (ptx[:, None] < maxx[o][None])
            & (pty[:, None] >= miny[o][None]) & (pty[:, None] < maxy[o][None]))  # [N,P]
    dx = px[o][None] - pxf[:, None]
    dy = py[o][None] - pyf[:, None]
    power = -0.5 * (cA[o][None] * dx * dx + cC[o][None] * dy * dy) - cB[o][None] * dx * dy
    op = opacities.to(f64).reshape(-1)[o][None]
    power_m = torch.where(gate & (power.detach() <= 0), power, torch.full_like(power, -1e3))
    alpha_raw = op * torch.exp(power_m)
    alpha = alpha_raw + (torch.clamp_max(alpha_raw, 0.99) - alpha_raw).detach()  # A-1
    contrib = gate & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha_e = torch.where(contrib, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha_e
    Tincl = torch.cumprod(one_m, dim=1)
    Tbefore = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], 1)
    stop = contrib & (Tincl.detach() < 1e-4)
    done = torch.cummax(stop.to(torch.int8), dim=1).values.bool()  # from the first stopping entry on
    use = contrib & ~done
    w = torch.where(use, alpha_e * Tbefore, torch.zeros_like(alpha_e))
    # transmittance after the last blended entry
    T_final = torch.where(use, one_m, torch.ones_like(one_m)).prod(dim=1)
    bgv = bg.to(f64).reshape(3)
    color = (w @ rgb[o]) + T_final[:, None] * bgv[None]
    with torch.no_grad():  # A-4
        depth = (w @ pv[:, 2][o][:, None])
    color = color.t().reshape(3, H, W)
    depth = depth.reshape(1, H, W)
    return color, radii, depth, means2D_pix


def finite_difference(fn, x, eps=1e-6):
    """Central finite differences of scalar fn wrt fp64 tensor x (test helper)."""
    g = torch.zeros_like(x)
    flat = x.view(-1)
    gf = g.view(-1)
    for i in range(flat.numel()):
        old = flat[i].item()
        flat[i] = old + eps
        fp = fn(x).item()
        flat[i] = old - eps
        fm = fn(x).item()
        flat[i] = old
        gf[i] = (fp - fm) / (2 * eps)
    return g


def quat_normalize(q):
    return q / q.norm(dim=1, keepdim=True)


__all__ = ["dense_rasterize", "finite_difference", "quat_normalize", "math"]
