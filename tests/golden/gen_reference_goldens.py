"""Generate ``reference_host.npz``: golden vectors captured by IMPORTING the Python reference in the
build container (it cannot travel to the GPU box).  Run once, here:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_reference_goldens.py

What is pinned (caller-side pieces of the hot path):
  * ``setup_camera``      /root/reference/src/tracking/helpers.py:10-33   (4 ring cameras + 4 demo cameras)
  * ``params2rendervar``  /root/reference/src/tracking/helpers.py:36-45
  * ``l1_loss_v1/v2``, ``weighted_l2_loss_v1/v2``, ``quat_mult``  .../helpers.py:71-94
  * ``build_rotation``, ``calc_ssim``, ``calc_psnr`` (values and input gradients)
                          /root/reference/src/tracking/external.py:25-42, 54-135
  * the t>0 rigidity block of ``get_loss``  /root/reference/src/tracking/train_utils.py:198-232 (re-evaluated
    with the reference's own helper functions on seeded tensors)
The rasterizer itself is NOT importable (absent third-party CUDA extension): its parity is unpinned.
Only inputs and outputs are stored -- data, no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_host.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("open3d", "cv2", "dgl", "ipdb"):
        sys.modules.setdefault(name, types.ModuleType(name))
    dgr = types.ModuleType("diff_gaussian_rasterization")

    class _Settings:  # records the kwargs the reference passes
        def __init__(self, **kw):
            self.__dict__.update(kw)
    dgr.GaussianRasterizationSettings = _Settings
    dgr.GaussianRasterizer = object
    sys.modules["diff_gaussian_rasterization"] = dgr
    torch.Tensor.cuda = lambda self, *a, **k: self          # CPU container: .cuda() -> identity
    _orig_tensor, _orig_zeros, _orig_zeros_like = torch.tensor, torch.zeros, torch.zeros_like

    def _strip(fn):
        def inner(*a, **k):
            if "device" in k and str(k["device"]).startswith("cuda"):
                k["device"] = "cpu"
            return fn(*a, **k)
        return inner
    torch.tensor, torch.zeros, torch.zeros_like = _strip(_orig_tensor), _strip(_orig_zeros), _strip(_orig_zeros_like)
    sys.path.insert(0, os.path.join(REF, "tracking"))
    import helpers as ref_helpers  # noqa
    import external as ref_external  # noqa
    return ref_helpers, ref_external


def ring_w2c(v, V, radius=4.0, height=0.8):
    th = 2 * np.pi * v / V + 0.3
    c = np.array([radius * np.cos(th), height, radius * np.sin(th)])
    f = -c / np.linalg.norm(c)
    r = np.cross(np.array([0.0, 1.0, 0.0]), f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    R = np.stack([r, u, f])
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ c
    return w2c


def main():
    H_, E_ = import_reference()
    out = {}
    # ---- setup_camera: ring cameras of SynthScene-v1 and the demo cameras
    W, H = 800, 800
    cams_in, cams_out = [], []
    for v in range(4):
        w2c = ring_w2c(v, 4)
        k = np.array([[W, 0, W / 2], [0, W, H / 2], [0, 0, 1.0]])
        cam = H_.setup_camera(W, H, k, w2c, near=0.01, far=100)
        cams_in.append(np.concatenate([w2c.reshape(-1), k.reshape(-1), [W, H, 0.01, 100]]))
        cams_out.append(np.concatenate([cam.viewmatrix.reshape(-1).numpy(), cam.projmatrix.reshape(-1).numpy(),
                                        cam.campos.reshape(-1).numpy(), [cam.tanfovx, cam.tanfovy]]))
    R_list = np.load("/root/reference/assets/demo/R_list.npy")
    t_list = np.load("/root/reference/assets/demo/t_list.npy")
    intr = np.load("/root/reference/assets/demo/intr_list.npy")
    for i in range(R_list.shape[0]):
        w2c = np.concatenate([np.concatenate([R_list[i], t_list[i].reshape(3, 1)], 1), np.array([[0, 0, 0, 1.0]])], 0)
        k = intr[i]
        cam = H_.setup_camera(640, 480, k, w2c, near=1.0, far=100)
        cams_in.append(np.concatenate([w2c.reshape(-1), np.asarray(k, dtype=np.float64).reshape(-1), [640, 480, 1.0, 100]]))
        cams_out.append(np.concatenate([cam.viewmatrix.reshape(-1).numpy(), cam.projmatrix.reshape(-1).numpy(),
                                        cam.campos.reshape(-1).numpy(), [cam.tanfovx, cam.tanfovy]]))
    out["cam_in"] = np.stack(cams_in)
    out["cam_out"] = np.stack(cams_out).astype(np.float64)

    # ---- params2rendervar
    g = torch.Generator().manual_seed(7)
    P = 33
    params = {"means3D": torch.randn(P, 3, generator=g), "rgb_colors": torch.rand(P, 3, generator=g),
              "unnorm_rotations": torch.randn(P, 4, generator=g), "logit_opacities": torch.randn(P, 1, generator=g),
              "log_scales": torch.randn(P, 3, generator=g) * 0.3 - 3}
    rv = H_.params2rendervar(params)
    for k_, v_ in params.items():
        out["p2r_in_" + k_] = v_.numpy()
    for k_ in ("rotations", "opacities", "scales", "means2D", "colors_precomp"):
        out["p2r_out_" + k_] = rv[k_].detach().numpy()

    # ---- scalar losses + quat_mult + build_rotation
    a = torch.randn(17, 20, 3, generator=g); b = torch.randn(17, 20, 3, generator=g); w = torch.rand(17, 20, generator=g)
    out["loss_a"], out["loss_b"], out["loss_w"] = a.numpy(), b.numpy(), w.numpy()
    out["l1_v1"] = H_.l1_loss_v1(a, b).numpy()
    out["l1_v2"] = H_.l1_loss_v2(a, b).numpy()
    out["wl2_v1"] = H_.weighted_l2_loss_v1(a[..., 0], b[..., 0], w).numpy()
    out["wl2_v2"] = H_.weighted_l2_loss_v2(a, b, w).numpy()
    q1 = torch.randn(19, 4, generator=g); q2 = torch.randn(19, 4, generator=g)
    out["q1"], out["q2"] = q1.numpy(), q2.numpy()
    out["quat_mult"] = H_.quat_mult(q1, q2).numpy()
    import inspect
    src = inspect.getsource(E_.build_rotation)
    ns = {"torch": torch}
    exec(src.replace("device='cuda'", "device='cpu'"), ns)   # the only change: where the zeros live
    out["build_rotation"] = ns["build_rotation"](q1).numpy()

    # ---- SSIM / PSNR values and input gradients
    im1 = torch.rand(3, 64, 48, generator=g, requires_grad=True)
    im2 = torch.rand(3, 64, 48, generator=g)
    s = E_.calc_ssim(im1, im2)
    s.backward()
    out["ssim_im1"], out["ssim_im2"] = im1.detach().numpy(), im2.numpy()
    out["ssim"] = s.detach().numpy()
    out["ssim_grad"] = im1.grad.numpy()
    out["psnr"] = E_.calc_psnr(im1.detach(), im2).numpy()
    comb = 0.8 * H_.l1_loss_v1(im1.detach(), im2) + 0.2 * (1.0 - E_.calc_ssim(im1.detach(), im2))
    out["im_term"] = comb.numpy()

    # ---- rigidity block of get_loss (train_utils.py:198-232), evaluated with the reference's helpers
    n, kk = 64, 20
    fg_pts = torch.randn(n, 3, generator=g).requires_grad_(True)
    fg_rot_un = torch.randn(n, 4, generator=g).requires_grad_(True)
    fg_rot = torch.nn.functional.normalize(fg_rot_un)
    prev_inv = torch.nn.functional.normalize(torch.randn(n, 4, generator=g))
    nbr = torch.randint(0, n, (n, kk), generator=g)
    prev_offset = torch.randn(n, kk, 3, generator=g) * 0.1
    nw = torch.rand(n, kk, generator=g)
    nd = torch.rand(n, kk, generator=g)
    rel_rot = H_.quat_mult(fg_rot, prev_inv)
    rot = ns["build_rotation"](rel_rot)
    neighbor_pts = fg_pts[nbr]
    curr_offset = neighbor_pts - fg_pts[:, None]
    coipc = (rot.transpose(2, 1)[:, None] @ curr_offset[:, :, :, None]).squeeze(-1)
    l_rigid = H_.weighted_l2_loss_v2(coipc, prev_offset, nw)
    l_rot = H_.weighted_l2_loss_v2(rel_rot[nbr], rel_rot[:, None], nw)
    mag = torch.sqrt((curr_offset ** 2).sum(-1) + 1e-20)
    l_iso = H_.weighted_l2_loss_v1(mag, nd, nw)
    l_floor = torch.clamp(fg_pts[:, 1], min=0).mean()
    total = 200.0 * l_rigid + 4.0 * l_rot + 1000.0 * l_iso + 2.0 * l_floor
    total.backward()
    for k_, v_ in dict(rig_fg_pts=fg_pts, rig_fg_rot_un=fg_rot_un, rig_prev_inv=prev_inv, rig_prev_offset=prev_offset,
                       rig_nw=nw, rig_nd=nd).items():
        out[k_] = v_.detach().numpy()
    out["rig_nbr"] = nbr.numpy()
    out["rig_losses"] = np.array([l_rigid.item(), l_rot.item(), l_iso.item(), l_floor.item(), total.item()])
    out["rig_grad_pts"] = fg_pts.grad.numpy()
    out["rig_grad_rot"] = fg_rot_un.grad.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
