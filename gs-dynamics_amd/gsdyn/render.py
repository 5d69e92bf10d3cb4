"""Forward-only caller of the path: mirror of the reference's ``Renderer``
(/root/reference/src/render/renderer.py:6-50) -- builds the settings record per call and renders under
``no_grad``; ``predict.py`` drives it per (camera, frame) and renders a second time with colours = 1 as an
alpha mask (/root/reference/src/predict.py:115-123)."""
from __future__ import annotations

import numpy as np
import torch

from diff_gaussian_rasterization import GaussianRasterizer, rasterize_gaussians_views

from .camera import setup_camera


class Renderer:
    def __init__(self, device, w: int = 1280, h: int = 720, near: float = 0.01, far: float = 100.0):
        self.near, self.far = near, far
        self.w, self.h = w, h
        self.device = device
        self._cameras = {}

    def _camera(self, w2c, k, bg):
        """The settings record of a camera, built once per distinct (pose, intrinsics, background): ``setup_camera`` costs
        three host->device copies and a device-side 4x4 inverse per call, which the reference pays on every render of every
        frame although predict.py's cameras are fixed (/root/reference/src/render/renderer.py:25-50)."""
        key = (np.asarray(w2c, dtype=np.float64).tobytes(), np.asarray(k, dtype=np.float64).tobytes(), tuple(float(b) for b in bg))
        cam = self._cameras.get(key)
        if cam is None:
            if len(self._cameras) >= 256:
                self._cameras.clear()
            cam = self._cameras[key] = setup_camera(self.w, self.h, k, w2c, near=self.near, far=self.far, bg=bg, device=self.device)
        return cam

    @torch.no_grad()
    def render(self, w2c, k, timestep_data, bg=(0.7, 0.7, 0.7)):
        timestep_data = {key: v.to(self.device) for key, v in timestep_data.items()}
        cam = self._camera(w2c, k, bg)
        im, _, depth = GaussianRasterizer(raster_settings=cam)(**timestep_data)
        return im, depth

    @torch.no_grad()
    def render_with_mask(self, w2c, k, timestep_data, bg=(0.0, 0.0, 0.0)):
        """Colour render + the all-ones 'mask' render of predict.py, as ONE multi-view call: the two renders share the
        camera, so the second one reuses the first one's tile lists (per-view colours, ``geometry_of``)."""
        ims, depths, masks = self.render_cameras_with_mask([(w2c, k)], timestep_data, bg=bg)
        return ims[0], depths[0], masks[0]

    @torch.no_grad()
    def render_cameras_with_mask(self, cameras, timestep_data, bg=(0.0, 0.0, 0.0), mask_from_alpha: bool = True):
        """All cameras of a frame, colour + mask each, in one rasterizer call (predict.py renders 4 cameras x 2 per
        frame, /root/reference/src/predict.py:100-123).  ``cameras``: list of (w2c, k).  Returns image, depth and mask lists.

        The mask render differs from the colour render in ``colors_precomp = 1`` only (predict.py:119-121), so each of its
        channels is sum_i alpha_i T_i + T_final bg = (1 - T_final) + T_final bg: the colour render's final transmittance says
        it all (sum_i alpha_i T_i telescopes to 1 - T_final over the blended entries, early stop included).  With
        ``mask_from_alpha`` (default) the mask comes from the forward's per-pixel T_final -- ONE plain blend pass per camera
        instead of a six-channel one -- and equals the second render up to fp32 rounding of the two summation orders (~1e-6;
        tests/test_dynamics_gpu.py bounds it against the second render).  ``mask_from_alpha = False`` blends the ones as well
        (fused pair: the two renders share tile lists and records)."""
        d = {key: v.to(self.device) for key, v in timestep_data.items()}
        cams = [self._camera(w2c, k, bg) for w2c, k in cameras]
        n = len(cams)
        col = d["colors_precomp"].float()
        P = d["means3D"].shape[0]
        if mask_from_alpha and P > 0:
            from diff_gaussian_rasterization import _hip
            out, _, depth, states = _hip.rasterize_forward_batch(
                cams, d["means3D"].float().contiguous(), d["opacities"].float().contiguous(), col.contiguous(), None,
                d["scales"].float().contiguous(), d["rotations"].float().contiguous(), None, prepare_backward=False, forward_only=True)
            black = all(float(b) == 0.0 for b in bg)                         # (decided on the host: no device round trip)
            masks = []
            for i in range(n):
                T = _hip.final_transmittance(states[i])                     # [H, W], a view of the state's image buffer
                if black:
                    masks.append((1.0 - T).unsqueeze(0).expand(3, -1, -1))  # three equal channels: no copy
                else:
                    masks.append(1.0 - T.unsqueeze(0) * (1.0 - cams[i].bg.view(3, 1, 1)))
            return [out[i] for i in range(n)], [depth[i] for i in range(n)], masks
        views = [c for c in cams for _ in (0, 1)]
        colours = torch.stack([col, torch.ones_like(col)]).repeat(n, 1, 1)
        out, _, depth = rasterize_gaussians_views(views, d["means3D"], torch.zeros((len(views), P, 3), device=self.device),
                                                  d["opacities"], colors_precomp=colours, scales=d["scales"], rotations=d["rotations"])
        return [out[2 * i] for i in range(n)], [depth[2 * i] for i in range(n)], [out[2 * i + 1] for i in range(n)]
