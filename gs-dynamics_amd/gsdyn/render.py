"""Forward-only caller of the path: mirror of the reference's ``Renderer``
(/root/reference/src/render/renderer.py:6-50) -- builds the settings record per call and renders under
``no_grad``; ``predict.py`` drives it per (camera, frame) and renders a second time with colours = 1 as an
alpha mask (/root/reference/src/predict.py:115-123)."""
from __future__ import annotations

import torch

from diff_gaussian_rasterization import GaussianRasterizer

from .camera import setup_camera


class Renderer:
    def __init__(self, device, w: int = 1280, h: int = 720, near: float = 0.01, far: float = 100.0):
        self.near, self.far = near, far
        self.w, self.h = w, h
        self.device = device

    @torch.no_grad()
    def render(self, w2c, k, timestep_data, bg=(0.7, 0.7, 0.7)):
        timestep_data = {key: v.to(self.device) for key, v in timestep_data.items()}
        cam = setup_camera(self.w, self.h, k, w2c, near=self.near, far=self.far, bg=bg, device=self.device)
        im, _, depth = GaussianRasterizer(raster_settings=cam)(**timestep_data)
        return im, depth

    @torch.no_grad()
    def render_with_mask(self, w2c, k, timestep_data, bg=(0.0, 0.0, 0.0)):
        """Colour render + the all-ones 'mask' render of predict.py."""
        im, depth = self.render(w2c, k, timestep_data, bg=bg)
        ones = dict(timestep_data)
        ones["colors_precomp"] = torch.ones_like(timestep_data["colors_precomp"])
        mask, _ = self.render(w2c, k, ones, bg=(0.0, 0.0, 0.0))
        return im, depth, mask
