"""GPU box: time the fused image-term kernels (gsr_views_loss_*) on the get_loss shape: 4 cameras x (colour + seg) at 800 x 800."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import losses as L
dev = torch.device("cuda:0")
n, H, W = 8, 800, 800
g = torch.Generator(device=dev).manual_seed(0)
renders = torch.rand((n, 3, H, W), device=dev, generator=g).requires_grad_(True)
targets = [torch.rand((3, H, W), device=dev, generator=g) for _ in range(n)]
cam_m = torch.zeros((4, 3), device=dev, requires_grad=True)
cam_c = torch.zeros((4, 3), device=dev, requires_grad=True)
rows = [0, -1, 1, -1, 2, -1, 3, -1]
wts = [50.0, 200.0] * 4
for it in range(3):
    if it == 2:
        _hip.profile_begin()
    for _ in range(5):
        total, _ = L.views_image_loss(renders, targets, rows, wts, cam_m, cam_c)
        total.backward()
    torch.cuda.synchronize()
for k, (ms, cnt) in sorted(_hip.profile_end().items()):
    print(f"{k:20s} {ms / cnt * 1e3:8.1f} us x {cnt}")
