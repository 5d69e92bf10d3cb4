#!/usr/bin/env python3
"""BASELINE.json configs[4]-shaped forward-only timing: 500k Gaussians, one 1920x1080 view (per GPU), per-kernel us."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
P, W, H = 500_000, 1920, 1080
params = synth_scene_params(P, device=dev)
cam = synth_ring_cameras(8, W, H, device=dev)[0]
with torch.no_grad():
    rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
st = {}
orig = _hip.rasterize_forward
def spy(*a, **k):
    out = orig(*a, **k); st["s"] = out[3]; return out
_hip.rasterize_forward = spy
def run():
    with torch.no_grad():
        return GaussianRasterizer(raster_settings=cam)(**rv)
for _ in range(3):
    run()
torch.cuda.synchronize()
v = _hip.debug_views(st["s"])
r = v["ranges"].cpu().numpy().astype(np.int64); n = r[:, 1] - r[:, 0]
print("D", st["s"].num_rendered, "tiles", len(n), "busy", int((n > 0).sum()), "max list", int(n.max()),
      "lists >512:", int((n > 512).sum()), ">2048:", int((n > 2048).sum()), ">4096:", int((n > 4096).sum()))
_hip.rasterize_forward = orig
N = 20
t0 = time.perf_counter()
for _ in range(N):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"forward {dt*1e3:.3f} ms/view  {W*H/dt/1e6:.0f} Mpix/s")
_hip.profile_begin()
for _ in range(5):
    run()
torch.cuda.synchronize()
prof = _hip.profile_end()
print({k: round(1e3 * ms / max(nn, 1), 1) for k, (ms, nn) in prof.items()})
