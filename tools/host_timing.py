#!/usr/bin/env python3
"""Host-side cost of the batched forward / backward calls vs the GPU time of the step (is the step launch-bound?)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip, rasterize_gaussians_views
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params

dev = torch.device("cuda:0")
P, W, H, V = 100_000, 800, 800, 4
params = synth_scene_params(P, device=dev)
cams = synth_ring_cameras(V, W, H, device=dev)
with torch.no_grad():
    rv = {k: v.detach().clone() for k, v in params2rendervar(params).items()}
for k in ("means3D", "rotations", "opacities", "scales"):
    rv[k].requires_grad_(True)
dL = torch.rand((V, 3, H, W), device=dev) * 2 - 1
m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
tf = tb = 0.0
of, ob = _hip.rasterize_forward_batch, _hip.rasterize_backward_batch
acc = {"f": 0.0, "b": 0.0, "n": 0}
def f(*a, **k):
    t = time.perf_counter(); r = of(*a, **k); acc["f"] += time.perf_counter() - t; return r
def b(*a, **k):
    t = time.perf_counter(); r = ob(*a, **k); acc["b"] += time.perf_counter() - t; return r
_hip.rasterize_forward_batch, _hip.rasterize_backward_batch = f, b
def step():
    for t in rv.values():
        t.grad = None
    m2.grad = None
    im, radii, depth = rasterize_gaussians_views(cams, rv["means3D"], m2, rv["opacities"], colors_precomp=rv["colors_precomp"],
                                                 scales=rv["scales"], rotations=rv["rotations"])
    im.backward(gradient=dL)
for _ in range(5):
    step()
torch.cuda.synchronize()
acc.update(f=0.0, b=0.0)
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per step: wall {1e6*t_all/N:.0f} us, host issue loop {1e6*t_issue/N:.0f} us, inside forward_batch {1e6*acc['f']/N:.0f} us "
      f"(includes the stage-1 sync), inside backward_batch {1e6*acc['b']/N:.0f} us")
