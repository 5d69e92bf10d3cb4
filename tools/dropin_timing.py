#!/usr/bin/env python3
"""The UNCHANGED-caller path: one `GaussianRasterizer(raster_settings=cam)(**rendervar)` forward + autograd backward
per render, as /root/reference/src/tracking/train_utils.py:174-192 calls it -- through the torch C++ layer (_C.so, one native call each
way) and through the ctypes binding (GSR_NO_TORCH_EXT=1), at 100k Gaussians / 800x800, in subprocesses.  Reports wall time per
render, the host time to ISSUE it, and the GPU-busy time from the library's own kernel events: is the drop-in surface GPU-bound?"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys, time
import torch
sys.path.insert(0, os.path.join(%(root)r, "gs-dynamics_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
cam = synth_ring_cameras(4, 800, 800, device=dev)[0]
dL = torch.rand((3, 800, 800), device=dev) * 2 - 1
def render():
    for p in params.values():
        p.grad = None
    rv = params2rendervar(params)                     # the reference's activations, torch ops
    rv["means2D"].retain_grad()
    im, radius, depth = GaussianRasterizer(raster_settings=cam)(**rv)
    im.backward(gradient=dL)
for _ in range(10):
    render()
torch.cuda.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N):
    render()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
_hip.profile_begin()
for _ in range(10):
    render()
torch.cuda.synchronize()
busy = sum(ms for ms, n in _hip.profile_end().values()) / 10 * 1e3
print(json.dumps({"torch_ext": dgr._C is not None, "wall_us": 1e6 * t_all / N, "host_issue_us": 1e6 * t_issue / N, "gsr_kernels_busy_us": busy}))
'''


def main():
    out = {}
    for name, env in (("torch C++ layer (_C.so)", {}), ("ctypes binding", {"GSR_NO_TORCH_EXT": "1"})):
        r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[name] = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
    for k, v in out.items():
        print(k, v)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
