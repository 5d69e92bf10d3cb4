"""GPU box: per-kernel time of the multi-view step's kernels without the optimiser (fixed scene), full vs frozen-colour backward.
V=<views> FROZEN=0|1 GSR_HIP_LIB=<library>"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
V = int(os.environ.get("V", "8"))
frozen = os.environ.get("FROZEN", "0") == "1"
params = synth_scene_params(100_000, device=dev)
cams = synth_ring_cameras(max(V, 4), 800, 800, device=dev)[:V]
dL = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (V, 3, 800, 800)).astype(np.float32), device=dev)
for _ in range(5):
    render_step_views(params, cams, dL, want_colour_grad=not frozen)
torch.cuda.synchronize()
_hip.profile_begin()
N = 10
for _ in range(N):
    render_step_views(params, cams, dL, want_colour_grad=not frozen)
torch.cuda.synchronize()
rows = _hip.profile_end()
print(os.path.basename(os.environ.get("GSR_HIP_LIB", "libgsr_hip.so")), "frozen" if frozen else "full",
      " ".join("%s=%.1f" % (k, 1e3 * t / n) for k, (t, n) in sorted(rows.items())))
