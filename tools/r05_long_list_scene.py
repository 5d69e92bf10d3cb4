"""GPU box: the scene of tests/test_lists_gpu.py::test_one_very_long_list_in_an_ordinary_scene, forward only, N times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from hipcheck import _settings
from util import random_gaussians, ring_camera
from diff_gaussian_rasterization import _hip
dev = torch.device("cuda:0")
cam = ring_camera(800, 800, bg=(0.1, 0.1, 0.1))
base = random_gaussians(20_000, seed=71, scale_lo=0.005, scale_hi=0.03)
clu = random_gaussians(3_000, seed=72, scale_lo=0.0015, scale_hi=0.003, spread=1.0)
th = 0.3
eye = np.array([4.0 * np.cos(th), 0.8, 4.0 * np.sin(th)], np.float32)
ray = -eye / np.linalg.norm(eye)
rng = np.random.default_rng(73)
t = rng.uniform(-0.8, 0.8, (3_000, 1)).astype(np.float32)
clu["means3D"] = (t * ray[None] + rng.normal(0, 0.004, (3_000, 3))).astype(np.float32) + np.array([0.02, 0.02, 0.0], np.float32)
clu["opacities"][:] = 0.02
both = {k: np.concatenate([base[k], clu[k]]) for k in base}
g = both if (len(sys.argv) < 2 or sys.argv[1] != "base") else base
rs = _settings(cam, dev)
tt = {k: torch.tensor(v, device=dev) for k, v in g.items()}
for _ in range(20):
    _hip.rasterize_forward(rs, tt["means3D"], tt["opacities"], tt["colors_precomp"], None, tt["scales"], tt["rotations"], None)
torch.cuda.synchronize()
