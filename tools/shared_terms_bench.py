"""GPU box: per-kernel times of the shared t > 0 terms (gsr_shared_terms_*) at the get_loss shape (100k Gaussians, 70k fg x 20 nbrs)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import synth_scene_params
from gsdyn.losses import shared_terms
from gsdyn.step import make_rigidity_variables
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
variables = make_rigidity_variables(params, num_knn=20)
m = (params["means3D"].detach() + 0.01 * torch.randn(100_000, 3, device=dev)).requires_grad_(True)
r = torch.nn.functional.normalize(params["unnorm_rotations"].detach() + 0.05 * torch.randn(100_000, 4, device=dev)).requires_grad_(True)
w5 = [200.0, 4.0, 1000.0, 2.0, 200.0]
for it in range(3):
    if it == 2:
        _hip.profile_begin()
    for _ in range(5):
        m.grad = None; r.grad = None
        total, _ = shared_terms(m, r, variables, w5)
        total.backward()
    torch.cuda.synchronize()
for k, (ms, cnt) in sorted(_hip.profile_end().items()):
    print(f"{k:24s} {ms / cnt * 1e3:8.1f} us x {cnt}")
