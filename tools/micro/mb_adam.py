"""GPU box: cost of the optimiser step of the tracking loop (8 parameter groups, 100k Gaussians): default vs fused Adam."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import synth_scene_params
dev = torch.device("cuda:0")
lrs = {"means3D": 0.00016 * 4, "rgb_colors": 0.0, "seg_colors": 0.0, "unnorm_rotations": 0.001, "logit_opacities": 0.05,
       "log_scales": 0.001, "cam_m": 1e-4, "cam_c": 1e-4}
from gsdyn.optim import FusedAdam
for fused in (False, True, 'gsr'):
    params = synth_scene_params(100_000, device=dev)
    groups = [{"params": [v], "name": k, "lr": lrs[k]} for k, v in params.items()]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15) if fused == 'gsr' else torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=fused)
    for k, v in params.items():
        if v.requires_grad and k not in ("rgb_colors", "seg_colors"):
            v.grad = torch.randn_like(v)
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        opt.step()
    torch.cuda.synchronize()
    print({False: "default", True: "torch fused", "gsr": "gsr_adam_step"}[fused], "Adam step ms", (time.perf_counter() - t0) / 50 * 1e3)
