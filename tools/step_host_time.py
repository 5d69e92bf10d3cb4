"""GPU box: how long the HOST needs to issue one 8-view step (gsdyn.step.render_step_views + FusedAdam) -- the loop is timed without waiting
for the device, so a value below the step time means the host runs ahead -- and where it spends that time (cProfile, own time)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import initialize_optimizer, synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = synth_scene_params(100_000, device=dev)
params["rgb_colors"].requires_grad_(True)
cams = synth_ring_cameras(V, 800, 800, device=dev)
dL = torch.tensor(np.random.default_rng(1).uniform(-1, 1, (V, 3, 800, 800)).astype(np.float32), device=dev)
opt = initialize_optimizer(params, 4.0)
KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")
def step():
    _, g = render_step_views(params, cams, dL)
    for k in KEYS:
        params[k].grad = g.get(k)
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N): step()
t_issue = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / N
print(f"V={V}: host issue time per step {t_issue*1e6:.0f} us, with the device drained {t_all*1e6:.0f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
