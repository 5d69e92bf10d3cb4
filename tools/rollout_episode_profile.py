"""GPU box: where collect_scene_data (gsdyn/predict.py: rollout -> smoothing -> packing) spends its time at BASELINE configs[4] size."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import synth_scene_params
from gsdyn.dynamics import DynamicsPredictor
from gsdyn.predict import collect_scene_data
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
frames = 8
params = {k: v.detach() for k, v in synth_scene_params(P, device=dev).items()}
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
           rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = DynamicsPredictor(cfg, device=dev).eval()
eef = torch.tensor([[0.0, 0.2, 0.0]], device=dev) + torch.tensor([[0.02, 0.0, 0.01]], device=dev) * torch.arange(frames, device=dev, dtype=torch.float32)[:, None]
roll = dict(max_nobj=100, fps_radius=0.3, adj_thresh=0.6, topk=5, connect_all=False, dist_thresh=0.005, n_fps_all=1000, remove_outliers=False)
collect_scene_data(model, params, eef[:2], **roll)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, _, tm = collect_scene_data(model, params, eef, **roll)
torch.cuda.synchronize(); print("total ms", (time.perf_counter() - t0) * 1e3, tm)
pr = cProfile.Profile(); pr.enable()
collect_scene_data(model, params, eef, **roll)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
