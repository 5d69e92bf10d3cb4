"""GPU box: replay the graphed rollout step (gsdyn.dynamics._GraphedStep) of BASELINE configs[4] (500 k Gaussians, 100 bones, GNN width 512)
200 times -- the workload tools/r04_rollout_trace.sh traces per kernel."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from gsdyn import dynamics as D
from gsdyn import synth_scene_params
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
params = {k: v.detach() for k, v in synth_scene_params(P, device=dev).items()}
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
           rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = D.DynamicsPredictor(cfg, device=dev).eval()
xyz = params["means3D"]
quat = torch.nn.functional.normalize(params["unnorm_rotations"])
with torch.no_grad():
    track = D.farthest_point_sampler(xyz[None], 1000)[0]
    pos_track = xyz[track]
    hist = pos_track[None].repeat(3, 1, 1)
    eef_h = torch.zeros((3, 1, 3), device=dev)
    gs = D._graphed_step_for(model, P, 1000, 3, 100, 0.3, 0, 0.6, 5, dev)
    gs.load(track, pos_track, hist, eef_h, xyz, quat)
    eef = torch.tensor([[0.02, 0.0, 0.01]], device=dev)
    for _ in range(5):
        gs.step(eef)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for i in range(n):
        gs.step(eef * (1 + 0.01 * i))
    torch.cuda.synchronize()
    print("graphed rollout step: %.1f us per step (n_valid %d, bad %d)" % ((time.perf_counter() - t0) / n * 1e6, int(gs.n_valid), int(gs.bad)))
