"""GPU box: where a rollout step (row N4, bench.py extras `rollout_step_cfg1`) spends its time: wall clock, cProfile, per-section syncs."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import params2rendervar, synth_scene_params
from gsdyn.dynamics import DynamicsPredictor, farthest_point_sampler, rollout_step
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3,
           rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = DynamicsPredictor(cfg, device=dev).eval()
with torch.no_grad():
    rv = {k: v.detach() for k, v in params2rendervar(params).items()}
pick = farthest_point_sampler(rv["means3D"][None], 100, start_idx=0)[0]
bones = rv["means3D"][pick]
hist, eef = bones[None].repeat(3, 1, 1), torch.zeros((3, 1, 3), device=dev)
step = lambda: rollout_step(model, hist, eef, eef[-1] + 0.02, rv["means3D"], rv["rotations"], 0.5, 5)   # noqa: E731
for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)

# ---- section timing (synchronised)
from gsdyn.dynamics import construct_edges, interpolate_motions, relations_to_matrix, fit_bone_rotations
def sec(name, f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = f()
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e3:8.3f} ms"); return out
nobj = 100
states = torch.zeros((1, 3, nobj + 1, 3), device=dev); states[0, :, :nobj] = hist
mask = torch.ones(nobj + 1, dtype=torch.bool, device=dev); tool = torch.zeros(nobj + 1, dtype=torch.bool, device=dev); tool[nobj] = True
recv, send = sec("construct_edges", lambda: construct_edges(states[0, -1], 0.5, mask, tool, topk=5))
attrs = torch.zeros((1, nobj + 1, 2), device=dev); action = torch.zeros((1, nobj + 1, 3), device=dev)
with torch.no_grad():
    pred, _ = sec("model", lambda: model(state=states, attrs=attrs, p_instance=torch.ones((1, nobj, 1), device=dev), action=action, receivers=recv, senders=send))
rel = sec("relations_to_matrix", lambda: relations_to_matrix(recv, send, nobj + 1)[:nobj, :nobj])
R = sec("fit_bone_rotations", lambda: fit_bone_rotations(bones, pred[0] - bones, rel))
sec("interpolate_motions", lambda: interpolate_motions(bones, pred[0] - bones, rel, rv["means3D"], quat=rv["rotations"]))
print("n_rel", recv.shape[0])
