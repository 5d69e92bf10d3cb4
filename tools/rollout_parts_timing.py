"""GPU box: wall time per call (host-issue bound, synchronised at the end of each loop) of the pieces of one rollout step at
BASELINE configs[4] size (500 k Gaussians, 100 bones): where the 3.7 ms per frame of gsdyn.dynamics.rollout go."""
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
fps_all = D.farthest_point_sampler(xyz[None], 1000)[0]
pos1000 = xyz[fps_all]
bones, idx = D.downsample_vertices(pos1000, 100, 0.3)
nobj = bones.shape[0]
hist = bones[None].repeat(3, 1, 1)
eef_h = torch.zeros((3, 1, 3), device=dev)
eef_n = torch.tensor([[0.02, 0.0, 0.01]], device=dev)

def t(name, fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print("%-34s %8.1f us per call" % (name, (time.perf_counter() - t0) / n * 1e6))

with torch.no_grad():
    t("downsample_vertices (fps + thinning)", lambda: D.downsample_vertices(pos1000, 100, 0.3))
    states = torch.zeros((1, 3, nobj + 1, 3), device=dev); states[0, :, :nobj] = hist
    mask = torch.ones(nobj + 1, dtype=torch.bool, device=dev); tool = torch.zeros(nobj + 1, dtype=torch.bool, device=dev); tool[nobj] = True
    t("construct_edges", lambda: D.construct_edges(states[0, -1], 0.6, mask, tool, topk=5))
    recv, send = D.construct_edges(states[0, -1], 0.6, mask, tool, topk=5)
    attrs = torch.zeros((1, nobj + 1, 2), device=dev); attrs[0, :nobj, 0] = 1; attrs[0, nobj:, 1] = 1
    action = torch.zeros((1, nobj + 1, 3), device=dev)
    pin = torch.ones((1, nobj, 1), device=dev)
    t("DynamicsPredictor.forward", lambda: model(state=states, attrs=attrs, p_instance=pin, action=action, receivers=recv, senders=send))
    pred, _ = model(state=states, attrs=attrs, p_instance=pin, action=action, receivers=recv, senders=send)
    rel = D.relations_to_matrix(recv, send, nobj + 1)[:nobj, :nobj]
    mot = pred[0] - bones
    t("fit_bone_rotations", lambda: D.fit_bone_rotations(bones, mot, rel))
    R = D.fit_bone_rotations(bones, mot, rel)
    t("mat2quat + normalize", lambda: torch.nn.functional.normalize(D.mat2quat(R), dim=-1))
    t("interpolate_motions (all of it)", lambda: D.interpolate_motions(bones, mot, rel, xyz, quat=quat))
    t("rollout_step (all of it)", lambda: D.rollout_step(model, hist, eef_h, eef_n, xyz, quat, 0.6, 5))
    inl = torch.arange(P, device=dev)
    t("all_pos[inl][fps_all_idx] + cat", lambda: xyz[inl][fps_all])
    from diff_gaussian_rasterization import _hip
    Rf, qf, code = _hip.fit_bones(bones, mot, rel)
    print("codes:", {int(c): int((code == c).sum()) for c in code.unique()}, "rel neighbours per bone min/mean", int(rel.sum(1).min()), float(rel.sum(1).float().mean()))
    t("fit_bones (one launch)", lambda: _hip.fit_bones(bones, mot, rel))
