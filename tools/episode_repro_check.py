import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd")); sys.path.insert(0, ROOT)
from gsdyn import synth_scene_params
from gsdyn.dynamics import DynamicsPredictor
from gsdyn.predict import predict_episode, ring_poses
dev = torch.device("cuda:0")
gold = np.load(os.path.join(ROOT, "tests/golden/dynamics_host.npz"))
cfg = {str(k): int(v) for k, v in zip(gold["gnn_cfg_keys"], gold["gnn_cfg_vals"])}
model = DynamicsPredictor(cfg, device=dev).eval()
model.load_state_dict({k[len("gnn_w_"):]: torch.tensor(gold[k]) for k in gold.files if k.startswith("gnn_w_")})
P, W, H, CAMS, S = 30000, 480, 272, 4, 5
params = {k: v.detach() for k, v in synth_scene_params(P, device=dev, scale_lo=0.01, scale_hi=0.04).items()}
eef = torch.tensor([[0.0, 0.0, 0.0]], device=dev) + torch.tensor([[0.04, 0.0, 0.02]], device=dev) * torch.tensor([0.0, 1.0, 1.01, 2.0, 3.0], device=dev)[:, None]
roll = dict(max_nobj=100, fps_radius=0.3, adj_thresh=0.6, topk=5, connect_all=False, dist_thresh=0.005, n_fps_all=1000)
poses = ring_poses(CAMS, W, H)
sort = os.environ.get("SORT", "1") == "1"
import gsdyn.predict as gp
if not sort:
    orig = gp.collect_scene_data
    gp.collect_scene_data = lambda *a, **k: orig(*a, spatial_sort=False, **k)
ref_scene = []
ref, _, _ = predict_episode(model, params, eef, poses, W, H, rollout_cfg=roll, rank=0, world=1, scene_out=ref_scene)
for it in range(12):
    sc = []
    got, _, _ = predict_episode(model, params, eef, poses, W, H, rollout_cfg=roll, rank=0, world=1, scene_out=sc)
    worst = 0; info = None
    for k in ref:
        d = (got[k][0] - ref[k][0]).abs()
        if float(d.max()) > worst:
            worst = float(d.max()); info = (k, float(d.mean()), int((d > 1e-3).sum()))
    dpos = max(float((a["means3D"] - b["means3D"]).abs().max()) for a, b in zip(sc, ref_scene))
    print(it, "worst image diff %.5f" % worst, info, "max position diff %.3g" % dpos)
    if worst > 0.05 and os.environ.get("ORACLE", "1") == "1":   # (HIP == oracle to 3e-7 on both runs when tried: the inputs, not the rasterizer)
        # is it the inputs (a discrete decision of the algorithm flipping under 1e-6 position noise) or the rasterizer?  The CPU oracle
        # on each run's own inputs decides.  (Test infrastructure: oracle/ through tests/oracle_double.py.)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_double
        from gsdyn.render import Renderer
        f, c = info[0]
        cam = Renderer(dev, w=W, h=H)._camera(poses[c][0], poses[c][1], (0.0, 0.0, 0.0))
        for name, scene, imgs in (("ref", ref_scene, ref), ("this", sc, got)):
            d = {k: v.detach().cpu() for k, v in scene[f].items()}
            o = oracle_double.rasterize_forward(cam, d["means3D"], d["opacities"], d["colors_precomp"], None, d["scales"], d["rotations"], None)
            print("   ", name, "HIP vs oracle on its own inputs: max %.3g" % float((imgs[(f, c)][0].cpu() - o[0].cpu()).abs().max()))
        break
