"""GPU box: tile-list lengths of the frames of the configs[4] episode (bench.py --config 5 --with-rollout) -- how many lists the per-tile sort's
long-ticket path sees (> 2032 entries) and how long they are."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip
from gsdyn import synth_scene_params
from gsdyn.dynamics import DynamicsPredictor
from gsdyn.predict import FrameShard, collect_scene_data, ring_poses
dev = torch.device("cuda:0")
P, W, H, CAMS, frames = 500_000, 1920, 1080, 4, 30
params = {k: v.detach() for k, v in synth_scene_params(P, device=dev).items()}
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3, rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = DynamicsPredictor(cfg, device=dev).eval()
eef = torch.tensor([[0.0, 0.2, 0.0]], device=dev) + torch.tensor([[0.02, 0.0, 0.01]], device=dev) * torch.arange(frames, device=dev, dtype=torch.float32)[:, None]
roll = dict(max_nobj=100, fps_radius=0.3, adj_thresh=0.6, topk=5, connect_all=False, dist_thresh=0.005, n_fps_all=1000, remove_outliers=False)
with torch.no_grad():
    scene, vis, tm = collect_scene_data(model, params, eef, **roll)
shard = FrameShard(dev, W, H, ring_poses(CAMS, W, H), 0, 1)
seen = {}
orig = _hip.rasterize_forward_batch
def spy(*a, **k):
    out = orig(*a, **k)
    seen["s"] = out[3]
    return out
_hip.rasterize_forward_batch = spy
al = lambda x: (x + 255) // 256 * 256
N, T = H * W, ((H + 15) // 16) * ((W + 15) // 16)
for f in (0, 5, 11, 17, 23, frames - 1):
    shard.render_frame(f, scene[f])
    torch.cuda.synchronize()
    lens = []
    for st in seen["s"]:
        r = st.image[2 * al(4 * N):][:8 * T].view(torch.int32).reshape(T, 2).cpu().numpy().astype(np.int64)
        lens.append(np.clip(r[:, 1] - r[:, 0], 0, None))
    L = np.concatenate(lens)
    xyz = scene[f]["means3D"]
    print(f"frame {f}: entries {L.sum()}  tiles {L.size}  mean {L[L > 0].mean():.0f}  max {L.max()}  > 512: {(L > 512).sum()}  > 1024: {(L > 1024).sum()}  > 2032: {(L > 2032).sum()}  "
          f"> 4096: {(L > 4096).sum()}  > 8192: {(L > 8192).sum()};  cloud extent {(xyz.max(0).values - xyz.min(0).values).tolist()}")
