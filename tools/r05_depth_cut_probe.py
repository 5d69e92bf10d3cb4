"""GPU box: how well do the previous frame's per-tile depth proposals serve the NEXT frame of the configs[4] episode (500 k Gaussians, 1080p,
4 cameras, the bench's rollout)?  Per (margin, dilation): frames with a failing tile, failing tiles per frame, list entries kept.  Decides
the defaults of gsdyn.render.DepthCuts -- and whether frame-level redo can pay on a scene that moves this much."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import synth_scene_params
from gsdyn.dynamics import DynamicsPredictor
from gsdyn.predict import collect_scene_data, ring_poses
from gsdyn.render import DepthCuts, Renderer
dev = torch.device("cuda:0")
P, W, H, CAMS, frames = 500_000, 1920, 1080, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 20
params = {k: v.detach() for k, v in synth_scene_params(P, seed=0, device=dev).items()}
cfg = dict(nf_particle=512, nf_relation=512, nf_effect=512, attr_dim=2, state_dim=0, action_dim=3, pstep=3, rel_attr_dim=2, rel_group_dim=1, rel_distance_dim=3, n_his=3)
torch.manual_seed(0)
model = DynamicsPredictor(cfg, device=dev).eval()
eef = torch.tensor([[0.0, 0.2, 0.0]], device=dev) + torch.tensor([[0.02, 0.0, 0.01]], device=dev) * torch.arange(frames, device=dev, dtype=torch.float32)[:, None]
roll = dict(max_nobj=100, fps_radius=0.3, adj_thresh=0.6, topk=5, connect_all=False, dist_thresh=0.005, n_fps_all=1000, remove_outliers=False)
scene, _, _ = collect_scene_data(model, params, eef, **roll)
mv = [float((scene[t]["means3D"] - scene[t - 1]["means3D"]).norm(dim=-1).mean()) for t in range(1, frames)]
mx = [float((scene[t]["means3D"] - scene[t - 1]["means3D"]).norm(dim=-1).max()) for t in range(1, frames)]
print(f"per-frame displacement of the Gaussians: mean {np.mean(mv):.4f} (max over frames {np.max(mv):.4f}), largest single {np.max(mx):.3f}; camera distance ~4")
rdr = Renderer(dev, w=W, h=H)
cams = [rdr._camera(w2c, k, (0.0, 0.0, 0.0)) for w2c, k in ring_poses(CAMS, W, H)]
T = ((H + 15) // 16) * ((W + 15) // 16)


def entries(states):
    return sum(int((lambda rg: (rg[:, 1] - rg[:, 0]).sum())(_hip.debug_views(st)["ranges"])) for st in states)


def call(d, cuts=None):
    return _hip.rasterize_forward_batch(cams, d["means3D"].contiguous(), d["opacities"].contiguous(), d["colors_precomp"].contiguous(), None,
                                        d["scales"].contiguous(), d["rotations"].contiguous(), None, forward_only=True, depth_cuts=cuts)

full = [entries(call(scene[t])[3]) for t in (0, frames // 2, frames - 1)]
print("entries per frame without cuts (first / middle / last):", full)
for margin in (1.01, 1.05, 1.25):
    for dil in (0, 1, 2, 4):
        dc = DepthCuts(dilate=dil, margin=margin, adapt=False)
        fails, kept, bad_frames = [], [], 0
        for t in range(frames):
            armed = dc.arm("k", CAMS, H, W, dev, t)
            out = call(scene[t], armed)
            dc.sent(armed[2])
            f = armed[2].tolist()
            if t > 0:
                fails.append(sum(f)); bad_frames += any(x > 0 for x in f)
                kept.append(entries(out[3]))
        print(f"margin {margin:4.2f} dilate {dil}: frames with a failing tile {bad_frames:2d} / {frames - 1}; failing tiles per frame (of {CAMS * T}) mean {np.mean(fails):7.1f} "
              f"max {np.max(fails):5d}; entries kept {np.mean(kept) / np.mean(full):.2f} of all")

dc = DepthCuts()            # the defaults: dilate 2, adapting
trace, fails = [], 0
for t in range(frames):
    armed = dc.arm("k", CAMS, H, W, dev, t)
    call(scene[t], armed)
    dc.sent(armed[2])
    trace.append(dc.dilate)
bad = dc.failed()
print(f"adaptive (start at dilate 2): dilation per frame {trace}; frames with failing views {sorted(bad)}, views to repeat {sum(len(v) for v in bad.values())} of {CAMS * (frames - 1)}")
