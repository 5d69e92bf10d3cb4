import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from diff_gaussian_rasterization import _hip
from gsdyn import params2rendervar, synth_scene_params
from gsdyn.predict import ring_poses
from gsdyn.render import Renderer
dev = torch.device("cuda:0")
params = synth_scene_params(500_000, device=dev)
with torch.no_grad():
    d = {k: v.detach() for k, v in params2rendervar(params).items()}
r = Renderer(dev, w=1920, h=1080)
poses = ring_poses(4, 1920, 1080)
cams = [r._camera(w2c, k, (0, 0, 0)) for w2c, k in poses]
out, _, depth, states = _hip.rasterize_forward_batch(cams, d["means3D"], d["opacities"], d["colors_precomp"], None, d["scales"], d["rotations"], None, prepare_backward=False, forward_only=True)
for st in states[:2]:
    v = _hip.debug_views(st)
    rg = v["ranges"].cpu().numpy().astype(np.int64)
    n = np.maximum(rg[:, 1] - rg[:, 0], 0)
    tot = n.sum()
    for lo, hi in ((0, 1), (1, 512), (512, 1024), (1024, 2048), (2048, 4096), (4096, 10**9)):
        m = (n >= lo) & (n < hi)
        print(f"lists [{lo},{hi}): tiles {m.sum():5d}  entries {n[m].sum():9d} = {100.0*n[m].sum()/tot:5.1f} %")
    print("max", n.max(), "tiles", len(n), "entries", tot)
