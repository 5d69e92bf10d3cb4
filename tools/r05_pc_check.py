"""GPU box: gradients of one multi-view fwd+bwd, saved to OUT (npz) -- run once per build / environment setting and compare the files
bit for bit (tools/r05_pc_check.sh): the producer / consumer backward must write the records the barrier form writes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import rasterize_gaussians_views
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
out, V, P, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
FROZEN = os.environ.get("FROZEN", "0") == "1"
dev = torch.device("cuda:0")
params = synth_scene_params(P, device=dev)
cams = synth_ring_cameras(max(V, 4), S, S, device=dev)[:V]
with torch.no_grad():
    rv0 = params2rendervar(params)
rv = {k: v.detach().clone().requires_grad_(not (FROZEN and k == "colors_precomp")) for k, v in rv0.items()}
dLv = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (V, 3, S, S)).astype(np.float32), device=dev)
m2 = torch.zeros((V, P, 3), device=dev, requires_grad=True)
res = {}
for rep in range(2):
    for t in list(rv.values()) + [m2]:
        t.grad = None
    im, _, _ = rasterize_gaussians_views(cams, rv["means3D"], m2, rv["opacities"], colors_precomp=rv["colors_precomp"],
                                         scales=rv["scales"], rotations=rv["rotations"])
    im.backward(gradient=dLv)
    torch.cuda.synchronize()
    cur = {k: v.grad.detach().cpu().numpy() for k, v in rv.items() if v.grad is not None}
    cur["means2D_views"] = m2.grad.detach().cpu().numpy()
    if rep == 0:
        res = cur
    else:
        for k in cur:
            assert np.array_equal(cur[k], res[k]), f"rerun differs in {k}"
np.savez(out, **res)
print("saved", out, {k: float(np.abs(v).max()) for k, v in res.items()})
