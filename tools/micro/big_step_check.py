"""GPU box: the fused direct step against the autograd step at a large size (500k Gaussians, 1920x1080, 2 cameras, t > 0)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import LossWeights, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
from gsdyn.dp import init_variables
from gsdyn.step import make_rigidity_variables
dev = torch.device("cuda:0")
P, W, H = int(os.environ.get("BIG_P", "500000")), 1920, 1080
torch.manual_seed(0)
params = synth_scene_params(P, device=dev)
cams = synth_ring_cameras(4, W, H, device=dev)
im_gt, seg_gt = synth_targets(W, H, device=dev)
w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)
views = [dict(cam=cams[i], im=im_gt, seg=seg_gt, id=i) for i in (0, 2)]
t0 = time.time(); rig = make_rigidity_variables(params, num_knn=20); torch.cuda.synchronize(); print("knn s", time.time() - t0)
with torch.no_grad():
    params["means3D"].add_(0.002 * torch.randn_like(params["means3D"]))
def fresh():
    v = init_variables(P, dev); v.update(rig); return v
for p_ in params.values(): p_.grad = None
la, _, aux = get_loss_views(params, views, fresh(), False, w, frozen_colours=True); la.backward()
ga = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
for rep in range(3):     # 1st: synchronous (no capacity yet for this shape in the direct path? it shares the cache), then capacity mode
    for p_ in params.values(): p_.grad = None
    lb, _, auxb = loss_and_grads_views(params, views, fresh(), False, w)
    gb = {k: v.grad.clone() for k, v in params.items() if v.grad is not None}
    errs = {k: ((ga[k] - gb[k]).abs().max() / ga[k].abs().max()).item() for k in ga}
    print(rep, "loss rel", abs(float(la.detach()) - float(lb)) / abs(float(lb)), {k: f"{v:.1e}" for k, v in errs.items()})
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    for p_ in params.values(): p_.grad = None
    loss_and_grads_views(params, views, fresh(), False, w)
torch.cuda.synchronize(); print("ms/step (2 cameras)", (time.perf_counter() - t0) / 5 * 1e3)
