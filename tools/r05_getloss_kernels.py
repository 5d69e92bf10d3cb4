"""GPU box: per-kernel times of the fused get_loss step (gsdyn.step.loss_and_grads_views: 4 cameras x (colour + seg) in one call, fused
losses, backward), HIP events around every library launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import LossWeights, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
from gsdyn.dp import init_variables
from gsdyn.step import make_rigidity_variables
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
params = synth_scene_params(P, device=dev)
cams = synth_ring_cameras(4, W, H, device=dev)
im_gt, seg_gt = synth_targets(W, H, device=dev)
variables = init_variables(P, dev)
variables.update(make_rigidity_variables(params, num_knn=20))
w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)
views = [dict(cam=c, im=im_gt, seg=seg_gt, id=i) for i, c in enumerate(cams)]
for initial in (True, False):
    for _ in range(5):
        loss_and_grads_views(params, views, variables, initial, w)
    torch.cuda.synchronize()
    _hip.profile_begin()
    N = 10
    for _ in range(N):
        loss_and_grads_views(params, views, variables, initial, w)
    torch.cuda.synchronize()
    prof = _hip.profile_end()
    tot = sum(ms for ms, n in prof.values()) / N
    print(f"t {'= 0' if initial else '> 0'}: library kernels busy {1e3 * tot:.1f} us per 4-camera step: "
          + " ".join(f"{k}={1e3 * ms / N:.1f}" for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0])))
