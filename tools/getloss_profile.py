"""GPU box: run a few get_loss-shaped steps (for rocprofv3 --stats)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import LossWeights, get_loss, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
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
BATCHED = os.environ.get("GETLOSS_SEPARATE") != "1"
DIRECT = os.environ.get("GETLOSS_AUTOGRAD") != "1"       # library calls back to back (gsdyn.step.loss_and_grads_views)
FROZEN = os.environ.get("GETLOSS_COLOUR_GRADS") != "1"     # the tracking schedule: both colour groups have lr 0
def step(initial):
    for p in params.values():
        p.grad = None
    if BATCHED and DIRECT:
        loss_and_grads_views(params, views, variables, initial, w)
        return
    if BATCHED:
        loss, _, _ = get_loss_views(params, views, variables, initial, w, frozen_colours=FROZEN)
        loss.backward()
        return
    for d in views:
        loss, _ = get_loss(params, d, variables, initial, w)
        loss.backward()
MODES = {'t0': (True,), 't1': (False,)}.get(os.environ.get('GETLOSS_MODE', ''), (True, False))
for initial in MODES:
    for _ in range(2):
        step(initial)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        step(initial)
    torch.cuda.synchronize()
    print("is_initial_timestep", initial, "ms/view", (time.perf_counter() - t0) * 1e3 / 3 / 4)
