"""GPU box: cProfile of the host side of the reference-pattern step (one camera per iteration, colour + seg as a 2-view call)."""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from gsdyn import LossWeights, get_loss_views, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
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
initial = os.environ.get("GETLOSS_MODE", "t1") == "t0"
def step(i):
    for p in params.values():
        p.grad = None
    if os.environ.get("DIRECT", "1") == "1":
        loss_and_grads_views(params, [views[i % 4]], variables, initial, w)
        return
    loss, _, _ = get_loss_views(params, [views[i % 4]], variables, initial, w, frozen_colours=True)
    loss.backward()
for i in range(20):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    step(i)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 10)
import gc
gc.collect(); gc.freeze()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    step(i)
torch.cuda.synchronize()
print("ms/step after gc.freeze()", (time.perf_counter() - t0) * 10)
gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100):
    step(i)
torch.cuda.synchronize()
print("ms/step with gc disabled", (time.perf_counter() - t0) * 10)
gc.enable()
if os.environ.get("HOST_PROFILE_ONLY_TIMING"):
    sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    step(i)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
