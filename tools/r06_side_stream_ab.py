"""GPU box: the get_loss-shaped direct step (gsdyn.step.loss_and_grads_views, t > 0) with the view-independent loss terms on the main stream
(GSDYN_SHARED_SIDE_STREAM=0) or beside the render on a second stream (=1; the switch is read at import: one process per arm).
Prints ms per step for 4 cameras in one call and for the one-camera train iteration (+ FusedAdam)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from gsdyn import LossWeights, initialize_optimizer, loss_and_grads_views, synth_ring_cameras, synth_scene_params, synth_targets
from gsdyn.dp import init_variables
from gsdyn.step import make_rigidity_variables
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
w = LossWeights(im=50.0, seg=200.0, rigid=200.0, iso=1000.0, rot=4.0, bg=200.0)
im_gt, seg_gt = synth_targets(W, H, device=dev)
cams = synth_ring_cameras(4, W, H, device=dev)


def med(fn, iters=20, warm=5, reps=5):
    for _ in range(warm):
        fn()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / iters * 1e3)
    return sorted(out)[len(out) // 2]


res = {}
for name, ncam, adam in (("4 cameras, one call", 4, False), ("one camera + Adam", 1, True)):
    params = synth_scene_params(P, seed=0, device=dev)
    v = init_variables(P, dev); v.update(make_rigidity_variables(params, num_knn=20))
    views = [dict(cam=c, im=im_gt, seg=seg_gt, id=i) for i, c in enumerate(cams)]
    opt = initialize_optimizer(params, 4.0) if adam else None
    it = [0]

    def step():
        for p_ in params.values():
            p_.grad = None
        ds = views if ncam == 4 else [views[it[0] % 4]]
        it[0] += 1
        loss_and_grads_views(params, ds, v, False, w)
        if opt is not None:
            opt.step()
    res[name] = med(step)
    g = params["means3D"].grad.clone() if params["means3D"].grad is not None else None
print("GSDYN_SHARED_SIDE_STREAM=%s  " % os.environ.get("GSDYN_SHARED_SIDE_STREAM", "1") + "  ".join(f"{k}: {v_:.4f} ms" for k, v_ in res.items()))
