"""GPU box: the one-view step (the per-GPU share of configs[3] at 8 GPUs; the reference's own loop takes ONE camera per iteration,
/root/reference/src/tracking/train_utils.py:82-86) eager against the same step replayed from a hipGraph (gsdyn.step.GraphedRenderStep), with and
without the Adam step behind it.  VERDICT r05 item 4b.  Median of HIP-event times over 40 steps, 3 rounds."""
import os, statistics, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
from gsdyn import initialize_optimizer, synth_ring_cameras, synth_scene_params
from gsdyn.step import GraphedRenderStep, render_step_views
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
KEYS = ("means3D", "rgb_colors", "unnorm_rotations", "logit_opacities", "log_scales")


def timed(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    import time
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    return statistics.median(a.elapsed_time(b) for a, b in ev) * 1e3, wall


for V in (1, 2):
    cams = synth_ring_cameras(8, W, H, device=dev)[:V]
    dL = torch.tensor(np.random.default_rng(1234).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    for with_adam in (False, True):
        params = synth_scene_params(P, seed=0, device=dev)
        opt = initialize_optimizer(params, 4.0) if with_adam else None

        def eager():
            _, g = render_step_views(params, cams, dL)
            if opt is not None:
                for k in KEYS:
                    params[k].grad = g.get(k)
                opt.step()
        gs = GraphedRenderStep(params, cams, dL)

        def graphed():
            _, g = gs.replay()
            if opt is not None:
                for k in KEYS:
                    params[k].grad = g.get(k)
                opt.step()
        for r in range(3):
            e, ew = timed(eager)
            g_, gw = timed(graphed)
            print(f"V={V} adam={int(with_adam)} round {r}: eager {e:.1f} us (wall {ew:.1f})   graphed {g_:.1f} us (wall {gw:.1f})   fits={gs.ok()}")
