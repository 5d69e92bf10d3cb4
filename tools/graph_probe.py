"""Eager vs hipGraph replay of the direct step (gsdyn.step.render_step_views) for V = 1, 2, 4, 8 views: wall us per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from gsdyn import synth_ring_cameras, synth_scene_params
from gsdyn.step import GraphedRenderStep, render_step_views
dev = torch.device("cuda:0")
P, W, H = 100_000, 800, 800
for V in (1, 2, 4, 8):
    params = synth_scene_params(P, seed=0, device=dev)
    cams = synth_ring_cameras(8, W, H, device=dev)[:V]
    dL = torch.tensor(np.random.default_rng(1).uniform(-1, 1, (V, 3, H, W)).astype(np.float32), device=dev)
    def t(fn, n=40, w=8):
        for _ in range(w): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    eager = t(lambda: render_step_views(params, cams, dL))
    ims_e, g_e = render_step_views(params, cams, dL)
    gs = GraphedRenderStep(params, cams, dL)
    graph = t(gs.replay)
    ims_g, g_g = gs.replay(); torch.cuda.synchronize()
    same = torch.equal(ims_e, ims_g) and all(torch.equal(g_e[k], g_g[k]) for k in ("means3D", "unnorm_rotations", "logit_opacities", "log_scales", "rgb_colors"))
    print(f"V={V}: eager {eager:.1f} us, graph replay {graph:.1f} us, counts ok {gs.ok()}, bit-identical {same}", flush=True)
