"""GPU box: a scene well past the benchmark's sizes -- 3 M Gaussians, four 1920x1080 views (and 8 M, one view) -- through the multi-view step:
no overflow of 32-bit offsets or capacities, finite outputs, the batched forward equal to the single-view forward bit for bit, repeatable."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
for P, V, W, H in ((3_000_000, 4, 1920, 1080), (8_000_000, 1, 1920, 1080))[:1 if os.environ.get("BIG_ONLY_FIRST") else 2]:
    params = synth_scene_params(P, seed=0, device=dev, scale_lo=0.002, scale_hi=0.012)
    cams = synth_ring_cameras(max(V, 4), W, H, device=dev)[:V]
    dL = torch.rand((V, 3, H, W), device=dev) - 0.5
    ims, g = render_step_views(params, cams, dL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ims2, g2 = render_step_views(params, cams, dL)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok_rep = torch.equal(ims, ims2) and all(torch.equal(g[k], g2[k]) for k in g if isinstance(g[k], torch.Tensor))
    fin = bool(torch.isfinite(ims).all()) and all(bool(torch.isfinite(v).all()) for k, v in g.items() if isinstance(v, torch.Tensor) and v.is_floating_point())
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
        im0, rad0, _ = GaussianRasterizer(raster_settings=cams[0])(**rv)
        imb = _hip.rasterize_forward_batch(list(cams), rv["means3D"], rv["opacities"], rv["colors_precomp"], None, rv["scales"], rv["rotations"], None)[0]
    same = torch.equal(im0, imb[0])          # same activated inputs (the step above fuses the activations: other roundings)
    close = float((im0 - ims[0]).abs().max())
    seen = int((g["radii"] > 0).sum())
    print(f"P={P} V={V} {W}x{H}: step {1e3 * dt:.1f} ms, visible Gaussian-views {seen}, finite {fin}, repeatable {ok_rep}, batched == single-view forward {same} (step's fused activations: max |diff| {close:.2e}), "
          f"|grad means3D| max {float(g['means3D'].abs().max()):.3e}, mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del params, ims, g, ims2, g2
    torch.cuda.empty_cache()
