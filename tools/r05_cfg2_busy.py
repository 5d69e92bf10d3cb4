"""GPU box: the forward-only drop-in call of BASELINE configs[1] (50 k Gaussians, one 800x800 view) -- wall time per call against the
library kernels' busy time per call (HIP events around every launch), tile-list reuse off, capacity mode on / off."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
for P in (50_000, 100_000):
    params = synth_scene_params(P, seed=0, device=dev)
    cam = synth_ring_cameras(4, 800, 800, device=dev)[0]
    with torch.no_grad():
        rv = {k: v.detach() for k, v in params2rendervar(params).items()}
    dgr.layer_state(dev).list_reuse = False
    for cap in (True, False, True):
        dgr.layer_state(dev).capacity_mode = cap
        def fwd():
            with torch.no_grad():
                GaussianRasterizer(raster_settings=cam)(**rv)
        for _ in range(20):
            fwd()
        torch.cuda.synchronize()
        N = 300
        t0 = time.perf_counter()
        for _ in range(N):
            fwd()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        _hip.profile_begin()
        for _ in range(20):
            fwd()
        torch.cuda.synchronize()
        prof = _hip.profile_end()
        busy = sum(ms for ms, n in prof.values()) / 20
        print(f"P={P} capacity={cap}: wall {1e6 * t_all / N:.1f} us per call, host issue {1e6 * t_issue / N:.1f}, kernels busy {1e3 * busy:.1f} us: "
              + " ".join(f"{k}={1e3 * ms / 20:.1f}" for k, (ms, n) in sorted(prof.items())), flush=True)
