"""GPU box: cProfile of the host side of the unchanged-caller loop (params2rendervar + GaussianRasterizer + backward), by own time.
Measured: ~375 us of host time per render, of which the layer's forward call (launches + its one synchronisation) is ~75 us and the
autograd backward (the layer's backward plus the caller's torch ops) ~175 us: the loop is bound by torch-op overhead, not by the library."""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
cam = synth_ring_cameras(4, 800, 800, device=dev)[0]
dL = torch.rand((3, 800, 800), device=dev) * 2 - 1
def render():
    for p in params.values():
        p.grad = None
    rv = params2rendervar(params)
    rv["means2D"].retain_grad()
    im, radius, depth = GaussianRasterizer(raster_settings=cam)(**rv)
    im.backward(gradient=dL)
for _ in range(20): render()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): render()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
