"""GPU box: host cost of one GaussianRasterizer call -- the autograd node in C++ (_C.rasterize) against the Python torch.autograd.Function,
same process, alternating.  Small scene (the GPU is never the limit): wall time per forward + backward as the reference issues them."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-dynamics_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizer
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
if os.environ.get("REUSE") == "0":
    dgr.layer_state(dev).list_reuse = False
if os.environ.get("CAPACITY") == "0":
    dgr.layer_state(dev).capacity_mode = False
for P, S in ((2000, 128), (100_000, 800)):
    params = synth_scene_params(P, device=dev)
    cam = synth_ring_cameras(4, S, S, device=dev)[0]
    dL = torch.rand((3, S, S), device=dev)
    with torch.no_grad():
        rv0 = {k: v.detach() for k, v in params2rendervar(params).items()}

    def fwd_bwd():
        rv = {k: v.requires_grad_(True) for k, v in rv0.items()}
        im, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
        im.backward(gradient=dL)

    def fwd_only():
        with torch.no_grad():
            GaussianRasterizer(raster_settings=cam)(**rv0)

    for name, fn in (("forward + backward", fwd_bwd), ("forward under no_grad", fwd_only)):
        res = {}
        for rnd in range(3):
            for node in ("python", "c++"):
                dgr._PY_NODE = node == "python"
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                N = 300
                t0 = time.perf_counter()
                for _ in range(N):
                    fn()
                t_issue = time.perf_counter() - t0
                torch.cuda.synchronize()
                t_all = time.perf_counter() - t0
                res.setdefault(node, []).append((1e6 * t_issue / N, 1e6 * t_all / N))
        for node, v in res.items():
            print(f"P={P} {S}x{S} {name:24s} node={node:7s} host issue us/call: " + " ".join(f"{a:7.1f}" for a, _ in v) + "   wall us/call: " + " ".join(f"{b:7.1f}" for _, b in v))
