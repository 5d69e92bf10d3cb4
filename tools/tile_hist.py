import os, sys, numpy as np, torch
sys.path.insert(0, "gs-dynamics_amd")
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev); cams = synth_ring_cameras(4, 800, 800, device=dev)
st = {}
orig = _hip.rasterize_forward
def spy(*a, **k):
    out = orig(*a, **k); st["s"] = out[3]; return out
_hip.rasterize_forward = spy
with torch.no_grad():
    rv = params2rendervar(params)
    GaussianRasterizer(raster_settings=cams[0])(**rv)
v = _hip.debug_views(st["s"])
r = v["ranges"].cpu().numpy().astype(np.int64); n = r[:, 1] - r[:, 0]
print("tiles", len(n), "busy", (n > 0).sum(), "D", n.sum(), "max", n.max())
for lo, hi in [(1, 64), (65, 128), (129, 256), (257, 512), (513, 1024), (1025, 2048), (2049, 4096), (4097, 10**9)]:
    m = (n >= lo) & (n <= hi)
    print(f"  {lo:5d}..{hi:<10d} tiles {m.sum():5d}  entries {n[m].sum():8d}")
