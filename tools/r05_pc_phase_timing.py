"""GPU box, debug build (TIMING=1, GSR_TIMING_KERNEL=b): phase breakdown of render_bwd (wave 0 view)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
os.environ["GSR_TIMING_KERNEL"] = "b"
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
V = int(os.environ.get("V", "1"))
cams = synth_ring_cameras(max(V, 4), 800, 800, device=dev)[:V]
cam = cams[0]
with torch.no_grad():
    rv0 = params2rendervar(params)
FROZEN = os.environ.get("FROZEN", "0") == "1"   # colors_precomp without gradient: the six-sum backward
rv = {k: v.detach().clone().requires_grad_(not (FROZEN and k == "colors_precomp")) for k, v in rv0.items()}
dL = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (3, 800, 800)).astype(np.float32), device=dev)
dLv = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (V, 3, 800, 800)).astype(np.float32), device=dev)
from diff_gaussian_rasterization import rasterize_gaussians_views
m2 = torch.zeros((V, 100_000, 3), device=dev, requires_grad=True)
lib = _hip.load_library()
buf = (C.c_uint64 * 16)()
def run():
    if V == 1:
        im, _, _ = GaussianRasterizer(raster_settings=cam)(**rv)
        im.backward(gradient=dL)
    else:
        im, _, _ = rasterize_gaussians_views(cams, rv["means3D"], m2, rv["opacities"], colors_precomp=rv["colors_precomp"],
                                             scales=rv["scales"], rotations=rv["rotations"])
        im.backward(gradient=dLv)
for _ in range(3):
    run()
lib.gsr_debug_phase_timing(buf)
N = 10
for _ in range(N):
    run()
lib.gsr_debug_phase_timing(buf)
names = ["consumer0: waiting for a batch", "consumer0: tile head (pixel loads)", "consumer0: visits", "consumer0: combining (when last to arrive)", "stager: waiting for a ring slot", "stager: tile start chain (n_contrib, max)", "stager: chunk loads + gathers", "stager: appending + publishing", "stager: next ticket + zero fill + order entry"]
tot = sum(buf[i] for i in range(9))
print("tiles per launch", buf[15] / N)
for i, n in enumerate(names):
    print(f"{n:50s} {buf[i] / N / 1e3:10.2f} x10us  {100.0 * buf[i] / max(tot, 1):5.1f} %")
try:
    st = (C.c_uint64 * 2048)()
    lib.gsr_debug_pc_starts(st)
    a = np.array(st[:], dtype=np.int64); a = a[a > 0]
    d = (a - a.min()) * 0.01
    print(f"workgroup starts: {len(a)} workgroups, latest start {d.max():.1f} us after the first; started within 5 us: {(d < 5).sum()}, later than 20 us: {(d > 20).sum()}")
    err = C.c_uint32(0); lib.gsr_debug_pc_error(C.byref(err)); print("pc error word", err.value)
except Exception as e:
    print("no start census:", e)
try:
    print("occupancy API: workgroups per CU =", lib.gsr_debug_pc_occupancy())
except Exception as e:
    print("no occupancy export:", e)
