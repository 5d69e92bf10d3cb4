"""GPU box, debug build (make -C gs-dynamics_amd/csrc clean all TIMING=1): phase breakdown of render_fwd."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import GaussianRasterizer, _hip
from gsdyn import params2rendervar, synth_ring_cameras, synth_scene_params
dev = torch.device("cuda:0")
params = synth_scene_params(100_000, device=dev)
cam = synth_ring_cameras(4, 800, 800, device=dev)[0]
with torch.no_grad():
    rv = params2rendervar(params)
    lib = _hip.load_library()
    buf = (C.c_uint64 * 16)()
    for _ in range(3):
        GaussianRasterizer(raster_settings=cam)(**rv)
    lib.gsr_debug_phase_timing(buf)
    N = 10
    for _ in range(N):
        GaussianRasterizer(raster_settings=cam)(**rv)
    rc = lib.gsr_debug_phase_timing(buf)
names = ["setup+first gather", "wait __syncthreads_count", "classify + issue prefetch", "barrier after counts",
         "compaction writes + barrier", "blend loop", "output stores", "ticket (atomic + 2 barriers)"]
tot = sum(buf[i] for i in range(8))
print("rc", rc, "tiles processed per launch", buf[15] / N)
for i, n in enumerate(names):
    print(f"{n:32s} {buf[i] / N / 1e6:10.2f} Mcycles/launch  {100.0 * buf[i] / max(tot, 1):5.1f} %")
print("sum over 1024 workgroups (wave 0 view):", tot / N / 1e6, "Mcycles/launch ->", tot / N / 1024 / 2.4e3, "us per WG at 2.4 GHz")
