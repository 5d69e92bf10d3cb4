"""GPU box, GSR_HIP_LIB = a -DGSR_TILE_TIMING build (tools/build_full_variant.sh NAME "-DGSR_TILE_TIMING ..."): phase breakdown of render_fwd at 8 views."""
import ctypes as C, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-dynamics_amd")):
    sys.path.insert(0, p)
from diff_gaussian_rasterization import _hip
from gsdyn import synth_ring_cameras, synth_scene_params
from gsdyn.step import render_step_views
dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = synth_scene_params(100_000, device=dev)
cams = synth_ring_cameras(V, 800, 800, device=dev)
dL = torch.rand((V, 3, 800, 800), device=dev) * 2 - 1
lib = _hip.load_library()
buf = (C.c_uint64 * 16)()
for _ in range(3):
    render_step_views(params, cams, dL)
torch.cuda.synchronize(); lib.gsr_debug_phase_timing(buf)
N = 10
for _ in range(N):
    render_step_views(params, cams, dL)
torch.cuda.synchronize(); rc = lib.gsr_debug_phase_timing(buf)
names = ["setup+first gather", "wait __syncthreads_count", "classify + issue prefetch", "barrier after counts",
         "compaction writes + barrier", "blend loop", "output stores", "ticket (atomic + 2 barriers)"]
tot = sum(buf[i] for i in range(8))
print("rc", rc, "tiles processed per launch", buf[15] / N)
for i, n in enumerate(names):
    print(f"{n:32s} {buf[i] / N / 1e6:10.2f} Mcycles/launch  {100.0 * buf[i] / max(tot, 1):5.1f} %")
